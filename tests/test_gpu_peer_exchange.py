"""The sharded frame with the PEER exchange (mdh_peer_export / mdh_peer_init): several PROCESSES on the one GPU of the test
box, each a rank with its own renderer, exchanging their radiance slices device to device (hipIpcMemHandle, interprocess
frame numbers polled on the device, no collective library and no host copy).  RCCL refuses two ranks of a communicator on one device; this backend is
the device-resident exchange that more than one process can run here.  The ranks' framebuffers must add up to the whole
frame and every rank's atlases must be the whole frame's, bit for bit, with frames in flight."""
import multiprocessing as mp
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _rank(rank, world, probes_name, overlap, frames, sync, conn):
    """a fresh process: one rank of the exchange (handles travel through the parent's pipes)"""
    try:
        sys.path.insert(0, ROOT)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        if sync:
            os.environ["MADARCH_HIP_PEER_SYNC"] = sync
        import helpers
        from madarch_amd import _binding as B
        hip = B.hip_binding()
        R = helpers.make("global_illumination", 96, 64, hip, probes=getattr(helpers, probes_name))
        R.Set_Option(B.OPT_FRAME_OVERLAP, overlap)
        conn.send(R.Peer_Export())
        blobs = conn.recv()
        R.Peer_Init(blobs, rank, world)
        assert (R.Get_Option(B.OPT_RANK), R.Get_Option(B.OPT_WORLD)) == (rank, world)
        R.Set_Option(B.OPT_TIMING, 1)
        for _ in range(frames):
            R.Render()
        R.Finish()
        ms, n = R.Pass_Time(B.PASS_EXCHANGE)
        out = {"image": R.Read_Framebuffer(), "radiance": R.Read_Texture(B.TEX_RADIANCE), "irradiance": R.Read_Texture(B.TEX_IRRADIANCE),
               "exchange_ms": ms / max(n, 1), "exchanges": n}
        conn.send(out)
        conn.recv()  # every rank has read its results: leave together
        R.Comm_Destroy()
        R.Render(); R.Finish()  # a whole frame again, alone
        R.Destroy()
        conn.send("left")
    except Exception as e:  # noqa: BLE001
        import traceback
        conn.send("ERROR " + repr(e) + "\n" + traceback.format_exc())


@pytest.mark.parametrize("world,probes_name,overlap,sync", [(2, "SMALL_PROBES", 2, None), (2, "ODD_PROBES", 2, None), (3, "SMALL_PROBES", 0, None),
                                                            (4, "ODD_PROBES", 2, None)])
def test_peer_exchange_between_processes_on_one_gpu(hip, world, probes_name, overlap, sync):
    import helpers
    frames = 70  # (more than any ring of signals a runtime might keep per event: round 4's first form failed after some tens)
    want = helpers.snapshot(helpers.make("global_illumination", 96, 64, hip, probes=getattr(helpers, probes_name)), frames)
    ctx = mp.get_context("spawn")  # fresh processes: nothing of this process's GPU state is inherited
    pipes, procs = [], []
    for rank in range(world):
        a, b = ctx.Pipe()
        p = ctx.Process(target=_rank, args=(rank, world, probes_name, overlap, frames, sync, b), daemon=True)
        p.start()
        pipes.append(a); procs.append(p)

    def get(c, what):
        assert c.poll(240), "a rank did not send its " + what
        v = c.recv()
        assert not (isinstance(v, str) and v.startswith("ERROR")), v
        return v
    try:
        blobs = [get(c, "handles") for c in pipes]
        assert all(len(b) == 512 for b in blobs)
        for c in pipes:
            c.send(b"".join(blobs))
        outs = [get(c, "results") for c in pipes]
        for c in pipes:
            c.send("go")
        for c in pipes:
            assert get(c, "goodbye") == "left"
    finally:
        for p in procs:
            p.join(30)
            if p.is_alive():
                p.kill()
    image = np.sum([o["image"] for o in outs], axis=0)
    assert helpers.same_bits(image, want["image"]), "the ranks' tiles do not add up to the whole frame"
    for rank, o in enumerate(outs):
        assert helpers.same_bits(o["radiance"], want["radiance"]), "radiance atlas of rank %d" % rank
        assert helpers.same_bits(o["irradiance"], want["irradiance"]), "irradiance atlas of rank %d" % rank
        assert o["exchanges"] == frames and o["exchange_ms"] >= 0.0
    print("peer exchange, %d ranks on one GPU: %s ms per frame" % (world, ", ".join("%.4f" % o["exchange_ms"] for o in outs)))


def test_cpp_example_renders_with_the_peer_exchange(tmp_path):
    """examples/global_illumination.cpp --rank R --world 2 --peer-dir D --device 0: two C++ processes on the C ABI alone (no Python, no
    torch, no RCCL) render the example as a sharded frame on the ONE GPU; their tiles add up to the frame the plain program renders."""
    import subprocess
    exe = os.path.join(ROOT, "examples", "bin", "global_illumination")
    assert os.path.exists(exe), "build first (python -c 'import __graft_entry__ as g; g.build()')"
    W, H, frames = 96, 64, 5
    plain = str(tmp_path / "plain.f32")
    r = subprocess.run([exe, str(W), str(H), str(frames), plain], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    d = tmp_path / "peers"
    d.mkdir()
    out = str(tmp_path / "node.f32")
    procs = [subprocess.Popen([exe, str(W), str(H), str(frames), out, "--rank", str(q), "--world", "2", "--peer-dir", str(d), "--device", "0"],
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for q in range(2)]
    for p in procs:
        so, se = p.communicate(timeout=300)
        assert p.returncode == 0, so + se
    want = np.fromfile(plain, dtype=np.float32)
    got = sum(np.fromfile(out + ".rank%d" % q, dtype=np.float32) for q in range(2))
    assert np.array_equal(got, want, equal_nan=True)


def test_peer_exchange_refusals(hip):
    """what mdh_peer_* must refuse: joining before exporting, blobs that are none, a blob of this very process at another rank's place
    (a process cannot open its own interprocess handles), an atlas format change while peers hold the handles, the collectives of
    a communicator; and that leaving returns the renderer to rank 0 of 1."""
    import helpers
    from madarch_amd import _binding as B
    R = helpers.make("global_illumination", 32, 24, hip, probes=helpers.SMALL_PROBES)
    with pytest.raises(B.MadarchError) as e:
        R.Peer_Init(bytes(2 * B.PEER_BLOB_BYTES), 0, 2)
    assert e.value.status == B.MDH_E_STATE  # no export yet
    blob = R.Peer_Export()
    assert len(blob) == B.PEER_BLOB_BYTES and any(blob)
    with pytest.raises(B.MadarchError) as e:
        R.Peer_Init(blob + bytes(B.PEER_BLOB_BYTES), 0, 2)
    assert e.value.status == B.MDH_E_INVALID  # rank 1's blob is no blob
    R2 = helpers.make("global_illumination", 32, 24, hip, probes=helpers.SMALL_PROBES)
    blob2 = R2.Peer_Export()
    with pytest.raises(B.MadarchError) as e:
        R.Peer_Init(blob + blob2, 0, 2)
    assert e.value.status == B.MDH_E_INVALID and "one process" in str(e.value)
    with pytest.raises(B.MadarchError) as e:
        R.Peer_Init(blob2 + blob, 0, 2)
    assert e.value.status == B.MDH_E_INVALID  # the blob at this rank's place is another renderer's
    with pytest.raises(B.MadarchError) as e:
        R.Set_Option(B.OPT_ATLAS_FORMAT, 1)
    assert e.value.status == B.MDH_E_STATE  # the handles name the atlases as they are
    # a world of one: every rank's blob is its own, nothing to open -- Render is a whole frame, the exchange a no-op that is timed
    want = helpers.snapshot(helpers.make("global_illumination", 32, 24, hip, probes=helpers.SMALL_PROBES), 2)
    R.Peer_Init(R.Peer_Export(), 0, 1)
    assert (R.Get_Option(B.OPT_RANK), R.Get_Option(B.OPT_WORLD)) == (0, 1)
    for call in (R.Comm_Barrier, lambda: R.Comm_Max(1.0), R.Comm_Reduce_Framebuffer):
        with pytest.raises(B.MadarchError) as e:
            call()
        assert e.value.status == B.MDH_E_STATE
    with pytest.raises(B.MadarchError):
        R.Set_Option(B.OPT_IRRADIANCE_ALL, 0)
    got = helpers.snapshot(R, 2)
    for k in want:
        assert helpers.same_bits(got[k], want[k]), k
    R.Comm_Destroy()
    R.Set_Option(B.OPT_WORLD, 1)  # the caller's again
    R.Render(); R.Finish()
    R.Destroy(); R2.Destroy()
