"""The library's RCCL exchange with MORE THAN ONE rank: two processes, each with its own GPU, join a communicator
(mdh_comm_unique_id / mdh_comm_init) and render sharded frames in flight; the gathered atlases of every rank and the reduced
framebuffer must be a one-rank run's, bit for bit -- with a probe count the world divides (in-place ncclAllGather at
buf + count * rank) and with one it does not (grouped ncclBroadcasts over own_probes' uneven slices).

Needs TWO GPUs: skipped on the one-GPU boxes of this pipeline, where RCCL refuses two ranks on one device ("invalid usage";
tests/test_gpu_peer_exchange.py runs the same sharded frame there through the peer exchange).  UNTIL THIS TEST HAS RUN ON A
TWO-GPU MACHINE THE N > 1 RCCL PATH IS UNVERIFIED ON HARDWARE (DESIGN.md section 7 says so)."""
import ctypes as C
import multiprocessing as mp
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def gpu_count():
    try:
        n = C.c_int(0)
        return n.value if C.CDLL("libamdhip64.so").hipGetDeviceCount(C.byref(n)) == 0 else 0
    except OSError:
        return 0


def _rank(rank, world, probes_name, frames, conn):
    try:
        sys.path.insert(0, ROOT)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import helpers
        from madarch_amd import _binding as B
        hip = B.hip_binding()
        R = helpers.make("global_illumination", 96, 64, hip, probes=getattr(helpers, probes_name), Device=rank)
        if rank == 0:
            conn.send(R.Comm_Unique_Id())
        ident = conn.recv()
        R.Comm_Init(ident, rank, world)
        R.Set_Option(B.OPT_TIMING, 1)
        for _ in range(frames):
            R.Render()
        R.Comm_Barrier()
        own = R.Read_Framebuffer()
        R.Comm_Reduce_Framebuffer(0)
        out = {"own": own, "reduced": R.Read_Framebuffer(), "radiance": R.Read_Texture(B.TEX_RADIANCE),
               "irradiance": R.Read_Texture(B.TEX_IRRADIANCE), "exchanges": R.Pass_Time(B.PASS_EXCHANGE)[1]}
        conn.send(out)
        conn.recv()
        R.Comm_Destroy()
        R.Destroy()
        conn.send("left")
    except Exception as e:  # noqa: BLE001
        import traceback
        conn.send("ERROR " + repr(e) + "\n" + traceback.format_exc())


@pytest.mark.skipif(gpu_count() < 2, reason="RCCL refuses two ranks of a communicator on one device: needs two GPUs")
@pytest.mark.parametrize("probes_name", ["SMALL_PROBES", "ODD_PROBES"])
def test_two_rccl_ranks_equal_one(hip, probes_name):
    import helpers
    world, frames = 2, 6
    want = helpers.snapshot(helpers.make("global_illumination", 96, 64, hip, probes=getattr(helpers, probes_name)), frames)
    ctx = mp.get_context("spawn")
    pipes, procs = [], []
    for rank in range(world):
        a, b = ctx.Pipe()
        p = ctx.Process(target=_rank, args=(rank, world, probes_name, frames, b), daemon=True)
        p.start()
        pipes.append(a); procs.append(p)

    def get(c, what):
        assert c.poll(300), "a rank did not send its " + what
        v = c.recv()
        assert not (isinstance(v, str) and v.startswith("ERROR")), v
        return v
    try:
        ident = get(pipes[0], "communicator id")
        for c in pipes:
            c.send(ident)
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
    assert helpers.same_bits(np.sum([o["own"] for o in outs], axis=0), want["image"])
    assert helpers.same_bits(outs[0]["reduced"], want["image"]), "ncclReduce of the ranks' framebuffers on rank 0"
    for rank, o in enumerate(outs):
        assert helpers.same_bits(o["radiance"], want["radiance"]), "radiance atlas of rank %d" % rank
        assert helpers.same_bits(o["irradiance"], want["irradiance"]), "irradiance atlas of rank %d" % rank
        assert o["exchanges"] == frames
