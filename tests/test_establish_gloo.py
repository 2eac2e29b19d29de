"""sharding.establish -- how an N-rank run forms its communicator inside the library or falls back -- on the CPU
with two gloo ranks and a scripted engine: whatever fails, and on whichever rank, BOTH ranks must end up with the
same kind of exchange, without hanging and without a second process."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class _Engine:
    """Stands in for a Renderer: only what establish() touches.  `script` says what fails on this rank."""

    class _b:  # the binding: has comm_init unless the script says the engine has none
        pass

    def __init__(self, rank, script):
        from madarch_amd import _binding as B
        self.B, self.rank, self.script, self.calls, self.opts = B, rank, script, [], {}
        self._b = type("b", (), {})()
        if script != "no_comm" and not script.startswith("peer"):
            self._b.comm_init = True
        if script.startswith("peer"):  # no communicator, but a peer exchange ("peer_fine", "peer_init@1", "peer_render@0")
            self._b.peer_init = True

    def Peer_Export(self):
        self.calls.append("peer_export")
        return bytes([self.rank + 1]) * self.B.PEER_BLOB_BYTES

    def Peer_Init(self, blobs, rank, world):
        self.calls.append("peer_init")
        assert len(blobs) == self.B.PEER_BLOB_BYTES * world and all(blobs[q * self.B.PEER_BLOB_BYTES] == q + 1 for q in range(world))
        self._fail("peer_init")
        self.peer = True

    peer = False

    def Comm_Destroy(self):
        self.calls.append("destroy")
        self.peer = False

    def _fail(self, what):
        if self.script == what or self.script == "%s@%d" % (what, self.rank):
            raise self.B.MadarchError(self.B.MDH_E_COMM, "scripted failure of %s" % what)

    def Comm_Unique_Id(self):
        self.calls.append("id")
        self._fail("id")
        return bytes(range(1, 129))

    def Comm_Init(self, ident, rank, world):
        self.calls.append("init")
        assert ident == bytes(range(1, 129))
        self._fail("init")

    def Render(self):
        self.calls.append("render")
        self._fail("render")
        if self.peer:
            self._fail("peer_render")
        if self.script == "hang@%d" % self.rank:
            import time
            while not self.aborted:
                time.sleep(0.05)
            raise self.B.MadarchError(self.B.MDH_E_COMM, "aborted")

    aborted = False

    def Comm_Barrier(self):
        self.calls.append("barrier")

    def Comm_Abort(self):
        self.calls.append("abort")
        self.aborted = True

    def Finish(self):
        pass

    def Set_Option(self, o, v):
        self.opts[o] = v

    def Write_Texture(self, *a):
        pass

    def Texture_Shape(self, tex):
        return (4, 4, 3)


def _worker(rank, world, port, script, out_dir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from madarch_amd import sharding
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    E = _Engine(rank, script)
    exchange, how = sharding.establish(E, rank, world, dist, timeout_s=3.0)
    with open(os.path.join(out_dir, "r%d" % rank), "w") as f:
        f.write("%s|%s|%s" % (how if exchange is None else exchange.name, how, ",".join(E.calls)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("script,want", [
    ("fine", "rccl"),
    ("no_comm", "host exchange"),
    ("id@1", "host exchange"),      # rank 1 cannot load librccl: nobody enters the collective join
    ("init@0", "host exchange"),    # rank 0's join fails: rank 1, which joined, aborts its communicator
    ("render@1", "host exchange"),  # a trial frame fails on one rank
    ("hang@1", "host exchange"),    # a trial frame never returns on one rank: the watchdog aborts
    ("peer_fine", "peer"),          # no communicator anywhere, the peer exchange carries the run
    ("peer_init@1", "host exchange"),    # rank 1 cannot open the handles: rank 0, which could, leaves again
    ("peer_render@0", "host exchange"),  # a trial frame of the peer exchange fails on one rank
])
def test_both_ranks_agree(tmp_path, script, want):
    import torch.multiprocessing as mp
    port = 29300 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, script, str(tmp_path)), nprocs=2, join=True)
    got = [open(os.path.join(str(tmp_path), "r%d" % r)).read().split("|") for r in range(2)]
    assert got[0][0] == got[1][0] == want, got
    if script == "id@1":
        assert "init" not in got[0][2] and "init" not in got[1][2]
    if script == "init@0":
        assert "abort" in got[1][2]
    if script == "hang@1":
        assert "abort" in got[1][2]
    if want == "rccl":
        assert got[0][1] == "rccl" and got[0][2].count("render") == 2 and "barrier" in got[0][2]
    if want == "peer":
        assert got[0][1] == got[1][1] == "peer" and got[0][2].count("render") == 2 and "destroy" not in got[0][2]
    if script == "peer_init@1":
        assert "destroy" in got[0][2]
    if script == "peer_render@0":
        assert "destroy" in got[0][2] and "destroy" in got[1][2]
