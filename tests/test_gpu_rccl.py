"""The exchange of the sharded frame on the device.

1. The form the product uses: the communicator INSIDE the library (mdh_comm_unique_id / mdh_comm_init, RCCL opened
   by libmadarch_hip.so): Renderers.Render itself runs radiance slice -> all-gather in place on the probe stream ->
   irradiance -> screen tiles.  Driven through the C ABI from a process that never imports torch, and from the C++
   example program; with the one GPU of the test box the communicator has one rank (RCCL refuses two ranks on one
   device), which still drives the whole path: library load, join, the in-place collective on the probe stream of
   frames in flight, both forms of the exchange (all-gather / grouped broadcasts), barrier, reduction, leaving.
2. The earlier form (madarch_amd.sharding.DeviceExchange), kept for callers that own an RCCL group: the atlas
set of the open frame is wrapped as a torch tensor without a copy and all-gathered in place with
RCCL on the renderer's probe stream.  With one GPU on
the test box the group has a single rank, which still drives the whole code path (pointer
aliasing, stream hand-over, in-place all_gather_into_tensor); the 2-rank logic is covered on the
CPU by test_sharding_gloo.py and on the GPU through the host exchange by
test_gpu_parity.py::test_sharded_frame_equals_whole_frame."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import os, sys
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import numpy as np, torch, torch.distributed as dist
from helpers import SMALL_PROBES, make, same_bits, snapshot
from madarch_amd import _binding as B, sharding
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29611")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
hip = B.hip_binding()
want = snapshot(make("global_illumination", 64, 40, hip, probes=SMALL_PROBES), 2)
R = make("global_illumination", 64, 40, hip, probes=SMALL_PROBES)
ex = sharding.DeviceExchange(dist, R, torch.device("cuda", 0))
frame = sharding.ShardedFrame(R, 0, 1, ex)
for _ in range(2):
    frame.Render()                               # Frame_Begin, probe passes with the in-place all-gathers
                                                 # (one rank: must leave the atlas intact) on the probe stream, Frame_End
R.Finish()
torch.cuda.synchronize()
# the torch view really aliases the library's atlas
full, off, own, total = ex._view(R, B.TEX_IRRADIANCE)
assert own == total and full.numel() == total
host = R.Read_Texture(B.TEX_IRRADIANCE)
assert int(full.sum().item()) > 0
got = {"image": R.Read_Framebuffer(), "radiance": R.Read_Texture(B.TEX_RADIANCE), "irradiance": host}
for k in got:
    assert same_bits(got[k], want[k]), k
dist.destroy_process_group()
print("RCCL_PATH_OK")
""" % (ROOT, ROOT)


def test_device_exchange_single_rank():
    out = subprocess.run([sys.executable, "-c", SCRIPT], capture_output=True, text=True, timeout=300)
    assert "RCCL_PATH_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]


COMM_SCRIPT = r"""
import os, sys
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import numpy as np
from helpers import SMALL_PROBES, ODD_PROBES, make, same_bits, snapshot
from madarch_amd import _binding as B
hip = B.hip_binding()
for probes in (SMALL_PROBES, ODD_PROBES):
    want = snapshot(make("global_illumination", 64, 40, hip, probes=probes), 3)
    R = make("global_illumination", 64, 40, hip, probes=probes)
    ident = R.Comm_Unique_Id()
    assert len(ident) == 128 and any(ident)
    R.Comm_Init(ident, 0, 1)
    assert (R.Get_Option(B.OPT_RANK), R.Get_Option(B.OPT_WORLD)) == (0, 1)
    try:
        R.Set_Option(B.OPT_WORLD, 2)        # rank and world belong to the communicator now
        raise SystemExit("MDH_OPT_WORLD was accepted under a communicator")
    except B.MadarchError as e:
        assert e.status == B.MDH_E_STATE
    R.Set_Option(B.OPT_TIMING, 1)
    got = snapshot(R, 3)                    # frames in flight, the all-gather between the probe passes of each
    for k in want:
        assert same_bits(got[k], want[k]), k
    ms, n = R.Pass_Time(B.PASS_EXCHANGE)
    assert n == 3 and ms > 0.0, (ms, n)
    R.Set_Option(B.OPT_IRRADIANCE_ALL, 0)   # two exchanges per frame
    R.Render(); R.Comm_Barrier()
    assert R.Pass_Time(B.PASS_EXCHANGE)[1] == 5
    assert R.Comm_Max(1.25) == 1.25
    img = R.Read_Framebuffer()
    R.Comm_Reduce_Framebuffer(0)
    assert same_bits(R.Read_Framebuffer(), img)
    R.Comm_Destroy()
    R.Set_Option(B.OPT_WORLD, 1)            # ... and are the caller's again
    R.Render(); R.Finish()
    R.Destroy()
assert "torch" not in sys.modules
print("LIBRARY_COMM_OK")
""" % (ROOT, ROOT)


@pytest.mark.parametrize("form", ["allgather", "broadcast"])
def test_library_communicator_single_rank_without_torch(form):
    env = dict(os.environ, MADARCH_HIP_EXCHANGE=form)
    out = subprocess.run([sys.executable, "-c", COMM_SCRIPT], capture_output=True, text=True, timeout=300, env=env)
    assert "LIBRARY_COMM_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]


def test_cpp_example_joins_a_node(tmp_path):
    """examples/global_illumination.cpp --rank 0 --world 1 --id-file: a C++ host on the C ABI alone forms the communicator
    (id through a file) and renders the frames the plain program renders."""
    exe = os.path.join(ROOT, "examples", "bin", "global_illumination")
    assert os.path.exists(exe), "build first (python -c 'import __graft_entry__ as g; g.build()')"
    a, b = str(tmp_path / "plain.f32"), str(tmp_path / "node.f32")
    for out, extra in ((a, []), (b, ["--rank", "0", "--world", "1", "--id-file", str(tmp_path / "id")])):
        r = subprocess.run([exe, "96", "64", "3", out] + extra, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
    assert open(a, "rb").read() == open(b, "rb").read()
