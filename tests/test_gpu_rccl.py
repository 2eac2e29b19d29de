"""The device-side exchange of the sharded frame (madarch_amd.sharding.DeviceExchange): the atlas
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
