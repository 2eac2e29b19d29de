"""bench.py under the driver's own N-rank launch line on the ONE GPU of the test box (--one-device: every rank on device 0).
RCCL refuses two ranks on one device, sharding.establish falls to the peer exchange on every rank together, and the run must
complete with one JSON line that says so: the N > 1 control flow of bench.py (join, trial frames, pre-warm agreement, timed
region, serial segment, leaving) on real hardware, every round."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world", [2])
def test_bench_two_ranks_on_one_gpu(world):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--one-device", "--steps", "12", "--warmup", "3",
           "--prewarm-s", "0.05", "--no-cpu-baseline", "--comm-timeout-s", "60"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert out.returncode == 0 and len(lines) == 1, out.stdout[-2000:] + out.stderr[-3000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == world and d["value"] > 0 and d["scaling"] == "strong"
    assert d["config"]["exchange"] == "peer" and "ALL %d RANKS ON ONE GPU" % world in d["config"]["parallelism"]
    assert d["exchange"]["how"] == "peer" and d["exchange"]["ms_avg_max_over_ranks"] > 0
    assert d["passes"]["exchange"]["launches"] == 12
