"""bench.py's N-rank control flow, launched exactly as the driver launches it (python -m torch.distributed.run ...
bench.py --gpus N --steps K --warmup W), on the CPU: gloo, the oracle as the engine (--rehearse-cpu).  The frame the
ranks assemble must be the single-process frame, whatever N."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(n, port):
    cmd = [sys.executable]
    if n > 1:
        cmd += ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1", "--master-port", str(port)]
    cmd += [os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "2", "--warmup", "1", "--rehearse-cpu"]
    env = dict(os.environ, OMP_NUM_THREADS="2")
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert out.returncode == 0 and len(lines) == 1, out.stdout[-2000:] + out.stderr[-3000:]
    return json.loads(lines[0])


@pytest.mark.parametrize("n", [2, 3])
def test_bench_control_flow_with_n_ranks(n):
    one = _run(1, 0)
    many = _run(n, 29540 + n + (os.getpid() % 500))
    assert many["rehearsal"] and many["n_gpus"] == n and many["steps"] == 2
    assert many["frame_sha1"] == one["frame_sha1"]  # tiles and probe slices of n ranks = the whole frame
    # the oracle has no communicator: every rank agrees on the exchange through host memory, in the same process,
    # and the line says so; the pre-warm blocks and the serial segment ran with the same frame count on every rank
    # (or the ranks' collectives would have paired wrongly and the frames differed)
    assert "HOST EXCHANGE FALL-BACK" in many["parallelism"] and "no communicator" in many["parallelism"]
    assert many["serial_segment"] and one["serial_segment"]
