#!/bin/bash
# on the GPU box: the whole GPU suite, the schedule stress test and a fresh profile set (TAG = $1)
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1 || { tail -30 gpurun_out/gpu_tests.log; exit 1; }
tail -2 gpurun_out/gpu_tests.log
timeout -k 10 500 python scripts/stress_pipelining.py 6 150 > gpurun_out/stress.log 2>&1 || { tail -20 gpurun_out/stress.log; exit 1; }
grep -c identical gpurun_out/stress.log
bash scripts/collect_profiles.sh ${1:-i} > gpurun_out/collect.log 2>&1
python - <<PY
import json
d=json.loads(open("gpurun_out/profiles_${1:-i}/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["passes_serial"], d["valu_issue"]["frac"], d["cpu_baseline"]["value"])
PY
