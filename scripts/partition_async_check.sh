set -e
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1 || { tail -30 gpurun_out/gpu_tests.log; exit 1; }
tail -2 gpurun_out/gpu_tests.log
timeout -k 10 500 python scripts/stress_pipelining.py 9 200 > gpurun_out/stress.log 2>&1 || { tail -20 gpurun_out/stress.log; exit 1; }
tail -18 gpurun_out/stress.log
python scripts/ball_game_bench.py
