set -e
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q -k "swap or api_behaviour" > gpurun_out/swap_test.log 2>&1 || { tail -30 gpurun_out/swap_test.log; exit 1; }
tail -3 gpurun_out/swap_test.log
python scripts/swap_buffers_bench.py 400 abcd 0
python scripts/swap_buffers_bench.py 400 abcd 1
