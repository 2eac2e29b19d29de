#!/bin/bash
# on the GPU box: the slowest tiles of an ordered screen launch as four wavefronts each (MADARCH_HIP_SPLIT_FIRST, thousandths of the tiles)
cd "$(dirname "$0")/.." || exit 1
for w in ${WORKLOADS:-global_illumination_1080p_ddgi8x8x8 simple_scene_1080p_full simple_scene_1080p_direct}; do
  for pm in ${PERMILLE:-0 5 20 50 100 0}; do
    MADARCH_HIP_SPLIT_FIRST=$pm timeout -k 10 200 python bench.py --workload $w --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 |
      python -c "import sys,json; d=json.loads(sys.stdin.read()); q=d.get('passes_serial') or {}; print('%-40s first %4d/1000: %8.1f Mpix/s in flight | serial %8.1f  screen %.4f ms' % ('$w', $pm, d['value'], d.get('value_serial',0), q.get('screen',{}).get('ms_avg',0)))"
  done
done
