#!/bin/bash
# on the GPU box: every variant library x the bench workloads (MDH_OPT_SCREEN_ORDER as the library has it), and image order once
cd "$(dirname "$0")/.." || exit 1
line() { python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('%-10s %-38s %8.1f Mpix/s in flight, %8.1f serial | in flight %s | serial %s' % ('$1', d['config']['workload'], d['value'], d['value_serial'], {k: v['ms_avg'] for k, v in d['passes'].items()}, {k: v['ms_avg'] for k, v in d['passes_serial'].items()}))"; }
for r in 1 2; do
for w in global_illumination_1080p_ddgi8x8x8 simple_scene_1080p_direct light_shafts_1080p; do
  python bench.py --workload $w --no-cpu-baseline --screen-order 0 2>/dev/null | tail -1 | line image
  for lib in madarch_amd/csrc/variants/libmadarch_hip_*.so; do
    tag=$(basename $lib .so); tag=${tag#libmadarch_hip_}
    MADARCH_HIP_LIBRARY=$PWD/$lib python bench.py --workload $w --no-cpu-baseline 2>/dev/null | tail -1 | line $tag
  done
done; done
python scripts/ball_game_bench.py
