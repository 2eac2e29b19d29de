#!/bin/bash
# on the GPU box: MADARCH_HIP_ORDER_FLOOR (thousandths of the median duration below which tiles keep image order) x workloads
cd "$(dirname "$0")/.." || exit 1
line() { python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('%-10s %-38s %8.1f Mpix/s in flight, %8.1f serial | in flight %s | serial %s' % ('$1', d['config']['workload'], d['value'], d['value_serial'], {k: v['ms_avg'] for k, v in d['passes'].items()}, {k: v['ms_avg'] for k, v in d['passes_serial'].items()}))"; }
for r in 1 2; do
for w in global_illumination_1080p_ddgi8x8x8 simple_scene_1080p_direct light_shafts_1080p global_illumination_4096sq_ddgi8x8x8; do
  python bench.py --workload $w --no-cpu-baseline --screen-order 0 2>/dev/null | tail -1 | line image
  for f in ${FLOORS:-0 1500 2000 2500 3000}; do
    MADARCH_HIP_ORDER_FLOOR=$f python bench.py --workload $w --no-cpu-baseline 2>/dev/null | tail -1 | line floor$f
  done
done; done
