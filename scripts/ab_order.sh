#!/bin/bash
# on the GPU box: MDH_OPT_SCREEN_ORDER on / off for the bench workloads
cd "$(dirname "$0")/.." || exit 1
for r in 1 2; do
for w in global_illumination_1080p_ddgi8x8x8 simple_scene_1080p_direct light_shafts_1080p; do
for o in 0 1; do
  python bench.py --workload $w --no-cpu-baseline --screen-order $o $BENCH_ARGS 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('%-38s order %d: %8.1f Mpix/s in flight, %8.1f serial | in flight %s | serial %s' % ('$w', $o, d['value'], d['value_serial'], {k: v['ms_avg'] for k, v in d['passes'].items()}, {k: v['ms_avg'] for k, v in d['passes_serial'].items()}))"
done; done; done
