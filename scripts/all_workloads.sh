#!/bin/bash
# on the GPU box: every bench workload plus the example programs' own benches, one line each
cd "$(dirname "$0")/.." || exit 1
for w in global_illumination_1080p_ddgi8x8x8 simple_scene_1080p_direct light_shafts_1080p global_illumination_4096sq_ddgi8x8x8; do
  python bench.py --workload $w --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('%-40s %8.1f Mpix/s %7.3f ms  serial passes %s' % ('$w', d['value'], d['ms_per_step'], {k: v['ms_avg'] for k, v in (d.get('passes_serial') or {}).items()}))"
done
python bench.py --no-cpu-baseline --atlas f32 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('%-40s %8.1f Mpix/s %7.3f ms' % ('GI 1080p, fp32 atlases', d['value'], d['ms_per_step']))"
python bench.py --no-cpu-baseline --animate-light 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('%-40s %8.1f Mpix/s %7.3f ms' % ('GI 1080p, light set every frame', d['value'], d['ms_per_step']))"
python bench.py --no-cpu-baseline --swap-buffers 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('%-40s %8.1f Mpix/s %7.3f ms' % ('GI 1080p, Swap_Buffers every frame', d['value'], d['ms_per_step']))"
python bench.py --no-cpu-baseline --rehearse-rccl 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('%-40s %8.1f Mpix/s %7.3f ms' % ('GI 1080p, one-rank RCCL rehearsal', d['value'], d['ms_per_step']))"
python scripts/ball_game_bench.py
python scripts/custom_kinds_bench.py 2>&1 | tail -4
