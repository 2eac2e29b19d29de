#!/bin/bash
# On the GPU box, after profiles/*_traffic.json of this round are in place: the four bench lines again, so that each cites
# the traffic and instruction counts collected with the same kernels -> gpurun_out/rebench/<workload>.json
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/rebench
for w in global_illumination_1080p_ddgi8x8x8 simple_scene_1080p_direct light_shafts_1080p global_illumination_4096sq_ddgi8x8x8; do
  s=200; [ $w = global_illumination_4096sq_ddgi8x8x8 ] && s=60
  timeout -k 10 400 python bench.py --workload $w --steps $s --warmup 20 2>/dev/null | tail -1 > gpurun_out/rebench/$w.json || exit 1
  python3 -c "
import json
d=json.loads(open('gpurun_out/rebench/$w.json').read())
print('%-42s %8.1f Mpix/s %.4f ms | serial %8.1f | valu %.3f roofline %.4f cpu %s' % ('$w', d['value'], d['ms_per_step'], d.get('value_serial',0), (d.get('valu_issue') or {}).get('frac',0), d['roofline']['frac'], (d.get('cpu_baseline') or {}).get('value')))"
done
