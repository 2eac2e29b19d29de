#!/bin/bash
# On the GPU box: the headline workload's other bench lines (the driver's 20-step run, fp32 atlases, a light set anew
# every frame, frames delivered to the host, the one-rank RCCL exchange path), ball_game, and the shard emulation at
# both sizes -> gpurun_out/variants_<TAG>/ ; copy what is to be kept into profiles/.
TAG=${1:-r02}
cd "$(dirname "$0")/.." || exit 1
OUT=gpurun_out/variants_$TAG
mkdir -p $OUT
B="timeout -k 10 200 python bench.py --no-cpu-baseline"
$B --steps 20 --warmup 5 2>/dev/null | tail -1 > $OUT/bench_driver_20steps.json &&
$B --steps 200 --warmup 20 --atlas f32 2>/dev/null | tail -1 > $OUT/bench_f32_atlases.json &&
$B --steps 200 --warmup 20 --animate-light 2>/dev/null | tail -1 > $OUT/bench_animated_light.json &&
$B --steps 200 --warmup 20 --swap-buffers 2>/dev/null | tail -1 > $OUT/bench_swap_buffers.json &&
$B --steps 200 --warmup 20 --rehearse-rccl 2>/dev/null | tail -1 > $OUT/bench_rehearse_rccl.json &&
timeout -k 10 300 python scripts/emulate_shards.py 4096 4096 $OUT/shards_4096.json > $OUT/shards_4096.log 2>&1 &&
timeout -k 10 300 python scripts/emulate_shards.py 1920 1080 $OUT/shards_1080p.json > $OUT/shards_1080p.log 2>&1
for f in driver_20steps f32_atlases animated_light swap_buffers rehearse_rccl; do
  python3 -c "
import json
d=json.loads(open('$OUT/bench_$f.json').read())
print('%-16s %8.1f Mpix/s %.4f ms | serial %8.1f' % ('$f', d['value'], d['ms_per_step'], d.get('value_serial', 0)))"
done
cat $OUT/shards_4096.log $OUT/shards_1080p.log
