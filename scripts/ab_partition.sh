#!/bin/bash
# on the GPU box: the partitioned workloads for every variant library
cd "$(dirname "$0")/.." || exit 1
for r in 1 2; do
for lib in madarch_amd/csrc/variants/libmadarch_hip_*.so; do
  tag=$(basename $lib .so); tag=${tag#libmadarch_hip_}
  MADARCH_HIP_LIBRARY=$PWD/$lib python bench.py --workload simple_scene_1080p_direct --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('%-8s simple_scene %8.1f Mpix/s %7.3f ms serial screen %s' % ('$tag', d['value'], d['ms_per_step'], d['passes_serial']['screen']['ms_avg']))"
  echo -n "$tag "; MADARCH_HIP_LIBRARY=$PWD/$lib python scripts/ball_game_bench.py
done
done
