#!/bin/bash
# on the GPU box: rank 0's share of a sharded frame (scripts/emulate_shards.py) for every variant library
cd "$(dirname "$0")/.." || exit 1
for lib in madarch_amd/csrc/variants/libmadarch_hip_*.so; do
  tag=$(basename $lib .so); echo "== ${tag#libmadarch_hip_}"
  MADARCH_HIP_LIBRARY=$PWD/$lib timeout -k 10 200 python scripts/emulate_shards.py
done
