#!/bin/bash
# builds kernel variants for A/B runs: scripts/build_variants.sh tag1="-DFOO=1" tag2="-DBAR=2" ...
# -> madarch_amd/csrc/variants/libmadarch_hip_<tag>.so (run with MADARCH_HIP_LIBRARY=...)
cd "$(dirname "$0")/../madarch_amd/csrc" || exit 1
mkdir -p variants
for spec in "$@"; do
  tag="${spec%%=*}"; flags="${spec#*=}"
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math \
      -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-flush-denormals-to-zero -fno-vectorize -fno-slp-vectorize -Wno-unused-value -Wno-unused-function \
      $flags -shared -o variants/libmadarch_hip_$tag.so mdh_api.hip -ldl 2>&1 | grep -E "error" ; echo "built $tag" ) &
done
wait
