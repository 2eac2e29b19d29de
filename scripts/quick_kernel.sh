#!/bin/bash
# Resource usage (and, with ISA=path, the assembly) of ONE kernel instantiation, in seconds instead of the minute the
# whole library takes:   scripts/quick_kernel.sh 'k_screen<1, 2, false, false>' [extra -D flags]
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
K="$1"; shift
TMP=${TMPDIR:-/tmp}/mdh_quick_$$.hip
cat > $TMP <<EOT
#include "$ROOT/madarch_amd/csrc/mdh_kernels.h"
template __global__ void $K(${ARGS:-KScene, KProbes, KVolumetrics, KCamera, ScreenArgs});
EOT
OUT=${ISA:-/dev/null}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt \
  -fno-gpu-flush-denormals-to-zero -fno-vectorize -fno-slp-vectorize -Wno-unused-value "$@" \
  -Rpass-analysis=kernel-resource-usage --cuda-device-only -S -o $OUT $TMP 2>&1 | grep -E "Function Name|VGPRs:|SGPRs:|ScratchSize|Occupancy" | sed -e 's/.*remark: *//' -e 's/ \[-Rpass.*//' | tail -5 | tr '\n' ' '
echo
rm -f $TMP
