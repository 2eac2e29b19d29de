#!/bin/bash
# on the GPU box: k_irradiance alone (scripts/irr_scaling.py) under every variant library, then the wavefronts' cycles of the
# MDH_PHASES build (scripts/diag_irradiance.py)
cd "$(dirname "$0")/.." || exit 1
for lib in madarch_amd/csrc/libmadarch_hip.so madarch_amd/csrc/variants/libmadarch_hip_*.so; do
  echo "== $lib"
  MADARCH_HIP_LIBRARY=$PWD/$lib timeout -k 10 120 python scripts/irr_scaling.py 2>&1 | tail -6
done
if [ -f madarch_amd/csrc/variants/libmadarch_hip_phases.so ]; then
  MADARCH_HIP_LIBRARY=$PWD/madarch_amd/csrc/variants/libmadarch_hip_phases.so timeout -k 10 120 python scripts/diag_irradiance.py 2>&1 | tail -5
fi
