#!/bin/bash
# prints "kernel VGPRs SGPRs scratch occupancy LDS" for every kernel of mdh_api.hip
cd "$(dirname "$0")"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math \
  -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-flush-denormals-to-zero -fno-vectorize -fno-slp-vectorize -Wno-unused-value $EXTRA \
  -Rpass-analysis=kernel-resource-usage --cuda-device-only -c -o /dev/null mdh_api.hip 2>&1 |
python3 -c '
import re,sys
cur=None
for line in sys.stdin:
    m=re.search(r"Function Name: (\S+)",line)
    if m: cur=m.group(1); vals={}; continue
    m=re.search(r"remark:\s+(\w[\w ]*\w)(?: \[bytes/lane\]| \[waves/SIMD\]| \[bytes/block\])?: (\d+)",line)
    if m and cur:
        vals[m.group(1)]=m.group(2)
        if m.group(1).startswith("LDS Size"):
            print("%-75s VGPR %3s AGPR %3s SGPR %3s scratch %5s occ %s"%(cur[:75],vals.get("VGPRs"),vals.get("AGPRs"),vals.get("TotalSGPRs"),vals.get("ScratchSize"),vals.get("Occupancy")))
'
