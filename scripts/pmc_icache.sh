#!/bin/bash
# On the GPU box: instruction-cache requests / hits / misses per kernel (serial and pipelined schedule)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_icache
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for mode in serial piped; do
  B="python3 $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-serial-segment"
  [ $mode = serial ] && B="$B --serial"
  timeout -k 10 300 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_IFETCH --kernel-trace --output-format csv -d $OUT/$mode -- $B > /dev/null 2>&1
  python3 - "$OUT/$mode" $mode <<'PY'
import collections, csv, glob, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
        if k.startswith("k_"): per[(k, r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
    for (k, _), c in per.items():
        for n, v in c.items(): agg[k][n].append(v)
for k, c in sorted(agg.items()):
    m = {n: sum(v) / len(v) for n, v in c.items()}
    print(sys.argv[2], k, {n: round(v) for n, v in m.items()})
PY
done
