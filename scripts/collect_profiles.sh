#!/bin/bash
# On the GPU box: bench line + rocprofv3 kernel stats + HBM traffic counters of the same command.
# Usage: scripts/collect_profiles.sh TAG   -> gpurun_out/profiles_TAG/{bench.json,kernel_stats.csv,traffic.json}
TAG=${1:-run}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/profiles_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $ROOT/bench.py --steps 30 --warmup 8 > $OUT/bench.json 2> $OUT/bench.err || exit 1
B="python3 $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $B > /dev/null 2>&1
cp $OUT/trace/*/*kernel_stats.csv $OUT/kernel_stats.csv
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- $B > /dev/null 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- $B > /dev/null 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY --kernel-trace --output-format csv -d $OUT/pmc_sq -- $B > /dev/null 2>&1
python3 - "$OUT" <<'PY'
import collections, csv, glob, json, sys
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pmc_*/**/*counter_collection.csv", recursive=True):
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
        if k.startswith("k_"): per[(k, r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
    for (k, _), c in per.items():
        for n, v in c.items(): agg[k][n].append(v)
res = {}
for k, c in agg.items():
    m = {n: sum(v) / len(v) for n, v in c.items()}
    m["hbm_bytes_per_launch"] = (m.get("FETCH_SIZE", 0) + m.get("WRITE_SIZE", 0)) * 1024
    res[k] = m
json.dump(res, open(out + "/traffic.json", "w"), indent=1, sort_keys=True)
print(json.dumps(res, indent=1, sort_keys=True))
PY
rm -rf $OUT/trace $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq
cat $OUT/bench.json
