#!/bin/bash
# On the GPU box: the bench line, the rocprofv3 kernel stats of the same command and the PMC
# counters of its kernels (separate passes, --kernel-trace only, as MI355X_MICROARCH.md prescribes).
# Usage: scripts/collect_profiles.sh TAG [WORKLOAD]  -> gpurun_out/profiles_TAG/{bench.json,kernel_stats.csv,
#        kernel_stats_serial.csv,traffic.json}   (WORKLOAD: a --workload of bench.py; default: the headline config)
TAG=${1:-run}
WL=${2:-global_illumination_1080p_ddgi8x8x8}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/profiles_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $ROOT/bench.py --workload $WL > $OUT/bench.json 2> $OUT/bench.err || exit 1
B="python3 $ROOT/bench.py --workload $WL --steps 20 --warmup 3 --prewarm-s 0.02 --no-cpu-baseline --no-serial-segment"
# kernel durations of the default (pipelined) schedule and of the serial one
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $B > /dev/null 2>&1
cp $OUT/trace/*/*kernel_stats.csv $OUT/kernel_stats.csv
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_s -- $B --serial > /dev/null 2>&1
cp $OUT/trace_s/*/*kernel_stats.csv $OUT/kernel_stats_serial.csv
# counters: serial schedule, one kernel on the chip at a time
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- $B --serial > /dev/null 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- $B --serial > /dev/null 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY --kernel-trace --output-format csv -d $OUT/pmc_sq -- $B --serial > /dev/null 2>&1
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
kern = {}
for k, c in agg.items():
    m = {n: sum(v) / len(v) for n, v in c.items()}
    # FETCH_SIZE / WRITE_SIZE are in KiB.  MI355X_MICROARCH.md (HBM): on gfx950 FETCH_SIZE reports half
    # of the bytes of a wide coalesced read -- doubled here as the guide prescribes; for the narrow
    # gathers and scratch reloads of these kernels (uncalibrated widths) that makes it an upper bound.
    m["hbm_read_bytes_per_launch"] = 2.0 * m.get("FETCH_SIZE", 0) * 1024
    m["hbm_write_bytes_per_launch"] = m.get("WRITE_SIZE", 0) * 1024
    m["hbm_bytes_per_launch"] = m["hbm_read_bytes_per_launch"] + m["hbm_write_bytes_per_launch"]
    kern[k] = m
bench = json.loads(open(out + "/bench.json").read().strip().splitlines()[-1])
res = {"workload": bench["config"]["workload"], "n_gpus": bench["n_gpus"], "schedule": "serial (MDH_OPT_FRAME_OVERLAP = 0)",
       "units": "per launch, averaged over the launches of one bench run; SQ_* cycle counters count quad-cycles summed over all waves/CUs",
       "kernels": kern}
json.dump(res, open(out + "/traffic.json", "w"), indent=1, sort_keys=True)
PY
rm -rf $OUT/trace $OUT/trace_s $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq
cat $OUT/bench.json
