#!/bin/bash
# On the GPU box: where the wave cycles of each kernel go (serial schedule), three PMC passes.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_breakdown${1:+_$1}
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $ROOT/bench.py ${WORKLOAD:+--workload $WORKLOAD} --steps 10 --warmup 3 --prewarm-s 0.02 --no-cpu-baseline --no-serial-segment --serial"
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --kernel-trace --output-format csv -d $OUT/a -- $B > /dev/null 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH --kernel-trace --output-format csv -d $OUT/b -- $B > /dev/null 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_IFETCH SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_FLAT SQ_INSTS_FLAT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/c -- $B > /dev/null 2>&1
python3 - "$OUT" <<'PY'
import collections, csv, glob, json, sys
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/*/**/*counter_collection.csv", recursive=True):
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
        if k.startswith("k_"): per[(k, r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
    for (k, _), c in per.items():
        for n, v in c.items(): agg[k][n].append(v)
res = {k: {n: sum(v) / len(v) for n, v in c.items()} for k, c in agg.items()}
json.dump(res, open(out + "/breakdown.json", "w"), indent=1, sort_keys=True)
for k, m in sorted(res.items()):
    wc = m.get("SQ_WAVE_CYCLES", 1)
    print(k)
    for n in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_MISC", "SQ_ACTIVE_INST_FLAT", "SQ_INST_CYCLES_SALU"):
        if n in m: print("   %-22s %6.1f %% of wave cycles" % (n, 100 * m[n] / wc))
    for n in ("SQ_INSTS_VALU", "SQ_INSTS_VALU_TRANS_F32", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_BRANCH", "SQ_INSTS_FLAT", "SQ_IFETCH", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"):
        if n in m: print("   %-24s %12.0f  (%.3f per VALU instruction)" % (n, m[n], m[n] / max(m.get("SQ_INSTS_VALU", 1), 1)))
PY
rm -rf $OUT/a $OUT/b $OUT/c
