"""Aggregate rocprofv3 --pmc counter_collection.csv files per kernel (mean per dispatch)."""
import collections, csv, glob, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for path in sys.argv[1:]:
    for f in glob.glob(path + "/**/*counter_collection.csv", recursive=True):
        per = collections.defaultdict(lambda: collections.defaultdict(float))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if not k.startswith("k_"): continue
            per[(k, r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
        for (k, _), c in per.items():
            for n, v in c.items(): agg[k][n].append(v)
for k, c in sorted(agg.items()):
    m = {n: sum(v) / len(v) for n, v in c.items()}
    print(k)
    for n in sorted(m): print("   %-28s %.5g" % (n, m[n]))
    if "SQ_INSTS_VALU" in m and "SQ_WAVES" in m: print("   => VALU instr per wave            %.0f" % (m["SQ_INSTS_VALU"] / m["SQ_WAVES"]))
    if "SQ_THREAD_CYCLES_VALU" in m and "SQ_ACTIVE_INST_VALU" in m: print("   => active lanes per VALU instr     %.1f / 64" % (m["SQ_THREAD_CYCLES_VALU"] / m["SQ_ACTIVE_INST_VALU"]))
    if "SQ_ACTIVE_INST_VALU" in m and "SQ_BUSY_CU_CYCLES" in m: print("   => VALU busy (ACTIVE_INST_VALU*4/BUSY_CU_CYCLES/4simd) %.3f" % (m["SQ_ACTIVE_INST_VALU"] * 4 / m["SQ_BUSY_CU_CYCLES"] / 4))
