"""Differential fuzzing: random scenes (tests/fuzz_scenes.py) rendered by the HIP library and by the CPU oracle
must agree like the parity tests demand.  Usage: python scripts/fuzz_parity.py [first seed] [seeds]"""
import os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle_engine import oracle_binding
from madarch_amd import _binding as B
from fuzz_scenes import build, compare

orc = oracle_binding()
hip = orc if os.environ.get("FUZZ_ORACLE_ONLY") else B.hip_binding()  # (oracle only: a dry run of the generator on a CPU)
first, count = (int(sys.argv[1]) if len(sys.argv) > 1 else 0), (int(sys.argv[2]) if len(sys.argv) > 2 else 50)
bad = []
for seed in range(first, first + count):
    try:
        compare(build(seed, hip), build(seed, orc))
        print("seed %d ok" % seed, flush=True)
    except Exception as e:  # noqa: BLE001
        bad.append(seed)
        print("seed %d FAILED: %s: %s" % (seed, type(e).__name__, str(e).splitlines()[0] if str(e) else ""), flush=True)
        if os.environ.get("FUZZ_TRACE"):
            traceback.print_exc()
print("failed seeds:", bad)
sys.exit(1 if bad else 0)
