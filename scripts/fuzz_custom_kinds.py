"""Differential fuzzing of user-defined kinds: random Distance / Normal expression trees (tests/fuzz_kinds.py) must give
the same distances (Eval_Distance_To: the kernels' interpreter against the oracle's tree walk) and the same frames
(hiprtc build, interpreter build, oracle).  Usage: python scripts/fuzz_custom_kinds.py [first seed] [seeds]"""
import os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle_engine import oracle_binding
from madarch_amd import _binding as B
from fuzz_kinds import build, compare

orc = oracle_binding()
hip = orc if os.environ.get("FUZZ_ORACLE_ONLY") else B.hip_binding()
first, count = (int(sys.argv[1]) if len(sys.argv) > 1 else 0), (int(sys.argv[2]) if len(sys.argv) > 2 else 20)
bad = []
for seed in range(first, first + count):
    try:
        want = build(seed, orc, 0)
        for jit in ((0,) if hip is orc else (1, 0)):
            compare(build(seed, hip, jit), want, "(jit %d)" % jit)
        print("seed %d ok" % seed, flush=True)
    except Exception as e:  # noqa: BLE001
        bad.append(seed)
        print("seed %d FAILED: %s: %s" % (seed, type(e).__name__, str(e).splitlines()[0] if str(e) else ""), flush=True)
        if os.environ.get("FUZZ_TRACE"):
            traceback.print_exc()
print("failed seeds:", bad)
sys.exit(1 if bad else 0)
