"""What the exact work eliminations of DESIGN.md section 4 remove, per pass, at a bench workload's full size: the oracle's work
(orc_work_counters: rays, march steps, SDF evaluations -- what the reference's shaders do) against the shipped kernels'
(`make -C madarch_amd/csrc counters`: the same kernels with the counters in, selected here through MADARCH_HIP_LIBRARY)
and the lanes alive per march loop.  Run on the GPU box:  python scripts/work_counters.py [workload]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("MADARCH_HIP_LIBRARY", os.path.join(ROOT, "madarch_amd", "csrc", "libmadarch_hip_counters.so"))
import bench  # noqa: E402
from madarch_amd import _binding as B  # noqa: E402
from oracle_engine import ORC_OPT_THREADS, oracle_binding  # noqa: E402

workload = sys.argv[1] if len(sys.argv) > 1 else "global_illumination_1080p_ddgi8x8x8"
hip, orc = B.hip_binding(), oracle_binding()
assert hasattr(hip.lib, "mdh_diag_work"), "needs the counters build"
Rg, Ro = bench.make_renderer(workload, hip), bench.make_renderer(workload, orc)
Ro.Set_Option(ORC_OPT_THREADS, len(os.sched_getaffinity(0)))
for R in (Rg, Ro):
    for _ in range(3):
        R.Render()
Rg.Finish()
lanes = (C.c_ulonglong * 16)()
print("%s: work per frame, oracle (= the reference's shaders) against the shipped kernels" % workload)
tot_o = tot_g = 0
for p in (B.PASS_RADIANCE, B.PASS_VISIBILITY, B.PASS_SCATTERING, B.PASS_SCREEN):
    o, g = (C.c_uint64 * 3)(), (C.c_ulonglong * 4)()
    orc.lib.orc_work_counters(Ro._h, p, o)
    hip.lib.mdh_diag_work(Rg._h, p, g)
    if not o[2]:
        continue
    tot_o += o[2]; tot_g += g[2]
    print("   %-11s oracle: %11d rays %12d march steps %12d SDF evaluations | kernels: %12d march steps %12d SDF evaluations (%d of them the arg-min at hit points) = %.1f %% of the oracle's evaluations" % (
        B.PASS_NAMES[p], o[0], o[1], o[2], g[1], g[2], g[3], 100.0 * g[2] / o[2]))
print("   frame: %d SDF evaluations in the oracle, %d in the kernels: %.1f %% eliminated" % (tot_o, tot_g, 100.0 * (1.0 - tot_g / tot_o)))
