"""Lanes alive per march loop and, through the space partition, lookups and candidate-pair iterations of ANY bench
workload's screen and radiance passes (needs a -DMDH_DIAG build: `make -C madarch_amd/csrc counters`, selected here
through MADARCH_HIP_LIBRARY).  Run on the GPU box:  python scripts/diag_workload.py [workload]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MADARCH_HIP_LIBRARY", os.path.join(ROOT, "madarch_amd", "csrc", "libmadarch_hip_counters.so"))
import bench
from madarch_amd import _binding as B
workload = sys.argv[1] if len(sys.argv) > 1 else "simple_scene_1080p_full"
hb = B.hip_binding()
R = bench.make_renderer(workload, hb)
buf = (C.c_ulonglong * 16)()
names = {0: "hit rays (primary / reflection)", 1: "soft shadow, first point", 2: "soft shadow, second point", 3: "probe visibility, first point (or queue)",
         4: "probe visibility, second point", 5: "partition lookups reaching a cell", 6: "candidate-pair iterations (walk_bits)", 7: "kind visits (general bit walk)"}
for f in range(3): R.Render()
R.Finish(); hb.lib.mdh_diag_read(buf)
print(workload)
for p, pname, units in ((B.PASS_RADIANCE, "radiance", None), (B.PASS_SCREEN, "screen", R.Width * R.Height / 64)):
    R.Render_Pass(p); R.Finish(); hb.lib.mdh_diag_read(buf)
    print(" ", pname)
    tot_e = tot_l = 0
    for t, n in names.items():
        e, l = buf[2 * t], buf[2 * t + 1]
        if t < 5: tot_e += e; tot_l += l
        if e: print("   %-44s wave-level %11d%s  lanes per %5.1f" % (n, e, (" (%7.1f per wavefront)" % (e / units)) if units else "", l / e))
    print("   march steps in all: %d wave-level, mean lanes %.1f" % (tot_e, tot_l / max(tot_e, 1)))
    g = (C.c_ulonglong * 4)()
    hb.lib.mdh_diag_work(R._h, p, g)
    print("   lanes: rays %d, march steps %d, SDF evaluations %d (arg-min at hits %d)" % (g[0], g[1], g[2], g[3]))
