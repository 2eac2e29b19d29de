"""Wall cycles per region of the pixel program of a bench workload (needs a -DMDH_PHASES build selected with
MADARCH_HIP_LIBRARY):  python scripts/diag_phases.py [workload]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from madarch_amd import _binding as B
hb = B.hip_binding()
workload = sys.argv[1] if len(sys.argv) > 1 else "global_illumination_1080p_ddgi8x8x8"
R = bench.make_renderer(workload, hb)
print(workload)
buf = (C.c_ulonglong * 16)()
names = {0: "hit march (primary/reflection)", 1: "hit setup (sdf_info, primitive_info)", 2: "first step + light sampling + BRDF",
         3: "soft shadow march", 4: "probe corner setup", 5: "probe visibility march / queue", 6: "probe weights + atlas taps",
         7: "reflection radiance tap", 8: "combine (indirect lighting, AO)", 11: "whole wave"}
for f in range(3): R.Render()
R.Finish(); hb.lib.mdh_diag_phases(buf)
for p, pname in ((B.PASS_RADIANCE, "radiance"), (B.PASS_SCREEN, "screen")):
    R.Render_Pass(p); R.Finish(); hb.lib.mdh_diag_phases(buf)
    tot = buf[11]
    print(pname, "wave-cycles total %.1fM" % (tot / 1e6))
    acc = 0
    for k in sorted(names):
        if k == 11: continue
        acc += buf[k]
        print("   %-40s %6.1f %%" % (names[k], 100.0 * buf[k] / tot))
    print("   %-40s %6.1f %%" % ("other (prologue, epilogue, stamps)", 100.0 * (tot - acc) / tot))
