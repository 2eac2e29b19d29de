"""Lane utilisation per march-loop type (needs a -DMDH_DIAG build selected with MADARCH_HIP_LIBRARY)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from madarch_amd import examples, _binding as B
hb = B.hip_binding()
R = examples.global_illumination(1920, 1080, Probes=examples.GI_8X8X8_PROBES, Binding=hb)
buf = (C.c_ulonglong * 16)()
names = ["hit rays (primary/reflection)", "soft shadow ctx0", "soft shadow ctx1", "probe visibility ctx0", "probe visibility ctx1"]
for f in range(2): R.Render()
R.Finish(); hb.lib.mdh_diag_read(buf)
for p, pname in ((B.PASS_RADIANCE, "radiance"), (B.PASS_SCREEN, "screen")):
    R.Render_Pass(p); R.Finish(); hb.lib.mdh_diag_read(buf)
    print(pname)
    tot_e = tot_l = 0
    for t, n in enumerate(names):
        e, l = buf[2 * t], buf[2 * t + 1]
        tot_e += e; tot_l += l
        if e: print("   %-32s wave-evals %10d  lanes/eval %5.1f  share of wave-evals %.3f" % (n, e, l / e, 0))
    print("   total wave-evals %d, mean lanes %.1f" % (tot_e, tot_l / max(tot_e, 1)))
