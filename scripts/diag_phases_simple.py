"""Wall cycles per region of the pixel program for simple_scene's screen pass (mode 2, space partition); needs a
-DMDH_PHASES build selected with MADARCH_HIP_LIBRARY."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from madarch_amd import examples, _binding as B
hb = B.hip_binding()
R = examples.simple_scene(1920, 1080, Binding=hb)
R.Set_Option(B.OPT_SCREEN_MODE, 2)
buf = (C.c_ulonglong * 16)()
names = {0: "hit march (primary)", 1: "hit setup (sdf_info, primitive_info)", 2: "first step + light sampling + BRDF",
         3: "soft shadow march", 8: "combine (AO taps)", 11: "whole wave"}
for f in range(3): R.Render()
R.Finish(); hb.lib.mdh_diag_phases(buf)
R.Render_Pass(B.PASS_SCREEN); R.Finish(); hb.lib.mdh_diag_phases(buf)
tot = buf[11]
waves = 1920 * 1080 / 64
print("screen: wave-cycles total %.1fM = %.0f (100 MHz ticks?) per wave" % (tot / 1e6, tot / waves))
acc = 0
for k in sorted(names):
    if k == 11: continue
    acc += buf[k]
    print("   %-40s %6.1f %%   %8.0f per wave" % (names[k], 100.0 * buf[k] / tot, buf[k] / waves))
print("   %-40s %6.1f %%" % ("other (staging, prologue, epilogue, stamps)", 100.0 * (tot - acc) / tot))
