"""What a wavefront that has its SIMD to itself spends per march step: the radiance and screen passes of a frame so small that
every wavefront runs alone (16 probes x 64 rays, 64 x 64 pixels), from a -DMDH_PHASES -DMDH_DIAG build (MADARCH_HIP_LIBRARY)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from madarch_amd import examples, renderers, _binding as B
hb = B.hip_binding()
probes = renderers.Probe_Settings(Radiance_Resolution=8, Irradiance_Resolution=4, Probe_Count=(4, 4), Grid_Dimensions=(4, 2, 2), Grid_Spacing=(2.0, 3.0, 3.0))
R = examples.global_illumination(64, 64, Probes=probes, Binding=hb)
R.Set_Option(B.OPT_TIMING, 1)
ph, ev = (C.c_ulonglong * 16)(), (C.c_ulonglong * 16)()
names = {0: "hit march", 1: "hit setup", 2: "first step + lights + BRDF", 3: "soft shadow march", 4: "probe corner setup", 5: "probe visibility march / queue",
         6: "probe weights + atlas taps", 7: "reflection radiance tap", 8: "combine", 11: "whole wave"}
for f in range(3): R.Render()
R.Finish(); hb.lib.mdh_diag_phases(ph); hb.lib.mdh_diag_read(ev)
for p, pname, waves in ((B.PASS_RADIANCE, "radiance", 16), (B.PASS_SCREEN, "screen", 64)):
    R.Reset_Pass_Times()
    R.Render_Pass(p); R.Finish(); hb.lib.mdh_diag_phases(ph); hb.lib.mdh_diag_read(ev)
    ms, n = R.Pass_Time(p)
    evals = sum(ev[2 * t] for t in range(5))
    print("%s: kernel %.1f us; per wavefront %.0f shader cycles (= %.2f GHz if the kernel is one wavefront long), %.1f SDF evaluations" %
          (pname, ms * 1e3 / max(n, 1), ph[11] / waves, ph[11] / waves / (ms * 1e3 / max(n, 1)) / 1e3, evals / waves))
    for k in sorted(names):
        if k != 11 and ph[k]: print("   %-36s %8.0f cycles per wavefront" % (names[k], ph[k] / waves))
    for t, nm in enumerate(("hit rays", "soft shadow ctx0", "soft shadow ctx1", "probe visibility ctx0", "probe visibility ctx1")):
        if ev[2 * t]: print("   evals %-24s %6.1f per wavefront at %.1f lanes" % (nm, ev[2 * t] / waves, ev[2 * t + 1] / ev[2 * t]))
