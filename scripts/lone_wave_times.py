"""Kernel durations of a frame so small that every wavefront has its SIMD to itself (16 probes x 64 rays, 64 x 64 pixels):
the latency of one wavefront's whole pixel program -- what the tail of every pass and a rank's probe passes at N = 8 run at."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from madarch_amd import examples, renderers, _binding as B
probes = renderers.Probe_Settings(Radiance_Resolution=8, Irradiance_Resolution=4, Probe_Count=(4, 4), Grid_Dimensions=(4, 2, 2), Grid_Spacing=(2.0, 3.0, 3.0))
R = examples.global_illumination(64, 64, Probes=probes)
R.Set_Option(B.OPT_FRAME_OVERLAP, 0)
for _ in range(50): R.Render()
R.Finish()
R.Set_Option(B.OPT_TIMING, 1)
R.Reset_Pass_Times()
for _ in range(200): R.Render()
R.Finish()
print(os.path.basename(os.environ.get("MADARCH_HIP_LIBRARY", "shipped")), {n: round(R.Pass_Time(p)[0] / max(R.Pass_Time(p)[1], 1) * 1e3, 1) for n, p in (("radiance us", B.PASS_RADIANCE), ("irradiance us", B.PASS_IRRADIANCE), ("screen us", B.PASS_SCREEN))}, flush=True)
