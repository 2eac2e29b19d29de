"""k_irradiance's duration against the number of probes (one workgroup per probe): what a lone workgroup takes and how
workgroups share a CU.  Run on the GPU box:  python scripts/irr_scaling.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from madarch_amd import examples, renderers, _binding as B
hb = B.hip_binding()
for pc, grid in (((4, 4), (4, 2, 2)), ((8, 8), (4, 4, 4)), ((16, 16), (8, 8, 4)), ((32, 16), (8, 8, 8)), ((32, 32), (16, 8, 8)), ((64, 32), (16, 16, 8))):
    P = renderers.Probe_Settings(Probe_Count=pc, Grid_Dimensions=grid, Grid_Spacing=(0.5, 0.5, 0.5))
    R = examples.global_illumination(64, 64, Probes=P, Binding=hb)
    R.Set_Option(B.OPT_FRAME_OVERLAP, 0)
    for _ in range(3): R.Render()
    R.Finish(); R.Set_Option(B.OPT_TIMING, 1); R.Reset_Pass_Times()
    for _ in range(20): R.Render_Pass(B.PASS_IRRADIANCE)
    R.Finish()
    ms, n = R.Pass_Time(B.PASS_IRRADIANCE)
    print("%5d probes: irradiance pass %.4f ms" % (pc[0] * pc[1], ms / n))
    R.Destroy()
