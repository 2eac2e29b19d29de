"""The radiance pass of rank 0's probe slice at world sizes 4 and 8 (BASELINE config 3), serial schedule, by itself."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from madarch_amd import examples, _binding as B
R = examples.global_illumination(1920, 1080, Probes=examples.GI_8X8X8_PROBES)
R.Set_Option(B.OPT_FRAME_OVERLAP, 0)
if os.environ.get("RAD_ORDER") is not None: R.Set_Option(B.OPT_RADIANCE_ORDER, int(os.environ["RAD_ORDER"]))
for _ in range(5): R.Render()
out = {}
for world in (8, 4, 2, 1):
    R.Set_Option(B.OPT_WORLD, world); R.Set_Option(B.OPT_RANK, 0)
    for _ in range(20): R.Render_Pass(B.PASS_RADIANCE)
    R.Finish(); R.Set_Option(B.OPT_TIMING, 1); R.Reset_Pass_Times()
    for _ in range(100): R.Render_Pass(B.PASS_RADIANCE)
    R.Finish()
    ms, n = R.Pass_Time(B.PASS_RADIANCE)
    out[world] = round(ms / n * 1e3, 1)
    R.Set_Option(B.OPT_TIMING, 0)
print("rays sorted %s: radiance pass of rank 0's slice, us: %s" % (R.Get_Option(B.OPT_RADIANCE_ORDER), out), flush=True)
