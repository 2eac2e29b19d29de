"""Host time of one mdh_render call (frames in flight, a frame so small that the GPU never is the limit): what bounds the
frame rate of a rank whose share of the frame is tiny (N = 8 at 1080p).  With --comm the same through a one-rank
communicator (the in-place all-gather's enqueue included)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from madarch_amd import examples, renderers, _binding as B
probes = renderers.Probe_Settings(Radiance_Resolution=8, Irradiance_Resolution=4, Probe_Count=(4, 4), Grid_Dimensions=(4, 2, 2), Grid_Spacing=(2.0, 3.0, 3.0))
R = examples.global_illumination(64, 64, Probes=probes)
if "--comm" in sys.argv:
    R.Comm_Init(R.Comm_Unique_Id(), 0, 1)
for overlap in (2, 0):
    R.Set_Option(B.OPT_FRAME_OVERLAP, overlap)
    for _ in range(200): R.Render()
    R.Finish()
    n = 3000
    t = time.perf_counter()
    for _ in range(n): R.Render()
    host = (time.perf_counter() - t) / n
    R.Finish()
    total = (time.perf_counter() - t) / n
    print("overlap %d%s: host %.1f us per mdh_render, %.1f us per frame with the GPU drained (64x64 pixels, 16 probes)" % (overlap, " + communicator" if "--comm" in sys.argv else "", host * 1e6, total * 1e6), flush=True)
if "--comm" in sys.argv:
    R.Comm_Destroy()
R.Destroy()
