"""Frame rate with the window's pixels delivered to the host (Swap_Buffers / Front_Buffer), GI 1080p 8x8x8:
 a) frames only   b) + Swap_Buffers per frame   c) + Front_Buffer of the frame before, per frame
 d) as c, pixels copied out of the pinned buffer as well.
Usage: swap_buffers_bench.py [frames] [modes] [MDH_OPT_WINDOW]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from madarch_amd import examples

N = int(sys.argv[1]) if len(sys.argv) > 1 else 400
modes = sys.argv[2] if len(sys.argv) > 2 else "abcd"
from madarch_amd import _binding as B
R = examples.global_illumination(1920, 1080, Probes=examples.GI_8X8X8_PROBES)
R.Set_Option(B.OPT_WINDOW, int(sys.argv[3]) if len(sys.argv) > 3 else 0)
for mode in modes:
    for _ in range(30):
        R.Render()
    R.Finish()
    t0 = time.perf_counter()
    for f in range(N):
        R.Render()
        if mode != "a":
            if mode in "cd" and f:
                R.Front_Buffer(copy=(mode == "d"))
            R.Swap_Buffers()
    if mode != "a":
        R.Front_Buffer()
    R.Finish()
    dt = (time.perf_counter() - t0) / N
    print("%s: %.4f ms/frame  %.0f Mpixels/s" % (mode, dt * 1e3, 1920 * 1080 / dt / 1e6), flush=True)
