"""Rank 0's share of a sharded frame on one GPU, without the collectives: what the probe chain and
the screen share cost per frame at world sizes 1..8 (serial schedule, as ShardedFrame runs it)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from madarch_amd import examples, _binding as B
hb = B.hip_binding()
R = examples.global_illumination(1920, 1080, Probes=examples.GI_8X8X8_PROBES, Binding=hb)
for world in (1, 2, 4, 8):
    R.Set_Option(B.OPT_WORLD, world)
    R.Set_Option(B.OPT_RANK, 0)
    for it in range(2):
        if it == 1:
            R.Set_Option(B.OPT_TIMING, 1); R.Reset_Pass_Times()
        R.Finish(); t0 = time.perf_counter()
        for _ in range(30):
            for p in (B.PASS_RADIANCE, B.PASS_IRRADIANCE, B.PASS_SCREEN):
                R.Render_Pass(p)
        R.Finish(); dt = (time.perf_counter() - t0) / 30
    times = {B.PASS_NAMES[p]: round(R.Pass_Time(p)[0] / max(R.Pass_Time(p)[1], 1), 4) for p in (B.PASS_RADIANCE, B.PASS_IRRADIANCE, B.PASS_SCREEN)}
    R.Set_Option(B.OPT_TIMING, 0)
    print("world %d: %.3f ms/frame without collectives -> %.0f Mpix/s; passes %s" % (world, dt * 1e3, 1920 * 1080 / dt / 1e6, times), flush=True)
