"""Rank 0's share of a sharded frame on one GPU, without the collectives: what the probe chain and
the screen share cost per frame at world sizes 1..8 -- frames kept in flight as ShardedFrame runs
them, and the strictly serial schedule with per-pass times beside it."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from madarch_amd import examples, sharding, _binding as B
hb = B.hip_binding()
R = examples.global_illumination(1920, 1080, Probes=examples.GI_8X8X8_PROBES, Binding=hb)
for world in (1, 2, 4, 8):
    frame = sharding.ShardedFrame(R, 0, world, None)
    out = []
    for overlap in (2, 0):
        R.Set_Option(B.OPT_FRAME_OVERLAP, overlap)
        for it in range(2):
            if it == 1 and overlap == 0:
                R.Set_Option(B.OPT_TIMING, 1); R.Reset_Pass_Times()
            R.Finish(); t0 = time.perf_counter()
            for _ in range(100):
                frame.Render()
            R.Finish(); dt = (time.perf_counter() - t0) / 100
        out.append(dt)
    times = {B.PASS_NAMES[p]: round(R.Pass_Time(p)[0] / max(R.Pass_Time(p)[1], 1), 4) for p in (B.PASS_RADIANCE, B.PASS_IRRADIANCE, B.PASS_SCREEN)}
    R.Set_Option(B.OPT_TIMING, 0)
    print("world %d: %.3f ms/frame in flight (%.0f Mpix/s), %.3f serial; serial passes %s" % (world, out[0] * 1e3, 1920 * 1080 / out[0] / 1e6, out[1] * 1e3, times), flush=True)
