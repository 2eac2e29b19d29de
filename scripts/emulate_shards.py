"""Rank 0's SHARE of a sharded frame on one GPU, without the collectives: what its probe slice and its tiles cost per
frame at world sizes 1..8 -- frames kept in flight as ShardedFrame runs them, and the strictly serial schedule with
per-pass times beside it.  This is not a multi-GPU measurement (no exchange runs, no other rank exists): it bounds
what strong scaling can reach, since a real rank adds one all-gather to the probe chain of every frame.
Usage: python scripts/emulate_shards.py [width height [out.json]]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from madarch_amd import examples, sharding, _binding as B
W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)
hb = B.hip_binding()
R = examples.global_illumination(W, H, Probes=examples.GI_8X8X8_PROBES, Binding=hb)
if os.environ.get("MADARCH_SPLIT") is not None:  # (experiments: MDH_OPT_SCREEN_SPLIT's limit)
    R.Set_Option(B.OPT_SCREEN_SPLIT, int(os.environ["MADARCH_SPLIT"]))
rows = []
for world in (1, 2, 4, 8):
    frame = sharding.ShardedFrame(R, 0, world, None)
    out = []
    n = 100 if W * H <= 1920 * 1080 else 30
    for overlap in (2, 0):
        R.Set_Option(B.OPT_FRAME_OVERLAP, overlap)
        for it in range(2):
            if it == 1 and overlap == 0:
                R.Set_Option(B.OPT_TIMING, 1); R.Reset_Pass_Times()
            R.Finish(); t0 = time.perf_counter()
            for _ in range(n):
                frame.Render()
            R.Finish(); dt = (time.perf_counter() - t0) / n
        out.append(dt)
    times = {B.PASS_NAMES[p]: round(R.Pass_Time(p)[0] / max(R.Pass_Time(p)[1], 1), 4) for p in (B.PASS_RADIANCE, B.PASS_IRRADIANCE, B.PASS_SCREEN)}
    R.Set_Option(B.OPT_TIMING, 0)
    rows.append({"world": world, "rank0_ms_in_flight": round(out[0] * 1e3, 4), "rank0_ms_serial": round(out[1] * 1e3, 4), "serial_pass_ms": times})
    print("world %d: rank 0's share %.3f ms/frame in flight, %.3f serial; serial passes %s" % (world, out[0] * 1e3, out[1] * 1e3, times), flush=True)
if len(sys.argv) > 3:
    json.dump({"what": "rank 0's share of a %dx%d global_illumination frame (DDGI 8x8x8) on ONE MI355X, no collectives, no other ranks: an upper bound on strong scaling, not a multi-GPU measurement" % (W, H),
               "rows": rows}, open(sys.argv[3], "w"), indent=1)
