"""examples/ball_game at 1920x1080: ten balls, partition rebuilt and scene edited every frame."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from madarch_amd import examples, _binding as B
G = examples.ball_game(1920, 1080, Probes=examples.GI_8X8X8_PROBES, Binding=B.hip_binding())
for i in range(10):
    G.Throw_Ball(); G.Move_Camera((0.2, 0.05, 0.0))
    for _ in range(3): G.Frame()
G.R.Finish(); t = time.perf_counter()
for _ in range(60): G.Frame()
G.R.Finish(); dt = (time.perf_counter() - t) / 60
G.R.Set_Option(B.OPT_TIMING, 1); G.R.Reset_Pass_Times()
for _ in range(10): G.Frame()
G.R.Finish()
print("ball_game 1080p, 10 balls: %.2f ms/frame, %.0f Mpix/s; passes %s" % (dt * 1e3, 1920 * 1080 / dt / 1e6,
      {B.PASS_NAMES[p]: round(G.R.Pass_Time(p)[0] / max(G.R.Pass_Time(p)[1], 1), 3) for p in range(len(B.PASS_NAMES)) if G.R.Pass_Time(p)[1]}))
