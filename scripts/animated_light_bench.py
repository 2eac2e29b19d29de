"""The global_illumination example as its main loop runs it (main.adb:219-232): the spot light's direction is
set anew before every frame.  Compares with the static frames bench.py times."""
import math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from madarch_amd import examples, _binding as B
from madarch_amd.lights import spot_lights
R = examples.global_illumination(1920, 1080, Probes=examples.GI_8X8X8_PROBES, Binding=B.hip_binding())
for animated in (False, True):
    t_light = 0.0
    for it in range(2):
        R.Finish(); t0 = time.perf_counter()
        for _ in range(60):
            if animated:
                t_light += 0.01
                R.Set_Light(1, spot_lights.Spot_Light, spot_lights.Create((3.5, 5.0, 2.0), (math.cos(t_light), math.sin(t_light), 0.0), 3.1415 / 4.0, (0.9, 0.9, 0.8)))
            R.Render()
        R.Finish(); dt = (time.perf_counter() - t0) / 60
    print("%-28s %.3f ms/frame  %.0f Mpix/s" % ("light set every frame" if animated else "static scene", dt * 1e3, 1920 * 1080 / dt / 1e6))
