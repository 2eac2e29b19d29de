"""MDH_OPT_SCREEN_SPLIT measured: the screen pass of small launches with every tile as one, two or four wavefronts.
Run on the GPU box:  python scripts/split_tiles_experiment.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from madarch_amd import examples, _binding as B
hb = B.hip_binding()
for name, make, w, h in (("global_illumination 8x8x8", lambda w, h: examples.global_illumination(w, h, Probes=examples.GI_8X8X8_PROBES, Binding=hb), 640, 360),
                         ("global_illumination 8x8x8", lambda w, h: examples.global_illumination(w, h, Probes=examples.GI_8X8X8_PROBES, Binding=hb), 256, 256),
                         ("simple_scene full", lambda w, h: examples.simple_scene(w, h, Binding=hb), 640, 360)):
    for limit in (0, 2 * ((w + 7) // 8) * ((h + 7) // 8), 4 * ((w + 7) // 8) * ((h + 7) // 8)):
        R = make(w, h)
        R.Set_Option(B.OPT_FRAME_OVERLAP, 0)
        R.Set_Option(B.OPT_SCREEN_SPLIT, limit)
        for _ in range(5): R.Render()
        R.Finish(); R.Set_Option(B.OPT_TIMING, 1); R.Reset_Pass_Times()
        for _ in range(30): R.Render_Pass(B.PASS_SCREEN)
        R.Finish()
        ms, n = R.Pass_Time(B.PASS_SCREEN)
        print("%-28s %4dx%-4d %5d tiles, limit %6d: screen pass %.4f ms" % (name, w, h, ((w + 7) // 8) * ((h + 7) // 8), limit, ms / n), flush=True)
        R.Destroy()
