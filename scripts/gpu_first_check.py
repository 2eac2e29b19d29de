"""First on-device check: HIP library vs CPU oracle on small frames (bit-level diff report)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from madarch_amd import examples, renderers, _binding as B
from oracle_engine import oracle_binding

ob = oracle_binding()
hb = B.hip_binding()
print(hb.version(), ob.version())

def diff(name, a, b):
    a = np.asarray(a); b = np.asarray(b)
    same = (a.view(np.uint32) == b.view(np.uint32)) if a.dtype == np.float32 else (a == b)
    nan_mismatch = (np.isnan(a) != np.isnan(b)).sum() if a.dtype == np.float32 else 0
    with np.errstate(all="ignore"):
        rel = np.abs(a.astype(np.float64) - b) / np.maximum(np.abs(b.astype(np.float64)), 1e-5)
    print("  %-12s bit-identical %.6f  max rel %.3g  nan-mismatch %d" % (name, same.mean(), np.nanmax(rel) if rel.size else 0, nan_mismatch))

def run(scene, W, H, mode, frames, **kw):
    print("== %s %dx%d mode %d frames %d %s" % (scene, W, H, mode, frames, kw))
    out = []
    for b in (hb, ob):
        R = examples.SCENES[scene](W, H, Binding=b, **kw)
        R.Set_Option(B.OPT_SCREEN_MODE, mode)
        R.Set_Option(B.OPT_GBUFFER, 1)
        t = time.time()
        for f in range(frames):
            R.Render()
        img = R.Read_Framebuffer()
        dt = time.time() - t
        gb = R.Read_Gbuffer()
        tex = [R.Read_Texture(i) for i in range(4)] if mode == 0 else []
        out.append((img, gb, tex, dt))
    (i1, g1, t1, d1), (i2, g2, t2, d2) = out
    print("  time hip %.3fs oracle %.3fs" % (d1, d2))
    diff("image", i1, i2); diff("gb.index", g1[0], g2[0]); diff("gb.t", g1[1], g2[1]); diff("gb.steps", g1[2], g2[2])
    for n, a, b in zip(("radiance", "irradiance", "visibility", "scattering"), t1, t2):
        diff(n, a, b)

run("global_illumination", 64, 64, 1, 1)
run("global_illumination", 64, 64, 2, 1)
run("global_illumination", 64, 64, 0, 3)
run("simple_scene", 64, 64, 2, 1)
run("simple_scene", 64, 64, 0, 2)
run("light_shafts", 48, 48, 0, 2, Volumetrics=renderers.Volumetrics_Settings(Visibility_Resolution=(24, 24, 24), Scattering_Resolution=(32, 32)))
