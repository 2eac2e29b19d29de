import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np
from madarch_amd import examples, _binding as B
hb = B.hip_binding()
for (w, h) in ((640, 360), (1920, 1080)):
    R = examples.simple_scene(w, h, Binding=hb)
    R.Set_Option(B.OPT_GBUFFER, 1)
    R.Render(); R.Finish()
    idx, t, steps = R.Read_Gbuffer()
    steps = np.asarray(steps).reshape(h, w)
    ty, tx = (h + 7) // 8, (w + 7) // 8
    pad = np.zeros((ty * 8, tx * 8), dtype=steps.dtype); pad[:h, :w] = steps
    tiles = pad.reshape(ty, 8, tx, 8).max(axis=(1, 3))
    print("%dx%d: primary steps per pixel mean %.1f max %d; per-tile max: mean %.1f, 99%% %d, max %d; tiles with max > 200: %d of %d" % (w, h, steps.mean(), steps.max(), tiles.mean(), np.percentile(tiles, 99), tiles.max(), (tiles > 200).sum(), tiles.size))
    top = np.argsort(tiles.ravel())[-5:]
    print("   slowest tiles (ty, tx, max steps):", [(int(k // tx), int(k % tx), int(tiles.ravel()[k])) for k in top])
    R.Destroy()
