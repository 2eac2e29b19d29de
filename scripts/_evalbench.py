import os, sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import custom_kinds as ck
from test_custom_kinds import room, EXTRA
from madarch_amd import _binding as B
hb = B.hip_binding()
R = room(hb, False, extra=EXTRA)
pts = np.random.default_rng(1).uniform(-1, 7, (1 << 20, 3)).astype(np.float32)
from madarch_amd.primitives import spheres
for kinds, name in (([ck.Torus], "torus (39 + 191 words)"), ([spheres.Sphere], "3 built-in spheres")):
    R.Eval_Distances_To(pts[:1000], kinds)
    t = time.perf_counter(); R.Eval_Distances_To(pts, kinds); dt = time.perf_counter() - t
    print(name, "%.1f ms for 1M points" % (dt * 1e3))
