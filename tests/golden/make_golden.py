"""Generates the golden vectors under tests/golden/ from the CPU oracle.

    python tests/golden/make_golden.py

The reference (Ada + GLSL) cannot run in this pipeline and holds no numeric fixture
(SURVEY.md sections 4, 8c), so these vectors come from the build's own restatement
(oracle/); they pin it against accidental change and are the data the HIP path is
compared with on the GPU box.  Cases follow SURVEY.md section 8c (i)-(vi)."""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from helpers import SEED, SMALL_PROBES, make, seeded_points, snapshot  # noqa: E402
from oracle_engine import oracle_binding  # noqa: E402

# name -> (scene, W, H, mode, atlas, frames, probes)
FRAME_CASES = {
    "c1_simple_scene_primary": ("simple_scene", 64, 64, 1, 0, 1, None),
    "c2_simple_scene_direct": ("simple_scene", 64, 48, 2, 0, 1, None),
    "c3_global_illumination_rgb8": ("global_illumination", 48, 32, 0, 0, 3, SMALL_PROBES),
    "c3_global_illumination_f32": ("global_illumination", 48, 32, 0, 1, 3, SMALL_PROBES),
    "c4_light_shafts": ("light_shafts", 40, 32, 0, 0, 2, SMALL_PROBES),
    "simple_scene_full": ("simple_scene", 40, 32, 0, 0, 2, SMALL_PROBES),
}


def cf(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def ray_table(orc, R, n, seed):
    rng = np.random.RandomState(seed & 0x7FFFFFFF)
    org = (np.array([0.0, 0.0, -4.0]) + np.array([6.0, 6.0, 9.0]) * rng.rand(n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    hit, idx, steps = (np.zeros(n, np.int32) for _ in range(3))
    t = np.zeros(n, np.float32)
    orc.lib.orc_probe_raycast(R._h, n, cf(org), cf(d), hit.ctypes.data_as(C.c_void_p), idx.ctypes.data_as(C.c_void_p),
                              cf(t), steps.ctypes.data_as(C.c_void_p))
    return {"org": org, "dir": d, "hit": hit, "index": idx, "t": t, "steps": steps}


def main():
    orc = oracle_binding()
    for name, (scene, W, H, mode, atlas, frames, probes) in FRAME_CASES.items():
        R = make(scene, W, H, orc, mode=mode, atlas=atlas, probes=probes)
        out = snapshot(R, frames)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print(name, {k: v.shape for k, v in out.items()})
    # ray tables and closest-primitive tables per scene (iii), (ii)
    for scene in ("simple_scene", "global_illumination"):
        R = make(scene, 8, 8, orc)
        rays = ray_table(orc, R, 256, SEED)
        pts = seeded_points(256, -1.0, (7.0, 7.0, 7.0))
        dist, idx = np.zeros(256, np.float32), np.zeros(256, np.int32)
        orc.lib.orc_probe_closest(R._h, 256, cf(pts), 0, cf(dist), idx.ctypes.data_as(C.c_void_p))
        d_eval, n_eval = R.Eval_Distances_To(pts, [p for p, _ in R.Scene.Prims_Count])
        np.savez_compressed(os.path.join(HERE, "tables_%s.npz" % scene), pts=pts, closest=dist, closest_index=idx,
                            eval_dist=d_eval, eval_normal=n_eval, **{"ray_" + k: v for k, v in rays.items()})
    # partition tables of simple_scene for the three builders (a13)
    tabs = {}
    for method, mname in ((0, "cpu_best"), (1, "cpu_fast"), (2, "gpu_fast")):
        R = make("simple_scene", 8, 8, orc, Partitioning_Method=method)
        tabs[mname] = R.Read_Partitioning()
    np.savez_compressed(os.path.join(HERE, "partition_simple_scene.npz"), **tabs)
    print("done")


if __name__ == "__main__":
    main()
