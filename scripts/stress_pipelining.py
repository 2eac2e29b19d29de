"""Randomised schedule check: the same random sequence of API calls (frames, camera moves, scene edits,
single passes, reads, option flips) on a serial renderer and on a pipelined one must give the same
bits at every read.  Usage: python scripts/stress_pipelining.py [seeds] [operations per seed]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import SMALL_PROBES, make, same_bits  # noqa: E402
from madarch_amd import _binding as B  # noqa: E402
from madarch_amd.primitives import spheres  # noqa: E402

hb = B.hip_binding()
seeds, n_ops = int(sys.argv[1]) if len(sys.argv) > 1 else 6, int(sys.argv[2]) if len(sys.argv) > 2 else 250


def drive(seed, overlap):
    rng = np.random.default_rng(seed)
    scene = ("global_illumination", "light_shafts", "simple_scene")[seed % 3]  # the last one with the space partition
    part = scene == "simple_scene"
    W, H = (int(v) for v in os.environ.get("STRESS_SIZE", "256x144").split("x"))
    R = make(scene, W, H, hb, probes=SMALL_PROBES if os.environ.get("STRESS_PROBES", "small") == "small" else None)
    R.Set_Option(B.OPT_GBUFFER, int(rng.integers(0, 2)))
    R.Set_Option(B.OPT_FRAME_OVERLAP, overlap)
    out = []
    for _ in range(n_ops):
        op = rng.integers(0, 100)
        if op < 60:
            for _ in range(int(rng.integers(1, 6))):
                R.Render()
        elif op < 70:
            R.Set_Camera_Position(tuple(rng.uniform(0.5, 5.0, 3)))
        elif op < 74:
            R.Set_Primitive(spheres.Sphere, 1, spheres.Create(tuple(rng.uniform(1.0, 5.0, 3)), float(rng.uniform(0.4, 1.2)), 3))
            if part and rng.integers(0, 4):  # a rebuilt partition table rides along with the frames in flight
                R.Update_Partitioning(int(rng.integers(0, 3)))
                if rng.integers(0, 4) == 0:
                    out.append(np.array([R.Partition_Warnings()], dtype=np.int32))
        elif op < 76:  # a burst of edits, each committed by a query or a frame: wraps the ring of table buffers
            for _ in range(int(rng.integers(3, 9))):
                R.Set_Primitive(spheres.Sphere, 1, spheres.Create(tuple(rng.uniform(1.0, 5.0, 3)), float(rng.uniform(0.4, 1.2)), 3))
                if part:
                    R.Update_Partitioning(int(rng.integers(0, 3)))
                if rng.integers(0, 2):
                    out.append(R.Eval_Distances_To(rng.uniform(0.0, 6.0, (17, 3)).astype(np.float32), [spheres.Sphere])[0])
                else:
                    R.Render()
        elif op < 82:
            R.Render_Pass(int(rng.choice([B.PASS_RADIANCE, B.PASS_IRRADIANCE, B.PASS_SCREEN])))
        elif op < 88:
            out.append(R.Read_Framebuffer())
        elif op < 92:
            out.append(R.Read_Texture(int(rng.choice([B.TEX_RADIANCE, B.TEX_IRRADIANCE]))))
        elif op < 95:
            t = R.Read_Texture(B.TEX_IRRADIANCE)
            R.Write_Texture(B.TEX_IRRADIANCE, t * np.float32(0.5))
        elif op < 97:
            R.Set_Option(B.OPT_GBUFFER, int(rng.integers(0, 2)))
        elif op < 98:
            R.Finish()
        elif op < 99:
            if part:
                out.append(R.Read_Partitioning())
        else:
            R.Set_Option(B.OPT_SCREEN_MODE, int(rng.integers(0, 3)))
    out.append(R.Read_Framebuffer()); out.append(R.Read_Texture(B.TEX_RADIANCE)); out.append(R.Read_Texture(B.TEX_IRRADIANCE))
    if part:
        out.append(R.Read_Partitioning())
    return out


bad = 0
for seed in range(seeds):
    ref = drive(seed, 0)
    for overlap in (1, 2):
        got = drive(seed, overlap)
        ok = len(ref) == len(got) and all(same_bits(a, b) for a, b in zip(ref, got))
        bad += not ok
        print("seed %d overlap %d: %d reads %s" % (seed, overlap, len(ref), "identical" if ok else "DIFFER"), flush=True)
sys.exit(1 if bad else 0)
