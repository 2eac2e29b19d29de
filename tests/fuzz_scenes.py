"""Random scenes for differential testing (scripts/fuzz_parity.py, tests/test_gpu_fuzz.py): kinds and counts, open or
closed rooms, tilted planes, materials, lights, camera, probe / partition / volumetrics settings, screen mode, atlas
format, then a second act of distance queries, scene edits, partition rebuilds, frames in flight and Swap_Buffers."""
import os

import numpy as np

from helpers import assert_parity, same_bits, snapshot
from madarch_amd import _binding as B, materials, renderers, scenes, windows
from madarch_amd.lights import point_lights, spot_lights
from madarch_amd.primitives import boxes, planes, spheres, triangles

PROBES = [
    renderers.Probe_Settings(Radiance_Resolution=16, Irradiance_Resolution=8, Probe_Count=(6, 6), Grid_Dimensions=(4, 3, 3), Grid_Spacing=(2.0, 3.0, 3.0)),
    renderers.Probe_Settings(Radiance_Resolution=12, Irradiance_Resolution=6, Probe_Count=(15, 5), Grid_Dimensions=(5, 5, 3), Grid_Spacing=(1.6, 1.9, 2.5)),
    renderers.Probe_Settings(Radiance_Resolution=8, Irradiance_Resolution=4, Probe_Count=(8, 8), Grid_Dimensions=(4, 4, 4), Grid_Spacing=(2.0, 2.0, 2.0)),
    renderers.Probe_Settings(Radiance_Resolution=10, Irradiance_Resolution=5, Probe_Count=(2, 1), Grid_Dimensions=(2, 1, 1), Grid_Spacing=(3.0, 3.0, 3.0)),
]


def build(seed, binding):
    rng = np.random.default_rng(seed)
    u = lambda lo, hi, n=None: rng.uniform(lo, hi, n)
    closed = rng.integers(0, 4) > 0
    part_on = bool(rng.integers(0, 2))
    dims = tuple(int(v) for v in rng.integers(2, 7, 3))
    part = scenes.Partitioning_Settings(Enable=part_on, Index_Count=int(rng.integers(3, 12)), Grid_Dimensions=dims,
                                        Grid_Spacing=tuple(float(v) for v in rng.choice([1.0, 1.5, 2.0, 2.5], 3)),
                                        Grid_Offset=tuple(float(v) for v in u(-3.0, -1.0, 3)),
                                        Border_Behavior=int(rng.integers(0, 2)) if hasattr(scenes, "Fallback") else scenes.Clamp)
    maxc = [int(rng.integers(1, 7)), int(rng.integers(1, 9)), int(rng.integers(1, 5)), int(rng.integers(1, 4))]
    lmax = [int(rng.integers(1, 4)), int(rng.integers(1, 4))]
    scene = scenes.Compile([(spheres.Sphere, maxc[0]), (planes.Plane, maxc[1]), (boxes.Box, maxc[2]), (triangles.Triangle, maxc[3])],
                           [(point_lights.Point_Light, lmax[0]), (spot_lights.Spot_Light, lmax[1])], Partitioning=part)
    W, H = int(rng.integers(1, int(os.environ.get("FUZZ_MAX_W", 40)) + 1)), int(rng.integers(1, int(os.environ.get("FUZZ_MAX_H", 28)) + 1))  # (larger for soak runs)
    vol = renderers.No_Volumetrics
    if rng.integers(0, 3) == 0:  # light shafts: froxel visibility + scattering passes
        vol = renderers.Volumetrics_Settings(Visibility_Resolution=tuple(int(v) for v in rng.integers(3, 22, 3)), Visibility_Step_Size=float(rng.choice([0.1, 0.25, 0.4])),
                                             Scattering_Resolution=tuple(int(v) for v in rng.integers(3, 30, 2)), Scattering_Step_Size=float(rng.choice([0.1, 0.3])))
    R = renderers.Create(windows.Open(W, H), scene, Probes=PROBES[int(rng.integers(0, len(PROBES)))], Volumetrics=vol, Binding=binding)
    nmat = int(rng.integers(1, 6))
    for m in range(nmat):
        R.Set_Material(m, materials.Create(tuple(u(0.0, 1.0, 3)), float(rng.choice([0.0, 0.5, 0.9])), float(rng.choice([0.1, 0.4, 0.8]))))
    mat = lambda: int(rng.integers(0, nmat))
    walls = [((0, 1, 0), 1.0), ((0, -1, 0), 7.0), ((1, 0, 0), 1.0), ((-1, 0, 0), 7.0), ((0, 0, 1), 6.0), ((0, 0, -1), 7.0)]
    npl = min(maxc[1], 6) if closed else int(rng.integers(0, min(maxc[1], 3) + 1))
    for n, o in walls[:npl]:
        R.Add_Primitive(planes.Plane, planes.Create(n, o, mat()))
    for _ in range(maxc[1] - npl if rng.integers(0, 2) else 0):  # tilted planes fill the rest
        n = u(-1.0, 1.0, 3); n /= np.linalg.norm(n)
        R.Add_Primitive(planes.Plane, planes.Create(tuple(float(v) for v in n), float(u(0.5, 4.0)), mat()))
    n_spheres = int(rng.integers(0, maxc[0] + 1))
    for _ in range(n_spheres):
        R.Add_Primitive(spheres.Sphere, spheres.Create(tuple(u(0.0, 6.0, 3)), float(u(0.2, 1.3)), mat()))
    for _ in range(int(rng.integers(0, maxc[2] + 1))):
        R.Add_Primitive(boxes.Box, boxes.Create(tuple(u(0.0, 6.0, 3)), tuple(u(0.2, 1.2, 3)), mat()))
    for _ in range(int(rng.integers(0, maxc[3] + 1))):
        a = u(0.0, 6.0, 3)
        R.Add_Primitive(triangles.Triangle, triangles.Create(tuple(a), tuple(a + u(-2.0, 2.0, 3)), tuple(a + u(-2.0, 2.0, 3)), mat()))
    for i in range(int(rng.integers(0, lmax[0] + 1))):
        R.Set_Light(i + 1, point_lights.Point_Light, point_lights.Create(tuple(u(0.0, 6.0, 3)), tuple(u(0.1, 1.0, 3))))
    for i in range(int(rng.integers(0, lmax[1] + 1))):
        d = u(-1.0, 1.0, 3); d /= np.linalg.norm(d)
        R.Set_Light(i + 1, spot_lights.Spot_Light, spot_lights.Create(tuple(u(0.0, 6.0, 3)), tuple(float(v) for v in d), float(u(0.2, 1.2)), tuple(u(0.1, 1.0, 3))))
    R.Set_Camera_Position(tuple(u(-0.5, 6.5, 3)))
    if rng.integers(0, 2):
        a, b = u(-1.0, 1.0), u(-0.6, 0.6)
        ca, sa, cb, sb = np.cos(a), np.sin(a), np.cos(b), np.sin(b)
        ry = np.array([[ca, 0, sa], [0, 1, 0], [-sa, 0, ca]]); rx = np.array([[1, 0, 0], [0, cb, -sb], [0, sb, cb]])
        R.Set_Camera_Orientation((ry @ rx).astype(np.float32).tolist())
    R.Set_Option(B.OPT_SCREEN_MODE, int(rng.choice([0, 0, 0, 1, 2])))
    R.Set_Option(B.OPT_ATLAS_FORMAT, int(rng.integers(0, 2)))
    R.Set_Option(B.OPT_AO_STEPS, int(rng.choice([0, 3, 5])))
    R.Set_Option(B.OPT_GBUFFER, 1)
    if part_on and rng.integers(0, 5) > 0:
        R.Update_Partitioning(int(rng.integers(0, 3)))
    R.Set_Option(B.OPT_FRAME_OVERLAP, int(rng.choice([0, 1, 2])))  # (the schedule must not show)
    R.Set_Option(B.OPT_WINDOW, int(rng.integers(0, 2)))
    # (a generator of its own: the scenes of the seeds run before this option existed stay what they were)
    R.Set_Option(B.OPT_INDIRECT_SPECULAR, int(np.random.default_rng(int(seed) ^ 0x5BEC).choice([2, 2, 0, 1, 3])))
    # (seeds from 6000 on: the optional mip chain of the radiance atlas, where the drawn resolution is a power of two)
    if int(seed) >= 6000 and np.random.default_rng(int(seed) ^ 0x3195).integers(0, 3) == 0 and (R.Probes.Radiance_Resolution & (R.Probes.Radiance_Resolution - 1)) == 0:
        R.Set_Option(B.OPT_RADIANCE_MIPS, 1)
    frames = int(rng.integers(1, 4))
    out = snapshot(R, frames)
    if part_on:
        out["partition"] = R.Read_Partitioning()
    # a second act: distance queries, a scene edit, frames without a read in between, the window's pixels
    kinds = [k for k, on in zip((spheres.Sphere, planes.Plane, boxes.Box, triangles.Triangle), rng.integers(0, 2, 4)) if on] or [planes.Plane]
    out["eval_d"], out["eval_n"] = R.Eval_Distances_To(u(-1.0, 7.0, (int(rng.integers(1, 40)), 3)).astype(np.float32), kinds)
    if rng.integers(0, 2) and n_spheres:
        R.Set_Primitive(spheres.Sphere, int(rng.integers(1, n_spheres + 1)), spheres.Create(tuple(u(0.0, 6.0, 3)), float(u(0.2, 1.3)), mat()))
        if part_on and rng.integers(0, 2):
            R.Update_Partitioning(int(rng.integers(0, 3)))
    if rng.integers(0, 2):
        R.Set_Light(1, point_lights.Point_Light, point_lights.Create(tuple(u(0.0, 6.0, 3)), tuple(u(0.1, 1.0, 3))))
    for f in range(int(rng.integers(1, 5))):
        R.Set_Camera_Position(tuple(u(-0.5, 6.5, 3)))
        R.Render()
        R.Swap_Buffers()
    act2 = snapshot(R, 0)
    for k, v in act2.items():
        out["act2_" + k] = v
    out["window"] = R.Front_Buffer()
    return out



def compare(got, want):
    """the parity bar of the tests: integer buffers, atlases, queries and tables bit for bit, colours within 1e-4"""
    assert_parity(got, want)
    assert_parity({k[5:]: v for k, v in got.items() if k.startswith("act2_")}, {k[5:]: v for k, v in want.items() if k.startswith("act2_")})
    for k in ("partition", "eval_d", "eval_n"):
        if k in want:
            assert same_bits(got[k], want[k]), k
    dw = np.abs(got["window"].astype(np.int16) - want["window"].astype(np.int16))
    assert dw.max() <= 1 and (dw == 0).mean() > 0.99, "window"
