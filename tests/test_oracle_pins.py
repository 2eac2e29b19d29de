"""Pins of the CPU oracle.  The reference holds no numeric fixture for this path (SURVEY.md
section 8c: parity unpinned), so the oracle is pinned here by
  (i)  hand-computed known answers of the closed forms the reference's sources state,
  (ii) an independent numpy float32 restatement of those forms (same operation order =>
       bit-identical), and
  (iii) the two interpretations of the expression trees agreeing: the tree evaluator
       (Madarch.Exprs.Eval) against the closed forms (what To_GLSL emits).
"""
import ctypes as C

import numpy as np
import pytest

from helpers import SEED, seeded_points
from madarch_amd import examples

f32 = np.float32


def cf(a):
    return np.ascontiguousarray(a, dtype=np.float32).ctypes.data_as(C.POINTER(C.c_float))


def orc_sdf(orc, type_, a, b, c, p):
    return orc.lib.orc_sdf(type_, cf(a), cf(b), cf(c), cf(p))


def orc_normal(orc, type_, a, b, c, p):
    out = np.zeros(3, dtype=np.float32)
    orc.lib.orc_sdf_normal(type_, cf(a), cf(b), cf(c), cf(p), cf(out))
    return out


def orc_exprs(orc, type_, a, b, c, p, ada_div=1):
    n = np.zeros(3, dtype=np.float32)
    d = orc.lib.orc_exprs_sdf(type_, cf(a), cf(b), cf(c), cf(p), ada_div, cf(n))
    return d, n


Z3 = np.zeros(3, dtype=np.float32)

# ---- numpy float32 restatement (every operation rounds to binary32, same order) ----------


def np_dot(a, b):
    return f32(f32(f32(a[0] * b[0]) + f32(a[1] * b[1])) + f32(a[2] * b[2]))


def np_length(a):
    return np.sqrt(np_dot(a, a), dtype=np.float32)


def np_sphere(c, r, p):  # madarch-primitives-spheres.ads:13-14
    return f32(np_length((c - p).astype(np.float32)) - f32(r))


def np_plane(n, o, p):  # madarch-primitives-planes.ads:13-14
    return f32(np_dot(n, p) + f32(o))


def np_box(c, s, p):  # madarch-primitives-boxes.adb:7-15
    q = (np.abs((c - p).astype(np.float32)) - s).astype(np.float32)
    outside = np_length(np.maximum(q, f32(0)))
    inside = min(max(q[0], max(q[1], q[2])), f32(0))
    return f32(outside + f32(inside))


def test_known_answers(orc):
    one = np.array([1, 0, 0], dtype=np.float32)
    assert orc_sdf(orc, 0, Z3, [1.0], Z3, [3, 4, 0]) == 4.0            # |(3,4,0)| - 1
    assert orc_sdf(orc, 0, Z3, [1.0], Z3, [0, 0, 0]) == -1.0
    assert orc_sdf(orc, 1, [0, 1, 0], [1.0], Z3, [5, 2, 7]) == 3.0     # y + 1
    assert orc_sdf(orc, 2, Z3, [1, 1, 1], Z3, [2, 0, 0]) == 1.0        # outside a face
    assert orc_sdf(orc, 2, Z3, [1, 1, 1], Z3, [0.5, 0, 0]) == -0.5     # inside
    assert orc_sdf(orc, 2, Z3, [1, 1, 1], Z3, [4, 5, 1]) == 5.0        # edge: |(3,4,0)|
    assert np.array_equal(orc_normal(orc, 0, Z3, [1.0], Z3, [0, 0, 2]), [0, 0, 1])
    assert np.array_equal(orc_normal(orc, 1, [0, 1, 0], [1.0], Z3, [5, 2, 7]), [0, 1, 0])
    assert np.array_equal(orc_normal(orc, 2, Z3, [1, 1, 1], Z3, [2, 0.1, 0.2]), one)
    # box normal near an edge: both axes within 0.002 of the largest (boxes.adb:5,23-26)
    n = orc_normal(orc, 2, Z3, [1, 1, 1], Z3, [1.0, 0.999, 0.0])
    assert np.allclose(n, [2 ** -0.5, 2 ** -0.5, 0], atol=1e-7)
    # triangle in the z = 0 plane: distance of a point above its interior is its height
    tri = ([0, 0, 0], [1, 0, 0], [0, 1, 0])
    assert abs(orc_sdf(orc, 3, *tri, [0.25, 0.25, 2.0]) - 2.0) < 1e-6
    assert abs(orc_sdf(orc, 3, *tri, [-3.0, 0.0, 4.0]) - 5.0) < 1e-6   # nearest is vertex v1


def test_numpy_restatement_is_bit_identical(orc):
    pts = seeded_points(256, -8.0, (8.0, 8.0, 8.0))
    rng = np.random.RandomState(7)
    for p in pts:
        c = rng.uniform(-3, 3, 3).astype(np.float32)
        s = rng.uniform(0.1, 2, 3).astype(np.float32)
        r = f32(rng.uniform(0.1, 2))
        n = rng.normal(size=3).astype(np.float32)
        assert orc_sdf(orc, 0, c, [r], Z3, p) == np_sphere(c, r, p)
        assert orc_sdf(orc, 1, n, [r], Z3, p) == np_plane(n, r, p)
        assert orc_sdf(orc, 2, c, s, Z3, p) == np_box(c, s, p)


def test_tree_evaluator_agrees_with_closed_forms(orc):
    """Exprs.Eval (madarch-exprs.adb:322-716) == the closed forms, bit for bit, on
    Sphere / Plane / Box, distance and normal."""
    pts = seeded_points(192, -6.0, (6.0, 6.0, 6.0), seed=SEED + 1)
    rng = np.random.RandomState(11)
    for p in pts:
        c = rng.uniform(-3, 3, 3).astype(np.float32)
        s = rng.uniform(0.2, 2, 3).astype(np.float32)
        r = f32(rng.uniform(0.2, 2))
        for type_, a, b in ((0, c, [r]), (1, s, [r]), (2, c, s)):
            d, n = orc_exprs(orc, type_, a, b, Z3, p)
            assert d == orc_sdf(orc, type_, a, b, Z3, p)
            assert np.array_equal(n, orc_normal(orc, type_, a, b, Z3, p), equal_nan=True)


def test_values_division_bug_is_reproduced(orc):
    """Madarch.Values."/" on two floats returns L + R (madarch-values.adb:112): the Ada
    evaluator's Triangle distance differs from the GLSL one; with a true division the tree
    evaluator and the closed form agree."""
    tri = ([0, 0, 0], [2, 0, 0], [0, 2, 0])
    p = [1.0, -1.0, 0.5]  # projects onto the middle of edge v1-v2: the quotient matters
    glsl = orc_sdf(orc, 3, *tri, p)
    d_true, _ = orc_exprs(orc, 3, *tri, p, ada_div=0)
    d_ada, _ = orc_exprs(orc, 3, *tri, p, ada_div=1)
    assert d_true == glsl
    assert d_ada != glsl


def test_octahedral_map_round_trip(orc):
    rng = np.random.RandomState(3)
    v = rng.normal(size=(512, 3)).astype(np.float32)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    for d in v:
        e = np.zeros(2, dtype=np.float32)
        orc.lib.orc_oct_encode(cf(d), cf(e))
        assert 0.0 <= e[0] <= 1.0 and 0.0 <= e[1] <= 1.0
        back = np.zeros(3, dtype=np.float32)
        orc.lib.orc_oct_decode(cf(e), cf(back))
        assert np.allclose(back, d, atol=2e-6)
    # the pole and a lower-hemisphere direction (probe_utils.glsl:64-78)
    e = np.zeros(2, dtype=np.float32)
    orc.lib.orc_oct_encode(cf([0, 0, 1]), cf(e))
    assert np.array_equal(e, [0.5, 0.5])


def test_cook_torrance_against_float64(orc):
    """glsl/cook_torrance_brdf.glsl:1-52 restated in float64 numpy."""
    rng = np.random.RandomState(5)
    PI = 3.14159265358
    for _ in range(128):
        N = rng.normal(size=3); N /= np.linalg.norm(N)
        V = rng.normal(size=3); V /= np.linalg.norm(V)
        L = rng.normal(size=3); L /= np.linalg.norm(L)
        if N @ V < 0.05 or N @ L < 0.05:
            continue
        albedo = rng.uniform(0, 1, 3); metallic = rng.uniform(0, 1); rough = rng.uniform(0.05, 1)
        H = (V + L) / np.linalg.norm(V + L)
        NdotV, NdotL = max(N @ V, 0), max(N @ L, 0)
        F0 = 0.04 * (1 - metallic) + albedo * metallic
        a2 = rough ** 4
        NDF = a2 / (PI * (max(N @ H, 0) ** 2 * (a2 - 1) + 1) ** 2)
        k = (rough + 1) ** 2 / 8
        G = (NdotV / (NdotV * (1 - k) + k)) * (NdotL / (NdotL * (1 - k) + k))
        F = F0 + (1 - F0) * (1.001 - max(H @ V, 0)) ** 5
        kD = (1 - F) * (1 - metallic)
        kS = np.minimum(NDF * G * F / max(4 * NdotV * NdotL, 0.001), 1.0)
        okD, okS = np.zeros(3, np.float32), np.zeros(3, np.float32)
        orc.lib.orc_cook_torrance(cf(N), cf(V), cf(L), cf(albedo), C.c_float(metallic), C.c_float(rough), cf(okD), cf(okS))
        assert np.allclose(okD, kD, rtol=2e-5, atol=1e-6)
        assert np.allclose(okS, kS, rtol=2e-5, atol=1e-6)


def test_lights_known_answers(orc):
    # point light: color / (dist^2 * 0.03)  (madarch-lights-point_lights.ads:20-22)
    R = examples.simple_scene(8, 8, Binding=orc, Partitioning_Method=None)
    rad, d, dist = np.zeros(3, np.float32), np.zeros(3, np.float32), C.c_float()
    orc.lib.orc_probe_light(R._h, 0, cf([0, 0, 0]), cf(rad), cf(d), C.byref(dist))
    assert dist.value == 3.0 and np.array_equal(d, [0, 1, 0])
    assert np.allclose(rad, 0.9 / (9 * 0.03), rtol=1e-6)
    # spot light on its axis: min(1/(d^2 0.03), 1.5) * color (madarch-lights-spot_lights.adb:5-24)
    R = examples.global_illumination(8, 8, Binding=orc)
    orc.lib.orc_probe_light(R._h, 0, cf([5.5, 5.0, 2.0]), cf(rad), cf(d), C.byref(dist))
    assert dist.value == 2.0 and np.array_equal(d, [-1, 0, 0])
    assert np.allclose(rad, np.array([0.9, 0.9, 0.8]) * 1.5, rtol=1e-6)
    # outside the cone (angle >= aperture): no light
    orc.lib.orc_probe_light(R._h, 0, cf([3.5, 0.0, 2.0]), cf(rad), cf(d), C.byref(dist))
    assert np.array_equal(rad, [0, 0, 0])


def test_raycast_known_answers(orc):
    """The camera's centre ray of the global_illumination scene runs down +z from (2,2,0)
    to the wall z = 7 (plane 6 of 6 => flat index 20 + 5, madarch-scenes.adb:656-666)."""
    R = examples.global_illumination(8, 8, Binding=orc)
    org = np.array([[2, 2, 0]], np.float32); d = np.array([[0, 0, 1]], np.float32)
    hit, idx, steps = np.zeros(1, np.int32), np.zeros(1, np.int32), np.zeros(1, np.int32)
    t = np.zeros(1, np.float32)
    orc.lib.orc_probe_raycast(R._h, 1, cf(org), cf(d), hit.ctypes.data_as(C.c_void_p), idx.ctypes.data_as(C.c_void_p),
                              cf(t), steps.ctypes.data_as(C.c_void_p))
    assert hit[0] == 1 and idx[0] == 25
    assert 7.0 - 1e-3 - 1e-5 <= t[0] <= 7.0  # stops within epsilon of the surface (raymarching.glsl:29)
    # a ray that leaves through nothing: straight up from above the ceiling plane
    org[0] = [2, 8, 0]; d[0] = [0, 1, 0]
    orc.lib.orc_probe_raycast(R._h, 1, cf(org), cf(d), hit.ctypes.data_as(C.c_void_p), idx.ctypes.data_as(C.c_void_p),
                              cf(t), steps.ctypes.data_as(C.c_void_p))
    assert hit[0] == 1  # starts behind the ceiling plane: negative distance is an immediate hit


def test_partitioning_lookup_agrees_with_full_scan(orc):
    """simple_scene: the grid lookup (scenes.adb:839-1118) returns the full-scan distance at
    points near surfaces, for the three table builders."""
    from madarch_amd import renderers
    pts = seeded_points(400, -0.9, (6.9, 6.9, 6.9), seed=SEED + 2)
    pts[:, 2] = pts[:, 2] * 1.8 - 5.5
    n = len(pts)
    for method in (renderers.CPU_Best, renderers.CPU_Fast, renderers.GPU_Fast):
        R = examples.simple_scene(8, 8, Binding=orc, Partitioning_Method=method)
        # CPU_Best (what examples/simple_scene uses) fits Index_Count = 20; the two fast builders
        # overflow a few cells (the reference only prints a warning there, renderers.adb:593-598)
        assert R.Partition_Warnings() == (0 if method == renderers.CPU_Best else 5)
        table = R.Read_Partitioning()
        cell = np.floor(pts - np.array([-1.5, -1.5, -10.0], np.float32)).astype(int)
        full_cell = table[cell[:, 0] * 200 + cell[:, 1] * 20 + cell[:, 2], :3].sum(axis=1) >= 20
        d_full, i_full = np.zeros(n, np.float32), np.zeros(n, np.int32)
        d_part, i_part = np.zeros(n, np.float32), np.zeros(n, np.int32)
        orc.lib.orc_probe_closest(R._h, n, cf(pts), 0, cf(d_full), i_full.ctypes.data_as(C.c_void_p))
        orc.lib.orc_probe_closest(R._h, n, cf(pts), 1, cf(d_part), i_part.ctypes.data_as(C.c_void_p))
        near = (d_full < 0.4) & ~full_cell  # the table keeps everything within a cell diagonal of the closest
        assert near.sum() > 50
        assert np.array_equal(d_full[near], d_part[near])
        assert np.array_equal(i_full[near], i_part[near])


def test_transcendental_algorithms_against_libm(orc):
    """acos / exp / pow are fixed as explicit fp32 algorithms (oracle/orc_math.h) that the HIP
    kernels repeat operation for operation; here they are held against libm in float64."""
    lib = orc.lib
    for f in (lib.orc_acos, lib.orc_exp, lib.orc_pow):
        f.restype = C.c_float
    lib.orc_acos.argtypes = [C.c_float]
    lib.orc_exp.argtypes = [C.c_float]
    lib.orc_pow.argtypes = [C.c_float, C.c_float]
    rng = np.random.RandomState(9)
    for x in np.concatenate([rng.uniform(-1, 1, 2000), [0.0, 1.0, -1.0, 0.5, 1e-7]]).astype(np.float32):
        assert abs(lib.orc_acos(x) - np.arccos(np.float64(x))) < 5e-7
    for x in np.concatenate([rng.uniform(-12, 2, 2000), [0.0, -0.1, -2.0]]).astype(np.float32):
        want = np.exp(np.float64(x))
        assert abs(lib.orc_exp(x) - want) <= 1.2e-6 * want
    # the tonemap: (c / (c + 1)) ^ 0.4545 on (0, 1)  (draw_screen.glsl:29)
    for x in np.concatenate([rng.uniform(0, 1, 2000), 10.0 ** rng.uniform(-8, 0, 500), [1.0]]).astype(np.float32):
        want = np.float64(x) ** np.float64(np.float32(0.4545))
        assert abs(lib.orc_pow(x, np.float32(0.4545)) - want) <= 2e-6 * want
    assert lib.orc_pow(0.0, 0.4545) == 0.0
    assert np.isnan(lib.orc_pow(-0.5, 0.4545)) and np.isnan(lib.orc_pow(float("nan"), 0.4545))
    assert lib.orc_exp(0.0) == 1.0 and lib.orc_acos(1.0) == 0.0
    # the trigonometric builtins of the user-defined kinds (MDH_X_SIN .. MDH_X_ATAN)
    for name, ref, lo, hi, tol in (("sin", np.sin, -50, 50, 2e-7), ("cos", np.cos, -50, 50, 2e-7), ("atan", np.arctan, -100, 100, 3e-7),
                                   ("asin", np.arcsin, -1, 1, 6e-7), ("tan", np.tan, -1.5, 1.5, 3e-6)):
        f = getattr(lib, "orc_" + name)
        f.restype, f.argtypes = C.c_float, [C.c_float]
        for x in rng.uniform(lo, hi, 3000).astype(np.float32):
            assert abs(f(float(x)) - ref(np.float64(x))) < tol, (name, x)


# ------------------------------------------------------------------------------------------------
# Independent float64 restatements, written from the GLSL, of the three pieces of the path that no
# known answer above covers: soft shadows, sample_irradiance and the irradiance fold.  They share no
# code with the oracle (own SDF of the global_illumination room, own octahedral maps, own bilinear
# filter) and run in float64, so agreement is within fp32 rounding, not to the bit.
GI_PLANES = [((0.0, 1.0, 0.0), 1.0), ((0.0, -1.0, 0.0), 7.0), ((1.0, 0.0, 0.0), 1.0), ((-1.0, 0.0, 0.0), 7.0), ((0.0, 0.0, 1.0), 6.0), ((0.0, 0.0, -1.0), 7.0)]


def gi_sdf64(p):
    """min over the primitives of examples/global_illumination/main.adb:40-74, float64"""
    d = min(np.dot(n, p) + o for n, o in GI_PLANES)
    d = min(d, np.linalg.norm(np.array([3.0, 4.0, 3.0]) - p) - 1.0)                      # spheres.ads:13-14
    q = np.abs(np.array([3.0, 0.0, 4.0]) - p) - np.array([1.5, 1.5, 1.5])                # boxes.adb:7-15
    return min(d, np.linalg.norm(np.maximum(q, 0.0)) + min(max(q[0], q[1], q[2]), 0.0), 20.0)


def softshadows64(o, d, tmin, tmax, k):  # raymarching.glsl:4-23
    res, prev, t = 1.0, 1e20, tmin
    while t < tmax:
        s = gi_sdf64(o + d * t)
        if s < 0.001:
            return 0.0
        y = s * s / (2.0 * prev)
        e = np.sqrt(s * s - y * y)
        den = max(0.0, t - y)
        res = min(res, k * e / den) if den > 0.0 else res
        prev = s
        t += s
    return res


def visibility64(o, d, tmax):  # raymarching.glsl:39-56
    t = 0.0
    while t < tmax:
        s = gi_sdf64(o + d * t)
        if s < 0.001:
            return 0.0
        t += s
    return 1.0


def oct_encode64(v):  # probe_utils.glsl:58-70, 88-92
    p = v[:2] / np.abs(v).sum()
    if v[2] <= 0.0:
        p = (1.0 - np.abs(p[::-1])) * np.where(p >= 0.0, 1.0, -1.0)
    return (p + 1.0) * 0.5


def oct_decode64(e):  # probe_utils.glsl:72-86
    e = e * 2.0 - 1.0
    v = np.array([e[0], e[1], 1.0 - abs(e[0]) - abs(e[1])])
    if v[2] < 0.0:
        v[:2] = (1.0 - np.abs(v[1::-1])) * np.where(v[:2] >= 0.0, 1.0, -1.0)
    return v / np.linalg.norm(v)


def bilinear64(img, cx, cy):
    """GL_LINEAR with GL_MIRRORED_REPEAT on an (H, W, 3) image (render_passes.adb:111-114)"""
    H, W = img.shape[:2]
    px, py = cx * W - 0.5, cy * H - 0.5
    x0, y0 = int(np.floor(px)), int(np.floor(py))
    fx, fy = px - x0, py - y0

    def mir(i, n):
        m = i % (2 * n)
        return 2 * n - 1 - m if m >= n else m
    t = lambda x, y: img[mir(y, H), mir(x, W)].astype(np.float64)  # noqa: E731
    return (t(x0, y0) * (1 - fx) * (1 - fy) + t(x0 + 1, y0) * fx * (1 - fy)) + t(x0, y0 + 1) * (1 - fx) * fy + t(x0 + 1, y0 + 1) * fx * fy


def _gi_renderer(orc, atlas=1):
    from helpers import SMALL_PROBES
    from madarch_amd import _binding as B
    R = examples.global_illumination(8, 8, Probes=SMALL_PROBES, Binding=orc)
    R.Set_Option(B.OPT_ATLAS_FORMAT, atlas)
    return R, SMALL_PROBES


@pytest.mark.parametrize("k", [64.0, 2.0])
def test_softshadows_against_float64(orc, k):
    """raymarching.glsl:4-23 from seeded points of the room towards the light and towards random directions, with
    the renderer's k = 64 (lighting.glsl:29) and a wide penumbra (k = 2)."""
    R, _ = _gi_renderer(orc)
    rng = np.random.RandomState(11)
    pts = seeded_points(160, (-0.5, -0.5, -5.0), (6.5, 6.5, 6.5), seed=SEED + 7).astype(np.float64)
    light = np.array([3.5, 5.0, 2.0])
    rows = []
    for i, p in enumerate(pts):
        if gi_sdf64(p) < 0.3:
            continue
        if i % 2:
            d = light - p
            tmax = np.linalg.norm(d)
            d = d / tmax
        else:
            d = rng.normal(size=3)
            d /= np.linalg.norm(d)
            tmax = rng.uniform(1.0, 6.0)
        rows.append((p, d, tmax, softshadows64(p, d, 0.0, tmax, k)))
    n = len(rows)
    assert n > 80
    out = np.zeros(n, np.float32)
    org = np.array([r[0] for r in rows], np.float32)
    dirs = np.array([r[1] for r in rows], np.float32)
    tmax = np.array([r[2] for r in rows], np.float32)
    orc.lib.orc_probe_softshadow(R._h, n, cf(org), cf(dirs), cf(tmax), C.c_float(k), cf(out))
    want = np.array([r[3] for r in rows])
    # a ray that grazes a surface can end blocked in one precision and graze past in the other: those aside,
    # the penumbra value agrees to fp32 accuracy of a ~20-step march
    agree = np.isclose(out, want, rtol=2e-3, atol=2e-4)
    assert agree.mean() > 0.97
    assert (want == 0.0).sum() > 5 and (want == 1.0).sum() > 5
    if k < 64.0:
        assert ((want > 0.0) & (want < 1.0)).sum() > 20


def test_sample_irradiance_against_float64(orc):
    """render_probes.glsl:6-69 on a random fp32 irradiance atlas: cage probes, visibility rays, the crushed and
    trilinear weights, clamped bilinear taps, the sqrt / square."""
    from madarch_amd import _binding as B
    R, P = _gi_renderer(orc)
    rng = np.random.RandomState(13)
    W, H = P.Probe_Count[0] * P.Irradiance_Resolution, P.Probe_Count[1] * P.Irradiance_Resolution
    atlas = rng.uniform(0.0, 1.0, size=(H, W, 3)).astype(np.float32)
    R.Write_Texture(B.TEX_IRRADIANCE, atlas)
    sp, dims, ires, pc = np.array(P.Grid_Spacing, np.float64), np.array(P.Grid_Dimensions), P.Irradiance_Resolution, np.array(P.Probe_Count)
    pts = seeded_points(60, (0.2, 0.2, 0.2), (6.0, 5.5, 5.5), seed=SEED + 9).astype(np.float64)
    pos, nrm, want = [], [], []
    for p in pts:
        if gi_sdf64(p) < 0.4:
            continue
        n = rng.normal(size=3)
        n /= np.linalg.norm(n)
        g = np.floor(p / sp).astype(int)
        alpha = p / sp - g
        acc, wsum = np.zeros(3), 0.0
        for i in range(8):
            off = np.array([i & 1, (i >> 1) & 1, (i >> 2) & 1])
            c = np.clip(g + off, 0, dims - 1)
            h = c * sp - p
            dist = np.linalg.norm(h)
            dp = h / dist
            w = ((dp @ n + 1.0) * 0.5) ** 2 + 0.2
            w *= visibility64(p + n * 0.05 * 5.0, dp, dist - 0.05 * 5.0)
            if w < 0.2:
                w *= w * w / 0.04
            tri = np.where(off == 1, alpha, 1.0 - alpha)
            w *= tri.prod()
            pid = c[2] * dims[0] * dims[1] + c[1] * dims[0] + c[0]
            base = np.array([pid % pc[0], pid // pc[0]]) / pc
            rid = np.clip(oct_encode64(n), 0.5 / ires, 1.0 - 0.5 / ires)
            tex = bilinear64(atlas, *(base + rid / pc))
            acc += np.sqrt(tex) * w
            wsum += w
        pos.append(p); nrm.append(n); want.append((acc / wsum) ** 2 if wsum else np.zeros(3))
    n = len(pos)
    assert n > 30
    out = np.zeros((n, 3), np.float32)
    orc.lib.orc_probe_sample_irradiance(R._h, n, cf(np.array(pos, np.float32)), cf(np.array(nrm, np.float32)), cf(out))
    ok = np.isclose(out, np.array(want), rtol=3e-4, atol=1e-5).all(axis=1)
    assert ok.mean() > 0.95  # (a visibility ray that grazes the sphere or the box may flip between precisions)


def test_irradiance_fold_against_float64(orc):
    """update_probe_irradiance.glsl:8-43 on a random fp32 radiance atlas: every texel of a few probes, with the
    corner-sample bleed into the neighbouring tiles."""
    from madarch_amd import _binding as B
    R, P = _gi_renderer(orc)
    rng = np.random.RandomState(17)
    rres, ires, pc = P.Radiance_Resolution, P.Irradiance_Resolution, np.array(P.Probe_Count)
    rad = rng.uniform(0.0, 1.0, size=(pc[1] * rres, pc[0] * rres, 3)).astype(np.float32)
    R.Write_Texture(B.TEX_RADIANCE, rad)
    R.Render_Pass(B.PASS_IRRADIANCE)
    got = R.Read_Texture(B.TEX_IRRADIANCE)
    step = 1.0 / pc / rres
    for probe in (0, 7, 20, 35):  # a corner tile, an edge tile, inner tiles
        ty, tx = divmod(probe, pc[0])
        base = np.array([tx, ty]) / pc
        taps = []
        for y in range(rres):
            for x in range(rres):
                c = np.clip(base + np.array([x, y]) * step, step, 1.0 - step)
                taps.append((bilinear64(rad, c[0], c[1]), oct_decode64((c * pc) % 1.0)))
        for ky in range(ires):
            for kx in range(ires):
                i, j = tx * ires + kx, ty * ires + ky
                nc = (np.array([(2 * i + 1) / (pc[0] * ires) - 1.0, (2 * j + 1) / (pc[1] * ires) - 1.0]) + 1.0) * 0.5
                irr_dir = oct_decode64((nc * pc) % 1.0)
                acc, wsum = np.zeros(3), 0.0
                for r_, d_ in taps:
                    w = max(irr_dir @ d_, 0.0)
                    acc += r_ * w
                    wsum += w
                assert np.allclose(got[j, i], acc / wsum, rtol=2e-4, atol=1e-6), (probe, kx, ky)


def test_byte_over_255_by_fma_is_the_correctly_rounded_quotient():
    """The kernels decode an RGB8 texel without a division or a table (mdh_device.h: u8_unorm): 1/255 = c_hi + c_lo with c_hi
    its nearest float, and fma (k, c_hi, fl (k c_lo)) must be fl (k / 255) -- what the oracle's division gives -- for every
    byte k.  Checked here in exact rational arithmetic, each fp32 operation rounded once."""
    from fractions import Fraction

    def fl(x):  # an exact rational to the nearest float32, ties to even
        c = np.float32(float(x))
        best = None
        for cand in (np.nextafter(c, np.float32(-np.inf)), c, np.nextafter(c, np.float32(np.inf))):
            key = (abs(Fraction(float(cand)) - x), int(np.float32(cand).view(np.uint32)) & 1)
            if best is None or key < best[0]:
                best = (key, np.float32(cand))
        return best[1]

    c_hi, c_lo = np.float32(float.fromhex("0x1.010102p-8")), np.float32(float.fromhex("-0x1.fdfdfep-33"))
    assert c_hi == fl(Fraction(1, 255)) and c_lo == fl(Fraction(1, 255) - Fraction(float(c_hi)))
    for k in range(256):
        want = np.float32(k) / np.float32(255.0)
        assert want == fl(Fraction(k, 255))
        low = fl(Fraction(k) * Fraction(float(c_lo)))
        assert fl(Fraction(k) * Fraction(float(c_hi)) + Fraction(float(low))) == want, k
