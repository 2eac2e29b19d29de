"""More independent checks of the CPU oracle (it can never be pinned: the reference holds no numeric fixture, SURVEY.md
section 8c).  Everything here is a float64 numpy restatement written from the reference's GLSL / Ada text; it shares
no code with oracle/ (own room geometry with arg-min and normals, own marches, own octahedral maps, own mirrored
bilinear filter, own light, own BRDF) and is compared with the oracle's functions through its probe exports:

  * the volumetric passes: henvey_greenstein_phase and sample_lights (compute_frustrum_visibility.glsl:8-42), one
    scattering texel (accumulate_scattering.glsl:9-48), the 3 x 3 depth-aware pick (volumetrics.glsl:34-54);
  * the spot light's cone (madarch-lights-spot_lights.adb:5-24);
  * the three indirect-specular bodies: the best-probe choice of mode 2 (render_probes.glsl:138-209), modes 1 and 3
    (:71-136, :211-244);
  * the acceptance order of the CPU_Best partition builder (madarch-renderers.adb:609-755);
  * the tone map (draw_screen.glsl:29).

Agreement is to fp32 rounding of the oracle's arithmetic (rays that graze a surface may end differently in the two
precisions: those tests ask for a large majority, like the ones of test_oracle_pins.py)."""
import ctypes as C

import numpy as np
import pytest

from helpers import SEED, SMALL_PROBES, SMALL_VOL, seeded_points
from madarch_amd import _binding as B
from madarch_amd import examples, renderers
from test_oracle_pins import GI_PLANES, bilinear64, cf, gi_sdf64, oct_encode64, softshadows64, visibility64

PI = 3.14159265358   # maths.glsl:1
TAU = 0.1            # volumetrics.glsl:12
MSS = 0.05           # raymarching.glsl:1
# materials of examples/global_illumination/main.adb:40-58: albedo, metallic, roughness
GI_MATS = [((0.0, 0.0, 0.0), 0.0, 0.6), ((1.0, 0.0, 0.0), 0.0, 0.6), ((0.0, 0.0, 1.0), 0.0, 0.6), ((0.1, 0.1, 0.1), 0.9, 0.1), ((0.0, 1.0, 0.0), 0.8, 0.3)]
GI_PLANE_MATS = [0, 0, 1, 2, 0, 0]
SPHERE_C, BOX_C, BOX_S = np.array([3.0, 4.0, 3.0]), np.array([3.0, 0.0, 4.0]), np.array([1.5, 1.5, 1.5])


def ip(a):
    return np.ascontiguousarray(a, dtype=np.int32).ctypes.data_as(C.POINTER(C.c_int32))


def gi_info64(p):
    """closest_primitive_info + primitive_info (scenes.adb:631-729) of the room: normal and material of the arg-min"""
    best, nrm, mat = 20.0, np.zeros(3), 0
    d = np.linalg.norm(SPHERE_C - p) - 1.0
    if d < best:
        best, nrm, mat = d, (p - SPHERE_C) / np.linalg.norm(p - SPHERE_C), 3
    for (n, o), m in zip(GI_PLANES, GI_PLANE_MATS):
        d = np.dot(n, p) + o
        if d < best:
            best, nrm, mat = d, np.array(n), m
    q = np.abs(BOX_C - p) - BOX_S
    d = np.linalg.norm(np.maximum(q, 0.0)) + min(q.max(), 0.0)
    if d < best:  # boxes.adb:17-41
        dd = (p - BOX_C) / BOX_S
        r = np.abs(dd)
        n = np.array([float(r[k] > r[(k + 1) % 3] - 0.002) * float(r[k] > r[(k + 2) % 3] - 0.002) * np.sign(dd[k]) for k in range(3)])
        best, nrm, mat = d, n / np.linalg.norm(n), 4
    return nrm, mat


def raycast64(o, d, tmax=20.0):  # raymarching.glsl:25-51
    t = 0.0
    while t < tmax:
        s = gi_sdf64(o + d * t)
        if s < 0.001:
            return o + d * t
        t += s
    return None


def spot64(pos):  # spot_lights.adb:5-24 wrapped as scenes.adb:497-549: radiance, dir, dist
    lp, ld, ap, col = np.array([3.5, 5.0, 2.0]), np.array([1.0, 0.0, 0.0]), np.float64(np.float32(3.1415 / 4.0)), np.array([0.9, 0.9, 0.8])
    v = lp - pos
    dist = np.linalg.norm(v)
    v = v / dist
    theta = np.arccos(max(np.dot(-v, ld), 0.0))
    ratio = min(max(theta / ap, 0.0), 1.0)
    return col * min(1.0 / (dist * dist * 0.03), 1.5) * (1.0 - ratio ** 8), v, dist


def cook_torrance64(N, V, L, albedo, metallic, rough):  # cook_torrance_brdf.glsl:1-52
    H = (V + L) / np.linalg.norm(V + L)
    NdotV, NdotL = max(N @ V, 0.0), max(N @ L, 0.0)
    F0 = 0.04 * (1 - metallic) + np.array(albedo) * metallic
    a2 = rough ** 4
    NDF = a2 / (PI * (max(N @ H, 0.0) ** 2 * (a2 - 1) + 1) ** 2)
    k = (rough + 1) ** 2 / 8
    G = (NdotV / (NdotV * (1 - k) + k)) * (NdotL / (NdotL * (1 - k) + k))
    F = F0 + (1 - F0) * (1.001 - max(H @ V, 0.0)) ** 5
    return (1 - F) * (1 - metallic), np.minimum(NDF * G * F / max(4 * NdotV * NdotL, 0.001), 1.0)


def direct64(pos, N, dir_, albedo, metallic, rough):  # lighting.glsl:1-40 (screen shader: specular kept)
    rad, L, dist = spot64(pos)
    NdotL = max(N @ L, 0.0)
    kD, kS = cook_torrance64(N, -dir_, L, albedo, metallic, rough)
    sh = softshadows64(pos + N * MSS * 5.0, L, 0.0, dist, 64.0) if NdotL > 0.001 else 0.0
    return (kD * np.array(albedo) / PI + kS) * rad * NdotL * sh


def hg64(a, b):  # volumetrics.glsl:21-30
    return (1.0 - TAU * TAU) / (4.0 * PI * (1.0 + TAU * TAU - 2.0 * TAU * np.dot(a, b)) ** 1.5)


# -------------------------------------------------------------------------------------- the spot light's cone
def test_spot_light_cone_against_float64(orc):
    """madarch-lights-spot_lights.adb:5-24: attenuation capped at 1.5, the acos / aperture ratio to the eighth power."""
    R = examples.global_illumination(8, 8, Binding=orc)
    rad, d, dist = np.zeros(3, np.float32), np.zeros(3, np.float32), C.c_float()
    pts = seeded_points(300, (3.0, 1.5, -1.5), (7.0, 7.0, 5.5), seed=SEED + 21).astype(np.float64)  # in front of and beside the light
    inside = 0
    for p in pts:
        want, wd, wdist = spot64(p)
        orc.lib.orc_probe_light(R._h, 0, cf(p), cf(rad), cf(d), C.byref(dist))
        assert np.allclose(d, wd, atol=2e-6) and abs(dist.value - wdist) < 1e-5
        assert np.allclose(rad, want, rtol=3e-4, atol=2e-6), (p, rad, want)  # (ratio ** 8 amplifies the fp32 acos eightfold)
        inside += want.max() > 0.0
    assert 40 < inside < 260  # both sides of the cone are sampled


# --------------------------------------------------------------------------------------------- the tone map
def test_tone_map_against_float64(orc):
    """draw_screen.glsl:29: pow (c / (c + 1), 0.4545): the oracle's fp32 division and pow_ against float64 over nine
    decades of colour, and a rendered frame's sanity under it (range, monotone in the light's power)."""
    orc.lib.orc_pow.restype, orc.lib.orc_pow.argtypes = C.c_float, [C.c_float, C.c_float]
    rng = np.random.RandomState(3)
    for c in np.concatenate([10.0 ** rng.uniform(-6, 3, 3000), [0.0, 1.0]]):
        c32 = np.float32(c)
        q = np.float32(c32 / np.float32(c32 + np.float32(1.0)))
        got = orc.lib.orc_pow(q, np.float32(0.4545))
        want = (np.float64(c32) / (np.float64(c32) + 1.0)) ** np.float64(np.float32(0.4545))
        assert abs(got - want) <= 3e-6 * want + 1e-30
    # and a frame applies exactly that map: simple_scene's direct light in mode 2 (direct * ao, tonemapped) stays within
    # [0, 1) and is monotone in the light's power
    from madarch_amd.lights import point_lights
    outs = []
    for power in (0.3, 0.9):
        R = examples.simple_scene(24, 16, Binding=orc)
        R.Set_Option(B.OPT_SCREEN_MODE, 2)
        R.Set_Light(1, point_lights.Point_Light, point_lights.Create((0.0, 3.0, 0.0), (power, power, power)))
        R.Render()
        outs.append(R.Read_Framebuffer())
    assert (outs[0] >= 0).all() and (outs[1] < 1.0).all() and (outs[1] >= outs[0]).all() and (outs[1] > outs[0]).mean() > 0.5


# ---------------------------------------------------------------------------------------- the volumetric passes
def _shafts(orc):
    R = examples.light_shafts(40, 24, Binding=orc, Volumetrics=SMALL_VOL)
    R.Render()
    return R


def point_light64(pos):  # point_lights.ads:20-22 wrapped as scenes.adb:497-549 (light_shafts/main.adb:59)
    v = np.array([5.0, 3.0, 6.0]) - pos
    dist = np.linalg.norm(v)
    return np.array([0.9, 0.9, 0.9]) / (dist * dist * 0.03), v / dist, dist


def camera_ray64(u, v):  # draw_screen.glsl:20-24 with the examples' camera: position (2, 2, 0), identity orientation
    f = np.array([u, v, 0.0])
    d = f - np.array([0.0, 0.0, -1.5])
    return f + np.array([2.0, 2.0, 0.0]), d / np.linalg.norm(d)


def test_froxel_texels_against_float64(orc):
    """compute_frustrum_visibility.glsl:8-42: a froxel's in-scattered light = sum over lights of exp (-Ld tau) *
    raycast_visibility * radiance * tau * HG (L, dir), at the camera ray through (x, fract height) advanced by depth *
    step.  (The light_shafts room is the global_illumination room with another light.)"""
    R = _shafts(orc)
    vis = R.Read_Texture(B.TEX_VISIBILITY)  # (vh * vz, vw, 3), row 0 = normalised y of 0
    vw, vh, vz = SMALL_VOL.Visibility_Resolution
    step = float(np.float32(SMALL_VOL.Visibility_Step_Size))
    rng = np.random.RandomState(23)
    ok = total = lit = 0
    for _ in range(400):
        X, Y = rng.randint(vw), rng.randint(vh * vz)
        px, py = -1.0 + 2.0 * (X + 0.5) / vw, -1.0 + 2.0 * (Y + 0.5) / (vh * vz)
        th = (py + 1.0) * 0.5 * vz
        depth = np.floor(th)
        o, d = camera_ray64(px, (th - depth) * 2.0 - 1.0)
        p = o + d * depth * step
        if abs(gi_sdf64(p)) < 0.02:
            continue  # (a sample point on a surface: the visibility ray is blocked at once in one precision only)
        rad, L, Ld = point_light64(p)
        want = np.exp(-Ld * TAU) * visibility64(p, L, Ld) * rad * TAU * hg64(L, d)
        total += 1
        lit += want.max() > 0
        ok += np.allclose(vis[Y, X], want, rtol=3e-4, atol=1e-7)
    assert total > 300 and lit > 100 and ok > 0.97 * total, (ok, total, lit)


def test_scattering_texels_against_float64(orc):
    """accumulate_scattering.glsl:9-48: len = min (|hit - origin|, Rz * step) (a miss keeps the far point), then
    L = step_s * sum over f = 0, step_s, ... < len of bilinear (froxels, (nx, (ny + floor (f / step_v)) / Rz)) * exp (-f tau),
    the froxel texture filtered GL_LINEAR with mirrored repeat; the texel stores (L, len)."""
    R = _shafts(orc)
    vis = R.Read_Texture(B.TEX_VISIBILITY).astype(np.float64)
    scat = R.Read_Texture(B.TEX_SCATTERING)
    sw, sh = SMALL_VOL.Scattering_Resolution
    vz = SMALL_VOL.Visibility_Resolution[2]
    step_v, step_s = float(np.float32(SMALL_VOL.Visibility_Step_Size)), float(np.float32(SMALL_VOL.Scattering_Step_Size))
    rng = np.random.RandomState(29)
    ok = total = 0
    for _ in range(60):
        X, Y = rng.randint(sw), rng.randint(sh)
        u, v = -1.0 + 2.0 * (X + 0.5) / sw, -1.0 + 2.0 * (Y + 0.5) / sh
        o, d = camera_ray64(u, v)
        hit = raycast64(o, d)
        to = hit if hit is not None else o + d * (step_v * vz)
        length = min(np.linalg.norm(to - o), step_v * vz)
        nx, ny = 0.5 * (u + 1.0), 0.5 * (v + 1.0)
        # f advances in fp32 in the shader (f += scattering_step_size): the loop count is that of the fp32 sum
        L, f = np.zeros(3), np.float32(0.0)
        while f < length:
            # (with equal step sizes every sample lies ON a slice boundary: which slice it reads is decided by the shader's
            #  fp32 quotient, so that one operation is taken in fp32 here too)
            rel = np.floor(np.float32(f) / np.float32(SMALL_VOL.Visibility_Step_Size))
            L += bilinear64(vis, nx, (ny + np.float64(rel)) / vz) * np.exp(-np.float64(f) * TAU)
            f = np.float32(f + np.float32(step_s))
        L *= step_s
        total += 1
        ok += abs(scat[Y, X, 3] - length) < 2e-4 and np.allclose(scat[Y, X, :3], L, rtol=5e-4, atol=1e-7)
    assert ok >= total - 2, (ok, total)


def test_depth_aware_pick_against_float64(orc):
    """volumetrics.glsl:34-54: of the 3 x 3 scattering texels round the fragment, the one whose stored ray length is
    closest to this pixel's (first strictly closer wins, starting from max_dist); out = L exp (-len tau) + its fog."""
    R = _shafts(orc)
    sw, sh = SMALL_VOL.Scattering_Resolution
    rng = np.random.RandomState(31)
    scat = rng.uniform(0.0, 1.0, size=(sh, sw, 4)).astype(np.float32)
    scat[..., 3] = rng.uniform(0.5, 9.0, size=(sh, sw))
    R.Write_Texture(B.TEX_SCATTERING, scat)
    n = 200
    L = rng.uniform(0, 2, size=(n, 3)).astype(np.float32)
    frm = np.tile(np.array([2.0, 2.0, 0.0], np.float32), (n, 1))
    to = (frm + rng.normal(size=(n, 3)) * 3.0).astype(np.float32)
    hit = (rng.uniform(size=n) < 0.85).astype(np.int32)
    frag = rng.uniform(-0.95, 0.95, size=(n, 2)).astype(np.float32)
    out = np.zeros((n, 3), np.float32)
    orc.lib.orc_probe_render_volumetrics(R._h, n, cf(L), cf(frm), cf(to), ip(hit), cf(frag), cf(out))
    s64 = scat.astype(np.float64)
    for q in range(n):
        tc = (frag[q].astype(np.float64) + 1.0) * 0.5
        length = np.linalg.norm(to[q].astype(np.float64) - frm[q]) if hit[q] else 20.0  # (a miss: len = max_dist)
        closest, fog = 20.0, np.zeros(3)
        for x in (-1, 0, 1):
            for y in (-1, 0, 1):
                # four-channel bilinear tap, mirrored repeat
                rgb = bilinear64(s64[..., :3], tc[0] + x / sw, tc[1] + y / sh)
                a = bilinear64(np.repeat(s64[..., 3:], 3, axis=2), tc[0] + x / sw, tc[1] + y / sh)[0]
                if abs(a - length) < closest:
                    closest, fog = abs(a - length), rgb
        want = L[q].astype(np.float64) * np.exp(-length * TAU) + fog
        assert np.allclose(out[q], want, rtol=2e-5, atol=2e-6), q


# ------------------------------------------------------------------------------ the indirect-specular bodies
def _gi(orc):
    R = examples.global_illumination(8, 8, Probes=SMALL_PROBES, Binding=orc)
    R.Set_Option(B.OPT_ATLAS_FORMAT, 1)
    rng = np.random.RandomState(37)
    P = SMALL_PROBES
    rad = rng.uniform(0, 1, size=(P.Probe_Count[1] * P.Radiance_Resolution, P.Probe_Count[0] * P.Radiance_Resolution, 3)).astype(np.float32)
    irr = rng.uniform(0, 1, size=(P.Probe_Count[1] * P.Irradiance_Resolution, P.Probe_Count[0] * P.Irradiance_Resolution, 3)).astype(np.float32)
    R.Write_Texture(B.TEX_RADIANCE, rad)
    R.Write_Texture(B.TEX_IRRADIANCE, irr)
    return R, rad.astype(np.float64), irr.astype(np.float64)


def _surface_points(n, seed):
    """points on the room's surfaces with their normals and a reflected direction: march a random ray to its hit"""
    rng = np.random.RandomState(seed)
    out = []
    while len(out) < n:
        o = rng.uniform((0.5, 0.5, -3.0), (6.0, 6.0, 6.0))
        if gi_sdf64(o) < 0.5:
            continue
        d = rng.normal(size=3)
        d /= np.linalg.norm(d)
        h = raycast64(o, d)
        if h is None:
            continue
        nrm, mat = gi_info64(h)
        r = d - nrm * (2.0 * np.dot(nrm, d))  # reflect (GLSL 4.30 section 8.5)
        out.append((h, nrm, r, mat))
    return out


def _probe_tap64(atlas, q, dims, pc, res, direction, lo=None):
    pid = q[2] * dims[0] * dims[1] + q[1] * dims[0] + q[0]                       # probe_utils.glsl:42-50
    base = np.array([pid % pc[0], pid // pc[0]]) / pc                            # probe_utils.glsl:52-56
    lo = 0.5 / res if lo is None else lo
    rid = np.clip(oct_encode64(direction), lo, 1.0 - lo)
    return bilinear64(atlas, *(base + rid / pc))


def _call_specular(orc, R, mode, pts, rough=None):
    n = len(pts)
    out = np.zeros((n, 3), np.float32)
    rough = np.zeros(n, np.float32) if rough is None else np.asarray(rough, np.float32)
    orc.lib.orc_probe_specular(R._h, mode, n, cf(np.array([p[0] for p in pts], np.float32)), cf(np.array([p[1] for p in pts], np.float32)),
                               cf(np.array([p[2] for p in pts], np.float32)), cf(rough), cf(out))
    return out


def test_specular_mode_2_best_probe_against_float64(orc):
    """sample_radiance_no_specular (render_probes.glsl:138-209): of the eight clamped cage probes of the reflection's hit,
    the one with the largest dot (probe_to_spec, -spec_normal) * visibility, strictly, from -2; one radiance tap at its
    clamped octahedral texel; plus that point's direct light with albedo 0 (M_ADD_INDIRECT_SPECULAR = 1)."""
    R, rad, _ = _gi(orc)
    P = SMALL_PROBES
    sp, dims, pc, rres = np.array(P.Grid_Spacing, np.float64), np.array(P.Grid_Dimensions), np.array(P.Probe_Count), P.Radiance_Resolution
    pts = _surface_points(70, 41)
    got = _call_specular(orc, R, 2, pts)
    ok = 0
    for (pos, nrm, r, _), g in zip(pts, got):
        spec = raycast64(pos + nrm * MSS * 5.0, r)
        if spec is None:
            want = np.zeros(3)
        else:
            sn, smat = gi_info64(spec)
            gp = np.floor(spec / sp).astype(int)
            best, bq, bpts = -2.0, None, None
            for i in range(8):
                q = np.clip(gp + np.array([i & 1, (i >> 1) & 1, (i >> 2) & 1]), 0, dims - 1)
                pts_ = spec - q * sp
                dist = np.linalg.norm(pts_)
                pts_ = pts_ / dist
                w = np.dot(pts_, -sn) * visibility64(spec + sn * MSS * 5.0, -pts_, dist - MSS * 5.0)
                if w > best:
                    best, bq, bpts = w, q, pts_
            want = _probe_tap64(rad, bq, dims, pc, rres, bpts) + direct64(spec, sn, r, (0.0, 0.0, 0.0), GI_MATS[smat][1], GI_MATS[smat][2])
        ok += np.allclose(g, want, rtol=2e-3, atol=2e-4)
    assert ok >= 0.9 * len(pts), ok  # (two marches per point, each of which may graze)


def test_specular_mode_1_against_float64(orc):
    """sample_radiance_with_specular (render_probes.glsl:71-136): the eight cage probes of the SHADED point light the
    reflection's hit, each weighted by max (softshadows (spec_pos, -probe_to_spec, 0.25, dist - 0.25, 0.5), 0.001) and
    the trilinear factor; lod = mix (0, log2 (rres), 2 roughness) only narrows the clamp of the tap (one level)."""
    R, rad, _ = _gi(orc)
    P = SMALL_PROBES
    sp, dims, pc, rres = np.array(P.Grid_Spacing, np.float64), np.array(P.Grid_Dimensions), np.array(P.Probe_Count), P.Radiance_Resolution
    pts = _surface_points(50, 43)
    rough = np.random.RandomState(5).uniform(0.0, 0.7, len(pts)).astype(np.float32)
    got = _call_specular(orc, R, 1, pts, rough)
    ok = 0
    for (pos, nrm, r, _), g, ro in zip(pts, got, rough.astype(np.float64)):
        spec = raycast64(pos + nrm * MSS * 5.0, r)
        if spec is None:
            want = np.zeros(3)
        else:
            gp = np.floor(pos / sp).astype(int)
            alpha = pos / sp - gp
            lod = float(int(np.log2(rres))) * (ro * 2.0)
            new_res = rres // int(lod + 1.0)
            acc, wsum = np.zeros(3), 0.0
            for i in range(8):
                off = np.array([i & 1, (i >> 1) & 1, (i >> 2) & 1])
                q = np.clip(gp + off, 0, dims - 1)
                pts_ = (pos - q * sp) + (spec - pos)
                dist = np.linalg.norm(pts_)
                pts_ = pts_ / dist
                w = max(softshadows64(spec, -pts_, MSS * 5.0, dist - MSS * 5.0, 0.5), 0.001)
                w *= np.where(off == 1, alpha, 1.0 - alpha).prod()
                acc += _probe_tap64(rad, q, dims, pc, rres, pts_, lo=0.5 / new_res) * w
                wsum += w
            want = acc / wsum if wsum else np.zeros(3)
        ok += np.allclose(g, want, rtol=3e-3, atol=3e-4)
    assert ok >= 0.85 * len(pts), ok  # (nine marches per point)


def test_specular_mode_3_against_float64(orc):
    """compute_indirect_specular (render_probes.glsl:211-244): the reflection's hit shaded in full -- direct light with its
    own albedo + kD irradiance / pi of sample_irradiance there (no specular of its own); the sky on a miss."""
    from test_oracle_pins import visibility64 as vis64
    R, _, irr = _gi(orc)
    P = SMALL_PROBES
    sp, dims, pc, ires = np.array(P.Grid_Spacing, np.float64), np.array(P.Grid_Dimensions), np.array(P.Probe_Count), P.Irradiance_Resolution
    pts = _surface_points(40, 47)
    got = _call_specular(orc, R, 3, pts)
    ok = 0
    for (pos, nrm, r, _), g in zip(pts, got):
        spec = raycast64(pos + nrm * MSS * 5.0, r)
        if spec is None:
            want = np.array([0.30, 0.36, 0.60]) - r[1] * 0.7
        else:
            sn, smat = gi_info64(spec)
            albedo, metallic, rough = GI_MATS[smat]
            gp = np.floor(spec / sp).astype(int)
            alpha = spec / sp - gp
            acc, wsum = np.zeros(3), 0.0
            for i in range(8):  # sample_irradiance, render_probes.glsl:6-69
                off = np.array([i & 1, (i >> 1) & 1, (i >> 2) & 1])
                q = np.clip(gp + off, 0, dims - 1)
                h = q * sp - spec
                dist = np.linalg.norm(h)
                dp = h / dist
                w = ((dp @ sn + 1.0) * 0.5) ** 2 + 0.2
                w *= vis64(spec + sn * MSS * 5.0, dp, dist - MSS * 5.0)
                if w < 0.2:
                    w *= w * w / 0.04
                w *= np.where(off == 1, alpha, 1.0 - alpha).prod()
                acc += np.sqrt(_probe_tap64(irr, q, dims, pc, ires, sn)) * w
                wsum += w
            irradiance = (acc / wsum) ** 2 if wsum else np.zeros(3)
            kD, _ = cook_torrance64(sn, -r, r - sn * (2.0 * np.dot(sn, r)), albedo, metallic, rough)
            want = kD * irradiance / PI + direct64(spec, sn, r, albedo, metallic, rough)
        ok += np.allclose(g, want, rtol=3e-3, atol=3e-4)
    assert ok >= 0.85 * len(pts), ok


# ------------------------------------------------------------------------- the CPU_Best builder's acceptance order
def test_cpu_best_acceptance_order_against_restatement(orc):
    """Update_Partitioning_CPU with Find_Candidates (madarch-renderers.adb:609-755) on simple_scene: per cell, the
    primitives closer to the centre than the closest one + the cell's diagonal are the candidates; 27 sample points
    (3 x 3 x 3 over the cell, corners included) each ACCEPT the candidate closest to them, and a kind's indices are
    written in the order of first acceptance.  Restated here in float64 with the scene's own primitive list; cells
    where two candidates are within 1e-5 of each other at a sample point (the fp32 arg-min may differ) are skipped."""
    R = examples.simple_scene(8, 8, Binding=orc)  # (builds the table with CPU_Best, as examples/simple_scene/main.adb:122)
    table = R.Read_Partitioning()
    part = R.Scene.Partitioning_Config
    dims, spc, off, icount = np.array(part.Grid_Dimensions), np.array(part.Grid_Spacing, np.float64), np.array(part.Grid_Offset, np.float64), part.Index_Count
    # the scene's primitives in kind order (Sphere, Plane, Box), from the std140 image the renderer keeps
    ubo = R.Read_Scene_Buffer()
    prims = []
    for k, (kind, _) in enumerate(R.Scene.Prims_Count):
        count_off, array_off, stride, _ = R.Scene_Layout(False, k)
        n = int(np.frombuffer(ubo[count_off:count_off + 4].tobytes(), np.int32)[0])
        for i in range(n):
            f = np.frombuffer(ubo[array_off + stride * i:array_off + stride * i + 32].tobytes(), np.float32).astype(np.float64)
            prims.append((k, i, kind.name, f))

    def dist(pr, p):
        _, _, name, f = pr
        if name == "Sphere":
            return np.linalg.norm(f[:3] - p) - f[3]
        if name == "Plane":
            return np.dot(f[:3], p) + f[3]
        q = np.abs(f[:3] - p) - f[4:7]
        return np.linalg.norm(np.maximum(q, 0.0)) + min(q.max(), 0.0)

    nk = len(R.Scene.Prims_Count)
    rng = np.random.RandomState(53)
    checked = 0
    for cell in rng.permutation(dims.prod())[:500]:
        X, Y, Z = cell // (dims[1] * dims[2]), (cell // dims[2]) % dims[1], cell % dims[2]
        grid_pos = np.array([X, Y, Z]) * spc + off
        centre = grid_pos + spc * 0.5
        ds = np.array([dist(pr, centre) for pr in prims])
        cands = [pr for pr, d in zip(prims, ds) if d < ds.min() + np.linalg.norm(spc)]
        accepted, ambiguous = [], False
        for sx in range(3):
            for sy in range(3):
                for sz in range(3):
                    pt = np.array([sx, sy, sz]) / 2.0 * spc + grid_pos
                    dd = np.sort([dist(pr, pt) for pr in cands])
                    ambiguous |= len(dd) > 1 and dd[1] - dd[0] < 1e-5
                    best = min(cands, key=lambda pr: dist(pr, pt))
                    if best[:2] not in [a[:2] for a in accepted]:
                        accepted.append(best)
        if ambiguous:
            continue
        want_counts = [sum(1 for a in accepted if a[0] == k) for k in range(nk)]
        want_idx = [a[1] for k in range(nk) for a in accepted if a[0] == k]
        if sum(want_counts) > icount:
            continue  # (an overflowing cell is cut: covered by the parity tests)
        assert table[cell, :nk].tolist() == want_counts, (cell, table[cell], want_counts)
        assert table[cell, nk:nk + len(want_idx)].tolist() == want_idx, (cell, table[cell], want_idx)
        checked += 1
    assert checked > 200


# --------------------------------------------------------------------------- MDH_OPT_RADIANCE_MIPS (an optional switch)
def _mips64(rad):
    """levels 0 .. log2 (res) of the atlas image: 2 x 2 box filter of the level below"""
    levels = [rad]
    while levels[-1].shape[0] % 2 == 0 and levels[-1].shape[1] % 2 == 0 and len(levels) <= int(np.log2(SMALL_PROBES.Radiance_Resolution)):
        a = levels[-1]
        levels.append((a[0::2, 0::2] + a[0::2, 1::2] + a[1::2, 0::2] + a[1::2, 1::2]) / 4.0)
    return levels


def _lod_tap64(levels, q, dims, pc, res, direction, lod, lo=None):
    """textureLod as GL_LINEAR_MIPMAP_LINEAR: lod clamped to the chain, two bilinear taps mixed by its fraction"""
    d = min(max(lod, 0.0), float(len(levels) - 1))
    l0 = int(np.floor(d))
    f = d - l0
    a = _probe_tap64(levels[l0], q, dims, pc, res, direction, lo=lo)
    return a if f == 0.0 else a * (1.0 - f) + _probe_tap64(levels[l0 + 1], q, dims, pc, res, direction, lo=lo) * f


def test_radiance_mip_levels_against_float64(orc):
    """The switch's chain: every level is the 2 x 2 box filter of the one below over the whole atlas image, down to one
    texel per probe; refused for a resolution that is no power of two; off by default."""
    R, rad, _ = _gi(orc)
    assert R.Get_Option(B.OPT_RADIANCE_MIPS) == 0
    with pytest.raises(B.MadarchError):
        R.Read_Texture(B.TEX_RADIANCE_MIP0 + 1)
    R.Set_Option(B.OPT_RADIANCE_MIPS, 1)
    levels = _mips64(rad)
    assert len(levels) == 5  # 16, 8, 4, 2, 1 texels per probe
    for l in range(1, len(levels)):
        got = R.Read_Texture(B.TEX_RADIANCE_MIP0 + l)
        assert got.shape == levels[l].shape
        assert np.allclose(got, levels[l], rtol=1e-6, atol=1e-7), l
    with pytest.raises(B.MadarchError):
        R.Read_Texture(B.TEX_RADIANCE_MIP0 + len(levels))
    from helpers import ODD_PROBES
    R2 = examples.global_illumination(8, 8, Probes=ODD_PROBES, Binding=orc)
    with pytest.raises(B.MadarchError):
        R2.Set_Option(B.OPT_RADIANCE_MIPS, 1)  # 12 texels per probe: no chain without boxes across tiles


def test_specular_taps_over_the_mip_chain_against_float64(orc):
    """With the switch on, mode 2's tap reads level 1 (textureLod (.., 1.0), render_probes.glsl:197) and mode 1's reads
    lod = mix (0, radiance_lods, 2 roughness) between two levels (render_probes.glsl:84-86,131)."""
    R, rad, _ = _gi(orc)
    R.Set_Option(B.OPT_RADIANCE_MIPS, 1)
    levels = _mips64(rad)
    P = SMALL_PROBES
    sp, dims, pc, rres = np.array(P.Grid_Spacing, np.float64), np.array(P.Grid_Dimensions), np.array(P.Probe_Count), P.Radiance_Resolution
    # mode 2
    pts = _surface_points(50, 47)
    got = _call_specular(orc, R, 2, pts)
    ok = differs = 0
    for (pos, nrm, r, _), g in zip(pts, got):
        spec = raycast64(pos + nrm * MSS * 5.0, r)
        if spec is None:
            want = want0 = np.zeros(3)
        else:
            sn, smat = gi_info64(spec)
            gp = np.floor(spec / sp).astype(int)
            best, bq, bpts = -2.0, None, None
            for i in range(8):
                q = np.clip(gp + np.array([i & 1, (i >> 1) & 1, (i >> 2) & 1]), 0, dims - 1)
                pts_ = spec - q * sp
                dist = np.linalg.norm(pts_)
                pts_ = pts_ / dist
                w = np.dot(pts_, -sn) * visibility64(spec + sn * MSS * 5.0, -pts_, dist - MSS * 5.0)
                if w > best:
                    best, bq, bpts = w, q, pts_
            direct = direct64(spec, sn, r, (0.0, 0.0, 0.0), GI_MATS[smat][1], GI_MATS[smat][2])
            want = _lod_tap64(levels, bq, dims, pc, rres, bpts, 1.0) + direct
            want0 = _probe_tap64(rad, bq, dims, pc, rres, bpts) + direct
        ok += np.allclose(g, want, rtol=2e-3, atol=2e-4)
        differs += not np.allclose(want, want0, rtol=2e-3, atol=2e-4)
    assert ok >= 0.9 * len(pts), ok
    assert differs >= 0.5 * len(pts)  # (level 1 of a random atlas is not level 0: the test tells them apart)
    # mode 1
    pts = _surface_points(40, 53)
    rough = np.random.RandomState(6).uniform(0.0, 0.7, len(pts)).astype(np.float32)
    got = _call_specular(orc, R, 1, pts, rough)
    ok = 0
    for (pos, nrm, r, _), g, ro in zip(pts, got, rough.astype(np.float64)):
        spec = raycast64(pos + nrm * MSS * 5.0, r)
        if spec is None:
            want = np.zeros(3)
        else:
            gp = np.floor(pos / sp).astype(int)
            alpha = pos / sp - gp
            lod = float(int(np.log2(rres))) * (ro * 2.0)
            new_res = rres // int(lod + 1.0)
            acc, wsum = np.zeros(3), 0.0
            for i in range(8):
                off = np.array([i & 1, (i >> 1) & 1, (i >> 2) & 1])
                q = np.clip(gp + off, 0, dims - 1)
                pts_ = (pos - q * sp) + (spec - pos)
                dist = np.linalg.norm(pts_)
                pts_ = pts_ / dist
                w = max(softshadows64(spec, -pts_, MSS * 5.0, dist - MSS * 5.0, 0.5), 0.001)
                w *= np.where(off == 1, alpha, 1.0 - alpha).prod()
                acc += _lod_tap64(levels, q, dims, pc, rres, pts_, lod, lo=0.5 / new_res) * w
                wsum += w
            want = acc / wsum if wsum else np.zeros(3)
        ok += np.allclose(g, want, rtol=3e-3, atol=3e-4)
    assert ok >= 0.85 * len(pts), ok
