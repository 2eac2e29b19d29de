"""user-defined primitive kinds.  A kind's Distance / Normal / Material
expressions (madarch_amd.exprs, the mirror of Madarch.Exprs) are compiled to the MDH_X register
programs of include/madarch_hip.h and interpreted by the kernels (and, independently, by the oracle).

The strongest check needs no new ground truth: the four built-in kinds, restated under other names
with the reference's own expressions (tests/custom_kinds.py), must give the hand-written built-in
paths bit for bit."""
import numpy as np
import pytest

import custom_kinds as ck
from helpers import SMALL_PROBES, same_bits, snapshot
from madarch_amd import _binding as B
from madarch_amd import exprs, materials, renderers, scenes, values, windows
from madarch_amd.lights import point_lights, spot_lights
from madarch_amd.primitives import boxes, planes, spheres, triangles

ROOM = (((0.0, 1.0, 0.0), 1.0), ((0.0, -1.0, 0.0), 7.0), ((1.0, 0.0, 0.0), 1.0), ((-1.0, 0.0, 0.0), 7.0), ((0.0, 0.0, 1.0), 6.0), ((0.0, 0.0, -1.0), 7.0))
TRIS = (((1.0, 0.5, 2.5), (2.5, 0.8, 3.0), (1.6, 2.2, 2.8)), ((4.0, 3.0, 4.0), (5.0, 3.2, 3.5), (4.4, 4.1, 4.6)))


def room(binding, custom, W=56, H=40, mode=0, partition=False, extra=(), custom_lights=False, lamp=False):
    """A room of planes with spheres, boxes and triangles: built-in kinds, or their restatements."""
    K = (ck.My_Sphere, ck.My_Plane, ck.My_Box, ck.My_Triangle) if custom else (spheres.Sphere, planes.Plane, boxes.Box, triangles.Triangle)
    mk = (ck.sphere, ck.plane, ck.box, ck.triangle) if custom else (spheres.Create, planes.Create, boxes.Create, triangles.Create)
    part = scenes.Partitioning_Settings(Enable=partition, Index_Count=12, Grid_Dimensions=(6, 6, 8), Grid_Spacing=(1.5, 1.5, 1.75), Grid_Offset=(-1.5, -1.5, -6.5))
    Scene = scenes.Compile(All_Primitives=[(K[0], 6), (K[1], 8), (K[2], 4), (K[3], 3)] + [(k, 3) for k, _ in extra],
                           All_Lights=[(ck.My_Point_Light if custom_lights else point_lights.Point_Light, 2),
                                       (ck.My_Spot_Light if custom_lights else spot_lights.Spot_Light, 3)] + ([(ck.Lamp, 6)] if lamp else []),
                           Partitioning=part)
    R = renderers.Create(windows.Open(W, H, "custom"), Scene, Probes=SMALL_PROBES, Volumetrics=renderers.No_Volumetrics, Binding=binding)
    for (n, o), m in zip(ROOM, (0, 0, 1, 2, 0, 0)):
        R.Add_Primitive(K[1], mk[1](n, o, m))
    for c, rad in (((3.0, 4.0, 3.0), 1.0), ((1.0, 1.0, 2.0), 0.5), ((5.0, 2.0, 4.0), 0.7)):
        R.Add_Primitive(K[0], mk[0](c, rad, 3))
    for c, s in (((3.0, 0.0, 4.0), (1.5, 1.5, 1.5)), ((0.5, 3.0, 3.0), (0.4, 0.6, 0.5))):
        R.Add_Primitive(K[2], mk[2](c, s, 2))
    for a, b, c in TRIS:
        R.Add_Primitive(K[3], mk[3](a, b, c, 1))
    for k, ents in extra:
        for e in ents:
            R.Add_Primitive(k, e)
    R.Set_Material(0, materials.Create((0.3, 0.3, 0.3), 0.0, 0.6))
    R.Set_Material(1, materials.Create((1.0, 0.0, 0.0), 0.0, 0.6))
    R.Set_Material(2, materials.Create((0.0, 0.0, 1.0), 0.2, 0.5))
    R.Set_Material(3, materials.Create((0.1, 0.1, 0.1), 0.9, 0.1))
    # Set_Light (Index, ...) sets the kind's count AND total_light_count to Index (renderers.adb:478-482), and the
    # light loop walks the kinds by cumulative counts: the sequence below leaves point 1, point 2, spot 1, spot 3
    # (and the lamp) reachable
    PL, SL = (ck.My_Point_Light, ck.My_Spot_Light) if custom_lights else (point_lights.Point_Light, spot_lights.Spot_Light)
    mkp, mks = (ck.point_light, ck.spot_light) if custom_lights else (point_lights.Create, spot_lights.Create)
    R.Set_Light(1, PL, mkp((3.0, 5.0, 1.0), (0.9, 0.9, 0.8)))
    R.Set_Light(2, PL, mkp((5.5, 1.0, 5.0), (0.2, 0.3, 0.6)))
    R.Set_Light(1, SL, mks((1.0, 6.0, 3.0), (0.3, -0.9, 0.1), 0.7, (0.8, 0.6, 0.9)))
    R.Set_Light(3, SL, mks((4.0, 6.0, 1.0), (-0.2, -0.9, 0.3), 0.5, (0.9, 0.9, 0.5)))
    if lamp:
        R.Set_Light(1, ck.Lamp, ck.lamp((5.0, 6.5, 4.0), 1.5, (1.2, 1.1, 0.9)))
        R.Set_Light(6, ck.Lamp, ck.lamp((0.0, 0.0, 0.0), 0.0, (0.0, 0.0, 0.0)))
    R.Set_Camera_Position((2.0, 2.0, 0.0))
    R.Set_Option(B.OPT_SCREEN_MODE, mode)
    R.Set_Option(B.OPT_GBUFFER, 1)
    if partition:
        R.Update_Partitioning(Method=renderers.CPU_Best)
    return R


EXTRA = ((ck.Torus, [ck.torus((3.0, 1.2, 2.5), 0.8, 0.25, 3)]),
         (ck.Ripple, [ck.ripple(-0.6, 0.15, 2.5, 1)]),
         (ck.Capsule, [ck.capsule((1.0, 3.0, 4.0), (2.0, 4.5, 3.0), 0.2, 1, 2), ck.capsule((4.5, 0.5, 2.0), (5.5, 1.5, 2.2), 0.4, 1, 2)]))


def assert_same(a, b):
    assert a.keys() == b.keys()
    for k in a:
        assert same_bits(a[k], b[k]), "%s differs" % k


# --------------------------------------------------------------------------------- CPU: the oracle
def test_compiler_lowers_in_the_order_of_the_glsl_contract():
    """(x x' + y y') + z z': Length of a difference is SUB x3, MUL, MUL, ADD, MUL, ADD, SQRT."""
    S, P = exprs.Struct_Identifier("prim"), exprs.Value_Identifier("x")
    words = exprs.compile_program((S.Get(ck.S_Center) - P).Length(), ck.My_Sphere.comps, values.Float_Kind, "x")
    ops = [w & 255 for w in words]
    X = exprs
    assert ops == [X.X_POINT] * 3 + [X.X_COMP] * 3 + [X.X_SUB] * 3 + [X.X_MUL, X.X_MUL, X.X_ADD, X.X_MUL, X.X_ADD, X.X_SQRT, X.X_MOV]
    # literals go through Single'Image (6 digits) like every literal of the generated GLSL
    assert exprs.image_roundtrip(0.1234567) == np.float32(0.123457) and exprs.image_roundtrip(0.000001) == np.float32(1e-6)
    with pytest.raises(exprs.Unsupported_Expr):
        exprs.compile_program(exprs.External_Call("noise", [S], [P]), ck.My_Sphere.comps, values.Float_Kind, "x")
    with pytest.raises(exprs.Type_Inference_Error):
        exprs.compile_program(P + exprs.Value_Identifier("nobody"), ck.My_Sphere.comps, values.Vector3_Kind, "x")


@pytest.mark.parametrize("mode,partition", [(0, False), (2, True)])
def test_oracle_restated_kinds_equal_the_built_in_kinds(orc, mode, partition):
    assert_same(snapshot(room(orc, False, mode=mode, partition=partition), 2), snapshot(room(orc, True, mode=mode, partition=partition), 2))


def test_oracle_new_kinds_known_answers(orc):
    R = room(orc, False, extra=EXTRA)
    R.Set_Option(B.OPT_ADA_EVAL_DIV, 0)
    pts = np.array([[3.0, 1.2, 2.5], [3.8, 1.2, 2.5], [3.0, 2.0, 2.5], [1.5, 3.75, 3.5], [0.0, 3.0, 4.0]], dtype=np.float32)
    d, n = R.Eval_Distances_To(pts, [ck.Torus])
    q = pts - np.float32([3.0, 1.2, 2.5])
    want = np.hypot(np.hypot(q[:, 0], q[:, 2]) - 0.8, q[:, 1]) - 0.25
    assert np.allclose(d, want, atol=2e-6)
    assert np.allclose(n[1], [0.0, 0.0, 0.0], atol=1.0) and abs(np.linalg.norm(n[2]) - 1.0) < 1e-3  # a unit normal away from the ring
    d, _ = R.Eval_Distances_To(pts, [ck.Capsule])
    a, b = np.float32([1.0, 3.0, 4.0]), np.float32([2.0, 4.5, 3.0])
    pa, ba = pts - a, b - a
    h = np.clip(pa @ ba / (ba @ ba), 0.0, 1.0)
    first = np.linalg.norm(pa - np.outer(h, ba), axis=1) - 0.2
    a2, b2 = np.float32([4.5, 0.5, 2.0]), np.float32([5.5, 1.5, 2.2])
    pa2, ba2 = pts - a2, b2 - a2
    second = np.linalg.norm(pa2 - np.outer(np.clip(pa2 @ ba2 / (ba2 @ ba2), 0.0, 1.0), ba2), axis=1) - 0.4
    assert np.allclose(d, np.minimum(first, second), atol=2e-6)
    # Madarch.Values."/" on two floats adds (values.adb:112): the capsule's h becomes clamp (dot + dot2, 0, 1)
    R.Set_Option(B.OPT_ADA_EVAL_DIV, 1)
    d_ada, _ = R.Eval_Distances_To(pts[3:4], [ck.Capsule])
    h_ada = np.clip(pa[3] @ ba + ba @ ba, 0.0, 1.0)
    assert abs(d_ada[0] - min(np.linalg.norm(pa[3] - h_ada * ba) - 0.2, 1e9)) < 1e-5 or d_ada[0] <= second[3] + 1e-5
    # the Material program picks an id by the radius: the thin capsule is red (1), the thick one blue (2)
    img = snapshot(R, 1)
    assert np.isfinite(img["image"]).all()


def test_oracle_restated_lights_equal_the_built_in_lights(orc):
    assert_same(snapshot(room(orc, False), 2), snapshot(room(orc, False, custom_lights=True), 2))
    # and the lamp really lights the scene: its frame differs from the one without it
    a, b = snapshot(room(orc, False), 1), snapshot(room(orc, False, lamp=True), 1)
    assert not same_bits(a["image"], b["image"]) and np.isfinite(b["image"]).all()


def test_invalid_programs_are_rejected(orc):
    good = ck.My_Sphere.programs

    def broken(words):
        k = type(ck.My_Sphere)("Broken", ck.My_Sphere.comps, ck.My_Sphere.distance, ck.My_Sphere.normal, ck.My_Sphere.material)
        d, n, m = good()
        k.programs = lambda: (words(d), n, m)
        return k

    for mutate in (lambda d: [99] + d[1:],                      # unknown instruction
                   lambda d: d[:-1] + [exprs.X_COMP | 0 << 8 | 200 << 16],  # component float past the instance
                   lambda d: d + [exprs.X_LIT]):                # literal without its word
        Scene = scenes.Compile(All_Primitives=[(broken(mutate), 2)], All_Lights=[(point_lights.Point_Light, 1)],
                               Partitioning=scenes.Partitioning_Settings(Enable=False))
        with pytest.raises(B.MadarchError) as e:
            renderers.Create(windows.Open(8, 8, "x"), Scene, Probes=SMALL_PROBES, Volumetrics=renderers.No_Volumetrics, Binding=orc)
        assert e.value.status == B.MDH_E_UNSUPPORTED_KIND


# --------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("mode,partition,jit", [(0, False, 0), (2, True, 0), (0, True, 1), (2, False, 1)])
def test_hip_restated_kinds_equal_the_built_in_kinds(hip, mode, partition, jit):
    R = room(hip, True, mode=mode, partition=partition)
    R.Set_Option(B.OPT_JIT, jit)
    assert_same(snapshot(room(hip, False, mode=mode, partition=partition), 2), snapshot(R, 2))
    assert R.Get_Option(B.OPT_JIT) == jit  # (a failed hiprtc build would have switched it off)


@pytest.mark.gpu
@pytest.mark.parametrize("partition", [False, True])
def test_hip_new_kinds_against_the_oracle(hip, orc, partition):
    outs = [snapshot(room(b, False, W=72, H=48, partition=partition, extra=EXTRA), 2) for b in (hip, orc)]
    assert_same(*outs)
    assert (outs[0]["gb_index"] >= 21).any()  # pixels whose primary hit is a torus, the ripple or a capsule (flat index after 6 + 8 + 4 + 3)
    pts = np.random.default_rng(7).uniform(-1.0, 7.0, (500, 3)).astype(np.float32)
    for ada in (0, 1):
        res = []
        for b in (hip, orc):
            R = room(b, True, extra=EXTRA)
            R.Set_Option(B.OPT_ADA_EVAL_DIV, ada)
            res.append(R.Eval_Distances_To(pts, [ck.My_Triangle, ck.Torus, ck.Capsule, ck.My_Box, ck.Ripple]))
        assert same_bits(res[0][0], res[1][0]) and same_bits(res[0][1], res[1][1])


@pytest.mark.gpu
def test_hip_user_defined_lights(hip, orc):
    assert_same(snapshot(room(hip, False), 2), snapshot(room(hip, False, custom_lights=True), 2))
    assert_same(snapshot(room(hip, False, W=72, H=48, lamp=True, extra=EXTRA[:1]), 2), snapshot(room(orc, False, W=72, H=48, lamp=True, extra=EXTRA[:1]), 2))


@pytest.mark.gpu
def test_hip_jit_equals_the_interpreter(hip):
    """MDH_OPT_JIT: the same programs compiled into the kernels with hiprtc give the interpreter's frames bit for bit
    (and so the built-in paths, for the restated kinds)."""
    outs = []
    for jit in (0, 1):
        R = room(hip, True, W=72, H=48, custom_lights=True, lamp=True, extra=EXTRA)
        R.Set_Option(B.OPT_JIT, jit)
        outs.append(snapshot(R, 2))
    assert_same(*outs)


@pytest.mark.gpu
def test_compiled_distance_programs_join_the_wave_level_culling(hip, tmp_path, monkeypatch):
    """A Distance program that ends in  sqrt (A) - R,  sqrt (A) + S  or  sqrt (A)  is compiled in a min form whose root is
    culled like the built-in spheres' (mdh_api.hip: jit_min_form): the generated header shows which kinds have one, and the
    frames are the interpreter's bit for bit (no culling there)."""
    dump = tmp_path / "mdh_jit_kinds.h"
    monkeypatch.setenv("MADARCH_HIP_JIT_DUMP", str(dump))
    outs = []
    for jit in (1, 0):
        R = room(hip, True, W=96, H=64, extra=EXTRA)
        R.Set_Option(B.OPT_JIT, jit)
        outs.append(snapshot(R, 2))
        assert R.Get_Option(B.OPT_JIT) == jit
    assert_same(*outs)
    text = dump.read_text()
    has_min = {k for k in range(8) if "float jit_p%d_min(" % k in text}
    # kinds in table order: My_Sphere, My_Plane, My_Box, My_Triangle, Torus, Ripple, Capsule
    assert {0, 4, 6} <= has_min, has_min  # sqrt (dot (v, v)) - radius
    assert 1 not in has_min and 5 not in has_min  # a plane has no root; the ripple's distance does not end in one
    assert "jit_closest_all" in text


@pytest.mark.gpu
@pytest.mark.parametrize("jit", [1, 0])
def test_hip_new_kinds_over_the_radiance_mip_chain(hip, orc, jit):
    """MDH_OPT_RADIANCE_MIPS with user-defined kinds: the screen kernel's variant for the optional paths, compiled around the
    scene's programs by hiprtc (or interpreting them), against the oracle."""
    outs = []
    for b in (hip, orc):
        R = room(b, False, W=72, H=48, extra=EXTRA)
        R.Set_Option(B.OPT_JIT, jit)
        R.Set_Option(B.OPT_RADIANCE_MIPS, 1)
        outs.append(snapshot(R, 2))
        if b is hip:
            assert R.Get_Option(B.OPT_JIT) == jit
    assert_same(*outs)
    plain = snapshot(room(hip, False, W=72, H=48, extra=EXTRA), 2)
    assert not same_bits(plain["image"], outs[0]["image"])  # (the chain is read: reflecting spheres and capsules are in view)


@pytest.mark.gpu
def test_hip_rejects_invalid_programs(hip):
    k = type(ck.My_Sphere)("Broken", ck.My_Sphere.comps, ck.My_Sphere.distance, ck.My_Sphere.normal, ck.My_Sphere.material)
    d, n, m = ck.My_Sphere.programs()
    k.programs = lambda: ([exprs.X_MOV | 70 << 8] + d, n, m)  # register 70 does not exist
    Scene = scenes.Compile(All_Primitives=[(k, 2)], All_Lights=[(point_lights.Point_Light, 1)], Partitioning=scenes.Partitioning_Settings(Enable=False))
    with pytest.raises(B.MadarchError) as e:
        renderers.Create(windows.Open(8, 8, "x"), Scene, Probes=SMALL_PROBES, Volumetrics=renderers.No_Volumetrics, Binding=hip)
    assert e.value.status == B.MDH_E_UNSUPPORTED_KIND


# ---- a kind is what its expressions say, not what it is called (madarch-primitives.ads:24-30)
def _ring_scene(binding, kind, light_kind=None):
    """A floor, a torus of the given kind and one light; the primary-ray geometry of a 40 x 28 frame."""
    Scene = scenes.Compile(All_Primitives=[(planes.Plane, 2), (kind, 2)], All_Lights=[(light_kind or point_lights.Point_Light, 2)],
                           Partitioning=scenes.Partitioning_Settings(Enable=False))
    R = renderers.Create(windows.Open(40, 28, "ring"), Scene, Probes=SMALL_PROBES, Volumetrics=renderers.No_Volumetrics, Binding=binding)
    R.Add_Primitive(planes.Plane, planes.Create((0.0, 1.0, 0.0), 1.0, 0))
    R.Add_Primitive(kind, ck.torus((2.5, 1.5, 4.0), 1.2, 0.4, 1))
    R.Set_Material(0, materials.Create((0.4, 0.4, 0.4), 0.0, 0.6))
    R.Set_Material(1, materials.Create((0.8, 0.2, 0.1), 0.3, 0.4))
    if light_kind is None:
        R.Set_Light(1, point_lights.Point_Light, point_lights.Create((3.0, 5.0, 1.0), (0.9, 0.9, 0.8)))
    else:
        R.Set_Light(1, light_kind, ck.lamp((3.0, 6.0, 2.0), 1.0, (1.2, 1.1, 0.9)))
    R.Set_Camera_Position((2.0, 2.0, 0.0))
    R.Set_Option(B.OPT_GBUFFER, 1)
    return R


# the torus of tests/custom_kinds.py under the NAME of a built-in kind, and the lamp light called "PointLight"
Sphere_Named_Torus = ck.primitives.Create("Sphere", ck.Torus.comps, ck.Torus.distance, ck.Torus.normal, ck.Torus.material)
Point_Light_Named_Lamp = ck.lights.Create("PointLight", ck.Lamp.comps, ck.Lamp.sample, ck.Lamp.position)


def test_oracle_kind_named_sphere_runs_its_own_expressions(orc):
    a = snapshot(_ring_scene(orc, ck.Torus), 1)
    b = snapshot(_ring_scene(orc, Sphere_Named_Torus), 1)
    for k in a:
        assert same_bits(a[k], b[k]), k
    # ... and the hole of the ring shows the floor (a sphere of that centre would cover it)
    assert len(np.unique(a["gb_index"])) >= 2
    c = snapshot(_ring_scene(orc, ck.Torus, ck.Lamp), 1)
    d = snapshot(_ring_scene(orc, Sphere_Named_Torus, Point_Light_Named_Lamp), 1)
    for k in c:
        assert same_bits(c[k], d[k]), k
    assert not same_bits(a["image"], c["image"])


@pytest.mark.gpu
@pytest.mark.parametrize("jit", [1, 0])
def test_hip_kind_named_sphere_runs_its_own_expressions(hip, orc, jit):
    """A user kind called "Sphere" whose Distance is a torus (and a light called "PointLight" whose Sample is the
    lamp's) must go through its MDH_X programs -- compiled or interpreted -- not through the hand-written sphere."""
    want = snapshot(_ring_scene(orc, Sphere_Named_Torus, Point_Light_Named_Lamp), 2)
    R = _ring_scene(hip, Sphere_Named_Torus, Point_Light_Named_Lamp)
    R.Set_Option(B.OPT_JIT, jit)
    got = snapshot(R, 2)
    from helpers import assert_parity
    assert_parity(got, want)
    assert same_bits(got["image"], snapshot(_ring_scene(hip, ck.Torus, ck.Lamp), 2)["image"])
