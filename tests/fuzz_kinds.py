"""Random user-defined kinds for differential testing (scripts/fuzz_custom_kinds.py, tests/test_gpu_fuzz.py): Distance
and Normal expression trees over the point and the kind's components -- arithmetic, min / max / clamp, Let_In,
If_Then_Else, vector algebra, the trigonometric builtins -- lowered to MDH_X by madarch_amd/exprs.py."""
import numpy as np

from helpers import SMALL_PROBES, assert_parity, same_bits, snapshot
from madarch_amd import _binding as B, components, entities, materials, primitives, renderers, scenes, values, windows
from madarch_amd.exprs import Construct_Vector3, Forward_Difference, If_Then_Else, Let_In, Literal, Value_Identifier
from madarch_amd.lights import point_lights
from madarch_amd.primitives import planes
from madarch_amd.primitives.materials import Material_Id

V3K, FK = values.Vector3_Kind, values.Float_Kind
C_V1, C_V2 = components.Create("va", V3K), components.Create("vb", V3K)
C_F1, C_F2 = components.Create("fa", FK), components.Create("fb", FK)
LF = lambda x: Literal(values.Float(float(x)))
LV = lambda v: Literal(values.Vector3(tuple(float(x) for x in v)))


class Gen:
    """typed random expressions over the point P and the kind's components"""

    def __init__(self, rng, S, P):
        self.rng, self.S, self.P, self.lets = rng, S, P, 0

    def f(self, depth):
        r, S, P = self.rng, self.S, self.P
        if depth <= 0 or r.integers(0, 6) == 0:
            c = r.integers(0, 5)
            if c == 0: return LF(r.uniform(-2.0, 2.0))
            if c == 1: return S.Get(C_F1 if r.integers(0, 2) else C_F2)
            if c == 2: return P.Get(int(r.integers(0, 3)))
            if c == 3: return S.Get(C_V1 if r.integers(0, 2) else C_V2).Get(int(r.integers(0, 3)))
            return (P - S.Get(C_V1)).Length()
        c = r.integers(0, 19)
        a = lambda: self.f(depth - 1)
        v = lambda: self.v(depth - 1)
        if c == 0: return a() + a()
        if c == 1: return a() - a()
        if c == 2: return a() * a()
        if c == 3: return a() / (a().Abs_Value() + LF(0.25))
        if c == 4: return a().Min(a())
        if c == 5: return a().Max(a())
        if c == 6: return a().Abs_Value()
        if c == 7: return -a()
        if c == 8: return a().Abs_Value().Sqrt()
        if c == 9: return (a() * LF(0.7)).Sin()
        if c == 10: return (a() * LF(0.7)).Cos()
        if c == 11: return a().Atan()
        if c == 12: return a().Floor() * LF(0.25)
        if c == 13: return a().Clamp(LF(-1.0), LF(1.5))
        if c == 14: return v().Length()
        if c == 15: return v().Dot(v())
        if c == 16: return If_Then_Else(a() < a(), a(), a())
        if c == 17:
            self.lets += 1
            name = "t%d" % self.lets
            return a().Let_In(FK, name, Value_Identifier(name) * Value_Identifier(name) - a())
        return a().Clamp(LF(-0.9), LF(0.9)).Asin()

    def v(self, depth):
        r, S, P = self.rng, self.S, self.P
        if depth <= 0 or r.integers(0, 5) == 0:
            c = r.integers(0, 4)
            if c == 0: return P
            if c == 1: return S.Get(C_V1 if r.integers(0, 2) else C_V2)
            if c == 2: return LV(r.uniform(-1.0, 1.0, 3))
            return P - S.Get(C_V2)
        c = r.integers(0, 9)
        a = lambda: self.f(depth - 1)
        v = lambda: self.v(depth - 1)
        if c == 0: return v() + v()
        if c == 1: return v() - v()
        if c == 2: return v() * a()
        if c == 3: return v() / (a().Abs_Value() + LF(0.5))
        if c == 4: return v().Abs_Value()
        if c == 5: return v().Max(LV((0.0, 0.0, 0.0)))
        if c == 6: return v().Cross(v())
        if c == 7: return Construct_Vector3(a(), a(), a())
        return (v() + LV((0.1, 0.2, 0.3))).Normalize()


ROOTED_SEEDS = 100000


def make_kind(seed):
    def distance(S, P):
        g = Gen(np.random.default_rng(seed), S, P)
        if seed >= ROOTED_SEEDS:
            # distances that END in a square root (seeds of their own, so that the earlier ones keep their scenes): the forms the
            # hiprtc build compiles with the root culled at wave level (mdh_api.hip: jit_min_form) -- root - R, root + S, S + root, root
            v = (P - S.Get(C_V1)) + g.v(3).Max(LV((-0.3, -0.3, -0.3))).Min(LV((0.3, 0.3, 0.3))) * LF(0.5)
            t = g.f(3).Clamp(LF(-0.4), LF(0.4)) * LF(0.5)
            form = seed % 4
            if form == 0: return v.Length() - (LF(0.8) + t)
            if form == 1: return v.Length() + (t - LF(0.9))
            if form == 2: return (t - LF(0.9)) + v.Length()
            return (v.Dot(v) * LF(0.25) + t * t).Sqrt()
        # a sphere shell keeps the field a sane distance bound far away; the random term deforms it nearby
        return ((P - S.Get(C_V1)).Length() - LF(1.0)) + g.f(4).Clamp(LF(-0.4), LF(0.4)) * LF(0.5)

    def normal(S, P):
        if seed % 2:
            return Forward_Difference(distance(S, Value_Identifier("DX")), "DX", P, 0.0005).Normalize()
        g = Gen(np.random.default_rng(seed + 7), S, P)
        return (g.v(3) + (P - S.Get(C_V1))).Normalize()

    return primitives.Create("Blob%d" % seed, (C_V1, C_V2, C_F1, C_F2, Material_Id), distance, normal, lambda S: S.Get(Material_Id))


def blob(rng, m):
    return entities.Create([(C_V1, values.Vector3(tuple(rng.uniform(1.0, 5.0, 3)))), (C_V2, values.Vector3(tuple(rng.uniform(-1.0, 1.0, 3)))),
                            (C_F1, values.Float(float(rng.uniform(-1.0, 1.0)))), (C_F2, values.Float(float(rng.uniform(0.1, 2.0)))), (Material_Id, values.Int(m))])


def build(seed, binding, jit):
    rng = np.random.default_rng(seed)
    Kind = make_kind(seed)
    part = scenes.Partitioning_Settings(Enable=bool(seed % 3 == 0), Index_Count=8, Grid_Dimensions=(5, 5, 7), Grid_Spacing=(2.0, 2.0, 2.0), Grid_Offset=(-2.0, -2.0, -7.0))
    scene = scenes.Compile([(planes.Plane, 6), (Kind, 3)], [(point_lights.Point_Light, 2)], Partitioning=part)
    R = renderers.Create(windows.Open(24, 16), scene, Probes=SMALL_PROBES, Volumetrics=renderers.No_Volumetrics, Binding=binding)
    R.Set_Option(B.OPT_JIT, jit)
    for m, alb in enumerate(((0.6, 0.6, 0.6), (0.9, 0.2, 0.1), (0.1, 0.2, 0.9))):
        R.Set_Material(m, materials.Create(alb, 0.5 if m == 2 else 0.0, 0.3 if m == 2 else 0.6))
    for (n, o), m in zip((((0, 1, 0), 1.0), ((0, -1, 0), 7.0), ((1, 0, 0), 1.0), ((-1, 0, 0), 7.0), ((0, 0, 1), 6.0), ((0, 0, -1), 7.0)), (0, 0, 1, 2, 0, 0)):
        R.Add_Primitive(planes.Plane, planes.Create(n, o, m))
    for i in range(int(rng.integers(1, 4))):
        R.Add_Primitive(Kind, blob(rng, int(rng.integers(0, 3))))
    R.Set_Light(1, point_lights.Point_Light, point_lights.Create((3.0, 5.0, 1.0), (0.9, 0.9, 0.8)))
    R.Set_Camera_Position((2.5, 2.0, -1.0))
    R.Set_Option(B.OPT_GBUFFER, 1)
    if part.Enable:
        R.Update_Partitioning(int(rng.integers(0, 3)))
    out = {}
    pts = rng.uniform(-1.0, 7.0, (64, 3)).astype(np.float32)
    for ada in (0, 1):  # GLSL division, and Madarch.Values."/" as Exprs.Eval has it
        R.Set_Option(B.OPT_ADA_EVAL_DIV, ada)
        out["eval_d%d" % ada], out["eval_n%d" % ada] = R.Eval_Distances_To(pts, [Kind, planes.Plane])
    out.update(snapshot(R, 2))
    if part.Enable:
        out["partition"] = R.Read_Partitioning()
    return out



def compare(got, want, what=""):
    assert_parity(got, want)
    for k in want:
        if k.startswith("eval") or k == "partition":
            assert same_bits(got[k], want[k]), "%s %s" % (k, what)
