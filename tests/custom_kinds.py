"""User-defined primitive kinds for the tests: the four built-in kinds restated with the reference's
own expressions (madarch-primitives-{spheres,planes}.ads, -{boxes,triangles}.adb) under other names,
and two kinds the reference does not have (a torus and a capsule)."""
from madarch_amd import components, entities, exprs, lights, primitives, values
from madarch_amd.exprs import (Construct_Vector3, Forward_Difference, If_Then_Else, Let_In, Literal, Min, Value_Identifier)
from madarch_amd.primitives.materials import Material_Id

V3K, FK = values.Vector3_Kind, values.Float_Kind
X, Y, Z = 0, 1, 2


def _material(S):  # Materials.Get_Material_Id (madarch-primitives-materials.ads)
    return S.Get(Material_Id)


# ---- Sphere (spheres.ads:13-17)
S_Center = components.Create("center", V3K)
S_Radius = components.Create("radius", FK)
My_Sphere = primitives.Create(
    "MySphere", (S_Center, S_Radius, Material_Id),
    lambda S, P: (S.Get(S_Center) - P).Length() - S.Get(S_Radius),
    lambda S, P: (P - S.Get(S_Center)).Normalize(), _material)


def sphere(c, r, m):
    return entities.Create([(S_Center, values.Vector3(c)), (S_Radius, values.Float(r)), (Material_Id, values.Int(m))])


# ---- Plane (planes.ads:13-17)
P_Normal = components.Create("normal", V3K)
P_Offset = components.Create("offset", FK)
My_Plane = primitives.Create(
    "MyPlane", (P_Normal, P_Offset, Material_Id),
    lambda S, P: S.Get(P_Normal).Dot(P) + S.Get(P_Offset),
    lambda S, P: S.Get(P_Normal), _material)


def plane(n, o, m):
    return entities.Create([(P_Normal, values.Vector3(n)), (P_Offset, values.Float(o)), (Material_Id, values.Int(m))])


# ---- Box (boxes.adb:7-41)
B_Center = components.Create("center", V3K)
B_Side = components.Create("side", V3K)
_ZV, _ZF, _E = Literal(values.Vector3((0.0, 0.0, 0.0))), Literal(values.Float(0.0)), Literal(values.Float(0.002))


def _box_distance(S, P):
    Q = Value_Identifier("q")
    return ((S.Get(B_Center) - P).Abs_Value() - S.Get(B_Side)).Let_In(
        V3K, "q", Q.Max(_ZV).Length() + Q.Get(X).Max(Q.Get(Y).Max(Q.Get(Z))).Min(_ZF))


def _box_normal(S, P):
    D, RX, RY, RZ = (Value_Identifier(n) for n in ("d", "rx", "ry", "rz"))
    nd = Construct_Vector3((RX > RY - _E).To_Float() * (RX > RZ - _E).To_Float() * D.Get(X).Sign(),
                           (RY > RX - _E).To_Float() * (RY > RZ - _E).To_Float() * D.Get(Y).Sign(),
                           (RZ > RX - _E).To_Float() * (RZ > RY - _E).To_Float() * D.Get(Z).Sign())
    return ((P - S.Get(B_Center)) / S.Get(B_Side)).Let_In(
        V3K, "d", D.Get(X).Abs_Value().Let_In(FK, "rx", D.Get(Y).Abs_Value().Let_In(FK, "ry", D.Get(Z).Abs_Value().Let_In(FK, "rz", nd.Normalize()))))


My_Box = primitives.Create("MyBox", (B_Center, B_Side, Material_Id), _box_distance, _box_normal, _material)


def box(c, s, m):
    return entities.Create([(B_Center, values.Vector3(c)), (B_Side, values.Vector3(s)), (Material_Id, values.Int(m))])


# ---- Triangle (triangles.adb:16-56)
T_V1, T_V2, T_V3 = (components.Create(n, V3K) for n in ("v1", "v2", "v3"))
_F0, _F1, _F2 = (Literal(values.Float(v)) for v in (0.0, 1.0, 2.0))


def _tri_distance(S, P):
    V21, V32, V13, P1, P2, P3, Nor = (Value_Identifier(n) for n in ("V21", "V32", "V13", "P1", "P2", "P3", "Nor"))
    cond = (V21.Cross(Nor).Dot(P1).Sign() + V32.Cross(Nor).Dot(P2).Sign() + V13.Cross(Nor).Dot(P3).Sign()) < _F2
    thn = Min((V21 * (V21.Dot(P1) / V21.Dot2()).Clamp(_F0, _F1) - P1).Dot2(),
              (V32 * (V32.Dot(P2) / V32.Dot2()).Clamp(_F0, _F1) - P2).Dot2(),
              (V13 * (V13.Dot(P3) / V13.Dot2()).Clamp(_F0, _F1) - P3).Dot2())
    els = Nor.Dot(P1) * Nor.Dot(P1) / Nor.Dot2()
    return Let_In([(V3K, "V21", S.Get(T_V2) - S.Get(T_V1)), (V3K, "V32", S.Get(T_V3) - S.Get(T_V2)), (V3K, "V13", S.Get(T_V1) - S.Get(T_V3)),
                   (V3K, "P1", P - S.Get(T_V1)), (V3K, "P2", P - S.Get(T_V2)), (V3K, "P3", P - S.Get(T_V3)), (V3K, "Nor", V21.Cross(V13))],
                  If_Then_Else(cond, thn, els).Sqrt())


def _tri_normal(S, P):
    return Forward_Difference(_tri_distance(S, Value_Identifier("DX")), "DX", P).Normalize()


My_Triangle = primitives.Create("MyTriangle", (T_V1, T_V2, T_V3, Material_Id), _tri_distance, _tri_normal, _material)


def triangle(a, b, c, m):
    return entities.Create([(T_V1, values.Vector3(a)), (T_V2, values.Vector3(b)), (T_V3, values.Vector3(c)), (Material_Id, values.Int(m))])


# ---- kinds the reference does not have
# torus around the y axis: length (vec2 (length (p.xz) - R, p.y)) - r, written with vec3s
R_Center = components.Create("center", V3K)
R_Major = components.Create("major", FK)
R_Minor = components.Create("minor", FK)


def _torus_distance(S, P):
    D = Value_Identifier("d")
    ring = Construct_Vector3(D.Get(X), _ZF, D.Get(Z)).Length() - S.Get(R_Major)
    return (P - S.Get(R_Center)).Let_In(V3K, "d", Construct_Vector3(ring, D.Get(Y), _ZF).Length() - S.Get(R_Minor))


Torus = primitives.Create(
    "Torus", (R_Center, R_Major, R_Minor, Material_Id), _torus_distance,
    lambda S, P: Forward_Difference(_torus_distance(S, Value_Identifier("DX")), "DX", P, 0.0005).Normalize(), _material)


def torus(c, major, minor, m):
    return entities.Create([(R_Center, values.Vector3(c)), (R_Major, values.Float(major)), (R_Minor, values.Float(minor)), (Material_Id, values.Int(m))])


# capsule between a and b: length (pa - ba clamp (dot (pa, ba) / dot2 (ba), 0, 1)) - r; the material comes from an
# expression (two ids chosen by the radius) to exercise the Material program
C_A, C_B = components.Create("a", V3K), components.Create("b", V3K)
C_Radius = components.Create("radius", FK)
C_Mat_Thin, C_Mat_Thick = components.Create("mat_thin", values.Int_Kind), components.Create("mat_thick", values.Int_Kind)


def _capsule_distance(S, P):
    PA, BA = Value_Identifier("pa"), Value_Identifier("ba")
    h = (PA.Dot(BA) / BA.Dot2()).Clamp(_F0, _F1)
    return Let_In([(V3K, "pa", P - S.Get(C_A)), (V3K, "ba", S.Get(C_B) - S.Get(C_A))], (PA - BA * h).Length() - S.Get(C_Radius))


Capsule = primitives.Create(
    "Capsule", (C_A, C_B, C_Radius, C_Mat_Thin, C_Mat_Thick), _capsule_distance,
    lambda S, P: Forward_Difference(_capsule_distance(S, Value_Identifier("DX")), "DX", P, 0.0005).Normalize(),
    lambda S: If_Then_Else(S.Get(C_Radius) < Literal(values.Float(0.3)), S.Get(C_Mat_Thin), S.Get(C_Mat_Thick)))


def capsule(a, b, r, thin, thick):
    return entities.Create([(C_A, values.Vector3(a)), (C_B, values.Vector3(b)), (C_Radius, values.Float(r)),
                            (C_Mat_Thin, values.Int(thin)), (C_Mat_Thick, values.Int(thick))])


# a rippled slab that exercises the trigonometric builtins: 0.4 (p.y - h - a sin (f p.x) cos (f p.z)) plus terms
# with tan, asin and atan of bounded arguments (the factor keeps the field a conservative bound)
W_Height, W_Amp, W_Freq = components.Create("height", FK), components.Create("amp", FK), components.Create("freq", FK)


def _ripple_distance(S, P):
    f = S.Get(W_Freq)
    wave = (P.Get(X) * f).Sin() * (P.Get(Z) * f).Cos()
    extra = ((P.Get(X) * Literal(values.Float(0.1))).Tan() + (wave * Literal(values.Float(0.5))).Asin() + P.Get(Z).Atan()) * Literal(values.Float(0.01))
    return (P.Get(Y) - S.Get(W_Height) - S.Get(W_Amp) * wave - extra) * Literal(values.Float(0.4))


Ripple = primitives.Create(
    "Ripple", (W_Height, W_Amp, W_Freq, Material_Id), _ripple_distance,
    lambda S, P: Forward_Difference(_ripple_distance(S, Value_Identifier("DX")), "DX", P, 0.0005).Normalize(), _material)


def ripple(height, amp, freq, m):
    return entities.Create([(W_Height, values.Float(height)), (W_Amp, values.Float(amp)), (W_Freq, values.Float(freq)), (Material_Id, values.Int(m))])


# ---- lights: PointLight and SpotLight restated (point_lights.ads:20-22, spot_lights.adb:5-24), and a lamp of our own
L_Position, L_Color = components.Create("position", V3K), components.Create("color", V3K)
L_Direction, L_Aperture = components.Create("direction", V3K), components.Create("aperture", FK)
_F15, _F003, _F8 = Literal(values.Float(1.5)), Literal(values.Float(0.03)), Literal(values.Float(8.0))

My_Point_Light = lights.Create(
    "MyPointLight", (L_Position, L_Color),
    lambda L, Pos, Normal, Dir, Dist: L.Get(L_Color) / (Dist * Dist * _F003), lambda L: L.Get(L_Position))


def _spot_sample(L, Pos, Normal, Dir, Dist):
    attenuation = _F1 / (Dist * Dist * _F003)
    theta = (-Dir).Dot(L.Get(L_Direction)).Max(_F0).Acos()
    ratio = (theta / L.Get(L_Aperture)).Clamp(_F0, _F1)
    return L.Get(L_Color) * attenuation.Min(_F15) * (_F1 - ratio ** _F8)


My_Spot_Light = lights.Create("MySpotLight", (L_Position, L_Direction, L_Aperture, L_Color), _spot_sample, lambda L: L.Get(L_Position))


def point_light(p, c):
    return entities.Create([(L_Position, values.Vector3(p)), (L_Color, values.Vector3(c))])


def spot_light(p, d, a, c):
    return entities.Create([(L_Position, values.Vector3(p)), (L_Direction, values.Vector3(d)), (L_Aperture, values.Float(a)), (L_Color, values.Vector3(c))])


# a lamp that hangs `drop` below its anchor, with a linear falloff and a term in the surface normal
M_Anchor, M_Drop, M_Power = components.Create("anchor", V3K), components.Create("drop", FK), components.Create("power", V3K)
Lamp = lights.Create(
    "Lamp", (M_Anchor, M_Drop, M_Power),
    lambda L, Pos, Normal, Dir, Dist: L.Get(M_Power) * (Normal.Dot(Dir).Max(Literal(values.Float(0.25))) / (_F1 + Dist)),
    lambda L: L.Get(M_Anchor) - Construct_Vector3(_F0, L.Get(M_Drop), _F0))


def lamp(anchor, drop, power):
    return entities.Create([(M_Anchor, values.Vector3(anchor)), (M_Drop, values.Float(drop)), (M_Power, values.Vector3(power))])
