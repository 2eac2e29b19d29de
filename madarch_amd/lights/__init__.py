"""Host mirror of Madarch.Lights (reference madarch/madarch-lights.ads:7-37): a light KIND is a
name, its component list and -- for a user-defined kind -- the two expression builders of
Lights.Create (madarch-lights.ads:20-24):
    Sample   (L : Struct_Expr; Pos, Normal, Dir, Dist : Expr) -> Expr      radiance (vector)
    Position (L : Struct_Expr)                                  -> Expr      vector
PointLight and SpotLight (the library's own kind objects, which bring no expressions) are hand-written device code; any other kind, whatever its name, is
compiled to MDH_X programs (madarch_amd/exprs.py) that the kernels interpret."""
from .. import exprs, values

BUILT_IN = ("PointLight", "SpotLight")


class Light:
    def __init__(self, name, comps, sample=None, position=None):
        self.name = name
        self.comps = list(comps)
        self.sample, self.position = sample, position

    def __repr__(self):
        return "Light(%r)" % self.name

    def is_user_defined(self):
        # by content, not by name: a kind that brings expressions runs them even if it is called "PointLight"
        return bool(self.sample or self.position) or self.name not in BUILT_IN

    # Get_Sample_Expr / Get_Position_Expr (madarch-lights.ads:31-37)
    def Get_Sample_Expr(self, Inst, Pos, Normal, Dir, Dist):
        return self.sample(Inst, Pos, Normal, Dir, Dist)

    def Get_Position_Expr(self, Inst):
        return self.position(Inst)

    def programs(self):
        """(sample, position) as MDH_X words; names as in the generated sample_<Light> (scenes.adb:500-516)."""
        L = exprs.Struct_Identifier("l")
        pos, normal, d, dist = (exprs.Value_Identifier(n) for n in ("pos", "normal", "dir", "dist"))
        V, F = values.Vector3_Kind, values.Float_Kind
        return (exprs.compile_program(self.sample(L, pos, normal, d, dist), self.comps, V,
                                      args=[("pos", V, 0), ("normal", V, 3), ("dir", V, 6), ("dist", F, 9)]),
                exprs.compile_program(self.position(L), self.comps, V))


def Create(Name, Comps, Sample=None, Position=None):
    return Light(Name, Comps, Sample, Position)


def Get_Name(l):
    return l.name


def Get_Components(l):
    return list(l.comps)


from . import point_lights, spot_lights  # noqa: E402,F401
