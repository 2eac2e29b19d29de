"""Host mirror of Madarch.Primitives (reference madarch/madarch-primitives.ads:13-60).

A primitive KIND is a name, its component list and -- for a user-defined kind -- the three
expression builders of Primitives.Create (madarch-primitives.ads:24-30):
    Distance (S : Struct_Expr; P : Expr) -> Expr      float
    Normal   (S : Struct_Expr; P : Expr) -> Expr      vector
    Material (S : Struct_Expr)           -> Expr      int
The four built-in kinds (Sphere, Plane, Box, Triangle) are hand-written device functions in
madarch_amd/csrc, taken by the library's own kind objects of those names (which bring no expressions); any other kind -- also one that merely shares such a name -- is compiled from its expressions
to the MDH_X register programs of include/madarch_hip.h (madarch_amd/exprs.py) that the kernels
interpret -- the analogue of the GLSL the reference generates from the same trees.
"""
from .. import exprs, values

BUILT_IN = ("Sphere", "Plane", "Box", "Triangle")


class Primitive:
    def __init__(self, name, comps, distance=None, normal=None, material=None):
        self.name = name
        self.comps = list(comps)
        self.distance, self.normal, self.material = distance, normal, material

    def __repr__(self):
        return "Primitive(%r)" % self.name

    def is_user_defined(self):
        # by content, not by name: a kind that brings expressions runs them even if it is called "Sphere"
        return bool(self.distance or self.normal or self.material) or self.name not in BUILT_IN

    # Get_Dist_Expr / Get_Normal_Expr / Get_Material_Expr (madarch-primitives.ads:37-50)
    def Get_Dist_Expr(self, Inst, Point):
        return self.distance(Inst, Point)

    def Get_Normal_Expr(self, Inst, Point):
        return self.normal(Inst, Point)

    def Get_Material_Expr(self, Inst):
        return self.material(Inst)

    def programs(self):
        """(distance, normal, material) as MDH_X words."""
        if not (self.distance and self.normal and self.material):
            raise exprs.Unsupported_Expr("kind %r is not built in and has no Distance / Normal / Material" % self.name)
        S, P = exprs.Struct_Identifier("prim"), exprs.Value_Identifier("x")
        return (exprs.compile_program(self.distance(S, P), self.comps, values.Float_Kind, "x"),
                exprs.compile_program(self.normal(S, P), self.comps, values.Vector3_Kind, "x"),
                exprs.compile_program(self.material(S), self.comps, values.Int_Kind))


def Create(Name, Comps, Distance=None, Normal=None, Material=None):
    return Primitive(Name, Comps, Distance, Normal, Material)


def Get_Name(p):
    return p.name


def Get_Components(p):
    return list(p.comps)


from . import materials, spheres, planes, boxes, triangles  # noqa: E402,F401
