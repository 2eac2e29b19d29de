"""Host mirror of Madarch.Primitives (reference madarch/madarch-primitives.ads:13-60).

A primitive KIND is a name plus its component list.  The reference also carries
the distance/normal/material expression builders; the MI355X back end has the
four built-in kinds as hand-written device functions (madarch_amd/csrc), so a
kind here is identified by its name: Sphere, Plane, Box, Triangle.
"""


class Primitive:
    def __init__(self, name, comps):
        self.name = name
        self.comps = list(comps)

    def __repr__(self):
        return "Primitive(%r)" % self.name


def Create(Name, Comps):
    return Primitive(Name, Comps)


def Get_Name(p):
    return p.name


def Get_Components(p):
    return list(p.comps)


from . import materials, spheres, planes, boxes, triangles  # noqa: E402,F401
