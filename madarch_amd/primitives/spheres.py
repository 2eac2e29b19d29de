"""Madarch.Primitives.Spheres (reference madarch/madarch-primitives-spheres.ads:10-33)."""
from .. import components, entities, values
from . import Create as _Create
from .materials import Material_Id

Center = components.Create("center", values.Vector3_Kind)
Radius = components.Create("radius", values.Float_Kind)

Sphere = _Create("Sphere", (Center, Radius, Material_Id))


def Create(Instance_Center, Instance_Radius, Instance_Material_Id):
    return entities.Create([(Center, values.Vector3(Instance_Center)),
                            (Radius, values.Float(Instance_Radius)),
                            (Material_Id, values.Int(Instance_Material_Id))])
