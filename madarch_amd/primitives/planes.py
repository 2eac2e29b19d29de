"""Madarch.Primitives.Planes (reference madarch/madarch-primitives-planes.ads:10-33)."""
from .. import components, entities, values
from . import Create as _Create
from .materials import Material_Id

Normal = components.Create("normal", values.Vector3_Kind)
Offset = components.Create("offset", values.Float_Kind)

Plane = _Create("Plane", (Normal, Offset, Material_Id))


def Create(Instance_Normal, Instance_Offset, Instance_Material_Id):
    return entities.Create([(Normal, values.Vector3(Instance_Normal)),
                            (Offset, values.Float(Instance_Offset)),
                            (Material_Id, values.Int(Instance_Material_Id))])
