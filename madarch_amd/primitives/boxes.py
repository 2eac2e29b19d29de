"""Madarch.Primitives.Boxes (reference madarch/madarch-primitives-boxes.ads:10-31)."""
from .. import components, entities, values
from . import Create as _Create
from .materials import Material_Id

Center = components.Create("center", values.Vector3_Kind)
Side = components.Create("side", values.Vector3_Kind)

Box = _Create("Box", (Center, Side, Material_Id))


def Create(Instance_Center, Instance_Side, Instance_Material_Id):
    return entities.Create([(Center, values.Vector3(Instance_Center)),
                            (Side, values.Vector3(Instance_Side)),
                            (Material_Id, values.Int(Instance_Material_Id))])
