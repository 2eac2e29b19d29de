"""Madarch.Primitives.Triangles (reference madarch/madarch-primitives-triangles.ads:10-35)."""
from .. import components, entities, values
from . import Create as _Create
from .materials import Material_Id

V1 = components.Create("v1", values.Vector3_Kind)
V2 = components.Create("v2", values.Vector3_Kind)
V3 = components.Create("v3", values.Vector3_Kind)

Triangle = _Create("Triangle", (V1, V2, V3, Material_Id))


def Create(Instance_V1, Instance_V2, Instance_V3, Instance_Material_Id):
    return entities.Create([(V1, values.Vector3(Instance_V1)),
                            (V2, values.Vector3(Instance_V2)),
                            (V3, values.Vector3(Instance_V3)),
                            (Material_Id, values.Int(Instance_Material_Id))])
