"""Madarch.Lights.Point_Lights (reference madarch/madarch-lights-point_lights.ads:14-35)."""
from .. import components, entities, values
from . import Create as _Create

Position = components.Create("position", values.Vector3_Kind)
Color = components.Create("color", values.Vector3_Kind)

Point_Light = _Create("PointLight", (Position, Color))


def Create(Instance_Position, Instance_Color):
    return entities.Create([(Position, values.Vector3(Instance_Position)),
                            (Color, values.Vector3(Instance_Color))])
