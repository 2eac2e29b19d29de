"""Madarch.Lights.Spot_Lights (reference madarch/madarch-lights-spot_lights.ads:14-41)."""
from .. import components, entities, values
from . import Create as _Create

Position = components.Create("position", values.Vector3_Kind)
Direction = components.Create("direction", values.Vector3_Kind)
Aperture = components.Create("aperture", values.Float_Kind)
Color = components.Create("color", values.Vector3_Kind)

Spot_Light = _Create("SpotLight", (Position, Direction, Aperture, Color))


def Create(Instance_Position, Instance_Direction, Instance_Aperture, Instance_Color):
    return entities.Create([(Position, values.Vector3(Instance_Position)),
                            (Direction, values.Vector3(Instance_Direction)),
                            (Aperture, values.Float(Instance_Aperture)),
                            (Color, values.Vector3(Instance_Color))])
