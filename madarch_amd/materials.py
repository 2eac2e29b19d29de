"""Host mirror of Madarch.Materials (reference madarch/madarch-materials.ads:10-25)."""
from . import components, entities, values

Albedo = components.Create("albedo", values.Vector3_Kind)
Metallic = components.Create("metallic", values.Float_Kind)
Roughness = components.Create("roughness", values.Float_Kind)


def Create(Instance_Albedo, Instance_Metallic, Instance_Roughness):
    return entities.Create([(Albedo, values.Vector3(Instance_Albedo)),
                            (Metallic, values.Float(Instance_Metallic)),
                            (Roughness, values.Float(Instance_Roughness))])
