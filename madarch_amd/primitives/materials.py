"""Madarch.Primitives.Materials (reference madarch/madarch-primitives-materials.ads:8)."""
from .. import components, values

Material_Id = components.Create("material_id", values.Int_Kind)
