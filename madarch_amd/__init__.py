"""madarch_amd -- host side of the MI355X-native Madarch renderer.

Mirrors the reference's Ada packages (Values, Components, Entities, Materials,
Primitives.*, Lights.*, Scenes, Renderers, Windows) over the C ABI of
include/madarch_hip.h.  The device code lives in csrc/ (hand-written HIP for
gfx950); importing this package does not load it, creating a Renderer does.
"""
from . import components, entities, gpu_types, lights, materials, primitives, renderers, scenes, values, windows  # noqa: F401
from ._binding import MadarchError, hip_binding  # noqa: F401

__all__ = ["components", "entities", "gpu_types", "lights", "materials", "primitives", "renderers",
           "scenes", "values", "windows", "MadarchError", "hip_binding"]
