"""Host mirror of GPU_Types (reference madarch/support/gpu_types-base.ads:21-37,
gpu_types-structs.adb:11-38, gpu_types-fixed_arrays.adb:17-39): the std140
size / alignment / offset calculator, and the entity -> element blob writer that
stands for Renderers.Write_Entity (madarch-renderers.adb:335-347)."""
import struct

from . import values


def _pad(x, amount):  # gpu_types.adb:2-8
    while x % amount:
        x += 1
    return x


class Base:
    def __init__(self, alignment, size):
        self.alignment, self.size = alignment, size


Int = Base(4, 4)
IVec_2 = Base(8, 8)
IVec_3 = Base(16, 12)
Float = Base(4, 4)
Vec_2 = Base(8, 8)
Vec_3 = Base(16, 12)


def of_kind(kind):  # Values.GPU_Type, madarch-values.adb:362-372
    return (Vec_3, Float, Int)[kind]


class Struct:
    alignment = 16  # gpu_types-structs.ads:11

    def __init__(self, named_components):
        self.components = list(named_components)  # [(name, type)]

    @property
    def size(self):  # gpu_types-structs.adb:11-21
        total = 0
        for _, t in self.components:
            total = _pad(total, t.alignment) + t.size
        return total

    def offset_of(self, name):  # gpu_types-structs.adb:23-38
        off = 0
        for n, t in self.components:
            off = _pad(off, t.alignment)
            if n == name:
                return off, t
            off += t.size
        raise KeyError(name)


class Fixed_Array:
    alignment = 16  # gpu_types-fixed_arrays.adb:12-15

    def __init__(self, length, component):
        self.length, self.component = length, component

    @property
    def stride(self):
        return _pad(self.component.size, 16)

    @property
    def size(self):  # gpu_types-fixed_arrays.adb:17-24
        return self.stride * self.length

    def offset_of(self, index1):  # gpu_types-fixed_arrays.adb:26-37 (1-based)
        return self.stride * (index1 - 1), self.component


def struct_of_components(comps):
    """Compute_Prim_Struct_Type / Compute_Light_Struct_Type (madarch-scenes.adb:1272-1306)."""
    return Struct([(c.name, of_kind(c.kind)) for c in comps])


def entity_blob(struct_type, entity):
    """Element image of one entity: every component written at its std140 offset
    (Write_Entity + Write_Value, madarch-renderers.adb:323-347)."""
    buf = bytearray(struct_type.size)

    def write(comp, val):
        off, _ = struct_type.offset_of(comp.name)
        if val.kind == values.Vector3_Kind:
            struct.pack_into("<3f", buf, off, *[float(x) for x in val.data])
        elif val.kind == values.Float_Kind:
            struct.pack_into("<f", buf, off, float(val.data))
        else:
            struct.pack_into("<i", buf, off, int(val.data))

    entity.Foreach(write)
    return bytes(buf)
