"""Host mirror of Madarch.Scenes (reference madarch/madarch-scenes.ads:13-76).

`Compile` keeps the reference's arguments.  Where the reference emits GLSL text
for the scene (madarch-scenes.adb:1189-1266), this back end has the built-in
kinds as hand-written HIP device functions, so a compiled scene is its
description (kinds, declared counts, partitioning, max distance) plus the std140
layout of its uniform block (Compute_Scene_GPU_Type, scenes.adb:1268-1345).
"""
import ctypes as C

from . import _binding as B
from . import gpu_types

Clamp, Fallback = 0, 1  # Partitioning_Border_Behavior, scenes.ads:28
Split, Unify = 0, 1     # Codegen_Loop_Strategy, scenes.ads:45


class Partitioning_Settings:  # scenes.ads:30-41
    def __init__(self, Enable=True, Index_Count=20, Border_Behavior=Clamp,
                 Grid_Dimensions=(10, 10, 20), Grid_Spacing=(1.0, 1.0, 1.0),
                 Grid_Offset=(-1.5, -1.5, -10.0)):
        self.Enable = bool(Enable)
        self.Index_Count = Index_Count
        self.Border_Behavior = Border_Behavior
        self.Grid_Dimensions = tuple(Grid_Dimensions)
        self.Grid_Spacing = tuple(Grid_Spacing)
        self.Grid_Offset = tuple(Grid_Offset)


Default_Partitioning_Settings = Partitioning_Settings()


class Scene:
    def __init__(self, prims, lights, partitioning, max_dist, loop_strategy):
        self.Prims_Count = list(prims)    # [(Primitive, declared count)]
        self.Lights_Count = list(lights)  # [(Light, declared count)]
        self.Partitioning_Config = partitioning
        self.Max_Dist = float(max_dist)
        self.Loop_Strategy = loop_strategy
        # Compute_Scene_GPU_Type (scenes.adb:1268-1345)
        comps = []
        self._prim_struct, self._light_struct = {}, {}
        for p, n in self.Prims_Count:
            st = gpu_types.struct_of_components(p.comps)
            self._prim_struct[p] = st
            comps.append(("prim_%s_count" % p.name, gpu_types.Int))
            comps.append(("prim_%s_array" % p.name, gpu_types.Fixed_Array(n, st)))
        for l, n in self.Lights_Count:
            st = gpu_types.struct_of_components(l.comps)
            self._light_struct[l] = st
            comps.append(("light_%s_count" % l.name, gpu_types.Int))
            comps.append(("light_%s_array" % l.name, gpu_types.Fixed_Array(n, st)))
        comps.append(("total_light_count", gpu_types.Int))
        self.GPU_Type = gpu_types.Struct(comps)

    # the ctypes description handed to <prefix>create
    def _desc(self):
        keep = []

        def decls(items):
            arr = (B.mdh_kind_decl * max(1, len(items)))()
            for i, (k, n) in enumerate(items):
                cs = (B.mdh_component * len(k.comps))()
                for j, c in enumerate(k.comps):
                    cs[j].name = c.name.encode()
                    cs[j].kind = c.kind
                keep.append(cs)
                arr[i].name = k.name.encode()
                arr[i].max_count = n
                arr[i].n_components = len(k.comps)
                arr[i].components = cs
                user = getattr(k, "is_user_defined", None) and k.is_user_defined()
                if user and (getattr(k, "sample", None) and k.position if hasattr(k, "sample") else k.distance and k.normal and k.material):
                    # the kind's expressions as MDH_X programs (the analogue of To_GLSL, scenes.adb:1189-1266);
                    # a light's Sample and Position travel in the dist / normal fields
                    for field, words in zip(("dist", "normal", "material"), k.programs()):
                        code = (C.c_int32 * max(1, len(words)))(*words)
                        keep.append(code)
                        setattr(arr[i], field + "_code", code)
                        setattr(arr[i], field + "_len", len(words))
            keep.append(arr)
            return arr

        d = B.mdh_scene_desc()
        d.n_prim_kinds = len(self.Prims_Count)
        d.prim_kinds = decls(self.Prims_Count)
        d.n_light_kinds = len(self.Lights_Count)
        d.light_kinds = decls(self.Lights_Count)
        p = self.Partitioning_Config
        d.partitioning.enable = 1 if p.Enable else 0
        d.partitioning.index_count = p.Index_Count
        d.partitioning.border_behavior = p.Border_Behavior
        d.partitioning.grid_dimensions = (C.c_int32 * 3)(*p.Grid_Dimensions)
        d.partitioning.grid_spacing = (C.c_float * 3)(*p.Grid_Spacing)
        d.partitioning.grid_offset = (C.c_float * 3)(*p.Grid_Offset)
        d.max_dist = self.Max_Dist
        d.loop_strategy = self.Loop_Strategy
        return d, keep

    def prim_kind_index(self, prim):
        for i, (p, _) in enumerate(self.Prims_Count):
            if p is prim:
                return i
        raise KeyError(prim)

    def light_kind_index(self, light):
        for i, (l, _) in enumerate(self.Lights_Count):
            if l is light:
                return i
        raise KeyError(light)


def Compile(All_Primitives, All_Lights, Partitioning=None, Max_Dist=20.0,
            Loop_Strategy=Unify, Print_GLSL=False):  # scenes.ads:47-53
    if Partitioning is None:
        Partitioning = Default_Partitioning_Settings
    return Scene(All_Primitives, All_Lights, Partitioning, Max_Dist, Loop_Strategy)


def Get_GPU_Type(S):  # scenes.ads:57
    return S.GPU_Type


def Get_Partitioning_Settings(S):  # scenes.ads:59
    return S.Partitioning_Config


def Get_Primitives_Location(S, Prim):  # scenes.adb:1435-1446 -> (array offset, count offset)
    a, _ = S.GPU_Type.offset_of("prim_%s_array" % Prim.name)
    c, _ = S.GPU_Type.offset_of("prim_%s_count" % Prim.name)
    return a, c


def Get_Lights_Location(S, Lit):  # scenes.adb:1448-1462 -> (array, count, total)
    a, _ = S.GPU_Type.offset_of("light_%s_array" % Lit.name)
    c, _ = S.GPU_Type.offset_of("light_%s_count" % Lit.name)
    t, _ = S.GPU_Type.offset_of("total_light_count")
    return a, c, t


def Get_Primitives(S):  # scenes.ads:76
    return [p for p, _ in S.Prims_Count]
