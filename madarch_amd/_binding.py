"""ctypes view of the C ABI declared in include/madarch_hip.h.

A `Binding` is (shared library, symbol prefix).  The product binding is
`hip_binding()`: libmadarch_hip.so with the prefix ``mdh_``; it raises when the
library is missing -- there is no CPU fallback anywhere in this package.  Tests
build a second Binding over the CPU oracle (prefix ``orc_``, see
tests/oracle_engine.py) to drive the very same host code against it.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# MADARCH_HIP_LIBRARY selects another build of the same library (kernel A/B runs)
HIP_LIBRARY = os.environ.get("MADARCH_HIP_LIBRARY") or os.path.join(_HERE, "csrc", "libmadarch_hip.so")

# status codes (include/madarch_hip.h)
MDH_OK, MDH_E_INVALID, MDH_E_PROBE_MISMATCH, MDH_E_UNSUPPORTED_KIND = 0, 1, 2, 3
MDH_E_INDEX, MDH_E_DEVICE, MDH_E_NO_DEVICE, MDH_E_STATE, MDH_E_COMM = 4, 5, 6, 7, 8

MDH_VEC3, MDH_FLOAT, MDH_INT = 0, 1, 2

OPT_ATLAS_FORMAT, OPT_SCREEN_MODE, OPT_AO_STEPS, OPT_GBUFFER = 0, 1, 2, 3
OPT_RANK, OPT_WORLD, OPT_TIMING, OPT_ADA_EVAL_DIV, OPT_FRAME_OVERLAP, OPT_JIT, OPT_IRRADIANCE_ALL, OPT_WINDOW = 4, 5, 6, 7, 8, 9, 10, 11
OPT_INDIRECT_SPECULAR, OPT_HYSTERESIS_PERMILLE, OPT_RADIANCE_ORDER, OPT_SCREEN_ORDER, OPT_NUMERICS, OPT_RADIANCE_MIPS, OPT_SCREEN_SPLIT = 12, 13, 14, 15, 16, 17, 18

PASS_RADIANCE, PASS_IRRADIANCE, PASS_VISIBILITY, PASS_SCATTERING, PASS_SCREEN, PASS_EXCHANGE = range(6)
PASS_NAMES = ("radiance", "irradiance", "visibility", "scattering", "screen", "exchange")
COMM_ID_BYTES = 128
PEER_BLOB_BYTES = 512
TEX_RADIANCE, TEX_IRRADIANCE, TEX_VISIBILITY, TEX_SCATTERING = range(4)
TEX_RADIANCE_MIP0 = 16  # + l: level l >= 1 of the radiance atlas (OPT_RADIANCE_MIPS), Read_Texture only


class mdh_component(C.Structure):
    _fields_ = [("name", C.c_char_p), ("kind", C.c_int32)]


class mdh_kind_decl(C.Structure):
    _fields_ = [("name", C.c_char_p), ("max_count", C.c_int32), ("n_components", C.c_int32),
                ("components", C.POINTER(mdh_component)),
                ("dist_code", C.POINTER(C.c_int32)), ("dist_len", C.c_int32),
                ("normal_code", C.POINTER(C.c_int32)), ("normal_len", C.c_int32),
                ("material_code", C.POINTER(C.c_int32)), ("material_len", C.c_int32)]


class mdh_partitioning(C.Structure):
    _fields_ = [("enable", C.c_int32), ("index_count", C.c_int32), ("border_behavior", C.c_int32),
                ("grid_dimensions", C.c_int32 * 3), ("grid_spacing", C.c_float * 3),
                ("grid_offset", C.c_float * 3)]


class mdh_probe_settings(C.Structure):
    _fields_ = [("radiance_resolution", C.c_int32), ("irradiance_resolution", C.c_int32),
                ("probe_count", C.c_int32 * 2), ("grid_dimensions", C.c_int32 * 3),
                ("grid_spacing", C.c_float * 3)]


class mdh_volumetrics(C.Structure):
    _fields_ = [("enabled", C.c_int32), ("visibility_resolution", C.c_int32 * 3),
                ("visibility_step_size", C.c_float), ("scattering_resolution", C.c_int32 * 2),
                ("scattering_step_size", C.c_float)]


class mdh_scene_desc(C.Structure):
    _fields_ = [("n_prim_kinds", C.c_int32), ("prim_kinds", C.POINTER(mdh_kind_decl)),
                ("n_light_kinds", C.c_int32), ("light_kinds", C.POINTER(mdh_kind_decl)),
                ("partitioning", mdh_partitioning), ("max_dist", C.c_float),
                ("loop_strategy", C.c_int32)]


class MadarchError(RuntimeError):
    """Raised where the Ada body raises Program_Error / Constraint_Error."""

    def __init__(self, status, message):
        super().__init__("status %d: %s" % (status, message))
        self.status = status


_P = C.c_void_p
_I = C.c_int32
_F = C.c_float
_PF = C.POINTER(C.c_float)
_PI = C.POINTER(C.c_int32)

# name -> (restype, argtypes); exactly the exports of include/madarch_hip.h
ABI = {
    "create": (_I, [_I, _I, C.POINTER(mdh_scene_desc), C.POINTER(mdh_probe_settings),
                    C.POINTER(mdh_volumetrics), _I, C.POINTER(_P)]),
    "destroy": (_I, [_P]),
    "set_option": (_I, [_P, _I, _I]),
    "get_option": (_I, [_P, _I, _PI]),
    "set_material": (_I, [_P, _I, _PF, _F, _F]),
    "add_material": (_I, [_P, _PF, _F, _F, _PI]),
    "set_primitive": (_I, [_P, _I, _I, _P, _I]),
    "add_primitive": (_I, [_P, _I, _P, _I, _PI]),
    "set_light": (_I, [_P, _I, _I, _P, _I]),
    "set_camera_position": (_I, [_P, _PF]),
    "set_camera_orientation": (_I, [_P, _PF]),
    "update_partitioning": (_I, [_P, _I]),
    "render": (_I, [_P]),
    "render_pass": (_I, [_P, _I]),
    "frame_begin": (_I, [_P]),
    "frame_probe_pass": (_I, [_P, _I]),
    "frame_end": (_I, [_P]),
    "finish": (_I, [_P]),
    "read_framebuffer": (_I, [_P, _P]),
    "swap_buffers": (_I, [_P]),
    "front_buffer": (_I, [_P, C.POINTER(_P), C.POINTER(C.c_int64)]),
    "read_gbuffer": (_I, [_P, _P, _P, _P]),
    "read_texture": (_I, [_P, _I, _P, _PI, _PI, _PI]),
    "write_texture": (_I, [_P, _I, _P, _I, _I, _I]),
    "read_atlas_slice": (_I, [_P, _I, _I, _I, _P]),
    "write_atlas_slice": (_I, [_P, _I, _I, _I, _P]),
    "eval_distance_to": (_I, [_P, _I, _P, _P, _I, _P, _P]),
    "pass_time": (_I, [_P, _I, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "reset_pass_times": (_I, [_P]),
    "scene_layout": (_I, [_P, _I, _I, _PI, _PI, _PI, _PI]),
    "scene_buffer_size": (_I, [_P, _PI, _PI]),
    "read_scene_buffer": (_I, [_P, _P, _I]),
    "read_partitioning": (_I, [_P, _P, _I]),
    "partition_warnings": (_I, [_P]),
    "last_error": (C.c_char_p, []),
    "version": (C.c_char_p, []),
}
# exports only the HIP library has: the communicator of a sharded run (RCCL inside the library), and
# device pointers / streams for callers that bring an exchange of their own
HIP_ONLY_ABI = {
    "frame_exchange": (_I, [_P, _I]),
    "comm_unique_id": (_I, [_P]),
    "comm_available": (_I, []),
    "comm_init": (_I, [_P, _P, _I, _I]),
    "comm_destroy": (_I, [_P]),
    "comm_abort": (_I, [_P]),
    "comm_barrier": (_I, [_P]),
    "comm_max_f64": (_I, [_P, C.POINTER(C.c_double)]),
    "comm_reduce_framebuffer": (_I, [_P, _I]),
    # the peer exchange: the same sharded frame with device-to-device copies between processes of one node
    "peer_export": (_I, [_P, _P]),
    "peer_init": (_I, [_P, _P, _I, _I]),
    "atlas_device_ptr": (_I, [_P, _I, C.POINTER(_P), C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                              C.POINTER(C.c_int64)]),
    "stream": (_I, [_P, C.POINTER(_P)]),
    "set_stream": (_I, [_P, _P]),
    "probe_stream": (_I, [_P, C.POINTER(_P)]),
}


class Binding:
    def __init__(self, lib, prefix, extra=None):
        self.lib = lib
        self.prefix = prefix
        table = dict(ABI)
        if extra:
            table.update(extra)
        for name, (res, args) in table.items():
            fn = getattr(lib, prefix + name)  # AttributeError = missing export: loud
            fn.restype = res
            fn.argtypes = args
            setattr(self, name, fn)

    def check(self, status):
        if status != MDH_OK:
            raise MadarchError(status, (self.last_error() or b"").decode("utf-8", "replace"))


_hip = None


def hip_binding():
    """The product binding.  Fails loudly when the HIP library is not built."""
    global _hip
    if _hip is None:
        if not os.path.exists(HIP_LIBRARY):
            raise ImportError(
                "libmadarch_hip.so is not built (%s): run `python -c 'import __graft_entry__ as g; "
                "g.build()'` or `make -C madarch_amd/csrc`; madarch_amd has no CPU fallback" % HIP_LIBRARY)
        _hip = Binding(C.CDLL(HIP_LIBRARY), "mdh_", HIP_ONLY_ABI)
    return _hip
