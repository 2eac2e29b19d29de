"""Host mirror of Madarch.Renderers (reference madarch/madarch-renderers.ads:21-97).

Same operations, names and argument meaning as the Ada package; the body calls
the C ABI of include/madarch_hip.h where the reference's body calls OpenGL.
Ada exceptions become `MadarchError` (status in `.status`).  Everything past
`Update_Partitioning` below is what the headless build adds in place of the
window: frame read-back, DDGI state access, per-pass control and timing.
"""
import ctypes as C

import numpy as np

from . import _binding as B
from . import gpu_types, materials, scenes

CPU_Best, CPU_Fast, GPU_Fast = 0, 1, 2  # Partitioning_Update_Method, renderers.ads:93


class Probe_Settings:  # renderers.ads:23-29
    def __init__(self, Radiance_Resolution=32, Irradiance_Resolution=8, Probe_Count=(6, 6),
                 Grid_Dimensions=(4, 3, 3), Grid_Spacing=(2.0, 3.0, 3.0)):
        self.Radiance_Resolution = Radiance_Resolution
        self.Irradiance_Resolution = Irradiance_Resolution
        self.Probe_Count = tuple(Probe_Count)
        self.Grid_Dimensions = tuple(Grid_Dimensions)
        self.Grid_Spacing = tuple(Grid_Spacing)


class Volumetrics_Settings:  # renderers.ads:33-41
    def __init__(self, Enabled=True, Visibility_Resolution=(100, 100, 100), Visibility_Step_Size=0.1,
                 Scattering_Resolution=(250, 250), Scattering_Step_Size=0.1):
        self.Enabled = bool(Enabled)
        self.Visibility_Resolution = tuple(Visibility_Resolution)
        self.Visibility_Step_Size = Visibility_Step_Size
        self.Scattering_Resolution = tuple(Scattering_Resolution)
        self.Scattering_Step_Size = Scattering_Step_Size


Default_Probe_Settings = Probe_Settings()
Default_Volumetrics_Settings = Volumetrics_Settings()
No_Volumetrics = Volumetrics_Settings(Enabled=False)  # renderers.ads:142-143


def _fp(a):
    return a.ctypes.data_as(C.c_void_p)


class Renderer:
    def __init__(self, binding, handle, window, scene, probes, volumetrics):
        self._b, self._h = binding, handle
        self.Window, self.Scene, self.Probes, self.Volumetrics = window, scene, probes, volumetrics
        self.Width, self.Height = window.Width, window.Height

    # ---- the reference's operations ------------------------------------------------
    def Render(self):  # renderers.adb:302-321
        self._b.check(self._b.render(self._h))

    def Set_Material(self, Index, Entity):  # renderers.adb:349-367
        alb = np.asarray(Entity.Get(materials.Albedo).data, dtype=np.float32)
        self._b.check(self._b.set_material(
            self._h, int(Index), alb.ctypes.data_as(C.POINTER(C.c_float)),
            float(Entity.Get(materials.Metallic).data), float(Entity.Get(materials.Roughness).data)))

    def Add_Material(self, Entity):  # renderers.adb:369-377
        alb = np.asarray(Entity.Get(materials.Albedo).data, dtype=np.float32)
        out = C.c_int32(-1)
        self._b.check(self._b.add_material(
            self._h, alb.ctypes.data_as(C.POINTER(C.c_float)),
            float(Entity.Get(materials.Metallic).data), float(Entity.Get(materials.Roughness).data),
            C.byref(out)))
        return out.value

    def Set_Primitive(self, Prim, Index, Entity):  # renderers.adb:379-398
        blob = gpu_types.entity_blob(self.Scene._prim_struct[Prim], Entity)
        self._b.check(self._b.set_primitive(self._h, self.Scene.prim_kind_index(Prim), int(Index),
                                            blob, len(blob)))

    def Add_Primitive(self, Prim, Entity):  # renderers.adb:435-456
        blob = gpu_types.entity_blob(self.Scene._prim_struct[Prim], Entity)
        out = C.c_int32(0)
        self._b.check(self._b.add_primitive(self._h, self.Scene.prim_kind_index(Prim), blob, len(blob),
                                            C.byref(out)))
        return out.value

    def Set_Light(self, Index, Lit, Entity):  # renderers.adb:458-483
        blob = gpu_types.entity_blob(self.Scene._light_struct[Lit], Entity)
        self._b.check(self._b.set_light(self._h, int(Index), self.Scene.light_kind_index(Lit), blob,
                                        len(blob)))

    def Set_Camera_Position(self, Position):  # renderers.adb:485-490
        p = np.asarray(Position, dtype=np.float32).reshape(3)
        self._b.check(self._b.set_camera_position(self._h, p.ctypes.data_as(C.POINTER(C.c_float))))

    def Set_Camera_Orientation(self, Orientation):  # renderers.adb:492-497
        """Orientation[i][j] = row i, column j of the 3x3 matrix; sent column-major."""
        m = np.asarray(Orientation, dtype=np.float32).reshape(3, 3)
        cm = np.ascontiguousarray(m.T).reshape(9)
        self._b.check(self._b.set_camera_orientation(self._h, cm.ctypes.data_as(C.POINTER(C.c_float))))

    def Eval_Distance_To(self, Position, Prims):  # renderers.adb:499-526 -> (distance, normal)
        d, n = self.Eval_Distances_To(np.asarray(Position, dtype=np.float32).reshape(1, 3), Prims)
        return float(d[0]), n[0]

    def Update_Partitioning(self, Method=GPU_Fast):  # renderers.adb:757-775
        self._b.check(self._b.update_partitioning(self._h, int(Method)))

    # ---- headless additions -------------------------------------------------------
    def Eval_Distances_To(self, Positions, Prims):
        pts = np.ascontiguousarray(Positions, dtype=np.float32).reshape(-1, 3)
        kinds = np.asarray([self.Scene.prim_kind_index(p) for p in Prims], dtype=np.int32)
        dist = np.empty(len(pts), dtype=np.float32)
        nrm = np.empty((len(pts), 3), dtype=np.float32)
        self._b.check(self._b.eval_distance_to(self._h, len(pts), _fp(pts), _fp(kinds), len(kinds),
                                               _fp(nrm), _fp(dist)))
        return dist, nrm

    def Render_Pass(self, Pass):
        self._b.check(self._b.render_pass(self._h, int(Pass)))

    # Render in three steps, for the sharded runs that exchange atlas slices between the passes
    def Frame_Begin(self):
        self._b.check(self._b.frame_begin(self._h))

    def Frame_Probe_Pass(self, Pass):
        self._b.check(self._b.frame_probe_pass(self._h, int(Pass)))

    def Frame_Exchange(self, Tex):
        """all-gather of the ranks' slices of an atlas, inside the library (needs Comm_Init; nothing without)"""
        self._b.check(self._b.frame_exchange(self._h, int(Tex)))

    def Frame_End(self):
        self._b.check(self._b.frame_end(self._h))

    # ---- one frame on N GPUs, one process per GPU: the communicator lives inside the library (RCCL)
    def Comm_Unique_Id(self):
        """rank 0: the 128 bytes every rank hands to Comm_Init"""
        buf = (C.c_uint8 * B.COMM_ID_BYTES)()
        self._b.check(self._b.comm_unique_id(buf))
        return bytes(buf)

    def Comm_Init(self, Id, Rank, World):
        """collective: from here on Render of every rank is one frame of the sharded schedule"""
        if len(Id) != B.COMM_ID_BYTES:
            raise ValueError("a communicator id is %d bytes" % B.COMM_ID_BYTES)
        buf = (C.c_uint8 * B.COMM_ID_BYTES).from_buffer_copy(Id)
        self._b.check(self._b.comm_init(self._h, buf, int(Rank), int(World)))

    def Comm_Available(self):
        """can this process load librccl at all?  (what a rank other than 0 asks before the collective join)"""
        self._b.check(self._b.comm_available())

    def Peer_Export(self):
        """this rank's interprocess handles (radiance atlases, events, a shared-memory block): 512 bytes for every rank"""
        buf = (C.c_uint8 * B.PEER_BLOB_BYTES)()
        self._b.check(self._b.peer_export(self._h, buf))
        return bytes(buf)

    def Peer_Init(self, Blobs, Rank, World):
        """every rank, with all ranks' Peer_Export blobs in rank order: Render is then the sharded frame, its exchange
        being device-to-device copies out of the peers' atlases (Comm_Destroy leaves)"""
        Blobs = b"".join(Blobs) if not isinstance(Blobs, (bytes, bytearray)) else bytes(Blobs)
        if len(Blobs) != B.PEER_BLOB_BYTES * int(World):
            raise ValueError("one %d-byte blob per rank" % B.PEER_BLOB_BYTES)
        buf = (C.c_uint8 * len(Blobs)).from_buffer_copy(Blobs)
        self._b.check(self._b.peer_init(self._h, buf, int(Rank), int(World)))

    def Comm_Destroy(self):
        self._b.check(self._b.comm_destroy(self._h))

    def Comm_Abort(self):
        self._b.check(self._b.comm_abort(self._h))

    def Comm_Barrier(self):
        self._b.check(self._b.comm_barrier(self._h))

    def Comm_Max(self, Value):
        v = C.c_double(float(Value))
        self._b.check(self._b.comm_max_f64(self._h, C.byref(v)))
        return v.value

    def Comm_Reduce_Framebuffer(self, Root=0):
        self._b.check(self._b.comm_reduce_framebuffer(self._h, int(Root)))

    def Finish(self):
        self._b.check(self._b.finish(self._h))

    def Set_Option(self, Option, Value):
        self._b.check(self._b.set_option(self._h, int(Option), int(Value)))

    def Get_Option(self, Option):
        v = C.c_int32(0)
        self._b.check(self._b.get_option(self._h, int(Option), C.byref(v)))
        return v.value

    def Read_Framebuffer(self):
        out = np.empty((self.Height, self.Width, 3), dtype=np.float32)
        self._b.check(self._b.read_framebuffer(self._h, _fp(out)))
        return out

    def Swap_Buffers(self):
        """Glfw.Windows.Context.Swap_Buffers at the end of Render (madarch-renderers.adb:320): the last
        frame becomes the window's RGBA8 pixels, converted on the device and copied to pinned host
        memory behind the frame; returns without waiting, frames stay in flight."""
        self._b.check(self._b.swap_buffers(self._h))

    def Front_Buffer(self, copy=True):
        """The pixels of the most recent Swap_Buffers: (H, W, 4) uint8, R G B A, row 0 = top.  Waits for
        that swap only.  copy=False returns a view of the renderer's buffer, valid until the second
        next Swap_Buffers."""
        ptr, n = C.c_void_p(), C.c_int64()
        self._b.check(self._b.front_buffer(self._h, C.byref(ptr), C.byref(n)))
        buf = (C.c_uint8 * (self.Height * self.Width * 4)).from_address(ptr.value)
        img = np.frombuffer(buf, dtype=np.uint8).reshape(self.Height, self.Width, 4)
        return img.copy() if copy else img

    def Read_Gbuffer(self):
        idx = np.empty((self.Height, self.Width), dtype=np.int32)
        t = np.empty((self.Height, self.Width), dtype=np.float32)
        steps = np.empty((self.Height, self.Width), dtype=np.int32)
        self._b.check(self._b.read_gbuffer(self._h, _fp(idx), _fp(t), _fp(steps)))
        return idx, t, steps

    def Texture_Shape(self, Tex):
        w, h, c = C.c_int32(), C.c_int32(), C.c_int32()
        self._b.check(self._b.read_texture(self._h, int(Tex), None, C.byref(w), C.byref(h), C.byref(c)))
        return h.value, w.value, c.value

    def Read_Texture(self, Tex):
        shape = self.Texture_Shape(Tex)
        out = np.empty(shape, dtype=np.float32)
        w, h, c = C.c_int32(), C.c_int32(), C.c_int32()
        self._b.check(self._b.read_texture(self._h, int(Tex), _fp(out), C.byref(w), C.byref(h), C.byref(c)))
        return out

    def Write_Texture(self, Tex, Data):
        a = np.ascontiguousarray(Data, dtype=np.float32)
        self._b.check(self._b.write_texture(self._h, int(Tex), _fp(a), a.shape[1], a.shape[0], a.shape[2]))

    def Probe_Total(self):
        return self.Probes.Probe_Count[0] * self.Probes.Probe_Count[1]

    def Read_Atlas_Slice(self, Tex, Probe_Begin, N_Probes):
        res = self.Probes.Radiance_Resolution if Tex == B.TEX_RADIANCE else self.Probes.Irradiance_Resolution
        out = np.empty((N_Probes, res, res, 3), dtype=np.float32)
        self._b.check(self._b.read_atlas_slice(self._h, int(Tex), int(Probe_Begin), int(N_Probes), _fp(out)))
        return out

    def Write_Atlas_Slice(self, Tex, Probe_Begin, Data):
        a = np.ascontiguousarray(Data, dtype=np.float32)
        self._b.check(self._b.write_atlas_slice(self._h, int(Tex), int(Probe_Begin), a.shape[0], _fp(a)))

    def Pass_Time(self, Pass):
        ms, n = C.c_double(0.0), C.c_int64(0)
        self._b.check(self._b.pass_time(self._h, int(Pass), C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def Reset_Pass_Times(self):
        self._b.check(self._b.reset_pass_times(self._h))

    def Scene_Layout(self, Is_Light, Kind_Ix):
        v = [C.c_int32() for _ in range(4)]
        self._b.check(self._b.scene_layout(self._h, int(Is_Light), int(Kind_Ix), *[C.byref(x) for x in v]))
        return tuple(x.value for x in v)  # count offset, array offset, stride, element size

    def Scene_Buffer_Size(self):
        s, t = C.c_int32(), C.c_int32()
        self._b.check(self._b.scene_buffer_size(self._h, C.byref(s), C.byref(t)))
        return s.value, t.value

    def Read_Scene_Buffer(self):
        n, _ = self.Scene_Buffer_Size()
        out = np.empty(n, dtype=np.uint8)
        self._b.check(self._b.read_scene_buffer(self._h, _fp(out), n))
        return out

    def Read_Partitioning(self):
        p = self.Scene.Partitioning_Config
        cells = p.Grid_Dimensions[0] * p.Grid_Dimensions[1] * p.Grid_Dimensions[2]
        width = len(self.Scene.Prims_Count) + p.Index_Count
        out = np.empty((cells, width), dtype=np.int32)
        self._b.check(self._b.read_partitioning(self._h, _fp(out), out.size))
        return out

    def Partition_Warnings(self):
        return self._b.partition_warnings(self._h)

    def Destroy(self):
        if self._h is not None:
            self._b.destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.Destroy()
        except Exception:
            pass


def Create(Window, Scene, Probes=None, Volumetrics=None, Device=0, Binding=None):
    """Renderers.Create (madarch-renderers.adb:91-300).  `Binding` defaults to the
    HIP library; there is no CPU implementation in this package."""
    b = Binding if Binding is not None else B.hip_binding()
    Probes = Probes if Probes is not None else Default_Probe_Settings
    Volumetrics = Volumetrics if Volumetrics is not None else Default_Volumetrics_Settings
    desc, keep = Scene._desc()
    ps = B.mdh_probe_settings()
    ps.radiance_resolution = Probes.Radiance_Resolution
    ps.irradiance_resolution = Probes.Irradiance_Resolution
    ps.probe_count = (C.c_int32 * 2)(*Probes.Probe_Count)
    ps.grid_dimensions = (C.c_int32 * 3)(*Probes.Grid_Dimensions)
    ps.grid_spacing = (C.c_float * 3)(*Probes.Grid_Spacing)
    vs = B.mdh_volumetrics()
    vs.enabled = 1 if Volumetrics.Enabled else 0
    vs.visibility_resolution = (C.c_int32 * 3)(*Volumetrics.Visibility_Resolution)
    vs.visibility_step_size = Volumetrics.Visibility_Step_Size
    vs.scattering_resolution = (C.c_int32 * 2)(*Volumetrics.Scattering_Resolution)
    vs.scattering_step_size = Volumetrics.Scattering_Step_Size
    handle = C.c_void_p()
    b.check(b.create(Window.Width, Window.Height, C.byref(desc), C.byref(ps), C.byref(vs), int(Device),
                     C.byref(handle)))
    del keep
    return Renderer(b, handle, Window, Scene, Probes, Volumetrics)
