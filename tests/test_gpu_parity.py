"""Parity of the HIP path with the CPU oracle on the same seeded inputs, through the C ABI
(run with -m gpu on the MI355X box).  Small frames are compared whole; at BASELINE.json's
full size (1920x1080, DDGI 8x8x8) the oracle renders a sample of the 8x8 tiles of the very
same frame, and size-independent properties are checked on the whole frame."""
import numpy as np
import pytest

from helpers import ODD_PROBES, SMALL_PROBES, assert_parity, make, same_bits, snapshot
from madarch_amd import _binding as B
from madarch_amd import examples, renderers, sharding

pytestmark = pytest.mark.gpu

CASES = [  # scene, W, H, mode, atlas, frames, probes
    ("simple_scene", 96, 64, 1, 0, 1, None),              # BASELINE config 1: primary rays only
    ("simple_scene", 96, 64, 2, 0, 1, None),              # config 2: direct PBR + AO, partitioned
    ("global_illumination", 80, 56, 0, 0, 3, None),       # config 3 shape, reference-default 4x3x3 probes
    ("global_illumination", 80, 56, 0, 1, 3, None),       # fp32 atlases
    ("global_illumination", 61, 37, 0, 0, 2, SMALL_PROBES),  # ragged: sizes that are no multiple of 8
    ("light_shafts", 64, 48, 0, 0, 2, SMALL_PROBES),      # config 4: volumetrics
    ("simple_scene", 64, 48, 0, 0, 2, SMALL_PROBES),      # partitioned scene through the full path
    ("global_illumination", 56, 40, 0, 0, 3, ODD_PROBES),  # 75 probes, 12 / 6 texel tiles, 15 x 5 atlas: no power of two anywhere
    ("global_illumination", 56, 40, 0, 1, 2, ODD_PROBES),
    ("light_shafts", 48, 40, 0, 0, 2, ODD_PROBES),
]


@pytest.mark.parametrize("scene,W,H,mode,atlas,frames,probes", CASES)
def test_frame_parity(hip, orc, scene, W, H, mode, atlas, frames, probes):
    got = snapshot(make(scene, W, H, hip, mode=mode, atlas=atlas, probes=probes), frames)
    want = snapshot(make(scene, W, H, orc, mode=mode, atlas=atlas, probes=probes), frames)
    assert_parity(got, want)
    # stronger than the stated tolerance: the two images agree to the bit almost everywhere
    assert (got["image"].view(np.uint32) == want["image"].view(np.uint32)).mean() > 0.999


def test_moved_camera_and_light(hip, orc):
    """A rotated camera (column-major matrix through the ABI) and a moved light."""
    from madarch_amd.lights import spot_lights
    outs = []
    c, s = np.float32(np.cos(0.4)), np.float32(np.sin(0.4))
    rot_y = [[c, 0, s], [0, 1, 0], [-s, 0, c]]
    for b in (hip, orc):
        R = make("global_illumination", 64, 40, b, probes=SMALL_PROBES)
        R.Set_Camera_Position((3.0, 2.5, -2.0))
        R.Set_Camera_Orientation(rot_y)
        R.Set_Light(1, spot_lights.Spot_Light, spot_lights.Create((3.5, 5.0, 2.0), (-1.0, 0.0, 0.0), 3.1415 / 4.0, (0.9, 0.9, 0.8)))
        outs.append(snapshot(R, 3))
    assert_parity(*outs)


@pytest.mark.parametrize("rres,ires", [(64, 8), (80, 4), (48, 16)])
def test_large_radiance_tiles(hip, orc, rres, ires):
    """Large radiance tiles: with up to 64 irradiance texels per probe one wavefront folds and the taps go through
    two small LDS buffers in chunks (any tile size); with more texels all taps are staged at once (48 x 48 x 32 B =
    72 KiB of LDS here)."""
    big = renderers.Probe_Settings(Radiance_Resolution=rres, Irradiance_Resolution=ires, Probe_Count=(4, 2), Grid_Dimensions=(2, 2, 2),
                                   Grid_Spacing=(4.0, 4.0, 5.0))
    outs = [snapshot(make("global_illumination", 40, 24, b, probes=big), 2) for b in (hip, orc)]
    assert_parity(*outs)


def test_radiance_tiles_beyond_the_lds(hip):
    """... and past 160 KiB that path refuses loudly."""
    too_big = renderers.Probe_Settings(Radiance_Resolution=80, Irradiance_Resolution=16, Probe_Count=(4, 2), Grid_Dimensions=(2, 2, 2))
    R = make("global_illumination", 16, 8, hip, probes=too_big)
    with pytest.raises(B.MadarchError):
        R.Render()


@pytest.mark.parametrize("steps", [0, 1, 8])
def test_ambient_occlusion_steps(hip, orc, steps):
    """M_AMBIENT_OCCLUSION_STEPS (lighting.glsl:51-69) other than the reference's 3 (madarch-renderers.adb:139), including none."""
    outs = []
    for b in (hip, orc):
        R = make("global_illumination", 48, 32, b, probes=SMALL_PROBES)
        R.Set_Option(B.OPT_AO_STEPS, steps)
        outs.append(snapshot(R, 2))
    assert_parity(*outs)


def test_empty_scene_and_sky(hip, orc):
    """No primitive added: every ray misses and returns the sky (render_probes.glsl:287)."""
    from madarch_amd import scenes, windows
    from madarch_amd.lights import point_lights
    from madarch_amd.primitives import spheres
    outs = []
    for b in (hip, orc):
        scene = scenes.Compile([(spheres.Sphere, 4)], [(point_lights.Point_Light, 2)],
                               Partitioning=scenes.Partitioning_Settings(Enable=False))
        R = renderers.Create(windows.Open(32, 24), scene, Volumetrics=renderers.No_Volumetrics, Binding=b)
        R.Set_Option(B.OPT_GBUFFER, 1)
        outs.append(snapshot(R, 1))
    assert_parity(*outs)
    assert (outs[0]["gb_index"] == -1).all()


@pytest.mark.parametrize("partitioned,method", [(False, 0), (True, 0), (True, 2)])
@pytest.mark.parametrize("W,H", [(1, 1), (9, 3), (40, 24)])
def test_every_kind_at_its_declared_maximum(hip, orc, partitioned, method, W, H):
    """Every built-in kind filled to its declared count, two kinds of lights at theirs, images down to one
    pixel: spheres, tilted and axis planes, boxes and triangles through the whole frame, with and without
    the space partition (a small Index_Count makes cells overflow and cut their lists)."""
    from madarch_amd import materials, scenes, windows
    from madarch_amd.lights import point_lights, spot_lights
    from madarch_amd.primitives import boxes, planes, spheres, triangles
    outs = []
    for b in (hip, orc):
        part = scenes.Partitioning_Settings(Enable=partitioned, Index_Count=6, Grid_Dimensions=(5, 5, 7), Grid_Spacing=(2.0, 2.0, 2.0),
                                            Grid_Offset=(-2.0, -2.0, -7.0))
        scene = scenes.Compile([(spheres.Sphere, 5), (planes.Plane, 7), (boxes.Box, 3), (triangles.Triangle, 2)],
                               [(point_lights.Point_Light, 2), (spot_lights.Spot_Light, 4)], Partitioning=part)
        R = renderers.Create(windows.Open(W, H), scene, Probes=SMALL_PROBES, Volumetrics=renderers.No_Volumetrics, Binding=b)
        for m, alb in enumerate(((0.8, 0.8, 0.8), (0.9, 0.1, 0.1), (0.1, 0.1, 0.9), (0.2, 0.2, 0.2))):
            R.Set_Material(m, materials.Create(alb, 0.9 if m == 3 else 0.0, 0.15 if m == 3 else 0.6))
        for n, o, m in (((0, 1, 0), 1.0, 0), ((0, -1, 0), 7.0, 0), ((1, 0, 0), 1.0, 1), ((-1, 0, 0), 7.0, 2), ((0, 0, 1), 6.0, 0),
                        ((0, 0, -1), 7.0, 0), ((0.6, 0.8, 0.0), 0.5, 1)):  # the last one is not axis-aligned
            R.Add_Primitive(planes.Plane, planes.Create(n, o, m))
        for i in range(5):
            R.Add_Primitive(spheres.Sphere, spheres.Create((1.0 + 1.1 * i, 0.5 + 0.7 * i, 3.0 + 0.5 * i), 0.45 + 0.1 * i, 3 if i % 2 else 1))
        for i in range(3):
            R.Add_Primitive(boxes.Box, boxes.Create((1.0 + 2.0 * i, 0.0, 5.0 - i), (0.6, 0.9 - 0.2 * i, 0.5), 2 if i else 3))
        R.Add_Primitive(triangles.Triangle, triangles.Create((0.0, 3.0, 5.0), (2.0, 5.0, 5.5), (3.0, 3.0, 4.0), 1))
        R.Add_Primitive(triangles.Triangle, triangles.Create((4.0, 1.0, 2.5), (5.5, 2.0, 3.0), (5.0, 0.5, 3.5), 3))
        R.Set_Light(1, point_lights.Point_Light, point_lights.Create((1.0, 5.0, 1.0), (0.7, 0.7, 0.6)))
        R.Set_Light(2, point_lights.Point_Light, point_lights.Create((5.0, 2.0, 0.0), (0.2, 0.3, 0.6)))
        # Set_Light leaves the kind's count AND the total at Index (renderers.adb:478-482): four spot entries make the
        # total 4, which the walk by cumulative counts reads as the two point lights and the first two spots
        for i, (pos, d, ap, col) in enumerate((((3.5, 6.0, 2.0), (0.0, -1.0, 0.2), 0.9, (0.9, 0.9, 0.8)), ((0.0, 1.0, -3.0), (0.3, 0.1, 1.0), 0.5, (0.5, 0.2, 0.2)),
                                               ((6.0, 6.0, 6.0), (-0.5, -0.7, -0.5), 0.7, (0.1, 0.5, 0.1)), ((2.0, 6.5, 5.0), (0.0, -1.0, 0.0), 0.4, (0.3, 0.3, 0.3)))):
            R.Set_Light(i + 1, spot_lights.Spot_Light, spot_lights.Create(pos, d, ap, col))
        R.Set_Camera_Position((2.5, 2.0, -2.0))
        R.Set_Option(B.OPT_GBUFFER, 1)
        if partitioned:
            R.Update_Partitioning(method)
        out = snapshot(R, 2)
        if partitioned:
            out["partition"], out["warnings"] = R.Read_Partitioning(), np.array([R.Partition_Warnings()])
        outs.append(out)
    assert_parity(*outs)
    if partitioned:
        assert same_bits(outs[0]["partition"], outs[1]["partition"]) and same_bits(outs[0]["warnings"], outs[1]["warnings"])
        if method == 0:
            assert outs[0]["warnings"][0] > 0  # Index_Count = 6 is too small for this room


def test_open_scene_with_few_hits_per_wavefront(hip, orc):
    """A floor, two spheres and the sky: most probe rays of a wavefront leave the scene and only a few lanes
    shade a point (the radiance pass's visibility queue has to start with any number of lanes)."""
    from madarch_amd import materials, scenes, windows
    from madarch_amd.lights import point_lights
    from madarch_amd.primitives import planes, spheres
    outs = []
    for b in (hip, orc):
        scene = scenes.Compile([(spheres.Sphere, 4), (planes.Plane, 2)], [(point_lights.Point_Light, 2)],
                               Partitioning=scenes.Partitioning_Settings(Enable=False))
        R = renderers.Create(windows.Open(48, 32), scene, Probes=SMALL_PROBES, Volumetrics=renderers.No_Volumetrics, Binding=b)
        R.Set_Material(0, materials.Create((0.7, 0.7, 0.7), 0.0, 0.6))
        R.Set_Material(1, materials.Create((0.9, 0.2, 0.1), 0.8, 0.2))
        R.Add_Primitive(planes.Plane, planes.Create((0, 1, 0), 1.0, 0))
        R.Add_Primitive(spheres.Sphere, spheres.Create((2.0, 0.5, 4.0), 1.2, 1))
        R.Add_Primitive(spheres.Sphere, spheres.Create((4.5, 2.5, 3.0), 0.6, 0))
        R.Set_Light(1, point_lights.Point_Light, point_lights.Create((3.0, 6.0, 1.0), (0.9, 0.9, 0.8)))
        R.Set_Camera_Position((2.5, 1.5, -1.0))
        R.Set_Option(B.OPT_GBUFFER, 1)
        outs.append(snapshot(R, 3))
    assert_parity(*outs)
    assert 0.05 < (outs[0]["gb_index"] == -1).mean() < 0.95  # sky and hits both


def test_set_primitive_updates_device_tables(hip, orc):
    """Set_Primitive between frames (what the examples' loops do with lights) reaches the kernels."""
    from madarch_amd.primitives import spheres
    outs = []
    for b in (hip, orc):
        R = make("global_illumination", 48, 32, b, probes=SMALL_PROBES)
        R.Render()
        R.Set_Primitive(spheres.Sphere, 1, spheres.Create((2.0, 2.0, 3.0), 0.8, 4))
        outs.append(snapshot(R, 2))
    assert_parity(*outs)


@pytest.mark.parametrize("probes,N,irr_all", [(SMALL_PROBES, 2, 1), (ODD_PROBES, 4, 0), (ODD_PROBES, 3, 1)])
def test_sharded_frame_equals_whole_frame(hip, probes, N, irr_all):
    """N ranks' worth of tiles and probe slices (uneven when N does not divide the probe count), exchanged
    through the host, give the whole frame bit for bit (the multi-GPU path with all 'ranks' on this one GPU)."""
    whole = snapshot(make("global_illumination", 72, 40, hip, probes=probes), 2)
    Rs = [make("global_illumination", 72, 40, hip, probes=probes) for _ in range(N)]
    for r, R in enumerate(Rs):
        R.Set_Option(B.OPT_RANK, r)
        R.Set_Option(B.OPT_WORLD, N)
        R.Set_Option(B.OPT_IRRADIANCE_ALL, irr_all)
    P = Rs[0].Probe_Total()
    bounds = [(P * r // N, P * (r + 1) // N) for r in range(N)]

    def exchange(tex):
        parts = [R.Read_Atlas_Slice(tex, lo, hi - lo) for R, (lo, hi) in zip(Rs, bounds)]
        for r, R in enumerate(Rs):
            for q, (lo, hi) in enumerate(bounds):
                if q != r:
                    R.Write_Atlas_Slice(tex, lo, parts[q])
    for _ in range(2):
        for R in Rs:
            R.Render_Pass(B.PASS_RADIANCE)
        exchange(B.TEX_RADIANCE)
        for R in Rs:
            R.Render_Pass(B.PASS_IRRADIANCE)
        if not irr_all:  # own probes only: the slices have to travel as well
            exchange(B.TEX_IRRADIANCE)
        for R in Rs:
            R.Render_Pass(B.PASS_SCREEN)
    img = sum(R.Read_Framebuffer() for R in Rs)
    assert same_bits(img, whole["image"])
    for R in Rs:
        assert same_bits(R.Read_Texture(B.TEX_IRRADIANCE), whole["irradiance"])
        assert same_bits(R.Read_Texture(B.TEX_RADIANCE), whole["radiance"])


@pytest.mark.parametrize("irr_all,mips", [(1, 0), (0, 0), (1, 1)])
def test_sharded_frames_in_flight_equal_whole_frames(hip, irr_all, mips):
    """The same two ranks through the three-step frame (Frame_Begin / Frame_Probe_Pass / Frame_End) that
    madarch_amd.sharding drives: frames stay in flight (two atlas sets, three streams per renderer) while
    the slices are exchanged inside the open frame."""
    frames = 5
    W = make("global_illumination", 120, 72, hip, probes=SMALL_PROBES)
    W.Set_Option(B.OPT_GBUFFER, 0)
    W.Set_Option(B.OPT_FRAME_OVERLAP, 0)
    W.Set_Option(B.OPT_RADIANCE_MIPS, mips)  # (the optional mip chain: every rank builds it from the whole, exchanged atlas)
    for _ in range(frames):
        W.Render()
    want = {"image": W.Read_Framebuffer(), "irradiance": W.Read_Texture(B.TEX_IRRADIANCE), "radiance": W.Read_Texture(B.TEX_RADIANCE)}
    Rs = [make("global_illumination", 120, 72, hip, probes=SMALL_PROBES) for _ in range(2)]
    for r, R in enumerate(Rs):
        R.Set_Option(B.OPT_GBUFFER, 0)
        R.Set_Option(B.OPT_RANK, r)
        R.Set_Option(B.OPT_WORLD, 2)
        R.Set_Option(B.OPT_IRRADIANCE_ALL, irr_all)
        R.Set_Option(B.OPT_RADIANCE_MIPS, mips)
    P = Rs[0].Probe_Total()
    for _ in range(frames):
        for R in Rs:
            R.Frame_Begin()
        for p, tex in ((B.PASS_RADIANCE, B.TEX_RADIANCE), (B.PASS_IRRADIANCE, B.TEX_IRRADIANCE)):
            for R in Rs:
                R.Frame_Probe_Pass(p)
            if tex == B.TEX_IRRADIANCE and irr_all:
                continue  # every rank updated every probe: nothing to exchange
            lo = Rs[0].Read_Atlas_Slice(tex, 0, P // 2)
            hi = Rs[1].Read_Atlas_Slice(tex, P // 2, P - P // 2)
            Rs[0].Write_Atlas_Slice(tex, P // 2, hi)
            Rs[1].Write_Atlas_Slice(tex, 0, lo)
        for R in Rs:
            R.Frame_End()
    img = Rs[0].Read_Framebuffer() + Rs[1].Read_Framebuffer()
    assert same_bits(img, want["image"])
    assert same_bits(Rs[0].Read_Texture(B.TEX_IRRADIANCE), want["irradiance"])
    assert same_bits(Rs[1].Read_Texture(B.TEX_RADIANCE), want["radiance"])


@pytest.mark.gpu
@pytest.mark.parametrize("overlap,gbuffer,scene", [(1, 0, "global_illumination"), (2, 0, "global_illumination"), (2, 1, "global_illumination"),
                                                   (2, 0, "light_shafts")])
def test_pipelined_frames_equal_serial_frames(hip, overlap, gbuffer, scene):
    """MDH_OPT_FRAME_OVERLAP only reschedules: bursts of pipelined frames (three streams, two atlas sets,
    two framebuffers) with camera and scene edits, single passes and read-backs in between give what
    the strictly serial renderer gives, bit for bit, at every point where something is read."""
    from madarch_amd.primitives import spheres

    def drive(level):
        R = make(scene, 200, 120, hip, probes=SMALL_PROBES)
        R.Set_Option(B.OPT_GBUFFER, gbuffer)
        R.Set_Option(B.OPT_FRAME_OVERLAP, level)
        seen = []
        for burst in range(4):
            for f in range(1 + 2 * burst):  # 1, 3, 5, 7 frames without a host read in between
                R.Set_Camera_Position((2.0 + 0.05 * f, 2.0, 0.1 * burst))
                R.Render()
            seen.append(R.Read_Framebuffer())
            seen.append(R.Read_Texture(B.TEX_IRRADIANCE))
            if burst == 1:  # a scene edit drains the pipeline
                R.Set_Primitive(spheres.Sphere, 1, spheres.Create((2.5, 3.0, 3.0), 0.9, 4))
            if burst == 2:  # single passes work in place on the atlas set the last frame wrote
                R.Render_Pass(B.PASS_RADIANCE)
                R.Render_Pass(B.PASS_IRRADIANCE)
                R.Render_Pass(B.PASS_SCREEN)
                seen.append(R.Read_Framebuffer())
                seen.append(R.Read_Texture(B.TEX_RADIANCE))
        R.Render()
        R.Render()
        seen.append(R.Read_Texture(B.TEX_RADIANCE))
        seen.append(R.Read_Framebuffer())
        return seen

    serial, piped = drive(0), drive(overlap)
    assert len(serial) == len(piped)
    for i, (a, b) in enumerate(zip(serial, piped)):
        assert same_bits(a, b), "output %d differs between the serial and the pipelined schedule" % i


@pytest.mark.gpu
def test_pipelined_frames_across_screen_modes(hip):
    """Frames of modes 1 and 2 run no probe passes: pipelined, they must leave the atlas sets alone (and still
    alternate the screen streams), so that what mode 0 reads afterwards is what the serial schedule reads."""
    def drive(level):
        R = make("global_illumination", 160, 96, hip, probes=SMALL_PROBES)
        R.Set_Option(B.OPT_GBUFFER, 0)
        R.Set_Option(B.OPT_FRAME_OVERLAP, level)
        seen = []
        for mode, frames in ((0, 3), (2, 3), (0, 2), (1, 1), (0, 1), (2, 2)):
            R.Set_Option(B.OPT_SCREEN_MODE, mode)
            for f in range(frames):
                R.Set_Camera_Position((2.0 + 0.1 * f, 2.0, 0.05 * mode))
                R.Render()
            seen += [R.Read_Framebuffer(), R.Read_Texture(B.TEX_IRRADIANCE), R.Read_Texture(B.TEX_RADIANCE)]
        return seen

    serial = drive(0)
    for level in (1, 2):
        for i, (a, b) in enumerate(zip(serial, drive(level))):
            assert same_bits(a, b), "output %d differs at overlap level %d" % (i, level)


def test_caller_supplied_stream(hip):
    """mdh_set_stream: on a stream of the caller frames run serially behind the caller's work; handing the renderer
    back to its own streams resumes frames in flight.  Same bits throughout."""
    import ctypes as C
    from madarch_amd.primitives import spheres
    rt = C.CDLL("libamdhip64.so")  # the HIP runtime the library itself is linked against

    def drive(use_stream):
        R = make("simple_scene", 96, 64, hip, probes=SMALL_PROBES)
        R.Set_Option(B.OPT_GBUFFER, 0)
        st = C.c_void_p()
        if use_stream:
            assert rt.hipStreamCreate(C.byref(st)) == 0
        seen = []
        for phase in range(3):
            if use_stream:
                hip.check(hip.set_stream(R._h, C.c_void_p(st.value if phase != 1 else 0)))  # phase 1: back on its own streams
                if phase != 1:
                    cur = C.c_void_p()
                    hip.check(hip.stream(R._h, C.byref(cur)))
                    assert cur.value == st.value
            for f in range(4):
                R.Set_Camera_Position((2.0 + 0.1 * f, 2.0, 0.05 * phase))
                R.Render()
                R.Swap_Buffers()
            R.Set_Primitive(spheres.Sphere, 2, spheres.Create((2.0, 3.0, 2.5 + phase), 0.6, 3))
            R.Update_Partitioning(phase)
            seen.append(R.Eval_Distances_To(np.linspace(0.0, 5.0, 30, dtype=np.float32).reshape(10, 3), [spheres.Sphere])[0])
            R.Render()
            seen += [R.Read_Framebuffer(), R.Read_Texture(B.TEX_IRRADIANCE), R.Read_Partitioning(), R.Front_Buffer()]
        if use_stream:
            hip.check(hip.set_stream(R._h, C.c_void_p(0)))
            R.Destroy()
            assert rt.hipStreamDestroy(st) == 0
        return seen

    for a, b in zip(drive(False), drive(True)):
        assert same_bits(a, b)


def window_pixels(img):
    """float framebuffer -> the RGBA8 default framebuffer of the reference's window (OpenGL 4.3 core 2.3.5.1)"""
    with np.errstate(invalid="ignore"):
        x = np.where(np.isnan(img), np.float32(0.0), img)
        lv = np.rint(np.clip(x, np.float32(0.0), np.float32(1.0)) * np.float32(255.0)).astype(np.uint8)
    return np.concatenate([lv, np.full(lv.shape[:2] + (1,), 255, np.uint8)], axis=2)


@pytest.mark.gpu
@pytest.mark.parametrize("overlap,window", [(0, 0), (2, 0), (0, 1), (2, 1), (2, 2)])
def test_swap_buffers_shows_each_frame(hip, orc, overlap, window):
    """Swap_Buffers (renderers.adb:320): the RGBA8 pixels of every frame, converted on the device and copied
    behind the frame while later frames are already in flight, are the conversion of that frame's colours bit
    for bit -- whichever stream and framebuffer drew it -- and match the oracle's window."""
    frames = 7
    R = make("global_illumination", 200, 120, hip, probes=SMALL_PROBES)
    R.Set_Option(B.OPT_GBUFFER, 0)
    R.Set_Option(B.OPT_FRAME_OVERLAP, overlap)
    R.Set_Option(B.OPT_WINDOW, window)  # 1: the screen pass stores the pixels in pinned host memory itself
    S = make("global_illumination", 200, 120, hip, probes=SMALL_PROBES)  # the same frames, read one by one
    S.Set_Option(B.OPT_GBUFFER, 0)
    S.Set_Option(B.OPT_FRAME_OVERLAP, 0)
    O = make("global_illumination", 200, 120, orc, probes=SMALL_PROBES)
    views, want, late = [], [], None
    for f in range(frames):
        for X in (R, S, O):
            X.Set_Camera_Position((2.0 + 0.07 * f, 2.0, 0.02 * f))
            X.Render()
        want.append(window_pixels(S.Read_Framebuffer()))
        R.Swap_Buffers()
        if f % 3 == 0:
            views.append((f, R.Front_Buffer()))
        elif f % 3 == 1:  # a view of the renderer's buffer stays valid until the second next swap ...
            late = (f, R.Front_Buffer(copy=False))
        else:             # ... so it is still this after one more frame and swap
            views.append((late[0], late[1].copy()))
    views.append((frames - 1, R.Front_Buffer()))
    assert len(views) == 6
    for f, got in views:
        assert got.shape == (120, 200, 4) and got.dtype == np.uint8
        assert (got == want[f]).all(), "frame %d" % f
    # the oracle's window for the last frame
    O.Swap_Buffers()
    o = O.Front_Buffer()
    assert (o == window_pixels(O.Read_Framebuffer())).all()
    d = np.abs(o.astype(np.int16) - want[-1].astype(np.int16))
    assert d.max() <= 1 and (d == 0).mean() > 0.999
    # a single pass outside a frame, then a swap: ordered after everything in flight
    R.Render()
    R.Render()
    R.Render_Pass(B.PASS_SCREEN)
    R.Swap_Buffers()
    assert (R.Front_Buffer() == window_pixels(R.Read_Framebuffer())).all()
    # switching where the conversion happens, between frames; a swap shows the frame before it, not later ones
    R.Set_Option(B.OPT_WINDOW, 1 - window if window < 2 else 0)
    R.Render()
    R.Swap_Buffers()
    shown = window_pixels(R.Read_Framebuffer())
    R.Set_Camera_Position((2.3, 2.1, 0.0))
    R.Render()
    assert (R.Front_Buffer() == shown).all()
    R.Swap_Buffers()
    assert (R.Front_Buffer() == window_pixels(R.Read_Framebuffer())).all()
    assert not (R.Front_Buffer() == shown).all()


@pytest.mark.gpu
@pytest.mark.parametrize("window", [0, 1, 2])
def test_swap_buffers_of_a_rank(hip, window):
    """In a sharded run the window of a rank holds its own 8x8 tiles and zeros elsewhere, like its framebuffer,
    also after the rank changes."""
    R = make("global_illumination", 120, 72, hip, probes=SMALL_PROBES)
    R.Set_Option(B.OPT_WINDOW, window)
    R.Render()
    R.Swap_Buffers()
    whole = R.Front_Buffer()
    ty, tx = np.meshgrid(np.arange(72) // 8, np.arange(120) // 8, indexing="ij")
    for rank in (1, 2, 0):
        R.Set_Option(B.OPT_WORLD, 3)
        R.Set_Option(B.OPT_RANK, rank)
        for _ in range(5):  # round the ring of host buffers
            R.Render_Pass(B.PASS_SCREEN)
        R.Swap_Buffers()
        mine = (ty * 15 + tx) % 3 == rank
        px = R.Front_Buffer()
        assert (px[mine] == whole[mine]).all() and (px[~mine] == 0).all()


@pytest.mark.gpu
def test_ball_game_frames(hip, orc):
    """examples/ball_game: per frame a physics step (Eval_Distance_To against planes and boxes, Set_Primitive per
    ball), a partition rebuild and a frame.  Ball states and frames must be those of the oracle bit for bit."""
    runs = []
    for b in (hip, orc):
        G = examples.ball_game(96, 64, Probes=SMALL_PROBES, Binding=b)
        G.R.Set_Option(B.OPT_GBUFFER, 1)
        for f in range(45):
            if f in (0, 6, 11):
                G.Throw_Ball()
                G.Move_Camera((0.3, 0.1, 0.0))
            G.Frame()
        runs.append((G, snapshot(G.R, 1)))
    (Gh, sh), (Go, so) = runs
    for bh, bo in zip(Gh.Ball_Bodies, Go.Ball_Bodies):
        assert bh[0] == bo[0] and same_bits(bh[1], bo[1]) and same_bits(bh[2], bo[2])
    assert any(body[2][1] > 0.0 for body in Go.Ball_Bodies)  # a ball has bounced off the box
    assert_parity(sh, so)
    assert same_bits(np.asarray(Gh.R.Read_Partitioning()), np.asarray(Go.R.Read_Partitioning()))


def test_timing_with_thousands_of_frames_in_flight(hip):
    """MDH_OPT_TIMING keeps an event pair per pass and folds them lazily.  The bound on that list (4096 pairs) is
    reached in the middle of a burst of pipelined frames: only pairs whose kernels have finished may be folded there
    (nothing is waited for, no event is handed out twice), and the totals must count every launch once."""
    R = make("global_illumination", 32, 24, hip, probes=SMALL_PROBES)
    R.Set_Option(B.OPT_GBUFFER, 0)
    R.Set_Option(B.OPT_TIMING, 1)
    frames = 1500  # x 3 passes = 4500 pairs
    for _ in range(frames):
        R.Render()
    for p in (B.PASS_RADIANCE, B.PASS_IRRADIANCE, B.PASS_SCREEN):
        ms, n = R.Pass_Time(p)
        assert n == frames and ms > 0.0
    R.Reset_Pass_Times()
    for _ in range(10):
        R.Render()
    assert R.Pass_Time(B.PASS_SCREEN)[1] == 10
    R.Destroy()


def test_state_errors_of_open_frames_and_ranks(hip):
    """What an open frame has latched cannot change under it, and a rank outside its world is refused where a
    slice is used (both MDH_E_STATE, as a Program_Error of the Ada body would be)."""
    import ctypes as C
    R = make("global_illumination", 32, 24, hip, probes=SMALL_PROBES)
    R.Frame_Begin()
    for opt, val in ((B.OPT_ATLAS_FORMAT, 1), (B.OPT_RANK, 0), (B.OPT_WORLD, 2), (B.OPT_FRAME_OVERLAP, 0), (B.OPT_SCREEN_MODE, 2)):
        with pytest.raises(B.MadarchError) as e:
            R.Set_Option(opt, val)
        assert e.value.status == B.MDH_E_STATE
    R.Frame_Probe_Pass(B.PASS_RADIANCE)
    R.Frame_Probe_Pass(B.PASS_IRRADIANCE)
    R.Frame_End()
    want = R.Read_Framebuffer()
    R.Set_Option(B.OPT_RANK, 2)  # world is still 1
    for call in (R.Render, lambda: R.Render_Pass(B.PASS_RADIANCE), R.Frame_Begin):
        with pytest.raises(B.MadarchError) as e:
            call()
        assert e.value.status == B.MDH_E_STATE
    ptr, total, off, own = C.c_void_p(), C.c_int64(), C.c_int64(), C.c_int64()
    assert hip.atlas_device_ptr(R._h, B.TEX_RADIANCE, C.byref(ptr), C.byref(total), C.byref(off), C.byref(own)) == B.MDH_E_STATE
    R.Set_Option(B.OPT_RANK, 0)
    assert same_bits(R.Read_Framebuffer(), want)
    R.Destroy()


def test_library_slices_are_the_exchanges_slices(hip):
    """mdh_atlas_device_ptr (what the in-place RCCL all-gather slices the atlas by) against sharding.slice_bytes
    (what the host exchange reads and writes), even and uneven splits, both texel formats."""
    import ctypes as C
    for probes in (SMALL_PROBES, ODD_PROBES):
        for atlas in (0, 1):
            R = make("global_illumination", 16, 8, hip, probes=probes, atlas=atlas)
            P = R.Probe_Total()
            for world in (1, 2, 3, 4, 8):
                R.Set_Option(B.OPT_WORLD, world)
                for rank in range(world):
                    R.Set_Option(B.OPT_RANK, rank)
                    for tex, res in ((B.TEX_RADIANCE, probes.Radiance_Resolution), (B.TEX_IRRADIANCE, probes.Irradiance_Resolution)):
                        ptr, total, off, own = C.c_void_p(), C.c_int64(), C.c_int64(), C.c_int64()
                        hip.check(hip.atlas_device_ptr(R._h, tex, C.byref(ptr), C.byref(total), C.byref(off), C.byref(own)))
                        texel = 4 if atlas == 0 else 16
                        assert total.value == P * res * res * texel
                        assert (off.value, own.value) == sharding.slice_bytes(P, res, texel, rank, world)
                R.Set_Option(B.OPT_RANK, 0)
            R.Destroy()


@pytest.mark.parametrize("spec", [0, 1, 3])
@pytest.mark.parametrize("scene,W,H,probes,atlas", [("global_illumination", 80, 56, SMALL_PROBES, 0), ("global_illumination", 56, 40, ODD_PROBES, 1),
                                                     ("simple_scene", 64, 48, SMALL_PROBES, 0), ("light_shafts", 48, 40, None, 0)])
def test_indirect_specular_modes(hip, orc, spec, scene, W, H, probes, atlas):
    """M_COMPUTE_INDIRECT_SPECULAR other than the 2 the reference's renderer fixes (render_probes.glsl:264-272):
    0 none, 1 sample_radiance_with_specular (:71-136), 3 compute_indirect_specular (:211-244) -- through the
    brute-force scan, the space partition and the volumetric composite, RGB8 and fp32 atlases, power-of-two and odd
    atlas dimensions."""
    outs = []
    for b in (hip, orc):
        R = make(scene, W, H, b, atlas=atlas, probes=probes)
        R.Set_Option(B.OPT_INDIRECT_SPECULAR, spec)
        assert R.Get_Option(B.OPT_INDIRECT_SPECULAR) == spec
        outs.append(snapshot(R, 3))
    assert_parity(*outs)
    assert (outs[0]["image"].view(np.uint32) == outs[1]["image"].view(np.uint32)).mean() > 0.999
    # the mode really is another image than the default's
    R = make(scene, W, H, hip, atlas=atlas, probes=probes)
    assert R.Get_Option(B.OPT_INDIRECT_SPECULAR) == 2
    # (every material of light_shafts has roughness 1: no pixel sends a reflection ray, whatever the mode)
    assert same_bits(snapshot(R, 3)["image"], outs[0]["image"]) == (scene == "light_shafts")
    with pytest.raises(B.MadarchError):
        R.Set_Option(B.OPT_INDIRECT_SPECULAR, 4)


@pytest.mark.parametrize("scene,W,H,spec,atlas,overlap", [("global_illumination", 80, 56, 2, 0, 2), ("global_illumination", 72, 48, 1, 1, 0),
                                                         ("simple_scene", 64, 48, 2, 1, 2), ("light_shafts", 48, 40, 1, 0, 2)])
def test_radiance_mips_switch(hip, orc, scene, W, H, spec, atlas, overlap):
    """MDH_OPT_RADIANCE_MIPS (off by default; the reference's atlases have one level): a mip chain of the radiance atlas,
    rebuilt before every screen pass, read by mode 2's tap at level 1 and by mode 1's at mix (0, radiance_lods, 2 roughness)
    -- RGB8 and fp32 atlases, frames in flight or not, through the space partition and the volumetric composite.  The
    levels themselves are read back and compared bit for bit."""
    outs, levels = [], []
    for b in (hip, orc):
        R = make(scene, W, H, b, atlas=atlas, probes=SMALL_PROBES)
        R.Set_Option(B.OPT_FRAME_OVERLAP, overlap)
        R.Set_Option(B.OPT_INDIRECT_SPECULAR, spec)
        assert R.Get_Option(B.OPT_RADIANCE_MIPS) == 0
        R.Set_Option(B.OPT_RADIANCE_MIPS, 1)
        assert R.Get_Option(B.OPT_RADIANCE_MIPS) == 1
        outs.append(snapshot(R, 3))
        levels.append([R.Read_Texture(B.TEX_RADIANCE_MIP0 + l) for l in range(1, 5)])
    assert_parity(*outs)
    assert (outs[0]["image"].view(np.uint32) == outs[1]["image"].view(np.uint32)).mean() > 0.999
    for l, (g, w) in enumerate(zip(*levels), 1):
        assert g.shape == (outs[0]["radiance"].shape[0] >> l, outs[0]["radiance"].shape[1] >> l, 3)
        assert same_bits(g, w), "level %d" % l
    # the switch really changes the image (where a pixel sends a reflection ray: every material of light_shafts has roughness 1) ...
    R = make(scene, W, H, hip, atlas=atlas, probes=SMALL_PROBES)
    R.Set_Option(B.OPT_FRAME_OVERLAP, overlap)
    R.Set_Option(B.OPT_INDIRECT_SPECULAR, spec)
    plain = snapshot(R, 3)
    assert same_bits(plain["image"], outs[0]["image"]) == (scene == "light_shafts")
    # ... and switching it off again gives the plain frames back
    R2 = make(scene, W, H, hip, atlas=atlas, probes=SMALL_PROBES)
    R2.Set_Option(B.OPT_FRAME_OVERLAP, overlap)
    R2.Set_Option(B.OPT_INDIRECT_SPECULAR, spec)
    R2.Set_Option(B.OPT_RADIANCE_MIPS, 1)
    R2.Render()
    R2.Set_Option(B.OPT_RADIANCE_MIPS, 0)
    R3 = make(scene, W, H, hip, atlas=atlas, probes=SMALL_PROBES)
    R3.Set_Option(B.OPT_FRAME_OVERLAP, overlap)
    R3.Set_Option(B.OPT_INDIRECT_SPECULAR, spec)
    R3.Render()
    assert same_bits(snapshot(R2, 2)["image"], snapshot(R3, 2)["image"])
    R4 = make(scene, W, H, hip, atlas=atlas, probes=ODD_PROBES)
    with pytest.raises(B.MadarchError):
        R4.Set_Option(B.OPT_RADIANCE_MIPS, 1)  # 12 texels per probe


@pytest.mark.parametrize("atlas,overlap,probes", [(0, 2, SMALL_PROBES), (1, 0, ODD_PROBES), (1, 2, SMALL_PROBES)])
def test_hysteresis_blends_with_the_previous_frames_irradiance(hip, orc, atlas, overlap, probes):
    """MDH_OPT_HYSTERESIS_PERMILLE (a deviation the survey lists, off by default): stored = mix (fresh, previous, h),
    `previous` being the other atlas set when frames are in flight and the texel itself when the pass runs in place."""
    outs = []
    for b in (hip, orc):
        R = make("global_illumination", 56, 40, b, atlas=atlas, probes=probes)
        assert R.Get_Option(B.OPT_HYSTERESIS_PERMILLE) == 0
        R.Set_Option(B.OPT_HYSTERESIS_PERMILLE, 850)
        R.Set_Option(B.OPT_FRAME_OVERLAP, overlap)
        outs.append(snapshot(R, 5))
    assert_parity(*outs)
    plain = snapshot(make("global_illumination", 56, 40, hip, atlas=atlas, probes=probes), 5)
    assert not same_bits(plain["irradiance"], outs[0]["irradiance"])
    # five frames at h = 0.85 have let in 1 - 0.85^5 = 56 % of the light: dimmer than the unblended atlas
    assert outs[0]["irradiance"].mean() < plain["irradiance"].mean()
    with pytest.raises(B.MadarchError):
        make("global_illumination", 16, 8, hip).Set_Option(B.OPT_HYSTERESIS_PERMILLE, 1000)


MANY_PROBES = renderers.Probe_Settings(Radiance_Resolution=8, Irradiance_Resolution=4, Probe_Count=(80, 64), Grid_Dimensions=(16, 16, 20),
                                       Grid_Spacing=(0.45, 0.35, 0.5))


# tiles of 12 x 12 texels (no power of two) in a launch of more than one wavefront per SIMD of an MI355X
BIG_ODD_PROBES = renderers.Probe_Settings(Radiance_Resolution=12, Irradiance_Resolution=6, Probe_Count=(25, 20), Grid_Dimensions=(10, 10, 5),
                                          Grid_Spacing=(0.7, 0.7, 1.4))


@pytest.mark.parametrize("probes,world", [(examples.GI_8X8X8_PROBES, 1), (BIG_ODD_PROBES, 1), (ODD_PROBES, 1), (MANY_PROBES, 1), (examples.GI_8X8X8_PROBES, 2)])
def test_radiance_ray_order_changes_no_texel(hip, probes, world):
    """MDH_OPT_RADIANCE_ORDER: from the second frame on the radiance pass takes its rays sorted by the previous frame's
    primary-march lengths (power-of-two and other tile sizes, a rank's slice, a light that moves between frames so that
    the order is stale): WHICH lane computes a texel, never what it holds -- atlases and image are the bits of the pass
    in probe order, frame after frame."""
    from madarch_amd.lights import spot_lights
    outs = []
    for order in (1, 0):
        R = make("global_illumination", 64, 40, hip, probes=probes)
        assert R.Get_Option(B.OPT_RADIANCE_ORDER) == 1
        R.Set_Option(B.OPT_RADIANCE_ORDER, order)
        if world > 1:
            R.Set_Option(B.OPT_WORLD, world)
            R.Set_Option(B.OPT_RANK, 1)
        frames = []
        for f in range(4):
            if f == 2:
                R.Set_Light(1, spot_lights.Spot_Light, spot_lights.Create((3.5, 5.0, 2.0), (-1.0, 0.0, 0.0), 3.1415 / 4.0, (0.9, 0.9, 0.8)))
            frames.append(snapshot(R, 1))
        outs.append(frames)
    for a, b in zip(*outs):
        for key in ("radiance", "irradiance", "image"):
            assert same_bits(a[key], b[key]), key


@pytest.mark.parametrize("scene,mode,world,overlap", [("global_illumination", 0, 1, 2), ("simple_scene", 2, 1, 2), ("global_illumination", 0, 3, 2),
                                                     ("simple_scene", 2, 1, 0), ("light_shafts", 0, 1, 2)])
def test_screen_tile_order_changes_no_pixel(hip, scene, mode, world, overlap):
    """MDH_OPT_SCREEN_ORDER: the first screen pass records every tile's wavefront duration, later passes start the tiles
    slowest first (image large enough for the option to engage: 2048 tiles and more; frames in flight on both screen
    streams and serial; a rank's tiles; a camera that moves, so that the tiles are sorted again while frames are in
    flight): WHERE in the launch a tile is drawn, never what it holds -- framebuffer, geometry buffer and window pixels
    are those of the pass in image order, frame after frame."""
    outs = []
    for order in (1, 0):
        # 65 x 41 = 2665 tiles, the last column and row partial (three ranks: 129 x 81 tiles, 3483 of them this rank's)
        R = make(scene, 520 * (2 if world > 1 else 1) - (8 if world > 1 else 0), 328 * (2 if world > 1 else 1) - (8 if world > 1 else 0), hip, mode=mode, probes=SMALL_PROBES)
        assert R.Get_Option(B.OPT_SCREEN_ORDER) == 1
        R.Set_Option(B.OPT_SCREEN_ORDER, order)
        R.Set_Option(B.OPT_FRAME_OVERLAP, overlap)
        if world > 1:
            R.Set_Option(B.OPT_WORLD, world)
            R.Set_Option(B.OPT_RANK, 1)
        frames = []
        for f in range(22):
            if f in (3, 12):
                R.Set_Camera_Position((2.0 + 0.1 * f, 2.0, 0.0))
            R.Render()
            if f in (0, 1, 2, 11, 12, 20, 21):
                R.Swap_Buffers()
                frames.append((R.Read_Framebuffer(), R.Read_Gbuffer(), R.Front_Buffer()))
        outs.append(frames)
        R.Destroy()
    for (img_a, gb_a, px_a), (img_b, gb_b, px_b) in zip(*outs):
        assert same_bits(img_a, img_b)
        if world == 1:  # (the geometry buffer of other ranks' tiles is never written, nor cleared)
            assert all(same_bits(x, y) for x, y in zip(gb_a, gb_b))
        assert (px_a == px_b).all()


@pytest.mark.parametrize("sres,sstep,vres,vstep", [((23, 21), 0.0125, (20, 20, 24), 0.1),   # 192 steps: two chunks of the fold's 128; texel count no multiple of 16
                                                   ((17, 5), 0.3, (24, 16, 12), 0.25),         # ten coarse steps, a ragged froxel volume (no 8x8 tiles)
                                                   ((40, 40), 0.007, (16, 16, 30), 0.1)])      # 429 steps: four chunks
def test_scattering_steps_beyond_one_chunk(hip, orc, sres, sstep, vres, vstep):
    """The scattering pass's two kernels (k_scat_march inside the visibility launch, k_scat_fold: a texel's steps spread over lanes
    in chunks of 128, folded in step order) for step counts and texture sizes the default settings never reach; froxel and
    scattering textures whole, bit for bit, serial frames and frames in flight, and a scattering pass outside a frame."""
    vol = renderers.Volumetrics_Settings(Visibility_Resolution=vres, Visibility_Step_Size=vstep, Scattering_Resolution=sres, Scattering_Step_Size=sstep)
    outs = []
    for b in (hip, orc):
        R = make("light_shafts", 72, 40, b, probes=SMALL_PROBES, Volumetrics=vol)
        out = snapshot(R, 3)
        # single passes outside a frame after a camera move: the scattering pass marches its own camera rays then
        R.Set_Camera_Position((2.5, 2.5, -0.5))
        R.Render_Pass(B.PASS_VISIBILITY)
        R.Render_Pass(B.PASS_SCATTERING)
        out["visibility2"], out["scattering2"] = R.Read_Texture(B.TEX_VISIBILITY), R.Read_Texture(B.TEX_SCATTERING)
        outs.append(out)
    assert_parity(*outs)
    for k in ("visibility2", "scattering2"):
        assert same_bits(outs[0][k], outs[1][k]), k
    assert outs[1]["scattering"][..., :3].max() > 0


@pytest.mark.parametrize("method", [0, 1, 2])
def test_partition_build_with_more_instances_than_a_wavefront(hip, orc, method):
    """k_partition_build (one wavefront per cell; lanes = the instances of a kind, 64 at a time) for kinds with more instances than
    lanes and cells whose pre-candidates run into hundreds: 150 spheres and 70 boxes in a coarse grid, all three builders, the tables
    and warning counts bit for bit, and frames through the general (not the small-scene) bit lookup."""
    from madarch_amd import materials, scenes, windows
    from madarch_amd.lights import point_lights
    from madarch_amd.primitives import boxes, planes, spheres
    rng = np.random.RandomState(7)
    cs = (rng.rand(150, 3) * np.array([7.0, 7.0, 12.0]) + np.array([-0.5, -0.5, -5.5])).astype(np.float32)
    cb = (rng.rand(70, 3) * np.array([7.0, 7.0, 12.0]) + np.array([-0.5, -0.5, -5.5])).astype(np.float32)
    outs = []
    for b in (hip, orc):
        part = scenes.Partitioning_Settings(Enable=True, Index_Count=12, Grid_Dimensions=(3, 3, 4), Grid_Spacing=(3.0, 3.0, 3.5), Grid_Offset=(-1.0, -1.0, -6.0))
        scene = scenes.Compile([(spheres.Sphere, 160), (planes.Plane, 8), (boxes.Box, 70)], [(point_lights.Point_Light, 2)], Partitioning=part)
        R = renderers.Create(windows.Open(48, 32), scene, Probes=SMALL_PROBES, Volumetrics=renderers.No_Volumetrics, Binding=b)
        for m, alb in enumerate(((0.8, 0.8, 0.8), (0.9, 0.1, 0.1), (0.1, 0.1, 0.9))):
            R.Set_Material(m, materials.Create(alb, 0.0, 0.6))
        for n, o in (((0, 1, 0), 1.0), ((0, -1, 0), 7.0), ((1, 0, 0), 1.0), ((-1, 0, 0), 7.0), ((0, 0, 1), 6.0), ((0, 0, -1), 7.0)):
            R.Add_Primitive(planes.Plane, planes.Create(n, o, 0))
        for i, c in enumerate(cs):
            R.Add_Primitive(spheres.Sphere, spheres.Create(tuple(float(v) for v in c), 0.12 + 0.02 * (i % 5), 1 + i % 2))
        for i, c in enumerate(cb):
            R.Add_Primitive(boxes.Box, boxes.Create(tuple(float(v) for v in c), (0.1 + 0.03 * (i % 4), 0.15, 0.1), 1 + i % 2))
        R.Set_Light(1, point_lights.Point_Light, point_lights.Create((3.0, 6.0, 0.0), (0.9, 0.9, 0.9)))
        R.Set_Camera_Position((3.0, 3.0, -4.0))
        R.Set_Option(B.OPT_GBUFFER, 1)
        R.Update_Partitioning(method)
        out = snapshot(R, 2)
        out["partition"], out["warnings"] = R.Read_Partitioning(), np.array([R.Partition_Warnings()])
        outs.append(out)
    assert same_bits(outs[0]["partition"], outs[1]["partition"]) and same_bits(outs[0]["warnings"], outs[1]["warnings"])
    if method != 0:
        assert outs[1]["warnings"][0] > 0  # Index_Count = 12 is too small for what the fast builders collect in cells of this grid
    assert_parity(*outs)


@pytest.mark.parametrize("scene,mode,world", [("global_illumination", 0, 1), ("simple_scene", 0, 1), ("simple_scene", 2, 1), ("light_shafts", 0, 1),
                                              ("global_illumination", 0, 3), ("simple_scene", 1, 1)])
def test_split_tiles_change_no_pixel(hip, scene, mode, world):
    """MDH_OPT_SCREEN_SPLIT: a screen launch that leaves the chip's wavefront slots empty gives every 8x8 tile to two
    wavefronts of 8x4 pixels or four of 4x4 (the other lanes idle).  HOW MANY wavefronts draw a tile, never what it holds:
    framebuffer, geometry buffer and window pixels are those of one wavefront per tile -- both split factors, partial
    tiles at the right and lower edge, frames in flight; a rank's tiles of a sharded frame are never split (the same
    pixels whatever the limit)."""
    outs = []
    # 27 x 18 = 486 tiles, the last column and row partial: 4 x 486 <= 2560 (quadrants), 2 x 486 <= 1000 (halves), 0 (whole tiles)
    for limit in (0, 2560, 1000):
        R = make(scene, 212, 140, hip, mode=mode, probes=SMALL_PROBES)
        assert R.Get_Option(B.OPT_SCREEN_SPLIT) == 2560
        R.Set_Option(B.OPT_SCREEN_SPLIT, limit)
        if world > 1:
            R.Set_Option(B.OPT_WORLD, world)
            R.Set_Option(B.OPT_RANK, 1)
        frames = []
        for f in range(4):
            if f == 2:
                R.Set_Camera_Position((2.2, 2.0, 0.0))
            R.Render()
            R.Swap_Buffers()
            frames.append((R.Read_Framebuffer(), R.Read_Gbuffer(), R.Front_Buffer()))
        outs.append(frames)
        R.Destroy()
    for other in outs[1:]:
        for (img_a, gb_a, px_a), (img_b, gb_b, px_b) in zip(outs[0], other):
            assert same_bits(img_a, img_b)
            if world == 1:  # (the geometry buffer of other ranks' tiles is never written, nor cleared)
                assert all(same_bits(x, y) for x, y in zip(gb_a, gb_b))
            assert (px_a == px_b).all()
    with pytest.raises(B.MadarchError):
        make(scene, 16, 16, hip, mode=mode, probes=SMALL_PROBES).Set_Option(B.OPT_SCREEN_SPLIT, -1)


@pytest.mark.parametrize("scene", ["global_illumination", "light_shafts", "simple_scene"])
def test_census_variants_follow_the_scene(hip, orc, scene):
    """The kernels exist once more for the censuses the reference's scenes have (MDH_PF_ROOM: every plane folded, one sphere,
    one box; MDH_PF_PSMALL: the partition's small form), chosen per pass from the committed scene.  Frames in flight while
    the rooms LEAVE their census -- a second sphere, then a second box: the general kernels take over between two passes --
    and while simple_scene is edited and its tables rebuilt inside its own: every stage is the oracle's."""
    from madarch_amd.primitives import boxes, spheres
    runs = []
    for b in (hip, orc):
        R = make(scene, 88, 56, b, probes=SMALL_PROBES)
        shots = [snapshot(R, 3)]
        if scene == "simple_scene":  # (stays within the small form: its variant under edits and rebuilt tables)
            R.Set_Primitive(spheres.Sphere, 1, spheres.Create((1.5, 1.0, 4.5), 0.6, 1))  # (every declared instance is in use: moved, not added)
            R.Update_Partitioning(Method=renderers.GPU_Fast)
            shots.append(snapshot(R, 3))
            R.Set_Primitive(boxes.Box, 1, boxes.Create((4.5, 0.5, 1.5), (0.5, 0.5, 0.5), 2))
            R.Update_Partitioning(Method=renderers.CPU_Fast)
            shots.append(snapshot(R, 3))
        else:
            s2 = R.Add_Primitive(spheres.Sphere, spheres.Create((1.5, 1.0, 4.5), 0.6, 1))
            shots.append(snapshot(R, 3))  # two spheres: no room any more
            b2 = R.Add_Primitive(boxes.Box, boxes.Create((4.5, 0.5, 1.5), (0.5, 0.5, 0.5), 2))
            shots.append(snapshot(R, 3))
            assert s2 is not None and b2 is not None
        runs.append(shots)
        R.Destroy()
    for got, want in zip(*runs):
        assert_parity(got, want)
    assert not same_bits(runs[1][0]["image"], runs[1][1]["image"])  # (the edits are visible)
