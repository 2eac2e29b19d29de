"""Every BASELINE.json config at the size it names, HIP path against the CPU oracle (run with -m gpu).

  C1  256x256 simple_scene, primary rays, the oracle evaluating the SDFs with the Madarch.Exprs tree walker
  C2  1920x1080 simple_scene, direct PBR + AO through the space partition
  C3  1920x1080 global_illumination, DDGI 8x8x8
  C4  1920x1080 light_shafts, default volumetrics (100^3 froxels, 250^2 scattering texels)
  C5  4096x4096 global_illumination split over 8 ranks (tiles + probe slices), all ranks on this one GPU

The oracle renders the same frames in full (a 1080p frame takes it about a second on the box's 16 threads);
at 4096^2 it renders one rank's share of the tiles.  Size-independent properties are checked on top."""
import numpy as np
import pytest

from helpers import same_bits
from madarch_amd import _binding as B
from madarch_amd import examples, renderers
from oracle_engine import ORC_OPT_SDF_MODE

pytestmark = pytest.mark.gpu

GI = examples.GI_8X8X8_PROBES


def tile_mask(W, H, world, rank):
    """pixels of the 8x8 tiles t with t mod world == rank (the screen pass's dealing, mdh_kernels.h: k_screen)"""
    ty, tx = np.meshgrid(np.arange(H) // 8, np.arange(W) // 8, indexing="ij")
    return (ty * ((W + 7) // 8) + tx) % world == rank


def assert_pixels(got_img, got_gb, want_img, want_gb, mine=None, bit_equal=0.999):
    """the parity bar on a set of pixels: geometry buffer bit-exact, colours within 1e-4 relative
    (BASELINE.json north_star) and equal to the bit almost everywhere"""
    sel = (lambda a: a[mine]) if mine is not None else (lambda a: a)
    for a, b, name in zip(got_gb, want_gb, ("index", "t", "steps")):
        assert same_bits(sel(a), sel(b)), "geometry buffer: " + name
    g, w = sel(got_img), sel(want_img)
    ok = np.isclose(g, w, rtol=1e-4, atol=1e-5, equal_nan=True)
    assert ok.all(), "%d of %d colour values outside 1e-4 relative" % ((~ok).sum(), ok.size)
    assert (g.view(np.uint32) == w.view(np.uint32)).mean() > bit_equal


# ------------------------------------------------------------------------------------ C1
def test_c1_256sq_primary_rays_against_the_exprs_evaluator(hip, orc):
    """BASELINE config 1 as written: 256x256 simple_scene, primary rays only (screen mode 1), the CPU side
    evaluating every SDF and normal by walking the Madarch.Exprs trees (madarch-exprs.adb:322-716 through
    Primitives.Eval_Dist, madarch-renderers.adb:499-526) instead of the closed forms."""
    outs = []
    for b, exprs in ((hip, 0), (orc, 1), (orc, 0)):
        R = examples.simple_scene(256, 256, Binding=b)
        R.Set_Option(B.OPT_SCREEN_MODE, 1)
        R.Set_Option(B.OPT_GBUFFER, 1)
        if exprs:
            R.Set_Option(ORC_OPT_SDF_MODE, 1)
        R.Render()
        outs.append((R.Read_Framebuffer(), R.Read_Gbuffer()))
        R.Destroy()
    (img_g, gb_g), (img_x, gb_x), (img_c, gb_c) = outs
    assert_pixels(img_g, gb_g, img_x, gb_x, bit_equal=0.9999)
    # the tree walker and the closed forms are the same arithmetic: bit for bit
    assert same_bits(img_x, img_c) and all(same_bits(a, b) for a, b in zip(gb_x, gb_c))
    assert (gb_g[0] >= 0).all()  # a closed room


# ------------------------------------------------------------------------------------ C2
def test_c2_simple_scene_1080p_direct(hip, orc):
    """config 2 at its size: primary march through the space partition (CPU_Best tables, built on the device),
    normals, direct PBR with soft shadows, AO."""
    outs = []
    for b in (hip, orc):
        R = examples.simple_scene(1920, 1080, Binding=b)
        R.Set_Option(B.OPT_SCREEN_MODE, 2)
        R.Set_Option(B.OPT_GBUFFER, 1)
        R.Render()
        outs.append((R.Read_Framebuffer(), R.Read_Gbuffer(), np.asarray(R.Read_Partitioning())))
        R.Destroy()
    (img_g, gb_g, part_g), (img_o, gb_o, part_o) = outs
    assert same_bits(part_g, part_o)
    assert_pixels(img_g, gb_g, img_o, gb_o)
    assert len(np.unique(gb_g[0])) > 20  # planes, most spheres and boxes are in view


def test_simple_scene_1080p_full(hip, orc):
    """The reference's simple_scene AS IT RUNS (examples/simple_scene/main.adb:36-39,122 with the renderer's fixed screen
    macros, madarch-renderers.adb:136-143): full pixel_color_probes -- direct specular, mode-2 indirect specular, 3
    occlusion steps -- THROUGH the space partition (CPU_Best) with the default 36 probes (renderers.ads:23-29), at 1080p:
    two frames of probe feedback, both atlases and the geometry buffer to the bit, the image within 1e-4."""
    outs = []
    for b in (hip, orc):
        R = examples.simple_scene(1920, 1080, Binding=b)
        R.Set_Option(B.OPT_GBUFFER, 1)
        for _ in range(2):
            R.Render()
        outs.append((R.Read_Framebuffer(), R.Read_Gbuffer(), R.Read_Texture(B.TEX_RADIANCE), R.Read_Texture(B.TEX_IRRADIANCE),
                     np.asarray(R.Read_Partitioning())))
        R.Destroy()
    (img_g, gb_g, rad_g, irr_g, part_g), (img_o, gb_o, rad_o, irr_o, part_o) = outs
    assert same_bits(part_g, part_o)
    assert same_bits(rad_g, rad_o), "radiance atlas"
    assert same_bits(irr_g, irr_o), "irradiance atlas"
    assert rad_o.max() > 0 and irr_o.max() > 0
    assert_pixels(img_g, gb_g, img_o, gb_o)
    assert len(np.unique(gb_g[0])) > 20


def test_global_illumination_1080p_default_probes(hip, orc):
    """global_illumination with the reference's OWN probe settings (4x3x3 probes as a 6x6 atlas, spacing (2, 3, 3):
    madarch-renderers.ads:23-29; BASELINE config 3 names an 8x8x8 grid instead) at 1080p."""
    outs = []
    for b in (hip, orc):
        R = examples.global_illumination(1920, 1080, Binding=b)
        R.Set_Option(B.OPT_GBUFFER, 1)
        for _ in range(2):
            R.Render()
        outs.append((R.Read_Framebuffer(), R.Read_Gbuffer(), R.Read_Texture(B.TEX_RADIANCE), R.Read_Texture(B.TEX_IRRADIANCE)))
        R.Destroy()
    (img_g, gb_g, rad_g, irr_g), (img_o, gb_o, rad_o, irr_o) = outs
    assert same_bits(rad_g, rad_o) and same_bits(irr_g, irr_o)
    assert_pixels(img_g, gb_g, img_o, gb_o)


# ------------------------------------------------------------------------------------ C3
@pytest.fixture(scope="module")
def gi_oracle(orc):
    """The oracle's DDGI state after two frames of the global_illumination scene with the 8x8x8 grid.  The probe
    passes depend on the scene and the probe settings only (renderers.adb:306-308 sets the camera on the screen and
    volumetric programs alone), so configs 3 and 5 share it."""
    Ro = examples.global_illumination(1920, 1080, Probes=GI, Binding=orc)
    Ro.Set_Option(B.OPT_GBUFFER, 1)
    for _ in range(2):
        Ro.Render_Pass(B.PASS_RADIANCE)
        Ro.Render_Pass(B.PASS_IRRADIANCE)
    return Ro, Ro.Read_Texture(B.TEX_RADIANCE), Ro.Read_Texture(B.TEX_IRRADIANCE)


@pytest.fixture(scope="module")
def full_size(hip):
    R = examples.global_illumination(1920, 1080, Probes=GI, Binding=hip)
    R.Set_Option(B.OPT_GBUFFER, 1)
    for _ in range(2):
        R.Render()
    return R


def test_c3_global_illumination_1080p(full_size, gi_oracle):
    """config 3 (the headline): both atlases whole and the whole 1920x1080 frame."""
    Rg = full_size
    Ro, rad_o, irr_o = gi_oracle
    assert same_bits(Rg.Read_Texture(B.TEX_IRRADIANCE), irr_o)
    assert same_bits(Rg.Read_Texture(B.TEX_RADIANCE), rad_o)
    Ro.Render_Pass(B.PASS_SCREEN)
    assert_pixels(Rg.Read_Framebuffer(), Rg.Read_Gbuffer(), Ro.Read_Framebuffer(), Ro.Read_Gbuffer())


def test_c3_properties(full_size, hip):
    """Size-independent properties on the whole 1080p frame: rendering is deterministic; the
    image does not depend on how tiles are dealt to ranks; every pixel of a closed room hits;
    the tonemapped image is in [0, 1]."""
    Rg = full_size
    img = Rg.Read_Framebuffer()
    idx, t, steps = Rg.Read_Gbuffer()
    assert (idx >= 0).all() and (steps >= 1).all()
    finite = np.isfinite(img)
    assert finite.mean() > 0.9999
    assert (img[finite] >= 0).all() and (img[finite] <= 1).all()
    # determinism + tile dealing: 3 'ranks' render their tiles of the same frame from the same atlases
    Rg.Render_Pass(B.PASS_SCREEN)
    assert same_bits(Rg.Read_Framebuffer(), img)
    acc = np.zeros_like(img)
    for r in range(3):
        Rg.Set_Option(B.OPT_WORLD, 3)
        Rg.Set_Option(B.OPT_RANK, r)
        Rg.Render_Pass(B.PASS_SCREEN)
        acc += Rg.Read_Framebuffer()
    Rg.Set_Option(B.OPT_RANK, 0)
    Rg.Set_Option(B.OPT_WORLD, 1)
    assert same_bits(acc, img)


# ------------------------------------------------------------------------------------ C4
def test_c4_light_shafts_1080p_default_volumetrics(hip, orc):
    """config 4 at its size with the volumetric settings the config names (the reference's defaults,
    madarch-renderers.ads:33-41): 100 x 100 x 100 froxels, 250 x 250 scattering texels, the default 4x3x3 probes.
    Froxel and scattering textures whole, atlases whole, the whole frame."""
    outs = []
    for b in (hip, orc):
        R = examples.light_shafts(1920, 1080, Binding=b)
        assert R.Volumetrics.Visibility_Resolution == (100, 100, 100) and R.Volumetrics.Scattering_Resolution == (250, 250)
        R.Set_Option(B.OPT_GBUFFER, 1)
        for _ in range(2):
            R.Render()
        outs.append({"img": R.Read_Framebuffer(), "gb": R.Read_Gbuffer(), "rad": R.Read_Texture(B.TEX_RADIANCE), "irr": R.Read_Texture(B.TEX_IRRADIANCE),
                     "vis": R.Read_Texture(B.TEX_VISIBILITY), "scat": R.Read_Texture(B.TEX_SCATTERING)})
        R.Destroy()
    g, o = outs
    assert g["vis"].shape == (100 * 100, 100, 3) and g["scat"].shape == (250, 250, 4)
    for k in ("rad", "irr", "vis", "scat"):
        assert same_bits(g[k], o[k]), k
    assert (g["vis"] > 0).mean() > 0.3 and (g["scat"][..., :3] > 0).mean() > 0.9  # lit froxels, fog along the rays
    assert_pixels(g["img"], g["gb"], o["img"], o["gb"])


# ------------------------------------------------------------------------------------ C5
def test_c5_4096sq_split_over_eight_ranks(hip, orc, gi_oracle):
    """config 5 on one GPU: eight renderers stand for the eight ranks -- rank r updates the radiance of probes
    [64 r, 64 r + 64), the slices are exchanged (through the host here, RCCL on a node), every rank folds the
    irradiance, and rank r draws the 8x8 tiles t = r (mod 8) of the 4096x4096 image.  The eight framebuffers add up to
    the frame of a single renderer bit for bit, the atlases of every rank are the oracle's, and the oracle renders
    rank 3's share of the tiles (32768 tiles, 2 M pixels)."""
    N, W, H = 8, 4096, 4096
    _, rad_o, irr_o = gi_oracle
    whole = examples.global_illumination(W, H, Probes=GI, Binding=hip)
    whole.Set_Option(B.OPT_GBUFFER, 1)
    for _ in range(2):
        whole.Render()
    img_w, gb_w = whole.Read_Framebuffer(), whole.Read_Gbuffer()
    assert same_bits(whole.Read_Texture(B.TEX_RADIANCE), rad_o) and same_bits(whole.Read_Texture(B.TEX_IRRADIANCE), irr_o)
    whole.Destroy()
    Rs = [examples.global_illumination(W, H, Probes=GI, Binding=hip) for _ in range(N)]
    for r, R in enumerate(Rs):
        R.Set_Option(B.OPT_WORLD, N)
        R.Set_Option(B.OPT_RANK, r)
        R.Set_Option(B.OPT_GBUFFER, 1)
    P = Rs[0].Probe_Total()
    assert P == 512
    for _ in range(2):
        for R in Rs:
            R.Render_Pass(B.PASS_RADIANCE)
        parts = [R.Read_Atlas_Slice(B.TEX_RADIANCE, P * r // N, P // N) for r, R in enumerate(Rs)]
        for r, R in enumerate(Rs):
            for q in range(N):
                if q != r:
                    R.Write_Atlas_Slice(B.TEX_RADIANCE, P * q // N, parts[q])
        for R in Rs:
            R.Render_Pass(B.PASS_IRRADIANCE)  # MDH_OPT_IRRADIANCE_ALL: every rank, every probe
        for R in Rs:
            R.Render_Pass(B.PASS_SCREEN)
    acc = np.zeros_like(img_w)
    for r, R in enumerate(Rs):
        img = R.Read_Framebuffer()
        mine = tile_mask(W, H, N, r)
        assert not img[~mine].any()  # a rank draws its own tiles only
        acc += img
        assert same_bits(R.Read_Texture(B.TEX_RADIANCE), rad_o) and same_bits(R.Read_Texture(B.TEX_IRRADIANCE), irr_o)
        if r == 3:
            gb3 = R.Read_Gbuffer()
            for a, b in zip(gb3, gb_w):
                assert same_bits(a[mine], b[mine])
    assert same_bits(acc, img_w)
    for R in Rs:
        R.Destroy()
    # the oracle on rank 3's tiles, from the same atlases
    Ro = examples.global_illumination(W, H, Probes=GI, Binding=orc)
    Ro.Set_Option(B.OPT_GBUFFER, 1)
    Ro.Write_Texture(B.TEX_RADIANCE, rad_o)
    Ro.Write_Texture(B.TEX_IRRADIANCE, irr_o)
    Ro.Set_Option(B.OPT_WORLD, N)
    Ro.Set_Option(B.OPT_RANK, 3)
    Ro.Render_Pass(B.PASS_SCREEN)
    mine = tile_mask(W, H, N, 3)
    assert mine.sum() == W * H // N
    assert_pixels(img_w, gb_w, Ro.Read_Framebuffer(), Ro.Read_Gbuffer(), mine)
    Ro.Destroy()
    # properties of the whole 4096^2 frame
    assert (gb_w[0] >= 0).all() and np.isfinite(img_w).mean() > 0.9999
