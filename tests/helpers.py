"""Shared helpers of the parity tests."""
import numpy as np

from madarch_amd import _binding as B
from madarch_amd import examples, renderers

SEED = 0x4D414441  # "MADA", SURVEY.md section 8d

SMALL_PROBES = renderers.Probe_Settings(Radiance_Resolution=16, Irradiance_Resolution=8, Probe_Count=(6, 6),
                                        Grid_Dimensions=(4, 3, 3), Grid_Spacing=(2.0, 3.0, 3.0))
# nothing a power of two, more than one wavefront of probes and not a multiple of 64: the divisions by the
# resolutions and the probe counts, the last partial wavefront of the radiance pass
ODD_PROBES = renderers.Probe_Settings(Radiance_Resolution=12, Irradiance_Resolution=6, Probe_Count=(15, 5),
                                      Grid_Dimensions=(5, 5, 3), Grid_Spacing=(1.6, 1.9, 2.5))
SMALL_VOL = renderers.Volumetrics_Settings(Visibility_Resolution=(20, 20, 24), Scattering_Resolution=(24, 24))


def make(scene, W, H, binding, mode=0, atlas=0, probes=None, **kw):
    if scene == "light_shafts":
        kw.setdefault("Volumetrics", SMALL_VOL)
    R = examples.SCENES[scene](W, H, Binding=binding, Probes=probes, **kw)
    R.Set_Option(B.OPT_SCREEN_MODE, mode)
    R.Set_Option(B.OPT_ATLAS_FORMAT, atlas)
    R.Set_Option(B.OPT_GBUFFER, 1)
    return R


def snapshot(R, frames):
    """Render `frames` frames and collect every observable output."""
    for _ in range(frames):
        R.Render()
    out = {"image": R.Read_Framebuffer()}
    out["gb_index"], out["gb_t"], out["gb_steps"] = R.Read_Gbuffer()
    if R.Get_Option(B.OPT_SCREEN_MODE) == 0:
        out["radiance"] = R.Read_Texture(B.TEX_RADIANCE)
        out["irradiance"] = R.Read_Texture(B.TEX_IRRADIANCE)
        if R.Volumetrics.Enabled:
            out["visibility"] = R.Read_Texture(B.TEX_VISIBILITY)
            out["scattering"] = R.Read_Texture(B.TEX_SCATTERING)
    return out


def same_bits(a, b):
    """Equal as numbers, NaN == NaN (the sign of a zero is not compared)."""
    return np.array_equal(np.asarray(a), np.asarray(b), equal_nan=True)


def assert_parity(got, want, rtol=1e-4, atol=1e-5):
    """The parity bar of BASELINE.json: integer buffers and DDGI state bit-exact,
    colours within 1e-4 relative per channel (abs floor 1e-5)."""
    for k in ("gb_index", "gb_steps"):
        assert np.array_equal(got[k], want[k]), k
    assert same_bits(got["gb_t"], want["gb_t"]), "gb_t"
    for k in ("radiance", "irradiance", "visibility", "scattering"):
        if k in want:
            assert same_bits(got[k], want[k]), k
    ok = np.isclose(got["image"], want["image"], rtol=rtol, atol=atol, equal_nan=True)
    assert ok.all(), "image: %d of %d values outside 1e-4 relative" % ((~ok).sum(), ok.size)


def seeded_points(n, lo, hi, seed=SEED):
    rng = np.random.RandomState(seed & 0x7FFFFFFF)
    return (lo + (np.asarray(hi) - lo) * rng.rand(n, 3)).astype(np.float32)
