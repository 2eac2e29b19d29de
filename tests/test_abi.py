"""The C-ABI library loads (no GPU needed) and exports every symbol include/madarch_hip.h
declares; the oracle exports the same operations under the orc_ prefix."""
import ctypes
import os
import re

from madarch_amd import _binding

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "madarch_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mdh_[a-z_0-9]+)\s*\(", text)))


def test_header_declares_the_survey_exports():
    names = declared_functions()
    for required in ("mdh_create", "mdh_destroy", "mdh_set_material", "mdh_set_primitive", "mdh_add_primitive",
                     "mdh_set_light", "mdh_set_camera_position", "mdh_set_camera_orientation",
                     "mdh_update_partitioning", "mdh_render", "mdh_read_framebuffer", "mdh_eval_distance_to",
                     "mdh_last_error"):
        assert required in names


def test_hip_library_exports_every_declared_symbol():
    assert os.path.exists(_binding.HIP_LIBRARY), "build first: python -c 'import __graft_entry__ as g; g.build()'"
    lib = ctypes.CDLL(_binding.HIP_LIBRARY)
    missing = [n for n in declared_functions() if not hasattr(lib, n)]
    assert not missing, missing
    b = _binding.hip_binding()
    assert b.version().startswith(b"madarch-hip")


def test_binding_table_matches_header():
    declared = set(declared_functions())
    bound = {"mdh_" + n for n in list(_binding.ABI) + list(_binding.HIP_ONLY_ABI)}
    assert bound == declared


def test_oracle_exports_the_shared_operations(orc):
    for name in _binding.ABI:
        assert hasattr(orc.lib, "orc_" + name)


def test_no_device_is_an_error_not_a_fallback(hip):
    """Without a GPU mdh_create must fail with MDH_E_NO_DEVICE (the product has no CPU path)."""
    import pytest
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    from madarch_amd import examples
    with pytest.raises(_binding.MadarchError) as e:
        examples.global_illumination(16, 16, Binding=hip)
    assert e.value.status == _binding.MDH_E_NO_DEVICE
