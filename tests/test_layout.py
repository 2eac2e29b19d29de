"""std140 layout: the three independent calculators (host mirror madarch_amd.gpu_types,
CPU oracle, and -- on the GPU box -- libmadarch_hip) against the offsets SURVEY.md section 8d
derives from the reference's rules (support/gpu_types-base.ads:21-37,
gpu_types-structs.adb:11-38, gpu_types-fixed_arrays.adb:17-39, madarch-scenes.adb:1268-1345)."""
import pytest

from madarch_amd import examples, gpu_types, materials
from madarch_amd.lights import point_lights, spot_lights
from madarch_amd.primitives import boxes, planes, spheres, triangles

# (count offset, array offset) per kind, then total_light_count offset and block size
SCENE_LAYOUTS = {
    "simple_scene": ([(0, 16), (656, 672), (992, 1008)], [(1648, 1664)], 1792, 1796),
    "global_illumination": ([(0, 16), (656, 672), (992, 1008)], [(1328, 1344)], 1536, 1540),
    "light_shafts": ([(0, 16), (656, 672), (992, 1008)], [(1328, 1344)], 1472, 1476),
}
ELEMENTS = [  # kind, {component: offset}, element size, array stride
    (spheres.Sphere, {"center": 0, "radius": 12, "material_id": 16}, 20, 32),
    (planes.Plane, {"normal": 0, "offset": 12, "material_id": 16}, 20, 32),
    (boxes.Box, {"center": 0, "side": 16, "material_id": 28}, 32, 32),
    (triangles.Triangle, {"v1": 0, "v2": 16, "v3": 32, "material_id": 44}, 48, 48),
    (point_lights.Point_Light, {"position": 0, "color": 16}, 28, 32),
    (spot_lights.Spot_Light, {"position": 0, "direction": 16, "aperture": 28, "color": 32}, 44, 48),
]


@pytest.mark.parametrize("kind,offsets,size,stride", ELEMENTS, ids=[e[0].name for e in ELEMENTS])
def test_element_layout_host(kind, offsets, size, stride):
    st = gpu_types.struct_of_components(kind.comps)
    assert st.size == size
    assert gpu_types.Fixed_Array(3, st).stride == stride
    for name, off in offsets.items():
        assert st.offset_of(name)[0] == off


def test_materials_and_probes_blocks_host():
    mat = gpu_types.struct_of_components((materials.Albedo, materials.Metallic, materials.Roughness))
    assert [mat.offset_of(n)[0] for n in ("albedo", "metallic", "roughness")] == [0, 12, 16]
    block = gpu_types.Struct([("material_count", gpu_types.Int), ("materials", gpu_types.Fixed_Array(20, mat))])
    assert block.offset_of("materials")[0] == 16 and block.size == 656  # renderers.adb:77-89
    probes = gpu_types.Struct([("probe_count", gpu_types.IVec_2), ("grid_dimensions", gpu_types.IVec_3),
                               ("grid_spacing", gpu_types.Vec_3)])  # renderers.adb:38-41
    assert [probes.offset_of(n)[0] for n in ("probe_count", "grid_dimensions", "grid_spacing")] == [0, 16, 32]
    assert probes.size == 44


def _check_scene(R, name):
    prims, lights, total, size = SCENE_LAYOUTS[name]
    for k, (c, a) in enumerate(prims):
        assert R.Scene_Layout(False, k)[:2] == (c, a)
    for k, (c, a) in enumerate(lights):
        assert R.Scene_Layout(True, k)[:2] == (c, a)
    assert R.Scene_Buffer_Size() == (size, total)
    # the host mirror agrees
    assert R.Scene.GPU_Type.size == size
    assert R.Scene.GPU_Type.offset_of("total_light_count")[0] == total


@pytest.mark.parametrize("name", sorted(SCENE_LAYOUTS))
def test_scene_block_layout_oracle(orc, name):
    R = examples.SCENES[name](16, 16, Binding=orc)
    _check_scene(R, name)


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(SCENE_LAYOUTS))
def test_scene_block_layout_hip(hip, orc, name):
    R = examples.SCENES[name](16, 16, Binding=hip)
    _check_scene(R, name)
    # and the uploaded std140 image is byte-identical to the oracle's
    Ro = examples.SCENES[name](16, 16, Binding=orc)
    assert bytes(R.Read_Scene_Buffer()) == bytes(Ro.Read_Scene_Buffer())
