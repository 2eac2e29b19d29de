"""The three benchmark scenes of BASELINE.json, written as the reference's example
programs write them (same data, same call order), minus the window loop.

  simple_scene        -- reference examples/simple_scene/main.adb:28-122
  global_illumination -- reference examples/global_illumination/main.adb:29-74,149-161
  light_shafts        -- reference examples/light_shafts/main.adb:29-59,140-155

Each returns the Renderer after the last call the example makes before its
first `Renderers.Render`.  `Binding=None` is the HIP library.
"""
from . import lights, materials, primitives, renderers, scenes, windows
from .lights import point_lights, spot_lights
from .primitives import boxes, planes, spheres

# BASELINE config 3: DDGI 8x8x8 probe grid = 512 probes as a 32x16 atlas; the grid
# has no offset (glsl/probe_utils.glsl:38-40), this spacing keeps it in the room
# 
GI_8X8X8_PROBES = renderers.Probe_Settings(Probe_Count=(32, 16), Grid_Dimensions=(8, 8, 8),
                                           Grid_Spacing=(0.95, 0.95, 0.9))

_ROOM_PLANES = (((0.0, 1.0, 0.0), 1.0), ((0.0, -1.0, 0.0), 7.0), ((1.0, 0.0, 0.0), 1.0),
                ((-1.0, 0.0, 0.0), 7.0), ((0.0, 0.0, 1.0), 6.0), ((0.0, 0.0, -1.0), 7.0))


def simple_scene(Width=1000, Height=1000, Probes=None, Binding=None, Device=0,
                 Partitioning_Method=renderers.CPU_Best):
    Scene = scenes.Compile(
        All_Primitives=[(spheres.Sphere, 20), (planes.Plane, 10), (boxes.Box, 20)],
        All_Lights=[(point_lights.Point_Light, 4)])
    Window = windows.Open(Width, Height, "Simple_Scene")
    R = renderers.Create(Window, Scene, Probes=Probes, Volumetrics=renderers.No_Volumetrics,
                         Device=Device, Binding=Binding)
    Point_Light_Instance = point_lights.Create((0.0, 3.0, 0.0), (0.9, 0.9, 0.9))
    plane_mats = (0, 0, 1, 2, 0, 0)
    Planes = [planes.Create(n, o, m) for (n, o), m in zip(_ROOM_PLANES, plane_mats)]
    Spheres = [spheres.Create((x, y, z), 0.5, 3) for (y, z, xs) in (
        (3.5, 2.0, (0.5, 1.5, 2.5, 3.5, 4.5, 5.5)), (0.5, 2.0, (0.5, 1.5, 2.5, 3.5, 4.5, 5.5)),
        (3.5, 5.0, (0.5, 1.5, 2.5, 3.5, 4.5, 5.5)), (0.5, 5.0, (0.5, 1.5))) for x in xs]
    Boxes = [boxes.Create(c, s, 2) for c, s in (
        ((3.0, 1.0, 2.0), (0.5, 0.5, 0.5)), ((0.0, 1.0, 2.0), (0.3, 0.3, 0.5)),
        ((3.0, 1.0, 4.0), (0.5, 0.5, 0.5)), ((4.0, 2.0, 2.0), (0.5, 0.5, 0.5)),
        ((2.0, 2.0, 2.0), (0.5, 0.5, 0.5)), ((1.0, 1.0, 6.0), (0.5, 0.5, 0.5)),
        ((3.0, 1.0, 6.0), (0.5, 0.5, 0.5)), ((3.0, 1.0, -2.0), (0.5, 0.5, 0.5)),
        ((1.0, 1.0, -2.0), (0.3, 0.3, 0.5)), ((3.0, 1.0, -4.0), (0.5, 0.5, 0.5)),
        ((4.0, 2.0, -2.0), (0.5, 0.5, 0.5)), ((2.0, 2.0, -2.0), (0.5, 0.5, 0.5)),
        ((1.0, 1.0, -6.0), (0.5, 0.5, 0.5)), ((3.0, 1.0, -6.0), (0.5, 0.5, 0.5)))]
    for Plane in Planes:
        R.Add_Primitive(planes.Plane, Plane)
    for Sphere in Spheres:
        R.Add_Primitive(spheres.Sphere, Sphere)
    for Box in Boxes:
        R.Add_Primitive(boxes.Box, Box)
    R.Set_Material(0, materials.Create((0.0, 0.0, 0.0), 0.0, 0.6))
    R.Set_Material(1, materials.Create((1.0, 0.0, 0.0), 0.0, 0.6))
    R.Set_Material(2, materials.Create((0.0, 0.0, 1.0), 0.0, 0.6))
    R.Set_Material(3, materials.Create((0.1, 0.1, 0.1), 0.9, 0.1))
    R.Set_Light(1, point_lights.Point_Light, Point_Light_Instance)
    R.Set_Camera_Position((2.0, 2.0, 0.0))
    if Partitioning_Method is not None:
        R.Update_Partitioning(Method=Partitioning_Method)
    return R


def global_illumination(Width=1000, Height=1000, Probes=None, Binding=None, Device=0):
    Scene = scenes.Compile(
        All_Primitives=[(spheres.Sphere, 20), (planes.Plane, 10), (boxes.Box, 10)],
        All_Lights=[(spot_lights.Spot_Light, 4)],
        Partitioning=scenes.Partitioning_Settings(Enable=False))
    Window = windows.Open(Width, Height, "Global_Illumination")
    R = renderers.Create(Window, Scene, Probes=Probes, Volumetrics=renderers.No_Volumetrics,
                         Device=Device, Binding=Binding)
    Spot_Light_Instance = spot_lights.Create((3.5, 5.0, 2.0), (1.0, 0.0, 0.0), 3.1415 / 4.0,
                                             (0.9, 0.9, 0.8))
    Wall_Mat_1 = R.Add_Material(materials.Create((0.0, 0.0, 0.0), 0.0, 0.6))
    Wall_Mat_2 = R.Add_Material(materials.Create((1.0, 0.0, 0.0), 0.0, 0.6))
    Wall_Mat_3 = R.Add_Material(materials.Create((0.0, 0.0, 1.0), 0.0, 0.6))
    Sphere_Mat = R.Add_Material(materials.Create((0.1, 0.1, 0.1), 0.9, 0.1))
    Box_Mat = R.Add_Material(materials.Create((0.0, 1.0, 0.0), 0.8, 0.3))
    plane_mats = (Wall_Mat_1, Wall_Mat_1, Wall_Mat_2, Wall_Mat_3, Wall_Mat_1, Wall_Mat_1)
    for (n, o), m in zip(_ROOM_PLANES, plane_mats):
        R.Add_Primitive(planes.Plane, planes.Create(n, o, m))
    R.Add_Primitive(spheres.Sphere, spheres.Create((3.0, 4.0, 3.0), 1.0, Sphere_Mat))
    R.Add_Primitive(boxes.Box, boxes.Create((3.0, 0.0, 4.0), (1.5, 1.5, 1.5), Box_Mat))
    R.Set_Camera_Position((2.0, 2.0, 0.0))  # Move_Camera with a zero offset, main.adb:80-85
    R.Set_Light(1, spot_lights.Spot_Light, Spot_Light_Instance)
    return R


def light_shafts(Width=1000, Height=1000, Probes=None, Volumetrics=None, Binding=None, Device=0):
    Scene = scenes.Compile(
        All_Primitives=[(spheres.Sphere, 20), (planes.Plane, 10), (boxes.Box, 10)],
        All_Lights=[(point_lights.Point_Light, 4)],
        Partitioning=scenes.Partitioning_Settings(Enable=False))
    Window = windows.Open(Width, Height, "Light_Shafts")
    R = renderers.Create(Window, Scene, Probes=Probes, Volumetrics=Volumetrics, Device=Device,
                         Binding=Binding)
    Point_Light_Instance = point_lights.Create((5.0, 3.0, 6.0), (0.9, 0.9, 0.9))
    plane_mats = (0, 0, 1, 2, 0, 0)
    for (n, o), m in zip(_ROOM_PLANES, plane_mats):
        R.Add_Primitive(planes.Plane, planes.Create(n, o, m))
    R.Add_Primitive(spheres.Sphere, spheres.Create((3.0, 4.0, 3.0), 1.0, 3))
    R.Add_Primitive(boxes.Box, boxes.Create((3.0, 0.0, 4.0), (1.5, 1.5, 1.5), 2))
    R.Set_Material(0, materials.Create((0.0, 0.0, 0.0), 0.0, 1.0))
    R.Set_Material(1, materials.Create((1.0, 0.0, 0.0), 0.0, 1.0))
    R.Set_Material(2, materials.Create((0.0, 1.0, 0.0), 0.0, 1.0))
    R.Set_Material(3, materials.Create((0.0, 0.0, 1.0), 0.0, 1.0))
    R.Set_Camera_Position((2.0, 2.0, 0.0))
    R.Set_Light(1, point_lights.Point_Light, Point_Light_Instance)
    return R


SCENES = {"simple_scene": simple_scene, "global_illumination": global_illumination,
          "light_shafts": light_shafts}
