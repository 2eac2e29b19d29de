"""The three benchmark scenes of BASELINE.json, written as the reference's example
programs write them (same data, same call order), minus the window loop.

  simple_scene        -- reference examples/simple_scene/main.adb:28-122
  global_illumination -- reference examples/global_illumination/main.adb:29-74,149-161
  light_shafts        -- reference examples/light_shafts/main.adb:29-59,140-155

Each returns the Renderer after the last call the example makes before its
first `Renderers.Render`.  `Binding=None` is the HIP library.
"""
import numpy as np

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


class Ball_Game:
    """examples/ball_game/main.adb: the global_illumination room with the default space partition, balls
    thrown from the camera that bounce off planes and boxes.  The physics step is the reference's: per
    ball one Eval_Distance_To against (Plane, Box) -- the CPU expression evaluator there, the batched device
    query here -- then Set_Primitive; every frame rebuilds the partition (Update_Partitioning) and renders.
    The interactive parts (keys, mouse) are the methods Throw_Ball / Move_Camera."""

    Ball_Radius = np.float32(0.2)
    Gravity = np.array([0.0, -9.81, 0.0], dtype=np.float32)

    def __init__(self, Width=1000, Height=1000, Probes=None, Binding=None, Device=0):
        self.Scene = scenes.Compile(  # main.adb:30-35: the default partitioning settings (enabled)
            All_Primitives=[(spheres.Sphere, 20), (planes.Plane, 10), (boxes.Box, 10)],
            All_Lights=[(spot_lights.Spot_Light, 4)])
        R = self.R = renderers.Create(windows.Open(Width, Height, "Ball_Game"), self.Scene, Probes=Probes,
                                      Volumetrics=renderers.No_Volumetrics, Device=Device, Binding=Binding)
        mats = [R.Add_Material(materials.Create(a, m, r)) for a, m, r in (
            ((0.0, 0.0, 0.0), 0.0, 1.0), ((1.0, 0.0, 0.0), 0.0, 1.0), ((0.0, 0.0, 1.0), 0.0, 1.0),
            ((0.1, 0.1, 0.1), 0.9, 0.1), ((0.0, 1.0, 0.0), 0.8, 0.3))]
        self.Box_Mat = mats[4]
        for (n, o), m in zip(_ROOM_PLANES, (mats[0], mats[0], mats[1], mats[2], mats[0], mats[0])):
            R.Add_Primitive(planes.Plane, planes.Create(n, o, m))
        R.Add_Primitive(spheres.Sphere, spheres.Create((3.0, 4.0, 3.0), 1.0, mats[3]))
        R.Add_Primitive(boxes.Box, boxes.Create((3.0, 0.0, 4.0), (1.5, 1.5, 1.5), mats[4]))
        R.Set_Light(1, spot_lights.Spot_Light, spot_lights.Create((3.5, 6.0, 2.0), (0.0, -1.0, 0.0), 3.1415 / 2.0, (1.0, 1.0, 1.0)))
        self.Camera_Position = np.array([2.0, 2.0, 0.0], dtype=np.float32)
        self.Camera_Orientation = np.eye(3, dtype=np.float32)
        R.Set_Camera_Position(self.Camera_Position)
        self.Ball_Bodies = []  # [index, position, velocity]

    def Move_Camera(self, Offset):  # main.adb:110-115
        self.Camera_Position = (self.Camera_Position + self.Camera_Orientation @ np.asarray(Offset, dtype=np.float32)).astype(np.float32)
        self.R.Set_Camera_Position(self.Camera_Position)

    def Throw_Ball(self):  # main.adb:98-108
        vel = (self.Camera_Orientation @ np.array([0.0, 0.0, 1.0], dtype=np.float32) * np.float32(10.0)).astype(np.float32)
        self.R.Add_Primitive(spheres.Sphere, spheres.Create(self.Camera_Position, self.Ball_Radius, self.Box_Mat))
        self.Ball_Bodies.append([len(self.Ball_Bodies) + 2, self.Camera_Position.copy(), vel])

    def Step_Physics(self, Dt=0.01):  # main.adb:196-228
        """One Eval_Distance_To per ball in the reference; the balls do not see each other (the query
        is against planes and boxes only), so all of them go to the device in ONE batched query."""
        if not self.Ball_Bodies:
            return
        Dt = np.float32(Dt)
        new_vel = [(b[2] + self.Gravity * Dt).astype(np.float32) for b in self.Ball_Bodies]
        new_pos = [(b[1] + v * Dt).astype(np.float32) for b, v in zip(self.Ball_Bodies, new_vel)]
        dists, normals = self.R.Eval_Distances_To(np.stack(new_pos), (planes.Plane, boxes.Box))
        for body, vel, pos, dist, normal in zip(self.Ball_Bodies, new_vel, new_pos, dists, normals):
            if np.float32(dist) <= self.Ball_Radius:
                normal = np.asarray(normal, dtype=np.float32)
                d = np.float32((vel[0] * normal[0] + vel[1] * normal[1]) + vel[2] * normal[2])
                vel = (vel - np.float32(2.0) * d * normal).astype(np.float32)  # Math_Utils.Reflect, math_utils.adb:38-42
                pos = body[1]
            self.R.Set_Primitive(spheres.Sphere, body[0], spheres.Create(pos, self.Ball_Radius, self.Box_Mat))
            body[1], body[2] = pos, vel

    def Frame(self, Dt=0.01):  # the loop body, main.adb:244-252
        self.Step_Physics(Dt)
        self.R.Update_Partitioning()
        self.R.Render()


def ball_game(Width=1000, Height=1000, Probes=None, Binding=None, Device=0):
    return Ball_Game(Width, Height, Probes, Binding, Device)


SCENES = {"simple_scene": simple_scene, "global_illumination": global_illumination,
          "light_shafts": light_shafts}
