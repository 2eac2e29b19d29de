--  madarch-scenes-hip.ads -- Madarch.Scenes.HIP: what the HIP body of Madarch.Renderers
--  (ada/madarch-renderers.adb) needs to know about a compiled scene, read from the
--  private part of Madarch.Scenes (hence a child unit):
--
--    * the position of a kind in Scenes.Compile's arguments -- the kind index of
--      include/madarch_hip.h (mdh_add_primitive, mdh_set_light, ...);
--    * the std140 type of one element of a kind's array, for Write_Entity's traversal
--      (madarch/madarch-renderers.adb:335-347);
--    * the mdh_scene_desc of the scene: kind names, components, declared counts and, for
--      every kind that is not one of the library's own six objects (Spheres.Sphere,
--      Planes.Plane, Boxes.Box, Triangles.Triangle, Point_Lights.Point_Light,
--      Spot_Lights.Spot_Light), its expressions as MDH_X programs (Madarch.Exprs.MDH_X) --
--      by identity of the kind object, never by its name.
--
--  One addition to the parent is needed, listed in INTEGRATION.md section 4 and applied by
--  ada/scenes_patch.sh: Scene_Internal keeps the Max_Dist that Compile was called with
--  (the reference bakes it into the GLSL text only, madarch-scenes.adb:1200-1201).
--
--  SOURCE ONLY: never compiled in this pipeline (no Ada toolchain in the image).

with Madarch_HIP;

package Madarch.Scenes.HIP is
   --  0-based position in All_Primitives / All_Lights; Constraint_Error if the scene
   --  was not compiled with that kind
   function Kind_Index (S : Scene; Prim : Primitives.Primitive) return Natural;
   function Kind_Index (S : Scene; Lit : Lights.Light) return Natural;

   --  the struct type of one element of prim_<Name>_array / light_<Name>_array
   --  (Compute_Scene_GPU_Type, madarch-scenes.adb:1268-1345)
   function Get_Primitive_Element_Type
     (S : Scene; Prim : Primitives.Primitive) return GPU_Types.GPU_Type;
   function Get_Light_Element_Type
     (S : Scene; Lit : Lights.Light) return GPU_Types.GPU_Type;

   --  the Count the kind was declared with in Scenes.Compile
   function Declared_Count (S : Scene; Prim : Primitives.Primitive) return Natural;
   function Declared_Count (S : Scene; Lit : Lights.Light) return Natural;

   --  the kinds whose behaviour is hand-written device code in libmadarch_hip.so
   function Is_Library_Kind (Prim : Primitives.Primitive) return Boolean;
   function Is_Library_Kind (Lit : Lights.Light) return Boolean;

   --  The C view of the scene.  The arrays it points to live in Keep until Free:
   --  mdh_create copies everything it needs before it returns.
   type Description is limited private;

   procedure Describe
     (S    : Scene;
      Desc : out Madarch_HIP.Scene_Desc;
      Keep : out Description);

   procedure Free (Keep : in out Description);
private
   type Kind_Decls_Access is access Madarch_HIP.Kind_Decl_Array;
   type Components_Access is access Madarch_HIP.Component_Array;
   type Components_Access_Array is array (Positive range <>) of Components_Access;
   type Components_Access_Array_Access is access Components_Access_Array;

   type Description is limited record
      Prim_Kinds, Light_Kinds : Kind_Decls_Access;
      Prim_Comps, Light_Comps : Components_Access_Array_Access;
   end record;
end Madarch.Scenes.HIP;
