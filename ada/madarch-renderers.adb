--  madarch-renderers.adb -- replacement body of Madarch.Renderers over
--  libmadarch_hip.so (MI355X).  The SPEC's public part
--  (madarch/madarch-renderers.ads:21-97) stays as it is; its private part shrinks to
--
--     type Renderer_Internal is record
--        Window : Windows.Window;
--        Scene  : Scenes.Scene;
--        Handle : Madarch_HIP.Handle;
--        Last_Material_Index : Materials.Id := 0;   --  as in the reference's record
--     end record;
--
--  SOURCE ONLY: written against GNAT conventions but never compiled in this
--  pipeline (no Ada toolchain in the image).  Each subprogram names the lines of the
--  OpenGL body it replaces.

with Ada.Containers.Vectors;
with Interfaces.C; use Interfaces.C;
with Interfaces.C.Strings;
with System;

with GPU_Types;
with Madarch.Components;
with Madarch.Scenes.HIP;
with Madarch_HIP;

package body Madarch.Renderers is
   package HIP renames Madarch_HIP;

   procedure Check (S : HIP.Status) is
   begin
      if S = HIP.E_Index then
         raise Constraint_Error with Strings.Value (HIP.Last_Error);
      elsif S /= HIP.OK then
         raise Program_Error with Strings.Value (HIP.Last_Error);
      end if;
   end Check;

   --  One std140 element image of an entity: the traversal of Write_Entity
   --  (madarch-renderers.adb:335-347) into a byte buffer instead of glBufferSubData.
   type Byte_Array is array (Natural range <>) of aliased Interfaces.Unsigned_8;

   function Element_Blob
     (Element_Type : GPU_Types.GPU_Type; Ent : Entities.Entity) return Byte_Array
   is
      Blob : Byte_Array (0 .. Natural (Element_Type.Size) - 1) := (others => 0);

      procedure Put (Offset : Natural; Item : System.Address; N : Natural) is
         Src : Byte_Array (0 .. N - 1) with Import, Address => Item;
      begin
         Blob (Offset .. Offset + N - 1) := Src;
      end Put;

      procedure Write_Component (C : Components.Component; V : Values.Value) is
         Off : constant Natural := Natural
           (Element_Type.Address.Component (Components.Get_Name (C)).Offset);
      begin
         case V.Kind is
            when Values.Vector3_Kind => Put (Off, V.Vector3_Value'Address, 12);
            when Values.Float_Kind   => Put (Off, V.Float_Value'Address, 4);
            when Values.Int_Kind     => Put (Off, V.Int_Value'Address, 4);
         end case;
      end Write_Component;
   begin
      Entities.Foreach (Ent, Write_Component'Unrestricted_Access);
      return Blob;
   end Element_Blob;

   --  Renderers.Create: madarch-renderers.adb:91-300.  Scenes.Compile's GLSL output is
   --  not used; the kind names, components and declared counts are handed over instead.
   function Create
     (Window      : Windows.Window;
      Scene       : Scenes.Scene;
      Probes      : Probe_Settings := Default_Probe_Settings;
      Volumetrics : Volumetrics_Settings := Default_Volumetrics_Settings)
      return Renderer
   is
      H    : aliased HIP.Handle;
      Desc : aliased HIP.Scene_Desc;       --  kinds, components, declared counts, MDH_X programs
      Keep : Scenes.HIP.Description;       --  owns what Desc points to until mdh_create has copied it
      S    : HIP.Status;
      P    : aliased HIP.Probe_Settings :=
        (int (Probes.Radiance_Resolution), int (Probes.Irradiance_Resolution),
         (int (Probes.Probe_Count (GL.X)), int (Probes.Probe_Count (GL.Y))),
         (int (Probes.Grid_Dimensions (GL.X)), int (Probes.Grid_Dimensions (GL.Y)),
          int (Probes.Grid_Dimensions (GL.Z))),
         (C_float (Probes.Grid_Spacing (GL.X)), C_float (Probes.Grid_Spacing (GL.Y)),
          C_float (Probes.Grid_Spacing (GL.Z))));
      V    : aliased HIP.Volumetrics :=
        ((if Volumetrics.Enabled then 1 else 0),
         (int (Volumetrics.Visibility_Resolution (GL.X)),
          int (Volumetrics.Visibility_Resolution (GL.Y)),
          int (Volumetrics.Visibility_Resolution (GL.Z))),
         C_float (Volumetrics.Visibility_Step_Size),
         (int (Volumetrics.Scattering_Resolution (GL.X)),
          int (Volumetrics.Scattering_Resolution (GL.Y))),
         C_float (Volumetrics.Scattering_Step_Size));
   begin
      Scenes.HIP.Describe (Scene, Desc, Keep);   --  ada/madarch-scenes-hip.ads
      S := HIP.Create (int (Window.Width), int (Window.Height),
                       Desc'Access, P'Access, V'Access, 0, H'Access);
      Scenes.HIP.Free (Keep);
      Check (S);
      return new Renderer_Internal'
        (Window => Window, Scene => Scene, Handle => H, Last_Material_Index => 0);
   end Create;

   --  madarch-renderers.adb:302-321 (five passes + Swap_Buffers)
   procedure Render (Self : Renderer) is
   begin
      Check (HIP.Render (Self.Handle));
      --  renderers.adb:320: the frame's RGBA8 pixels travel to pinned host memory behind it; no host
      --  wait here, frames stay in flight (HIP.Front_Buffer waits for the last swap and returns them)
      Check (HIP.Swap_Buffers (Self.Handle));
   end Render;

   --  madarch-renderers.adb:349-367
   procedure Set_Material
     (Self : in out Renderer; Index : Materials.Id; Entity : Entities.Entity)
   is
      A : constant Singles.Vector3 := Entities.Get (Entity, Materials.Albedo).Vector3_Value;
      Albedo : aliased constant HIP.Float3 :=
        (C_float (A (GL.X)), C_float (A (GL.Y)), C_float (A (GL.Z)));
   begin
      Check (HIP.Set_Material
        (Self.Handle, int (Index), Albedo'Access,
         C_float (Entities.Get (Entity, Materials.Metallic).Float_Value),
         C_float (Entities.Get (Entity, Materials.Roughness).Float_Value)));
      if Index >= Self.Last_Material_Index then
         Self.Last_Material_Index := Index + 1;
      end if;
   end Set_Material;

   --  madarch-renderers.adb:369-377
   function Add_Material
     (Self : in out Renderer; Entity : Entities.Entity) return Materials.Id
   is
      Index : constant Materials.Id := Self.Last_Material_Index;
   begin
      Set_Material (Self, Index, Entity);
      return Index;
   end Add_Material;

   --  madarch-renderers.adb:379-398
   procedure Set_Primitive
     (Self : in out Renderer; Prim : Primitives.Primitive; Index : Positive;
      Entity : Entities.Entity)
   is
      Blob : aliased constant Byte_Array :=
        Element_Blob (Scenes.HIP.Get_Primitive_Element_Type (Self.Scene, Prim), Entity);
   begin
      Check (HIP.Set_Primitive
        (Self.Handle, int (Scenes.HIP.Kind_Index (Self.Scene, Prim)), int (Index),
         Blob'Address, Blob'Length));
   end Set_Primitive;

   --  madarch-renderers.adb:435-456
   procedure Add_Primitive
     (Self : in out Renderer; Prim : Primitives.Primitive; Entity : Entities.Entity)
   is
      Blob  : aliased constant Byte_Array :=
        Element_Blob (Scenes.HIP.Get_Primitive_Element_Type (Self.Scene, Prim), Entity);
      Count : aliased int;
   begin
      Check (HIP.Add_Primitive
        (Self.Handle, int (Scenes.HIP.Kind_Index (Self.Scene, Prim)),
         Blob'Address, Blob'Length, Count'Access));
   end Add_Primitive;

   --  madarch-renderers.adb:458-483
   procedure Set_Light
     (Self : in out Renderer; Index : Positive; Lit : Lights.Light;
      Entity : Entities.Entity)
   is
      Blob : aliased constant Byte_Array :=
        Element_Blob (Scenes.HIP.Get_Light_Element_Type (Self.Scene, Lit), Entity);
   begin
      Check (HIP.Set_Light
        (Self.Handle, int (Index), int (Scenes.HIP.Kind_Index (Self.Scene, Lit)),
         Blob'Address, Blob'Length));
   end Set_Light;

   --  madarch-renderers.adb:485-497
   procedure Set_Camera_Position
     (Self : in out Renderer; Position : Singles.Vector3)
   is
      P : aliased constant HIP.Float3 :=
        (C_float (Position (GL.X)), C_float (Position (GL.Y)), C_float (Position (GL.Z)));
   begin
      Check (HIP.Set_Camera_Position (Self.Handle, P'Access));
   end Set_Camera_Position;

   procedure Set_Camera_Orientation
     (Self : in out Renderer; Orientation : Singles.Matrix3)
   is
      --  OpenGLAda matrices are indexed (column, row); the C side takes column-major
      M : aliased HIP.Float9;
      K : Natural := 0;
   begin
      for Col in GL.X .. GL.Z loop
         for Row in GL.X .. GL.Z loop
            M (K) := C_float (Orientation (Col, Row));
            K := K + 1;
         end loop;
      end loop;
      Check (HIP.Set_Camera_Orientation (Self.Handle, M'Access));
   end Set_Camera_Orientation;

   --  madarch-renderers.adb:499-526
   function Eval_Distance_To
     (Self : Renderer; Position : Singles.Vector3;
      Prims : Primitives.Primitive_Array; Normal : out Singles.Vector3) return Single
   is
      P    : aliased constant HIP.Float3 :=
        (C_float (Position (GL.X)), C_float (Position (GL.Y)), C_float (Position (GL.Z)));
      N    : aliased HIP.Float3;
      D    : aliased C_float;
      Ixs  : aliased array (Prims'Range) of aliased int;
   begin
      for I in Prims'Range loop
         Ixs (I) := int (Scenes.HIP.Kind_Index (Self.Scene, Prims (I)));
      end loop;
      Check (HIP.Eval_Distance_To
        (Self.Handle, 1, P'Address, Ixs'Address, Ixs'Length, N'Address, D'Address));
      Normal := (Single (N (0)), Single (N (1)), Single (N (2)));
      return Single (D);
   end Eval_Distance_To;

   --  madarch-renderers.adb:757-775 (all three methods run on the device)
   procedure Update_Partitioning
     (Self : in out Renderer; Method : Partitioning_Update_Method := GPU_Fast) is
   begin
      Check (HIP.Update_Partitioning
        (Self.Handle, Partitioning_Update_Method'Pos (Method)));
   end Update_Partitioning;
end Madarch.Renderers;
