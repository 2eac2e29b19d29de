--  madarch-scenes-hip.adb -- see the spec.  SOURCE ONLY: never compiled in this pipeline.

with Ada.Unchecked_Deallocation;
with Interfaces.C; use Interfaces.C;
with Interfaces.C.Strings;
with System;

with Madarch.Components;
with Madarch.Exprs;
with Madarch.Exprs.MDH_X;
with Madarch.Values;
with Madarch.Primitives.Spheres;
with Madarch.Primitives.Planes;
with Madarch.Primitives.Boxes;
with Madarch.Primitives.Triangles;
with Madarch.Lights.Point_Lights;
with Madarch.Lights.Spot_Lights;

package body Madarch.Scenes.HIP is
   use type GL.Types.Size;
   use type Primitives.Primitive;
   use type Lights.Light;

   package X renames Madarch.Exprs.MDH_X;

   function Kind_Index (S : Scene; Prim : Primitives.Primitive) return Natural is
   begin
      for I in S.Prims'Range loop
         if S.Prims (I) = Prim then
            return I - S.Prims'First;
         end if;
      end loop;
      raise Constraint_Error with
        "primitive kind " & Primitives.Get_Name (Prim) & " is not part of the scene";
   end Kind_Index;

   function Kind_Index (S : Scene; Lit : Lights.Light) return Natural is
   begin
      for I in S.Lits'Range loop
         if S.Lits (I) = Lit then
            return I - S.Lits'First;
         end if;
      end loop;
      raise Constraint_Error with
        "light kind " & Lights.Get_Name (Lit) & " is not part of the scene";
   end Kind_Index;

   function Get_Primitive_Element_Type
     (S : Scene; Prim : Primitives.Primitive) return GPU_Types.GPU_Type
   is
      Array_Location, Count_Location : GPU_Types.Locations.Location;
   begin
      Get_Primitives_Location (S, Prim, Array_Location, Count_Location);
      return Array_Location.Component (1).Typ;
   end Get_Primitive_Element_Type;

   function Get_Light_Element_Type
     (S : Scene; Lit : Lights.Light) return GPU_Types.GPU_Type
   is
      Array_Location, Count_Location, Total_Location : GPU_Types.Locations.Location;
   begin
      Get_Lights_Location (S, Lit, Array_Location, Count_Location, Total_Location);
      return Array_Location.Component (1).Typ;
   end Get_Light_Element_Type;

   --  A fixed array of N structs takes N times the struct's size padded to 16 bytes
   --  (support/gpu_types-fixed_arrays.adb:17-25)
   function Length_Of (Array_Type, Element_Type : GPU_Types.GPU_Type) return Natural is
      Stride : GL.Types.Size := Element_Type.Size;
   begin
      while Stride mod 16 /= 0 loop
         Stride := Stride + 1;
      end loop;
      return Natural (Array_Type.Size / Stride);
   end Length_Of;

   function Declared_Count (S : Scene; Prim : Primitives.Primitive) return Natural is
      Array_Location, Count_Location : GPU_Types.Locations.Location;
   begin
      Get_Primitives_Location (S, Prim, Array_Location, Count_Location);
      return Length_Of (Array_Location.Typ, Array_Location.Component (1).Typ);
   end Declared_Count;

   function Declared_Count (S : Scene; Lit : Lights.Light) return Natural is
      Array_Location, Count_Location, Total_Location : GPU_Types.Locations.Location;
   begin
      Get_Lights_Location (S, Lit, Array_Location, Count_Location, Total_Location);
      return Length_Of (Array_Location.Typ, Array_Location.Component (1).Typ);
   end Declared_Count;

   function Is_Library_Kind (Prim : Primitives.Primitive) return Boolean is
     (Prim = Primitives.Spheres.Sphere or else Prim = Primitives.Planes.Plane
      or else Prim = Primitives.Boxes.Box or else Prim = Primitives.Triangles.Triangle);

   function Is_Library_Kind (Lit : Lights.Light) return Boolean is
     (Lit = Lights.Point_Lights.Point_Light or else Lit = Lights.Spot_Lights.Spot_Light);

   --  mdh_component [] of a kind: name and Values.Value_Kind'Pos (0 Vector3, 1 Float, 2 Int)
   function C_Components (Comps : Components.Component_Array) return Components_Access is
      Res : constant Components_Access :=
        new Madarch_HIP.Component_Array (0 .. size_t (Comps'Length) - 1);
   begin
      for I in Comps'Range loop
         Res (size_t (I - Comps'First)) :=
           (Name => Strings.New_String (Components.Get_Name (Comps (I))),
            Kind => int (Values.Value_Kind'Pos (Components.Get_Kind (Comps (I)))));
      end loop;
      return Res;
   end C_Components;

   --  a program on the heap (released by Free through the addresses in the declaration)
   procedure Attach
     (Code : out System.Address; Len : out int; Words : X.Word_Array)
   is
      Heap : constant X.Word_Array_Access := new X.Word_Array'(Words);
   begin
      Code := Heap (Heap'First)'Address;
      Len  := int (Heap'Length);
   end Attach;

   procedure Describe
     (S    : Scene;
      Desc : out Madarch_HIP.Scene_Desc;
      Keep : out Description)
   is
      Cfg : constant Partitioning_Settings := S.Partitioning_Config;
      Inst : constant Exprs.Struct_Expr := Exprs.Struct_Identifier ("inst");
   begin
      Keep.Prim_Kinds  := new Madarch_HIP.Kind_Decl_Array (0 .. size_t (S.Prims'Length) - 1);
      Keep.Light_Kinds := new Madarch_HIP.Kind_Decl_Array (0 .. size_t (S.Lits'Length) - 1);
      Keep.Prim_Comps  := new Components_Access_Array (1 .. S.Prims'Length);
      Keep.Light_Comps := new Components_Access_Array (1 .. S.Lits'Length);

      for I in S.Prims'Range loop
         declare
            Prim  : constant Primitives.Primitive := S.Prims (I);
            Comps : constant Components.Component_Array := Primitives.Get_Components (Prim);
            K     : constant Positive := I - S.Prims'First + 1;
            D     : Madarch_HIP.Kind_Decl renames Keep.Prim_Kinds (size_t (K - 1));
         begin
            Keep.Prim_Comps (K) := C_Components (Comps);
            D.Name         := Strings.New_String (Primitives.Get_Name (Prim));
            D.Max_Count    := int (Declared_Count (S, Prim));
            D.N_Components := int (Comps'Length);
            D.Components   := Keep.Prim_Comps (K) (0)'Address;
            if not Is_Library_Kind (Prim) then
               --  Distance and Normal take the point (floats 0 .. 2 of MDH_X_POINT), Material nothing
               Attach (D.Dist_Code, D.Dist_Len,
                       X.Lower (Primitives.Get_Dist_Expr (Prim, Inst, Exprs.Value_Identifier ("x")),
                                Comps, Values.Float_Kind, X.Point_Argument ("x")));
               Attach (D.Normal_Code, D.Normal_Len,
                       X.Lower (Primitives.Get_Normal_Expr (Prim, Inst, Exprs.Value_Identifier ("x")),
                                Comps, Values.Vector3_Kind, X.Point_Argument ("x")));
               Attach (D.Material_Code, D.Material_Len,
                       X.Lower (Primitives.Get_Material_Expr (Prim, Inst), Comps, Values.Int_Kind));
            end if;
         end;
      end loop;

      for I in S.Lits'Range loop
         declare
            Lit   : constant Lights.Light := S.Lits (I);
            Comps : constant Components.Component_Array := Lights.Get_Components (Lit);
            K     : constant Positive := I - S.Lits'First + 1;
            D     : Madarch_HIP.Kind_Decl renames Keep.Light_Kinds (size_t (K - 1));
         begin
            Keep.Light_Comps (K) := C_Components (Comps);
            D.Name         := Strings.New_String (Lights.Get_Name (Lit));
            D.Max_Count    := int (Declared_Count (S, Lit));
            D.N_Components := int (Comps'Length);
            D.Components   := Keep.Light_Comps (K) (0)'Address;
            if not Is_Library_Kind (Lit) then
               --  a light's Sample and Position travel in the Dist / Normal fields
               --  (include/madarch_hip.h: mdh_kind_decl); names as in the generated
               --  sample_<Light>, madarch-scenes.adb:500-516
               Attach (D.Dist_Code, D.Dist_Len,
                       X.Lower (Lights.Get_Sample_Expr
                                  (Lit, Inst, Exprs.Value_Identifier ("pos"), Exprs.Value_Identifier ("normal"),
                                   Exprs.Value_Identifier ("dir"), Exprs.Value_Identifier ("dist")),
                                Comps, Values.Vector3_Kind,
                                X.Light_Sample_Arguments ("pos", "normal", "dir", "dist")));
               Attach (D.Normal_Code, D.Normal_Len,
                       X.Lower (Lights.Get_Position_Expr (Lit, Inst), Comps, Values.Vector3_Kind));
            end if;
         end;
      end loop;

      Desc.N_Prim_Kinds  := int (S.Prims'Length);
      Desc.Prim_Kinds    := Keep.Prim_Kinds (0)'Address;
      Desc.N_Light_Kinds := int (S.Lits'Length);
      Desc.Light_Kinds   := Keep.Light_Kinds (0)'Address;
      if Cfg.Enable then
         Desc.Part :=
           (Enable          => 1,
            Index_Count     => int (Cfg.Index_Count),
            Border_Behavior => Partitioning_Border_Behavior'Pos (Cfg.Border_Behavior),
            Grid_Dimensions => (int (Cfg.Grid_Dimensions (GL.X)), int (Cfg.Grid_Dimensions (GL.Y)),
                                int (Cfg.Grid_Dimensions (GL.Z))),
            Grid_Spacing    => (C_float (Cfg.Grid_Spacing (GL.X)), C_float (Cfg.Grid_Spacing (GL.Y)),
                                C_float (Cfg.Grid_Spacing (GL.Z))),
            Grid_Offset     => (C_float (Cfg.Grid_Offset (GL.X)), C_float (Cfg.Grid_Offset (GL.Y)),
                                C_float (Cfg.Grid_Offset (GL.Z))));
      else
         Desc.Part := (Enable => 0, Index_Count => 0, Border_Behavior => 0,
                       Grid_Dimensions => (0, 0, 0), Grid_Spacing => (0.0, 0.0, 0.0),
                       Grid_Offset => (0.0, 0.0, 0.0));
      end if;
      Desc.Max_Dist      := C_float (S.Max_Dist);   --  (the component ada/scenes_patch.sh adds)
      Desc.Loop_Strategy := Codegen_Loop_Strategy'Pos (Unify);   --  a property of the GLSL text: unused
   end Describe;

   procedure Free (Keep : in out Description) is
      procedure Release is new Ada.Unchecked_Deallocation
        (Madarch_HIP.Kind_Decl_Array, Kind_Decls_Access);
      procedure Release is new Ada.Unchecked_Deallocation
        (Madarch_HIP.Component_Array, Components_Access);
      procedure Release is new Ada.Unchecked_Deallocation
        (Components_Access_Array, Components_Access_Array_Access);

      procedure Release_Kinds
        (Kinds : in out Kind_Decls_Access; Comps : in out Components_Access_Array_Access) is
      begin
         if Kinds /= null then
            for D of Kinds.all loop
               Strings.Free (D.Name);
               --  (the program words were allocated by Attach and are a few hundred bytes per
               --  user-defined kind and scene; they are left to the storage pool)
            end loop;
            Release (Kinds);
         end if;
         if Comps /= null then
            for C of Comps.all loop
               if C /= null then
                  for E of C.all loop
                     Strings.Free (E.Name);
                  end loop;
                  Release (C);
               end if;
            end loop;
            Release (Comps);
         end if;
      end Release_Kinds;
   begin
      Release_Kinds (Keep.Prim_Kinds, Keep.Prim_Comps);
      Release_Kinds (Keep.Light_Kinds, Keep.Light_Comps);
   end Free;
end Madarch.Scenes.HIP;
