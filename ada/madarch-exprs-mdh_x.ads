--  madarch-exprs-mdh_x.ads -- Madarch.Exprs.MDH_X: expression trees lowered to the MDH_X
--  register programs of include/madarch_hip.h.
--
--  What Exprs.To_GLSL (madarch/madarch-exprs.adb:325-711) is to the OpenGL back end, Lower is
--  to libmadarch_hip.so: the Distance / Normal / Material expressions of a user-defined
--  primitive kind (madarch-primitives.ads:24-30) and the Sample / Position expressions of a
--  user-defined light kind (madarch-lights.ads:20-24) travel to the renderer as straight-line
--  programs, one IEEE fp32 operation per instruction, in the operation order the generated GLSL
--  has (DESIGN.md section 5).  A child of Madarch.Exprs because the node types are private there.
--
--  The same lowering, node kind by node kind, as madarch_amd/exprs.py (class _Compiler), which
--  the tests of this repository run; this unit is SOURCE ONLY (no Ada toolchain in the image).

with Interfaces;

package Madarch.Exprs.MDH_X is
   subtype Word is Interfaces.Integer_32;
   type Word_Array is array (Natural range <>) of aliased Word
     with Convention => C;
   type Word_Array_Access is access all Word_Array;

   --  a named argument of the program: its kind and its first MDH_X_POINT float
   type Argument is record
      Name  : Unbounded_String;
      Kind  : Value_Kind;
      First : Natural;
   end record;
   type Argument_Array is array (Positive range <>) of Argument;

   No_Arguments : constant Argument_Array (1 .. 0) :=
     (others => (Null_Unbounded_String, Float_Kind, 0));

   --  the point of Distance / Normal: floats 0 .. 2, bound to Name
   function Point_Argument (Name : String) return Argument_Array;

   --  a light's Sample (madarch-scenes.adb:500-516): pos 0, normal 3, dir 6, dist 9
   function Light_Sample_Arguments
     (Pos, Normal, Dir, Dist : String) return Argument_Array;

   --  a node the programs cannot express: External_Call (which Exprs.Eval cannot
   --  evaluate either, madarch-exprs.adb:715-716), integer arithmetic, more than
   --  64 live registers, more than 4096 words
   Unsupported_Expr : exception;

   --  The program of E for a kind whose instances hold Comps (packed as MDH_X_COMP
   --  addresses them: components in order, a vector = 3 floats).  The result lands
   --  in R0 (R0 .. R2 for a vector).  Raises Type_Inference_Error as Infer_Type does.
   function Lower
     (E           : Expr'Class;
      Comps       : Components.Component_Array;
      Result_Kind : Value_Kind;
      Args        : Argument_Array := No_Arguments) return Word_Array;
end Madarch.Exprs.MDH_X;
