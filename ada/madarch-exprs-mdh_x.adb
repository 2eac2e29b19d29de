--  madarch-exprs-mdh_x.adb -- see the spec.  The walk mirrors madarch_amd/exprs.py
--  (_Compiler.expr / bin_op / builtin) case for case; the instruction encoding is
--  include/madarch_hip.h: op | dst << 8 | a << 16 | b << 24, LIT and SEL take a second word.
--  SOURCE ONLY: never compiled in this pipeline.

with Ada.Containers.Vectors;
with Ada.Unchecked_Conversion;

with GL.Types;

package body Madarch.Exprs.MDH_X is
   use Interfaces;
   use type GL.Types.Single;
   use type GL.Types.Int;

   --  enum MDH_X_* of include/madarch_hip.h
   X_LIT   : constant := 0;   X_MOV   : constant := 1;   X_COMP  : constant := 2;
   X_POINT : constant := 3;   X_ADD   : constant := 4;   X_SUB   : constant := 5;
   X_MUL   : constant := 6;   X_DIV   : constant := 7;   X_DIVF  : constant := 8;
   X_NEG   : constant := 9;   X_ABS   : constant := 10;  X_FLOOR : constant := 11;
   X_SIGN  : constant := 12;  X_MIN   : constant := 13;  X_MAX   : constant := 14;
   X_SQRT  : constant := 15;  X_POW   : constant := 16;  X_LT    : constant := 17;
   X_GT    : constant := 18;  X_LE    : constant := 19;  X_GE    : constant := 20;
   X_SEL   : constant := 21;  X_ITOF  : constant := 22;  X_ACOS  : constant := 23;
   X_SIN   : constant := 24;  X_COS   : constant := 25;  X_TAN   : constant := 26;
   X_ASIN  : constant := 27;  X_ATAN  : constant := 28;

   X_Regs      : constant := 64;
   X_Max_Words : constant := 4096;

   --  a comparison result is a kind of its own: usable as an If_Then_Else condition
   --  and by To_Float only
   type X_Kind is (K_Vector3, K_Float, K_Int, K_Bool);

   function To_X (K : Value_Kind) return X_Kind is
     (case K is
         when Vector3_Kind => K_Vector3,
         when Float_Kind   => K_Float,
         when Int_Kind     => K_Int);

   subtype Reg is Natural range 0 .. X_Regs - 1;
   type Reg_List is array (1 .. 3) of Reg;

   --  what an expression evaluates to: its kind, the registers that hold it and
   --  whether this expression owns them (an identifier aliases its variable's)
   type Operand is record
      Kind  : X_Kind;
      N     : Natural range 1 .. 3;
      R     : Reg_List;
      Owned : Boolean;
   end record;

   function Width (K : X_Kind) return Natural is
     (if K = K_Vector3 then 3 else 1);

   package Word_Vectors is new Ada.Containers.Vectors (Natural, Unsigned_32);

   type Binding is record
      Name : Unbounded_String;
      Kind : X_Kind;
      N    : Natural range 1 .. 3;
      R    : Reg_List;
   end record;
   package Binding_Vectors is new Ada.Containers.Vectors (Positive, Binding);

   function Bits is new Ada.Unchecked_Conversion (GL.Types.Single, Unsigned_32);
   function Bits is new Ada.Unchecked_Conversion (Integer_32, Unsigned_32);
   function To_Word is new Ada.Unchecked_Conversion (Unsigned_32, Integer_32);

   --  a float literal as it survives Single'Image in the generated GLSL text
   --  (madarch-exprs.adb:330-336): six significant digits
   function Image_Roundtrip (X : GL.Types.Single) return GL.Types.Single is
     (if X'Valid then GL.Types.Single'Value (GL.Types.Single'Image (X)) else X);

   function Point_Argument (Name : String) return Argument_Array is
     (1 => (To_Unbounded_String (Name), Vector3_Kind, 0));

   function Light_Sample_Arguments
     (Pos, Normal, Dir, Dist : String) return Argument_Array is
     ((To_Unbounded_String (Pos), Vector3_Kind, 0),
      (To_Unbounded_String (Normal), Vector3_Kind, 3),
      (To_Unbounded_String (Dir), Vector3_Kind, 6),
      (To_Unbounded_String (Dist), Float_Kind, 9));

   function Lower
     (E           : Expr'Class;
      Comps       : Components.Component_Array;
      Result_Kind : Value_Kind;
      Args        : Argument_Array := No_Arguments) return Word_Array
   is
      Code : Word_Vectors.Vector;

      --  free registers as a stack: R3 is handed out first, R0 .. R2 hold the result
      Free     : array (1 .. X_Regs) of Reg;
      Free_Top : Natural := 0;

      procedure Emit
        (Op : Natural; Dst : Reg; A : Natural := 0; B : Natural := 0)
      is
      begin
         Code.Append
           (Unsigned_32 (Op) or Shift_Left (Unsigned_32 (Dst), 8)
            or Shift_Left (Unsigned_32 (A), 16) or Shift_Left (Unsigned_32 (B), 24));
      end Emit;

      function Alloc return Reg is
      begin
         if Free_Top = 0 then
            raise Unsupported_Expr with "expression needs more than 64 registers";
         end if;
         Free_Top := Free_Top - 1;
         return Free (Free_Top + 1);
      end Alloc;

      function Alloc (K : X_Kind) return Operand is
         Res : Operand := (K, Width (K), (others => 0), True);
      begin
         for I in 1 .. Res.N loop
            Res.R (I) := Alloc;
         end loop;
         return Res;
      end Alloc;

      procedure Release (R : Reg) is
      begin
         Free_Top := Free_Top + 1;
         Free (Free_Top) := R;
      end Release;

      --  (in reverse, so that the next Alloc hands the same registers out in the same order)
      procedure Release (O : Operand) is
      begin
         if O.Owned then
            for I in reverse 1 .. O.N loop
               Release (O.R (I));
            end loop;
         end if;
      end Release;

      function Lit_Reg (B : Unsigned_32) return Reg is
         R : constant Reg := Alloc;
      begin
         Emit (X_LIT, R);
         Code.Append (B);
         return R;
      end Lit_Reg;

      --  packed instance layout of MDH_X_COMP
      function Comp_Offset (C : Component; K : out X_Kind) return Natural is
         Off : Natural := 0;
      begin
         for I in Comps'Range loop
            if Comps (I) = C then
               K := To_X (Get_Kind (C));
               return Off;
            end if;
            Off := Off + (if Get_Kind (Comps (I)) = Vector3_Kind then 3 else 1);
         end loop;
         raise Type_Inference_Error with
           "component " & Get_Name (C) & " is not one of the kind's components";
      end Comp_Offset;

      --  GLSL's implicit int -> float conversion of a scalar operand
      function As_Float (O : Operand) return Operand is
      begin
         if O.Kind /= K_Int then
            return O;
         end if;
         declare
            Res : constant Operand := Alloc (K_Float);
         begin
            Emit (X_ITOF, Res.R (1), O.R (1));
            Release (O);
            return Res;
         end;
      end As_Float;

      --  an owned copy of an aliased value
      function Own (O : Operand) return Operand is
      begin
         if O.Owned then
            return O;
         end if;
         declare
            Res : constant Operand := Alloc (O.Kind);
         begin
            for I in 1 .. O.N loop
               Emit (X_MOV, Res.R (I), O.R (I));
            end loop;
            return Res;
         end;
      end Own;

      --  (ax bx + ay by) + az bz into a fresh register
      function Dot_Reg (A, B : Operand) return Reg is
         T0 : constant Reg := Alloc;
         T1 : constant Reg := Alloc;
         D  : constant Reg := Alloc;
      begin
         Emit (X_MUL, T0, A.R (1), B.R (1));
         Emit (X_MUL, T1, A.R (2), B.R (2));
         Emit (X_ADD, T0, T0, T1);
         Emit (X_MUL, T1, A.R (3), B.R (3));
         Emit (X_ADD, D, T0, T1);
         Release (T1);
         Release (T0);
         return D;
      end Dot_Reg;

      function Scalar (K : X_Kind; R : Reg) return Operand is
        ((K, 1, (R, 0, 0), True));

      --  component I of a vector operand, or the scalar itself (GLSL broadcast)
      function Lane (O : Operand; I : Positive) return Reg is
        (if O.Kind = K_Vector3 then O.R (I) else O.R (1));

      function Walk
        (X : Expr'Class; Env : Binding_Vectors.Vector) return Operand;

      function Walk_Bin_Op
        (B : Bin_Op; Env : Binding_Vectors.Vector) return Operand
      is
         L : constant Operand := As_Float (Walk (B.Lhs, Env));
         R : constant Operand := As_Float (Walk (B.Rhs, Env));
      begin
         if B.Op in Bin_Lt .. Bin_Gte then
            if L.Kind /= K_Float or else R.Kind /= K_Float then
               raise Type_Inference_Error with "comparison of non-scalars";
            end if;
            declare
               D : constant Reg := Alloc;
            begin
               Emit ((case B.Op is
                         when Bin_Lt  => X_LT,
                         when Bin_Gt  => X_GT,
                         when Bin_Lte => X_LE,
                         when others  => X_GE), D, L.R (1), R.R (1));
               Release (R);
               Release (L);
               return Scalar (K_Bool, D);
            end;
         end if;
         if L.Kind not in K_Vector3 | K_Float
           or else R.Kind not in K_Vector3 | K_Float
         then
            raise Type_Inference_Error with "binary operation on a condition";
         end if;
         declare
            Vec : constant Boolean := L.Kind = K_Vector3 or else R.Kind = K_Vector3;
            --  a scalar "/" is DIVF: the GLSL "/" in the kernels, Madarch.Values."/"
            --  (L + R, madarch-values.adb:112) in Eval_Distance_To under MDH_OPT_ADA_EVAL_DIV
            Op  : constant Natural :=
              (case B.Op is
                  when Bin_Add => X_ADD,
                  when Bin_Sub => X_SUB,
                  when Bin_Mul => X_MUL,
                  when others  => (if Vec then X_DIV else X_DIVF));
            Res : constant Operand := Alloc (if Vec then K_Vector3 else K_Float);
         begin
            for I in 1 .. Res.N loop
               Emit (Op, Res.R (I), Lane (L, I), Lane (R, I));
            end loop;
            Release (R);
            Release (L);
            return Res;
         end;
      end Walk_Bin_Op;

      function Walk_Builtin
        (B : Builtin_Call; Env : Binding_Vectors.Vector) return Operand
      is
         A : array (B.Args'Range) of Operand;

         procedure Want (I : Positive; K : X_Kind) is
         begin
            if K = K_Float and then A (I).Kind = K_Int then
               A (I) := As_Float (A (I));
            elsif A (I).Kind /= K then
               raise Type_Inference_Error with
                 "builtin " & B.Builtin'Image & ": argument" & I'Image & " has the wrong kind";
            end if;
         end Want;

         function Done (Res : Operand) return Operand is
         begin
            for I in reverse A'Range loop
               Release (A (I));
            end loop;
            return Res;
         end Done;

         --  a literal integer exponent between 2 and 16, or 0
         function Small_Integer_Power return Natural is
         begin
            if B.Args (2).Value.all in Lit'Class then
               declare
                  V : Value renames Lit (B.Args (2).Value.all).V;
               begin
                  if V.Kind = Int_Kind and then V.Int_Value in 2 .. 16 then
                     return Natural (V.Int_Value);
                  elsif V.Kind = Float_Kind
                    and then V.Float_Value = GL.Types.Single'Floor (V.Float_Value)
                    and then V.Float_Value in 2.0 .. 16.0
                  then
                     return Natural (V.Float_Value);
                  end if;
               end;
            end if;
            return 0;
         end Small_Integer_Power;
      begin
         for I in A'Range loop
            A (I) := Walk (B.Args (I), Env);
         end loop;
         case B.Builtin is
            when Builtin_Vec3 =>
               Want (1, K_Float); Want (2, K_Float); Want (3, K_Float);
               declare
                  Res : constant Operand := Alloc (K_Vector3);
               begin
                  for I in 1 .. 3 loop
                     Emit (X_MOV, Res.R (I), A (I).R (1));
                  end loop;
                  return Done (Res);
               end;
            when Builtin_Dot =>
               Want (1, K_Vector3); Want (2, K_Vector3);
               return Done (Scalar (K_Float, Dot_Reg (A (1), A (2))));
            when Builtin_Dot2 =>
               Want (1, K_Vector3);
               return Done (Scalar (K_Float, Dot_Reg (A (1), A (1))));
            when Builtin_Len =>
               Want (1, K_Vector3);
               declare
                  D : constant Reg := Dot_Reg (A (1), A (1));
               begin
                  Emit (X_SQRT, D, D);
                  return Done (Scalar (K_Float, D));
               end;
            when Builtin_Norm =>   --  v / length (v), support/math_utils.ads:77-83
               Want (1, K_Vector3);
               declare
                  D   : constant Reg := Dot_Reg (A (1), A (1));
                  Res : Operand;
               begin
                  Emit (X_SQRT, D, D);
                  Res := Alloc (K_Vector3);
                  for I in 1 .. 3 loop
                     Emit (X_DIV, Res.R (I), A (1).R (I), D);
                  end loop;
                  Release (D);
                  return Done (Res);
               end;
            when Builtin_Cross =>   --  a_p c_q - a_q c_p for (p, q) = (y, z), (z, x), (x, y)
               Want (1, K_Vector3); Want (2, K_Vector3);
               declare
                  Res : constant Operand := Alloc (K_Vector3);
                  T   : constant Reg := Alloc;
                  P   : constant array (1 .. 3) of Positive := (2, 3, 1);
                  Q   : constant array (1 .. 3) of Positive := (3, 1, 2);
               begin
                  for I in 1 .. 3 loop
                     Emit (X_MUL, Res.R (I), A (1).R (P (I)), A (2).R (Q (I)));
                     Emit (X_MUL, T, A (1).R (Q (I)), A (2).R (P (I)));
                     Emit (X_SUB, Res.R (I), Res.R (I), T);
                  end loop;
                  Release (T);
                  return Done (Res);
               end;
            when Builtin_Neg | Builtin_Abs | Builtin_Floor =>
               if A (1).Kind not in K_Vector3 | K_Float then
                  raise Type_Inference_Error with "builtin of a non-float";
               end if;
               declare
                  Res : constant Operand := Alloc (A (1).Kind);
                  Op  : constant Natural :=
                    (case B.Builtin is
                        when Builtin_Neg => X_NEG,
                        when Builtin_Abs => X_ABS,
                        when others      => X_FLOOR);
               begin
                  for I in 1 .. Res.N loop
                     Emit (Op, Res.R (I), A (1).R (I));
                  end loop;
                  return Done (Res);
               end;
            when Builtin_Sign | Builtin_Sqrt | Builtin_Acos | Builtin_Sin
               | Builtin_Cos | Builtin_Tan | Builtin_Asin | Builtin_Atan =>
               Want (1, K_Float);
               declare
                  D : constant Reg := Alloc;
               begin
                  Emit ((case B.Builtin is
                            when Builtin_Sign => X_SIGN,
                            when Builtin_Sqrt => X_SQRT,
                            when Builtin_Acos => X_ACOS,
                            when Builtin_Sin  => X_SIN,
                            when Builtin_Cos  => X_COS,
                            when Builtin_Tan  => X_TAN,
                            when Builtin_Asin => X_ASIN,
                            when others       => X_ATAN), D, A (1).R (1));
                  return Done (Scalar (K_Float, D));
               end;
            when Builtin_Min | Builtin_Max =>   --  component-wise, a scalar operand is broadcast
               if A (1).Kind = K_Int then Want (1, K_Float); end if;
               if A (2).Kind = K_Int then Want (2, K_Float); end if;
               if A (1).Kind not in K_Vector3 | K_Float
                 or else A (2).Kind not in K_Vector3 | K_Float
               then
                  raise Type_Inference_Error with "min / max of a non-float";
               end if;
               declare
                  Vec : constant Boolean :=
                    A (1).Kind = K_Vector3 or else A (2).Kind = K_Vector3;
                  Res : constant Operand := Alloc (if Vec then K_Vector3 else K_Float);
               begin
                  for I in 1 .. Res.N loop
                     Emit ((if B.Builtin = Builtin_Min then X_MIN else X_MAX),
                           Res.R (I), Lane (A (1), I), Lane (A (2), I));
                  end loop;
                  return Done (Res);
               end;
            when Builtin_Pow =>
               declare
                  N : constant Natural := Small_Integer_Power;
                  D : Reg;
               begin
                  Want (1, K_Float);
                  if N /= 0 then
                     --  repeated multiplication, squaring from the top bit (DESIGN.md section 5)
                     D := Alloc;
                     Emit (X_MOV, D, A (1).R (1));
                     declare
                        Top : Natural := 0;
                     begin
                        while 2 ** (Top + 1) <= N loop
                           Top := Top + 1;
                        end loop;
                        for Bit in reverse 0 .. Top - 1 loop
                           Emit (X_MUL, D, D, D);
                           if (N / 2 ** Bit) mod 2 = 1 then
                              Emit (X_MUL, D, D, A (1).R (1));
                           end if;
                        end loop;
                     end;
                  else
                     Want (2, K_Float);
                     D := Alloc;
                     Emit (X_POW, D, A (1).R (1), A (2).R (1));
                  end if;
                  return Done (Scalar (K_Float, D));
               end;
            when Builtin_Clamp =>   --  min (max (x, lb), ub)
               Want (1, K_Float); Want (2, K_Float); Want (3, K_Float);
               declare
                  D : constant Reg := Alloc;
               begin
                  Emit (X_MAX, D, A (1).R (1), A (2).R (1));
                  Emit (X_MIN, D, D, A (3).R (1));
                  return Done (Scalar (K_Float, D));
               end;
            when Builtin_Float =>   --  float (int); float (bool) is 1.0 / 0.0 already
               if A (1).Kind = K_Vector3 then
                  raise Type_Inference_Error with "To_Float of a vector";
               end if;
               declare
                  D : constant Reg := Alloc;
               begin
                  Emit ((if A (1).Kind = K_Int then X_ITOF else X_MOV), D, A (1).R (1));
                  return Done (Scalar (K_Float, D));
               end;
         end case;
      end Walk_Builtin;

      function Walk
        (X : Expr'Class; Env : Binding_Vectors.Vector) return Operand
      is
         N : Expr_Node'Class renames X.Value.all;
      begin
         if N in Lit'Class then
            declare
               V : Value renames Lit (N).V;
            begin
               case V.Kind is
                  when Vector3_Kind =>
                     declare
                        Res : Operand := (K_Vector3, 3, (others => 0), True);
                     begin
                        Res.R (1) := Lit_Reg (Bits (Image_Roundtrip (V.Vector3_Value (GL.X))));
                        Res.R (2) := Lit_Reg (Bits (Image_Roundtrip (V.Vector3_Value (GL.Y))));
                        Res.R (3) := Lit_Reg (Bits (Image_Roundtrip (V.Vector3_Value (GL.Z))));
                        return Res;
                     end;
                  when Float_Kind =>
                     return Scalar (K_Float, Lit_Reg (Bits (Image_Roundtrip (V.Float_Value))));
                  when Int_Kind =>
                     return Scalar (K_Int, Lit_Reg (Bits (Integer_32 (V.Int_Value))));
               end case;
            end;
         elsif N in Ident'Class then
            for I in reverse 1 .. Natural (Env.Length) loop   --  the innermost binding wins
               if Env (I).Name = Ident (N).Name then
                  return (Env (I).Kind, Env (I).N, Env (I).R, False);
               end if;
            end loop;
            raise Type_Inference_Error with
              "unbound identifier " & To_String (Ident (N).Name);
         elsif N in Get_Component'Class then
            declare
               K   : X_Kind;
               Off : constant Natural := Comp_Offset (Get_Component (N).Suffix, K);
               Res : constant Operand := Alloc (K);
            begin
               for I in 1 .. Res.N loop
                  Emit (X_COMP, Res.R (I), Off + I - 1);
               end loop;
               return Res;
            end;
         elsif N in Project_Axis'Class then
            declare
               V : constant Operand := Walk (Project_Axis (N).E, Env);
               R : Reg;
            begin
               if V.Kind /= K_Vector3 then
                  raise Type_Inference_Error with "axis projection of a non-vector";
               end if;
               R := Alloc;
               Emit (X_MOV, R, V.R (GL.Index_3D'Pos (Project_Axis (N).A) + 1));
               Release (V);
               return Scalar (K_Float, R);
            end;
         elsif N in Var_Body'Class then
            declare
               VB    : Var_Body renames Var_Body (N);
               Env2  : Binding_Vectors.Vector := Env;
               Bound : array (VB.Decls'Range) of Operand;
               Res   : Operand;
            begin
               for I in VB.Decls'Range loop
                  declare
                     O : Operand := Walk (VB.Decls (I).Value, Env2);
                  begin
                     if O.Kind = K_Int and then VB.Decls (I).Kind = Float_Kind then
                        O := As_Float (O);
                     end if;
                     if O.Kind /= To_X (VB.Decls (I).Kind) then
                        raise Type_Inference_Error with
                          "declaration " & To_String (VB.Decls (I).Name) & ": kind mismatch";
                     end if;
                     O := Own (O);   --  an alias of another variable: own a copy
                     Env2.Append ((VB.Decls (I).Name, O.Kind, O.N, O.R));
                     Bound (I) := O;
                  end;
               end loop;
               --  the body may be one of the variables: move it out before they die
               Res := Own (Walk (VB.In_Body, Env2));
               for I in Bound'Range loop
                  Release (Bound (I));
               end loop;
               return Res;
            end;
         elsif N in Condition'Class then
            declare
               C : constant Operand := Walk (Condition (N).Cond, Env);
            begin
               if C.Kind not in K_Bool | K_Int then   --  madarch-exprs.adb:663-665
                  raise Type_Inference_Error with "Invalid value for ternary condition";
               end if;
               declare
                  T : constant Operand := Walk (Condition (N).Thn, Env);
                  F : constant Operand := Walk (Condition (N).Els, Env);
               begin
                  if T.Kind /= F.Kind then
                     raise Type_Inference_Error with "if expression: kind mismatch";
                  end if;
                  declare
                     Res : constant Operand := Alloc (T.Kind);
                  begin
                     for I in 1 .. Res.N loop   --  SEL d, cond, then ; else
                        Emit (X_SEL, Res.R (I), C.R (1), T.R (I));
                        Code.Append (Unsigned_32 (F.R (I)));
                     end loop;
                     Release (F);
                     Release (T);
                     Release (C);
                     return Res;
                  end;
               end;
            end;
         elsif N in Bin_Op'Class then
            return Walk_Bin_Op (Bin_Op (N), Env);
         elsif N in Builtin_Call'Class then
            return Walk_Builtin (Builtin_Call (N), Env);
         else   --  Unchecked_Call: no program form (and no Eval either, madarch-exprs.adb:715-716)
            raise Unsupported_Expr with "External_Call cannot be compiled to MDH_X";
         end if;
      end Walk;

      Env : Binding_Vectors.Vector;
   begin
      for R in reverse 3 .. X_Regs - 1 loop   --  (R3 on top of the stack)
         Release (R);
      end loop;
      for A of Args loop
         declare
            O : constant Operand := Alloc (To_X (A.Kind));
         begin
            for I in 1 .. O.N loop
               Emit (X_POINT, O.R (I), A.First + I - 1);
            end loop;
            Env.Append ((A.Name, O.Kind, O.N, O.R));
         end;
      end loop;
      declare
         Res : Operand := Walk (E, Env);
      begin
         if Result_Kind = Float_Kind and then Res.Kind = K_Int then
            Res := As_Float (Res);
         end if;
         if Res.Kind /= To_X (Result_Kind) then
            raise Type_Inference_Error with "expression has the wrong kind for this program";
         end if;
         for I in 1 .. Res.N loop
            Emit (X_MOV, I - 1, Res.R (I));
         end loop;
      end;
      if Natural (Code.Length) > X_Max_Words then
         raise Unsupported_Expr with "program longer than 4096 words";
      end if;
      return Words : Word_Array (0 .. Natural (Code.Length) - 1) do
         for I in Words'Range loop
            Words (I) := To_Word (Code (I));
         end loop;
      end return;
   end Lower;
end Madarch.Exprs.MDH_X;
