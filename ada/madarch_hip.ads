--  madarch_hip.ads -- thin Ada binding of include/madarch_hip.h (libmadarch_hip.so).
--
--  SOURCE ONLY: this pipeline has no Ada toolchain (no gnat1 / gprbuild), so this
--  unit has never been compiled here.  It is the binding a Madarch maintainer adds
--  next to madarch/madarch-renderers.ads; the replacement body that uses it is
--  ada/madarch-renderers.adb.  Link with -lmadarch_hip.

with Interfaces.C;
with Interfaces.C.Strings;
with System;

package Madarch_HIP is
   use Interfaces.C;

   subtype Status is int;   --  0 = ok; see the MDH_E_* codes of madarch_hip.h
   subtype Handle is System.Address;   --  mdh_renderer *

   OK             : constant Status := 0;
   E_Probe_Mismatch : constant Status := 2;   --  Program_Error, renderers.adb:63-65
   E_Index        : constant Status := 4;     --  Constraint_Error

   type Component is record
      Name : Strings.chars_ptr;
      Kind : int;           --  0 Vector3, 1 Float, 2 Int = Values.Value_Kind'Pos
   end record with Convention => C;
   type Component_Array is array (size_t range <>) of aliased Component
     with Convention => C;

   type Kind_Decl is record
      Name         : Strings.chars_ptr;
      Max_Count    : int;
      N_Components : int;
      Components   : System.Address;   --  const mdh_component *
      --  user-defined kinds (Primitives.Create): the Distance, Normal and Material
      --  expressions as MDH_X register programs; Null_Address / 0 for built-in kinds
      Dist_Code     : System.Address := System.Null_Address;   --  const int32_t *
      Dist_Len      : int := 0;
      Normal_Code   : System.Address := System.Null_Address;
      Normal_Len    : int := 0;
      Material_Code : System.Address := System.Null_Address;
      Material_Len  : int := 0;
   end record with Convention => C;
   type Kind_Decl_Array is array (size_t range <>) of aliased Kind_Decl
     with Convention => C;

   type Int3 is array (0 .. 2) of int with Convention => C;
   type Int2 is array (0 .. 1) of int with Convention => C;
   type Float3 is array (0 .. 2) of C_float with Convention => C;
   type Float9 is array (0 .. 8) of C_float with Convention => C;

   type Partitioning is record
      Enable, Index_Count, Border_Behavior : int;
      Grid_Dimensions : Int3;
      Grid_Spacing, Grid_Offset : Float3;
   end record with Convention => C;

   type Probe_Settings is record
      Radiance_Resolution, Irradiance_Resolution : int;
      Probe_Count     : Int2;
      Grid_Dimensions : Int3;
      Grid_Spacing    : Float3;
   end record with Convention => C;

   type Volumetrics is record
      Enabled               : int;
      Visibility_Resolution : Int3;
      Visibility_Step_Size  : C_float;
      Scattering_Resolution : Int2;
      Scattering_Step_Size  : C_float;
   end record with Convention => C;

   type Scene_Desc is record
      N_Prim_Kinds  : int;
      Prim_Kinds    : System.Address;
      N_Light_Kinds : int;
      Light_Kinds   : System.Address;
      Part          : Partitioning;
      Max_Dist      : C_float;
      Loop_Strategy : int;
   end record with Convention => C;

   function Create
     (Width, Height : int;
      Scene  : access constant Scene_Desc;
      Probes : access constant Probe_Settings;
      Vol    : access constant Volumetrics;
      Device : int;
      Result : access Handle) return Status
     with Import, Convention => C, External_Name => "mdh_create";

   function Destroy (R : Handle) return Status
     with Import, Convention => C, External_Name => "mdh_destroy";

   function Set_Material
     (R : Handle; Id0 : int; Albedo : access constant Float3;
      Metallic, Roughness : C_float) return Status
     with Import, Convention => C, External_Name => "mdh_set_material";

   function Set_Primitive
     (R : Handle; Kind_Ix, Index1 : int; Blob : System.Address; N_Bytes : int)
      return Status
     with Import, Convention => C, External_Name => "mdh_set_primitive";

   function Add_Primitive
     (R : Handle; Kind_Ix : int; Blob : System.Address; N_Bytes : int;
      Out_Count : access int) return Status
     with Import, Convention => C, External_Name => "mdh_add_primitive";

   function Set_Light
     (R : Handle; Index1, Light_Kind_Ix : int; Blob : System.Address;
      N_Bytes : int) return Status
     with Import, Convention => C, External_Name => "mdh_set_light";

   function Set_Camera_Position (R : Handle; P : access constant Float3)
      return Status
     with Import, Convention => C, External_Name => "mdh_set_camera_position";

   function Set_Camera_Orientation (R : Handle; M : access constant Float9)
      return Status
     with Import, Convention => C, External_Name => "mdh_set_camera_orientation";

   function Update_Partitioning (R : Handle; Method : int) return Status
     with Import, Convention => C, External_Name => "mdh_update_partitioning";

   function Render (R : Handle) return Status
     with Import, Convention => C, External_Name => "mdh_render";

   function Finish (R : Handle) return Status
     with Import, Convention => C, External_Name => "mdh_finish";

   function Read_Framebuffer (R : Handle; RGB_Out : System.Address) return Status
     with Import, Convention => C, External_Name => "mdh_read_framebuffer";

   --  the window's RGBA8 pixels (Swap_Buffers, madarch-renderers.adb:320): enqueue, then fetch
   function Swap_Buffers (R : Handle) return Status
     with Import, Convention => C, External_Name => "mdh_swap_buffers";

   function Front_Buffer
     (R : Handle; RGBA : access System.Address; Swap_Count : access Interfaces.C.long) return Status
     with Import, Convention => C, External_Name => "mdh_front_buffer";

   function Eval_Distance_To
     (R : Handle; N : int; Points : System.Address; Kind_Ixs : System.Address;
      N_Kinds : int; Normals_Out, Dist_Out : System.Address) return Status
     with Import, Convention => C, External_Name => "mdh_eval_distance_to";

   --  One frame on the N GPUs of a node, one process per GPU (madarch_hip.h, mdh_comm_*):
   --  every process creates its Renderer on its own device, rank 0 makes a 128-byte id and
   --  hands it to the others (a file, an environment variable), all call Comm_Init, and
   --  from then on Render of every rank is one frame of the sharded schedule -- probe
   --  slices, RCCL all-gather inside the library, interleaved screen tiles.
   Comm_Id_Bytes : constant := 128;
   type Comm_Id is array (0 .. Comm_Id_Bytes - 1) of Interfaces.C.unsigned_char
     with Convention => C;

   function Comm_Unique_Id (Id_Out : access Comm_Id) return Status
     with Import, Convention => C, External_Name => "mdh_comm_unique_id";

   function Comm_Init
     (R : Handle; Id : access constant Comm_Id; Rank, World : int) return Status
     with Import, Convention => C, External_Name => "mdh_comm_init";

   function Comm_Destroy (R : Handle) return Status
     with Import, Convention => C, External_Name => "mdh_comm_destroy";

   function Comm_Barrier (R : Handle) return Status
     with Import, Convention => C, External_Name => "mdh_comm_barrier";

   --  the ranks' tiles of the last frame summed into Root's framebuffer (what its window shows)
   function Comm_Reduce_Framebuffer (R : Handle; Root : int) return Status
     with Import, Convention => C, External_Name => "mdh_comm_reduce_framebuffer";

   function Comm_Available return Status
     with Import, Convention => C, External_Name => "mdh_comm_available";

   function Comm_Abort (R : Handle) return Status
     with Import, Convention => C, External_Name => "mdh_comm_abort";

   --  The same sharded frame without a collective library (madarch_hip.h, mdh_peer_*): every
   --  rank exports 512 bytes of interprocess handles, the host hands all ranks' blobs to every
   --  rank, Peer_Init opens them, and Render's exchange is device-to-device copies ordered on
   --  the device.  One process per rank, one node; the form several ranks can run on ONE GPU.
   Peer_Blob_Bytes : constant := 512;
   type Peer_Blob is array (0 .. Peer_Blob_Bytes - 1) of Interfaces.C.unsigned_char
     with Convention => C;
   type Peer_Blobs is array (Natural range <>) of Peer_Blob
     with Convention => C;

   function Peer_Export (R : Handle; Blob_Out : access Peer_Blob) return Status
     with Import, Convention => C, External_Name => "mdh_peer_export";

   --  Blobs: the address of World blobs in rank order
   function Peer_Init
     (R : Handle; Blobs : System.Address; Rank, World : int) return Status
     with Import, Convention => C, External_Name => "mdh_peer_init";

   function Last_Error return Strings.chars_ptr
     with Import, Convention => C, External_Name => "mdh_last_error";
end Madarch_HIP;
