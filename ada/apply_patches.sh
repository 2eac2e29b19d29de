#!/bin/bash
# The three edits the HIP back end needs in Madarch's own sources, applied to a checkout of
# Roldak/Madarch.  Usage: ada/apply_patches.sh <madarch checkout> [<output dir>]
# Without an output directory the files are edited in place; with one, patched copies of the three files are
# written there and the checkout is left alone (what scripts/check_ada_sources.py does).
#   1. madarch/madarch-scenes.ads   Scene_Internal remembers the Max_Dist Compile was called with
#   2. madarch/madarch-scenes.adb   Compile stores it
#   3. madarch/madarch-renderers.ads  the private Renderer_Internal record shrinks to what the HIP body uses
# Everything else under madarch/ stays as it is; ada/*.ad[sb] of this repository are added next to it and
# ada/madarch-renderers.adb replaces the OpenGL body.
set -e
SRC=${1:?usage: apply_patches.sh <madarch checkout> [<output dir>]}
OUT=${2:-$SRC/madarch}
M=$SRC/madarch
mkdir -p "$OUT"
for f in madarch-scenes.ads madarch-scenes.adb madarch-renderers.ads; do
   [ "$OUT" = "$M" ] || cp "$M/$f" "$OUT/$f"
done
# 1. one more component behind GPU_Type in Scene_Internal
awk '{ print } /^ *GPU_Type *: GPU_Types.GPU_Type;/ && !done { print "      Max_Dist : GL.Types.Single := 20.0;   --  (HIP back end) what Compile was called with"; done = 1 }' \
   "$OUT/madarch-scenes.ads" > "$OUT/.tmp" && mv "$OUT/.tmp" "$OUT/madarch-scenes.ads"
# 2. ... set in the aggregate of Compile, behind the GPU_Type association
awk '{ print } /GPU_Type => Compute_Scene_GPU_Type/ && !done { print "         Max_Dist => Max_Dist,"; done = 1 }' \
   "$OUT/madarch-scenes.adb" > "$OUT/.tmp" && mv "$OUT/.tmp" "$OUT/madarch-scenes.adb"
# 3. the renderer's record: window, scene, the library's handle, the material counter
awk '
   /^ *type Renderer_Internal is record/ && !done {
      print "   type Renderer_Internal is record"
      print "      Window : Windows.Window;"
      print "      Scene  : Scenes.Scene;"
      print "      Handle : System.Address;   --  Madarch_HIP.Handle (mdh_renderer *)"
      print "      Last_Material_Index : Materials.Id := 0;"
      print "   end record;"
      skip = 1; done = 1; next
   }
   skip && /end record;/ { skip = 0; next }
   skip { next }
   /^package Madarch.Renderers is/ { print "with System;"; print; next }
   { print }' "$OUT/madarch-renderers.ads" > "$OUT/.tmp" && mv "$OUT/.tmp" "$OUT/madarch-renderers.ads"
grep -q "Max_Dist : GL.Types.Single" "$OUT/madarch-scenes.ads"
grep -q "Max_Dist => Max_Dist," "$OUT/madarch-scenes.adb"
grep -q "Handle : System.Address" "$OUT/madarch-renderers.ads"
echo "patched: $OUT/madarch-scenes.ads $OUT/madarch-scenes.adb $OUT/madarch-renderers.ads"
