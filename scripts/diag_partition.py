"""Where the space-partition lookups of simple_scene's screen pass go (needs a -DMDH_DIAG build selected with
MADARCH_HIP_LIBRARY): SDF evaluations per march loop, lookups, candidate-pair iterations and the lanes alive in them."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from madarch_amd import examples, _binding as B
hb = B.hip_binding()
R = examples.simple_scene(1920, 1080, Binding=hb)
R.Set_Option(B.OPT_SCREEN_MODE, 2)
buf = (C.c_ulonglong * 16)()
for f in range(2): R.Render()
R.Finish(); hb.lib.mdh_diag_read(buf)
R.Render_Pass(B.PASS_SCREEN); R.Finish(); hb.lib.mdh_diag_read(buf)
names = {0: "primary march steps", 1: "soft shadow steps", 5: "lookups reaching a cell", 6: "candidate-pair iterations", 7: "kind visits (lanes = candidates of all lanes)"}
waves = 1920 * 1080 / 64
for t, n in names.items():
    e, l = buf[2 * t], buf[2 * t + 1]
    if e: print("%-48s wave-level %10d (%.1f per wave)  lanes per %5.1f" % (n, e, e / waves, l / e))
