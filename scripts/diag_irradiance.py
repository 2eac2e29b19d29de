"""Cycles of k_irradiance's four wavefronts in their pipeline stage and at the chunk barrier (needs a -DMDH_PHASES build
selected with MADARCH_HIP_LIBRARY)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from madarch_amd import examples, _binding as B
hb = B.hip_binding()
R = examples.global_illumination(64, 64, Probes=examples.GI_8X8X8_PROBES, Binding=hb)
R.Set_Option(B.OPT_FRAME_OVERLAP, 0)
buf = (C.c_ulonglong * 16)()
for _ in range(3): R.Render()
R.Finish(); hb.lib.mdh_diag_phases(buf)
R.Render_Pass(B.PASS_IRRADIANCE); R.Finish(); hb.lib.mdh_diag_phases(buf)
for w in range(4):
    print("wavefront %d: loop work %8.0f cycles per probe, barrier wait %8.0f, staging before the loop %8.0f" % (w, buf[2 * w] / 512, buf[2 * w + 1] / 512, buf[8 + w] / 512))
