"""Wave timeline of simple_scene's screen pass (mode 2, space partition); needs a -DMDH_TIMELINE build selected
with MADARCH_HIP_LIBRARY."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from madarch_amd import examples, _binding as B
hb = B.hip_binding()
R = examples.global_illumination(1920, 1080, Probes=examples.GI_8X8X8_PROBES, Binding=hb)

for f in range(3): R.Render()
R.Finish()
R.Render_Pass(B.PASS_SCREEN); R.Finish()
n = 240 * 135
buf = np.zeros(3 * n, np.uint64)
hb.lib.mdh_diag_waves(buf.ctypes.data_as(C.c_void_p), n)
t = buf.reshape(n, 3)
t0, t1 = t[:, 0].astype(np.int64), t[:, 1].astype(np.int64)
base = t0.min()
start, end = (t0 - base) / 100.0, (t1 - base) / 100.0  # us
dur = end - start
print("kernel span %.1f us; wave duration mean %.1f us, median %.1f, p90 %.1f, p99 %.1f, max %.1f; sum of durations / span = %.0f waves resident on average" % (end.max(), dur.mean(), np.median(dur), np.percentile(dur, 90), np.percentile(dur, 99), dur.max(), dur.sum() / end.max()))
print("start time: p50 %.1f p75 %.1f p90 %.1f max %.1f" % tuple(np.percentile(start, [50, 75, 90, 100])))
edges = np.linspace(0, end.max(), 21)[:-1]
print("resident waves at t (us):", [(int(e), int(((start <= e) & (end > e)).sum())) for e in edges])
rows = dur.reshape(135, 240)
print("mean wave duration by tile row (top to bottom, groups of 15 rows):", [round(float(rows[i:i + 15].mean()), 1) for i in range(0, 135, 15)])
