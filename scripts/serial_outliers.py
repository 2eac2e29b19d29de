"""On the GPU box: 600 device-synchronised frames of the headline workload one at a time, their median, mean and maximum and
the frames that took more than 1.3 x the median (a re-sort of the probe rays shows as +0.04 ms; anything larger is the box)."""
import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from madarch_amd import examples, _binding as B
hb = B.hip_binding()
R = examples.global_illumination(1920, 1080, Probes=examples.GI_8X8X8_PROBES, Binding=hb)
R.Set_Option(B.OPT_FRAME_OVERLAP, 0)
for _ in range(50): R.Render()
R.Finish()
ts = []
for f in range(600):
    t0 = time.perf_counter(); R.Render(); R.Finish(); ts.append(time.perf_counter() - t0)
ts = np.array(ts) * 1e3
med = np.median(ts)
print("median %.4f ms, mean %.4f, max %.4f" % (med, ts.mean(), ts.max()))
print("frames over 1.3x median:", [(i, round(float(t), 3)) for i, t in enumerate(ts) if t > 1.3 * med][:40])
