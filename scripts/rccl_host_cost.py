"""Host time of one sharded frame through the RCCL exchange path (one rank), split by call."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
import torch, torch.distributed as dist
from madarch_amd import examples, sharding, _binding as B
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, **({"device_id": torch.device("cuda", 0)} if os.environ.get("WITH_DEVICE_ID") else {}))
R = examples.global_illumination(1920, 1080, Probes=examples.GI_8X8X8_PROBES)
ex = sharding.DeviceExchange(dist, R, torch.device("cuda", 0))
frame = sharding.ShardedFrame(R, 0, 1, ex)
for _ in range(30): frame.Render()
R.Finish(); torch.cuda.synchronize()
N = 300
if os.environ.get("WITH_TIMING"): R.Set_Option(B.OPT_TIMING, 1)
acc = [0.0] * 6
t0 = time.perf_counter()
for _ in range(N):
    a = time.perf_counter(); R.Frame_Begin()
    b = time.perf_counter(); R.Frame_Probe_Pass(B.PASS_RADIANCE)
    c = time.perf_counter(); ex.all_gather(R, B.TEX_RADIANCE, 0, 1)
    d = time.perf_counter(); R.Frame_Probe_Pass(B.PASS_IRRADIANCE)
    e = time.perf_counter(); R.Frame_End()
    f = time.perf_counter()
    for i, v in enumerate((b - a, c - b, d - c, e - d, f - e)): acc[i] += v
host = (time.perf_counter() - t0) / N
R.Finish(); torch.cuda.synchronize()
total = (time.perf_counter() - t0) / N
print("host per frame %.1f us (begin %.1f, radiance %.1f, all_gather %.1f, irradiance %.1f, end %.1f); frame %.1f us" %
      ((host * 1e6,) + tuple(v / N * 1e6 for v in acc[:5]) + (total * 1e6,)))
dist.destroy_process_group()
