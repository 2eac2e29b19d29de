"""The N > 1 frame (madarch_amd/sharding.py) with world_size 2 over gloo on the CPU: each rank
drives the CPU oracle through the same host code the GPU ranks use (probe slices, tile
dealing, atlas exchange), and the result must equal the single-rank frame bit for bit."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _probes(name):
    from helpers import ODD_PROBES, SMALL_PROBES
    from madarch_amd import renderers
    if name == "two":  # fewer probes than ranks: a rank with an empty slice
        return renderers.Probe_Settings(Radiance_Resolution=8, Irradiance_Resolution=4, Probe_Count=(2, 1), Grid_Dimensions=(1, 1, 2), Grid_Spacing=(3.0, 3.0, 4.0))
    return {"small": SMALL_PROBES, "odd": ODD_PROBES}[name]


def _worker(rank, world, port, scene, irr_all, probes, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    from helpers import make
    from madarch_amd import _binding as B
    from madarch_amd import sharding
    from oracle_engine import ORC_OPT_THREADS, oracle_binding
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    R = make(scene, 56, 40, oracle_binding(), probes=_probes(probes))
    R.Set_Option(ORC_OPT_THREADS, 2)
    R.Set_Option(B.OPT_IRRADIANCE_ALL, irr_all)  # 1: every rank updates all probes, one exchange per frame; 0: two
    frame = sharding.ShardedFrame(R, rank, world, sharding.HostExchange(dist))
    for _ in range(2):
        frame.Render()
    img = frame.Gather_Framebuffer(dist)
    if rank == 0:
        np.savez(os.path.join(out_dir, "sharded.npz"), image=img, radiance=R.Read_Texture(B.TEX_RADIANCE),
                 irradiance=R.Read_Texture(B.TEX_IRRADIANCE))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("scene,irr_all,probes,world", [
    ("global_illumination", 1, "small", 2), ("global_illumination", 0, "small", 2), ("light_shafts", 1, "small", 2),
    ("global_illumination", 0, "odd", 2),   # 75 probes over 2 ranks: slices of 37 and 38
    ("global_illumination", 0, "two", 3),   # 2 probes over 3 ranks: rank 0 owns none
])
def test_ranks_equal_one(orc, tmp_path, scene, irr_all, probes, world):
    import torch.multiprocessing as mp
    from helpers import make, same_bits, snapshot
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, scene, irr_all, probes, str(tmp_path)), nprocs=world, join=True)
    want = snapshot(make(scene, 56, 40, orc, probes=_probes(probes)), 2)
    with np.load(os.path.join(str(tmp_path), "sharded.npz")) as got:
        for k in ("image", "radiance", "irradiance"):
            assert same_bits(got[k], want[k]), k


def test_device_and_host_exchange_agree_on_the_slices():
    """DeviceExchange gathers in place: rank r's input is the byte range [offset, offset + size) of the atlas.  That
    range must be exactly the probes the host exchange reads and writes for the same rank, for even and uneven
    splits, and the ranges of all ranks must tile the atlas in rank order (what an all-gather of equal counts needs
    when the split is even)."""
    from madarch_amd import sharding
    for P in (512, 36, 75, 37, 2):
        for world in (1, 2, 3, 4, 8):
            for res, texel in ((32, 4), (8, 16), (12, 4)):
                per = res * res * texel
                end = 0
                for rank, (lo, hi) in enumerate(sharding.slice_bounds(P, world)):
                    off, size = sharding.slice_bytes(P, res, texel, rank, world)
                    assert (off, size) == (per * lo, per * (hi - lo)) and off == end
                    end = off + size
                assert end == P * per
                if P % world == 0:  # equal counts: the in-place form sendbuff = recvbuff + rank * count
                    assert all(sharding.slice_bytes(P, res, texel, r, world) == (r * (P // world) * per, (P // world) * per) for r in range(world))


def test_probe_slices_cover_all_probes():
    for P in (36, 512, 37):
        for world in (1, 2, 3, 4, 8):
            bounds = [(P * r // world, P * (r + 1) // world) for r in range(world)]
            assert bounds[0][0] == 0 and bounds[-1][1] == P
            assert all(a[1] == b[0] for a, b in zip(bounds, bounds[1:]))
