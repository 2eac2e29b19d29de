"""The N > 1 frame (madarch_amd/sharding.py) with world_size 2 over gloo on the CPU: each rank
drives the CPU oracle through the same host code the GPU ranks use (probe slices, tile
dealing, atlas exchange), and the result must equal the single-rank frame bit for bit."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, scene, irr_all, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    from helpers import SMALL_PROBES, make
    from madarch_amd import _binding as B
    from madarch_amd import sharding
    from oracle_engine import ORC_OPT_THREADS, oracle_binding
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    R = make(scene, 56, 40, oracle_binding(), probes=SMALL_PROBES)
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


@pytest.mark.parametrize("scene,irr_all", [("global_illumination", 1), ("global_illumination", 0), ("light_shafts", 1)])
def test_two_ranks_equal_one(orc, tmp_path, scene, irr_all):
    import torch.multiprocessing as mp
    from helpers import SMALL_PROBES, make, same_bits, snapshot
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, scene, irr_all, str(tmp_path)), nprocs=2, join=True)
    want = snapshot(make(scene, 56, 40, orc, probes=SMALL_PROBES), 2)
    with np.load(os.path.join(str(tmp_path), "sharded.npz")) as got:
        for k in ("image", "radiance", "irradiance"):
            assert same_bits(got[k], want[k]), k


def test_probe_slices_cover_all_probes():
    for P in (36, 512, 37):
        for world in (1, 2, 3, 4, 8):
            bounds = [(P * r // world, P * (r + 1) // world) for r in range(world)]
            assert bounds[0][0] == 0 and bounds[-1][1] == P
            assert all(a[1] == b[0] for a, b in zip(bounds, bounds[1:]))
