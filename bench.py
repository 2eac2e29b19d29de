"""bench.py -- Mpixels/s of one full frame (Renderers.Render: DDGI radiance + irradiance
passes, then the per-pixel screen pass) of the global_illumination scene at 1920x1080,
BASELINE.json's metric and config 3, on N MI355X of one node.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = one frame.  The scene, atlases and tables are resident in HBM before the timed
region; frame read-back is excluded.  N > 1: image tiles and probe slices are split over
the ranks (fixed frame => strong scaling) with one in-place RCCL all-gather per frame (the
radiance slices; every rank folds the irradiance itself, MDH_OPT_IRRADIANCE_ALL) issued by
libmadarch_hip.so itself: the ranks' renderers join a communicator (mdh_comm_init) and
Renderers.Render is the sharded frame.  torch.distributed (gloo, CPU tensors) is the control
plane only -- the communicator id, barriers, max over ranks -- and, should the communicator not
form, the fall-back exchange through host memory (config.parallelism says which ran).
Prints one JSON line on rank 0.

`value` is the throughput of the library's default schedule (consecutive frames in flight,
MDH_OPT_FRAME_OVERLAP); `value_serial` / `ms_per_step_serial` beside it are SURVEY.md section
8(d)'s metric to the letter: the wall time of one device-synchronised Renderers.Render, all
passes, nothing else in flight.  `roofline` prices the dominant kernel on its duration inside the
timed region (as BENCH_r01 did), `roofline_serial` on its duration with the chip to itself (what
BENCH_r02's `roofline` held): "schema": 3 marks lines with this meaning.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec

WORKLOADS = {
    # name: (scene, width, height, probes, screen mode)
    "global_illumination_1080p_ddgi8x8x8": ("global_illumination", 1920, 1080, "gi8", 0),
    "simple_scene_1080p_direct": ("simple_scene", 1920, 1080, None, 2),  # BASELINE config 2: a build-defined subset (direct light + occlusion)
    "light_shafts_1080p": ("light_shafts", 1920, 1080, None, 0),
    "global_illumination_4096sq_ddgi8x8x8": ("global_illumination", 4096, 4096, "gi8", 0),
    # the reference's examples as they actually run: the renderer's fixed screen macros
    # (madarch-renderers.adb:136-143: direct specular, mode-2 indirect specular, 3 occlusion steps) with the default
    # probe settings (madarch-renderers.ads:23-29: 4x3x3 probes as a 6x6 atlas)
    "simple_scene_1080p_full": ("simple_scene", 1920, 1080, None, 0),  # through the space partition, CPU_Best (examples/simple_scene/main.adb:36-39,122)
    "global_illumination_1080p_default_probes": ("global_illumination", 1920, 1080, None, 0),  # "report the reference-default 4x3x3 once"
    # examples/ball_game/main.adb:39-40,196-252: ten balls; every step = physics query + ten Set_Primitive + Update_Partitioning + Render
    "ball_game_1080p": ("ball_game", 1920, 1080, None, 0),
    "ball_game_1080p_ddgi8x8x8": ("ball_game", 1920, 1080, "gi8", 0),  # (what scripts/ball_game_bench.py measured in rounds 2 and 3)
    # --rehearse-cpu only: the control flow of this file with N ranks on the CPU (gloo, the oracle as the engine)
    "rehearsal_small": ("global_illumination", 96, 64, None, 0),
}


def make_renderer(workload, binding, device=0, settle_frames=True):
    from madarch_amd import _binding as B
    from madarch_amd import examples
    scene, W, H, probes, mode = WORKLOADS[workload]
    P = examples.GI_8X8X8_PROBES if probes == "gi8" else None
    if scene == "ball_game":  # the game object rides on the renderer: make_step() below drives its loop body
        G = examples.ball_game(W, H, Probes=P, Binding=binding, Device=device)
        for _ in range(10):  # ten balls in the air, thrown from a moving camera (as scripts/ball_game_bench.py did)
            G.Throw_Ball()
            G.Move_Camera((0.2, 0.05, 0.0))
            for _ in range(3):
                if settle_frames:
                    G.Frame()
                else:  # (the CPU baseline: the same ball positions without thirty oracle frames)
                    G.Step_Physics()
        if not settle_frames:
            G.R.Update_Partitioning()
        R = G.R
        R.Game = G
    else:
        R = examples.SCENES[scene](W, H, Probes=P, Binding=binding, Device=device)
    R.Set_Option(B.OPT_SCREEN_MODE, mode)
    return R


def mode_of(R):
    from madarch_amd import _binding as B
    return R.Get_Option(B.OPT_SCREEN_MODE)


def make_step(R, frame):
    """One step of the workload: Renderers.Render -- or, for ball_game, the example's whole loop body
    (examples/ball_game/main.adb:244-252: physics with its distance queries, the scene edits, Update_Partitioning, Render)."""
    G = getattr(R, "Game", None)
    if G is None:
        return frame.Render

    def step():
        G.Step_Physics()
        R.Update_Partitioning()
        frame.Render()
    return step


def algorithmic_bytes_screen(R):
    """Compulsory HBM bytes of ONE screen-pass launch (with the
    reference's RGB8 atlases as RGBA8 texels): per pixel 16 B written + 8 bilinear
    irradiance taps + 1 bilinear radiance tap (4 texels x 4 B each) + the scene tables
    staged once per 256-pixel workgroup."""
    from madarch_amd import _binding as B
    texel = 4 if R.Get_Option(B.OPT_ATLAS_FORMAT) == 0 else 16
    mode = R.Get_Option(B.OPT_SCREEN_MODE)
    per_px = 16 + ((8 * 4 + 4) * texel if mode == 0 else 0)
    n_ubo, _ = R.Scene_Buffer_Size()
    tables = n_ubo + 656  # scene block + materials block (std140 sizes of the reference)
    world = R.Get_Option(B.OPT_WORLD)
    px = R.Width * R.Height / world
    return px * per_px + (px / 256.0) * tables, per_px, tables


def host_cpu_share():
    """Threads this process may really use: the scheduler affinity, cut by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        pass
    return max(1, n)


def measured_traffic(workload, world):
    """HBM bytes per k_screen launch from the PMC passes of the committed profile set (collected with
    scripts/collect_profiles.sh on the same command), or None when no profile matches this run."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")), reverse=True):
        try:
            t = json.load(open(path))
            if t.get("workload") == workload and t.get("n_gpus") == world:
                return int(t["kernels"]["k_screen"]["hbm_bytes_per_launch"]), os.path.basename(path)
        except (OSError, KeyError, ValueError):
            continue
    return None, None


VALU_PEAK_WAVE_INSTS = 256 * 4 * 2.4e9 / 2  # 256 CUs x 4 SIMDs, one wave64 VALU instruction per 2 cycles at 2.4 GHz (MI355X_MICROARCH.md)


def measured_valu(workload, world, passes):
    """Wave-level VALU instructions per k_screen launch (SQ_INSTS_VALU of the committed PMC pass) against
    the chip's VALU issue peak over the kernel's duration in this run -- the roofline that really bounds
    the path.  None when no profile matches."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")), reverse=True):
        try:
            t = json.load(open(path))
            if t.get("workload") == workload and t.get("n_gpus") == world:
                insts = float(t["kernels"]["k_screen"]["SQ_INSTS_VALU"])
                lanes = float(t["kernels"]["k_screen"]["SQ_THREAD_CYCLES_VALU"]) / float(t["kernels"]["k_screen"]["SQ_ACTIVE_INST_VALU"])
                ms = passes["screen"]["ms_avg"]
                rate = insts / (ms * 1e-3)
                return {"kernel": "k_screen", "wave_insts_per_launch": int(insts), "achieved_wave_insts_per_s": round(rate, 1),
                        "peak_wave_insts_per_s": VALU_PEAK_WAVE_INSTS, "frac": round(rate / VALU_PEAK_WAVE_INSTS, 4),
                        "active_lanes_per_inst": round(lanes, 1), "kernel_ms": ms, "source": os.path.basename(path)}
        except (OSError, KeyError, ValueError, ZeroDivisionError):
            continue
    return None


def cpu_baseline(workload):
    """The CPU oracle (a C restatement of the reference's path -- the Ada/GLSL reference
    cannot be built here) on the host cores: one warm-up frame, then whole frames for about ten seconds."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_engine import ORC_OPT_THREADS, oracle_binding
    R = make_renderer(workload, oracle_binding(), settle_frames=False)
    R.Set_Option(ORC_OPT_THREADS, host_cpu_share())
    cores = R.Get_Option(ORC_OPT_THREADS)
    R.Render()
    frames, t = 0, time.perf_counter()
    while True:  # about ten seconds of CPU work (the oracle does ~3 Mpixels/s on 16 threads)
        R.Render()
        frames += 1
        dt = time.perf_counter() - t
        if dt > 10.0 or frames >= 64:
            break
    mpix = R.Width * R.Height * frames / dt / 1e6
    # the work of one frame as the reference's shaders do it: SDF evaluations over all passes
    import ctypes as C
    lib = oracle_binding().lib
    evals = 0
    for p in range(5):
        out = (C.c_uint64 * 3)()
        lib.orc_work_counters(R._h, p, out)
        evals += int(out[2])
    R.Destroy()
    return {"value": round(mpix, 4), "unit": "Mpixels/s", "cores": cores, "kind": "port", "sdf_evals_per_frame": evals,
            "sample": "%d full %dx%d frames of the same workload (%.1f s) after 1 warm-up frame, all passes, OpenMP over rows" % (frames, R.Width, R.Height, dt)}


def cpu_baseline_exprs():
    """BASELINE config 1 as written: 256x256 simple_scene, primary rays only, every SDF and normal evaluated by the
    oracle's restatement of the Madarch.Exprs tree walker (Exprs.Eval / Primitives.Eval_Dist, reference
    madarch-exprs.adb:322-716, madarch-renderers.adb:499-526: heap context nodes, component lookup by name) --
    the CPU path north_star names, timed on the host cores."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from madarch_amd import _binding as B
    from madarch_amd import examples
    from oracle_engine import ORC_OPT_SDF_MODE, ORC_OPT_THREADS, oracle_binding
    R = examples.simple_scene(256, 256, Binding=oracle_binding())
    R.Set_Option(B.OPT_SCREEN_MODE, 1)
    R.Set_Option(ORC_OPT_SDF_MODE, 1)
    R.Set_Option(ORC_OPT_THREADS, host_cpu_share())
    cores = R.Get_Option(ORC_OPT_THREADS)
    R.Render()
    frames, t = 0, time.perf_counter()
    while True:
        R.Render()
        frames += 1
        dt = time.perf_counter() - t
        if dt > 5.0 or frames >= 64:
            break
    mpix = 256 * 256 * frames / dt / 1e6
    R.Destroy()
    return {"value": round(mpix, 4), "unit": "Mpixels/s", "cores": cores, "kind": "port", "evaluator": "Madarch.Exprs tree walker (oracle/orc_exprs.c)",
            "sample": "%d frames of 256x256 simple_scene, screen mode 1 (primary rays, normal colour), space partition on (%.1f s) after 1 warm-up frame" % (frames, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="global_illumination_1080p_ddgi8x8x8", choices=sorted(WORKLOADS))
    ap.add_argument("--atlas", default="rgb8", choices=("rgb8", "f32"))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--prewarm-s", type=float, default=0.4, help="seconds of untimed frames before the warm-up (clock ramp)")
    ap.add_argument("--mode", type=int, default=None, help="override the screen mode (ablation runs only)")
    ap.add_argument("--serial", action="store_true", help="MDH_OPT_FRAME_OVERLAP = 0: one pass after the other (per-kernel timing runs)")
    ap.add_argument("--overlap", type=int, default=None, help="MDH_OPT_FRAME_OVERLAP value (default: the library's)")
    ap.add_argument("--no-serial-segment", action="store_true", help="skip the untimed serial frames that give clean per-kernel durations")
    ap.add_argument("--swap-buffers", action="store_true", help="Swap_Buffers after every frame and fetch the window's RGBA8 pixels of the frame before (the PCIe-inclusive rate; never the default)")
    ap.add_argument("--animate-light", action="store_true", help="set the light anew before every frame, as the example's main loop does (global_illumination/main.adb:219-232)")
    ap.add_argument("--rehearse-rccl", action="store_true", help="one rank, but with a (one-rank) communicator: every frame runs the library's RCCL exchange (rehearsal of the N > 1 code path on one GPU)")
    ap.add_argument("--rehearse-cpu", action="store_true", help="TEST ONLY: this file's N-rank control flow on the CPU -- gloo, the oracle as the engine, a tiny frame; prints a line marked as a rehearsal, never a result")
    ap.add_argument("--numerics", default="exact", choices=("exact", "fast", "hybrid"), help="hybrid = the second labelled experiment (exact march loops, hardware arithmetic in shading only); fast = the LABELLED EXPERIMENT build (make -C madarch_amd/csrc fast: hardware sqrt / rcp / log / exp, tolerance only); the line then carries config.numerics = fast and is no result of the product")
    ap.add_argument("--screen-order", type=int, default=None, help="MDH_OPT_SCREEN_ORDER value (default: the library's)")
    ap.add_argument("--comm-timeout-s", type=float, default=120.0, help="watchdog of the communicator's join and trial frames (N > 1)")
    ap.add_argument("--exchange", default="auto", choices=("auto", "rccl", "peer", "host"), help="N > 1: the exchange backend (auto: rccl, then the peer exchange, then host memory; each falls back to the next)")
    ap.add_argument("--one-device", action="store_true", help="REHEARSAL: every rank on device 0 (a one-GPU box: the peer exchange between processes sharing the chip); the line says so")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (args.gpus, args.gpus))
        args.gpus = world
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL's peer-to-peer setup needs on this pool

    # The CONTROL plane of an N-rank run: torch.distributed over gloo, CPU tensors only (hands the communicator id
    # round, agrees on the fall-back, barriers, max over ranks).  The DATA plane -- the all-gather of the atlas slices --
    # is RCCL inside libmadarch_hip.so (mdh_comm_init): torch never sees a device pointer or a stream, and a one-rank run
    # does not import it at all.  (torch is imported BEFORE the library is loaded: INTEGRATION.md section 5.)
    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group("gloo")

    def vmax(x):  # the same value on every rank
        if dist is None:
            return x
        t = torch.tensor([x], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    if args.numerics != "exact":  # (selected before the binding reads the variable)
        os.environ["MADARCH_HIP_LIBRARY"] = os.path.join(ROOT, "madarch_amd", "csrc", "libmadarch_hip_%s.so" % args.numerics)
    from madarch_amd import _binding as B
    from madarch_amd import sharding

    rehearsal = args.rehearse_cpu
    if rehearsal:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from oracle_engine import ORC_OPT_THREADS, oracle_binding
        args.workload, args.no_cpu_baseline, args.prewarm_s = "rehearsal_small", True, 0.0
        R = make_renderer(args.workload, oracle_binding())
        R.Set_Option(ORC_OPT_THREADS, 2)
    else:
        R = make_renderer(args.workload, B.hip_binding(), device=0 if args.one_device else local_rank)
    R.Set_Option(B.OPT_ATLAS_FORMAT, 0 if args.atlas == "rgb8" else 1)
    if args.mode is not None:
        R.Set_Option(B.OPT_SCREEN_MODE, args.mode)
    if args.serial:
        R.Set_Option(B.OPT_FRAME_OVERLAP, 0)
    elif args.overlap is not None:
        R.Set_Option(B.OPT_FRAME_OVERLAP, args.overlap)
    if args.screen_order is not None:
        R.Set_Option(B.OPT_SCREEN_ORDER, args.screen_order)

    # N ranks: the renderers join a communicator inside the library and Renderers.Render of every rank becomes one frame
    # of the sharded schedule; if that fails anywhere, every rank falls back to the exchange through host memory over the
    # control plane -- in this process, and the line says so.
    exchange, how = None, "single rank"
    if world > 1:
        order = {"auto": ("rccl", "peer", "host"), "rccl": ("rccl", "peer", "host"), "peer": ("peer", "host"), "host": ("host",)}[args.exchange]
        exchange, how = sharding.establish(R, rank, world, dist, timeout_s=args.comm_timeout_s, backends=order,
                                           log=lambda m: print(m, file=sys.stderr, flush=True))
    elif args.rehearse_rccl:
        R.Comm_Init(R.Comm_Unique_Id(), 0, 1)
        how = "rccl"
    frame = sharding.ShardedFrame(R, rank, world, exchange)
    step = make_step(R, frame)

    def sync():  # barrier + device synchronisation: every pass this rank has enqueued is done, then every rank is here
        R.Finish()
        if dist is not None:
            dist.barrier()

    animate = None
    if args.animate_light and WORKLOADS[args.workload][0] == "global_illumination":
        import math
        from madarch_amd.lights import spot_lights
        clock = [0.0]

        def animate():
            clock[0] += 0.01
            R.Set_Light(1, spot_lights.Spot_Light, spot_lights.Create((3.5, 5.0, 2.0), (math.cos(clock[0]), math.sin(clock[0]), 0.0), 3.1415 / 4.0, (0.9, 0.9, 0.8)))
    if args.swap_buffers:
        R.Set_Option(B.OPT_WINDOW, 1)
    # untimed pre-warm: the clocks of a box that has just been handed over ramp up over the first few hundred
    # milliseconds of load (the driver's 5 warm-up frames are ~3 ms of GPU work), so frames run for at least
    # 0.4 s before the official warm-up starts.  Every frame of a sharded run is a collective: the ranks agree on
    # the elapsed time after each block (max over ranks), so all of them run the same number of blocks.
    t_pre = time.perf_counter()
    while True:
        for _ in range(2 if rehearsal else 25):
            step()
        R.Finish()
        if vmax(time.perf_counter() - t_pre) >= args.prewarm_s:
            break
    for _ in range(args.warmup):
        step()
    sync()
    R.Set_Option(B.OPT_TIMING, 1)
    R.Reset_Pass_Times()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        if animate:
            animate()
        step()
        if args.swap_buffers:  # pixels of the frame before reach the host while this one is drawn
            if _:
                R.Front_Buffer(copy=False)
            R.Swap_Buffers()
    if args.swap_buffers:
        R.Front_Buffer(copy=False)
    sync()
    dt = vmax(time.perf_counter() - t0)

    def pass_times():
        out = {}
        for p, name in enumerate(B.PASS_NAMES):
            ms, n = R.Pass_Time(p)
            if n:
                out[name] = {"ms_avg": round(ms / n, 4), "launches": n}
        return out

    passes = pass_times()
    # N > 1: the exchange's time per frame as the MAX over the ranks (every rank calls this; HIP events on the probe stream)
    exchange_ms = vmax(passes.get("exchange", {}).get("ms_avg", 0.0)) if world > 1 or "exchange" in passes else None
    overlap = R.Get_Option(B.OPT_FRAME_OVERLAP)
    passes_serial, serial_dt = None, None
    if (overlap or rehearsal) and not args.no_serial_segment:  # (the oracle has no schedule; the rehearsal walks the segment's control flow all the same)
        # Pipelined frames share the chip between kernels, which stretches every launch.  A short
        # untimed run of the strictly serial schedule gives each kernel's duration on its own.
        R.Set_Option(B.OPT_FRAME_OVERLAP, 0)
        for _ in range(1 if rehearsal else 3):
            step()
        sync()
        R.Reset_Pass_Times()
        for _ in range(1 if rehearsal else 10):
            step()
        sync()
        passes_serial = pass_times()
        # wall time of ONE device-synchronised Renderers.Render (all passes), nothing else in flight
        R.Set_Option(B.OPT_TIMING, 0)
        n_serial = 2 if rehearsal else max(10, min(args.steps, 50))
        sync()
        ts = time.perf_counter()
        for _ in range(n_serial):
            step()
            R.Finish()
        serial_dt = vmax((time.perf_counter() - ts) / n_serial)
        # SURVEY.md 8(d)'s steady-state protocol to the letter: atlases zeroed, 8 warm-up frames, frames 9 .. 40 timed one
        # by one (each device-synchronised), the MEDIAN reported
        if not rehearsal and mode_of(R) == 0 and getattr(R, "Game", None) is None:
            import numpy as np
            for tex in (B.TEX_RADIANCE, B.TEX_IRRADIANCE):
                R.Write_Texture(tex, np.zeros(R.Texture_Shape(tex), dtype=np.float32))
        per_frame = []
        for f in range(2 if rehearsal else 40):
            sync()
            tf = time.perf_counter()
            step()
            R.Finish()
            per_frame.append(vmax(time.perf_counter() - tf))
        steady = sorted(per_frame[(0 if rehearsal else 8):])
        steady_median = steady[len(steady) // 2] if len(steady) % 2 else 0.5 * (steady[len(steady) // 2 - 1] + steady[len(steady) // 2])
        R.Set_Option(B.OPT_FRAME_OVERLAP, overlap)

    parallelism = "tiles+probes/%d" % world
    if how == "rccl":
        parallelism += ", RCCL all-gather inside libmadarch_hip" + (" (one-rank rehearsal)" if world == 1 else "")
    elif how == "peer":
        parallelism += ", peer exchange inside libmadarch_hip (device-to-device copies between the ranks' processes, ordered by frame numbers polled on the device)"
    elif world > 1:
        parallelism += ", HOST EXCHANGE FALL-BACK over gloo (%s)" % how
    if args.one_device and world > 1:
        parallelism += " -- REHEARSAL: ALL %d RANKS ON ONE GPU" % world
    if rehearsal:
        img = frame.Gather_Framebuffer(dist)
        if rank == 0:
            import hashlib
            print(json.dumps({"rehearsal": True, "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
                              "frame_sha1": hashlib.sha1(img.tobytes()).hexdigest(), "parallelism": parallelism,
                              "serial_segment": passes_serial is not None, "engine": "CPU oracle over gloo (control flow only)"}), flush=True)
    elif rank == 0:
        W, H = R.Width, R.Height
        scene, _, _, probes, mode = WORKLOADS[args.workload]
        value = W * H * args.steps / dt / 1e6
        alg_bytes, per_px, tables = algorithmic_bytes_screen(R)
        screen_ms_piped = passes.get("screen", {}).get("ms_avg", float("nan"))
        # `roofline` prices the dominant kernel on its average duration inside the TIMED region (the schedule `value`
        # is measured on); `roofline_serial` on its duration with the chip to itself (the untimed serial segment):
        # inside pipelined frames two screen passes and the probe passes share the chip and every launch stretches
        screen_ms_serial = passes_serial["screen"]["ms_avg"] if passes_serial else None
        achieved = alg_bytes / (screen_ms_piped * 1e-3) / 1e9
        traffic, traffic_src = measured_traffic(args.workload, world) if args.atlas == "rgb8" and args.mode is None else (None, None)
        out = {
            "metric": "Mpixels/sec at %dx%d %s scene (one Renderers.Render frame: every pass of renderers.adb:302-321)" % (W, H, scene),
            "value": round(value, 3), "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic", "schema": 4,
            "config": {"workload": args.workload, "scene": scene, "width": W, "height": H,
                       "probe_grid": "8x8x8" if probes == "gi8" else "4x3x3 (the reference's default, renderers.ads:23-29)", "atlas": args.atlas,
                       "step": "physics query + scene edits + Update_Partitioning + Renderers.Render (examples/ball_game/main.adb:244-252)" if scene == "ball_game" else "Renderers.Render",
                       "screen_mode": mode, "parallelism": parallelism, "exchange": how,
                       "frame_overlap": overlap, "animated_light": bool(animate), "swap_buffers": bool(args.swap_buffers),
                       "radiance_order": int(R.Get_Option(B.OPT_RADIANCE_ORDER)), "screen_order": int(R.Get_Option(B.OPT_SCREEN_ORDER)),
                       "numerics": ("exact", "fast", "hybrid")[R.Get_Option(B.OPT_NUMERICS)]},
            # schema 4: `roofline` = the kernel's algorithmic bytes of ALL timed launches over the timed
            # region -- the figure that follows from the driver-timed `value` (consecutive frames draw on two streams, so
            # launches of this kernel overlap one another and each is stretched by its neighbours); the per-launch duration
            # inside the region is under `roofline_per_launch`, the kernel with the chip to itself under `roofline_serial`
            "roofline": {"bound": "hbm", "kernel": "k_screen", "achieved": round(alg_bytes * args.steps / dt / 1e9, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(alg_bytes * args.steps / dt / 1e9 / HBM_PEAK_GBS, 6), "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": int(alg_bytes), "bytes_per_pixel": per_px, "launches": args.steps,
                         "region_ms": round(dt * 1e3, 4),
                         "note": "the path is fp32-VALU bound (sphere tracing), not HBM bound; see DESIGN.md 'Roofline' and valu_issue below; achieved = algorithmic bytes of the k_screen launches of the timed region / the region's wall time"},
            "roofline_per_launch": {"bound": "hbm", "kernel": "k_screen", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                    "frac": round(achieved / HBM_PEAK_GBS, 6), "kernel_ms_avg": screen_ms_piped,
                                    "kernel_ms_source": "timed region, HIP events on the kernel's stream (launches of neighbouring frames overlap: each is stretched)",
                                    "launches_overlapping": round(screen_ms_piped * 1e-3 * args.steps / dt, 3)},
            "passes": passes,
        }
        if exchange_ms is not None:
            out["exchange"] = {"ms_avg_max_over_ranks": round(exchange_ms, 4), "how": how}
        if passes_serial:
            ach_s = alg_bytes / (screen_ms_serial * 1e-3) / 1e9
            out["roofline_serial"] = {"bound": "hbm", "kernel": "k_screen", "achieved": round(ach_s, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                      "frac": round(ach_s / HBM_PEAK_GBS, 6), "kernel_ms_avg": screen_ms_serial,
                                      "kernel_ms_source": "untimed serial segment after the timed region (MDH_OPT_FRAME_OVERLAP = 0), HIP events on the kernel's stream"}
            out["passes_serial"] = passes_serial
            out["value_serial"] = round(W * H / serial_dt / 1e6, 3)
            out["ms_per_step_serial"] = round(serial_dt * 1e3, 4)
            out["serial_note"] = "one device-synchronised Renderers.Render at a time (MDH_OPT_FRAME_OVERLAP = 0, host wait after every frame), averaged over %d frames: SURVEY.md 8(d)'s frame time; `value` keeps consecutive frames in flight" % n_serial
            out["steady_state"] = {"protocol": "SURVEY.md 8(d): atlases zeroed, 8 warm-up frames, frames 9..40 one device-synchronised frame at a time, median",
                                   "ms_median": round(steady_median * 1e3, 4), "value_median": round(W * H / steady_median / 1e6, 3),
                                   "ms_min": round(steady[0] * 1e3, 4), "ms_max": round(steady[-1] * 1e3, 4), "frames": len(steady)}
        valu = measured_valu(args.workload, world, passes_serial or passes)
        if valu:
            out["valu_issue"] = valu
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args.workload)
            out["cpu_baseline_exprs"] = cpu_baseline_exprs()
            # the reference's work per second: what the oracle evaluates for a frame / the GPU's frame time (the kernels do
            # less: DESIGN.md section 4 "Exact work elimination"; tests/test_gpu_work_counters.py holds the counts)
            out["useful_sdf_evals_per_s"] = round(out["cpu_baseline"]["sdf_evals_per_frame"] / (dt / args.steps), 1)
        print(json.dumps(out), flush=True)
    # leave in order: the communicator while every rank is still here, then the renderer, then the control plane
    if how == "rccl":
        R.Comm_Barrier()
        R.Comm_Destroy()
    elif how == "peer":
        R.Finish()
        dist.barrier()  # (nobody closes handles a peer still copies from)
        R.Comm_Destroy()
    if dist is not None:
        dist.barrier()
    R.Destroy()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
