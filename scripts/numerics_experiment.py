"""A LABELLED EXPERIMENT, never the default: what BASELINE.json's tolerance (1e-4 relative per channel) would buy over this
build's own contract (the oracle's bits).  `make -C madarch_amd/csrc fast` builds the same sources with the hardware's
v_sqrt_f32 / v_rcp_f32 / v_log_f32 / v_exp_f32, fused multiply-adds and the irradiance fold in four partial sums
(mdh_device.h: MDH_FAST_NUMERICS).  This script renders BASELINE config 3 (1920x1080, DDGI 8x8x8) for FRAMES frames with
the exact library, the fast library and the oracle, and reports

  * the gate of BASELINE.md section 4 against the EXACT oracle: the share of pixels within 1e-4 relative (absolute floor
    1e-5) on every channel, the largest error, geometry-buffer index mismatches, atlas texels that differ;
  * Mpixels/s with frames in flight and the serial per-pass times of both libraries.

Run on the GPU box:  python scripts/numerics_experiment.py [rgb8|f32] [frames]"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
OUT = os.environ.get("TMPDIR", "/tmp")  # (three 1080p frames with their buffers: too large for gpurun_out)


def render(which, atlas, frames, out):
    import numpy as np
    from madarch_amd import _binding as B
    from madarch_amd import examples
    if which == "oracle":
        from oracle_engine import ORC_OPT_THREADS, oracle_binding
        binding = oracle_binding()
    else:
        binding = B.hip_binding()
    R = examples.global_illumination(1920, 1080, Probes=examples.GI_8X8X8_PROBES, Binding=binding)
    R.Set_Option(B.OPT_ATLAS_FORMAT, 0 if atlas == "rgb8" else 1)
    R.Set_Option(B.OPT_GBUFFER, 1)
    if which == "oracle":
        R.Set_Option(ORC_OPT_THREADS, len(os.sched_getaffinity(0)))
    for _ in range(frames):
        R.Render()
    idx, t, steps = R.Read_Gbuffer()
    np.savez(out, image=R.Read_Framebuffer(), gb_index=idx, gb_steps=steps, gb_t=t, radiance=R.Read_Texture(B.TEX_RADIANCE), irradiance=R.Read_Texture(B.TEX_IRRADIANCE))
    info = {"which": which, "version": (binding.version() or b"").decode() if which != "oracle" else "oracle", "numerics": R.Get_Option(B.OPT_NUMERICS)}
    if which != "oracle":
        R.Set_Option(B.OPT_GBUFFER, 0)
        for _ in range(100):
            R.Render()
        R.Finish()
        t0 = time.perf_counter()
        for _ in range(400):
            R.Render()
        R.Finish()
        info["mpix_in_flight"] = round(1920 * 1080 * 400 / (time.perf_counter() - t0) / 1e6, 1)
        R.Set_Option(B.OPT_FRAME_OVERLAP, 0)
        for _ in range(5):
            R.Render()
        R.Finish()
        R.Set_Option(B.OPT_TIMING, 1)
        R.Reset_Pass_Times()
        t0 = time.perf_counter()
        for _ in range(40):
            R.Render()
            R.Finish()
        info["mpix_serial"] = round(1920 * 1080 * 40 / (time.perf_counter() - t0) / 1e6, 1)
        info["passes_serial_ms"] = {B.PASS_NAMES[p]: round(R.Pass_Time(p)[0] / max(R.Pass_Time(p)[1], 1), 4) for p in range(len(B.PASS_NAMES)) if R.Pass_Time(p)[1]}
    print(json.dumps(info), flush=True)


def main():
    import numpy as np
    atlas = sys.argv[1] if len(sys.argv) > 1 else "rgb8"
    frames = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    libs = {"exact": os.path.join(ROOT, "madarch_amd", "csrc", "libmadarch_hip.so"), "fast": os.path.join(ROOT, "madarch_amd", "csrc", "libmadarch_hip_fast.so"),
            "hybrid": os.path.join(ROOT, "madarch_amd", "csrc", "libmadarch_hip_hybrid.so")}
    libs = {k: v for k, v in libs.items() if os.path.exists(v)}
    infos = {}
    for which in ("oracle",) + tuple(libs):
        env = dict(os.environ)
        if which in libs:
            env["MADARCH_HIP_LIBRARY"] = libs[which]
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "--render", which, atlas, str(frames), os.path.join(OUT, "numerics_%s.npz" % which)],
                             capture_output=True, text=True, env=env, timeout=900)
        assert out.returncode == 0, out.stderr[-3000:]
        infos[which] = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    want = np.load(os.path.join(OUT, "numerics_oracle.npz"))
    print("BASELINE config 3, 1920x1080, DDGI 8x8x8, %s atlases, frame %d of the probe feedback; the gate: every channel within 1e-4 relative (floor 1e-5) of the EXACT oracle" % (atlas, frames))
    for which in libs:
        got, info = np.load(os.path.join(OUT, "numerics_%s.npz" % which)), infos[which]
        img_ok = np.isclose(got["image"], want["image"], rtol=1e-4, atol=1e-5, equal_nan=True).all(axis=2)
        rel = np.abs(got["image"] - want["image"]) / np.maximum(np.abs(want["image"]), 1e-5)
        print("%-5s  %s" % (which, info["version"]))
        print("       pixels inside the gate %.4f %% (%d of %d outside), bit-equal %.4f %%, largest relative error %.3g, 99.9th percentile %.3g" % (
            100.0 * img_ok.mean(), (~img_ok).sum(), img_ok.size, 100.0 * (got["image"] == want["image"]).all(axis=2).mean(), float(np.nanmax(rel)), float(np.nanpercentile(rel, 99.9))))
        bad = np.argwhere(~img_ok)
        if 0 < len(bad) <= 12:
            print("       outside the gate (row, column): " + ", ".join("(%d, %d)" % (y, x) for y, x in bad))
        print("       geometry buffer: %d pixels with another primitive index, %d with another step count or march length; atlas texels that differ: radiance %d of %d, irradiance %d of %d" % (
            (got["gb_index"] != want["gb_index"]).sum(), ((got["gb_steps"] != want["gb_steps"]) | (got["gb_t"] != want["gb_t"])).sum(),
            (got["radiance"] != want["radiance"]).any(axis=2).sum(), want["radiance"].shape[0] * want["radiance"].shape[1],
            (got["irradiance"] != want["irradiance"]).any(axis=2).sum(), want["irradiance"].shape[0] * want["irradiance"].shape[1]))
        print("       %.1f Mpixels/s with frames in flight, %.1f one synchronised frame at a time; passes (serial, ms) %s" % (info["mpix_in_flight"], info["mpix_serial"], info["passes_serial_ms"]))
    e = infos["exact"]
    for which in libs:
        if which != "exact":
            f = infos[which]
            print("%s / exact: %.3f in flight, %.3f serial" % (which, f["mpix_in_flight"] / e["mpix_in_flight"], f["mpix_serial"] / e["mpix_serial"]))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--render":
        render(sys.argv[2], sys.argv[3], int(sys.argv[4]), sys.argv[5])
    else:
        main()
