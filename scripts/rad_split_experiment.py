"""the radiance pass as two kernels (MADARCH_HIP_RAD_SPLIT=1) against the shipped one-kernel pass: both
atlases bit for bit, the pass's serial time at BASELINE config 3 and on rank 0's slice of an 8-way sharded frame, the frame
rate in flight.  Run on the GPU box:  python scripts/rad_split_experiment.py"""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(tag):
    import numpy as np
    from madarch_amd import _binding as B, examples
    hb = B.hip_binding()
    out = {"tag": tag}
    for name, world in (("whole", 1), ("slice_1_of_8", 8)):
        R = examples.global_illumination(1920, 1080, Probes=examples.GI_8X8X8_PROBES, Binding=hb)
        R.Set_Option(B.OPT_WORLD, world); R.Set_Option(B.OPT_RANK, 0)
        for _ in range(12): R.Render()
        R.Finish()
        if world == 1:
            np.savez(os.path.join(os.environ.get("TMPDIR", "/tmp"), "rad_split_%s.npz" % tag), rad=R.Read_Texture(B.TEX_RADIANCE), irr=R.Read_Texture(B.TEX_IRRADIANCE), img=R.Read_Framebuffer())
            t0 = time.perf_counter()
            for _ in range(300): R.Render()
            R.Finish()
            out["mpix_in_flight"] = round(1920 * 1080 * 300 / (time.perf_counter() - t0) / 1e6, 1)
        R.Set_Option(B.OPT_FRAME_OVERLAP, 0)
        for _ in range(5): R.Render()
        R.Finish(); R.Set_Option(B.OPT_TIMING, 1); R.Reset_Pass_Times()
        for _ in range(40): R.Render()
        R.Finish()
        ms, n = R.Pass_Time(B.PASS_RADIANCE)
        out["radiance_ms_" + name] = round(ms / n, 4)
        R.Destroy()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        run(sys.argv[1])
    else:
        import numpy as np
        res = {}
        for tag, env in (("one_kernel", {}), ("two_kernels", {"MADARCH_HIP_RAD_SPLIT": "1"})):
            o = subprocess.run([sys.executable, os.path.abspath(__file__), tag], capture_output=True, text=True, env=dict(os.environ, **env), timeout=600)
            assert o.returncode == 0, o.stderr[-2000:]
            res[tag] = json.loads([l for l in o.stdout.splitlines() if l.startswith("{")][-1])
            print(res[tag])
        a, b = (np.load(os.path.join(os.environ.get("TMPDIR", "/tmp"), "rad_split_%s.npz" % t)) for t in ("one_kernel", "two_kernels"))
        print("bit-equal: radiance %s, irradiance %s, image %s" % tuple(bool(np.array_equal(a[k], b[k], equal_nan=True)) for k in ("rad", "irr", "img")))
