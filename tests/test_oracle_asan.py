"""The CPU oracle under AddressSanitizer + UndefinedBehaviorSanitizer (`make -C oracle asan`; SURVEY.md section 5: sanitizers
on the CPU side -- the GPU pool runs none).  A child process preloads the sanitizer runtime, loads the instrumented
oracle and drives it through what the parity tests drive the two engines through: frames of every example scene (all
passes, the space partition with its three builders, volumetrics, both atlas formats, the four indirect-specular modes),
distance queries through the Madarch.Exprs tree walker, a scene of user-defined kinds, a sharded frame, and the pins of
tests/test_oracle_pins*.py.  Any report aborts the child."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ASAN_LIB = os.path.join(ROOT, "oracle", "libmadarch_oracle_asan.so")

FRAMES = r"""
import os, sys
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import numpy as np
from helpers import ODD_PROBES, SMALL_PROBES, make, snapshot
from madarch_amd import _binding as B, renderers
from madarch_amd.primitives import spheres, boxes
from oracle_engine import ORC_OPT_SDF_MODE, oracle_binding
orc = oracle_binding()
for scene, mode, atlas, probes in (("global_illumination", 0, 0, SMALL_PROBES), ("global_illumination", 0, 1, ODD_PROBES), ("simple_scene", 2, 0, None),
                                   ("simple_scene", 0, 0, SMALL_PROBES), ("light_shafts", 0, 0, None), ("simple_scene", 1, 0, None)):
    R = make(scene, 40, 24, orc, mode=mode, atlas=atlas, probes=probes)
    snapshot(R, 2)
    if scene == "simple_scene":
        for method in (renderers.CPU_Best, renderers.CPU_Fast, renderers.GPU_Fast):
            R.Update_Partitioning(method); R.Render(); R.Read_Partitioning(); R.Partition_Warnings()
        R.Eval_Distances_To(np.random.RandomState(1).uniform(-1, 6, (64, 3)).astype(np.float32), [spheres.Sphere, boxes.Box])
        R.Set_Option(ORC_OPT_SDF_MODE, 1); R.Render()   # every SDF through the Madarch.Exprs tree walker
    R.Swap_Buffers(); R.Front_Buffer()
    R.Destroy()
for spec in (0, 1, 3):
    R = make("global_illumination", 24, 16, orc, probes=SMALL_PROBES)
    R.Set_Option(B.OPT_INDIRECT_SPECULAR, spec); R.Set_Option(B.OPT_HYSTERESIS_PERMILLE, 300)
    snapshot(R, 2); R.Destroy()
R = make("global_illumination", 40, 24, orc, probes=SMALL_PROBES)   # a rank's share of a sharded frame
R.Set_Option(B.OPT_WORLD, 3); R.Set_Option(B.OPT_RANK, 2)
R.Render(); R.Read_Atlas_Slice(B.TEX_RADIANCE, 5, 7); R.Destroy()
import test_custom_kinds as tck                                       # user-defined kinds and lights: MDH_X programs interpreted by the oracle
for partition in (False, True):
    R = tck.room(orc, True, W=32, H=20, partition=partition, custom_lights=True)
    if partition: R.Update_Partitioning(renderers.GPU_Fast)
    R.Render(); R.Destroy()
print("ASAN_FRAMES_OK")
""" % (ROOT, ROOT)


def _env():
    runtime = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    ubsan = subprocess.run(["gcc", "-print-file-name=libubsan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(runtime) or not os.path.exists(runtime):
        pytest.skip("no libasan in this toolchain")
    env = dict(os.environ, LD_PRELOAD=runtime + (":" + ubsan if os.path.isabs(ubsan) and os.path.exists(ubsan) else ""),
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1",
               MADARCH_ORACLE_LIBRARY=ASAN_LIB, OMP_NUM_THREADS="2")
    return env


def _build():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "asan"])


def test_frames_of_every_scene_under_the_sanitizers():
    _build()
    out = subprocess.run([sys.executable, "-c", FRAMES], capture_output=True, text=True, timeout=900, env=_env(), cwd=ROOT)
    assert out.returncode == 0 and "ASAN_FRAMES_OK" in out.stdout, out.stdout[-1500:] + out.stderr[-4000:]
    assert "runtime error" not in out.stderr and "AddressSanitizer" not in out.stderr, out.stderr[-4000:]


def test_the_pins_under_the_sanitizers():
    _build()
    out = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", os.path.join(ROOT, "tests", "test_oracle_pins.py"),
                          os.path.join(ROOT, "tests", "test_oracle_pins64.py"), os.path.join(ROOT, "tests", "test_layout.py"), "-m", "not gpu"],
                         capture_output=True, text=True, timeout=1500, env=_env(), cwd=ROOT)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert "runtime error" not in out.stderr and "AddressSanitizer" not in out.stderr, out.stderr[-4000:]
