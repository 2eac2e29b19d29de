"""The C++ host mirror (include/madarch.hpp) and the three example programs restated on it
(examples/*.cpp): they build against the C ABI, fail loudly without a device, and on the GPU
box produce the very frames the Python mirror produces for the same calls."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "examples", "bin")
NAMES = ("global_illumination", "simple_scene", "light_shafts")


def build():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "examples")])


def test_examples_build_and_refuse_to_run_without_a_gpu():
    import torch
    build()
    for n in NAMES:
        assert os.path.exists(os.path.join(BIN, n))
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    out = subprocess.run([os.path.join(BIN, "global_illumination"), "16", "16", "1"], capture_output=True, text=True)
    assert out.returncode == 1 and "no HIP device" in out.stderr  # Program_Error from MDH_E_NO_DEVICE


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_cpp_example_matches_python_mirror(hip, tmp_path, name):
    from helpers import same_bits
    from madarch_amd import examples
    build()
    W, H, frames = 72, 48, 2
    path, ppm = str(tmp_path / (name + ".f32")), str(tmp_path / (name + ".ppm"))
    out = subprocess.run([os.path.join(BIN, name), str(W), str(H), str(frames), path, ppm], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    got = np.fromfile(path, dtype=np.float32).reshape(H, W, 3)
    R = examples.SCENES[name](W, H, Binding=hip)
    for _ in range(frames):
        R.Render()
        R.Swap_Buffers()
    assert same_bits(got, R.Read_Framebuffer())
    raw = open(ppm, "rb").read()
    head = ("P6\n%d %d\n255\n" % (W, H)).encode()
    assert raw.startswith(head) and len(raw) == len(head) + W * H * 3
    assert (np.frombuffer(raw[len(head):], dtype=np.uint8).reshape(H, W, 3) == R.Front_Buffer()[..., :3]).all()
