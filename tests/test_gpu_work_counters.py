""""instrument the oracle to emit exact sdf_evals, march_steps, rays per pass; the kernel must report
the same counters (debug build) -- that equality is itself a parity check".

The shipped kernels do LESS than the oracle on purpose (DESIGN.md section 4 "Exact work elimination": folded cage corners,
null rays, the shared first step, twins, the ray queue), so the equality is held by the LITERAL diagnostic build
(`make -C madarch_amd/csrc literal`: the same kernels with every elimination switched off and the work counters in):
per pass, lanes of the kernel = calls of the oracle for

    rays          raycast / raycast_hit_position / raycast_visibility / softshadows started
    march steps   SDF evaluations inside their loops
    SDF evals     all evaluations (march steps, occlusion taps) -- the kernels evaluate the arg-min once at a hit point
                  instead of carrying it through every step of raycast (raymarching.glsl:25-37): those evaluations are
                  counted apart and taken off

and the literal build's frames are the oracle's, bit for bit, like the shipped build's.  scripts/work_counters.py reports
what the eliminations remove (the `counters` build against this one)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LITERAL = os.path.join(ROOT, "madarch_amd", "csrc", "libmadarch_hip_literal.so")

SCRIPT = r"""
import ctypes as C, os, sys
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import numpy as np
from helpers import ODD_PROBES, SMALL_PROBES, assert_parity, make, snapshot
from madarch_amd import _binding as B
from oracle_engine import oracle_binding
hip, orc = B.hip_binding(), oracle_binding()
assert hasattr(hip.lib, "mdh_diag_work"), "not the literal build"
def gpu_work(R, p):
    out = (C.c_ulonglong * 4)()
    assert hip.lib.mdh_diag_work(R._h, p, out) == 0
    return list(out)
def orc_work(R, p):
    out = (C.c_uint64 * 3)()
    assert orc.lib.orc_work_counters(R._h, p, out) == 0
    return list(out)
cases = [("global_illumination", 0, SMALL_PROBES, 2), ("global_illumination", 0, ODD_PROBES, 2), ("simple_scene", 2, None, 2), ("simple_scene", 0, SMALL_PROBES, 2),
         ("light_shafts", 0, None, 2), ("global_illumination", 0, SMALL_PROBES, 1), ("global_illumination", 0, SMALL_PROBES, 3), ("simple_scene", 1, None, 2)]
for scene, mode, probes, spec in cases:
    Rs = []
    for b in (hip, orc):
        R = make(scene, 72, 40, b, mode=mode, probes=probes)
        R.Set_Option(B.OPT_INDIRECT_SPECULAR, spec)
        Rs.append(R)
    got, want = snapshot(Rs[0], 2), snapshot(Rs[1], 2)
    assert_parity(got, want)
    passes = [B.PASS_SCREEN] + ([B.PASS_RADIANCE] if mode == 0 else []) + ([B.PASS_VISIBILITY, B.PASS_SCATTERING] if scene == "light_shafts" else [])
    for p in passes:
        g, o = gpu_work(Rs[0], p), orc_work(Rs[1], p)
        assert o[0] > 0 and o[1] > 0, (scene, p, o)
        assert g[0] == o[0], ("rays", scene, mode, spec, B.PASS_NAMES[p], g, o)
        assert g[1] == o[1], ("march steps", scene, mode, spec, B.PASS_NAMES[p], g, o)
        assert g[2] - g[3] == o[2], ("SDF evaluations", scene, mode, spec, B.PASS_NAMES[p], g, o)
    print(scene, mode, spec, "ok", {B.PASS_NAMES[p]: orc_work(Rs[1], p) for p in passes})
print("WORK_COUNTERS_OK")
""" % (ROOT, ROOT)


def test_literal_kernels_do_the_oracles_work():
    assert os.path.exists(LITERAL), "build first (python -c 'import __graft_entry__ as g; g.build()')"
    env = dict(os.environ, MADARCH_HIP_LIBRARY=LITERAL)
    out = subprocess.run([sys.executable, "-c", SCRIPT], capture_output=True, text=True, timeout=600, env=env)
    assert "WORK_COUNTERS_OK" in out.stdout, out.stdout[-3000:] + out.stderr[-3000:]
