"""A slice of the differential fuzzers in the GPU suite: random scenes (tests/fuzz_scenes.py) on the HIP library
against the oracle.  scripts/fuzz_parity.py runs any number of further seeds."""
import pytest

import fuzz_kinds
from fuzz_scenes import build, compare

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", range(20000, 20016))
def test_random_scene(hip, orc, seed):
    compare(build(seed, hip), build(seed, orc))


@pytest.mark.parametrize("seed", range(30000, 30003))
def test_random_user_defined_kind(hip, orc, seed):
    """random expression trees: hiprtc build, interpreter build and oracle"""
    want = fuzz_kinds.build(seed, orc, 0)
    for jit in (1, 0):
        fuzz_kinds.compare(fuzz_kinds.build(seed, hip, jit), want, "(jit %d)" % jit)
