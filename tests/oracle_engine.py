"""TEST INFRASTRUCTURE: a `Binding` over the CPU oracle (oracle/libmadarch_oracle.so,
prefix orc_), so tests can drive the same host code (madarch_amd.renderers) against
the oracle and against the HIP library and compare.  Never imported by the product."""
import ctypes as C
import os
import subprocess

from madarch_amd import _binding as B

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
# MADARCH_ORACLE_LIBRARY selects another build of the oracle (the sanitizer build, tests/test_oracle_asan.py)
ORACLE_LIB = os.environ.get("MADARCH_ORACLE_LIBRARY") or os.path.join(ORACLE_DIR, "libmadarch_oracle.so")

ORC_OPT_SDF_MODE, ORC_OPT_THREADS = 100, 101

_binding = None


def build_oracle():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])


def oracle_lib():
    if os.environ.get("MADARCH_ORACLE_LIBRARY"):
        return C.CDLL(ORACLE_LIB)
    if not os.path.exists(ORACLE_LIB) or any(
            os.path.getmtime(os.path.join(ORACLE_DIR, f)) > os.path.getmtime(ORACLE_LIB)
            for f in os.listdir(ORACLE_DIR) if f.endswith((".c", ".h"))):
        build_oracle()
    return C.CDLL(ORACLE_LIB)


def oracle_binding():
    global _binding
    if _binding is None:
        lib = oracle_lib()
        _binding = B.Binding(lib, "orc_")
        lib.orc_sdf_evals.restype = C.c_uint64
        lib.orc_sdf_evals.argtypes = [C.c_void_p]
        lib.orc_sdf.restype = C.c_float
        lib.orc_exprs_sdf.restype = C.c_float
    return _binding
