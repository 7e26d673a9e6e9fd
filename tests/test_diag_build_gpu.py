"""The DIAGNOSTIC build of the library (tools/build_diag.sh -> tools/_build/libjcdf_hip_diag.so; built by build.sh beside the
product) under test in the driver's suite: tests/test_diagnostic_paths.py — the two-stage tridiagonalisation, the Q replay,
the register-staged predecessor kernels, the W ablation switches — skips itself when the product library is loaded, so it is
run here in a CHILD process with JCDF_LIB_PATH pointing at the diagnostic build (no code in the tree
that the round's record does not exercise)."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DIAG = os.path.join(ROOT, "tools", "_build", "libjcdf_hip_diag.so")


def test_diagnostic_library_is_built_and_exports_the_product_abi_plus_its_own():
    """CPU side: build.sh produced the diagnostic library; it exports every product symbol + the entry points of csrc/jcdf_diag.h"""
    import ctypes
    from juliachem_jl_amd import _lib
    assert os.path.exists(DIAG), "run ./build.sh (tools/build_diag.sh)"
    lib = ctypes.CDLL(DIAG)
    for name in list(_lib.PROTOTYPES) + list(_lib.DIAG_PROTOTYPES):
        assert hasattr(lib, name), name


@pytest.mark.gpu
def test_diagnostic_paths_against_the_diagnostic_build():
    env = dict(os.environ, JCDF_LIB_PATH=DIAG)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_diagnostic_paths.py"), "-q", "-m", "gpu", "-x",
                        "-p", "no:cacheprovider"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    m = re.search(r"(\d+) passed", r.stdout)
    assert m and int(m.group(1)) >= 25 and "skipped" not in r.stdout.splitlines()[-1], tail
