import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


@pytest.fixture(scope="session")
def oracle_c():
    """ctypes handle of the plain-C oracle twin (oracle/c/jcdf_oracle.c), built on demand."""
    import ctypes
    import subprocess
    so = os.path.join(ROOT, "oracle", "_build", "libjcdf_oracle.so")
    src = os.path.join(ROOT, "oracle", "c", "jcdf_oracle.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        os.makedirs(os.path.dirname(so), exist_ok=True)
        subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-o", so, src])
    lib = ctypes.CDLL(so)
    i64, p = ctypes.c_int64, ctypes.c_void_p
    lib.jcdf_oracle_fock_dense.argtypes = [i64, i64, i64, p, p, p, ctypes.c_int, p]
    lib.jcdf_oracle_fock_dense.restype = ctypes.c_int
    lib.jcdf_oracle_form_B.argtypes = [i64, i64, p, p]
    lib.jcdf_oracle_form_B.restype = ctypes.c_int
    return lib
