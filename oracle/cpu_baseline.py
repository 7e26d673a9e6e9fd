"""CPU baseline of bench.py: the reference's two CPU Fock-build modes — dense
`df_rhf_fock_build_BLAS!` (DensityFitting.jl:111-125,185-224) and the default screened / blocked
`df_rhf_fock_build_screened!` (ScreenedDF.jl:80-132) with the reference's threading scheme — restated in C
(oracle/c/jcdf_cpu_baseline.c) on the best host BLAS found at run time.

TEST / BENCH INFRASTRUCTURE: imported by bench.py's `cpu_baseline` leg and by tests/ only."""
import ctypes as C
import glob
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "_build", "libjcdf_cpu_baseline.so")
_P, _I64 = C.c_void_p, C.c_int64


def _candidates():
    out = []
    for p in ("/opt/conda/lib/libmkl_rt.so", "/opt/conda/lib/libmkl_rt.so.1", "/opt/conda/lib/libmkl_rt.so.2"):
        if os.path.exists(p):
            out.append((p, 0))
            break
    for p in glob.glob(os.path.join(os.path.dirname(np.__file__), "..", "numpy.libs", "libscipy_openblas64_*.so")):
        out.append((os.path.abspath(p), 1))
    return out


def host_cores() -> int:
    """CPUs this process may actually use: the affinity mask, cut down to the cgroup's CPU quota where one is set (a
    GPU box of the pool shows all of the host's hardware threads in the mask but grants a share of them)."""
    n = len(os.sched_getaffinity(0))
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(float(txt[0]) / float(txt[1]) + 0.999)))
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, (q + per - 1) // per))
            break
        except Exception:
            continue
    return n


# numpy's OpenBLAS is compiled for at most 64 threads (its thread-metadata table overflows beyond: heap corruption)
MAX_THREADS = {0: 1 << 30, 1: 64}


class CpuBaseline:
    def __init__(self, threads=None, calibrate_n=4096):
        self.lib = C.CDLL(LIB)
        self.lib.jcbl_load_blas.argtypes = [C.c_char_p, C.c_int]
        self.lib.jcbl_blas_name.restype = C.c_char_p
        self.lib.jcbl_dgemm_calibration.restype = C.c_double
        self.lib.jcbl_dgemm_calibration.argtypes = [_I64, C.c_int, C.c_int]
        self.lib.jcbl_fock_dense.argtypes = [_I64, _I64, _I64, _P, _P, _P, _P, _P, C.c_int]
        self.lib.jcbl_fock_screened.argtypes = [_I64, _I64, _I64, _I64, _P, _P, _P, _P, _P, _P, _P, _P, _P, C.c_int, C.c_int]
        want = int(threads or host_cores())
        self.calibration = {}
        best = None
        for path, kind in _candidates():
            if self.lib.jcbl_load_blas(path.encode(), kind) != 0:
                continue
            name = self.lib.jcbl_blas_name().decode()
            nthr = min(want, MAX_THREADS[kind])
            gf = float(self.lib.jcbl_dgemm_calibration(calibrate_n, 2, nthr))
            self.calibration[name] = {"gflops": gf, "threads": nthr}
            if best is None or gf > best[2]:
                best = (path, kind, gf, name, nthr)
        if best is None:
            raise OSError("no host BLAS could be loaded (looked for libmkl_rt and numpy's OpenBLAS)")
        assert self.lib.jcbl_load_blas(best[0].encode(), best[1]) == 0
        self.blas, self.dgemm_gflops, self.calibrate_n, self.threads = best[3], best[2], calibrate_n, best[4]

    def fock_dense(self, B, C_occ, H):
        """B (Q, N, N) Fortran order, C_occ (N, o), H (N, N) -> (F, times{density,V,J,W,K})"""
        Q, N, _ = B.shape
        o = C_occ.shape[1]
        B = np.asfortranarray(B); Co = np.asfortranarray(C_occ); H = np.asfortranarray(H)
        F = np.zeros((N, N), order="F")
        t = np.zeros(5)
        rc = self.lib.jcbl_fock_dense(N, Q, o, B.ctypes.data, Co.ctypes.data, H.ctypes.data, F.ctypes.data, t.ctypes.data, self.threads)
        if rc != 0:
            raise RuntimeError("jcbl_fock_dense rc=%d" % rc)
        return F, dict(zip(("density", "V", "J", "W", "K"), t.tolist()))

    def fock_screened(self, Bp, sd, C_occ, H, n_blocks=10):
        """Bp (Q, P) Fortran order packed by `sd` (oracle.df_fock.ScreeningData), C_occ (N, o)"""
        Q, P = Bp.shape
        N, o = C_occ.shape
        Bp = np.asfortranarray(Bp); Co = np.asfortranarray(C_occ); H = np.asfortranarray(H)
        start = np.ascontiguousarray(sd.sparse_p_start_indices, dtype=np.int64)
        count = np.ascontiguousarray(sd.non_screened_p_indices_count, dtype=np.int64)
        qlist = np.ascontiguousarray(sd.pq_q, dtype=np.int64)
        diag = np.ascontiguousarray(np.diag(sd.sparse_pq_index_map), dtype=np.int64)
        assert (diag >= 0).all(), "the screened CPU mode needs every (p,p) kept (ScreenedDF.jl:318-365 starts its runs at map[p,p])"
        F = np.zeros((N, N), order="F")
        t = np.zeros(5)
        rc = self.lib.jcbl_fock_screened(N, Q, o, P, Bp.ctypes.data, start.ctypes.data, count.ctypes.data, qlist.ctypes.data,
                                         diag.ctypes.data, Co.ctypes.data, H.ctypes.data, F.ctypes.data, t.ctypes.data, self.threads, n_blocks)
        if rc != 0:
            raise RuntimeError("jcbl_fock_screened rc=%d" % rc)
        return F, dict(zip(("density", "V", "J", "W", "K"), t.tolist()))
