"""Device D&C tridiagonal eigensolver (jcdf_stedc_device) vs scipy/LAPACK, correctness + timing."""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import scipy.linalg as sla
import juliachem_jl_amd
from juliachem_jl_amd import _lib
lib = _lib.load()
dev = torch.device("cuda", 0)
f64 = dict(dtype=torch.float64, device=dev)

def solve(d, e, reps=0):
    n = len(d)
    wb = int(lib.jcdf_stedc_workspace_bytes(n)); assert wb >= 0
    work = torch.empty(wb // 8 + 8, **f64)
    D = torch.as_tensor(d, **f64).clone(); E = torch.as_tensor(np.append(e, 0.0), **f64).clone()
    Z = torch.empty((n, n), **f64)
    st = torch.cuda.current_stream().cuda_stream
    p = lambda t: C.c_void_p(t.data_ptr())
    D0 = D.clone()
    rc = lib.jcdf_stedc_device(C.c_void_p(st), n, p(D), p(E), p(Z), n, p(work), wb); assert rc == 0, rc
    torch.cuda.synchronize()
    ms = None
    if reps:
        ts = []
        for _ in range(reps):
            D.copy_(D0); torch.cuda.synchronize(); t0 = time.perf_counter()
            lib.jcdf_stedc_device(C.c_void_p(st), n, p(D), p(E), p(Z), n, p(work), wb); torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        ms = np.median(ts)
    return D.cpu().numpy(), Z.cpu().numpy().T, ms      # Z rows are eigenvectors (column-major) -> columns after .T

def check(name, d, e, reps=0):
    n = len(d)
    w, V, ms = solve(d, e, reps)
    T = np.diag(d) + np.diag(e, 1) + np.diag(e, -1)
    wref = sla.eigvalsh_tridiagonal(d, e) if n > 1 else np.array(d)
    sc = max(1.0, np.abs(wref).max())
    werr = np.abs(w - wref).max() / sc
    orth = np.abs(V.T @ V - np.eye(n)).max()
    res = np.abs(T @ V - V * w[None, :]).max() / sc
    ok = werr < 1e-13 * max(n, 10) and orth < 1e-13 * max(n, 10) and res < 1e-13 * max(n, 10) and np.all(np.diff(w) >= 0)
    print("%-28s n=%4d  werr %.1e  orth %.1e  resid %.1e  %s%s" % (name, n, werr, orth, res, "ok" if ok else "FAIL",
          "  %.3f ms" % ms if ms else ""), flush=True)
    return ok

rng = np.random.default_rng(0)
allok = True
for n in (1, 2, 3, 5, 8, 17, 33, 64, 100, 255, 510, 700):
    allok &= check("random", rng.standard_normal(n), rng.standard_normal(max(n - 1, 0)))
allok &= check("identity", np.ones(64), np.zeros(63))
allok &= check("zero offdiag, repeated", np.repeat(np.arange(8.0), 8), np.zeros(63))
allok &= check("1-2-1 Toeplitz", 2 * np.ones(200), -np.ones(199))
allok &= check("Wilkinson W21+", np.abs(np.arange(-10, 11)).astype(float), np.ones(20))
gl = np.tile(np.abs(np.arange(-10, 11)).astype(float), 10); ge = np.ones(len(gl) - 1); ge[20::21] = 1e-8
allok &= check("glued Wilkinson", gl, ge)
allok &= check("graded", 10.0 ** -np.arange(0, 60, 0.5), 10.0 ** -np.arange(0.25, 59.5, 0.5)[:119])
allok &= check("clustered", 1.0 + 1e-10 * rng.standard_normal(300), 1e-10 * rng.standard_normal(299))
allok &= check("tiny couplings", rng.standard_normal(128), 1e-14 * rng.standard_normal(127))
# the matrices of the product: tridiagonalised random symmetric (as in the SCF bench) and a Fock-like spectrum
for n in (240, 510, 1250):
    A = rng.standard_normal((n, n)); A = A + A.T
    Hh, Qh = sla.hessenberg(A, calc_q=True)
    allok &= check("sytrd(random symmetric)", np.diag(Hh).copy(), np.diag(Hh, 1).copy(), reps=5)
print("ALL OK" if allok else "SOME FAILED")
