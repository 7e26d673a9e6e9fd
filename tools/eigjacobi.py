"""Probe: rocSOLVER syevj (Jacobi), cold and warm-started in the previous eigenbasis."""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import juliachem_jl_amd
from juliachem_jl_amd.eigh import DeviceEigh
dev = torch.device("cuda", 0)
for n in (510,):
    rng = np.random.default_rng(0); A = rng.standard_normal((n, n)); A = A + A.T
    Ag = torch.as_tensor(A, device=dev)
    eg = DeviceEigh(n, dev)
    w_ref, U_ref = eg(Ag); torch.cuda.synchronize(); w_ref = w_ref.clone(); U_ref = U_ref.clone()
    p = lambda x: C.c_void_p(x.data_ptr())
    i32 = dict(dtype=torch.int32, device=dev); f64 = dict(dtype=torch.float64, device=dev)
    resid = torch.zeros(1, **f64); nsw = torch.zeros(1, **i32); W = torch.zeros(n, **f64); info = torch.zeros(1, **i32)
    Aw = torch.zeros((n, n), **f64)
    eg.rs.rocsolver_dsyevj.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_double, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    def t(f, reps=3):
        f(); torch.cuda.synchronize(); ts = []
        for _ in range(reps):
            torch.cuda.synchronize(); t0 = time.perf_counter(); f(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
        return np.median(ts)
    for label, M in (("cold", Ag),
                     ("warm 1e-2", U_ref.T @ (Ag + 1e-2 * torch.as_tensor(rng.standard_normal((n, n)), device=dev).triu().add(0)) @ U_ref),
                     ("warm 1e-4", U_ref.T @ (Ag + 1e-4 * torch.as_tensor(rng.standard_normal((n, n)), device=dev)) @ U_ref)):
        M = 0.5 * (M + M.T)
        def jac():
            Aw.copy_(M)
            rc = eg.rs.rocsolver_dsyevj(eg.handle, 252, 211, 122, n, p(Aw), n, 0.0, p(resid), 30, p(nsw), p(W), p(info))
            assert rc == 0, rc
        ms = t(jac)
        wr = torch.linalg.eigvalsh(M)
        print("n=%d syevj %s: %.2f ms  sweeps=%d info=%d  werr %.2e" % (n, label, ms, int(nsw.item()), int(info.item()), float((W - wr).abs().max())), flush=True)
