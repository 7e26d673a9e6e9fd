"""Probe: rocSOLVER stedc vs the (unlisted) stedcj / stedcx on the tridiagonal matrix of sytrd."""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import juliachem_jl_amd
from juliachem_jl_amd.eigh import DeviceEigh
dev = torch.device("cuda", 0)
for n, nev in ((510, 81), (1250, 250)):
    rng = np.random.default_rng(0); A = rng.standard_normal((n, n)); A = A + A.T
    Ag = torch.as_tensor(A, device=dev)
    eg = DeviceEigh(n, dev)
    w_ref, U_ref = eg(Ag); torch.cuda.synchronize(); w_ref = w_ref.clone()
    p = lambda x: C.c_void_p(x.data_ptr())
    i32 = dict(dtype=torch.int32, device=dev); f64 = dict(dtype=torch.float64, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    eg.A.copy_(Ag); eg.lib.jcdf_sytrd_device(C.c_void_p(st), n, p(eg.A), n, p(eg.D), p(eg.E), p(eg.TAU), p(eg.work), eg.wb)
    torch.cuda.synchronize()
    Dk, Ek = eg.D.clone(), eg.E.clone()
    D2 = torch.zeros(n, **f64); E2 = torch.zeros(n, **f64); Z = torch.zeros((n, n), **f64); info = torch.zeros(1, **i32)
    W = torch.zeros(n, **f64); nevd = torch.zeros(1, **i32)
    T = torch.diag(Dk) + torch.diag(Ek[:n - 1], 1) + torch.diag(Ek[:n - 1], -1)
    def t(f, reps=5):
        f(); torch.cuda.synchronize(); ts = []
        for _ in range(reps):
            torch.cuda.synchronize(); t0 = time.perf_counter(); f(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
        return np.median(ts)
    def stedc():
        D2.copy_(Dk); E2.copy_(Ek)
        assert eg.rs.rocsolver_dstedc(eg.handle, 212, n, p(D2), p(E2), p(Z), n, p(info)) == 0
    def stedcj():
        D2.copy_(Dk); E2.copy_(Ek)
        assert eg.rs.rocsolver_dstedcj(eg.handle, 212, n, p(D2), p(E2), p(Z), n, p(info)) == 0
    eg.rs.rocsolver_dstedcx.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, C.c_int] + [C.c_void_p] * 5 + [C.c_int, C.c_void_p]
    def stedcx():
        D2.copy_(Dk); E2.copy_(Ek)
        rc = eg.rs.rocsolver_dstedcx(eg.handle, 212, 233, n, 0.0, 0.0, 1, nev, p(D2), p(E2), p(nevd), p(W), p(Z), n, p(info)); assert rc == 0, rc
    for name, f, k in (("stedc", stedc, n), ("stedcj", stedcj, n), ("stedcx", stedcx, nev)):
        try:
            ms = t(f)
            Zc = Z.T[:, :k]
            w = (W if name == "stedcx" else D2)[:k]
            print("n=%d %-7s %.2f ms  info=%d  orth %.1e  resid %.1e  werr %.1e" % (n, name, ms, int(info.item()),
                  float((Zc.T @ Zc - torch.eye(k, **f64)).abs().max()), float((T @ Zc - Zc * w).abs().max()), float((w - w_ref[:k]).abs().max())), flush=True)
        except Exception as e:
            print(name, "failed", repr(e), flush=True)
