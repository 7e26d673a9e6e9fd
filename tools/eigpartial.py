"""Probe: rocSOLVER stebz + stein (partial spectrum) vs stedc on the tridiagonal matrix of sytrd."""
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
    w_ref, U_ref = eg(Ag); torch.cuda.synchronize()
    w_ref = w_ref.clone()
    st = torch.cuda.current_stream().cuda_stream
    p = lambda x: C.c_void_p(x.data_ptr())
    def sytrd():
        eg.A.copy_(Ag); eg.lib.jcdf_sytrd_device(C.c_void_p(st), n, p(eg.A), n, p(eg.D), p(eg.E), p(eg.TAU), p(eg.work), eg.wb)
    i32 = dict(dtype=torch.int32, device=dev); f64 = dict(dtype=torch.float64, device=dev)
    nevd = torch.zeros(1, **i32); nsplit = torch.zeros(1, **i32); W = torch.zeros(n, **f64)
    iblock = torch.zeros(n, **i32); isplit = torch.zeros(n, **i32); info = torch.zeros(1, **i32)
    Z = torch.zeros((nev, n), **f64); ifail = torch.zeros(n, **i32)
    D2 = torch.zeros(n, **f64); E2 = torch.zeros(n, **f64)
    eg.rs.rocsolver_dstebz.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, C.c_int, C.c_double] + [C.c_void_p] * 8
    def t(f, reps=5):
        f(); torch.cuda.synchronize(); ts = []
        for _ in range(reps):
            torch.cuda.synchronize(); t0 = time.perf_counter(); f(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
        return np.median(ts)
    sytrd(); torch.cuda.synchronize()
    Dk, Ek = eg.D.clone(), eg.E.clone()
    def stebz():
        D2.copy_(Dk); E2.copy_(Ek)
        rc = eg.rs.rocsolver_dstebz(eg.handle, 233, 241, n, 0.0, 0.0, 1, nev, 0.0, p(D2), p(E2), p(nevd), p(nsplit), p(W), p(iblock), p(isplit), p(info))
        assert rc == 0, rc
    def stein():
        rc = eg.rs.rocsolver_dstein(eg.handle, n, p(D2), p(E2), p(nevd), p(W), p(iblock), p(isplit), p(Z), n, p(ifail), p(info))
        assert rc == 0, rc
    def stedc():
        D2.copy_(Dk); E2.copy_(Ek)
        rc = eg.rs.rocsolver_dstedc(eg.handle, 212, n, p(D2), p(E2), p(eg.Cm), n, p(eg.info)); assert rc == 0
    def ormtr_full():
        eg.rs.rocsolver_dormtr(eg.handle, 141, 122, 111, n, n, p(eg.A), n, p(eg.TAU), p(eg.Cm), n)
    def ormtr_part():
        eg.rs.rocsolver_dormtr(eg.handle, 141, 122, 111, n, nev, p(eg.A), n, p(eg.TAU), p(Z), n)
    t_bz = t(stebz); t_in = t(stein)
    print("n=%d nev=%d: sytrd %.2f | stebz %.2f stein %.2f ormtr(nev) %.2f | stedc %.2f ormtr(n) %.2f  [nev=%d nsplit=%d info=%d werr=%.2e]"
          % (n, nev, t(sytrd), t_bz, t_in, t(ormtr_part), t(stedc), t(ormtr_full), int(nevd.item()), int(nsplit.item()), int(info.item()),
             float((W[:nev] - w_ref[:nev]).abs().max())), flush=True)
    # orthogonality / residual of stein vectors on the tridiagonal matrix
    stebz(); stein(); torch.cuda.synchronize()
    Zc = Z.T   # (n, nev) column-major view
    T = torch.diag(Dk) + torch.diag(Ek[:n - 1], 1) + torch.diag(Ek[:n - 1], -1)
    print("   stein: orth %.2e  resid %.2e" % (float((Zc.T @ Zc - torch.eye(nev, **f64)).abs().max()), float((T @ Zc - Zc * W[:nev]).abs().max())))
