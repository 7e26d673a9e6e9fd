import sys, time; sys.path.insert(0, "/root/repo")
import numpy as np, torch
import juliachem_jl_amd
from juliachem_jl_amd.eigh import DeviceEigh
for n in (1700, 2000, 2500):
    rng = np.random.default_rng(n); A = rng.standard_normal((n, n)); A = A + A.T
    Ag = torch.as_tensor(A, device="cuda")
    eg = DeviceEigh(n, torch.device("cuda", 0))
    w, U = eg(Ag); torch.cuda.synchronize()
    t0 = time.perf_counter(); w, U = eg(Ag); torch.cuda.synchronize(); ms = (time.perf_counter() - t0) * 1e3
    ok = eg.check()
    res = float((Ag @ U - U * w).abs().max()); orth = float((U.T @ U - torch.eye(n, device="cuda", dtype=torch.float64)).abs().max())
    print("n=%d ok=%s with_q=%s own_stedc=%s fallbacks=%d reason=%s  %.1f ms  resid %.1e orth %.1e" % (n, eg.ok, getattr(eg, "with_q", None), getattr(eg, "own_stedc", None), eg.fallbacks, getattr(eg, "reason", ""), ms, res, orth))
