import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import juliachem_jl_amd
from juliachem_jl_amd.eigh import DeviceEigh
dev = torch.device("cuda", 0)
for n in (240, 510, 1250):
    rng = np.random.default_rng(0); A = rng.standard_normal((n, n)); A = A + A.T
    Ag = torch.as_tensor(A, device=dev)
    def t(f, reps=8):
        f(); torch.cuda.synchronize(); ts = []
        for _ in range(reps):
            t0 = time.perf_counter(); f(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
        return np.median(ts)
    base = t(lambda: torch.linalg.eigh(Ag))
    out = ["n=%d torch.eigh %.2f ms" % (n, base)]
    for G in ("", "16", "32", "64", "128", "256"):
        if G: os.environ["JCDF_SYTRD_G"] = G
        else: os.environ.pop("JCDF_SYTRD_G", None)
        eg = DeviceEigh(n, dev)
        ms = t(lambda: eg(Ag))
        # time the sytrd kernel alone
        st = torch.cuda.current_stream().cuda_stream
        import ctypes as C
        p = lambda x: C.c_void_p(x.data_ptr())
        def sy():
            eg.A.copy_(Ag); eg.lib.jcdf_sytrd_device(C.c_void_p(st), n, p(eg.A), n, p(eg.D), p(eg.E), p(eg.TAU), p(eg.work), eg.wb)
        ms_s = t(sy)
        ok = eg.check()
        out.append("G=%s: eigh %.2f (sytrd %.2f) ok=%s" % (G or "auto", ms, ms_s, ok))
    print(" | ".join(out))
