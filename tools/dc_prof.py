"""rocprofv3 target: the device D&C solver on the tridiagonal form of a random symmetric N x N matrix."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, scipy.linalg as sla
import juliachem_jl_amd
from juliachem_jl_amd import _lib
lib = _lib.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 510
rng = np.random.default_rng(0); A = rng.standard_normal((n, n)); A = A + A.T
Hh = sla.hessenberg(A)
f64 = dict(dtype=torch.float64, device="cuda")
d0 = torch.as_tensor(np.diag(Hh).copy(), **f64); E = torch.as_tensor(np.append(np.diag(Hh, 1), 0.0), **f64)
wb = int(lib.jcdf_stedc_workspace_bytes(n)); work = torch.empty(wb // 8 + 8, **f64); Z = torch.empty((n, n), **f64)
p = lambda t: C.c_void_p(t.data_ptr()); st = torch.cuda.current_stream().cuda_stream
for _ in range(20):
    D = d0.clone()
    lib.jcdf_stedc_device(C.c_void_p(st), n, p(D), p(E), p(Z), n, p(work), wb)
torch.cuda.synchronize()
