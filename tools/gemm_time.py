#!/usr/bin/env python3
"""us per jcdf_gemm_tn_device / jcdf_gemm_nt_device call in a dependent chain (diagnostic build: JCDF_GEMM_SK = 0 / 2 / 4).
usage: gemm_time.py [n ...]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from juliachem_jl_amd import _lib
lib = _lib.load()
dev = torch.device("cuda", 0)
st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
p = lambda t: C.c_void_p(t.data_ptr())
for n in [int(a) for a in sys.argv[1:]] or [512, 704, 960]:
    A = torch.randn((n, n), dtype=torch.float64, device=dev) / n ** 0.5
    B = torch.randn((n, n), dtype=torch.float64, device=dev) / n ** 0.5
    Cm = torch.empty((n, n), dtype=torch.float64, device=dev)
    ref = A.T @ B
    assert lib.jcdf_gemm_tn_device(st, n, n, n, 1.0, p(A), n, p(B), n, p(Cm), n) == 0
    err = (Cm - ref).abs().max().item()
    for name, call in (("tn", lambda X, Y, Z: lib.jcdf_gemm_tn_device(st, n, n, n, 1.0, p(X), n, p(Y), n, p(Z), n)),
                       ("nt", lambda X, Y, Z: lib.jcdf_gemm_nt_device(st, n, n, n, p(X), n, p(Y), n, p(Z), n))):
        for _ in range(5):
            call(A, B, Cm)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        X, Z = A, Cm
        for _ in range(100):                     # dependent chain: the output of one call is an operand of the next
            call(X, B, Z)
            X, Z = Z, X
        e1.record()
        torch.cuda.synchronize()
        print("n=%4d %s  %.2f us per call   (SK=%s, tn err %.1e)" % (n, name, e0.elapsed_time(e1) * 10.0, os.environ.get("JCDF_GEMM_SK", "lib"), err), flush=True)
