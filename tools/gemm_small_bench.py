#!/usr/bin/env python3
"""ms per call of jcdf_gemm_tn_device / jcdf_gemm_nt_device on small square products (back to back, device events).
usage: gemm_small_bench.py [n ...]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from juliachem_jl_amd import _lib
lib = _lib.load()
dev = torch.device("cuda", 0)
for n in [int(a) for a in sys.argv[1:]] or [512, 1280]:
    A = torch.randn((n, n), dtype=torch.float64, device=dev); B = torch.randn((n, n), dtype=torch.float64, device=dev); Cm = torch.empty_like(A)
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    p = lambda t: C.c_void_p(t.data_ptr())
    for name, call in (("tn", lambda: lib.jcdf_gemm_tn_device(st, n, n, n, 1.0, p(A), n, p(B), n, p(Cm), n)),
                       ("nt", lambda: lib.jcdf_gemm_nt_device(st, n, n, n, p(A), n, p(B), n, p(Cm), n))):
        for _ in range(10):
            assert call() == 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 200
        e0.record()
        for _ in range(reps):
            call()
        e1.record(); torch.cuda.synchronize()
        ref = (A.T @ B) if name == "tn" else (A @ B.T)
        err = float((Cm - ref).abs().max() / ref.abs().max())
        us = e0.elapsed_time(e1) / reps * 1e3
        print("n=%5d %s: %7.2f us per call, %5.1f TF, rel err %.1e" % (n, name, us, 2.0 * n ** 3 / us / 1e6, err), flush=True)
