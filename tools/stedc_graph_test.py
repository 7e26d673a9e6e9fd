#!/usr/bin/env python3
"""Does replaying the divide & conquer's launch sequence as a captured graph shorten it?  (29 dependent launches of 5-35 us at n = 510)"""
import os, sys, ctypes as C
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from juliachem_jl_amd import _lib
lib = _lib.load()
dev = torch.device("cuda", 0)
for n in (510, 1250):
    rng = np.random.default_rng(n)
    d = torch.as_tensor(rng.standard_normal(n), device=dev); e = torch.as_tensor(rng.standard_normal(n), device=dev)
    D = d.clone(); E = e.clone()
    npad = (n + 31) // 32 * 32
    Z = torch.zeros((npad, npad), dtype=torch.float64, device=dev)
    wb = int(lib.jcdf_stedc_workspace_bytes(n)); work = torch.zeros(wb // 8 + 8, dtype=torch.float64, device=dev)
    p = lambda t: C.c_void_p(t.data_ptr())
    def call(stream):
        D.copy_(d)
        rc = lib.jcdf_stedc_device(C.c_void_p(stream.cuda_stream), n, p(D), p(E), p(Z), npad, p(work), wb)
        assert rc == 0
    s = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(s):
        for _ in range(3):
            call(s)
    torch.cuda.synchronize()
    wref = D.clone()
    def timeit(fn, reps=30):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(s):
            e0.record(s)
            for _ in range(reps):
                fn()
            e1.record(s)
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps
    t_plain = timeit(lambda: call(s))
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        call(s)
    torch.cuda.synchronize()
    t_graph = timeit(lambda: g.replay())
    print("n=%d  plain %.3f ms  graph %.3f ms  same eigenvalues: %s" % (n, t_plain, t_graph, bool(torch.equal(D, wref))), flush=True)
