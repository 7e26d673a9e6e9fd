#!/usr/bin/env python3
"""Reduction only, with and without Q (for rocprofv3 --kernel-trace --stats): splits the one-workgroup tail into its two phases.
usage: python tools/sytrd_tail_time.py [n]"""
import ctypes as C
import os
import sys
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from juliachem_jl_amd import _lib   # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 510
lib = _lib.load()
dev = torch.device("cuda", 0)
rng = np.random.default_rng(n)
A0 = rng.standard_normal((n, n)); A0 = torch.as_tensor(0.5 * (A0 + A0.T), device=dev)
npad = (n + 31) // 32 * 32
wb = int(lib.jcdf_sytrd_workspace_bytes(n))
work = torch.zeros(wb // 8 + 1, dtype=torch.float64, device=dev)
D, E, TAU = (torch.zeros(n, dtype=torch.float64, device=dev) for _ in range(3))
Q = torch.zeros(npad, npad, dtype=torch.float64, device=dev)
p = lambda t: C.c_void_p(t.data_ptr())
st = torch.cuda.current_stream().cuda_stream
for withq in (0, 1):
    for _ in range(10):
        A = A0.clone()
        if withq:
            rc = lib.jcdf_sytrd_q_device(C.c_void_p(st), n, p(A), n, p(D), p(E), p(TAU), p(Q), npad, p(work), wb)
        else:
            rc = lib.jcdf_sytrd_device(C.c_void_p(st), n, p(A), n, p(D), p(E), p(TAU), p(work), wb)
        assert rc == 0, rc
    torch.cuda.synchronize()
print("done")
