#!/usr/bin/env python3
"""Eigensolve time with the persistent kernels launched cooperatively (mode 1) or plainly (mode 0): ms per call."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from juliachem_jl_amd import _lib
from juliachem_jl_amd.eigh import DeviceEigh
lib = _lib.load()
dev = torch.device("cuda", 0)
for n in (510, 1250):
    rng = np.random.default_rng(n)
    A = rng.standard_normal((n, n)); A = 0.5 * (A + A.T)
    dA = torch.as_tensor(A, device=dev)
    eg = DeviceEigh(n, dev)
    for mode in (1, 0, 1, 0):
        lib.jcdf_set_persistent_launch_mode(mode)
        for _ in range(3):
            eg(dA)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            eg(dA)
        e1.record()
        torch.cuda.synchronize()
        print("n=%d mode=%d  %.3f ms per eigensolve  ok=%s" % (n, mode, e0.elapsed_time(e1) / 20, eg.check()), flush=True)
