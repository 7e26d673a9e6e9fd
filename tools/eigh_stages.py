#!/usr/bin/env python3
"""DeviceEigh stage by stage (tridiagonalisation, tridiagonal solver, back-transformation): ms per call.
usage: python tools/eigh_stages.py [n ...]"""
import os
import sys
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from juliachem_jl_amd.eigh import DeviceEigh   # noqa: E402

dev = torch.device("cuda", 0)
for n in [int(a) for a in sys.argv[1:]] or [1250, 1536, 1537, 1700, 1915]:
    rng = np.random.default_rng(n)
    A = rng.standard_normal((n, n)); A = 0.5 * (A + A.T)
    dA = torch.as_tensor(A, device=dev)
    eg = DeviceEigh(n, dev)
    for _ in range(3):
        eg(dA)
    torch.cuda.synchronize()
    eg.timing = True
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    reps = 8
    for _ in range(reps):
        w, U = eg(dA)
    e1.record()
    torch.cuda.synchronize()
    tot = e0.elapsed_time(e1) / reps
    a, b = eg.stage_ms()
    ok = eg.check()
    res = float((dA @ U - U * w[None, :]).abs().max())
    orth = float((U.T @ U - torch.eye(n, device=dev, dtype=U.dtype)).abs().max())
    print("n=%4d  total %.3f ms = sytrd %.3f + stedc %.3f + back-transformation %.3f   with_q=%s ok=%s residual %.1e orth %.1e"
          % (n, tot, a, b, tot - a - b, eg.with_q, ok, res, orth), flush=True)
