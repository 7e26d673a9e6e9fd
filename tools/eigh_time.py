#!/usr/bin/env python3
"""DeviceEigh end to end (reduction + divide & conquer + back-transformation), one-stage vs two-stage: ms per call.
usage: python tools/eigh_time.py [n ...]"""
import os
import sys
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from juliachem_jl_amd.eigh import DeviceEigh   # noqa: E402

dev = torch.device("cuda", 0)
for n in [int(a) for a in sys.argv[1:]] or [257, 510, 590]:
    rng = np.random.default_rng(n)
    A = rng.standard_normal((n, n)); A = 0.5 * (A + A.T)
    dA = torch.as_tensor(A, device=dev)
    for mode in ("0", "1"):
        os.environ["JCDF_EIGH_TWO_STAGE"] = mode
        eg = DeviceEigh(n, dev)
        for _ in range(5):
            eg(dA)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            w, U = eg(dA)
        e1.record()
        torch.cuda.synchronize()
        ok = eg.check()
        res = float((dA @ U - U * w[None, :]).abs().max())
        orth = float((U.T @ U - torch.eye(n, device=dev, dtype=U.dtype)).abs().max())
        werr = float((w - torch.linalg.eigvalsh(dA)).abs().max())
        print("n=%4d two_stage=%s  %.3f ms per eigh  ok=%s residual %.1e orthogonality %.1e eigenvalues %.1e"
              % (n, mode, e0.elapsed_time(e1) / 20, ok, res, orth, werr), flush=True)
