#!/usr/bin/env python3
"""Two-stage tridiagonalisation (csrc/jcdf_sbr.hpp) against numpy: residuals of A = Q T Q^T per size, and timings of the
stages (torch events).  usage: python tools/eig2_check.py [n ...]"""
import ctypes as C
import os
import sys
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import juliachem_jl_amd as jc   # noqa: E402

lib = jc._lib.load()
dev = torch.device("cuda", 0)
sizes = [int(a) for a in sys.argv[1:]] or [3, 5, 17, 18, 19, 33, 34, 40, 64, 100, 257, 510, 590]
p = lambda t: C.c_void_p(t.data_ptr())
print("max n", lib.jcdf_sytrd2_max_n())
for n in sizes:
    rng = np.random.default_rng(n)
    A = rng.standard_normal((n, n)); A = 0.5 * (A + A.T)
    f64 = dict(dtype=torch.float64, device=dev)
    dA0 = torch.as_tensor(A, device=dev)
    wb = int(lib.jcdf_sytrd2_workspace_bytes(n))
    work = torch.zeros(wb // 8 + 8, **f64)
    D = torch.zeros(n, **f64); E = torch.zeros(n, **f64); Q = torch.zeros((n, n), **f64)
    st = torch.cuda.current_stream(dev).cuda_stream
    times = []
    for rep in range(3):
        dA = dA0.clone()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        ev[0].record()
        rc = lib.jcdf_sytrd2_device(C.c_void_p(st), n, p(dA), n, p(D), p(E), p(Q), n, p(work), wb)
        ev[1].record()
        rc2 = lib.jcdf_sytrd2_apply_q_device(C.c_void_p(st), n, p(Q), n, p(work), wb)
        ev[2].record()
        torch.cuda.synchronize()
        times.append((ev[0].elapsed_time(ev[1]), ev[1].elapsed_time(ev[2])))
    err = int(work[1:2].view(torch.int32)[0].item())
    Qh = Q.cpu().numpy(); Dh = D.cpu().numpy(); Eh = E.cpu().numpy()[: n - 1]
    T = np.diag(Dh) + np.diag(Eh, 1) + np.diag(Eh, -1)
    e1 = np.abs(Qh.T @ A @ Qh - T).max()
    e2 = np.abs(Qh.T @ Qh - np.eye(n)).max()
    e3 = np.abs(np.linalg.eigvalsh(T) - np.linalg.eigvalsh(A)).max()
    print("n=%4d rc=%d/%d err=%d |Q^T A Q - T| %.2e |Q^T Q - I| %.2e eig %.2e   reduce %.3f ms  apply_q %.3f ms"
          % (n, rc, rc2, err, e1, e2, e3, min(t[0] for t in times), min(t[1] for t in times)), flush=True)
