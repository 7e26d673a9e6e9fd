#!/usr/bin/env python3
"""jcdf_sp2_device against the eigensolver's projector, and timing.  usage: python tools/sp2_test.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from juliachem_jl_amd.eigh import DeviceSP2

dev = torch.device("cuda", 0)
torch.manual_seed(1)
cases = [(int(a), int(b), float(c)) for a, b, c in (x.split(",") for x in sys.argv[1:])] if len(sys.argv) > 1 else None
for n, o, gap in cases or [(64, 10, 0.5), (100, 37, 0.1), (130, 5, 0.3), (510, 81, 0.5), (510, 81, 0.01), (500, 100, 0.5), (1250, 250, 0.5), (1900, 300, 0.3)]:
    Q, _ = torch.linalg.qr(torch.randn(n, n, dtype=torch.float64, device=dev))
    ev = torch.cat([torch.linspace(-20.0, -0.5, o, dtype=torch.float64, device=dev),
                    torch.linspace(-0.5 + gap, 6.0, n - o, dtype=torch.float64, device=dev)])
    F = (Q * ev) @ Q.T
    F = 0.5 * (F + F.T)
    Pref = Q[:, :o] @ Q[:, :o].T
    sp = DeviceSP2(n, o, dev)
    P = sp(F, 120).clone()
    info = sp.info.cpu().tolist()
    err = (P - Pref).abs().max().item()
    its = int(info[0])
    # timing with exactly the needed count + 4
    for _ in range(3):
        sp(F, its + 4)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    R = 20
    for _ in range(R):
        sp(F, its + 4)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / R * 1e3
    print("n %4d o %3d gap %.2f: squarings %3d finished %d  tr %.12f  idem %.1e  max|P-Pref| %.2e  sym %.1e  %.3f ms (%.1f us / squaring)" % (
        n, o, gap, its, int(info[1]), info[2], info[3], err, (P - P.T).abs().max().item(), dt, dt * 1e3 / (its + 4)))
