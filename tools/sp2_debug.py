#!/usr/bin/env python3
"""Per-step record of the SP2 path of DeviceSCF on a synthetic shape (diagnostic): usage sp2_debug.py N,Q,o [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from juliachem_jl_amd.engine import DeviceFockBuilder, DeviceSCF

N, Q, o = (int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "510,200,81").split(","))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 16
rng = np.random.default_rng(1); dev = torch.device("cuda", 0)
fb = DeviceFockBuilder(N, Q, o, [1] * Q, device=0)
fb.h.set_metric_inverse(np.eye(Q))
Hs = rng.standard_normal((N, N)); H = 0.5 * (Hs + Hs.T)
fb.set_core_hamiltonian(H)
g = torch.Generator(device=dev); g.manual_seed(7)
for s0 in range(0, Q, 256):
    s1 = min(Q, s0 + 256)
    A = torch.randn((N, N, s1 - s0), dtype=torch.float64, device=dev, generator=g) * 0.1
    fb.push_three_center_device(s0, s1, (0.5 * (A + A.transpose(0, 1))).contiguous().reshape(-1))
scf = DeviceSCF(fb, H, np.eye(N), 0.0, density_solver="sp2")
for it in range(steps):
    skip = scf.sp2_skip
    E, dE, drms = scf.step()
    rec = scf.tail_out.cpu().tolist()
    li = scf.lowdin.info.cpu().tolist()
    Gd = scf.Cpt[:o, :N] @ scf.Cpt[:o, :N].T
    print("it %2d sp2 %s E %.8f drms %.3e rec %s | lowdin info %s steps_next %d | ||CptCpt^T - I|| %.2e pad %.2e | %s retries %d" % (
        it, not skip, E, drms, ["%.3g" % x for x in rec], ["%.3g" % x for x in li], scf.lowdin.steps,
        (Gd - torch.eye(o, device=dev, dtype=torch.float64)).abs().max().item(),
        max(scf.Cpt[o:].abs().max().item() if scf.op > o else 0.0, scf.Cop[o:].abs().max().item() if scf.op > o else 0.0),
        scf.sp2_reasons, scf.sp2_basis_retries), flush=True)
