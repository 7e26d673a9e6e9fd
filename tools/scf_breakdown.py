#!/usr/bin/env python3
"""Per-segment device time of DeviceSCF.step() on the C20H42 shape (diagnostic)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import juliachem_jl_amd as jc
from juliachem_jl_amd import synthetic
from juliachem_jl_amd.engine import DeviceFockBuilder, DeviceSCF

N, Q, o = synthetic.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "C20H42"]
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
scf = DeviceSCF(fb, H, np.eye(N), 0.0)
scf.profile = True
for _ in range(3): scf.step()
scf.seg = {}
torch.cuda.synchronize(); t0 = time.perf_counter()
n = 10
for _ in range(n): scf.step()
torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / n * 1e3
print("wall per step %.3f ms" % wall)
for k, v in scf.seg.items():
    print("  %-12s %.3f ms" % (k, sum(v) / len(v)))
