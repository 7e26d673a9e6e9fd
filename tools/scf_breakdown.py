#!/usr/bin/env python3
"""Per-segment device time of DeviceSCF.step() on the C20H42 shape (diagnostic)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import juliachem_jl_amd as jc
from juliachem_jl_amd import synthetic
from juliachem_jl_amd.engine import DeviceFockBuilder, DeviceSCF

cfg = sys.argv[1] if len(sys.argv) > 1 else "C20H42"                    # name or N,Q,o ; argv[2]: eigh | sp2
N, Q, o = (tuple(int(x) for x in cfg.split(",")) if "," in cfg else synthetic.CONFIGS[cfg])
solver = sys.argv[2] if len(sys.argv) > 2 else "eigh"
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
scf = DeviceSCF(fb, H, np.eye(N), 0.0, density_solver=solver)
for _ in range(8 if solver == "sp2" else 3): scf.step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): scf.step()
torch.cuda.synchronize(); print("%s N=%d o=%d: %.3f ms per step unprofiled (sp2 steps %d, fallbacks %d %s retries %d; squarings %s, NS steps %s)" % (solver, N, o, (time.perf_counter() - t0) / 10 * 1e3, scf.sp2_steps, scf.sp2_fallbacks, scf.sp2_reasons, scf.sp2_basis_retries, scf.sp2.iterations if scf.sp2 else None, scf.lowdin.steps if scf.sp2 else None))
scf.profile = True
for _ in range(2): scf.step()
scf.seg = {}
torch.cuda.synchronize(); t0 = time.perf_counter()
n = 10
for _ in range(n): scf.step()
torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / n * 1e3
print("wall per step %.3f ms" % wall)
for k, v in scf.seg.items():
    print("  %-12s %.3f ms" % (k, sum(v) / len(v)))
