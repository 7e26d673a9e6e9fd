#!/usr/bin/env python3
"""Diagnostic build: Fock-build kernel times behind an eigensolve whose tridiagonalisation runs on G1 workgroups (JCDF_SYTRD_G1)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from juliachem_jl_amd import synthetic
from juliachem_jl_amd.engine import DeviceFockBuilder
from juliachem_jl_amd.eigh import DeviceEigh
N, Q, o = synthetic.CONFIGS["C20H42"]
rng = np.random.default_rng(1); dev = torch.device("cuda", 0)
fb = DeviceFockBuilder(N, Q, o, [1] * Q, device=0)
g = torch.Generator(device=dev); g.manual_seed(7)
for c0 in range(0, N * N, 16384):
    c1 = min(N * N, c0 + 16384)
    blk = torch.randn((c1 - c0, len(fb.rows)), dtype=torch.float64, device=dev, generator=g) * 0.05
    fb.h.set_B_columns_device(c0, c1, blk.data_ptr())
fb.set_core_hamiltonian(np.eye(N))
C, _ = np.linalg.qr(rng.standard_normal((N, N)))
Ct = torch.as_tensor(np.ascontiguousarray(C[:, :o].T), device=dev)
eig = DeviceEigh(N, dev)
S = torch.randn((N, N), dtype=torch.float64, device=dev); S = S + S.T
for _ in range(3):
    eig(S); fb.build(Ct)
torch.cuda.synchronize()
fb.h.kernel_stats_total(reset=True)
evs = []
for _ in range(15):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); eig(S); e1.record(); evs.append((e0, e1))
    fb.build(Ct)
torch.cuda.synchronize()
recs, n, fock = fb.h.kernel_stats_total(reset=True)
k = {r["name"]: r["seconds"] / max(n, 1) * 1e3 for r in recs}
print("G1=%s  W %.3f K %.3f J %.3f fock %.3f | eigensolve %.3f ms" % (os.environ.get("JCDF_SYTRD_G1", "default"), k["k_exchange_W"], k["k_exchange_K"],
      k["k_coulomb_J"], fock / max(n, 1) * 1e3, sum(a.elapsed_time(b) for a, b in evs) / len(evs)), flush=True)
