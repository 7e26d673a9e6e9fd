#!/usr/bin/env python3
"""Runs only the HIP Fock build (no SCF wrapper) on a synthetic shape — the target
of `rocprofv3 --kernel-trace --stats` / `--pmc` runs.
usage: prof_fock.py [config] [n_builds]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import juliachem_jl_amd as jc
from juliachem_jl_amd import synthetic
from juliachem_jl_amd.engine import DeviceFockBuilder

cfg = sys.argv[1] if len(sys.argv) > 1 else "C20H42"
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 5
kept = float(sys.argv[3]) if len(sys.argv) > 3 else None      # Schwarz-kept pair fraction (band mask); None = dense map
N, Q, o = (tuple(int(x) for x in cfg.split(",")) if "," in cfg else synthetic.CONFIGS[cfg])      # name or N,Q,o
rng = np.random.default_rng(1)
dev = torch.device("cuda", 0)
pq = (None, None)
if kept is not None:
    sd = jc.get_screening_metadata(synthetic.band_mask(N, kept, rng))
    pq = jc.packed_pq_lists(sd)
    print("screened map: kept pair fraction %.3f (P = %d of %d)" % (sd.screened_indices_count / N ** 2, sd.screened_indices_count, N * N))
    sel = torch.as_tensor(pq[0] * N + pq[1], device=dev)
fb = DeviceFockBuilder(N, Q, o, [1] * Q, device=0, pq=pq)
fb.h.set_metric_inverse(np.eye(Q))            # B == T: setup cost is irrelevant here
Hs = rng.standard_normal((N, N))
fb.set_core_hamiltonian(0.5 * (Hs + Hs.T))
g = torch.Generator(device=dev); g.manual_seed(7)
step = 256
for s0 in range(0, Q, step):
    s1 = min(Q, s0 + step)
    A = torch.randn((N, N, s1 - s0), dtype=torch.float64, device=dev, generator=g) * 0.1
    T = (0.5 * (A + A.transpose(0, 1))).contiguous()
    T = T.reshape(N * N, s1 - s0)[sel].contiguous().reshape(-1) if kept is not None else T.reshape(-1)
    fb.push_three_center_device(s0, s1, T)
torch.cuda.synchronize()
if os.environ.get("JCDF_PROF_NO_OVERLAP") == "1":        # PMC passes: every kernel alone on the device, so its counters are its own
    fb.h.set_overlap(False)
C, _ = np.linalg.qr(rng.standard_normal((N, N)))
Ct = torch.as_tensor(np.ascontiguousarray(C[:, :o].T), device=dev)
for _ in range(nb):
    fb.build(Ct)
torch.cuda.synchronize()
t = fb.h.synchronize()
print("fock_time %.3f ms  W %.3f  J %.3f  K %.3f  assemble %.3f" % (t.fock_time * 1e3, t.W_time * 1e3, t.J_time * 1e3,
                                                                       t.K_time * 1e3, t.copy_J_time * 1e3))
for ks in fb.h.kernel_stats():
    if ks["seconds"] > 0:
        print("%-18s %.3f ms  exec %.1f TF  alg %.1f TF  alg %.0f GB/s" % (ks["name"], ks["seconds"] * 1e3,
              ks["flops"] / ks["seconds"] / 1e12, ks["alg_flops"] / ks["seconds"] / 1e12, ks["alg_bytes"] / ks["seconds"] / 1e9))
fb.close()
