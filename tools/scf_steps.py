#!/usr/bin/env python3
"""A few device SCF iterations on a synthetic shape with a scattered kept-pair map — the target of a rocprofv3 kernel trace that
counts the launches of ONE step by family (library / torch / vendor).  usage: scf_steps.py [config] [kept] [steps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import juliachem_jl_amd as jc
from juliachem_jl_amd import synthetic
from juliachem_jl_amd.engine import DeviceFockBuilder, DeviceSCF

cfg = sys.argv[1] if len(sys.argv) > 1 else "gly10_vtz"
kept = float(sys.argv[2]) if len(sys.argv) > 2 else 0.30
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
N, Q, o = synthetic.CONFIGS[cfg]
rng = np.random.default_rng(synthetic.SEED + 50)
dev = torch.device("cuda", 0)
sd = jc.get_screening_metadata(synthetic.cluster_mask(N, kept, rng))
p, q = jc.packed_pq_lists(sd)
P = len(p)
Hs = rng.standard_normal((N, N)); H = 0.5 * (Hs + Hs.T)
fb = DeviceFockBuilder(N, Q, o, synthetic.aux_shells(Q, rng), device=0, pq=(p, q))
fb.set_core_hamiltonian(H)
g = torch.Generator(device=dev); g.manual_seed(1)
g1 = torch.randn((Q, N), dtype=torch.float64, device=dev, generator=g) * 0.05
g2 = torch.randn((Q, N), dtype=torch.float64, device=dev, generator=g) * 0.05
pd, qd = torch.as_tensor(p, device=dev), torch.as_tensor(q, device=dev)
for c0 in range(0, P, 8192):
    c1 = min(P, c0 + 8192)
    blk = (g1[:, pd[c0:c1]] * g2[:, qd[c0:c1]] + g1[:, qd[c0:c1]] * g2[:, pd[c0:c1]]).t().contiguous()
    torch.cuda.synchronize()
    fb.h.set_B_columns_device(c0, c1, blk.data_ptr())
del g1, g2, blk
scf = DeviceSCF(fb, H, np.eye(N), 0.0)
torch.cuda.synchronize()
for _ in range(steps):
    E, dE, drms = scf.step()
torch.cuda.synchronize()
print("config %s N=%d Q=%d o=%d kept %.3f: %d steps, E = %.10f, eigensolver %s" % (cfg, N, Q, o, P / N ** 2, steps, E, scf.solver_report()))
fb.close()
