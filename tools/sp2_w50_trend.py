#!/usr/bin/env python3
"""Per-iteration record of the spectral-projection density solver on the (H2O)50-shaped synthetic SCF (13 %-kept map): squarings,
accelerated or not, ||F' - F_ref||_F, energy change — how fast the synthetic problem reaches the regime the bench line quotes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
args = bench.parse_args(["--no-cpu-baseline"])
args.steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
import juliachem_jl_amd as jc
from juliachem_jl_amd import synthetic
from juliachem_jl_amd.engine import DeviceFockBuilder, DeviceSCF
dev = torch.device("cuda", 0)
N, Q, o = synthetic.CONFIGS["w50"]
rng = np.random.default_rng(synthetic.SEED + 50)
KEPT = float(sys.argv[2]) if len(sys.argv) > 2 else 0.13          # 0: the unscreened map
if KEPT > 0:
    sd = jc.get_screening_metadata(synthetic.cluster_mask(N, KEPT, rng))
    p, q = jc.packed_pq_lists(sd)
    pq = (p, q)
else:
    p = np.repeat(np.arange(N, dtype=np.int64), N); q = np.tile(np.arange(N, dtype=np.int64), N)
    pq = (None, None)
P = len(p)
shells = synthetic.aux_shells(Q, rng)
Hs = rng.standard_normal((N, N)); H = 0.5 * (Hs + Hs.T)
fb = DeviceFockBuilder(N, Q, o, shells, device=0, pq=pq)
fb.set_core_hamiltonian(H)
g = torch.Generator(device=dev); g.manual_seed(synthetic.SEED + 1000)
g1 = torch.randn((Q, N), dtype=torch.float64, device=dev, generator=g) * 0.05
g2 = torch.randn((Q, N), dtype=torch.float64, device=dev, generator=g) * 0.05
pd, qd = torch.as_tensor(p, device=dev), torch.as_tensor(q, device=dev)
for c0 in range(0, P, 8192):
    c1 = min(P, c0 + 8192)
    blk = (g1[:, pd[c0:c1]] * g2[:, qd[c0:c1]] + g1[:, qd[c0:c1]] * g2[:, pd[c0:c1]]).t().contiguous()
    torch.cuda.synchronize()
    fb.h.set_B_columns_device(c0, c1, blk.data_ptr())
del g1, g2, blk
scf = DeviceSCF(fb, H, np.eye(N), 0.0, density_solver="sp2")
for it in range(args.steps):
    E, dE, drms = scf.step()
    si = scf.sp2.info.cpu().tolist()
    print("it %2d  E %.8f dE %9.2e drms %9.2e | sp2 steps %d fallbacks %d | squarings %2d accel %d delta %.3e  homo-lumo(ref) %.3e" % (
        it + 1, E, dE, drms, scf.sp2_steps, scf.sp2_fallbacks, int(si[0]), int(si[6]), si[7],
        float(scf.ref_eigs[2] - scf.ref_eigs[1])), flush=True)
fb.close()
