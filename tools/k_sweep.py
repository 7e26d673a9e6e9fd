#!/usr/bin/env python3
"""Split-K sweep of the exchange-K kernel (k_exchange_K64) on a synthetic shape: K time alone (J after K) and beside J.
usage: k_sweep.py [config|N,Q,o] [n_builds] [m1,m2,...]    (m = "k_slices_per_xcd" of jcdf_set_tuning; 0 = library rule)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from juliachem_jl_amd import synthetic
from juliachem_jl_amd.engine import DeviceFockBuilder

cfg = sys.argv[1] if len(sys.argv) > 1 else "C20H42"
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 8
ms = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [0, 5, 6, 7, 8, 14]
N, Q, o = (tuple(int(x) for x in cfg.split(",")) if "," in cfg else synthetic.CONFIGS[cfg])
rng = np.random.default_rng(1)
dev = torch.device("cuda", 0)
C, _ = np.linalg.qr(rng.standard_normal((N, N)))
Ct = torch.as_tensor(np.ascontiguousarray(C[:, :o].T), device=dev)
for m in ms:
    fb = DeviceFockBuilder(N, Q, o, [1] * Q, device=0, tuning={"k_slices_per_xcd": m})
    g = torch.Generator(device=dev); g.manual_seed(7)
    for c0 in range(0, N * N, 16384):                      # B itself: random packed columns (its values do not matter for timing)
        c1 = min(N * N, c0 + 16384)
        blk = torch.randn((c1 - c0, len(fb.rows)), dtype=torch.float64, device=dev, generator=g) * 0.05
        fb.h.set_B_columns_device(c0, c1, blk.data_ptr())
    out = []
    for overlap in (False, True):
        fb.h.set_overlap(overlap)
        fb.h.kernel_stats_total(reset=True)
        for _ in range(nb):
            fb.build(Ct)
        torch.cuda.synchronize()
        recs, n, fock = fb.h.kernel_stats_total(reset=True)
        k = {r["name"]: r["seconds"] / max(n, 1) * 1e3 for r in recs}
        out.append("%s: K %.3f J %.3f W %.3f fock %.3f ms" % ("beside J" if overlap else "alone", k["k_exchange_K"], k["k_coulomb_J"],
                                                              k["k_exchange_W"], fock / max(n, 1) * 1e3))
    print("m = %2d  %s" % (m, "   ".join(out)), flush=True)
    fb.close()
