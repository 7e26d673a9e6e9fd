#!/usr/bin/env python3
"""Probe: trace-correcting purification (SP2) of F' = X F X next to the eigensolve in a real SCF run ((H2O)n / cc-pVDZ):
iterations needed, error of the density against the eigensolver's, time with plain torch ops.
usage: python tools/sp2_probe.py [n_waters]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import juliachem_jl_amd as jc
from juliachem_jl_amd import rhf, engine

nw = int(sys.argv[1]) if len(sys.argv) > 1 else 20
g = json.load(open(os.path.join(ROOT, "tests", "golden", "w50_geometry.json")))
b = json.load(open(os.path.join(ROOT, "tests", "golden", "water_ccpvdz_rifit.json")))
xyz = np.asarray(g["geometry"]).reshape(-1, 3)[:3 * nw] * g["angstrom_to_bohr"]
atoms = [{"symbol": s, "center": list(map(float, r))} for s, r in zip(g["symbols"][:3 * nw], xyz)]


def sp2(Fp, o, kmax=80, tol=1e-11):
    n = Fp.shape[0]
    r = torch.sum(torch.abs(Fp), dim=1) - torch.abs(torch.diagonal(Fp))
    lo = torch.min(torch.diagonal(Fp) - r)
    hi = torch.max(torch.diagonal(Fp) + r)
    X = (hi * torch.eye(n, dtype=Fp.dtype, device=Fp.device) - Fp) / (hi - lo)
    hist = []
    for k in range(kmax):
        X2 = X @ X
        t, t2 = torch.trace(X), torch.trace(X2)
        idem = (t - t2).item()
        hist.append(idem)
        if abs(idem) < tol and k > 4:
            break
        if abs(t2 - o) < abs(2 * t - t2 - o):
            X = X2
        else:
            X = 2 * X - X2
    return X, k + 1, hist, (lo.item(), hi.item())


orig = engine.DeviceSCF._diag
log = []


def patched(self, use_sp2=False):
    orig(self, use_sp2)
    E = 0.5 * torch.sum(self.D * (self.F + self.H))
    Fp = self.X @ self.F @ self.X
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    P, k, hist, (lo, hi) = sp2(Fp, self.n_occ)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    D2 = 2.0 * (self.X @ P @ self.X)
    err = (D2 - self.D).abs().max().item()
    dE = (0.5 * (torch.sum(D2 * self.F) + torch.sum(D2 * self.H)) - E).item()
    eps = self.eps
    log.append((k, err, dE, dt * 1e3, lo, hi, (eps[self.n_occ] - eps[self.n_occ - 1]).item(), eps[0].item(), eps[-1].item()))


engine.DeviceSCF._diag = patched
res = rhf.run(atoms, b["charges"], b["basis"], b["aux_basis"], {"dele": 1e-6, "rmsd": 1e-6, "niter": 40}, output=0)
print("(H2O)%d N=%d E=%.10f its=%d" % (nw, res["Overlap"].shape[0], res["Energy"], res["Iterations"]))
for i, l in enumerate(log):
    print("diag %2d: sp2 its %2d  max|dD| %.2e  dE %.2e  %.2f ms  gersh [%.1f, %.1f]  gap %.3f  spec [%.2f, %.2f]" % ((i,) + l))
