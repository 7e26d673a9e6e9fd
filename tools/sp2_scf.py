#!/usr/bin/env python3
"""SCF step time and trail with density_solver = "eigh" vs "sp2" on the bench's C20H42 shape (synthetic) and on a real
(H2O)n / cc-pVDZ run.  usage: python tools/sp2_scf.py [n_waters]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from juliachem_jl_amd import rhf

nw = int(sys.argv[1]) if len(sys.argv) > 1 else 20
g = json.load(open(os.path.join(ROOT, "tests", "golden", "w50_geometry.json")))
b = json.load(open(os.path.join(ROOT, "tests", "golden", "water_ccpvdz_rifit.json")))
xyz = np.asarray(g["geometry"]).reshape(-1, 3)[:3 * nw] * g["angstrom_to_bohr"]
atoms = [{"symbol": s, "center": list(map(float, r))} for s, r in zip(g["symbols"][:3 * nw], xyz)]
res = {}
for solver in ("eigh", "sp2"):
    t0 = time.perf_counter()
    r = rhf.run(atoms, b["charges"], b["basis"], b["aux_basis"], {"dele": 1e-8, "rmsd": 1e-8, "niter": 60, "density_solver": solver}, output=0)
    res[solver] = r
    print("%-5s (H2O)%d N=%d: E = %.10f  its %d  converged %s  wall %.2f s  %s" % (solver, nw, r["Overlap"].shape[0], r["Energy"], r["Iterations"], r["Converged?"], time.perf_counter() - t0, r["Density Solver"]))
a, c = res["eigh"]["Trail"], res["sp2"]["Trail"]
print("max |dE| along the trail: %.2e   final dE %.2e   max|dD| %.2e  max|dF| %.2e  max|d eps| %.2e" % (
    max(abs(x[1] - y[1]) for x, y in zip(a, c)), abs(res["eigh"]["Energy"] - res["sp2"]["Energy"]),
    np.abs(res["eigh"]["Density"] - res["sp2"]["Density"]).max(), np.abs(res["eigh"]["Fock"] - res["sp2"]["Fock"]).max(),
    np.abs(res["eigh"]["Orbital Energies"] - res["sp2"]["Orbital Energies"]).max()))
import juliachem_jl_amd.engine as eng
for x, y in zip(a, c):
    print("it %2d  E_eigh %.10f  dE(sp2-eigh) %+.2e   drms %.2e / %.2e" % (x[0], x[1], y[1] - x[1], x[3], y[3]))
