#!/usr/bin/env python3
"""The benchmark molecule itself, with real integrals: n-eicosane C20H42 (idealised all-trans geometry: C-C 1.53 A,
C-C-C 112 deg, C-H 1.09 A, H-C-H 107 deg) / 6-31G(2df,p) / cc-pVTZ-JKFIT — the carbon and hydrogen tables of the
reference's S22 log (tests/golden/s22_10_benzene_methane_631g2dfp_jkfit.json; the reference's cc-pVDZ tables for carbon
are not in the snapshot) — DF-RHF on one MI355X through rhf.run: 956 AO, 3390 auxiliary functions, 81 occupied
orbitals, Schwarz-screened packed layout.
usage: python tools/run_c20h42.py [n_carbons] [density_solver]"""
import json, math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from juliachem_jl_amd import rhf

ANG = 1.0 / 0.52917724924


def n_alkane(nc):
    cc, ch, ccc, hch = 1.53, 1.09, math.radians(112.0), math.radians(107.0)
    dx, dz = cc * math.sin(ccc / 2), cc * math.cos(ccc / 2)
    atoms = []
    C = [np.array([i * dx, 0.0, 0.5 * dz * (1 if i % 2 == 0 else -1)]) for i in range(nc)]
    for i, c in enumerate(C):
        atoms.append(("C", c))
    for i, c in enumerate(C):
        up = 1.0 if i % 2 == 0 else -1.0                        # side of the zigzag this carbon sticks out to
        hy, hz = ch * math.sin(hch / 2), ch * math.cos(hch / 2)
        atoms.append(("H", c + np.array([0.0, hy, up * hz])))
        atoms.append(("H", c + np.array([0.0, -hy, up * hz])))
        if i in (0, nc - 1):                                     # methyl ends: third hydrogen continues the zigzag
            s = -1.0 if i == 0 else 1.0
            atoms.append(("H", c + ch * np.array([s * math.sin(ccc / 2), 0.0, -up * math.cos(ccc / 2)])))
    return [{"symbol": s, "center": list(map(float, r * ANG))} for s, r in atoms]


nc = int(sys.argv[1]) if len(sys.argv) > 1 else 20
solver = sys.argv[2] if len(sys.argv) > 2 else "eigh"
b = json.load(open(os.path.join(ROOT, "tests", "golden", "s22_10_benzene_methane_631g2dfp_jkfit.json")))
atoms = n_alkane(nc)
t0 = time.perf_counter()
res = rhf.run(atoms, b["charges"], b["basis"], b["aux_basis"],
              {"dele": 1e-6, "rmsd": 1e-6, "niter": 50, "density_solver": solver, "df_use_adaptive": False}, output=2)
wall = time.perf_counter() - t0
tm = res["Timings"]
N = res["Overlap"].shape[0]
o = (6 * nc + 2 * nc + 2) // 2
print("C%dH%d  N=%d  converged=%s in %d iterations  E = %.10f Eh  density solver %s" % (nc, 2 * nc + 2, N, res["Converged?"], res["Iterations"], res["Energy"], res["Density Solver"]))
print("wall %.1f s: two-centre %.2f s, Schwarz + packing %.2f s (kept pairs %s of %d = %.1f %%), three-centre %.1f s" % (
    wall, tm.timings.get("two_eri_time", 0.0), tm.timings.get("screening_time", 0.0), tm.non_timing_data.get("screened_indices_count", "all"), N * N,
    100.0 * float(tm.non_timing_data.get("screened_indices_count", N * N)) / (N * N), tm.timings.get("three_eri_time", 0.0)))
eps = res["Orbital Energies"]
print("HOMO %.6f  LUMO %.6f  gap %.6f Eh" % (eps[o - 1], eps[o], eps[o] - eps[o - 1]))
print("last Fock build: " + ", ".join("%s %.2f ms" % (k["name"], 1e3 * k["seconds"]) for k in res["Kernel Stats"]) +
      "; device memory %.1f GB" % (res["Device Bytes"] / 1e9))
