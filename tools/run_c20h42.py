#!/usr/bin/env python3
"""The benchmark molecule itself, with real integrals: n-eicosane C20H42 (idealised all-trans geometry: C-C 1.53 A,
C-C-C 112 deg, C-H 1.09 A, H-C-H 107 deg) / 6-31G(2df,p) / cc-pVTZ-JKFIT — the carbon and hydrogen tables of the
reference's S22 log (tests/golden/s22_10_benzene_methane_631g2dfp_jkfit.json; the reference's cc-pVDZ tables for carbon
are not in the snapshot) — DF-RHF on one MI355X through rhf.run: 956 AO, 3390 auxiliary functions, 81 occupied
orbitals, Schwarz-screened packed layout.
usage: python tools/run_c20h42.py [n_carbons] [density_solver]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from juliachem_jl_amd import rhf

from juliachem_jl_amd.synthetic import n_alkane

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
