#!/usr/bin/env python3
"""A real molecule at the scale of BASELINE config 4: (H2O)50 (geometry of the reference's example_inputs/w50.json,
tests/golden/w50_geometry.json) / cc-pVDZ / cc-pVDZ-RIFIT (basis data of the reference's water log), DF-RHF on one
MI355X through rhf.run: 1250 AO, 4800 auxiliary functions, Schwarz-screened packed layout, 61 GB of B in HBM.
usage: python tools/run_w50.py [n_waters] [eigh|sp2] [xs|dense]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import juliachem_jl_amd as jc
from juliachem_jl_amd import rhf

nw = int(sys.argv[1]) if len(sys.argv) > 1 else 50
g = json.load(open(os.path.join(ROOT, "tests", "golden", "w50_geometry.json")))
b = json.load(open(os.path.join(ROOT, "tests", "golden", "water_ccpvdz_rifit.json")))
xyz = np.asarray(g["geometry"]).reshape(-1, 3)[:3 * nw] * g["angstrom_to_bohr"]
atoms = [{"symbol": s, "center": list(map(float, r))} for s, r in zip(g["symbols"][:3 * nw], xyz)]
t0 = time.perf_counter()
flags = {"dele": 1e-6, "rmsd": 1e-6, "niter": 40}
if len(sys.argv) > 2:
    flags["density_solver"] = sys.argv[2]                       # eigh | sp2
if len(sys.argv) > 3 and sys.argv[3] == "xs":
    flags["df_exchange_screen"] = True                          # the reference's block-screened exchange (ScreenedDF.jl:431-447)
if len(sys.argv) > 3 and sys.argv[3] == "dense":
    flags["df_force_dense"] = True                              # the unscreened (Q, N^2) map (DensityFitting.jl:78-90; 60 GB of B at N = 1250)
res = rhf.run(atoms, b["charges"], b["basis"], b["aux_basis"], flags, output=2)
wall = time.perf_counter() - t0
tm = res["Timings"]
N = res["Overlap"].shape[0]
print("(H2O)%d  N=%d  converged=%s in %d iterations  E = %.10f Eh  (%.6f Eh per molecule)" % (nw, N, res["Converged?"], res["Iterations"], res["Energy"], res["Energy"] / nw))
print("wall %.1f s: two-centre %.2f s, Schwarz + packing %.2f s (kept pairs %s of %d), three-centre %.1f s" % (
    wall, tm.timings.get("two_eri_time", 0.0), tm.timings.get("screening_time", 0.0), tm.non_timing_data.get("screened_indices_count", "all"), N * N,
    tm.timings.get("three_eri_time", 0.0)))
eps = res["Orbital Energies"]; o = 5 * nw
print("HOMO %.6f  LUMO %.6f  gap %.6f Eh" % (eps[o - 1], eps[o], eps[o] - eps[o - 1]))
print("density solver %s; exchange screening blocks %s" % (res["Density Solver"], tm.non_timing_data.get("df_exchange_screen_blocks")))
print("last Fock build: " + ", ".join("%s %.2f ms" % (k["name"], 1e3 * k["seconds"]) for k in res["Kernel Stats"]) +
      "; device memory %.1f GB" % (res["Device Bytes"] / 1e9))
it = res["Iteration Times"]
print("iteration wall times (ms): " + " ".join("%.1f" % (1e3 * t) for t in it))
tail = sorted(it[len(it) // 2:])
print("median of the second half: %.2f ms per iteration" % (1e3 * tail[len(tail) // 2]))
