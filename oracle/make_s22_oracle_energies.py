#!/usr/bin/env python3
"""Runs the CPU oracle (library host integrals -> oracle/df_fock.py dense DF Fock build -> oracle/scf.py SCF loop,
dele 1e-9 / rmsd 1e-8, core guess) on the ten S22 complexes of tests/golden/s22_cho.json and stores the converged
density-fitted RHF energies in tests/golden/s22_cho_oracle.json, so that the GPU tests of the three largest complexes
need not repeat a minute of CPU SCF each.  tests/test_s22_cho.py re-derives the small ones on every CPU run."""
import json
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import s22_case

d, basis, aux, charges = s22_case.load()
out = {"source": "oracle/make_s22_oracle_energies.py: CPU oracle, dele 1e-9, rmsd 1e-8, guess hcore, dense DF", "energies": {}}
for key in d["complexes"]:
    atoms, c = s22_case.atoms_of(d, key)
    n_occ = (sum(charges[a["symbol"]] for a in atoms) - c["molecular_charge"]) // 2
    t0 = time.time()
    res, N, Q = s22_case.oracle_energy(atoms, basis, aux, charges, n_occ)
    assert res.converged
    out["energies"][key] = {"energy": res.energy, "iterations": len(res.trail), "N": N, "Q": Q, "n_occ": n_occ}
    print(key, c["name"], N, Q, n_occ, "%.10f" % res.energy, "dE(GAMESS RHF) %.2e" % (res.energy - c["gamess_rhf_energy"]), "%.0f s" % (time.time() - t0), flush=True)
json.dump(out, open(os.path.join(ROOT, "tests", "golden", "s22_cho_oracle.json"), "w"), indent=1)
