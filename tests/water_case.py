"""Shared builder of the two water cases from the golden fixtures (tests/golden/*.json, data extracted
from the reference's own logs by oracle/make_water_golden.py):
  "ccpvdz"  water / cc-pVDZ / cc-pVDZ-RIFIT       (25 AO, 96 aux; d functions)
  "631g2dfp" water / 6-31G(2df,p) / cc-pVTZ-JKFIT  (47 AO, 166 aux; sp shells, f and g functions)
Integrals come from the oracle's host integral code."""
import functools
import json
import os

import numpy as np

from oracle import integrals as gi

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FIXTURES = {"ccpvdz": "water_ccpvdz_rifit.json", "631g2dfp": "water_631g2dfp_jkfit.json"}


@functools.lru_cache(maxsize=2)
def water(case: str = "ccpvdz"):
    d = json.load(open(os.path.join(GOLDEN, FIXTURES[case])))
    atoms = d["atoms"]
    prim = gi.build_shells(atoms, d["basis"])
    aux = gi.build_shells(atoms, d["aux_basis"])
    Z = [d["charges"][a["symbol"]] for a in atoms]
    R = np.array([a["center"] for a in atoms])
    S, T, V = gi.one_electron(prim, Z, R)
    out = dict(golden=d, S=S, H=T + V, J2c=gi.two_center(aux), T3=gi.three_center(aux, prim),
               E_nuc=gi.nuclear_repulsion(Z, R), n_occ=int(d["settings"]["Number of electrons"]) // 2,
               aux_shell_nbas=[s.nbas for s in aux], prim_shell_nbas=[s.nbas for s in prim])
    return out
