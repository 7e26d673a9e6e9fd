"""Shared builder of the water / cc-pVDZ / cc-pVDZ-RIFIT case from the golden fixture
(tests/golden/water_ccpvdz_rifit.json, data extracted from the reference's own log by
oracle/make_water_golden.py).  Integrals come from the oracle's host integral code."""
import functools
import json
import os

import numpy as np

from oracle import integrals as gi

FIXTURE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "water_ccpvdz_rifit.json")


@functools.lru_cache(maxsize=1)
def water():
    d = json.load(open(FIXTURE))
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
