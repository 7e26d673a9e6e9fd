"""S22 complexes made of C, H and O in the basis pair the reference's logs hold tables for (tests/golden/s22_cho.json,
extracted by oracle/make_s22_golden.py): input builders shared by the CPU and GPU tests."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load():
    d = json.load(open(os.path.join(GOLDEN, "s22_cho.json")))
    w = json.load(open(os.path.join(GOLDEN, "water_631g2dfp_jkfit.json")))
    c = json.load(open(os.path.join(GOLDEN, "s22_10_benzene_methane_631g2dfp_jkfit.json")))
    assert w["basis"]["H"] == c["basis"]["H"] and w["aux_basis"]["H"] == c["aux_basis"]["H"]       # one table per element
    basis = {"H": w["basis"]["H"], "O": w["basis"]["O"], "C": c["basis"]["C"]}
    aux = {"H": w["aux_basis"]["H"], "O": w["aux_basis"]["O"], "C": c["aux_basis"]["C"]}
    charges = {"H": 1, "C": 6, "O": 8}
    return d, basis, aux, charges


def atoms_of(d, key):
    c = d["complexes"][str(key)]
    xyz = np.asarray(c["geometry_angstrom"], dtype=np.float64).reshape(-1, 3) * d["angstrom_to_bohr"]     # JCBasis.jl:61
    return [{"symbol": s, "center": list(map(float, r))} for s, r in zip(c["symbols"], xyz)], c


def oracle_energy(atoms, basis, aux, charges, n_occ, dele=1e-9, rmsd=1e-8, niter=80):
    """CPU reference: the library's host integrals feeding the oracle's dense DF SCF (DensityFitting.jl:111-224,
    SCF.jl:399-573)."""
    from juliachem_jl_amd.integrals import HostIntegralEngine
    from oracle import df_fock as orc, scf as oscf
    eng = HostIntegralEngine(atoms, basis, aux, charges)
    N, Q = eng.prim.nbf, eng.aux.nbf
    S, T, V = eng.one_electron()
    H = T + V
    E_nuc = eng.nuclear_repulsion()
    J = eng.calculate_two_center_intgrals()
    B = orc.calculate_B(J + np.tril(J, -1).T, np.asarray(eng.calculate_three_center_integrals(range(0, Q), None)).reshape(Q, N, N, order="F"))
    eng.close()
    res = oscf.rhf_df_scf(H, S, E_nuc, n_occ, lambda C, it: H + orc.df_rhf_fock_build_BLAS(B, C[:, :n_occ]), dele=dele, rmsd=rmsd, niter=niter)
    return res, N, Q
