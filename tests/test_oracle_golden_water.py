"""Pins the CPU oracle against the reference's OWN golden output (SURVEY.md 8c,
golden #1): water / cc-pVDZ / cc-pVDZ-RIFIT, contraction_mode dense, hcore guess,
dele = rmsd = 1e-6.  /root/reference/water_ccpvdz_out.log prints 11 iterations
(E, dE, ||dD||) and the converged energy of the 12th; the oracle (integrals ->
B = L^-1 T -> dense DF Fock build -> DIIS/damping/eigensolve loop) must reproduce
every line.  Tolerance: the log prints 10 decimals and the basis 6 decimals;
observed agreement is <= 1e-8 Eh on every iteration and 1e-10 on the final energy."""
import numpy as np

from oracle import df_fock as orc, scf as oscf
from water_case import water


def test_sizes_match_log():
    w = water()
    g = w["golden"]["settings"]
    assert w["H"].shape[0] == int(g["Number of basis functions"]) == 25
    assert w["J2c"].shape[0] == int(g["Number of auxillary basis functions"]) == 96
    assert np.allclose(np.diag(w["S"]), 1.0, atol=1e-14)            # every Cartesian function unit-normalised
    assert np.allclose(w["T3"], w["T3"].transpose(0, 2, 1))
    assert np.allclose(w["J2c"], w["J2c"].T)


def test_oracle_reproduces_reference_scf_trail():
    w = water()
    g = w["golden"]
    o = w["n_occ"]
    B = orc.calculate_B(w["J2c"], w["T3"])
    res = oscf.rhf_df_scf(w["H"], w["S"], w["E_nuc"], o,
                          lambda C, it: w["H"] + orc.df_rhf_fock_build_BLAS(B, C[:, :o]),
                          dele=1e-6, rmsd=1e-6, niter=50)
    assert res.converged and res.iterations == len(g["trail"]) + 1   # the converged iteration is not printed (SCF.jl:527-547)
    for (it, E, dE, drms), (git, gE, gdE, gdrms) in zip(res.trail, g["trail"]):
        assert it == git
        assert abs(E - gE) < 2e-8, (it, E, gE)
        assert abs(dE - gdE) < 4e-8, (it, dE, gdE)
        assert abs(drms - gdrms) < 1e-8, (it, drms, gdrms)
    assert abs(res.energy - g["final_energy"]) < 1e-9, res.energy


def test_screened_and_sharded_oracle_paths_give_the_same_energy():
    """The packed path with an all-true mask and the 3-shard sum reproduce the same
    converged energy (partition-/layout-invariance on real integrals)."""
    w = water()
    o = w["n_occ"]
    sd = orc.setup_unscreened_screening_matricies(25)
    offs = orc.shard_offsets(w["aux_shell_nbas"], 3)
    shards = [orc.pack_three_center(orc.calculate_B(w["J2c"], w["T3"], range(int(offs[r]), int(offs[r + 1]))), sd)
              for r in range(3)]
    res = oscf.rhf_df_scf(w["H"], w["S"], w["E_nuc"], o,
                          lambda C, it: orc.df_rhf_fock_build(shards, C, o, w["H"], sd, "screened"),
                          dele=1e-6, rmsd=1e-6, niter=50)
    assert abs(res.energy - w["golden"]["final_energy"]) < 1e-9


def test_oracle_reproduces_second_reference_trail_with_f_and_g_functions():
    """Golden #2: water / 6-31G(2df,p) / cc-pVTZ-JKFIT from the reference's
    test/water_new_algo-4-8.log — sp shells, Cartesian f functions in the AO basis and f, g functions in the
    auxiliary basis, 13 printed iterations (this older log also prints the converged one).  The basis is printed
    with 6 decimals: away from convergence the energy is first-order sensitive to that (<= 1e-5 Eh in iterations
    1-3, 2e-7 in iteration 4), at convergence second-order: every later line and the final energy agree to 1e-7 Eh."""
    w = water("631g2dfp")
    g = w["golden"]
    o = w["n_occ"]
    assert w["H"].shape[0] == int(g["settings"]["Number of basis functions"]) == 47
    assert w["J2c"].shape[0] == int(g["settings"]["Number of auxillary basis functions"]) == 166
    assert np.allclose(np.diag(w["S"]), 1.0, atol=1e-14)
    B = orc.calculate_B(w["J2c"], w["T3"])
    res = oscf.rhf_df_scf(w["H"], w["S"], w["E_nuc"], o,
                          lambda C, it: w["H"] + orc.df_rhf_fock_build_BLAS(B, C[:, :o]),
                          dele=1e-6, rmsd=1e-6, niter=20)
    assert res.converged and res.iterations == len(g["trail"]) == 13
    for (it, E, dE, drms), (git, gE, gdE, gdrms) in zip(res.trail, g["trail"]):
        assert it == git
        assert abs(E - gE) < (1e-5 if it <= 3 else (2e-7 if it == 4 else 1e-7)), (it, E, gE)
        assert abs(drms - gdrms) < (2e-5 if it <= 3 else 2e-7), (it, drms, gdrms)
    assert abs(res.energy - g["final_energy"]) < 1e-7, res.energy
