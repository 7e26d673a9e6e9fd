"""Real molecules — (H2O)n / cc-pVDZ / cc-pVDZ-RIFIT (the molecule of BASELINE config 4 at n = 50: 1250 AO, 4800 auxiliary
functions) and n-alkanes / 6-31G(2df,p) / cc-pVTZ-JKFIT (n-C20H42: the real molecule of bench.py) — in the Schwarz-screened packed
layout through the HIP path against the CPU oracle ON THE SAME INTEGRALS:

  (a) one Fock build with the core-Hamiltonian guess orbitals: F elementwise to 1e-11 max|F| (times sqrt(cond(J2c) / 1e6) for
      the ill-conditioned JKFIT metric);
  (c) the Hartree-Fock energy functional E[D] = E_nuc + <D, H + F(D)>/2 at the converged density of the device SCF, F(D) once
      from the device and once from the oracle: 1e-8 Eh (variational in D: this is the operators compared in energy units);
  (b) the SCF energies of the first iterations of `rhf.run` against the oracle's SCF loop with the oracle's screened Fock
      build, both from the hcore guess: 1e-8 Eh per iteration (north_star's bar) once the density has settled, 1e-11 |E|
      while it still moves by |dD| >= 1 (the iteration energy is not variational there).

(H2O)6 and n-butane run in the default suite (seconds).  The full size is minutes of host BLAS (B formation 9e12 flop, every oracle Fock build
4e12) and runs only with JCDF_RUN_SLOW=1 (JCDF_SLOW_WATERS, JCDF_SLOW_ITERS choose the size and the number of compared
iterations; the n-C20H42 case runs its whole SCF on both sides); the outcomes on MI355X are recorded in
profiles/r03_w50_oracle_parity.txt and profiles/r03_c20h42_oracle_parity.txt.  What this pins: the real-molecule numbers of
DESIGN.md against the reference's algorithm as restated by the oracle — not against published energies (the reference holds
none for these molecules)."""
import json
import os
import time

import numpy as np
import pytest

import juliachem_jl_amd as jc
from juliachem_jl_amd import rhf
from juliachem_jl_amd.df import get_screening_metadata, packed_pq_lists
from juliachem_jl_amd.integrals import HostIntegralEngine
from oracle import df_fock as orc
from oracle import scf as oscf

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
SLOW = bool(os.environ.get("JCDF_RUN_SLOW"))


def _cluster(nw):
    g = json.load(open(os.path.join(HERE, "golden", "w50_geometry.json")))
    b = json.load(open(os.path.join(HERE, "golden", "water_ccpvdz_rifit.json")))
    xyz = np.asarray(g["geometry"]).reshape(-1, 3)[:3 * nw] * g["angstrom_to_bohr"]
    atoms = [{"symbol": s, "center": list(map(float, r))} for s, r in zip(g["symbols"][:3 * nw], xyz)]
    return atoms, b


def _alkane(nc):
    from juliachem_jl_amd.synthetic import n_alkane
    b = json.load(open(os.path.join(HERE, "golden", "s22_10_benzene_methane_631g2dfp_jkfit.json")))   # C and H: 6-31G(2df,p) / cc-pVTZ-JKFIT
    return n_alkane(nc), b


def _check(nw, n_iter, log=print, molecule=None, compare_trail=True, dele=1e-6, rmsd=1e-6):
    import torch
    from juliachem_jl_amd.engine import DeviceFockBuilder
    atoms, b = molecule if molecule is not None else _cluster(nw)
    atoms = [atoms[i] for i in rhf.spatial_order(atoms)]            # rhf.run's internal order, so that both sides see one AO order
    t0 = time.perf_counter()
    eng = HostIntegralEngine(atoms, b["basis"], b["aux_basis"], b["charges"])
    N, Q, o = eng.prim.nbf, eng.aux.nbf, int(round(float(np.sum(eng.Z)))) // 2
    S, T, V = eng.one_electron()
    H = T + V
    E_nuc = eng.nuclear_repulsion()
    J2c = eng.calculate_two_center_intgrals()
    # B = L^-1 T inherits the conditioning of the metric: the entries of L^-1 reach sqrt(cond), the sums over the aux index cancel
    # by that factor, and two summation orders (MFMA tiles, host BLAS) differ by ~eps sqrt(cond) in B, and F follows.  (Taking
    # L^-1 from the host's LAPACK instead of the device factorisation does not change it: 1.1e-10 vs 7.9e-11 on n-C20H42.)
    # cc-pVDZ-RIFIT on water: cond ~ 1e6; cc-pVTZ-JKFIT on alkanes: 1.6e10.
    wj = np.linalg.eigvalsh(np.tril(J2c) + np.tril(J2c, -1).T)
    cond = float(wj[-1] / wj[0])
    ftol = 1e-11 * max(1.0, np.sqrt(cond / 1e6))
    mask = eng.schwarz_mask(jc.create_scf_options({"scf_type": "df"}).df_screening_sigma, float(np.max(np.diag(J2c))))
    sd = get_screening_metadata(mask)
    osd = orc.get_screening_metadata(mask)
    pq = packed_pq_lists(sd)
    assert np.array_equal(pq[0], osd.pq_p) and np.array_equal(pq[1], osd.pq_q)         # product and oracle pack alike
    Tp = eng.calculate_three_center_integrals(range(Q), sd)         # (Q, P) column-major
    log("%s: N=%d Q=%d n_occ=%d kept pairs %.1f %%, integrals %.1f s" % ("(H2O)%d" % nw if molecule is None else "n-C%dH%d" % (nw, 2 * nw + 2), N, Q, o,
                                                                        100.0 * mask.mean(), time.perf_counter() - t0))
    # ---- oracle side
    t0 = time.perf_counter()
    Bp = orc.calculate_B(J2c, np.ascontiguousarray(Tp))
    log("oracle B formation %.1f s" % (time.perf_counter() - t0))
    X = oscf.build_orthogonalizer(S)
    w, U = np.linalg.eigh(X.T @ H @ X)
    C0 = X @ U
    t0 = time.perf_counter()
    F_ref = H + orc.df_rhf_fock_build_screened(Bp, C0[:, :o], osd)
    t_build = time.perf_counter() - t0
    log("oracle Fock build %.1f s" % t_build)
    # ---- (a) the same build on the device
    fb = DeviceFockBuilder(N, Q, o, eng.aux.shell_nbas, device=0, pq=pq)
    fb.set_metric(J2c)
    fb.set_core_hamiltonian(H)
    fb.push_three_center_device(0, Q, torch.as_tensor(np.ravel(Tp, order="K"), device=fb.device))
    Ct = torch.as_tensor(np.ascontiguousarray(C0[:, :o].T), device=fb.device)       # (n_occ, N) row-major, DensityFitting.jl:49
    F_dev = fb.build(Ct).cpu().numpy().reshape(N, N)
    rel = np.abs(F_dev - F_ref).max() / np.abs(F_ref).max()
    log("(a) Fock build, hcore-guess orbitals: max|F_hip - F_oracle| / max|F| = %.2e   (metric condition number %.1e: bar %.1e)" % (rel, cond, ftol))
    assert rel < ftol
    assert np.array_equal(F_dev, F_dev.T)
    # ---- (c) the Hartree-Fock energy functional at the CONVERGED density of the device SCF, with either Fock operator:
    #      E[D] = E_nuc + <D, H + F(D)>/2 is variational in D, so this compares the operators in energy units where it matters
    conv = rhf.run(atoms, b["charges"], b["basis"], b["aux_basis"], {"dele": 1e-6, "rmsd": 1e-6, "niter": 60, "reorder_atoms": False,
                                                                       "df_use_adaptive": False})
    assert conv["Converged?"]
    Cc = np.ascontiguousarray(conv["MO Coeff"][:, :o])
    D = 2.0 * (Cc @ Cc.T)
    t0 = time.perf_counter()
    Fc_ref = H + orc.df_rhf_fock_build_screened(Bp, Cc, osd)
    Fc_dev = fb.build(torch.as_tensor(np.ascontiguousarray(Cc.T), device=fb.device)).cpu().numpy().reshape(N, N)
    E_ref = E_nuc + 0.5 * (np.vdot(D, Fc_ref) + np.vdot(D, H))
    E_dev = E_nuc + 0.5 * (np.vdot(D, Fc_dev) + np.vdot(D, H))
    relc = np.abs(Fc_dev - Fc_ref).max() / np.abs(Fc_ref).max()
    log("(c) converged density (%d iterations, E_scf = %.10f): E[D] with F_hip %.10f, with F_oracle %.10f, diff %.1e Eh; "
        "max|dF|/max|F| = %.2e  (%.1f s)" % (conv["Iterations"], conv["Energy"], E_dev, E_ref, E_dev - E_ref, relc, time.perf_counter() - t0))
    assert relc < ftol and abs(E_dev - E_ref) < 1e-8 and abs(conv["Energy"] - E_ref) < 1e-5
    fb.close()
    eng.close()
    # ---- (b) SCF trails
    t0 = time.perf_counter()
    gaps = []

    def oracle_fock(C, it):
        F = H + orc.df_rhf_fock_build_screened(Bp, C[:, :o], osd)
        if it <= 3:
            e = np.linalg.eigvalsh(X.T @ F @ X)
            gaps.append(float(e[o] - e[o - 1]))
        return F
    ref = oscf.rhf_df_scf(H, S, E_nuc, o, oracle_fock, dele=dele, rmsd=rmsd, niter=n_iter)
    log("oracle SCF, %d iterations: %.1f s" % (len(ref.trail), time.perf_counter() - t0))
    log("    HOMO-LUMO gap of the first Fock matrices (oracle): " + ", ".join("%.2e" % g for g in gaps))
    res = rhf.run(atoms, b["charges"], b["basis"], b["aux_basis"], {"dele": dele, "rmsd": rmsd, "niter": n_iter, "reorder_atoms": False,
                                                                      "df_use_adaptive": False})      # the screened path at every size
    worst = 0.0
    bad = []
    for (i1, e1, d1, r1), (i2, e2, d2, r2) in zip(res["Trail"], ref.trail):
        assert i1 == i2
        worst = max(worst, abs(e1 - e2))
        log("   iteration %2d   hip %.10f   oracle %.10f   diff %.1e   |dD| %.3e / %.3e" % (i1, e1, e2, e1 - e2, r1, r2))
        # The iteration energy is <D_new, F_extrapolated + H>/2: not variational, so far from convergence it follows the
        # differences of F (rounding, times the conditioning of the metric) to first order.  1e-8 Eh once the density has settled
        # (|dD| < 1); before that the bar of (a) relative to |E| ((H2O)50: 1e-9 ... 3e-8 Eh at |E| = 3500, |dD| ~ 100).
        tol = 1e-8 if r2 < 1.0 else max(1e-8, ftol * abs(e2))
        if not (abs(e1 - e2) < tol and abs(r1 - r2) < 1e-7 * max(1.0, r2)):
            bad.append((i1, e1, e2, tol))
    if compare_trail:
        assert not bad, bad
        assert len(res["Trail"]) == len(ref.trail) and (len(ref.trail) == n_iter or (res["Converged?"] and ref.converged))
    if ref.converged and res["Converged?"]:
        log("    converged on both sides after %d / %d iterations (dele %.0e, rmsd %.0e): E_hip - E_oracle = %.1e Eh"
            % (len(res["Trail"]), len(ref.trail), dele, rmsd, res["Energy"] - ref.energy))
        assert abs(res["Energy"] - ref.energy) < max(1e-8, 10.0 * dele if compare_trail else 1e-8)
    log("(b) %d SCF iterations: max |E_hip - E_oracle| = %.1e Eh" % (min(len(res["Trail"]), len(ref.trail)), worst))
    return rel, worst


def test_water_hexamer_against_the_oracle():
    _check(6, 8)


@pytest.mark.skipif(not SLOW, reason="minutes of host BLAS: set JCDF_RUN_SLOW=1 (recorded in profiles/r03_w50_oracle_parity.txt)")
def test_full_water_cluster_against_the_oracle():
    nw = int(os.environ.get("JCDF_SLOW_WATERS", "50"))
    out = os.environ.get("JCDF_SLOW_LOG")
    lines = []

    def log(s):
        print(s, flush=True)
        lines.append(s)
        if out:
            with open(out, "w") as f:
                f.write("\n".join(lines) + "\n")
    _check(nw, int(os.environ.get("JCDF_SLOW_ITERS", "6")), log)


@pytest.mark.skipif(not SLOW, reason="minutes of host BLAS: set JCDF_RUN_SLOW=1 (recorded in profiles/r03_c20h42_oracle_parity.txt)")
def test_eicosane_whole_scf_against_the_oracle():
    """n-C20H42 / 6-31G(2df,p) / cc-pVTZ-JKFIT (956 AO, 27 % of the pairs kept: the real molecule of bench.py): checks (a) and
    (c), and the whole SCF on both sides, hcore guess to a TIGHT convergence (1e-9 Eh, 1e-8), compared at the end: 1e-8 Eh.
    The iteration energies are logged side by side but not compared: the core guess of the chain leaves the first Fock matrices
    with a HOMO-LUMO gap small enough for the two eigensolvers to occupy different orbitals at iteration 2 (2.8 Eh apart), after
    which the trails are different SCF paths to the same minimum."""
    out = os.environ.get("JCDF_SLOW_LOG")
    lines = []

    def log(s):
        print(s, flush=True)
        lines.append(s)
        if out:
            with open(out, "w") as f:
                f.write("\n".join(lines) + "\n")
    _check(20, 90, log, molecule=_alkane(20), compare_trail=False, dele=1e-9, rmsd=1e-8)


def test_butane_against_the_oracle():
    _check(4, 40, molecule=_alkane(4))
