"""CPU oracle for the SCF iteration that calls the DF Fock build.

TEST INFRASTRUCTURE ONLY (see oracle/df_fock.py header).

Restates /root/reference/src/rhf/energy/SCF.jl:69-262 (rhf_kernel),
:340-592 (scf_cycles_kernel), :1072-1125 (iteration) and
EnergyHelpers.jl:234-258 (DIIS) in DF mode with guess "hcore".
The Fock builder is a callable so the same loop drives the numpy oracle and
the HIP product path (tests compare the two trails).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Callable, List, Optional, Tuple

import numpy as np
import scipy.linalg as sla


def build_orthogonalizer(S: np.ndarray, threshold: float = 1.0e-6) -> np.ndarray:
    """X = U_k diag(s_k^-1/2) U_k^T, dropping s < 1e-6 (SCF.jl:142-162)."""
    s, U = np.linalg.eigh(S)
    keep = s >= threshold
    return (U[:, keep] * (s[keep] ** -0.5)) @ U[:, keep].T


def iteration(F: np.ndarray, H: np.ndarray, X: np.ndarray, n_occ: int
              ) -> Tuple[float, np.ndarray, np.ndarray, np.ndarray]:
    """One diagonalisation step (SCF.jl:1072-1125): F' = X F X, eigh, C = X U,
    D = 2 C_o C_o^T, E_elec = (<D,F> + <D,H>)/2.  Returns (E_elec, eps, C, D)."""
    Fp = X.T @ F @ X
    eps, U = np.linalg.eigh(0.5 * (Fp + Fp.T))
    C = X @ U
    Co = C[:, :n_occ]
    D = 2.0 * (Co @ Co.T)
    E_elec = 0.5 * (np.vdot(D, F) + np.vdot(D, H))
    return float(E_elec), eps, C, D


def DIIS(e_array: List[np.ndarray], F_array: List[np.ndarray], B_dim: int) -> np.ndarray:
    """Pulay extrapolation (EnergyHelpers.jl:234-258): B_ij = <e_i,e_j>, border
    -1, rhs (0..0,-1), solved with LAPACK sysv('U'); newest entry first."""
    B = np.empty((B_dim + 1, B_dim + 1))
    for i in range(B_dim):
        for j in range(B_dim):
            B[i, j] = np.vdot(e_array[i], e_array[j])
        B[i, B_dim] = -1.0
        B[B_dim, i] = -1.0
    B[B_dim, B_dim] = 0.0
    rhs = np.zeros(B_dim + 1)
    rhs[B_dim] = -1.0
    sysv, = sla.get_lapack_funcs(("sysv",), (B, rhs))
    _, _, coeff, info = sysv(B, rhs, lower=0)
    if info != 0:
        raise np.linalg.LinAlgError("sysv info=%d" % info)
    F = np.zeros_like(F_array[0])
    for i in range(B_dim):
        F += coeff[i] * F_array[i]
    return F


@dataclass
class SCFResult:
    energy: float
    converged: bool
    iterations: int
    trail: List[Tuple[int, float, float, float]] = field(default_factory=list)  # (iter, E, dE, Drms)
    F: Optional[np.ndarray] = None
    D: Optional[np.ndarray] = None
    C: Optional[np.ndarray] = None
    eps: Optional[np.ndarray] = None


def rhf_df_scf(H: np.ndarray, S: np.ndarray, E_nuc: float, n_occ: int,
               fock_build: Callable[[np.ndarray, int], np.ndarray],
               dele: float = 1.0e-6, rmsd: float = 1.0e-6, niter: int = 10,
               ndiis: int = 10) -> SCFResult:
    """DF-RHF SCF with hcore guess.

    `fock_build(C, iter)` returns the full F = H + 2J - K built from the first
    n_occ columns of C (the df_rhf_fock_build! contract, DensityFitting.jl:23-76).
    Convergence: |dE| <= dele and ||dD||_F <= rmsd (SCF.jl:527-547); iteration
    cap `niter` = df_max_iterations (SCF.jl:596-598)."""
    X = build_orthogonalizer(S)
    F = H.copy()
    _, eps, C, D = iteration(F, H, X, n_occ)            # "iteration 0", SCF.jl:178-181
    F_old = F.copy()
    E_old = 0.0
    dE = 1.0
    B_dim = 1
    e_hist: List[np.ndarray] = []
    F_hist: List[np.ndarray] = []
    res = SCFResult(0.0, False, 0)
    it = 1
    while True:
        F = np.array(fock_build(C, it), dtype=np.float64, copy=True)       # SCF.jl:463
        if ndiis > 0:                                                      # SCF.jl:472-501
            FDS = (F @ D) @ S
            e = FDS - FDS.T
            e_hist = [e.copy()] + e_hist[:ndiis - 1]
            F_hist = [F.copy()] + F_hist[:ndiis - 1]
            if it > 1:
                B_dim = min(B_dim + 1, ndiis)
                try:
                    F = DIIS(e_hist, F_hist, B_dim)
                except Exception:                                           # "Faulty DIIS!"
                    B_dim = 2
        x = 1.0 / math.log(50.0 * dE, 50.0) if dE >= 1.0 else 1.0           # SCF.jl:504
        F = (1.0 - x) * F_old + x * F
        F_old = F.copy()
        D_old = D
        E_elec, eps, C, D = iteration(F, H, X, n_occ)                       # SCF.jl:513
        D_rms = float(np.sqrt(np.vdot(D - D_old, D - D_old)))               # Frobenius (:521-522)
        E = E_elec + E_nuc
        dE = E - E_old
        res.trail.append((it, E, dE, D_rms))
        res.energy, res.iterations = E, it
        res.F, res.D, res.C, res.eps = F, D, C, eps
        if abs(dE) <= dele and D_rms <= rmsd:                               # SCF.jl:527-547
            res.converged = True
            break
        if it >= niter:                                                     # SCF.jl:565-568
            res.converged = False
            break
        it += 1
        E_old = E
    return res
