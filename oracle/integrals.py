"""Small host Gaussian-integral code (McMurchie-Davidson) for the ORACLE only.

TEST INFRASTRUCTURE ONLY.  Its one purpose is to pin the CPU oracle
(oracle/df_fock.py + oracle/scf.py) against the reference's own golden log
`/root/reference/water_ccpvdz_out.log` (water / cc-pVDZ / cc-pVDZ-RIFIT, dense DF,
11 printed iterations + final energy -75.9911548795 Eh): the reference obtains
its integrals from Libint 2.7.0 (libint_jll 2.7.0+0, Manifest.toml:756-760; call
sites deps/src/jeri-df-tei.hpp:56-58,82, jeri-oei.hpp:61,106,155), which is not in
the image, so the published McMurchie-Davidson scheme is restated here
(Helgaker, Jorgensen, Olsen, "Molecular Electronic-Structure Theory", ch. 9).

Conventions reproduced from the reference:
  * Cartesian functions in Libint order (xx,xy,xz,yy,yz,zz; ...), `pure = false`
    (jeri-core.hpp:53-54), Cartesian counts (l+1)(l+2)/2 (BasisStructs.jl:31-33);
  * every Cartesian function individually unit-normalised: Libint normalises the
    axial function of a contracted shell, JuliaChem's axial_normalization_factor
    then rescales xy, xxy, xyz ... (Globals.jl:6-28, EnergyHelpers.jl:260-411);
  * two-centre (P|Q), three-centre (P|mu nu) Coulomb integrals, overlap, kinetic,
    nuclear attraction; E_nuc = sum Z_i Z_j / r_ij (EnergyHelpers.jl:5-23).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Sequence, Tuple

import numpy as np
from scipy.special import hyp1f1


def cart_list(l: int) -> List[Tuple[int, int, int]]:
    return [(lx, ly, l - lx - ly) for lx in range(l, -1, -1) for ly in range(l - lx, -1, -1)]


def dfact(n: int) -> float:
    return 1.0 if n <= 0 else float(np.prod(np.arange(n, 0, -2, dtype=np.float64)))


@dataclass
class GShell:
    l: int
    exps: np.ndarray
    coefs: np.ndarray            # raw contraction coefficients (of normalised primitives)
    center: np.ndarray

    @property
    def nbas(self) -> int:
        return (self.l + 1) * (self.l + 2) // 2


def prim_norm(alpha: np.ndarray, lx: int, ly: int, lz: int) -> np.ndarray:
    l = lx + ly + lz
    return ((2.0 * alpha / math.pi) ** 0.75 * (4.0 * alpha) ** (l / 2.0)
            / math.sqrt(dfact(2 * lx - 1) * dfact(2 * ly - 1) * dfact(2 * lz - 1)))


def contracted_coefs(sh: GShell) -> np.ndarray:
    """(ncart, nprim) coefficients of UNnormalised primitives x^lx y^ly z^lz e^{-a r^2}
    such that every contracted Cartesian function has unit self-overlap."""
    out = np.empty((sh.nbas, sh.exps.size))
    a = sh.exps
    for k, (lx, ly, lz) in enumerate(cart_list(sh.l)):
        c = sh.coefs * prim_norm(a, lx, ly, lz)
        l = sh.l
        # same-centre overlap of two primitives with equal (lx,ly,lz)
        p = a[:, None] + a[None, :]
        s = (math.pi / p) ** 1.5 * dfact(2 * lx - 1) * dfact(2 * ly - 1) * dfact(2 * lz - 1) / (2.0 * p) ** l
        out[k] = c / math.sqrt(c @ s @ c)
    return out


# ---- Hermite expansion coefficients -------------------------------------------------
def hermite_E(la: int, lb: int, a: np.ndarray, b: np.ndarray, XAB: float) -> np.ndarray:
    """E[i, j, t, n] for one Cartesian direction; a, b are flattened primitive-pair
    exponent arrays (b may be 0 for a one-centre 'pair')."""
    p = a + b
    mu = a * b / p
    XPA = -b / p * XAB
    XPB = a / p * XAB
    n = a.size
    E = np.zeros((la + 1, lb + 1, la + lb + 2, n))
    E[0, 0, 0] = np.exp(-mu * XAB * XAB)
    for i in range(la):
        for t in range(i + 2):
            E[i + 1, 0, t] = XPA * E[i, 0, t] + (t + 1) * E[i, 0, t + 1]
            if t > 0:
                E[i + 1, 0, t] += E[i, 0, t - 1] / (2.0 * p)
    for i in range(la + 1):
        for j in range(lb):
            for t in range(i + j + 2):
                E[i, j + 1, t] = XPB * E[i, j, t] + (t + 1) * E[i, j, t + 1]
                if t > 0:
                    E[i, j + 1, t] += E[i, j, t - 1] / (2.0 * p)
    return E[:, :, :la + lb + 1]


def boys(nmax: int, x: np.ndarray) -> np.ndarray:
    """F_n(x), n = 0..nmax (downward recursion from the Kummer form)."""
    x = np.asarray(x, dtype=np.float64)
    F = np.empty((nmax + 1,) + x.shape)
    F[nmax] = hyp1f1(nmax + 0.5, nmax + 1.5, -x) / (2 * nmax + 1)
    ex = np.exp(-x)
    for n in range(nmax, 0, -1):
        F[n - 1] = (2.0 * x * F[n] + ex) / (2 * n - 1)
    return F


def hermite_R(L: int, alpha: np.ndarray, R: np.ndarray) -> np.ndarray:
    """R[t,u,v,n] = R^0_{tuv}(alpha, R) for t+u+v <= L; R is (3, n)."""
    n = alpha.size
    r2 = (R * R).sum(axis=0)
    F = boys(L, alpha * r2)
    # Rn[m][t,u,v]
    cur = np.zeros((L + 1, L + 1, L + 1, L + 1, n))          # index [m, t, u, v]
    for m in range(L + 1):
        cur[m, 0, 0, 0] = (-2.0 * alpha) ** m * F[m]
    X, Y, Z = R
    for t in range(L + 1):
        for u in range(L + 1 - t):
            for v in range(L + 1 - t - u):
                if t == u == v == 0:
                    continue
                mmax = L - (t + u + v)
                for m in range(mmax + 1):
                    if t > 0:
                        val = X * cur[m + 1, t - 1, u, v]
                        if t > 1:
                            val = val + (t - 1) * cur[m + 1, t - 2, u, v]
                    elif u > 0:
                        val = Y * cur[m + 1, t, u - 1, v]
                        if u > 1:
                            val = val + (u - 1) * cur[m + 1, t, u - 2, v]
                    else:
                        val = Z * cur[m + 1, t, u, v - 1]
                        if v > 1:
                            val = val + (v - 1) * cur[m + 1, t, u, v - 2]
                    cur[m, t, u, v] = val
    return cur[0]


# ---- shell-pair Hermite densities -----------------------------------------------------
@dataclass
class PairData:
    L: int                 # la + lb
    p: np.ndarray          # (n,)
    P: np.ndarray          # (3, n)
    H: np.ndarray          # (ncart_a * ncart_b, L+1, L+1, L+1, n)  contracted-coefficient weighted
    na: int
    nb: int


def pair_data(sa: GShell, sb: GShell = None) -> PairData:
    """Hermite expansion of the product of two shells (sb None: a single shell)."""
    ca = contracted_coefs(sa)
    if sb is None:
        a = sa.exps
        b = np.zeros_like(a)
        cb = np.ones((1, 1))
        lb, B, nb, w = 0, sa.center, 1, ca[:, None, :]                       # (na, 1, n)
        carts_b = [(0, 0, 0)]
    else:
        a = np.repeat(sa.exps, sb.exps.size)
        b = np.tile(sb.exps, sa.exps.size)
        cbm = contracted_coefs(sb)
        lb, B, nb = sb.l, sb.center, sb.nbas
        w = (ca[:, None, :, None] * cbm[None, :, None, :]).reshape(sa.nbas, nb, -1)
        carts_b = cart_list(sb.l)
    la, A, na = sa.l, sa.center, sa.nbas
    p = a + b
    P = (a[None, :] * A[:, None] + b[None, :] * B[:, None]) / p[None, :]
    E = [hermite_E(la, lb, a, b, A[d] - B[d]) for d in range(3)]
    L = la + lb
    H = np.zeros((na * nb, L + 1, L + 1, L + 1, a.size))
    for ia, (ax, ay, az) in enumerate(cart_list(la)):
        for ib, (bx, by, bz) in enumerate(carts_b):
            ex = E[0][ax, bx][:ax + bx + 1]
            ey = E[1][ay, by][:ay + by + 1]
            ez = E[2][az, bz][:az + bz + 1]
            H[ia * nb + ib, :ax + bx + 1, :ay + by + 1, :az + bz + 1] = (
                ex[:, None, None, :] * ey[None, :, None, :] * ez[None, None, :, :] * w[ia, ib][None, None, None, :])
    return PairData(L, p, P, H, na, nb)


def eri_block(bra: PairData, ket: PairData) -> np.ndarray:
    """Contracted Coulomb integrals (bra|ket), shape (nbra_funcs, nket_funcs)."""
    nb_, nk_ = bra.p.size, ket.p.size
    p = np.repeat(bra.p, nk_)
    q = np.tile(ket.p, nb_)
    alpha = p * q / (p + q)
    PQ = np.repeat(bra.P, nk_, axis=1) - np.tile(ket.P, (1, nb_))
    L = bra.L + ket.L
    R = hermite_R(L, alpha, PQ)                                   # (L+1,)*3 + (n,)
    pref = 2.0 * math.pi ** 2.5 / (p * q * np.sqrt(p + q))
    R = R * pref
    Lb, Lk = bra.L, ket.L
    tb = np.arange(Lb + 1)
    tk = np.arange(Lk + 1)
    idx = tb[:, None] + tk[None, :]                               # (Lb+1, Lk+1)
    Rexp = R[idx[:, None, None, :, None, None], idx[None, :, None, None, :, None], idx[None, None, :, None, None, :]]
    # Rexp[t,u,v,x,y,z,n]
    sign = (-1.0) ** (tk[:, None, None] + tk[None, :, None] + tk[None, None, :])
    Hk = ket.H * sign[None, :, :, :, None]
    Rexp = Rexp.reshape(Lb + 1, Lb + 1, Lb + 1, Lk + 1, Lk + 1, Lk + 1, nb_, nk_)
    return np.einsum("atuvb,kxyzn,tuvxyzbn->ak", bra.H, Hk, Rexp, optimize=True)


# ---- one-electron integrals --------------------------------------------------------------
def _overlap_1d(E: np.ndarray, i: int, j: int, p: np.ndarray) -> np.ndarray:
    return E[i, j, 0] * np.sqrt(math.pi / p)


def one_electron(shells: Sequence[GShell], charges: Sequence[float], centers: np.ndarray
                 ) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """Overlap S, kinetic T and nuclear attraction V over unit-normalised Cartesians."""
    pos = np.concatenate([[0], np.cumsum([s.nbas for s in shells])])
    n = int(pos[-1])
    S = np.zeros((n, n)); T = np.zeros((n, n)); V = np.zeros((n, n))
    for ia, sa in enumerate(shells):
        ca = contracted_coefs(sa)
        for ib, sb in enumerate(shells[:ia + 1]):
            cb = contracted_coefs(sb)
            a = np.repeat(sa.exps, sb.exps.size)
            b = np.tile(sb.exps, sa.exps.size)
            p = a + b
            A, B = sa.center, sb.center
            E = [hermite_E(sa.l, sb.l + 2, a, b, A[d] - B[d]) for d in range(3)]
            P = (a[None, :] * A[:, None] + b[None, :] * B[:, None]) / p[None, :]
            Lh = sa.l + sb.l
            Rn = [hermite_R(Lh, p, P - C[:, None]) for C in centers]
            for ka, (ax, ay, az) in enumerate(cart_list(sa.l)):
                for kb, (bx, by, bz) in enumerate(cart_list(sb.l)):
                    w = (ca[ka][:, None] * cb[kb][None, :]).reshape(-1)
                    sx = _overlap_1d(E[0], ax, bx, p); sy = _overlap_1d(E[1], ay, by, p); sz = _overlap_1d(E[2], az, bz, p)

                    def kin1d(Ed, i, j):
                        t = -2.0 * b * b * _overlap_1d(Ed, i, j + 2, p) + b * (2 * j + 1) * _overlap_1d(Ed, i, j, p)
                        if j >= 2:
                            t = t - 0.5 * j * (j - 1) * _overlap_1d(Ed, i, j - 2, p)
                        return t
                    s_val = np.sum(w * sx * sy * sz)
                    t_val = np.sum(w * (kin1d(E[0], ax, bx) * sy * sz + sx * kin1d(E[1], ay, by) * sz
                                        + sx * sy * kin1d(E[2], az, bz)))
                    v_val = 0.0
                    for Z, R in zip(charges, Rn):
                        acc = np.zeros_like(p)
                        for t in range(ax + bx + 1):
                            for u in range(ay + by + 1):
                                for v in range(az + bz + 1):
                                    acc += E[0][ax, bx, t] * E[1][ay, by, u] * E[2][az, bz, v] * R[t, u, v]
                        v_val -= Z * np.sum(w * 2.0 * math.pi / p * acc)
                    i, j = pos[ia] + ka, pos[ib] + kb
                    S[i, j] = S[j, i] = s_val
                    T[i, j] = T[j, i] = t_val
                    V[i, j] = V[j, i] = v_val
    return S, T, V


def two_center(aux: Sequence[GShell]) -> np.ndarray:
    pos = np.concatenate([[0], np.cumsum([s.nbas for s in aux])])
    n = int(pos[-1])
    J = np.zeros((n, n))
    pd = [pair_data(s) for s in aux]
    for i in range(len(aux)):
        for j in range(i + 1):
            blk = eri_block(pd[i], pd[j])
            J[pos[i]:pos[i + 1], pos[j]:pos[j + 1]] = blk
            J[pos[j]:pos[j + 1], pos[i]:pos[i + 1]] = blk.T
    return J


def three_center(aux: Sequence[GShell], prim: Sequence[GShell]) -> np.ndarray:
    """(Q | mu nu) dense, shape (Naux, N, N)."""
    pa = np.concatenate([[0], np.cumsum([s.nbas for s in aux])])
    pb = np.concatenate([[0], np.cumsum([s.nbas for s in prim])])
    T = np.zeros((int(pa[-1]), int(pb[-1]), int(pb[-1])))
    pda = [pair_data(s) for s in aux]
    for m in range(len(prim)):
        for n in range(m + 1):
            ket = pair_data(prim[m], prim[n])
            for q in range(len(aux)):
                blk = eri_block(pda[q], ket).reshape(aux[q].nbas, prim[m].nbas, prim[n].nbas)
                T[pa[q]:pa[q + 1], pb[m]:pb[m + 1], pb[n]:pb[n + 1]] = blk
                T[pa[q]:pa[q + 1], pb[n]:pb[n + 1], pb[m]:pb[m + 1]] = blk.transpose(0, 2, 1)
    return T


def nuclear_repulsion(charges: Sequence[float], centers: np.ndarray) -> float:
    e = 0.0
    for i in range(len(charges)):
        for j in range(i):
            e += charges[i] * charges[j] / np.linalg.norm(centers[i] - centers[j])
    return float(e)


def build_shells(atoms: Sequence[Dict], basis: Dict[str, List[Dict]]) -> List[GShell]:
    """atoms: [{'symbol','center'}]; basis[symbol] = [{'l', 'exps', 'coefs'}] in input order."""
    out = []
    for at in atoms:
        for sh in basis[at["symbol"]]:
            out.append(GShell(int(sh["l"]), np.asarray(sh["exps"], float), np.asarray(sh["coefs"], float),
                              np.asarray(at["center"], float)))
    return out
