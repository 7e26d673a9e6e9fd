"""Seeded synthetic inputs of the exact shapes of the metric configs
(SURVEY.md 8d): the reference cannot produce real integrals in this image
(no Julia/Libint, basis blobs missing), so benches and large tests run on
synthetic tensors; same seed -> same bits on every box."""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Tuple

import numpy as np

SEED = 20241024

# name -> (N AO, Q aux, n_occ)   SURVEY.md 8 size table
CONFIGS = {
    "water": (25, 96, 5),
    "benzene_dimer": (240, 972, 42),
    "C20H42": (510, 1950, 81),
    "w50": (1250, 4800, 250),
    "gly10_vtz": (1915, 5261, 155),
}


@dataclass
class SyntheticDF:
    N: int
    Q: int
    n_occ: int
    J2c: np.ndarray            # (Q, Q) SPD metric
    T: np.ndarray              # (Q, N, N) symmetric in the last two
    C: np.ndarray              # (N, N) orthonormal columns ("MO coefficients")
    H: np.ndarray              # (N, N) symmetric
    mask: Optional[np.ndarray]  # bool (N, N) symmetric keep-mask or None
    aux_shell_nbas: List[int]


def aux_shells(Q: int, rng: np.random.Generator) -> List[int]:
    """Cartesian shell sizes 1,3,6,10 summing to Q (aux sets are s..f)."""
    out: List[int] = []
    left = Q
    sizes = np.array([1, 3, 6, 10])
    while left > 0:
        s = int(rng.choice(sizes[sizes <= left]))
        out.append(s)
        left -= s
    return out


def band_mask(N: int, kept_fraction: float, rng: np.random.Generator) -> np.ndarray:
    """Symmetric band |p-q| <= w plus seeded raggedness so that kept/N^2 ~ kept_fraction
    (C20H42 picture of the reference: ~0.47)."""
    w = max(1, int(round(N * (1.0 - np.sqrt(max(0.0, 1.0 - kept_fraction))))))
    i = np.arange(N)
    d = np.abs(i[:, None] - i[None, :])
    m = d <= w
    salt = np.triu(rng.random((N, N)) < 0.02, 1)
    near = (d > w) & (d <= w + max(2, w // 8))
    m = m | ((salt | salt.T) & near)
    pepper = np.triu(rng.random((N, N)) < 0.02, 1)
    drop = (pepper | pepper.T) & (d >= max(1, w - max(2, w // 8))) & (d <= w)
    m = m & ~drop
    np.fill_diagonal(m, True)
    return m


def cluster_mask(N: int, kept_fraction: float, rng: np.random.Generator, per_site: int = 5) -> np.ndarray:
    """Scattered symmetric keep-mask of a compact 3-D cluster, the pattern a Schwarz test leaves on something like
    (H2O)50: `per_site` consecutive functions share a random site in a cube (sites sorted along a Morton curve, as
    rhf.spatial_order sorts atoms), a pair is kept iff its sites are closer than a cut-off chosen by bisection so that
    kept/N^2 ~ kept_fraction.  Every row keeps a different, non-contiguous set of partners."""
    ns = (N + per_site - 1) // per_site
    R = rng.random((ns, 3))
    cell = np.minimum((R * 8).astype(np.int64), 7)
    key = np.zeros(ns, dtype=np.int64)
    for bit in range(3):
        for d in range(3):
            key |= ((cell[:, d] >> bit) & 1) << (3 * bit + d)
    R = R[np.argsort(key, kind="stable")]
    site = np.minimum(np.arange(N) // per_site, ns - 1)
    d2 = ((R[:, None, :] - R[None, :, :]) ** 2).sum(-1)
    lo, hi = 0.0, 3.0
    for _ in range(40):
        mid = 0.5 * (lo + hi)
        frac = (d2[np.ix_(site, site)] < mid).mean()
        lo, hi = (mid, hi) if frac < kept_fraction else (lo, mid)
    m = d2[np.ix_(site, site)] < hi
    np.fill_diagonal(m, True)
    return m


def make(N: int, Q: int, n_occ: int, seed: int = SEED, kept_fraction: Optional[float] = None,
         dtype=np.float64) -> SyntheticDF:
    rng = np.random.default_rng(seed)
    A = rng.standard_normal((Q, N, N)) * 0.1
    T = 0.5 * (A + A.transpose(0, 2, 1))
    del A
    M = rng.standard_normal((Q, Q))
    J2c = M @ M.T + Q * np.eye(Q)
    Cfull, _ = np.linalg.qr(rng.standard_normal((N, N)))
    Hs = rng.standard_normal((N, N))
    H = 0.5 * (Hs + Hs.T)
    shells = aux_shells(Q, rng)
    mask = None
    if kept_fraction is not None:
        mask = band_mask(N, kept_fraction, rng)
        T = T * mask[None, :, :]
    return SyntheticDF(N, Q, n_occ, J2c, np.ascontiguousarray(T), Cfull, H, mask, shells)


def make_config(name: str, **kw) -> SyntheticDF:
    N, Q, o = CONFIGS[name]
    return make(N, Q, o, **kw)


def n_alkane(nc: int) -> List[dict]:
    """Idealised all-trans n-alkane C_nc H_(2nc+2) (C-C 1.53 A, C-C-C 112 deg, C-H 1.09 A, H-C-H 107 deg), atoms as
    {"symbol", "center" in bohr} — the geometry of tools/run_c20h42.py and of bench.py's real-molecule object (the
    reference's own C20H42 input is not in the snapshot)."""
    import math
    ang = 1.0 / 0.52917724924
    cc, ch, ccc, hch = 1.53, 1.09, math.radians(112.0), math.radians(107.0)
    dx, dz = cc * math.sin(ccc / 2), cc * math.cos(ccc / 2)
    atoms = []
    Cs = [np.array([i * dx, 0.0, 0.5 * dz * (1 if i % 2 == 0 else -1)]) for i in range(nc)]
    for c in Cs:
        atoms.append(("C", c))
    for i, c in enumerate(Cs):
        up = 1.0 if i % 2 == 0 else -1.0                        # side of the zigzag this carbon sticks out to
        hy, hz = ch * math.sin(hch / 2), ch * math.cos(hch / 2)
        atoms.append(("H", c + np.array([0.0, hy, up * hz])))
        atoms.append(("H", c + np.array([0.0, -hy, up * hz])))
        if i in (0, nc - 1):                                     # methyl ends: third hydrogen continues the zigzag
            sgn = -1.0 if i == 0 else 1.0
            atoms.append(("H", c + ch * np.array([sgn * math.sin(ccc / 2), 0.0, -up * math.cos(ccc / 2)])))
    return [{"symbol": sym, "center": list(map(float, r * ang))} for sym, r in atoms]
