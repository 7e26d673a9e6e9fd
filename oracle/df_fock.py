"""CPU oracle for the density-fitted RHF Fock build of JuliaChem.jl.

TEST INFRASTRUCTURE ONLY.  Nothing on the product path may import this module:
only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of
`bench.py` use it, and only as the checker / the timed CPU baseline.

This is a numpy restatement (not a copy) of the reference algorithm; every
function cites the reference file:line it follows (paths relative to
/root/reference/src/rhf/energy/DensityFitting/ unless stated).  Arrays use the
reference's *mathematical* index conventions; where the reference's memory
order matters (the packed pq order) it is reproduced exactly.

Parity pinning: see oracle/README.md — the Fock-build algebra + SCF loop are
pinned end-to-end against the reference's own golden log
`water_ccpvdz_out.log` (energy trail of 11 iterations) through
oracle/integrals.py + oracle/scf.py (tests/test_oracle_golden_water.py).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import numpy as np
import scipy.linalg as sla


# --------------------------------------------------------------------------
# shard rule  (DynamicLoad.jl:160-203, GPUDF.jl:1026-1056)
# --------------------------------------------------------------------------
def get_df_static_shell_indices(n_aux_shells: int, n_ranks: int, rank: int) -> range:
    """0-based half-open aux *shell* range of `rank` (DynamicLoad.jl:160-171):
    floor(S/n) shells each, the last rank takes the remainder."""
    n_indices = n_aux_shells // n_ranks
    begin = n_indices * rank
    end = begin + n_indices
    if rank == n_ranks - 1:
        end = n_aux_shells
    return range(begin, end)


def static_load_rank_indicies(rank: int, n_ranks: int, aux_shell_nbas: Sequence[int]) -> Tuple[range, range]:
    """(shell range, aux-function range) of a shard, both 0-based half-open
    (DynamicLoad.jl:174-203).  Function range = first..last function of the
    shard's shells; shells are contiguous so this is a contiguous range."""
    pos = np.concatenate([[0], np.cumsum(np.asarray(aux_shell_nbas, dtype=np.int64))])
    sh = get_df_static_shell_indices(len(aux_shell_nbas), n_ranks, rank)
    if len(sh) == 0:
        return sh, range(0, 0)
    return sh, range(int(pos[sh.start]), int(pos[sh.stop]))


def shard_offsets(aux_shell_nbas: Sequence[int], n_shards: int) -> np.ndarray:
    """q0[s] for s=0..n_shards (GPUDF.jl:1026-1056 applied to all global device ids)."""
    out = np.zeros(n_shards + 1, dtype=np.int64)
    for r in range(n_shards):
        _, fr = static_load_rank_indicies(r, n_shards, aux_shell_nbas)
        out[r] = fr.start
        out[r + 1] = fr.stop
    return out


# --------------------------------------------------------------------------
# packing rule (SchwarzScreening.jl:72-81, :97-111; ScreenedDF.jl:16-77)
# --------------------------------------------------------------------------
@dataclass
class ScreeningData:
    """Mirror of the reference's ScreeningData fields used on the path
    (shared/SCFData.jl:1-17).  All indices are 0-based here; `sparse_pq_index_map`
    holds -1 for screened pairs (reference: 0 in a 1-based map)."""
    basis_function_screen_matrix: np.ndarray          # bool N x N, symmetric
    sparse_pq_index_map: np.ndarray                   # int64 N x N  [q, p] -> packed idx
    screened_indices_count: int
    sparse_p_start_indices: np.ndarray                # int64 N
    non_screened_p_indices_count: np.ndarray          # int64 N   (K_p)
    non_zero_ranges: List[List[range]] = field(default_factory=list)
    pq_p: np.ndarray = None                           # packed idx -> p (outer)
    pq_q: np.ndarray = None                           # packed idx -> q (inner)
    K_block_width: int = 0
    exchange_batch_indexes: List[Tuple[int, int]] = field(default_factory=list)


def build_sparse_pq_index_map(mask: np.ndarray) -> Tuple[np.ndarray, int]:
    """Running index over kept pairs, OUTER loop p, INNER loop q, stored at
    map[q, p]  (SchwarzScreening.jl:72-81)."""
    n = mask.shape[0]
    mp = -np.ones((n, n), dtype=np.int64)
    idx = 0
    for p in range(n):
        keep = np.nonzero(mask[:, p])[0]
        mp[keep, p] = idx + np.arange(keep.size)
        idx += keep.size
    return mp, idx


def get_screening_metadata(mask: np.ndarray) -> ScreeningData:
    """Packed-layout metadata (ScreenedDF.jl:16-77) from a symmetric keep-mask."""
    mask = np.asarray(mask, dtype=bool)
    assert mask.shape[0] == mask.shape[1] and np.array_equal(mask, mask.T)
    n = mask.shape[0]
    mp, count = build_sparse_pq_index_map(mask)
    start = np.zeros(n, dtype=np.int64)
    kp = mask.sum(axis=0).astype(np.int64)
    ranges: List[List[range]] = []
    for p in range(n):
        keep = np.nonzero(mask[:, p])[0]
        start[p] = mp[keep[0], p] if keep.size else 0
        rs: List[range] = []
        if keep.size:
            brk = np.nonzero(np.diff(keep) != 1)[0]
            lo = 0
            for b in list(brk) + [keep.size - 1]:
                rs.append(range(int(keep[lo]), int(keep[b]) + 1))
                lo = b + 1
        ranges.append(rs)
    pq_p = np.empty(count, dtype=np.int64)
    pq_q = np.empty(count, dtype=np.int64)
    qq, pp = np.nonzero(mp >= 0)
    pq_p[mp[qq, pp]] = pp
    pq_q[mp[qq, pp]] = qq
    return ScreeningData(mask, mp, count, start, kp, ranges, pq_p, pq_q)


def setup_unscreened_screening_matricies(n: int) -> ScreeningData:
    """Dense special case P = N^2 (SchwarzScreening.jl:97-111).  The reference
    fills map[pp,qq] = pp + (qq-1)*N, i.e. packed index = q_row + N*p_col with
    the row index fastest — identical to build_sparse_pq_index_map(all-true)."""
    return get_screening_metadata(np.ones((n, n), dtype=bool))


def pack_three_center(T_dense: np.ndarray, sd: ScreeningData) -> np.ndarray:
    """(Q, N, N) dense -> (Q, P) packed, column idx = map[q, p]
    (ThreeCenterIntegralsScreened.jl:8-85 writes exactly these slots)."""
    return np.ascontiguousarray(T_dense[:, sd.pq_q, sd.pq_p])


# --------------------------------------------------------------------------
# B formation (DensityFitting.jl:128-183; ScreenedDF.jl:89-103,134-190)
# --------------------------------------------------------------------------
def form_J_AB_inv(two_center_integrals: np.ndarray) -> np.ndarray:
    """L^-1 with L = chol((P|Q)) lower; only the lower triangle of the input is
    referenced and the result's upper triangle is exactly zero
    (potrf!('L') + trtri!('L','N'), DensityFitting.jl:137-140;
    TwoCenterIntegrals.jl:150-162 zeroes the upper triangle of the input)."""
    a = np.tril(np.asarray(two_center_integrals, dtype=np.float64))
    L = sla.cholesky(a + np.tril(a, -1).T, lower=True)
    Linv = sla.solve_triangular(L, np.eye(L.shape[0]), lower=True)
    return np.tril(Linv)


def calculate_B(two_center_integrals: np.ndarray, T: np.ndarray,
                rows: Optional[range] = None) -> np.ndarray:
    """B = L^-1 . T  (trmm, DensityFitting.jl:152 / ScreenedDF.jl:103).
    `T` is (Q, P) (any packed or dense-flattened pq).  With `rows`, only that
    aux shard of B is returned: B[rows] = sum_s Linv[rows, rows_s] T[rows_s]
    (DensityFitting.jl:153-174, GPUDF.jl:918-997)."""
    Linv = form_J_AB_inv(two_center_integrals)
    T2 = T.reshape(T.shape[0], -1)
    if rows is None:
        return (Linv @ T2).reshape(T.shape)
    return (Linv[rows.start:rows.stop, :] @ T2).reshape((len(rows),) + T.shape[1:])


# --------------------------------------------------------------------------
# dense path (DensityFitting.jl:185-224) == DenseGPUDF.jl:94-113
# --------------------------------------------------------------------------
def calculate_coulomb_dense(B: np.ndarray, C_occ: np.ndarray) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """B: (Q, N, N); C_occ: (N, o).  Returns (2J-part F, V, density)
    DensityFitting.jl:193-198: D~ = C_o C_o^T (no factor 2); V = B.vec(D~);
    F = 2 B^T V."""
    Q, n, _ = B.shape
    density = C_occ @ C_occ.T
    B2 = B.reshape(Q, n * n)
    V = B2 @ density.reshape(n * n)
    F = 2.0 * (B2.T @ V)
    return F.reshape(n, n), V, density


def calculate_exchange_dense(B: np.ndarray, C_occ: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """W[i,Q,mu] = sum_nu C[nu,i] B[Q,mu,nu] (gemm 'T','T', :216);
    K = W^T W over (i,Q) (:219, applied with alpha=-1 onto F).
    Written as the two large GEMMs of the reference (no einsum temporaries), so that the
    timed CPU baseline is a fair restatement: (Q*N x N)(N x o), then (N x Q*o)(Q*o x N)."""
    Q, n, _ = B.shape
    o = C_occ.shape[1]
    W3 = (B.reshape(Q * n, n) @ C_occ).reshape(Q, n, o)          # W3[Q, mu, i]  (B symmetric in mu,nu)
    Wm = np.ascontiguousarray(W3.transpose(1, 0, 2)).reshape(n, Q * o)
    K = Wm @ Wm.T
    return K, W3.transpose(2, 0, 1)                               # W as (o, Q, N), a view


def df_rhf_fock_build_BLAS(B: np.ndarray, C_occ: np.ndarray) -> np.ndarray:
    """Two-electron Fock 2J - K of the dense CPU mode (DensityFitting.jl:111-125)."""
    F, _, _ = calculate_coulomb_dense(B, C_occ)
    K, _ = calculate_exchange_dense(B, C_occ)
    return F - K


# --------------------------------------------------------------------------
# screened / packed path (ScreenedDF.jl:80-132, 242-378, 385-457, 548-641)
# --------------------------------------------------------------------------
def calculate_exchange_block_screen_matrix(n: int, n_blocks: int) -> Tuple[int, int, List[Tuple[int, int]]]:
    """(K_block_width, n_blocks, [(i,j) j<=i]) — ScreenedDF.jl:385-420.
    N < 100 -> a single block."""
    if n < 100:
        return n, 1, [(0, 0)]
    bw = n // n_blocks
    idx = [(i, j) for i in range(n_blocks) for j in range(i + 1)]
    return bw, n_blocks, idx


def calculate_W_screened(Bp: np.ndarray, C_occ_T: np.ndarray, sd: ScreeningData) -> np.ndarray:
    """W[:, :, p] (Q x o) = B[:, start_p:start_p+K_p] . nz_p^T with
    nz_p = C_o[:, kept q of p]  (ScreenedDF.jl:242-289).  C_occ_T is (o, N)
    (permuted at :84).  Returns W as (Q, o, N)."""
    Q = Bp.shape[0]
    o, n = C_occ_T.shape
    W = np.zeros((Q, o, n))
    for p in range(n):
        kp = int(sd.non_screened_p_indices_count[p])
        if kp == 0:
            continue
        s = int(sd.sparse_p_start_indices[p])
        keep = np.nonzero(sd.basis_function_screen_matrix[:, p])[0]
        nz = C_occ_T[:, keep]                       # o x K_p
        W[:, :, p] = Bp[:, s:s + kp] @ nz.T
    return W


def calculate_K_lower_diagonal_block_no_screen(W: np.ndarray, n_blocks: int) -> np.ndarray:
    """Returns the array the reference writes into two_electron_fock with
    beta=0: -W^T W assembled from lower-triangle blocks, mirrored, plus the
    ragged remainder strip (ScreenedDF.jl:548-641)."""
    Q, o, n = W.shape
    W2 = W.reshape(Q * o, n)
    bw, n_blocks, idx = calculate_exchange_block_screen_matrix(n, n_blocks)
    F = np.zeros((n, n))
    for (bi, bj) in idx:
        pr = slice(bi * bw, (bi + 1) * bw)
        qr = slice(bj * bw, (bj + 1) * bw)
        blk = -(W2[:, pr].T @ W2[:, qr])
        F[pr, qr] = blk
        if bi != bj:
            F[qr, pr] = blk.T
    rem = n % n_blocks
    if rem != 0:
        qs = slice(n - rem, n)
        strip = -(W2.T @ W2[:, qs])
        F[:, qs] = strip
        F[qs, :] = strip.T
    return F


def exchange_block_screen(mask: np.ndarray, n_blocks: int, screen: bool = True) -> Tuple[int, int, np.ndarray]:
    """(K_block_width, n_blocks, block_screen_matrix) of calculate_exchange_block_screen_matrix, ScreenedDF.jl:385-457:
    block (pp >= qq) of the lower triangle is kept when basis_function_screen_matrix has a true entry inside
    [p_range, q_range] (:435-441), or always without df_screen_exchange (:443-445).  N < 100: one block (:392-394)."""
    n = mask.shape[0]
    bw, n_blocks, idx = calculate_exchange_block_screen_matrix(n, n_blocks)
    bs = np.zeros((n_blocks, n_blocks), dtype=bool)
    for (pp, qq) in idx:
        if not screen or mask[pp * bw:(pp + 1) * bw, qq * bw:(qq + 1) * bw].sum() != 0:
            bs[pp, qq] = True
    return bw, n_blocks, bs


def calculate_K_lower_diagonal_block(W: np.ndarray, sd: ScreeningData, n_blocks: int) -> np.ndarray:
    """The df_exchange_screen form of the exchange (ScreenedDF.jl:459-545): only the kept blocks of the lower triangle
    are computed (-W^T W, assigned and mirrored, :488-508), then the ragged strip of N mod n_blocks columns against all
    rows (:518-545).  Returned on a FRESH (zero) two_electron_fock, i.e. what iteration 1 leaves: the reference's skipped
    blocks are never written, so from iteration 2 on they would carry the previous iteration's DIIS-mixed entries plus
    another H (DensityFitting.jl:62-65) - a defect of the snapshot (SURVEY Appendix B rule: not replicated); zero is the
    value the algorithm means (no kept pair in the block: K screened, J absent)."""
    Q, o, n = W.shape
    W2 = W.reshape(Q * o, n)
    bw, n_blocks, bs = exchange_block_screen(sd.basis_function_screen_matrix, n_blocks, True)
    F = np.zeros((n, n))
    for pp in range(n_blocks):
        for qq in range(pp + 1):
            if not bs[pp, qq]:
                continue
            pr = slice(pp * bw, (pp + 1) * bw)
            qr = slice(qq * bw, (qq + 1) * bw)
            blk = -(W2[:, pr].T @ W2[:, qr])
            F[pr, qr] = blk
            if pp != qq:
                F[qr, pr] = blk.T
    rem = n % n_blocks
    if rem != 0:
        qs = slice(n - rem, n)
        strip = -(W2.T @ W2[:, qs])
        F[:, qs] = strip
        F[qs, :] = strip.T
    return F


def copy_screened_density_to_array(density: np.ndarray, sd: ScreeningData) -> np.ndarray:
    """density_array[map[i,j]] = 2 D~[i,j] (i>j kept), D~[i,i]; every other
    slot stays 0 (ScreenedDF.jl:305-316)."""
    out = np.zeros(sd.screened_indices_count)
    lower = sd.pq_q >= sd.pq_p           # row index (q, inner) >= column (p, outer)
    i = sd.pq_q[lower]
    j = sd.pq_p[lower]
    out[np.nonzero(lower)[0]] = np.where(i != j, 2.0, 1.0) * density[i, j]
    return out


def calculate_coulomb_screened(Bp: np.ndarray, C_occ_T: np.ndarray, sd: ScreeningData
                               ) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """V and packed J via the lower-triangle runs map[p,p] .. start[p+1]-1
    (ScreenedDF.jl:318-365).  Returns (J_packed, V, density_array)."""
    n = C_occ_T.shape[1]
    density = C_occ_T.T @ C_occ_T
    d = copy_screened_density_to_array(density, sd)
    Q = Bp.shape[0]
    V = np.zeros(Q)
    J = np.zeros(sd.screened_indices_count)
    mp = sd.sparse_pq_index_map
    for p in range(n):
        lo = int(mp[p, p])
        hi = int(sd.sparse_p_start_indices[p + 1]) if p + 1 < n else sd.screened_indices_count
        V += Bp[:, lo:hi] @ d[lo:hi]
    for p in range(n):
        lo = int(mp[p, p])
        hi = int(sd.sparse_p_start_indices[p + 1]) if p + 1 < n else sd.screened_indices_count
        J[lo:hi] += 2.0 * (Bp[:, lo:hi].T @ V)
    return J, V, d


def copy_screened_coulomb_to_fock(F: np.ndarray, J: np.ndarray, sd: ScreeningData) -> None:
    """F[i,j] += J[map[i,j]] for kept i>=j, then F[j,i] = F[i,j]
    (ScreenedDF.jl:367-378)."""
    lower = sd.pq_q >= sd.pq_p
    i = sd.pq_q[lower]
    j = sd.pq_p[lower]
    F[i, j] += J[np.nonzero(lower)[0]]
    F[j, i] = F[i, j]


def df_rhf_fock_build_screened(Bp: np.ndarray, C_occ: np.ndarray, sd: ScreeningData,
                               n_blocks: int = 10, screen_exchange: bool = False) -> np.ndarray:
    """2J - K of the default CPU mode: exchange first (assign, beta=0), then
    Coulomb added (ScreenedDF.jl:130-131).  Bp: (Q, P) packed; C_occ: (N, o).
    screen_exchange = scf flag df_exchange_screen (ScreenedDF.jl:231-235)."""
    C_T = np.ascontiguousarray(C_occ.T)
    W = calculate_W_screened(Bp, C_T, sd)
    if screen_exchange:
        F = calculate_K_lower_diagonal_block(W, sd, n_blocks)
    else:
        F = calculate_K_lower_diagonal_block_no_screen(W, n_blocks)
    J, _, _ = calculate_coulomb_screened(Bp, C_T, sd)
    copy_screened_coulomb_to_fock(F, J, sd)
    return F


# --------------------------------------------------------------------------
# dispatcher + shard reduction (DensityFitting.jl:23-76)
# --------------------------------------------------------------------------
def df_rhf_fock_build(B_shards: Sequence[np.ndarray], coefficients: np.ndarray, n_occ: int,
                      H: np.ndarray, sd: Optional[ScreeningData] = None,
                      contraction_mode: str = "dense") -> np.ndarray:
    """Full F = H + sum_shards (2J_s - K_s).  C_o = C[:, :n_occ] (:49); H added
    on shard 0 only (:62-65); sum over shards == MPI.Allreduce! (:68-71).
    B_shards[s] is (Q_s, N, N) for "dense" or (Q_s, P) for "screened"."""
    C_occ = np.ascontiguousarray(coefficients[:, :n_occ])
    n = coefficients.shape[0]
    F = np.zeros((n, n))
    for s, Bs in enumerate(B_shards):
        if contraction_mode == "dense":
            part = df_rhf_fock_build_BLAS(Bs, C_occ)
        else:
            part = df_rhf_fock_build_screened(Bs, C_occ, sd)
        if s == 0:
            part = part + H
        F += part
    return F


def fock_from_definition(B: np.ndarray, C_occ: np.ndarray, H: np.ndarray) -> np.ndarray:
    """Independent check: F = H + sum_Q [2 B_Q tr(B_Q D~) - B_Q D~ B_Q]."""
    D = C_occ @ C_occ.T
    tr = np.einsum("qmn,mn->q", B, D)
    J = np.einsum("q,qmn->mn", tr, B)
    K = np.einsum("qma,ab,qbn->mn", B, D, B, optimize=True)
    return H + 2.0 * J - K
