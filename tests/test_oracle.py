"""CPU tests of the oracle itself: the numpy restatement of the dense path, the
packed/screened path, the sharded sum, the plain-C twin and the definition of
the DF Fock matrix must all agree (SURVEY.md 8c "(i) algebraically")."""
import ctypes

import numpy as np
import pytest

import juliachem_jl_amd as jc
from juliachem_jl_amd import synthetic
from oracle import df_fock as orc


def _inputs(N, Q, o, kept=None, seed=7):
    s = synthetic.make(N, Q, o, seed=seed, kept_fraction=kept)
    B = orc.calculate_B(s.J2c, s.T)
    return s, B


@pytest.mark.parametrize("N,Q,o", [(7, 11, 2), (25, 96, 5), (40, 60, 13)])
def test_dense_equals_definition(N, Q, o):
    s, B = _inputs(N, Q, o)
    Co = s.C[:, :o]
    F = s.H + orc.df_rhf_fock_build_BLAS(B, Co)
    F2 = orc.fock_from_definition(B, Co, s.H)
    assert np.allclose(F, F2, rtol=0, atol=1e-11 * np.abs(F2).max())
    assert np.allclose(F, F.T, atol=1e-11 * np.abs(F).max())


@pytest.mark.parametrize("N,Q,o,nb", [(25, 30, 5, 10), (107, 40, 9, 10), (130, 33, 12, 4)])
def test_screened_equals_dense_with_zeros(N, Q, o, nb):
    """Packed path on a screened tensor == dense path on the same tensor with
    zeros at screened pairs (what the HIP layout stores)."""
    s, B = _inputs(N, Q, o, kept=0.5)
    sd = orc.get_screening_metadata(s.mask)
    Bp = orc.pack_three_center(B, sd)
    Co = s.C[:, :o]
    Fs = orc.df_rhf_fock_build_screened(Bp, Co, sd, n_blocks=nb)
    Fd = orc.df_rhf_fock_build_BLAS(B, Co)
    assert np.allclose(Fs, Fd, rtol=0, atol=1e-11 * np.abs(Fd).max())


def test_unscreened_map_is_dense_layout():
    sd = orc.setup_unscreened_screening_matricies(6)
    q, p = 4, 2
    assert sd.sparse_pq_index_map[q, p] == q + 6 * p          # SchwarzScreening.jl:97-111
    assert sd.screened_indices_count == 36
    assert np.all(sd.non_screened_p_indices_count == 6)


def test_packing_order_outer_p_inner_q():
    m = np.array([[1, 1, 0], [1, 1, 1], [0, 1, 1]], dtype=bool)
    sd = orc.get_screening_metadata(m)
    # p = 0: q = 0,1 -> 0,1 ; p = 1: q = 0,1,2 -> 2,3,4 ; p = 2: q = 1,2 -> 5,6
    assert sd.sparse_pq_index_map.tolist() == [[0, 2, -1], [1, 3, 5], [-1, 4, 6]]
    assert sd.sparse_p_start_indices.tolist() == [0, 2, 5]
    assert sd.non_screened_p_indices_count.tolist() == [2, 3, 2]
    assert [list(r) for r in sd.non_zero_ranges[1]] == [[0, 1, 2]]


@pytest.mark.parametrize("n_shards", [1, 2, 3, 8])
def test_shard_sum_invariance(n_shards):
    s = synthetic.make(30, 64, 6, seed=3)
    offs = orc.shard_offsets(s.aux_shell_nbas, n_shards)
    assert offs[0] == 0 and offs[-1] == 64 and np.all(np.diff(offs) >= 0)
    Bfull = orc.calculate_B(s.J2c, s.T)
    shards = [orc.calculate_B(s.J2c, s.T, range(int(offs[r]), int(offs[r + 1]))) for r in range(n_shards)]
    assert np.allclose(np.concatenate(shards, axis=0), Bfull, atol=1e-13)
    F1 = orc.df_rhf_fock_build([Bfull], s.C, 6, s.H)
    Fn = orc.df_rhf_fock_build(shards, s.C, 6, s.H)
    assert np.allclose(F1, Fn, rtol=0, atol=1e-12 * np.abs(F1).max())


def test_shard_rule_matches_reference_example():
    # GPUDF.jl:1011-1020 comment: 16 devices, A = 63 one-function shells -> 1:4, ..., 61:63 (1-based)
    nb = [1] * 63
    offs = orc.shard_offsets(nb, 16)
    # floor(63/16) = 3 shells per device, the last takes the remainder
    assert offs[1] - offs[0] == 3 and offs[16] == 63 and offs[15] == 45
    sh, fr = orc.static_load_rank_indicies(15, 16, nb)
    assert (sh.start, sh.stop) == (45, 63) and (fr.start, fr.stop) == (45, 63)


def test_c_twin_matches_numpy(oracle_c):
    N, Q, o = 13, 17, 4
    s, B = _inputs(N, Q, o)
    Co = np.asfortranarray(s.C[:, :o])
    Bf = np.asfortranarray(B)                       # (Q, N, N) column-major, Q fastest
    Hf = np.asfortranarray(s.H)
    F = np.zeros((N, N), order="F")
    rc = oracle_c.jcdf_oracle_fock_dense(N, Q, o, Bf.ctypes.data, Co.ctypes.data, Hf.ctypes.data, 1, F.ctypes.data)
    assert rc == 0
    ref = s.H + orc.df_rhf_fock_build_BLAS(B, Co)
    assert np.allclose(F, ref, rtol=0, atol=1e-12 * np.abs(ref).max())
    # B formation twin
    T = np.asfortranarray(s.T.reshape(Q, N * N))
    J = np.asfortranarray(s.J2c)
    assert oracle_c.jcdf_oracle_form_B(Q, N * N, J.ctypes.data, T.ctypes.data) == 0
    assert np.allclose(T.reshape(Q, N, N), B, rtol=0, atol=1e-12 * np.abs(B).max())


def test_block_screened_exchange_of_the_oracle():
    """oracle.calculate_K_lower_diagonal_block (ScreenedDF.jl:459-545): equals the unscreened block form wherever the
    reference's block screen keeps a block or in the ragged strip, and is zero in the skipped blocks."""
    from juliachem_jl_amd import synthetic
    N, Q, o, nbk = 143, 20, 6, 7
    s = synthetic.make(N, Q, o, seed=4, kept_fraction=0.2)
    sd = orc.get_screening_metadata(s.mask)
    Bp = orc.pack_three_center(orc.calculate_B(s.J2c, s.T), sd)
    W = orc.calculate_W_screened(Bp, np.ascontiguousarray(s.C[:, :o].T), sd)
    full = orc.calculate_K_lower_diagonal_block_no_screen(W, nbk)
    scr = orc.calculate_K_lower_diagonal_block(W, sd, nbk)
    bw, nb, bs = orc.exchange_block_screen(sd.basis_function_screen_matrix, nbk, True)
    assert (bw, nb) == (N // nbk, nbk) and not bs[np.triu_indices(nb, 1)].any() and bs.diagonal().all()
    assert (~bs[np.tril_indices(nb)]).sum() > 0
    for pp in range(nb):
        for qq in range(pp + 1):
            blk = (slice(pp * bw, (pp + 1) * bw), slice(qq * bw, (qq + 1) * bw))
            if bs[pp, qq]:
                assert np.array_equal(scr[blk], full[blk])
            else:
                assert not scr[blk].any() and not scr[blk[::-1]].any()
    assert N % nbk != 0 and np.array_equal(scr[:, nb * bw:], full[:, nb * bw:]) and np.array_equal(scr, scr.T)
    # one block below N = 100 (ScreenedDF.jl:392-394): nothing can be screened
    assert orc.exchange_block_screen(np.eye(50, dtype=bool), 10, True)[1] == 1
