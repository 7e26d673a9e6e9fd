"""world_size-2 (and 3) CPU tests of the N > 1 path with the gloo backend: the aux
shard rule, the one-time B-formation exchange (point-to-point, block s only to the ranks
behind it: the lower triangle; bytes per rank asserted) and the single N x N all-reduce, driven through the SAME helper
functions the GPU engine uses (juliachem_jl_amd.engine), with the oracle standing
in for the per-shard device arithmetic (no GPU in this container)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import juliachem_jl_amd  # noqa: F401  (import shim)
from juliachem_jl_amd import synthetic
from juliachem_jl_amd.engine import allreduce_fock, exchange_three_center_blocks, shard_ranges
from oracle import df_fock as orc


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, N, Q, o, out, block=None):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        s = synthetic.make(N, Q, o, seed=31)
        ranges = shard_ranges(s.aux_shell_nbas, world)
        rows = ranges[rank]
        Linv = orc.form_J_AB_inv(s.J2c)
        T = s.T.reshape(Q, N * N)
        B_local = np.zeros((len(rows), N * N))
        pushed = []

        def push(s0, s1, blk):                       # what jcdf_push_three_center_device does on the GPU
            if s0 >= rows.stop:
                return                               # Linv[rows, s0:s1] == 0
            pushed.append((s0, s1))
            Tb = blk.numpy().reshape(N * N, s1 - s0).T          # the device layout: (rows, P) column-major
            B_local[:] += Linv[rows.start:rows.stop, s0:s1] @ Tb

        T_own = torch.from_numpy(np.ascontiguousarray(T[rows.start:rows.stop].T).reshape(-1))
        stats = {}
        allocs = []

        def alloc(n):
            allocs.append(n)
            return torch.empty(n, dtype=torch.float64)
        exchange_three_center_blocks(ranges, rank, world, dist, T_own, alloc, push, block=block, stats=stats)
        # the lower triangle of the exchange: rank r receives exactly the blocks s < r (in pieces of <= block rows), sends its
        # own block to the ranks behind it, holds ONE receive buffer of one piece
        P = N * N
        step = lambda r: min(len(r), block) if block else len(r)
        assert stats["received"] == P * sum(len(r) for r in ranges[:rank])
        assert stats["sent"] == P * len(rows) * (world - 1 - rank)
        assert len(allocs) == (1 if rank > 0 else 0) and stats["recv_buffer"] == (P * max(step(r) for r in ranges) if rank > 0 else 0)
        want = []
        for r in ranges[:rank]:
            want += [(a, min(r.stop, a + step(r))) for a in range(r.start, r.stop, step(r))]
        assert sorted(pushed) == sorted(want + [(rows.start, rows.stop)])
        Bref = orc.calculate_B(s.J2c, s.T, rows).reshape(len(rows), N * N)
        assert np.allclose(B_local, Bref, rtol=0, atol=1e-12 * np.abs(Bref).max())
        part = orc.df_rhf_fock_build_BLAS(B_local.reshape(len(rows), N, N), s.C[:, :o])
        if rank == 0:
            part = part + s.H                        # H on rank 0 only (GPUDF.jl:221-225)
        F = allreduce_fock(torch.from_numpy(part.copy()), world, dist).numpy()
        ref = orc.df_rhf_fock_build([orc.calculate_B(s.J2c, s.T)], s.C, o, s.H)
        assert np.allclose(F, ref, rtol=0, atol=1e-11 * np.abs(ref).max())
        out.put((rank, float(np.abs(F - ref).max())))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,block", [(2, None), (3, None), (3, 5)])
def test_sharded_fock_build_gloo(world, block):
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, 23, 40, 4, out, block)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    got = sorted(out.get(timeout=5)[0] for _ in range(world))
    assert got == list(range(world))


def test_shard_ranges_cover_and_follow_reference_rule():
    nb = [1, 3, 6, 10, 1, 3, 6, 1, 1]                 # 9 aux shells, 32 functions
    r = shard_ranges(nb, 4)                           # 9 // 4 = 2 shells each, the last takes 3
    assert [(x.start, x.stop) for x in r] == [(0, 4), (4, 20), (20, 24), (24, 32)]
    assert [(x.start, x.stop) for x in shard_ranges(nb, 1)] == [(0, 32)]
