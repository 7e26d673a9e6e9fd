"""Device-resident driver around the HIP Fock build: one process per GPU,
`torch.distributed` (backend "nccl" == RCCL over xGMI) for the one collective the
path has — the all-reduce of the N x N partial Fock matrix that replaces
MPI.Allreduce! (DensityFitting.jl:68-71) and the host axpy over devices
(GPUDF.jl:267-277).  torch is plumbing here (device memory, streams, the
collective, the replicated eigensolve); every kernel of the Fock build is in
libjcdf_hip.so.

`DeviceFockBuilder`   aux-sharded Fock build, inputs/outputs stay in HBM.
`DeviceSCF`           the SCF loop body of scf_cycles_kernel (SCF.jl:399-573) and
                      `iteration` (SCF.jl:1072-1125) with everything on the device:
                      the caller of the hot path.
"""
from __future__ import annotations

import ctypes
import math
import os
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from .df import JCDFGroup, JCDFHandle, lapack_potrf_trtri
from .eigh import DeviceEigh


def _dist():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size(), dist
    return 0, 1, None


def shard_ranges(aux_shell_nbas: Sequence[int], n_shards: int) -> List[range]:
    """Contiguous aux-shell shards, floor(S/n) shells each, remainder to the last
    (DynamicLoad.jl:160-203)."""
    pos = np.concatenate([[0], np.cumsum(np.asarray(aux_shell_nbas, dtype=np.int64))])
    S = len(aux_shell_nbas)
    per = S // n_shards
    out = []
    for r in range(n_shards):
        b = per * r
        e = S if r == n_shards - 1 else b + per
        out.append(range(int(pos[b]), int(pos[e])))
    return out


def _staged(dist, t) -> bool:
    """gloo cannot move device tensors: stage through the host (test/rehearsal path only;
    the production backend is nccl == RCCL, which works on the device buffers directly)."""
    return t.is_cuda and dist.get_backend() == "gloo"


def _broadcast(dist, t, src: int) -> None:
    if _staged(dist, t):
        tmp = t.cpu()
        dist.broadcast(tmp, src)
        t.copy_(tmp)
    else:
        dist.broadcast(t, src)


def _all_reduce(dist, t, op=None) -> None:
    kw = {} if op is None else {"op": op}
    if _staged(dist, t):
        tmp = t.cpu()
        dist.all_reduce(tmp, **kw)
        t.copy_(tmp)
    else:
        dist.all_reduce(t, **kw)


def _p2p(dist, ops) -> None:
    """a batch of sends / receives ((kind, tensor, peer) triples) as ONE group (RCCL: ncclGroupStart/End, so the transfers
    to different peers run over their own xGMI links at the same time); device tensors under gloo are staged through the host"""
    if not ops:
        return
    staged = []
    reqs = []
    for kind, t, peer in ops:
        buf = t
        if _staged(dist, t):
            buf = t.cpu() if kind == "send" else torch.empty(t.shape, dtype=t.dtype)
            if kind == "recv":
                staged.append((t, buf))
        reqs.append(dist.P2POp(dist.isend if kind == "send" else dist.irecv, buf, peer))
    for w in dist.batch_isend_irecv(reqs):
        w.wait()
    for t, buf in staged:
        t.copy_(buf)


def exchange_three_center_blocks(ranges: Sequence[range], rank: int, world: int, dist, T_own, alloc, push,
                                 block: Optional[int] = None, stats: Optional[dict] = None) -> None:
    """One-time B formation across ranks (reference: GPUDF.jl:918-997, host-staged MPI.Send/Recv! of every block to every
    rank).  Every rank owns the three-centre integrals of its own aux rows, `T_own` = the (rows, P) column-major block
    (flat: aux index fastest).  B[rows_r] = sum_s L^-1[rows_r, rows_s] T[rows_s] and L^-1 is lower triangular, so block s
    travels ONLY to the ranks r > s — point-to-point, the lower triangle of the all-to-all — and is offered to
    `push(s0, s1, chunk)` there; rank s pushes its own block itself.  `block`: at most this many aux rows travel (and are
    held in the receive buffer) at a time; `alloc(n)` returns a receive buffer of n doubles, allocated once.
    `stats` (optional dict) counts doubles sent / received and the largest receive buffer.
    Backend-agnostic (RCCL on device tensors, gloo on CPU tensors in the tests)."""
    R_own = len(ranges[rank])
    P = T_own.numel() // R_own
    if stats is not None:
        stats.update(sent=0, received=0, recv_buffer=0)
    step_max = max((min(len(r), block) if block else len(r)) for r in ranges)
    buf = alloc(step_max * P) if (world > 1 and rank > 0) else None
    if stats is not None and buf is not None:
        stats["recv_buffer"] = int(buf.numel())
    for s, rows in enumerate(ranges):
        R = len(rows)
        if s == rank:
            push(rows.start, rows.stop, T_own)
        if world == 1 or rank < s:
            continue
        step = min(R, block) if block else R
        for a0 in range(0, R, step):
            a1 = min(R, a0 + step)
            if rank == s:
                if s == world - 1:
                    break                                            # the last block has no receiver
                chunk = T_own if (a0 == 0 and a1 == R) else T_own.view(P, R)[:, a0:a1].contiguous().view(-1)
                _p2p(dist, [("send", chunk, r) for r in range(s + 1, world)])
                if stats is not None:
                    stats["sent"] += int(chunk.numel()) * (world - 1 - s)
            else:
                part = buf[:(a1 - a0) * P]
                _p2p(dist, [("recv", part, s)])
                push(rows.start + a0, rows.start + a1, part)
                if stats is not None:
                    stats["received"] += int(part.numel())


def allreduce_fock(F, world: int, dist):
    """Sum of the per-shard partial Fock matrices: MPI.Allreduce!(two_electron_fock)
    of DensityFitting.jl:68-71 as ONE collective on the N x N buffer."""
    if world > 1:
        _all_reduce(dist, F)
    return F


class DeviceFockBuilder:
    """F = H + sum_shards (2 J_s - K_s) with this rank's shard on this rank's GPU."""

    def __init__(self, N: int, Q_total: int, n_occ: int, aux_shell_nbas: Sequence[int],
                 device: Optional[int] = None, pq: Tuple[Optional[np.ndarray], Optional[np.ndarray]] = (None, None),
                 exchange_screen_blocks: int = 0, tuning: Optional[dict] = None):
        self.rank, self.world, self.dist = _dist()
        if device is None:
            device = int(os.environ.get("LOCAL_RANK", "0"))
        self.device = torch.device("cuda", device)
        torch.cuda.set_device(self.device)
        self.N, self.Q_total, self.n_occ = N, Q_total, n_occ
        self.ranges = shard_ranges(aux_shell_nbas, self.world)
        self.rows = self.ranges[self.rank]
        if len(self.rows) == 0:
            raise ValueError("empty auxiliary shard on rank %d" % self.rank)
        self.h = JCDFHandle(device)
        # every library operation goes on torch's current stream: ordered with the torch ops
        # that produce its inputs (C_occ, the T blocks) and consume its output (F)
        self.h.set_stream(torch.cuda.current_stream(self.device).cuda_stream)
        self.h.set_exchange_screening(exchange_screen_blocks)      # df_exchange_screen (ScreenedDF.jl:431-447); 0 = off
        for key, value in (tuning or {}).items():                  # jcdf_set_tuning keys (include/jcdf.h)
            self.h.set_tuning(key, value)
        self.h.configure(N, Q_total, self.rows.start, self.rows.stop, n_occ, pq[0], pq[1])
        self.F = torch.zeros((N, N), dtype=torch.float64, device=self.device)
        self.time_collectives = False      # bench: device events around the broadcast of C and the all-reduce of F
        self.collective_events: list = []

    # ---- setup -----------------------------------------------------------------
    def set_metric(self, J2c: np.ndarray) -> None:
        self.h.set_metric(J2c)   # device potrf/trtri (DenseGPUDF.jl:185-193; host LAPACK at GPUDF.jl:890-891)

    def set_core_hamiltonian(self, H: np.ndarray) -> None:
        self.h.set_core_hamiltonian(H if self.rank == 0 else None)     # GPUDF.jl:158-161

    def push_three_center_device(self, s0: int, s1: int, T_dev: torch.Tensor) -> None:
        """T_dev: device tensor holding the (s1-s0, P) column-major block."""
        if s0 < self.rows.stop:
            self.h.push_three_center_device(s0, s1, T_dev.data_ptr())

    def exchange_three_center(self, T_own: torch.Tensor, block: Optional[int] = None) -> dict:
        """One-time B formation across ranks (GPUDF.jl:918-997): every rank owns the three-centre integrals of its own aux
        rows; block s goes point-to-point over RCCL to the ranks r > s only (L^-1 is lower triangular) in pieces of at most
        `block` aux rows (default: 2 GiB of receive buffer) and is accumulated there.  Returns the traffic counters."""
        P = T_own.numel() // len(self.rows)
        if block is None:
            block = max(16, ((1 << 28) // max(P, 1)) // 16 * 16)
        stats: dict = {}
        exchange_three_center_blocks(
            self.ranges, self.rank, self.world, self.dist, T_own,
            lambda n: torch.empty(n, dtype=torch.float64, device=self.device), self.push_three_center_device,
            block=block, stats=stats)
        torch.cuda.synchronize(self.device)
        return stats

    # ---- per iteration -----------------------------------------------------------
    def build(self, C_occ_dev: torch.Tensor) -> torch.Tensor:
        """C_occ_dev: (n_occ, N) row-major device tensor == (N, n_occ) column-major,
        the layout of DensityFitting.jl:49.  Returns the reduced F (device).
        With more than one rank the coefficients are first broadcast from rank 0 — MPI.Bcast!(C, 0) of SCF.jl:462 —
        so that every shard is built from the same bits even if a rank's replicated eigensolve took another route."""
        if self.world > 1:
            ev = self._collective_begin()
            _broadcast(self.dist, C_occ_dev, 0)
            self._collective_end(ev, "bcast")
        self.h.fock_build_device(C_occ_dev.data_ptr(), self.F.data_ptr())
        ev = self._collective_begin()
        out = allreduce_fock(self.F, self.world, self.dist)    # RCCL ncclAllReduce(N^2 fp64) over xGMI
        self._collective_end(ev, "allreduce")
        return out

    def build_ld(self, C_pad: torch.Tensor, F_pad: torch.Tensor) -> torch.Tensor:
        """The same for a caller that keeps its matrices zero padded: C_pad (>= n_occ rows, row stride ldc >= N: orbital i in
        row i), F_pad (square, row stride ldf >= N): this rank's partial Fock matrix is written into the N x N corner of
        F_pad and all-reduced there (the padding is zero on every rank and stays zero) — no repacking launches."""
        if self.world > 1:
            ev = self._collective_begin()
            _broadcast(self.dist, C_pad, 0)
            self._collective_end(ev, "bcast")
        self.h.fock_build_device_ld(C_pad.data_ptr(), C_pad.stride(0), F_pad.data_ptr(), F_pad.stride(0))
        ev = self._collective_begin()
        out = allreduce_fock(F_pad, self.world, self.dist)
        self._collective_end(ev, "allreduce")
        return out

    def _collective_begin(self):
        if not (self.time_collectives and self.world > 1):
            return None
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(torch.cuda.current_stream(self.device))
        return ev

    def _collective_end(self, ev, tag: str = "allreduce") -> None:
        if ev is not None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record(torch.cuda.current_stream(self.device))
            self.collective_events.append((tag, ev, e1))

    def collective_ms(self, split: bool = False):
        """sum of the recorded collective durations (ms); clears the record.  Call after a device synchronise.
        split = True: {"bcast": ms, "allreduce": ms} (the broadcast of C and the all-reduce of F separately)."""
        parts = {"bcast": 0.0, "allreduce": 0.0}
        for tag, a, b in self.collective_events:
            parts[tag] = parts.get(tag, 0.0) + float(a.elapsed_time(b))
        self.collective_events = []
        return parts if split else float(sum(parts.values()))

    def close(self) -> None:
        self.h.close()


class GroupFockBuilder:
    """The same interface as `DeviceFockBuilder` for ONE process that drives several GPUs (include/jcdf.h, jcdf_group_*):
    member i of the group holds aux shard i on devices[i]; the SCF loop (DIIS, eigensolve, density: `DeviceSCF`) lives on
    devices[0] only — nothing is replicated — and every Fock build is sharded over all devices: C_occ is fetched by the
    other devices device-to-device, the partial Fock matrices are summed on the devices (RCCL reduce / peer-mapped slice
    sums over xGMI) and gathered where the caller's F lives (`jcdf_group_fock_build_device_ld`).  From torch's point of
    view this is a single-rank job: rank 0, world 1, no process group."""

    def __init__(self, N: int, Q_total: int, n_occ: int, aux_shell_nbas: Sequence[int], devices: Sequence[int],
                 pq: Tuple[Optional[np.ndarray], Optional[np.ndarray]] = (None, None), exchange_screen_blocks: int = 0,
                 tuning: Optional[dict] = None, transport: Optional[str] = None):
        self.rank, self.world, self.dist = 0, 1, None
        self.devices = [int(d) for d in devices]
        self.device = torch.device("cuda", self.devices[0])
        torch.cuda.set_device(self.device)
        self.N, self.Q_total, self.n_occ = N, Q_total, n_occ
        self.ranges = shard_ranges(aux_shell_nbas, len(self.devices))
        if any(len(r) == 0 for r in self.ranges):
            raise ValueError("empty auxiliary shard: more devices than auxiliary shells")
        self.rows = range(0, Q_total)                      # this process holds every shard
        self.g = JCDFGroup(self.devices)
        if transport:
            self.g.set_transport(transport)
        self.g.set_exchange_screening(exchange_screen_blocks)
        for key, value in (tuning or {}).items():
            for m in self.g.members:
                m.set_tuning(key, value)
        self.g.configure(N, Q_total, [r.start for r in self.ranges] + [Q_total], n_occ, pq[0], pq[1])
        self.h = self.g.members[0]                          # kernel statistics / overlap switch of member 0 (its shard)
        # member 0 works on torch's current stream, like DeviceFockBuilder's handle: no event hop between the SCF kernels and
        # its shard of the build; the other members keep their own streams (ordered by the group's events)
        self.h.set_stream(torch.cuda.current_stream(self.device).cuda_stream)
        self.F = torch.zeros((N, N), dtype=torch.float64, device=self.device)
        self.time_collectives = False
        self.collective_events: list = []

    def set_metric(self, J2c: np.ndarray) -> None:
        self.g.set_metric(J2c)

    def set_core_hamiltonian(self, H: np.ndarray) -> None:
        self.g.set_core_hamiltonian(H)

    def push_three_center_device(self, s0: int, s1: int, T_dev: torch.Tensor) -> None:
        """T_dev: device tensor (any device of this process) holding the (s1-s0, P) column-major block"""
        torch.cuda.synchronize(T_dev.device)
        self.g.push_three_center_device(s0, s1, T_dev.data_ptr())

    def exchange_three_center(self, T_own: torch.Tensor, block: Optional[int] = None) -> dict:
        """every shard's block is already in this process: pushed shard by shard (the library uploads / fetches each block
        once and hands it to the members behind it device-to-device) — no inter-process traffic"""
        P = T_own.numel() // self.Q_total
        Tm = T_own.view(P, self.Q_total)                    # (rows, P) column-major == [c][row]
        for r in self.ranges:
            self.push_three_center_device(r.start, r.stop, Tm[:, r.start:r.stop].contiguous().view(-1))
        return {"sent": 0, "received": 0, "recv_buffer": 0}

    def build(self, C_occ_dev: torch.Tensor) -> torch.Tensor:
        st = torch.cuda.current_stream(self.device).cuda_stream
        self.g.fock_build_device_ld(C_occ_dev.data_ptr(), C_occ_dev.stride(0), self.F.data_ptr(), self.N, st)
        return self.F

    def build_ld(self, C_pad: torch.Tensor, F_pad: torch.Tensor) -> torch.Tensor:
        st = torch.cuda.current_stream(self.device).cuda_stream
        self.g.fock_build_device_ld(C_pad.data_ptr(), C_pad.stride(0), F_pad.data_ptr(), F_pad.stride(0), st)
        return F_pad

    def collective_ms(self, split: bool = False):
        return {"bcast": 0.0, "allreduce": 0.0} if split else 0.0

    def group_timings(self) -> dict:
        """device-side times of the LAST build (after a synchronise): C fetch, longest member build, reduce, gather"""
        t, gt = self.g.synchronize()
        return {"bcast_ms": gt.bcast_time * 1e3, "build_ms": gt.build_time * 1e3, "reduce_ms": gt.reduce_time * 1e3,
                "gather_ms": gt.d2h_time * 1e3, "member_fock_ms": [x.fock_time * 1e3 for x in t]}

    def close(self) -> None:
        self.g.close()


class DeviceSCF:
    """SCF iteration with all matrices on the device (hcore guess, DIIS, damping).  Matrices are
    symmetric, so row/column-major coincide.  Every dense product of an iteration runs on the library's own fp64 MFMA
    cores (`jcdf_gemm_tn_device` / `_nt_`, csrc/jcdf_blas.hpp): the N x N matrices live zero padded in (Np x Np)
    buffers, Np = N rounded up to 32; `F`, `D`, `C` are the unpadded views."""

    def __init__(self, fb: DeviceFockBuilder, H: np.ndarray, S: np.ndarray, E_nuc: float, ndiis: int = 10,
                 density_solver: Optional[str] = None):
        """density_solver: "eigh" (default; the reference's eigen() per iteration) or "sp2": the occupied-space
        projector by spectral projection (eigh.DeviceSP2) and an orthonormal basis of it from the previous
        iteration's occupied orbitals — same density, energy and DIIS error; orbital energies and canonical
        orbitals only after canonical_orbitals().  $JCDF_DENSITY_SOLVER overrides."""
        self.fb = fb
        dev = fb.device
        self.N, self.n_occ = fb.N, fb.n_occ
        self.Np = (self.N + 31) // 32 * 32
        self.op = (self.n_occ + 31) // 32 * 32
        from . import _lib
        self._lib = _lib.load()
        self._f64 = dict(dtype=torch.float64, device=dev)
        self.Hp = self._padded(H)
        self.Sp = self._padded(S)
        self.eigh = DeviceEigh(self.N, dev)      # persistent-kernel tridiagonalisation + divide & conquer + one GEMM / compact-WY
        # X = U s^-1/2 U^T (SCF.jl:142-162) from the library's own eigensolver as well (round 3: torch.linalg.eigh here);
        # a failure of it is counted like any other (solver_report) and the vendor routine takes this one decomposition
        s, U = self.eigh(self.Sp[:self.N, :self.N])
        if not self.eigh.check():
            s, U = torch.linalg.eigh(self.Sp[:self.N, :self.N])
        s, U = s.clone(), U.clone()
        keep = s >= 1.0e-6                                          # SCF.jl:142-162
        self.Xp = self._padded((U[:, keep] * s[keep].rsqrt()) @ U[:, keep].T)
        self.E_nuc = E_nuc
        if not 0 <= int(ndiis) <= 15:
            # the DIIS history lives in device ring buffers and the bordered Pulay system is solved by one workgroup
            # (jcdf_diis_device: at most 15 vectors); the reference's default is 10 (SCF.jl:364)
            raise ValueError("ndiis = %r: the device DIIS keeps 0..15 error vectors" % (ndiis,))
        self.ndiis = int(ndiis)
        self.density_solver = (os.environ.get("JCDF_DENSITY_SOLVER") or density_solver or "eigh").lower()
        if self.density_solver not in ("eigh", "sp2"):
            raise ValueError("density_solver %r: \"eigh\" or \"sp2\"" % self.density_solver)
        self.sp2 = None
        if self.density_solver == "sp2" and 0 < self.n_occ < self.N and 2 <= self.N <= 4096:     # outside: eigensolver
            from .eigh import DeviceLowdin, DeviceSP2
            self.sp2 = DeviceSP2(self.N, self.n_occ, dev)
            self.lowdin = DeviceLowdin(self.n_occ, self.N, dev)
            self.Cpt = torch.zeros((self.op, self.Np), **self._f64)     # occupied orbitals in the orthogonal basis (rows), zero padded
            self.Yt = torch.zeros((self.op, self.Np), **self._f64)
            # the last matrix X F X that was diagonalised and four of its eigenvalues {lowest, HOMO, LUMO, highest}: spectral bounds
            # (Weyl) and the gap bracket of the accelerated projection for the iterations that follow (jcdf_sp2_ref_device)
            self.Fref = torch.zeros((self.Np, self.Np), **self._f64)
            self.ref_eigs = torch.zeros(4, **self._f64)
            self.ref_idx = torch.tensor([0, self.n_occ - 1, self.n_occ, self.N - 1], dtype=torch.int64, device=dev)
            self.have_ref = False
        self.sp2_pivot = None
        self.tail_work = torch.zeros(256, **self._f64)
        self.tail_out = torch.zeros(8, **self._f64)
        self.sp2_steps = self.sp2_fallbacks = self.sp2_basis_retries = self.sp2_refreshes = 0
        self.sp2_refresh = False                 # next step: one eigensolve to renew the reference decomposition of the projections
        self.sp2_refresh_below = 0.05            # ... asked for when the density change falls below this while the recursion is not accelerated
        self.sp2_accelerated = 0                 # projections that ran the accelerated recursion
        self.sp2_reasons = {}
        self.sp2_skip = True                     # first step, and while the density still changes wholesale: eigensolver
        self.canonical = True                    # self.C / self.eps are the eigenvectors / eigenvalues of self.F
        self.diis_on_host = bool(os.environ.get("JCDF_DIIS_HOST"))   # debug: Pulay system solved by numpy (one more sync per iteration)
        # work matrices of the products (padded, zero outside N x N by construction of their factors)
        self.T1 = torch.zeros((self.Np, self.Np), **self._f64)
        self.T2 = torch.zeros((self.Np, self.Np), **self._f64)
        self.Fpr = torch.zeros((self.Np, self.Np), **self._f64)     # X F X
        self.Up = torch.zeros((self.Np, self.Np), **self._f64)      # eigenvectors of X F X (columns), padded
        self.Ctp = torch.zeros((self.Np, self.Np), **self._f64)     # (X U)^T: row i = orbital i in the AO basis
        self.Cop = torch.zeros((self.op, self.Np), **self._f64)     # occupied rows of it, zero rows up to a multiple of 32
        # F and D live in two buffers each, used in turn: the buffer that is not written this iteration still holds last
        # iteration's matrix — F_old of the damping step and D_old of the convergence test — so neither is ever copied
        self.Fbuf = [torch.zeros((self.Np, self.Np), **self._f64) for _ in range(2)]
        self.Dbuf = [torch.zeros((self.Np, self.Np), **self._f64) for _ in range(2)]
        self.fi = self.di = 0
        self.eigh.scratch = True                  # X F X is a work matrix: tridiagonalised in place
        self.reset()

    # ---- plumbing --------------------------------------------------------------------------------------------
    def _padded(self, M) -> torch.Tensor:
        out = torch.zeros((self.Np, self.Np), **self._f64)
        out[:self.N, :self.N].copy_(torch.as_tensor(M, **self._f64))
        return out

    def _st(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.Hp.device).cuda_stream)

    def _gemm_tn(self, A: torch.Tensor, B: torch.Tensor, out: torch.Tensor, alpha: float = 1.0) -> torch.Tensor:
        """out[m][n] = alpha sum_k A[k][m] B[k][n] on the library's MFMA core (A symmetric: out = alpha A B)."""
        K, M = A.shape
        N = B.shape[1]
        rc = self._lib.jcdf_gemm_tn_device(self._st(), M, N, K, alpha, ctypes.c_void_p(A.data_ptr()), A.stride(0),
                                           ctypes.c_void_p(B.data_ptr()), B.stride(0), ctypes.c_void_p(out.data_ptr()), out.stride(0))
        if rc != 0:
            raise RuntimeError("jcdf_gemm_tn_device failed (status %d)" % rc)
        return out

    def _gemm_nt(self, A: torch.Tensor, B: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
        """out[m][n] = sum_k A[m][k] B[n][k] on the library's MFMA core (B symmetric: out = A B)."""
        M, K = A.shape
        Nn = B.shape[0]
        rc = self._lib.jcdf_gemm_nt_device(self._st(), M, Nn, K, ctypes.c_void_p(A.data_ptr()), A.stride(0), ctypes.c_void_p(B.data_ptr()),
                                           B.stride(0), ctypes.c_void_p(out.data_ptr()), out.stride(0))
        if rc != 0:
            raise RuntimeError("jcdf_gemm_nt_device failed (status %d)" % rc)
        return out

    @property
    def F(self) -> torch.Tensor:
        return self.Fp_[:self.N, :self.N]

    @property
    def D(self) -> torch.Tensor:
        return self.Dp[:self.N, :self.N]

    @property
    def C(self) -> torch.Tensor:
        """MO coefficients, columns = orbitals (view of the transposed buffer the products leave)"""
        return self.Ctp[:self.N, :self.N].t()

    @property
    def H(self) -> torch.Tensor:
        return self.Hp[:self.N, :self.N]

    @property
    def S(self) -> torch.Tensor:
        return self.Sp[:self.N, :self.N]

    @property
    def X(self) -> torch.Tensor:
        return self.Xp[:self.N, :self.N]

    @property
    def Fp_(self) -> torch.Tensor:
        return self.Fbuf[self.fi]

    @property
    def Dp(self) -> torch.Tensor:
        return self.Dbuf[self.di]

    @property
    def Co_t(self) -> torch.Tensor:
        """(n_occ, N) contiguous: the occupied orbitals as the plain Fock-build entry takes them"""
        return self.Cop[:self.n_occ, :self.N].contiguous()

    def reset(self) -> None:
        self.sp2_skip = True
        self.fi = self.di = 0
        self.Fbuf[0].copy_(self.Hp)
        self.Fbuf[1].copy_(self.Hp)                                # F_old of the first iteration (SCF.jl:186)
        self.Dbuf[0].zero_()
        self.Dbuf[1].zero_()
        self._checked_diag()                                       # "iteration 0", SCF.jl:178-181
        self.E_old, self.dE, self.B_dim, self.iter = 0.0, 1.0, 1, 1
        # DIIS history: ring buffers on the device (slot of the newest entry = head); the small
        # Pulay matrix is solved on the device and gets ONE new row of dot products per iteration
        nd = max(self.ndiis, 1)
        self.e_hist = torch.zeros((nd, self.N * self.N), **self._f64)
        self.F_hist = torch.zeros((nd, self.N * self.N), **self._f64)
        self.head, self.n_hist = -1, 0
        self.Bmat = np.zeros((nd, nd))
        self.Bmat_d = torch.zeros((nd, nd), **self._f64)
        self.coef_d = torch.zeros(nd, **self._f64)
        self.dots_d = torch.zeros(nd, **self._f64)
        self.dots_work = torch.zeros(64 * nd, **self._f64)
        self.diis_flag = torch.zeros(1, dtype=torch.int32, device=self.Hp.device)
        self.trail: List[Tuple[int, float, float, float]] = []

    def _diag(self, use_sp2: bool = False) -> None:
        """SCF.jl:1072-1125: F' = X F X, eigh, C = X U, D = 2 C_o C_o^T (the energy: _tail)."""
        N, o = self.N, self.n_occ
        self._gemm_tn(self.Fp_, self.Xp, self.T1)                   # F X      (F symmetric)
        self._gemm_tn(self.Xp, self.T1, self.Fpr)                   # X (F X)
        if use_sp2:
            # occupied-space projector P of F' by spectral projection; orthonormal basis of its range from the previous
            # occupied orbitals Cp (orthogonal basis): Y = P Cp, Cp_new = Y (Y^T Y)^-1/2 (Loewdin), so that
            # Cp_new Cp_new^T = P exactly when P is a projector and Y has full rank (the Newton-Schulz iteration for the
            # inverse square root converges then and only then: its status word goes into the iteration's record).
            # Every product runs on the library's MFMA cores (jcdf_gemm_nt_device, jcdf_lowdin_rows_device).
            self.sp2(self.Fpr[:N, :N], ref=(self.Fref[:N, :N], self.ref_eigs) if self.have_ref else None)    # -> self.sp2.Pp (Np x Np, zero padded)
            self._gemm_nt(self.Cpt, self.sp2.Pp, self.Yt)           # (o, N) = (P Cp)^T   (P symmetric)
            self._sp2_basis()
            return
        else:
            if self.sp2 is not None:
                self.Fref.copy_(self.Fpr)                           # the reference of the projections that follow (Fpr itself is consumed)
            try:
                self.eps, U = self.eigh(self.Fpr[:N, :N])           # (destroys Fpr: eigh.scratch)
            except RuntimeError:
                # a stage of the library failed after X F X had been overwritten: rebuild it; the eigensolver has switched to
                # the vendor routine (counted in solver_report), which leaves its argument alone
                self._gemm_tn(self.Fp_, self.Xp, self.T1)
                self._gemm_tn(self.Xp, self.T1, self.Fpr)
                self.eps, U = self.eigh(self.Fpr[:N, :N])
            Up = self.eigh.U_padded                                 # the library path leaves U zero padded (Np x Np) already
            if Up is None:
                self.Up[:N, :N].copy_(U)
                Up = self.Up
            self._gemm_tn(Up, self.Xp, self.Ctp)                    # (X U)^T[i][m] = sum_k U[k][i] X[k][m]
            if self.sp2 is not None:
                self.Cpt[:o, :N].copy_(Up[:N, :o].t())              # occupied orbitals in the orthogonal basis, (o, N)
                torch.index_select(self.eps, 0, self.ref_idx, out=self.ref_eigs)
                self.have_ref = True
            self.Cop[:o].copy_(self.Ctp[:o])
            self.canonical = True
        self._gemm_tn(self.Cop, self.Cop, self.Dp, 2.0)             # D = 2 Co^T Co (zero rows of Cop add nothing)

    def _sp2_basis(self) -> None:
        """second half of the SP2 step: orthonormal basis of range(P) from Y = (P Cp)^T (self.Yt is left untouched, so the
        step can be repeated with more Newton-Schulz steps), orbitals in the AO basis, density."""
        N, o = self.N, self.n_occ
        self.lowdin(self.Yt, self.Cpt)
        self._gemm_nt(self.Cpt, self.Xp, self.Cop)                  # (o, N): rows = occupied orbitals in the AO basis (X symmetric)
        self.sp2_pivot = self.lowdin.info[1:2]                      # Newton-Schulz steps needed, 0 = no convergence
        self.canonical = False
        self._gemm_tn(self.Cop, self.Cop, self.Dp, 2.0)

    def _checked_diag(self) -> None:
        """Eigensolve whose status is read at once (one host sync): used where no scf tail follows — iteration 0 and
        canonical_orbitals().  A hand-off timeout / stedc failure is counted (eigh.fallbacks), reported by
        `solver_report()`, and the step is redone with the vendor solver; all ranks redo together."""
        self._diag(False)
        bad = self.eigh.status()
        if self.fb.world > 1:
            _all_reduce(self.fb.dist, bad, self.fb.dist.ReduceOp.MAX)
        if float(bad.item()) != 0.0:
            self.eigh.check()                                      # flips the failing rank to the vendor solver, counted
            self._diag(False)

    def solver_report(self) -> dict:
        """What the replicated eigensolves of this SCF actually ran on (no silent fallback)."""
        e = self.eigh
        return {"eigensolves": e.calls, "library_path": bool(e.ok), "vendor_fallbacks": e.fallbacks,
                "reason": getattr(e, "reason", None)}

    def _tail(self, D_old: torch.Tensor, use_sp2: bool) -> List[float]:
        """E_elec, ||D - D_old|| and the iteration's status words in one 64-byte record (`jcdf_scf_tail_device`), read
        with ONE device-to-host copy — the only host synchronisation of the iteration.  (The padded buffers are summed
        whole: the padding is zero in all four.)"""
        p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
        err, info = self.eigh.status_tensors()
        rc = self._lib.jcdf_scf_tail_device(self._st(), self.Np, p(self.Dp), p(D_old), p(self.Fp_), p(self.Hp), p(self.diis_flag),
                                            p(err), p(info), p(self.sp2.info) if use_sp2 else None,
                                            p(self.sp2_pivot) if use_sp2 else None, p(self.tail_work), p(self.tail_out))
        if rc != 0:
            raise RuntimeError("jcdf_scf_tail_device failed (status %d)" % rc)
        if self.fb.world > 1:
            # every rank must take the same decisions (convergence, DIIS reset, fallbacks) or the next collective hangs:
            # trouble flags are OR-ed over the ranks, everything else is rank 0's record — ONE small all-reduce (sum): the
            # other ranks contribute zeros to the record part, every rank its own flags
            rec, dist = self.tail_out, self.fb.dist
            buf = torch.zeros(13, dtype=torch.float64, device=rec.device)
            if self.fb.rank == 0:
                buf[:8] = rec
                if use_sp2:
                    buf[11:13] = self.sp2.info[6:8]                  # accelerated?, ||F' - F_ref||: rank 0's, like the rest of the record
            buf[8] = (rec[3] != 0).to(torch.float64)
            if use_sp2:
                # two separate flags: trouble with the PROJECTOR (squarings not finished / trace off / not finite)
                # and trouble with the BASIS only (Newton-Schulz steps too few) — the second has its own, cheaper retry in
                # step(), which every rank must enter together, and must not double the squarings enqueued next time
                proj_bad = ~((rec[4] == 1.0) & ((rec[5] - self.n_occ).abs() < 1e-6) & torch.isfinite(rec[0]))
                buf[9] = proj_bad.to(torch.float64)
                buf[10] = (~(rec[6] >= 1.0)).to(torch.float64)
            _all_reduce(dist, buf)
            rec.copy_(buf[:8])
            rec[3] = buf[8]
            if use_sp2:
                rec[4] = torch.where(buf[9] != 0, torch.zeros_like(rec[4]), rec[4])     # some rank's projector failed: unfinished everywhere
                rec[6] = torch.where(buf[10] != 0, torch.zeros_like(rec[6]), rec[6])    # some rank's basis failed: retried everywhere
                return torch.cat([rec, buf[11:13]]).cpu().tolist()
            return rec.cpu().tolist() + [0.0, 0.0]
        if use_sp2:
            return torch.cat([self.tail_out, self.sp2.info[6:8]]).cpu().tolist()    # + {accelerated, delta} of the projection: still ONE copy
        return self.tail_out.cpu().tolist() + [0.0, 0.0]

    def canonical_orbitals(self) -> None:
        """Eigenvectors / eigenvalues of the current Fock matrix into self.C / self.eps (what the reference has after
        every iteration; with density_solver = "sp2" only on request)."""
        if not self.canonical:
            D = self.Dp.clone()
            self._checked_diag()                                   # sets C, eps (and D, Co_t: the same space)
            self.Dp.copy_(D)

    def _diis_on_host(self, Fp: torch.Tensor) -> None:
        """debug path (JCDF_DIIS_HOST=1): history on the device, the Pulay system solved by numpy (one more sync per iteration)"""
        N, nd = self.N, self.ndiis
        p = lambda t: ctypes.c_void_p(t.data_ptr())
        self._gemm_tn(self.Dp, Fp, self.T1)
        self._gemm_tn(self.Sp, self.T1, self.T2)
        self.head = (self.head + 1) % nd
        self.n_hist = min(self.n_hist + 1, nd)
        rc = self._lib.jcdf_diis_push_device(self._st(), N, self.Np, p(self.T2), p(Fp), p(self.e_hist[self.head]), p(self.F_hist[self.head]))
        rc = rc or self._lib.jcdf_diis_dots_device(self._st(), nd, self.head, N * N, p(self.e_hist), p(self.dots_d), p(self.dots_work))
        if rc != 0:
            raise RuntimeError("DIIS history kernels failed (status %d)" % rc)
        dots = self.dots_d.cpu().numpy()
        self.Bmat[self.head, :] = dots
        self.Bmat[:, self.head] = dots
        if self.iter <= 1:
            return
        self.B_dim = min(self.B_dim + 1, nd)
        n = self.B_dim
        order = [(self.head - k) % nd for k in range(n)]           # newest first, like the reference's vcat
        Bm = -np.ones((n + 1, n + 1))
        Bm[:n, :n] = self.Bmat[np.ix_(order, order)]
        Bm[n, n] = 0.0
        rhs = np.zeros(n + 1)
        rhs[n] = -1.0
        try:
            c = np.linalg.solve(Bm, rhs)[:n]
            if not np.all(np.isfinite(c)):
                raise np.linalg.LinAlgError("non-finite DIIS coefficients")
            cfull = np.zeros(nd)
            cfull[order] = c
            self.coef_d.copy_(torch.as_tensor(cfull, device=self.coef_d.device))
            rc = self._lib.jcdf_diis_mix_device(self._st(), nd, N, self.Np, p(self.F_hist), p(self.coef_d), p(Fp))
            if rc != 0:
                raise RuntimeError("jcdf_diis_mix_device failed (status %d)" % rc)
        except np.linalg.LinAlgError:                              # "Faulty DIIS!" SCF.jl:493-499
            self.B_dim = 2

    profile = False

    def _mark(self, name: str) -> None:
        """diagnostic: wall time (device drained) since the previous mark, per segment"""
        if self.profile:
            import time
            torch.cuda.synchronize()
            now = time.perf_counter()
            self.seg.setdefault(name, []).append((now - self._t) * 1e3)
            self._t = now

    def step(self) -> Tuple[float, float, float]:
        """One pass of the loop body SCF.jl:399-573.  Returns (E, dE, D_rms)."""
        if self.profile:
            import time
            torch.cuda.synchronize()
            self._t = time.perf_counter()
            if not hasattr(self, "seg"):
                self.seg = {}
        N = self.N
        p = lambda t: ctypes.c_void_p(t.data_ptr())
        F_old = self.Fbuf[self.fi]                                 # last iteration's final Fock matrix stays where it is
        Fp = self.Fbuf[self.fi ^ 1]
        self.fb.build_ld(self.Cop, Fp)                             # SCF.jl:463: F = H + 2J - K straight into the padded buffer
        self._mark("fock")
        x = 1.0 / math.log(50.0 * self.dE, 50.0) if self.dE >= 1.0 else 1.0     # SCF.jl:504
        if self.ndiis > 0 and not self.diis_on_host:               # SCF.jl:472-505: DIIS + damping, four launches
            nd = self.ndiis
            self._gemm_tn(self.Dp, Fp, self.T1)                     # D F
            self._gemm_tn(self.Sp, self.T1, self.T2)                # S D F = (F D S)^T
            self.head = (self.head + 1) % nd
            self.n_hist = min(self.n_hist + 1, nd)
            solve = self.iter > 1
            if solve:
                self.B_dim = min(self.B_dim + 1, nd)
            rc = self._lib.jcdf_diis_step_device(self._st(), nd, self.head, self.B_dim if solve else 1, 1 if solve else 0, N, self.Np,
                                                 p(self.T2), p(Fp), p(self.e_hist), p(self.F_hist), p(self.Bmat_d), p(self.coef_d),
                                                 p(self.diis_flag), p(self.dots_work), p(F_old) if x != 1.0 else None, x)
            if rc != 0:
                raise RuntimeError("jcdf_diis_step_device failed (status %d)" % rc)
        else:
            if self.ndiis > 0:
                self._diis_on_host(Fp)
            if x != 1.0:
                Fp.mul_(x).add_(F_old, alpha=1.0 - x)
        self._mark("diis")
        self.fi ^= 1                                               # self.Fp_ is the new matrix, the other buffer holds F_old
        D_old = self.Dbuf[self.di]
        self.di ^= 1                                               # _diag writes the new density into the other buffer
        self._mark("damp")
        use_sp2 = self.sp2 is not None and not self.sp2_skip and not self.sp2_refresh
        if self.sp2_refresh:                                       # this step's eigensolve renews the reference (Fref, frontier eigenvalues)
            self.sp2_refresh = False
            self.sp2_refreshes += 1
        self._diag(use_sp2)
        self._mark("diag")
        e_h, drms, faulty, eig_bad, finished, trace, pivot, used, accel, ref_delta = self._tail(D_old, use_sp2)
        if faulty:                                                 # "Faulty DIIS!" SCF.jl:493-499 (seen one sync later)
            self.B_dim = 2
            self.diis_flag.zero_()
        if use_sp2:
            proj_ok = finished == 1.0 and abs(trace - self.n_occ) < 1e-6
            self.sp2.adapt(used, finished == 1.0)
            self.sp2_steps += 1
            if proj_ok and not pivot >= 1.0 and self.lowdin.steps < 40:
                # the projector is fine, the Newton-Schulz iteration for the basis had too few steps (the occupied space
                # turned further than last time): once more with the maximum, before an eigensolve is spent on it
                self.sp2_basis_retries += 1
                self.lowdin.steps = 40
                self._sp2_basis()
                e_h, drms, _, _, _, _, pivot, _ = self._tail(D_old, True)[:8]
            self.lowdin.adapt(pivot if math.isfinite(pivot) else 0.0)
            good = proj_ok and pivot >= 1.0 and math.isfinite(e_h)
            if not good:                                           # not converged in the squarings enqueued / basis lost: eigensolve
                self.sp2_fallbacks += 1
                why = "unfinished" if finished != 1.0 else "trace" if abs(trace - self.n_occ) >= 1e-6 else "basis" if not pivot >= 1.0 else "nan"
                self.sp2_reasons[why] = self.sp2_reasons.get(why, 0) + 1
                # a projection / orthonormalisation that did not converge may have left non-finite numbers in the zero
                # padding of the buffers the eigensolver path only partly overwrites
                self.Cop.zero_()
                self.Cpt.zero_()
                self._diag(False)
                e_h, drms, _, eig_bad = self._tail(D_old, False)[:4]
        if eig_bad:                                                # hand-off timeout / stedc failure on some rank: all ranks redo,
            self.eigh.check()                                      # the failing one with the vendor solver (counted, solver_report)
            self._diag(False)
            e_h, drms = self._tail(D_old, False)[:2]
        if use_sp2:
            self.sp2_accelerated += int(accel == 1.0)
            # The accelerated recursion (half the squarings) needs ||F' - F_ref|| below half the HOMO-LUMO gap of the last matrix
            # that was DIAGONALISED — normally the one of the first iteration, far from where the SCF has moved.  Once the density
            # has settled (its change below a threshold that drops tenfold with every request) one eigensolve is spent on a new
            # reference: 7 ms against ~1.4 ms saved in every projection that follows at N = 1250.
            if accel != 1.0 and drms < self.sp2_refresh_below and not self.sp2_skip:
                self.sp2_refresh = True
                self.sp2_refresh_below = 0.1 * drms
        self.sp2_skip = not (drms < 15.0)       # occupied space still turning by ~90 degrees somewhere: no basis to project
        E = e_h + self.E_nuc
        dE = E - self.E_old
        self.trail.append((self.iter, E, dE, drms))
        self.dE, self.E_old = dE, E
        self.iter += 1
        self._mark("energy")
        return E, dE, drms
