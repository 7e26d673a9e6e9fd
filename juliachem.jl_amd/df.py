"""Host-side mirror of JuliaChem's DF Fock-build operator interface, on top of the
C ABI (include/jcdf.h).  Julia is not available in this image, so the host side
is Python; names, argument meaning and error behaviour follow the reference so
the parity tests read like the reference's own drivers:

    df_rhf_fock_build!      DensityFitting.jl:23-76    -> df_rhf_fock_build
    run_gpu_fock_build!     DensityFitting.jl:78-90    -> run_gpu_fock_build
    df_rhf_fock_build_GPU!  GPUDF.jl:11-304            -> df_rhf_fock_build_GPU
    calculate_B_GPU!        GPUDF.jl:828-1008          -> calculate_B_GPU
    SCFData / ScreeningData shared/SCFData.jl:1-44
    SCFGPUData_cuda         shared/GPUData_cuda.jl:4-38 -> SCFGPUData_hip
    SCFOptions              shared/SCFOptions.jl:2-139
    JCTiming / JCTC keys    shared/JCTiming.jl:3-145

(paths relative to /root/reference/src/rhf/energy/DensityFitting/ or src/shared/).
The Julia glue with the same structure is julia/JCDFHip.jl.

This module never imports oracle/: all arithmetic happens in libjcdf_hip.so.
"""
from __future__ import annotations

import ctypes as C
import time
from dataclasses import dataclass, field
from typing import Any, Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import _lib
from ._lib import JCDFError, jcdf_group_timings, jcdf_kernel_stat, jcdf_timings


# --------------------------------------------------------------------------
# JCTiming (shared/JCTiming.jl)
# --------------------------------------------------------------------------
class JCTC:
    """Timing-key constants, same strings as module JCTC (JCTiming.jl:3-109)."""
    fock_time = "fock_time-"
    iteration_time = "iteration_time-"
    form_J_AB_inv_time = "form_J_AB_inv_time"
    density_time = "density_time-"
    H_add_time = "H_add_time-"
    K_time = "K_time-"
    W_time = "W_time-"
    J_time = "J_time-"
    V_time = "V_time-"
    screening_time = "screening_time"
    fock_MPI_time = "fock_MPI_time-"
    contraction_algorithm = "contraction_algorithm"
    B_time = "B_time"
    three_eri_time = "three_eri_time"
    two_eri_time = "two_eri_time"
    screened_indices_count = "screened_indices_count"
    GPU_num_devices = "GPU_num_devices"
    GPU_data_size_MB = "GPU_-N-_data_size_MB"
    GPU_density_time = "GPU_-N-_density_time-"
    GPU_V_time = "GPU_-N-_V_time-"
    GPU_J_time = "GPU_-N-_J_time-"
    GPU_W_time = "GPU_-N-_W_time-"
    GPU_K_time = "GPU_-N-_K_time-"
    GPU_non_zero_coeff_time = "GPU_-N-_non_zero_coeff_time-"
    GPU_H_add_time = "GPU_-N-_H_add_time-"
    gpu_fock_time = "GPU_-N-_fock_time-"
    gpu_copy_J_time = "gpu_copy_J_time-"
    gpu_copy_sym_time = "gpu_copy_sym_time-"
    total_fock_gpu_time = "total_fock_gpu_time-"
    fock_gpu_cpu_copy_reduce_time = "fock_gpu_cpu_copy_reduce_time-"


def JCTiming_key(key: str, iteration: int) -> str:
    return key + str(iteration)                                   # JCTiming.jl:135-137


def JCTiming_GPUkey(key: str, device: int, iteration: Optional[int] = None) -> str:
    k = key.replace("-N-", str(device))                           # JCTiming.jl:139-145
    return k if iteration is None else k + str(iteration)


@dataclass
class JCTiming:
    run_time: float = 0.0
    run_name: str = ""
    converged: bool = False
    scf_energy: float = 0.0
    non_timing_data: Dict[str, str] = field(default_factory=dict)
    user_options: Dict[str, str] = field(default_factory=dict)
    options: Dict[str, str] = field(default_factory=dict)
    timings: Dict[str, float] = field(default_factory=dict)


def create_jctiming() -> JCTiming:
    return JCTiming()


# --------------------------------------------------------------------------
# SCFOptions (shared/SCFOptions.jl, shared/Constants.jl)
# --------------------------------------------------------------------------
@dataclass
class SCFOptions:
    density_fitting: bool = False
    contraction_mode: str = "default"
    load: str = "static"
    guess: str = "hcore"
    energy_convergence: float = 1e-3
    density_convergence: float = 1e-3
    df_energy_convergence: float = 1e-3
    df_density_convergence: float = 1e-3
    max_iterations: int = 10
    df_max_iterations: int = 10
    df_exchange_n_blocks: int = 0
    df_screening_sigma: float = 1e-5
    df_screen_exchange: bool = False
    df_force_dense: bool = False
    df_use_adaptive: bool = True
    num_devices: int = 1
    df_use_K_sym: bool = False
    df_K_sym_type: str = "square"


DF_EXCHANGE_N_BLOCKS_CPU_DEFAULT = 10      # Constants.jl:13 — the mode of the reference that screens exchange blocks


def exchange_screen_blocks(scf_options: "SCFOptions") -> int:
    """n_blocks for jcdf_set_exchange_screening: 0 unless df_exchange_screen is set (SCFOptions.jl:92-93); then the
    user's df_exchange_n_blocks or the default of the reference's screened mode (ScreenedDF.jl:120-122, 385-389).
    The other exchange knobs of the reference's GPU path — df_use_K_sym, df_K_sym_type (square / rect forms of the same
    lower-triangle K, GPUDF.jl:35,669-826) and, without df_exchange_screen, df_exchange_n_blocks (block width of its
    cuBLAS calls, GPUDF.jl:61-72) — select among algorithms with identical results; this library has one K kernel
    (64 x 64 blocks of the lower triangle, split-K chosen by jcdf_configure), so they are accepted and have no effect."""
    if not scf_options.df_screen_exchange:
        return 0
    return int(scf_options.df_exchange_n_blocks) or DF_EXCHANGE_N_BLOCKS_CPU_DEFAULT


def create_scf_options(scf_flags: Dict[str, Any]) -> SCFOptions:
    """JSON keywords.scf -> SCFOptions with the reference's defaults and its
    df_* aliasing rules (SCFOptions.jl:47-139, Constants.jl:3-78)."""
    g = scf_flags.get
    guess = g("guess", "hcore")
    do_df = str(g("scf_type", "rhf")).lower() == "df"
    dele = g("dele", 1e-3)
    rmsd = g("rmsd", 1e-3)
    if guess == "df":
        df_dele, df_rmsd = g("df_dele", 1e-3), g("df_rmsd", 1e-3)
    else:
        df_dele, df_rmsd = dele, rmsd
    niter = int(g("niter", 10))
    df_niter = niter if do_df else (int(g("df_niter", 10)) if guess == "df" else 0)
    return SCFOptions(do_df, g("contraction_mode", "default"), g("load", "static"), guess,
                      dele, rmsd, df_dele, df_rmsd, niter, df_niter,
                      int(g("df_exchange_n_blocks", 0)), float(g("df_sigma", 1e-5)),
                      bool(g("df_exchange_screen", False)), bool(g("df_force_dense", False)),
                      bool(g("df_use_adaptive", True)), int(g("num_devices", 1)),
                      bool(g("df_use_K_sym", False)), g("df_K_sym_type", "square"))


# --------------------------------------------------------------------------
# data structures the path keeps (modules/BasisStructs.jl, shared/SCFData.jl)
# --------------------------------------------------------------------------
@dataclass
class Shell:
    """The fields of Shell the DF path reads (BasisStructs.jl:3-29): `pos` is the
    1-based index of the shell's first basis function, `nbas` its Cartesian count."""
    pos: int
    nbas: int


@dataclass
class Basis:
    """BasisStructs.jl:170-180: shells, norb (# functions), nels (# electrons)."""
    shells: List[Shell]
    norb: int
    nels: int = 0

    def __len__(self) -> int:
        return len(self.shells)

    def __getitem__(self, i: int) -> Shell:
        return self.shells[i]


def basis_from_shell_sizes(nbas: Sequence[int], nels: int = 0) -> Basis:
    shells, pos = [], 1
    for n in nbas:
        shells.append(Shell(pos, int(n)))
        pos += int(n)
    return Basis(shells, pos - 1, nels)


@dataclass
class CalculationBasisSets:
    """BasisStructs.jl:182-185 (field name `auxillary` spelled as in the reference)."""
    primary: Basis
    auxillary: Basis


@dataclass
class ScreeningData:
    """shared/SCFData.jl:1-17, 0-based.  sparse_pq_index_map[q, p] = packed index
    or -1 (reference: 1-based with 0 for screened)."""
    sparse_pq_index_map: Optional[np.ndarray] = None
    basis_function_screen_matrix: Optional[np.ndarray] = None
    sparse_p_start_indices: Optional[np.ndarray] = None
    non_screened_p_indices_count: Optional[np.ndarray] = None
    screened_indices_count: int = 0
    K_block_width: int = 0


class SCFGPUData:
    """abstract parent (shared/GPUData.jl:4)."""


class SCFGPUDataNoGPU(SCFGPUData):
    pass


class SCFGPUData_hip(SCFGPUData):
    """Counterpart of SCFGPUData_cuda (GPUData_cuda.jl:4-38): instead of ~30
    CuArrays it owns opaque jcdf handles, one per device of this process."""

    def __init__(self) -> None:
        self.handles: List["JCDFHandle"] = []
        self.group: Optional["JCDFGroup"] = None      # num_devices > 1: the handles are the group's members
        self.device_Q_range_lengths: List[int] = []
        self.device_Q_indices: List[range] = []
        self.number_of_devices_used: int = 0

    def close(self) -> None:
        for h in self.handles:
            h.close()
        self.handles = []
        if self.group is not None:
            self.group.close()
            self.group = None


def get_default_gpu_data_hip() -> SCFGPUData_hip:
    return SCFGPUData_hip()


@dataclass
class SCFData:
    """shared/SCFData.jl:19-37 — only the fields the GPU DF path touches."""
    gpu_data: SCFGPUData
    two_electron_fock: Optional[np.ndarray] = None
    screening_data: ScreeningData = field(default_factory=ScreeningData)
    mu: int = 0          # reference field is `μ`
    occ: int = 0
    A: int = 0
    scf_iteration: int = 0


# --------------------------------------------------------------------------
# shard + packing rules (host logic; also declared in include/jcdf.h comments)
# --------------------------------------------------------------------------
def get_df_static_shell_indices(basis_sets: CalculationBasisSets, n_ranks: int, rank: int) -> range:
    """DynamicLoad.jl:160-171 (0-based half-open)."""
    n_shells = len(basis_sets.auxillary)
    n_indices = n_shells // n_ranks
    begin = n_indices * rank
    end = n_shells if rank == n_ranks - 1 else begin + n_indices
    return range(begin, end)


def static_load_rank_indicies(rank: int, n_ranks: int, basis_sets: CalculationBasisSets
                              ) -> Tuple[range, range]:
    """(aux shell range, aux function range), 0-based half-open (DynamicLoad.jl:174-203)."""
    sh = get_df_static_shell_indices(basis_sets, n_ranks, rank)
    aux = basis_sets.auxillary
    if len(sh) == 0:
        return sh, range(0, 0)
    first = aux[sh.start].pos - 1
    last = aux[sh.stop - 1].pos - 1 + aux[sh.stop - 1].nbas
    return sh, range(first, last)


def calculate_device_ranges_GPU(num_devices: int, n_ranks: int, basis_sets: CalculationBasisSets
                                ) -> List[range]:
    """aux function range per global device id = rank*num_devices + dev (GPUDF.jl:1026-1056)."""
    total = num_devices * n_ranks
    return [static_load_rank_indicies(g, total, basis_sets)[1] for g in range(total)]


def sparse_pq_index_map_from_mask(mask: np.ndarray) -> Tuple[np.ndarray, int]:
    """SchwarzScreening.jl:72-81: running index, outer loop p, inner loop q, at map[q, p]."""
    keep_q, keep_p = np.nonzero(np.asarray(mask, dtype=bool))           # row = q, col = p
    order = np.lexsort((keep_q, keep_p))                                   # sort by p, then q
    mp = -np.ones(mask.shape, dtype=np.int64)
    mp[keep_q[order], keep_p[order]] = np.arange(order.size)
    return mp, int(order.size)


def setup_unscreened_screening_matricies(n: int) -> ScreeningData:
    """SchwarzScreening.jl:97-111: map[q, p] = q + N*p."""
    mask = np.ones((n, n), dtype=bool)
    mp = (np.arange(n)[:, None] + n * np.arange(n)[None, :]).astype(np.int64)
    return ScreeningData(mp, mask, n * np.arange(n, dtype=np.int64),
                         np.full(n, n, dtype=np.int64), n * n, 0)


def get_screening_metadata(mask: np.ndarray) -> ScreeningData:
    """ScreenedDF.jl:16-77 given the Schwarz keep-mask (SchwarzScreening.jl:9-71
    needs 4-centre integrals and stays with the host integral code)."""
    mask = np.asarray(mask, dtype=bool)
    if mask.ndim != 2 or mask.shape[0] != mask.shape[1] or not np.array_equal(mask, mask.T):
        raise ValueError("basis_function_screen_matrix must be square and symmetric")
    mp, count = sparse_pq_index_map_from_mask(mask)
    n = mask.shape[0]
    kp = mask.sum(axis=0).astype(np.int64)
    start = np.zeros(n, dtype=np.int64)
    for p in range(n):
        col = mp[:, p]
        kept = col[col >= 0]
        start[p] = kept[0] if kept.size else 0
    return ScreeningData(mp, mask, start, kp, count, 0)


def packed_pq_lists(sd: ScreeningData) -> Tuple[np.ndarray, np.ndarray]:
    """inverse of sparse_pq_index_map: (pq_p, pq_q) per packed index
    (create_sparse_to_p_q_kernel, GPUDF.jl:422-438)."""
    qq, pp = np.nonzero(sd.sparse_pq_index_map >= 0)
    idx = sd.sparse_pq_index_map[qq, pp]
    pq_p = np.empty(sd.screened_indices_count, dtype=np.int64)
    pq_q = np.empty(sd.screened_indices_count, dtype=np.int64)
    pq_p[idx] = pp
    pq_q[idx] = qq
    return pq_p, pq_q


# --------------------------------------------------------------------------
# thin OO wrapper of one jcdf handle
# --------------------------------------------------------------------------
def _f64(a: np.ndarray, order: str = "F") -> np.ndarray:
    return np.require(a, dtype=np.float64, requirements=["F" if order == "F" else "C", "A"])


class JCDFHandle:
    """One HIP device == one aux shard (include/jcdf.h)."""

    def __init__(self, device_id: int = 0, _borrowed: Optional[int] = None) -> None:
        self._lib = _lib.load()
        self._owned = _borrowed is None
        if _borrowed is not None:                    # a member of a JCDFGroup: the group destroys it
            hp = C.c_void_p(_borrowed)
        else:
            hp = C.c_void_p()
            rc = self._lib.jcdf_create(C.byref(hp), int(device_id))
            if rc != 0:
                raise JCDFError(rc, (self._lib.jcdf_last_error(None) or b"").decode())
        self._h = hp
        self.device_id = device_id
        self.N = self.o = self.Ql = self.P = 0

    def _check(self, rc: int) -> None:
        if rc != 0:
            raise JCDFError(rc, (self._lib.jcdf_last_error(self._h) or b"").decode())

    def close(self) -> None:
        if getattr(self, "_h", None):
            if self._owned:
                self._lib.jcdf_destroy(self._h)
            self._h = None

    def __del__(self) -> None:
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, stream: Optional[int]) -> None:
        """stream: a hipStream_t value (0 = legacy default stream); None = the handle's own stream."""
        if stream is None:
            self._check(self._lib.jcdf_set_stream(self._h, None, 1))
        else:
            self._check(self._lib.jcdf_set_stream(self._h, stream or None, 0))

    def set_tuning(self, key: str, value: int) -> None:
        """jcdf_set_tuning: "k_slices_per_xcd", "w_chunk_stages", "host_cholesky" (before configure; 0 = library rule)."""
        self._check(self._lib.jcdf_set_tuning(self._h, key.encode(), int(value)))

    def set_exchange_screening(self, n_blocks: int) -> None:
        """df_exchange_screen of the reference (ScreenedDF.jl:431-447, 459-545): K blocks of width N / n_blocks without a
        kept pair are not computed (K = 0 there).  Before configure; 0 = off."""
        self._check(self._lib.jcdf_set_exchange_screening(self._h, int(n_blocks)))

    def configure(self, N: int, Q_total: int, q0: int, q1: int, n_occ: int,
                  pq_p: Optional[np.ndarray] = None, pq_q: Optional[np.ndarray] = None) -> None:
        if pq_p is None:
            P, pp, pq = N * N, None, None
        else:
            self._pq_p = np.ascontiguousarray(pq_p, dtype=np.int64)
            self._pq_q = np.ascontiguousarray(pq_q, dtype=np.int64)
            P, pp, pq = self._pq_p.size, self._pq_p.ctypes.data, self._pq_q.ctypes.data
        self._check(self._lib.jcdf_configure(self._h, N, Q_total, q0, q1, n_occ, P, pp, pq))
        self.N, self.o, self.Ql, self.P, self.Qtot, self.q0, self.q1 = N, n_occ, q1 - q0, P, Q_total, q0, q1

    def set_metric(self, J2c: np.ndarray) -> None:
        a = _f64(J2c)
        assert a.shape == (self.Qtot, self.Qtot)
        self._check(self._lib.jcdf_set_metric(self._h, a.ctypes.data))

    def set_metric_inverse(self, Linv: np.ndarray) -> None:
        a = _f64(Linv)
        assert a.shape == (self.Qtot, self.Qtot)
        self._check(self._lib.jcdf_set_metric_inverse(self._h, a.ctypes.data))

    def push_three_center(self, s0: int, s1: int, T: np.ndarray) -> None:
        a = _f64(T)
        assert a.shape == (s1 - s0, self.P), (a.shape, (s1 - s0, self.P))
        self._check(self._lib.jcdf_push_three_center(self._h, s0, s1, a.ctypes.data))

    def push_three_center_device(self, s0: int, s1: int, d_ptr: int) -> None:
        self._check(self._lib.jcdf_push_three_center_device(self._h, s0, s1, d_ptr))

    def set_B(self, B: np.ndarray) -> None:
        a = _f64(B)
        assert a.shape == (self.Ql, self.P)
        self._check(self._lib.jcdf_set_B(self._h, a.ctypes.data))

    def set_B_columns_device(self, c0: int, c1: int, d_ptr: int) -> None:
        """packed columns [c0,c1) of B, (Ql x (c1-c0)) column-major, already on the device"""
        self._check(self._lib.jcdf_set_B_columns_device(self._h, c0, c1, d_ptr))

    def get_B(self) -> np.ndarray:
        out = np.empty((self.Ql, self.P), dtype=np.float64, order="F")
        self._check(self._lib.jcdf_get_B(self._h, out.ctypes.data))
        return out

    def set_core_hamiltonian(self, H: Optional[np.ndarray]) -> None:
        if H is None:
            self._check(self._lib.jcdf_set_core_hamiltonian(self._h, None))
        else:
            a = _f64(H)
            assert a.shape == (self.N, self.N)
            self._check(self._lib.jcdf_set_core_hamiltonian(self._h, a.ctypes.data))

    def fock_build(self, C_occ: np.ndarray) -> Tuple[np.ndarray, jcdf_timings]:
        c = _f64(C_occ)
        if self.N and c.shape != (self.N, self.o):
            raise JCDFError(1, "C_occ has shape %s, expected (N, n_occ) = %s" % (c.shape, (self.N, self.o)))
        F = np.empty((self.N, self.N), dtype=np.float64, order="F")
        t = jcdf_timings()
        self._check(self._lib.jcdf_fock_build(self._h, c.ctypes.data, F.ctypes.data, C.byref(t)))
        return F, t

    def fock_build_begin(self, C_occ: np.ndarray) -> None:
        c = _f64(C_occ)
        if self.N and c.shape != (self.N, self.o):
            raise JCDFError(1, "C_occ has shape %s, expected (N, n_occ) = %s" % (c.shape, (self.N, self.o)))
        self._check(self._lib.jcdf_fock_build_begin(self._h, c.ctypes.data))

    def fock_build_finish(self) -> Tuple[np.ndarray, jcdf_timings]:
        F = np.empty((self.N, self.N), dtype=np.float64, order="F")
        t = jcdf_timings()
        self._check(self._lib.jcdf_fock_build_finish(self._h, F.ctypes.data, C.byref(t)))
        return F, t

    def fock_build_device(self, d_C_occ: int, d_F: int, stream: int = 0) -> None:
        self._check(self._lib.jcdf_fock_build_device(self._h, d_C_occ, d_F, stream or None))

    def fock_build_device_ld(self, d_C_occ: int, ldc: int, d_F: int, ldf: int, stream: int = 0) -> None:
        """jcdf_fock_build_device_ld: orbital i at d_C_occ + ldc i, column p of F at d_F + ldf p (zero padded caller matrices)"""
        self._check(self._lib.jcdf_fock_build_device_ld(self._h, d_C_occ, int(ldc), d_F, int(ldf), stream or None))

    def synchronize(self) -> jcdf_timings:
        t = jcdf_timings()
        self._check(self._lib.jcdf_synchronize(self._h, C.byref(t)))
        return t

    def get_V(self) -> np.ndarray:
        out = np.empty(self.Ql, dtype=np.float64)
        self._check(self._lib.jcdf_get_V(self._h, out.ctypes.data))
        return out

    def get_W(self) -> np.ndarray:
        out = np.empty((self.Ql, self.o, self.N), dtype=np.float64, order="F")
        self._check(self._lib.jcdf_get_W(self._h, out.ctypes.data))
        return out

    def set_overlap(self, overlap_jk: bool) -> None:
        """J beside K on a side stream (default) or one after the other (stand-alone kernel timings)."""
        self._check(self._lib.jcdf_set_overlap(self._h, 1 if overlap_jk else 0))

    def device_bytes(self) -> int:
        return int(self._lib.jcdf_device_bytes(self._h))

    def kernel_stats(self) -> List[Dict[str, Any]]:
        arr = (jcdf_kernel_stat * 16)()
        n = self._lib.jcdf_kernel_stats(self._h, arr, 16)
        return [dict(name=arr[i].name.decode(), seconds=arr[i].seconds, flops=arr[i].flops,
                     alg_flops=arr[i].alg_flops, alg_bytes=arr[i].alg_bytes) for i in range(n)]


def _kernel_stats_total(self, reset: bool = False):
    """(records with `seconds` summed over the builds since the last reset, number of builds, summed whole-build seconds)"""
    arr = (jcdf_kernel_stat * 16)()
    nb, fs = C.c_int64(0), C.c_double(0.0)
    n = self._lib.jcdf_kernel_stats_total(self._h, arr, 16, C.byref(nb), C.byref(fs), 1 if reset else 0)
    recs = [dict(name=arr[i].name.decode(), seconds=arr[i].seconds, flops=arr[i].flops, alg_flops=arr[i].alg_flops,
                 alg_bytes=arr[i].alg_bytes) for i in range(n)]
    return recs, int(nb.value), float(fs.value)


JCDFHandle.kernel_stats_total = _kernel_stats_total


class JCDFGroup:
    """All devices of this process behind one call (include/jcdf.h, jcdf_group_*): member i = aux shard i on
    device_ids[i]; C_occ goes up once, the partial Fock matrices are summed ON THE DEVICES (RCCL reduce-scatter or the
    library's peer-mapped slice sums) and F comes down once — instead of the reference's one task, one H2D, one D2H per
    device and a host axpy! (GPUDF.jl:188-193, 267-277)."""

    def __init__(self, device_ids: Sequence[int]) -> None:
        self._lib = _lib.load()
        ids = (C.c_int32 * len(device_ids))(*[int(d) for d in device_ids])
        gp = C.c_void_p()
        rc = self._lib.jcdf_group_create(C.byref(gp), len(device_ids), ids)
        if rc != 0:
            raise JCDFError(rc, (self._lib.jcdf_group_last_error(None) or b"").decode())
        self._g = gp
        self.device_ids = [int(d) for d in device_ids]
        self.n = len(self.device_ids)
        self.members = [JCDFHandle(d, _borrowed=self._lib.jcdf_group_handle(gp, i)) for i, d in enumerate(self.device_ids)]
        self.N = self.o = self.P = self.Qtot = 0

    def _check(self, rc: int) -> None:
        if rc != 0:
            raise JCDFError(rc, (self._lib.jcdf_group_last_error(self._g) or b"").decode())

    def close(self) -> None:
        if getattr(self, "_g", None):
            for m in self.members:
                m.close()                             # borrowed: forgets the pointer
            self._lib.jcdf_group_destroy(self._g)
            self._g = None

    def __del__(self) -> None:
        try:
            self.close()
        except Exception:
            pass

    def set_transport(self, name: str) -> None:
        """"auto" | "rccl" | "peer" """
        self._check(self._lib.jcdf_group_set_transport(self._g, name.encode()))

    def transport(self) -> str:
        return (self._lib.jcdf_group_transport(self._g) or b"").decode()

    def set_exchange_screening(self, n_blocks: int) -> None:
        self._check(self._lib.jcdf_group_set_exchange_screening(self._g, int(n_blocks)))

    def configure(self, N: int, Q_total: int, shard_q0: Sequence[int], n_occ: int,
                  pq_p: Optional[np.ndarray] = None, pq_q: Optional[np.ndarray] = None) -> None:
        if len(shard_q0) != self.n + 1:
            raise JCDFError(1, "shard_q0 needs n_devices + 1 entries")
        if pq_p is None:
            P, pp, pq = N * N, None, None
        else:
            self._pq_p = np.ascontiguousarray(pq_p, dtype=np.int64)
            self._pq_q = np.ascontiguousarray(pq_q, dtype=np.int64)
            P, pp, pq = self._pq_p.size, self._pq_p.ctypes.data, self._pq_q.ctypes.data
        q0 = (C.c_int64 * (self.n + 1))(*[int(x) for x in shard_q0])
        self._check(self._lib.jcdf_group_configure(self._g, N, Q_total, q0, n_occ, P, pp, pq))
        self.N, self.o, self.P, self.Qtot = N, n_occ, P, Q_total
        for i, m in enumerate(self.members):
            m.N, m.o, m.P, m.Qtot = N, n_occ, P, Q_total
            m.q0, m.q1 = int(shard_q0[i]), int(shard_q0[i + 1])
            m.Ql = m.q1 - m.q0

    def set_metric(self, J2c: np.ndarray) -> None:
        a = _f64(J2c)
        assert a.shape == (self.Qtot, self.Qtot)
        self._check(self._lib.jcdf_group_set_metric(self._g, a.ctypes.data))

    def push_three_center(self, s0: int, s1: int, T: np.ndarray) -> None:
        a = _f64(T)
        assert a.shape == (s1 - s0, self.P), (a.shape, (s1 - s0, self.P))
        self._check(self._lib.jcdf_group_push_three_center(self._g, s0, s1, a.ctypes.data))

    def push_three_center_device(self, s0: int, s1: int, d_ptr: int) -> None:
        """the same entry point: the library copies with hipMemcpyDefault, so T may live on any device of the process"""
        self._check(self._lib.jcdf_group_push_three_center(self._g, s0, s1, d_ptr))

    def set_core_hamiltonian(self, H: Optional[np.ndarray]) -> None:
        if H is None:
            self._check(self._lib.jcdf_group_set_core_hamiltonian(self._g, None))
        else:
            a = _f64(H)
            assert a.shape == (self.N, self.N)
            self._check(self._lib.jcdf_group_set_core_hamiltonian(self._g, a.ctypes.data))

    def fock_build(self, C_occ: np.ndarray) -> Tuple[np.ndarray, List[jcdf_timings], jcdf_group_timings]:
        c = _f64(C_occ)
        if c.shape != (self.N, self.o):
            raise JCDFError(1, "C_occ has shape %s, expected (N, n_occ) = %s" % (c.shape, (self.N, self.o)))
        F = np.empty((self.N, self.N), dtype=np.float64, order="F")
        t = (jcdf_timings * self.n)()
        gt = jcdf_group_timings()
        self._check(self._lib.jcdf_group_fock_build(self._g, c.ctypes.data, F.ctypes.data, t, C.byref(gt)))
        return F, list(t), gt

    def fock_build_device_ld(self, d_C_occ: int, ldc: int, d_F: int, ldf: int, stream: int = 0) -> None:
        self._check(self._lib.jcdf_group_fock_build_device_ld(self._g, d_C_occ, int(ldc), d_F, int(ldf), stream or None))

    def synchronize(self) -> Tuple[List[jcdf_timings], jcdf_group_timings]:
        t = (jcdf_timings * self.n)()
        gt = jcdf_group_timings()
        self._check(self._lib.jcdf_group_synchronize(self._g, t, C.byref(gt)))
        return list(t), gt


def group_reduce_plan(count: int, n: int) -> Tuple[int, List[int]]:
    """jcdf_group_reduce_plan: (chunk, n + 1 slice offsets of the N*N elements) — pure host code of the library"""
    off = (C.c_int64 * (n + 1))()
    chunk = int(_lib.load().jcdf_group_reduce_plan(int(count), int(n), off))
    if chunk < 0:
        raise JCDFError(1, "jcdf_group_reduce_plan: bad arguments")
    return chunk, [int(x) for x in off]


def lapack_potrf_trtri(J2c: np.ndarray) -> np.ndarray:
    """L^-1 exactly as the reference forms it on the host: LAPACK.potrf!('L') + trtri!('L','N')
    (GPUDF.jl:890-891, DensityFitting.jl:137-140), through scipy's LAPACK.  Upper triangle zeroed
    (TwoCenterIntegrals.jl:150-162).  Falls back to the library's own host routine without scipy."""
    try:
        from scipy.linalg import lapack
    except Exception:
        return host_potrf_trtri(J2c)
    a = np.array(J2c, dtype=np.float64, order="F", copy=True)
    c, info = lapack.dpotrf(a, lower=1, overwrite_a=1)
    if info != 0:
        raise JCDFError(5, "metric not positive definite at pivot %d" % info)
    inv, info = lapack.dtrtri(c, lower=1, overwrite_c=1)
    if info != 0:
        raise JCDFError(5, "dtrtri info=%d" % info)
    return np.asfortranarray(np.tril(inv))


def device_potrf_trtri(J2c: np.ndarray, device_id: int = 0) -> np.ndarray:
    """L^-1 via the library's blocked fp64-MFMA device factorisation (what jcdf_set_metric runs;
    CUSOLVER.potrf!/trtri! at DenseGPUDF.jl:185-193)."""
    a = np.array(J2c, dtype=np.float64, order="F", copy=True)
    if a.ndim != 2 or a.shape[0] != a.shape[1] or a.shape[0] == 0:
        raise JCDFError(1, "device_potrf_trtri: square non-empty matrix expected")
    rc = _lib.load().jcdf_device_potrf_trtri(_physical_device(device_id), a.ctypes.data, a.shape[0])
    if rc != 0:
        raise JCDFError(rc, "device potrf/trtri failed (status %d%s)" % (rc, ": not positive definite" if rc == 5 else ""))
    return a


def host_potrf_trtri(J2c: np.ndarray) -> np.ndarray:
    """L^-1 via the library's dependency-free host Cholesky/inverse (JCDF_HOST_CHOLESKY=1 path of jcdf_set_metric)."""
    a = np.array(J2c, dtype=np.float64, order="F", copy=True)
    rc = _lib.load().jcdf_host_potrf_trtri(a.ctypes.data, a.shape[0])
    if rc != 0:
        raise JCDFError(5, "metric not positive definite at pivot %d" % rc)
    return a


# --------------------------------------------------------------------------
# integral source: what the reference gets from the JERI/Libint engines
# (jeri_engine_thread_df, jeri_engine_thread).  Integrals stay on the host.
# --------------------------------------------------------------------------
class DFIntegralEngine:
    """Host provider of the DF integrals, standing where `jeri_engine_thread_df`
    stands in the reference signatures.  Subclasses implement:

    calculate_two_center_intgrals() -> (A, A) lower triangle valid
        (TwoCenterIntegrals.jl:7-29)
    calculate_three_center_integrals(aux_range, screening_data) -> (len, P)
        packed exactly like ThreeCenterIntegralsScreened.jl:8-85
    schwarz_mask(sigma, max_P_P) -> bool (N, N) or None for "keep all"
        (SchwarzScreening.jl:9-71)
    """

    def calculate_two_center_intgrals(self) -> np.ndarray:
        raise NotImplementedError

    def calculate_three_center_integrals(self, aux_range: range, sd: ScreeningData) -> np.ndarray:
        raise NotImplementedError

    def schwarz_mask(self, sigma: float, max_P_P: float) -> Optional[np.ndarray]:
        return None


class TensorIntegralEngine(DFIntegralEngine):
    """Integrals already in memory: J2c (A,A) and dense T (A,N,N) — used with
    synthetic tensors and with the small host integral code of the tests."""

    def __init__(self, J2c: np.ndarray, T_dense: np.ndarray, mask: Optional[np.ndarray] = None):
        self.J2c, self.T, self.mask = J2c, T_dense, mask

    def calculate_two_center_intgrals(self) -> np.ndarray:
        return np.tril(self.J2c)

    def calculate_three_center_integrals(self, aux_range: range, sd: ScreeningData) -> np.ndarray:
        pq_p, pq_q = packed_pq_lists(sd)
        return np.asfortranarray(self.T[aux_range.start:aux_range.stop][:, pq_q, pq_p])

    def schwarz_mask(self, sigma: float, max_P_P: float) -> Optional[np.ndarray]:
        return self.mask


# --------------------------------------------------------------------------
# communicator shim: torch.distributed when initialised, else single rank
# (reference: MPI.COMM_WORLD)
# --------------------------------------------------------------------------
def _comm() -> Tuple[int, int, Any]:
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist.get_rank(), dist.get_world_size(), dist
    except Exception:
        pass
    return 0, 1, None


# --------------------------------------------------------------------------
# the operator
# --------------------------------------------------------------------------
def calculate_B_GPU(scf_data: SCFData, engine: DFIntegralEngine, two_center_integrals: np.ndarray,
                    num_devices: int, basis_sets: CalculationBasisSets, jc_timing: JCTiming,
                    exchange_stats: Optional[dict] = None) -> None:
    """B = L^-1 (Q|pq) on the devices (GPUDF.jl:828-1008).  potrf/trtri on the device (reference: host LAPACK at
    :890-891, cuSOLVER at DenseGPUDF.jl:185-193) — with num_devices > 1 once, on the group's first device.  Every T row
    block s is produced by its owner and accumulated on every shard r with rows_r >= rows_s (L^-1 lower triangular).
    Across processes the blocks travel point-to-point, block s only to the ranks behind it
    (`engine.exchange_three_center_blocks`: RCCL send / recv on device tensors with the nccl backend, host tensors with
    gloo) — the reference stages every block to every rank through the host (MPI.Send/Recv!, :918-997)."""
    rank, n_ranks, dist = _comm()
    gd: SCFGPUData_hip = scf_data.gpu_data
    ranges = calculate_device_ranges_GPU(num_devices, n_ranks, basis_sets)
    gd.device_Q_indices = ranges
    gd.device_Q_range_lengths = [len(r) for r in ranges]
    sd = scf_data.screening_data
    t0 = time.perf_counter()
    if gd.group is not None:
        gd.group.set_metric(two_center_integrals)              # potrf + trtri once, L^-1 rows device-to-device
    else:
        for h in gd.handles:
            h.set_metric(two_center_integrals)
    jc_timing.timings[JCTC.form_J_AB_inv_time] = time.perf_counter() - t0

    def push_host(s0: int, s1: int, T: np.ndarray) -> None:
        if gd.group is not None:
            gd.group.push_three_center(s0, s1, T)              # ONE H2D, the other members fetch device-to-device
        else:
            for h in gd.handles:
                h.push_three_center(s0, s1, T)

    t_eri = 0.0
    t0 = time.perf_counter()
    if n_ranks == 1:
        for rows in ranges:
            t1 = time.perf_counter()
            T = engine.calculate_three_center_integrals(rows, sd)
            t_eri += time.perf_counter() - t1
            push_host(rows.start, rows.stop, T)
    else:
        import torch
        from .engine import exchange_three_center_blocks
        # this rank's rows = the union of its devices' (contiguous) shards; its own T block, (rows, P) column-major, flat
        rank_ranges = [range(ranges[r * num_devices].start, ranges[r * num_devices + num_devices - 1].stop) for r in range(n_ranks)]
        own = rank_ranges[rank]
        t1 = time.perf_counter()
        T_own = np.asfortranarray(engine.calculate_three_center_integrals(own, sd))
        t_eri += time.perf_counter() - t1
        P = sd.screened_indices_count
        on_device = dist.get_backend() == "nccl"
        flat = torch.from_numpy(T_own.reshape(-1, order="F"))
        if on_device:
            dev0 = torch.device("cuda", gd.handles[0].device_id)
            flat = flat.to(dev0)

        def push(s0: int, s1: int, chunk) -> None:
            if on_device:
                torch.cuda.synchronize(chunk.device)
                if gd.group is not None:
                    gd.group.push_three_center_device(s0, s1, chunk.data_ptr())
                else:
                    for h in gd.handles:
                        local = chunk if chunk.device.index == h.device_id else chunk.to(torch.device("cuda", h.device_id))
                        h.push_three_center_device(s0, s1, local.data_ptr())
            else:
                push_host(s0, s1, chunk.numpy().reshape((s1 - s0, P), order="F"))

        alloc = (lambda n: torch.empty(n, dtype=torch.float64, device=dev0)) if on_device else (lambda n: torch.empty(n, dtype=torch.float64))
        block = max(16, ((1 << 28) // max(P, 1)) // 16 * 16)
        stats: dict = {}
        exchange_three_center_blocks(rank_ranges, rank, n_ranks, dist, flat, alloc, push, block=block, stats=stats)
        if exchange_stats is not None:
            exchange_stats.update(stats)
        jc_timing.non_timing_data["B_exchange_doubles_sent"] = str(stats.get("sent", 0))
        jc_timing.non_timing_data["B_exchange_doubles_received"] = str(stats.get("received", 0))
    jc_timing.timings[JCTC.B_time] = time.perf_counter() - t0 - t_eri
    jc_timing.timings[JCTC.three_eri_time] = t_eri


def df_rhf_fock_build_GPU(scf_data: SCFData, jeri_engine_thread_df: DFIntegralEngine,
                          jeri_engine_thread: Any, basis_sets: CalculationBasisSets,
                          occupied_orbital_coefficients: np.ndarray, iteration: int,
                          scf_options: SCFOptions, H: np.ndarray, jc_timing: JCTiming,
                          force_dense: bool = False) -> None:
    """Signature and side effects of df_rhf_fock_build_GPU! (GPUDF.jl:11-14):
    scf_data.two_electron_fock <- this process's sum over its devices of
    2J - K (+ H on rank 0, device 1).  num_devices > 1: the devices are one `JCDFGroup` — C_occ uploaded once, the
    partial Fock matrices summed on the devices, one pass of D2H (reference: a task, an H2D and a D2H per device and a
    host axpy!, GPUDF.jl:188-193, 206, 267-277)."""
    rank, n_ranks, _ = _comm()
    num_devices = scf_options.num_devices
    gd = scf_data.gpu_data
    if not isinstance(gd, SCFGPUData_hip):
        raise JCDFError(1, "contraction_mode requires scf_data.gpu_data::SCFGPUData_hip")
    gd.number_of_devices_used = num_devices
    n, n_occ = scf_data.mu, scf_data.occ
    if iteration == 1:
        eng = jeri_engine_thread_df
        t0 = time.perf_counter()
        two_center = eng.calculate_two_center_intgrals()
        jc_timing.timings[JCTC.two_eri_time] = time.perf_counter() - t0
        t0 = time.perf_counter()
        mask = None if force_dense else eng.schwarz_mask(
            scf_options.df_screening_sigma, float(np.max(np.abs(np.diag(two_center)))))
        sd = setup_unscreened_screening_matricies(n) if mask is None else get_screening_metadata(mask)
        scf_data.screening_data = sd
        jc_timing.timings[JCTC.screening_time] = time.perf_counter() - t0
        jc_timing.timings[JCTC.screened_indices_count] = sd.screened_indices_count
        ranges = calculate_device_ranges_GPU(num_devices, n_ranks, basis_sets)
        pq = (None, None) if mask is None else packed_pq_lists(sd)
        gd.close()
        local = ranges[rank * num_devices:(rank + 1) * num_devices]
        if any(len(r) == 0 for r in local):
            raise JCDFError(1, "more devices than auxiliary shells: empty shard")
        devices = [_physical_device(dev if n_ranks == 1 else _local_device(dev, num_devices)) for dev in range(num_devices)]
        if num_devices > 1:
            g = JCDFGroup(devices)
            g.set_exchange_screening(exchange_screen_blocks(scf_options))
            # this rank's devices hold a contiguous part of the auxiliary basis (global device id = rank * num_devices + dev)
            g.configure(n, scf_data.A, [r.start for r in local] + [local[-1].stop], n_occ, pq[0], pq[1])
            g.set_core_hamiltonian(H if rank == 0 else None)                  # member 0 only (GPUDF.jl:158-161)
            gd.group = g
            gd.handles = list(g.members)
        else:
            h = JCDFHandle(devices[0])
            h.set_exchange_screening(exchange_screen_blocks(scf_options))
            h.configure(n, scf_data.A, local[0].start, local[0].stop, n_occ, pq[0], pq[1])
            h.set_core_hamiltonian(H if rank == 0 else None)                  # GPUDF.jl:158-161
            gd.handles.append(h)
        calculate_B_GPU(scf_data, eng, two_center, num_devices, basis_sets, jc_timing)
        jc_timing.non_timing_data[JCTC.contraction_algorithm] = "dense hip" if mask is None else "screened hip"
        jc_timing.non_timing_data[JCTC.GPU_num_devices] = str(num_devices)
        if gd.group is not None:
            jc_timing.non_timing_data["GPU_reduce_transport"] = gd.group.transport()
        for dev, h in enumerate(gd.handles):
            jc_timing.non_timing_data[JCTiming_GPUkey(JCTC.GPU_data_size_MB, dev + 1)] = str(h.device_bytes() / 1024 ** 2)

    t0 = time.perf_counter()
    if gd.group is not None:
        # all devices behind one call; the sum over devices happens on the devices
        F, per_dev, gt = gd.group.fock_build(occupied_orbital_coefficients)
        scf_data.two_electron_fock = F
        copy_reduce = gt.bcast_time + gt.reduce_time + gt.d2h_time
        jc_timing.non_timing_data["GPU_reduce_transport"] = gd.group.transport()
    else:
        F, t = gd.handles[0].fock_build(occupied_orbital_coefficients)
        per_dev = [t]
        scf_data.two_electron_fock = F
        copy_reduce = t.copy_time
    jc_timing.timings[JCTiming_key(JCTC.total_fock_gpu_time, iteration)] = time.perf_counter() - t0
    for dev, t in enumerate(per_dev, start=1):
        for key, val in ((JCTC.GPU_W_time, t.W_time), (JCTC.GPU_V_time, t.V_time), (JCTC.GPU_J_time, t.J_time),
                         (JCTC.GPU_K_time, t.K_time), (JCTC.GPU_density_time, t.density_time),
                         (JCTC.gpu_fock_time, t.fock_time),
                         (JCTC.GPU_non_zero_coeff_time, t.non_zero_coeff_time)):
            jc_timing.timings[JCTiming_GPUkey(key, dev, iteration)] = val
        jc_timing.timings[JCTiming_GPUkey(JCTC.gpu_copy_J_time, dev, iteration)] = t.copy_J_time
    for key, attr in ((JCTC.K_time, "K_time"), (JCTC.W_time, "W_time"), (JCTC.V_time, "V_time"),
                      (JCTC.J_time, "J_time"), (JCTC.fock_time, "fock_time")):
        jc_timing.timings[JCTiming_key(key, iteration)] = max(getattr(t, attr) for t in per_dev)
    jc_timing.timings[JCTiming_GPUkey(JCTC.GPU_H_add_time, 1, iteration)] = per_dev[0].H_add_time
    jc_timing.timings[JCTiming_key(JCTC.fock_gpu_cpu_copy_reduce_time, iteration)] = copy_reduce


def _physical_device(index: int) -> int:
    """Test hook: with JCDF_ALLOW_DEVICE_WRAP=1 logical devices wrap onto the physical ones, so the
    num_devices > 1 code path can be exercised on a one-GPU box (each handle is still one shard)."""
    import os
    if os.environ.get("JCDF_ALLOW_DEVICE_WRAP") == "1":
        import torch
        return index % max(1, torch.cuda.device_count())
    return index


def _local_device(dev: int, num_devices: int) -> int:
    import os
    return int(os.environ.get("LOCAL_RANK", "0")) * num_devices + dev


def run_gpu_fock_build(scf_data, jeri_engine_thread_df, jeri_engine_thread, basis_sets,
                       occupied_orbital_coefficients, iteration, scf_options, H, jc_timing) -> None:
    """DensityFitting.jl:78-90.  The reference picks its dense kernel set when
    df_force_dense / denseGPU, or adaptively for N < 800 on one rank; here "dense"
    only means the unscreened pq map — the same kernels run either way."""
    rank, n_ranks, _ = _comm()
    force_dense = scf_options.df_force_dense or scf_options.contraction_mode == "denseGPU"
    adaptive_dense = scf_options.df_use_adaptive and scf_data.mu < 800 and rank == 0 and n_ranks == 1
    df_rhf_fock_build_GPU(scf_data, jeri_engine_thread_df, jeri_engine_thread, basis_sets,
                          occupied_orbital_coefficients, iteration, scf_options, H, jc_timing,
                          force_dense=force_dense or adaptive_dense)


def df_rhf_fock_build(scf_data: SCFData, jeri_engine_thread_df: DFIntegralEngine, jeri_engine_thread: Any,
                      basis_sets: CalculationBasisSets, coefficients: np.ndarray, iteration: int,
                      scf_options: SCFOptions, H: np.ndarray, jc_timing: JCTiming) -> np.ndarray:
    """df_rhf_fock_build! (DensityFitting.jl:23-76): returns the full Fock matrix
    F = H + 2J - K reduced over all shards; the returned array IS
    scf_data.two_electron_fock (the caller's DIIS mutates it in place)."""
    rank, n_ranks, dist = _comm()
    if iteration == 1:
        scf_data.mu = basis_sets.primary.norb
        scf_data.A = basis_sets.auxillary.norb
        scf_data.occ = int(basis_sets.primary.nels) // 2
        scf_data.two_electron_fock = np.zeros((scf_data.mu, scf_data.mu), order="F")
    occupied_orbital_coefficients = np.asfortranarray(coefficients[:, :scf_data.occ])
    if scf_options.contraction_mode in ("GPU", "denseGPU", "HIP"):
        run_gpu_fock_build(scf_data, jeri_engine_thread_df, jeri_engine_thread, basis_sets,
                           occupied_orbital_coefficients, iteration, scf_options, H, jc_timing)
    else:
        raise JCDFError(1, "contraction_mode %r is a CPU mode of the reference; this package only "
                           "provides the GPU path (GPU / denseGPU / HIP) and has no CPU fallback"
                        % scf_options.contraction_mode)
    if n_ranks > 1:                                               # DensityFitting.jl:68-71
        import torch
        t0 = time.perf_counter()
        Ft = torch.from_numpy(np.ascontiguousarray(scf_data.two_electron_fock))
        if dist.get_backend() == "nccl":
            d = Ft.cuda()
            dist.all_reduce(d)
            Ft = d.cpu()
        else:
            dist.all_reduce(Ft)
        scf_data.two_electron_fock = np.asfortranarray(Ft.numpy())
        jc_timing.timings[JCTiming_key(JCTC.fock_MPI_time, iteration)] = time.perf_counter() - t0
    return scf_data.two_electron_fock
