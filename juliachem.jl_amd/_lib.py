"""ctypes binding of libjcdf_hip.so (include/jcdf.h).  The product path has no
CPU fallback: if the HIP library is missing or no gfx950 device exists, every
entry point raises JCDFError."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("JCDF_LIB_PATH") or os.path.join(_HERE, "lib", "libjcdf_hip.so")   # (override: diagnostic builds of tools/)


def is_diagnostic_build() -> bool:
    """True when the loaded library is a -DJCDF_DIAGNOSTIC build (tools/build_diag.sh): experiment entry points present."""
    return hasattr(load(), "jcdf_sytrd2_device")


class JCDFError(RuntimeError):
    """A non-zero jcdf_* status, converted the way the Julia glue converts it
    to error() (reference convention: GPUDF.jl:39-41)."""

    def __init__(self, code: int, msg: str):
        super().__init__("jcdf status %d: %s" % (code, msg))
        self.code = code


class jcdf_timings(C.Structure):
    _fields_ = [(n, C.c_double) for n in (
        "non_zero_coeff_time", "W_time", "K_time", "V_time", "J_time", "density_time",
        "H_add_time", "copy_J_time", "fock_time", "copy_time")]


class jcdf_group_timings(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("bcast_time", "build_time", "reduce_time", "d2h_time", "total_time")]


class jcdf_kernel_stat(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("seconds", C.c_double), ("flops", C.c_double),
                ("alg_flops", C.c_double), ("alg_bytes", C.c_double)]


_P = C.c_void_p
_D = C.POINTER(C.c_double)
_I64 = C.c_int64

# every symbol include/jcdf.h declares: name -> (restype, argtypes)
PROTOTYPES = {
    "jcdf_create": (C.c_int32, [C.POINTER(_P), C.c_int32]),
    "jcdf_destroy": (C.c_int32, [_P]),
    "jcdf_last_error": (C.c_char_p, [_P]),
    "jcdf_abi_version": (C.c_int32, []),
    "jcdf_set_stream": (C.c_int32, [_P, _P, C.c_int32]),
    "jcdf_configure": (C.c_int32, [_P, _I64, _I64, _I64, _I64, _I64, _I64, _P, _P]),
    "jcdf_set_tuning": (C.c_int32, [_P, C.c_char_p, _I64]),
    "jcdf_set_exchange_screening": (C.c_int32, [_P, _I64]),
    "jcdf_set_metric": (C.c_int32, [_P, _P]),
    "jcdf_set_metric_inverse": (C.c_int32, [_P, _P]),
    "jcdf_push_three_center": (C.c_int32, [_P, _I64, _I64, _P]),
    "jcdf_push_three_center_device": (C.c_int32, [_P, _I64, _I64, _P]),
    "jcdf_set_B": (C.c_int32, [_P, _P]),
    "jcdf_get_B": (C.c_int32, [_P, _P]),
    "jcdf_set_B_columns_device": (C.c_int32, [_P, C.c_int64, C.c_int64, _P]),
    "jcdf_set_core_hamiltonian": (C.c_int32, [_P, _P]),
    "jcdf_fock_build": (C.c_int32, [_P, _P, _P, C.POINTER(jcdf_timings)]),
    "jcdf_fock_build_begin": (C.c_int32, [_P, _P]),
    "jcdf_fock_build_finish": (C.c_int32, [_P, _P, C.POINTER(jcdf_timings)]),
    "jcdf_fock_build_device": (C.c_int32, [_P, _P, _P, _P]),
    "jcdf_fock_build_device_ld": (C.c_int32, [_P, _P, _I64, _P, _I64, _P]),
    "jcdf_set_overlap": (C.c_int32, [_P, C.c_int32]),
    "jcdf_synchronize": (C.c_int32, [_P, C.POINTER(jcdf_timings)]),
    "jcdf_get_V": (C.c_int32, [_P, _P]),
    "jcdf_get_W": (C.c_int32, [_P, _P]),
    "jcdf_host_potrf_trtri": (C.c_int32, [_P, _I64]),
    "jcdf_device_potrf_trtri": (C.c_int32, [C.c_int32, _P, _I64]),
    "jcdf_sytrd_workspace_bytes": (_I64, [_I64]),
    "jcdf_sytrd_device": (C.c_int32, [_P, _I64, _P, _I64, _P, _P, _P, _P, _I64]),
    "jcdf_sytrd_q_device": (C.c_int32, [_P, _I64, _P, _I64, _P, _P, _P, _P, _I64, _P, _I64]),
    "jcdf_sytrd_max_n": (_I64, [C.c_int32]),
    "jcdf_set_persistent_launch_mode": (C.c_int32, [C.c_int32]),
    "jcdf_ormtr_workspace_bytes": (_I64, [_I64]),
    "jcdf_ormtr_device": (C.c_int32, [_P, _I64, _P, _I64, _P, _P, _I64, _P, _I64, _P, _I64]),
    "jcdf_diis_device": (C.c_int32, [_P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, _P]),
    "jcdf_stedc_workspace_bytes": (_I64, [_I64]),
    "jcdf_stedc_device": (C.c_int32, [_P, _I64, _P, _P, _P, _I64, _P, _I64]),
    "jcdf_scf_tail_device": (C.c_int32, [_P, _I64, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "jcdf_lowdin_workspace_bytes": (_I64, [_I64]),
    "jcdf_lowdin_rows_device": (C.c_int32, [_P, _I64, _I64, _P, _I64, _P, _I64, C.c_int32, _P, _I64, _P]),
    "jcdf_sp2_workspace_bytes": (_I64, [_I64]),
    "jcdf_sp2_device": (C.c_int32, [_P, _I64, _I64, _P, _I64, _P, _I64, C.c_int32, _P, _I64, _P]),
    "jcdf_sp2_ref_device": (C.c_int32, [_P, _I64, _I64, _P, _I64, _P, _I64, C.c_int32, _P, _I64, _P, _P, _I64, _P]),
    "jcdf_gemm_tn_device": (C.c_int32, [_P, _I64, _I64, _I64, C.c_double, _P, _I64, _P, _I64, _P, _I64]),
    "jcdf_gemm_nt_device": (C.c_int32, [_P, _I64, _I64, _I64, _P, _I64, _P, _I64, _P, _I64]),
    "jcdf_diis_push_device": (C.c_int32, [_P, _I64, _I64, _P, _P, _P, _P]),
    "jcdf_diis_dots_device": (C.c_int32, [_P, C.c_int32, C.c_int32, _I64, _P, _P, _P]),
    "jcdf_diis_mix_device": (C.c_int32, [_P, C.c_int32, _I64, _I64, _P, _P, _P]),
    "jcdf_diis_step_device": (C.c_int32, [_P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _I64, _I64, _P, _P, _P, _P, _P, _P, _P, _P, _P, C.c_double]),
    "jcdf_device_bytes": (_I64, [_P]),
    "jcdf_kernel_stats": (C.c_int32, [_P, C.POINTER(jcdf_kernel_stat), C.c_int32]),
    "jcdf_kernel_stats_total": (C.c_int32, [_P, C.POINTER(jcdf_kernel_stat), C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_double), C.c_int32]),
    # multi-device group (all devices of one process behind one call; F reduced on the devices)
    "jcdf_group_create": (C.c_int32, [C.POINTER(_P), C.c_int32, C.POINTER(C.c_int32)]),
    "jcdf_group_destroy": (C.c_int32, [_P]),
    "jcdf_group_last_error": (C.c_char_p, [_P]),
    "jcdf_group_size": (C.c_int32, [_P]),
    "jcdf_group_handle": (_P, [_P, C.c_int32]),
    "jcdf_group_set_transport": (C.c_int32, [_P, C.c_char_p]),
    "jcdf_group_transport": (C.c_char_p, [_P]),
    "jcdf_group_reduce_plan": (_I64, [_I64, C.c_int32, C.POINTER(C.c_int64)]),
    "jcdf_group_configure": (C.c_int32, [_P, _I64, _I64, C.POINTER(C.c_int64), _I64, _I64, _P, _P]),
    "jcdf_group_set_exchange_screening": (C.c_int32, [_P, _I64]),
    "jcdf_group_set_metric": (C.c_int32, [_P, _P]),
    "jcdf_group_push_three_center": (C.c_int32, [_P, _I64, _I64, _P]),
    "jcdf_group_set_core_hamiltonian": (C.c_int32, [_P, _P]),
    "jcdf_group_fock_build": (C.c_int32, [_P, _P, _P, C.POINTER(jcdf_timings), C.POINTER(jcdf_group_timings)]),
    "jcdf_group_fock_build_device_ld": (C.c_int32, [_P, _P, _I64, _P, _I64, _P]),
    "jcdf_group_synchronize": (C.c_int32, [_P, C.POINTER(jcdf_timings), C.POINTER(jcdf_group_timings)]),
}

# entry points of DIAGNOSTIC builds only (csrc/jcdf_diag.h; tools/build_diag.sh + JCDF_LIB_PATH): bound when present
DIAG_PROTOTYPES = {
    "jcdf_keepalive_device": (C.c_int32, [_P, C.c_int32, C.c_int32, C.c_double, C.c_int32, C.c_int32, _P, _P]),
    "jcdf_sytrd_replay_q_device": (C.c_int32, [_P, _I64, _P, _I64, _P, _P, _I64]),
    "jcdf_sytrd2_max_n": (_I64, []),
    "jcdf_sytrd2_workspace_bytes": (_I64, [_I64]),
    "jcdf_sytrd2_device": (C.c_int32, [_P, _I64, _P, _I64, _P, _P, _P, _I64, _P, _I64]),
    "jcdf_sytrd2_apply_q_device": (C.c_int32, [_P, _I64, _P, _I64, _P, _I64]),
    "jcdf_w_stall_cycles": (_I64, [_P, _P, _I64]),
}

_lib = None


def load() -> C.CDLL:
    """Load libjcdf_hip.so (built in-tree by build.sh / __graft_entry__.build())."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise JCDFError(-1, "HIP library not built: %s (run ./build.sh); there is no CPU fallback" % LIB_PATH)
        # One HIP runtime per process: PyTorch bundles its own libamdhip64.so.7; importing it
        # first makes libjcdf_hip.so (NEEDED libamdhip64.so.7) bind to that same copy instead of
        # loading /opt/rocm's next to it (two runtimes -> "No HIP GPUs are available").
        try:
            import torch  # noqa: F401
        except Exception:
            pass
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        for name, (res, args) in DIAG_PROTOTYPES.items():
            fn = getattr(lib, name, None)
            if fn is not None:
                fn.restype = res
                fn.argtypes = args
        _lib = lib
    return _lib
