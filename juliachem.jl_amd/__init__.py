"""juliachem.jl_amd — MI355X-native density-fitted RHF Fock build behind
JuliaChem.jl's GPU-DF operator boundary.

Layout:  csrc/   HIP kernels + C ABI (include/jcdf.h)  -> lib/libjcdf_hip.so
         df.py   host-side mirror of the reference interface (ctypes over the C ABI)
         engine.py  device-resident driver used by bench.py (torch = plumbing only)
"""
from . import _lib
from ._lib import JCDFError, LIB_PATH
from .df import (JCDFHandle, JCDFGroup, group_reduce_plan, JCTC, JCTiming, JCTiming_GPUkey, JCTiming_key, SCFData, SCFGPUData_hip,
                 SCFOptions, ScreeningData, Basis, Shell, CalculationBasisSets, DFIntegralEngine,
                 TensorIntegralEngine, basis_from_shell_sizes, create_jctiming, create_scf_options,
                 df_rhf_fock_build, df_rhf_fock_build_GPU, get_default_gpu_data_hip,
                 get_screening_metadata, host_potrf_trtri, device_potrf_trtri, lapack_potrf_trtri, packed_pq_lists,
                 setup_unscreened_screening_matricies, static_load_rank_indicies,
                 calculate_device_ranges_GPU)

__all__ = [n for n in dir() if not n.startswith("_")]
