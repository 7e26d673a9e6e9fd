// jcdf_scf.hpp — the scalar tail of one SCF iteration on the device (caller-side helper, like k_diis_solve):
//   E_elec = 1/2 sum D o (F + H)      (SCF.jl:1116-1123)
//   ||D - D_old||_F                   (SCF.jl:559-563 takes the rms of the same difference)
// and the iteration's status words packed beside them, so that the host's ONE copy per iteration reads one 64-byte
// record instead of the results of ~25 five-microsecond launches (sum, mul, norm, abs, casts, stack).
// Two launches: per-workgroup partial sums over a fixed contiguous partition, then one workgroup adds them in index
// order (bit-reproducible) and writes the record.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace jcdf {

constexpr int SCF_TAIL_GROUPS = 128;

__global__ __launch_bounds__(256) void k_scf_tail_partial(const double *__restrict__ D, const double *__restrict__ Dold,
                                                          const double *__restrict__ F, const double *__restrict__ H, int64_t nn,
                                                          double *__restrict__ part)
{
    __shared__ double red[2][256];
    const int64_t per = (nn + gridDim.x - 1) / gridDim.x;
    const int64_t lo = per * blockIdx.x, hi = (lo + per < nn) ? lo + per : nn;
    double e = 0.0, r = 0.0;
    for (int64_t i = lo + threadIdx.x; i < hi; i += 256) {
        const double d = D[i], dd = d - Dold[i];
        e += d * (F[i] + H[i]);
        r += dd * dd;
    }
    red[0][threadIdx.x] = e;
    red[1][threadIdx.x] = r;
    __syncthreads();
    for (int h = 128; h > 0; h >>= 1) {
        if ((int)threadIdx.x < h) {
            red[0][threadIdx.x] += red[0][threadIdx.x + h];
            red[1][threadIdx.x] += red[1][threadIdx.x + h];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        part[2 * blockIdx.x] = red[0][0];
        part[2 * blockIdx.x + 1] = red[1][0];
    }
}

// out[0] = E_elec, out[1] = ||dD||_F, out[2] = DIIS flag, out[3] = eigensolver status (|err| + |info|),
// out[4..7] = SP2 {finished, tr P, smallest pivot of the basis Cholesky, squarings}; absent inputs read as 0
__global__ __launch_bounds__(64) void k_scf_tail_final(const double *__restrict__ part, int nparts, const int32_t *diis_flag,
                                                       const int32_t *eig_err, const int32_t *eig_info, const double *sp2_info,
                                                       const double *pivot, double *__restrict__ out)
{
    // one wave: lane l adds partials l, l + 64, ... then a butterfly — the same order on every run
    double e = 0.0, r = 0.0;
    for (int p = threadIdx.x; p < nparts; p += 64) {
        e += part[2 * p];
        r += part[2 * p + 1];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        e += __shfl_xor(e, o, 64);
        r += __shfl_xor(r, o, 64);
    }
    if (threadIdx.x != 0) return;
    out[0] = 0.5 * e;
    out[1] = sqrt(r);
    out[2] = diis_flag ? (double)*diis_flag : 0.0;
    out[3] = (eig_err ? fabs((double)*eig_err) : 0.0) + (eig_info ? fabs((double)*eig_info) : 0.0);
    out[4] = sp2_info ? sp2_info[1] : 0.0;
    out[5] = sp2_info ? sp2_info[2] : 0.0;
    out[6] = pivot ? *pivot : 0.0;
    out[7] = sp2_info ? sp2_info[0] : 0.0;
}

}  // namespace jcdf

namespace jcdf {

#ifdef JCDF_DIAGNOSTIC
// ---- k_keepalive: waves that keep the CUs occupied for a given time (experiment: what the clock governor looks at) ----------
// After ~1 ms without load on most CUs (the replicated eigensolve: 64 polling workgroups) the next Fock build runs 11-17 %
// slower at IDENTICAL cycle counts per phase (tools/w_stall.py with W_STALL_GAP_MS): the shader clock has dropped and climbs
// back over several ms.  mode 0: the waves only sleep (occupancy without issue), 1: a dependent fp64 FMA chain per wave,
// 2: fp64 MFMAs back to back.  `until` = wall-clock ticks (100 MHz) to stay; `stop` (optional): leave as soon as *stop != 0.
__global__ __launch_bounds__(256) void k_keepalive(unsigned long long ticks, int mode, const int *stop, double *sink, int pause)
{
    const unsigned long long t0 = wall_clock64();
    double x = 1.0 + threadIdx.x * 1e-9;
    double4_t acc = {0.0, 0.0, 0.0, 0.0};
    for (;;) {
        if (mode == 0) {
            __builtin_amdgcn_s_sleep(127);
        } else if (mode == 1) {
#pragma unroll
            for (int i = 0; i < 64; ++i) x = x * 1.0000001 + 1e-12;
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, acc, 0, 0, 0);
        }
        for (int i = 0; i < pause; ++i) __builtin_amdgcn_s_sleep(1);    // duty cycle: `pause` x 64 clocks between two bursts
        if (wall_clock64() - t0 >= ticks) break;
        if (stop && __hip_atomic_load(stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
    }
    if (x + acc[0] == 123.456) sink[0] = x;                          // keeps the arithmetic alive
}
#endif  // JCDF_DIAGNOSTIC

}  // namespace jcdf
