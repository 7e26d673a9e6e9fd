// jcdf_eig.hpp — k_sytrd_lower: Householder tridiagonalisation of a symmetric fp64 matrix
// in ONE persistent kernel launch (caller side of the hot path, SURVEY 8 row f1: the
// replicated eigensolve of `iteration`, /root/reference/src/rhf/energy/SCF.jl:1080-1083).
//
// Why: rocSOLVER's syevd spends ~9 of its 12 ms (N = 510) in ~4000 tiny latrd/symv/syr2
// launches of the tridiagonalisation (profiles/r01_kernel_stats_bench.txt).  The algorithm
// is LAPACK dsytd2 (unblocked, 'L'): N-2 dependent steps, each a symv and a rank-2 update of
// the trailing matrix — only ~2.7e8 flops in total at N = 510, pure latency.  Here the matrix
// lives in LDS, distributed column-cyclically over G workgroups (G <= #CUs, one per CU, all
// co-resident), and the N-2 steps run inside one launch with two grid barriers per step.
//
// Inter-workgroup hand-off (cdna_hip_programming.md Guideline 16 / MI355X_MICROARCH "Valid
// forms", table row 1): every handed-off double is written with an agent-scope relaxed
// atomic store (sc1, write-through) and read with an agent-scope relaxed atomic load (sc1,
// bypasses L1); each storing wave drains with s_waitcnt vmcnt(0), the workgroup barriers, one
// lane adds to a monotonic agent-scope counter and polls it; the other waves continue after a
// workgroup barrier.  Every spin is bounded by a wall-clock timeout that raises an error word.
//
// Output is LAPACK-compatible (dsytrd 'L'): D, E, TAU and the Householder vectors below the
// sub-diagonal of A, so rocSOLVER's stedc + ormtr finish the eigendecomposition.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace jcdf {

__device__ __forceinline__ void st_sc1(double *p, double v)
{
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), (unsigned long long)__double_as_longlong(v),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double ld_sc1(const double *p)
{
    return __longlong_as_double((long long)__hip_atomic_load(
        reinterpret_cast<const unsigned long long *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

// Monotonic-counter grid barrier.  Returns false on timeout (error word set).
// `flag` is one double of the kernel's dynamic LDS (no static __shared__: it would shift the
// dynamic base off its 16-B alignment, Guideline 17).
__device__ __forceinline__ bool grid_barrier(unsigned long long *bar, unsigned long long target, int *err, double *flag)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every storing wave drains its sc1 stores
    __syncthreads();
    if (threadIdx.x == 0) {
        int ok = 1;
        __hip_atomic_fetch_add(bar, 1ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long t0 = wall_clock64();                       // 100 MHz
        while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (wall_clock64() - t0 > 5000000ULL) {                         // 50 ms: a peer is gone
                __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = 0;
                break;
            }
            if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { ok = 0; break; }
        }
        *flag = ok ? 1.0 : 0.0;
    }
    __syncthreads();
    return *flag != 0.0;
}

__device__ __forceinline__ double block_sum(double x, double *red)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = x;
    __syncthreads();
    double s = 0.0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += red[w];
    return s;
}

// Wait until the agent-scope word *flag reaches `want` (one lane polls, bounded).
__device__ __forceinline__ bool flag_wait(const unsigned long long *flag, unsigned long long want, int *err, double *lflag)
{
    if (threadIdx.x == 0) {
        int ok = 1;
        const unsigned long long t0 = wall_clock64();
        while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
            __builtin_amdgcn_s_sleep(1);
            if (wall_clock64() - t0 > 5000000ULL) {
                __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = 0;
                break;
            }
            if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { ok = 0; break; }
        }
        *lflag = ok ? 1.0 : 0.0;
    }
    __syncthreads();
    return *lflag != 0.0;
}

// A: n x n symmetric (full storage, lda >= n).  Workspace (device; the 64-byte header holding
// `bar`, `err`, `vflag[2]` is zeroed before launch):
//   vbuf, ybuf: 2*n doubles each (slot n-1 of a vbuf half carries tau); dots: 2*gridDim.x doubles.
// LDS: (ncol_max * n + 2 n + 16) doubles, ncol_max = ceil(n / gridDim.x).
//
// Per step k: [owner of column k has published its reflector v_k (flag)] -> everyone: y = tau A22 v
// for its own columns -> ONE grid barrier -> everyone: w, rank-2 update of its own columns; the
// owner of column k+1 updates that column FIRST, builds and publishes v_{k+1}, and only then
// updates the rest, so the next step's reflector is in flight while the others still update.
__global__ __launch_bounds__(256) void k_sytrd_lower(double *__restrict__ A, int lda, int n, double *__restrict__ D,
                                                     double *__restrict__ E, double *__restrict__ TAU,
                                                     double *vbuf, double *ybuf, double *dots,
                                                     unsigned long long *bar, unsigned long long *vflag, int *err)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int G = gridDim.x, g = blockIdx.x, tid = threadIdx.x, nthr = blockDim.x;
    const int ncol_max = (n + G - 1) / G;
    const int nc = (n - g + G - 1) / G > 0 ? (n - g + G - 1) / G : 0;       // my columns: g, g+G, ...
    double *slab = lds;                                   // column c (global j = g + c G) at slab + c*n
    double *vs = lds + (size_t)ncol_max * n;
    double *ws = vs + n;
    double *red = ws + n;                                 // 8 doubles for reductions + 1 barrier flag
    double *bflag = red + 8;

    for (int c = 0; c < nc; ++c)
        for (int i = tid; i < n; i += nthr) slab[(size_t)c * n + i] = A[(size_t)(g + c * G) * lda + i];
    __syncthreads();

    // dlarfg on x = A[k+1:n, k] (column k is local column k / G of its owner); publishes v_k, tau_k
    auto reflector = [&](int k) {
        const int m = n - k - 1, buf = k & 1;
        double *vb = vbuf + (size_t)buf * n;
        double *x = slab + (size_t)(k / G) * n + (k + 1);
        double part = 0.0;
        for (int i = 1 + tid; i < m; i += nthr) part += x[i] * x[i];
        const double xnorm2 = block_sum(part, red);
        const double alpha = x[0];
        double tau = 0.0, beta = alpha, scale = 0.0;
        if (xnorm2 != 0.0) {
            beta = -copysign(sqrt(alpha * alpha + xnorm2), alpha);
            tau = (beta - alpha) / beta;
            scale = 1.0 / (alpha - beta);
        }
        __syncthreads();
        for (int i = tid; i < m; i += nthr) {
            const double v = (i == 0) ? 1.0 : x[i] * scale;
            st_sc1(vb + i, v);
            if (i > 0) x[i] = v;                          // LAPACK storage of the reflector
        }
        if (tid == 0) {
            x[0] = beta;                                  // E[k] lives on the sub-diagonal
            st_sc1(vb + (n - 1), tau);
            E[k] = beta;
            TAU[k] = tau;
            D[k] = slab[(size_t)(k / G) * n + k];
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave drains, then ONE lane flags
        __syncthreads();
        if (tid == 0)
            __hip_atomic_store(vflag + buf, (unsigned long long)(k + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };

    if (n > 1 && g == 0) reflector(0);
    unsigned long long phase = 0;
    for (int k = 0; k < n - 1; ++k) {
        const int m = n - k - 1;                          // rows k+1 .. n-1
        const int buf = k & 1;
        const double *vb = vbuf + (size_t)buf * n;
        double *yb = ybuf + (size_t)buf * n;
        if (!flag_wait(vflag + buf, (unsigned long long)(k + 1), err, bflag)) return;

        // ---- everyone: v, tau -> LDS; y_j = tau * A22[:, j] . v for my columns j > k
        for (int i = tid; i < m; i += nthr) vs[i] = ld_sc1(vb + i);
        if (tid == 0) vs[n - 1] = ld_sc1(vb + (n - 1));
        __syncthreads();
        const double tau = vs[n - 1];
        const int c0 = (k + 1 - g + G - 1) / G;           // first local column with j > k
        double dpart = 0.0;
        if (tau != 0.0) {
            const int wave = tid >> 6, lane = tid & 63, nw = nthr >> 6;
            for (int c = c0 + wave; c < nc; c += nw) {
                const double *col = slab + (size_t)c * n + (k + 1);
                double s = 0.0;
                for (int i = lane; i < m; i += 64) s += col[i] * vs[i];
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
                if (lane == 0) {
                    const int j = g + c * G;
                    const double y = tau * s;
                    st_sc1(yb + (j - (k + 1)), y);
                    dpart += vs[j - (k + 1)] * y;
                }
            }
        }
        dpart = block_sum(dpart, red);                    // lane 0 of each wave carried its partial
        if (tid == 0) st_sc1(dots + (size_t)buf * G + g, dpart);
        if (!grid_barrier(bar, (++phase) * (unsigned long long)G, err, bflag)) return;

        // ---- everyone: w = y - (tau/2)(y.v) v ; A22[:, j] -= v w_j + w v_j for my columns j > k
        const bool next_owner = (k + 1 < n - 1) && (g == (k + 1) % G);
        if (tau != 0.0) {
            double dl = 0.0;
            for (int q = tid; q < G; q += nthr) dl += ld_sc1(dots + (size_t)buf * G + q);
            const double al = -0.5 * tau * block_sum(dl, red);
            for (int i = tid; i < m; i += nthr) ws[i] = ld_sc1(yb + i) + al * vs[i];
            __syncthreads();
            int cfirst = c0;
            if (next_owner) {                             // column k+1 first, then its reflector goes out
                double *col = slab + (size_t)c0 * n + (k + 1);           // local column c0 is j = k+1
                for (int i = tid; i < m; i += nthr) col[i] -= vs[i] * ws[0] + ws[i] * vs[0];
                __syncthreads();
                reflector(k + 1);
                cfirst = c0 + 1;
            }
            {   // one wave per column, lanes along the rows (no integer division in the hot loop)
                const int wave = tid >> 6, lane = tid & 63, nw = nthr >> 6;
                for (int c = cfirst + wave; c < nc; c += nw) {
                    const int jj = g + c * G - (k + 1);
                    const double wj = ws[jj], vj = vs[jj];
                    double *col = slab + (size_t)c * n + (k + 1);
                    for (int i = lane; i < m; i += 64) col[i] -= vs[i] * wj + ws[i] * vj;
                }
            }
        } else if (next_owner) {
            reflector(k + 1);
        }
        __syncthreads();
    }
    if (g == (n - 1) % G && tid == 0) D[n - 1] = slab[(size_t)((n - 1) / G) * n + (n - 1)];
    __syncthreads();
    for (int c = 0; c < nc; ++c)
        for (int i = tid; i < n; i += nthr) A[(size_t)(g + c * G) * lda + i] = slab[(size_t)c * n + i];
}

}  // namespace jcdf
