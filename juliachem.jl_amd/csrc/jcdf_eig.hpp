// jcdf_eig.hpp — k_sytrd_lower: Householder tridiagonalisation of a symmetric fp64 matrix
// in ONE persistent kernel launch (caller side of the hot path, SURVEY 8 row f1: the
// replicated eigensolve of `iteration`, /root/reference/src/rhf/energy/SCF.jl:1080-1083).
//
// Why: rocSOLVER's syevd spends ~9 of its 12 ms (N = 510) in ~4000 tiny latrd/symv/syr2
// launches of the tridiagonalisation (profiles/r01_kernel_stats_bench.txt).  The algorithm
// is LAPACK dsytd2 (unblocked, 'L'): N-2 dependent steps, each a symv and a rank-2 update of
// the trailing matrix — only ~2.7e8 flops in total at N = 510, pure latency.  Here the matrix
// lives in LDS, distributed column-cyclically over G workgroups (G <= #CUs, one per CU, all
// co-resident), and the N-2 steps run inside one launch.
//
// Inter-workgroup hand-off: tagged 8-byte granules (see below) — no grid barrier, no fence; every
// spin is bounded by a wall-clock timeout that raises an error word.  Per column two all-to-all
// exchanges remain (the reflector v, then y = tau A v): ~7 us per column on MI355X, which is the
// fabric's all-gather latency (MI355X_MICROARCH price list: 8 KB all-gather ~2.4-3 us), not
// arithmetic.  A first version with two counter grid barriers per column measured the same 7.5 us.
//
// Output is LAPACK-compatible (dsytrd 'L'): D, E, TAU and the Householder vectors below the
// sub-diagonal of A, so rocSOLVER's stedc + ormtr finish the eigendecomposition.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace jcdf {

// ---- tagged hand-off granules -----------------------------------------------------------
// A handed-off double travels as two naturally aligned 8-byte words {tag32, half32}, each written
// by ONE agent-scope relaxed atomic store (sc1, write-through) and read by agent-scope relaxed
// atomic loads (sc1, bypass L1): "the data IS the flag" (cdna_hip_programming.md Guideline 16,
// recipe R2: 8-byte {tag, value} granules need no flag, no fence and no ordering).  The tag is
// the step number, so a consumer simply re-reads until both tags match; stale contents of the
// reused buffers can never be mistaken for fresh data.  This removes every grid barrier from the
// tridiagonalisation: per column there are two store->load hops (v, then y) instead of six.
typedef unsigned long long u64;

__device__ __forceinline__ void pub(u64 *g, int idx, double v, unsigned tag)
{
    const u64 bits = (u64)__double_as_longlong(v);
    __hip_atomic_store(g + 2 * idx, ((u64)tag << 32) | (bits & 0xffffffffULL), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(g + 2 * idx + 1, ((u64)tag << 32) | (bits >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Returns false on timeout / peer failure (error word set).
__device__ __forceinline__ bool sub(const u64 *g, int idx, unsigned tag, double *out, int *err)
{
    u64 a, b;
    unsigned spins = 0;
    u64 t0 = 0;
    for (;;) {
        a = __hip_atomic_load(g + 2 * idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        b = __hip_atomic_load(g + 2 * idx + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((unsigned)(a >> 32) == tag && (unsigned)(b >> 32) == tag) break;
        if (++spins == 64) t0 = wall_clock64();
        if (spins > 64 && (spins & 63) == 0) {
            if (wall_clock64() - t0 > 5000000ULL) {                              // 50 ms at 100 MHz
                __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return false;
            }
            if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return false;
        }
        __builtin_amdgcn_s_sleep(1);
    }
    *out = __longlong_as_double((long long)(((b & 0xffffffffULL) << 32) | (a & 0xffffffffULL)));
    return true;
}

// Batched form: elements idx = first + e*stride (e < 8, idx < count) are requested together, so a
// thread pays one memory round trip for all of them instead of one per element.
__device__ __forceinline__ bool sub8(const u64 *g, int first, int stride, int count, unsigned tag, double *dst, int *err)
{
    unsigned pending = 0;
#pragma unroll
    for (int e = 0; e < 8; ++e)
        if (first + e * stride < count) pending |= 1u << e;
    unsigned spins = 0;
    u64 t0 = 0;
    while (pending) {
        u64 a[8], b[8];
#pragma unroll
        for (int e = 0; e < 8; ++e)
            if (pending & (1u << e)) {
                const int idx = first + e * stride;
                a[e] = __hip_atomic_load(g + 2 * idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                b[e] = __hip_atomic_load(g + 2 * idx + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
#pragma unroll
        for (int e = 0; e < 8; ++e)
            if ((pending & (1u << e)) && (unsigned)(a[e] >> 32) == tag && (unsigned)(b[e] >> 32) == tag) {
                dst[first + e * stride] = __longlong_as_double((long long)(((b[e] & 0xffffffffULL) << 32) | (a[e] & 0xffffffffULL)));
                pending &= ~(1u << e);
            }
        if (!pending) break;
        if (++spins == 64) t0 = wall_clock64();
        if (spins > 64 && (spins & 63) == 0) {
            if (wall_clock64() - t0 > 5000000ULL) {
                __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return false;
            }
            if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return false;
        }
        __builtin_amdgcn_s_sleep(1);
    }
    return true;
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

// true iff every thread of the workgroup passes `ok`
__device__ __forceinline__ bool block_all(bool ok, double *red)
{
    return block_sum(ok ? 0.0 : 1.0, red) == 0.0;
}

// A: n x n symmetric (full storage, lda >= n).  Workspace: `err` word + granule buffers
//   vg: 2 x (n+1) granule pairs (slot n of a half carries tau), yg: 2 x n, dg: 2 x gridDim.x,
// all zeroed before launch (tags start at 1).
// LDS: (ncol_max * n + 2 n + 16) doubles, ncol_max = ceil(n / gridDim.x).
//
// Per column k:  owner publishes v_k, tau_k  ->  everyone: y = tau A22 v for its own columns,
// publishes y and its partial v.y  ->  everyone: w, rank-2 update of its own columns; the owner of
// column k+1 updates that column FIRST and publishes v_{k+1} before it updates the rest.
// Buffer reuse (parity of k) is safe without barriers: v_{k+2} can only be formed after every
// workgroup has published y_{k+1}, i.e. after it has consumed v_{k+1}, v_k and y_k.
__global__ __launch_bounds__(256) void k_sytrd_lower(double *__restrict__ A, int lda, int n, double *__restrict__ D,
                                                     double *__restrict__ E, double *__restrict__ TAU,
                                                     u64 *vg, u64 *yg, u64 *dg, int *err)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int G = gridDim.x, g = blockIdx.x, tid = threadIdx.x, nthr = blockDim.x;
    const int ncol_max = (n + G - 1) / G;
    const int nc = (n - g + G - 1) / G > 0 ? (n - g + G - 1) / G : 0;       // my columns: g, g+G, ...
    double *slab = lds;                                   // column c (global j = g + c G) at slab + c*n
    double *vs = lds + (size_t)ncol_max * n;
    double *ws = vs + n;
    double *red = ws + n;                                 // 8 doubles for reductions

    for (int c = 0; c < nc; ++c)
        for (int i = tid; i < n; i += nthr) slab[(size_t)c * n + i] = A[(size_t)(g + c * G) * lda + i];
    __syncthreads();

    // dlarfg on x = A[k+1:n, k] (column k is local column k / G of its owner); publishes v_k, tau_k
    auto reflector = [&](int k) {
        const int m = n - k - 1, buf = k & 1;
        const unsigned tag = (unsigned)(k + 1);
        u64 *vb = vg + (size_t)buf * 2 * (n + 1);
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
        if (tid == 0) pub(vb, n, tau, tag);
        for (int i = tid; i < m; i += nthr) {
            const double v = (i == 0) ? 1.0 : x[i] * scale;
            pub(vb, i, v, tag);
            if (i > 0) x[i] = v;                          // LAPACK storage of the reflector
        }
        if (tid == 0) {
            x[0] = beta;                                  // E[k] lives on the sub-diagonal
            E[k] = beta;
            TAU[k] = tau;
            D[k] = slab[(size_t)(k / G) * n + k];
        }
    };

    if (n > 1 && g == 0) reflector(0);
    for (int k = 0; k < n - 1; ++k) {
        const int m = n - k - 1;                          // rows k+1 .. n-1
        const int buf = k & 1;
        const unsigned tag = (unsigned)(k + 1);
        const u64 *vb = vg + (size_t)buf * 2 * (n + 1);
        u64 *yb = yg + (size_t)buf * 2 * n;
        u64 *db = dg + (size_t)buf * 2 * G;

        // ---- everyone: v, tau -> LDS (each thread waits for its own elements)
        bool ok = true;
        for (int i0 = tid; i0 < m && ok; i0 += 8 * nthr) ok = sub8(vb, i0, nthr, m, tag, vs, err);
        if (tid == 0 && ok) ok = sub(vb, n, tag, vs + (n - 1), err);
        if (!block_all(ok, red)) return;
        const double tau = vs[n - 1];
        const int c0 = (k + 1 - g + G - 1) / G;           // first local column with j > k
        const int wave = tid >> 6, lane = tid & 63, nw = nthr >> 6;

        // ---- y_j = tau * A22[:, j] . v for my columns j > k ; partial v.y
        double dpart = 0.0;
        if (tau != 0.0) {
            for (int c = c0 + wave; c < nc; c += nw) {
                const double *col = slab + (size_t)c * n + (k + 1);
                double s = 0.0;
                for (int i = lane; i < m; i += 64) s += col[i] * vs[i];
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
                if (lane == 0) {
                    const int j = g + c * G;
                    const double y = tau * s;
                    pub(yb, j - (k + 1), y, tag);
                    dpart += vs[j - (k + 1)] * y;
                }
            }
            dpart = block_sum(dpart, red);                // lane 0 of each wave carried its partial
            if (tid == 0) pub(db, g, dpart, tag);
        }

        // ---- everyone: w = y - (tau/2)(y.v) v ; A22[:, j] -= v w_j + w v_j for my columns j > k
        const bool next_owner = (k + 1 < n - 1) && (g == (k + 1) % G);
        if (tau != 0.0) {
            double dl = 0.0;
            for (int q = tid; q < G && ok; q += nthr) {
                double dq;
                ok = sub(db, q, tag, &dq, err);
                dl += dq;
            }
            for (int i0 = tid; i0 < m && ok; i0 += 8 * nthr) ok = sub8(yb, i0, nthr, m, tag, ws, err);   // ws = y for now
            if (!block_all(ok, red)) return;
            const double al = -0.5 * tau * block_sum(dl, red);
            for (int i = tid; i < m; i += nthr) ws[i] += al * vs[i];
            __syncthreads();
            int cfirst = c0;
            if (next_owner) {                             // column k+1 first, then its reflector goes out
                double *col = slab + (size_t)c0 * n + (k + 1);           // local column c0 is j = k+1
                for (int i = tid; i < m; i += nthr) col[i] -= vs[i] * ws[0] + ws[i] * vs[0];
                __syncthreads();
                reflector(k + 1);
                cfirst = c0 + 1;
            }
            for (int c = cfirst + wave; c < nc; c += nw) {              // one wave per column, lanes along the rows
                const int jj = g + c * G - (k + 1);
                const double wj = ws[jj], vj = vs[jj];
                double *col = slab + (size_t)c * n + (k + 1);
                for (int i = lane; i < m; i += 64) col[i] -= vs[i] * wj + ws[i] * vj;
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
