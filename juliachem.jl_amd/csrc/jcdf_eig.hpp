// jcdf_eig.hpp — k_sytrd_lower / k_sytrd_onehop (+ k_sytd2_tail): Householder tridiagonalisation of a symmetric
// fp64 matrix in ONE persistent kernel launch and a one-workgroup finish (caller side of the hot path, SURVEY 8 row f1: the
// replicated eigensolve of `iteration`, /root/reference/src/rhf/energy/SCF.jl:1080-1083).
//
// Why: rocSOLVER's syevd spends ~9 of its 12 ms (N = 510) in ~4000 tiny latrd/symv/syr2
// launches of the tridiagonalisation.  The algorithm is LAPACK dsytd2 (unblocked, 'L'): N-2
// dependent steps, each a symv and a rank-2 update of the trailing matrix — only ~2.7e8 flops in
// total at N = 510, pure latency.  Here the matrix lives in LDS, distributed column-cyclically over
// G workgroups (G <= #CUs, one per CU, all co-resident), and the N-2 steps run inside one launch.
//
// Inter-workgroup hand-off: tagged 8-byte granules (see below) — no grid barrier, no fence; every
// spin is bounded by a wall-clock timeout that raises an error word.  One round of agent-scope polls
// is a ~1.4 us round trip inside these kernels, so what counts is the number of DEPENDENT poll
// rounds per column: k_sytrd_lower has two (reflector v + tau, then y = tau A v), k_sytrd_onehop one
// (every workgroup forms the reflector itself from a column that was broadcast a step ahead).
// History at N = 510, us per column: 7.5 (two counter grid barriers) -> 7.0 (granules) -> 5.9 (reductions
// by DPP instead of ds_bpermute, Q accumulated in the kernel) -> 5.0 (tau travels with v) -> 4.6 (one hop)
// -> 4.1 over the whole matrix once the LAST 128 columns are reduced by one workgroup from its register file
// (k_sytd2_tail below: 1.7 us per column, no hand-off; k_q_tail_reflect applies its reflectors to Q)
// -> 3.5 with 512 threads in the one-hop kernel: one 64-lane wave per column in the fused pass and per row of Q — what
// a step waits for is its dependent chain, and the per-lane length of these two loops was a third of it (the polled volume
// and the sharing of its cache lines are not: profiles/r03_sytd2_tail.txt).
//
// Output is LAPACK-compatible (dsytrd 'L'): D, E, TAU and the Householder vectors below the
// sub-diagonal of A (handed-off values rounded to 50 mantissa bits, see below), plus optionally the
// accumulated orthogonal factor Q, so the eigenvectors are one GEMM Q Z after the tridiagonal solve.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace jcdf {

// ---- tagged hand-off granules -----------------------------------------------------------
// A handed-off double travels as ONE naturally aligned 8-byte word written by ONE agent-scope relaxed
// atomic store (sc1, write-through) and read by agent-scope relaxed atomic loads (sc1, bypass L1):
// "the data IS the flag" (cdna_hip_programming.md Guideline 16, recipe R2: tagged 8-byte granules need
// no flag, no fence and no ordering).  The tag is a 2-bit sequence number kept in the two lowest
// mantissa bits: sender and receivers all compute with the value whose two low bits are zero (a
// relative perturbation <= 3 * 2^-52, the size of one rounding error of the dot products that
// produce these numbers), so every workgroup sees bit-identical data.  A slot of buffer (k & 1) is
// rewritten at EVERY step k (also when tau == 0), so the word it held before carries the tag of step
// k - 2, which differs: stale contents can never be mistaken for fresh data, and the zero-filled
// initial state carries tag 0 while steps 0 and 1 use tag 1.  Half the words of a {tag32, half32}
// encoding: the hand-offs are bound by the number of polled words (tools/xcd_pingpong.hip).
typedef unsigned long long u64;

#ifdef JCDF_SYTRD_PROFILE   // tools/sytrd_prof.hip: per-phase wall-clock ticks (100 MHz) of workgroup 1, summed over columns
__device__ u64 g_sytrd_prof[8];
#define SYTRD_TICK(slot) do { if (g == 1 && tid == 0) { const u64 t_ = wall_clock64(); g_sytrd_prof[slot] += t_ - tprof; tprof = t_; } } while (0)
#else
#define SYTRD_TICK(slot) do { } while (0)
#endif

__device__ __forceinline__ unsigned step_tag(int k) { return (unsigned)((k >> 1) + 1) & 3u; }

__device__ __forceinline__ double tag_trunc(double v)
{
    return __longlong_as_double(__double_as_longlong(v) & ~3LL);
}

__device__ __forceinline__ void pub(u64 *g, int idx, double v, unsigned tag)
{
    __hip_atomic_store(g + idx, ((u64)__double_as_longlong(v) & ~3ULL) | tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// bounded spin bookkeeping shared by the subscribers: false = give up (timeout or a peer failed)
__device__ __forceinline__ bool spin_ok(unsigned &spins, u64 &t0, int *err)
{
    if (++spins == 64) t0 = wall_clock64();
    if (spins > 64 && (spins & 63) == 0) {
        if (wall_clock64() - t0 > 5000000ULL) {                                  // 50 ms at 100 MHz
            __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return false;
        }
        if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return false;
    }
    __builtin_amdgcn_s_sleep(1);
    return true;
}

// granule pairs (first + e*stride, +1), e < NP, below `count`; `beat` (optional): one more word, only its tag matters.
// (16-byte polls — two granules per request — were measured too: no gain inside this kernel, tools/xcd_pingpong.hip
// has the stand-alone numbers; every poll here is an 8-byte agent-scope atomic load.)
// SYTRD_POLL_DEPTH attempts can be kept in flight, SYTRD_POLL_GAP s_sleep units apart (a poll is a ~1.4 us round
// trip, so a word that lands just after an attempt was issued is seen a whole round trip later) — measured: every
// extra attempt in flight makes the kernel SLOWER (depth 2: +7 %, depth 4: +15 %), and so does any delay before the
// first attempt; polls on a line that is about to be written compete with that write.  Depth 1, no delay.
#ifndef SYTRD_POLL_DEPTH
#define SYTRD_POLL_DEPTH 1
#endif
#ifndef SYTRD_POLL_GAP
#define SYTRD_POLL_GAP 4
#endif
template <int NP>
__device__ __forceinline__ bool sub_pairs(const u64 *g, int first, int stride, int count, unsigned tag, double *dst, int *err,
                                          const u64 *beat, unsigned btag)
{
    unsigned pending = beat ? 1u << 8 : 0u;
    const u64 *p[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const bool in = e < NP && first + e * stride < count;
        p[e] = g + (in ? first + e * stride : 0);                             // slots beyond the end re-read pair 0 (unused)
        if (in) pending |= 1u << (2 * e);
        if (in && first + e * stride + 1 < count) pending |= 2u << (2 * e);
    }
    unsigned spins = 0;
    u64 t0 = 0;
    while (pending) {
        u64 lo[SYTRD_POLL_DEPTH][NP], hi[SYTRD_POLL_DEPTH][NP], b[SYTRD_POLL_DEPTH];
#pragma unroll
        for (int a = 0; a < SYTRD_POLL_DEPTH; ++a) {
            if (a > 0) __builtin_amdgcn_s_sleep(SYTRD_POLL_GAP);
            b[a] = (pending & (1u << 8)) ? __hip_atomic_load(beat, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
#pragma unroll
            for (int e = 0; e < NP; ++e) {
                lo[a][e] = (pending & (1u << (2 * e))) ? __hip_atomic_load(p[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
                hi[a][e] = (pending & (2u << (2 * e))) ? __hip_atomic_load(p[e] + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
            }
        }
#pragma unroll
        for (int a = 0; a < SYTRD_POLL_DEPTH; ++a) {                          // oldest attempt first (loads return in order)
#pragma unroll
            for (int e = 0; e < NP; ++e) {
                if ((pending & (1u << (2 * e))) && (unsigned)(lo[a][e] & 3ULL) == tag) {
                    dst[first + e * stride] = __longlong_as_double((long long)(lo[a][e] & ~3ULL));
                    pending &= ~(1u << (2 * e));
                }
                if ((pending & (2u << (2 * e))) && (unsigned)(hi[a][e] & 3ULL) == tag) {
                    dst[first + e * stride + 1] = __longlong_as_double((long long)(hi[a][e] & ~3ULL));
                    pending &= ~(2u << (2 * e));
                }
            }
            if ((pending & (1u << 8)) && (unsigned)(b[a] & 3ULL) == btag) pending &= ~(1u << 8);
            if (!pending) break;
        }
        if (!pending) break;
        if (!spin_ok(spins, t0, err)) return false;
    }
    return true;
}

// all `count` granules of a buffer (+ optional heartbeat word) by the whole workgroup, adjacent pairs per lane
__device__ __forceinline__ bool sub_all(const u64 *g, int count, unsigned tag, double *dst, int *err, const u64 *beat = nullptr,
                                        unsigned btag = 0)
{
    const int tid = threadIdx.x, nthr = blockDim.x;
    bool ok = true;
    if (count <= 2 * nthr) return sub_pairs<1>(g, 2 * tid, 2 * nthr, count, tag, dst, err, beat, btag);
    if (count <= 4 * nthr) return sub_pairs<2>(g, 2 * tid, 2 * nthr, count, tag, dst, err, beat, btag);
    for (int i0 = 2 * tid, first = 1; (i0 < count || first) && ok; i0 += 8 * nthr, first = 0)
        ok = sub_pairs<4>(g, i0, 2 * nthr, count, tag, dst, err, first ? beat : nullptr, btag);
    return ok;
}

// ---- reductions on the VALU (DPP), not through the LDS crossbar ------------------------------
// (a double __shfl_xor is two ds_bpermute_b32; with four waves reducing eight values each the LDS
// pipe, not arithmetic, set the time of a step: tools/sytrd_prof.hip)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_f64(double x)
{
    const long long b = __double_as_longlong(x);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffLL), CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, ROW_MASK, 0xf, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);       // +0.0 in rows outside ROW_MASK
}

// after this every lane holds the sum over its row of 16 lanes
__device__ __forceinline__ double row16_sum(double x)
{
    x += dpp_f64<0xB1, 0xf>(x);      // quad_perm [1,0,3,2]
    x += dpp_f64<0x4E, 0xf>(x);      // quad_perm [2,3,0,1]
    x += dpp_f64<0x141, 0xf>(x);     // row_half_mirror
    x += dpp_f64<0x140, 0xf>(x);     // row_mirror
    return x;
}

// lane 31 <- sum over lanes 0..31, lane 63 <- sum over lanes 32..63 (other lanes: partial values)
__device__ __forceinline__ double half_sums(double x)
{
    x = row16_sum(x);
    x += dpp_f64<0x142, 0xa>(x);     // row_bcast:15 into rows 1 and 3
    return x;
}

// sum over the 64 lanes, returned in every lane
__device__ __forceinline__ double wave_sum(double x)
{
    x = half_sums(x);
    x += dpp_f64<0x143, 0xc>(x);     // row_bcast:31 into rows 2 and 3
    const long long b = __double_as_longlong(x);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffLL), 63);
    const int hi = __builtin_amdgcn_readlane((int)(b >> 32), 63);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}

// Block-wide sums with ONE barrier: `red` holds two alternating banks of 16 doubles, `rs` is a
// per-thread call counter (call N+2 may overwrite bank N&1 because every thread passed the barrier
// of call N+1 after it read bank N&1).
__device__ __forceinline__ double block_sum(double x, double *red, unsigned &rs)
{
    double *bank = red + 16 * (rs++ & 1);
    x = wave_sum(x);
    if ((threadIdx.x & 63) == 0) bank[threadIdx.x >> 6] = x;
    __syncthreads();
    double s = 0.0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += bank[w];
    return s;
}

__device__ __forceinline__ void block_sum2(double &x, double &y, double *red, unsigned &rs)
{
    double *bank = red + 16 * (rs++ & 1);
    x = wave_sum(x);
    y = wave_sum(y);
    if ((threadIdx.x & 63) == 0) {
        bank[threadIdx.x >> 6] = x;
        bank[8 + (threadIdx.x >> 6)] = y;
    }
    __syncthreads();
    double sx = 0.0, sy = 0.0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) {
        sx += bank[w];
        sy += bank[8 + w];
    }
    x = sx;
    y = sy;
}

// true iff every thread of the workgroup passes `ok`
__device__ __forceinline__ bool block_all(bool ok, double *red, unsigned &rs)
{
    return block_sum(ok ? 0.0 : 1.0, red, rs) == 0.0;
}

// dlarfg scalars without IEEE division / square root (their expansions are ~25 dependent instructions each and this
// chain is on the critical path of every step): y = rsqrt(alpha^2 + sigma) and r = 1 / (|alpha| + norm) by the
// hardware estimates + two Newton steps (full double precision), then
//   beta = -sign(alpha) norm,  tau = (beta - alpha) / beta = 1 + |alpha| y,  scale = 1 / (alpha - beta) = sign(alpha) r.
__device__ __forceinline__ void house_scalars(double alpha, double sigma, double &tau, double &beta, double &scale)
{
    tau = 0.0;
    beta = alpha;
    scale = 0.0;
    if (sigma != 0.0) {
        const double x = alpha * alpha + sigma;
        double y = __builtin_amdgcn_rsq(x);
        y = y * (1.5 - 0.5 * x * y * y);
        y = y * (1.5 - 0.5 * x * y * y);
        const double norm = x * y, aa = fabs(alpha);
        const double dn = aa + norm;
        double r = __builtin_amdgcn_rcp(dn);
        r = r * (2.0 - dn * r);
        r = r * (2.0 - dn * r);
        beta = -copysign(norm, alpha);
        tau = 1.0 + aa * y;
        scale = copysign(r, alpha);
    }
}

constexpr int SYTRD_CB = 8;      // local columns processed together (independent accumulators)

// A: n x n symmetric (full storage, lda >= n).  Workspace: `err` word + granule buffers
//   vg: 2 x (n+2) granules (v_k in slots 0..m-1, tau_k in slot m), yg: 2 x roundup(n,2), hg: 2 x gridDim.x (heartbeats),
// all zeroed before launch.
// LDS: ((Qout ? 2 : 1) * ncol_max * n + 2 n + 32) doubles, ncol_max = ceil(n / gridDim.x).
// Qout (optional, n x n row-major): the orthogonal matrix Q = H_0 H_1 ... H_{n-3} of A = Q T Q^T,
// accumulated on the fly (rows distributed like the columns of A), so eigenvectors of A are Q Z.
//
// Per column k:  owner publishes v_k, tau_k  ->  everyone: y = tau A22 v for its own columns (32 lanes
// per column, SYTRD_CB columns at a time, DPP reduction, no barrier), publishes y  ->  everyone: y.v
// (redundantly, no exchange), w, rank-2 update of its own columns; the owner of column k+1 updates that column FIRST
// (the norm of the new reflector is accumulated in the same pass) and publishes v_{k+1} before it
// updates the rest.
// Buffer reuse (parity of k) is safe without barriers: whoever publishes an item (k+1) has taken y_k, and
// a workgroup publishes y_k only after it has consumed v_k and every item (k-1).  A workgroup whose
// columns are all finished publishes no y any more but still needs every later reflector for its rows
// of Q and must not be overrun: from then on it publishes a heartbeat (k) instead, and the others take
// its heartbeat (k-1) together with y_k (one step old when asked for, so it costs no waiting, and only
// the last ~gridDim.x steps have such workgroups at all).
__global__ __launch_bounds__(512) void k_sytrd_lower(double *__restrict__ A, int lda, int n, double *__restrict__ D,
                                                     double *__restrict__ E, double *__restrict__ TAU,
                                                     u64 *vg, u64 *yg, u64 *hg, int *err, double *__restrict__ Qout, int ldq, int kstop)
{
    // kstop: columns 0 .. kstop-1 are reduced here (kstop >= n-1: all of them); with kstop < n-1 the kernel leaves the
    // trailing block A[kstop:, kstop:] with every update applied, for k_sytd2_tail
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int G = gridDim.x, g = blockIdx.x, tid = threadIdx.x, nthr = blockDim.x;
    const int ncol_max = (n + G - 1) / G;
    const int nc = (n - g + G - 1) / G > 0 ? (n - g + G - 1) / G : 0;       // my columns: g, g+G, ...
    const int Ga = n < G ? n : G;                         // workgroups that own columns and take part
    const int kend = kstop < n - 1 ? kstop : n - 1;
    if (nc == 0) return;                                  // owns nothing; nobody waits for it
    double *slab = lds;                                   // column c (global j = g + c G) at slab + c*n
    double *vs = lds + (size_t)ncol_max * n;
    double *ws = vs + n;
    double *red = ws + n;                                 // 2 x 16 doubles for reductions
    double *qrow = red + 32;                              // Qout != nullptr: my rows g, g+G, ... of Q, row r at qrow + r*n
    unsigned rs = 0;
    if (Qout)
        for (int r = 0; r < nc; ++r)
            for (int i = tid; i < n; i += nthr) qrow[(size_t)r * n + i] = (i == g + r * G) ? 1.0 : 0.0;

    for (int c = 0; c < nc; ++c)
        for (int i = tid; i < n; i += nthr) slab[(size_t)c * n + i] = A[(size_t)(g + c * G) * lda + i];
    __syncthreads();

    // dlarfg on x = A[k+1:n, k] (column k is local column k / G of its owner) given xnorm2 = |x[1:]|^2;
    // publishes v_k, tau_k.  Called by every thread of the owner.
    auto reflector = [&](int k, double xnorm2) {
        const int m = n - k - 1, buf = k & 1;
        const unsigned tag = step_tag(k);
        u64 *vb = vg + (size_t)buf * (n + 2);                               // (n + 2: even, 16-byte aligned halves)
        double *x = slab + (size_t)(k / G) * n + (k + 1);
        const double alpha = x[0];
        double tau = 0.0, beta = alpha, scale = 0.0;
        if (xnorm2 != 0.0) {
            beta = -copysign(sqrt(alpha * alpha + xnorm2), alpha);
            tau = tag_trunc((beta - alpha) / beta);
            scale = 1.0 / (alpha - beta);
        }
        __syncthreads();                                  // everyone has read x[0]
        if (tid == 0) pub(vb, m, tau, tag);                                // slot m, right behind the m entries of v
        for (int i = tid; i < m; i += nthr) {
            const double v = (i == 0) ? 1.0 : tag_trunc(x[i] * scale);
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

#ifdef JCDF_SYTRD_PROFILE
    u64 tprof = wall_clock64();
#endif
    if (n > 1 && kend > 0 && g == 0) {
        double part = 0.0;
        for (int i = 2 + tid; i < n; i += nthr) part += slab[i] * slab[i];
        reflector(0, block_sum(part, red, rs));
    }
    for (int k = 0; k < kend; ++k) {
        const int m = n - k - 1;                          // rows k+1 .. n-1
        const int buf = k & 1;
        const unsigned tag = step_tag(k);
        const u64 *vb = vg + (size_t)buf * (n + 2);
        u64 *yb = yg + (size_t)buf * ((n + 1) & ~1);                        // even stride: 16-byte aligned halves

        // ---- everyone: v, tau -> LDS (each thread waits for its own elements)
        bool ok = true;
        ok = sub_all(vb, m + 1, tag, vs, err);                                // v and tau (slot m)
        if (!block_all(ok, red, rs)) return;
        SYTRD_TICK(0);                                    // waited for v
        const double tau = vs[m];
        const int c0 = (k + 1 - g + G - 1) / G;           // first local column with j > k
        const bool next_owner = (k + 1 < kend) && (g == (k + 1) % G);

        {   // tau == 0 (nothing to annihilate) runs the same exchange with y = 0: every slot is rewritten every step
            // ---- y_j = tau * A22[:, j] . v for my columns j > k: one 64-lane WAVE per column (as in k_sytrd_onehop: the per-lane
            //      length of this loop is on the dependent chain of every step; a 32-lane group per column left half the waves idle)
            {
                const int ln = tid & 63, wv = tid >> 6, nwv = nthr >> 6;      // nwv columns per pass
                for (int cb = c0; cb < nc; cb += nwv) {
                    const bool have = cb + wv < nc;
                    const double *col = slab + (size_t)(have ? cb + wv : c0) * n + (k + 1);
                    double s0 = 0.0, s1 = 0.0;
                    int i = ln;
                    for (; i + 64 < m; i += 128) {
                        s0 += col[i] * vs[i];
                        s1 += col[i + 64] * vs[i + 64];
                    }
                    if (i < m) s0 += col[i] * vs[i];
                    const double s = wave_sum(s0 + s1);
                    if (ln == 0 && have) {
                        const int j = g + (cb + wv) * G;
                        pub(yb, j - (k + 1), tau * s, tag);
                    }
                }
            }
            SYTRD_TICK(1);                                // y for my columns

            // ---- while y travels: Q <- Q H_k on my rows of Q (needs only v_k; same lanes own the same
            //      elements in every step, so no barrier).  Replaces the ormtr back-transformation by one GEMM.
            if (Qout) {
                const int ln = tid & 63, wv = tid >> 6, nwv = nthr >> 6;      // one wave per row of Q
                for (int rb = 0; rb < nc; rb += nwv) {
                    const bool have = rb + wv < nc;
                    double *q = qrow + (size_t)(have ? rb + wv : 0) * n + (k + 1);
                    double s0 = 0.0, s1 = 0.0;
                    int i = ln;
                    for (; i + 64 < m; i += 128) {
                        s0 += q[i] * vs[i];
                        s1 += q[i + 64] * vs[i + 64];
                    }
                    if (i < m) s0 += q[i] * vs[i];
                    const double sc = tau * wave_sum(s0 + s1);
                    if (have)
                        for (int i2 = ln; i2 < m; i2 += 64) q[i2] -= sc * vs[i2];
                }
            }
            SYTRD_TICK(4);                                // Q accumulation

            // ---- everyone: full y -> LDS, y.v, w = y - (tau/2)(y.v) v
            ok = sub_all(yb, m, tag, ws, err,                                  // ws = y for now
                         (k >= 1 && tid < Ga && tid + ((n - 1 - tid) / G) * G <= k)    // workgroup `tid` has no column > k
                             ? hg + (size_t)((k - 1) & 1) * G + tid : nullptr, step_tag(k - 1));
            double bad = ok ? 0.0 : 1.0, dot = 0.0;
            if (ok)
                for (int i = 2 * tid; i < m; i += 2 * nthr) {               // each thread re-reads only what it wrote (sub_all)
                    dot += ws[i] * vs[i];
                    if (i + 1 < m) dot += ws[i + 1] * vs[i + 1];
                }
            block_sum2(bad, dot, red, rs);
            if (bad != 0.0) return;
            if (tid == 0 && g + (nc - 1) * G <= k + 1)                      // from the step before my last y on:
                pub(hg + (size_t)(k & 1) * G, g, 0.0, tag);                 // heartbeat "v_k and y_k are behind me"
            SYTRD_TICK(2);                                // waited for y
            const double al = -0.5 * tau * dot;
            for (int i = tid; i < m; i += nthr) ws[i] += al * vs[i];
            __syncthreads();

            // ---- A22[:, j] -= v w_j + w v_j for my columns j > k
            int cfirst = c0;
            if (next_owner) {                             // column k+1 first, then its reflector goes out
                double *col = slab + (size_t)c0 * n + (k + 1);           // local column c0 is j = k+1
                const double w0 = ws[0], v0 = vs[0];
                double part = 0.0;
                for (int i = tid; i < m; i += nthr) {
                    const double x = col[i] - (vs[i] * w0 + ws[i] * v0);
                    col[i] = x;
                    if (i >= 2) part += x * x;
                }
                reflector(k + 1, block_sum(part, red, rs));
                cfirst = c0 + 1;
            }
            for (int cb = cfirst; cb < nc; cb += SYTRD_CB) {
                const int ne = nc - cb;
                double wj[SYTRD_CB], vj[SYTRD_CB];
#pragma unroll
                for (int e = 0; e < SYTRD_CB; ++e) {
                    const int jj = g + (cb + (e < ne ? e : 0)) * G - (k + 1);
                    wj[e] = ws[jj];
                    vj[e] = vs[jj];
                }
                double *base = slab + (size_t)cb * n + (k + 1);
                for (int i = tid; i < m; i += nthr) {
                    const double vi = vs[i], wi = ws[i];
#pragma unroll
                    for (int e = 0; e < SYTRD_CB; ++e)
                        if (e < ne) base[(size_t)e * n + i] -= vi * wj[e] + wi * vj[e];
                }
            }
        }
        __syncthreads();
        SYTRD_TICK(3);                                    // rank-2 update (+ reflector when next owner)
    }
    if (kend == n - 1 && g == (n - 1) % G && tid == 0) D[n - 1] = slab[(size_t)((n - 1) / G) * n + (n - 1)];
    __syncthreads();
    for (int c = 0; c < nc; ++c)
        for (int i = tid; i < n; i += nthr) A[(size_t)(g + c * G) * lda + i] = slab[(size_t)c * n + i];
    if (Qout)
        for (int r = 0; r < nc; ++r)
            for (int i = tid; i < n; i += nthr) Qout[(size_t)(g + r * G) * ldq + i] = qrow[(size_t)r * n + i];
}

// ---- the same, two buffers polled together (one round trip): pairs of both, + optional heartbeat ---------------
// NB = pairs-per-buffer a thread takes (count <= 2 NB blockDim): ALL of them are in flight at once — one poll round trip per
// step, not one per pair (with three sequential rounds per step the one-exchange kernel lost to the two-exchange kernel
// above n ~ 700).
template <int NB>
__device__ __forceinline__ bool sub_two_n(const u64 *g1, unsigned tag1, double *dst1, const u64 *g2, unsigned tag2, double *dst2,
                                          int count, int *err, const u64 *beat, unsigned btag)
{
    const int tid = threadIdx.x, nthr = blockDim.x;
    unsigned pending[NB];
    bool any = false;
#pragma unroll
    for (int e = 0; e < NB; ++e) {
        const int i0 = 2 * tid + 2 * nthr * e;
        pending[e] = 0u;
        if (i0 < count) pending[e] |= 1u | (g2 ? 4u : 0u);
        if (i0 + 1 < count) pending[e] |= 2u | (g2 ? 8u : 0u);
        any = any || pending[e] != 0u;
    }
    bool hb_pending = beat != nullptr;
    unsigned spins = 0;
    u64 t0 = 0;
    while (any || hb_pending) {
        u64 a0[NB], a1[NB], b0[NB], b1[NB];
#pragma unroll
        for (int e = 0; e < NB; ++e) {
            const int i0 = 2 * tid + 2 * nthr * e;
            a0[e] = (pending[e] & 1u) ? __hip_atomic_load(g1 + i0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
            a1[e] = (pending[e] & 2u) ? __hip_atomic_load(g1 + i0 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
            b0[e] = (pending[e] & 4u) ? __hip_atomic_load(g2 + i0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
            b1[e] = (pending[e] & 8u) ? __hip_atomic_load(g2 + i0 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
        }
        const u64 hb = hb_pending ? __hip_atomic_load(beat, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
        any = false;
#pragma unroll
        for (int e = 0; e < NB; ++e) {
            const int i0 = 2 * tid + 2 * nthr * e;
            if ((pending[e] & 1u) && (unsigned)(a0[e] & 3ULL) == tag1) { dst1[i0] = __longlong_as_double((long long)(a0[e] & ~3ULL)); pending[e] &= ~1u; }
            if ((pending[e] & 2u) && (unsigned)(a1[e] & 3ULL) == tag1) { dst1[i0 + 1] = __longlong_as_double((long long)(a1[e] & ~3ULL)); pending[e] &= ~2u; }
            if ((pending[e] & 4u) && (unsigned)(b0[e] & 3ULL) == tag2) { dst2[i0] = __longlong_as_double((long long)(b0[e] & ~3ULL)); pending[e] &= ~4u; }
            if ((pending[e] & 8u) && (unsigned)(b1[e] & 3ULL) == tag2) { dst2[i0 + 1] = __longlong_as_double((long long)(b1[e] & ~3ULL)); pending[e] &= ~8u; }
            any = any || pending[e] != 0u;
        }
        if (hb_pending && (unsigned)(hb & 3ULL) == btag) hb_pending = false;
        if (!any && !hb_pending) break;
        if (!spin_ok(spins, t0, err)) return false;
    }
    return true;
}

__device__ __forceinline__ bool sub_two(const u64 *g1, unsigned tag1, double *dst1, const u64 *g2, unsigned tag2, double *dst2,
                                        int count, int *err, const u64 *beat, unsigned btag)
{
    const int per = 2 * (int)blockDim.x;
    if (count <= per) return sub_two_n<1>(g1, tag1, dst1, g2, tag2, dst2, count, err, beat, btag);
    if (count <= 2 * per) return sub_two_n<2>(g1, tag1, dst1, g2, tag2, dst2, count, err, beat, btag);
    if (count <= 3 * per) return sub_two_n<3>(g1, tag1, dst1, g2, tag2, dst2, count, err, beat, btag);
    return sub_two_n<4>(g1, tag1, dst1, g2, tag2, dst2, count, err, beat, btag);      // count <= 8 blockDim
}

// ---- k_sytrd_onehop: ONE all-to-all exchange per column (experimental twin of k_sytrd_lower, same interface) -----
// At step k every workgroup holds v_k, tau_k and has published y_k = tau_k A v_k for its own columns.  Then
//   a. (while y travels) Q <- Q H_k on its rows of Q;
//   b. it takes column k+1 (state before update k), which its owner broadcast ONE STEP AHEAD, together with
//   c. the full y_k (+ heartbeats), forms y.v and w = y - (tau/2)(y.v) v itself;
//   d. it applies update k to its copy of column k+1 and forms the reflector v_{k+1}, tau_{k+1} itself — every
//      workgroup runs the same instructions on the same bits, so all copies are identical and the reflector never
//      travels; the owner also keeps it in LAPACK storage;
//   e. ONE pass over its columns j > k+1: A[:, j] -= v w_j + w v_j fused with y_{k+1,j} = tau A[:, j].v_{k+1},
//      published at once; the owner of column k+2 broadcasts that column (now in the state step k+1 needs).
// Buffers (parity of the item index), tags and the heartbeat rule as in k_sytrd_lower; here the heartbeat (k) is
// fresh: a workgroup with no column > k publishes it where the others publish y_k.
// LDS: (ncol_max * n + 5 n + 32) doubles; requires n <= 32 NR and at most 8 columns per workgroup; 512 threads (8 waves:
// wave w takes my column c0 + w in the fused pass e. and holds my w-th row of Q).
template <int NR>
__global__ __launch_bounds__(512) void k_sytrd_onehop(double *__restrict__ A, int lda, int n, double *__restrict__ D,
                                                      double *__restrict__ E, double *__restrict__ TAU,
                                                      u64 *cg, u64 *yg, u64 *hg, int *err, double *__restrict__ Qout, int ldq, int kstop)
{
    // kstop: as in k_sytrd_lower (columns 0 .. kstop-1 are reduced; kstop < n-1 leaves A[kstop:, kstop:] updated for k_sytd2_tail)
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int G = gridDim.x, g = blockIdx.x, tid = threadIdx.x, nthr = blockDim.x;
    const int ncol_max = (n + G - 1) / G;
    const int nc = (n - g + G - 1) / G > 0 ? (n - g + G - 1) / G : 0;       // my columns: g, g+G, ...
    const int Ga = n < G ? n : G;
    if (nc == 0) return;
    const int lastcol = g + (nc - 1) * G;
    double *slab = lds;
    double *vs = lds + (size_t)ncol_max * n;              // v_k      (index 0 <-> row k+1)
    double *vnext = vs + n;                               // v_{k+1}  (index 0 <-> row k+2)
    double *ys = vnext + n;                               // y_k
    double *ws = ys + n;                                  // w_k
    double *cs = ws + n;                                  // column k+1, then x = that column after update k
    double *red = cs + n;
    unsigned rs = 0;
    const int seg = tid & 31, ce = tid >> 5, cpp = nthr >> 5;
    const int ne = (n + 1) & ~1, nv = n + 2;              // buffer strides (even)
    // my rows of Q live in REGISTERS: wave `wq` holds row g + wq*G (nc <= 8 = waves per workgroup, n <= 32 NR), lane `lq` its
    // elements lq, lq+64, ... — the Q update then only reads v from LDS
    constexpr int NRW = (NR + 1) / 2;
    const int wq = tid >> 6, lq = tid & 63;
    const bool haveq = Qout && wq < nc;
    double qreg[NRW];
#pragma unroll
    for (int r = 0; r < NRW; ++r) qreg[r] = (haveq && lq + 64 * r == g + wq * G) ? 1.0 : 0.0;
    for (int c = 0; c < nc; ++c)
        for (int i = tid; i < n; i += nthr) slab[(size_t)c * n + i] = A[(size_t)(g + c * G) * lda + i];
    for (int i = tid; i < n; i += nthr) vs[i] = ws[i] = 0.0;                // "update -1" is empty
    __syncthreads();
    if (g == 0)                                           // column 0 as it is; column 1 goes out in the first fused pass
        for (int i = tid; i < n; i += nthr) pub(cg, i, slab[i], step_tag(0));

#ifdef JCDF_SYTRD_PROFILE
    u64 tprof = wall_clock64();
#endif
    double tau = 0.0;
    for (int k = -1; k < n - 1; ++k) {
        const int r0 = k + 1, m = n - r0;                 // rows r0 .. n-1, relative index i <-> row r0 + i
        // ---- a. while y_k travels: Q <- Q H_k on my row of Q (registers; v_k from LDS, index = column - r0)
        if (Qout && k >= 0) {
            double vr[NRW], sq = 0.0;
#pragma unroll
            for (int r = 0; r < NRW; ++r) {
                const int c = lq + 64 * r;
                vr[r] = (64 * r + 63 >= r0 && c >= r0 && c < n) ? vs[c - r0] : 0.0;
                sq += qreg[r] * vr[r];
            }
            const double sc = tau * wave_sum(sq);
#pragma unroll
            for (int r = 0; r < NRW; ++r) qreg[r] -= sc * vr[r];
        }
        SYTRD_TICK(4);                                    // Q accumulation
        // ---- b + c. column r0 and y_k (+ heartbeat of the workgroups without a column > k), ONE round of polls
        const u64 *cb = cg + (size_t)(r0 & 1) * nv;
        const u64 *yb = k >= 0 ? yg + (size_t)(k & 1) * ne : nullptr;
        const u64 *hb = (k >= 0 && tid < Ga && tid + ((n - 1 - tid) / G) * G <= k) ? hg + (size_t)(k & 1) * G + tid : nullptr;
        bool ok = sub_two(cb, step_tag(r0), cs, yb, step_tag(k), ys, m, err, hb, step_tag(k));
        double dot = 0.0;
        if (k >= 0 && ok)
            for (int i = 2 * tid; i < m; i += 2 * nthr) {                   // each thread re-reads only what it wrote
                dot += ys[i] * vs[i];
                if (i + 1 < m) dot += ys[i + 1] * vs[i + 1];
            }
        double bad = ok ? 0.0 : 1.0;
        block_sum2(bad, dot, red, rs);
        if (bad != 0.0) return;
        SYTRD_TICK(2);                                    // waited for the column and y
        // ---- d. w, column r0 after update k, its reflector — identical in every workgroup
        const double al = -0.5 * tau * dot;
        const double v0 = vs[0], w0 = (k >= 0 ? ys[0] : 0.0) + al * v0;
        double part = 0.0;
        for (int i = tid; i < m; i += nthr) {
            const double wi = (k >= 0 ? ys[i] : 0.0) + al * vs[i];
            ws[i] = wi;
            const double x = cs[i] - (vs[i] * w0 + wi * v0);
            cs[i] = x;
            if (i >= 2) part += x * x;
        }
        const double xnorm2 = block_sum(part, red, rs);   // barrier: ws and cs (= x) are visible
        const bool mine = g == r0 % G;
        double *own = slab + (size_t)(r0 / G) * n + r0;   // my column r0 from its diagonal (owner only)
        if (r0 == n - 1) {
            if (mine && tid == 0) {
                own[0] = cs[0];
                D[r0] = cs[0];
            }
            break;
        }
        if (r0 == kstop) {
            // early stop: column r0 is stored as it is after update k (no reflector), update k goes into my other columns
            // (no y, nothing sent), and the trailing block is left to the one-workgroup kernel
            if (mine)
                for (int i = tid; i < m; i += nthr) own[i] = cs[i];
            const int c0 = (r0 + 1 - g + G - 1) / G;
            for (int cbase = c0; cbase < nc; cbase += cpp) {
                if (cbase + ce < nc) {
                    const int j = g + (cbase + ce) * G;
                    double *col = slab + (size_t)(cbase + ce) * n + r0;
                    const double wj = ws[j - r0], vj = vs[j - r0];
                    for (int i = seg; i < m; i += 32) col[i] -= vs[i] * wj + ws[i] * vj;
                }
            }
            break;
        }
        const double alpha = cs[1];
        double tau_next, beta, scale;                     // (rsq / rcp + Newton: ~130 instead of ~270 cycles of dependent fp64 ops)
        house_scalars(alpha, xnorm2, tau_next, beta, scale);
        for (int i = tid; i < m - 1; i += nthr) {         // v_{k+1}: index i <-> row r0 + 1 + i
            const double v = (i == 0) ? 1.0 : cs[i + 1] * scale;
            vnext[i] = v;
            if (mine && i > 0) own[i + 1] = v;
        }
        if (mine && tid == 0) {
            own[0] = cs[0];
            own[1] = beta;
            D[r0] = cs[0];
            E[r0] = beta;
            TAU[r0] = tau_next;
        }
        SYTRD_TICK(0);                                    // reflector (every workgroup)
        // ---- e. fused: rank-2 update k of my columns j > r0 and y_{k+1}; column r0+1 is sent on by its owner
        {
            const int c0 = (r0 + 1 - g + G - 1) / G;
            u64 *ynb = yg + (size_t)((k + 1) & 1) * ne;
            u64 *cnb = cg + (size_t)((r0 + 1) & 1) * nv;
            const unsigned ytag = step_tag(k + 1), ctag = step_tag(r0 + 1);
            // one WAVE per column (512 threads: eight columns at once), 64 rows per pass, the column's y by one wave sum
            const int wv = tid >> 6, ln = tid & 63, nwv = nthr >> 6;
            for (int cbase = c0; cbase < nc; cbase += nwv) {
                const bool have = cbase + wv < nc;
                const int j = g + (have ? cbase + wv : c0) * G;
                double *col = slab + (size_t)(have ? cbase + wv : c0) * n + r0;
                const double wj = ws[j - r0], vj = vs[j - r0];
                double s0 = 0.0;
                // four passes' operands are fetched before the first result is stored: the compiler cannot move a load above the
                // store of the pass before (all of it is one LDS array to it), and a lane's passes would run one LDS latency each
                for (int i = ln; i < m; i += 256) {
                    double cc[4], vv[4], wwv[4], zz[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int ii = i + 64 * u;
                        const bool in = ii < m;
                        cc[u] = in ? col[ii] : 0.0;
                        vv[u] = in ? vs[ii] : 0.0;
                        wwv[u] = in ? ws[ii] : 0.0;
                        zz[u] = in ? cs[ii] : 0.0;
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int ii = i + 64 * u;
                        if (ii < m) {
                            const double x0 = cc[u] - (vv[u] * wj + wwv[u] * vj);
                            if (have) col[ii] = x0;
                            if (ii >= 1) s0 += x0 * (ii == 1 ? 1.0 : zz[u] * scale);
                        }
                    }
                }
                const double sy = wave_sum(s0);
                if (ln == 0 && have) pub(ynb, j - (r0 + 1), tau_next * sy, ytag);
            }
            // column r0+1 goes out from its owner — by ALL its waves, from the slab the pass above has just updated: published
            // inside the pass by the one wave that owns the column it was m / 64 store instructions on that wave alone, and every
            // step has one such workgroup whose lateness the whole grid then waits for in the next round of polls
            if (r0 + 1 < n && g == (r0 + 1) % G) {         // (uniform over the workgroup)
                __syncthreads();
                const double *nxt = slab + (size_t)((r0 + 1) / G) * n + r0;
                for (int i = 1 + tid; i < m; i += nthr) pub(cnb, i - 1, nxt[i], ctag);
            }
            if (tid == 0 && lastcol <= k + 1) pub(hg + (size_t)((k + 1) & 1) * G, g, 0.0, ytag);     // heartbeat (k+1)
        }
        __syncthreads();
        double *t = vs;
        vs = vnext;
        vnext = t;
        tau = tau_next;
        SYTRD_TICK(3);                                    // fused update + y
    }
    __syncthreads();
    for (int c = 0; c < nc; ++c)
        for (int i = tid; i < n; i += nthr) A[(size_t)(g + c * G) * lda + i] = slab[(size_t)c * n + i];
    if (haveq) {
#pragma unroll
        for (int r = 0; r < NRW; ++r)
            if (lq + 64 * r < n) Qout[(size_t)(g + wq * G) * ldq + lq + 64 * r] = qreg[r];
    }
}

// ---- k_sytd2_tail: the LAST columns of the tridiagonalisation in ONE workgroup -------------------------------------------------
// The chip-wide kernels above pay one hand-off per column (4.6-7 us) whatever is left of the matrix.  The trailing block
// A[k0:, k0:] of T = n - k0 <= 128 rows fits the REGISTER FILE of one workgroup: 512 threads x 32 doubles.  Wave w, lane l holds
// rows w, w + 8, w + 16, ... (16 of them) of the columns l and l + 64 (both triangles, kept symmetric): rows are dealt to the waves
// cyclically, so the part still to be reduced stays spread over all eight as it shrinks.  The persistent kernel stops at column
// k0 (kstop) and this kernel finishes the job — LAPACK dsytd2 on the block: D, E, TAU for rows k0 .., reflectors back into A.
// Per column: the column goes to LDS (as ROW j of the symmetric block: two elements per lane of one wave), its norm in every wave
// (no block reduction), the reflector's scalars in every thread, y = tau A v as register FMAs + an eight-way sum through LDS, y.v
// again in every wave, the rank-2 update with w = y - (tau y.v / 2) v folded into the column factor,
//   A -= v (y + 2 al v)_c^T + y v_c^T,
// four barriers in all; only vectors travel through LDS (a first version with the block itself in LDS ran 9800 cycles per
// column, bound by LDS instruction issue), and groups of four registers whose rows are already reduced are skipped, as is the
// first column of a lane once j has passed 63.  The reflectors are kept in LDS (T x 128 doubles) exactly as they were applied
// and go to A at the end, LAPACK storage (beta on the sub-diagonal, v below it): scaling the register copy of column j instead
// would store a v that differs from the applied one (taken from ROW j) by the rounding asymmetry of the two triangles — for a
// graded matrix enough to cost Q two digits of orthogonality (water / 6-31G(2df,p) core Hamiltonian: 4e-14 instead of 1e-15).
constexpr int SYTD2_TAIL_T = 128;
constexpr int SYTD2_TAIL_NT = 512;
__global__ __launch_bounds__(SYTD2_TAIL_NT) void k_sytd2_tail(double *__restrict__ A, int lda, int n, int k0, double *__restrict__ D,
                                                             double *__restrict__ E, double *__restrict__ TAU)
{
    constexpr int TM = SYTD2_TAIL_T, NW = SYTD2_TAIL_NT / 64, RW = TM / NW;       // 8 waves, RW = 16 rows of a column per thread
    // vectors in natural order (index = row) and dealt by wave (row i at (i % 8) * 16 + i / 8: a thread's 16 rows are contiguous)
    __shared__ __attribute__((aligned(16))) double vn[TM], vd[TM], yn[TM], yd[TM], xs[TM], yp[NW][TM];
    __shared__ double dl[TM], el[TM], tl[TM];            // D, E, TAU of the block: to global memory once, at the end (a global store
                                                         // in front of a barrier makes the workgroup wait for it)
    extern __shared__ __attribute__((aligned(16))) double vall[];              // reflector j at vall + j * TM, natural order
    const int tid = threadIdx.x, T = n - k0, lane = tid & 63, w = tid >> 6;
    const int c0 = lane, c1 = lane + 64;
    auto dealt = [](int i) { return (i & (NW - 1)) * RW + (i >> 3); };
    double a0[RW], a1[RW];
#pragma unroll
    for (int r = 0; r < RW; ++r) {
        const int row = NW * r + w;
        a0[r] = (c0 < T && row < T) ? A[(size_t)(k0 + c0) * lda + k0 + row] : 0.0;
        a1[r] = (c1 < T && row < T) ? A[(size_t)(k0 + c1) * lda + k0 + row] : 0.0;
    }
#ifdef JCDF_TAIL_TICKS
    long long tk[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = clock64(), w0 = wall_clock64();
#define TT(i) { const long long tn = clock64(); tk[i] += tn - tprev; tprev = tn; }
#else
#define TT(i)
#endif
    TT(0);
    for (int j = 0; j + 1 < T; ++j) {
        // ---- row j (= column j) -> LDS: wave j % 8 holds it, register j / 8, two elements per lane
        if (w == (j & (NW - 1))) {
            double x0, x1;
            switch (j >> 3) {
#define JCDF_ROWSEL(r) case r: x0 = a0[r]; x1 = a1[r]; break;
                JCDF_ROWSEL(0) JCDF_ROWSEL(1) JCDF_ROWSEL(2) JCDF_ROWSEL(3) JCDF_ROWSEL(4) JCDF_ROWSEL(5) JCDF_ROWSEL(6) JCDF_ROWSEL(7)
                JCDF_ROWSEL(8) JCDF_ROWSEL(9) JCDF_ROWSEL(10) JCDF_ROWSEL(11) JCDF_ROWSEL(12) JCDF_ROWSEL(13) JCDF_ROWSEL(14)
#undef JCDF_ROWSEL
            default: x0 = a0[RW - 1]; x1 = a1[RW - 1]; break;
            }
            xs[c0] = x0;
            xs[c1] = x1;
        }
        __syncthreads();
        TT(1);
        // (every wave sums the whole column itself, two elements per lane: the same bits everywhere, and no second trip through LDS)
        const int e0 = 2 * lane, e1 = 2 * lane + 1;
        const double2_t xp = *reinterpret_cast<const double2_t *>(xs + e0);
        const double xnorm2 = wave_sum(((e0 >= j + 2 && e0 < T) ? xp.x * xp.x : 0.0) + ((e1 >= j + 2 && e1 < T) ? xp.y * xp.y : 0.0));
        const double alpha = xs[j + 1], diag = xs[j];
        double tau, beta, scale;
        TT(2);
        house_scalars(alpha, xnorm2, tau, beta, scale);                        // the same scalars in every thread
        if (w == 0) {
            double2_t v;
            v.x = (e0 <= j || e0 >= T) ? 0.0 : ((e0 == j + 1) ? 1.0 : xp.x * scale);
            v.y = (e1 <= j || e1 >= T) ? 0.0 : ((e1 == j + 1) ? 1.0 : xp.y * scale);
            *reinterpret_cast<double2_t *>(vn + e0) = v;
            *reinterpret_cast<double2_t *>(vall + (size_t)j * TM + e0) = v;
            vd[dealt(e0)] = v.x;
            vd[dealt(e1)] = v.y;
            if (tid == 0) {
                dl[j] = diag;
                el[j] = beta;
                tl[j] = tau;
            }
        }
        __syncthreads();
        TT(3);
        if (tau != 0.0) {                                                      // (uniform: every thread holds the same tau)
            const bool both = j < 63;                                          // columns 0 .. 63 are behind j otherwise: v_c = w_c = 0 there
            // ---- y_c = tau sum_i A[i][c] v_i: my rows from registers, the eight waves' parts through LDS
            {
                double s0 = 0.0, s1 = 0.0;
#pragma unroll
                for (int g = 0; g < RW; g += 4) {
                    if (NW * (g + 3) + w > j) {                                // (wave-uniform) the group's last row is still to be reduced
                        const double2_t va = *reinterpret_cast<const double2_t *>(vd + w * RW + g);
                        const double2_t vb = *reinterpret_cast<const double2_t *>(vd + w * RW + g + 2);
                        s1 += a1[g] * va.x + a1[g + 1] * va.y + a1[g + 2] * vb.x + a1[g + 3] * vb.y;
                        if (both) s0 += a0[g] * va.x + a0[g + 1] * va.y + a0[g + 2] * vb.x + a0[g + 3] * vb.y;
                    }
                }
                yp[w][c0] = s0;
                yp[w][c1] = s1;
            }
            __syncthreads();
            TT(4);
            // y and y.v in every wave (two elements per lane), y to LDS from the first wave: one barrier instead of a block reduction
            double dot;
            {
                double2_t y = {0.0, 0.0};
#pragma unroll
                for (int q = 0; q < NW; ++q) {
                    const double2_t p = *reinterpret_cast<const double2_t *>(&yp[q][e0]);
                    y.x += p.x;
                    y.y += p.y;
                }
                const double2_t vp = *reinterpret_cast<const double2_t *>(vn + e0);
                y.x = (e0 > j && e0 < T) ? tau * y.x : 0.0;
                y.y = (e1 > j && e1 < T) ? tau * y.y : 0.0;
                if (w == 0) {
                    *reinterpret_cast<double2_t *>(yn + e0) = y;
                    yd[dealt(e0)] = y.x;
                    yd[dealt(e1)] = y.y;
                }
                dot = wave_sum(y.x * vp.x + y.y * vp.y);
            }
            const double al2 = -tau * dot;                                     // 2 al, al = -tau y.v / 2
            __syncthreads();
            TT(5);
            // ---- A -= v w^T + w v^T, w = y + al v: v and y vanish up to row / column j, so the reduced part (and the reflector) stay.
            //      (The LDS pipe, not its latency, is what these loops wait for: every wave fetches its own 16 v_i and y_i, 1 KB per
            //      instruction whether the lanes read the same address or not; fetching all of them ahead of the branches was slower.)
            {
                const double vc0 = vn[c0], vc1 = vn[c1], wc0 = yn[c0] + al2 * vc0, wc1 = yn[c1] + al2 * vc1;
#pragma unroll
                for (int g = 0; g < RW; g += 4) {
                    if (NW * (g + 3) + w > j) {
                        const double2_t va = *reinterpret_cast<const double2_t *>(vd + w * RW + g);
                        const double2_t vb = *reinterpret_cast<const double2_t *>(vd + w * RW + g + 2);
                        const double2_t ya = *reinterpret_cast<const double2_t *>(yd + w * RW + g);
                        const double2_t yb = *reinterpret_cast<const double2_t *>(yd + w * RW + g + 2);
                        a1[g] -= va.x * wc1 + ya.x * vc1;
                        a1[g + 1] -= va.y * wc1 + ya.y * vc1;
                        a1[g + 2] -= vb.x * wc1 + yb.x * vc1;
                        a1[g + 3] -= vb.y * wc1 + yb.y * vc1;
                        if (both) {
                            a0[g] -= va.x * wc0 + ya.x * vc0;
                            a0[g + 1] -= va.y * wc0 + ya.y * vc0;
                            a0[g + 2] -= vb.x * wc0 + yb.x * vc0;
                            a0[g + 3] -= vb.y * wc0 + yb.y * vc0;
                        }
                    }
                }
            }
        }
        TT(6);
        // (no barrier here: xs is rewritten next, and nobody reads it after the barrier that published v; v, yp and y are rewritten
        //  two barriers further on)
    }
#ifdef JCDF_TAIL_TICKS
    if (tid == 0)
        printf("tail ticks: load %lld | col->lds %lld norm %lld house+v %lld matvec %lld y,dot %lld update %lld | wall %lld (100 MHz)\n", tk[0], tk[1],
               tk[2], tk[3], tk[4], tk[5], tk[6], wall_clock64() - w0);
#endif
#undef TT
    if (w == ((T - 1) & (NW - 1))) {                                           // the last diagonal element
#pragma unroll
        for (int r = 0; r < RW; ++r)
            if (NW * r + w == T - 1) {
                if (c0 == T - 1) dl[T - 1] = a0[r];
                if (c1 == T - 1) dl[T - 1] = a1[r];
            }
    }
    __syncthreads();
    if (tid < T) {
        D[k0 + tid] = dl[tid];
        if (tid + 1 < T) {
            E[k0 + tid] = el[tid];
            TAU[k0 + tid] = tl[tid];
        }
    }
    // LAPACK storage of the lower triangle: D on the diagonal, beta on the sub-diagonal, the reflector below it (the upper triangle
    // of the block keeps what the chip-wide kernel left there: nobody reads it)
    for (int e = tid; e < T * TM; e += SYTD2_TAIL_NT) {
        const int c = e / TM, row = e % TM;
        if (row >= c && row < T)
            A[(size_t)(k0 + c) * lda + k0 + row] = (row == c) ? dl[c] : ((row == c + 1) ? el[c] : vall[(size_t)c * TM + row]);
    }
}

// Q[:, k0 : k0 + T] <- Q[:, k0 : k0 + T] H_k0 H_k0+1 ... (the reflectors k_sytd2_tail left in A, LAPACK storage) for all rows of Q
// (row-major, leading dimension ldq): every row is e_r^T Q H H ..., independent of the others, so the whole chip takes part.
// The T - 2 reflectors go to LDS once per workgroup (126 x 128 doubles); a wave holds two rows in registers, two columns per
// lane, and applies a reflector with one 16-byte LDS read, four FMAs and two wave sums.
constexpr int Q_TAIL_ROWS = 16;                           // rows per workgroup (8 waves x 2)
constexpr int Q_TAIL_NT = 512;
__global__ __launch_bounds__(Q_TAIL_NT) void k_q_tail_reflect(double *__restrict__ Q, int ldq, int nrows, int k0, int T, const double *__restrict__ A,
                                                        int lda, const double *__restrict__ TAU, int q_is_unit)
{
    constexpr int TM = SYTD2_TAIL_T;
    extern __shared__ __attribute__((aligned(16))) double vall[];          // reflector j at vall + j * TM (v[i] for block row i), then tau[TM]
    double *taus = vall + (size_t)(TM - 2) * TM;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nref = T - 2;
#pragma unroll 8
    for (int e = tid; e < nref * (TM / 2); e += Q_TAIL_NT) {
        const int j = e / (TM / 2), i = (e % (TM / 2)) * 2;
        double2_t v = {0.0, 0.0};
        if (i + 1 < T) {                                                   // (k0 + i even offsets: lda and k0 are not known to be, so two loads)
            const double *col = A + (size_t)(k0 + j) * lda + k0;
            v.x = (i <= j) ? 0.0 : ((i == j + 1) ? 1.0 : col[i]);
            v.y = (i + 1 <= j) ? 0.0 : ((i + 1 == j + 1) ? 1.0 : col[i + 1]);
        } else if (i < T) {
            v.x = (i <= j) ? 0.0 : ((i == j + 1) ? 1.0 : A[(size_t)(k0 + j) * lda + k0 + i]);
        }
        *reinterpret_cast<double2_t *>(vall + (size_t)j * TM + i) = v;
    }
    if (tid < TM) taus[tid] = (tid < nref) ? TAU[k0 + tid] : 0.0;
    __syncthreads();
    const int r0 = blockIdx.x * Q_TAIL_ROWS + wave * 2, r1 = r0 + 1, cc = 2 * lane;
    // (q_is_unit: nothing has written Q yet — the block is the whole matrix — and it starts as the unit matrix)
    auto ld = [&](int r, int col) { return (r < nrows && col < T) ? (q_is_unit ? (r == k0 + col ? 1.0 : 0.0) : Q[(size_t)r * ldq + k0 + col]) : 0.0; };
    double q00 = ld(r0, cc), q01 = ld(r0, cc + 1), q10 = ld(r1, cc), q11 = ld(r1, cc + 1);
    for (int j = 0; j < nref; ++j) {
        const double2_t v = *reinterpret_cast<const double2_t *>(vall + (size_t)j * TM + cc);
        const double tj = taus[j];
        const double s0 = tj * wave_sum(q00 * v.x + q01 * v.y), s1 = tj * wave_sum(q10 * v.x + q11 * v.y);
        q00 -= s0 * v.x;
        q01 -= s0 * v.y;
        q10 -= s1 * v.x;
        q11 -= s1 * v.y;
    }
    if (r0 < nrows) {
        if (cc < T) Q[(size_t)r0 * ldq + k0 + cc] = q00;
        if (cc + 1 < T) Q[(size_t)r0 * ldq + k0 + cc + 1] = q01;
    }
    if (r1 < nrows) {
        if (cc < T) Q[(size_t)r1 * ldq + k0 + cc] = q10;
        if (cc + 1 < T) Q[(size_t)r1 * ldq + k0 + cc + 1] = q11;
    }
}

#ifdef JCDF_DIAGNOSTIC
// ---- k_sytrd_replay_q: Q = H_0 H_1 ... H_{n-3} from the stored reflectors (LAPACK dorgtr's result, row-major) -----------------
// Optional (DeviceEigh: JCDF_EIGH_Q_REPLAY=1): Q is not needed by the tridiagonal solver, so instead of accumulating it inside the
// persistent kernel it can be rebuilt afterwards, row-parallel, on a side stream beside the divide & conquer: row r of Q
// is e_r^T H_0 H_1 ..., two rows per wave in registers (lane l holds columns l, l+64, ...), reflector k read from column k of A
// (contiguous), the next one prefetched while this one is applied.  ~0.1 ms at n = 510 on a side stream beside the divide &
// conquer (0.4 ms).  A: n x n column-major LAPACK storage (v_k below the sub-diagonal of column k), TAU[n].
template <int NR>
__global__ __launch_bounds__(256) void k_sytrd_replay_q(const double *__restrict__ A, int lda, int n, const double *__restrict__ TAU,
                                                        double *__restrict__ Q, int ldq)
{
    __shared__ double stau[640];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < n; i += 256) stau[i] = (i < n - 2) ? TAU[i] : 0.0;
    __syncthreads();
    const int row0 = (blockIdx.x * 4 + wave) * 2;
    if (row0 >= n) return;
    double q0[NR], q1[NR], v[NR], v1[NR], v2[NR];                   // reflectors k, k+1, k+2 (two L2 latencies ahead)
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int c = lane + 64 * r;
        q0[r] = (c == row0) ? 1.0 : 0.0;
        q1[r] = (c == row0 + 1) ? 1.0 : 0.0;
    }
    auto fetch = [&](int k, double (&dst)[NR]) {                    // v_k: 1 at k+1, A[k][c] for c >= k+2, 0 elsewhere
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int c = lane + 64 * r;
            dst[r] = (k < n - 2 && c >= k + 2 && c < n) ? A[(size_t)k * lda + c] : ((c == k + 1 && k < n - 2) ? 1.0 : 0.0);
        }
    };
    // apply H_k from `use` while reflector k+2 lands in `next` (three register sets in rotation: no copy ever waits for a load)
    auto step = [&](int k, const double (&use)[NR], double (&next)[NR]) {
        fetch(k + 2, next);
        const double tau = stau[k < n - 2 ? k : 0];
        double d0a = 0.0, d0b = 0.0, d1a = 0.0, d1b = 0.0;
#pragma unroll
        for (int r = 0; r < NR; r += 2) {
            d0a += q0[r] * use[r];
            d1a += q1[r] * use[r];
            if (r + 1 < NR) {
                d0b += q0[r + 1] * use[r + 1];
                d1b += q1[r + 1] * use[r + 1];
            }
        }
        const double d0 = tau * wave_sum(d0a + d0b), d1 = tau * wave_sum(d1a + d1b);
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            q0[r] -= d0 * use[r];
            q1[r] -= d1 * use[r];
        }
    };
    fetch(0, v);
    fetch(1, v1);
    for (int k = 0; k < n - 2; k += 3) {                           // reflectors past n-3 are fetched as zero: H = I
        step(k, v, v2);
        step(k + 1, v1, v);
        step(k + 2, v2, v1);
    }
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int c = lane + 64 * r;
        if (c < n) {
            Q[(size_t)row0 * ldq + c] = q0[r];
            if (row0 + 1 < n) Q[(size_t)(row0 + 1) * ldq + c] = q1[r];
        }
    }
}
#endif  // JCDF_DIAGNOSTIC


// ---- k_diis_solve: the Pulay step of the SCF wrapper without a round trip to the host ----------------------------
// (reference: DIIS in src/rhf/energy/EnergyHelpers.jl:234-258 — B matrix of error-vector dot products bordered by
// -1, LAPACK.sysv!('U'), "Faulty DIIS" -> history cut to 2; called from SCF.jl:472-501.)
// Bmat: nd x nd ring buffer of <e_i, e_j> (symmetric), dots[nd]: <e_slot, e_newest> for every slot (written into row
// and column `head` first), n: vectors in use, newest-first order slot_k = (head - k) mod nd.  One workgroup; the
// (n+1) x (n+1) system is solved by Gaussian elimination with partial pivoting in LDS (n <= 15).
// coef[slot] = c (0 for unused slots).  A singular or non-finite system sets flag[0] = 1 and returns the unit vector
// on the newest entry (no extrapolation this iteration), like the reference's exception path.
// nparts > 0: dots[s] is first formed from nparts partial sums per slot (dots[s * nparts + k], fixed order) — the
// k_diis_dots_partial output consumed directly, one launch less per iteration.
__global__ __launch_bounds__(64) void k_diis_solve(double *__restrict__ Bmat, const double *__restrict__ dots, int nd, int head,
                                                  int n, int solve, double *__restrict__ coef, int *__restrict__ flag, int nparts = 0)
{
    __shared__ double M[16][17];
    __shared__ double rhs[16];
    __shared__ int piv_row;
    const int tid = threadIdx.x;
    if (tid < nd) {
        double d = 0.0;
        if (nparts > 0) {
            // (fixed order; eight loads in flight at a time instead of one after the other)
            int k = 0;
            for (; k + 8 <= nparts; k += 8) {
                double t[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) t[u] = dots[tid * nparts + k + u];
#pragma unroll
                for (int u = 0; u < 8; ++u) d += t[u];
            }
            for (; k < nparts; ++k) d += dots[tid * nparts + k];
        } else
            d = dots[tid];
        Bmat[(size_t)head * nd + tid] = d;
        Bmat[(size_t)tid * nd + head] = d;
    }
    __syncthreads();
    if (tid < nd) coef[tid] = (tid == head) ? 1.0 : 0.0;
    if (!solve) return;
    const int m = n + 1;
    for (int idx = tid; idx < m * m; idx += 64) {
        const int i = idx / m, j = idx % m;
        double v;
        if (i < n && j < n) v = Bmat[(size_t)((head - i + nd) % nd) * nd + (head - j + nd) % nd];
        else v = (i == n && j == n) ? 0.0 : -1.0;
        M[i][j] = v;
    }
    if (tid < m) rhs[tid] = (tid == n) ? -1.0 : 0.0;
    __syncthreads();
    bool bad = false;
    for (int c = 0; c < m; ++c) {
        // partial pivoting, by the whole wave: lane r offers |M[r][c]|, the first lane holding the maximum is the pivot row
        // (the row a serial search with a strict '>' finds)
        int p;
        {
            const bool in = tid >= c && tid < m;
            const double mine = in ? fabs(M[tid][c]) : -1.0;
            double best = mine;                                                  // (fmax passes over a NaN: it fails the solve later)
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) best = fmax(best, __shfl_xor(best, off, 64));
            const unsigned long long who = __ballot(in && mine == best && best > 0.0 && isfinite(best));
            p = who ? (int)__builtin_ctzll(who) : -1;
        }
        if (p < 0) { bad = true; break; }
        if (p != c) {
            if (tid < m) { const double t = M[c][tid]; M[c][tid] = M[p][tid]; M[p][tid] = t; }
            if (tid == 0) { const double t = rhs[c]; rhs[c] = rhs[p]; rhs[p] = t; }
        }
        __syncthreads();
        const double d = M[c][c];
        if (tid > c && tid < m) {                         // row `tid`
            const double f = M[tid][c] / d;
            for (int j = c; j < m; ++j) M[tid][j] -= f * M[c][j];
            rhs[tid] -= f * rhs[c];
        }
        __syncthreads();
    }
    if (!bad) {                                           // back substitution, column-oriented: every lane forms x_r, lane j < r updates rhs[j]
        for (int r = m - 1; r >= 0; --r) {
            const double x = rhs[r] / M[r][r];
            if (!isfinite(x)) bad = true;                 // (the same value in every lane)
            __syncthreads();
            if (tid == r) rhs[r] = x;
            if (tid < r) rhs[tid] -= M[tid][r] * x;
            __syncthreads();
        }
        if (tid == 0) piv_row = bad ? -1 : 0;
    }
    __syncthreads();
    if (bad || piv_row < 0) {
        if (tid == 0) flag[0] = 1;
        return;                                           // coef already = unit vector on the newest entry
    }
    if (tid < nd) coef[tid] = 0.0;
    __syncthreads();
    if (tid < n) coef[(head - tid + nd) % nd] = rhs[tid];
}

}  // namespace jcdf
