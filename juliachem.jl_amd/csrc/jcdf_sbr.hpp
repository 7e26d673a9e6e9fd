// jcdf_sbr.hpp — two-stage tridiagonalisation of a symmetric fp64 matrix (caller side of the hot path, SURVEY 8
// row f1: the replicated eigensolve of `iteration`, /root/reference/src/rhf/energy/SCF.jl:1080-1083).
//
// Why two stages: the one-stage Householder reduction (jcdf_eig.hpp) needs one chip-wide exchange per COLUMN — n
// dependent all-to-all hand-offs at 4.6-7 us each, 2.4 ms at n = 510, the largest single item of an SCF iteration.
//   stage 1  dense -> band of half-width SB = 16 (successive band reduction): per PANEL of 16 columns one Householder
//            QR inside ONE workgroup (its column steps synchronise through LDS, ~0.4 us each) and one two-sided
//            block-reflector update A22 <- (I - V T V^T)^T A22 (I - V T V^T) = A22 - V W^T - W V^T on MFMA;
//            the chip-wide dependencies drop from n to 3 n/16 kernel boundaries;
//   stage 2  band -> tridiagonal by bulge chasing, one column per sweep, in ONE workgroup: the band (n x 2 SB
//            doubles) lives in LDS, every wave owns a sweep and follows the sweep in front of it at the classical
//            distance of two blocks; the waves synchronise through progress counters in LDS (~0.1 us, no
//            chip-wide traffic at all);
//   Q        the orthogonal factor is accumulated forwards, row-wise: Q1 = prod (I - V T V^T) inside the stage-1
//            update launch, then the stage-2 reflectors are replayed from a log on the rows of Q1 by a third kernel
//            that can run beside the tridiagonal eigensolver (it needs only D and E).
// tools/sbr_proto.py is the numpy statement of the same algorithm with the same index conventions.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "jcdf_eig.hpp"      // dpp reductions, double4_t comes from jcdf_gemm.hpp via the including file

namespace jcdf {

constexpr int SB = 16;             // half bandwidth after stage 1 = panel width = reflector length of stage 2
constexpr int SBW = 2 * SB;        // doubles per column of the band storage AB[j][d] = A[j+d][j], d < 2 SB (bulges included)

// ---------------------------------------------------------------------------------------------------------------
// stage 1
// ---------------------------------------------------------------------------------------------------------------

// Householder QR of the panel P = A[r0:n, j0:j0+16] (m x 16, r0 = j0 + 16) inside one workgroup of 256 threads.
// A is symmetric and fully stored, so P[i][c] is read as A[j0+c][r0+i] (contiguous in i).  Thread t holds rows
// t, t+256, ... (NROW of them) in registers.  Per column ONE workgroup reduction of 16 values: h[c'] = sum_{i>c}
// x_i P[i][c'] with x = P[c+1:, c] gives the norm (c' = c), the products v^T P[:, c'] (c' > c) and the entries
// V[:, c']^T v of the T recurrence (c' < c) at once, because v = (1, x * scale).
// Out: Vbuf[m][16] (unit lower trapezoidal), Tbuf[16][16] (upper triangular, Q = I - V T V^T), and R / zeros into
// A[j0+c][r0+i] (the upper-triangle image of the panel; the band is read from there by k_sbr_extract).
template <int NROW>
__global__ __launch_bounds__(256) void k_sbr_panel(double *__restrict__ A, int lda, int n, int k, double *__restrict__ Vbuf,
                                                   double *__restrict__ Tbuf)
{
    __shared__ double part[16][256];
    __shared__ double tot[16];
    __shared__ double prow[2][16];
    const int tid = threadIdx.x;
    const int j0 = k * SB, r0 = j0 + SB, m = n - r0;
    double p[NROW][16];
#pragma unroll
    for (int q = 0; q < NROW; ++q) {
        const int i = tid + 256 * q;
#pragma unroll
        for (int c = 0; c < 16; ++c) p[q][c] = (i < m) ? A[(size_t)(j0 + c) * lda + r0 + i] : 0.0;
    }
    double trow[16];                                      // row (tid & 15) of T, redundantly in every 16-lane group
#pragma unroll
    for (int c = 0; c < 16; ++c) trow[c] = 0.0;
    const int tr = tid & 15;

#pragma unroll
    for (int c = 0; c < 16; ++c) {
        // ---- h[c'] = sum_{i > c} x_i P[i][c'], all 16 columns
        double h[16];
#pragma unroll
        for (int cc = 0; cc < 16; ++cc) h[cc] = 0.0;
#pragma unroll
        for (int q = 0; q < NROW; ++q) {
            const int i = tid + 256 * q;
            const double x = (i > c) ? p[q][c] : 0.0;
#pragma unroll
            for (int cc = 0; cc < 16; ++cc) h[cc] += x * p[q][cc];
        }
#pragma unroll
        for (int cc = 0; cc < 16; ++cc) part[cc][tid] = h[cc];
        if (tid == c) {
#pragma unroll
            for (int cc = 0; cc < 16; ++cc) prow[c & 1][cc] = p[0][cc];
        }
        __syncthreads();
        {
            const int cq = tid >> 4, ch = tid & 15;
            double s = 0.0;
#pragma unroll
            for (int e = 0; e < 16; ++e) s += part[cq][e * 16 + ch];
            s = row16_sum(s);
            if (ch == 0) tot[cq] = s;
        }
        __syncthreads();
        double hs[16], pr[16];
#pragma unroll
        for (int cc = 0; cc < 16; ++cc) {
            hs[cc] = tot[cc];
            pr[cc] = prow[c & 1][cc];
        }
        // ---- dlarfg on (alpha, x)
        const double alpha = pr[c], sigma = hs[c];
        double tau = 0.0, beta = alpha, scale = 0.0;
        if (sigma != 0.0) {
            beta = -copysign(sqrt(alpha * alpha + sigma), alpha);
            tau = (beta - alpha) / beta;
            scale = 1.0 / (alpha - beta);
        }
        // ---- P[:, c'] -= v w[c'], w[c'] = tau (P[c][c'] + scale h[c'])  (c' > c);  column c <- (beta, v)
#pragma unroll
        for (int q = 0; q < NROW; ++q) {
            const int i = tid + 256 * q;
            const double vi = (i > c) ? p[q][c] * scale : ((i == c) ? 1.0 : 0.0);
#pragma unroll
            for (int cc = 0; cc < 16; ++cc)
                if (cc > c) p[q][cc] -= vi * (tau * (pr[cc] + scale * hs[cc]));
            if (i > c) p[q][c] = vi;
            else if (i == c) p[q][c] = beta;
        }
        // ---- column c of T: T[:c, c] = -tau T[:c, :c] S[:c, c],  S[c'][c] = V[:, c']^T v_c = P[c][c'] + scale h[c']
        {
            double acc = 0.0;
#pragma unroll
            for (int qq = 0; qq < 16; ++qq)
                if (qq < c) acc += ((qq >= tr) ? trow[qq] : 0.0) * (pr[qq] + scale * hs[qq]);
            trow[c] = (tr < c) ? -tau * acc : ((tr == c) ? tau : 0.0);
        }
    }
    // ---- out
#pragma unroll
    for (int q = 0; q < NROW; ++q) {
        const int i = tid + 256 * q;
        if (i < m) {
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                Vbuf[(size_t)i * 16 + c] = (i > c) ? p[q][c] : ((i == c) ? 1.0 : 0.0);
                A[(size_t)(j0 + c) * lda + r0 + i] = (i <= c) ? p[q][c] : 0.0;
            }
        }
    }
    if (tid < 16) {
#pragma unroll
        for (int c = 0; c < 16; ++c) Tbuf[tid * 16 + c] = trow[c];
    }
}

// Y = A22 V (m x 16), A22 = A[r0:, r0:] symmetric and fully stored; one workgroup per 16 rows, its four waves split
// the contraction; MFMA operands straight from global memory (A22[kk][i] for A22[i][kk]: 128-byte segments; V rows
// are 512 contiguous bytes per k step).  Also the tile's part of M1 = V^T Y (16 x 16) into M1p[tile].
__global__ __launch_bounds__(256) void k_sbr_y(const double *__restrict__ A, int lda, int n, int r0, const double *__restrict__ Vbuf,
                                               double *__restrict__ Ybuf, double *__restrict__ M1p)
{
    __shared__ double red[4][4][64];
    __shared__ double ytile[16][17];
    const int m = n - r0, i0 = blockIdx.x * 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 15, lk = lane >> 4;
    const int nks = (m + 3) / 4;
    double4_t acc = {0.0, 0.0, 0.0, 0.0};
    const bool rowok = i0 + lr < m;
    for (int ks = wave; ks < nks; ks += 4) {
        const int kk = 4 * ks + lk;
        const bool ok = kk < m;
        const double a = (ok && rowok) ? A[(size_t)(r0 + kk) * lda + r0 + i0 + lr] : 0.0;
        const double b = ok ? Vbuf[(size_t)kk * 16 + lr] : 0.0;
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave][r][lane] = acc[r];
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const double y = (red[0][r][lane] + red[1][r][lane]) + (red[2][r][lane] + red[3][r][lane]);
            const int row = lk + 4 * r;                              // accumulator layout: row = lk + 4 r, col = lr
            ytile[row][lr] = y;
            if (i0 + row < m) Ybuf[(size_t)(i0 + row) * 16 + lr] = y;
        }
        // M1 part: sum_i V[i0+i][c'] Y[i0+i][c]: A operand [row = c'][k = i], B operand [k = i][col = c]
        double4_t ma = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s_ = 0; s_ < 4; ++s_) {
            const int i = 4 * s_ + lk;
            const double a = (i0 + i < m) ? Vbuf[(size_t)(i0 + i) * 16 + lr] : 0.0;
            const double b = ytile[i][lr];
            ma = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, ma, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) M1p[(size_t)blockIdx.x * 256 + (lk + 4 * r) * 16 + lr] = ma[r];
    }
}

// W = (Y - 1/2 V (T^T M1)) T, one workgroup per 16 rows (M1 = sum of the tile parts, fixed order).
__global__ __launch_bounds__(256) void k_sbr_w(int m, const double *__restrict__ Vbuf, const double *__restrict__ Ybuf,
                                               const double *__restrict__ Tbuf, const double *__restrict__ M1p, int ntile,
                                               double *__restrict__ Wbuf)
{
    __shared__ double sT[16][17], sM[16][17], sN[16][17], sX[16][17];
    const int tid = threadIdx.x, r = tid >> 4, c = tid & 15, i0 = blockIdx.x * 16;
    double m1 = 0.0;
    for (int t = 0; t < ntile; ++t) m1 += M1p[(size_t)t * 256 + tid];
    sM[r][c] = m1;
    sT[r][c] = Tbuf[tid];
    __syncthreads();
    double nn = 0.0;
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) nn += sT[kk][r] * sM[kk][c];        // (T^T M1)[r][c]
    sN[r][c] = nn;
    __syncthreads();
    const bool ok = i0 + r < m;
    double x = ok ? Ybuf[(size_t)(i0 + r) * 16 + c] : 0.0;
    if (ok) {
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) x -= 0.5 * Vbuf[(size_t)(i0 + r) * 16 + kk] * sN[kk][c];
    }
    sX[r][c] = x;
    __syncthreads();
    double w = 0.0;
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) w += sX[r][kk] * sT[kk][c];
    if (ok) Wbuf[(size_t)(i0 + r) * 16 + c] = w;
}

// blocks [0, nt2): A22 <- A22 - V W^T - W V^T on 32 x 32 tiles (4 waves x one 16 x 16 MFMA tile, K = 2 x 16);
// blocks [nt2, nt2 + ceil(n/16)): 16 rows of Q:  Q[I, r0:] <- Q[I, r0:] - ((Q[I, r0:] V) T) V^T.
__global__ __launch_bounds__(256) void k_sbr_update(double *__restrict__ A, int lda, int n, int r0, const double *__restrict__ Vbuf,
                                                    const double *__restrict__ Wbuf, const double *__restrict__ Tbuf,
                                                    double *__restrict__ Q, int ldq, int nt1)
{
    const int m = n - r0;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 15, lk = lane >> 4;
    const int nt2 = nt1 * nt1;
    if ((int)blockIdx.x < nt2) {
        const int I = (blockIdx.x / nt1) * 32 + (wave >> 1) * 16, J = (blockIdx.x % nt1) * 32 + (wave & 1) * 16;
        if (I >= m || J >= m) return;
        double4_t acc;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = I + lk + 4 * r, col = J + lr;
            acc[r] = (row < m && col < m) ? A[(size_t)(r0 + row) * lda + r0 + col] : 0.0;
        }
        const bool iok = I + lr < m, jok = J + lr < m;
#pragma unroll
        for (int s_ = 0; s_ < 4; ++s_) {
            const int kk = 4 * s_ + lk;
            const double vi = iok ? Vbuf[(size_t)(I + lr) * 16 + kk] : 0.0, wi = iok ? Wbuf[(size_t)(I + lr) * 16 + kk] : 0.0;
            const double vj = jok ? Vbuf[(size_t)(J + lr) * 16 + kk] : 0.0, wj = jok ? Wbuf[(size_t)(J + lr) * 16 + kk] : 0.0;
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-vi, wj, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-wi, vj, acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = I + lk + 4 * r, col = J + lr;
            if (row < m && col < m) A[(size_t)(r0 + row) * lda + r0 + col] = acc[r];
        }
        return;
    }
    // ---- rows of Q
    __shared__ double red[4][4][64];
    __shared__ double zt[16][17];
    const int I = ((int)blockIdx.x - nt2) * 16;
    const bool rowok = I + lr < n;
    {
        // Z = Q[I, r0:] V: A operand [row = i][k = j] = Q[I+i][r0+j], B operand [k = j][col = c] = V[j][c]
        double4_t acc = {0.0, 0.0, 0.0, 0.0};
        const int nks = (m + 3) / 4;
        for (int ks = wave; ks < nks; ks += 4) {
            const int kk = 4 * ks + lk;
            const bool ok = kk < m;
            const double a = (ok && rowok) ? Q[(size_t)(I + lr) * ldq + r0 + kk] : 0.0;
            const double b = ok ? Vbuf[(size_t)kk * 16 + lr] : 0.0;
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wave][r][lane] = acc[r];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            zt[lk + 4 * r][lr] = (red[0][r][lane] + red[1][r][lane]) + (red[2][r][lane] + red[3][r][lane]);
    }
    __syncthreads();
    // ZT = Z T (every wave for itself): A operand [row = i][k] = Z[i][k], B operand [k][col = c] = T[k][c]
    double4_t ztacc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s_ = 0; s_ < 4; ++s_) {
        const int kk = 4 * s_ + lk;
        ztacc = __builtin_amdgcn_mfma_f64_16x16x4f64(zt[lr][kk], Tbuf[kk * 16 + lr], ztacc, 0, 0, 0);
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) zt[lk + 4 * r][lr] = ztacc[r];
    }
    __syncthreads();
    // Q[I, r0 + J] -= ZT V[J]^T: A operand [row = i][k = c] = ZT[i][c], B operand [k = c][col = j] = V[J+j][c]
    double za[4];
#pragma unroll
    for (int s_ = 0; s_ < 4; ++s_) za[s_] = -zt[lr][4 * s_ + lk];
    const int ntj = (m + 15) / 16;
    for (int tj = wave; tj < ntj; tj += 4) {
        const int J = tj * 16;
        double4_t acc;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = I + lk + 4 * r, col = J + lr;
            acc[r] = (row < n && col < m) ? Q[(size_t)row * ldq + r0 + col] : 0.0;
        }
        const bool jok = J + lr < m;
#pragma unroll
        for (int s_ = 0; s_ < 4; ++s_) {
            const double b = jok ? Vbuf[(size_t)(J + lr) * 16 + 4 * s_ + lk] : 0.0;
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(za[s_], b, acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = I + lk + 4 * r, col = J + lr;
            if (row < n && col < m) Q[(size_t)row * ldq + r0 + col] = acc[r];
        }
    }
}

// AB[j][d] = A[j][j+d] for d <= SB (the upper-triangle image holds the panels' R factors), 0 for SB < d < 2 SB;
// Q = identity when `Q` is given (before stage 1).
__global__ __launch_bounds__(256) void k_sbr_extract(const double *__restrict__ A, int lda, int n, double *__restrict__ AB)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= n * SBW) return;
    const int j = idx / SBW, d = idx % SBW;
    AB[idx] = (d <= SB && j + d < n) ? A[(size_t)j * lda + j + d] : 0.0;
}

__global__ __launch_bounds__(256) void k_set_identity(double *__restrict__ Q, int ldq, int n)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= n * n) return;
    const int r = idx / n, c = idx % n;
    Q[(size_t)r * ldq + c] = (r == c) ? 1.0 : 0.0;
}

// ---------------------------------------------------------------------------------------------------------------
// stage 2
// ---------------------------------------------------------------------------------------------------------------

__device__ __forceinline__ double mk_f64(unsigned lo, unsigned hi) { return __longlong_as_double(((long long)hi << 32) | lo); }

// x + x(lane ^ 16): v_permlane16_swap exchanges the odd rows of its first operand with the even rows of its second
__device__ __forceinline__ double xor16_sum(double x)
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(x);
    const auto l = __builtin_amdgcn_permlane16_swap((unsigned)b, (unsigned)b, false, false);
    const auto h = __builtin_amdgcn_permlane16_swap((unsigned)(b >> 32), (unsigned)(b >> 32), false, false);
    return mk_f64(l[0], h[0]) + mk_f64(l[1], h[1]);       // (even row's value) + (odd row's value) in both rows
}

// x + x(lane ^ 32)
__device__ __forceinline__ double xor32_sum(double x)
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(x);
    const auto l = __builtin_amdgcn_permlane32_swap((unsigned)b, (unsigned)b, false, false);
    const auto h = __builtin_amdgcn_permlane32_swap((unsigned)(b >> 32), (unsigned)(b >> 32), false, false);
    return mk_f64(l[0], h[0]) + mk_f64(l[1], h[1]);       // (lower half's value) + (upper half's value) in both halves
}

// sum over the four 16-lane rows of the wave, the same bits in every row
__device__ __forceinline__ double rows_sum(double x) { return xor32_sum(xor16_sum(x)); }

__device__ __forceinline__ double lane0_f64(double x)
{
    const long long b = __double_as_longlong(x);
    return mk_f64((unsigned)__builtin_amdgcn_readfirstlane((int)(b & 0xffffffffLL)), (unsigned)__builtin_amdgcn_readfirstlane((int)(b >> 32)));
}

constexpr int SB2ST_DONE = 1 << 30;

// Band -> tridiagonal in one workgroup of NW waves.  LDS: band (n + 16) x 32 doubles (16 zero rows behind the matrix:
// blocks that reach past the end need no masks), NW x 48 doubles of transposition scratch, n progress counters.
// Lane map inside a 16 x 16 block: lane = (j = lane & 15 column, g = lane >> 4), rows i = 4 g + r, r < 4 ("layout C");
// sums over rows are 4 FMAs + two row exchanges, sums over columns use a second register image of the block with the
// roles of rows and columns swapped ("layout R": lane = (row lane & 15, columns 4 g + r)) — no 16-lane DPP reduction
// of four values anywhere.  Vectors change between "indexed by lane & 15" and "indexed by 4 g + r" through 16 doubles of
// per-wave LDS scratch.
// Sweep s (column s), step t: reflector H_t on rows R_t = s+1+16t .. +15:
//   D_t <- H_t D_t H_t (symmetric 16 x 16 at R_t), B_t <- B_t H_t (block below it), H_{t+1} from B_t[:, 0], B_t <- H_{t+1} B_t.
// Step (s, t) may start when step (s-1, t+1) is complete: prog[s-1] >= t+2.
// log[(s * tmax + t) * 16 + j] = sqrt(tau) v[j]  (H = I - (sqrt(tau) v)(sqrt(tau) v)^T), for k_sb2st_apply_q.
template <int NW>
__global__ __launch_bounds__(NW * 64) void k_sb2st_chase(const double *__restrict__ ABin, int n, double *__restrict__ D,
                                                         double *__restrict__ E, double *__restrict__ vlog, int tmax, int *err)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double *band = lds;                                             // (n + 16) * 32
    double *scr = band + (size_t)(n + 16) * SBW + (threadIdx.x >> 6) * 48;  // per wave: w | z | x
    int *prog = (int *)(band + (size_t)(n + 16) * SBW + NW * 48);   // n ints (+1 abort word)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, g = lane >> 4;
    for (int i = tid; i < (n + 16) * SBW; i += NW * 64) band[i] = (i < n * SBW) ? ABin[i] : 0.0;
    for (int i = tid; i <= n; i += NW * 64) prog[i] = 0;
    __syncthreads();
    int *abortw = prog + n;

    int offD[4], offC[4], offR[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = 4 * g + r;
        offD[r] = (i >= j) ? j * SBW + (i - j) : i * SBW + (j - i);
        offC[r] = j * SBW + SB + i - j;                             // B[i][j], layout C
        const int c = 4 * g + r;                                    // layout R: row = j (lane & 15), column c
        offR[r] = c * SBW + SB + j - c;
    }
    const int off0 = SB + j;                                        // B[row = lane & 15][0]

    auto wait_for = [&](int s_, int need) -> bool {
        if (s_ < 0) return true;
        unsigned spins = 0;
        while (__hip_atomic_load(prog + s_, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < need) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 22) || __hip_atomic_load(abortw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0) {
                __hip_atomic_store(abortw, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                return false;
            }
        }
        return true;
    };
    // dlarfg from x given in both index forms; returns v in both forms (v[0] = 1), tau, beta
    auto house = [&](double xj, const double (&xr)[4], double &vj, double (&vr)[4], double &tau, double &beta) {
        const double alpha = lane0_f64(xj);
        double sp = 0.0;
#pragma unroll
        for (int r = 0; r < 4; ++r) sp += (4 * g + r >= 1) ? xr[r] * xr[r] : 0.0;
        const double sigma = rows_sum(sp);
        double scale = 0.0;
        tau = 0.0;
        beta = alpha;
        if (sigma != 0.0) {
            beta = -copysign(sqrt(alpha * alpha + sigma), alpha);
            tau = (beta - alpha) / beta;
            scale = 1.0 / (alpha - beta);
        }
        vj = (j == 0) ? 1.0 : xj * scale;
#pragma unroll
        for (int r = 0; r < 4; ++r) vr[r] = (4 * g + r == 0) ? 1.0 : xr[r] * scale;
    };

    bool alive = true;
    for (int s = wave; s < n - 2 && alive; s += NW) {
        const int nst = (n - s - 3) / SB + 1;
        if (!wait_for(s - 1, 2)) break;
        // ---- reflector that clears column s below the sub-diagonal
        double vj, vr[4], tau, beta;
        {
            double *col = band + (size_t)s * SBW + 1;
            const double xj = col[j];
            double xr[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) xr[r] = col[4 * g + r];
            house(xj, xr, vj, vr, tau, beta);
            if (g == 0) col[j] = (j == 0) ? beta : 0.0;
        }
        int r0 = s + 1;
        for (int t = 0; t < nst; ++t, r0 += SB) {
            if (t > 0 && !wait_for(s - 1, t + 2)) { alive = false; break; }
            double *base = band + (size_t)r0 * SBW;
            if (g == 0) vlog[((size_t)s * tmax + t) * 16 + j] = sqrt(tau) * vj;
            double d[4], bc[4], br[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                d[r] = base[offD[r]];
                bc[r] = base[offC[r]];
                br[r] = base[offR[r]];
            }
            const double b0 = base[off0];
            // u[j] = sum_i D[i][j] v[i] (D symmetric), z[row] = sum_c B[row][c] v[c]
            double up = 0.0, zp = 0.0;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                up += d[r] * vr[r];
                zp += br[r] * vr[r];
            }
            const double u = rows_sum(up), z = rows_sum(zp);
            const double gamma = row16_sum(u * vj);
            const double w = tau * u - 0.5 * tau * tau * gamma * vj;
            const double x = b0 - tau * z;                          // first column of B H (v[0] = 1)
            if (g == 0) {
                scr[j] = w;
                scr[16 + j] = z;
                scr[32 + j] = x;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            double wr[4], zr[4], xr[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                wr[r] = scr[4 * g + r];
                zr[r] = scr[16 + 4 * g + r];
                xr[r] = scr[32 + 4 * g + r];
            }
            // D <- D - v w^T - w v^T (lower triangle goes back)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                d[r] -= vr[r] * w + wr[r] * vj;
                if (4 * g + r >= j) base[offD[r]] = d[r];
            }
            // B <- B - tau z v^T
#pragma unroll
            for (int r = 0; r < 4; ++r) bc[r] -= tau * zr[r] * vj;
            // next reflector from the first column x of B
            double v2j, v2r[4], tau2, beta2;
            house(x, xr, v2j, v2r, tau2, beta2);
            double gp = 0.0;
#pragma unroll
            for (int r = 0; r < 4; ++r) gp += v2r[r] * bc[r];
            const double gg = rows_sum(gp);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                bc[r] -= tau2 * v2r[r] * gg;
                if (j == 0) bc[r] = (4 * g + r == 0) ? beta2 : 0.0;
                base[offC[r]] = bc[r];
            }
            __hip_atomic_store(prog + s, t + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            vj = v2j;
            tau = tau2;
#pragma unroll
            for (int r = 0; r < 4; ++r) vr[r] = v2r[r];
        }
        __hip_atomic_store(prog + s, SB2ST_DONE, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __syncthreads();
    if (tid == 0 && *abortw != 0) *err = 2;
    for (int i = tid; i < n; i += NW * 64) {
        D[i] = band[(size_t)i * SBW];
        if (i < n - 1) E[i] = band[(size_t)i * SBW + 1];
    }
}

// Q <- Q H_(0,0) H_(0,1) ... H_(1,0) ... (the order of the sequential algorithm) on 4 RPL rows per wave, rows in LDS.
// Lane = (column j of the 16-wide window, row group); consecutive steps of one sweep touch disjoint windows.
template <int RPL>
__global__ __launch_bounds__(256) void k_sb2st_apply_q(double *__restrict__ Q, int ldq, int n, const double *__restrict__ vlog, int tmax)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int ldr = n + 17;                                        // 16 zero columns behind every row
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, g = lane >> 4;
    const int rows_wg = 16 * RPL, row0 = blockIdx.x * rows_wg;
    for (int idx = tid; idx < rows_wg * ldr; idx += 256) {
        const int r = idx / ldr, c = idx % ldr;
        lds[idx] = (row0 + r < n && c < n) ? Q[(size_t)(row0 + r) * ldq + c] : 0.0;
    }
    __syncthreads();
    double *myrow[RPL];
#pragma unroll
    for (int e = 0; e < RPL; ++e) myrow[e] = lds + (size_t)(wave * 4 * RPL + 4 * e + g) * ldr + j;
    for (int s = 0; s < n - 2; ++s) {
        const int nst = (n - s - 3) / SB + 1;
        const double *vl = vlog + (size_t)s * tmax * 16 + j;
        int c0 = s + 1;
        int t = 0;
        for (; t + 2 <= nst; t += 2, c0 += 2 * SB) {
            const double va = vl[(size_t)t * 16], vb = vl[(size_t)(t + 1) * 16];
            double qa[RPL], qb[RPL];
#pragma unroll
            for (int e = 0; e < RPL; ++e) {
                qa[e] = myrow[e][c0];
                qb[e] = myrow[e][c0 + SB];
            }
#pragma unroll
            for (int e = 0; e < RPL; ++e) {
                const double da = row16_sum(qa[e] * va), db = row16_sum(qb[e] * vb);
                myrow[e][c0] = qa[e] - da * va;
                myrow[e][c0 + SB] = qb[e] - db * vb;
            }
        }
        if (t < nst) {
            const double va = vl[(size_t)t * 16];
#pragma unroll
            for (int e = 0; e < RPL; ++e) {
                const double qa = myrow[e][c0];
                const double da = row16_sum(qa * va);
                myrow[e][c0] = qa - da * va;
            }
        }
    }
    __syncthreads();
    for (int idx = tid; idx < rows_wg * n; idx += 256) {
        const int r = idx / n, c = idx % n;
        if (row0 + r < n) Q[(size_t)(row0 + r) * ldq + c] = lds[(size_t)r * ldr + c];
    }
}

}  // namespace jcdf
