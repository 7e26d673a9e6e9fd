// jcdf_sbr.hpp — two-stage tridiagonalisation of a symmetric fp64 matrix (caller side of the hot path, SURVEY 8
// row f1: the replicated eigensolve of `iteration`, /root/reference/src/rhf/energy/SCF.jl:1080-1083).
//
// Why two stages: the one-stage Householder reduction (jcdf_eig.hpp) needs one chip-wide exchange per COLUMN — n
// dependent all-to-all hand-offs at 4.6-7 us each, 2.4 ms at n = 510, the largest single item of an SCF iteration.
//   stage 1  dense -> band of half-width SB = 16 (successive band reduction): per PANEL of 16 columns one Householder
//            QR inside ONE workgroup (its column steps synchronise through LDS; measured 1.2 us each) and one two-sided
//            block-reflector update A22 <- (I - V T V^T)^T A22 (I - V T V^T) = A22 - V W^T - W V^T on MFMA;
//            the chip-wide dependencies drop from n to 3 n/16 kernel boundaries;
//   stage 2  band -> tridiagonal by bulge chasing, one column per sweep, in ONE workgroup: the band (n x 2 SB
//            doubles) lives in LDS, every wave owns a sweep and follows the sweep in front of it at the classical
//            distance of two blocks; the waves synchronise through progress counters in LDS (no chip-wide traffic
//            at all); measured: bound by the instruction issue of that one CU, 1.0 ms at n = 510;
//   Q        the orthogonal factor is accumulated forwards, row-wise: Q1 = prod (I - V T V^T) by one kernel per panel on
//            a side stream beside the chase, then the stage-2 reflectors are replayed from a log on the rows of Q1 by a
//            kernel that runs beside the tridiagonal eigensolver (it needs only D and E).
// Result (profiles/r02_two_stage_eigh.txt): on par with the one-stage kernel at n <= 590 (2.72-2.79 vs 2.74 ms per
// eigensolve at n = 510), not faster — optional (DeviceEigh: JCDF_EIGH_TWO_STAGE=1), the one-stage kernel is the default.
// tools/sbr_proto.py is the numpy statement of the same algorithm with the same index conventions.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "jcdf_eig.hpp"      // dpp reductions, double4_t comes from jcdf_gemm.hpp via the including file

namespace jcdf {

constexpr int SB = 16;             // half bandwidth after stage 1 = panel width = reflector length of stage 2
constexpr int SBW = 2 * SB;        // doubles per column of the band storage AB[j][d] = A[j+d][j], d < 2 SB (bulges included)

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


// a <- a + b where afterwards the lower half-wave holds (a summed over both halves) and the upper half-wave (b summed
// over both halves): TWO cross-half reductions for one addition (v_permlane32_swap exchanges a's upper with b's lower half)
__device__ __forceinline__ double pair_sum32(double a, double b)
{
    const unsigned long long ba = (unsigned long long)__double_as_longlong(a), bb = (unsigned long long)__double_as_longlong(b);
    const auto l = __builtin_amdgcn_permlane32_swap((unsigned)ba, (unsigned)bb, false, false);
    const auto h = __builtin_amdgcn_permlane32_swap((unsigned)(ba >> 32), (unsigned)(bb >> 32), false, false);
    return mk_f64(l[0], h[0]) + mk_f64(l[1], h[1]);
}
// the same across the two 16-lane rows of each half: even rows end with a's sum, odd rows with b's
__device__ __forceinline__ double pair_sum16(double a, double b)
{
    const unsigned long long ba = (unsigned long long)__double_as_longlong(a), bb = (unsigned long long)__double_as_longlong(b);
    const auto l = __builtin_amdgcn_permlane16_swap((unsigned)ba, (unsigned)bb, false, false);
    const auto h = __builtin_amdgcn_permlane16_swap((unsigned)(ba >> 32), (unsigned)(bb >> 32), false, false);
    return mk_f64(l[0], h[0]) + mk_f64(l[1], h[1]);
}

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
template <int NT>
struct SbrPanelLds {
    double part[16][NT / 4];                              // [value][wave * 16 + lane & 15]: partial sums over 4 lanes each
    double tot[16];
    double prow[2][16];
    double sS[16][17];                                    // S[c'][c] = V[:, c']^T v_c (c' < c) and tau_c on the diagonal, for T
};

// QR of the panel held in registers (p[q][c] = P[tid + NT q][c], rows >= m are 0) by the NT threads of the workgroup
// (NT = 512: two waves per SIMD — a lone wave issues one instruction per 4 cycles, two share the SIMD at 2)
template <int NROW, int NT>
__device__ __forceinline__ void sbr_panel_core(double (&p)[NROW][16], SbrPanelLds<NT> &L, double *__restrict__ A, int lda, int j0, int r0,
                                               int m, double *__restrict__ Vbuf, double *__restrict__ Tbuf)
{
    double (&part)[16][NT / 4] = L.part;
    double (&tot)[16] = L.tot;
    double (&prow)[2][16] = L.prow;
    double (&sS)[16][17] = L.sS;
    const int tid = threadIdx.x;
    const int tr = tid & 15;
    const int gq = (tid >> 4) & 3;                        // 16-lane row of the wave
    const int myval = 2 * (gq & 1) + (gq >> 1);           // after the two pair sums this row holds values 4 q + myval
    double *pdst = &part[myval][(tid >> 6) * 16 + tr];

#pragma unroll
    for (int c = 0; c < 16; ++c) {
        // ---- h[c'] = sum_{i > c} x_i P[i][c'], all 16 columns
        double h[16];
#pragma unroll
        for (int cc = 0; cc < 16; ++cc) h[cc] = 0.0;
#pragma unroll
        for (int q = 0; q < NROW; ++q) {
            const int i = tid + NT * q;
            const double x = (i > c) ? p[q][c] : 0.0;
#pragma unroll
            for (int cc = 0; cc < 16; ++cc) h[cc] += x * p[q][cc];
        }
        // sums over the four 16-lane rows of the wave, two values per exchange: 16 -> 8 -> 4 values per lane
        double s1[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) s1[e] = pair_sum32(h[2 * e], h[2 * e + 1]);
#pragma unroll
        for (int e = 0; e < 4; ++e) pdst[(size_t)(4 * e) * (NT / 4)] = pair_sum16(s1[2 * e], s1[2 * e + 1]);
        if (tid == c) {
#pragma unroll
            for (int cc = 0; cc < 16; ++cc) prow[c & 1][cc] = p[0][cc];
        }
        __syncthreads();
        {
            const int cq = tid >> 4, ch = tid & 15;
            double sv = 0.0;
            if (NT == 256 || cq < 16) {
#pragma unroll
                for (int e = 0; e < NT / 64; ++e) sv += part[cq & 15][ch + 16 * e];
            }
            sv = row16_sum(sv);
            if (ch == 0 && (NT == 256 || cq < 16)) tot[cq & 15] = sv;
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
        double tau, beta, scale;
        house_scalars(alpha, sigma, tau, beta, scale);
        // ---- P[:, c'] -= v w[c'], w[c'] = tau (P[c][c'] + scale h[c'])  (c' > c);  column c <- (beta, v)
#pragma unroll
        for (int q = 0; q < NROW; ++q) {
            const int i = tid + NT * q;
            const double vi = (i > c) ? p[q][c] * scale : ((i == c) ? 1.0 : 0.0);
#pragma unroll
            for (int cc = 0; cc < 16; ++cc)
                if (cc > c) p[q][cc] -= vi * (tau * (pr[cc] + scale * hs[cc]));
            if (i > c) p[q][c] = vi;
            else if (i == c) p[q][c] = beta;
        }
        // ---- column c of S (for T, after the loop): S[c'][c] = V[:, c']^T v_c = P[c][c'] + scale h[c']
        if (tid < 16) sS[tid][c] = (tid < c) ? pr[tid] + scale * hs[tid] : ((tid == c) ? tau : 0.0);
    }
    // ---- T: T[:c, c] = -tau_c T[:c, :c] S[:c, c]; thread r < 16 owns row r
    __syncthreads();
    double trow[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        double acc = 0.0;
#pragma unroll
        for (int qq = 0; qq < 16; ++qq)
            if (qq < c) acc += ((qq >= tr) ? trow[qq] : 0.0) * sS[qq][c];
        const double tau = sS[c][c];
        trow[c] = (tr < c) ? -tau * acc : ((tr == c) ? tau : 0.0);
    }
    // ---- out
#pragma unroll
    for (int q = 0; q < NROW; ++q) {
        const int i = tid + NT * q;
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

template <int NROW, int NT>
__global__ __launch_bounds__(NT) void k_sbr_panel(double *__restrict__ A, int lda, int n, int k, double *__restrict__ Vbuf,
                                                  double *__restrict__ Tbuf)
{
    __shared__ SbrPanelLds<NT> L;
    const int tid = threadIdx.x;
    const int j0 = k * SB, r0 = j0 + SB, m = n - r0;
    double p[NROW][16];
#pragma unroll
    for (int q = 0; q < NROW; ++q) {
        const int i = tid + NT * q;
#pragma unroll
        for (int c = 0; c < 16; ++c) p[q][c] = (i < m) ? A[(size_t)(j0 + c) * lda + r0 + i] : 0.0;
    }
    sbr_panel_core<NROW, NT>(p, L, A, lda, j0, r0, m, Vbuf, Tbuf);
}

// acc += sum over the k steps [ks0, ks1) of A-operand x B-operand, eight k steps of loads in flight at a time (the
// operands come straight from L2: one load round trip per MFMA would otherwise bound these loops)
template <class FA, class FB>
__device__ __forceinline__ void mfma_stream(double4_t &acc, int ks0, int ks1, FA fa, FB fb)
{
    for (int ks = ks0; ks < ks1; ks += 8) {
        double a[8], b[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const bool in = ks + e < ks1;
            a[e] = in ? fa(ks + e) : 0.0;
            b[e] = in ? fb(ks + e) : 0.0;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[e], b[e], acc, 0, 0, 0);
    }
}

// Y = A22 V (m x 16), A22 = A[r0:, r0:] symmetric and fully stored; one workgroup per 16 rows, its eight waves split
// the contraction; MFMA operands straight from global memory (A22[kk][i] for A22[i][kk]: 128-byte segments; V rows
// are 512 contiguous bytes per k step).  Also the tile's part of M1 = V^T Y (16 x 16) into M1p[tile].
constexpr int SBR_YW = 8;
__global__ __launch_bounds__(SBR_YW * 64) void k_sbr_y(const double *__restrict__ A, int lda, int n, int r0,
                                                      const double *__restrict__ Vbuf, double *__restrict__ Ybuf,
                                                      double *__restrict__ M1p)
{
    __shared__ double red[SBR_YW][4][64];
    __shared__ double ytile[16][17];
    const int m = n - r0, i0 = blockIdx.x * 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 15, lk = lane >> 4;
    const int nks = (m + 3) / 4, per = (nks + SBR_YW - 1) / SBR_YW;
    double4_t acc = {0.0, 0.0, 0.0, 0.0};
    const bool rowok = i0 + lr < m;
    const double *Ab = A + (size_t)r0 * lda + r0 + i0 + lr;
    mfma_stream(acc, wave * per, min(nks, (wave + 1) * per),
                [&](int ks) { const int kk = 4 * ks + lk; return (kk < m && rowok) ? Ab[(size_t)kk * lda] : 0.0; },
                [&](int ks) { const int kk = 4 * ks + lk; return kk < m ? Vbuf[(size_t)kk * 16 + lr] : 0.0; });
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave][r][lane] = acc[r];
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            double y = 0.0;
#pragma unroll
            for (int w = 0; w < SBR_YW; ++w) y += red[w][r][lane];
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

// blocks [0, nt2): A22 <- A22 - V W^T - W V^T on 32 x 32 tiles, W = Y T - V N2, N2 = 1/2 (T^T M1) T (M1 = sum of the
// tile parts of k_sbr_y, fixed order): every workgroup forms N2 and the 2 x 32 rows of W it needs itself (one more
// launch per panel would cost more), then 4 waves x one 16 x 16 MFMA tile with K = 2 x 16;
// NROWN > 0 (optional, JCDF_SBR_FUSE): one more block runs the QR of the NEXT panel (columns r0 .. r0+15, rows r0+16 ..) inside this launch — the
// one-workgroup QR is the longest item of a panel step and needs only its own 16 columns updated, which it does itself:
//   P[i][c] = A[r0+c][r0+16+i] - sum_q V[16+i][q] G1[q][c] + Y[16+i][q] G2[q][c],
//   G1 = W_top^T - N2 V_top^T,  G2 = T V_top^T  (V_top, W_top: the first 16 rows)
// while the tile blocks leave the two strips (rows < 16) != (cols < 16) of A22 alone (the upper one receives R, the lower one
// is dead).  The next panel's V and T go to the other buffer pair.
template <int NROWN>
__global__ __launch_bounds__(256) void k_sbr_update(double *__restrict__ A, int lda, int n, int r0, const double *__restrict__ Vbuf,
                                                    const double *__restrict__ Ybuf, const double *__restrict__ Tbuf,
                                                    const double *__restrict__ M1p, int ntile, int nt1, int ntb,
                                                    double *__restrict__ Vnext, double *__restrict__ Tnext)
{
    // block roles: [panel block (NROWN > 0)] [ntb tile blocks, each takes the tiles bid, bid + ntb, ...]
    const int m = n - r0;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 15, lk = lane >> 4;
    const int nt2 = nt1 * nt1;
    __shared__ double sT[16][17], sM[16][17], sN[16][17];
    __shared__ double sW[4][16][17];
    const bool is_panel = NROWN > 0 && blockIdx.x == 0;
    const int bid = (int)blockIdx.x - (NROWN > 0 ? 1 : 0);
    if (bid < ntb || is_panel) {
        {
            const int r = tid >> 4, c = tid & 15;
            double m1 = 0.0;
            for (int t = 0; t < ntile; t += 40) {                               // fixed order; all loads of a pass in flight (one L2 round trip)
                double e[40];
#pragma unroll
                for (int u = 0; u < 40; ++u) e[u] = (t + u < ntile) ? M1p[(size_t)(t + u) * 256 + tid] : 0.0;
#pragma unroll
                for (int u = 0; u < 40; ++u) m1 += e[u];
            }
            sM[r][c] = m1;
            sT[r][c] = Tbuf[tid];
            __syncthreads();
            double nn = 0.0;
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) nn += sT[kk][r] * sM[kk][c];        // (T^T M1)[r][c]
            sN[r][c] = nn;
            __syncthreads();
            double n2 = 0.0;
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) n2 += sN[r][kk] * sT[kk][c];
            sM[r][c] = -0.5 * n2;                                               // -N2
            __syncthreads();
        }
        if constexpr (NROWN > 0) {
            if (is_panel) {
                __shared__ SbrPanelLds<256> L;
                __shared__ __attribute__((aligned(16))) double sG1[16][16], sG2[16][16];
                const int r = tid >> 4, c = tid & 15;
                // W_top[r][c] and V_top[r][c] (rows r < 16 of A22; m >= 18 here)
                double wt = 0.0;
#pragma unroll
                for (int e = 0; e < 16; ++e) wt += Ybuf[r * 16 + e] * sT[e][c] + Vbuf[r * 16 + e] * sM[e][c];
                sW[0][r][c] = wt;
                sW[1][r][c] = Vbuf[r * 16 + c];
                __syncthreads();
                // G1[q][c] = W_top[c][q] + sum_e (-N2)[q][e] V_top[c][e],  G2[q][c] = sum_e T[q][e] V_top[c][e]   (q = r here)
                double g1 = sW[0][c][r], g2 = 0.0;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    g1 += sM[r][e] * sW[1][c][e];
                    g2 += sT[r][e] * sW[1][c][e];
                }
                sG1[r][c] = g1;
                sG2[r][c] = g2;
                __syncthreads();
                const int mp = m - 16;                                          // rows of the next panel
                double p[NROWN][16];
                double vr_[NROWN][16], yr_[NROWN][16];
#pragma unroll
                for (int q = 0; q < NROWN; ++q) {
                    const int i = tid + 256 * q;
                    const bool in = i < mp;
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        vr_[q][e] = in ? Vbuf[(size_t)(16 + i) * 16 + e] : 0.0;
                        yr_[q][e] = in ? Ybuf[(size_t)(16 + i) * 16 + e] : 0.0;
                        p[q][e] = in ? A[(size_t)(r0 + e) * lda + r0 + 16 + i] : 0.0;
                    }
                }
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    double g1r[16], g2r[16];
#pragma unroll
                    for (int cc = 0; cc < 16; cc += 2) {
                        const double2_t a = *reinterpret_cast<const double2_t *>(&sG1[e][cc]);
                        const double2_t b = *reinterpret_cast<const double2_t *>(&sG2[e][cc]);
                        g1r[cc] = a[0]; g1r[cc + 1] = a[1];
                        g2r[cc] = b[0]; g2r[cc + 1] = b[1];
                    }
#pragma unroll
                    for (int q = 0; q < NROWN; ++q)
#pragma unroll
                        for (int cc = 0; cc < 16; ++cc) p[q][cc] -= vr_[q][e] * g1r[cc] + yr_[q][e] * g2r[cc];
                }
                sbr_panel_core<NROWN, 256>(p, L, A, lda, r0, r0 + 16, mp, Vnext, Tnext);
                return;
            }
        }
        for (int tile = bid; tile < nt2; tile += ntb) {
            const int Ib = (tile / nt1) * 32, Jb = (tile % nt1) * 32;
            // wave w forms the 16 rows of W starting at R_w = (Ib, Ib+16, Jb, Jb+16)[w]: [Y | V] (16 x 32) x [T ; -N2] (32 x 16)
            {
                const int R = (wave < 2 ? Ib : Jb) + (wave & 1) * 16;
                const bool ok = R + lr < m;
                double4_t wacc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int s_ = 0; s_ < 4; ++s_) {
                    const int kk = 4 * s_ + lk;
                    const double y = ok ? Ybuf[(size_t)(R + lr) * 16 + kk] : 0.0, v = ok ? Vbuf[(size_t)(R + lr) * 16 + kk] : 0.0;
                    wacc = __builtin_amdgcn_mfma_f64_16x16x4f64(y, sT[kk][lr], wacc, 0, 0, 0);
                    wacc = __builtin_amdgcn_mfma_f64_16x16x4f64(v, sM[kk][lr], wacc, 0, 0, 0);
                }
                __syncthreads();                                                // the previous tile's readers of sW are done
#pragma unroll
                for (int r = 0; r < 4; ++r) sW[wave][lk + 4 * r][lr] = wacc[r];
            }
            __syncthreads();
            const int wi = wave >> 1, wj = wave & 1;                            // tile (I, J) = (Ib + 16 wi, Jb + 16 wj)
            const int I = Ib + wi * 16, J = Jb + wj * 16;
            if (I < m && J < m) {
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
                    const double vi = iok ? Vbuf[(size_t)(I + lr) * 16 + kk] : 0.0, wi_ = sW[wi][lr][kk];
                    const double vj = jok ? Vbuf[(size_t)(J + lr) * 16 + kk] : 0.0, wj_ = sW[2 + wj][lr][kk];
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-vi, wj_, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-wi_, vj, acc, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = I + lk + 4 * r, col = J + lr;
                    if (row < m && col < m && (NROWN == 0 || (row < 16) == (col < 16))) A[(size_t)(r0 + row) * lda + r0 + col] = acc[r];
                }
            }
        }
        return;
    }
}

// 16 rows of Q per block:  Q[I, r0:] <- Q[I, r0:] - ((Q[I, r0:] V) T) V^T.  Off the critical path of stage 1 (nothing there
// reads Q): launched on a side stream behind the panel's QR, it runs beside the next panels' kernels.
__global__ __launch_bounds__(256) void k_sbr_qupdate(int n, int r0, const double *__restrict__ Vbuf, const double *__restrict__ Tbuf,
                                                     double *__restrict__ Q, int ldq)
{
    const int m = n - r0;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 15, lk = lane >> 4;
    __shared__ double zt[16][17];
    __shared__ double red[4][4][64];
    const int I = (int)blockIdx.x * 16;
    const bool rowok = I + lr < n;
    {
        // Z = Q[I, r0:] V: A operand [row = i][k = j] = Q[I+i][r0+j], B operand [k = j][col = c] = V[j][c]
        double4_t acc = {0.0, 0.0, 0.0, 0.0};
        const int nks = (m + 3) / 4, per = (nks + 3) / 4;
        const double *Qb = Q + (size_t)(I + lr) * ldq + r0;
        mfma_stream(acc, wave * per, min(nks, (wave + 1) * per),
                    [&](int ks) { const int kk = 4 * ks + lk; return (kk < m && rowok) ? Qb[kk] : 0.0; },
                    [&](int ks) { const int kk = 4 * ks + lk; return kk < m ? Vbuf[(size_t)kk * 16 + lr] : 0.0; });
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
    for (int tj0 = wave; tj0 < ntj; tj0 += 16) {                  // four column tiles of this wave at a time: their loads overlap
        double4_t acc[4];
        double b[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int J = (tj0 + 4 * u) * 16;
            const bool jok = J + lr < m;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = I + lk + 4 * r, col = J + lr;
                acc[u][r] = (row < n && col < m) ? Q[(size_t)row * ldq + r0 + col] : 0.0;
                b[u][r] = jok ? Vbuf[(size_t)(J + lr) * 16 + 4 * r + lk] : 0.0;
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int J = (tj0 + 4 * u) * 16;
#pragma unroll
            for (int s_ = 0; s_ < 4; ++s_) acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(za[s_], b[u][s_], acc[u], 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = I + lk + 4 * r, col = J + lr;
                if (row < n && col < m) Q[(size_t)row * ldq + r0 + col] = acc[u][r];
            }
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

constexpr int SB2ST_DONE = 1 << 30;

#ifdef JCDF_SB2ST_PROFILE   // diagnostic build (tools/sb2st_prof.py): shader cycles per wave {waiting, in steps, steps, prologues, whole kernel}
__device__ unsigned long long g_sb2st_prof[16][8];
#define SB2ST_T(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#define SB2ST_ADD(slot, val) do { if (lane == 0) g_sb2st_prof[wave][slot] += (val); } while (0)
#else
#define SB2ST_T(var) do { } while (0)
#define SB2ST_ADD(slot, val) do { } while (0)
#endif

// Band -> tridiagonal in one workgroup of NW waves.  LDS: band (n + 16) x 32 doubles (16 zero rows behind the matrix:
// blocks that reach past the end need no masks), NW x 48 doubles of transposition scratch, n progress counters.
// Lane map inside a 16 x 16 block: lane = (j = lane & 15 column, g = lane >> 4), rows i = 4 g + r, r < 4 ("layout C");
// sums over rows are 4 FMAs + two row exchanges, sums over columns use a second register image of the block with the
// roles of rows and columns swapped ("layout R": lane = (row lane & 15, columns 4 g + r)) — no 16-lane DPP reduction
// of four values anywhere.  Vectors change between "indexed by lane & 15" and "indexed by 4 g + r" through 16 doubles of
// per-wave LDS scratch.  The kernel is bound by instruction issue (16 waves on the 4 SIMDs of one CU), so the step is
// written for instruction count.
// Sweep s (column s), step t: reflector H_t on rows R_t = s+1+16t .. +15:
//   D_t <- H_t D_t H_t (symmetric 16 x 16 at R_t), B_t <- B_t H_t (block below it), H_{t+1} from B_t[:, 0], B_t <- H_{t+1} B_t.
// Step (s, t) may start when step (s-1, t+1) is complete: prog[s-1] >= t+2.
// vlog[(s * tmax + t) * 16 + j] = v[j] for j >= 1, tau in slot 0 (v[0] = 1), for k_sb2st_apply_q.
template <int NW>
__global__ __launch_bounds__(NW * 64) void k_sb2st_chase(const double *__restrict__ ABin, int n, double *__restrict__ D,
                                                         double *__restrict__ E, double *__restrict__ vlog, int tmax, int *err)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double *band = lds;                                             // (n + 16) * 32
    double *scr = band + (size_t)(n + 16) * SBW + (threadIdx.x >> 6) * 48;  // per wave: w | tau z | x
    int *prog = (int *)(band + (size_t)(n + 16) * SBW + NW * 48);   // n ints (+1 abort word)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, g = lane >> 4;
    for (int i = tid; i < (n + 16) * SBW; i += NW * 64) band[i] = (i < n * SBW) ? ABin[i] : 0.0;
    for (int i = tid; i <= n; i += NW * 64) prog[i] = 0;
    __syncthreads();
    int *abortw = prog + n;

    int offD[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = 4 * g + r;
        offD[r] = (i >= j) ? j * SBW + (i - j) : i * SBW + (j - i);
    }
    const int offC = j * SBW + SB + 4 * g - j;                      // B[4 g + r][j] at offC + r (layout C)
    const int offR = 4 * g * SBW + SB + j - 4 * g;                  // B[j][4 g + r] at offR + r (SBW - 1) (layout R)
    const int off0 = SB + j;                                        // B[row = lane & 15][0]
    const bool g0 = g == 0, j0 = j == 0;

    // wave-uniform wait on the sweep in front; false = gave up
    auto wait_for = [&](int s_, int need) -> bool {
        if (s_ < 0) return true;
        for (unsigned spins = 0;; ++spins) {
            const int p = __builtin_amdgcn_readfirstlane(__hip_atomic_load(prog + s_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
            if (p >= need) break;
            __builtin_amdgcn_s_sleep(1);
            if ((spins & 1023u) == 1023u) {
                const int ab = __builtin_amdgcn_readfirstlane(__hip_atomic_load(abortw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
                if (ab != 0 || spins > (1u << 24)) {
                    __hip_atomic_store(abortw, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    return false;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        return true;
    };
    // reflector from x given in both index forms (xj = x[lane & 15], xr[r] = x[4 g + r])
    auto house = [&](double xj, const double (&xr)[4], double &vj, double (&vr)[4], double &tau, double &beta) {
        const double alpha = lane0_f64(xj);
        const double x0 = g0 ? 0.0 : xr[0];
        const double sigma = rows_sum(x0 * x0 + xr[1] * xr[1] + xr[2] * xr[2] + xr[3] * xr[3]);
        double scale;
        house_scalars(alpha, sigma, tau, beta, scale);
        vj = j0 ? 1.0 : xj * scale;
        vr[0] = g0 ? 1.0 : xr[0] * scale;
#pragma unroll
        for (int r = 1; r < 4; ++r) vr[r] = xr[r] * scale;
    };

    bool alive = true;
    SB2ST_T(tk0);
    for (int s = wave; s < n - 2 && alive; s += NW) {
        const int nst = (n - s - 3) / SB + 1;
        SB2ST_T(tw0);
        if (!wait_for(s - 1, 2)) break;
        SB2ST_T(tw1);
        SB2ST_ADD(0, tw1 - tw0);
        // ---- reflector that clears column s below the sub-diagonal
        double vj, vr[4], tau, beta;
        {
            double *col = band + (size_t)s * SBW + 1;
            const double xj = col[j];
            double xr[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) xr[r] = col[4 * g + r];
            house(xj, xr, vj, vr, tau, beta);
            if (g0) col[j] = j0 ? beta : 0.0;
        }
        SB2ST_T(tp1);
        SB2ST_ADD(3, tp1 - tw1);
        double *base = band + (size_t)(s + 1) * SBW;
        double *lg = vlog + (size_t)s * tmax * 16 + j;
        for (int t = 0; t < nst; ++t, base += SB * SBW, lg += 16) {
            SB2ST_T(ts0);
            if (t > 0 && !wait_for(s - 1, t + 2)) { alive = false; break; }
            SB2ST_T(ts1);
            SB2ST_ADD(0, ts1 - ts0);
            double d[4], bc[4], br[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                d[r] = base[offD[r]];
                bc[r] = base[offC + r];
                br[r] = base[offR + r * (SBW - 1)];
            }
            const double b0 = base[off0];
            if (g0) *lg = j0 ? tau : vj;
            // u[j] = sum_i D[i][j] v[i] (D symmetric), z[row] = sum_c B[row][c] v[c]
            const double u = rows_sum(d[0] * vr[0] + d[1] * vr[1] + d[2] * vr[2] + d[3] * vr[3]);
            const double z = rows_sum(br[0] * vr[0] + br[1] * vr[1] + br[2] * vr[2] + br[3] * vr[3]);
            const double gamma = row16_sum(u * vj);
            const double w = tau * (u - (0.5 * tau * gamma) * vj);
            const double tz = tau * z;
            const double x = b0 - tz;                               // first column of B H (v[0] = 1)
            if (g0) {
                scr[j] = w;
                scr[16 + j] = tz;
                scr[32 + j] = x;
            }
            double wr[4], zr[4], xr[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                wr[r] = scr[4 * g + r];
                zr[r] = scr[16 + 4 * g + r];
                xr[r] = scr[32 + 4 * g + r];
            }
            // D <- D - v w^T - w v^T: (v_i w_j) + (w_i v_j) is the same number in the lanes of (i, j) and (j, i), both store
#pragma unroll
            for (int r = 0; r < 4; ++r) base[offD[r]] = d[r] - __dadd_rn(__dmul_rn(vr[r], w), __dmul_rn(wr[r], vj));   // (no fma: it would break the symmetry)
            // B <- B - tau z v^T
#pragma unroll
            for (int r = 0; r < 4; ++r) bc[r] -= zr[r] * vj;
            // next reflector from the first column x of B
            double v2j, v2r[4], tau2, beta2;
            house(x, xr, v2j, v2r, tau2, beta2);
            const double tg = tau2 * rows_sum(v2r[0] * bc[0] + v2r[1] * bc[1] + v2r[2] * bc[2] + v2r[3] * bc[3]);
#pragma unroll
            for (int r = 0; r < 4; ++r) bc[r] -= v2r[r] * tg;
            if (j0) {
                bc[0] = g0 ? beta2 : 0.0;
                bc[1] = bc[2] = bc[3] = 0.0;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) base[offC + r] = bc[r];
            __hip_atomic_store(prog + s, t + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            SB2ST_T(ts2);
            SB2ST_ADD(1, ts2 - ts1);
            SB2ST_ADD(2, 1ULL);
            vj = v2j;
            tau = tau2;
#pragma unroll
            for (int r = 0; r < 4; ++r) vr[r] = v2r[r];
        }
        __hip_atomic_store(prog + s, SB2ST_DONE, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    SB2ST_T(tk1);
    SB2ST_ADD(4, tk1 - tk0);
    __syncthreads();
    if (tid == 0 && *abortw != 0) *err = 2;
    for (int i = tid; i < n; i += NW * 64) {
        D[i] = band[(size_t)i * SBW];
        if (i < n - 1) E[i] = band[(size_t)i * SBW + 1];
    }
}

// ---- the same chase with TWO waves per sweep ---------------------------------------------------------------------------
// The time of k_sb2st_chase is the dependency chain "step (s, t+1) complete -> step (s+1, t) may start", 2 n steps long,
// each step one wave's ~250 dependent-ish instructions (~2000 cycles; the waves wait two thirds of the time).  Only the
// reflector chain of a sweep is sequential inside the sweep:  v_t -> z = B_t v_t -> x = B_t[:, 0] - tau z -> v_{t+1};
// the updates of D_t and B_t are needed by the NEXT sweep, not by the next step.  So a sweep is run by a pair of waves:
//   C (chain):  waits for the sweep in front, reads B_t once (layout R), forms z, x and the next reflector, and passes
//               {v, tau z, x, tau, tau', scale', beta'} to its partner through a small ring in LDS — it never waits for
//               its partner's updates (only for ring space);
//   U (update): D_t <- H D_t H,  B_t <- H' (B_t H), stores the blocks and publishes the progress counter of the sweep.
// A sweep now trails the one in front by 2 max(C, U) instead of 2 (C + U).
constexpr int SB2ST_SLOT = 72;       // doubles per ring slot: v[16] | tau z[16] | x[16] | w[16] (U's own) | tau, tau', scale', beta', ...
template <int NP, int RING>
__global__ __launch_bounds__(NP * 128) void k_sb2st_chase2(const double *__restrict__ ABin, int n, double *__restrict__ D,
                                                           double *__restrict__ E, double *__restrict__ vlog, int tmax, int *err)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double *band = lds;                                             // (n + 16) * 32
    double *ringbase = band + (size_t)(n + 16) * SBW;               // NP * RING * SB2ST_SLOT
    int *prog = (int *)(ringbase + NP * RING * SB2ST_SLOT);         // n progress counters, abort word, 2 NP sequence counters
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int pair = wave >> 1;
    const bool is_c = (wave & 1) == 0;
    const int j = lane & 15, g = lane >> 4;
    for (int i = tid; i < (n + 16) * SBW; i += NP * 128) band[i] = (i < n * SBW) ? ABin[i] : 0.0;
    for (int i = tid; i < n + 1 + 2 * NP; i += NP * 128) prog[i] = 0;
    __syncthreads();
    int *abortw = prog + n, *cseq = prog + n + 1 + pair, *useq = prog + n + 1 + NP + pair;
    double *ring = ringbase + (size_t)pair * RING * SB2ST_SLOT;
    const bool g0 = g == 0, j0 = j == 0;

    // wave-uniform wait until *word >= need; false = gave up
    auto wait_ge = [&](const int *word, int need) -> bool {
        for (unsigned spins = 0;; ++spins) {
            const int p = __builtin_amdgcn_readfirstlane(__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
            if (p >= need) break;
            __builtin_amdgcn_s_sleep(1);
            if ((spins & 1023u) == 1023u) {
                const int ab = __builtin_amdgcn_readfirstlane(__hip_atomic_load(abortw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
                if (ab != 0 || spins > (1u << 24)) {
                    __hip_atomic_store(abortw, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    return false;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        return true;
    };

    int seq = 0;                                                    // steps of this pair so far (both waves count alike)
    SB2ST_T(tk0);
    if (is_c) {
        const int offR = 4 * g * SBW + SB + j - 4 * g;              // B[j][4 g + r] at offR + r (SBW - 1) (layout R)
        const int off0 = SB + j;                                    // B[row = lane & 15][0]
        bool alive = true;
        for (int s = pair; s < n - 2 && alive; s += NP) {
            const int nst = (n - s - 3) / SB + 1;
            if (s > 0 && !wait_ge(prog + s - 1, 2)) break;
            // ---- reflector that clears column s below the sub-diagonal
            double vj, vr[4], tau;
            {
                double *col = band + (size_t)s * SBW + 1;
                const double xj = col[j];
                double xr[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) xr[r] = col[4 * g + r];
                const double alpha = lane0_f64(xj);
                const double x0 = g0 ? 0.0 : xr[0];
                const double sigma = rows_sum(x0 * x0 + xr[1] * xr[1] + xr[2] * xr[2] + xr[3] * xr[3]);
                double beta, scale;
                house_scalars(alpha, sigma, tau, beta, scale);
                vj = j0 ? 1.0 : xj * scale;
                vr[0] = g0 ? 1.0 : xr[0] * scale;
#pragma unroll
                for (int r = 1; r < 4; ++r) vr[r] = xr[r] * scale;
                if (g0) col[j] = j0 ? beta : 0.0;
            }
            const double *base = band + (size_t)(s + 1) * SBW;
            double *lg = vlog + (size_t)s * tmax * 16 + j;
            for (int t = 0; t < nst; ++t, base += SB * SBW, lg += 16, ++seq) {
                SB2ST_T(tc0);
                if (t > 0 && s > 0 && !wait_ge(prog + s - 1, t + 2)) { alive = false; break; }
                SB2ST_T(tc1);
                SB2ST_ADD(0, tc1 - tc0);
                double br[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) br[r] = base[offR + r * (SBW - 1)];
                const double b0 = base[off0];
                if (g0) *lg = j0 ? tau : vj;
                const double tz = tau * rows_sum(br[0] * vr[0] + br[1] * vr[1] + br[2] * vr[2] + br[3] * vr[3]);
                const double x = b0 - tz;                           // first column of B H (v[0] = 1), indexed by lane & 15
                SB2ST_T(tc2);
                if (seq >= RING && !wait_ge(useq, seq - RING + 1)) { alive = false; break; }
                SB2ST_T(tc3);
                SB2ST_ADD(5, tc3 - tc2);
                double *slot = ring + (size_t)(seq % RING) * SB2ST_SLOT;
                if (g0) {
                    slot[j] = vj;
                    slot[16 + j] = tz;
                    slot[32 + j] = x;
                }
                double xr[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) xr[r] = slot[32 + 4 * g + r];
                const double alpha = lane0_f64(x);
                const double sigma = row16_sum(j0 ? 0.0 : x * x);
                double tau2, beta2, scale2;
                house_scalars(alpha, sigma, tau2, beta2, scale2);
                if (lane == 0) {
                    slot[64] = tau;
                    slot[65] = tau2;
                    slot[66] = scale2;
                    slot[67] = beta2;
                }
                __hip_atomic_store(cseq, seq + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                SB2ST_T(tc4);
                SB2ST_ADD(1, (tc4 - tc1) - (tc3 - tc2));
                SB2ST_ADD(2, 1ULL);
                vj = j0 ? 1.0 : x * scale2;
                vr[0] = g0 ? 1.0 : xr[0] * scale2;
#pragma unroll
                for (int r = 1; r < 4; ++r) vr[r] = xr[r] * scale2;
                tau = tau2;
            }
        }
    } else {
        int offD[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = 4 * g + r;
            offD[r] = (i >= j) ? j * SBW + (i - j) : i * SBW + (j - i);
        }
        const int offC = j * SBW + SB + 4 * g - j;                  // B[4 g + r][j] at offC + r (layout C)
        bool alive = true;
        for (int s = pair; s < n - 2 && alive; s += NP) {
            const int nst = (n - s - 3) / SB + 1;
            double *base = band + (size_t)(s + 1) * SBW;
            for (int t = 0; t < nst; ++t, base += SB * SBW, ++seq) {
                SB2ST_T(tu0);
                if (!wait_ge(cseq, seq + 1)) { alive = false; break; }
                SB2ST_T(tu1);
                SB2ST_ADD(0, tu1 - tu0);
                double *slot = ring + (size_t)(seq % RING) * SB2ST_SLOT;
                const double vj = slot[j], tzj = slot[16 + j];
                double vr[4], tzr[4], xr[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    vr[r] = slot[4 * g + r];
                    tzr[r] = slot[16 + 4 * g + r];
                    xr[r] = slot[32 + 4 * g + r];
                }
                const double tau = slot[64], tau2 = slot[65], scale2 = slot[66], beta2 = slot[67];
                double d[4], bc[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    d[r] = base[offD[r]];
                    bc[r] = base[offC + r];
                }
                // D <- H D H: u = D v (D symmetric: column sums), gamma = v^T u, w = tau u - (tau^2 gamma / 2) v
                const double u = rows_sum(d[0] * vr[0] + d[1] * vr[1] + d[2] * vr[2] + d[3] * vr[3]);
                const double gamma = row16_sum(u * vj);
                const double w = tau * (u - (0.5 * tau * gamma) * vj);
                if (g0) slot[48 + j] = w;
                // B <- B - (tau z) v^T, then H' from the left: B <- B - v' (tau' v'^T B), first column (beta', 0, ...)
#pragma unroll
                for (int r = 0; r < 4; ++r) bc[r] -= tzr[r] * vj;
                double v2r[4];
                v2r[0] = g0 ? 1.0 : xr[0] * scale2;
#pragma unroll
                for (int r = 1; r < 4; ++r) v2r[r] = xr[r] * scale2;
                const double tg = tau2 * rows_sum(v2r[0] * bc[0] + v2r[1] * bc[1] + v2r[2] * bc[2] + v2r[3] * bc[3]);
#pragma unroll
                for (int r = 0; r < 4; ++r) bc[r] -= v2r[r] * tg;
                if (j0) {
                    bc[0] = g0 ? beta2 : 0.0;
                    bc[1] = bc[2] = bc[3] = 0.0;
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) base[offC + r] = bc[r];
                double wr[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) wr[r] = slot[48 + 4 * g + r];
                // (v_i w_j) + (w_i v_j) is the same number in the lanes of (i, j) and (j, i): both store (no fma: symmetry)
#pragma unroll
                for (int r = 0; r < 4; ++r) base[offD[r]] = d[r] - __dadd_rn(__dmul_rn(vr[r], w), __dmul_rn(wr[r], vj));
                __hip_atomic_store(prog + s, t + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_store(useq, seq + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                SB2ST_T(tu2);
                SB2ST_ADD(1, tu2 - tu1);
                SB2ST_ADD(2, 1ULL);
                (void)tzj;
            }
            if (alive) __hip_atomic_store(prog + s, SB2ST_DONE, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    SB2ST_T(tk1);
    SB2ST_ADD(4, tk1 - tk0);
    __syncthreads();
    if (tid == 0 && *abortw != 0) *err = 2;
    for (int i = tid; i < n; i += NP * 128) {
        D[i] = band[(size_t)i * SBW];
        if (i < n - 1) E[i] = band[(size_t)i * SBW + 1];
    }
}

// ---- two waves per sweep, taking its steps in turn ("ping-pong") ---------------------------------------------------------
// Step t+1 of a sweep needs from step t only the next reflector, which is known a third of the way into the step (after
// z = B v, x and the dlarfg scalars); the rest of step t (the updates of D_t and B_t and their stores) is needed by the NEXT
// sweep.  So the steps of a sweep alternate between the two waves of a pair: a wave publishes v_{t+1} through a 17-double
// mailbox in LDS as soon as it has it, and its partner starts step t+1 while it finishes step t.  Both waves run the same
// code; a sweep now advances one step per max(chain part + hand-off, step / 2).  Progress counters stay ordered: the wave
// of step t publishes t+1 only after it has seen t.
constexpr int SB2ST_MBOX = 24;       // doubles per mailbox slot: v[16] | tau
template <int NP>
__global__ __launch_bounds__(NP * 128) void k_sb2st_chase3(const double *__restrict__ ABin, int n, double *__restrict__ D,
                                                           double *__restrict__ E, double *__restrict__ vlog, int tmax, int *err)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    constexpr int RING = 4;
    double *band = lds;                                             // (n + 16) * 32
    double *scr = band + (size_t)(n + 16) * SBW + (threadIdx.x >> 6) * 48;   // per wave: w | tau z | x
    double *mbase = band + (size_t)(n + 16) * SBW + 2 * NP * 48;    // NP * RING * SB2ST_MBOX
    int *prog = (int *)(mbase + NP * RING * SB2ST_MBOX);            // n progress counters, abort word, NP mailbox counters
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int pair = wave >> 1, me = wave & 1;
    const int j = lane & 15, g = lane >> 4;
    for (int i = tid; i < (n + 16) * SBW; i += NP * 128) band[i] = (i < n * SBW) ? ABin[i] : 0.0;
    for (int i = tid; i < n + 1 + NP; i += NP * 128) prog[i] = 0;
    __syncthreads();
    int *abortw = prog + n, *mready = prog + n + 1 + pair;
    double *mbox = mbase + (size_t)pair * RING * SB2ST_MBOX;

    int offD[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = 4 * g + r;
        offD[r] = (i >= j) ? j * SBW + (i - j) : i * SBW + (j - i);
    }
    const int offC = j * SBW + SB + 4 * g - j;                      // B[4 g + r][j] at offC + r (layout C)
    const int offR = 4 * g * SBW + SB + j - 4 * g;                  // B[j][4 g + r] at offR + r (SBW - 1) (layout R)
    const int off0 = SB + j;                                        // B[row = lane & 15][0]
    const bool g0 = g == 0, j0 = j == 0;

    auto wait_ge = [&](const int *word, int need) -> bool {
        for (unsigned spins = 0;; ++spins) {
            const int p = __builtin_amdgcn_readfirstlane(__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
            if (p >= need) break;
            __builtin_amdgcn_s_sleep(1);
            if ((spins & 1023u) == 1023u) {
                const int ab = __builtin_amdgcn_readfirstlane(__hip_atomic_load(abortw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
                if (ab != 0 || spins > (1u << 24)) {
                    __hip_atomic_store(abortw, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    return false;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        return true;
    };

    bool alive = true;
    int k = 0;                                                      // steps of this pair so far; step k belongs to wave k & 1
    SB2ST_T(tk0);
    for (int s = pair; s < n - 2 && alive; s += NP) {
        const int nst = (n - s - 3) / SB + 1;
        double *base = band + (size_t)(s + 1) * SBW;
        double *lg = vlog + (size_t)s * tmax * 16 + j;
        for (int t = 0; t < nst; ++t, base += SB * SBW, lg += 16, ++k) {
            if ((k & 1) != me) continue;
            SB2ST_T(ts0);
            if (s > 0 && !wait_ge(prog + s - 1, t + 2)) { alive = false; break; }
            double d[4], bc[4], br[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                d[r] = base[offD[r]];
                bc[r] = base[offC + r];
                br[r] = base[offR + r * (SBW - 1)];
            }
            const double b0 = base[off0];
            double vj, vr[4], tau;
            if (t == 0) {                                           // the reflector that clears column s below the sub-diagonal
                double *col = band + (size_t)s * SBW + 1;
                const double xj = col[j];
                double xr[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) xr[r] = col[4 * g + r];
                const double alpha = lane0_f64(xj);
                const double x0 = g0 ? 0.0 : xr[0];
                const double sigma = rows_sum(x0 * x0 + xr[1] * xr[1] + xr[2] * xr[2] + xr[3] * xr[3]);
                double beta, scale;
                house_scalars(alpha, sigma, tau, beta, scale);
                vj = j0 ? 1.0 : xj * scale;
                vr[0] = g0 ? 1.0 : xr[0] * scale;
#pragma unroll
                for (int r = 1; r < 4; ++r) vr[r] = xr[r] * scale;
                if (g0) col[j] = j0 ? beta : 0.0;
            } else {                                                // from the partner's step t - 1
                if (!wait_ge(mready, k)) { alive = false; break; }
                const double *mb = mbox + (size_t)(k % RING) * SB2ST_MBOX;
                vj = mb[j];
                tau = mb[16];
#pragma unroll
                for (int r = 0; r < 4; ++r) vr[r] = mb[4 * g + r];
            }
            SB2ST_T(ts1);
            SB2ST_ADD(0, ts1 - ts0);
            if (g0) *lg = j0 ? tau : vj;
            // ---- chain part: z = B v, x = first column of B H, the next reflector -> mailbox
            const double tz = tau * rows_sum(br[0] * vr[0] + br[1] * vr[1] + br[2] * vr[2] + br[3] * vr[3]);
            const double x = b0 - tz;
            const double u = rows_sum(d[0] * vr[0] + d[1] * vr[1] + d[2] * vr[2] + d[3] * vr[3]);
            if (g0) {
                scr[16 + j] = tz;
                scr[32 + j] = x;
            }
            const double alpha2 = lane0_f64(x);
            const double sigma2 = row16_sum(j0 ? 0.0 : x * x);
            double tau2, beta2, scale2;
            house_scalars(alpha2, sigma2, tau2, beta2, scale2);
            const double v2j = j0 ? 1.0 : x * scale2;
            if (t + 1 < nst) {
                double *mb = mbox + (size_t)((k + 1) % RING) * SB2ST_MBOX;
                if (g0) mb[j] = v2j;
                if (lane == 0) mb[16] = tau2;
                __hip_atomic_store(mready, k + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            SB2ST_T(ts2);
            SB2ST_ADD(5, ts2 - ts1);
            // ---- update part
            const double gamma = row16_sum(u * vj);
            const double w = tau * (u - (0.5 * tau * gamma) * vj);
            if (g0) scr[j] = w;
            double wr[4], zr[4], xr[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                wr[r] = scr[4 * g + r];
                zr[r] = scr[16 + 4 * g + r];
                xr[r] = scr[32 + 4 * g + r];
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) base[offD[r]] = d[r] - __dadd_rn(__dmul_rn(vr[r], w), __dmul_rn(wr[r], vj));   // (no fma: symmetry)
#pragma unroll
            for (int r = 0; r < 4; ++r) bc[r] -= zr[r] * vj;
            double v2r[4];
            v2r[0] = g0 ? 1.0 : xr[0] * scale2;
#pragma unroll
            for (int r = 1; r < 4; ++r) v2r[r] = xr[r] * scale2;
            const double tg = tau2 * rows_sum(v2r[0] * bc[0] + v2r[1] * bc[1] + v2r[2] * bc[2] + v2r[3] * bc[3]);
#pragma unroll
            for (int r = 0; r < 4; ++r) bc[r] -= v2r[r] * tg;
            if (j0) {
                bc[0] = g0 ? beta2 : 0.0;
                bc[1] = bc[2] = bc[3] = 0.0;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) base[offC + r] = bc[r];
            SB2ST_T(ts3);
            if (t > 0 && !wait_ge(prog + s, t)) { alive = false; break; }              // the partner's step t - 1 is complete
            __hip_atomic_store(prog + s, (t + 1 < nst) ? t + 1 : SB2ST_DONE, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            SB2ST_T(ts4);
            SB2ST_ADD(1, ts3 - ts2);
            SB2ST_ADD(3, ts4 - ts3);
            SB2ST_ADD(2, 1ULL);
        }
    }
    SB2ST_T(tk1);
    SB2ST_ADD(4, tk1 - tk0);
    __syncthreads();
    if (tid == 0 && *abortw != 0) *err = 2;
    for (int i = tid; i < n; i += NP * 128) {
        D[i] = band[(size_t)i * SBW];
        if (i < n - 1) E[i] = band[(size_t)i * SBW + 1];
    }
}

// Q <- Q H_(0,0) H_(0,1) ... H_(1,0) ... (the order of the sequential algorithm): ONE ROW of Q per wave, in LDS.  The steps
// of one sweep touch disjoint 16-column windows, so the four 16-lane rows of the wave take four consecutive steps of
// the sweep at once: lane l works on column s + 1 + 64 grp + l with the log entry at the same offset — every access is
// 64 consecutive doubles.  n waves of 2 n^2 / 64 wave-instructions each (a wave with four rows and one step per
// instruction — 4 x fewer, 4 x longer waves — took 1.03 ms at n = 510).
// The log entries of sweep s + 1 are fetched (MAXG loads per lane) while sweep s is applied, and the MAXG groups of a sweep
// are independent chains: neither the L2 latency of the log nor the LDS round trip of a window is paid per group.
template <int NWQ, int MAXG>
__global__ __launch_bounds__(NWQ * 64) void k_sb2st_apply_q(double *__restrict__ Q, int ldq, int n, const double *__restrict__ vlog,
                                                           int tmax)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int ldr = n + 144;                                       // 80 zero columns behind the row + 64 of scratch for idle groups
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row = blockIdx.x * NWQ + wave;
    if (row >= n) return;
    double *q = lds + (size_t)wave * ldr;
    for (int c = lane; c < ldr; c += 64) q[c] = c < n ? Q[(size_t)row * ldq + c] : 0.0;
    const bool j0 = (lane & 15) == 0;
    const int dummy = n + 80 + lane;
    auto fetch = [&](int s, double (&v)[MAXG]) {
        const int nst = (s < n - 2) ? (n - s - 3) / SB + 1 : 0;
        const double *vl = vlog + (size_t)s * tmax * 16 + lane;
#pragma unroll
        for (int e = 0; e < MAXG; ++e) v[e] = (4 * e + (lane >> 4) < nst) ? vl[64 * e] : 0.0;   // steps past the end: H = I
    };
    // G groups of the current sweep: G independent chains (read window, 16-lane dot product, update, write back)
    auto apply = [&](auto G_, int s, int nst, const double (&v)[MAXG]) {
        constexpr int G = decltype(G_)::value;
        int idx[G];
        double x[G];
#pragma unroll
        for (int e = 0; e < G; ++e) {
            idx[e] = (4 * e < nst) ? s + 1 + 64 * e + lane : dummy;
            x[e] = q[idx[e]];
        }
#pragma unroll
        for (int e = 0; e < G; ++e) {
            const double tau = dpp_f64<0x150, 0xf>(v[e]);                  // row_newbcast:0 — slot 0 of an entry holds tau
            const double ve = j0 ? 1.0 : v[e];
            const double dt = tau * row16_sum(x[e] * ve);
            q[idx[e]] = x[e] - dt * ve;
        }
    };
    double v0[MAXG], v1[MAXG], v2[MAXG], v3[MAXG];                         // the log three sweeps ahead (L2 latency ~ 1 us)
    fetch(0, v0);
    fetch(1, v1);
    fetch(2, v2);
    for (int s = 0; s < n - 2; ++s) {
        fetch(s + 3, v3);
        const int nst = (n - s - 3) / SB + 1;
        if (nst > 24) apply(std::integral_constant<int, MAXG>(), s, nst, v0);
        else if (nst > 12) apply(std::integral_constant<int, 6>(), s, nst, v0);
        else apply(std::integral_constant<int, 3>(), s, nst, v0);
#pragma unroll
        for (int e = 0; e < MAXG; ++e) {
            v0[e] = v1[e];
            v1[e] = v2[e];
            v2[e] = v3[e];
        }
    }
    for (int c = lane; c < n; c += 64) Q[(size_t)row * ldq + c] = q[c];
}

}  // namespace jcdf
