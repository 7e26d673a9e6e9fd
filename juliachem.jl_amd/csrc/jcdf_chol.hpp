// jcdf_chol.hpp — device Cholesky factorisation + triangular inverse of the DF metric (P|Q)
// (reference: LAPACK.potrf!('L') + trtri!('L','N') on the host at
// GPUDF.jl:890-891, CUSOLVER.potrf!/trtri! at DenseGPUDF.jl:185-193).
//
// Memory view: the reference's column-major lower triangle L[r + Q c] (r >= c) IS the row-major
// upper triangle U = L^T, U[c][r] at c*ld + r.  Working on U row-major makes every operand of the
// blocked algorithm k-major for the fp64 MFMA GEMM core (jcdf_gemm.hpp) and every store coalesced:
//   factor :  for each 64-row block i:  U11 = chol(R11);  U12 = U11^-T R12 (substitution);  R22 -= U12^T U12
//   inverse:  V = U^-1 (upper, row-major) == (L^-1)^T — exactly the k-major "LinvT" the metric-apply
//             kernel consumes.  Block rows from the bottom up:
//             V[i][i] = inv(U_ii),   V[i][j>i] = -U_ii^-1 ( U[i][i+1:] * V[i+1:][j] )  (substitution).
// The matrix is padded to a multiple of 128 (identity on the padding) plus 64 columns/rows of
// zero slack, so every tile of every kernel is full and in bounds.  The O(Q^3) work (trailing update,
// U*V products) is in MFMA GEMM kernels; the O(Q^2 * 64) triangular solves are register-resident
// substitutions; the 64 x 64 diagonal blocks are factored and inverted by one workgroup in LDS.
#pragma once
#include "jcdf_gemm.hpp"

namespace jcdf {

constexpr int CH_NB = 64;

// identity on the padding diagonal [Q, n_pad) (the rest of R was zero-filled / copied from the host:
// row r of the row-major upper view == column r of the caller's column-major lower triangle)
__global__ void k_chol_pad_diag(double *__restrict__ R, int64_t ld, int Q, int n_pad)
{
    const int i = Q + blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_pad) R[(int64_t)i * ld + i] = 1.0;
}

// One workgroup: U11 = chol_upper(R11) in place, inv(U11) -> invU (row-major 64x64).
// err[0] = 1-based index of the first non-positive pivot (0 = ok).
__global__ __launch_bounds__(256) void k_chol_diag(double *__restrict__ R, int64_t ld, int i0,
                                                   double *__restrict__ invU, int *__restrict__ err)
{
    extern __shared__ __attribute__((aligned(16))) double sm[];
    double(*a)[CH_NB + 1] = reinterpret_cast<double(*)[CH_NB + 1]>(sm);
    double(*v)[CH_NB + 1] = reinterpret_cast<double(*)[CH_NB + 1]>(sm + CH_NB * (CH_NB + 1));
    const int tid = threadIdx.x;
    if (err[0] != 0) return;                              // an earlier block already failed
    for (int idx = tid; idx < CH_NB * CH_NB; idx += 256) {
        const int r = idx / CH_NB, c = idx % CH_NB;
        a[r][c] = (c >= r) ? R[(int64_t)(i0 + r) * ld + i0 + c] : 0.0;
        v[r][c] = 0.0;
    }
    __syncthreads();
    for (int k = 0; k < CH_NB; ++k) {                     // right-looking: scale row k, rank-1 update below
        const double d = a[k][k];
        if (!(d > 0.0)) {                                 // uniform: every thread reads the same LDS word
            if (tid == 0) err[0] = i0 + k + 1;
            return;
        }
        const double s = 1.0 / sqrt(d);
        __syncthreads();
        for (int c = k + tid; c < CH_NB; c += 256) a[k][c] *= s;
        __syncthreads();
        const int w = CH_NB - k - 1;
        for (int idx = tid; idx < w * w; idx += 256) {
            const int r = k + 1 + idx / w, c = k + 1 + idx % w;
            if (c >= r) a[r][c] -= a[k][r] * a[k][c];
        }
        __syncthreads();
    }
    if (tid < CH_NB) {                                    // column c of U^-1 by back substitution
        const int c = tid;
        for (int r = c; r >= 0; --r) {
            double s = (r == c) ? 1.0 : 0.0;
            for (int k = r + 1; k <= c; ++k) s -= a[r][k] * v[k][c];
            v[r][c] = s / a[r][r];
        }
    }
    __syncthreads();
    for (int idx = tid; idx < CH_NB * CH_NB; idx += 256) {
        const int r = idx / CH_NB, c = idx % CH_NB;
        if (c >= r) R[(int64_t)(i0 + r) * ld + i0 + c] = a[r][c];
        invU[r * CH_NB + c] = v[r][c];
    }
}

using C64Cfg = GemmCfg<2, 2, 2, 4, 16>;      // 64 x 128 tile, 8 waves
using C128Cfg = GemmCfg<4, 4, 2, 2, 16>;     // 128 x 128 tile, 4 waves

// Triangular solve with a 64 x 64 diagonal block, one thread per right-hand-side column (its 64
// unknowns live in registers, the triangle is broadcast from LDS) — substitution, not a product
// with the explicit inverse, so the factorisation keeps LAPACK's backward stability on the
// ill-conditioned metrics of diffuse auxiliary sets.
//   FORWARD : X <- U11^-T X        (panel solve  U12 = U11^-T R12, in place)
//   !FORWARD: X <- -U_ii^-1 Y      (inverse step V[i][j>i] = -U_ii^-1 T)
// Utri = the upper-triangular block (row-major, leading dimension ld) at R + i0*ld + i0.
template <bool FORWARD>
__global__ __launch_bounds__(64) void k_chol_trsm(const double *__restrict__ Utri, int64_t ld, const double *Y,
                                                  int64_t ldy, double *X, int64_t ldx, int64_t ncols)
{
    __shared__ double tri[CH_NB][CH_NB + 1];
    const int tid = threadIdx.x;
    for (int r = 0; r < CH_NB; ++r) tri[r][tid] = (tid >= r) ? Utri[(int64_t)r * ld + tid] : 0.0;
    __syncthreads();
    const int64_t c = (int64_t)blockIdx.x * 64 + tid;
    if (c >= ncols) return;
    double x[CH_NB];
#pragma unroll
    for (int r = 0; r < CH_NB; ++r) x[r] = FORWARD ? Y[(int64_t)r * ldy + c] : -Y[(int64_t)r * ldy + c];
    if (FORWARD) {
#pragma unroll
        for (int j = 0; j < CH_NB; ++j) {
            x[j] /= tri[j][j];
#pragma unroll
            for (int k = j + 1; k < CH_NB; ++k) x[k] -= tri[j][k] * x[j];
        }
    } else {
#pragma unroll
        for (int j = CH_NB - 1; j >= 0; --j) {
            x[j] /= tri[j][j];
#pragma unroll
            for (int r = 0; r < j; ++r) x[r] -= tri[r][j] * x[j];
        }
    }
#pragma unroll
    for (int r = 0; r < CH_NB; ++r) X[(int64_t)r * ldx + c] = x[r];
}

// R22[r][c] -= sum_{k < 64} P[k][r] P[k][c] for the upper block-triangle of 128 x 128 tiles
// (P = the 64-row panel U12, R22 = trailing matrix; both start at column/row `off`).
__global__ __launch_bounds__(256, 2) void k_chol_syrk(const double *__restrict__ P, double *__restrict__ R22,
                                                      int64_t ld, int ntile)
{
    using Cfg = C128Cfg;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    int t = blockIdx.x, tr = 0;                            // upper-triangle tile (tr <= tc)
    while (t >= ntile - tr) { t -= ntile - tr; ++tr; }
    const int tc = tr + t;
    double4_t acc[Cfg::WM][Cfg::WN];
#pragma unroll
    for (int m = 0; m < Cfg::WM; ++m)
#pragma unroll
        for (int n = 0; n < Cfg::WN; ++n) acc[m][n] = double4_t{0.0, 0.0, 0.0, 0.0};
    gemm_tn_core<Cfg, false>(P + (int64_t)tr * 128, ld, P + (int64_t)tc * 128, ld, CH_NB / 16, acc, smem);
#pragma unroll
    for (int m = 0; m < Cfg::WM; ++m)
#pragma unroll
        for (int n = 0; n < Cfg::WN; ++n)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                double *dst = R22 + ((int64_t)tr * 128 + tile_row<Cfg>(m, j)) * ld + (int64_t)tc * 128 + tile_col<Cfg>(n);
                *dst -= acc[m][n][j];
            }
}

// UT[k][r] = U[i0 + r][k0 + k]  (64-row panel -> k-major), k < nk
__global__ __launch_bounds__(256) void k_chol_transpose_panel(const double *__restrict__ U, int64_t ld, int i0,
                                                              int64_t k0, int64_t nk, double *__restrict__ UT)
{
    __shared__ double tile[64][65];
    const int64_t kb = (int64_t)blockIdx.x * 64;
    for (int idx = threadIdx.x; idx < 64 * 64; idx += 256) {
        const int r = idx / 64, k = idx % 64;
        tile[r][k] = (kb + k < nk) ? U[(int64_t)(i0 + r) * ld + k0 + kb + k] : 0.0;
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < 64 * 64; idx += 256) {
        const int k = idx / 64, r = idx % 64;
        if (kb + k < nk) UT[(kb + k) * 64 + r] = tile[r][k];
    }
}

// T[r][x] = sum_{k} UT[k][r] * V[k0 + k][x0 + x]  with k limited to the rows where V is non-zero
// for this column tile (V upper triangular): k < (tile end).  64 x 128 tile per workgroup.
__global__ __launch_bounds__(512) void k_chol_inv_gemm(const double *__restrict__ UT, const double *__restrict__ V,
                                                       int64_t ld, int64_t k0, double *__restrict__ T, int64_t ldt)
{
    using Cfg = C64Cfg;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int64_t x0 = (int64_t)blockIdx.x * Cfg::TN;      // column offset relative to k0
    const int nchunks = (int)((x0 + Cfg::TN) / 16);         // V[k0+k][k0+x] == 0 for k > x
    double4_t acc[Cfg::WM][Cfg::WN];
#pragma unroll
    for (int m = 0; m < Cfg::WM; ++m)
#pragma unroll
        for (int n = 0; n < Cfg::WN; ++n) acc[m][n] = double4_t{0.0, 0.0, 0.0, 0.0};
    gemm_tn_core<Cfg, false>(UT, CH_NB, V + k0 * ld + k0 + x0, ld, nchunks, acc, smem);
#pragma unroll
    for (int m = 0; m < Cfg::WM; ++m)
#pragma unroll
        for (int n = 0; n < Cfg::WN; ++n)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                T[(int64_t)tile_row<Cfg>(m, j) * ldt + x0 + tile_col<Cfg>(n)] = acc[m][n][j];
}

// V[i0 + r][i0 + c] = invU[r][c]   (diagonal block of the inverse)
__global__ void k_chol_put_diag(const double *__restrict__ invU, double *__restrict__ V, int64_t ld, int i0)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= CH_NB * CH_NB) return;
    const int r = idx / CH_NB, c = idx % CH_NB;
    V[(int64_t)(i0 + r) * ld + i0 + c] = invU[idx];
}

}  // namespace jcdf
