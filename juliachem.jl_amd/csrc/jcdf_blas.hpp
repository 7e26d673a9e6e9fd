// jcdf_blas.hpp — the small dense products of the device-resident SCF iteration (caller side of the hot path, SURVEY 8
// row f1) on the library's own fp64 MFMA cores, so that a step issues no vendor BLAS kernel:
//   F D S - (F D S)^T   SCF.jl:473-481        X F X, C = X U        SCF.jl:1080-1100
//   D = 2 C_o C_o^T     SCF.jl:1106-1108      U = Q Z (back-transformation of the eigensolver, jcdf_eig.hpp / jcdf_dc.hpp)
//   the two reductions over the DIIS history (EnergyHelpers.jl:234-258): dots <e_s, e_new>, F = sum_s c_s F_s
// All matrices are row-major with a leading dimension that is a multiple of 32 and zero padding up to it (symmetric
// ones: row-major == column-major); 32 x 32 output tiles, one workgroup per tile over the whole contraction length
// (N = 510: 256 tiles = one per CU; the same shape as the spectral-projection kernel, 14 us per N^3 product where the
// vendor GEMM of the same size takes 40-60 us).
#pragma once
#include "jcdf_gemm.hpp"

namespace jcdf {

typedef GemmCfg<1, 1, 2, 2, 32> BlasTNCfg;       // 32 x 32 tile, 4 waves of 16 x 16, 32 k rows per LDS stage
typedef GemmCfg<1, 1, 2, 2, 16> BlasNTCfg;       // the same tile on the NT core (16 k per stage)

// C[m][n] = alpha * sum_k A[k][m] B[k][n]   (both operands k-major); M, N, K multiples of 32.
// sym: only tiles with tm >= tn are computed and mirrored (C symmetric by construction, e.g. X (F X), D = 2 Co^T Co).
__global__ __launch_bounds__(BlasTNCfg::NT) void k_blas_gemm_tn(const double *__restrict__ A, int64_t lda, const double *__restrict__ B,
                                                                int64_t ldb, double *__restrict__ C, int64_t ldc, int kchunks,
                                                                double alpha, int n_tn)
{
    using Cfg = BlasTNCfg;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int tm = blockIdx.x / n_tn, tn = blockIdx.x % n_tn;
    double4_t acc[1][1];
    acc[0][0] = double4_t{0.0, 0.0, 0.0, 0.0};
    gemm_tn_core<Cfg, false, 0, 2>(A + tm * 32, lda, B + tn * 32, ldb, kchunks, acc, smem);
    const int col = tn * 32 + tile_col<Cfg>(0);
#pragma unroll
    for (int j = 0; j < 4; ++j) C[(int64_t)(tm * 32 + tile_row<Cfg>(0, j)) * ldc + col] = alpha * acc[0][0][j];
}

// C[m][n] = sum_k A[m][k] B[n][k]   (both operands k-contiguous); M, N multiples of 32, K of 16.
__global__ __launch_bounds__(BlasNTCfg::NT) void k_blas_gemm_nt(const double *__restrict__ A, int64_t lda, const double *__restrict__ B,
                                                                int64_t ldb, double *__restrict__ C, int64_t ldc, int kchunks, int n_tn)
{
    using Cfg = BlasNTCfg;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int tm = blockIdx.x / n_tn, tn = blockIdx.x % n_tn;
    double4_t acc[1][1];
    acc[0][0] = double4_t{0.0, 0.0, 0.0, 0.0};
    gemm_nt_core<Cfg>(A + (int64_t)tm * 32 * lda, lda, B + (int64_t)tn * 32 * ldb, ldb, kchunks, acc, smem);
    const int col = tn * 32 + tile_col<Cfg>(0);
#pragma unroll
    for (int j = 0; j < 4; ++j) C[(int64_t)(tm * 32 + tile_row<Cfg>(0, j)) * ldc + col] = acc[0][0][j];
}

// DIIS bookkeeping of one iteration: e = T^T - T (T = S D F = (F D S)^T, ld), packed N x N into slot `head` of the error
// history, the Fock matrix into slot `head` of the Fock history.
__global__ __launch_bounds__(256) void k_diis_push(const double *__restrict__ T, const double *__restrict__ F, int64_t ld, int n,
                                                   double *__restrict__ e_slot, double *__restrict__ f_slot)
{
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)n * n) return;
    const int r = (int)(idx / n), c = (int)(idx % n);
    e_slot[idx] = T[(int64_t)c * ld + r] - T[(int64_t)r * ld + c];
    f_slot[idx] = F[(int64_t)r * ld + c];
}

// dots[s] = <e_hist[s], e_hist[head]>: DIIS_DOT_PARTS workgroups per slot leave partial sums, a second launch adds them
// in fixed order (one workgroup per slot alone took 260 us at N = 510: 10 workgroups on a 256-CU chip).
constexpr int DIIS_DOT_PARTS = 64;
__global__ __launch_bounds__(256) void k_diis_dots_partial(const double *__restrict__ e_hist, int64_t len, int head,
                                                           double *__restrict__ part)
{
    __shared__ double red[256];
    const int s_ = blockIdx.y;
    const double *a = e_hist + (int64_t)s_ * len, *b = e_hist + (int64_t)head * len;
    const int64_t per = (len + DIIS_DOT_PARTS - 1) / DIIS_DOT_PARTS, i0 = blockIdx.x * per, i1 = min(len, i0 + per);
    double s = 0.0;
    for (int64_t i = i0 + threadIdx.x; i < i1; i += 256) s += a[i] * b[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int h = 128; h > 0; h >>= 1) {
        if ((int)threadIdx.x < h) red[threadIdx.x] += red[threadIdx.x + h];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[s_ * DIIS_DOT_PARTS + blockIdx.x] = red[0];
}

__global__ void k_diis_dots_final(const double *__restrict__ part, int nd, double *__restrict__ dots)
{
    const int s_ = threadIdx.x;
    if (s_ >= nd) return;
    double s = 0.0;
    for (int k = 0; k < DIIS_DOT_PARTS; ++k) s += part[s_ * DIIS_DOT_PARTS + k];
    dots[s_] = s;
}

// F[r][c] (ld) = sum_s coef[s] F_hist[s][r*n + c]  (slots with coef 0 are skipped: unused history)
__global__ __launch_bounds__(256) void k_diis_mix(const double *__restrict__ f_hist, int64_t len, int nd, const double *__restrict__ coef,
                                                  int n, double *__restrict__ F, int64_t ld)
{
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= len) return;
    double s = 0.0;
    for (int k = 0; k < nd; ++k) {
        const double c = coef[k];
        if (c != 0.0) s += c * f_hist[(int64_t)k * len + idx];
    }
    F[(int64_t)(idx / n) * ld + (idx % n)] = s;
}

}  // namespace jcdf
