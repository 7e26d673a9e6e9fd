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

// C[m][n] = alpha * sum_k A[k][m] B[k][n]   (both operands k-major); M, N multiples of the tile edge (32; 64 for the
// 8-wave form used on large products: half the operand traffic per flop, see Sp2Cfg64), K a multiple of 32.
typedef GemmCfg<2, 1, 2, 4, 32> BlasTN64Cfg;     // 64 x 64 tile, 8 waves of 32 x 16
template <class Cfg>
__global__ __launch_bounds__(Cfg::NT) void k_blas_gemm_tn(const double *__restrict__ A, int64_t lda, const double *__restrict__ B,
                                                          int64_t ldb, double *__restrict__ C, int64_t ldc, int kchunks,
                                                          double alpha, int n_tn)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int tm = blockIdx.x / n_tn, tn = blockIdx.x % n_tn;
    double4_t acc[Cfg::WM][Cfg::WN];
#pragma unroll
    for (int m = 0; m < Cfg::WM; ++m)
#pragma unroll
        for (int n = 0; n < Cfg::WN; ++n) acc[m][n] = double4_t{0.0, 0.0, 0.0, 0.0};
    gemm_tn_core<Cfg, false, 0, 2>(A + tm * Cfg::TM, lda, B + tn * Cfg::TN, ldb, kchunks, acc, smem);
#pragma unroll
    for (int m = 0; m < Cfg::WM; ++m)
#pragma unroll
        for (int n = 0; n < Cfg::WN; ++n) {
            const int col = tn * Cfg::TN + tile_col<Cfg>(n);
#pragma unroll
            for (int j = 0; j < 4; ++j) C[(int64_t)(tm * Cfg::TM + tile_row<Cfg>(m, j)) * ldc + col] = alpha * acc[m][n][j];
        }
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

// C = alpha A^T B + diag I for square operands, written in BOTH orientations: C[tm][tn] and Ct[tn][tm] = C^T — so that a
// chain of products never needs an explicit transpose (TN core: the left factor enters transposed).  Two independent
// problems per launch (blockIdx.y); optional per-tile partial sum of ||C - I||_F^2.  smem after the product: 32 x 33 doubles.
struct NsProblem { const double *At, *B; double *C, *Ct; };
__global__ __launch_bounds__(BlasTNCfg::NT) void k_ns_gemm(NsProblem p0, NsProblem p1, int64_t ld, int kchunks, int n_tn, double alpha, double diag,
                                                           double *__restrict__ part)
{
    using Cfg = BlasTNCfg;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ double red[256];
    const NsProblem pr = blockIdx.y ? p1 : p0;
    const int tm = blockIdx.x / n_tn, tn = blockIdx.x % n_tn;
    double4_t acc[1][1];
    acc[0][0] = double4_t{0.0, 0.0, 0.0, 0.0};
    gemm_tn_core<Cfg, false, 0, 2>(pr.At + tm * 32, ld, pr.B + tn * 32, ld, kchunks, acc, smem);
    __syncthreads();
    double (*T)[33] = reinterpret_cast<double (*)[33]>(smem);
    const int col = tile_col<Cfg>(0);
    double s = 0.0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = tile_row<Cfg>(0, j);
        const bool on_diag = tm == tn && row == col;
        const double v = alpha * acc[0][0][j] + (on_diag ? diag : 0.0);
        pr.C[(int64_t)(tm * 32 + row) * ld + tn * 32 + col] = v;
        T[col][row] = v;
        const double e = v - (on_diag ? 1.0 : 0.0);
        s += e * e;
    }
    __syncthreads();
    {
        const int c = threadIdx.x >> 3, rr = (threadIdx.x & 7) * 4;            // row c of the transposed tile, 4 of its columns
        double *dst = pr.Ct + (int64_t)(tn * 32 + c) * ld + tm * 32 + rr;
        *reinterpret_cast<double2_t *>(dst) = double2_t{T[c][rr], T[c][rr + 1]};
        *reinterpret_cast<double2_t *>(dst + 2) = double2_t{T[c][rr + 2], T[c][rr + 3]};
    }
    if (part) {
        red[threadIdx.x] = s;
        __syncthreads();
        for (int h = 128; h > 0; h >>= 1) {
            if ((int)threadIdx.x < h) red[threadIdx.x] += red[threadIdx.x + h];
            __syncthreads();
        }
        if (threadIdx.x == 0) part[blockIdx.x] = red[0];
    }
}

// ---- Loewdin orthonormalisation of o row vectors by Newton-Schulz (caller-side helper of the SP2 step for any number of
// occupied orbitals; the one-workgroup Cholesky of jcdf_scf.hpp holds at most 128 rows in LDS) ------------------------------
// Y (o x n rows, zero padded to op x np): G = Y Y^T;  coupled iteration Y_0 = G, Z_0 = I, T_k = (3 I - Z_k Y_k) / 2,
// Y_{k+1} = Y_k T_k, Z_{k+1} = T_k Z_k  ->  Z -> G^{-1/2} (quadratically once ||I - Z Y|| < 1; the eigenvalues of G are
// the cos^2 of the angles between the old and the new occupied space, in (0, 1]);  out = Z Y: orthonormal rows with the
// span of Y's, and of all such bases the one closest to Y.  Everything is an op x op product on the MFMA cores.
// The products are taken as written (k_ns_gemm keeps every iterate in both orientations): this coupled form is the
// numerically stable one (Higham 1997).  Measured alternatives that are NOT: A^T B in place of A B for the nearly
// symmetric factors amplifies the antisymmetric rounding error by 3/2 per step (3e-10 after 40 steps), and forcing
// symmetry by mirroring the lower triangle of each product makes the converged iteration drift away again (residual
// 1e-15 at step 12, 5e-4 at step 40).
// k_lowdin_prepare: Y_0 = Y_0^T = G made exactly symmetric (lower triangle mirrored) and padded with a unit diagonal (rows >= o
// are zero rows of Y), Z_0 = Z_0^T = I, ||I - G||_F^2 partial sums.
__global__ __launch_bounds__(256) void k_lowdin_prepare(const double *__restrict__ Gin, double *__restrict__ Y, double *__restrict__ Yt,
                                                        double *__restrict__ Z, double *__restrict__ Zt, int o, int op, double *__restrict__ part)
{
    __shared__ double red[256];
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    double s = 0.0;
    if (idx < (int64_t)op * op) {
        const int r = (int)(idx / op), c = (int)(idx % op);
        double g = (r >= c) ? Gin[idx] : Gin[(int64_t)c * op + r];
        if (r >= o && r == c) g = 1.0;
        Y[idx] = Yt[idx] = g;
        Z[idx] = Zt[idx] = (r == c) ? 1.0 : 0.0;
        const double d = ((r == c) ? 1.0 : 0.0) - g;
        s = d * d;
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int h = 128; h > 0; h >>= 1) {
        if ((int)threadIdx.x < h) red[threadIdx.x] += red[threadIdx.x + h];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}

// info = {||I - G||_F, iterations needed (first k with ||I - Z_k Y_k||_F < 2e-7, + 1; 0: not reached), ||I - Z Y||_F before the
// last step, iterations run}.  Fixed-order sums (bit-reproducible).
__global__ __launch_bounds__(64) void k_lowdin_info(const double *__restrict__ part0, int n0, const double *__restrict__ part, int ntile,
                                                    int iterations, double *__restrict__ info)
{
    __shared__ double res[64];
    const int t = threadIdx.x;
    // lane t sums the tiles of step t (and every 64th partial of ||I - G||^2), each in a fixed order
    double r = 0.0;
    if (t < iterations)
        for (int k = 0; k < ntile; ++k) r += part[(int64_t)t * ntile + k];
    res[t] = (t < iterations) ? 2.0 * sqrt(r) : 0.0;
    double s = 0.0;
    for (int i = t; i < n0; i += 64) s += part0[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    __syncthreads();
    if (t != 0) return;
    int used = 0;
    for (int k = 0; k < iterations; ++k)
        if (!used && res[k] < 2e-7) used = k + 1;
    info[0] = sqrt(s);
    info[1] = (double)used;
    info[2] = res[iterations - 1];
    info[3] = (double)iterations;
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

// F[r][c] (ld) = sum_s coef[s] F_hist[s][r*n + c]  (slots with coef 0 are skipped: unused history); with F_old (same ld) and
// x != 1 the dynamic damping of SCF.jl:504-505 is applied in the same pass: F = (1 - x) F_old + x sum_s ...
__global__ __launch_bounds__(256) void k_diis_mix(const double *__restrict__ f_hist, int64_t len, int nd, const double *__restrict__ coef,
                                                  int n, double *__restrict__ F, int64_t ld, const double *__restrict__ F_old = nullptr,
                                                  double x = 1.0)
{
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= len) return;
    double s = 0.0;
    for (int k = 0; k < nd; ++k) {
        const double c = coef[k];
        if (c != 0.0) s += c * f_hist[(int64_t)k * len + idx];
    }
    const int64_t at = (int64_t)(idx / n) * ld + (idx % n);
    F[at] = F_old ? (1.0 - x) * F_old[at] + x * s : s;
}

}  // namespace jcdf
