// jcdf_wy.hpp — back-transformation of the replicated eigensolve for matrices whose orthogonal factor cannot be accumulated
// inside the tridiagonalisation kernel (n > 1536: rows of Q no longer fit the register file of a workgroup, jcdf_eig.hpp).
// Reference step: eigen!(Hermitian(.)) at /root/reference/src/rhf/energy/SCF.jl:1083 (LAPACK dsyevd: dsytrd + dstedc +
// dormtr); this file is the dormtr('L','L','N') part — C <- Q C, Q = H_0 H_1 ... H_{n-3} from the reflectors
// jcdf_sytrd_device leaves below the sub-diagonal of A — on the library's own fp64 MFMA cores (round 3 called the vendor's
// rocsolver_dormtr here: 5.8 of the 20.2 ms of an n = 1915 eigensolve, profiles/r04_eigh_stages.txt).
//
// Blocked compact-WY (Schreiber / Van Loan; LAPACK dlarft 'F','C' + dlarfb): WY_NB consecutive reflectors are one factor
//   H_{j0} ... H_{j0+NB-1} = I - V T V^T,   T upper triangular,  T[0:j, j] = -tau_j T[0:j,0:j] (V[:,0:j]^T v_j),  T[j][j] = tau_j,
// applied to the eigenvector matrix as  Z <- Z - U (V^T Z)  with  U = V T  — two GEMMs per block, the blocks from the last
// to the first.  The eigenvectors are held TRANSPOSED (Zt[j][k] = Z[k][j], what the divide & conquer writes), so every
// operand is k-contiguous:
//   k_wy_extract   Vt[j][k] = v_j[k]  (0 for k <= j, 1 at k = j+1, A(k,j) below), zero padded            one launch
//   k_wy_S, k_wy_T, k_wy_U   all blocks at once: S = V^T V (NT MFMA core), T from the recurrence (one workgroup per block),
//                  Ut[m'][k] = sum_m T[m][m'] Vt[m][k] (TN MFMA core)                                      three launches
//                  (a first version did S and Ut with VALU loops over LDS slabs inside the T kernel: 1.6 of the 3.0 ms)
//   per block b:   W[m][j]  = sum_k Vt_b[m][k] Zt[j][k]        NT MFMA core (k_blas_gemm_nt), k from the block's first row on
//                  Zt[j][k] -= sum_m W[m][j] Ut_b[m][k]         TN MFMA core, accumulate (k_wy_update)
// Work: 2 n^3 flop in all (the leading zeros of the reflectors are skipped), 2 launches per WY_NB = 128 reflectors.
#pragma once
#include "jcdf_blas.hpp"

namespace jcdf {

constexpr int WY_NB = 128;         // reflectors per block (64: twice the launches, 1.36 instead of 0.8 ms at n = 1915)

// Vt[j][k], j < nrp (reflector), k < npad: cleaned reflectors from LAPACK storage (A column-major, v_j below the sub-diagonal of column j)
__global__ __launch_bounds__(256) void k_wy_extract(const double *__restrict__ A, int64_t lda, int n, double *__restrict__ Vt, int64_t ldv,
                                                    int nrp, int npad)
{
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)nrp * npad) return;
    const int j = (int)(idx / npad), k = (int)(idx % npad);
    double v = 0.0;
    if (j < n - 2 && k < n) v = (k <= j) ? 0.0 : (k == j + 1 ? 1.0 : A[(int64_t)j * lda + k]);
    Vt[(int64_t)j * ldv + k] = v;
}

// S_b = V_b^T V_b for every block in one launch: grid (tiles of 32 x 32 in the NB x NB matrix, blocks); NT MFMA core over the
// rows the block's reflectors touch (v_j[k] = 0 for k <= j).  Sm: [nblk][NB][NB] row-major.
__global__ __launch_bounds__(BlasNTCfg::NT) void k_wy_S(const double *__restrict__ Vt, int64_t ldv, int npad, double *__restrict__ Sm)
{
    using Cfg = BlasNTCfg;
    constexpr int NB = WY_NB, TPR = NB / 32;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int b = blockIdx.y, tm = blockIdx.x / TPR, tn = blockIdx.x % TPR;
    const int64_t j0 = (int64_t)b * NB;
    const int kbeg = (int)(j0 / 16 * 16);
    const double *Vb = Vt + j0 * ldv + kbeg;
    double4_t acc[1][1];
    acc[0][0] = double4_t{0.0, 0.0, 0.0, 0.0};
    gemm_nt_core<Cfg>(Vb + (int64_t)tm * 32 * ldv, ldv, Vb + (int64_t)tn * 32 * ldv, ldv, (npad - kbeg) / 16, acc, smem);
    const int col = tn * 32 + tile_col<Cfg>(0);
#pragma unroll
    for (int j = 0; j < 4; ++j) Sm[((int64_t)b * NB + tm * 32 + tile_row<Cfg>(0, j)) * NB + col] = acc[0][0][j];
}

// T_b from S_b and tau (dlarft forward / columnwise), one workgroup per block, in place in LDS: column j of S is read only at
// step j, so T overwrites S column by column.  Four lanes per row share the triangular dot product of a step (the NB steps are one
// dependent chain: 379 -> ~130 us at NB = 128).  Tm: [nblk][NB][NB] row-major (T[m][m'], upper triangular).
__global__ __launch_bounds__(4 * WY_NB) void k_wy_T(const double *__restrict__ Sm, const double *__restrict__ TAU, int n, double *__restrict__ Tm)
{
    constexpr int NB = WY_NB, LS = NB + 1, NTH = 4 * NB;
    extern __shared__ __attribute__((aligned(16))) double S[];          // NB x (NB + 1)
    __shared__ double tau[NB];
    const int b = blockIdx.x, j0 = b * NB, tid = threadIdx.x, row = tid >> 2, part = tid & 3;
    if (tid < NB) tau[tid] = (j0 + tid < n - 2) ? TAU[j0 + tid] : 0.0;
    for (int e = tid; e < NB * NB; e += NTH) S[(e / NB) * LS + (e % NB)] = Sm[(int64_t)b * NB * NB + e];
    __syncthreads();
    for (int j = 0; j < NB; ++j) {
        double t = 0.0;
        if (row < j)
            for (int l = row + part; l < j; l += 4) t += S[row * LS + l] * S[l * LS + j];   // T[row][l] (already T) * (V^T v_j)[l]
        t += __shfl_xor(t, 1, 64);
        t += __shfl_xor(t, 2, 64);
        __syncthreads();
        if (part == 0) S[row * LS + j] = (row < j) ? -tau[j] * t : (row == j ? tau[j] : 0.0);
        __syncthreads();
    }
    for (int e = tid; e < NB * NB; e += NTH) Tm[(int64_t)b * NB * NB + e] = S[(e / NB) * LS + (e % NB)];
}

// Ut_b[m'][k] = sum_m T_b[m][m'] Vt_b[m][k] for every block in one launch: grid (NB / 32 x npad / 32 tiles, blocks); TN MFMA core
__global__ __launch_bounds__(BlasTNCfg::NT) void k_wy_U(const double *__restrict__ Tm, const double *__restrict__ Vt, int64_t ldv, int npad,
                                                        double *__restrict__ Ut)
{
    using Cfg = BlasTNCfg;
    constexpr int NB = WY_NB;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int b = blockIdx.y, n_tn = npad / 32, tm = blockIdx.x / n_tn, tn = blockIdx.x % n_tn;
    const int64_t j0 = (int64_t)b * NB;
    double4_t acc[1][1];
    acc[0][0] = double4_t{0.0, 0.0, 0.0, 0.0};
    if (tn * 32 + 31 > j0)                                              // the block's reflectors are zero up to row j0
        gemm_tn_core<Cfg, false, 0, 1>(Tm + (int64_t)b * NB * NB + tm * 32, NB, Vt + j0 * ldv + tn * 32, ldv, NB / 32, acc, smem);
    const int col = tn * 32 + tile_col<Cfg>(0);
#pragma unroll
    for (int j = 0; j < 4; ++j) Ut[(j0 + tm * 32 + tile_row<Cfg>(0, j)) * ldv + col] = acc[0][0][j];
}

// C[m][n] += alpha * sum_k A[k][m] B[k][n]  (TN core, 32 x 32 tiles; K a multiple of 32)
__global__ __launch_bounds__(BlasTNCfg::NT) void k_wy_update(const double *__restrict__ A, int64_t lda, const double *__restrict__ B, int64_t ldb,
                                                             double *__restrict__ C, int64_t ldc, int kchunks, double alpha, int n_tn)
{
    using Cfg = BlasTNCfg;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int tm = blockIdx.x / n_tn, tn = blockIdx.x % n_tn;
    double4_t acc[1][1];
    acc[0][0] = double4_t{0.0, 0.0, 0.0, 0.0};
    gemm_tn_core<Cfg, false, 0, 1>(A + tm * 32, lda, B + tn * 32, ldb, kchunks, acc, smem);
    const int col = tn * 32 + tile_col<Cfg>(0);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        double *c = C + (int64_t)(tm * 32 + tile_row<Cfg>(0, j)) * ldc + col;
        *c += alpha * acc[0][0][j];
    }
}

// dst[c][r] = src[r][c] on n x n (32 x 32 tiles through LDS; dst is written up to its zero padding: np x np tiles)
__global__ __launch_bounds__(256) void k_wy_transpose(const double *__restrict__ src, int64_t lds_, double *__restrict__ dst, int64_t ldd, int n)
{
    __shared__ double t[32][33];
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    for (int e = threadIdx.x; e < 1024; e += 256) {
        const int r = e >> 5, c = e & 31;
        t[r][c] = (r0 + r < n && c0 + c < n) ? src[(int64_t)(r0 + r) * lds_ + c0 + c] : 0.0;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 1024; e += 256) {
        const int c = e >> 5, r = e & 31;
        dst[(int64_t)(c0 + c) * ldd + r0 + r] = t[r][c];
    }
}

}  // namespace jcdf
