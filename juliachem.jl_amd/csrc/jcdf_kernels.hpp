// jcdf_kernels.hpp — device kernels of the MI355X DF-RHF Fock build.
//
// HBM layout (all fp64):
//   B     [Ql][Nk][Np]   this shard of B = L^-1 (Q|qp); row q, column p, p fastest.
//                        Symmetric in (q,p); screened-out pairs and padding are 0.
//                        Nk = roundup(N,16) rows, Np = roundup(N,128) columns.
//   Cpad  [Np][opad]     occupied MO coefficients C[q][i], i fastest, zero padded.
//   W     [Wrows][Np]    exchange intermediate W[(Q*o + i)][p]; rows >= Ql*o are 0.
//   vpart [Ql][nvp]      per-workgroup partial sums of V[Q] (deterministic).
//   Jpart [SJ][Nk][Np]   per-aux-slice partial Coulomb, lower triangle (p <= q).
//   Kslab [S][ntri][128][128]  split-K partial exchange tiles (lower block-triangle).
#pragma once
#include "jcdf_gemm.hpp"

namespace jcdf {

constexpr int KC = 16;          // k rows per LDS stage
constexpr int TILE_P = 128;     // p-tile of every MFMA kernel == padding unit of Np

// ---------------------------------------------------------------------------
// C_occ (N x o, column-major, reference layout DensityFitting.jl:49) -> Cpad.
// Replaces the per-p gather buffer of build_non_zero_coefficients_kernel
// (GPUDF.jl:459-480): the W kernel reads C rows from LDS directly.
// ---------------------------------------------------------------------------
// Also writes Cperm: the same numbers pre-permuted into the fp64 MFMA accumulator
// layout of the W kernel's output tiles, so that its fused-V epilogue reads
// C[p][i] with fully coalesced 16-B loads:
//   Cperm[mt][ct][t][lane][e] = C[p = 16 ct + (lane&15)][i = mt*TMw + 16 (t/2) + (lane>>4) + 4 (2 (t&1) + e)]
// (ct = 16-column tile, t < 2*WMw, e < 2), i.e. element e of lane's t-th double2
// pairs with accumulator acc[m = t/2][.][j = 2 (t&1) + e].
__global__ void k_prep_C(const double *__restrict__ C, int N, int o, int Np, int opad, int WMw, int n_mtiles,
                         double *__restrict__ Cpad, double *__restrict__ Cperm)
{
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)Np * opad) return;
    const int q = (int)(idx / opad), i = (int)(idx % opad);
    const double v = (q < N && i < o) ? C[q + (int64_t)N * i] : 0.0;
    Cpad[idx] = v;
    const int TMw = 16 * WMw;
    if (i >= TMw * n_mtiles) return;                      // remainder orbitals (VALU path) are not in Cperm
    const int mt = i / TMw, il = i % TMw;
    const int m = il >> 4, lk = il & 3, j = (il & 15) >> 2;
    const int t = 2 * m + (j >> 1), e = j & 1;
    const int ct = q >> 4, lane = (q & 15) | (lk << 4);
    Cperm[((((int64_t)mt * (Np / 16) + ct) * (2 * WMw) + t) * 64 + lane) * 2 + e] = v;
}

// ---------------------------------------------------------------------------
// k_exchange_W: W[(Q,i)][p] = sum_q C[q][i] B[Q][q][p]  (one pass over B)
// fused with V[Q] = sum_{p,i} W[(Q,i)][p] C[p][i]  (== B_Q . D~, D~ = C C^T).
// Reference: calculate_W_screened_GPU (GPUDF.jl:637-667; N small GEMMs) /
// DenseGPUDF.jl:107 (W) and GPUDF.jl:539-542 / DenseGPUDF.jl:99 (V gemv, which
// costs the reference one extra pass over B).
//
// Workgroup = 4 waves (one per SIMD) side by side along p, each WM x 2 MFMA tiles:
// TN = 128 columns, all WM*16 orbitals of one aux index Q.  Two such workgroups
// are co-resident per CU, so the two waves sharing a SIMD's matrix pipe belong to
// different workgroups and do not hit their barriers together.
// Variants measured in one process on the C20H42 shape (tools/w_ablate.hip,
// profiles/r01_w_ablate.txt): this one 57-61 TF executed; 8 waves x (WM x 1) 59 TF;
// 8 waves, 256 columns 54 TF; B fragments loaded straight from HBM into registers
// (no LDS for B, barrier every 32 rows) 55-58 TF; MFMA + ds_read only (no loads,
// no barrier) 70 TF = the ceiling of this loop at the ~2.1 GHz the chip holds.
// The kernel is not MFMA-issue-bound: doing 80 of the 81 orbitals of C20H42 with 5
// MFMA row tiles and the 81st with VALU FMAs (1/6 fewer MFMAs) did not make it faster,
// so n_occ is simply padded to a multiple of 16.
// ---------------------------------------------------------------------------
// WVM = 1: 4 waves side by side, tile (16 WM) x 128.  WVM = 2 (more than 128 occupied
// orbitals): 8 waves as 2 x 4, tile (32 WM) x 128 — the B tile is staged once for both
// orbital halves instead of being re-read by a second workgroup.
template <int WM, int WVM = 1>
using WCfg = GemmCfg<WM, 2, WVM, 4, KC>;

template <int WM, int WVM = 1, bool FUSE_V = true>
__global__ __launch_bounds__(256 * WVM, (WVM == 1 && WM <= 6) ? 2 : ((WVM == 2) ? 2 : 1)) void k_exchange_W(
    const double *__restrict__ B, const double *__restrict__ Cpad, const double *__restrict__ Cperm,
    double *__restrict__ W, double *__restrict__ vpart, int Ql, int o, int Nk, int Np, int opad,
    int n_mtiles, int n_ntiles, const int *__restrict__ kptr, const int *__restrict__ klist)
{
    using Cfg = WCfg<WM, WVM>;
    extern __shared__ __attribute__((aligned(16))) double smem[];

    // XCD-aware decode: blocks b and b+8 share an XCD/L2; keep the m-tiles that
    // re-read the same B_Q tile on one XCD (speed only, never correctness).
    const int b = blockIdx.x;
    const int xcd = b & 7, r = b >> 3;
    const int mt = r % n_mtiles;
    const int64_t outer = (int64_t)(r / n_mtiles) * 8 + xcd;
    if (outer >= (int64_t)Ql * n_ntiles) return;          // whole workgroup exits together
    const int Q = (int)(outer / n_ntiles);
    const int nt = (int)(outer % n_ntiles);

    double4_t acc[WM][2];
#pragma unroll
    for (int m = 0; m < WM; ++m) acc[m][0] = acc[m][1] = double4_t{0.0, 0.0, 0.0, 0.0};

    const double *Ag = Cpad + mt * Cfg::TM_MFMA;
    const int emt = mt * WVM + (int)(threadIdx.x >> 6) / 4;           // 16*WM-row tile index of this wave (Cperm)
    const double *Bg = B + (int64_t)Q * Nk * Np + nt * Cfg::TN;
    // block sparsity: only the 16-row k stages in which this column tile has a kept (q,p) pair
    // (the Schwarz pattern does not depend on the aux index); dense map: all of them (kptr == null)
    const int k0 = kptr ? kptr[nt] : 0;
    const int nk = kptr ? kptr[nt + 1] - k0 : Nk / KC;
    // 2-stage-deep register prefetch everywhere: for <7,2>/<8,2> the second register set costs a few
    // dozen scratch spills but measured 3 % faster than the 1-deep variant on the (H2O)50 shape
    if (nk > 0) gemm_tn_core<Cfg, true, 0, 2>(Ag, opad, Bg, Np, nk, acc, smem, kptr ? klist + k0 : nullptr);

    double vsum = 0.0;
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const int p = nt * Cfg::TN + tile_col<Cfg>(n);
        if (FUSE_V) {   // C[p][i] in accumulator layout: 2*WM coalesced 16-B loads (zero where i >= o)
            const double2_t *cp = reinterpret_cast<const double2_t *>(Cperm) +
                                  ((int64_t)emt * (Np / 16) + (p >> 4)) * (2 * WM) * 64 + (threadIdx.x & 63);
#pragma unroll
            for (int t = 0; t < 2 * WM; ++t) {
                const double2_t c = cp[t * 64];
                vsum += acc[t >> 1][n][2 * (t & 1)] * c.x + acc[t >> 1][n][2 * (t & 1) + 1] * c.y;
            }
        }
#pragma unroll
        for (int m = 0; m < WM; ++m)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int i = mt * Cfg::TM_MFMA + tile_row<Cfg>(m, j);
                if (i < o) W[((int64_t)Q * o + i) * Np + p] = acc[m][n][j];
            }
    }
    if (!FUSE_V) return;
    // deterministic workgroup reduction: butterfly inside the wave, fixed order across waves
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) vsum += __shfl_xor(vsum, off, 64);
    if ((threadIdx.x & 63) == 0) smem[threadIdx.x >> 6] = vsum;   // all waves passed the core's last barrier
    __syncthreads();
    if (threadIdx.x == 0) {
        double sum = (smem[0] + smem[1]) + (smem[2] + smem[3]);
        if (WVM == 2) sum += (smem[4] + smem[5]) + (smem[6] + smem[7]);
        vpart[(int64_t)Q * (n_ntiles * n_mtiles) + nt * n_mtiles + mt] = sum;
    }
}

// ---------------------------------------------------------------------------
// k_coulomb_J: Jpart[s][q][p] = sum_{Q in slice s} V[Q] B[Q][q][p], p <= q only
// (B symmetric: the lower triangle is a contiguous prefix of every row, so this
// pass streams half of B).  HBM-bound, no MFMA.  Also finalises V.
// Reference: calculate_J_screened_GPU (GPUDF.jl:544-547) / DenseGPUDF.jl:103.
// ---------------------------------------------------------------------------
// Register budget: beside the K kernel (two 216-VGPR waves per SIMD) 80 of a SIMD's 512 registers are free, so this kernel
// must stay <= 80 VGPRs to run WHILE K runs (J beside K) and wants as many loads in flight as that allows:
// 4 rows x 3 aux indices = 12 loads of 16 B, 71 VGPRs.  (8 rows x 2: 86 VGPRs, not resident beside K; 8 x 1: 58 VGPRs,
// resident but latency-starved; 8 x 2 forced to 80: spills.)
#ifndef JCDF_J_ROWS
#define JCDF_J_ROWS 4
#endif
constexpr int J_ROWS = JCDF_J_ROWS;      // q rows per workgroup: J_ROWS rows x Np doubles contiguous per aux index
#ifndef JCDF_J_QUNROLL
#define JCDF_J_QUNROLL 3
#endif
constexpr int J_QUNROLL = JCDF_J_QUNROLL;   // aux indices in flight per thread (x J_ROWS loads of 16 B)

#ifndef JCDF_J_BLOCKS_PER_CU
#define JCDF_J_BLOCKS_PER_CU 6
#endif
__global__ __launch_bounds__(256, JCDF_J_BLOCKS_PER_CU) void k_coulomb_J(
    const double *__restrict__ B, const double *__restrict__ vpart, int nvp, int Ql, int Nk, int Np,
    int QS, double *__restrict__ Jpart, double *__restrict__ V, const unsigned long long *__restrict__ jmask)
{
    extern __shared__ __attribute__((aligned(16))) double Vs[];
    const int q0 = blockIdx.x * J_ROWS;
    const int s = blockIdx.y;
    const int Qb = s * QS;
    const int nQ = min(QS, Ql - Qb);
    for (int k = threadIdx.x; k < nQ; k += blockDim.x) {
        double v = 0.0;
        for (int t = 0; t < nvp; ++t) v += vpart[(int64_t)(Qb + k) * nvp + t];
        Vs[k] = v;
        if (blockIdx.x == 0) V[Qb + k] = v;
    }
    __syncthreads();

    // The slab stride (Nk*Np*8 B) is a large power of two for the common sizes, so a
    // thread that walked the aux index with one row per step would touch a new DRAM
    // page / TLB entry on every load (measured: 96 GB/s).  Instead each workgroup
    // consumes J_ROWS full rows (J_ROWS*Np*8 B contiguous) of one slab before it
    // moves to the next aux index.
    const int64_t slab2 = (int64_t)Nk * Np / 2;             // slab stride in double2 units
    const int np2 = Np / 2;
    const int qlast = q0 + J_ROWS - 1;
    // jmask[row block]: bit t set <=> the 128-column tile t has a kept pair in these rows (null: dense)
    const unsigned long long tmask = jmask ? jmask[blockIdx.x] : ~0ULL;
    for (int c2 = threadIdx.x; 2 * c2 <= qlast && c2 < np2; c2 += blockDim.x) {
        if (!((tmask >> (c2 >> 6)) & 1ULL)) continue;      // whole tile screened: its Jpart stays 0
        const double2_t *ptr = reinterpret_cast<const double2_t *>(B) + ((int64_t)Qb * Nk + q0) * np2 + c2;
        double2_t acc[J_ROWS];
#pragma unroll
        for (int r = 0; r < J_ROWS; ++r) acc[r] = double2_t{0.0, 0.0};
        int k = 0;
#pragma unroll 1
        for (; k + J_QUNROLL <= nQ; k += J_QUNROLL) {
            double2_t v[J_QUNROLL][J_ROWS];
#pragma unroll
            for (int u = 0; u < J_QUNROLL; ++u)
#pragma unroll
                for (int r = 0; r < J_ROWS; ++r)
                    v[u][r] = __builtin_nontemporal_load(ptr + (int64_t)(k + u) * slab2 + (int64_t)r * np2);
#pragma unroll
            for (int u = 0; u < J_QUNROLL; ++u) {
                const double vq = Vs[k + u];
#pragma unroll
                for (int r = 0; r < J_ROWS; ++r) {
                    acc[r].x += vq * v[u][r].x;
                    acc[r].y += vq * v[u][r].y;
                }
            }
        }
        for (; k < nQ; ++k) {
            const double vq = Vs[k];
#pragma unroll
            for (int r = 0; r < J_ROWS; ++r) {
                const double2_t v = __builtin_nontemporal_load(ptr + (int64_t)k * slab2 + (int64_t)r * np2);
                acc[r].x += vq * v.x;
                acc[r].y += vq * v.y;
            }
        }
        double2_t *out = reinterpret_cast<double2_t *>(Jpart) + ((int64_t)s * Nk + q0) * np2 + c2;
#pragma unroll
        for (int r = 0; r < J_ROWS; ++r) out[(int64_t)r * np2] = acc[r];
    }
}

// ---------------------------------------------------------------------------
// k_exchange_K: Kslab[s][t] = sum_{k in slice s} W[k][ti-tile]^T W[k][tj-tile]
// for lower block-triangle tiles t = (ti >= tj).  SYRK with a huge contraction
// (Ql*o) and a tiny output -> split-K over slices, deterministic slab reduce in
// k_fock_assemble.  Reference: calcululate_K_no_sym_GPU! / lower-triangle block
// GEMMs (GPUDF.jl:669-672, 758-826) / DenseGPUDF.jl:111.
// ---------------------------------------------------------------------------
using KCfg = GemmCfg<4, 2, 2, 4, KC>;    // 128 x 128 tile, 8 waves of 64 x 32
using KCfg4 = GemmCfg<4, 4, 2, 2, KC>;   // 128 x 128 tile, 4 waves of 64 x 64 (twice the MFMAs per barrier and per LDS read)

template <class Cfg>
__global__ __launch_bounds__(Cfg::NT, (Cfg::NT == 256) ? 2 : 1) void k_exchange_K(
    const double *__restrict__ W, int Np, int ntri, int S, int KS, double *__restrict__ Kslab)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    // all tiles of one k-slice on one XCD: they re-read the same W rows through that L2
    const int b = blockIdx.x;
    const int xcd = b & 7, r = b >> 3;
    const int t = r % ntri;
    const int s = (r / ntri) * 8 + xcd;
    if (s >= S) return;
    int ti = 0;
    while ((ti + 1) * (ti + 2) / 2 <= t) ++ti;
    const int tj = t - ti * (ti + 1) / 2;

    double4_t acc[Cfg::WM][Cfg::WN];
#pragma unroll
    for (int m = 0; m < Cfg::WM; ++m)
#pragma unroll
        for (int n = 0; n < Cfg::WN; ++n) acc[m][n] = double4_t{0.0, 0.0, 0.0, 0.0};

    const double *base = W + (int64_t)s * KS * Np;
    gemm_tn_core<Cfg, false>(base + ti * Cfg::TM, Np, base + tj * Cfg::TN, Np, KS / KC, acc, smem);

    double *out = Kslab + ((int64_t)s * ntri + t) * (Cfg::TM * Cfg::TN);
#pragma unroll
    for (int m = 0; m < Cfg::WM; ++m)
#pragma unroll
        for (int n = 0; n < Cfg::WN; ++n)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                out[tile_row<Cfg>(m, j) * Cfg::TN + tile_col<Cfg>(n)] = acc[m][n][j];
}

// ---------------------------------------------------------------------------
// k_fock_assemble: F = 2 J - K (+ H), symmetric, written in the reference's
// N x N column-major layout.  Sums the J slices and K slabs in fixed order.
// Replaces copy_screened_J_to_fock_upper_triangle + copy_upper_to_lower_kernel
// + axpy!(H) (GPUDF.jl:482-536, 221-225).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_fock_assemble(
    const double *__restrict__ Jpart, int SJ, const double *__restrict__ Kslab, int S, int ntri,
    const double *__restrict__ H, int N, int Nk, int Np, double *__restrict__ F)
{
    const int q = blockIdx.y;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p > q || q >= N) return;
    double j = 0.0;
    for (int s = 0; s < SJ; ++s) j += Jpart[((int64_t)s * Nk + q) * Np + p];
    const int ti = q >> 7, tj = p >> 7;
    const int t = ti * (ti + 1) / 2 + tj;
    const int64_t off = (int64_t)t * (128 * 128) + (q & 127) * 128 + (p & 127);
    double k = 0.0;
    for (int s = 0; s < S; ++s) k += Kslab[(int64_t)s * ntri * (128 * 128) + off];
    const double v = 2.0 * j - k;
    const int64_t lo = q + (int64_t)N * p, up = p + (int64_t)N * q;
    F[lo] = v + (H ? H[lo] : 0.0);
    if (p != q) F[up] = v + (H ? H[up] : 0.0);
}

// ---------------------------------------------------------------------------
// setup kernels
// ---------------------------------------------------------------------------
// raw (R x P, column-major, reference layout ThreeCenterIntegralsScreened.jl:25)
//   -> dst[a][p_c][q_c]   (row = outer index p, column = inner index q: by the
//   (q,p) symmetry this equals dst[a][q][p] and makes the writes contiguous
//   along the packed index).  32 x 32 LDS transpose tile.
__global__ __launch_bounds__(256) void k_scatter_T(
    const double *__restrict__ raw, int64_t R, int64_t P, const int64_t *__restrict__ pq_p,
    const int64_t *__restrict__ pq_q, int N, int Nk, int Np, double *__restrict__ dst)
{
    __shared__ double tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;     // 32 x 8
    const int64_t a0 = (int64_t)blockIdx.x * 32, c0 = (int64_t)blockIdx.y * 32;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int64_t a = a0 + tx, c = c0 + ty + 8 * k;
        tile[ty + 8 * k][tx] = (a < R && c < P) ? raw[a + R * c] : 0.0;
    }
    __syncthreads();
    const int64_t c = c0 + tx;
    if (c >= P) return;
    int64_t pp, qq;
    if (pq_p) { pp = pq_p[c]; qq = pq_q[c]; } else { pp = c / N; qq = c % N; }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int64_t a = a0 + ty + 8 * k;
        if (a < R) dst[(a * Nk + pp) * Np + qq] = tile[tx][ty + 8 * k];
    }
}

// inverse of k_scatter_T (jcdf_get_B)
__global__ __launch_bounds__(256) void k_gather_T(
    const double *__restrict__ src, int64_t R, int64_t P, const int64_t *__restrict__ pq_p,
    const int64_t *__restrict__ pq_q, int N, int Nk, int Np, double *__restrict__ raw)
{
    __shared__ double tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int64_t a0 = (int64_t)blockIdx.x * 32, c0 = (int64_t)blockIdx.y * 32;
    const int64_t c = c0 + tx;
    int64_t pp = 0, qq = 0;
    if (c < P) { if (pq_p) { pp = pq_p[c]; qq = pq_q[c]; } else { pp = c / N; qq = c % N; } }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int64_t a = a0 + ty + 8 * k;
        tile[tx][ty + 8 * k] = (a < R && c < P) ? src[(a * Nk + pp) * Np + qq] : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int64_t a = a0 + tx, cc = c0 + ty + 8 * k;
        if (a < R && cc < P) raw[a + R * cc] = tile[ty + 8 * k][tx];
    }
}

// k_metric_apply: Bout[r][x] += sum_s LinvT[s][r] T[s][x]   ("(B|Q)^-1 metric solve")
// r < M rows of this shard, x over the flattened Nk*Np slab.  Replaces
// CUBLAS.trmm!/gemm! of GPUDF.jl:907,939-943 and DenseGPUDF.jl:210,270.
using MCfg = GemmCfg<4, 2, 2, 4, KC>;

__global__ __launch_bounds__(512) void k_metric_apply(
    const double *__restrict__ LinvT, int64_t ldl, const double *__restrict__ T, int64_t slab,
    int Kpad, int M, int n_xtiles, int mt0, double *__restrict__ Bout)
{
    using Cfg = MCfg;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int xt = blockIdx.x % n_xtiles;
    const int mt = mt0 + blockIdx.x / n_xtiles;           // row tiles above the diagonal of L^-1 are skipped

    double4_t acc[Cfg::WM][Cfg::WN];
#pragma unroll
    for (int m = 0; m < Cfg::WM; ++m)
#pragma unroll
        for (int n = 0; n < Cfg::WN; ++n) acc[m][n] = double4_t{0.0, 0.0, 0.0, 0.0};

    gemm_tn_core<Cfg, true>(LinvT + mt * Cfg::TM, ldl, T + (int64_t)xt * Cfg::TN, slab, Kpad / KC,
                            acc, smem);
#pragma unroll
    for (int m = 0; m < Cfg::WM; ++m)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = mt * Cfg::TM + tile_row<Cfg>(m, j);
            if (r < M) {
#pragma unroll
                for (int n = 0; n < Cfg::WN; ++n) {
                    double *dst = Bout + (int64_t)r * slab + (int64_t)xt * Cfg::TN + tile_col<Cfg>(n);
                    *dst += acc[m][n][j];
                }
            }
        }
}

// W (internal [(Q,i)][p]) -> reference GPU layout (Ql, o, N) column-major (GPUDF.jl:140)
__global__ void k_export_W(const double *__restrict__ W, int Ql, int o, int N, int Np,
                           double *__restrict__ out)
{
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = (int64_t)Ql * o * N;
    if (idx >= total) return;
    const int Q = (int)(idx % Ql);
    const int i = (int)((idx / Ql) % o);
    const int p = (int)(idx / ((int64_t)Ql * o));
    out[idx] = W[((int64_t)Q * o + i) * Np + p];
}

}  // namespace jcdf
