// jcdf_gemm.hpp — fp64 MFMA GEMM cores for gfx950 (MI355X).
//
// "TN" core (first half of this file):   acc[m][n] += sum_k A[k][m] * B[k][n]
// Both operands are stored k-major (row k contiguous in m resp. n):
//   W pass (register-staged form)   : A = C_occ [q][i]        B = Bp [(q,p)][Q]   (rows gathered per stage)
//   Cholesky / D&C / SP2 helpers    : see jcdf_chol.hpp, jcdf_dc.hpp, jcdf_sp2.hpp
// "NT" core (second half): both operands k-contiguous — the K pass and the metric apply.
//
// Hardware mapping (CDNA4): v_mfma_f64_16x16x4_f64, one f64 of A and of B per
// lane: A[row = lane&15][k = lane>>4], B[k = lane>>4][col = lane&15]; the
// result has col = lane&15, row = (lane>>4) + 4*reg  (NOT the f32 C/D map).
// A workgroup owns a TM x TN output tile, TM = 16*WM*WAVES_M, TN = 16*WN*WAVES_N;
// each 64-lane wave owns WM x WN MFMA tiles.  k is consumed in chunks of KC rows
// staged through LDS (double buffered, one barrier per chunk).  LDS rows are
// padded so that (row stride in bytes) % 256 == 128: the two k-rows a
// ds_read_b64 half-wave touches then fall on disjoint bank halves
// (conflict-free; MI355X_MICROARCH §LDS).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace jcdf {

typedef double double4_t __attribute__((ext_vector_type(4)));
typedef double double2_t __attribute__((ext_vector_type(2)));

template <int WM_, int WN_, int WAVES_M_, int WAVES_N_, int KC_>
struct GemmCfg {
    static constexpr int WM = WM_, WN = WN_, WAVES_M = WAVES_M_, WAVES_N = WAVES_N_, KC = KC_;
    static constexpr int TM_MFMA = 16 * WM * WAVES_M;
    static constexpr int TM = TM_MFMA;                            // staged A width
    static constexpr int TN = 16 * WN * WAVES_N;
    static constexpr int NT = 64 * WAVES_M * WAVES_N;
    static constexpr int LDAS = ((TM + 15) / 32) * 32 + 16;       // doubles; >= TM and stride*8 % 256 == 128
    static constexpr int LDBS = TN + ((TN % 32 == 16) ? 0 : 16);
    static constexpr int STAGE_DOUBLES = KC * (LDAS + LDBS);
    static constexpr int SMEM_BYTES = 2 * STAGE_DOUBLES * 8;
    static constexpr int A_VEC = KC * TM / 2;                        // double2 per A stage
    static constexpr int B_VEC = KC * TN / 2;
    static constexpr int A_PER_THREAD = (A_VEC + NT - 1) / NT;
    static constexpr int B_PER_THREAD = (B_VEC + NT - 1) / NT;
    static_assert(KC % 4 == 0, "KC must be a multiple of the MFMA k (4)");
};

// Ag -> A[k = 0][m0], Bg -> B[k = 0][n0]; lda/ldb in doubles (even, 16-B aligned rows).
// nchunks = K / KC.  All tile loads must be in bounds (buffers are padded).
// STREAM_B: the B operand is read exactly once from HBM by the whole grid (the
// 3-index tensor) -> non-temporal loads keep it from evicting the reused operand.
// ABL: timing-only ablation bits (0 in the product; round 1's tools/w_ablate.hip, profiles/r01_w_ablate.txt):
//   1 = no A global loads after the first stage, 2 = no B global loads after the first
//   stage, 4 = no LDS re-staging (ds_write) after the first stage, 8 = no barrier in the loop.
// PREFETCH = 2: two register sets, every global load has two compute phases to land (for the
// HBM-streaming W kernel); PREFETCH = 1: one set (operands served from L2, fewer VGPRs).
template <class Cfg, bool STREAM_B, int ABL = 0, int PREFETCH = 1>
__device__ __forceinline__ void gemm_tn_core(const double *__restrict__ Ag, int64_t lda,
                                             const double *__restrict__ Bg, int64_t ldb,
                                             int nchunks, double4_t (&acc)[Cfg::WM][Cfg::WN],
                                             double *smem, const int *__restrict__ klist = nullptr)
{
    // klist (optional): stage t covers k rows [klist[t]*KC, +KC) instead of [t*KC, +KC) — block-sparse
    // contraction: all-zero stages (fully Schwarz-screened tiles) are simply not in the list.
    constexpr int WM = Cfg::WM, WN = Cfg::WN, KC = Cfg::KC;
    constexpr int TM = Cfg::TM, TN = Cfg::TN, NT = Cfg::NT;
    constexpr int LDAS = Cfg::LDAS, LDBS = Cfg::LDBS;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / Cfg::WAVES_N;
    const int wn = wave % Cfg::WAVES_N;
    const int lr = lane & 15;   // row (A) / col (B) inside the MFMA tile
    const int lk = lane >> 4;   // k inside the MFMA step

    // Two register sets: the loads of stage t+2 are issued before the MFMAs of stage t and are
    // written to LDS after the MFMAs of stage t+1, i.e. every global load has two full compute
    // phases to land.  (With one set the W kernel was latency-bound: replacing 1/6 of its MFMAs
    // by nothing did not make it faster; HBM latency under this load is ~3 us.)
    double2_t ra0[Cfg::A_PER_THREAD], rb0[Cfg::B_PER_THREAD];
    double2_t ra1[Cfg::A_PER_THREAD], rb1[Cfg::B_PER_THREAD];

    auto load_stage = [&](double2_t (&ra)[Cfg::A_PER_THREAD], double2_t (&rb)[Cfg::B_PER_THREAD], int chunk) {
        const int kc = klist ? klist[chunk] : chunk;
        const double *Ap = Ag + (int64_t)kc * KC * lda;
        const double *Bp = Bg + (int64_t)kc * KC * ldb;
        if (!((ABL & 1) && chunk > 0))
#pragma unroll
        for (int i = 0; i < Cfg::A_PER_THREAD; ++i) {
            const int idx = tid + i * NT;
            if (Cfg::A_VEC % NT == 0 || idx < Cfg::A_VEC) {
                const int r = idx / (TM / 2), c = idx % (TM / 2);
                ra[i] = *reinterpret_cast<const double2_t *>(Ap + (int64_t)r * lda + 2 * c);
            }
        }
        if (!((ABL & 2) && chunk > 0))
#pragma unroll
        for (int i = 0; i < Cfg::B_PER_THREAD; ++i) {
            const int idx = tid + i * NT;
            if (Cfg::B_VEC % NT == 0 || idx < Cfg::B_VEC) {
                const int r = idx / (TN / 2), c = idx % (TN / 2);
                const double2_t *src = reinterpret_cast<const double2_t *>(Bp + (int64_t)r * ldb + 2 * c);
                rb[i] = STREAM_B ? __builtin_nontemporal_load(src) : *src;
            }
        }
    };
    auto store_stage = [&](const double2_t (&ra)[Cfg::A_PER_THREAD], const double2_t (&rb)[Cfg::B_PER_THREAD], int buf) {
        double *As = smem + buf * Cfg::STAGE_DOUBLES;
        double *Bs = As + KC * LDAS;
#pragma unroll
        for (int i = 0; i < Cfg::A_PER_THREAD; ++i) {
            const int idx = tid + i * NT;
            if (Cfg::A_VEC % NT == 0 || idx < Cfg::A_VEC) {
                const int r = idx / (TM / 2), c = idx % (TM / 2);
                *reinterpret_cast<double2_t *>(As + r * LDAS + 2 * c) = ra[i];
            }
        }
#pragma unroll
        for (int i = 0; i < Cfg::B_PER_THREAD; ++i) {
            const int idx = tid + i * NT;
            if (Cfg::B_VEC % NT == 0 || idx < Cfg::B_VEC) {
                const int r = idx / (TN / 2), c = idx % (TN / 2);
                *reinterpret_cast<double2_t *>(Bs + r * LDBS + 2 * c) = rb[i];
            }
        }
    };
    auto compute_stage = [&](int buf) {
        const double *As = smem + buf * Cfg::STAGE_DOUBLES + wm * (WM * 16) + lr;
        const double *Bs = smem + buf * Cfg::STAGE_DOUBLES + KC * LDAS + wn * (WN * 16) + lr;
#pragma unroll
        for (int ks = 0; ks < KC / 4; ++ks) {
            double a[WM], b[WN];
#pragma unroll
            for (int m = 0; m < WM; ++m) a[m] = As[(ks * 4 + lk) * LDAS + m * 16];
#pragma unroll
            for (int n = 0; n < WN; ++n) b[n] = Bs[(ks * 4 + lk) * LDBS + n * 16];
#pragma unroll
            for (int m = 0; m < WM; ++m)
#pragma unroll
                for (int n = 0; n < WN; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m], b[n], acc[m][n], 0, 0, 0);
        }
    };

    load_stage(ra0, rb0, 0);
    store_stage(ra0, rb0, 0);
    if (PREFETCH == 2) {
        if (nchunks > 1) load_stage(ra1, rb1, 1);
        __syncthreads();
        for (int t = 0; t < nchunks; t += 2) {
            // even stage t: LDS buffer 0; set 1 holds stage t+1 (in flight); set 0 is free
            if (t + 2 < nchunks) load_stage(ra0, rb0, t + 2);
            compute_stage(0);
            if (t + 1 < nchunks && !(ABL & 4)) store_stage(ra1, rb1, 1);   // buffer 1: last read before the previous barrier
            if (!(ABL & 8)) __syncthreads();
            if (t + 1 >= nchunks) break;
            // odd stage t+1: LDS buffer 1; set 0 holds stage t+2 (in flight); set 1 is free
            if (t + 3 < nchunks) load_stage(ra1, rb1, t + 3);
            compute_stage((ABL & 4) ? 0 : 1);
            if (t + 2 < nchunks && !(ABL & 4)) store_stage(ra0, rb0, 0);
            if (!(ABL & 8)) __syncthreads();
        }
    } else {
        __syncthreads();
        int cur = 0;
        for (int t = 0; t < nchunks; ++t) {
            const bool more = (t + 1 < nchunks);
            if (more) load_stage(ra0, rb0, t + 1);      // latency hides under this stage's MFMAs
            compute_stage(cur);
            if (more && !(ABL & 4)) store_stage(ra0, rb0, cur ^ 1);   // other buffer: last read before the previous barrier
            if (!(ABL & 8)) __syncthreads();
            if (!(ABL & 4)) cur ^= 1;
        }
    }
}

// ---------------------------------------------------------------------------
// "NT" variant: both operands are stored with the CONTRACTION index contiguous,
//
//   acc[m][n] += sum_k A[m][k] * B[n][k]
//
//   K pass  : A = Wb [p ][(i,Q)]     B = Wb   [p'][(i,Q)]   (SYRK over the blocked exchange intermediate)
//   metric  : A = T  [c ][s]         B = Linv [r ][s]       (B = L^-1 T in the reference's (Q_d, P) layout)
//
// A stage is TM (resp. TN) rows x KC = 16 contiguous doubles (one 128-B line per row); every thread moves 16 B
// along k.  The LDS image keeps the global orientation with UNPADDED 128-B rows and an XOR swizzle inside the row:
// double k of row r sits at position  (((k >> 1) ^ (r & 7)) << 1) | ((k & 1) ^ ((r >> 3) & 1)),  i.e. the 16-B chunk
// index is XOR-ed with the low row bits and the two halves of a chunk swap for rows 8..15 of every 16.  The 16 lanes of
// an MFMA operand read (16 rows, one k) then cover all banks in the 32-bank mode of ds_read2_b64 and in the 64-bank
// mode of ds_read_b64; the stores stay whole 16-B chunks (ds_write_b128).  (Rows padded to 18 doubles instead: 41 % of
// the K kernel's LDS cycles were bank conflicts, rocprofv3 SQ_LDS_BANK_CONFLICT 4.6e7 of SQ_LDS_IDX_ACTIVE 1.1e8 — rows
// r and r + 8 share their banks when the compiler pairs the reads into ds_read2_b64.)
template <class Cfg>
struct GemmNT {
    static constexpr int KC = Cfg::KC, TM = Cfg::TM, TN = Cfg::TN, NT = Cfg::NT;
    static_assert(KC == 16, "the swizzle is written for 16-double rows");
    static constexpr int LDK = KC;
    static constexpr int STAGE_DOUBLES = (TM + TN) * LDK;
    static constexpr int SMEM_BYTES = 2 * STAGE_DOUBLES * 8;
    static constexpr int A_VEC = TM * KC / 2, B_VEC = TN * KC / 2;
    static constexpr int A_PER_THREAD = (A_VEC + NT - 1) / NT, B_PER_THREAD = (B_VEC + NT - 1) / NT;
    static_assert(A_VEC % NT == 0 && B_VEC % NT == 0, "tile does not divide over the threads");
};

// Ag -> A[m0][k0], Bg -> B[n0][k0]; lda/ldb = row strides in doubles (even); nchunks = K / KC.
// STREAM_A: the A operand is read once from HBM by the whole grid (non-temporal loads).
// same: the A and B tiles are the same rows (diagonal tile of a SYRK): B is neither loaded nor staged.
// BLOCKED: the operands are stored in blocks of (tile rows x KC) (chunk t of a tile is one contiguous block, lda/ldb = KC):
// consecutive chunks are TM*KC (resp. TN*KC) doubles apart.
// mfma_on = false: this wave issues no MFMA (its accumulators are not needed: upper block of a diagonal SYRK tile);
// it still loads, stages and meets every barrier.
template <class Cfg, bool STREAM_A = false, bool BLOCKED = false>
__device__ __forceinline__ void gemm_nt_core(const double *__restrict__ Ag, int64_t lda,
                                             const double *__restrict__ Bg, int64_t ldb, int nchunks,
                                             double4_t (&acc)[Cfg::WM][Cfg::WN], double *smem, bool same = false,
                                             bool mfma_on = true)
{
    using G = GemmNT<Cfg>;
    constexpr int WM = Cfg::WM, WN = Cfg::WN, KC = Cfg::KC, NT = Cfg::NT, LDK = G::LDK;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / Cfg::WAVES_N, wn = wave % Cfg::WAVES_N;
    const int lr = lane & 15, lk = lane >> 4;

    double2_t ra[G::A_PER_THREAD], rb[G::B_PER_THREAD];
    auto load_stage = [&](int chunk) {
#pragma unroll
        for (int i = 0; i < G::A_PER_THREAD; ++i) {
            const int idx = tid + i * NT, r = idx / (KC / 2), c = idx % (KC / 2);
            const double2_t *src = reinterpret_cast<const double2_t *>(Ag + (int64_t)r * lda + (int64_t)chunk * (BLOCKED ? Cfg::TM * KC : KC) + 2 * c);
            ra[i] = STREAM_A ? __builtin_nontemporal_load(src) : *src;
        }
        if (!same)
#pragma unroll
            for (int i = 0; i < G::B_PER_THREAD; ++i) {
                const int idx = tid + i * NT, r = idx / (KC / 2), c = idx % (KC / 2);
                rb[i] = *reinterpret_cast<const double2_t *>(Bg + (int64_t)r * ldb + (int64_t)chunk * (BLOCKED ? Cfg::TN * KC : KC) + 2 * c);
            }
    };
    // chunk c of row r -> chunk c ^ (r & 7), halves swapped when bit 3 of r is set
    auto store_one = [&](double *base, int r, int c, double2_t v) {
        if (r & 8) v = double2_t{v.y, v.x};
        *reinterpret_cast<double2_t *>(base + r * LDK + 2 * (c ^ (r & 7))) = v;
    };
    auto store_stage = [&](int buf) {
        double *As = smem + buf * G::STAGE_DOUBLES;
        double *Bs = As + Cfg::TM * LDK;
#pragma unroll
        for (int i = 0; i < G::A_PER_THREAD; ++i) {
            const int idx = tid + i * NT;
            store_one(As, idx / (KC / 2), idx % (KC / 2), ra[i]);
        }
        if (!same)
#pragma unroll
            for (int i = 0; i < G::B_PER_THREAD; ++i) {
                const int idx = tid + i * NT;
                store_one(Bs, idx / (KC / 2), idx % (KC / 2), rb[i]);
            }
    };
    // position of k = 4 ks + lk in this lane's rows (all rows of a lane have the same low 4 bits, lr)
    int koff[KC / 4];
#pragma unroll
    for (int ks = 0; ks < KC / 4; ++ks) {
        const int k = 4 * ks + lk;
        koff[ks] = (((k >> 1) ^ (lr & 7)) << 1) | ((k & 1) ^ ((lr >> 3) & 1));
    }
    auto compute_stage = [&](int buf) {
        if (!mfma_on) return;
        const double *As = smem + buf * G::STAGE_DOUBLES + (wm * (WM * 16) + lr) * LDK;
        const double *Bs = smem + buf * G::STAGE_DOUBLES + (same ? 0 : Cfg::TM * LDK) + (wn * (WN * 16) + lr) * LDK;
#pragma unroll
        for (int ks = 0; ks < KC / 4; ++ks) {
            double a[WM], b[WN];
#pragma unroll
            for (int m = 0; m < WM; ++m) a[m] = As[m * 16 * LDK + koff[ks]];
#pragma unroll
            for (int n = 0; n < WN; ++n) b[n] = Bs[n * 16 * LDK + koff[ks]];
#pragma unroll
            for (int m = 0; m < WM; ++m)
#pragma unroll
                for (int n = 0; n < WN; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m], b[n], acc[m][n], 0, 0, 0);
        }
    };

    load_stage(0);
    store_stage(0);
    __syncthreads();
    int cur = 0;
    for (int t = 0; t < nchunks; ++t) {
        const bool more = (t + 1 < nchunks);
        if (more) load_stage(t + 1);
        compute_stage(cur);
        if (more) store_stage(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }
}

// Element coordinates of acc[m][n][j] inside the workgroup tile.
template <class Cfg>
__device__ __forceinline__ int tile_row(int m, int j)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    return (wave / Cfg::WAVES_N) * (Cfg::WM * 16) + m * 16 + (lane >> 4) + 4 * j;
}
template <class Cfg>
__device__ __forceinline__ int tile_col(int n)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    return (wave % Cfg::WAVES_N) * (Cfg::WN * 16) + n * 16 + (lane & 15);
}

}  // namespace jcdf
