// jcdf_kernels.hpp — device kernels of the MI355X DF-RHF Fock build.
//
// HBM layout (all fp64).  The 3-index tensor is kept in the REFERENCE's packed
// layout, device_B (Q_d, P) column-major (GPUDF.jl:111-155): one row per kept (Schwarz-unscreened)
// pair, the auxiliary index contiguous.  The dense map is the special case P = N*N.
//   Bp    [P + slack][ldq]    Bp[c][Q] = (L^-1 (Q|pq))[Q], c = packed pair index (outer p, inner q;
//                             SchwarzScreening.jl:72-81), ldq = roundup(Ql, 16); columns >= Ql are 0.
//   Cpad  [Np + 16][opad]     occupied MO coefficients C[q][i], i fastest; rows >= N are 0 (row N is the
//                             "zero row" the padding slots of a stage point at).
//   Cv    [N][n_mt][WVM][4][4 WM]   the same numbers permuted into the fp64 MFMA accumulator layout for
//                             the fused-V epilogue of the W kernel.
//   Wb    [Np/128][Wld/16][128][16]   exchange intermediate W[p][k], k = i*ldq + Q (the reference's W (Q_d, o, N),
//                             GPUDF.jl:140, with the orbital index moved outside the aux index), blocked so that
//                             the 128 p of a K-kernel tile x 16 consecutive k are one contiguous 16 KB block:
//                             element (p, k) at ((p/128 * Wld/16 + k/16) * 128 + p%128) * 16 + k%16.
//                             Wld = S*KS >= o*ldq; the k tail and the rows p >= N are 0.
//   vpart [n_chunks*n_mt][vld]  per-workgroup-chunk partial sums of V[Q] (deterministic order).
//   J     [Plow]              Coulomb matrix on the kept pairs with q >= p.
//   Kslab [S][ntri][128][128] split-K partial exchange tiles (lower block-triangle).
#pragma once
#include "jcdf_gemm.hpp"

namespace jcdf {

constexpr int KC = 16;          // k rows per LDS stage
constexpr int TILE_P = 128;     // p-tile of the K kernel == padding unit of Np
constexpr int TILE_Q = 128;     // aux-index tile of the W kernel
typedef __attribute__((address_space(3))) void lds_void_t;        // operands of __builtin_amdgcn_global_load_lds
typedef __attribute__((address_space(1))) const void glb_void_t;
typedef __attribute__((address_space(3))) const volatile double lds_cvdouble_t;   // an LDS read the compiler may not merge with its neighbour

// ---------------------------------------------------------------------------
// C_occ (N x o, column-major, reference layout DensityFitting.jl:49) -> Cpad, Cv.
// Replaces the per-p gather buffer of build_non_zero_coefficients_kernel
// (GPUDF.jl:459-480): the W kernel gathers the kept C rows itself.
//   Cv[(((p*n_mt + mt)*WVM + wm)*4 + g)*4WM + 4m + j] = C[p][i = (mt*WVM + wm)*16WM + 16m + 4j + g]
// i.e. what lane group g = lane>>4 of wave row wm multiplies its accumulators acc[m][.][j] with.
// ---------------------------------------------------------------------------
__global__ void k_prep_C(const double *__restrict__ C, int64_t ldc, int N, int o, int Np, int opad, int WM, int rows_per_p,
                         double *__restrict__ Cpad, double *__restrict__ Cv)
{
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)Np * opad) return;
    const int q = (int)(idx / opad), i = (int)(idx % opad);
    const double v = (q < N && i < o) ? C[q + ldc * i] : 0.0;
    Cpad[idx] = v;
    if (q >= N || i >= 16 * WM * rows_per_p) return;      // trailing orbitals of the VALU remainder path are not in Cv
    const int wrow = i / (16 * WM), il = i % (16 * WM);       // wrow = mt*WVM + wm < rows_per_p
    const int m = il >> 4, g = il & 3, j = (il & 15) >> 2;
    Cv[(((int64_t)q * rows_per_p + wrow) * 4 + g) * (4 * WM) + 4 * m + j] = v;
}

// ---------------------------------------------------------------------------
// k_exchange_W_dma: for every p, Wt[p][(i,Q)] = sum_{q kept with p} C[q][i] Bp[(q,p)][Q]   (one pass over B)
// fused with V[Q] = sum_{p,i} W[Q,i,p] C[p][i]  (== B_Q . D~, D~ = C C^T).
// Reference: calculate_W_screened_GPU (GPUDF.jl:637-667; one GEMM per p: (Q_d x K_p)(K_p x o), 2 Q P o flop) /
// DenseGPUDF.jl:107 (W) and GPUDF.jl:539-542 / DenseGPUDF.jl:99 (V gemv, which costs the reference one extra
// pass over B).  Same shape here: MFMA M = orbitals, N = 128 aux indices (contiguous in Bp), K = the K_p kept
// q of this p in stages of 8 - the work is 2 Q P o up to the rounding of K_p to 8.
//
// A workgroup owns one 128-wide aux tile and a CHUNK of consecutive p (host-balanced to ~equal stage counts) and
// streams through the chunk's stages without draining the pipeline at a p boundary; at the last stage of a p the
// accumulators are stored and reset.  Stage table (built once by jcdf_configure, the pattern does not depend on the
// aux index): stg_c[8 t + r] = packed row of slot r, stg_q[8 t + r] = its C row (N = the zero row for padding
// slots), stg_p[t] = p if t is the last stage of p, else -1.
// WVM = 1: 4 waves side by side along Q, tile (16 WM) x 128.  WVM = 2 (more than 128 occupied orbitals): 8 waves
// as 2 x 4, tile (32 WM) x 128 - the B stage feeds both orbital halves.
// (The register-staged predecessors k_exchange_W / k_exchange_K live in jcdf_kernels_diag.hpp: diagnostic builds only.)
//
// k_exchange_W_dma: the same contraction with LDS-DMA staging (global_load_lds_dwordx4: global -> LDS with no
// VGPR in between).  Stages of KCD = 8 slots in a ring of 4 LDS buffers; the DMA of stage t+3 is issued at the
// start of phase t, a counted `s_waitcnt vmcnt` at the end of phase t retires stage t+1 (stages t+2 and t+3 stay in
// flight across the raw s_barrier), and stage t+1 is read in phase t+1 — one phase after the wait that retires it.
// One wave instruction moves one 1 KB row piece: a B row (128 aux indices) or 128 orbitals of a C row; LDS rows are
// padded to 144 / 272 doubles (conflict-free ds_read_b64, no piece crosses a row).  No staging registers, no
// ds_write, no branch around a load; the stage table has 8-slot granularity (K_p rounded up to 8 instead of 16).
// ---------------------------------------------------------------------------
constexpr int KCD = 8;

#ifndef JCDF_W_DMA_WAVES3
#define JCDF_W_DMA_WAVES3 1
#endif
// WN = 16-column MFMA tiles per wave along the aux index: 2 (aux tile 128) or 4 (aux tile 256: every staged C row
// feeds twice as many MFMAs — up to 96 orbitals the kernel is bound by the bytes it moves into LDS, half of which
// are C rows, not by MFMA issue: skipping 5 % of the MFMAs did not change its time).
// REM = 1..3 trailing orbitals (n_occ mod 16) that do not fill an MFMA row tile are contracted by VALU FMAs from the
// same LDS stage instead of being padded to 16 MFMA rows: C20H42 has 81 occupied orbitals = 5 MFMA tiles + 1 (a 6th
// tile would be 15/16 padding, a sixth of the kernel's MFMA work).
template <int WM, int WVM, int WN, int REM = 0>
struct WDmaCfg {
    static constexpr int TM = 16 * WM * WVM, TN = 64 * WN;          // MFMA rows
    static constexpr int TMA = TM + (REM ? 16 : 0);                 // staged orbitals of a C row
    static_assert(REM == 0 || (WVM == 1 && WN == 2 && TMA <= 128), "VALU remainder: 4-wave form only");
    static constexpr int AH = (TMA <= 128) ? 1 : 2;                 // 1 KB pieces per C row
    static constexpr int BH = TN / 128;                             // 1 KB pieces per B row
    // LDS row strides (doubles): >= the row, and = 16 mod 32 (conflict-free ds_read_b64 across the 4 k rows of an MFMA step)
    static constexpr int LDAS = ((TMA + 15) / 32) * 32 + 16, LDBS = TN + 16;
    // up to 96 orbitals the 128-wide form fits 3 workgroups per CU (<= 168 VGPRs, 3 x 48 KB of LDS with a ring of 3)
    static constexpr bool THREE = JCDF_W_DMA_WAVES3 && WVM == 1 && WM <= 6 && WN == 2;
    static constexpr int RING = (THREE || WN == 4) ? 3 : 4;
    static constexpr int WAVES_PER_SIMD = THREE ? 3 : 2;
    static constexpr int STAGE_DOUBLES = KCD * (LDAS + LDBS);
    static constexpr int SMEM_BYTES = (RING * STAGE_DOUBLES + (REM ? 128 : 0)) * 8;
    static constexpr int NW = 4 * WVM, NT = 64 * NW;
    static constexpr int A_PER = KCD * AH / NW, B_PER = KCD * BH / NW;   // pieces per wave and stage
    static constexpr int G = A_PER + B_PER;
    static constexpr int STORES = WM * 4 * WN;                      // W stores per wave in an epilogue
    static_assert((KCD * AH) % NW == 0 && (KCD * BH) % NW == 0, "stage rows do not divide over the waves");
};

template <int N> __device__ __forceinline__ void wait_vmcnt()
{
    static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit field");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// ABL: timing-only ablation bits (tools/prof_fock.py with JCDF_W_ABLATE, C20H42-shaped <6,1,2> only; results are wrong
// with them; they are template bits because a run-time flag around the DMA issue or the wait de-pipelines the loop:
// the same kernel took 5.05 instead of 1.70 ms): 2 = no DMA after the prologue, 4 = no wait / barrier in the loop,
// 8 = no W stores, 16 = no MFMA.
template <int WM, int WVM, int WN, int ABL = 0, int REM = 0>
__global__ __launch_bounds__(256 * WVM, (WDmaCfg<WM, WVM, WN, REM>::WAVES_PER_SIMD)) void k_exchange_W_dma(
    const double *__restrict__ Bp, int64_t ldq, const double *__restrict__ Cpad, const double *__restrict__ Cv,
    double *__restrict__ Wt, int64_t Wld, double *__restrict__ vpart, int vld, int o, int opad, int n_mt, int n_qt,
    const int *__restrict__ wchunk, const int *__restrict__ stg_c, const int *__restrict__ stg_q,
    const int *__restrict__ stg_p, int skip_partial, unsigned long long *__restrict__ stall)
{
    using D = WDmaCfg<WM, WVM, WN, REM>;
    constexpr int TM = D::TM, TMA = D::TMA, TN = D::TN, LDAS = D::LDAS, LDBS = D::LDBS, NW = D::NW, AH = D::AH, BH = D::BH;
    constexpr int A_PER = D::A_PER, B_PER = D::B_PER, G = D::G, RING = D::RING;
    extern __shared__ __attribute__((aligned(16))) double smem[];

    const int b = blockIdx.x;
    const int qt = b % n_qt, mt = (b / n_qt) % n_mt, chunk = b / (n_qt * n_mt);
    const int t0 = wchunk[chunk], nst = wchunk[chunk + 1] - t0;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / 4, wn = wave % 4;
    const int lr = lane & 15, lk = lane >> 4;

    double4_t acc[WM][WN];
#pragma unroll
    for (int m = 0; m < WM; ++m)
#pragma unroll
        for (int n = 0; n < WN; ++n) acc[m][n] = double4_t{0.0, 0.0, 0.0, 0.0};
    double vacc[WN];
#pragma unroll
    for (int n = 0; n < WN; ++n) vacc[n] = 0.0;

    // a wave whose aux columns all lie past the end of the rows (last, partial aux tile) issues no MFMA and no
    // store; it still moves its DMA pieces and meets every barrier
    const bool active = !skip_partial || qt * TN + wn * (16 * WN) < ldq;
    // VALU remainder: thread tid < 128 owns aux column tid of the tile for the REM trailing orbitals
    double racc[REM ? REM : 1], rv = 0.0;
#pragma unroll
    for (int r = 0; r < (REM ? REM : 1); ++r) racc[r] = 0.0;
    const double *Ag = Cpad + mt * TM + 2 * lane;
    const double *Bg = Bp + (int64_t)qt * TN + 2 * lane;

    int iq[A_PER], ic[B_PER];                        // wave-uniform gather indices of the next stage to issue
    auto next_idx = [&](int t) {
        const int ts = min(t, nst - 1);              // past the chunk's end the last stage is issued again (nobody reads it)
        const int *sq = stg_q + (int64_t)(t0 + ts) * KCD, *sc = stg_c + (int64_t)(t0 + ts) * KCD;
#pragma unroll
        for (int i = 0; i < A_PER; ++i) iq[i] = sq[(wave + i * NW) / AH];
#pragma unroll
        for (int i = 0; i < B_PER; ++i) ic[i] = sc[(wave + i * NW) / BH];
    };
    auto issue = [&](int buf) {
        double *As = smem + buf * D::STAGE_DOUBLES;
        double *Bs = As + KCD * LDAS;
#pragma unroll
        for (int i = 0; i < A_PER; ++i) {
            const int s = wave + i * NW, row = s / AH, half = s % AH;
            // lanes past the row's orbitals are masked off: their 16 bytes would land in the next LDS row
            if (TMA % 128 == 0 || half * 64 + lane < TMA / 2)
                __builtin_amdgcn_global_load_lds((glb_void_t *)(Ag + (int64_t)iq[i] * opad + half * 128),
                                                 (lds_void_t *)(As + row * LDAS + half * 128), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < B_PER; ++i) {
            const int s = wave + i * NW, row = s / BH, half = s % BH;
            __builtin_amdgcn_global_load_lds((glb_void_t *)(Bg + (int64_t)ic[i] * ldq + half * 128),
                                             (lds_void_t *)(Bs + row * LDBS + half * 128), 16, 0, 2);   // nt: read once
        }
    };
    auto compute_stage = [&](int buf) {
        if (!active) return;
        const double *As = smem + buf * D::STAGE_DOUBLES + wm * (WM * 16) + lr;
        const double *Bs = smem + buf * D::STAGE_DOUBLES + KCD * LDAS + wn * (16 * WN) + lr;
#pragma unroll
        for (int ks = 0; ks < KCD / 4; ++ks) {
            double a[WM], bb[WN];
#pragma unroll
            for (int m = 0; m < WM; ++m) a[m] = As[(ks * 4 + lk) * LDAS + m * 16];
#pragma unroll
            for (int n = 0; n < WN; ++n) bb[n] = Bs[(ks * 4 + lk) * LDBS + n * 16];
#pragma unroll
            for (int m = 0; m < WM; ++m)
#pragma unroll
                for (int n = 0; n < WN; ++n) {
                    if constexpr (ABL & 16) {       // keep the LDS reads alive without the matrix pipe
                        asm volatile("" ::"v"(a[m]), "v"(bb[n]));
                    } else {
                        acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m], bb[n], acc[m][n], 0, 0, 0);
                    }
                }
        }
    };
    auto compute_rem = [&](int buf) {               // waves 0 and 1: one aux column per thread, all KCD rows of the stage
        if constexpr (REM > 0) {
            if (wave < 2) {
                const double *Ar = smem + buf * D::STAGE_DOUBLES + TM;
                const double *Br = smem + buf * D::STAGE_DOUBLES + KCD * LDAS + tid;
#pragma unroll
                for (int k = 0; k < KCD; ++k) {
                    const double bk = Br[k * LDBS];
#pragma unroll
                    for (int r = 0; r < REM; ++r) racc[r] += Ar[k * LDAS + r] * bk;
                }
            }
        }
    };
    const int qcol = qt * TN + wn * (16 * WN) + lr;         // aux column of acc[.][0]; acc[.][n] is 16 n further
    const int ibase = (mt * WVM + wm) * (WM * 16) + lk;     // orbital of acc[0][.][0]
    auto epilogue = [&](int p) {
        // V: C[p][i] in accumulator layout, 2 WM loads of 16 B, the same address in all 16 lanes of a group; taken two m
        // tiles at a time (the scheduler would otherwise hold all 4 WM values in registers next to the accumulators)
        const double2_t *cv = reinterpret_cast<const double2_t *>(Cv) +
                              ((((int64_t)p * n_mt + mt) * WVM + wm) * 4 + lk) * (2 * WM);
#pragma unroll
        for (int m0 = 0; m0 < WM; m0 += 2) {
#pragma unroll
            for (int m = m0; m < m0 + 2 && m < WM; ++m) {
                const double2_t c01 = cv[2 * m], c23 = cv[2 * m + 1];
#pragma unroll
                for (int n = 0; n < WN; ++n)
                    vacc[n] += (acc[m][n][0] * c01.x + acc[m][n][1] * c01.y) + (acc[m][n][2] * c23.x + acc[m][n][3] * c23.y);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // (p, k = i*ldq + Q) -> block (p/128, k/16), row p%128: the 16 lanes of a group write one 128-B row of a block;
        // one orbital further = ldq/16 blocks further, and this lane's orbitals are ibase + 4 (4m + j)
        const int64_t istep4 = (ldq >> 4) * (4 * 2048);
        double *wp = Wt + ((int64_t)(p >> 7) * (Wld >> 4) * 128 + (p & 127)) * 16 + (int64_t)(qcol >> 4) * 2048 + lr +
                     (int64_t)ibase * (istep4 >> 2);
#pragma unroll
        for (int m = 0; m < WM; ++m)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (ibase + m * 16 + 4 * j < o && !(ABL & 8)) {
#pragma unroll
                    for (int n = 0; n < WN; ++n)
                        if (qcol + 16 * n < ldq) wp[2048 * n] = acc[m][n][j];
                }
                wp += istep4;
            }
#pragma unroll
        for (int m = 0; m < WM; ++m)
#pragma unroll
            for (int n = 0; n < WN; ++n) acc[m][n] = double4_t{0.0, 0.0, 0.0, 0.0};
    };

    auto epilogue_rem = [&](int p) {                 // trailing orbitals: W[p][(TM + r, Q)] and their share of V
        if constexpr (REM > 0) {
            if (wave < 2) {
                const int Qc = qt * TN + tid;
                const double *cp = Cpad + (int64_t)p * opad + mt * TM + TM;
#pragma unroll
                for (int r = 0; r < REM; ++r) {
                    rv += racc[r] * cp[r];
                    const int64_t k = (int64_t)(mt * TM + TM + r) * ldq + Qc;
                    if (Qc < ldq) Wt[(((int64_t)(p >> 7) * (Wld >> 4) + (k >> 4)) * 128 + (p & 127)) * 16 + (k & 15)] = racc[r];
                    racc[r] = 0.0;
                }
            }
        }
    };

    // prologue: stages 0 .. RING-2 in flight, stage 0 retired
    constexpr int AHEAD = RING - 1;                  // the DMA of stage t + AHEAD is issued at the start of phase t
    constexpr int FLY = (AHEAD - 1) * G;             // pieces that stay in flight across a barrier
    next_idx(0);
#pragma unroll
    for (int s0 = 0; s0 < AHEAD; ++s0) {
        issue(s0);
        next_idx(s0 + 1);
    }
    wait_vmcnt<FLY>();
    __builtin_amdgcn_s_barrier();
    int cur = 0, nxt = AHEAD;                        // ring slots of stage t and of stage t + AHEAD
    // ABL & 32: diagnostic build with s_memtime stamps around the five segments of a phase (shader cycles per wave, summed
    // over the chunk's phases; tools/w_stall.py, profiles/r02_w_stall.txt).  The stamps cost cycles themselves.
    unsigned long long seg[5] = {0, 0, 0, 0, 0}, tprev = 0;
    auto stamp = [&](int k) {
        if constexpr (ABL & 32) {
            const unsigned long long now = __builtin_amdgcn_s_memtime();
            seg[k] += now - tprev;
            tprev = now;
        }
    };
    if constexpr (ABL & 32) tprev = __builtin_amdgcn_s_memtime();
    for (int t = 0; t < nst; ++t) {
        if constexpr (!(ABL & 2)) issue(nxt);        // -> the buffer phase t-1 has finished reading
        stamp(0);                                    // DMA issue (address arithmetic + 2 G global_load_lds)
        compute_stage(cur);
        compute_rem(cur);
        stamp(1);                                    // LDS operand reads + MFMA issue (+ the VALU remainder)
        const int p = stg_p[t0 + t];
        next_idx(t + AHEAD + 1);
        if (p >= 0) epilogue_rem(p);
        if (p >= 0 && active) {
            epilogue(p);
            stamp(2);                                // scalar index loads, epilogue at the end of a p
            // behind stage t+1's pieces the counter now holds FLY pieces and this wave's W stores
            if constexpr (!(ABL & 4)) wait_vmcnt<(FLY + D::STORES > 63) ? 63 : FLY + D::STORES>();
        } else {
            stamp(2);
            if constexpr (!(ABL & 4)) wait_vmcnt<FLY>();
        }
        stamp(3);                                    // counted vmcnt wait (stage t+1's DMA)
        if constexpr (!(ABL & 4)) __builtin_amdgcn_s_barrier();
        stamp(4);                                    // barrier
        cur = (cur + 1 == RING) ? 0 : cur + 1;
        nxt = (nxt + 1 == RING) ? 0 : nxt + 1;
    }
    if constexpr (ABL & 32) {
        if (stall && lane == 0) {
            unsigned long long *o_ = stall + ((size_t)blockIdx.x * NW + wave) * 6;
#pragma unroll
            for (int k = 0; k < 5; ++k) o_[k] = seg[k];
            o_[5] = (unsigned long long)nst;
        }
    }
    wait_vmcnt<0>();                                 // the re-issued tail stages still write this workgroup's LDS

#pragma unroll
    for (int n = 0; n < WN; ++n) {
        vacc[n] += __shfl_xor(vacc[n], 16, 64);
        vacc[n] += __shfl_xor(vacc[n], 32, 64);
    }
    if (WVM == 2) {
        __builtin_amdgcn_s_barrier();                // every wave's DMA has landed (vmcnt(0) above) before the LDS is reused
        if (wm == 1 && lk == 0) {
#pragma unroll
            for (int n = 0; n < WN; ++n) smem[wn * (16 * WN) + 16 * n + lr] = vacc[n];
        }
        __syncthreads();
        if (wm == 0) {
#pragma unroll
            for (int n = 0; n < WN; ++n) vacc[n] += smem[wn * (16 * WN) + 16 * n + lr];
        }
    }
    if constexpr (REM > 0) {                         // the remainder's share of V, column by column, through the scratch row
        double *scr = smem + RING * D::STAGE_DOUBLES;
        __builtin_amdgcn_s_barrier();
        if (wave < 2) scr[tid] = rv;
        __syncthreads();
        if (lk == 0) {
#pragma unroll
            for (int n = 0; n < WN; ++n) vacc[n] += scr[wn * (16 * WN) + 16 * n + lr];
        }
    }
    if (wm == 0 && lk == 0) {
        double *vp = vpart + ((int64_t)chunk * n_mt + mt) * vld + qcol;
#pragma unroll
        for (int n = 0; n < WN; ++n) vp[16 * n] = vacc[n];
    }
}

// V[Q] = sum over the W kernel's partials, fixed order; entries Ql <= Q < ldq are set to 0.
// 32 aux columns x 8 slices of the partials per workgroup (one thread per column and slice, slices added in fixed order
// through LDS): with one thread per column walking all ~200-500 partial rows alone the launch took 55-60 us.
__global__ __launch_bounds__(256) void k_reduce_V(const double *__restrict__ vpart, int nparts, int vld, int Ql, int ldq,
                                                  double *__restrict__ V)
{
    __shared__ double part[8][32];
    const int col = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const int Q = blockIdx.x * 32 + col;
    double v = 0.0;
    if (Q < Ql) {
        const int per = (nparts + 7) / 8, t0 = sl * per, t1 = min(nparts, t0 + per);
        for (int t = t0; t < t1; ++t) v += vpart[(int64_t)t * vld + Q];
    }
    part[sl][col] = v;
    __syncthreads();
    if (sl == 0 && Q < ldq) {
        double sum = 0.0;
#pragma unroll
        for (int k = 0; k < 8; ++k) sum += part[k][col];
        V[Q] = sum;
    }
}

// ---------------------------------------------------------------------------
// k_coulomb_J: J[j] = sum_Q V[Q] Bp[jrow[j]][Q] for the kept pairs with q >= p (B is symmetric in (q,p):
// this pass streams half of B).  HBM-bound, no MFMA: one wave per group of J_ROWS packed rows, every row a
// contiguous run of ldq doubles; V lives in LDS.
// Reference: calculate_J_screened_GPU (GPUDF.jl:544-547) / DenseGPUDF.jl:103 / the lower-triangle runs of
// calculate_coulomb_screened (ScreenedDF.jl:318-365).
// ---------------------------------------------------------------------------
// Register budget: beside the K kernel (two ~216-VGPR waves per SIMD) 80 of a SIMD's 512 registers are free, so this
// kernel must stay <= 80 VGPRs to run WHILE K runs (J beside K): 4 rows x 3 steps = 12 loads of 16 B in flight.
#ifndef JCDF_J_ROWS
#define JCDF_J_ROWS 4
#endif
constexpr int J_ROWS = JCDF_J_ROWS;
#ifndef JCDF_J_UNROLL
#define JCDF_J_UNROLL 3
#endif
constexpr int J_UNROLL = JCDF_J_UNROLL;
#ifndef JCDF_J_BLOCKS_PER_CU
#define JCDF_J_BLOCKS_PER_CU 6
#endif
#ifndef JCDF_J_PRIO
#define JCDF_J_PRIO 3
#endif
__global__ __launch_bounds__(256, JCDF_J_BLOCKS_PER_CU) void k_coulomb_J(
    const double *__restrict__ Bp, int64_t ldq, const double *__restrict__ V, const int *__restrict__ jrow,
    int64_t nrows, double *__restrict__ J)
{
    extern __shared__ __attribute__((aligned(16))) double Vs[];
    // Beside the K kernel (two 216-VGPR MFMA waves per SIMD) this kernel's one wave per SIMD was starved of issue slots at the
    // default priority: 53 % of its bytes in the 0.72 ms K needs, the rest behind it (window 0.87 ms).  At priority 3 it
    // streams while K multiplies and both end together: window 0.76 ms, K itself 0.72 -> 0.75 (profiles/r04_jk_phase.txt).
    __builtin_amdgcn_s_setprio(JCDF_J_PRIO);
    for (int k = threadIdx.x; k < (int)ldq; k += blockDim.x) Vs[k] = V[k];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
    const int n2 = (int)(ldq / 2);
    const double2_t *V2 = reinterpret_cast<const double2_t *>(Vs);
    for (int64_t g = wave0 * J_ROWS; g < nrows; g += nwaves * J_ROWS) {
        const double2_t *ptr[J_ROWS];
#pragma unroll
        for (int r = 0; r < J_ROWS; ++r) {
            const int64_t jr = (g + r < nrows) ? g + r : nrows - 1;
            ptr[r] = reinterpret_cast<const double2_t *>(Bp + (int64_t)__builtin_amdgcn_readfirstlane(jrow[jr]) * ldq);
        }
        double acc[J_ROWS];
#pragma unroll
        for (int r = 0; r < J_ROWS; ++r) acc[r] = 0.0;
        int x = lane;
#pragma unroll 1
        for (; x + 64 * (J_UNROLL - 1) < n2; x += 64 * J_UNROLL) {
            double2_t v[J_UNROLL][J_ROWS];
#pragma unroll
            for (int u = 0; u < J_UNROLL; ++u)
#pragma unroll
                for (int r = 0; r < J_ROWS; ++r) v[u][r] = __builtin_nontemporal_load(ptr[r] + x + 64 * u);
#pragma unroll
            for (int u = 0; u < J_UNROLL; ++u) {
                const double2_t vq = V2[x + 64 * u];
#pragma unroll
                for (int r = 0; r < J_ROWS; ++r) acc[r] += vq.x * v[u][r].x + vq.y * v[u][r].y;
            }
        }
        for (; x < n2; x += 64) {
            const double2_t vq = V2[x];
#pragma unroll
            for (int r = 0; r < J_ROWS; ++r) {
                const double2_t v = __builtin_nontemporal_load(ptr[r] + x);
                acc[r] += vq.x * v.x + vq.y * v.y;
            }
        }
#pragma unroll
        for (int r = 0; r < J_ROWS; ++r)
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) acc[r] += __shfl_xor(acc[r], off, 64);
        if (lane == 0) {
#pragma unroll
            for (int r = 0; r < J_ROWS; ++r)
                if (g + r < nrows) J[g + r] = acc[r];
        }
    }
}

// ---------------------------------------------------------------------------
// k_exchange_K64: Kslab[s][blk] = sum_{k in slice s} W[row-block bi][k] W[row-block bj][k] for the 64 x 64 blocks
// (bi >= bj) of the lower triangle that are needed, k = (i, Q).  SYRK with a huge contraction (o*ldq) and a tiny output
// -> split-K over slices, deterministic slab reduce in k_fock_assemble.  Reference: calcululate_K_no_sym_GPU! /
// lower-triangle block GEMMs (GPUDF.jl:669-672, 758-826) / DenseGPUDF.jl:111; with exchange screening the block list
// of calculate_exchange_block_screen_matrix + calculate_K_lower_diagonal_block (ScreenedDF.jl:431-447, 459-545).
//
// Unit of work = one WAVE computing one 64 x 64 block (16 MFMA tiles, 128 accumulator registers).  A workgroup is a
// GROUP of up to four such blocks that together touch at most four 64-row blocks of W (jcdf_configure builds the
// groups: an off-diagonal 128 x 128 tile is one group of 2 x 2 blocks sharing 2 + 2 row blocks; the 3-block diagonal
// 128-tiles are packed four blocks to a group across neighbouring tiles, so no wave idles and no upper-triangle
// block is computed: N = 510 -> 36 blocks in 9 groups where 128-tiles needed 10 workgroups with 4 idle waves).
// Staging: one stage = 16 k of the group's four row blocks = 4 x 8 KB, each a CONTIGUOUS 8 KB half of a 16 KB block
// of Wb, copied by LDS-DMA (global_load_lds_dwordx4, 8 wave instructions of 1 KB per row block, wave w moves row
// block w) with the 16-B chunks of every row XOR-swizzled ON THE SOURCE SIDE (the DMA's LDS side is lane-linear):
// chunk c of row r lands at chunk position c ^ ((r >> 1) & 7).  Rows are 128 B = half a bank row, so the 16 rows one
// 32-lane half of a ds_read_b64 touches (same k chunk) fall on 2 x 8 distinct 16-B slots: conflict-free (the earlier
// key r & 7 left rows r and r + 8 on the same banks: SQ_LDS_BANK_CONFLICT = 50 % of SQ_LDS_IDX_ACTIVE).
// Two LDS buffers: the DMA of stage t+1 is issued at the start of phase t into the buffer phase t-1 has finished
// reading and has the whole phase (64 MFMAs per wave) to land; operands come from L2 (all groups of a k-slice run on
// one XCD).  No staging registers, no ds_write.
// Group descriptor (16 ints): rb[4] = staged row blocks (-1: slot unused), then per wave {slot a, slot b, output
// block index or -1 (idle wave)}.
// ---------------------------------------------------------------------------
constexpr int KB64 = 64;                          // block edge
constexpr int K64_STAGE = 4 * KB64 * KC;          // doubles per stage (4 row blocks x 64 rows x 16 k)
constexpr int K64_SMEM_BYTES = 2 * K64_STAGE * 8; // two buffers

__global__ __launch_bounds__(256, 2) void k_exchange_K64(const double *__restrict__ Wt, int64_t Wld, const int *__restrict__ groups,
                                                         int ngroups, int S, int KS, double *__restrict__ Kslab, int nblk64)
{
    constexpr int WM = 4, WN = 4;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int b = blockIdx.x;
    const int xcd = b & 7, r = b >> 3;
    const int g = r % ngroups;
    const int s = (r / ngroups) * 8 + xcd;
    if (s >= S) return;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, lk = lane >> 4;
    const int *G = groups + 16 * g;
    const int my_rb = G[wave];                                        // row block this wave stages (-1: none)
    const int pa = G[4 + 3 * wave], pb = G[5 + 3 * wave], ob = G[6 + 3 * wave];
    const bool mfma_on = ob >= 0;

    double4_t acc[WM][WN];
#pragma unroll
    for (int m = 0; m < WM; ++m)
#pragma unroll
        for (int n = 0; n < WN; ++n) acc[m][n] = double4_t{0.0, 0.0, 0.0, 0.0};

    const int64_t nkb = Wld / KC;
    const int nchunks = KS / KC;
    // piece i (8 rows) of this wave's row block: lane l fills LDS row 8 i + l/8, chunk position l%8, from source chunk
    // (l%8) ^ key(row), key(row) = (row >> 1) & 7 = (4 i + (l >> 4)) & 7: l >> 4 for even i, 4 + (l >> 4) for odd i
    const int lrow = (lane >> 3) * KC;
    const int src_even = lrow + (((lane & 7) ^ (lane >> 4)) << 1);
    const int src_odd = lrow + (((lane & 7) ^ (4 + (lane >> 4))) << 1);
    const int rbs = my_rb < 0 ? 0 : my_rb;
    const double *Sb = Wt + (((int64_t)(rbs >> 1) * nkb + (int64_t)s * nchunks) * 128 + (rbs & 1) * 64) * KC;
    auto issue = [&](int buf, int chunk) {
        if (my_rb < 0) return;                                        // wave-uniform
        double *Ls = smem + buf * K64_STAGE + wave * (KB64 * KC);
        const double *src = Sb + (int64_t)chunk * (128 * KC);
#pragma unroll
        for (int i = 0; i < 8; ++i)
            __builtin_amdgcn_global_load_lds((glb_void_t *)(src + i * 128 + ((i & 1) ? src_odd : src_even)), (lds_void_t *)(Ls + i * 128), 16, 0,
                                             0);
    };
    int koff[KC / 4];                                 // position of k = 4 ks + lk in this lane's rows (key = lr >> 1 for every m)
#pragma unroll
    for (int ks = 0; ks < KC / 4; ++ks) {
        const int k = 4 * ks + lk;
        koff[ks] = (((k >> 1) ^ (lr >> 1)) << 1) | (k & 1);
    }
    auto compute_stage = [&](int buf) {
        if (!mfma_on) return;
        const double *As = smem + buf * K64_STAGE + pa * (KB64 * KC) + lr * KC;
        const double *Bs = smem + buf * K64_STAGE + pb * (KB64 * KC) + lr * KC;
#pragma unroll
        for (int ks = 0; ks < KC / 4; ++ks) {
            double a[WM], bb[WN];
#pragma unroll
            // volatile: keeps these as eight ds_read_b64 (64-bank mode, 2 LDS cycles each, conflict-free with the swizzle above).
            // Merged into ds_read2st64_b64 by the compiler they are banked modulo 32 in groups of 16 lanes: rows r and r + 1
            // collide (SQ_LDS_BANK_CONFLICT = 50 % of SQ_LDS_IDX_ACTIVE) and a pair costs 16 cycles instead of 4.
            for (int m = 0; m < WM; ++m) a[m] = *(lds_cvdouble_t *)(As + m * 16 * KC + koff[ks]);
#pragma unroll
            for (int n = 0; n < WN; ++n) bb[n] = *(lds_cvdouble_t *)(Bs + n * 16 * KC + koff[ks]);
#pragma unroll
            for (int m = 0; m < WM; ++m)
#pragma unroll
                for (int n = 0; n < WN; ++n) acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m], bb[n], acc[m][n], 0, 0, 0);
        }
    };

    issue(0, 0);
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    for (int c = 0; c < nchunks; ++c) {
        issue((c + 1) & 1, min(c + 1, nchunks - 1));       // past the end: the last block again (nobody reads it)
        compute_stage(c & 1);
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
    }

    if (mfma_on) {
        double *out = Kslab + ((int64_t)s * nblk64 + ob) * (KB64 * KB64);
#pragma unroll
        for (int m = 0; m < WM; ++m)
#pragma unroll
            for (int n = 0; n < WN; ++n)
#pragma unroll
                for (int j = 0; j < 4; ++j) out[(m * 16 + lk + 4 * j) * KB64 + n * 16 + lr] = acc[m][n][j];
    }
}

// ---------------------------------------------------------------------------
// k_fock_assemble: F = 2 J - K (+ H), symmetric, written in the reference's
// N x N column-major layout.  Sums the K slabs in fixed order; J is scattered from the
// packed lower pairs through cmap[q + N p] (index into J, -1: screened pair).
// Replaces copy_screened_J_to_fock_upper_triangle + copy_upper_to_lower_kernel
// + axpy!(H) (GPUDF.jl:482-536, 221-225).
// kblk[bi (bi+1)/2 + bj] = slab index of the 64 x 64 block (bi >= bj), -1: block not computed.  Exchange screening
// (bscr != NULL; ScreenedDF.jl:431-447): element (q, p), q >= p, takes its K only if the reference's K block
// (q / bsw, p / bsw) is kept or q lies in the ragged strip q >= nbs * bsw the reference always computes
// (ScreenedDF.jl:518-545); elsewhere K = 0, which is what the reference's skipped blocks hold on a fresh Fock array.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_fock_assemble(
    const double *__restrict__ J, const int *__restrict__ cmap, const double *__restrict__ Kslab, int S, int nblk64,
    const int *__restrict__ kblk, const unsigned char *__restrict__ bscr, int bsw, int nbs,
    const double *__restrict__ H, int N, double *__restrict__ F, int64_t ldf)
{
    const int q = blockIdx.y;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p > q || q >= N) return;
    const int jc = cmap[q + (int64_t)N * p];
    const double j = jc >= 0 ? J[jc] : 0.0;
    const int bi = q >> 6, bj = p >> 6;
    const int blk = kblk[bi * (bi + 1) / 2 + bj];
    bool kept = blk >= 0;
    if (bscr && kept) kept = q >= nbs * bsw || bscr[(q / bsw) * nbs + p / bsw] != 0;
    double k = 0.0;
    if (kept) {
        const int64_t off = (int64_t)blk * (KB64 * KB64) + (q & 63) * KB64 + (p & 63);
        for (int s = 0; s < S; ++s) k += Kslab[(int64_t)s * nblk64 * (KB64 * KB64) + off];
    }
    const double v = 2.0 * j - k;
    const int64_t lo = q + (int64_t)N * p, up = p + (int64_t)N * q;
    F[q + ldf * p] = v + (H ? H[lo] : 0.0);
    if (p != q) F[p + ldf * q] = v + (H ? H[up] : 0.0);
}

// ---------------------------------------------------------------------------
// setup kernels
// ---------------------------------------------------------------------------
// k_metric_apply: Bp[c][r] += sum_s T[c][s] Linv[r][s]   ("(B|Q)^-1 metric solve", B = L^-1 T in the
// reference's own (Q_d, P) layout).  T: the pushed block of three-centre integrals, (R x P) column-major with
// the rows padded to Rpad (zeros) -> row c contiguous in s; Linv: rows r of this shard, s contiguous.
// One launch per push, so B is read and written once per push.  Replaces CUBLAS.trmm!/gemm! of
// GPUDF.jl:907,939-943 and DenseGPUDF.jl:210,270.
using MCfg = GemmCfg<4, 4, 2, 2, KC>;

__global__ __launch_bounds__(256, 2) void k_metric_apply(
    const double *__restrict__ T, int64_t ldt, int64_t P, const double *__restrict__ Linv, int64_t ldl, int nchunks,
    int Ql, int n_ctiles, int rt0, double *__restrict__ Bp, int64_t ldq)
{
    using Cfg = MCfg;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int ct = blockIdx.x % n_ctiles;
    const int rt = rt0 + blockIdx.x / n_ctiles;           // row tiles above the diagonal of L^-1 are skipped

    double4_t acc[Cfg::WM][Cfg::WN];
#pragma unroll
    for (int m = 0; m < Cfg::WM; ++m)
#pragma unroll
        for (int n = 0; n < Cfg::WN; ++n) acc[m][n] = double4_t{0.0, 0.0, 0.0, 0.0};

    gemm_nt_core<Cfg, true>(T + (int64_t)ct * Cfg::TM * ldt, ldt, Linv + (int64_t)rt * Cfg::TN * ldl, ldl, nchunks, acc, smem);
#pragma unroll
    for (int m = 0; m < Cfg::WM; ++m)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t c = (int64_t)ct * Cfg::TM + tile_row<Cfg>(m, j);
            if (c < P) {
#pragma unroll
                for (int n = 0; n < Cfg::WN; ++n) {
                    const int r = rt * Cfg::TN + tile_col<Cfg>(n);
                    if (r < Ql) Bp[c * ldq + r] += acc[m][n][j];
                }
            }
        }
}

// dst[r][s] = src[s][r] (n x n blocks of a row-major matrix), 32 x 32 LDS tiles: L^-T (row-major upper, what the
// device factorisation leaves) -> rows of L^-1 with s contiguous
__global__ __launch_bounds__(256) void k_transpose(const double *__restrict__ src, int64_t lds_, int64_t r0, int64_t nr,
                                                   int64_t ns, double *__restrict__ dst, int64_t ldd)
{
    __shared__ double tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int64_t rb = (int64_t)blockIdx.x * 32, sb = (int64_t)blockIdx.y * 32;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int64_t s = sb + ty + 8 * k, r = rb + tx;
        tile[ty + 8 * k][tx] = (s < ns && r < nr) ? src[s * lds_ + r0 + r] : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int64_t r = rb + ty + 8 * k, s = sb + tx;
        if (r < nr && s < ns) dst[r * ldd + s] = tile[tx][ty + 8 * k];
    }
}

// Wt (internal [p][i*ldq + Q]) -> reference GPU layout (Ql, o, N) column-major (GPUDF.jl:140)
__global__ void k_export_W(const double *__restrict__ Wt, int64_t Wld, int64_t ldq, int Ql, int o, int N,
                           double *__restrict__ out)
{
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = (int64_t)Ql * o * N;
    if (idx >= total) return;
    const int Q = (int)(idx % Ql);
    const int i = (int)((idx / Ql) % o);
    const int p = (int)(idx / ((int64_t)Ql * o));
    const int64_t k = (int64_t)i * ldq + Q;
    out[idx] = Wt[(((int64_t)(p >> 7) * (Wld >> 4) + (k >> 4)) * 128 + (p & 127)) * 16 + (k & 15)];
}

}  // namespace jcdf
