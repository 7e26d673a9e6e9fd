// jcdf_kernels_diag.hpp - the register-staged predecessor of k_exchange_W_dma.  NOT part of the
// shipping library: compiled only with -DJCDF_DIAGNOSTIC (tools/build_diag.sh) for A/B timing against the LDS-DMA form
// (the round-1/2 K kernels on 128 x 128 tiles are in the git history: their slab layout differs from k_exchange_K64's).
#pragma once
#include "jcdf_kernels.hpp"

namespace jcdf {

// ---------------------------------------------------------------------------
// k_exchange_W: for every p, Wt[p][(i,Q)] = sum_{q kept with p} C[q][i] Bp[(q,p)][Q]   (one pass over B)
// fused with V[Q] = sum_{p,i} W[Q,i,p] C[p][i]  (== B_Q . D~, D~ = C C^T).
// Reference: calculate_W_screened_GPU (GPUDF.jl:637-667; one GEMM per p: (Q_d x K_p)(K_p x o), 2 Q P o flop) /
// DenseGPUDF.jl:107 (W) and GPUDF.jl:539-542 / DenseGPUDF.jl:99 (V gemv, which costs the reference one extra
// pass over B).  Same shape here: MFMA M = orbitals, N = 128 aux indices (contiguous in Bp), K = the K_p kept
// q of this p in stages of 16 — the work is 2 Q P o up to the rounding of K_p to 16.
//
// A workgroup owns one 128-wide aux tile and a CHUNK of consecutive p (host-balanced to ~equal stage counts) and
// streams through the chunk's stages without draining the pipeline at a p boundary: global -> registers two stages
// ahead, registers -> LDS one stage ahead, one barrier per stage; at the last stage of a p the accumulators are
// stored and reset.  Stage table (built once by jcdf_configure, the pattern does not depend on the aux index):
// stg_c[16 t + r] = packed row of slot r, stg_q[16 t + r] = its C row (N = the zero row for padding slots),
// stg_p[t] = p if t is the last stage of p, else -1.
// WVM = 1: 4 waves side by side along Q, tile (16 WM) x 128, two workgroups per CU.  WVM = 2 (more than 128
// occupied orbitals): 8 waves as 2 x 4, tile (32 WM) x 128 — the B stage feeds both orbital halves.
// ---------------------------------------------------------------------------
template <int WM, int WVM = 1>
using WCfg = GemmCfg<WM, 2, WVM, 4, KC>;

template <int WM, int WVM>
__global__ __launch_bounds__(256 * WVM, (WVM == 1 && WM <= 6) ? 2 : ((WVM == 2) ? 2 : 1)) void k_exchange_W(
    const double *__restrict__ Bp, int64_t ldq, const double *__restrict__ Cpad, const double *__restrict__ Cv,
    double *__restrict__ Wt, int64_t Wld, double *__restrict__ vpart, int vld, int o, int opad, int n_mt, int n_qt,
    const int *__restrict__ wchunk, const int *__restrict__ stg_c, const int *__restrict__ stg_q,
    const int *__restrict__ stg_p)
{
    using Cfg = WCfg<WM, WVM>;
    constexpr int TM = Cfg::TM, LDAS = Cfg::LDAS, LDBS = Cfg::LDBS, NW = 4 * WVM;
    constexpr int AH = (TM / 2 + 63) / 64;            // wave instructions per A row (16 B per lane)
    constexpr int A_PER = 16 * AH / NW, B_PER = 16 / NW;
    static_assert((16 * AH) % NW == 0 && 16 % NW == 0, "stage rows do not divide over the waves");
    extern __shared__ __attribute__((aligned(16))) double smem[];

    const int b = blockIdx.x;
    const int qt = b % n_qt, mt = (b / n_qt) % n_mt, chunk = b / (n_qt * n_mt);
    const int t0 = wchunk[chunk], nst = wchunk[chunk + 1] - t0;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / 4, wn = wave % 4;
    const int lr = lane & 15, lk = lane >> 4;

    double4_t acc[WM][2];
#pragma unroll
    for (int m = 0; m < WM; ++m) acc[m][0] = acc[m][1] = double4_t{0.0, 0.0, 0.0, 0.0};
    double vacc[2] = {0.0, 0.0};

    const double *Ag = Cpad + mt * TM;
    const double *Bg = Bp + (int64_t)qt * TILE_Q + 2 * lane;

    // this wave's rows of a stage: A slot s = wave + i NW -> row s / AH, half s % AH; B row wave + i NW.
    // B (HBM, read once) travels through two register sets, two phases ahead of its use; A (the C rows, L2 resident)
    // through one set, one phase ahead — with two sets for both the 96..128-orbital forms spill.
    double2_t ra[A_PER], rb0[B_PER], rb1[B_PER];
    int iq[A_PER], ic[B_PER];                        // wave-uniform gather indices of the NEXT stage to load (A resp. B)
    auto load_idx_A = [&](int t) {
        const int *sq = stg_q + (int64_t)(t0 + t) * 16;
#pragma unroll
        for (int i = 0; i < A_PER; ++i) iq[i] = sq[(wave + i * NW) / AH];
    };
    auto load_idx_B = [&](int t) {
        const int *sc = stg_c + (int64_t)(t0 + t) * 16;
#pragma unroll
        for (int i = 0; i < B_PER; ++i) ic[i] = sc[wave + i * NW];
    };
    auto load_A = [&]() {
        // no per-lane predicate (a branch around a load makes the compiler drain the vm counter): lanes past the end
        // of the row repeat its last 16 bytes
#pragma unroll
        for (int i = 0; i < A_PER; ++i) {
            const int col2 = min(((wave + i * NW) % AH) * 64 + lane, TM / 2 - 1);
            ra[i] = *reinterpret_cast<const double2_t *>(Ag + (int64_t)iq[i] * opad + 2 * col2);
        }
    };
    auto load_B = [&](double2_t (&rb)[B_PER]) {
#pragma unroll
        for (int i = 0; i < B_PER; ++i)
            rb[i] = __builtin_nontemporal_load(reinterpret_cast<const double2_t *>(Bg + (int64_t)ic[i] * ldq));
    };
    auto store_stage = [&](const double2_t (&rb)[B_PER], int buf) {
        double *As = smem + buf * Cfg::STAGE_DOUBLES;
        double *Bs = As + KC * LDAS;
#pragma unroll
        for (int i = 0; i < A_PER; ++i) {
            const int s = wave + i * NW, col2 = min((s % AH) * 64 + lane, TM / 2 - 1);
            *reinterpret_cast<double2_t *>(As + (s / AH) * LDAS + 2 * col2) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < B_PER; ++i) *reinterpret_cast<double2_t *>(Bs + (wave + i * NW) * LDBS + 2 * lane) = rb[i];
    };
    auto compute_stage = [&](int buf) {
        const double *As = smem + buf * Cfg::STAGE_DOUBLES + wm * (WM * 16) + lr;
        const double *Bs = smem + buf * Cfg::STAGE_DOUBLES + KC * LDAS + wn * 32 + lr;
#pragma unroll
        for (int ks = 0; ks < KC / 4; ++ks) {
            double a[WM], bb[2];
#pragma unroll
            for (int m = 0; m < WM; ++m) a[m] = As[(ks * 4 + lk) * LDAS + m * 16];
#pragma unroll
            for (int n = 0; n < 2; ++n) bb[n] = Bs[(ks * 4 + lk) * LDBS + n * 16];
#pragma unroll
            for (int m = 0; m < WM; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m], bb[n], acc[m][n], 0, 0, 0);
        }
    };
    const int qcol = qt * TILE_Q + wn * 32 + lr;           // aux column of acc[.][0]; acc[.][1] is 16 further
    const int ibase = (mt * WVM + wm) * (WM * 16) + lk;    // orbital of acc[0][.][0]
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
                for (int n = 0; n < 2; ++n)
                    vacc[n] += (acc[m][n][0] * c01.x + acc[m][n][1] * c01.y) + (acc[m][n][2] * c23.x + acc[m][n][3] * c23.y);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // (p, k = i*ldq + Q) -> block (p/128, k/16), row p%128: the 16 lanes of a group write one 128-B row of a block;
        // one orbital further = ldq/16 blocks further, and this lane's orbitals are ibase + 4 (4m + j)
        const int64_t istep4 = (ldq >> 4) * (4 * 2048);
        double *wp = Wt + ((int64_t)(p >> 7) * (Wld >> 4) * 128 + (p & 127)) * 16 + (int64_t)(qcol >> 4) * 2048 + lr +
                     (int64_t)ibase * (istep4 >> 2);
        const bool c0 = qcol < ldq, c1 = qcol + 16 < ldq;
#pragma unroll
        for (int m = 0; m < WM; ++m)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (ibase + m * 16 + 4 * j < o) {
                    if (c0) wp[0] = acc[m][0][j];
                    if (c1) wp[2048] = acc[m][1][j];
                }
                wp += istep4;
            }
#pragma unroll
        for (int m = 0; m < WM; ++m) acc[m][0] = acc[m][1] = double4_t{0.0, 0.0, 0.0, 0.0};
    };

    // No load or LDS store of the loop is conditional: past the chunk's last stage the stage index is clamped (the
    // last stage is loaded and staged again, nobody reads it).  A branch around a load or around the store that
    // retires it makes the compiler's wait-count pass drain the vm counter at every phase.
    // The gather indices of a phase's loads are fetched (scalar loads) at the END of the previous phase, just before
    // the LDS stores and the barrier: they share the lgkm counter with the LDS reads, and a scalar load in flight at
    // the first ds_read of a phase would hold that read's wait.
    auto next_idx = [&](int tA, int tB) {
        load_idx_A(min(tA, nst - 1));
        load_idx_B(min(tB, nst - 1));
    };
    next_idx(0, 0);
    load_A();                                        // stage 0
    load_B(rb0);                                     // stage 0
    next_idx(1, 1);
    store_stage(rb0, 0);
    load_B(rb1);                                     // stage 1
    next_idx(1, 2);
    __syncthreads();
    for (int t = 0;; t += 2) {
        // even stage t: LDS buffer 0; B set 1 holds stage t+1 (in flight); B set 0 and the A set are free
        load_A();                                    // stage t+1.  A before B: the vm counter is in order, and the A set
        load_B(rb0);                                 // stage t+2   is waited for at the end of THIS phase, the B set a phase later
        compute_stage(0);
        {
            const int p = stg_p[t0 + t];
            if (p >= 0) epilogue(p);
        }
        next_idx(t + 2, t + 3);
        store_stage(rb1, 1);
        __syncthreads();
        if (t + 1 >= nst) break;
        // odd stage t+1: LDS buffer 1; B set 0 holds stage t+2 (in flight); B set 1 and the A set are free
        load_A();                                    // stage t+2
        load_B(rb1);                                 // stage t+3
        compute_stage(1);
        {
            const int p = stg_p[t0 + t + 1];
            if (p >= 0) epilogue(p);
        }
        next_idx(t + 3, t + 4);
        store_stage(rb0, 0);
        __syncthreads();
        if (t + 2 >= nst) break;
    }

    // V partial of this (chunk, m tile, aux tile): lane groups and wave rows in fixed order
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        vacc[n] += __shfl_xor(vacc[n], 16, 64);
        vacc[n] += __shfl_xor(vacc[n], 32, 64);
    }
    if (WVM == 2) {
        if (wm == 1 && lk == 0) { smem[wn * 32 + lr] = vacc[0]; smem[wn * 32 + 16 + lr] = vacc[1]; }
        __syncthreads();
        if (wm == 0) { vacc[0] += smem[wn * 32 + lr]; vacc[1] += smem[wn * 32 + 16 + lr]; }
    }
    if (wm == 0 && lk == 0) {
        double *vp = vpart + ((int64_t)chunk * n_mt + mt) * vld + qcol;
        vp[0] = vacc[0];
        vp[16] = vacc[1];
    }
}

}  // namespace jcdf
