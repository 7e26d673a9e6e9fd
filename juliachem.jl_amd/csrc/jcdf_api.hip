// jcdf_api.hip — C ABI (include/jcdf.h) of the MI355X DF-RHF Fock build.
// Host-side orchestration only: buffer management, kernel launches on one HIP
// stream, HIP-event timing.  All arithmetic of the hot path lives in
// jcdf_kernels.hpp.  There is deliberately NO CPU fallback: without a HIP device
// jcdf_create fails with JCDF_ERR_NO_DEVICE.
#include "../../include/jcdf.h"
#include "jcdf_kernels.hpp"
#include "jcdf_host_lapack.hpp"
#include "jcdf_eig.hpp"
#include "jcdf_chol.hpp"
#include "jcdf_dc.hpp"
#include "jcdf_sp2.hpp"
#include "jcdf_scf.hpp"
#include "jcdf_blas.hpp"
#include "jcdf_wy.hpp"
#ifdef JCDF_DIAGNOSTIC            // tools/build_diag.sh: ablations, experiment kernels and the variant knobs; never in the shipping library
#include "jcdf_kernels_diag.hpp"
#include "jcdf_sbr.hpp"
#include "jcdf_diag.h"
#endif

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <map>
#include <mutex>
#include <tuple>
#include <string>
#include <vector>

using namespace jcdf;

namespace {

std::string g_create_error;

inline int64_t roundup(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

// Variant / experiment knobs are environment variables of DIAGNOSTIC builds only (-DJCDF_DIAGNOSTIC, tools/build_diag.sh);
// the shipping library reads no environment: what a caller may tune goes through jcdf_set_tuning.
#ifdef JCDF_DIAGNOSTIC
inline const char *diag_env(const char *name) { return getenv(name); }
#else
inline const char *diag_env(const char *) { return nullptr; }
#endif

struct KernelRec {
    const char *name;
    hipEvent_t e0, e1;       // the two events of the most recent build's set that bracket this kernel
    int i0, i1;              // their indices in a set
    double flops, alg_flops, alg_bytes;
    double sum_seconds;      // accumulated over the builds folded so far (jcdf_kernel_stats_total)
};

}  // namespace

struct jcdf_handle {
    int device = -1;
    int num_cu = 256;
    hipStream_t stream = nullptr;       // stream every library operation is enqueued on (jcdf_set_stream)
    hipStream_t own_stream = nullptr;   // created by jcdf_create, the default for `stream`
    std::string err;

    // sizes
    int64_t N = 0, Qtot = 0, q0 = 0, q1 = 0, Ql = 0, o = 0, P = 0;
    int64_t Np = 0, ldq = 0, Wld = 0, Plow = 0;
    int WMw = 0, WVMw = 1, n_mtiles = 0, opad = 0, n_qt = 0, vld = 0;
    int w_rem = 0;                      // trailing orbitals (n_occ mod 16) contracted by VALU FMAs in the DMA W kernel
    int n_chunks = 0, n_stages = 0;
    int w_ablate = 0;                  // diagnostic builds: timing-only ablation bits of the DMA kernel (JCDF_W_ABLATE)
    bool w_skip_partial = true;        // DMA kernel: waves past the end of the aux rows (partial last tile) issue no MFMA
    bool w_dma = true;                 // LDS-DMA staging (k_exchange_W_dma); false only in diagnostic builds (register-staged k_exchange_W)
    int tq = TILE_Q;                   // aux-index tile of the W kernel: 128, or 256 for the DMA kernel up to 96 orbitals
    int kcw = KCD;                     // slots per stage of the W kernel's stage table (8 resp. 16)
    int ngroups = 0, nblk64 = 0, S = 0, KS = 0;      // K kernel: workgroup groups of up to four 64 x 64 blocks, blocks computed, split-K
    int64_t xs_blocks = 0;                 // jcdf_set_exchange_screening: the reference's df_exchange_n_blocks with df_exchange_screen (0: off)
    int xs_width = 0, xs_nb = 0;           // K_block_width and block count in effect (ScreenedDF.jl:392-396)
    bool configured = false, have_metric = false, have_B = false, have_H = false, pushed_any = false;
    bool dense_map = true;
    // jcdf_set_tuning (persist across jcdf_configure); 0 = the library's own rule
    int64_t tune_k_slices_per_xcd = 0, tune_w_chunk_stages = 0, tune_host_cholesky = 0, tune_j_workgroups = 0, tune_k_first = 0;

    // device buffers
    double *dB = nullptr, *dCpad = nullptr, *dCv = nullptr, *dWt = nullptr, *dVpart = nullptr, *dV = nullptr;
    double *dJ = nullptr, *dKslab = nullptr, *dH = nullptr, *dF = nullptr, *dC = nullptr;
    double *dLinv = nullptr;                             // rows [q0,q1) of L^-1, row-major (s contiguous)
    int64_t ldl = 0, linv_rows = 0;
    int *dWchunk = nullptr, *dStgC = nullptr, *dStgQ = nullptr, *dStgP = nullptr;   // stage table of the W kernel
    int *dJrow = nullptr, *dCmap = nullptr;              // packed rows with q >= p; (q,p) -> index into J
    int *dKgroups = nullptr, *dKblk = nullptr;           // K kernel group descriptors; (bi,bj) -> slab index of the 64 x 64 block
    unsigned char *dBscr = nullptr;                      // exchange screening: the reference's block_screen_matrix (row-major nb x nb)
    unsigned long long *dStall = nullptr;                // diagnostic builds, JCDF_W_ABLATE=32: per-wave segment cycles of the W kernel
    double *dStage = nullptr;                            // setup staging for pushed three-centre blocks, freed after setup
    int64_t stage_doubles = 0;
    int64_t bytes = 0;

    // timing
    std::vector<KernelRec> recs;
    // Device time stamps of a build: EIGHT events, shared between neighbouring kernels (the end of one is the start of the next;
    // every record is a barrier packet on the queue, ~5 us of bubble each: 15 records per build cost 70 us, 8 cost half), in
    // TWO sets used in turn, so that the times of build k are read while build k+1 is already enqueued.
    //   0 begin = prep_C start | 1 prep_C end = W start | 2 W end | 3 fork = K start = J start | 4 J end (join) | 5 K end |
    //   6 assemble start (behind the join) | 7 assemble end = end of the build
    static constexpr int NEV = 8;
    hipEvent_t evset[2][NEV] = {};
    int evcur = 0;                      // set of the most recent build
    bool set_unfolded[2] = {false, false};
    hipEvent_t ev_begin = nullptr, ev_end = nullptr;     // = evset[evcur][0] / [7] of the most recent build
    hipEvent_t ev_h2d = nullptr, ev_d2h = nullptr;
    hipStream_t side = nullptr;                      // the HBM-bound J pass runs beside the MFMA-bound K pass (overlap_jk)
    bool overlap_jk = true;
    bool timed_host_copy = false;
    bool pending = false;
    int64_t builds_folded = 0;
    double fock_sum_seconds = 0.0;
};

namespace {

#define JCDF_HIP(h, call)                                                                    \
    do {                                                                                     \
        hipError_t e_ = (call);                                                              \
        if (e_ != hipSuccess) {                                                              \
            (h)->err = std::string(#call) + ": " + hipGetErrorString(e_);                    \
            return JCDF_ERR_HIP;                                                             \
        }                                                                                    \
    } while (0)

hipError_t ensure_device_attributes();

int32_t fail(jcdf_handle *h, int32_t code, const std::string &msg)
{
    if (h) h->err = msg;
    return code;
}

template <class T>
int32_t dev_alloc(jcdf_handle *h, T **p, int64_t count, bool zero)
{
    *p = nullptr;
    if (count <= 0) count = 1;
    hipError_t e = hipMalloc((void **)p, (size_t)count * sizeof(T));
    if (e != hipSuccess) {
        h->err = "hipMalloc(" + std::to_string(count * (int64_t)sizeof(T)) + " B): " + hipGetErrorString(e);
        return JCDF_ERR_ALLOC;
    }
    h->bytes += count * (int64_t)sizeof(T);
    if (zero) JCDF_HIP(h, hipMemsetAsync(*p, 0, (size_t)count * sizeof(T), h->stream));
    return JCDF_OK;
}

template <class T>
int32_t dev_upload(jcdf_handle *h, T **p, const std::vector<T> &v)
{
    int32_t rc = dev_alloc(h, p, (int64_t)std::max<size_t>(v.size(), 1), false);
    if (rc) return rc;
    if (!v.empty()) JCDF_HIP(h, hipMemcpyAsync(*p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, h->stream));
    return JCDF_OK;
}

template <class T>
void dev_free(jcdf_handle *h, T **p, int64_t count)
{
    if (*p) {
        (void)hipFree(*p);
        h->bytes -= std::max<int64_t>(count, 1) * (int64_t)sizeof(T);
        *p = nullptr;
    }
}

void free_all(jcdf_handle *h)
{
    (void)hipSetDevice(h->device);
    double **bufs[] = {&h->dB, &h->dCpad, &h->dCv, &h->dWt, &h->dVpart, &h->dV, &h->dJ, &h->dKslab,
                       &h->dH, &h->dF, &h->dC, &h->dLinv, &h->dStage};
    for (auto b : bufs)
        if (*b) { (void)hipFree(*b); *b = nullptr; }
    int **ibufs[] = {&h->dWchunk, &h->dStgC, &h->dStgQ, &h->dStgP, &h->dJrow, &h->dCmap, &h->dKgroups, &h->dKblk};
    for (auto b : ibufs)
        if (*b) { (void)hipFree(*b); *b = nullptr; }
    if (h->dBscr) { (void)hipFree(h->dBscr); h->dBscr = nullptr; }
    if (h->dStall) { (void)hipFree(h->dStall); h->dStall = nullptr; }
    h->recs.clear();                    // (their events belong to h->evset)
    h->set_unfolded[0] = h->set_unfolded[1] = false;
    h->pending = false;
    h->bytes = 0;
    h->stage_doubles = 0;
    h->configured = h->have_metric = h->have_B = h->have_H = h->pushed_any = false;
}

// ---- W kernel dispatch: WMw 16-orbital MFMA row tiles per wave, WVMw wave rows ------------
template <int WM, int WVM, int WN>
hipError_t launch_W_dma_t(jcdf_handle *h, hipStream_t st, bool set_attr_only)
{
    using D = WDmaCfg<WM, WVM, WN>;
    if (set_attr_only)     // per device: jcdf_configure runs on the handle's device
        return hipFuncSetAttribute((const void *)k_exchange_W_dma<WM, WVM, WN>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   D::SMEM_BYTES);
    const int64_t nblk = (int64_t)h->n_chunks * h->n_mtiles * h->n_qt;
    hipLaunchKernelGGL((k_exchange_W_dma<WM, WVM, WN>), dim3((unsigned)nblk), dim3(D::NT), D::SMEM_BYTES, st, h->dB, h->ldq,
                       h->dCpad, h->dCv, h->dWt, h->Wld, h->dVpart, h->vld, (int)h->o, h->opad, h->n_mtiles, h->n_qt,
                       h->dWchunk, h->dStgC, h->dStgQ, h->dStgP, h->w_skip_partial ? 1 : 0, h->dStall);
    return hipSuccess;
}

#ifdef JCDF_DIAGNOSTIC
// timing-only ablations of the C20H42-shaped form (JCDF_W_ABLATE; see k_exchange_W_dma): WRONG RESULTS, diagnostic builds only
template <int ABL>
hipError_t launch_W_ablate_t(jcdf_handle *h, hipStream_t st, bool set_attr_only)
{
    using D = WDmaCfg<6, 1, 2>;
    if (set_attr_only)
        return hipFuncSetAttribute((const void *)k_exchange_W_dma<6, 1, 2, ABL>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   D::SMEM_BYTES);
    const int64_t nblk = (int64_t)h->n_chunks * h->n_mtiles * h->n_qt;
    hipLaunchKernelGGL((k_exchange_W_dma<6, 1, 2, ABL>), dim3((unsigned)nblk), dim3(D::NT), D::SMEM_BYTES, st, h->dB, h->ldq,
                       h->dCpad, h->dCv, h->dWt, h->Wld, h->dVpart, h->vld, (int)h->o, h->opad, h->n_mtiles, h->n_qt,
                       h->dWchunk, h->dStgC, h->dStgQ, h->dStgP, h->w_skip_partial ? 1 : 0, h->dStall);
    return hipSuccess;
}
#endif

// VALU remainder forms: WM full MFMA row tiles + REM trailing orbitals
template <int WM, int REM>
hipError_t launch_W_rem_t(jcdf_handle *h, hipStream_t st, bool set_attr_only)
{
    using D = WDmaCfg<WM, 1, 2, REM>;
    if (set_attr_only)
        return hipFuncSetAttribute((const void *)k_exchange_W_dma<WM, 1, 2, 0, REM>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   D::SMEM_BYTES);
    const int64_t nblk = (int64_t)h->n_chunks * h->n_mtiles * h->n_qt;
    hipLaunchKernelGGL((k_exchange_W_dma<WM, 1, 2, 0, REM>), dim3((unsigned)nblk), dim3(D::NT), D::SMEM_BYTES, st, h->dB, h->ldq,
                       h->dCpad, h->dCv, h->dWt, h->Wld, h->dVpart, h->vld, (int)h->o, h->opad, h->n_mtiles, h->n_qt,
                       h->dWchunk, h->dStgC, h->dStgQ, h->dStgP, h->w_skip_partial ? 1 : 0, h->dStall);
    return hipSuccess;
}

template <int WM>
hipError_t launch_W_rem(jcdf_handle *h, hipStream_t st, bool attr)
{
    switch (h->w_rem) {
        case 1: return launch_W_rem_t<WM, 1>(h, st, attr);
        case 2: return launch_W_rem_t<WM, 2>(h, st, attr);
        default: return launch_W_rem_t<WM, 3>(h, st, attr);
    }
}

template <int WM, int WVM>
hipError_t launch_W_t(jcdf_handle *h, hipStream_t st, bool set_attr_only)
{
    if (h->w_dma && h->w_rem) {
        if constexpr (WVM == 1 && WM >= 3 && WM <= 7) return launch_W_rem<WM>(h, st, set_attr_only);
    }
#ifndef JCDF_DIAGNOSTIC
    return launch_W_dma_t<WM, WVM, 2>(h, st, set_attr_only);
#else
    if (h->w_dma) {
        if constexpr (WVM == 1 && WM <= 6) {
            if (h->tq == 256) return launch_W_dma_t<WM, WVM, 4>(h, st, set_attr_only);
        }
        return launch_W_dma_t<WM, WVM, 2>(h, st, set_attr_only);
    }
    using Cfg = WCfg<WM, WVM>;
    if (set_attr_only)
        return hipFuncSetAttribute((const void *)k_exchange_W<WM, WVM>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   Cfg::SMEM_BYTES);
    const int64_t nblk = (int64_t)h->n_chunks * h->n_mtiles * h->n_qt;
    hipLaunchKernelGGL((k_exchange_W<WM, WVM>), dim3((unsigned)nblk), dim3(Cfg::NT), Cfg::SMEM_BYTES, st, h->dB, h->ldq,
                       h->dCpad, h->dCv, h->dWt, h->Wld, h->dVpart, h->vld, (int)h->o, h->opad, h->n_mtiles, h->n_qt,
                       h->dWchunk, h->dStgC, h->dStgQ, h->dStgP);
    return hipSuccess;
#endif
}

hipError_t launch_W(jcdf_handle *h, hipStream_t st, bool attr = false)
{
#ifdef JCDF_DIAGNOSTIC
    // (the ablation forms exist for 6 MFMA row tiles without the VALU remainder: C20H42 shape with JCDF_W_REM=0)
    if (h->w_ablate && h->w_dma && !h->w_rem && h->WVMw == 1 && h->WMw == 6 && h->tq == TILE_Q) {
        switch (h->w_ablate) {
            case 2: return launch_W_ablate_t<2>(h, st, attr);
            case 4: return launch_W_ablate_t<4>(h, st, attr);
            case 6: return launch_W_ablate_t<6>(h, st, attr);
            case 8: return launch_W_ablate_t<8>(h, st, attr);
            case 14: return launch_W_ablate_t<14>(h, st, attr);
            case 16: return launch_W_ablate_t<16>(h, st, attr);
            case 20: return launch_W_ablate_t<20>(h, st, attr);
            case 30: return launch_W_ablate_t<30>(h, st, attr);
            case 32: return launch_W_ablate_t<32>(h, st, attr);
            default: break;
        }
    }
#endif
    if (h->WVMw == 2) {
        switch (h->WMw) {
            case 5: return launch_W_t<5, 2>(h, st, attr);
            case 6: return launch_W_t<6, 2>(h, st, attr);
            case 7: return launch_W_t<7, 2>(h, st, attr);
            default: return launch_W_t<8, 2>(h, st, attr);
        }
    }
    switch (h->WMw) {
        case 1: return launch_W_t<1, 1>(h, st, attr);
        case 2: return launch_W_t<2, 1>(h, st, attr);
        case 3: return launch_W_t<3, 1>(h, st, attr);
        case 4: return launch_W_t<4, 1>(h, st, attr);
        case 5: return launch_W_t<5, 1>(h, st, attr);
        case 6: return launch_W_t<6, 1>(h, st, attr);
        case 7: return launch_W_t<7, 1>(h, st, attr);
        default: return launch_W_t<8, 1>(h, st, attr);
    }
}

// every hip* status of a build is folded into one word and reported by enqueue_fock
struct HipAcc {
    hipError_t first = hipSuccess;
    void operator()(hipError_t e) { if (first == hipSuccess && e != hipSuccess) first = e; }
};

// record slot idx of the build being enqueued: name and the two events of the current set that bracket it
KernelRec &rec_slot(jcdf_handle *h, size_t idx, const char *name, int e0, int e1)
{
    while (h->recs.size() <= idx) h->recs.push_back(KernelRec{});
    KernelRec &r = h->recs[idx];
    r.name = name;
    r.flops = r.alg_flops = r.alg_bytes = 0.0;
    r.e0 = h->evset[h->evcur][e0];
    r.e1 = h->evset[h->evcur][e1];
    r.i0 = e0;
    r.i1 = e1;
    return r;
}

double elapsed_s(hipEvent_t a, hipEvent_t b);

// Adds the event times of the build recorded in `set` to the running sums (jcdf_kernel_stats_total).  wait = false: only if
// that build has already finished (never blocks the enqueue path); returns false if it was still running.
bool fold_set(jcdf_handle *h, int set, bool wait)
{
    if (!h->set_unfolded[set] || h->recs.empty()) return true;
    hipEvent_t *E = h->evset[set];
    if (wait) {
        if (hipEventSynchronize(E[jcdf_handle::NEV - 1]) != hipSuccess) return false;
    } else if (hipEventQuery(E[jcdf_handle::NEV - 1]) != hipSuccess) {
        return false;
    }
    for (auto &r : h->recs) r.sum_seconds += elapsed_s(E[r.i0], E[r.i1]);
    h->fock_sum_seconds += elapsed_s(E[0], E[jcdf_handle::NEV - 1]);
    h->builds_folded++;
    h->set_unfolded[set] = false;
    return true;
}

// J beside K (side stream) or one after the other: the handle's switch; diagnostic builds: JCDF_OVERLAP_JK overrides
bool overlap_env_now(const jcdf_handle *h)
{
    static const int overlap_env = [] { const char *e = diag_env("JCDF_OVERLAP_JK"); return e ? atoi(e) : -1; }();
    return overlap_env >= 0 ? overlap_env != 0 : h->overlap_jk;
}

int32_t enqueue_fock(jcdf_handle *h, const double *dC, int64_t ldc, double *dF, int64_t ldf, hipStream_t st)
{
    // the set this build records into held the build before last: its times are read now at the latest (a build that is
    // still running when its events are needed again is left out of the sums)
    const int prev = h->evcur, set = h->evcur ^ 1;
    if (!fold_set(h, set, false)) h->set_unfolded[set] = false;
    h->evcur = set;
    hipEvent_t *E = h->evset[set];
    h->ev_begin = E[0];
    h->ev_end = E[jcdf_handle::NEV - 1];
    const double N = (double)h->N, Ql = (double)h->Ql, o = (double)h->o, P = (double)h->P;
    HipAcc ok;
    size_t k = 0;
    ok(hipEventRecord(E[0], st));
    {
        KernelRec &r = rec_slot(h, k++, "k_prep_C", 0, 1);
        const int64_t tot = h->Np * h->opad;
        hipLaunchKernelGGL(k_prep_C, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, dC, ldc, (int)h->N,
                           (int)h->o, (int)h->Np, h->opad, h->WMw, h->n_mtiles * h->WVMw, h->dCpad, h->dCv);
        r.alg_bytes = 8.0 * N * o;
        ok(hipEventRecord(E[1], st));
    }
    {
        KernelRec &r = rec_slot(h, k++, "k_exchange_W", 1, 2);
        ok(launch_W(h, st));
        // incl. all padding: orbitals to opad, K_p to whole stages, the aux index to whole 32-column wave tiles (DMA kernel)
        r.flops = 2.0 * (double)h->kcw * (double)h->n_stages *
                  ((double)(h->opad - (h->w_rem ? 16 : 0)) *
                       (h->w_dma && h->w_skip_partial ? (double)roundup(h->ldq, h->tq / 4) : (double)h->n_qt * h->tq) +
                   (double)h->w_rem * (double)h->n_qt * h->tq);               // MFMA row tiles + the VALU remainder orbitals
        r.alg_flops = 2.0 * Ql * P * o + 2.0 * Ql * N * o;           // the reference's 2 Q P o (+ fused V from W)
        r.alg_bytes = 8.0 * Ql * P + 8.0 * Ql * o * N;               // B read once + W written once
        ok(hipEventRecord(E[2], st));
    }
    // V from the W kernel's partials: only the Coulomb pass needs it, so it goes in front of J (on J's stream)
    auto run_reduce_V = [&](hipStream_t sv) {
        const int nparts = h->n_chunks * h->n_mtiles;
        hipLaunchKernelGGL(k_reduce_V, dim3((unsigned)((h->ldq + 31) / 32)), dim3(256), 0, sv, h->dVpart, nparts, h->vld,
                           (int)h->Ql, (int)h->ldq, h->dV);
    };
    // e0 / e1: indices of the events that bracket the launch; the caller records them
    const bool overlapped = overlap_env_now(h);
    auto run_J = [&](size_t slot, hipStream_t st, int e0, int e1) {
        KernelRec &r = rec_slot(h, slot, "k_coulomb_J", e0, e1);
        const int64_t groups = (h->Plow + 4 * J_ROWS - 1) / (4 * J_ROWS);
        int64_t cap = (int64_t)h->num_cu * JCDF_J_BLOCKS_PER_CU;
        if (overlapped) {
            // Beside K there is room for ONE workgroup of this kernel per CU (K: two workgroups of four 216-VGPR waves, 128 KB of
            // LDS).  A grid of exactly that many workgroups is resident from start to end and evenly loaded; a larger grid leaves
            // a tail of workgroups that start when the first ones finish (256: window 0.76 ms, 288: 0.94, 1536: 0.82-0.87;
            // profiles/r04_jk_phase.txt).  Only while K is the longer of the two: a J pass that outlasts K (few occupied
            // orbitals) keeps the full grid and fills the chip once K is gone.
            const double k_est = 2.0 * (double)h->nblk64 * 64.0 * 64.0 * (double)h->S * (double)h->KS / 55.0e12;
            const double j_est = 8.0 * Ql * (double)h->Plow / 2.5e12;              // at the rate it reaches beside K
            if (k_est >= j_est) cap = h->num_cu;
            if (h->tune_j_workgroups > 0) cap = h->tune_j_workgroups;
        }
        const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(groups, cap));
        hipLaunchKernelGGL(k_coulomb_J, dim3(grid), dim3(256), (size_t)h->ldq * sizeof(double), st, h->dB, h->ldq, h->dV,
                           h->dJrow, h->Plow, h->dJ);
        r.flops = r.alg_flops = 2.0 * Ql * (double)h->Plow;
        r.alg_bytes = 8.0 * Ql * (double)h->Plow;                    // the kept pairs with q >= p, once
    };
    auto run_K = [&](size_t slot, int e0, int e1) {
        KernelRec &r = rec_slot(h, slot, "k_exchange_K", e0, e1);
        const int nwg = (int)(roundup(h->S, 8) * h->ngroups);
        hipLaunchKernelGGL(k_exchange_K64, dim3((unsigned)nwg), dim3(256), K64_SMEM_BYTES, st, h->dWt, h->Wld, h->dKgroups, h->ngroups,
                           h->S, h->KS, h->dKslab, h->nblk64);
        // one wave per needed 64 x 64 block of the lower triangle (diagonal blocks are computed whole)
        r.flops = 2.0 * (double)h->nblk64 * 64.0 * 64.0 * (double)h->S * (double)h->KS;
        r.alg_flops = 2.0 * Ql * o * N * N;                          // dense formula
        r.alg_bytes = 8.0 * Ql * o * N;                              // W read once
    };
    // record slots stay fixed (2 = J, 3 = K) whatever the launch order
    static const bool k_first = [] { const char *e = diag_env("JCDF_K_BEFORE_J"); return e && atoi(e) != 0; }();
    run_reduce_V(st);                                  // 13 us alone, 36 us when it has to start beside the K kernel's workgroups
    ok(hipEventRecord(E[3], st));
    if (overlapped) {
        // J (streams half of B, no MFMA) on a side stream while K (MFMA, W out of L2) runs: both only need W's outputs.
        // The fork and the join are the timing events themselves.
        if (!h->side) ok(hipStreamCreateWithFlags(&h->side, hipStreamNonBlocking));
        ok(hipStreamWaitEvent(h->side, E[3], 0));
        if (!h->tune_k_first) {                            // default: J's latency ramp in front of K's first stage
            run_J(2, h->side, 3, 4);
            ok(hipEventRecord(E[4], h->side));
            run_K(3, 3, 5);
            ok(hipEventRecord(E[5], st));
        } else {
            run_K(3, 3, 5);
            ok(hipEventRecord(E[5], st));
            run_J(2, h->side, 3, 4);
            ok(hipEventRecord(E[4], h->side));
        }
        ok(hipStreamWaitEvent(st, E[4], 0));
    } else if (k_first) {
        run_K(3, 3, 5);
        ok(hipEventRecord(E[5], st));
        run_J(2, st, 5, 4);
        ok(hipEventRecord(E[4], st));
    } else {
        run_J(2, st, 3, 4);
        ok(hipEventRecord(E[4], st));
        run_K(3, 4, 5);
        ok(hipEventRecord(E[5], st));
    }
    ok(hipEventRecord(E[6], st));
    k = 4;
    {
        KernelRec &r = rec_slot(h, k++, "k_fock_assemble", 6, 7);
        hipLaunchKernelGGL(k_fock_assemble, dim3((unsigned)((h->N + 255) / 256), (unsigned)h->N), dim3(256), 0,
                           st, h->dJ, h->dCmap, h->dKslab, h->S, h->nblk64, h->dKblk, h->dBscr, h->xs_width, h->xs_nb,
                           h->have_H ? h->dH : nullptr, (int)h->N, dF, ldf);
        r.alg_bytes = 8.0 * N * N * (h->have_H ? 2.0 : 1.0);
    }
    ok(hipEventRecord(E[7], st));
    ok(hipGetLastError());
    if (ok.first != hipSuccess) return fail(h, JCDF_ERR_HIP, std::string("Fock build enqueue: ") + hipGetErrorString(ok.first));
    h->pending = true;
    h->set_unfolded[set] = true;
    // the previous build has normally finished by now (the caller read its result): its times are added up here, behind the
    // enqueue of this one, not in front of it
    (void)fold_set(h, prev, false);
    return JCDF_OK;
}

double elapsed_s(hipEvent_t a, hipEvent_t b)
{
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, a, b) != hipSuccess) return 0.0;
    return (double)ms * 1e-3;
}

void release_stage(jcdf_handle *h)
{
    dev_free(h, &h->dStage, h->stage_doubles);
    h->stage_doubles = 0;
}

constexpr int64_t STAGE_MAX_DOUBLES = (int64_t)1 << 28;      // 2 GiB of pushed three-centre rows on the device at a time

// Accumulate Bp[:, r] += sum_{s in [s0,s1)} T[s][:] Linv[q0 + r][s].  `T` is the caller's block, column-major
// ((s1-s0) x P); on_device selects the copy kind.  One metric-apply launch per sub-block of rows that fits the
// staging buffer (all of them for the usual block sizes), so B is read and written once per push.
int32_t push_block(jcdf_handle *h, int64_t s0, int64_t s1, const double *T, bool on_device)
{
    const int64_t Rtot = s1 - s0;
    const int64_t Prows = roundup(h->P, MCfg::TM);
    int64_t Rmax = std::max<int64_t>(KC, (STAGE_MAX_DOUBLES / Prows) / KC * KC);
    Rmax = std::min(Rmax, roundup(Rtot, KC));
    if (h->stage_doubles < Prows * Rmax) {
        release_stage(h);
        int32_t rc = dev_alloc(h, &h->dStage, Prows * Rmax, false);
        if (rc) return rc;
        h->stage_doubles = Prows * Rmax;
    }
    const int n_ctiles = (int)(Prows / MCfg::TM);
    const int n_rt = (int)(roundup(h->Ql, MCfg::TN) / MCfg::TN);
    for (int64_t a0 = 0; a0 < Rtot; a0 += Rmax) {
        const int64_t R = std::min(Rmax, Rtot - a0), Rpad = roundup(R, KC);
        // rows [a0, a0+R) of every column -> staging row c = [T[s0+a0 .. ][c], 0 ...] (leading dimension Rpad)
        if (Rpad > R)
            JCDF_HIP(h, hipMemset2DAsync(h->dStage + R, (size_t)Rpad * 8, 0, (size_t)(Rpad - R) * 8, (size_t)h->P, h->stream));
        JCDF_HIP(h, hipMemcpy2DAsync(h->dStage, (size_t)Rpad * 8, T + a0, (size_t)Rtot * 8, (size_t)R * 8, (size_t)h->P,
                                     on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, h->stream));
        // L^-1 is lower triangular: rows of this shard above the block's first column get nothing
        const int rt0 = (int)(std::max<int64_t>(0, s0 + a0 - h->q0) / MCfg::TN);
        if (rt0 < n_rt)
            hipLaunchKernelGGL(k_metric_apply, dim3((unsigned)((int64_t)n_ctiles * (n_rt - rt0))), dim3(MCfg::NT),
                               GemmNT<MCfg>::SMEM_BYTES, h->stream, h->dStage, Rpad, h->P, h->dLinv + (s0 + a0), h->ldl,
                               (int)(Rpad / KC), (int)h->Ql, n_ctiles, rt0, h->dB, h->ldq);
        if (!on_device) JCDF_HIP(h, hipStreamSynchronize(h->stream));   // host buffer may be reused by caller
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(h, JCDF_ERR_HIP, std::string("push_block: ") + hipGetErrorString(e));
    return JCDF_OK;
}

// ---- device Cholesky + triangular inverse of the metric (jcdf_chol.hpp) ------------------------
constexpr int CHOL_DIAG_LDS = 2 * CH_NB * (CH_NB + 1) * 8;
struct CholBuffers {
    double *R = nullptr, *V = nullptr, *invU = nullptr, *UT = nullptr, *T = nullptr;
    int *err = nullptr;
    int64_t n_pad = 0, ld = 0, nblk = 0;
    ~CholBuffers()
    {
        (void)hipFree(R); (void)hipFree(V); (void)hipFree(invU); (void)hipFree(UT); (void)hipFree(T); (void)hipFree(err);
    }
};

// Factor the Q x Q SPD matrix whose column-major lower triangle is at host pointer J and leave
// V = (L^-1)^T (row-major upper, leading dimension w.ld) in w.V.  *info = 0 or the 1-based index
// of the first non-positive pivot.  Everything runs on `st`; returns after a stream sync.
hipError_t chol_inverse_device(hipStream_t st, const double *J, int64_t Q, CholBuffers &w, int *info)
{
#define CH(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return e_; } while (0)
    (void)hipGetLastError();                 // the launches below are checked through it: start from a clean word
    w.n_pad = roundup(Q, 128);
    w.ld = w.n_pad + CH_NB;
    w.nblk = w.n_pad / CH_NB;
    const size_t mat = (size_t)(w.ld * w.ld) * 8;
    CH(hipMalloc((void **)&w.R, mat));
    CH(hipMalloc((void **)&w.V, mat));
    CH(hipMalloc((void **)&w.invU, (size_t)w.nblk * CH_NB * CH_NB * 8));
    CH(hipMalloc((void **)&w.UT, (size_t)w.ld * CH_NB * 8));
    CH(hipMalloc((void **)&w.T, (size_t)w.ld * CH_NB * 8));
    CH(hipMalloc((void **)&w.err, 64));
    CH(hipMemsetAsync(w.R, 0, mat, st));
    CH(hipMemsetAsync(w.V, 0, mat, st));
    CH(hipMemsetAsync(w.UT, 0, (size_t)w.ld * CH_NB * 8, st));
    CH(hipMemsetAsync(w.T, 0, (size_t)w.ld * CH_NB * 8, st));
    CH(hipMemsetAsync(w.err, 0, 64, st));
    CH(hipMemcpy2DAsync(w.R, (size_t)w.ld * 8, J, (size_t)Q * 8, (size_t)Q * 8, (size_t)Q, hipMemcpyHostToDevice, st));
    if (w.n_pad > Q)
        hipLaunchKernelGGL(k_chol_pad_diag, dim3((unsigned)((w.n_pad - Q + 127) / 128)), dim3(128), 0, st, w.R, w.ld,
                           (int)Q, (int)w.n_pad);
    CH(ensure_device_attributes());
    const int diag_lds = CHOL_DIAG_LDS;
    // factor: R -> U (upper, row-major)
    for (int64_t b = 0; b < w.nblk; ++b) {
        const int64_t i0 = b * CH_NB, k0 = i0 + CH_NB, nk = w.n_pad - k0;
        hipLaunchKernelGGL(k_chol_diag, dim3(1), dim3(256), diag_lds, st, w.R, w.ld, (int)i0,
                           w.invU + b * CH_NB * CH_NB, w.err);
        if (nk <= 0) continue;
        const unsigned nt = (unsigned)((nk + 127) / 128);
        double *R12 = w.R + i0 * w.ld + k0;
        hipLaunchKernelGGL(k_chol_trsm<true>, dim3((unsigned)(nk / 64)), dim3(64), 0, st, w.R + i0 * w.ld + i0, w.ld,
                           R12, w.ld, R12, w.ld, nk);
        hipLaunchKernelGGL(k_chol_syrk, dim3(nt * (nt + 1) / 2), dim3(C128Cfg::NT), C128Cfg::SMEM_BYTES, st, R12,
                           w.R + k0 * w.ld + k0, w.ld, (int)nt);
    }
    // inverse: V = U^-1, block rows bottom-up
    for (int64_t b = w.nblk - 1; b >= 0; --b) {
        const int64_t i0 = b * CH_NB, k0 = i0 + CH_NB, nk = w.n_pad - k0;
        hipLaunchKernelGGL(k_chol_put_diag, dim3(CH_NB * CH_NB / 256), dim3(256), 0, st, w.invU + b * CH_NB * CH_NB,
                           w.V, w.ld, (int)i0);
        if (nk <= 0) continue;
        const unsigned nt = (unsigned)((nk + 127) / 128);
        hipLaunchKernelGGL(k_chol_transpose_panel, dim3((unsigned)((nk + 63) / 64)), dim3(256), 0, st, w.R, w.ld,
                           (int)i0, k0, nk, w.UT);
        hipLaunchKernelGGL(k_chol_inv_gemm, dim3(nt), dim3(C64Cfg::NT), C64Cfg::SMEM_BYTES, st, w.UT, w.V, w.ld, k0,
                           w.T, w.ld);
        hipLaunchKernelGGL(k_chol_trsm<false>, dim3((unsigned)(nk / 64)), dim3(64), 0, st, w.R + i0 * w.ld + i0, w.ld,
                           w.T, w.ld, w.V + i0 * w.ld + k0, w.ld, nk);
    }
    CH(hipGetLastError());
    CH(hipMemcpyAsync(info, w.err, sizeof(int), hipMemcpyDeviceToHost, st));
    CH(hipStreamSynchronize(st));
#undef CH
    return hipSuccess;
}

// A new metric starts a new B: the next push zeroes it again (jcdf.h: "the first push zeroes B").
int32_t alloc_linv(jcdf_handle *h)
{
    h->ldl = roundup(h->Qtot, KC) + KC;                 // a pushed block's zero-padded tail may reach KC-1 columns past Qtot
    h->linv_rows = roundup(h->Ql, MCfg::TN);
    if (!h->dLinv) {
        int32_t rc = dev_alloc(h, &h->dLinv, h->linv_rows * h->ldl, true);
        if (rc) return rc;
    } else {
        JCDF_HIP(h, hipMemsetAsync(h->dLinv, 0, (size_t)(h->linv_rows * h->ldl) * 8, h->stream));
    }
    h->pushed_any = h->have_B = false;
    return JCDF_OK;
}

int32_t upload_linv(jcdf_handle *h, const double *Linv)
{
    // dLinv[r][s] = Linv[(q0 + r) + Qtot * s]: rows of this shard, s contiguous
    int32_t rc = alloc_linv(h);
    if (rc) return rc;
    std::vector<double> rows;
    try {
        rows.assign((size_t)(h->Ql * h->Qtot), 0.0);
    } catch (...) {
        return fail(h, JCDF_ERR_ALLOC, "jcdf_set_metric: out of host memory");
    }
    for (int64_t s = 0; s < h->Qtot; ++s)
        for (int64_t r = std::max<int64_t>(0, s - h->q0); r < h->Ql; ++r)      // lower triangular: q0 + r >= s
            rows[(size_t)(r * h->Qtot + s)] = Linv[(h->q0 + r) + h->Qtot * s];
    JCDF_HIP(h, hipMemcpy2DAsync(h->dLinv, (size_t)h->ldl * 8, rows.data(), (size_t)h->Qtot * 8, (size_t)h->Qtot * 8,
                                 (size_t)h->Ql, hipMemcpyHostToDevice, h->stream));
    JCDF_HIP(h, hipStreamSynchronize(h->stream));
    h->have_metric = true;
    return JCDF_OK;
}

// dLinv[r][s] = V[s][q0 + r] straight from the device factorisation (V = L^-T, row-major upper)
int32_t upload_linv_from_device(jcdf_handle *h, const double *V, int64_t ldv)
{
    int32_t rc = alloc_linv(h);
    if (rc) return rc;
    dim3 grid((unsigned)((h->Ql + 31) / 32), (unsigned)((h->Qtot + 31) / 32));
    hipLaunchKernelGGL(k_transpose, grid, dim3(256), 0, h->stream, V, ldv, h->q0, h->Ql, h->Qtot, h->dLinv, h->ldl);
    JCDF_HIP(h, hipGetLastError());
    JCDF_HIP(h, hipStreamSynchronize(h->stream));
    h->have_metric = true;
    return JCDF_OK;
}

// ---- divide & conquer plan (static per n): the merges of every level, bottom-up pairing ------------------
constexpr int DC_MFMA_MIN = 96;      // merges of a level whose largest merge has at least this many rows use the MFMA update
struct DcLevel {
    int nm = 0, maxm = 0, has_carry = 0;
    int64_t rows = 0;                // packed rows used by this level
    size_t merge_off = 0;            // index of this level's first DcMerge in the device array
};
struct DcPlan {
    std::vector<DcLevel> levels;
    DcMerge *d_merges = nullptr;
    int64_t max_rows = 0;
    int max_nm = 0;
};
std::mutex g_dc_mutex;
std::map<std::pair<int, int64_t>, DcPlan> g_dc_plans;       // (device, n)

constexpr int DC_PREPARE_LDS_MAX = 96 * 1024;            // k_dc_prepare's dynamic LDS (36 bytes per row of a merge)

const DcPlan *dc_plan(int64_t n)
{
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lock(g_dc_mutex);
    auto it = g_dc_plans.find({dev, n});
    if (it != g_dc_plans.end()) return &it->second;
    DcPlan plan;
    std::vector<DcMerge> all;
    std::vector<std::pair<int, int>> blocks;                // (start, size)
    for (int i = 0; i < (int)n; i += DC_LEAF) blocks.push_back({i, std::min<int>(DC_LEAF, (int)n - i)});   // leaves: k_dc_leaf
    while (blocks.size() > 1) {
        DcLevel lv;
        lv.merge_off = all.size();
        std::vector<std::pair<int, int>> next;
        for (size_t b = 0; b + 1 < blocks.size(); b += 2) lv.maxm = std::max(lv.maxm, blocks[b].second + blocks[b + 1].second);
        const bool pad = lv.maxm >= DC_MFMA_MIN;            // MFMA update: whole 16-row k stages per merge
        int64_t rows = 0;
        for (size_t b = 0; b + 1 < blocks.size(); b += 2) {
            DcMerge mg{blocks[b].first, blocks[b].second, blocks[b + 1].second, (int)rows};
            const int m = mg.n1 + mg.n2;
            rows += pad ? roundup(m, 16) : m;
            all.push_back(mg);
            lv.nm++;
            next.push_back({mg.s, m});
        }
        if (blocks.size() % 2) {                                // odd block out: carried to the next level unchanged;
            lv.has_carry = 1;                                   // its descriptor sits behind the level's merges
            all.push_back(DcMerge{blocks.back().first, blocks.back().second, 0, 0});
            next.push_back(blocks.back());
        }
        lv.rows = rows;
        plan.max_rows = std::max(plan.max_rows, rows);
        plan.max_nm = std::max(plan.max_nm, lv.nm);
        plan.levels.push_back(lv);
        blocks.swap(next);
    }
    if (!all.empty()) {
        if (hipMalloc((void **)&plan.d_merges, all.size() * sizeof(DcMerge)) != hipSuccess) return nullptr;
        if (hipMemcpy(plan.d_merges, all.data(), all.size() * sizeof(DcMerge), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
    }
    auto res = g_dc_plans.emplace(std::make_pair(dev, n), std::move(plan));
    return &res.first->second;
}

struct DcWork {                      // carve-up of the caller's workspace
    double *w2, *Zb, *X, *Zp, *G, *dl, *zl, *zhat, *mu, *defval, *rho, *sc;
    int *col, *defcol, *org, *K, *info;
    int64_t ldx, ldzb;
    int64_t bytes;
};
DcWork dc_carve(char *base, int64_t n, const DcPlan *plan)
{
    DcWork wk{};
    int64_t off = 0;
    auto take = [&](int64_t bytes) { char *p = base ? base + off : nullptr; off += (bytes + 255) / 256 * 256; return p; };
    const int64_t rows = plan->max_rows + 16;
    wk.ldx = roundup(n, 64) + 64;
    wk.ldzb = n;
    wk.info = (int *)take(256);                      // byte offset 0 of the workspace: non-zero = a leaf's QL iteration did not converge
    wk.w2 = (double *)take(n * 8);
    wk.Zb = (double *)take(n * n * 8);
    wk.X = (double *)take(rows * wk.ldx * 8);
    wk.Zp = (double *)take(rows * wk.ldx * 8);
    wk.G = (double *)take(rows * wk.ldx * 8);
    wk.dl = (double *)take(rows * 8);
    wk.zl = (double *)take(rows * 8);
    wk.zhat = (double *)take(rows * 8);
    wk.mu = (double *)take(rows * 8);
    wk.defval = (double *)take(rows * 8);
    wk.rho = (double *)take((plan->max_nm + 1) * 8);
    wk.sc = (double *)take(16);
    wk.col = (int *)take(rows * 4);
    wk.defcol = (int *)take(rows * 4);
    wk.org = (int *)take(rows * 4);
    wk.K = (int *)take((plan->max_nm + 1) * 4);
    wk.bytes = off;
    return wk;
}

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE attribute: set for every fixed-size kernel of the
// library each time a handle is created on a device (and by the stand-alone entry points on their current device).
hipError_t set_device_kernel_attributes()
{
    hipError_t first = hipSuccess;
    auto set = [&](const void *f, int bytes) {
        hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (first == hipSuccess && e != hipSuccess) first = e;
    };
    set((const void *)k_exchange_K64, K64_SMEM_BYTES);
    set((const void *)k_metric_apply, GemmNT<MCfg>::SMEM_BYTES);
    set((const void *)k_coulomb_J, 150 * 1024);
    set((const void *)k_chol_diag, CHOL_DIAG_LDS);
    set((const void *)k_q_tail_reflect, ((SYTD2_TAIL_T - 2) * SYTD2_TAIL_T + SYTD2_TAIL_T) * 8);
    set((const void *)k_sytd2_tail, SYTD2_TAIL_T * SYTD2_TAIL_T * 8);
    set((const void *)k_chol_inv_gemm, C64Cfg::SMEM_BYTES);
    set((const void *)k_chol_syrk, C128Cfg::SMEM_BYTES);
    set((const void *)k_dc_update_mfma<DcCfg>, DcCfg::SMEM_BYTES);
    set((const void *)k_dc_update_mfma<DcCfg32>, DcCfg32::SMEM_BYTES);
    set((const void *)k_dc_prepare, DC_PREPARE_LDS_MAX);
    set((const void *)k_blas_gemm_tn<BlasTNCfg>, BlasTNCfg::SMEM_BYTES);
    set((const void *)k_blas_gemm_tn<BlasTN64Cfg>, BlasTN64Cfg::SMEM_BYTES);
    set((const void *)k_ns_gemm, BlasTNCfg::SMEM_BYTES);
    set((const void *)k_sp2_fused<Sp2Cfg>, Sp2Cfg::SMEM_BYTES);
    set((const void *)k_sp2_fused<Sp2Cfg64>, Sp2Cfg64::SMEM_BYTES);
    set((const void *)k_blas_gemm_nt, GemmNT<BlasNTCfg>::SMEM_BYTES);
    set((const void *)k_wy_update, BlasTNCfg::SMEM_BYTES);
    set((const void *)k_wy_U, BlasTNCfg::SMEM_BYTES);
    set((const void *)k_wy_S, GemmNT<BlasNTCfg>::SMEM_BYTES);
    set((const void *)k_wy_T, WY_NB * (WY_NB + 1) * 8);
    return first;
}

// once per device of this process (jcdf_create and the stand-alone entry points, which run on the current device)
hipError_t ensure_device_attributes()
{
    static std::mutex mtx;
    static std::vector<int> done;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lock(mtx);
    if (std::find(done.begin(), done.end(), dev) != done.end()) return hipSuccess;
    e = set_device_kernel_attributes();
    if (e == hipSuccess) done.push_back(dev);
    return e;
}

}  // namespace

// ---- launch of a PERSISTENT kernel (every workgroup spins on hand-offs from its peers: all G must be co-resident) ----------
// First line: the capacity is checked, not assumed — hipOccupancyMaxActiveBlocksPerMultiprocessor x CUs >= G for this kernel
// with this block size and LDS (cached per device / kernel / shape) — and the launch goes through
// hipLaunchCooperativeKernel, the runtime's own co-residency contract (mode 2: always; mode 1, the default: when the grid is
// larger than half the chip), or a plain launch after the same check (mode 0).  Second line: the bounded spins inside the
// kernels (50 ms, error word).
static int g_persistent_mode = 1;
namespace {
struct OccKey {
    int dev; const void *fn; int threads; size_t lds;
    bool operator<(const OccKey &o) const { return std::tie(dev, fn, threads, lds) < std::tie(o.dev, o.fn, o.threads, o.lds); }
};
std::mutex g_occ_mutex;
std::map<OccKey, int> g_occ_cache;          // -> co-resident workgroups on the whole device
std::map<int, int> g_coop_ok;               // device -> hipDeviceAttributeCooperativeLaunch

template <class... KArgs, class... Args>
int32_t launch_persistent(void (*kernel)(KArgs...), int G, int threads, size_t lds, hipStream_t st, Args... args)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return JCDF_ERR_HIP;
    int capacity = 0, coop = 0;
    {
        std::lock_guard<std::mutex> lock(g_occ_mutex);
        const OccKey key{dev, (const void *)kernel, threads, lds};
        auto it = g_occ_cache.find(key);
        if (it == g_occ_cache.end()) {
            int per_cu = 0, cus = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)kernel, threads, lds) != hipSuccess ||
                hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
                return JCDF_ERR_HIP;
            it = g_occ_cache.emplace(key, per_cu * cus).first;
        }
        capacity = it->second;
        auto ic = g_coop_ok.find(dev);
        if (ic == g_coop_ok.end()) {
            int v = 0;
            if (hipDeviceGetAttribute(&v, hipDeviceAttributeCooperativeLaunch, dev) != hipSuccess) v = 0;
            ic = g_coop_ok.emplace(dev, v).first;
        }
        coop = ic->second;
    }
    if (capacity < G) return JCDF_ERR_INVALID;               // the grid can never be co-resident on this device: refuse, do not spin
    std::tuple<KArgs...> held(args...);                       // exact parameter types, addressable
    // mode 1: cooperative where two such kernels (another stream, another rank on the same card) could NOT be resident together —
    // G above half the CUs — so that they queue up behind each other instead of each holding part of the chip and spinning;
    // smaller grids fit side by side and keep the plain launch (the cooperative one costs ~50 us per eigensolve, measured)
    int cus = 0;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (coop && (g_persistent_mode == 2 || (g_persistent_mode == 1 && 2 * G > cus))) {
        void *argv[sizeof...(KArgs)];
        size_t i = 0;
        std::apply([&](auto &...a) { ((argv[i++] = (void *)&a), ...); }, held);
        return hipLaunchCooperativeKernel((const void *)kernel, dim3((unsigned)G), dim3((unsigned)threads), argv, (unsigned)lds, st) == hipSuccess
                   ? JCDF_OK : JCDF_ERR_HIP;
    }
    std::apply([&](auto &...a) { hipLaunchKernelGGL(kernel, dim3((unsigned)G), dim3((unsigned)threads), lds, st, a...); }, held);
    return JCDF_OK;
}
}  // namespace

// ============================================================================
extern "C" {

int32_t jcdf_abi_version(void) { return 1002; }

const char *jcdf_last_error(const jcdf_handle *h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int32_t jcdf_create(jcdf_handle **out, int32_t device_id)
{
    if (!out) { g_create_error = "jcdf_create: out == NULL"; return JCDF_ERR_INVALID; }
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
        g_create_error = "jcdf_create: no HIP device available (" +
                         std::string(e != hipSuccess ? hipGetErrorString(e) : "device count 0") +
                         "); libjcdf_hip has no CPU fallback";
        return JCDF_ERR_NO_DEVICE;
    }
    if (device_id < 0 || device_id >= count) {
        g_create_error = "jcdf_create: device_id " + std::to_string(device_id) + " out of range [0," +
                         std::to_string(count) + ")";
        return JCDF_ERR_INVALID;
    }
    jcdf_handle *h = new (std::nothrow) jcdf_handle();
    if (!h) { g_create_error = "jcdf_create: out of host memory"; return JCDF_ERR_ALLOC; }
    h->device = device_id;
    hipDeviceProp_t prop;
    if ((e = hipSetDevice(device_id)) != hipSuccess || (e = hipGetDeviceProperties(&prop, device_id)) != hipSuccess ||
        (e = hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking)) != hipSuccess) {
        g_create_error = std::string("jcdf_create: ") + hipGetErrorString(e);
        delete h;
        return JCDF_ERR_HIP;
    }
    if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0) {
        g_create_error = std::string("jcdf_create: device is ") + prop.gcnArchName +
                         ", this library contains gfx950 (MI355X) code objects only";
        (void)hipStreamDestroy(h->own_stream);
        delete h;
        return JCDF_ERR_NO_DEVICE;
    }
    h->stream = h->own_stream;
    h->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    for (int a = 0; a < 2 && e == hipSuccess; ++a)
        for (int b = 0; b < jcdf_handle::NEV && e == hipSuccess; ++b) e = hipEventCreate(&h->evset[a][b]);
    h->ev_begin = h->evset[0][0];
    h->ev_end = h->evset[0][jcdf_handle::NEV - 1];
    if (e != hipSuccess || (e = hipEventCreate(&h->ev_h2d)) != hipSuccess || (e = hipEventCreate(&h->ev_d2h)) != hipSuccess ||
        (e = ensure_device_attributes()) != hipSuccess) {
        g_create_error = std::string("jcdf_create: ") + hipGetErrorString(e);
        jcdf_destroy(h);
        return JCDF_ERR_HIP;
    }
    *out = h;
    return JCDF_OK;
}

int32_t jcdf_destroy(jcdf_handle *h)
{
    if (!h) return JCDF_OK;
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);
    free_all(h);
    for (int a = 0; a < 2; ++a)
        for (int b = 0; b < jcdf_handle::NEV; ++b)
            if (h->evset[a][b]) (void)hipEventDestroy(h->evset[a][b]);
    if (h->ev_h2d) (void)hipEventDestroy(h->ev_h2d);
    if (h->ev_d2h) (void)hipEventDestroy(h->ev_d2h);
    if (h->side) (void)hipStreamDestroy(h->side);
    if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
    delete h;
    return JCDF_OK;
}

int32_t jcdf_set_stream(jcdf_handle *h, void *stream, int32_t use_own)
{
    if (!h) return JCDF_ERR_INVALID;
    JCDF_HIP(h, hipSetDevice(h->device));
    JCDF_HIP(h, hipStreamSynchronize(h->stream));
    h->stream = use_own ? h->own_stream : (hipStream_t)stream;
    return JCDF_OK;
}

int32_t jcdf_set_tuning(jcdf_handle *h, const char *key, int64_t value)
{
    if (!h) return JCDF_ERR_INVALID;
    if (!key || value < 0) return fail(h, JCDF_ERR_INVALID, "jcdf_set_tuning: NULL key / negative value");
    const std::string k(key);
    if (k == "k_slices_per_xcd") {
        if (value > 64) return fail(h, JCDF_ERR_INVALID, "jcdf_set_tuning: k_slices_per_xcd in 0..64");
        h->tune_k_slices_per_xcd = value;
    } else if (k == "w_chunk_stages") {
        if (value > 1 << 20) return fail(h, JCDF_ERR_INVALID, "jcdf_set_tuning: w_chunk_stages too large");
        h->tune_w_chunk_stages = value;
    } else if (k == "host_cholesky") {
        h->tune_host_cholesky = value != 0;
    } else if (k == "j_workgroups") {
        if (value > 1 << 16) return fail(h, JCDF_ERR_INVALID, "jcdf_set_tuning: j_workgroups too large");
        h->tune_j_workgroups = value;
    } else if (k == "k_first") {
        h->tune_k_first = value != 0;
    } else {
        return fail(h, JCDF_ERR_INVALID, "jcdf_set_tuning: unknown key '" + k + "'");
    }
    return JCDF_OK;
}

int32_t jcdf_set_exchange_screening(jcdf_handle *h, int64_t n_blocks)
{
    if (!h) return JCDF_ERR_INVALID;
    if (n_blocks < 0 || n_blocks > 4096) return fail(h, JCDF_ERR_INVALID, "jcdf_set_exchange_screening: n_blocks in 0..4096");
    h->xs_blocks = n_blocks;
    return JCDF_OK;
}

int32_t jcdf_configure(jcdf_handle *h, int64_t N, int64_t Q_total, int64_t q0, int64_t q1, int64_t n_occ,
                       int64_t P, const int64_t *pq_p, const int64_t *pq_q)
{
    if (!h) return JCDF_ERR_INVALID;
    if (N <= 0 || Q_total <= 0 || q0 < 0 || q1 <= q0 || q1 > Q_total || n_occ <= 0 || n_occ > N || P <= 0)
        return fail(h, JCDF_ERR_INVALID, "jcdf_configure: invalid sizes");
    if ((pq_p == nullptr) != (pq_q == nullptr))
        return fail(h, JCDF_ERR_INVALID, "jcdf_configure: pq_p and pq_q must both be given or both NULL");
    if (!pq_p && P != N * N)
        return fail(h, JCDF_ERR_INVALID, "jcdf_configure: NULL pq map requires P == N*N (dense map)");
    if (P > N * N) return fail(h, JCDF_ERR_INVALID, "jcdf_configure: P > N*N");
    if (N > 46000) return fail(h, JCDF_ERR_INVALID, "jcdf_configure: N too large for 32-bit tile indices");
    if (pq_p) {                                        // the packed set must be symmetric and duplicate-free
        std::vector<uint8_t> seen((size_t)(N * N), 0);
        for (int64_t c = 0; c < P; ++c) {
            const int64_t p = pq_p[c], q = pq_q[c];
            if (p < 0 || p >= N || q < 0 || q >= N)
                return fail(h, JCDF_ERR_INVALID, "jcdf_configure: pq index out of range at packed index " + std::to_string(c));
            if (seen[(size_t)(q + N * p)]) return fail(h, JCDF_ERR_INVALID, "jcdf_configure: duplicate pq pair");
            seen[(size_t)(q + N * p)] = 1;
        }
        for (int64_t c = 0; c < P; ++c)
            if (!seen[(size_t)(pq_p[c] + N * pq_q[c])])
                return fail(h, JCDF_ERR_INVALID, "jcdf_configure: packed pq set is not symmetric");
    }
    JCDF_HIP(h, hipSetDevice(h->device));
    JCDF_HIP(h, hipStreamSynchronize(h->stream));
    free_all(h);

    h->N = N; h->Qtot = Q_total; h->q0 = q0; h->q1 = q1; h->Ql = q1 - q0; h->o = n_occ; h->P = P;
    h->dense_map = (pq_p == nullptr);
    h->Np = roundup(N, TILE_P);
    // row length of the packed tensor; a row stride that is a multiple of 4 KB would put the 16 rows of a stage on the
    // same HBM channel phase
    h->ldq = roundup(h->Ql, KC);
    if (h->ldq % 512 == 0) h->ldq += KC;
    if ((size_t)h->ldq * 8 > 150 * 1024) return fail(h, JCDF_ERR_INVALID, "jcdf_configure: aux shard too long for the Coulomb kernel's LDS copy of V (use more shards)");
    // orbital (M) tiling of the W kernel: up to 128 orbitals -> one 4-wave workgroup holds them all;
    // more -> 8-wave workgroups of up to 256 orbitals (two wave rows share the staged B tile)
    h->w_rem = 0;
    if (n_occ <= 128 || diag_env("JCDF_W_NO_WVM2")) {
        h->WVMw = 1;
        h->n_mtiles = (int)((n_occ + 127) / 128);
        h->WMw = (int)((((n_occ + h->n_mtiles - 1) / h->n_mtiles) + 15) / 16);
        // 1..3 orbitals past the last full MFMA row tile: VALU FMAs instead of a 16-row tile of padding (DMA kernel,
        // 48 <= n_occ <= 115; C20H42: 81 = 5 x 16 + 1).  JCDF_W_REM=0 pads as before.
        const char *e = diag_env("JCDF_W_REM"), *d = diag_env("JCDF_W_DMA");
        const int64_t r = n_occ % 16;
        if (!(e && atoi(e) == 0) && !(d && atoi(d) == 0) && h->n_mtiles == 1 && r >= 1 && r <= 3 && n_occ / 16 >= 3 && n_occ / 16 <= 7) {
            h->w_rem = (int)r;
            h->WMw = (int)(n_occ / 16);
        }
    } else {
        h->WVMw = 2;
        h->n_mtiles = (int)((n_occ + 255) / 256);
        h->WMw = (int)((((n_occ + h->n_mtiles - 1) / h->n_mtiles) + 31) / 32);    // per wave row
        if (h->WMw < 5) h->WMw = 5;
    }
    h->opad = h->n_mtiles * h->WVMw * h->WMw * 16 + (h->w_rem ? 16 : 0);
    {
        const char *e = diag_env("JCDF_W_DMA");
        h->w_dma = !(e && atoi(e) == 0);
        h->kcw = h->w_dma ? KCD : KC;
        e = diag_env("JCDF_W_SKIP_PARTIAL");
        h->w_skip_partial = !(e && atoi(e) == 0);
        e = diag_env("JCDF_W_ABLATE");
        h->w_ablate = e ? atoi(e) : 0;
        // 256-wide aux tiles (each staged C row feeds twice the MFMAs) measured 25 % SLOWER up to 96 orbitals (C20H42
        // shape 2.16 vs 1.70 ms: 255 VGPRs, two waves per SIMD): only on request
        e = diag_env("JCDF_W_TQ");
        h->tq = (h->w_dma && h->WVMw == 1 && h->WMw <= 6 && h->ldq > TILE_Q && e && atoi(e) == 256) ? 256 : TILE_Q;
    }
    h->n_qt = (int)((h->ldq + h->tq - 1) / h->tq);
    h->vld = h->n_qt * h->tq;
    JCDF_HIP(h, launch_W(h, nullptr, true));

    // ---- stage table of the W kernel + the Coulomb kernel's row list, from the packed pair list ----------------
    // For every p the kept (c, q) in packed order, in stages of KC slots; slots past K_p point at the zero row of C
    // (and at any valid row of B).  The pattern does not depend on the aux index, so this is built once.
    std::vector<int> stg_c, stg_q, stg_p, wchunk, jrow, cmap;
    const int64_t kcw = h->kcw;
    try {
        std::vector<int64_t> pstart((size_t)N + 1, 0);
        std::vector<int> order;                       // packed indices grouped by p (stable); empty: already grouped
        if (pq_p) {
            bool grouped = true;
            for (int64_t c = 0; c < P; ++c) {
                pstart[(size_t)pq_p[c] + 1]++;
                if (c && pq_p[c] < pq_p[c - 1]) grouped = false;
            }
            for (int64_t p = 0; p < N; ++p) pstart[(size_t)p + 1] += pstart[(size_t)p];
            if (!grouped) {
                order.resize((size_t)P);
                std::vector<int64_t> fill(pstart.begin(), pstart.end() - 1);
                for (int64_t c = 0; c < P; ++c) order[(size_t)fill[(size_t)pq_p[c]]++] = (int)c;
            }
        } else {
            for (int64_t p = 0; p <= N; ++p) pstart[(size_t)p] = p * N;
        }
        cmap.assign((size_t)(N * N), -1);
        jrow.reserve((size_t)(P / 2 + N));
        int64_t total_stages = 0;
        for (int64_t p = 0; p < N; ++p) total_stages += std::max<int64_t>(1, (pstart[(size_t)p + 1] - pstart[(size_t)p] + kcw - 1) / kcw);
        stg_c.reserve((size_t)total_stages * kcw);
        stg_q.reserve((size_t)total_stages * kcw);
        stg_p.reserve((size_t)total_stages);
        // Chunks of consecutive p closed greedily at a target stage count: ~6 workgroups per resident slot (2 per CU), at
        // least 256 contraction rows each — a workgroup's pipeline fill is paid once per chunk (C20H42 shape: 64 stages
        // per chunk 1.76 ms, 128: 1.66, 512: 1.64), and many chunks let the dispatcher even out unequal K_p.  Measured
        // against it on one box (gpurun_out/r02_pf15.txt) and not better: exactly one chunk per resident slot (C20H42
        // 1.60-1.68 vs 1.57-1.61 ms; (H2O)50 13 % kept 9.5 vs 9.5) and equalised chunk sizes (1.67-1.70: with 64 stages per
        // p the greedy rule happens to give 170 equal chunks of 3 p, the equalised one 2 and 3 p mixed).  The slots stay
        // busy either way: sum of workgroup times / slots = kernel time (profiles/r02_w_stall.txt).
        const int64_t tiles = total_stages * h->n_qt * h->n_mtiles;
        int64_t target = std::min<int64_t>(4096 / kcw, std::max<int64_t>(256 / kcw, tiles / (6 * 2 * (int64_t)h->num_cu)));
        if (h->tune_w_chunk_stages > 0) target = h->tune_w_chunk_stages;
        wchunk.push_back(0);
        int64_t in_chunk = 0;
        for (int64_t p = 0; p < N; ++p) {
            const int64_t k0 = pstart[(size_t)p], Kp = pstart[(size_t)p + 1] - k0;
            const int64_t ns = std::max<int64_t>(1, (Kp + kcw - 1) / kcw);
            int last_c = 0;
            for (int64_t j = 0; j < ns * kcw; ++j) {
                if (j < Kp) {
                    const int64_t c = order.empty() ? k0 + j : order[(size_t)(k0 + j)];
                    const int64_t q = pq_q ? pq_q[c] : j;
                    last_c = (int)c;
                    stg_c.push_back((int)c);
                    stg_q.push_back((int)q);
                    if (q >= p) { cmap[(size_t)(q + N * p)] = (int)jrow.size(); jrow.push_back((int)c); }
                } else {
                    stg_c.push_back(last_c);
                    stg_q.push_back((int)N);
                }
            }
            for (int64_t t = 0; t < ns; ++t) stg_p.push_back(t == ns - 1 ? (int)p : -1);
            in_chunk += ns;
            if (in_chunk >= target || p == N - 1) { wchunk.push_back((int)stg_p.size()); in_chunk = 0; }
        }
    } catch (...) {
        return fail(h, JCDF_ERR_ALLOC, "jcdf_configure: out of host memory for the stage table");
    }
    h->n_stages = (int)stg_p.size();
    h->n_chunks = (int)wchunk.size() - 1;
    h->Plow = (int64_t)jrow.size();

    // ---- K kernel: the needed 64 x 64 blocks of the lower triangle, packed four to a workgroup ------------------------
    // Exchange screening (jcdf_set_exchange_screening; calculate_exchange_block_screen_matrix, ScreenedDF.jl:385-457): the
    // reference's K blocks of width N / n_blocks (one block below N = 100); a block is kept when it holds a kept pair; the
    // ragged strip q >= n_blocks * width is always computed (ScreenedDF.jl:518-545).
    std::vector<int> kgroups, kblk;
    std::vector<unsigned char> bscr;
    h->xs_nb = h->xs_width = 0;
    try {
        const int Nb = (int)((N + 63) / 64);
        if (h->xs_blocks > 0) {
            h->xs_nb = N < 100 ? 1 : (int)std::min<int64_t>(h->xs_blocks, N);
            h->xs_width = N < 100 ? (int)N : (int)(N / h->xs_nb);
            bscr.assign((size_t)h->xs_nb * h->xs_nb, 0);
            for (int64_t c = 0; c < P; ++c) {
                const int64_t p = pq_p ? pq_p[c] : c / N, q = pq_q ? pq_q[c] : c % N;
                if (q < p) continue;
                const int64_t bq = q / h->xs_width, bp = p / h->xs_width;
                if (bq < h->xs_nb && bp < h->xs_nb) bscr[(size_t)(bq * h->xs_nb + bp)] = 1;
            }
        }
        auto needed = [&](int bi, int bj) {                  // does block (bi >= bj) hold an element whose K is kept?
            if (bscr.empty()) return true;
            const int q0 = bi * 64, q1 = std::min<int>((int)N, q0 + 64) - 1, p0 = bj * 64, p1 = std::min<int>((int)N, p0 + 64) - 1;
            if (q1 >= h->xs_nb * h->xs_width) return true;   // reaches into the strip
            for (int bq = q0 / h->xs_width; bq <= q1 / h->xs_width; ++bq)
                for (int bp = p0 / h->xs_width; bp <= std::min(p1 / h->xs_width, bq); ++bp)
                    if (bscr[(size_t)bq * h->xs_nb + bp]) return true;
            return false;
        };
        kblk.assign((size_t)Nb * (Nb + 1) / 2, -1);
        int nblk = 0;
        // a group: up to 4 blocks over at most 4 distinct row blocks
        std::vector<int> rows, blocks;                       // pending group: row blocks, (bi, bj) pairs
        auto flush = [&]() {
            if (blocks.empty()) return;
            int d[16];
            for (int k = 0; k < 4; ++k) d[k] = k < (int)rows.size() ? rows[k] : -1;
            for (int w = 0; w < 4; ++w) {
                if (2 * w < (int)blocks.size()) {
                    const int bi = blocks[2 * w], bj = blocks[2 * w + 1];
                    d[4 + 3 * w] = (int)(std::find(rows.begin(), rows.end(), bi) - rows.begin());
                    d[5 + 3 * w] = (int)(std::find(rows.begin(), rows.end(), bj) - rows.begin());
                    d[6 + 3 * w] = kblk[(size_t)bi * (bi + 1) / 2 + bj];
                } else {
                    d[4 + 3 * w] = d[5 + 3 * w] = 0;
                    d[6 + 3 * w] = -1;
                }
            }
            kgroups.insert(kgroups.end(), d, d + 16);
            rows.clear();
            blocks.clear();
        };
        auto add = [&](int bi, int bj) {
            std::vector<int> nr = rows;
            if (std::find(nr.begin(), nr.end(), bi) == nr.end()) nr.push_back(bi);
            if (std::find(nr.begin(), nr.end(), bj) == nr.end()) nr.push_back(bj);
            if (blocks.size() >= 8 || nr.size() > 4) {
                flush();
                nr.clear();
                nr.push_back(bi);
                if (bj != bi) nr.push_back(bj);
            }
            rows = nr;
            kblk[(size_t)bi * (bi + 1) / 2 + bj] = nblk++;
            blocks.push_back(bi);
            blocks.push_back(bj);
        };
        const int nT = (Nb + 1) / 2;
        std::vector<std::pair<int, int>> part;               // blocks of 128-tiles that are not a full 2 x 2 group
        for (int ti = 0; ti < nT; ++ti)
            for (int tj = 0; tj <= ti; ++tj) {
                std::vector<std::pair<int, int>> tb;
                for (int a = 0; a < 2; ++a)
                    for (int c = 0; c < 2; ++c) {
                        const int bi = 2 * ti + a, bj = 2 * tj + c;
                        if (bi < Nb && bj <= bi && needed(bi, bj)) tb.push_back({bi, bj});
                    }
                if (tb.size() == 4) {                        // full off-diagonal tile: its own group
                    flush();
                    for (auto &x : tb) add(x.first, x.second);
                    flush();
                } else {
                    part.insert(part.end(), tb.begin(), tb.end());
                }
            }
        // the partial tiles (diagonal ones, screened ones, the half tile of an odd block count) in row-major block order:
        // neighbours share row blocks
        std::sort(part.begin(), part.end());
        for (auto &x : part) add(x.first, x.second);
        flush();
        h->nblk64 = nblk;
        h->ngroups = (int)(kgroups.size() / 16);
        if (h->ngroups == 0) return fail(h, JCDF_ERR_INVALID, "jcdf_configure: exchange screening left no K block (empty pair list?)");
    } catch (...) {
        return fail(h, JCDF_ERR_ALLOC, "jcdf_configure: out of host memory for the K block list");
    }
    const int64_t Ktot = h->o * h->ldq;
    // Split-K: all tiles of one k-slice run on one XCD (they share W rows through that L2), so the slice count is a
    // multiple of 8: m slices per XCD.  Measured (tools/prof_fock.py, JCDF_K_SLICES_PER_XCD): when the lower triangle has
    // at most one tile per CU of an XCD, the best m is the largest for which all ngroups * m workgroups are resident at once
    // (2 per CU; N = 510: m = 6 0.91 ms, m = 8 1.18, m = 16 1.18 — a partial second round costs more than it balances);
    // with more tiles than that a single round leaves some CUs with two workgroups and the rest with one (N = 956:
    // m = 1 9.2 ms), and many short rounds balance better (m = 8 6.7 ms; N = 1250: 19.9 -> 18.3 ms).  Model: busiest
    // CU's workgroup count / m; multi-round forms are considered only when one slice per XCD is already more than one
    // workgroup per CU (N = 896, 28 tiles: m = 2 in one round 2.4 ms, m = 8 2.8 ms) and taken when the model says they win
    // (N = 1915, 120 tiles: m = 1 66 ms, m = 4 60 ms).
    const int64_t max_chunks = std::max<int64_t>(1, Ktot / (4 * KC));      // >= 4 LDS stages per slice
    static const double K_MULTI_ROUND_CHARGE = [] { const char *e = diag_env("JCDF_K_MULTI_CHARGE"); return e ? atof(e) : 1.0; }();
    const int64_t cus_per_xcd = std::max(1, h->num_cu / 8);
    const int64_t slots_per_xcd = 2 * cus_per_xcd;
    int64_t per_xcd = std::max<int64_t>(1, slots_per_xcd / h->ngroups);
    {
        const int64_t w1 = h->ngroups * per_xcd;                              // single round
        const double single = (double)((w1 + cus_per_xcd - 1) / cus_per_xcd) / (double)per_xcd;
        double best = 1e300;
        int64_t best_m = 0;
        for (int64_t m = 4; m <= 12 && h->ngroups > cus_per_xcd; ++m) {
            if (h->ngroups * m <= slots_per_xcd || 8 * m > max_chunks || 8 * m * (int64_t)h->nblk64 * 64 * 64 * 8 > (int64_t)512 << 20) continue;
            const double cost = K_MULTI_ROUND_CHARGE * (double)((h->ngroups * m + cus_per_xcd - 1) / cus_per_xcd) / (double)m;
            if (cost <= best) { best = cost; best_m = m; }                 // ties: the finer split
        }
        if (best_m && best < 0.99 * single) per_xcd = best_m;
        if (h->tune_k_slices_per_xcd > 0) per_xcd = h->tune_k_slices_per_xcd;
    }
    int64_t S = 8 * per_xcd;
    if (S > max_chunks) S = std::max<int64_t>(1, max_chunks);
    h->KS = (int)roundup((Ktot + S - 1) / S, KC);
    h->S = (int)((Ktot + h->KS - 1) / h->KS);
    h->Wld = (int64_t)h->S * h->KS;

    int32_t rc;
    // KC rows + one aux tile of slack: a partial last aux tile reads past the end of its rows
    if ((rc = dev_alloc(h, &h->dB, (h->P + KC) * h->ldq + 4 * TILE_Q, true))) return rc;
    if ((rc = dev_alloc(h, &h->dCpad, (h->Np + KC) * h->opad, true))) return rc;
    if ((rc = dev_alloc(h, &h->dCv, N * h->opad, true))) return rc;
    if ((rc = dev_alloc(h, &h->dWt, h->Np * h->Wld, true))) return rc;
    if ((rc = dev_alloc(h, &h->dVpart, (int64_t)h->n_chunks * h->n_mtiles * h->vld, true))) return rc;
    if ((rc = dev_alloc(h, &h->dV, h->ldq, true))) return rc;
    if ((rc = dev_alloc(h, &h->dJ, h->Plow, true))) return rc;
    if ((rc = dev_alloc(h, &h->dKslab, (int64_t)h->S * h->nblk64 * 64 * 64, true))) return rc;
    if ((rc = dev_alloc(h, &h->dH, N * N, true))) return rc;
    if ((rc = dev_alloc(h, &h->dF, N * N, true))) return rc;
    if ((rc = dev_alloc(h, &h->dC, N * n_occ, true))) return rc;
    if ((rc = dev_upload(h, &h->dWchunk, wchunk))) return rc;
    if ((rc = dev_upload(h, &h->dStgC, stg_c))) return rc;
    if ((rc = dev_upload(h, &h->dStgQ, stg_q))) return rc;
    if ((rc = dev_upload(h, &h->dStgP, stg_p))) return rc;
    if ((rc = dev_upload(h, &h->dJrow, jrow))) return rc;
    if ((rc = dev_upload(h, &h->dCmap, cmap))) return rc;
    if ((rc = dev_upload(h, &h->dKgroups, kgroups))) return rc;
    if ((rc = dev_upload(h, &h->dKblk, kblk))) return rc;
    if (!bscr.empty() && (rc = dev_upload(h, &h->dBscr, bscr))) return rc;
    if (h->w_ablate == 32 && (rc = dev_alloc(h, &h->dStall, (int64_t)h->n_chunks * h->n_mtiles * h->n_qt * 8 * 6, true))) return rc;
    JCDF_HIP(h, hipStreamSynchronize(h->stream));       // the host vectors above go out of scope
    h->configured = true;
    return JCDF_OK;
}

int32_t jcdf_set_metric_inverse(jcdf_handle *h, const double *Linv)
{
    if (!h) return JCDF_ERR_INVALID;
    if (!h->configured || !Linv) return fail(h, JCDF_ERR_INVALID, "jcdf_set_metric_inverse: configure first / NULL");
    JCDF_HIP(h, hipSetDevice(h->device));
    return upload_linv(h, Linv);
}

int32_t jcdf_set_metric(jcdf_handle *h, const double *J2c)
{
    if (!h) return JCDF_ERR_INVALID;
    if (!h->configured || !J2c) return fail(h, JCDF_ERR_INVALID, "jcdf_set_metric: configure first / NULL");
    JCDF_HIP(h, hipSetDevice(h->device));
    if (h->tune_host_cholesky) {              // host potrf/trtri (the reference's GPUDF.jl:890-891 placement)
        std::vector<double> L;
        try {
            L.assign(J2c, J2c + (size_t)(h->Qtot * h->Qtot));
        } catch (...) {
            return fail(h, JCDF_ERR_ALLOC, "jcdf_set_metric: out of host memory");
        }
        const int info = hostlapack::potrf_trtri_lower(L.data(), h->Qtot);
        if (info != 0)
            return fail(h, JCDF_ERR_NOT_SPD, "jcdf_set_metric: (P|Q) not positive definite at pivot " + std::to_string(info));
        return upload_linv(h, L.data());
    }
    CholBuffers w;                                   // device potrf/trtri (DenseGPUDF.jl:185-193 placement)
    int info = 0;
    hipError_t e = chol_inverse_device(h->stream, J2c, h->Qtot, w, &info);
    if (e == hipErrorOutOfMemory) return fail(h, JCDF_ERR_ALLOC, "jcdf_set_metric: out of device memory for the factorisation");
    if (e != hipSuccess) return fail(h, JCDF_ERR_HIP, std::string("jcdf_set_metric: ") + hipGetErrorString(e));
    if (info != 0)
        return fail(h, JCDF_ERR_NOT_SPD, "jcdf_set_metric: (P|Q) not positive definite at pivot " + std::to_string(info));
    return upload_linv_from_device(h, w.V, w.ld);
}

int32_t jcdf_device_potrf_trtri(int32_t device_id, double *A, int64_t n)
{
    if (!A || n <= 0) return JCDF_ERR_INVALID;
    if (hipSetDevice(device_id) != hipSuccess) return JCDF_ERR_NO_DEVICE;
    CholBuffers w;
    int info = 0;
    hipError_t e = chol_inverse_device(nullptr, A, n, w, &info);
    if (e == hipErrorOutOfMemory) return JCDF_ERR_ALLOC;
    if (e != hipSuccess) return JCDF_ERR_HIP;
    if (info != 0) return JCDF_ERR_NOT_SPD;
    // column-major lower L^-1[q + n s] = V[s][q]; the strict upper triangle of V's rows is what is
    // non-zero, so the returned matrix has an exactly-zero upper triangle (like the reference's trtri + zeroing)
    if (hipMemcpy2D(A, (size_t)n * 8, w.V, (size_t)w.ld * 8, (size_t)n * 8, (size_t)n, hipMemcpyDeviceToHost) != hipSuccess)
        return JCDF_ERR_HIP;
    return JCDF_OK;
}

int32_t jcdf_push_three_center(jcdf_handle *h, int64_t s0, int64_t s1, const double *T)
{
    if (!h) return JCDF_ERR_INVALID;
    if (!h->configured || !h->have_metric) return fail(h, JCDF_ERR_INVALID, "jcdf_push_three_center: configure + set_metric first");
    if (!T || s0 < 0 || s1 <= s0 || s1 > h->Qtot) return fail(h, JCDF_ERR_INVALID, "jcdf_push_three_center: bad row range");
    JCDF_HIP(h, hipSetDevice(h->device));
    if (!h->pushed_any) {
        JCDF_HIP(h, hipMemsetAsync(h->dB, 0, (size_t)((h->P + KC) * h->ldq + 4 * TILE_Q) * 8, h->stream));
        h->pushed_any = true;
    }
    if (s0 >= h->q1) return JCDF_OK;                 // Linv[q0:q1, s0:s1] == 0 (lower triangular)
    int32_t rc = push_block(h, s0, s1, T, false);
    if (rc) return rc;
    h->have_B = true;
    return JCDF_OK;
}

int32_t jcdf_push_three_center_device(jcdf_handle *h, int64_t s0, int64_t s1, const double *d_T)
{
    if (!h) return JCDF_ERR_INVALID;
    if (!h->configured || !h->have_metric) return fail(h, JCDF_ERR_INVALID, "jcdf_push_three_center_device: configure + set_metric first");
    if (!d_T || s0 < 0 || s1 <= s0 || s1 > h->Qtot) return fail(h, JCDF_ERR_INVALID, "jcdf_push_three_center_device: bad row range");
    JCDF_HIP(h, hipSetDevice(h->device));
    if (!h->pushed_any) {
        JCDF_HIP(h, hipMemsetAsync(h->dB, 0, (size_t)((h->P + KC) * h->ldq + 4 * TILE_Q) * 8, h->stream));
        h->pushed_any = true;
    }
    if (s0 >= h->q1) return JCDF_OK;
    int32_t rc = push_block(h, s0, s1, d_T, true);
    if (rc) return rc;
    JCDF_HIP(h, hipStreamSynchronize(h->stream));
    h->have_B = true;
    return JCDF_OK;
}

int32_t jcdf_set_B(jcdf_handle *h, const double *B)
{
    if (!h) return JCDF_ERR_INVALID;
    if (!h->configured || !B) return fail(h, JCDF_ERR_INVALID, "jcdf_set_B: configure first / NULL");
    JCDF_HIP(h, hipSetDevice(h->device));
    // the device layout IS the reference's (Q_d, P) column-major, only the leading dimension is padded
    JCDF_HIP(h, hipMemcpy2DAsync(h->dB, (size_t)h->ldq * 8, B, (size_t)h->Ql * 8, (size_t)h->Ql * 8, (size_t)h->P,
                                 hipMemcpyHostToDevice, h->stream));
    JCDF_HIP(h, hipStreamSynchronize(h->stream));
    h->have_B = true;
    h->pushed_any = true;
    return JCDF_OK;
}

int32_t jcdf_set_B_columns_device(jcdf_handle *h, int64_t c0, int64_t c1, const double *d_B)
{
    if (!h) return JCDF_ERR_INVALID;
    if (!h->configured || !d_B) return fail(h, JCDF_ERR_INVALID, "jcdf_set_B_columns_device: configure first / NULL");
    if (c0 < 0 || c1 <= c0 || c1 > h->P) return fail(h, JCDF_ERR_INVALID, "jcdf_set_B_columns_device: bad packed-index range");
    JCDF_HIP(h, hipSetDevice(h->device));
    JCDF_HIP(h, hipMemcpy2DAsync(h->dB + c0 * h->ldq, (size_t)h->ldq * 8, d_B, (size_t)h->Ql * 8, (size_t)h->Ql * 8,
                                 (size_t)(c1 - c0), hipMemcpyDeviceToDevice, h->stream));
    JCDF_HIP(h, hipStreamSynchronize(h->stream));
    h->have_B = true;
    h->pushed_any = true;
    return JCDF_OK;
}

int32_t jcdf_get_B(jcdf_handle *h, double *B_out)
{
    if (!h) return JCDF_ERR_INVALID;
    if (!h->configured || !h->have_B || !B_out) return fail(h, JCDF_ERR_INVALID, "jcdf_get_B: no B yet / NULL");
    JCDF_HIP(h, hipSetDevice(h->device));
    JCDF_HIP(h, hipMemcpy2DAsync(B_out, (size_t)h->Ql * 8, h->dB, (size_t)h->ldq * 8, (size_t)h->Ql * 8, (size_t)h->P,
                                 hipMemcpyDeviceToHost, h->stream));
    JCDF_HIP(h, hipStreamSynchronize(h->stream));
    return JCDF_OK;
}

int32_t jcdf_set_core_hamiltonian(jcdf_handle *h, const double *H)
{
    if (!h) return JCDF_ERR_INVALID;
    if (!h->configured) return fail(h, JCDF_ERR_INVALID, "jcdf_set_core_hamiltonian: configure first");
    JCDF_HIP(h, hipSetDevice(h->device));
    if (!H) { h->have_H = false; return JCDF_OK; }
    JCDF_HIP(h, hipMemcpyAsync(h->dH, H, (size_t)(h->N * h->N) * 8, hipMemcpyHostToDevice, h->stream));
    JCDF_HIP(h, hipStreamSynchronize(h->stream));
    h->have_H = true;
    return JCDF_OK;
}

int32_t jcdf_fock_build_device(jcdf_handle *h, const double *d_C_occ, double *d_F, void *stream)
{
    if (!h) return JCDF_ERR_INVALID;
    if (!h->configured || !h->have_B) return fail(h, JCDF_ERR_INVALID, "jcdf_fock_build_device: B not set");
    if (!d_C_occ || !d_F) return fail(h, JCDF_ERR_INVALID, "jcdf_fock_build_device: NULL pointer");
    JCDF_HIP(h, hipSetDevice(h->device));
    if (h->stage_doubles) release_stage(h);
    h->timed_host_copy = false;
    return enqueue_fock(h, d_C_occ, h->N, d_F, h->N, stream ? (hipStream_t)stream : h->stream);
}

int32_t jcdf_fock_build_device_ld(jcdf_handle *h, const double *d_C_occ, int64_t ldc, double *d_F, int64_t ldf, void *stream)
{
    if (!h) return JCDF_ERR_INVALID;
    if (!h->configured || !h->have_B) return fail(h, JCDF_ERR_INVALID, "jcdf_fock_build_device_ld: B not set");
    if (!d_C_occ || !d_F || ldc < h->N || ldf < h->N) return fail(h, JCDF_ERR_INVALID, "jcdf_fock_build_device_ld: NULL pointer / leading dimension < N");
    JCDF_HIP(h, hipSetDevice(h->device));
    if (h->stage_doubles) release_stage(h);
    h->timed_host_copy = false;
    return enqueue_fock(h, d_C_occ, ldc, d_F, ldf, stream ? (hipStream_t)stream : h->stream);
}

int32_t jcdf_set_overlap(jcdf_handle *h, int32_t overlap_jk)
{
    if (!h) return JCDF_ERR_INVALID;
    h->overlap_jk = overlap_jk != 0;
    return JCDF_OK;
}

int32_t jcdf_synchronize(jcdf_handle *h, jcdf_timings *t)
{
    if (!h) return JCDF_ERR_INVALID;
    JCDF_HIP(h, hipSetDevice(h->device));
    if (h->pending) {
        JCDF_HIP(h, hipEventSynchronize(h->ev_end));
        h->pending = false;
    }
    JCDF_HIP(h, hipStreamSynchronize(h->stream));
    if (t) {
        std::memset(t, 0, sizeof(*t));
        if (h->recs.size() >= 5) {
            t->non_zero_coeff_time = elapsed_s(h->recs[0].e0, h->recs[0].e1);
            t->W_time = elapsed_s(h->recs[1].e0, h->recs[1].e1);
            t->J_time = elapsed_s(h->recs[2].e0, h->recs[2].e1);
            t->K_time = elapsed_s(h->recs[3].e0, h->recs[3].e1);
            t->copy_J_time = elapsed_s(h->recs[4].e0, h->recs[4].e1);
            t->fock_time = elapsed_s(h->ev_begin, h->ev_end);
            if (h->timed_host_copy)
                t->copy_time = elapsed_s(h->ev_h2d, h->ev_begin) + elapsed_s(h->ev_end, h->ev_d2h);
        }
    }
    return JCDF_OK;
}

int32_t jcdf_fock_build_begin(jcdf_handle *h, const double *C_occ)
{
    if (!h) return JCDF_ERR_INVALID;
    if (!h->configured || !h->have_B) return fail(h, JCDF_ERR_INVALID, "jcdf_fock_build_begin: B not set");
    if (!C_occ) return fail(h, JCDF_ERR_INVALID, "jcdf_fock_build_begin: NULL pointer");
    JCDF_HIP(h, hipSetDevice(h->device));
    if (h->stage_doubles) release_stage(h);
    JCDF_HIP(h, hipEventRecord(h->ev_h2d, h->stream));
    // pageable host memory: the copy has left the caller's buffer when this returns
    JCDF_HIP(h, hipMemcpyAsync(h->dC, C_occ, (size_t)(h->N * h->o) * 8, hipMemcpyHostToDevice, h->stream));
    return enqueue_fock(h, h->dC, h->N, h->dF, h->N, h->stream);
}

int32_t jcdf_fock_build_finish(jcdf_handle *h, double *F_out, jcdf_timings *t)
{
    if (!h) return JCDF_ERR_INVALID;
    if (!h->configured || !F_out) return fail(h, JCDF_ERR_INVALID, "jcdf_fock_build_finish: not configured / NULL");
    JCDF_HIP(h, hipSetDevice(h->device));
    JCDF_HIP(h, hipMemcpyAsync(F_out, h->dF, (size_t)(h->N * h->N) * 8, hipMemcpyDeviceToHost, h->stream));
    JCDF_HIP(h, hipEventRecord(h->ev_d2h, h->stream));
    JCDF_HIP(h, hipStreamSynchronize(h->stream));
    h->timed_host_copy = true;
    return jcdf_synchronize(h, t);
}

int32_t jcdf_fock_build(jcdf_handle *h, const double *C_occ, double *F_out, jcdf_timings *t)
{
    if (!h) return JCDF_ERR_INVALID;
    if (!F_out) return fail(h, JCDF_ERR_INVALID, "jcdf_fock_build: NULL pointer");
    const int32_t rc = jcdf_fock_build_begin(h, C_occ);
    if (rc) return rc;
    return jcdf_fock_build_finish(h, F_out, t);
}

int32_t jcdf_get_V(jcdf_handle *h, double *V_out)
{
    if (!h) return JCDF_ERR_INVALID;
    if (!h->configured || !V_out) return fail(h, JCDF_ERR_INVALID, "jcdf_get_V: not configured / NULL");
    JCDF_HIP(h, hipSetDevice(h->device));
    JCDF_HIP(h, hipStreamSynchronize(h->stream));
    JCDF_HIP(h, hipMemcpy(V_out, h->dV, (size_t)h->Ql * 8, hipMemcpyDeviceToHost));
    return JCDF_OK;
}

int32_t jcdf_get_W(jcdf_handle *h, double *W_out)
{
    if (!h) return JCDF_ERR_INVALID;
    if (!h->configured || !W_out) return fail(h, JCDF_ERR_INVALID, "jcdf_get_W: not configured / NULL");
    JCDF_HIP(h, hipSetDevice(h->device));
    const int64_t total = h->Ql * h->o * h->N;
    double *tmp = nullptr;
    int32_t rc = dev_alloc(h, &tmp, total, false);
    if (rc) return rc;
    hipLaunchKernelGGL(k_export_W, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, h->stream, h->dWt, h->Wld, h->ldq,
                       (int)h->Ql, (int)h->o, (int)h->N, tmp);
    hipError_t e = hipMemcpyAsync(W_out, tmp, (size_t)total * 8, hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    dev_free(h, &tmp, total);
    if (e != hipSuccess) return fail(h, JCDF_ERR_HIP, std::string("jcdf_get_W: ") + hipGetErrorString(e));
    return JCDF_OK;
}

int32_t jcdf_host_potrf_trtri(double *A, int64_t n)
{
    if (!A || n <= 0) return -1;
    return hostlapack::potrf_trtri_lower(A, n);
}

int32_t jcdf_set_persistent_launch_mode(int32_t mode)
{
    if (mode < 0 || mode > 2) return JCDF_ERR_INVALID;
    g_persistent_mode = mode;
    return JCDF_OK;
}

// ---- replicated eigensolve helper (caller side) ----------------------------
static size_t sytrd_lds(int64_t n, int G, bool with_q)
{
    return (size_t)((with_q ? 2 : 1) * ((n + G - 1) / G) * n + 2 * n + 32) * 8;
}

static int sytrd_groups(int64_t n, bool with_q, size_t *lds_bytes)
{
    // fewest workgroups that hold the matrix (and, with_q, the rows of Q) in LDS, but at least the
    // measured sweet spot (tools/eigbench3.py: n = 240 -> 32, n = 510 -> 64 workgroups; flat between 32 and 128)
    int G = n >= 400 ? 64 : (n >= 100 ? 32 : (n >= 32 ? 8 : 1));
    while (sytrd_lds(n, G, with_q) > 150 * 1024 && G < 256) G *= 2;
    const char *env = diag_env("JCDF_SYTRD_G");
    if (env) {
        const int want = atoi(env);
        if (want >= 1 && want <= 256 && sytrd_lds(n, want, with_q) <= 160 * 1024) G = want;
    }
    *lds_bytes = sytrd_lds(n, G, with_q);
    return G;
}

// largest n of the one-exchange kernel k_sytrd_onehop: 8 columns of a workgroup + 5 vectors in LDS (104 n bytes <= 160 KB),
// rows of Q in registers (k_sytrd_onehop<48>: n <= 32 * 48)
constexpr int64_t SYTRD_ONEHOP_MAX_N = 1536;

static int64_t sytrd_granule_bytes(int64_t n)
{
    // err word (64 B header) + granule pairs: v 2(n+1), y 2n (+ 512 spare); 16 B per pair
    return 64 + (int64_t)(2 * (n + 1) + 2 * n + 2 * 256) * 16;
}

// the trailing SYTD2_TAIL_T columns go to the one-workgroup kernel (n = 130: 0.76 -> 0.57 ms per eigensolve, 256: 1.43 -> 1.06,
// 510: 2.69 -> 2.49, 1250: 9.54 -> 9.05); a matrix of up to SYTD2_TAIL_T rows goes there whole
// (diagnostic builds: JCDF_SYTRD_TAIL = smallest n that uses it, 0 = never)
static bool sytrd_use_tail(int64_t n)
{
    static const int64_t min_n = diag_env("JCDF_SYTRD_TAIL") ? atoll(diag_env("JCDF_SYTRD_TAIL")) : 32;
    return min_n > 0 && n >= std::max<int64_t>(min_n, 3);
}

int64_t jcdf_sytrd_workspace_bytes(int64_t n)
{
    if (n <= 0) return 0;
    return roundup(sytrd_granule_bytes(n), 256);          // the granules (zeroed every call)
}

int32_t jcdf_sytrd_q_device(void *stream, int64_t n, double *d_A, int64_t lda, double *d_D, double *d_E, double *d_TAU,
                            double *d_Q, int64_t ldq, void *d_work, int64_t work_bytes)
{
    if (n <= 0 || !d_A || lda < n || !d_D || !d_E || !d_TAU || !d_work || work_bytes < jcdf_sytrd_workspace_bytes(n) || (d_Q && ldq < n))
        return JCDF_ERR_INVALID;
    size_t lds = 0;
    const int G = sytrd_groups(n, d_Q != nullptr, &lds);
    // (n <= SYTRD_ONEHOP_MAX_N runs the one-exchange kernel, whose rows of Q live in registers: the LDS of the two-exchange kernel
    //  only limits the sizes above)
    if (n > SYTRD_ONEHOP_MAX_N && lds > 160 * 1024) return JCDF_ERR_INVALID;   // n too large for LDS residency
    hipStream_t st = (hipStream_t)stream;
    char *w = (char *)d_work;
    int *err = (int *)(w + 8);
    jcdf::u64 *vg = (jcdf::u64 *)(w + 64), *yg = vg + 2 * (n + 2), *hg = yg + 2 * ((n + 1) & ~(int64_t)1);
    // tags restart at 1 every call: all granules (and the error word) are zeroed first
    if (hipMemsetAsync(w, 0, (size_t)sytrd_granule_bytes(n), st) != hipSuccess) return JCDF_ERR_HIP;
    // the last 128 columns in one workgroup (k_sytd2_tail: ~1.7 us per column instead of one chip-wide hand-off of 4.6-7 us each)
    const bool tail = sytrd_use_tail(n);
    const int kstop = tail ? (int)std::max<int64_t>(n - SYTD2_TAIL_T, 0) : (int)n;
    auto finish = [&]() -> int32_t {
        if (tail) {
            if (ensure_device_attributes() != hipSuccess) return JCDF_ERR_HIP;
            hipLaunchKernelGGL(k_sytd2_tail, dim3(1), dim3(SYTD2_TAIL_NT), (size_t)SYTD2_TAIL_T * SYTD2_TAIL_T * 8, st, d_A, (int)lda, (int)n, kstop,
                               d_D, d_E, d_TAU);
            if (d_Q)                                                         // Q[:, kstop:] times the block's reflectors, row-parallel
                hipLaunchKernelGGL(k_q_tail_reflect, dim3((unsigned)((n + Q_TAIL_ROWS - 1) / Q_TAIL_ROWS)), dim3(Q_TAIL_NT),
                                   (size_t)((SYTD2_TAIL_T - 2) * SYTD2_TAIL_T + SYTD2_TAIL_T) * 8, st, d_Q, (int)ldq, (int)n, kstop, (int)n - kstop,
                                   (const double *)d_A, (int)lda, (const double *)d_TAU, kstop == 0 ? 1 : 0);
        }
        return hipGetLastError() == hipSuccess ? JCDF_OK : JCDF_ERR_HIP;
    };
    // one exchange per column (every workgroup forms the reflector itself) wins while the redundant work is small:
    // N = 240: 0.93 vs 1.08 ms, 510: 2.43 vs 2.59, 700: 3.93 vs 3.99, 1000: 6.90 vs 6.50 (tools/sytrd_prof.hip)
    static const int onehop_env = diag_env("JCDF_SYTRD_ONEHOP") ? atoi(diag_env("JCDF_SYTRD_ONEHOP")) : -1;
    // (with every poll of a thread in flight at once, sub_two_n, the one-exchange kernel also wins at n = 700: 3.86 vs 3.98 ms and
    //  956: 5.86 vs 6.11 ms; with 512 threads — one WAVE per column in the fused pass and per row of Q — it wins wherever its
    //  slabs fit the LDS: whole eigensolve n = 1000 6.53 -> 5.26 ms, n = 1250 9.01 (two exchanges) -> 7.66, n = 1500 13.3)
    const bool onehop = onehop_env >= 0 ? onehop_env != 0 : n <= SYTRD_ONEHOP_MAX_N;
    if (kstop == 0) return finish();                                     // the whole matrix is the tail (Q starts as the unit matrix there)
    if (onehop && n <= SYTRD_ONEHOP_MAX_N) {
        int G1 = (int)std::max<int64_t>(n >= 400 ? 64 : (n >= 100 ? 32 : (n >= 32 ? 8 : 1)), (n + 7) / 8);   // <= 8 columns each
        if (const char *e = diag_env("JCDF_SYTRD_G1")) G1 = std::max(G1, std::min(256, atoi(e)));                 // diagnostic builds: more workgroups
        const size_t lds1 = (size_t)(((n + G1 - 1) / G1) * n + 5 * n + 32) * 8;
#define JCDF_ONEHOP(NR)                                                                                                     \
    do {                                                                                                                    \
        if (hipFuncSetAttribute((const void *)k_sytrd_onehop<NR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1) != hipSuccess) \
            return JCDF_ERR_HIP;                                                                                            \
        const int32_t rc_ = launch_persistent(k_sytrd_onehop<NR>, G1, 512, lds1, st, d_A, (int)lda, (int)n, d_D, d_E,       \
                                              d_TAU, vg, yg, hg, err, d_Q, (int)ldq, kstop);                                \
        if (rc_) return rc_;                                                                                                \
    } while (0)
        if (n <= 64) JCDF_ONEHOP(2);
        else if (n <= 256) JCDF_ONEHOP(8);
        else if (n <= 512) JCDF_ONEHOP(16);
        else if (n <= 640) JCDF_ONEHOP(20);
        else if (n <= 1024) JCDF_ONEHOP(32);
        else if (n <= 1280) JCDF_ONEHOP(40);
        else JCDF_ONEHOP(48);
#undef JCDF_ONEHOP
        return finish();
    }
    if (lds > 160 * 1024) return JCDF_ERR_INVALID;                   // (reached below the one-exchange limit only by a diagnostic override)
    if (hipFuncSetAttribute((const void *)k_sytrd_lower, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return JCDF_ERR_HIP;
    if (!d_Q && n > SYTRD_ONEHOP_MAX_N && onehop_env != 0) {
        // Above the one-exchange kernel's size without Q (round 4; the back-transformation is jcdf_ormtr_device then): the
        // two-exchange kernel reduces only the first n - 1536 columns (~7 us each) and leaves the trailing 1536 x 1536 block
        // with every update applied — a symmetric matrix of the size the one-exchange kernel holds (5.7 us per column), which
        // continues on it in place, followed by the one-workgroup tail: n = 1915 13.5 -> 10.6 ms (profiles/r04_eigh_stages.txt).
        const int64_t k1 = n - SYTRD_ONEHOP_MAX_N, n2 = SYTRD_ONEHOP_MAX_N;
        if (int32_t rc1 = launch_persistent(k_sytrd_lower, G, 512, lds, st, d_A, (int)lda, (int)n, d_D, d_E, d_TAU, vg, yg, hg, err,
                                            (double *)nullptr, 0, (int)k1))
            return rc1;
        // fresh granules for the second kernel (tags restart at 1); the error word in the header stays as the first kernel left it
        if (hipMemsetAsync(w + 64, 0, (size_t)sytrd_granule_bytes(n) - 64, st) != hipSuccess) return JCDF_ERR_HIP;
        const int G2 = (int)((n2 + 7) / 8);
        const size_t lds2 = (size_t)(((n2 + G2 - 1) / G2) * n2 + 5 * n2 + 32) * 8;
        if (hipFuncSetAttribute((const void *)k_sytrd_onehop<48>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2) != hipSuccess) return JCDF_ERR_HIP;
        double *A2 = d_A + k1 * (lda + 1);
        jcdf::u64 *vg2 = (jcdf::u64 *)(w + 64), *yg2 = vg2 + 2 * (n2 + 2), *hg2 = yg2 + 2 * ((n2 + 1) & ~(int64_t)1);
        if (int32_t rc2 = launch_persistent(k_sytrd_onehop<48>, G2, 512, lds2, st, A2, (int)lda, (int)n2, d_D + k1, d_E + k1, d_TAU + k1,
                                            vg2, yg2, hg2, err, (double *)nullptr, 0, tail ? (int)(n2 - SYTD2_TAIL_T) : (int)n2))
            return rc2;
        return finish();
    }
    // 512 threads from n = 1000 on: half the dependent polls and half the elements per thread (n = 1250: 9.04 -> 8.69 ms; n = 700: equal)
    if (int32_t rc3 = launch_persistent(k_sytrd_lower, G, n >= 1000 ? 512 : 256, lds, st, d_A, (int)lda, (int)n, d_D, d_E, d_TAU, vg, yg,
                                        hg, err, d_Q, (int)ldq, kstop))
        return rc3;
    return finish();
}

int32_t jcdf_sytrd_device(void *stream, int64_t n, double *d_A, int64_t lda, double *d_D, double *d_E, double *d_TAU,
                          void *d_work, int64_t work_bytes)
{
    return jcdf_sytrd_q_device(stream, n, d_A, lda, d_D, d_E, d_TAU, nullptr, 0, d_work, work_bytes);
}

#ifdef JCDF_DIAGNOSTIC
int32_t jcdf_sytrd_replay_q_device(void *stream, int64_t n, const double *d_A, int64_t lda, const double *d_TAU, double *d_Q, int64_t ldq)
{
    if (n <= 0 || n > 640 || !d_A || lda < n || !d_TAU || !d_Q || ldq < n) return JCDF_ERR_INVALID;
    hipStream_t st = (hipStream_t)stream;
    const unsigned grid = (unsigned)((n + 7) / 8);
#define JCDF_REPLAY(NR) hipLaunchKernelGGL(k_sytrd_replay_q<NR>, dim3(grid), dim3(256), 0, st, d_A, (int)lda, (int)n, d_TAU, d_Q, (int)ldq)
    if (n <= 64) JCDF_REPLAY(1);
    else if (n <= 128) JCDF_REPLAY(2);
    else if (n <= 256) JCDF_REPLAY(4);
    else if (n <= 384) JCDF_REPLAY(6);
    else if (n <= 512) JCDF_REPLAY(8);
    else JCDF_REPLAY(10);
#undef JCDF_REPLAY
    return hipGetLastError() == hipSuccess ? JCDF_OK : JCDF_ERR_HIP;
}

int32_t jcdf_keepalive_device(void *stream, int32_t workgroups, int32_t threads, double microseconds, int32_t mode, int32_t pause,
                              const int32_t *d_stop, double *d_sink)
{
    if (workgroups < 1 || workgroups > 4096 || (threads != 64 && threads != 128 && threads != 256) || microseconds < 0.0 || microseconds > 1.0e6 ||
        mode < 0 || mode > 2 || pause < 0 || pause > 127 || !d_sink)
        return JCDF_ERR_INVALID;
    hipLaunchKernelGGL(k_keepalive, dim3((unsigned)workgroups), dim3((unsigned)threads), 0, (hipStream_t)stream,
                       (unsigned long long)(microseconds * 100.0), (int)mode, (const int *)d_stop, d_sink, (int)pause);
    return hipGetLastError() == hipSuccess ? JCDF_OK : JCDF_ERR_HIP;
}

#endif  // JCDF_DIAGNOSTIC

int64_t jcdf_sytrd_max_n(int32_t with_q)
{
    int64_t n = 64;
    while (sytrd_lds(n + 1, 256, with_q != 0) <= 160 * 1024) ++n;       // the two-exchange kernel (Q rows in LDS)
    return std::max<int64_t>(n, SYTRD_ONEHOP_MAX_N);                     // the one-exchange kernel (Q rows in registers)
}

// ---- back-transformation by blocked compact-WY on the MFMA cores (jcdf_wy.hpp) ----------------------------------
namespace {
struct WyWork {
    double *Vt, *Ut, *Tm, *Sm, *W;
    int64_t nrp, npad, nblk, bytes;
};
WyWork wy_carve(char *base, int64_t n)
{
    WyWork w;
    w.nrp = roundup(n, WY_NB);
    w.npad = roundup(n, 32);
    w.nblk = w.nrp / WY_NB;
    size_t off = 0;
    auto take = [&](size_t doubles) { char *p = base ? base + off : nullptr; off += (size_t)roundup((int64_t)doubles * 8, 256); return (double *)p; };
    w.Vt = take((size_t)w.nrp * w.npad);
    w.Ut = take((size_t)w.nrp * w.npad);
    w.Tm = take((size_t)w.nblk * WY_NB * WY_NB);
    w.Sm = take((size_t)w.nblk * WY_NB * WY_NB);
    w.W = take((size_t)WY_NB * w.npad);
    w.bytes = (int64_t)off;
    return w;
}
}  // namespace

int64_t jcdf_ormtr_workspace_bytes(int64_t n)
{
    if (n <= 0) return 0;
    return wy_carve(nullptr, n).bytes;
}

int32_t jcdf_ormtr_device(void *stream, int64_t n, const double *d_A, int64_t lda, const double *d_TAU, double *d_Ct, int64_t ldc,
                          double *d_Out, int64_t ldo, void *d_work, int64_t work_bytes)
{
    if (n <= 0 || !d_A || lda < n || !d_TAU || !d_Ct || !d_work) return JCDF_ERR_INVALID;
    WyWork w = wy_carve((char *)d_work, n);
    if (work_bytes < w.bytes || ldc < w.npad || (ldc & 1) || (d_Out && ldo < w.npad)) return JCDF_ERR_INVALID;
    if (ensure_device_attributes() != hipSuccess) return JCDF_ERR_HIP;
    hipStream_t st = (hipStream_t)stream;
    const int npad = (int)w.npad;
    hipLaunchKernelGGL(k_wy_extract, dim3((unsigned)((w.nrp * w.npad + 255) / 256)), dim3(256), 0, st, d_A, lda, (int)n, w.Vt, w.npad, (int)w.nrp, npad);
    hipLaunchKernelGGL(k_wy_S, dim3((unsigned)((WY_NB / 32) * (WY_NB / 32)), (unsigned)w.nblk), dim3(BlasNTCfg::NT), GemmNT<BlasNTCfg>::SMEM_BYTES, st,
                       (const double *)w.Vt, w.npad, npad, w.Sm);
    hipLaunchKernelGGL(k_wy_T, dim3((unsigned)w.nblk), dim3(4 * WY_NB), (size_t)WY_NB * (WY_NB + 1) * 8, st, (const double *)w.Sm, d_TAU, (int)n, w.Tm);
    hipLaunchKernelGGL(k_wy_U, dim3((unsigned)((WY_NB / 32) * (npad / 32)), (unsigned)w.nblk), dim3(BlasTNCfg::NT), BlasTNCfg::SMEM_BYTES, st,
                       (const double *)w.Tm, (const double *)w.Vt, w.npad, npad, w.Ut);
    for (int64_t b = w.nblk - 1; b >= 0; --b) {
        const int64_t j0 = b * WY_NB;
        if (j0 >= n - 2) continue;                                   // a block of identity reflectors
        const int64_t kq = j0 / 32 * 32;                             // the block's reflectors are zero above row j0 + 1
        const double *Vb = w.Vt + j0 * w.npad + kq, *Ub = w.Ut + j0 * w.npad + kq;
        // W[m][j] = sum_k Vt_b[m][k] Zt[j][k]
        hipLaunchKernelGGL(k_blas_gemm_nt, dim3((unsigned)((WY_NB / 32) * (npad / 32))), dim3(BlasNTCfg::NT), GemmNT<BlasNTCfg>::SMEM_BYTES, st, Vb,
                           w.npad, (const double *)(d_Ct + kq), ldc, w.W, w.npad, (int)((w.npad - kq) / 16), npad / 32);
        // Zt[j][k] -= sum_m W[m][j] Ut_b[m][k]
        const int n_tn = (int)((w.npad - kq) / 32);
        hipLaunchKernelGGL(k_wy_update, dim3((unsigned)((npad / 32) * n_tn)), dim3(BlasTNCfg::NT), BlasTNCfg::SMEM_BYTES, st, (const double *)w.W, w.npad, Ub,
                           w.npad, d_Ct + kq, ldc, WY_NB / 32, -1.0, n_tn);
    }
    if (d_Out)
        hipLaunchKernelGGL(k_wy_transpose, dim3((unsigned)(npad / 32), (unsigned)(npad / 32)), dim3(256), 0, st, (const double *)d_Ct, ldc, d_Out, ldo, (int)n);
    return hipGetLastError() == hipSuccess ? JCDF_OK : JCDF_ERR_HIP;
}

#ifdef JCDF_DIAGNOSTIC
// ---- two-stage tridiagonalisation (jcdf_sbr.hpp): dense -> band (16) -> tridiagonal, Q accumulated forwards -------------
namespace {
struct Sytrd2Layout {
    int64_t n, tmax, ntile, npanel;
    size_t off_v, off_y, off_t, off_m1, off_ab, off_log, total;
};
Sytrd2Layout sytrd2_layout(int64_t n)
{
    Sytrd2Layout L;
    L.n = n;
    L.tmax = n >= 3 ? (n - 3) / SB + 1 : 1;
    L.ntile = (n + 15) / 16;
    size_t o = 64;
    auto take = [&](size_t doubles) { const size_t at = o; o += (doubles * 8 + 63) / 64 * 64; return at; };
    L.npanel = std::max<int64_t>(1, n / SB);
    L.off_v = take((size_t)L.npanel * n * 16);                   // V and T of every panel (the Q update trails on a side stream)
    L.off_y = take((size_t)n * 16);
    L.off_t = take((size_t)L.npanel * 256);
    L.off_m1 = take((size_t)L.ntile * 256);
    L.off_ab = take((size_t)n * SBW);
    L.off_log = take((size_t)n * L.tmax * 16);
    L.total = o;
    return L;
}
struct Sytrd2Side {
    hipStream_t stream = nullptr;
    std::vector<hipEvent_t> events;
};
std::mutex g_sytrd2_mutex;
std::map<int, Sytrd2Side> g_sytrd2_side;                          // per device
Sytrd2Side *sytrd2_side(size_t nevents)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lock(g_sytrd2_mutex);
    Sytrd2Side &sd = g_sytrd2_side[dev];
    if (!sd.stream && hipStreamCreateWithFlags(&sd.stream, hipStreamNonBlocking) != hipSuccess) return nullptr;
    while (sd.events.size() < nevents) {
        hipEvent_t e;
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
        sd.events.push_back(e);
    }
    return &sd;
}
constexpr int SB2ST_WAVES = 16;
size_t sb2st_lds(int64_t n) { return (size_t)(n + 16) * SBW * 8 + SB2ST_WAVES * 48 * 8 + (size_t)(n + 2) * 4; }
size_t sb2st3_lds(int64_t n)
{
    return (size_t)(n + 16) * SBW * 8 + (size_t)16 * 48 * 8 + (size_t)8 * 4 * SB2ST_MBOX * 8 + (size_t)(n + 2 + 8) * 4;
}
constexpr int SB2ST_PAIRS = 8, SB2ST_RING = 4;
size_t sb2st2_lds(int64_t n)
{
    return (size_t)(n + 16) * SBW * 8 + (size_t)SB2ST_PAIRS * SB2ST_RING * SB2ST_SLOT * 8 + (size_t)(n + 2 + 2 * SB2ST_PAIRS) * 4;
}
}  // namespace

int64_t jcdf_sytrd2_max_n(void)
{
    int64_t n = 64;
    while (sb2st_lds(n + 1) <= 160 * 1024) ++n;
    return std::min<int64_t>(n, 256 * 5 + 16);                  // the panel kernel holds at most 5 rows per thread
}

int64_t jcdf_sytrd2_workspace_bytes(int64_t n)
{
    if (n <= 0) return 0;
    return (int64_t)sytrd2_layout(n).total;
}

int32_t jcdf_sytrd2_device(void *stream, int64_t n, double *d_A, int64_t lda, double *d_D, double *d_E, double *d_Q, int64_t ldq,
                           void *d_work, int64_t work_bytes)
{
    if (n <= 0 || n > jcdf_sytrd2_max_n() || !d_A || lda < n || !d_D || !d_E || !d_Q || ldq < n || !d_work ||
        work_bytes < jcdf_sytrd2_workspace_bytes(n))
        return JCDF_ERR_INVALID;
    if (ensure_device_attributes() != hipSuccess) return JCDF_ERR_HIP;
    hipStream_t st = (hipStream_t)stream;
    const Sytrd2Layout L = sytrd2_layout(n);
    char *w = (char *)d_work;
    int *err = (int *)(w + 8);
    double *V = (double *)(w + L.off_v), *Y = (double *)(w + L.off_y), *T = (double *)(w + L.off_t);
    double *M1p = (double *)(w + L.off_m1), *AB = (double *)(w + L.off_ab), *vlog = (double *)(w + L.off_log);
    if (hipMemsetAsync(w, 0, 64, st) != hipSuccess) return JCDF_ERR_HIP;
    const int ni = (int)n;
    hipLaunchKernelGGL(k_set_identity, dim3((unsigned)((n * n + 255) / 256)), dim3(256), 0, st, d_Q, (int)ldq, ni);
    // Q <- Q (I - V T V^T), panel after panel, needs only the stored V_k, T_k: all of it runs on a side stream of the device
    // BESIDE THE CHASE (one CU, ~1 ms) instead of inside stage 1 (per-panel events cost ~7 us each on the main stream:
    // measured); JCDF_SBR_Q_INLINE=1: right behind each panel's QR on `stream` itself
    static const bool q_inline = diag_env("JCDF_SBR_Q_INLINE") && atoi(diag_env("JCDF_SBR_Q_INLINE")) != 0;
    Sytrd2Side *sd = q_inline ? nullptr : sytrd2_side(2);
    if (!q_inline && !sd) return JCDF_ERR_HIP;
    // per panel k: [QR of panel k] -> Y = A22 V -> [A22 update (and, JCDF_SBR_FUSE=1, the QR of panel k+1 in the same launch)]
    auto Vb = [&](int k) { return V + (size_t)k * n * 16; };
    auto Tb = [&](int k) { return T + (size_t)k * 256; };
    bool hip_ok = true;
    auto q_update = [&](int k, hipStream_t qs) {
        hipLaunchKernelGGL(k_sbr_qupdate, dim3((unsigned)((ni + 15) / 16)), dim3(256), 0, qs, ni, (k + 1) * SB, Vb(k), Tb(k), d_Q, (int)ldq);
    };
    auto after_panel = [&](int k) {                                 // V_k, T_k are final on `st` here
        if (!sd) q_update(k, st);
    };
    auto launch_panel = [&](int k) {
        const int m = ni - (k + 1) * SB;
        // JCDF_SBR_PANEL_512=1: 512 threads above 256 rows (two waves per SIMD; measured slower: 256 registers per thread spill)
        static const bool wide = diag_env("JCDF_SBR_PANEL_512") && atoi(diag_env("JCDF_SBR_PANEL_512")) != 0;
#define JCDF_PANEL(NR, NT_) hipLaunchKernelGGL((k_sbr_panel<NR, NT_>), dim3(1), dim3(NT_), 0, st, d_A, (int)lda, ni, k, Vb(k), Tb(k))
        if (m <= 256) JCDF_PANEL(1, 256);
        else if (wide && m <= 512) JCDF_PANEL(1, 512);
        else if (wide && m <= 1024) JCDF_PANEL(2, 512);
        else if (m <= 512) JCDF_PANEL(2, 256);
        else if (m <= 768) JCDF_PANEL(3, 256);
        else JCDF_PANEL(5, 256);
#undef JCDF_PANEL
        after_panel(k);
    };
    // JCDF_SBR_FUSE=1: the QR of panel k+1 as one more block of panel k's update launch (2 launches per panel instead of 3);
    // measured no faster (n = 510: 38-42 us per fused launch against 22 + 14.5 us), so the plain sequence is the default
    static const bool fuse = diag_env("JCDF_SBR_FUSE") && atoi(diag_env("JCDF_SBR_FUSE")) != 0;
    if (ni - SB >= 2) launch_panel(0);
    for (int k = 0; ni - (k + 1) * SB >= 2; ++k) {
        const int r0 = (k + 1) * SB, m = ni - r0;
        const int ntile = (m + 15) / 16, nt1 = (m + 31) / 32;
        hipLaunchKernelGGL(k_sbr_y, dim3((unsigned)ntile), dim3(SBR_YW * 64), 0, st, d_A, (int)lda, ni, r0, Vb(k), Y, M1p);
        const bool next = ni - (k + 2) * SB >= 2;
        const int nrown = (next && fuse) ? (m - 16 + 255) / 256 : 0;
        const int ntb = nrown > 0 ? std::min(nt1 * nt1, 160) : nt1 * nt1;   // with a panel block: every block resident at once, that block first
        const unsigned nblk = (unsigned)(ntb + (nrown > 0 ? 1 : 0));
#define JCDF_UPD(NR) hipLaunchKernelGGL(k_sbr_update<NR>, dim3(nblk), dim3(256), 0, st, d_A, (int)lda, ni, r0, Vb(k), Y, Tb(k), M1p, ntile, \
                                        nt1, ntb, Vb(k + 1), Tb(k + 1))
        if (nrown == 0) JCDF_UPD(0);
        else if (nrown == 1) JCDF_UPD(1);
        else if (nrown == 2) JCDF_UPD(2);
        else if (nrown == 3) JCDF_UPD(3);
        else JCDF_UPD(5);
#undef JCDF_UPD
        if (next) {
            if (fuse) after_panel(k + 1);
            else launch_panel(k + 1);
        }
    }
    hipLaunchKernelGGL(k_sbr_extract, dim3((unsigned)((n * SBW + 255) / 256)), dim3(256), 0, st, d_A, (int)lda, ni, AB);
    if (sd) {
        hip_ok = hip_ok && hipEventRecord(sd->events[0], st) == hipSuccess && hipStreamWaitEvent(sd->stream, sd->events[0], 0) == hipSuccess;
        for (int k = 0; ni - (k + 1) * SB >= 2; ++k) q_update(k, sd->stream);
        hip_ok = hip_ok && hipEventRecord(sd->events[1], sd->stream) == hipSuccess;
    }
    // two waves per sweep (chain + update) where the ring fits beside the band; JCDF_SB2ST_ONE_WAVE=1: the one-wave kernel
    static const int variant = diag_env("JCDF_SB2ST_VARIANT") ? atoi(diag_env("JCDF_SB2ST_VARIANT")) : 1;   // 1: one wave per sweep (fastest measured), 2: chain + update waves, 3: ping-pong
    const bool one_wave = variant == 1;
    if (variant == 3 && sb2st3_lds(n) <= 160 * 1024) {
        const size_t lds = sb2st3_lds(n);
        if (hipFuncSetAttribute((const void *)k_sb2st_chase3<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return JCDF_ERR_HIP;
        hipLaunchKernelGGL(k_sb2st_chase3<8>, dim3(1), dim3(8 * 128), lds, st, AB, ni, d_D, d_E, vlog, (int)L.tmax, err);
    } else if (!one_wave && sb2st2_lds(n) <= 160 * 1024) {
        const size_t lds = sb2st2_lds(n);
        if (hipFuncSetAttribute((const void *)k_sb2st_chase2<SB2ST_PAIRS, SB2ST_RING>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds) != hipSuccess)
            return JCDF_ERR_HIP;
        hipLaunchKernelGGL((k_sb2st_chase2<SB2ST_PAIRS, SB2ST_RING>), dim3(1), dim3(SB2ST_PAIRS * 128), lds, st, AB, ni, d_D, d_E, vlog,
                           (int)L.tmax, err);
    } else {
        const size_t lds = sb2st_lds(n);
        static const int nw = diag_env("JCDF_SB2ST_NW") ? atoi(diag_env("JCDF_SB2ST_NW")) : SB2ST_WAVES;    // experiments only
#define JCDF_CHASE1(NW_)                                                                                                         \
    do {                                                                                                                         \
        if (hipFuncSetAttribute((const void *)k_sb2st_chase<NW_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) \
            return JCDF_ERR_HIP;                                                                                                 \
        hipLaunchKernelGGL(k_sb2st_chase<NW_>, dim3(1), dim3(NW_ * 64), lds, st, AB, ni, d_D, d_E, vlog, (int)L.tmax, err);      \
    } while (0)
        if (nw == 1) JCDF_CHASE1(1);
        else if (nw == 2) JCDF_CHASE1(2);
        else if (nw == 4) JCDF_CHASE1(4);
        else if (nw == 8) JCDF_CHASE1(8);
        else JCDF_CHASE1(16);
#undef JCDF_CHASE1
    }
    if (sd) hip_ok = hip_ok && hipStreamWaitEvent(st, sd->events[1], 0) == hipSuccess;   // d_Q is complete for whatever follows on `stream`
    return (hip_ok && hipGetLastError() == hipSuccess) ? JCDF_OK : JCDF_ERR_HIP;
}

#ifdef JCDF_SB2ST_PROFILE
int32_t jcdf_sb2st_profile(unsigned long long *out, int32_t reset)       // diagnostic builds only (not in jcdf.h)
{
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_sb2st_prof), sizeof(unsigned long long) * 16 * 8) != hipSuccess) return JCDF_ERR_HIP;
    if (reset) {
        static unsigned long long zero[16 * 8];
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_sb2st_prof), zero, sizeof(zero)) != hipSuccess) return JCDF_ERR_HIP;
    }
    return JCDF_OK;
}
#endif

int32_t jcdf_sytrd2_apply_q_device(void *stream, int64_t n, double *d_Q, int64_t ldq, const void *d_work, int64_t work_bytes)
{
    if (n <= 0 || n > jcdf_sytrd2_max_n() || !d_Q || ldq < n || !d_work || work_bytes < jcdf_sytrd2_workspace_bytes(n))
        return JCDF_ERR_INVALID;
    if (n < 3) return JCDF_OK;
    hipStream_t st = (hipStream_t)stream;
    const Sytrd2Layout L = sytrd2_layout(n);
    const double *vlog = (const double *)((const char *)d_work + L.off_log);
    constexpr int NWQ = 4, MAXG = 10;                               // one row of Q per wave; 4 MAXG >= tmax (n <= 590 + ...)
    if (L.tmax > 4 * MAXG) return JCDF_ERR_INVALID;
    const size_t lds = (size_t)NWQ * (n + 144) * 8;
    if (hipFuncSetAttribute((const void *)k_sb2st_apply_q<NWQ, MAXG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return JCDF_ERR_HIP;
    hipLaunchKernelGGL((k_sb2st_apply_q<NWQ, MAXG>), dim3((unsigned)((n + NWQ - 1) / NWQ)), dim3(NWQ * 64), lds, st, d_Q, (int)ldq, (int)n,
                       vlog, (int)L.tmax);
    return hipGetLastError() == hipSuccess ? JCDF_OK : JCDF_ERR_HIP;
}

#endif  // JCDF_DIAGNOSTIC

int32_t jcdf_diis_device(void *stream, int32_t nd, int32_t head, int32_t n, int32_t solve, double *d_Bmat, const double *d_dots,
                         double *d_coef, int32_t *d_flag)
{
    if (nd < 1 || nd > 15 || head < 0 || head >= nd || n < 1 || n > nd || !d_Bmat || !d_dots || !d_coef || !d_flag)
        return JCDF_ERR_INVALID;
    hipLaunchKernelGGL(k_diis_solve, dim3(1), dim3(64), 0, (hipStream_t)stream, d_Bmat, d_dots, (int)nd, (int)head, (int)n,
                       (int)solve, d_coef, d_flag);
    return hipGetLastError() == hipSuccess ? JCDF_OK : JCDF_ERR_HIP;
}

int32_t jcdf_scf_tail_device(void *stream, int64_t n, const double *d_D, const double *d_D_old, const double *d_F, const double *d_H,
                             const int32_t *d_diis_flag, const int32_t *d_eig_err, const int32_t *d_eig_info, const double *d_sp2_info,
                             const double *d_pivot, double *d_work, double *d_out)
{
    if (n < 1 || !d_D || !d_D_old || !d_F || !d_H || !d_work || !d_out) return JCDF_ERR_INVALID;
    hipStream_t st = (hipStream_t)stream;
    const int64_t nn = n * n;
    const int groups = (int)std::min<int64_t>(SCF_TAIL_GROUPS, (nn + 255) / 256);
    hipLaunchKernelGGL(k_scf_tail_partial, dim3(groups), dim3(256), 0, st, d_D, d_D_old, d_F, d_H, nn, d_work);
    hipLaunchKernelGGL(k_scf_tail_final, dim3(1), dim3(64), 0, st, d_work, groups, d_diis_flag, d_eig_err, d_eig_info, d_sp2_info, d_pivot,
                       d_out);
    return hipGetLastError() == hipSuccess ? JCDF_OK : JCDF_ERR_HIP;
}

// ---- SP2 density solver (jcdf_sp2.hpp) -----------------------------------------------------------
namespace {
struct Sp2Work {
    double *Xa, *Xb, *part, *partials;
    Sp2State *state;
    int64_t np, ld, bytes;
    int chunks, ntri, tile;
};
Sp2Work sp2_carve(char *base, int64_t n)
{
    Sp2Work w;
    w.np = roundup(n, SP2_PAD);
    w.ld = w.np;
    w.tile = w.np >= SP2_T64_MIN_NP ? 64 : 32;
    const int nt = (int)(w.np / w.tile);
    w.ntri = nt * (nt + 1) / 2;
    w.chunks = (int)(w.np / Sp2Cfg::KC);
    size_t off = 0;
    auto take = [&](size_t bytes) { char *p = base ? base + off : nullptr; off += roundup((int64_t)bytes, 256); return p; };
    w.Xa = (double *)take((size_t)w.np * w.ld * 8);
    w.Xb = (double *)take((size_t)w.np * w.ld * 8);
    w.part = (double *)take(3 * 1024 * 8);
    w.partials = (double *)take((size_t)2 * 2 * SP2_PART * 8);
    w.state = (Sp2State *)take(2 * sizeof(Sp2State));
    w.bytes = (int64_t)off;
    return w;
}
}  // namespace

int64_t jcdf_sp2_workspace_bytes(int64_t n)
{
    if (n <= 0) return 0;
    return sp2_carve(nullptr, n).bytes;
}

int32_t jcdf_sp2_device(void *stream, int64_t n, int64_t n_occ, const double *d_F, int64_t ldf, double *d_P, int64_t ldp,
                        int32_t iterations, void *d_work, int64_t work_bytes, double *d_info)
{
    return jcdf_sp2_ref_device(stream, n, n_occ, d_F, ldf, d_P, ldp, iterations, d_work, work_bytes, d_info, nullptr, 0, nullptr);
}

int32_t jcdf_sp2_ref_device(void *stream, int64_t n, int64_t n_occ, const double *d_F, int64_t ldf, double *d_P, int64_t ldp,
                            int32_t iterations, void *d_work, int64_t work_bytes, double *d_info, const double *d_Fref, int64_t ldr,
                            const double *d_ref_eigs)
{
    if (n < 2 || n > 4096 || n_occ < 1 || n_occ >= n || !d_F || !d_P || ldf < n || ldp < n || iterations < 1 || iterations > 1000 ||
        !d_work || !d_info || ((d_Fref != nullptr) != (d_ref_eigs != nullptr)) || (d_Fref && ldr < n))
        return JCDF_ERR_INVALID;
    Sp2Work w = sp2_carve((char *)d_work, n);
    if (work_bytes < w.bytes) return JCDF_ERR_INVALID;
    hipStream_t st = (hipStream_t)stream;
    const int nb = (int)std::min<int64_t>(1024, (n + 3) / 4);
    hipLaunchKernelGGL(k_sp2_bounds, dim3(nb), dim3(256), 0, st, d_F, ldf, (int)n, w.part, d_Fref, ldr);
    hipLaunchKernelGGL(k_sp2_init, dim3((unsigned)w.np), dim3(256), 0, st, d_F, ldf, (int)n, (int)w.np, w.part, nb, w.Xa, w.ld, w.state,
                       w.partials, d_ref_eigs);
    if (ensure_device_attributes() != hipSuccess) return JCDF_ERR_HIP;
    for (int k = 0; k < iterations; ++k) {
        if (w.tile == 64)
            hipLaunchKernelGGL(k_sp2_fused<Sp2Cfg64>, dim3((unsigned)(8 * ((w.ntri + 7) / 8))), dim3(Sp2Cfg64::NT), Sp2Cfg64::SMEM_BYTES, st, w.Xa, w.Xb, w.Xa,
                               w.Xb, w.ld, (int)n_occ, w.chunks, w.partials, (int)w.np, w.ntri, w.state, k, (int)n);
        else
            hipLaunchKernelGGL(k_sp2_fused<Sp2Cfg>, dim3((unsigned)(8 * ((w.ntri + 7) / 8))), dim3(Sp2Cfg::NT), Sp2Cfg::SMEM_BYTES, st, w.Xa, w.Xb, w.Xa, w.Xb,
                               w.ld, (int)n_occ, w.chunks, w.partials, (int)w.np, w.ntri, w.state, k, (int)n);
    }
    hipLaunchKernelGGL(k_sp2_finish, dim3((unsigned)std::min<int64_t>(n, 512)), dim3(256), 0, st, w.Xa, w.Xb, w.ld, (int)n, d_P, ldp,
                       w.state, (int)iterations, d_info);
    return hipGetLastError() == hipSuccess ? JCDF_OK : JCDF_ERR_HIP;
}

static size_t dc_prepare_lds(int maxm) { return (size_t)(2 * maxm + std::max(maxm, 256)) * 8 + (size_t)3 * maxm * 4; }

// the largest merge of the plan must fit k_dc_prepare's LDS (36 bytes per row: n <~ 2700) — checked before anything is enqueued
static bool dc_plan_fits(const DcPlan *plan)
{
    for (const DcLevel &lv : plan->levels)
        if (dc_prepare_lds(lv.maxm) > (size_t)DC_PREPARE_LDS_MAX) return false;
    return true;
}

int64_t jcdf_stedc_workspace_bytes(int64_t n)
{
    if (n <= 0) return 0;
    const DcPlan *plan = dc_plan(n);
    if (!plan || !dc_plan_fits(plan)) return -1;             // too large for this solver: the caller picks another one at construction
    return dc_carve(nullptr, n, plan).bytes;
}

int32_t jcdf_stedc_device(void *stream, int64_t n, double *d_D, double *d_E, double *d_Z, int64_t ldz, void *d_work,
                          int64_t work_bytes)
{
    if (n <= 0 || !d_D || !d_Z || ldz < n || (n > 1 && !d_E) || !d_work) return JCDF_ERR_INVALID;
    const DcPlan *plan = dc_plan(n);
    if (!plan) return JCDF_ERR_ALLOC;
    if (!dc_plan_fits(plan)) return JCDF_ERR_INVALID;                      // before the first launch (jcdf_stedc_workspace_bytes says -1 for such n)
    DcWork wk = dc_carve((char *)d_work, n, plan);
    if (work_bytes < wk.bytes) return JCDF_ERR_INVALID;
    hipStream_t st = (hipStream_t)stream;
    if (ensure_device_attributes() != hipSuccess) return JCDF_ERR_HIP;
    const int L = (int)plan->levels.size();
    // ping-pong so that the last level writes into the caller's buffers: eigenvalues end in d_D, vectors in d_Z
    double *Za = (L % 2 == 0) ? d_Z : wk.Zb, *Zn = (L % 2 == 0) ? wk.Zb : d_Z;
    int64_t lda = (L % 2 == 0) ? ldz : wk.ldzb, ldn = (L % 2 == 0) ? wk.ldzb : ldz;
    double *wa = (L % 2 == 0) ? d_D : wk.w2, *wn = (L % 2 == 0) ? wk.w2 : d_D;
    // (wa may be d_D itself: a leaf reads its own diagonal elements, and E, before it writes its eigenvalues over them)
    hipLaunchKernelGGL(k_dc_norm, dim3(1), dim3(256), 0, st, d_D, d_E, (int)n, wk.sc, wk.info);
    hipLaunchKernelGGL(k_dc_init, dim3((unsigned)((n * n + 255) / 256)), dim3(256), 0, st, (int)n, Za, lda, Zn, ldn);
    {
        const int64_t leaves = (n + DC_LEAF - 1) / DC_LEAF;
        hipLaunchKernelGGL(k_dc_leaf, dim3((unsigned)leaves), dim3(64), 0, st, (const double *)d_D,
                           (const double *)d_E, (int)n, wa, Za, lda, (const double *)wk.sc, L == 0 ? 1 : 0, wk.info);
    }
    for (int l = 0; l < L; ++l) {
        const DcLevel &lv = plan->levels[l];
        const DcMerge *mg = plan->d_merges + lv.merge_off;
        const int maxm = lv.maxm;
        const size_t prep_lds = dc_prepare_lds(maxm);
        static const int fuse_max = diag_env("JCDF_DC_FUSE_MAX") ? atoi(diag_env("JCDF_DC_FUSE_MAX")) : 32;
        if (maxm <= fuse_max) {       // tiny merges: one launch per level instead of six
            hipLaunchKernelGGL(k_dc_merge_small<4>, dim3((unsigned)lv.nm), dim3(256), prep_lds, st, mg, wa, d_E, Za, lda, wk.K, wk.rho,
                               wk.dl, wk.zl, wk.col, wk.defcol, wk.defval, wk.sc, wk.org, wk.mu, wk.zhat, wk.X, wk.Zp, wk.ldx, wk.G, Zn,
                               ldn, wn, l == L - 1 ? 1 : 0, (const int *)wk.info);
        } else {
            hipLaunchKernelGGL(k_dc_prepare, dim3((unsigned)lv.nm), dim3(256), prep_lds, st, mg, wa, d_E, Za, lda, wk.K, wk.rho,
                               wk.dl, wk.zl, wk.col, wk.defcol, wk.defval, wk.sc, (const int *)wk.info);
            // lanes per root / per zhat entry: enough workgroups at the big levels, no idle lanes at the small ones
            const unsigned nmu = (unsigned)lv.nm;
            if (maxm >= 256) {
                const unsigned gx = (unsigned)((maxm + 3) / 4);                   // 4 roots (64 lanes each) per block
                hipLaunchKernelGGL(k_dc_secular<64>, dim3(gx, nmu), dim3(256), (size_t)2 * maxm * 8, st, mg, wk.K, wk.rho, wk.dl, wk.zl, wk.org, wk.mu, (const int *)wk.info);
                hipLaunchKernelGGL(k_dc_zhat<64>, dim3(gx, nmu), dim3(256), 0, st, mg, wk.K, wk.dl, wk.zl, wk.org, wk.mu, wk.zhat, (const int *)wk.info);
            } else if (maxm >= 64) {
                const unsigned gx = (unsigned)((maxm + 15) / 16);                 // 16 roots (16 lanes each) per block
                hipLaunchKernelGGL(k_dc_secular<16>, dim3(gx, nmu), dim3(256), (size_t)2 * maxm * 8, st, mg, wk.K, wk.rho, wk.dl, wk.zl, wk.org, wk.mu, (const int *)wk.info);
                hipLaunchKernelGGL(k_dc_zhat<16>, dim3(gx, nmu), dim3(256), 0, st, mg, wk.K, wk.dl, wk.zl, wk.org, wk.mu, wk.zhat, (const int *)wk.info);
            } else {
                const unsigned gx = (unsigned)std::max(1, (maxm + 63) / 64);      // 64 roots (4 lanes each) per block
                hipLaunchKernelGGL(k_dc_secular<4>, dim3(gx, nmu), dim3(256), (size_t)2 * maxm * 8, st, mg, wk.K, wk.rho, wk.dl, wk.zl, wk.org, wk.mu, (const int *)wk.info);
                hipLaunchKernelGGL(k_dc_zhat<4>, dim3(gx, nmu), dim3(256), 0, st, mg, wk.K, wk.dl, wk.zl, wk.org, wk.mu, wk.zhat, (const int *)wk.info);
            }
            {
                const int64_t work = (int64_t)roundup(maxm, 16) * (roundup(maxm, 16) + maxm);
                const unsigned gx = (unsigned)std::max<int64_t>(1, std::min<int64_t>(512 / std::max(1, lv.nm) + 1, (work + 1023) / 1024));
                hipLaunchKernelGGL(k_dc_vectors, dim3(gx, nmu), dim3(256), 0, st, mg, wk.K, wk.dl, wk.org, wk.mu, wk.zhat, wk.col, Za, lda,
                                   wk.X, wk.Zp, wk.ldx, maxm >= DC_MFMA_MIN ? 1 : 0, (const int *)wk.info);
            }
            if (maxm >= DC_MFMA_MIN) {
                const unsigned tiles = (unsigned)(((maxm + 63) / 64) * ((maxm + 63) / 64));
                if ((int64_t)tiles * lv.nm < 128) {          // too few 64 x 64 tiles to fill the chip: 32 x 32
                    const unsigned t32 = (unsigned)(((maxm + 31) / 32) * ((maxm + 31) / 32));
                    hipLaunchKernelGGL(k_dc_update_mfma<DcCfg32>, dim3(t32, (unsigned)lv.nm), dim3(DcCfg32::NT), DcCfg32::SMEM_BYTES, st, mg,
                                       wk.K, wk.X, wk.Zp, wk.ldx, wk.G, (const int *)wk.info);
                } else {
                    hipLaunchKernelGGL(k_dc_update_mfma<DcCfg>, dim3(tiles, (unsigned)lv.nm), dim3(DcCfg::NT), DcCfg::SMEM_BYTES, st, mg, wk.K,
                                       wk.X, wk.Zp, wk.ldx, wk.G, (const int *)wk.info);
                }
            } else {
                const unsigned tiles = (unsigned)std::min(64, ((maxm + 15) / 16) * ((maxm + 15) / 16));
                hipLaunchKernelGGL(k_dc_update_simple, dim3(tiles, (unsigned)lv.nm), dim3(256), 0, st, mg, wk.K, wk.X, wk.Zp, wk.ldx,
                                   wk.G, (const int *)wk.info);
            }
            const unsigned fx = (unsigned)std::max(1, (maxm + 7) / 8);
            hipLaunchKernelGGL(k_dc_finish, dim3(fx, (unsigned)lv.nm), dim3(256), (size_t)maxm * 8, st, mg, wk.K, wk.dl, wk.org, wk.mu,
                               wk.defcol, wk.defval, wk.G, wk.ldx, Za, lda, Zn, ldn, wn, wk.sc, l == L - 1 ? 1 : 0, (const int *)wk.info);
        }
        if (lv.has_carry)
            hipLaunchKernelGGL(k_dc_carry, dim3(16, 1), dim3(256), 0, st, mg + lv.nm, Za, lda, Zn, ldn, wa, wn);
        std::swap(Za, Zn);
        std::swap(lda, ldn);
        std::swap(wa, wn);
    }
    return hipGetLastError() == hipSuccess ? JCDF_OK : JCDF_ERR_HIP;
}

// ---- small dense products of the device SCF iteration (jcdf_blas.hpp) -------------------------------------------
int32_t jcdf_gemm_tn_device(void *stream, int64_t M, int64_t N, int64_t K, double alpha, const double *d_A, int64_t lda,
                            const double *d_B, int64_t ldb, double *d_C, int64_t ldc)
{
    if (M <= 0 || N <= 0 || K <= 0 || M % 32 || N % 32 || K % 32 || !d_A || !d_B || !d_C || lda < M || ldb < N || ldc < N ||
        (lda & 1) || (ldb & 1))
        return JCDF_ERR_INVALID;
    if (ensure_device_attributes() != hipSuccess) return JCDF_ERR_HIP;
    // 64 x 64 tiles (8 waves) once they fill at least half the chip: half the operand traffic of the 32 x 32 form, which
    // is bound by the L2 from ~1000^3 on
    if (M % 64 == 0 && N % 64 == 0 && (M / 64) * (N / 64) >= 128) {
        const int n_tn = (int)(N / 64);
        hipLaunchKernelGGL(k_blas_gemm_tn<BlasTN64Cfg>, dim3((unsigned)((M / 64) * n_tn)), dim3(BlasTN64Cfg::NT), BlasTN64Cfg::SMEM_BYTES,
                           (hipStream_t)stream, d_A, lda, d_B, ldb, d_C, ldc, (int)(K / 32), alpha, n_tn);
    } else {
        const int n_tn = (int)(N / 32);
        hipLaunchKernelGGL(k_blas_gemm_tn<BlasTNCfg>, dim3((unsigned)((M / 32) * n_tn)), dim3(BlasTNCfg::NT), BlasTNCfg::SMEM_BYTES,
                           (hipStream_t)stream, d_A, lda, d_B, ldb, d_C, ldc, (int)(K / 32), alpha, n_tn);
    }
    return hipGetLastError() == hipSuccess ? JCDF_OK : JCDF_ERR_HIP;
}

int32_t jcdf_gemm_nt_device(void *stream, int64_t M, int64_t N, int64_t K, const double *d_A, int64_t lda, const double *d_B,
                            int64_t ldb, double *d_C, int64_t ldc)
{
    if (M <= 0 || N <= 0 || K <= 0 || M % 32 || N % 32 || K % 16 || !d_A || !d_B || !d_C || lda < K || ldb < K || ldc < N ||
        (lda & 1) || (ldb & 1))
        return JCDF_ERR_INVALID;
    if (ensure_device_attributes() != hipSuccess) return JCDF_ERR_HIP;
    const int n_tn = (int)(N / 32);
    hipLaunchKernelGGL(k_blas_gemm_nt, dim3((unsigned)((M / 32) * n_tn)), dim3(BlasNTCfg::NT), GemmNT<BlasNTCfg>::SMEM_BYTES,
                       (hipStream_t)stream, d_A, lda, d_B, ldb, d_C, ldc, (int)(K / 16), n_tn);
    return hipGetLastError() == hipSuccess ? JCDF_OK : JCDF_ERR_HIP;
}

// ---- Loewdin orthonormalisation of row vectors by Newton-Schulz (jcdf_blas.hpp) ----------------------------------------------
namespace {
constexpr int LOWDIN_MAX_ITER = 40;
struct LowdinWork {
    double *G0, *Y[2], *Yt[2], *Z[2], *Zt[2], *T, *Tt, *part0, *part;
    int64_t op, bytes;
    int ntile, n0;
};
LowdinWork lowdin_carve(char *base, int64_t o)
{
    LowdinWork w;
    w.op = roundup(o, 32);
    w.ntile = (int)((w.op / 32) * (w.op / 32));
    w.n0 = (int)((w.op * w.op + 255) / 256);
    size_t off = 0;
    auto take = [&](size_t doubles) { char *p = base ? base + off : nullptr; off += (size_t)roundup((int64_t)doubles * 8, 256); return (double *)p; };
    const size_t m = (size_t)w.op * w.op;
    w.G0 = take(m);
    for (int k = 0; k < 2; ++k) { w.Y[k] = take(m); w.Yt[k] = take(m); w.Z[k] = take(m); w.Zt[k] = take(m); }
    w.T = take(m); w.Tt = take(m);
    w.part0 = take((size_t)w.n0);
    w.part = take((size_t)LOWDIN_MAX_ITER * w.ntile);
    w.bytes = (int64_t)off;
    return w;
}
}  // namespace

int64_t jcdf_lowdin_workspace_bytes(int64_t o)
{
    if (o <= 0) return 0;
    return lowdin_carve(nullptr, o).bytes;
}

int32_t jcdf_lowdin_rows_device(void *stream, int64_t o, int64_t n, const double *d_Y, int64_t ldy, double *d_Z, int64_t ldz,
                                int32_t iterations, void *d_work, int64_t work_bytes, double *d_info)
{
    if (o < 1 || o > 4096 || n < 1 || !d_Y || !d_Z || !d_work || !d_info || iterations < 1 || iterations > LOWDIN_MAX_ITER) return JCDF_ERR_INVALID;
    const int64_t np = roundup(n, 32);
    if (ldy < np || ldz < np || (ldy & 1) || (ldz & 1)) return JCDF_ERR_INVALID;
    LowdinWork w = lowdin_carve((char *)d_work, o);
    if (work_bytes < w.bytes) return JCDF_ERR_INVALID;
    if (ensure_device_attributes() != hipSuccess) return JCDF_ERR_HIP;
    hipStream_t st = (hipStream_t)stream;
    const int op = (int)w.op, n_t = op / 32;
    const int64_t ld = op;
    // G = Y Y^T on the NT core (rows of Y are k-contiguous; the zero padding of Y adds nothing)
    hipLaunchKernelGGL(k_blas_gemm_nt, dim3((unsigned)(n_t * n_t)), dim3(BlasNTCfg::NT), GemmNT<BlasNTCfg>::SMEM_BYTES, st, d_Y, ldy, d_Y, ldy,
                       w.G0, ld, (int)(np / 16), n_t);
    // Y_0 = Y_0^T = G (lower triangle mirrored: exactly symmetric, unit diagonal in the padding), Z_0 = Z_0^T = I
    hipLaunchKernelGGL(k_lowdin_prepare, dim3((unsigned)w.n0), dim3(256), 0, st, w.G0, w.Y[0], w.Yt[0], w.Z[0], w.Zt[0], (int)o, op, w.part0);
    int c = 0;
    const NsProblem none{nullptr, nullptr, nullptr, nullptr};
    for (int k = 0; k < iterations; ++k) {
        // T = (3 I - Z Y) / 2 with the residual ||T - I||_F^2 per tile;  Y' = Y T and Z' = T Z in one launch
        hipLaunchKernelGGL(k_ns_gemm, dim3((unsigned)w.ntile, 1), dim3(BlasTNCfg::NT), BlasTNCfg::SMEM_BYTES, st,
                           NsProblem{w.Zt[c], w.Y[c], w.T, w.Tt}, none, ld, op / 32, n_t, -0.5, 1.5, w.part + (int64_t)k * w.ntile);
        hipLaunchKernelGGL(k_ns_gemm, dim3((unsigned)w.ntile, 2), dim3(BlasTNCfg::NT), BlasTNCfg::SMEM_BYTES, st,
                           NsProblem{w.Yt[c], w.T, w.Y[c ^ 1], w.Yt[c ^ 1]}, NsProblem{w.Tt, w.Z[c], w.Z[c ^ 1], w.Zt[c ^ 1]}, ld, op / 32, n_t, 1.0,
                           0.0, (double *)nullptr);
        c ^= 1;
    }
    // out[i][col] = sum_k Z[i][k] Y[k][col] = sum_k Zt[k][i] Y[k][col]
    hipLaunchKernelGGL(k_blas_gemm_tn<BlasTNCfg>, dim3((unsigned)(n_t * (np / 32))), dim3(BlasTNCfg::NT), BlasTNCfg::SMEM_BYTES, st, w.Zt[c], ld, d_Y,
                       ldy, d_Z, ldz, op / 32, 1.0, (int)(np / 32));
    hipLaunchKernelGGL(k_lowdin_info, dim3(1), dim3(64), 0, st, w.part0, w.n0, w.part, w.ntile, (int)iterations, d_info);
    return hipGetLastError() == hipSuccess ? JCDF_OK : JCDF_ERR_HIP;
}

int32_t jcdf_diis_push_device(void *stream, int64_t n, int64_t ld, const double *d_T, const double *d_F, double *d_e_slot, double *d_f_slot)
{
    if (n <= 0 || ld < n || !d_T || !d_F || !d_e_slot || !d_f_slot) return JCDF_ERR_INVALID;
    hipLaunchKernelGGL(k_diis_push, dim3((unsigned)((n * n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_T, d_F, ld, (int)n, d_e_slot,
                       d_f_slot);
    return hipGetLastError() == hipSuccess ? JCDF_OK : JCDF_ERR_HIP;
}

int32_t jcdf_diis_dots_device(void *stream, int32_t nd, int32_t head, int64_t len, const double *d_e_hist, double *d_dots, double *d_work)
{
    if (nd < 1 || nd > 64 || head < 0 || head >= nd || len <= 0 || !d_e_hist || !d_dots || !d_work) return JCDF_ERR_INVALID;
    hipLaunchKernelGGL(k_diis_dots_partial, dim3(DIIS_DOT_PARTS, (unsigned)nd), dim3(256), 0, (hipStream_t)stream, d_e_hist, len, (int)head,
                       d_work);
    hipLaunchKernelGGL(k_diis_dots_final, dim3(1), dim3(64), 0, (hipStream_t)stream, d_work, (int)nd, d_dots);
    return hipGetLastError() == hipSuccess ? JCDF_OK : JCDF_ERR_HIP;
}

int32_t jcdf_diis_step_device(void *stream, int32_t nd, int32_t head, int32_t n_use, int32_t solve, int64_t n, int64_t ld, const double *d_T,
                              double *d_F, double *d_e_hist, double *d_f_hist, double *d_Bmat, double *d_coef, int32_t *d_flag, double *d_work,
                              const double *d_F_old, double x)
{
    if (nd < 1 || nd > 15 || head < 0 || head >= nd || n_use < 1 || n_use > nd || n <= 0 || ld < n || !d_T || !d_F || !d_e_hist || !d_f_hist ||
        !d_Bmat || !d_coef || !d_flag || !d_work || (x != 1.0 && !d_F_old))
        return JCDF_ERR_INVALID;
    hipStream_t st = (hipStream_t)stream;
    const int64_t len = n * n;
    hipLaunchKernelGGL(k_diis_push, dim3((unsigned)((len + 255) / 256)), dim3(256), 0, st, d_T, d_F, ld, (int)n, d_e_hist + (int64_t)head * len,
                       d_f_hist + (int64_t)head * len);
    hipLaunchKernelGGL(k_diis_dots_partial, dim3(DIIS_DOT_PARTS, (unsigned)nd), dim3(256), 0, st, d_e_hist, len, (int)head, d_work);
    hipLaunchKernelGGL(k_diis_solve, dim3(1), dim3(64), 0, st, d_Bmat, (const double *)d_work, (int)nd, (int)head, (int)n_use, (int)solve, d_coef, d_flag,
                       DIIS_DOT_PARTS);
    if (solve || x != 1.0)
        hipLaunchKernelGGL(k_diis_mix, dim3((unsigned)((len + 255) / 256)), dim3(256), 0, st, d_f_hist, len, (int)nd, d_coef, (int)n, d_F, ld,
                           x != 1.0 ? d_F_old : nullptr, x);
    return hipGetLastError() == hipSuccess ? JCDF_OK : JCDF_ERR_HIP;
}

int32_t jcdf_diis_mix_device(void *stream, int32_t nd, int64_t n, int64_t ld, const double *d_f_hist, const double *d_coef, double *d_F)
{
    if (nd < 1 || nd > 64 || n <= 0 || ld < n || !d_f_hist || !d_coef || !d_F) return JCDF_ERR_INVALID;
    hipLaunchKernelGGL(k_diis_mix, dim3((unsigned)((n * n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_f_hist, n * n, (int)nd, d_coef,
                       (int)n, d_F, ld);
    return hipGetLastError() == hipSuccess ? JCDF_OK : JCDF_ERR_HIP;
}

#ifdef JCDF_DIAGNOSTIC
// Diagnostic (JCDF_W_ABLATE=32, C20H42-shaped form only): per-wave cycles of the five segments of the W kernel's phases
// from the last build: out[(block * waves + wave) * 6 + {issue, mfma, misc, vmcnt wait, barrier, phases}].  Returns the
// number of wave records written, 0 if the diagnostic build is not active.
int64_t jcdf_w_stall_cycles(jcdf_handle *h, unsigned long long *out, int64_t max_waves)
{
    if (!h || !h->dStall || !out) return 0;
    if (hipSetDevice(h->device) != hipSuccess || hipStreamSynchronize(h->stream) != hipSuccess) return 0;
    const int64_t waves = std::min<int64_t>(max_waves, (int64_t)h->n_chunks * h->n_mtiles * h->n_qt * 4);
    if (hipMemcpy(out, h->dStall, (size_t)waves * 6 * 8, hipMemcpyDeviceToHost) != hipSuccess) return 0;
    return waves;
}

#endif  // JCDF_DIAGNOSTIC

int64_t jcdf_device_bytes(const jcdf_handle *h) { return h ? h->bytes : 0; }

int32_t jcdf_kernel_stats(jcdf_handle *h, jcdf_kernel_stat *out, int32_t max_records)
{
    if (!h || !out || max_records <= 0) return 0;
    if (h->pending) {
        (void)hipEventSynchronize(h->ev_end);
        h->pending = false;
    }
    int32_t n = 0;
    for (auto &r : h->recs) {
        if (n >= max_records) break;
        std::memset(&out[n], 0, sizeof(out[n]));
        std::snprintf(out[n].name, sizeof(out[n].name), "%s", r.name ? r.name : "?");
        out[n].seconds = elapsed_s(r.e0, r.e1);
        out[n].flops = r.flops;
        out[n].alg_flops = r.alg_flops;
        out[n].alg_bytes = r.alg_bytes;
        ++n;
    }
    return n;
}

int32_t jcdf_kernel_stats_total(jcdf_handle *h, jcdf_kernel_stat *out, int32_t max_records, int64_t *n_builds, double *fock_seconds,
                                int32_t reset)
{
    if (!h) return 0;
    (void)fold_set(h, h->evcur ^ 1, true);
    (void)fold_set(h, h->evcur, true);
    int32_t n = 0;
    if (out)
        for (auto &r : h->recs) {
            if (n >= max_records) break;
            std::memset(&out[n], 0, sizeof(out[n]));
            std::snprintf(out[n].name, sizeof(out[n].name), "%s", r.name ? r.name : "?");
            out[n].seconds = r.sum_seconds;
            out[n].flops = r.flops;
            out[n].alg_flops = r.alg_flops;
            out[n].alg_bytes = r.alg_bytes;
            ++n;
        }
    if (n_builds) *n_builds = h->builds_folded;
    if (fock_seconds) *fock_seconds = h->fock_sum_seconds;
    if (reset) {
        for (auto &r : h->recs) r.sum_seconds = 0.0;
        h->builds_folded = 0;
        h->fock_sum_seconds = 0.0;
    }
    return n;
}

}  // extern "C"

#include "jcdf_group.hpp"          // the multi-device group (jcdf_group_*)
