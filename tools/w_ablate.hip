// w_ablate.hip — in-process A/B timing of W-kernel variants (same device, interleaved
// rounds; cdna guide rule 24).  Also the timing-only ablations of the LDS-staged GEMM core.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I juliachem.jl_amd/csrc tools/w_ablate.hip -o tools/w_ablate
#include "jcdf_kernels.hpp"
#include <algorithm>
#include <cstdio>
#include <functional>
#include <string>
#include <vector>
using namespace jcdf;

template <class Cfg, int ABL, int MINW, int PF = 1>
__global__ __launch_bounds__(Cfg::NT, MINW) void k_w_lds(const double *__restrict__ B, const double *__restrict__ Cpad,
                                                           double *__restrict__ W, int Ql, int o, int Nk, int Np, int opad,
                                                           int n_ntiles)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int b = blockIdx.x;
    if (b >= Ql * n_ntiles) return;
    const int Q = b / n_ntiles, nt = b % n_ntiles;
    double4_t acc[Cfg::WM][Cfg::WN];
    for (int m = 0; m < Cfg::WM; ++m)
        for (int n = 0; n < Cfg::WN; ++n) acc[m][n] = double4_t{0, 0, 0, 0};
    gemm_tn_core<Cfg, true, ABL, PF>(Cpad, opad, B + (int64_t)Q * Nk * Np + nt * Cfg::TN, Np, Nk / Cfg::KC, acc, smem);
    for (int n = 0; n < Cfg::WN; ++n) {
        const int p = nt * Cfg::TN + tile_col<Cfg>(n);
        for (int m = 0; m < Cfg::WM; ++m)
            for (int j = 0; j < 4; ++j) {
                const int i = tile_row<Cfg>(m, j);
                if (i < o) W[((int64_t)Q * o + i) * Np + p] = acc[m][n][j];
            }
    }
}

// ---- experimental variant kept for the record: B fragments straight from HBM -----
// (no LDS for B; C staged in 32-row blocks, barrier every 96 MFMAs per wave).
// Measured slower than the LDS-staged product kernel: 4 rows x 256 B per wave-load
// is a worse shape for the memory path than 1 KiB contiguous rows.
constexpr int XW_KB = 32;
template <int WM, int NW>
struct XCfg {
    static constexpr int TM = 16 * WM, TN = 32 * NW, NT = 64 * NW;
    static constexpr int LDAS = TM + ((TM % 32 == 16) ? 0 : 16);
    static constexpr int SMEM_BYTES = 2 * XW_KB * LDAS * 8;
    static constexpr int A_VEC = XW_KB * TM / 2;
    static constexpr int A_PER_THREAD = (A_VEC + NT - 1) / NT;
};
template <int WM, int NW, bool BNT>
__global__ __launch_bounds__(64 * NW, 2) void k_w_direct(const double *__restrict__ B, const double *__restrict__ Cpad,
                                                          double *__restrict__ W, int Ql, int o, int Nk, int Np, int opad, int n_ntiles)
{
    using Cfg = XCfg<WM, NW>;
    constexpr int LDAS = Cfg::LDAS, NT = Cfg::NT;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int b = blockIdx.x;
    if (b >= Ql * n_ntiles) return;
    const int Q = b / n_ntiles, nt = b % n_ntiles;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lc = lane & 15, lk = lane >> 4;
    const int n0 = nt * Cfg::TN + wave * 32;
    double4_t acc[WM][2];
    for (int m = 0; m < WM; ++m) acc[m][0] = acc[m][1] = double4_t{0, 0, 0, 0};
    const double2_t *Bl = reinterpret_cast<const double2_t *>(B + (int64_t)Q * Nk * Np + (int64_t)lk * Np + n0 + 2 * lc);
    const int64_t rowstep2 = (int64_t)Np / 2;
    auto load_B = [&](double2_t (&bf)[4], int chunk) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const double2_t *src = Bl + (int64_t)(chunk * 16 + 4 * s) * rowstep2;
            bf[s] = BNT ? __builtin_nontemporal_load(src) : *src;
        }
    };
    double2_t ra[Cfg::A_PER_THREAD];
    auto load_A = [&](int blk) {
#pragma unroll
        for (int i = 0; i < Cfg::A_PER_THREAD; ++i) {
            const int idx = tid + i * NT;
            if (Cfg::A_VEC % NT == 0 || idx < Cfg::A_VEC) {
                const int rr = idx / (Cfg::TM / 2), cc = idx % (Cfg::TM / 2);
                ra[i] = *reinterpret_cast<const double2_t *>(Cpad + (int64_t)(blk * XW_KB + rr) * opad + 2 * cc);
            }
        }
    };
    auto store_A = [&](int buf) {
        double *As = smem + buf * (XW_KB * LDAS);
#pragma unroll
        for (int i = 0; i < Cfg::A_PER_THREAD; ++i) {
            const int idx = tid + i * NT;
            if (Cfg::A_VEC % NT == 0 || idx < Cfg::A_VEC) {
                const int rr = idx / (Cfg::TM / 2), cc = idx % (Cfg::TM / 2);
                *reinterpret_cast<double2_t *>(As + rr * LDAS + 2 * cc) = ra[i];
            }
        }
    };
    auto compute = [&](const double *As, const double2_t (&bf)[4]) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            double a[WM];
#pragma unroll
            for (int m = 0; m < WM; ++m) a[m] = As[(4 * s + lk) * LDAS + m * 16 + lc];
#pragma unroll
            for (int m = 0; m < WM; ++m) {
                acc[m][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m], bf[s].x, acc[m][0], 0, 0, 0);
                acc[m][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m], bf[s].y, acc[m][1], 0, 0, 0);
            }
        }
    };
    const int nblk = Nk / XW_KB;
    double2_t b0[4], b1[4];
    load_B(b0, 0); load_A(0); store_A(0);
    __syncthreads();
    for (int kb = 0; kb < nblk; ++kb) {
        const bool more = (kb + 1 < nblk);
        const double *As = smem + (kb & 1) * (XW_KB * LDAS);
        if (more) load_A(kb + 1);
        load_B(b1, 2 * kb + 1);
        compute(As, b0);
        if (more) load_B(b0, 2 * kb + 2);
        compute(As + 16 * LDAS, b1);
        if (more) store_A((kb + 1) & 1);
        __syncthreads();
    }
    const int p = n0 + 2 * lc;
    for (int m = 0; m < WM; ++m)
        for (int j = 0; j < 4; ++j) {
            const int i = m * 16 + lk + 4 * j;
            if (i < o) *reinterpret_cast<double2_t *>(W + ((int64_t)Q * o + i) * Np + p) = double2_t{acc[m][0][j], acc[m][1][j]};
        }
}

struct Variant { std::string name; std::function<void()> launch; std::vector<float> ms; };

int main(int argc, char **argv)
{
    const int Ql = 1950, o = 81, Nk = 512, Np = 512, opad = 96;
    double *B, *C, *W, *vp;
    hipMalloc(&B, (size_t)Ql * Nk * Np * 8); hipMalloc(&C, (size_t)Np * opad * 8);
    hipMalloc(&W, (size_t)(Ql * o + 64) * Np * 8); hipMalloc(&vp, (size_t)Ql * 16 * 8);
    std::vector<double> h((size_t)Nk * Np);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (double)((i * 2654435761u) % 1000) / 1000.0 - 0.5;
    for (int q = 0; q < Ql; ++q) hipMemcpy(B + (size_t)q * Nk * Np, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(C, h.data(), (size_t)Np * opad * 8, hipMemcpyHostToDevice);

    std::vector<Variant> vs;
    auto add_lds = [&](const char *name, auto cfg, auto abl, auto minw) {
        using Cfg = decltype(cfg);
        constexpr int ABL = decltype(abl)::value, MINW = decltype(minw)::value;
        hipFuncSetAttribute((const void *)k_w_lds<Cfg, ABL, MINW>, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::SMEM_BYTES);
        const int nnt = Np / Cfg::TN;
        vs.push_back({name, [=] { hipLaunchKernelGGL((k_w_lds<Cfg, ABL, MINW>), dim3(Ql * nnt), dim3(Cfg::NT), Cfg::SMEM_BYTES, 0, B, C, W, Ql, o, Nk, Np, opad, nnt); }, {}});
    };
    auto add_dir = [&](const char *name, auto nw, auto bnt) {
        constexpr int NW = decltype(nw)::value; constexpr bool BNT = decltype(bnt)::value;
        using Cfg = XCfg<6, NW>;
        hipFuncSetAttribute((const void *)k_w_direct<6, NW, BNT>, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::SMEM_BYTES);
        const int nnt = Np / Cfg::TN;
        vs.push_back({name, [=] { hipLaunchKernelGGL((k_w_direct<6, NW, BNT>), dim3(Ql * nnt), dim3(Cfg::NT), Cfg::SMEM_BYTES, 0, B, C, W, Ql, o, Nk, Np, opad, nnt); }, {}});
    };
    {   // the product kernel itself (XCD decode, fused V epilogue)
        using Cfg = WCfg<6>;
        hipFuncSetAttribute((const void *)k_exchange_W<6>, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::SMEM_BYTES);
        hipFuncSetAttribute((const void *)k_exchange_W<6, 1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::SMEM_BYTES);
        const int nnt = Np / Cfg::TN;
        const int nblk = ((Ql * nnt + 7) / 8) * 8;
        vs.push_back({"PRODUCT k_exchange_W<6> without fused V", [=] { hipLaunchKernelGGL((k_exchange_W<6, 1, false>), dim3(nblk), dim3(256), Cfg::SMEM_BYTES, 0, B, C, C, W, vp, Ql, o, Nk, Np, opad, 1, nnt, (const int *)nullptr, (const int *)nullptr); }, {}});
        vs.push_back({"PRODUCT k_exchange_W<6>", [=] { hipLaunchKernelGGL((k_exchange_W<6>), dim3(nblk), dim3(256), Cfg::SMEM_BYTES, 0, B, C, C, W, vp, Ql, o, Nk, Np, opad, 1, nnt, (const int *)nullptr, (const int *)nullptr); }, {}});
    }
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
    using I4 = std::integral_constant<int, 4>; using I8 = std::integral_constant<int, 8>; using I15 = std::integral_constant<int, 15>;
    add_dir("direct-B 4w TN128 nt", I4{}, std::true_type{});
    add_dir("direct-B 4w TN128 plain loads", I4{}, std::false_type{});
    add_dir("direct-B 8w TN256 nt", I8{}, std::true_type{});
    add_dir("direct-B 8w TN256 plain loads", I8{}, std::false_type{});
    add_lds("LDS-B 4w 96x128 KC16", GemmCfg<6, 2, 1, 4, 16>{}, I0{}, I2{});
    add_lds("LDS-B 8w 96x128 KC16", GemmCfg<6, 1, 1, 8, 16>{}, I0{}, I1{});
    add_lds("LDS-B 8w 96x256 KC16", GemmCfg<6, 2, 1, 8, 16>{}, I0{}, I1{});
    add_lds("LDS-B 4w: MFMA+ds_read only (ceiling)", GemmCfg<6, 2, 1, 4, 16>{}, I15{}, I2{});

    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (auto &v : vs) { v.launch(); }
    hipDeviceSynchronize();
    const int rounds = 7;
    for (int r = 0; r < rounds; ++r)
        for (auto &v : vs) {
            hipEventRecord(e0); v.launch(); v.launch(); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); v.ms.push_back(ms / 2);
        }
    const double fl = 2.0 * Ql * (double)Nk * Np * opad;
    for (auto &v : vs) {
        std::sort(v.ms.begin(), v.ms.end());
        printf("%-40s median %.3f ms (min %.3f)  %.1f TF exec\n", v.name.c_str(), v.ms[rounds / 2], v.ms[0], fl / v.ms[rounds / 2] / 1e9);
    }
    printf("last error: %s\n", hipGetErrorString(hipGetLastError()));
    return 0;
}
