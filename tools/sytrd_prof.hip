// Phase timeline of k_sytrd_lower (workgroup 1): hipcc --offload-arch=gfx950 -O3 -DJCDF_SYTRD_PROFILE -o sytrd_prof sytrd_prof.hip
#include "../juliachem.jl_amd/csrc/jcdf_gemm.hpp"
#include "../juliachem.jl_amd/csrc/jcdf_eig.hpp"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
using namespace jcdf;
#ifndef JCDF_SYTRD_PROFILE
__device__ u64 g_sytrd_prof[8];
#endif
int main(int argc, char **argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 510;
    const int G = argc > 2 ? atoi(argv[2]) : 64;
    std::vector<double> A((size_t)n * n);
    srand(1);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j <= i; ++j) A[(size_t)i * n + j] = A[(size_t)j * n + i] = rand() / (double)RAND_MAX - 0.5;
    double *dA, *dD, *dE, *dT;
    char *w;
    const size_t wb = 64 + (size_t)(2 * (n + 1) + 2 * n + 2 * 256) * 16;
    hipMalloc(&dA, A.size() * 8); hipMalloc(&dD, n * 8); hipMalloc(&dE, n * 8); hipMalloc(&dT, n * 8); hipMalloc(&w, wb);
    const int ncol = (n + G - 1) / G;
    const bool withq = argc > 3 ? atoi(argv[3]) != 0 : true;
    (void)(argc > 4 ? atoi(argv[4]) : 0);                          // (argv[4] unused, kept for the positions of the others)
    const int nthr = argc > 5 ? atoi(argv[5]) : 256;
    const int onehop = argc > 6 ? atoi(argv[6]) : 0;             // 1: k_sytrd_onehop
    const size_t lds = onehop ? ((size_t)ncol * n + 5 * n + 32) * 8 : ((size_t)(withq ? 2 : 1) * ncol * n + 2 * n + 32) * 8;
    if (onehop && (ncol > 8 || n > 1000)) { printf("onehop needs <= 8 columns per workgroup and n <= 1000\n"); return 1; }
    double *dQ; hipMalloc(&dQ, A.size() * 8);
    // columns 0 .. kstop-1 in the chip-wide kernel (the library stops 128 columns early and hands the rest to k_sytd2_tail)
    const int kstop = argc > 7 ? atoi(argv[7]) : (n > 128 ? n - 128 : n);
    u64 *vg = (u64 *)(w + 64), *yg = vg + 2 * (n + 2), *hg = yg + 2 * ((n + 1) & ~1);
    for (int rep = 0; rep < 3; ++rep) {
        hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice);
        hipMemset(w, 0, wb);
        u64 zero[8] = {0};
        hipMemcpyToSymbol(HIP_SYMBOL(g_sytrd_prof), zero, sizeof(zero));
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
#define ONEHOP(NR)                                                                                                                   \
    do {                                                                                                                             \
        hipFuncSetAttribute((const void *)k_sytrd_onehop<NR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                 \
        hipLaunchKernelGGL(k_sytrd_onehop<NR>, dim3(G), dim3(nthr), lds, 0, dA, n, n, dD, dE, dT, vg, yg, hg, (int *)(w + 8),         \
                           withq ? dQ : nullptr, n, kstop);                                                                          \
    } while (0)
        if (onehop) {                                        // the library's choice of the rows-per-lane template (jcdf_sytrd_q_device)
            if (n <= 64) ONEHOP(2);
            else if (n <= 256) ONEHOP(8);
            else if (n <= 512) ONEHOP(16);
            else if (n <= 640) ONEHOP(20);
            else ONEHOP(32);
        } else {
            hipFuncSetAttribute((const void *)k_sytrd_lower, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL(k_sytrd_lower, dim3(G), dim3(nthr), lds, 0, dA, n, n, dD, dE, dT, vg, yg, hg, (int *)(w + 8),
                               withq ? dQ : nullptr, n, kstop);
        }
        hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        u64 p[8]; hipMemcpyFromSymbol(p, HIP_SYMBOL(g_sytrd_prof), sizeof(p));
        int err; hipMemcpy(&err, w + 8, 4, hipMemcpyDeviceToHost);
        std::vector<double> d(n), e(n);
        hipMemcpy(d.data(), dD, n * 8, hipMemcpyDeviceToHost); hipMemcpy(e.data(), dE, n * 8, hipMemcpyDeviceToHost);
        double tr = 0, trT = 0, fr = 0, frT = 0;
        for (int i = 0; i < n; ++i) { tr += A[(size_t)i * n + i]; trT += d[i]; frT += d[i] * d[i] + (i < n - 1 ? 2 * e[i] * e[i] : 0); }
        for (size_t i = 0; i < A.size(); ++i) fr += A[i] * A[i];
        if (kstop >= n) printf("   invariants: trace %.3e  frob^2 rel %.3e\n", tr - trT, (fr - frT) / fr);
        printf("n=%d G=%d T=%d lds=%zu: %.3f ms (%.2f us/col) err=%d | per column us: wait_v %.2f  y %.2f  Q %.2f  wait_y %.2f  update %.2f\n", n, G, nthr, lds, ms,
               1e3 * ms / std::min(n, kstop), err, p[0] / 100.0 / std::min(n, kstop), p[1] / 100.0 / std::min(n, kstop), p[4] / 100.0 / std::min(n, kstop),
               p[2] / 100.0 / std::min(n, kstop), p[3] / 100.0 / std::min(n, kstop));
    }
    if (withq && kstop >= n) {          // Q^T A Q == T and Q^T Q == I (complete reductions only)
        std::vector<double> Q(A.size()), d(n), e(n), AQ(A.size());
        hipMemcpy(Q.data(), dQ, A.size() * 8, hipMemcpyDeviceToHost);
        hipMemcpy(d.data(), dD, n * 8, hipMemcpyDeviceToHost); hipMemcpy(e.data(), dE, n * 8, hipMemcpyDeviceToHost);
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) { double s = 0; for (int k = 0; k < n; ++k) s += A[(size_t)i * n + k] * Q[(size_t)k * n + j]; AQ[(size_t)i * n + j] = s; }
        double emax = 0, omax = 0;
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) {
                double s = 0, o = 0;
                for (int k = 0; k < n; ++k) { s += Q[(size_t)k * n + i] * AQ[(size_t)k * n + j]; o += Q[(size_t)k * n + i] * Q[(size_t)k * n + j]; }
                const double t = i == j ? d[i] : (i == j + 1 ? e[j] : (j == i + 1 ? e[i] : 0.0));
                emax = fmax(emax, fabs(s - t)); omax = fmax(omax, fabs(o - (i == j)));
            }
        printf("   Q check: |Q^T A Q - T|max %.3e   |Q^T Q - I|max %.3e\n", emax, omax);
    }
    return 0;
}
