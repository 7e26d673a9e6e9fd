// Dependent-chain latencies (shader cycles, one wave alone on a SIMD) of the building blocks of the bulge-chasing kernels.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/lat_bench.hip -o tools/_build/lat_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../juliachem.jl_amd/csrc/jcdf_gemm.hpp"
#include "../juliachem.jl_amd/csrc/jcdf_sbr.hpp"
using namespace jcdf;

__global__ void k_lat(double *out, unsigned long long *cyc, double seed)
{
    __shared__ double sh[256];
    __shared__ int flag[4];
    const int lane = threadIdx.x & 63;
    double x = seed + lane * 1e-3;
    const int R = 64;
    unsigned long long t0, t1;
    sh[lane] = x; sh[64 + lane] = x;
    if (lane < 4) flag[lane] = 1;
    __syncthreads();
#define TIMEIT(slot, ...)                                      \
    t0 = __builtin_amdgcn_s_memtime();                         \
    for (int i = 0; i < R; ++i) { __VA_ARGS__; }                      \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        \
    t1 = __builtin_amdgcn_s_memtime();                         \
    if (lane == 0) cyc[slot] = (t1 - t0) / R;
    TIMEIT(0, x = x * 1.0000001 + 1e-9)                                     // fma chain
    TIMEIT(1, x = rows_sum(x) * 0.25)                                       // two permlane-swap stages
    TIMEIT(2, x = row16_sum(x) * 0.0625)                                    // four dpp stages
    TIMEIT(3, x = lane0_f64(x) + 1e-9 * lane)                               // readfirstlane x2 -> vgpr
    { double tau, beta, scale;
      TIMEIT(4, house_scalars(x, 0.5, tau, beta, scale); x = 1.0 + 0.1 * scale + 1e-3 * tau + 1e-9 * beta) }
    TIMEIT(5, sh[lane] = x; x = sh[(lane + 1) & 63] + 1e-9)                  // LDS write -> read round trip
    TIMEIT(6, x += sh[128 + ((int)x & 63)] )                                  // dependent LDS read
    TIMEIT(7, int p = __builtin_amdgcn_readfirstlane(__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)); x += p * 1e-9)   // poll
    TIMEIT(8, __hip_atomic_store(flag + 1, i, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); x += 1e-9)   // release store
    TIMEIT(9, x = sqrt(x * x + 0.5))                                          // IEEE sqrt
    TIMEIT(10, x = 1.0 / (x + 0.5))                                           // IEEE division
    TIMEIT(11, x = dpp_f64<0x150, 0xf>(x) + 1e-9 * lane)                      // row_newbcast
    TIMEIT(12, x = __builtin_amdgcn_rsq(x + 1.0) + 1.0)
    TIMEIT(13, double a = x, b = x + 1.0; x = pair_sum32(a, b) * 0.5)
    out[threadIdx.x] = x;
}

int main()
{
    double *out; unsigned long long *cyc;
    hipMalloc(&out, 64 * 8); hipMalloc(&cyc, 32 * 8);
    hipMemset(cyc, 0, 32 * 8);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k_lat, dim3(1), dim3(64), 0, 0, out, cyc, 1.0);
    unsigned long long h[32];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    const char *names[] = {"fma f64 (dependent)", "rows_sum (2 permlane stages) + mul", "row16_sum (4 dpp stages) + mul", "lane0_f64 (readfirstlane) + fma",
                           "house_scalars + 3 fma", "LDS write -> read", "dependent LDS read", "LDS poll (atomic load + readfirstlane)",
                           "LDS release store", "IEEE sqrt + fma", "IEEE division + add", "row_newbcast + fma", "v_rsq_f64 + 2 add", "pair_sum32 + add + mul"};
    for (int i = 0; i < 14; ++i) printf("%-45s %6llu cycles\n", names[i], h[i]);
    return 0;
}
