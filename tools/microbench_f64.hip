// microbench_f64.hip — measures, on the box, the numbers the roofline is priced
// against ("measure v_mfma_f64_16x16x4_f64 issue rate on the box first"):
//   1. fp64 MFMA issue rate (1 and 2 waves per SIMD, 4 independent accumulators)
//   2. fp64 VALU FMA rate, and MFMA + VALU co-issue on one SIMD
//   3. HBM streaming read bandwidth (16 B/lane, non-temporal)
// build: hipcc --offload-arch=gfx950 -O3 tools/microbench_f64.hip -o tools/microbench_f64
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double double4_t __attribute__((ext_vector_type(4)));
typedef double double2_t __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int NACC>
__global__ void k_mfma(double *out, int iters, double a0, double b0)
{
    double4_t acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = double4_t{0, 0, 0, 0};
    double a = a0 + threadIdx.x * 1e-9, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_valu(double *out, int iters, double a0, double b0)
{
    double x[16];
    for (int i = 0; i < 16; ++i) x[i] = a0 + i + threadIdx.x * 1e-9;
    const double b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 16; ++i) x[i] = __builtin_fma(x[i], b, b);
    }
    double s = 0;
    for (int i = 0; i < 16; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// waves with even wave id issue MFMA, odd issue VALU FMA (2 waves per SIMD: one of each)
__global__ void k_mixed(double *out, int iters, double a0, double b0)
{
    const int wave = threadIdx.x >> 6;
    const bool do_mfma = (__builtin_amdgcn_readfirstlane(wave) & 4) == 0;   // waves 0-3 -> MFMA, 4-7 -> VALU
    double s = 0;
    if (do_mfma) {
        double4_t acc[4];
        for (int i = 0; i < 4; ++i) acc[i] = double4_t{0, 0, 0, 0};
        double a = a0 + threadIdx.x * 1e-9, b = b0;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
        }
        for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    } else {
        double x[16];
        for (int i = 0; i < 16; ++i) x[i] = a0 + i + threadIdx.x * 1e-9;
        const double b = b0;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < 16; ++i) x[i] = __builtin_fma(x[i], b, b);
        }
        for (int i = 0; i < 16; ++i) s += x[i];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_stream(const double2_t *__restrict__ in, double *out, size_t n_vec)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    double2_t acc = double2_t{0, 0};
    for (; i + 3 * stride < n_vec; i += 4 * stride) {
        double2_t v0 = __builtin_nontemporal_load(in + i), v1 = __builtin_nontemporal_load(in + i + stride);
        double2_t v2 = __builtin_nontemporal_load(in + i + 2 * stride), v3 = __builtin_nontemporal_load(in + i + 3 * stride);
        acc += v0 + v1 + v2 + v3;
    }
    for (; i < n_vec; i += stride) acc += in[i];
    if (acc.x + acc.y == 1.2345e300) out[0] = acc.x;
}

template <class F>
double time_ms(F f, int reps)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    f();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) f();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}

int main()
{
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    printf("device %s, %d CUs, clock %d kHz\n", p.gcnArchName, cus, p.clockRate);
    double *out;
    CK(hipMalloc(&out, sizeof(double) * 1024 * 512 * 8));
    const int iters = 2000;
    // 1 wave/SIMD: 256 threads per block, 1 block per CU; 2 waves/SIMD: 512 threads
    for (int threads : {256, 512, 1024}) {
        double ms = time_ms([&] { hipLaunchKernelGGL(k_mfma<4>, dim3(cus), dim3(threads), 0, 0, out, iters, 1.0, 1e-3); }, 5);
        double nmfma = (double)cus * (threads / 64) * iters * 8 * 4;
        double tf = nmfma * 2048.0 / (ms * 1e-3) / 1e12;
        double cyc_per_mfma_per_simd = (ms * 1e-3) * 2.4e9 / ((double)(threads / 64) / 4.0 * iters * 32);
        printf("MFMA f64 16x16x4, %4d thr/CU (%d waves/SIMD), 4 acc: %.2f TFLOP/s, %.1f cycles/MFMA/SIMD @2.4GHz\n",
               threads, threads / 256, tf, cyc_per_mfma_per_simd);
    }
    {
        double ms = time_ms([&] { hipLaunchKernelGGL(k_mfma<1>, dim3(cus), dim3(256), 0, 0, out, iters, 1.0, 1e-3); }, 5);
        double nmfma = (double)cus * 4 * iters * 8;
        printf("MFMA f64 dependent chain (1 acc, 1 wave/SIMD): %.2f TFLOP/s, %.1f cycles/MFMA\n",
               nmfma * 2048.0 / (ms * 1e-3) / 1e12, (ms * 1e-3) * 2.4e9 / (iters * 8.0));
    }
    for (int threads : {256, 512, 1024}) {
        double ms = time_ms([&] { hipLaunchKernelGGL(k_valu, dim3(cus), dim3(threads), 0, 0, out, iters, 1.0, 0.999); }, 5);
        double nfma = (double)cus * threads * iters * 4 * 16;
        printf("VALU v_fma_f64, %4d thr/CU: %.2f TFLOP/s\n", threads, nfma * 2.0 / (ms * 1e-3) / 1e12);
    }
    {
        double ms = time_ms([&] { hipLaunchKernelGGL(k_mixed, dim3(cus), dim3(512), 0, 0, out, iters, 1.0, 0.999); }, 5);
        double mf = (double)cus * 4 * iters * 8 * 4 * 2048.0, vf = (double)cus * 256 * iters * 4 * 16 * 2.0;
        printf("MIXED 4 MFMA waves + 4 VALU waves per CU: %.3f ms -> MFMA %.2f + VALU %.2f = %.2f TFLOP/s\n", ms,
               mf / (ms * 1e-3) / 1e12, vf / (ms * 1e-3) / 1e12, (mf + vf) / (ms * 1e-3) / 1e12);
    }
    {
        const size_t bytes = (size_t)4 << 30;
        double2_t *buf;
        CK(hipMalloc(&buf, bytes));
        CK(hipMemset(buf, 1, bytes));
        for (int bpc : {4, 8, 16}) {
            double ms = time_ms([&] { hipLaunchKernelGGL(k_stream, dim3(cus * bpc), dim3(256), 0, 0, buf, out, bytes / 16); }, 5);
            printf("HBM stream read 4 GiB, %2d blocks/CU: %.1f GB/s\n", bpc, bytes / (ms * 1e-3) / 1e9);
        }
        hipFree(buf);
    }
    hipFree(out);
    return 0;
}
