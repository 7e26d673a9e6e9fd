// All-gather of tagged 8-byte granules among G workgroups (the hand-off of k_sytrd_*): does the ADDRESS LAYOUT of the granule
// buffer matter (all polls of a round hit the same few lines -> one memory channel)?  Line l of 16 granules is placed at
// byte offset l * line_stride.   hipcc --offload-arch=gfx950 -O3 tools/ag_spread.hip -o tools/_build/ag_spread
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned long long u64;
__global__ void allgather(u64 *buf, int G, int per, int rounds, u64 *out, int line_stride_g /*granules*/, size_t half /*granules between the two round buffers*/)
{
    const int g = blockIdx.x, tid = threadIdx.x, total = G * per;
    u64 t0 = wall_clock64();
    for (int r = 1; r <= rounds; ++r) {
        u64 *b = buf + (size_t)(r & 1) * half;
        for (int i = tid; i < per; i += blockDim.x) {
            const int slot = g * per + i;
            __hip_atomic_store(b + (size_t)(slot >> 4) * line_stride_g + (slot & 15), (u64)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        for (int slot = tid; slot < total; slot += blockDim.x) {
            const u64 *p = b + (size_t)(slot >> 4) * line_stride_g + (slot & 15);
            unsigned spins = 0;
            while (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (u64)r) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1u << 22)) break;
            }
        }
        __syncthreads();
    }
    if (tid == 0 && g == 0) out[0] = wall_clock64() - t0;
}
int main()
{
    u64 *buf, *out;
    const size_t bytes = 256u << 20;
    hipMalloc(&buf, bytes);
    hipMalloc(&out, 64);
    const int rounds = 2000;
    const int strides[] = {128, 256, 512, 1024, 2048, 4096, 4096 + 128, 8192 + 256, 65536 + 128, 1 << 20};
    for (int G : {64, 32})
        for (int per : {8, 16})
            for (int sb : strides) {
                const size_t lines = (size_t)(G * per + 15) / 16;
                const size_t half = (lines * sb / 8 + 4096) & ~(size_t)4095;
                if (2 * half * 8 > bytes) continue;
                hipMemset(buf, 0, 2 * half * 8);
                hipMemset(out, 0, 64);
                hipLaunchKernelGGL(allgather, dim3(G), dim3(256), 0, 0, buf, G, per, rounds, out, sb / 8, half);
                hipDeviceSynchronize();
                u64 h;
                hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost);
                printf("G=%2d per=%2d (%4d granules, %2zu lines) line stride %8d B: %.3f us per round\n", G, per, G * per, lines, sb, (double)h / 100.0 / rounds);
            }
    return 0;
}
