// Hop latency of the tagged-granule hand-off (jcdf_eig.hpp) between two workgroups on the SAME XCD and on
// DIFFERENT XCDs, agent-scope relaxed atomics in both cases.  hipcc --offload-arch=gfx950 -O3 -o xcd_pingpong xcd_pingpong.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned long long u64;
// bounded spin: gives up after ~20 ms so a protocol error can never hang the device
#define SPIN_UNTIL(cond) do { u64 t_ = wall_clock64(); while (!(cond)) { __builtin_amdgcn_s_sleep(1); if (wall_clock64() - t_ > 2000000ULL) break; } } while (0)

__device__ __forceinline__ unsigned xcc_id()
{
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xf;
}

// blocks a and b bounce a counter `rounds` times; every other block exits.  out[0] = ticks (100 MHz), out[1..2] = xcc ids
__global__ void pingpong(u64 *flag, int a, int b, int rounds, u64 *out, int nelem)
{
    const int me = blockIdx.x;
    if (me != a && me != b) return;
    const int tid = threadIdx.x;
    if (tid == 0) out[me == a ? 1 : 2] = xcc_id();
    u64 t0 = wall_clock64();
    for (int r = 1; r <= rounds; ++r) {
        // a publishes odd tags (2r-1), b answers with even tags (2r); nelem granules per message, one per thread
        if (me == a) {
            if (tid < nelem) __hip_atomic_store(flag + tid, (u64)(2 * r - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (tid < nelem) SPIN_UNTIL(__hip_atomic_load(flag + 1024 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (u64)(2 * r));
        } else {
            if (tid < nelem) SPIN_UNTIL(__hip_atomic_load(flag + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (u64)(2 * r - 1));
            if (tid < nelem) __hip_atomic_store(flag + 1024 + tid, (u64)(2 * r), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
    }
    if (tid == 0 && me == a) out[0] = wall_clock64() - t0;
}

// all-gather: G participating blocks (those with blockIdx % stride == 0), each publishes `per` granules per round and
// reads all G*per; rounds back to back.  out[0] = ticks
// SCOPE: __HIP_MEMORY_SCOPE_AGENT (sc1) or __HIP_MEMORY_SCOPE_WORKGROUP (sc0: coherent in the XCD's L2 only -> valid
// only when every participant runs on the same XCD)
// interleave != 0: granule i of a workgroup lives at slot i*G + g (neighbouring slots belong to different
// workgroups, as when slot == matrix column and columns are dealt out cyclically) instead of g*per + i
template <int SCOPE>
__global__ void allgather(u64 *buf, int G, int stride, int per, int rounds, u64 *out, int interleave)
{
    if (blockIdx.x % stride != 0) return;
    const int g = blockIdx.x / stride;
    if (g >= G) return;
    const int tid = threadIdx.x, total = G * per;
    u64 t0 = wall_clock64();
    for (int r = 1; r <= rounds; ++r) {
        u64 *b = buf + (size_t)(r & 1) * total;
        for (int i = tid; i < per; i += blockDim.x)
            __hip_atomic_store(b + (interleave ? i * G + g : g * per + i), (u64)r, __ATOMIC_RELAXED, SCOPE);
        for (int i = tid; i < total; i += blockDim.x)
            SPIN_UNTIL(__hip_atomic_load(b + i, __ATOMIC_RELAXED, SCOPE) == (u64)r);
        __syncthreads();
    }
    if (tid == 0 && g == 0) out[0] = wall_clock64() - t0;
}

// Two granules per request: one 16-byte sc1 load per lane.  Each 8-byte half carries its own tag, so only a tear
// INSIDE an aligned 8-byte half could hurt, and a 16-byte aligned lane access is served from one line read.
__device__ __forceinline__ void load2(const u64 *p, u64 &a, u64 &b)
{
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    u32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    a = ((u64)v.y << 32) | v.x;
    b = ((u64)v.w << 32) | v.z;
}

__global__ void allgather_wide(u64 *buf, int G, int stride, int per, int rounds, u64 *out)
{
    if (blockIdx.x % stride != 0) return;
    const int g = blockIdx.x / stride;
    if (g >= G) return;
    const int tid = threadIdx.x, total = G * per;
    u64 t0 = wall_clock64();
    for (int r = 1; r <= rounds; ++r) {
        u64 *b = buf + (size_t)(r & 1) * total;
        for (int i = tid; i < per; i += blockDim.x)
            __hip_atomic_store(b + g * per + i, ((u64)r << 32) | (u64)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int i = 2 * tid; i < total; i += 2 * blockDim.x) {
            u64 x = 0, y = 0;
            u64 t_ = wall_clock64();
            for (;;) {
                load2(b + i, x, y);
                // a granule is accepted on its LOW dword (where the sytrd tag lives): is the HIGH dword from the same store?
                if ((unsigned)x == (unsigned)r && (x >> 32) != (u64)r) atomicAdd((unsigned long long *)(out + 4), 1ULL);
                if ((unsigned)y == (unsigned)r && (y >> 32) != (u64)r) atomicAdd((unsigned long long *)(out + 4), 1ULL);
                if ((unsigned)x == (unsigned)r && (unsigned)y == (unsigned)r) break;
                __builtin_amdgcn_s_sleep(1);
                if (wall_clock64() - t_ > 2000000ULL) break;
            }
        }
        __syncthreads();
    }
    if (tid == 0 && g == 0) out[0] = wall_clock64() - t0;
}

int main()
{
    u64 *flag, *out;
    hipMalloc(&flag, 1 << 20);
    hipMalloc(&out, 64);
    const int rounds = 2000;
    struct { int a, b; const char *what; } cases[] = {{0, 8, "same XCD (blocks 0, 8)"}, {0, 1, "different XCD (blocks 0, 1)"},
                                                      {0, 16, "same XCD (blocks 0, 16)"}, {0, 4, "different XCD (blocks 0, 4)"}};
    for (int nelem : {1, 64, 256})
        for (auto &c : cases) {
            hipMemset(flag, 0, 1 << 20);
            hipMemset(out, 0, 64);
            hipLaunchKernelGGL(pingpong, dim3(64), dim3(256), 0, 0, flag, c.a, c.b, rounds, out, nelem);
            hipDeviceSynchronize();
            u64 h[3];
            hipMemcpy(h, out, 24, hipMemcpyDeviceToHost);
            printf("pingpong nelem=%3d %-30s xcc %llu/%llu : %.3f us per hop\n", nelem, c.what, h[1], h[2],
                   (double)h[0] / 100.0 / (2.0 * rounds));
        }
    struct { int G, stride, per; } ag[] = {{16, 8, 32}, {16, 1, 32}, {32, 8, 16}, {64, 1, 8}, {64, 4, 8}, {8, 8, 64}, {8, 1, 64}, {16, 8, 64}, {64, 1, 16}, {5, 1, 8}, {64, 1, 1}};
    for (int il = 0; il < 2; ++il)
    for (auto &c : ag) {
        hipMemset(flag, 0, 1 << 20);
        hipMemset(out, 0, 64);
        hipLaunchKernelGGL(allgather<__HIP_MEMORY_SCOPE_AGENT>, dim3(c.G * c.stride), dim3(256), 0, 0, flag, c.G, c.stride, c.per, rounds, out, il);
        hipDeviceSynchronize();
        u64 h[1];
        hipMemcpy(h, out, 8, hipMemcpyDeviceToHost);
        printf("allgather %s G=%2d stride=%d (%s) %3d granules each (%d total): %.3f us per round\n", il ? "interleaved" : "contiguous ", c.G, c.stride,
               c.stride % 8 == 0 ? "one XCD" : "spread", c.per, c.G * c.per, (double)h[0] / 100.0 / rounds);
    }
    struct { int G, per; } wd[] = {{64, 8}, {64, 16}, {16, 32}, {8, 64}};
    for (auto &c : wd) {
        hipMemset(flag, 0, 1 << 20);
        hipMemset(out, 0, 64);
        hipLaunchKernelGGL(allgather_wide, dim3(c.G), dim3(256), 0, 0, flag, c.G, 1, c.per, rounds, out);
        hipDeviceSynchronize();
        u64 h[5];
        hipMemcpy(h, out, 40, hipMemcpyDeviceToHost);
        printf("allgather 16-byte polls     G=%2d spread %3d granules each (%d total): %.3f us per round, torn granules: %llu\n", c.G, c.per,
               c.G * c.per, (double)h[0] / 100.0 / rounds, h[4]);
    }
    // (workgroup-scope (sc0) granules were tried here for participants on one XCD: the loads are served from the CU's
    //  L1 and never see the other CU's store - every spin ran into its timeout.  Agent scope is the only usable one.)
    return 0;
}
