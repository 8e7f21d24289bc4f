// Development probe: sustained L2 -> CU bandwidth per CU for (a) global_load_dwordx4 -> VGPR, (b) global_load_lds_dwordx4.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// each block streams `iters` x (NT x 16 B) from a region of `region` bytes (power of two), L2 resident
template <int UNROLL>
__global__ __launch_bounds__(256) void k_vgpr(const uint4* __restrict__ src, uint4* out, int iters, size_t region_vecs) {
    size_t base = ((size_t)blockIdx.x * 9973 * 256 + threadIdx.x) & (region_vecs - 1);
    uint4 acc = {0, 0, 0, 0};
    for (int i = 0; i < iters; i += UNROLL) {
        uint4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = src[(base + (size_t)(i + u) * 256) & (region_vecs - 1)];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) { acc.x ^= v[u].x; acc.y ^= v[u].y; acc.z ^= v[u].z; acc.w ^= v[u].w; }
    }
    if (acc.x == 0x12345678) out[0] = acc;
}
template <int UNROLL>
__global__ __launch_bounds__(256) void k_dma(const uint4* __restrict__ src, uint4* out, int iters, size_t region_vecs) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    size_t base = ((size_t)blockIdx.x * 9973 * 256 + threadIdx.x) & (region_vecs - 1);
    for (int i = 0; i < iters; i += UNROLL) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
            __builtin_amdgcn_global_load_lds((gptr_t)(src + ((base + (size_t)(i + u) * 256) & (region_vecs - 1))),
                                             (lptr_t)(lds + ((u * 4 + wave) * 1024)), 16, 0, 0);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(UNROLL) : "memory");  // keep ~UNROLL in flight
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lds[threadIdx.x] == 0x7b && iters < 0) out[0] = *(uint4*)lds;
}
template <typename F> float time_it(F launch, hipStream_t s) {
    launch(); CK(hipStreamSynchronize(s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < 5; ++i) launch();
    CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3f / 5;
}
int main() {
    hipStream_t s; CK(hipStreamCreate(&s));
    uint4 *src, *out; CK(hipMalloc(&src, 512 << 20)); CK(hipMalloc(&out, 1 << 20)); CK(hipMemset(src, 1, 512 << 20));
    CK(hipFuncSetAttribute((const void*)k_dma<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    const int iters = 2048;
    for (size_t region : {(size_t)1 << 20, (size_t)8 << 20, (size_t)256 << 20}) {
        for (int bpc : {1, 2, 4}) {
            int grid = 256 * bpc;
            double bytes = (double)grid * 256 * 16 * iters;
            float t = time_it([&] { hipLaunchKernelGGL(k_vgpr<8>, dim3(grid), dim3(256), 0, s, src, out, iters, region / 16); }, s);
            printf("VGPR  region=%4zu MB blocks/CU=%d: %.1f us  %.2f TB/s  %.1f B/clk/CU(@2.1GHz)\n", region >> 20, bpc, t, bytes / t / 1e6, bytes / t / 1e6 * 1e12 / 256 / 2.1e9);
            float t2 = time_it([&] { hipLaunchKernelGGL(k_dma<8>, dim3(grid), dim3(256), 32 * 1024, s, src, out, iters, region / 16); }, s);
            printf("DMA   region=%4zu MB blocks/CU=%d: %.1f us  %.2f TB/s  %.1f B/clk/CU(@2.1GHz)\n", region >> 20, bpc, t2, bytes / t2 / 1e6, bytes / t2 / 1e6 * 1e12 / 256 / 2.1e9);
        }
    }
    return 0;
}
