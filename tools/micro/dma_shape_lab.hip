// Development probe (round 3): LDS-DMA fill rate per CU as a function of the SHAPE of a 1-KiB piece -- RP rows x RB bytes out of a
// row-major matrix with an 8-KiB row pitch (what a GEMM stage fetches): 16 x 64 B (BK = 64), 8 x 128 B (BK = 128, full cache lines),
// 4 x 256 B, 1 x 1024 B -- for an L2-resident and a MALL/HBM-sized region, 8 waves per CU, `DEPTH` pieces per wave in flight.
// Optionally with a ds_read_b128 stream beside it (LDS port contention) -- second argument.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/micro/dma_shape_lab.hip -o build/dma_shape_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef int v4i __attribute__((ext_vector_type(4)));

template <int RB, int DEPTH, int READS, int SWZ = 0>
__global__ __launch_bounds__(512) void k_fill(const uint8_t* __restrict__ src, uint4* out, int iters, int rows_total, int ld) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    constexpr int LPR = RB / 16, RP = 1024 / RB;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // blocks of one XCD (b % 8) walk the same rows, like the tiles of one rasterization patch sharing A strips / B slabs
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int row0 = (xcd * 1237 + (j % 6) * 256) % rows_total;
    v4i accv = {0, 0, 0, 0};
    int kofs = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < DEPTH; ++u) {
            const int r = (row0 + (wave * DEPTH + u) * RP + lane / LPR) % rows_total;
            // SWZ: the 16-byte chunks of a row are fetched in XOR-permuted lane order (the source side of the GEMM's LDS swizzle)
            const int ch = SWZ ? ((lane % LPR) ^ ((r >> 1) & (LPR - 1))) : (lane % LPR);
            const uint8_t* s = src + (size_t)r * ld + ((kofs + ch * 16) & (ld - 1));
            __builtin_amdgcn_global_load_lds((gptr_t)s, (lptr_t)(lds + (((it & 1) * 8 * DEPTH + wave * DEPTH + u) * 1024)), 16, 0, 0);
        }
        if constexpr (READS > 0) {
#pragma unroll
            for (int q = 0; q < READS; ++q) {
                const v4i t = *(const v4i*)(lds + 96 * 1024 + ((wave * READS + q) & 31) * 1024 + lane * 16);
                accv += t;
            }
        }
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH) : "memory");  // the previous iteration's pieces have landed
        kofs += RB;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (accv[0] == 0x12345678 && iters < 0) out[0] = *(uint4*)lds;
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
template <int RB, int DEPTH, int READS, int SWZ = 0>
void run(const uint8_t* src, uint4* out, int rows_total, hipStream_t s) {
    const int iters = 512, ld = 8192;
    auto kern = k_fill<RB, DEPTH, READS, SWZ>;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    float t = time_it([&] { hipLaunchKernelGGL(kern, dim3(256), dim3(512), 128 * 1024, s, src, out, iters, rows_total, ld); }, s);
    const double bytes = 256.0 * 8 * DEPTH * 1024 * iters;
    printf("  %s rows x bytes %2d x %4d  depth %d  reads/iter %2d  region %4d MB: %8.1f us  %6.2f TB/s  %5.1f B/clk/CU (@2.1 GHz)\n", SWZ ? "swizzled" : "linear  ", 1024 / RB, RB, DEPTH, READS,
           (int)((size_t)rows_total * ld >> 20), t, bytes / t / 1e6, bytes / t / 1e6 * 1e12 / 256 / 2.1e9);
}
int main() {
    hipStream_t s; CK(hipStreamCreate(&s));
    uint8_t* src; uint4* out;
    CK(hipMalloc(&src, (size_t)512 << 20)); CK(hipMalloc(&out, 1 << 20)); CK(hipMemset(src, 1, (size_t)512 << 20));
    for (int rows : {256, 2048, 65536}) {  // 2 MB (L2 resident), 16 MB (MALL), 512 MB (HBM)
        run<64, 4, 0>(src, out, rows, s);  run<128, 4, 0>(src, out, rows, s);  run<256, 4, 0>(src, out, rows, s);  run<1024, 4, 0>(src, out, rows, s);
        run<64, 2, 0>(src, out, rows, s);  run<128, 2, 0>(src, out, rows, s);
        run<64, 4, 12>(src, out, rows, s); run<128, 4, 12>(src, out, rows, s); run<1024, 4, 12>(src, out, rows, s);
        run<128, 4, 0, 1>(src, out, rows, s); run<128, 2, 0, 1>(src, out, rows, s); run<64, 4, 0, 1>(src, out, rows, s);
    }
    return 0;
}
