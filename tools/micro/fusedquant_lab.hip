// What would "row-quantize inside the GEMM workgroup" (SURVEY 7 hard part 2, option 2: every CTA recomputes the amax of its rows)
// cost?  This lab times JUST that prologue in the GEMM's own geometry: a grid of tiles_m x tiles_n workgroups of 8 waves, each of
// which reads its 64 activation rows (bf16, K elements), reduces the row amax, quantizes the rows with the reference's arithmetic
// (IEEE division per element, rint, clamp) and parks the int8 codes in LDS where the MFMA loop would read them.  tiles_n workgroups
// repeat the same 64 rows -- that redundancy is the price of the fusion.  Compared against the stand-alone row-quantization launch
// (one wave per row, every row once) it would replace.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/micro/fusedquant_lab.hip -o build/fusedquant_lab
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#define HC(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// NP: 512-element passes per row (K <= NP * 512);  FAST: reciprocal multiply instead of the IEEE division (NOT bit-exact: lower bound)
template <int NP, bool FAST>
__global__ __launch_bounds__(512) void cta_quant(const uint16_t* __restrict__ x, int M, int K, int tiles_n, float* __restrict__ xs, int* __restrict__ sink) {
    extern __shared__ uint8_t lds[];  // [64][K] int8
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tile_m = blockIdx.x / tiles_n;
    int acc = 0;
    for (int rr = 0; rr < 8; ++rr) {
        const int r = wave * 8 + rr;
        int m = tile_m * 64 + r;
        if (m >= M) m = M - 1;
        const uint16_t* row = x + (size_t)m * K;
        uint4 raw[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int idx = p * 512 + lane * 8;
            raw[p] = *(const uint4*)(row + (idx < K ? idx : 0));
        }
        float v[NP][8];
        float amax = 0.0f;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const uint32_t w[4] = {raw[p].x, raw[p].y, raw[p].z, raw[p].w};
            const bool ok = p * 512 + lane * 8 < K;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                v[p][2 * i] = ok ? __uint_as_float(w[i] << 16) : 0.0f;
                v[p][2 * i + 1] = ok ? __uint_as_float(w[i] & 0xffff0000u) : 0.0f;
                amax = fmaxf(amax, fmaxf(fabsf(v[p][2 * i]), fabsf(v[p][2 * i + 1])));
            }
        }
        amax = wave_max(amax);
        const float scale = amax / 127.0f;
        const float rinv = 1.0f / scale;
        if (lane == 0 && blockIdx.x % tiles_n == 0) xs[m] = scale;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            uint32_t w0 = 0, w1 = 0;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float q = FAST ? v[p][e] * rinv : v[p][e] / scale;
                q = fminf(fmaxf(__builtin_rintf(q), -128.0f), 127.0f);
                const uint32_t b = (uint32_t)(int)q & 0xffu;
                if (e < 4) w0 |= b << (8 * e); else w1 |= b << (8 * (e - 4));
            }
            const int idx = p * 512 + lane * 8;
            if (idx < K) *(uint2*)(lds + (size_t)r * K + idx) = make_uint2(w0, w1);
            acc += (int)w0;
        }
    }
    __syncthreads();
    if (acc == 0x12345678) sink[0] = lds[threadIdx.x];  // keeps everything alive
}

template <int NP, bool FAST>
static void run(const char* what, const uint16_t* x, int M, int K, int N, float* xs, int* sink, hipStream_t s) {
    const int tiles_m = (M + 63) / 64, tiles_n = (N + 127) / 128;
    const size_t lds = (size_t)64 * K;
    HC(hipFuncSetAttribute((const void*)cta_quant<NP, FAST>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    auto launch = [&]() { hipLaunchKernelGGL((cta_quant<NP, FAST>), dim3(tiles_m * tiles_n), dim3(512), lds, s, x, M, K, tiles_n, xs, sink); };
    for (int i = 0; i < 3; ++i) launch();
    hipEvent_t e0, e1;
    HC(hipEventCreate(&e0)); HC(hipEventCreate(&e1));
    HC(hipEventRecord(e0, s));
    for (int i = 0; i < 50; ++i) launch();
    HC(hipEventRecord(e1, s));
    HC(hipEventSynchronize(e1));
    float ms;
    HC(hipEventElapsedTime(&ms, e0, e1));
    printf("  M=%5d K=%5d N=%5d  %3d x %2d workgroups  %-28s %6.2f us per launch (back-to-back, incl. the ~1.5 us boundary)\n", M, K, N, tiles_m, tiles_n, what, ms * 1e3 / 50);
}

int main() {
    hipStream_t s;
    HC(hipStreamCreate(&s));
    uint16_t* x; float* xs; int* sink;
    HC(hipMalloc(&x, (size_t)4096 * 2560 * 2)); HC(hipMalloc(&xs, 4096 * 4)); HC(hipMalloc(&sink, 4));
    HC(hipMemset(x, 0x3c, (size_t)4096 * 2560 * 2));
    printf("CTA-recomputes-amax prologue alone (no GEMM), 64-row x 128-column tile grid, 8 waves per workgroup:\n");
    run<3, false>("IEEE division (bit-exact)", x, 1024, 1280, 1280, xs, sink, s);
    run<3, true>("reciprocal multiply (NOT exact)", x, 1024, 1280, 1280, xs, sink, s);
    run<2, false>("IEEE division (bit-exact)", x, 4096, 640, 640, xs, sink, s);
    run<2, true>("reciprocal multiply (NOT exact)", x, 4096, 640, 640, xs, sink, s);
    run<3, false>("IEEE division (bit-exact)", x, 1024, 1280, 3840, xs, sink, s);
    run<3, false>("IEEE division (bit-exact)", x, 1024, 1280, 10240, xs, sink, s);
    return 0;
}
