// Register-resident MFMA issue-rate micro-benchmark ("verify the int8 rate with a register-resident MFMA loop and
// record the measured ceiling next to the datasheet number").  No memory traffic: every wave keeps NACC independent accumulators
// and issues MFMAs back to back.  Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o build/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));

constexpr int NACC = 4;

__device__ __forceinline__ unsigned hash32(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

// DATA 0: small integers (mostly-zero upper bytes, the r01 probe); 1: full-entropy bytes in every operand register, a different
// set per accumulator slot (what a GEMM on quantized tensors feeds the pipe: operand toggling costs power and the part clocks down)
template <int KIND, int DATA>
__global__ __launch_bounds__(256) void mfma_loop(int iters, float* sink, long long* cycles) {
    const int lane = threadIdx.x;
    v4i a = {lane, lane + 1, lane + 2, lane + 3}, b = {lane * 3, 7, 9, 11};
    v8i a8 = {lane, 1, 2, 3, 4, 5, 6, 7}, b8 = {lane, 7, 6, 5, 4, 3, 2, 1};
    v4i ar[NACC], br[NACC];
    v8i ar8[NACC], br8[NACC];
    for (int n = 0; n < NACC; ++n) {
        for (int r = 0; r < 4; ++r) {
            unsigned ha = hash32(lane * 64 + n * 8 + r + blockIdx.x * 4099u), hb = hash32(lane * 64 + n * 8 + r + 77777u);
            if (KIND == 2) { ha = (ha & 0x807f807fu) | 0x3f003f00u; hb = (hb & 0x807f807fu) | 0x3f003f00u; }  // bf16 in [-2,-1] u [1,2]: no inf / nan
            if (KIND == 1) { ha &= 0xf7f7f7f7u; hb &= 0xf7f7f7f7u; }  // e4m3: keep away from the NaN code
            ar[n][r] = (int)ha; br[n][r] = (int)hb;
            ar8[n][r] = (int)ha; br8[n][r] = (int)hb; ar8[n][r + 4] = (int)hash32(ha) & (int)0xf7f7f7f7u; br8[n][r + 4] = (int)hash32(hb) & (int)0xf7f7f7f7u;
        }
    }
    v16i ci[NACC];
    v16f cf[NACC];
    for (int n = 0; n < NACC; ++n)
        for (int r = 0; r < 16; ++r) { ci[n][r] = 0; cf[n][r] = 0.0f; }
    const long long t0 = wall_clock64();
    const long long c0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int n = 0; n < NACC; ++n) {
            const v4i xa = DATA ? ar[n] : a, xb = DATA ? br[n] : b;
            const v8i xa8 = DATA ? ar8[n] : a8, xb8 = DATA ? br8[n] : b8;
            if constexpr (KIND == 0) ci[n] = __builtin_amdgcn_mfma_i32_32x32x32_i8(xa, xb, ci[n], 0, 0, 0);
            else if constexpr (KIND == 1) cf[n] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(xa8, xb8, cf[n], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
            else cf[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(v8bf, xa), __builtin_bit_cast(v8bf, xb), cf[n], 0, 0, 0);
        }
    }
    const long long c1 = clock64();
    const long long t1 = wall_clock64();
    float s = 0.0f;
    for (int n = 0; n < NACC; ++n)
        for (int r = 0; r < 16; ++r) s += (float)ci[n][r] + cf[n][r];
    if (s == 123.456f) sink[0] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { cycles[0] = c1 - c0; cycles[1] = t1 - t0; }
}

template <int KIND, int DATA>
void run(const char* name, double ops_per_mfma, int waves_per_simd, int cus, int iters) {
    float* sink; long long* cyc;
    hipMalloc(&sink, 4); hipMalloc(&cyc, 16);
    const int blocks = cus * waves_per_simd;  // 256 threads = 4 waves = one per SIMD
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((mfma_loop<KIND, DATA>), dim3(blocks), dim3(256), 0, 0, 100, sink, cyc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((mfma_loop<KIND, DATA>), dim3(blocks), dim3(256), 0, 0, iters, sink, cyc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[2]; hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
    const double mfmas = (double)blocks * 4 * iters * NACC;
    const double per_wave_cycles = (double)h[0] / (iters * NACC);  // shader clocks per MFMA as seen by one wave
    const double shader_mhz = (double)h[0] / ((double)h[1] / 100.0);  // wall_clock64 ticks at 100 MHz
    printf("%-28s %s %7d iters waves/SIMD %d: %8.1f TOP/s  (%.3f ms; one wave issues an MFMA every %.1f shader clocks; shader clock %.0f MHz during the loop)\n",
           name, DATA ? "random-bytes" : "small-ints  ", iters, waves_per_simd, mfmas * ops_per_mfma / (ms * 1e-3) / 1e12, ms, per_wave_cycles, shader_mhz);
    hipFree(sink); hipFree(cyc);
}

int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("%s  CUs %d  clockRate %d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
    // short (about 1 ms) and long (tens of ms: long enough for the power management to settle) loops, both operand fills
    for (int iters : {20000, 400000}) {
        for (int w : {1, 2}) {
            run<0, 0>("v_mfma_i32_32x32x32_i8", 2.0 * 32 * 32 * 32, w, p.multiProcessorCount, iters);
            run<0, 1>("v_mfma_i32_32x32x32_i8", 2.0 * 32 * 32 * 32, w, p.multiProcessorCount, iters);
        }
        run<1, 0>("v_mfma_scale_f32_32x32x64_f8", 2.0 * 32 * 32 * 64, 2, p.multiProcessorCount, iters / 2);
        run<1, 1>("v_mfma_scale_f32_32x32x64_f8", 2.0 * 32 * 32 * 64, 2, p.multiProcessorCount, iters / 2);
        run<2, 0>("v_mfma_f32_32x32x16_bf16", 2.0 * 32 * 32 * 16, 2, p.multiProcessorCount, iters);
        run<2, 1>("v_mfma_f32_32x32x16_bf16", 2.0 * 32 * 32 * 16, 2, p.multiProcessorCount, iters);
    }
    return 0;
}
