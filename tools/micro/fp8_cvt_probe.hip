// Does gfx950's v_cvt_pk_fp8_f32 equal the software float -> e4m3fn (OCP, RNE) conversion of sdnq_dev.h on the clamped range?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I. tools/micro/fp8_cvt_probe.hip -o /tmp/fp8p && /tmp/fp8p
#include "../../sdnq_amd/csrc/sdnq_dev.h"
#include <cstdio>
__global__ void k(unsigned long long* bad, unsigned* first, unsigned lo, unsigned n) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float f = __uint_as_float(lo + i);
    for (int sgn = 0; sgn < 2; ++sgn) {
        const float v = sgn ? -f : f;
        const unsigned sw = f32_to_e4m3fn(v);
        const unsigned hw = (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(v, 0.0f, 0, false) & 0xffu;
        if (sw != hw) { if (atomicAdd(bad, 1ull) == 0) { first[0] = __float_as_uint(v); first[1] = sw; first[2] = hw; } }
    }
}
int main() {
    unsigned long long* bad; unsigned* first;
    hipMalloc(&bad, 8); hipMalloc(&first, 12); hipMemset(bad, 0, 8); hipMemset(first, 0, 12);
    // every float from 0 up to 448.0 (0x43e00000): 1.14e9 values, both signs
    const unsigned hi = 0x43e00000u;
    for (unsigned lo = 0; lo <= hi; lo += (1u << 26)) {
        const unsigned n = (hi - lo + 1) < (1u << 26) ? (hi - lo + 1) : (1u << 26);
        hipLaunchKernelGGL(k, dim3((n + 255) / 256), dim3(256), 0, 0, bad, first, lo, n);
    }
    hipDeviceSynchronize();
    unsigned long long b; unsigned f3[3];
    hipMemcpy(&b, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(f3, first, 12, hipMemcpyDeviceToHost);
    printf("mismatches over all floats in [-448, 448]: %llu  (first: bits %08x sw %02x hw %02x)\n", b, f3[0], f3[1], f3[2]);
    return 0;
}
