// How fast can 256 workgroups write a [M][N] bf16 matrix tile by tile?  (the epilogue pattern of the scaled matmul, without the matmul)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef int v4i __attribute__((ext_vector_type(4)));
#define HC(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int NT_STORE>
__global__ __launch_bounds__(512) void tile_write(uint8_t* out, int M, int N, int BM, int BN, int tiles_n) {
    const int tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;
    const int ppr = BN * 2 / 16;  // 16-byte pieces per tile row
    const v4i val = {(int)blockIdx.x, (int)threadIdx.x, 3, 4};
    for (int v = threadIdx.x; v < BM * ppr; v += blockDim.x) {
        const int r = v / ppr, c = v % ppr;
        const long long gm = (long long)tm * BM + r, gn = (long long)tn * BN + c * 8;
        if (gm >= M || gn >= N) continue;
        v4i* dst = (v4i*)(out + (gm * N + gn) * 2);
        if (NT_STORE) __builtin_nontemporal_store(val, dst);
        else *dst = val;
    }
}

int main() {
    const int M = 1024, N = 10240;
    uint8_t* out;
    HC(hipMalloc(&out, (size_t)M * N * 2));
    hipStream_t s;
    HC(hipStreamCreate(&s));
    const int cfg[][2] = {{64, 128}, {256, 160}, {128, 320}, {256, 256}, {64, 10240}, {16, 10240}, {4, 10240}};
    for (auto& c : cfg) {
        const int bm = c[0], bn = c[1], tiles_n = (N + bn - 1) / bn, tiles = ((M + bm - 1) / bm) * tiles_n;
        for (int nt = 0; nt < 2; ++nt) {
            hipEvent_t e0, e1;
            HC(hipEventCreate(&e0)); HC(hipEventCreate(&e1));
            auto launch = [&]() {
                if (nt) hipLaunchKernelGGL(tile_write<1>, dim3(tiles), dim3(512), 0, s, out, M, N, bm, bn, tiles_n);
                else hipLaunchKernelGGL(tile_write<0>, dim3(tiles), dim3(512), 0, s, out, M, N, bm, bn, tiles_n);
            };
            for (int i = 0; i < 3; ++i) launch();
            HC(hipEventRecord(e0, s));
            for (int i = 0; i < 50; ++i) launch();
            HC(hipEventRecord(e1, s));
            HC(hipEventSynchronize(e1));
            float ms;
            HC(hipEventElapsedTime(&ms, e0, e1));
            printf("tile %3d x %5d (%4d WGs) %s: %6.2f us  %.2f TB/s\n", bm, bn, tiles, nt ? "nt   " : "plain", ms * 1e3 / 50, (double)M * N * 2 / (ms / 50 * 1e-3) / 1e12);
        }
    }
    return 0;
}
