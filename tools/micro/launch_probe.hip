// Development probe: what does a launch cost on MI355X as a function of LDS size / grid / barriers / a dependent load chain?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__global__ __launch_bounds__(256) void k_empty(int* out, int nbar) {
    extern __shared__ int lds[];
    for (int i = 0; i < nbar; ++i) __builtin_amdgcn_s_barrier();
    if (nbar < 0) out[threadIdx.x] = lds[threadIdx.x];
}
// dependent chain of `depth` global loads (pointer chasing over a permutation): latency per hop
__global__ __launch_bounds__(256) void k_chain(const int* __restrict__ next, int* out, int depth) {
    int idx = blockIdx.x * 256 + threadIdx.x;
    for (int i = 0; i < depth; ++i) idx = next[idx];
    if (idx == -1) out[0] = idx;
}
// stream: each thread loads `n` 16-byte vectors (independent) and stores one
__global__ __launch_bounds__(256) void k_stream(const uint4* __restrict__ src, uint4* dst, int n, size_t stride) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    uint4 acc = {0, 0, 0, 0};
    for (int j = 0; j < n; ++j) { uint4 v = src[i + j * stride]; acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w; }
    dst[i] = acc;
}

template <typename F> float time_graph(F launch, int reps, hipStream_t s) {
    hipGraph_t g; hipGraphExec_t ge;
    launch(); CK(hipStreamSynchronize(s));
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (int i = 0; i < reps; ++i) launch();
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < 5; ++i) CK(hipGraphLaunch(ge, s));
    CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3f / (5 * reps);
}

int main() {
    hipStream_t s; CK(hipStreamCreate(&s));
    int* out; CK(hipMalloc(&out, 1 << 20));
    CK(hipFuncSetAttribute((const void*)k_empty, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    for (int grid : {64, 256, 320, 512, 1024, 4096})
        for (int lds : {0, 32768, 65536, 98304, 131072, 163840})
            printf("empty grid=%5d lds=%6d: %.2f us\n", grid, lds, time_graph([&] { hipLaunchKernelGGL(k_empty, dim3(grid), dim3(256), lds, s, out, 0); }, 100, s));
    for (int nbar : {10, 40, 160})
        printf("barriers grid=320 lds=64K nbar=%d: %.2f us\n", nbar, time_graph([&] { hipLaunchKernelGGL(k_empty, dim3(320), dim3(256), 65536, s, out, nbar); }, 100, s));
    // pointer chase
    const int N = 1 << 22;
    std::vector<int> h(N);
    for (int i = 0; i < N; ++i) h[i] = (int)(((long long)i * 1048583LL + 12345) % N);
    int* next; CK(hipMalloc(&next, N * 4)); CK(hipMemcpy(next, h.data(), N * 4, hipMemcpyHostToDevice));
    for (int depth : {0, 1, 2, 4, 8})
        printf("chain grid=256 depth=%d: %.2f us\n", depth, time_graph([&] { hipLaunchKernelGGL(k_chain, dim3(256), dim3(256), 0, s, next, out, depth); }, 50, s));
    // streaming reads of a 2.6 MB / 26 MB buffer
    uint4 *src, *dst; CK(hipMalloc(&src, 256 << 20)); CK(hipMalloc(&dst, 64 << 20)); CK(hipMemset(src, 1, 256 << 20));
    for (int blocks : {256, 640, 2560})
        for (int n : {1, 4})
            printf("stream blocks=%d x%d vec (%d KB): %.2f us\n", blocks, n, blocks * 256 * 16 * n / 1024,
                   time_graph([&] { hipLaunchKernelGGL(k_stream, dim3(blocks), dim3(256), 0, s, src, dst, n, (size_t)blocks * 256); }, 50, s));
    return 0;
}
