// Development probe: what does device-wide synchronisation INSIDE a kernel cost on MI355X (8 XCDs, private L2s)?
//   (a) an empty launch in a hipGraph chain (the cost a fused kernel would remove),
//   (b) a grid-wide barrier on a device-scope atomic counter,
//   (c) producer -> flag -> consumer hand-over of data between workgroups (different XCDs), checked for correctness:
//       each workgroup writes a block of bytes, releases a per-block flag, then reads ANOTHER workgroup's block after acquiring its flag.
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -o /tmp/sync_lab tools/micro/sync_lab.hip && /tmp/sync_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__global__ __launch_bounds__(256) void k_empty(int* out) { if (out == nullptr) __builtin_trap(); }

__global__ __launch_bounds__(256) void k_barrier(unsigned* ctr, int rounds, unsigned base) {
    for (int r = 0; r < rounds; ++r) {
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned want = base + (unsigned)(r + 1) * gridDim.x;
            while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < want) __builtin_amdgcn_s_sleep(1);
        }
        __syncthreads();
    }
}

// producer / consumer: block b writes 16 KB (value = f(b, epoch)), sets flag[b] = epoch; then waits for flag[(b + shift) % grid] and
// sums that block's data.  mode 0: plain stores + __threadfence (release) / acquire load;  mode 1: nontemporal stores.
template <int MODE>
__global__ __launch_bounds__(256) void k_handover(uint4* data, unsigned* flags, unsigned* sums, unsigned epoch, int shift, int vecs) {
    const int b = blockIdx.x;
    uint4* mine = data + (size_t)b * vecs;
    for (int i = threadIdx.x; i < vecs; i += 256) {
        const uint4 v = {(unsigned)b * 7919u + epoch, (unsigned)i, epoch, 1u};
        if (MODE == 1) __builtin_nontemporal_store(v.x, &mine[i].x), __builtin_nontemporal_store(v.y, &mine[i].y),
                       __builtin_nontemporal_store(v.z, &mine[i].z), __builtin_nontemporal_store(v.w, &mine[i].w);
        else mine[i] = v;
    }
    __syncthreads();  // all stores of the block issued ... and, with the release below, visible device-wide
    if (threadIdx.x == 0) __hip_atomic_store(&flags[b], epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    const int o = (b + shift) % gridDim.x;
    if (threadIdx.x == 0)
        while (__hip_atomic_load(&flags[o], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != epoch) __builtin_amdgcn_s_sleep(1);
    __syncthreads();
    __atomic_thread_fence(__ATOMIC_ACQUIRE);  // every wave of the block: drop stale lines before reading the peer's data
    const uint4* theirs = data + (size_t)o * vecs;
    unsigned bad = 0;
    for (int i = threadIdx.x; i < vecs; i += 256) {
        const uint4 v = theirs[i];
        bad += (v.x != (unsigned)o * 7919u + epoch) + (v.y != (unsigned)i) + (v.z != epoch);
    }
    if (bad) atomicAdd(&sums[0], bad);
}

template <typename F> float time_graph(F launch, int reps, hipStream_t s) {
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (int i = 0; i < reps; ++i) launch(i);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, s));
    CK(hipGraphLaunch(ge, s));
    CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3f / reps;
}

int main() {
    hipStream_t s; CK(hipStreamCreate(&s));
    int* out; CK(hipMalloc(&out, 1 << 20));
    unsigned* ctr; CK(hipMalloc(&ctr, 4096)); CK(hipMemset(ctr, 0, 4096));
    for (int grid : {160, 256, 512})
        printf("empty launch in a graph chain, grid=%d: %.2f us\n", grid, time_graph([&](int) { hipLaunchKernelGGL(k_empty, dim3(grid), dim3(256), 0, s, out); }, 200, s));
    // grid barrier: the counter keeps growing across launches; every launch gets its base from the host-side running total
    for (int grid : {160, 256, 512}) {
        for (int rounds : {1, 11}) {
            CK(hipMemset(ctr, 0, 4));
            // replays: graph executed twice (warm + timed) with `reps` launches each -> bases must continue; use separate counters per launch index instead
            unsigned* ctrs; CK(hipMalloc(&ctrs, 400 * 64 * 4)); CK(hipMemset(ctrs, 0, 400 * 64 * 4));
            // each launch uses its own counter; the graph runs twice, so the second run starts from rounds*grid: pass base via a device-side trick:
            // simpler: run the timed graph ONCE only after re-zeroing (time_graph launches warm + timed = 2 runs) -> use want relative to parity
            hipGraph_t g; hipGraphExec_t ge;
            CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
            for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k_barrier, dim3(grid), dim3(256), 0, s, ctrs + i * 64, rounds, 0u);
            CK(hipStreamEndCapture(s, &g));
            CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            CK(hipEventRecord(e0, s));
            CK(hipGraphLaunch(ge, s));
            CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("kernel with %2d grid barrier(s), grid=%d: %.2f us per launch (cold graph run)\n", rounds, grid, ms * 1e3f / 200);
            CK(hipFree(ctrs));
        }
    }
    // hand-over
    const int grid = 256, vecs = 1024;  // 16 KB per block
    uint4* data; CK(hipMalloc(&data, (size_t)grid * vecs * 16)); CK(hipMemset(data, 0, (size_t)grid * vecs * 16));
    unsigned *flags, *sums; CK(hipMalloc(&flags, grid * 4)); CK(hipMalloc(&sums, 64)); CK(hipMemset(flags, 0, grid * 4)); CK(hipMemset(sums, 0, 64));
    for (int mode = 0; mode < 2; ++mode) {
        for (int shift : {1, 37, 131}) {
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            static unsigned epoch = 1;
            CK(hipEventRecord(e0, s));
            for (int i = 0; i < 200; ++i, ++epoch) {
                if (mode) hipLaunchKernelGGL(k_handover<1>, dim3(grid), dim3(256), 0, s, data, flags, sums, epoch, shift, vecs);
                else hipLaunchKernelGGL(k_handover<0>, dim3(grid), dim3(256), 0, s, data, flags, sums, epoch, shift, vecs);
            }
            CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            unsigned bad; CK(hipMemcpy(&bad, sums, 4, hipMemcpyDeviceToHost));
            printf("hand-over mode=%d shift=%3d: %.2f us per launch, mismatches so far %u\n", mode, shift, ms * 1e3f / 200, bad);
        }
    }
    return 0;
}
