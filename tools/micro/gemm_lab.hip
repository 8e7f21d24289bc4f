// Development lab for the scaled-matmul kernel template (sdnq_amd/csrc/gemm.hip): a stand-alone executable that instantiates a
// handful of tile configurations only (seconds to compile instead of the library's two minutes), runs them on a list of shapes,
// checks every configuration's output against the first one bit for bit and times back-to-back launches with HIP events.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DLAB_...] tools/micro/gemm_lab.hip -o build/gemm_lab
//   build/gemm_lab "1024,10240,1280;1024,1280,5120" 1,12,13
#define SDNQ_LAB 1
#include "../../sdnq_amd/csrc/gemm.hip"

#include <stdio.h>
#include <string.h>

#include <string>
#include <vector>

#define HC(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

static int run_cfg(int id, GemmParams p, hipStream_t s) {
    constexpr int MM = SDNQ_MM_I8, OT = SDNQ_BF16, EP = EPI_BIAS1D;
    switch (id) {
        case 0: return launch_one<MM, OT, EP, 256, 256, 128, 64, 4, LD_PIPE, 64>(p, s);
        case 1: return launch_one<MM, OT, EP, 64, 128, 32, 32, 3, LD_DMA>(p, s);
        case 2: return launch_one<MM, OT, EP, 64, 64, 32, 32, 4, LD_PIPE>(p, s);
        case 3: return launch_one<MM, OT, EP, 256, 128, 64, 64, 3, LD_PIPE, 64>(p, s);
        case 10: return launch_one<MM, OT, EP, 128, 128, 64, 32, 4, LD_PIPE, 64>(p, s);
        case 13: return launch_one<MM, OT, EP, 256, 160, 32, 160, 3, LD_PIPE, 64>(p, s);
        case 18: return launch_one<MM_I8_16, OT, EP, 64, 80, 16, 80, 3, LD_DMA, 128>(p, s);
        case 19: return launch_one<MM_I8_16, OT, EP, 64, 80, 16, 80, 4, LD_DMA, 128>(p, s);
        case 24: return launch_one<MM, OT, EP, 128, 128, 64, 32, 3, LD_PIPE, 128>(p, s);
        case 30: return launch_one<MM, OT, EP, 256, 256, 128, 64, 4, LD_8P, 64>(p, s);
        case 31: return launch_one<MM, OT, EP, 256, 128, 64, 64, 3, LD_8P, 128>(p, s);
        case 32: return launch_one<MM, OT, EP, 256, 256, 128, 64, 2, LD_HT, 128>(p, s);
#ifdef LAB_EXTRA
        LAB_EXTRA
#endif
        default: return -100;
    }
}

__global__ void fill_uniform(int8_t* p, size_t n, unsigned seed) {  // LAB_UNIFORM=1: full-entropy bytes (the power-limited case)
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        unsigned x = (unsigned)i * 2654435761u + seed;
        x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
        p[i] = (int8_t)(x & 0xff);
    }
}
__global__ void fill_kernel(int8_t* p, size_t n, unsigned seed) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        unsigned x = (unsigned)i * 2654435761u + seed;
        x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
        // sum of four bytes, centred: a bell-shaped int8 like row-wise absmax-quantized tensors
        const int v = (int)(x & 63) + (int)((x >> 8) & 63) + (int)((x >> 16) & 63) + (int)((x >> 24) & 63) - 126;
        p[i] = (int8_t)(v < -127 ? -127 : (v > 127 ? 127 : v));
    }
}
__global__ void fill_f(float* p, size_t n, float base, float step) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < n) p[i] = base + step * (float)(i % 97);
}

int main(int argc, char** argv) {
    std::vector<std::array<int64_t, 3>> shapes;
    std::vector<int> ids;
    {
        std::string sh = argc > 1 ? argv[1] : "1024,10240,1280";
        size_t pos = 0;
        while (pos < sh.size()) {
            size_t e = sh.find(';', pos);
            if (e == std::string::npos) e = sh.size();
            long long m, n, k;
            sscanf(sh.substr(pos, e - pos).c_str(), "%lld,%lld,%lld", &m, &n, &k);
            shapes.push_back({m, n, k});
            pos = e + 1;
        }
        std::string il = argc > 2 ? argv[2] : "1,13";
        pos = 0;
        while (pos < il.size()) {
            size_t e = il.find(',', pos);
            if (e == std::string::npos) e = il.size();
            ids.push_back(atoi(il.substr(pos, e - pos).c_str()));
            pos = e + 1;
        }
    }
    const int reps = argc > 3 ? atoi(argv[3]) : 50;
    hipStream_t s;
    HC(hipStreamCreate(&s));
    for (auto& sh : shapes) {
        const int64_t m = sh[0], n = sh[1], k = sh[2];
        int8_t *a, *b;
        float *sa, *sb;
        uint16_t *bias, *out, *ref;
        const int64_t pad = getenv("LAB_PAD") ? atoi(getenv("LAB_PAD")) : 0;  // row pitch = K + pad bytes (L2 channel spread experiment)
        const int64_t kp = k + pad;
        HC(hipMalloc(&a, m * kp)); HC(hipMalloc(&b, n * kp)); HC(hipMalloc(&sa, m * 4)); HC(hipMalloc(&sb, n * 4));
        HC(hipMalloc(&bias, n * 2)); HC(hipMalloc(&out, m * n * 2)); HC(hipMalloc(&ref, m * n * 2));
        if (getenv("LAB_ZERO")) {
            HC(hipMemsetAsync(a, 0, m * kp, s));
            HC(hipMemsetAsync(b, 0, n * kp, s));
        } else if (getenv("LAB_UNIFORM")) {
            fill_uniform<<<2048, 256, 0, s>>>(a, m * kp, 1u);
            fill_uniform<<<2048, 256, 0, s>>>(b, n * kp, 77u);
        } else {
            fill_kernel<<<2048, 256, 0, s>>>(a, m * kp, 1u);
            fill_kernel<<<2048, 256, 0, s>>>(b, n * kp, 77u);
        }
        fill_f<<<(m + 255) / 256, 256, 0, s>>>(sa, m, 1e-3f, 1e-5f);
        fill_f<<<(n + 255) / 256, 256, 0, s>>>(sb, n, 2e-3f, 1e-5f);
        HC(hipMemsetAsync(bias, 0x3c, n * 2, s));
        HC(hipStreamSynchronize(s));
        printf("M=%lld N=%lld K=%lld:", (long long)m, (long long)n, (long long)k);
        bool have_ref = false;
        for (int id : ids) {
            GemmParams p{};
            p.a = (const uint8_t*)a; p.b = (const uint8_t*)b; p.sa = sa; p.sb = sb; p.bias = bias; p.out = out;
            p.M = m; p.N = n; p.K = k; p.lda = kp; p.ldb = kp; p.bias_ndim = 1; p.bias_dtype = SDNQ_BF16;
            HC(hipMemsetAsync(out, 0xff, m * n * 2, s));
            int st = run_cfg(id, p, s);
            if (st != 0) { printf(" %d:ERR%d", id, st); continue; }
            HC(hipStreamSynchronize(s));
            bool same = true;
            if (!have_ref) { HC(hipMemcpy(ref, out, m * n * 2, hipMemcpyDeviceToDevice)); have_ref = true; }
            else {
                std::vector<uint16_t> h1(m * n), h2(m * n);
                HC(hipMemcpy(h1.data(), out, m * n * 2, hipMemcpyDeviceToHost));
                HC(hipMemcpy(h2.data(), ref, m * n * 2, hipMemcpyDeviceToHost));
                same = memcmp(h1.data(), h2.data(), m * n * 2) == 0;
            }
            hipEvent_t e0, e1;
            HC(hipEventCreate(&e0)); HC(hipEventCreate(&e1));
            for (int i = 0; i < 3; ++i) run_cfg(id, p, s);
            HC(hipEventRecord(e0, s));
            for (int i = 0; i < reps; ++i) run_cfg(id, p, s);
            HC(hipEventRecord(e1, s));
            HC(hipEventSynchronize(e1));
            float ms;
            HC(hipEventElapsedTime(&ms, e0, e1));
            printf("  %d:%7.2f us%s", id, ms * 1e3 / reps, same ? "" : " MISMATCH");
#ifdef SDNQ_TRACE2
            if (id == 32) {
                unsigned h[64];
                HC(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_trace2), sizeof(h)));
                const double nk = (double)((k + 127) / 128);
                printf("\n   cycles per phase (issue | vmcnt | barrier1+lgkm | mfma | barrier2), per K tile average:\n");
                for (int g = 0; g < 2; ++g)
                    for (int q = 0; q < 4; ++q) {
                        printf("   group %d phase %d:", g, q);
                        for (int e = 0; e < 5; ++e) printf(" %7.1f", h[g * 20 + q * 5 + e] / nk);
                        printf("\n");
                    }
            }
#endif
        }
        printf("\n");
        hipFree(a); hipFree(b); hipFree(sa); hipFree(sb); hipFree(bias); hipFree(out); hipFree(ref);
    }
    return 0;
}
