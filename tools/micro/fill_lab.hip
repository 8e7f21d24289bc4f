// Development probe (round 6): what bounds the LDS-DMA fill of a GEMM tile's stages when EVERY CU runs one tile of the same problem --
// the access pattern of gemm_ks.hip (64 x 80 tiles of a 1024 x 1280 x K problem: 256 workgroups, 18 one-KiB pieces per 128-byte K stage,
// XCD-contiguous tile order in groups of 8 row blocks), pure fetch, no consumption.  Knobs: waves that issue, row pitch of the operands,
// shared vs private rows, the K phase of a tile (all tiles at the same K offset vs rotated per tile), 64 x 128 tiles on 160 workgroups.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/micro/fill_lab.hip -o build/fill_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef __attribute__((address_space(3))) void* lptr_t;

struct P {
    const uint8_t *a, *b;
    int pitch, nk, tiles_m, tiles_n, bm, bn, group_m, priv, rot, depth;
};

template <int NW, int BM, int BN, int DEPTH>
__global__ __launch_bounds__(NW * 64) void k_fill(P p) {
    extern __shared__ __attribute__((aligned(1024))) uint8_t lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwg = p.tiles_m * p.tiles_n;
    int bid = blockIdx.x;
    {
        const int q = nwg / 8, r = nwg % 8, xcd = bid % 8, j = bid / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    }
    const int per_group = p.group_m * p.tiles_n;
    const int gid = bid / per_group, first_m = gid * p.group_m;
    const int gsz = (p.tiles_m - first_m) < p.group_m ? (p.tiles_m - first_m) : p.group_m;
    const int in_g = bid - gid * per_group;
    int tile_n = in_g / gsz, tile_m = first_m + in_g - tile_n * gsz;
    if (p.priv) { tile_m = bid; tile_n = bid; }  // private rows: every workgroup its own row blocks of (larger) matrices
    constexpr int pa = BM / 8, pb = BN / 8, pcs = pa + pb;  // (compile-time: a run-time division per piece costs more than the piece)
    const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void*)(p.a + (size_t)tile_m * BM * p.pitch), 0, 0x7fffffff, 0x00020000);
    const auto rsB = __builtin_amdgcn_make_buffer_rsrc((void*)(p.b + (size_t)tile_n * BN * p.pitch), 0, 0x7fffffff, 0x00020000);
    const int sw0 = ((lane & 7) ^ ((lane >> 4) & 7)) << 4, sw1 = ((lane & 7) ^ ((4 + (lane >> 4)) & 7)) << 4;
    const int vo0 = (lane >> 3) * p.pitch + sw0, vo1 = (lane >> 3) * p.pitch + sw1;
    constexpr int stage_bytes = pcs * 1024;
    const int rot = p.rot ? (int)((blockIdx.x * 7u) % (unsigned)p.nk) : 0;
    const int total = pcs * p.nk;
    constexpr int window = DEPTH * pcs / NW;  // pieces per wave in flight
    int inflight = 0;
    for (int g = wave; g < total; g += NW) {
        const int st = g / pcs, pc = g - st * pcs;
        int ks = st + rot; if (ks >= p.nk) ks -= p.nk;
        uint8_t* slot = lds + (st % DEPTH) * stage_bytes;
        if (pc < pb) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lptr_t)(slot + pa * 1024 + pc * 1024), 16, (pc & 1) ? vo1 : vo0, ks * 128 + pc * 8 * p.pitch, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lptr_t)(slot + (pc - pb) * 1024), 16, (pc & 1) ? vo1 : vo0, ks * 128 + (pc - pb) * 8 * p.pitch, 0, 0);
        if (++inflight >= window) {  // keep ~depth stages in flight: wait for the oldest quarter of the window
            if (window >= 16) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            else if (window >= 8) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <typename F> float time_it(F launch, hipStream_t s) {
    launch(); CK(hipStreamSynchronize(s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < 20; ++i) launch();
    CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3f / 20;
}

int main() {
    hipStream_t s; CK(hipStreamCreate(&s));
    uint8_t *a, *b;
    const size_t SZ = (size_t)768 << 20;
    CK(hipMalloc(&a, SZ)); CK(hipMalloc(&b, SZ)); CK(hipMemset(a, 1, SZ)); CK(hipMemset(b, 2, SZ));
    struct Cfg { const char* name; int nw, bm, bn, tm, tn, K, pitch, priv, rot, depth; };
    const Cfg cfgs[] = {
        {"ks 64x80  4 waves  K1280", 4, 64, 80, 16, 16, 1280, 1280, 0, 0, 8},
        {"ks 64x80  4 waves  K32768", 4, 64, 80, 16, 16, 32768, 32768, 0, 0, 8},
        {"ks 64x80 16 waves  K32768", 16, 64, 80, 16, 16, 32768, 32768, 0, 0, 8},
        {"ks 64x80  8 waves  K1280", 8, 64, 80, 16, 16, 1280, 1280, 0, 0, 8},
        {"ks 64x80 16 waves  K1280", 16, 64, 80, 16, 16, 1280, 1280, 0, 0, 8},
        {"ks 64x80  8 waves  K1280 rotated", 8, 64, 80, 16, 16, 1280, 1280, 0, 1, 8},
        {"ks 64x80 16 waves  K1280 rotated", 16, 64, 80, 16, 16, 1280, 1280, 0, 1, 8},
        {"ks 64x80  8 waves  K1280 pitch 1408", 8, 64, 80, 16, 16, 1280, 1408, 0, 0, 8},
        {"ks 64x80  8 waves  K1280 pitch 8192", 8, 64, 80, 16, 16, 1280, 8192, 0, 0, 8},
        {"ks 64x80  8 waves  K1280 private rows", 8, 64, 80, 16, 16, 1280, 1280, 1, 0, 8},
        {"ks 64x80  8 waves  K1280 private rotated", 8, 64, 80, 16, 16, 1280, 1280, 1, 1, 8},
        {"ks 64x80  8 waves  K5120", 8, 64, 80, 16, 16, 5120, 5120, 0, 0, 8},
        {"ks 64x80  8 waves  K5120 rotated", 8, 64, 80, 16, 16, 5120, 5120, 0, 1, 8},
        {"ks 64x80 16 waves  K5120 rotated", 16, 64, 80, 16, 16, 5120, 5120, 0, 1, 8},
        {"ks 64x80  8 waves  K5120 pitch 5248", 8, 64, 80, 16, 16, 5120, 5248, 0, 0, 8},
        {"ks 64x80  8 waves  K5120 pitch 5248 rotated", 8, 64, 80, 16, 16, 5120, 5248, 0, 1, 8},
        {"64x128    8 waves  K1280 (160 wgs)", 8, 64, 128, 16, 10, 1280, 1280, 0, 0, 6},
        {"64x128    8 waves  K1280 rotated", 8, 64, 128, 16, 10, 1280, 1280, 0, 1, 6},
        {"64x128    8 waves  K5120 (160 wgs)", 8, 64, 128, 16, 10, 5120, 5120, 0, 0, 6},
        {"64x128    8 waves  K5120 rotated", 8, 64, 128, 16, 10, 5120, 5120, 0, 1, 6},
        {"ks 64x80  8 waves  K1280 depth 4", 8, 64, 80, 16, 16, 1280, 1280, 0, 0, 4},
        {"ks 64x80  8 waves  K32768 (steady state)", 8, 64, 80, 16, 16, 32768, 32768, 0, 0, 8},
        {"ks 64x80  8 waves  K32768 rotated", 8, 64, 80, 16, 16, 32768, 32768, 0, 1, 8},
    };
    for (const Cfg& c : cfgs) {
        P p{a, b, c.pitch, c.K / 128, c.tm, c.tn, c.bm, c.bn, 8, c.priv, c.rot, c.depth};
        const int nwg = c.tm * c.tn, ldsb = c.depth * (c.bm + c.bn) * 128;
        if (c.priv && (size_t)nwg * c.bn * c.pitch > SZ) { printf("%s: skipped (size)\n", c.name); continue; }
        float t = 0;
#define RUN(NWV, BMV, BNV, DV)                                                                                                      \
    if (c.nw == NWV && c.bm == BMV && c.bn == BNV && c.depth == DV) {                                                               \
        CK(hipFuncSetAttribute((const void*)k_fill<NWV, BMV, BNV, DV>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));     \
        t = time_it([&] { hipLaunchKernelGGL((k_fill<NWV, BMV, BNV, DV>), dim3(nwg), dim3(NWV * 64), ldsb, s, p); }, s);             \
    }
        RUN(8, 64, 80, 8) RUN(16, 64, 80, 8) RUN(8, 64, 80, 4) RUN(8, 64, 128, 6) RUN(4, 64, 80, 8)
        if (t == 0) { printf("%s: no instantiation\n", c.name); continue; }
        const double bytes = (double)nwg * (c.bm + c.bn) * c.K;
        printf("%-46s %4d wgs  %7.2f us  %6.2f TB/s  %5.1f B/clk/CU-in-use (@2.1 GHz)\n", c.name, nwg, t, bytes / t / 1e6, bytes / t / 1e6 * 1e12 / nwg / 2.1e9);
    }
    return 0;
}
