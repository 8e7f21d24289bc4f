// Development probe (round 3): what does an instruction of one wave cost while the OTHER wave of its SIMD streams MFMAs?
// 8 waves per CU (one workgroup per CU): waves 0-3 ("loaders") run one kind of instruction stream; waves 4-7 (their SIMD partners) are idle
// or run a register-resident v_mfma_i32_32x32x32_i8 loop.  Loader streams:
//   G  LDS-DMA pieces (8 rows x 128 B) with global_load_lds and per-piece 64-bit address arithmetic (what gemm.hip does)
//   B  the same pieces with buffer_load ... lds: constant per-lane voffset, the K advance in the SGPR soffset -- NO vector ALU work
//   R  ds_read_b128 (immediate offsets, no address arithmetic)
//   V  dependent v_add chains
//   RV ds_read_b128 whose address register is written by a vector-ALU add right before each read (what the generic K loops do)
//   V64 dependent 64-bit integer adds (v_lshl_add_u64), GS global_load_lds with an SGPR base + constant 32-bit VGPR offset (no VALU)
//   LV / LS plain global_load_dwordx4 into registers, 64-bit VGPR address (fixed) vs SGPR base + 32-bit VGPR offset;  SV / SS global_store_dwordx4 likewise
//   BV buffer_load ... lds whose voffset register is written by a v_mov / v_add right before each load
// Reports the loader wave's cycles per instruction (s_memtime stamps of wave 0, averaged over the CUs).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/micro/dma_mfma_lab.hip -o build/dma_mfma_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

__device__ unsigned long long g_stamp[2 * 256];

enum { S_G = 0, S_B = 1, S_R = 2, S_V = 3, S_RV = 4, S_BV = 5, S_V64 = 6, S_GS = 7, S_LV = 8, S_LS = 9, S_SV = 10, S_SS = 11 };
// PART 0: partners idle; 1: partners MFMA back to back; 2: partners MFMA with `GAP` s_nop-free VALU-free pauses (s_sleep-less): 1 MFMA + 1 idle slot
template <int STREAM, int PART, int DEPTH>
__global__ __launch_bounds__(512) void k_probe(const uint8_t* __restrict__ src, int* out, int iters, int mfma_iters, int ld, size_t bytes) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wave < 4) {
        const int row0 = ((blockIdx.x & 7) * 1237 + ((blockIdx.x >> 3) % 6) * 256) & 2047;
        unsigned long long t0 = 0, t1 = 0;
        long n_inst = 0;
        if constexpr (STREAM == S_G) {
            int kofs = 0;
            t0 = __builtin_amdgcn_s_memtime();
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int u = 0; u < DEPTH; ++u) {
                    const int r = (row0 + (wave * DEPTH + u) * 8 + (lane >> 3)) & 2047;
                    const uint8_t* s = src + (size_t)r * ld + ((kofs + (lane & 7) * 16) & (ld - 1));
                    __builtin_amdgcn_global_load_lds((gptr_t)s, (lptr_t)(lds + (((it & 1) * 4 * DEPTH + wave * DEPTH + u) * 1024)), 16, 0, 0);
                }
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH) : "memory");
                kofs += 128;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            t1 = __builtin_amdgcn_s_memtime();
            n_inst = (long)iters * DEPTH;
        } else if constexpr (STREAM == S_B) {
#if defined(__HIP_DEVICE_COMPILE__)  // the buffer-resource type only exists in the device pass
            auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, (int)bytes, 0x00020000);
            int voff[DEPTH];
#pragma unroll
            for (int u = 0; u < DEPTH; ++u) voff[u] = ((row0 + (wave * DEPTH + u) * 8 + (lane >> 3)) & 2047) * ld + (lane & 7) * 16;
            int kofs = 0;
            t0 = __builtin_amdgcn_s_memtime();
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int u = 0; u < DEPTH; ++u)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)(lds + (((it & 1) * 4 * DEPTH + wave * DEPTH + u) * 1024)), 16, voff[u], kofs, 0, 0);
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH) : "memory");
                kofs = (kofs + 128) & (ld - 1);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            t1 = __builtin_amdgcn_s_memtime();
            n_inst = (long)iters * DEPTH;
#endif
        } else if constexpr (STREAM == S_R) {
            v4i acc = {0, 0, 0, 0};
            const uint8_t* base = lds + lane * 16 + wave * 4096;
            t0 = __builtin_amdgcn_s_memtime();
            for (int it = 0; it < iters; ++it) {
                v4i t[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) t[u] = *(const v4i*)(base + u * 1024 + (it & 1) * 32768);
#pragma unroll
                for (int u = 0; u < 8; ++u) acc ^= t[u];
            }
            t1 = __builtin_amdgcn_s_memtime();
            n_inst = (long)iters * 8;
            if (acc[0] == 0x1234567) out[1] = acc[1];
        } else if constexpr (STREAM == S_RV) {
            v4i acc = {0, 0, 0, 0};
            int base = lane * 16 + wave * 4096;
            t0 = __builtin_amdgcn_s_memtime();
            for (int it = 0; it < iters; ++it) {
                v4i t[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    int a = base + u * 1024 + (it & 1) * 32768;
                    asm volatile("v_add_u32 %0, %1, 0" : "=v"(a) : "v"(a));  // a vector-ALU write of the address right before the read
                    t[u] = *(const v4i*)(lds + a);
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) acc ^= t[u];
            }
            t1 = __builtin_amdgcn_s_memtime();
            n_inst = (long)iters * 8;
            if (acc[0] == 0x1234567) out[1] = acc[1];
        } else if constexpr (STREAM == S_BV) {
#if defined(__HIP_DEVICE_COMPILE__)
            auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, (int)bytes, 0x00020000);
            int voff[DEPTH];
#pragma unroll
            for (int u = 0; u < DEPTH; ++u) voff[u] = ((row0 + (wave * DEPTH + u) * 8 + (lane >> 3)) & 2047) * ld + (lane & 7) * 16;
            int kofs = 0;
            t0 = __builtin_amdgcn_s_memtime();
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int u = 0; u < DEPTH; ++u) {
                    int vo = voff[u];
                    asm volatile("v_add_u32 %0, %1, 0" : "=v"(vo) : "v"(vo));
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)(lds + (((it & 1) * 4 * DEPTH + wave * DEPTH + u) * 1024)), 16, vo, kofs, 0, 0);
                }
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH) : "memory");
                kofs = (kofs + 128) & (ld - 1);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            t1 = __builtin_amdgcn_s_memtime();
            n_inst = (long)iters * DEPTH;
#endif
        } else if constexpr (STREAM == S_V64) {
            unsigned long long x = lane, y = (unsigned long long)lane << 33;
            t0 = __builtin_amdgcn_s_memtime();
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int u = 0; u < 8; ++u) { x = (x << 1) + y; y = (y << 2) + x; }
            }
            t1 = __builtin_amdgcn_s_memtime();
            n_inst = (long)iters * 8 * 2;
            if (x == 0x1234567) out[1] = (int)y;
        } else if constexpr (STREAM == S_GS) {
            int voff[DEPTH];
#pragma unroll
            for (int u = 0; u < DEPTH; ++u) voff[u] = ((row0 + (wave * DEPTH + u) * 8 + (lane >> 3)) & 2047) * ld + (lane & 7) * 16;
            int kofs = 0;
            t0 = __builtin_amdgcn_s_memtime();
            for (int it = 0; it < iters; ++it) {
                const uint8_t* sb = src + kofs;  // wave-uniform base
#pragma unroll
                for (int u = 0; u < DEPTH; ++u) {
                    const unsigned m0v = (unsigned)(size_t)(lptr_t)(lds + (((it & 1) * 4 * DEPTH + wave * DEPTH + u) * 1024));
                    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(m0v), "v"(voff[u]), "s"(sb) : "memory");
                }
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH) : "memory");
                kofs = (kofs + 128) & (ld - 1);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            t1 = __builtin_amdgcn_s_memtime();
            n_inst = (long)iters * DEPTH;
        } else if constexpr (STREAM == S_LV || STREAM == S_LS || STREAM == S_SV || STREAM == S_SS) {
            // addresses fixed before the loop (no VALU inside); only the addressing FORM differs
            const int vo = (((row0 + wave * 8 + (lane >> 3)) & 2047) * ld + (lane & 7) * 16);
            const uint8_t* pv = src + vo;
            v4i acc = {0, 0, 0, 0};
            t0 = __builtin_amdgcn_s_memtime();
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if constexpr (STREAM == S_LV) {
                        v4i t;
                        asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(t) : "v"(pv), "n"(0) : "memory");
                        asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
                    } else if constexpr (STREAM == S_LS) {
                        v4i t;
                        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(t) : "v"(vo), "s"(src) : "memory");
                        asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
                    } else if constexpr (STREAM == S_SV) {
                        asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(pv + ((size_t)32 << 20)), "v"(acc) : "memory");
                        asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
                    } else {
                        asm volatile("global_store_dwordx4 %0, %1, %2" ::"v"(vo + (32 << 20)), "v"(acc), "s"(src) : "memory");
                        asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
                    }
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            t1 = __builtin_amdgcn_s_memtime();
            n_inst = (long)iters * 4;
            if (acc[0] == 0x1234567) out[1] = acc[1];
        } else {
            int x = lane, y = lane * 3;
            t0 = __builtin_amdgcn_s_memtime();
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int u = 0; u < 8; ++u) { x = x * 3 + y; y = y ^ (x >> 3); }
            }
            t1 = __builtin_amdgcn_s_memtime();
            n_inst = (long)iters * 8 * 4;  // mul-add, shift, xor ~ 3-4 VALU per step
            if (x == 0x1234567) out[1] = y;
        }
        if (threadIdx.x == 0) { g_stamp[blockIdx.x * 2] = t1 - t0; g_stamp[blockIdx.x * 2 + 1] = (unsigned long long)n_inst; }
    } else if (PART != 0) {
        v16i acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][e] = 0;
        v4i a = {lane, lane * 3, lane * 5, lane * 7}, b = {lane * 11, lane * 13, lane * 17, lane * 19};
        for (int it = 0; it < mfma_iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc[i], 0, 0, 0);
                if constexpr (PART == 2) asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // ~32 idle issue cycles after every MFMA
            }
        }
        int sum = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) sum += acc[i][0] + acc[i][7];
        if (sum == 0x12345678) out[0] = sum;
    }
}
template <int STREAM, int PART, int DEPTH>
void run(const uint8_t* src, int* out, hipStream_t s, const char* what) {
    const int iters = 256, ld = 8192;
    const int mfma_iters = 256 * 4 * 4 * 60 / 32 / 4 * 4;
    auto kern = k_probe<STREAM, PART, DEPTH>;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(kern, dim3(256), dim3(512), 128 * 1024, s, src, out, iters, mfma_iters, ld, (size_t)64 << 20);
    CK(hipStreamSynchronize(s));
    unsigned long long h[512];
    CK(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_stamp), sizeof(h)));
    double cyc = 0, n = 0;
    for (int b = 0; b < 256; ++b) { cyc += (double)h[2 * b]; n += (double)h[2 * b + 1]; }
    printf("  %-72s %8.1f cycles per instruction of the loader wave\n", what, cyc / n);
}
int main() {
    hipStream_t s; CK(hipStreamCreate(&s));
    uint8_t* src; int* out;
    CK(hipMalloc(&src, (size_t)64 << 20)); CK(hipMalloc(&out, 1 << 20)); CK(hipMemset(src, 1, (size_t)64 << 20));
    run<S_G, 0, 2>(src, out, s, "G global_load_lds + address VALU, depth 2 | partners idle");
    run<S_G, 1, 2>(src, out, s, "G global_load_lds + address VALU, depth 2 | partners MFMA dense");
    run<S_G, 2, 2>(src, out, s, "G global_load_lds + address VALU, depth 2 | partners MFMA + 32 idle");
    run<S_B, 0, 2>(src, out, s, "B buffer_load lds, no VALU, depth 2        | partners idle");
    run<S_B, 1, 2>(src, out, s, "B buffer_load lds, no VALU, depth 2        | partners MFMA dense");
    run<S_B, 2, 2>(src, out, s, "B buffer_load lds, no VALU, depth 2        | partners MFMA + 32 idle");
    run<S_B, 1, 4>(src, out, s, "B buffer_load lds, no VALU, depth 4        | partners MFMA dense");
    run<S_R, 0, 2>(src, out, s, "R ds_read_b128, immediate offsets          | partners idle");
    run<S_R, 1, 2>(src, out, s, "R ds_read_b128, immediate offsets          | partners MFMA dense");
    run<S_RV, 0, 2>(src, out, s, "RV ds_read_b128, address from a VALU add   | partners idle");
    run<S_RV, 1, 2>(src, out, s, "RV ds_read_b128, address from a VALU add   | partners MFMA dense");
    run<S_BV, 0, 2>(src, out, s, "BV buffer_load lds, voffset from a VALU add | partners idle");
    run<S_BV, 1, 2>(src, out, s, "BV buffer_load lds, voffset from a VALU add | partners MFMA dense");
    run<S_V64, 0, 2>(src, out, s, "V64 dependent 64-bit shift-adds (per op)   | partners idle");
    run<S_V64, 1, 2>(src, out, s, "V64 dependent 64-bit shift-adds (per op)   | partners MFMA dense");
    run<S_GS, 0, 2>(src, out, s, "GS global_load_lds, SGPR base + const voff | partners idle");
    run<S_GS, 1, 2>(src, out, s, "GS global_load_lds, SGPR base + const voff | partners MFMA dense");
    run<S_LV, 0, 2>(src, out, s, "LV global_load_dwordx4 v[a:b], off          | partners idle");
    run<S_LV, 1, 2>(src, out, s, "LV global_load_dwordx4 v[a:b], off          | partners MFMA dense");
    run<S_LS, 0, 2>(src, out, s, "LS global_load_dwordx4 voff, s[base]        | partners idle");
    run<S_LS, 1, 2>(src, out, s, "LS global_load_dwordx4 voff, s[base]        | partners MFMA dense");
    run<S_SV, 0, 2>(src, out, s, "SV global_store_dwordx4 v[a:b], off         | partners idle");
    run<S_SV, 1, 2>(src, out, s, "SV global_store_dwordx4 v[a:b], off         | partners MFMA dense");
    run<S_SS, 0, 2>(src, out, s, "SS global_store_dwordx4 voff, s[base]       | partners idle");
    run<S_SS, 1, 2>(src, out, s, "SS global_store_dwordx4 voff, s[base]       | partners MFMA dense");
    run<S_V, 0, 2>(src, out, s, "V dependent VALU chain (per VALU op)       | partners idle");
    run<S_V, 1, 2>(src, out, s, "V dependent VALU chain (per VALU op)       | partners MFMA dense");
    run<S_V, 2, 2>(src, out, s, "V dependent VALU chain (per VALU op)       | partners MFMA + 32 idle");
    return 0;
}
