// The quantized matmul of 4-BIT group-wise weights with the re-quantization FUSED INTO THE GEMM (north_star N1: "the packed-weight unpack
// ... fused into the GEMM"; round 6):  out = cast(fma(f32(xq . Wq^T) * xs[m], ws[n], bias))  with Wq never materialized.
//
// Reference chain replaced (dequantizer.py:166-239 re_quantize_matmul + layers/linear/linear_int8.py:104-107): every forward unpacks the
// 4-bit codes, dequantizes them with their group scales (+ zero points), re-quantizes the row to int8 -- Wq[n][k] = round(value / ws[n]),
// ws[n] = amax_k |value| / 127 -- and multiplies.  Earlier rounds either kept Wq resident (cached mode: +1 B per weight, 3.6 GB
// instead of 1.4 GB for the SDXL UNet) or rebuilt it with a separate kernel per call (per-call mode: 443 extra launches per step,
// 13.3 vs 8.9 ms).  Here the stored codes ARE the GEMM's weight operand:
//   * within one 64-element group of one row the int8 value is a function of the 4-bit code alone: a 16-entry byte table per
//     (row, 64 columns), built ONCE at load with the very expressions of the re-quantizer (sdnq_hip_lut4_build: dequant.hip's
//     requant_lut4_kernel writing its tables instead of expanding them) -- 0.25 B per weight, so the layer keeps 0.5 + 0.25 B per weight
//     + scales resident instead of 0.5 + 1.0;
//   * a K stage of the weight operand in LDS is 128 rows x (128 B of codes + 4 tables of 16 B) = 24 KB for 256 columns (int8: 32 KB);
//   * inside the workgroup every 16-column fragment of the stage is expanded ONCE -- 8 code bytes -> 16 int8 values by byte permutes
//     through the fragment's table (nibble split, three v_perm_b32 per four codes, even / odd re-interleave: ~36 vector instructions) --
//     into a 32-KB int8 image of the stage in LDS, exactly the bytes the stand-alone kernel would have written to HBM (bit-identical by
//     construction and by tests/test_gemm_w4.py); the MFMA phase then reads both operands as the int8 GEMM does.
// The expansion is repeated by every row block that reads the weight tile (M / 64 times), so this form is for the few-row problems of
// the bs = 1 diffusion steps (sdnq_hip_scaled_mm_w4_supported: M <= 1024, K <= 1280 by default -- where it measured faster than the
// stand-alone re-quantization kernel + the int8 GEMM); FLUX-size M keeps the stand-alone kernel (profiles/r03_lut4_fused_loader_lab.txt).
//
// Tile 64 x 128, eight waves of 32 x 32 on v_mfma_i32_32x32x32_i8, K stages of 256 columns (full 128-byte lines of every operand):
// 16 + 16 + 8 one-KiB LDS-DMA pieces per stage = 5 per wave, 3-stage ring + the int8 image (152 KB: one workgroup per CU).
#include <atomic>
#include <cstdlib>
#include <type_traits>

#include "sdnq_dev.h"

int sdnq_internal_take_prefetch(int64_t room, int threads, const uint8_t* pf_ptr[4], int pf_lines[4]);  // gemm.hip

namespace {

constexpr int BM = 64, BN = 128, SK = 256, NW = 8, NT = NW * 64, NS = 3;
constexpr int A_STAGE = BM * SK, C_STAGE = BN * SK / 2, L_STAGE = BN * (SK / 64) * 16;  // 16 + 16 + 8 KiB
constexpr int STAGE = A_STAGE + C_STAGE + L_STAGE;
constexpr int PPW = 5;  // pieces per wave per stage
constexpr int W8_BYTES = BN * SK;  // the expanded int8 weight operand of ONE stage (32 KiB)
constexpr int LDS_BYTES = NS * STAGE + W8_BYTES + (2 * BN + BM) * 4;
static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");

struct W4Params {
    const float* xs;      // [M] activation row scales
    const float* ws;      // [N] weight row scales (of the re-quantized rows)
    const void* bias;     // [N] or null
    void* out;            // [M][ldc]
    int64_t ldc;
    int bias_dtype;
    const uint8_t* pf_ptr[4];  // weight prefetch hosted by this launch (sdnq_hip_prefetch_hint)
    int pf_lines[4];
};

template <int N> __device__ __forceinline__ void w4_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// 8 packed bytes (16 four-bit codes, element j = nibble j) -> the 16 int8 values, through the 16-entry table t (entry c = byte c & 3 of
// dword c >> 2).  The same look-up as requant_lut4_kernel (dequant.hip), on whole dwords: nibble split (even / odd elements), two
// byte permutes + a bit-3 blend per four codes, then the even / odd halves re-interleaved into element order.
__device__ __forceinline__ v4i expand16(const v2i& codes, const v4i& t) {
    v4i o;
#pragma unroll
    for (int d = 0; d < 2; ++d) {
        const u32 x = (u32)codes[d];
        u32 r[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const u32 c = h ? ((x >> 4) & 0x0f0f0f0fu) : (x & 0x0f0f0f0fu);  // elements 8d + {0,2,4,6} + h
            const u32 sel = c & 0x07070707u;
            const u32 lo = __builtin_amdgcn_perm((u32)t[1], (u32)t[0], sel);  // entries 0..7
            const u32 hi = __builtin_amdgcn_perm((u32)t[3], (u32)t[2], sel);  // entries 8..15
            // bit 3 of a code picks the byte of `hi` over the byte of `lo`: a third byte permute whose selector is i + 4 (bit 3 of code i)
            // (a 0xff-per-byte mask from `(c >> 3) * 0xff` costs a quarter-rate 32-bit multiply per four codes)
            r[h] = __builtin_amdgcn_perm(hi, lo, ((c >> 1) & 0x04040404u) | 0x03020100u);
        }
        o[2 * d] = (int)__builtin_amdgcn_perm(r[1], r[0], 0x05010400u);      // elements 8d + 0, 1, 2, 3
        o[2 * d + 1] = (int)__builtin_amdgcn_perm(r[1], r[0], 0x07030602u);  // elements 8d + 4, 5, 6, 7
    }
    return o;
}

// OUT_T: bf16 / f16 output
template <int OUT_T, bool HAS_BIAS>
__global__ __launch_bounds__(NT) void gemm_w4_kernel(const uint8_t* __restrict__ a, const uint8_t* __restrict__ codes, const uint8_t* __restrict__ lut, int lda, int M,
                                                     int N, int K, int tiles_m, int tiles_n, int group_m, W4Params p_) {
    SDNQ_KERNARGS_NOW("s"(a), "s"(codes), "s"(lut), "s"(lda), "s"(M), "s"(N), "s"(K), "s"(tiles_m), "s"(tiles_n), "s"(group_m));
    const W4Params& p = p_;
    extern __shared__ __attribute__((aligned(1024))) uint8_t lds[];
    float* const s_sb = (float*)(lds + NS * STAGE + W8_BYTES);
    float* const s_bias = s_sb + BN;
    float* const s_xs = s_bias + BN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nwg = tiles_m * tiles_n;
    int bid = blockIdx.x;
    if (bid >= nwg) {  // hosted weight prefetch: one dword of every 128-byte line of the next layers' weights (gemm.hip, launch_one)
        const int t = (bid - nwg) * NT + tid, stride = ((int)gridDim.x - nwg) * NT;
#pragma nounroll
        for (int r = 0; r < 4; ++r) {
            const uint8_t* base = p.pf_ptr[r];
            const int lines = p.pf_lines[r];
            for (int i = t; i < lines; i += stride) {
                int v;
                asm volatile("global_load_dword %0, %1, off" : "=v"(v) : "v"(base + (int64_t)i * 128) : "memory");
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }
    {   // block b runs on XCD b % 8 (private L2 each): every XCD walks a contiguous range of the tile sequence, groups of row blocks, m fastest
        const int q = nwg / 8, r = nwg % 8, xcd = bid % 8, j = bid / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    }
    int tile_m, tile_n;
    {
        const int per_group = group_m * tiles_n;
        const int gid = bid / per_group, first_m = gid * group_m;
        const int gsz = (tiles_m - first_m) < group_m ? (tiles_m - first_m) : group_m;
        const int in_g = bid - gid * per_group;
        tile_n = in_g / gsz;
        tile_m = first_m + in_g - tile_n * gsz;
    }
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int nk = (K + SK - 1) / SK;              // stages; the last one holds 128 columns when K % 256 == 128 (launcher: K % 128 == 0)
    const int last_ks = (K % SK) ? 4 : 8;          // K sub-steps (32 columns) of the last stage
    const int m_rows = (M - m0) < BM ? (M - m0) : BM, n_lim = (N - n0) < BN ? (N - n0) : BN;
    const int ldc_ = K >> 1, ldl = (K >> 6) << 4;  // row pitches of the code and table operands

    // ---- LDS-DMA assignment (piece = 1 KiB, lane-linear in LDS; the source chunk is the swizzle-inverse).  Per stage and wave w:
    //   activations: pieces w and w + 8 of [2 halves][64 rows][128 B] (8 rows x 128 B each; chunk ^= (row >> 1) & 7)
    //   codes:       pieces w and w + 8 of [128 rows][128 B]
    //   tables:      piece w of [128 rows][4 x 16 B] (16 rows x 64 B; table slot ^= (row >> 2) & 3)
    // Rows past the tile's valid rows are clamped (computed on valid memory, never stored); columns past K in the last stage belong to
    // the next row (never multiplied) or lie past the descriptor's extent (zeros).
    const auto rsA = SDNQ_MAKE_RSRC_N(a + (int64_t)m0 * lda, (int64_t)(m_rows - 1) * lda + K);
    const auto rsC = SDNQ_MAKE_RSRC_N(codes + (int64_t)n0 * ldc_, (int64_t)n_lim * ldc_);
    const auto rsL = SDNQ_MAKE_RSRC_N(lut + (int64_t)n0 * ldl, (int64_t)n_lim * ldl);
    int voA[2], voC[2], voL;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        {   // activation piece w + 8u: half h = (w + 8u) >> 3 = u, rows 8w .. 8w + 7
            const int r = wave * 8 + (lane >> 3);
            const int rc = r < m_rows ? r : m_rows - 1;
            voA[u] = rc * lda + u * 128 + (((lane & 7) ^ ((r >> 1) & 7)) << 4);
        }
        {   // code piece w + 8u: rows 8 (w + 8u) ..
            const int r = (wave + 8 * u) * 8 + (lane >> 3);
            const int rc = r < n_lim ? r : n_lim - 1;
            voC[u] = rc * ldc_ + (((lane & 7) ^ ((r >> 1) & 7)) << 4);
        }
    }
    {
        const int r = wave * 16 + (lane >> 2);
        const int rc = r < n_lim ? r : n_lim - 1;
        voL = rc * ldl + (((lane & 3) ^ ((r >> 2) & 3)) << 4);
    }
    auto issue_stage = [&](int st) {  // K stage st into ring slot st % NS; stages past the end are never requested
        uint8_t* slot = lds + (st % NS) * STAGE;
        SDNQ_DMA16(rsC, slot + A_STAGE + wave * 1024, voC[0], st * (SK / 2));
        SDNQ_DMA16(rsC, slot + A_STAGE + (wave + 8) * 1024, voC[1], st * (SK / 2));
        SDNQ_DMA16(rsL, slot + A_STAGE + C_STAGE + wave * 1024, voL, st * (SK / 64) * 16);
        SDNQ_DMA16(rsA, slot + wave * 1024, voA[0], st * SK);
        SDNQ_DMA16(rsA, slot + (wave + 8) * 1024, voA[1], st * SK);
    };
    int issued = 0;
#pragma nounroll
    for (; issued < nk && issued < NS - 1; ++issued) issue_stage(issued);
    __builtin_amdgcn_sched_barrier(0);
    SDNQ_KERNARGS_NOW("s"(p_.xs), "s"(p_.ws), "s"(p_.bias), "s"(p_.out), "s"(p_.ldc), "s"(p_.bias_dtype));
    // per-channel / per-row epilogue vectors: requested now (behind the prologue's pieces), parked in LDS after the loop.  Branch-free, from
    // clamped addresses, by every thread: a load under a condition makes the compiler wait for it at the join -- and with it for every
    // LDS-DMA piece in front of it in the queue.  16-bit bias elements are fetched as the aligned dword that holds them.
    const int vn = (tid & (BN - 1)) < n_lim ? (tid & (BN - 1)) : n_lim - 1, vm = (tid & (BM - 1)) < m_rows ? (tid & (BM - 1)) : m_rows - 1;
    const float ev_sb = p.ws[n0 + vn], ev_xs = p.xs[m0 + vm];
    u32 ev_bias = 0;
    if constexpr (HAS_BIAS) {
        const int64_t bo = (int64_t)(n0 + vn) * (p.bias_dtype == SDNQ_F32 ? 4 : 2);
        ev_bias = *(const u32*)((const uint8_t*)p.bias + (bo & ~(int64_t)3));
        // (the half is picked when the value is parked)
    }

    v16i acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0;
    const int wm = wave >> 2, wn = wave & 3;
    const int frow = lane & 31, fgrp = lane >> 5;
    uint8_t* const w8 = lds + NS * STAGE;  // the stage's weight operand as int8: [2 halves of 128 columns][128 rows][128 B], chunk ^= (row >> 1) & 7
    // ---- expansion assignment: the 2048 16-byte fragments of a stage (128 rows x 16 runs of 16 columns) are expanded ONCE per workgroup,
    // four per thread: thread t takes row t & 127, runs (t >> 7) + 4 j.  (Expanding in the MFMA lanes repeats every fragment in the
    // two wave rows of the tile: measured 11.3 us against 7.3 for the int8 GEMM at 1024 x 1280 x 1280, profiles/r06_w4_fused_lab.txt.)
    const int er = tid & 127, ec0 = tid >> 7, esw = (er >> 1) & 7;
    const int e_code = A_STAGE + er * 128, e_lut = A_STAGE + C_STAGE + er * 64, e_lsw = (er >> 2) & 3;
    // ---- MFMA fragment addresses: activation row wm * 32 + frow of the ring stage (half h, chunk q * 2 + fgrp), weight row wn * 32 + frow of w8
    const int ra = wm * 32 + frow, rb = wn * 32 + frow;
    const int aoff = ra * 128, asw = (ra >> 1) & 7, boff = rb * 128, bsw = (rb >> 1) & 7;
#pragma nounroll
    for (int kt = 0; kt < nk; ++kt) {
        // stage kt has landed (this wave's pieces; the barrier makes it everybody's): the stages requested behind it stay in flight
        // (the epilogue vectors' loads sit between the prologue's pieces and the stages requested inside the loop)
        constexpr int EV = HAS_BIAS ? 3 : 2;
        if (kt == 0 && issued >= 2) w4_wait_vmcnt<PPW + EV>();
        else if (issued - kt - 1 >= 1) w4_wait_vmcnt<PPW>();
        else w4_wait_vmcnt<0>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's fragment reads of stage kt - 1 (ring and w8) have retired
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (issued < nk) { issue_stage(issued); ++issued; }  // refills the slot stage kt - 1 occupied
        const uint8_t* slot = lds + (kt % NS) * STAGE;
        const bool full = kt < nk - 1 || last_ks == 8;  // (wave-uniform) false: the half stage at the end of K, 128 columns
        // ---- expand: codes + tables -> int8 operand, each fragment once
        {
            v2i cw[4];
            v4i tb[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c = ec0 + 4 * j;  // run of 16 columns: code bytes 8 c .. 8 c + 7, table c >> 2
                cw[j] = *(const v2i*)(slot + e_code + (((c >> 1) ^ esw) << 4) + (c & 1) * 8);
                tb[j] = *(const v4i*)(slot + e_lut + (((c >> 2) ^ e_lsw) << 4));
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c = ec0 + 4 * j;
                if (j >= 2 && !full) break;  // runs 8 .. 15 are the second 128 columns
                *(v4i*)(w8 + (c >> 3) * (BN * 128) + er * 128 + (((c & 7) ^ esw) << 4)) = expand16(cw[j], tb[j]);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // ---- multiply: 128 columns = 4 K sub-steps of 32
        auto half = [&](auto hc) {
            constexpr int h = decltype(hc)::value;
            v4i xa[4], wb[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                xa[q] = *(const v4i*)(slot + h * (BM * 128) + aoff + (((q * 2 + fgrp) ^ asw) << 4));
                wb[q] = *(const v4i*)(w8 + h * (BN * 128) + boff + (((q * 2 + fgrp) ^ bsw) << 4));
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(wb[q], xa[q], acc, 0, 0, 0);
        };
        half(std::integral_constant<int, 0>{});
        if (full) half(std::integral_constant<int, 1>{});
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // the epilogue vectors (their loads are older than every piece the loop waited for) -> LDS, beside the ring
    if (tid < BN) {
        s_sb[tid] = ev_sb;
        if constexpr (HAS_BIAS) {
            const u32 hb = ((n0 + vn) & 1) ? (ev_bias >> 16) : (ev_bias & 0xffffu);  // 16-bit element inside its dword
            s_bias[tid] = p.bias_dtype == SDNQ_F32 ? __uint_as_float(ev_bias) : (p.bias_dtype == SDNQ_BF16 ? __uint_as_float(hb << 16) : f16_bits_to_f32((uint16_t)hb));
        }
    }
    if (tid < BM) s_xs[tid] = ev_xs;
    __syncthreads();  // every wave is done with the ring before it becomes the output staging area (no piece is in flight: none past K was requested)

    // ---- epilogue in the MFMA register layout: lane owns row wm*32 + (lane & 31), channels wn*32 + (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5);
    // out = cast(fma(f32(acc) * xs, ws, bias)) (kernel_wrappers.py:132-144); final 16-bit values leave through LDS as 16-byte row pieces
    constexpr int OUT_ROW = BN * 2 + 16;
    {
        const int ml = wm * 32 + frow;
        const float sa = s_xs[ml];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int nl0 = wn * 32 + 8 * q + 4 * fgrp;
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float vv = (float)acc[4 * q + e] * sa;
                if constexpr (HAS_BIAS) o[e] = fmaf(vv, s_sb[nl0 + e], s_bias[nl0 + e]);
                else o[e] = vv * s_sb[nl0 + e];
            }
            *(v2i*)(lds + ml * OUT_ROW + nl0 * 2) = (v2i){(int)pack2<OUT_T>(o[0], o[1]), (int)pack2<OUT_T>(o[2], o[3])};
        }
    }
    __syncthreads();
    constexpr int PPR = BN * 2 / 16;  // 16-byte pieces per output row
#pragma unroll
    for (int v = tid; v < BM * PPR; v += NT) {
        const int r = v / PPR, c = v % PPR;
        if (r >= m_rows || c * 8 >= n_lim) continue;  // N % 8 == 0: a piece never straddles N
        const v4i val = *(const v4i*)(lds + r * OUT_ROW + c * 16);
        __builtin_nontemporal_store(val, (v4i*)((uint8_t*)p.out + ((int64_t)(m0 + r) * p.ldc + n0 + c * 8) * 2));
    }
}

inline int w4_cu_count() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    static std::atomic<int> cus[64];
    int v = cus[dev].load(std::memory_order_relaxed);
    if (v == 0) {
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        cus[dev].store(v, std::memory_order_relaxed);
    }
    return v;
}

template <int OUT_T, bool HAS_BIAS>
int launch_w4(const void* a, const void* codes, const void* lut, int64_t lda, int64_t m, int64_t n, int64_t k, W4Params p, hipStream_t s) {
    auto kern = gemm_w4_kernel<OUT_T, HAS_BIAS>;
    static std::atomic<uint64_t> attr_devices{0};  // (the attribute belongs to the function ON ONE DEVICE)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return SDNQ_ERR_LAUNCH;
    const uint64_t bit = 1ull << (dev & 63);
    if (!(attr_devices.load(std::memory_order_acquire) & bit)) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess) return SDNQ_ERR_LAUNCH;
        attr_devices.fetch_or(bit, std::memory_order_release);
    }
    const int tiles_m = (int)((m + BM - 1) / BM), tiles_n = (int)((n + BN - 1) / BN);
    const int64_t tiles = (int64_t)tiles_m * tiles_n;
    const int group_m = tiles_m < 8 ? tiles_m : 8;
    const int64_t slots = w4_cu_count();  // one workgroup per CU (120 KB of LDS)
    const int pf_wgs = sdnq_internal_take_prefetch(tiles < slots ? slots - tiles : 0, NT, p.pf_ptr, p.pf_lines);
    hipLaunchKernelGGL(kern, dim3((unsigned)(tiles + pf_wgs)), dim3(NT), LDS_BYTES, s, (const uint8_t*)a, (const uint8_t*)codes, (const uint8_t*)lut, (int)lda, (int)m,
                       (int)n, (int)k, tiles_m, tiles_n, group_m, p);
    SDNQ_CHECK_LAUNCH();
    return SDNQ_OK;
}

inline int64_t w4_env(const char* name, int64_t dflt) {
    const char* e = getenv(name);
    return e ? atoll(e) : dflt;
}

}  // namespace

extern "C" int sdnq_hip_scaled_mm_w4_supported(int mm_dtype, int out_dtype, int64_t m, int64_t n, int64_t k) {
    // where the one launch beats "re-quantize, then multiply" (profiles/r06_w4_fused_lab.txt, graph-replayed on distinct weights):
    // 1024 x 1280 x 1280 10.4 vs 12.4 us (the 232 c x c projections of an SDXL step), 1024 x 3840 x 1280 19.7 vs 19.7; it loses at
    // K = 5120 (31.9 vs 28.9), N = 10240 (48.0 vs 35.1) and from 2048 rows on (19.1 vs 15.3) -- every row block of 64 rows repeats the
    // expansion, and the expand / multiply phases of a stage do not overlap yet
    static const int64_t on = w4_env("SDNQ_HIP_FUSED_LUT4", 1), max_m = w4_env("SDNQ_HIP_FUSED_LUT4_MAX_M", 1024), min_m = w4_env("SDNQ_HIP_FUSED_LUT4_MIN_M", 33),
                         max_k = w4_env("SDNQ_HIP_FUSED_LUT4_MAX_K", 1280), max_n = w4_env("SDNQ_HIP_FUSED_LUT4_MAX_N", 3840);
    if (!on || mm_dtype != SDNQ_MM_I8) return 0;
    if (out_dtype != SDNQ_BF16 && out_dtype != SDNQ_F16) return 0;
    if (m < min_m || m > max_m || n <= 0 || n > max_n || (n % 8) != 0 || k < 128 || k > max_k || (k % 128) != 0) return 0;
    return 1;
}

extern "C" int sdnq_hip_scaled_mm_w4(const void* a, const void* codes, const void* lut, const float* sa, const float* sb, const void* bias, int bias_dtype,
                                     void* out, int out_dtype, int64_t m, int64_t n, int64_t k, int64_t lda, sdnq_stream_t stream) {
    if (!a || !codes || !lut || !sa || !sb || !out) return SDNQ_ERR_NULL;
    if (out_dtype != SDNQ_BF16 && out_dtype != SDNQ_F16) return SDNQ_ERR_UNSUPPORTED;
    if (bias && (bias_dtype < 0 || bias_dtype > 2)) return SDNQ_ERR_DTYPE;
    if (lda == 0) lda = k;
    if (m <= 0 || n <= 0 || k < 128 || (k % 128) != 0 || (n % 8) != 0 || lda < k) return SDNQ_ERR_SHAPE;
    if (m > 0x7fffffffll || n > 0x7fffffffll || 64 * lda + k > 0x7fffffffll || 128 * k > 0x7fffffffll) return SDNQ_ERR_SHAPE;
    if (((uintptr_t)a % 16) || ((uintptr_t)codes % 16) || ((uintptr_t)lut % 16) || ((uintptr_t)out % 16) || (lda % 16)) return SDNQ_ERR_ALIGN;
    W4Params p{};
    p.xs = sa; p.ws = sb; p.bias = bias; p.out = out; p.ldc = n; p.bias_dtype = bias_dtype;
    hipStream_t s = (hipStream_t)stream;
    if (out_dtype == SDNQ_BF16) return bias ? launch_w4<SDNQ_BF16, true>(a, codes, lut, lda, m, n, k, p, s) : launch_w4<SDNQ_BF16, false>(a, codes, lut, lda, m, n, k, p, s);
    return bias ? launch_w4<SDNQ_F16, true>(a, codes, lut, lda, m, n, k, p, s) : launch_w4<SDNQ_F16, false>(a, codes, lut, lda, m, n, k, p, s);
}
