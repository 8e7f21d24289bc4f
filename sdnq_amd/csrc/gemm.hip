// Scaled int8 / fp8 matmul for gfx950 (CDNA4):  out = cast(fma(f32(A.B^T) * sa[m], sb[n], bias))
//
// Replaces the reference's Triton kernel sdnq_scaled_mm_kernel (kernels/triton_scaled_mm.py:127-236,
// wrapper :239-275) and its eager twins int_scaled_mm_torch / fp8_scaled_mm_torch
// (kernel_wrappers.py:132-144).  Same contract, different machine mapping:
//   * both operands are K-contiguous in HBM (a [M][K]; b physical [N][K] == the reference's
//     b[K,N] with strides (1,K)), so every MFMA fragment is one 16-byte (int8) / 32-byte (fp8)
//     K-contiguous vector per lane -- no transposes anywhere;
//   * int8 -> v_mfma_i32_32x32x32_i8 (exact int32 accumulate, so any K order is bit-identical);
//     fp8  -> v_mfma_scale_f32_32x32x64_f8f6f4 with neutral E8M0 scales (the only full-rate fp8 MFMA);
//   * HBM -> LDS goes through global_load_lds_dwordx4 (LDS-DMA, no VGPR round trip) into an NS-deep ring of
//     stages; each stage is BK = 128 bytes of K for BM + BN rows.  The wave only waits with a COUNTED
//     s_waitcnt vmcnt(n) for the stage it is about to read, so NS-2 stages stay in flight across the single
//     raw s_barrier per K-step (a __syncthreads() would drain the DMA queue);
//   * LDS rows are 128 B with the 16-byte chunk index XOR-swizzled by (row >> 1) & 7: conflict-free
//     ds_read_b128 for the 32-row MFMA fragment pattern.  LDS-DMA writes lane-linearly, so the swizzle is
//     applied to the per-lane SOURCE address (and again on the read);
//   * MFMA computes the TRANSPOSED tile (A-operand = weight rows n, B-operand = activation rows m):
//     a lane then owns ONE activation row m and 16 output channels in runs of 4, so the epilogue
//     packs 4 results per 8-byte LDS store and the tile leaves the CU as full 16-byte row segments;
//   * epilogue: fma(f32(acc) * sa[m], sb[n], bias) -- single-rounding FMA like tl.fma
//     (triton_scaled_mm.py:225) and CPU addcmul; int32 -> f32 conversion is RNE above 2^24;
//   * optional fused low-rank (SVD) bias: bias2d[m][n] = cast_svd(f32(bias[n]) + sum_r t[m][r] * up[n][r])
//     (linear_int8.py:57-62) and zero-point term f32(rowsum[m]) * sa[m] * zp[n] (linear_int8.py:65-69)
//     are produced in the epilogue instead of materialising an [M][N] bias in HBM;
//   * block -> tile map is XCD-aware: the 8 XCDs (private L2s) each walk a contiguous range of tiles.
#include <atomic>
#include <vector>
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "sdnq_dev.h"

// gemm_ks.hip: the 64 x 80 tile with an in-workgroup K split (8 waves, partial sums reduced through LDS) -- tile id 28
bool sdnq_internal_ks_eligible(int64_t m, int64_t n, int64_t k, int64_t lda, int64_t ldb);
bool sdnq_internal_ks_preferred(int64_t m, int64_t n, int64_t k);
int sdnq_internal_scaled_mm_ks(const void* a, const void* b, const float* sa, const float* sb, const void* bias, int bias_dtype, void* out,
                               int out_dtype, int64_t m, int64_t n, int64_t k, int64_t lda, int64_t ldb, int64_t ldc, hipStream_t s);

namespace {

constexpr int BKB = 128;  // default bytes of K per LDS stage row (a full 128-byte line per row)

__device__ const uint4 g_zero16 = {0u, 0u, 0u, 0u};  // source of K-tail chunks for the LDS-DMA

#ifdef SDNQ_TRACE2
__device__ unsigned g_trace2[64];
#endif
#ifdef SDNQ_TRACE  // development build only: per-workgroup phase timestamps (shader clock), read back by tools/trace_gemm.py
__device__ unsigned long long g_trace[4096 * 8];
// launch filter (tools/trace_in_step.py): only launches of this M x N x K are recorded, so that after a replay of a whole step's graph the
// buffer holds the LAST such launch as it ran INSIDE the step (cold weights, the row quantizer in front of it); 0 = every launch
__device__ int g_trace_shape[4];
#define TRACE(slot)                                                                    \
    do {                                                                               \
        if (threadIdx.x == 0 && blockIdx.x < 4096 && tr_on) g_trace[blockIdx.x * 8 + (slot)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define TRACE(slot) do { } while (0)
#endif

struct GemmParams {
    const uint8_t* a;   // [M][K]
    const uint8_t* b;   // [N][K]
    const float* sa;    // [M]
    const float* sb;    // [N]
    const void* bias;   // [N] or [M][ld_bias] or null
    void* out;          // [M][N]
    const void* lr_t;   // [M][R]  low-rank activations (svd dtype) or null
    const void* lr_up;  // [N][R]
    const int32_t* zp_rowsum;  // [M] or null
    const float* zp;           // [N] or null
    const float* a_zp;         // [M] activation zero point (uint8 matmul) or null
    const float* wcs;          // [N] f32(colsum(b)) * sb (uint8 matmul) or null
    int64_t M, N, K;   // K in BYTES of one operand row (== elements for int8 / fp8)
    int64_t lda, ldb;  // operand row strides in bytes (0: K)
    int64_t ldc;       // output row stride in ELEMENTS (0: N); with out_hw > 0: channels per image of the conv output (0: N)
    int64_t out_hw;    // 0: out is [M][N].  > 0: conv output [B][N][out_hw] with m = b * out_hw + pixel (NCHW, conv_int8.py:81-87)
    int64_t ld_bias;
    int64_t zp_k;      // K of the uint8 matmul's K * (xzp * wzp) term when it is not this launch's K (one group of a grouped conv: the whole unfolded row); 0: K;
                       // < 0: -K in the CONV forwards' rounding order, (xzp * K) * wzp added unfused (conv_uint8.py:66)
    int bias_ndim;
    int bias_dtype;  // SdnqFloat of bias (and of lr_t / lr_up, which share the svd dtype)
    int rank;
    int tiles_m, tiles_n;
    int group_m;  // m-strips per rasterization group
    // host-computed multipliers for the divisions of the tile mapping: q = umulhi(n, ceil(2^32 / d)) is exact for n * d <= 2^32 (0 stands for
    // d = 1); `fastmap` says the grid is small enough.  A software 32-bit division is ~30 scalar instructions and the 64-bit one of the
    // grouped launch ~150 -- code a wave executes once, at instruction-fetch speed (the prologue is fetched from the last-level cache on
    // every launch of the bs = 1 steps), in front of its first LDS-DMA
    uint32_t mg_per_group, mg_group_m, mg_tail, mg_unit;
    int fastmap, fastunit;
    int swz;      // LDS chunk swizzle mask (7; 0 only for experiments)
    // several output tensors (linked projections: one GEMM over the stacked weights of to_q / to_k / to_v, each layer's output in
    // its own [M][seg_n] tensor): channel n goes to out_seg[n / seg_n]; seg_n % 8 == 0, so a 16-byte piece never straddles tensors
    void* out_seg[4];
    int64_t seg_n;  // 0: single output p.out
    // grouped launch (sdnq_hip_scaled_mm_grouped): the output channels are cut into units of unit_n channels, each with its own
    // weight rows / scales / bias (the layers' own parameters, no stacked copy); BN divides unit_n, so a tile lies inside one unit
    const SdnqGemmUnit* units;  // device table or null
    int64_t unit_n;
    // weight prefetch (sdnq_hip_prefetch_hint): workgroups past the last tile pull these ranges (128-byte lines) into the memory-side
    // cache while the tiles compute; see launch_one
    const uint8_t* pf_ptr[4];
    int pf_lines[4];
};

// Where one workgroup's BN output channels live: weight rows, per-channel vectors and the output matrix they belong to.
struct TileView {
    const uint8_t* b;      // weight row of the tile's first channel
    const float* sb;       // its scale
    const void* bias;      // 1-D bias vector the tile indexes with bias0 + i
    int64_t bias0;
    uint8_t* out;          // &out_matrix[0][first channel of the tile]
    int64_t out_ld;        // channels per row of that matrix
    int64_t n_lim;         // valid channels from the tile's first one (may exceed BN)
};

// the 16-byte store of a finished output piece.  SDNQ_STORE_MODE (lab builds): 1 non-temporal, 2 write-through (sc1), 3 sc0 sc1
// Default 1: the outputs are written once and not read again by this kernel; non-temporal stores measured -2..-9 % on the
// output-heavy GEMMs (1024 x 10240 x 1280: 24.9 -> 22.6 us, 4096 x 5120 x 640: 31.3 -> 28.9 us) and never slower (tools/micro/gemm_lab.hip).
#ifndef SDNQ_STORE_MODE
#define SDNQ_STORE_MODE 1
#endif
__device__ __forceinline__ void store16(void* dst, const uint4& v) {
#if SDNQ_STORE_MODE == 1
    __builtin_nontemporal_store((v4i){(int)v.x, (int)v.y, (int)v.z, (int)v.w}, (v4i*)dst);
#elif SDNQ_STORE_MODE == 2
    const v4i w = {(int)v.x, (int)v.y, (int)v.z, (int)v.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(dst), "v"(w) : "memory");
#elif SDNQ_STORE_MODE == 3
    const v4i w = {(int)v.x, (int)v.y, (int)v.z, (int)v.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(dst), "v"(w) : "memory");
#else
    *(uint4*)dst = v;
#endif
}

// address of output row gm, channel `c` of the tile (global channel gn0 = n0 + c)
__device__ __forceinline__ uint8_t* out_piece(const GemmParams& p, const TileView& tv, int64_t gm, int64_t gn0, int c, int out_b) {
    if (p.seg_n == 0) return tv.out + (gm * tv.out_ld + c) * out_b;
    const int64_t seg = gn0 / p.seg_n;
    return (uint8_t*)p.out_seg[seg] + (gm * p.seg_n + (gn0 - seg * p.seg_n)) * out_b;
}

template <int MM> struct MmaTraits;
template <> struct MmaTraits<SDNQ_MM_I8> {
    typedef v16i acc_t;
    static constexpr int MS = 32;  // the MFMA computes MS x MS output tiles
    static constexpr int KB = 32;  // bytes of K per MFMA per operand row
    static __device__ __forceinline__ void zero(acc_t& c) {
#pragma unroll
        for (int i = 0; i < 16; ++i) c[i] = 0;
    }
    static __device__ __forceinline__ float tof(const acc_t& c, int i) { return (float)c[i]; }
};
template <> struct MmaTraits<SDNQ_MM_FP8> {
    typedef v16f acc_t;
    static constexpr int MS = 32;
    static constexpr int KB = 64;
    static __device__ __forceinline__ void zero(acc_t& c) {
#pragma unroll
        for (int i = 0; i < 16; ++i) c[i] = 0.0f;
    }
    static __device__ __forceinline__ float tof(const acc_t& c, int i) { return c[i]; }
};

// Plain float GEMMs (the dequantize-then-F.linear branch at M > 32, layers/linear/forward.py:25-26): same tiles, same LDS
// image, K counted in BYTES.  32 bytes of a K row feed one v_mfma_f32_32x32x16_{bf16,f16} (8 elements per lane) or four
// v_mfma_f32_32x32x2_f32 (lane holds 4 consecutive floats; MFMA r consumes float r of both operands, so lanes < 32 cover
// k = 0..3 and lanes >= 32 cover k = 4..7 of the segment).  No scales: out = cast(acc + bias).
enum { MM_BF16 = 2, MM_F16 = 3, MM_F32 = 4 };
template <int MM> struct FloatMma {
    typedef v16f acc_t;
    static constexpr int MS = 32;
    static constexpr int KB = 32;
    static __device__ __forceinline__ void zero(acc_t& c) {
#pragma unroll
        for (int i = 0; i < 16; ++i) c[i] = 0.0f;
    }
    static __device__ __forceinline__ float tof(const acc_t& c, int i) { return c[i]; }
};
template <> struct MmaTraits<MM_BF16> : FloatMma<MM_BF16> {};
template <> struct MmaTraits<MM_F16> : FloatMma<MM_F16> {};
template <> struct MmaTraits<MM_F32> : FloatMma<MM_F32> {};
// The float16 quantized matmul (quantized_matmul_dtype="float16": layers/linear/linear_fp16.py:16-74, fp_scaled_mm_func
// kernel_wrappers.py:207-211): float16 operands -- activations row-quantized to float16 with a row scale, weights as float16 codes --
// on v_mfma_f32_32x32x16_f16 with the SCALED epilogue of the int8 / fp8 matmuls (out = cast(fma(acc * sa, sb, bias))).  K counted in
// bytes like the float GEMMs.  Round 6.
enum { MM_F16S = 10 };
template <> struct MmaTraits<MM_F16S> : FloatMma<MM_F16S> {};
// Fused dequantize + float GEMM (the reference's DEFAULT mode, use_quantized_matmul=False: dequantize_symmetric / _asymmetric
// then F.linear, dequantizer.py:52-84 + layers/linear/forward.py:25-26) for row-wise 8-bit weights: the B operand stays int8 /
// uint8 in HBM and in LDS (HALF the bytes of a dequantized bf16 copy, and no dequantize launch, no [N][K] float matrix in HBM);
// every lane turns the 8 codes of its weight row into the 8 bf16 / f16 values of its MFMA fragment between LDS and the matrix
// core -- W = cast(fma(u, s, c)) with u the byte, c = -128 s (signed) or the zero point (unsigned): bit for bit the value
// sdnq_hip_dequant writes -- and the activations are the plain 16-bit operand.  A stage row is BK bytes of A and BK / 2 of B.
// MM_W8*: signed codes (the int8 checkpoints), W = cast(f32(v) * s);  MM_W8*U: unsigned codes with a zero point, W = cast(fma(u, s, zp)).
// Two instantiations, not a run-time switch: a branch inside the conversion splits the basic block the MFMAs live in and the
// scheduler no longer overlaps one fragment's conversion with the previous fragment's MFMAs (measured: 14.3 against 13.1 ms per step).
enum { MM_W8BF16 = 5, MM_W8F16 = 6, MM_W8BF16U = 8, MM_W8F16U = 9 };
template <> struct MmaTraits<MM_W8BF16> : FloatMma<MM_W8BF16> {};
template <> struct MmaTraits<MM_W8F16> : FloatMma<MM_W8F16> {};
template <> struct MmaTraits<MM_W8BF16U> : FloatMma<MM_W8BF16U> {};
template <> struct MmaTraits<MM_W8F16U> : FloatMma<MM_W8F16U> {};
// int8 on v_mfma_i32_16x16x64_i8: 16 x 16 output tiles (4 accumulators per lane: lane l holds n = 4 (l >> 4) + 0..3 of m = l & 15 --
// again a run of 4 output channels of one row), 64 bytes of K per instruction.  Same LDS image and loaders; what it buys is
// tile shapes in multiples of 16: 64 x 80 tiles cut the 1024 x 1280 outputs of the SDXL attention / feed-forward projections into
// exactly 256 workgroups with 25 % fewer LDS-fill bytes per CU than 160 tiles of 64 x 128.
enum { MM_I8_16 = 7 };
template <> struct MmaTraits<MM_I8_16> {
    typedef v4i acc_t;
    static constexpr int MS = 16;
    static constexpr int KB = 64;
    static __device__ __forceinline__ void zero(acc_t& c) { c = (v4i){0, 0, 0, 0}; }
    static __device__ __forceinline__ float tof(const acc_t& c, int i) { return (float)c[i]; }
};
template <int MM> constexpr bool is_w8a16 = (MM == MM_W8BF16 || MM == MM_W8F16 || MM == MM_W8BF16U || MM == MM_W8F16U);
template <int MM> constexpr bool is_w8_bf16 = (MM == MM_W8BF16 || MM == MM_W8BF16U);
template <int MM> constexpr bool is_w8_signed = (MM == MM_W8BF16 || MM == MM_W8F16);
template <int MM> constexpr bool is_float_mm = ((MM >= MM_BF16 && MM <= 6) || is_w8a16<MM>);
struct WRow { float s, c; };  // scale and additive constant of this lane's weight row (MM_W8*)

// LDS byte offset of 16-byte chunk c (0..7) of tile row r; rows are 128 B, chunk XOR-swizzled.
// LDS byte offset of 16-byte chunk c of tile row r, XOR-swizzled so that the 16 lanes of a ds_read_b128 group (16
// distinct rows, same logical chunk) land on 16 distinct 16-byte slots of the 256-byte bank row:
//   128-byte rows (8 chunks, 2 rows per bank row): chunk ^= (r >> 1) & 7;   64-byte rows (4 chunks, 4 rows): chunk ^= (r >> 2) & 3
template <int BK> __device__ __forceinline__ int lds_off(int r, int c, int swz) {
    if constexpr (BK == 128) return r * 128 + ((c ^ ((r >> 1) & swz)) << 4);
    else return r * 64 + ((c ^ ((r >> 2) & (swz & 3))) << 4);
}

template <int T_ID> __device__ __forceinline__ float ldf(const void* p, int64_t i) { return FT<T_ID>::load(p, i); }

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// f(integral_constant<N-1>), ..., f(integral_constant<0>): a compile-time unrolled countdown
template <int N, typename F> __device__ __forceinline__ void static_for_down(F&& f) {
    if constexpr (N > 0) {
        f(std::integral_constant<int, N - 1>{});
        static_for_down<N - 1>(f);
    }
}

template <int N, int I = 0, typename F> __device__ __forceinline__ void static_for_up(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for_up<N, I + 1>(f);
    }
}

template <int N> __device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// EPI_LRFAST: EPI_LOWRANK restricted (by the launcher) to what the register-layout low-rank epilogue handles -- rank-32 16-bit factors,
// no zero-point terms, 16-bit output -- with the general staged path compiled OUT: the 256x256 tiles (128 live accumulators per lane)
// are only instantiated for it; with both paths in one kernel the allocator spilled 46-59 registers (round-2 verdict, cfg5's
// dominant kernel), tools/check_spills.py now keeps every kernel free of scratch.
enum { EPI_NONE = 0, EPI_BIAS1D = 1, EPI_BIAS2D = 2, EPI_LOWRANK = 3, EPI_LRFAST = 4 };
template <int EPI> constexpr bool is_lr = (EPI == EPI_LOWRANK || EPI == EPI_LRFAST);

// rows of final values the register-layout epilogues stage at a time (LDS: 160 KB minus <= 4 KB of per-channel vectors)
template <int EPI, int BM, int BN, int OUT_B> constexpr int epi_chunk_rows() {
    const int lr = is_lr<EPI> ? (BM + BN) * 64 : 0;
    return (BM <= 128 || lr + BM * (BN * OUT_B + 16) <= 155 * 1024) ? BM : 128;
}

// f32(acc) * sa, the first step of every scaled epilogue; LP: carried on bf16 tensors (see gemm_kernel)
template <bool LP>
__device__ __forceinline__ float acc_times_sa(float a, float sa) {
    if constexpr (LP) return FT<SDNQ_BF16>::round(FT<SDNQ_BF16>::round(a) * sa);
    else return a * sa;
}

__device__ __forceinline__ float ldf_rt(const void* p, int64_t i, int dt) {
    return dt == SDNQ_F32 ? ((const float*)p)[i] : (dt == SDNQ_BF16 ? bf16_bits_to_f32(((const uint16_t*)p)[i]) : f16_bits_to_f32(((const uint16_t*)p)[i]));
}

// One MFMA operand fragment (the K-contiguous bytes of tile row `r` this lane feeds to K sub-step `ks`) and the MFMA on it.
template <int MM> struct FragOps {
    typedef v4i frag_t;   // activation-side fragment
    typedef typename std::conditional<is_w8a16<MM>, v2i, v4i>::type fragb_t;  // weight-side fragment as read from LDS
    static constexpr int CPK = MmaTraits<MM>::KB / 16;  // 16-byte chunks of a row per K sub-step (= lane groups of the MFMA)
    template <int BK> static __device__ __forceinline__ frag_t load(const uint8_t* s, int r, int ks, int fgrp, int swz) {
        return *(const v4i*)(s + lds_off<BK>(r, ks * CPK + fgrp, swz));
    }
    template <int BKW> static __device__ __forceinline__ fragb_t loadb(const uint8_t* s, int r, int ks, int fgrp, int swz) {
        if constexpr (is_w8a16<MM>) return *(const v2i*)(s + lds_off<BKW>(r, ks, swz) + fgrp * 8);  // 8 codes = the lane's 8 k values
        else return *(const v4i*)(s + lds_off<BKW>(r, ks * CPK + fgrp, swz));
    }
    // 8 stored bytes -> the 8 16-bit values of the MFMA fragment.  The K loop of the fused dequantize GEMM is bound by THIS VALU work,
    // not by its MFMAs (two 32x32x16 = 64 matrix cycles per fragment of a 64-row wave tile against ~4 cycles per VALU instruction):
    //   signed codes (MM_W8BF16 / MM_W8F16; the int8 checkpoints): v_cvt_f32_i32_sdwa sext(BYTE_n) + v_mul_f32 + packed convert = 20 VALU;
    //   f32(v) * s is the reference's own expression (dequantizer.py:63), one rounding;
    //   unsigned codes with a zero point: v_cvt_f32_ubyteN + fma(u, s, zp) + packed convert = 20 VALU.
    // (Round 3: the signed codes used to take the unsigned route after an XOR with 0x80 -- u = v + 128, c = -128 s, same value after
    // the single rounding of the fma -- 22 VALU.  v_pk_fma_f32 / v_pk_mul_f32 do not help: the compiler un-packs those that sit behind
    // an MFMA, and forcing them packed with inline asm was slower, 13.9 against 13.4 ms per step: a packed fp32 instruction does not
    // overlap the MFMA in flight.)
    static __device__ __forceinline__ v4i dequant8(const v2i& raw, const WRow& wr, u32 flip) {
        float f[8];
        if constexpr (is_w8_signed<MM>) {
            const int w0 = raw[0], w1 = raw[1];
            f[0] = (float)(int)(signed char)(w0 & 0xff) * wr.s; f[1] = (float)(int)(signed char)((w0 >> 8) & 0xff) * wr.s;
            f[2] = (float)(int)(signed char)((w0 >> 16) & 0xff) * wr.s; f[3] = (float)(w0 >> 24) * wr.s;
            f[4] = (float)(int)(signed char)(w1 & 0xff) * wr.s; f[5] = (float)(int)(signed char)((w1 >> 8) & 0xff) * wr.s;
            f[6] = (float)(int)(signed char)((w1 >> 16) & 0xff) * wr.s; f[7] = (float)(w1 >> 24) * wr.s;
#pragma unroll
            for (int e = 0; e < 8; ++e) asm("" : "+v"(f[e]));  // keeps the multiplies scalar (the SLP vectorizer would pair them: see above)
        } else {
            const u32 w0 = (u32)raw[0], w1 = (u32)raw[1];
            f[0] = fmaf((float)(w0 & 0xffu), wr.s, wr.c); f[1] = fmaf((float)((w0 >> 8) & 0xffu), wr.s, wr.c);
            f[2] = fmaf((float)((w0 >> 16) & 0xffu), wr.s, wr.c); f[3] = fmaf((float)(w0 >> 24), wr.s, wr.c);
            f[4] = fmaf((float)(w1 & 0xffu), wr.s, wr.c); f[5] = fmaf((float)((w1 >> 8) & 0xffu), wr.s, wr.c);
            f[6] = fmaf((float)((w1 >> 16) & 0xffu), wr.s, wr.c); f[7] = fmaf((float)(w1 >> 24), wr.s, wr.c);
        }
        const uint4 pk = is_w8_bf16<MM> ? Vec16<SDNQ_BF16>::pack(f) : Vec16<SDNQ_F16>::pack(f);
        return (v4i){(int)pk.x, (int)pk.y, (int)pk.z, (int)pk.w};
    }
    // the weight-side MFMA operand of a fragment: converted ONCE per fragment (the callers multiply it with every row tile of the wave)
    static __device__ __forceinline__ v4i prep(const fragb_t& wb, const WRow& wr, u32 flip) {
        if constexpr (is_w8a16<MM>) return dequant8(wb, wr, flip);
        else return wb;
    }
    static __device__ __forceinline__ void mma(typename MmaTraits<MM>::acc_t& c, const v4i& w, const frag_t& x) {
        if constexpr (is_w8_bf16<MM>) {
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(v8bf, w), __builtin_bit_cast(v8bf, x), c, 0, 0, 0);
        } else if constexpr (is_w8a16<MM>) {
            c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(v8h, w), __builtin_bit_cast(v8h, x), c, 0, 0, 0);
        } else if constexpr (MM == MM_I8_16) {
            c = __builtin_amdgcn_mfma_i32_16x16x64_i8(w, x, c, 0, 0, 0);
        } else if constexpr (MM == SDNQ_MM_I8) {
            c = __builtin_amdgcn_mfma_i32_32x32x32_i8(w, x, c, 0, 0, 0);
        } else if constexpr (MM == MM_BF16) {
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(v8bf, w), __builtin_bit_cast(v8bf, x), c, 0, 0, 0);
        } else if constexpr (MM == MM_F16 || MM == MM_F16S) {
            c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(v8h, w), __builtin_bit_cast(v8h, x), c, 0, 0, 0);
        } else {
            const v4f wf = __builtin_bit_cast(v4f, w), xf = __builtin_bit_cast(v4f, x);
#pragma unroll
            for (int r = 0; r < 4; ++r) c = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[r], xf[r], c, 0, 0, 0);
        }
    }
};
template <> struct FragOps<SDNQ_MM_FP8> {
    typedef v8i frag_t;
    typedef v8i fragb_t;
    template <int BK> static __device__ __forceinline__ frag_t load(const uint8_t* s, int r, int ks, int fgrp, int swz) {
        const v4i lo = *(const v4i*)(s + lds_off<BK>(r, ks * 4 + fgrp * 2, swz));
        const v4i hi = *(const v4i*)(s + lds_off<BK>(r, ks * 4 + fgrp * 2 + 1, swz));
        return (v8i){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    }
    template <int BK> static __device__ __forceinline__ fragb_t loadb(const uint8_t* s, int r, int ks, int fgrp, int swz) {
        return load<BK>(s, r, ks, fgrp, swz);
    }
    static __device__ __forceinline__ const frag_t& prep(const fragb_t& wb, const WRow&, u32) { return wb; }
    static __device__ __forceinline__ void mma(v16f& c, const frag_t& w, const frag_t& x) {
        c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(w, x, c, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    }
};

// BM x BN block tile, WM x WN wave tile (multiples of 32), NW = (BM/WM)*(BN/WN) waves, NS LDS stages.
// EPI selects the epilogue at compile time.
//
// Code-size discipline: on this machine straight-line code that a wave executes ONCE runs at instruction-fetch
// speed (cold I-cache: ~600 cycles per ~100 instructions were measured in the unrolled pipeline-drain copies of an
// earlier version), and diffusion-size GEMMs only run 5-40 K-steps, so everything outside the K loop is kept
// compact: the DMA prologue and the epilogue are runtime loops, and the K loop has ONE body for fill, steady state
// and drain (stages past the end of K are issued as re-fetches of a valid stage that nobody consumes, which keeps the counted
// vmcnt a constant).
// LD selects how HBM/L2 -> LDS is done:
//   LD_DMA : global_load_lds_dwordx4 into an NS-deep LDS ring (no VGPR round trip, but ~100+ issue cycles per 1-KiB
//            piece and ~1 us latency: bytes in flight are bounded by the LDS ring);
//            (a register-ring loader -- global_load_dwordx4 into VGPRs, then ds_write_b128 -- measured the same and was removed)
//   LD_PIPE: the LD_DMA ring, plus software pipelining of the LDS->register fragment reads: while the MFMAs of K sub-step
//            ks run, the fragments of sub-step ks+1 (or of the next stage, right after the barrier) are already being
//            read into the other of two fragment sets, so neither the LDS latency nor the two waves of a SIMD reading
//            LDS in lock-step after every barrier leave the matrix pipe idle.
//   LD_PP  : the LD_DMA ring driven PING-PONG by the two halves of an 8-wave workgroup (waves 0-3 / 4-7: wave w and wave w+4
//            sit on the same SIMD).  Each half alternates a LOAD phase (all LDS->register fragment reads of one K stage, its
//            share of the LDS-DMA issue for the stage NS-1 ahead, counted vmcnt) with an MFMA phase (the stage's matrix
//            instructions back to back at raised priority), the second half running one phase behind the first, one raw
//            s_barrier per phase: while one wave of a SIMD feeds the matrix pipe, its partner does the issue-heavy work
//            (an LDS-DMA costs its wave ~60-180 issue cycles) that in the lock-step schedules above leaves the pipe idle.
//   LD_8P  : the fine-grained form of the ping-pong (round 3).  A K stage is consumed in TWO phases -- phase 0: every weight
//            fragment + the first half of the wave's activation row blocks, phase 1: the second half -- and each phase is
//            {LDS->register reads of THIS phase, then this wave's share of the LDS-DMA issue (half a stage), raw barrier,
//            lgkmcnt(0), the phase's MFMAs back to back at raised priority, raw barrier}.  The second half of the workgroup
//            (waves 4-7, the SIMD partners of waves 0-3) runs one barrier behind, so at any time one wave of every SIMD is
//            in its MFMA section while its partner does the slow-issuing work (an LDS-DMA piece occupies its wave's issue
//            port for 60-180 cycles, during which an in-order wave cannot feed the matrix pipe).  What differs from LD_PP:
//            reads come BEFORE the DMA issue and are waited for AFTER the barrier (their latency hides under the DMA issue
//            and the barrier instead of delaying the partner's release), the counted vmcnt sits once per stage, and the
//            phases are half as long (8 MFMAs), so neither role starves.  Hazards (slots = intervals between barriers;
//            group 0 reads/issues in even slots, group 1 in odd ones):
//              RAW  stage j is first read in phase 2j; every wave waits for ITS pieces of stage j (vmcnt leaving the AHEAD-1
//                   younger stages in flight) at the end of the read/issue section of phase 2j-1, in front of a barrier that
//                   both groups pass before phase 2j's reads;
//              WAR  the DMA for stage j+AHEAD lands in the ring slot of stage j-1.  A region read in phase P is safe to
//                   overwrite by DMAs issued in phase P+2 or later (the later group's reads retire at its lgkmcnt(0) one slot
//                   after it issued them; the next barrier orders them before any later issue).  Phase 2j (the first of
//                   stage j) therefore refills only the WEIGHT rows of slot j-1 (last read in phase 2j-2), phase 2j+1 the
//                   activation rows (second half last read in phase 2j-1).
//   LD_HT  : LD_8P on 128-BYTE K tiles staged in HALF-TILES, with NO vector-ALU work in the K loop (round 3; 256x256 tile, 8 waves of
//            128x64).  Two measurements shaped it (tools/micro/dma_shape_lab.hip, tools/micro/dma_mfma_lab.hip, profiles/r03_*):
//            (1) an LDS-DMA piece of 16 rows x 64 B (the 64-byte stages every 256-row tile used so far) moves at ~1/1.75 of the rate
//                of 8 rows x 128 B (full cache lines): 64 vs 100 GB/s per CU;
//            (2) a vector-memory instruction whose ADDRESS registers were written by a vector-ALU instruction of the same wave is
//                held back while the other wave of the SIMD streams MFMAs: global_load_lds behind its 64-bit address arithmetic
//                costs ~2000 cycles per piece beside a dense MFMA stream and 86 alone; buffer_load ... lds with a CONSTANT per-lane
//                voffset and the K advance in the scalar soffset costs 86 in both cases.  That -- not DMA throughput, not latency,
//                not the schedule -- is why every earlier K loop sat at ~60 % matrix utilisation whatever its structure.
//            So: rows are 128 B, the DMAs are buffer loads whose only per-K-tile operand is an SGPR, the LDS read addresses are
//            per-lane constants + immediate offsets (ring slot parity is a compile-time constant: two copies of the K-tile body),
//            and the ring is kept per HALF-TILE (128 rows x 128 B = 16 KiB, one per phase, two pieces per wave), refilled in the
//            order it is consumed:
//              HA0 / HA1 = first / second 64 activation rows of every wave row, HB0 / HB1 = first / second 32 weight rows of every
//              wave column; two slots each (8 x 16 KiB); K tile t (parity p) = phases 4t..4t+3:
//                0: read B0       acc[0][0..1]     refill HB1(t+1)          2: read A1        acc[1][2..3]     refill HA0(t+2)
//                1: read B1       acc[1][0..1]     refill HA1(t+1)          3: read A0(t+1)   acc[0][2..3]     refill HB0(t+2)
//              a half-tile is refilled two phases after its last read (the WAR rule of LD_8P) and needed six phases after that; every
//              phase ends its read/issue section with vmcnt(8): all but the four youngest half-tiles have landed, which covers the
//              reads of the next phase.  Needs K % 128 == 0 (a partial K tile would read the next row's bytes).
//   LD_OV  : the LD_DMA ring with every stream of a K stage OVERLAPPED (round 5; the one-round tiles of the bs = 1 steps).  In LD_DMA a
//            stage is {counted vmcnt, barrier, this wave's DMA pieces, all fragment reads, lgkmcnt, the MFMAs}: the ~85 issue cycles of
//            every LDS-DMA piece, the LDS reads (64 KB per stage of a 64x128 tile: 256 LDS cycles) and the matrix time of the two waves
//            of a SIMD ADD UP -- 907 cycles per 24.5-KB stage = 27 B/clk/CU where the fill path alone sustains 45-59
//            (profiles/r03_dma_shape.txt).  Here the fragments of stage kt + 1 are read (right after the barrier, into the other of two
//            register sets) while the MFMAs of stage kt run, and the DMA pieces of the stage NS ahead are issued BETWEEN those MFMAs --
//            one piece per matrix instruction or two, so the wave's issue port is busy with the piece while its MFMA occupies the pipe.
//            The ring slot of stage kt is free as soon as every wave's reads of it have retired (lgkmcnt(0) in front of the barrier that
//            opens stage kt), i.e. one stage EARLIER than in LD_DMA: NS slots carry NS stages of DMA in flight.
//   LD_OG  : LD_OV with the fragments read one K SUB-STEP ahead instead of one stage (wave tiles of several MFMA tiles; see the loop).
enum { LD_DMA = 0, LD_PIPE = 2, LD_PP = 3, LD_8P = 4, LD_HT = 5, LD_OV = 6, LD_OG = 7 };
constexpr int HT_BYTES = 16384, HT_SLOTS = 8;

// LP: dequantize_fp32=False with BFLOAT16 scales -- the reference's eager epilogue runs on bf16 tensors (kernel_wrappers.py:132-144:
// `int_mm_func(a, b, out_dtype=scale_a.dtype).mul_(scale_a)` then `.mul_(scale_b)` / addcmul): the accumulator is rounded to bf16,
// the product with the activation scale is rounded to bf16, and the last fma (fp32 op-math) is rounded by the bf16 store.
template <int MM, int OUT_T, int EPI, int BM, int BN, int WM, int WN, int NS, int LD, int BK, bool LP = false>
// Kernel arguments: the 14 dwords everything in front of the first LDS-DMA needs come FIRST, as scalars -- a kernarg-preloading build
// (-mllvm -amdgpu-kernarg-preload-count=14, build.sh) delivers them in SGPRs with the wave, so tile mapping, descriptors and the prologue
// DMAs run without a scalar-cache round trip; the rest of the parameter struct is fetched in one batch behind them.  hk_flags: bits 0-7
// group_m, 8-15 swz, 16 fastmap, 17 fastunit, 18 grouped launch.
__global__ __launch_bounds__((BM / WM) * (BN / WN) * 64) void gemm_kernel(const uint8_t* hk_a, const uint8_t* hk_b, int hk_lda, int hk_ldb, int hk_M,
                                                                          int hk_N, int hk_K, int hk_tiles_m, int hk_tiles_n, uint32_t hk_flags,
                                                                          uint32_t hk_mg_per_group, uint32_t hk_mg_group_m, const GemmParams p_) {
    const GemmParams& p = p_;  // the rest of the parameters stay where they are (kernarg memory: p.out_seg is indexed at run time)
    struct Hot {
        const uint8_t *a, *b;
        int64_t lda, ldb, M, N, K;
        int tiles_m, tiles_n, group_m, swz, fastmap, fastunit;
        uint32_t mg_per_group, mg_group_m;
        bool grouped;  // p.units != nullptr, known without the struct: the non-grouped path never waits for it
    };
    const Hot hk = {hk_a, hk_b, hk_lda, hk_ldb, hk_M, hk_N, hk_K, hk_tiles_m, hk_tiles_n, (int)(hk_flags & 0xffu), (int)((hk_flags >> 8) & 0xffu),
                   (int)((hk_flags >> 16) & 1u), (int)((hk_flags >> 17) & 1u), hk_mg_per_group, hk_mg_group_m, ((hk_flags >> 18) & 1u) != 0};
    typedef MmaTraits<MM> MT;
    constexpr int WAVES_M = BM / WM, WAVES_N = BN / WN, NW = WAVES_M * WAVES_N, NT = NW * 64;
    constexpr int MS = MT::MS;                 // MFMA output tile edge: 32, or 16 for MM_I8_16
    constexpr int TM = WM / MS, TN = WN / MS;
    static_assert(WM % MS == 0 && WN % MS == 0, "wave tile in whole MFMA tiles");
    static_assert(MS == 32 || (EPI <= EPI_BIAS1D && BM <= 128), "16x16 tiles: plain epilogues of the small tiles only");
    constexpr int RPP = 1024 / BK;   // tile rows per 1-KiB DMA piece (8 for 128-byte rows, 16 for 64-byte rows)
    constexpr int LPR = BK / 16;     // lanes (16-byte chunks) per row
    // the weight operand of the fused dequantize GEMM holds one byte per K element where the activations hold two
    constexpr int BKW = is_w8a16<MM> ? BK / 2 : BK;  // bytes per stage row of B
    constexpr int RPP_B = 1024 / BKW, LPR_B = BKW / 16;
    // DMA pieces per wave per stage.  Tiles whose A and B rows both split evenly over the waves keep separate A / B piece lists;
    // other tiles (BN = 160, 320: the shapes that cut N = 10240 / 5120 into exactly 256 / 512 workgroups) deal the pieces of the
    // combined [A rows | B rows] stage round-robin to the waves (JOINT): the last piece slot of a wave may be empty, and the
    // counted vmcnt of such a wave is one piece per stage lower.
    constexpr bool JOINT = (BM % (RPP * NW) != 0) || (BN % (RPP_B * NW) != 0);
    static_assert(!is_w8a16<MM> || (BK == 128 && !JOINT), "fused dequantize GEMM: 128-byte activation rows / 64-byte weight rows, even piece split");
    constexpr int A_TOT = BM / RPP, TOT = (BM + BN) / RPP;
    constexpr int A_PIECES = JOINT ? 0 : BM / RPP / NW, B_PIECES = JOINT ? 0 : BN / RPP_B / NW;
    constexpr int PPW = JOINT ? (TOT + NW - 1) / NW : A_PIECES + B_PIECES;
    constexpr int REM = JOINT ? TOT % NW : 0;  // JOINT: waves below REM own PPW pieces, the others PPW - 1 (0: all own PPW)
    static_assert(BM % RPP == 0 && BN % RPP == 0 && BM % 16 == 0, "tile rows must split into DMA pieces");
    static_assert(BK == 64 || BK == 128, "stage rows are 64 or 128 bytes");
    static_assert(LD == LD_HT || (PPW * (NS - 2) <= 63 && NS >= 2), "vmcnt field / stage count");
    constexpr int STAGE_BYTES = BM * BK + BN * BKW;
    constexpr int LDS_STAGES = NS;
    constexpr int OUT_B = FT<OUT_T>::bytes;
    constexpr int CH = BM > 128 ? 64 : BM, ECH = BM / CH;  // general low-rank epilogue: the tile leaves in ECH chunks of CH rows (LDS budget)
    // register-layout epilogues stage FINAL values: the whole tile at once when it fits next to the staged low-rank factor tiles (every
    // wave then finishes its sub-tiles at the same time), else two chunks of 128 rows (only the waves of that half work)
    constexpr int CHR = epi_chunk_rows<EPI, BM, BN, FT<OUT_T>::bytes>(), ECHR = BM / CHR;
    static_assert(ECHR == 1 || CHR % WM == 0, "all rows of a wave lie in one chunk");
    // epilogue staging: final values (simple epilogues) or raw accumulators + low-rank tile (EPI_LOWRANK)
    constexpr int MAIN_BYTES = (LD == LD_HT) ? HT_SLOTS * HT_BYTES : LDS_STAGES * STAGE_BYTES;
    // EPI_LOWRANK: [t tile BM x 64 B | svd_up tile BN x 64 B] (rank-32 factors staged for the low-rank MFMAs), then the raw accumulator
    // plane (32-bit) and the bias2d plane (16-bit: cast_svd(bias + low-rank), linear_int8.py:57-62) of one chunk
    constexpr int LR_BYTES = is_lr<EPI> ? (BM + BN) * 64 : 0;
    constexpr int B2_ROW = BN * 2 + 16;
    constexpr int EPI_GEN = LR_BYTES + CH * (BN * 4 + 16) + CH * B2_ROW, EPI_REG = LR_BYTES + CHR * (BN * OUT_B + 16);
    constexpr int EPI_BYTES = is_lr<EPI> ? ((EPI_GEN > EPI_REG || OUT_T == SDNQ_F32) ? EPI_GEN : EPI_REG) : EPI_REG;  // f32 low-rank outputs: general path only
    constexpr int VEC_OFF = MAIN_BYTES > EPI_BYTES ? MAIN_BYTES : EPI_BYTES;  // per-channel epilogue vectors live after the ring
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    float* s_sb = (float*)(lds + VEC_OFF);  // [BN] column scales
    float* s_bias = s_sb + BN;              // [BN] 1-D bias as f32
    float* s_zp = s_bias + BN;              // [BN] zero points (EPI_LOWRANK)
    float* s_wcs = s_zp + BN;               // [BN] scaled weight column sums (EPI_LOWRANK, uint8 matmul)

    // Every parameter the prologue needs, fetched in ONE batch: left to itself the compiler issues the s_loads of the by-value struct
    // where each field is first used, and the tile mapping -> TileView -> descriptor chain then waits for three or four DEPENDENT
    // round trips to a cold scalar cache (~600-800 cycles each) before the first LDS-DMA can be issued.  The empty asm makes all of
    // them live in SGPRs here, so the loads go out together behind a single s_waitcnt.
#ifndef SDNQ_NO_KERNARG_BATCH
    asm volatile("" ::"s"(hk_a), "s"(hk_b), "s"(hk_lda), "s"(hk_ldb), "s"(hk_M), "s"(hk_N), "s"(hk_K), "s"(hk_tiles_m), "s"(hk_tiles_n), "s"(hk_flags),
                 "s"(hk_mg_per_group), "s"(hk_mg_group_m));
#ifndef SDNQ_PRELOAD_GEMM  // without preload everything is fetched here; with it the struct's fields are requested below, behind the DMAs
    asm volatile("" ::"s"(p_.sb), "s"(p_.bias), "s"(p_.out), "s"(p_.units), "s"(p_.ldc), "s"(p_.unit_n), "s"(p_.mg_tail), "s"(p_.mg_unit),
                 "s"(p_.bias_dtype), "s"(p_.seg_n), "s"(p_.out_hw), "s"(p_.sa));
#endif
#endif
#ifdef SDNQ_TRACE
    const bool tr_on = g_trace_shape[0] == 0 || (g_trace_shape[0] == hk_M && g_trace_shape[1] == hk_N && g_trace_shape[2] == hk_K);
#endif
    TRACE(0);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;

    // L2-aware tile order. (1) block b runs on XCD b % 8 (private 4 MiB L2 each): give every XCD a CONTIGUOUS range of
    // the tile sequence. (2) the sequence itself is grouped: GROUP_M m-strips are walked together, m fastest, so the
    // ~32 workgroups resident on one XCD at a time cover a near-square GROUP_M x (32/GROUP_M) patch of tiles and
    // share both their A strips and their B slabs in that L2 (a 1 x 32 row of tiles would re-fetch every B slab from
    // MALL/HBM for each m-strip: measured 51% of wave cycles parked on vmcnt/barrier at 16384 x 8192 x 4096).
    const int nwg = hk.tiles_m * hk.tiles_n;
    int bid = blockIdx.x;
    if (bid >= nwg) {
        // a prefetch workgroup (launch_one appends them when the launch leaves workgroup slots free): one dword of every 128-byte line of
        // the NEXT layers' weights, nothing kept -- they are in the Infinity Cache when their own GEMM asks for them
        const int t = (bid - nwg) * (int)blockDim.x + (int)threadIdx.x, stride = ((int)gridDim.x - nwg) * (int)blockDim.x;
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
#ifdef SDNQ_TRACE
        if (threadIdx.x == 0 && blockIdx.x < 4096) g_trace[blockIdx.x * 8] = 0;  // (not a tile: tools/trace_*.py count rows with an entry stamp)
#endif
        return;
    }
    {
        const int q = nwg / 8, r = nwg % 8, xcd = bid % 8, j = bid / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    }
    int tile_m, tile_n;
    // (operands are wave-uniform; __umulhi is selected as a VECTOR multiply, so the quotient is pinned back to a scalar register --
    // left in a VGPR, a tile coordinate / unit index turned the unit-table load into vector loads and every buffer descriptor derived
    // from it into VGPRs, i.e. a readfirstlane "waterfall" loop around each LDS-DMA of the K loop)
    auto fdiv = [](uint32_t n, uint32_t mg) -> uint32_t { return mg ? (uint32_t)__builtin_amdgcn_readfirstlane((int)__umulhi(n, mg)) : n; };
    if (hk.fastmap) {
        const int per_group = hk.group_m * hk.tiles_n;
        const int gid = (int)fdiv((uint32_t)bid, hk.mg_per_group), first_m = gid * hk.group_m;
        const bool tail = (hk.tiles_m - first_m) < hk.group_m;
        const int gsz = tail ? (hk.tiles_m - first_m) : hk.group_m;
        const int in_g = bid - gid * per_group;
        if (tail) tile_n = (int)fdiv((uint32_t)in_g, p.mg_tail);  // (a branch, not a select: only the last group waits for the struct)
        else tile_n = (int)fdiv((uint32_t)in_g, hk.mg_group_m);
        tile_m = first_m + in_g - tile_n * gsz;
    } else {
        const int per_group = hk.group_m * hk.tiles_n;
        const int gid = bid / per_group, first_m = gid * hk.group_m;
        const int gsz = (hk.tiles_m - first_m) < hk.group_m ? (hk.tiles_m - first_m) : hk.group_m;
        const int in_g = bid - gid * per_group;
        tile_m = first_m + in_g % gsz;
        tile_n = in_g / gsz;
    }
    const int64_t m0 = (int64_t)tile_m * BM, n0 = (int64_t)tile_n * BN;
    const int K = (int)hk.K;
    // Where the tile's weight rows are -- all the prologue DMAs need of the TileView.  The rest of it (scale / bias / output of the
    // tile) is filled in AFTER the prologue DMAs are issued: those fields come out of the parameter struct, and a scalar load in front
    // of the DMAs means a round trip to the (cold) scalar cache before the first byte is requested.
    TileView tv;
    SdnqGemmUnit un = {};
    int64_t un_d = 0;
    if (hk.grouped) {  // grouped launch: this tile lies inside ONE unit of one layer (BN divides unit_n); wave-uniform loads
        const int64_t u = hk.fastunit ? (int64_t)fdiv((uint32_t)n0, p.mg_unit) : n0 / p.unit_n;
        un_d = n0 - u * p.unit_n;
        un = p.units[u];
        // The unit-table entry is read with vector loads (global memory the compiler cannot prove read-only), so the weight pointer of
        // this path lives in VGPRs -- and with it the merged value of both paths: every LDS-DMA of the weight operand then sat in a
        // readfirstlane "waterfall" loop (6-24 of them per kernel, the K loop's included).  The value is wave-uniform: pin it to SGPRs.
        const uint64_t bv = (uint64_t)((const uint8_t*)un.b + un_d * hk.ldb);
        const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)bv), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(bv >> 32));
        tv.b = (const uint8_t*)(((uint64_t)hi << 32) | lo);
        tv.n_lim = p.unit_n - un_d;
    } else {
        tv.b = hk.b + n0 * hk.ldb;
        tv.n_lim = hk.N - n0;
    }

    // ---- LDS-DMA assignment: piece = 8 tile rows x 128 B; lane l -> row l/8, physical chunk l%8 ----------------
    // wave w owns A pieces w, w+NW, ... and B pieces w, w+NW, ...; the source chunk is the swizzle-inverse of the
    // physical chunk so that LDS stays lane-linear (base + lane*16) as the DMA requires.
    // The DMAs are BUFFER loads (buffer_load_dwordx4 ... lds): a per-lane byte offset that never changes (row * pitch + chunk, relative to
    // the tile's first row) in a VGPR, the K advance in the scalar offset operand -- no vector-ALU instruction feeds a DMA address.
    // (round 3, tools/micro/dma_mfma_lab.hip: a global_load_lds whose 64-bit address comes out of vector-ALU arithmetic costs ~2000
    // cycles beside a wave that streams MFMAs on the same SIMD, 86 alone; the buffer form costs 86 in both cases.)
    // K tail (K % BK != 0): the partial stage is consumed FIRST -- it is fetched by the prologue (executed once, so its per-lane
    // "past K -> zeros" select may cost what it likes) and the K loop only ever fetches whole stages: the accumulation order over K
    // changes nothing for int8 (exact) and only the fp32 summation order for fp8 / float.  Logical stage j = 0 is K stage nk - 1,
    // j >= 1 is K stage j - 1; stages past the end re-fetch K stage 0 (never consumed; keeps the counted vmcnt a constant).
    int voff[PPW];
    // row of this lane inside a DMA piece and the (swizzle-inverse) 16-byte chunk it fetches, per operand geometry
    auto r8_of = [&](bool isA) { return lane / (isA ? LPR : LPR_B); };
    auto chunk_of = [&](int r, bool isA) {
        return (isA ? BK : BKW) == 128 ? ((lane & 7) ^ ((r >> 1) & hk.swz)) : ((lane & 3) ^ ((r >> 2) & (hk.swz & 3)));
    };
    const bool full = REM == 0 || wave < REM;  // wave-uniform: this wave owns PPW pieces (else PPW - 1)
    // (operand, piece inside the operand) of this wave's piece slot i
    auto piece_of = [&](int i, bool& isA) {
        if constexpr (JOINT) {
            int pc = i * NW + wave;
            if (pc >= TOT) pc = TOT - 1;  // the empty slot of a wave with PPW - 1 pieces: never issued
            isA = pc < A_TOT;
            return isA ? pc : pc - A_TOT;
        } else {
            isA = i < A_PIECES;
            return (isA ? i : i - A_PIECES) * NW + wave;
        }
    };
    const int nk = (K + BK - 1) / BK;
    const int K_B = is_w8a16<MM> ? K / 2 : K;  // bytes of a B row
    const bool has_tail = (K % BK) != 0;
    const uint8_t* baseA = hk.a + m0 * hk.lda;
    // Descriptors with the operands' TRUE extents from the tile's first row (the valid rows' bytes; clamped to the 31-bit field for
    // matrices beyond 2 GiB, whose stages are whole anyway): a DMA lane that runs past the end of the matrix reads zeros.  Round 4: with
    // K shorter than one stage (K = 32 on a 128-byte stage) the ring's filler fetches -- whole stages that nobody consumes, issued to keep
    // the counted vmcnt a constant -- read up to 96 bytes past the last row, and a tiny operand at the end of a mapped segment took the
    // GPU down with a memory access fault (found by tools/fuzz_modes.py; every shipped model has K >= 320).
    const int64_t extA = (hk.M - 1 - m0) * hk.lda + K, extB = (tv.n_lim - 1) * hk.ldb + (is_w8a16<MM> ? K / 2 : K);
    auto rsA = SDNQ_MAKE_RSRC_N(baseA, extA < 0x7fffffffll ? extA : 0x7fffffffll);
    auto rsB = SDNQ_MAKE_RSRC_N(tv.b, extB < 0x7fffffffll ? extB : 0x7fffffffll);
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        bool isA;
        const int piece = piece_of(i, isA);
        const int r = piece * (isA ? RPP : RPP_B) + r8_of(isA);
        const int c = chunk_of(r, isA);
        // clamp: rows past the edge are computed on valid memory and never stored
        if (isA) {
            int64_t g = r;
            if (m0 + g >= hk.M) g = hk.M - 1 - m0;
            voff[i] = (int)(g * hk.lda) + c * 16;
        } else {
            const int64_t g = r < tv.n_lim ? r : tv.n_lim - 1;
            voff[i] = (int)(g * hk.ldb) + c * 16;
        }
    }
    // logical K offset (bytes) of this lane's chunk inside a stage row (recomputed: 2 VALU, prologue only)
    auto kofs = [&](int i) {
        bool isA;
        const int piece = piece_of(i, isA);
        return chunk_of(piece * (isA ? RPP : RPP_B) + r8_of(isA), isA) << 4;
    };
    int slot_i = 0;  // ring slot the next issued stage goes to
    // this wave's piece slots [I0, I1) of LOGICAL stage j >= 1 into ring slot `slot`: whole K stages only
    auto issue_range = [&](int j, int slot, auto i0c, auto i1c) {
        constexpr int I0 = decltype(i0c)::value, I1 = decltype(i1c)::value;
        uint8_t* stage = lds + slot * STAGE_BYTES;
        int st = j - (has_tail ? 1 : 0);
        st = st < nk - (has_tail ? 1 : 0) ? st : 0;
#pragma unroll
        for (int i = I0; i < I1; ++i) {
            if (JOINT && i == PPW - 1 && !full) break;  // this wave's last slot is empty
            bool isA;
            const int piece = piece_of(i, isA);
            uint8_t* dst = stage + (isA ? 0 : BM * BK) + piece * 1024;
            if (isA) SDNQ_DMA16(rsA, dst, voff[i], st * BK);
            else SDNQ_DMA16(rsB, dst, voff[i], st * BKW);
        }
    };
    // logical stage 0 when K has a tail: K stage nk - 1, chunks past K from a 16-byte zero constant (plain global_load_lds)
    auto issue_tail = [&](int slot) {
        uint8_t* stage = lds + slot * STAGE_BYTES;
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            if (JOINT && i == PPW - 1 && !full) break;
            bool isA;
            const int piece = piece_of(i, isA);
            uint8_t* dst = stage + (isA ? 0 : BM * BK) + piece * 1024;
            const int k0 = (nk - 1) * (isA ? BK : BKW);
            const uint8_t* sp = (k0 + kofs(i) < (isA ? K : K_B)) ? (isA ? baseA : tv.b) + voff[i] + k0 : (const uint8_t*)&g_zero16;
            __builtin_amdgcn_global_load_lds((gptr_t)sp, (lptr_t)dst, 16, 0, 0);
        }
    };
    auto issue = [&](int j) {  // j >= 1 (the K loop), or j == 0 without a K tail
        const int slot = slot_i;
        slot_i = (slot_i + 1 == NS) ? 0 : slot_i + 1;
        issue_range(j, slot, std::integral_constant<int, 0>{}, std::integral_constant<int, PPW>{});
    };

    typename MT::acc_t acc[TN][TM];  // [n-subtile][m-subtile]; MFMA A-operand = weights (n), B-operand = activations (m)
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) MT::zero(acc[i][j]);

    constexpr int AHEAD = NS - 1;  // stages in flight ahead of the one being consumed (LD_DMA)
    // counted wait: everything but this wave's pieces of the AHEAD - 1 youngest stages has landed
    auto wait_ahead = [&]() {
        if (JOINT && !full) wait_vmcnt<(AHEAD - 1) * (PPW - (REM != 0 ? 1 : 0))>();
        else wait_vmcnt<(AHEAD - 1) * PPW>();
    };
    if constexpr (LD != LD_HT) {
        if (has_tail) { issue_tail(0); slot_i = 1; }
        else issue(0);
#pragma nounroll
        for (int s = 1; s < AHEAD; ++s) issue(s);
    }
    // LD_HT: buffer descriptors of the tile's activation / weight rows, this lane's constant byte offsets into them for its two
    // pieces (8 rows x 128 B each) of every half-tile, and the prologue DMAs.  Slots: HA0 0-1, HA1 2-3, HB0 4-5, HB1 6-7.
    int hvo[4][2] = {};  // [HA0, HA1, HB0, HB1][piece]: clamped row * pitch + (swizzle-inverse) chunk
    const int k_last = K - 128;  // K tiles past the end are fetched from the last one (never consumed; keeps the vmcnt a constant)
    auto issue_ht = [&](auto typec, int slot, int kt) {
        constexpr int ty = decltype(typec)::value;
        int k0 = kt * 128;
        k0 = k0 < k_last ? k0 : k_last;
#pragma unroll
        for (int u = 0; u < 2; ++u)
            SDNQ_DMA16(ty < 2 ? rsA : rsB, lds + slot * HT_BYTES + (2 * wave + u) * 1024, hvo[ty][u], k0);
    };
    if constexpr (LD == LD_HT) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int rho = (2 * wave + u) * 8 + (lane >> 3);  // row inside the half-tile
            const int ck = ((lane & 7) ^ ((rho >> 1) & hk.swz)) << 4;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                int64_t ra = (rho >> 6) * 128 + h * 64 + (rho & 63);  // tile row of this half-tile row
                if (m0 + ra >= hk.M) ra = hk.M - 1 - m0;
                hvo[h][u] = (int)(ra * hk.lda) + ck;
                int64_t rb = (rho >> 5) * 64 + h * 32 + (rho & 31);
                if (rb >= tv.n_lim) rb = tv.n_lim - 1;
                hvo[2 + h][u] = (int)(rb * hk.ldb) + ck;
            }
        }
        issue_ht(std::integral_constant<int, 0>{}, 0, 0);
        issue_ht(std::integral_constant<int, 2>{}, 4, 0);
        issue_ht(std::integral_constant<int, 3>{}, 6, 0);
        issue_ht(std::integral_constant<int, 1>{}, 2, 0);
        issue_ht(std::integral_constant<int, 0>{}, 1, 1);
        issue_ht(std::integral_constant<int, 2>{}, 5, 1);
    }
    TRACE(1);
#if defined(SDNQ_PRELOAD_GEMM) && !defined(SDNQ_NO_KERNARG_BATCH)
    asm volatile("" ::"s"(p_.sb), "s"(p_.bias), "s"(p_.out), "s"(p_.ldc), "s"(p_.bias_dtype), "s"(p_.seg_n), "s"(p_.out_hw), "s"(p_.sa));
#endif
    if (hk.grouped) {
        tv.sb = un.sb + un_d;
        tv.bias = un.bias;
        tv.bias0 = un_d;
        tv.out_ld = un.n_seg;
        tv.out = (uint8_t*)p.out + (hk.M * un.n_start + un.n_loc + un_d) * OUT_B;
    } else {
        tv.sb = p.sb + n0;
        tv.bias = p.bias;
        tv.bias0 = n0;
        tv.out_ld = p.ldc;
        tv.out = (uint8_t*)p.out + n0 * OUT_B;
    }

    // per-output-channel epilogue vectors -> LDS once per workgroup (after the DMA prologue so its load latency hides
    // under it; visible after the first barrier of the K loop)
    for (int i = tid; i < BN; i += NT) {
        const int64_t li = i < tv.n_lim ? i : tv.n_lim - 1;  // channel inside the tile, clamped to the last valid one
        const int64_t gn = n0 + li;
        if constexpr (!is_float_mm<MM> || is_w8a16<MM>) s_sb[i] = tv.sb[li];
        if constexpr (is_w8a16<MM>) s_zp[i] = p.zp ? p.zp[gn] : -128.0f * tv.sb[li];  // additive constant of the row's dequantization
        if constexpr (EPI == EPI_BIAS1D || is_lr<EPI>) s_bias[i] = tv.bias ? ldf_rt(tv.bias, tv.bias0 + li, p.bias_dtype) : 0.0f;
        if constexpr (is_lr<EPI>) { s_zp[i] = p.zp ? p.zp[gn] : 0.0f; s_wcs[i] = p.wcs ? p.wcs[gn] : 0.0f; }
    }

    const int frow = lane & (MS - 1), fgrp = lane / MS;  // row of the MFMA tile this lane feeds, and its 16-byte K chunk
    // fused dequantize GEMM: scale and additive constant of the TN weight rows this lane converts (one row per 32-channel block)
    // They come through the LDS vectors filled above, NOT straight from global memory: the first use of a global load inside
    // the K loop makes the compiler put an s_waitcnt vmcnt(0) there, which drains the LDS-DMA ring every stage (measured: the
    // whole fused step 23.3 ms instead of ...: the K loop ran one HBM round trip per stage).
    WRow wrow[TN];
    u32 wflip = 0;
    if constexpr (is_w8a16<MM>) {
        wflip = p.zp ? 0u : 0x80808080u;  // (signedness is the kernel's MM; kept for the timing-lab builds)
        __syncthreads();  // s_sb / s_zp written by other threads (also waits for the prologue DMAs: once, before the loop)
#pragma unroll
        for (int i = 0; i < TN; ++i) {
            wrow[i].s = s_sb[wn * WN + i * MS + frow];
            wrow[i].c = s_zp[wn * WN + i * MS + frow];
        }
    }
    int slot_c = 0;  // ring slot being consumed
    auto compute = [&]() {
        const uint8_t* sA = lds + slot_c * STAGE_BYTES;
        const uint8_t* sB = sA + BM * BK;
        slot_c = (slot_c + 1 == NS) ? 0 : slot_c + 1;
        constexpr int KS = BK / MT::KB;
        static_assert(KS >= 1, "stage row shorter than one MFMA K step");
        // all fragment reads of the stage are issued before the first MFMA, so LDS latency overlaps the matrix pipe
        typename FragOps<MM>::frag_t fa[KS][TM];
        typename FragOps<MM>::fragb_t fb[KS][TN];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
            for (int j = 0; j < TM; ++j) fa[ks][j] = FragOps<MM>::template load<BK>(sA, wm * WM + j * MS + frow, ks, fgrp, hk.swz);
#pragma unroll
            for (int i = 0; i < TN; ++i) fb[ks][i] = FragOps<MM>::template loadb<BKW>(sB, wn * WN + i * MS + frow, ks, fgrp, hk.swz);
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int i = 0; i < TN; ++i)
                { const auto wfrag = FragOps<MM>::prep(fb[ks][i], wrow[i], wflip);
#pragma unroll
                for (int j = 0; j < TM; ++j) FragOps<MM>::mma(acc[i][j], wfrag, fa[ks][j]); }
    };

    if constexpr (LD == LD_PIPE) {
        typedef typename FragOps<MM>::frag_t frag_t;
        constexpr int KS = BK / MT::KB;
        constexpr int U = (KS & 1) ? 2 : 1;  // stages per loop trip, so that the fragment-set parity is static
        frag_t fa[2][TM];
        typename FragOps<MM>::fragb_t fb[2][TN];
        auto load_set = [&](int slot, auto ksc, auto setc) {
            constexpr int ks = decltype(ksc)::value, st = decltype(setc)::value;
            const uint8_t* sA = lds + slot * STAGE_BYTES;
            const uint8_t* sB = sA + BM * BK;
#pragma unroll
            for (int j = 0; j < TM; ++j) fa[st][j] = FragOps<MM>::template load<BK>(sA, wm * WM + j * MS + frow, ks, fgrp, hk.swz);
#pragma unroll
            for (int i = 0; i < TN; ++i) fb[st][i] = FragOps<MM>::template loadb<BKW>(sB, wn * WN + i * MS + frow, ks, fgrp, hk.swz);
        };
        auto mma_set = [&](auto setc) {
            constexpr int st = decltype(setc)::value;
#pragma unroll
            for (int i = 0; i < TN; ++i)
                { const auto wfrag = FragOps<MM>::prep(fb[st][i], wrow[i], wflip);
#pragma unroll
                for (int j = 0; j < TM; ++j) FragOps<MM>::mma(acc[i][j], wfrag, fa[st][j]); }
        };
        // stage 0 landed for every wave -> top up the ring (slot NS-1) -> first fragment set
        wait_ahead();
        __builtin_amdgcn_s_barrier();
        issue(AHEAD);
        load_set(0, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
#pragma nounroll
        for (int kt = 0; kt < nk; kt += U) {
            static_for_up<U * KS>([&](auto subc) {
                constexpr int sub = decltype(subc)::value, u = sub / KS, ks = sub % KS, cur = sub & 1;
                if constexpr (ks + 1 < KS) {
                    load_set(slot_c, std::integral_constant<int, ks + 1>{}, std::integral_constant<int, cur ^ 1>{});
                } else {
                    // end of stage kt+u: stage kt+u+1 must have landed (counted vmcnt: the NS-2 younger stages stay in
                    // flight), this wave's own LDS reads of the finished stage must have completed (its slot is about
                    // to be refilled), one barrier, refill, and the first fragments of the next stage start flowing
                    // while the last MFMAs of this one run.
                    wait_ahead();
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    issue(kt + u + 1 + AHEAD);
                    slot_c = (slot_c + 1 == NS) ? 0 : slot_c + 1;
                    load_set(slot_c, std::integral_constant<int, 0>{}, std::integral_constant<int, cur ^ 1>{});
                }
                // (two stages per trip and an odd stage count: the second half of the last trip sees a stage past the end of K, which
                //  now holds re-fetched data instead of zeros -- skip its MFMAs; wave-uniform)
                if (U == 1 || kt + u < nk) mma_set(std::integral_constant<int, cur>{});
            });
        }
    } else if constexpr (LD == LD_PP) {
        static_assert(NW == 8, "ping-pong schedule: two halves of four waves");
        typedef typename FragOps<MM>::frag_t frag_t;
        constexpr int KS = BK / MT::KB;
        frag_t fa[KS][TM];
        typename FragOps<MM>::fragb_t fb[KS][TN];
        const int half = __builtin_amdgcn_readfirstlane(wave >> 2);
        // Time is cut into slots by workgroup-wide barriers.  Half 0 runs LOAD(j) in slot 2j and MFMA(j) in slot 2j+1, half 1
        // LOAD(j) in slot 2j+1 and MFMA(j) in slot 2j+2.
        //   RAW: stage j is first read in slot 2j; every wave waits for ITS pieces of stage j (counted vmcnt) before the barrier
        //        that ends slot 2j-1 (half 0: end of MFMA(j-1); half 1: end of LOAD(j-1)).
        //   WAR: the DMA for stage j+AHEAD overwrites the ring slot of stage j-1; it is issued in LOAD(j) (slots 2j / 2j+1), after
        //        the barrier that ends slot 2j-1, by which time both halves have finished reading stage j-1 (lgkmcnt(0) before
        //        the barrier that ends a LOAD phase).
        wait_ahead();  // own pieces of stage 0 (stages 1..AHEAD-1 stay in flight)
        __builtin_amdgcn_s_barrier();
        if (half == 1) __builtin_amdgcn_s_barrier();  // the stagger: half 1 sits out slot 0
#pragma nounroll
        for (int kt = 0; kt < nk; ++kt) {
            // ---- LOAD(kt)
            issue(kt + AHEAD);
            {
                const uint8_t* sA = lds + slot_c * STAGE_BYTES;
                const uint8_t* sB = sA + BM * BK;
                slot_c = (slot_c + 1 == NS) ? 0 : slot_c + 1;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
                    for (int i = 0; i < TN; ++i) fb[ks][i] = FragOps<MM>::template loadb<BKW>(sB, wn * WN + i * MS + frow, ks, fgrp, hk.swz);
#pragma unroll
                    for (int j = 0; j < TM; ++j) fa[ks][j] = FragOps<MM>::template load<BK>(sA, wm * WM + j * MS + frow, ks, fgrp, hk.swz);
                }
            }
            if (half == 1) wait_ahead();  // own pieces of stage kt+1, read by half 0 in the next slot
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            // ---- MFMA(kt)
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int i = 0; i < TN; ++i)
                    { const auto wfrag = FragOps<MM>::prep(fb[ks][i], wrow[i], wflip);
#pragma unroll
                for (int j = 0; j < TM; ++j) FragOps<MM>::mma(acc[i][j], wfrag, fa[ks][j]); }
            __builtin_amdgcn_s_setprio(0);
            if (half == 0) wait_ahead();  // own pieces of stage kt+1, read by this half right after the barrier
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
        if (half == 0) __builtin_amdgcn_s_barrier();  // matches half 1's extra barrier at the start
    } else if constexpr (LD == LD_8P) {
        static_assert(NW == 8 && !JOINT && (TM % 2) == 0 && NS >= 3 && A_PIECES >= 1 && B_PIECES >= 1, "fine ping-pong: 8 waves, even activation blocks");
        typedef typename FragOps<MM>::frag_t frag_t;
        constexpr int KS = BK / MT::KB, TH = TM / 2;
        frag_t fa0[KS][TH], fa1[KS][TH];
        typename FragOps<MM>::fragb_t fb[KS][TN];
        const int grp = __builtin_amdgcn_readfirstlane(wave >> 2);
        wait_ahead();  // own pieces of stage 0 (stages 1..AHEAD-1 stay in flight)
        __builtin_amdgcn_s_barrier();
        if (grp == 1) __builtin_amdgcn_s_barrier();  // the stagger: group 1 sits out the first slot
        int slot_r = AHEAD;  // ring slot stage kt + AHEAD goes to (the one stage kt - 1 occupied)
#pragma nounroll
        for (int kt = 0; kt < nk; ++kt) {
            const uint8_t* sA = lds + slot_c * STAGE_BYTES;
            const uint8_t* sB = sA + BM * BK;
            slot_c = (slot_c + 1 == NS) ? 0 : slot_c + 1;
            // ---- phase 0: weight fragments + first half of the activation blocks
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
                for (int i = 0; i < TN; ++i) fb[ks][i] = FragOps<MM>::template loadb<BKW>(sB, wn * WN + i * MS + frow, ks, fgrp, hk.swz);
#pragma unroll
                for (int j = 0; j < TH; ++j) fa0[ks][j] = FragOps<MM>::template load<BK>(sA, wm * WM + j * MS + frow, ks, fgrp, hk.swz);
            }
            __builtin_amdgcn_sched_barrier(0);
            issue_range(kt + AHEAD, slot_r, std::integral_constant<int, A_PIECES>{}, std::integral_constant<int, PPW>{});  // weight rows
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int i = 0; i < TN; ++i)
                    { const auto wfrag = FragOps<MM>::prep(fb[ks][i], wrow[i], wflip);
#pragma unroll
                for (int j = 0; j < TH; ++j) FragOps<MM>::mma(acc[i][j], wfrag, fa0[ks][j]); }
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            // ---- phase 1: second half of the activation blocks
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int j = 0; j < TH; ++j) fa1[ks][j] = FragOps<MM>::template load<BK>(sA, wm * WM + (TH + j) * MS + frow, ks, fgrp, hk.swz);
            __builtin_amdgcn_sched_barrier(0);
            issue_range(kt + AHEAD, slot_r, std::integral_constant<int, 0>{}, std::integral_constant<int, A_PIECES>{});  // activation rows
            slot_r = (slot_r + 1 == NS) ? 0 : slot_r + 1;
            wait_ahead();  // own pieces of stage kt + 1
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int i = 0; i < TN; ++i)
                    { const auto wfrag = FragOps<MM>::prep(fb[ks][i], wrow[i], wflip);
#pragma unroll
                for (int j = 0; j < TH; ++j) FragOps<MM>::mma(acc[i][TH + j], wfrag, fa1[ks][j]); }
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
        if (grp == 0) __builtin_amdgcn_s_barrier();  // matches group 1's extra barrier at the start
    } else if constexpr (LD == LD_HT) {
        static_assert(BM == 256 && BN == 256 && WM == 128 && WN == 64 && BK == 128 && MS == 32 && !is_w8a16<MM>, "half-tile ring: the 256x256 tile of 8 waves");
        typedef typename FragOps<MM>::frag_t frag_t;
        typedef typename FragOps<MM>::fragb_t fragb_t;
        constexpr int KS = BK / MT::KB;
        frag_t fa0[KS][2], fa1[KS][2];
        fragb_t fb0[KS], fb1[KS];
        const int grp = __builtin_amdgcn_readfirstlane(wave >> 2);
        const int nkt = K / 128;
        // this lane's rows inside the activation / weight half-tiles; the fragment addresses below are (row, k sub-step) constants of the
        // lane + immediates (slot, second row block): no address arithmetic inside the loop
        const uint8_t* ldsA = lds + (wm * 64 + frow) * 128;                  // slot 0 (HA0, parity 0)
        const uint8_t* ldsB = lds + 4 * HT_BYTES + (wn * 32 + frow) * 128;   // slot 4 (HB0, parity 0)
        const int rswA = ((wm * 64 + frow) >> 1) & hk.swz, rswB = ((wn * 32 + frow) >> 1) & hk.swz;
        // chunk offsets (swizzled) of this lane's 16-byte pieces: [k sub-step][piece of the fragment]
        constexpr int CPK = MT::KB / 16, NPC = (MM == SDNQ_MM_FP8) ? 2 : 1;  // chunks per sub-step; 16-byte reads per lane and fragment
        int coA[KS][NPC], coB[KS][NPC];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int c = 0; c < NPC; ++c) {
                const int ch = ks * CPK + fgrp * NPC + c;
                coA[ks][c] = (ch ^ rswA) << 4;
                coB[ks][c] = (ch ^ rswB) << 4;
            }
        auto ldA = [&](int ks, int imm) -> frag_t {
            if constexpr (NPC == 1) return *(const v4i*)(ldsA + coA[ks][0] + imm);
            else {
                const v4i lo = *(const v4i*)(ldsA + coA[ks][0] + imm), hi = *(const v4i*)(ldsA + coA[ks][1] + imm);
                return (v8i){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
        };
        auto ldB = [&](int ks, int imm) -> fragb_t {
            if constexpr (NPC == 1) return *(const v4i*)(ldsB + coB[ks][0] + imm);
            else {
                const v4i lo = *(const v4i*)(ldsB + coB[ks][0] + imm), hi = *(const v4i*)(ldsB + coB[ks][1] + imm);
                return (v8i){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
        };
#ifdef SDNQ_TRACE2  // development build: per-segment cycle totals of waves 0 and 4 (tools/micro/gemm_lab.hip prints them)
        unsigned tacc[4][5] = {};
        unsigned ts0 = (unsigned)__builtin_amdgcn_s_memtime(), ts1 = 0, ts2 = 0, ts3 = 0, ts4 = 0;
#define TS(v) v = (unsigned)__builtin_amdgcn_s_memtime()
#else
#define TS(v) do { } while (0)
#endif
        auto phase_sync = [&](bool reads) {
            // end of the read / issue section: counted vmcnt (four half-tiles stay in flight), barrier, then this phase's own reads
            __builtin_amdgcn_sched_barrier(0);
            TS(ts1);
            wait_vmcnt<8>();
            TS(ts2);
            __builtin_amdgcn_s_barrier();
            if (reads) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            TS(ts3);
            __builtin_amdgcn_s_setprio(1);
        };
        auto phase_end = [&](auto qc) {
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            TS(ts4);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
#ifdef SDNQ_TRACE2
            constexpr int q = decltype(qc)::value;
            const unsigned ts5 = (unsigned)__builtin_amdgcn_s_memtime();
            tacc[q][0] += ts1 - ts0; tacc[q][1] += ts2 - ts1; tacc[q][2] += ts3 - ts2; tacc[q][3] += ts4 - ts3; tacc[q][4] += ts5 - ts4;
            ts0 = ts5;
            __builtin_amdgcn_sched_barrier(0);
#endif
        };
#ifdef SDNQ_LAB_LUT4
        // TIMING-ONLY lab build (wrong results; tools/lut4_lab.sh): what a fused 4-bit loader would add to this K loop.  Every weight
        // fragment is treated as 8 bytes of packed codes + a 16-entry byte table of its (row, group) and expanded to the 16 int8 codes
        // of the MFMA operand with the cheapest sequence found (nibble split, two v_perm per 4 codes, bit-3 blend; the even / odd
        // interleave is assumed away by a k-permuted activation operand): 34 vector-ALU instructions per fragment, placed in the
        // read / issue section of the phase (where the partner wave's MFMA section can hide them).  No table build, no table traffic.
        v4i lut_t = *(const v4i*)(lds + VEC_OFF + (lane & 15) * 16);
        auto lut_expand = [&](fragb_t& f) {
            if constexpr (NPC == 1) {
                v4i o;
#pragma unroll
                for (int d = 0; d < 2; ++d) {
                    const u32 x = (u32)f[d];
                    const u32 cl = x & 0x0f0f0f0fu, chh = (x >> 4) & 0x0f0f0f0fu;
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const u32 c = h ? chh : cl;
                        const u32 sel = c & 0x07070707u;
                        const u32 p0 = __builtin_amdgcn_perm((u32)lut_t[1], (u32)lut_t[0], sel);
                        const u32 p1 = __builtin_amdgcn_perm((u32)lut_t[3], (u32)lut_t[2], sel);
                        const u32 m = ((c >> 3) & 0x01010101u) * 255u;
                        o[2 * d + h] = (int)((p1 & m) | (p0 & ~m));
                    }
                }
                f = o;
            }
        };
#define LUT_EXPAND(fb) do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); _Pragma("unroll") for (int ks = 0; ks < KS; ++ks) lut_expand(fb[ks]); } while (0)
#else
#define LUT_EXPAND(fb) do { } while (0)
#endif
        wait_vmcnt<8>();  // HA0(0), HB0(0) have landed: everything phase 0 reads
        __builtin_amdgcn_s_barrier();
        if (grp == 1) __builtin_amdgcn_s_barrier();  // the stagger: group 1 sits out the first slot
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)  // A0 of K tile 0 (every later one is read in the phase 3 before its K tile)
#pragma unroll
            for (int j = 0; j < 2; ++j) fa0[ks][j] = ldA(ks, j * 4096);
        // one K tile; PAR = t & 1 selects the ring slots at compile time (slot offsets become instruction immediates)
        auto ktile = [&](auto parc, int t) {
            constexpr int PAR = decltype(parc)::value;
            // ---- phase 0: B0 (A0 was read in the previous phase 3) -> acc[0][0..1]; refill HB1 for K tile t + 1
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) fb0[ks] = ldB(ks, PAR * HT_BYTES);
            LUT_EXPAND(fb0);
            __builtin_amdgcn_sched_barrier(0);
            issue_ht(std::integral_constant<int, 3>{}, 6 + (PAR ^ 1), t + 1);
            phase_sync(true);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                { const auto wfrag = FragOps<MM>::prep(fb0[ks], wrow[0], wflip);
#pragma unroll
                for (int j = 0; j < 2; ++j) FragOps<MM>::mma(acc[0][j], wfrag, fa0[ks][j]); }
            phase_end(std::integral_constant<int, 0>{});
            // ---- phase 1: B1 -> acc[1][0..1]; refill HA1 for K tile t + 1
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) fb1[ks] = ldB(ks, (2 + PAR) * HT_BYTES);
            LUT_EXPAND(fb1);
            __builtin_amdgcn_sched_barrier(0);
            issue_ht(std::integral_constant<int, 1>{}, 2 + (PAR ^ 1), t + 1);
            phase_sync(true);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                { const auto wfrag = FragOps<MM>::prep(fb1[ks], wrow[1], wflip);
#pragma unroll
                for (int j = 0; j < 2; ++j) FragOps<MM>::mma(acc[1][j], wfrag, fa0[ks][j]); }
            phase_end(std::integral_constant<int, 1>{});
            // ---- phase 2: A1 -> acc[1][2..3]; refill HA0 for K tile t + 2
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int j = 0; j < 2; ++j) fa1[ks][j] = ldA(ks, (2 + PAR) * HT_BYTES + j * 4096);
            __builtin_amdgcn_sched_barrier(0);
            issue_ht(std::integral_constant<int, 0>{}, PAR, t + 2);
            phase_sync(true);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                { const auto wfrag = FragOps<MM>::prep(fb1[ks], wrow[1], wflip);
#pragma unroll
                for (int j = 0; j < 2; ++j) FragOps<MM>::mma(acc[1][2 + j], wfrag, fa1[ks][j]); }
            phase_end(std::integral_constant<int, 2>{});
            // ---- phase 3: A0 of K tile t + 1 (its registers are free since phase 1; landed: it is the fifth-youngest half-tile) ->
            //      acc[0][2..3]; refill HB0 for K tile t + 2.  Reads per phase: 4 / 4 / 8 / 8 instead of 12 / 4 / 8 / 0.
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int j = 0; j < 2; ++j) fa0[ks][j] = ldA(ks, (PAR ^ 1) * HT_BYTES + j * 4096);
            __builtin_amdgcn_sched_barrier(0);
            issue_ht(std::integral_constant<int, 2>{}, 4 + PAR, t + 2);
            phase_sync(true);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                { const auto wfrag = FragOps<MM>::prep(fb0[ks], wrow[0], wflip);
#pragma unroll
                for (int j = 0; j < 2; ++j) FragOps<MM>::mma(acc[0][2 + j], wfrag, fa1[ks][j]); }
            phase_end(std::integral_constant<int, 3>{});
        };
#pragma nounroll
        for (int t = 0; t < nkt; t += 2) {
            ktile(std::integral_constant<int, 0>{}, t);
            if (t + 1 < nkt) ktile(std::integral_constant<int, 1>{}, t + 1);
        }
        if (grp == 0) __builtin_amdgcn_s_barrier();  // matches group 1's extra barrier at the start
#ifdef SDNQ_TRACE2
        if (blockIdx.x == 300 && (tid == 0 || tid == 256)) {
            for (int q = 0; q < 4; ++q)
                for (int e = 0; e < 5; ++e) g_trace2[(tid >> 8) * 20 + q * 5 + e] = tacc[q][e];
        }
#endif
#undef TS
#undef LUT_EXPAND
    } else if constexpr (LD == LD_OV) {
        static_assert(MS == 32, "overlapped ring: 32x32 MFMA tiles");
        typedef typename FragOps<MM>::frag_t frag_t;
        typedef typename FragOps<MM>::fragb_t fragb_t;
        constexpr int KS = BK / MT::KB, NM = KS * TN * TM;
        frag_t fa[2][KS][TM];
        fragb_t fb[2][KS][TN];
        // A 32x32 wave tile has ONE accumulator: its MFMAs form a dependent chain, and anything issued between two MFMAs on the same
        // accumulator costs the forwarding path (+43 cycles per gap, MI355X_MICROARCH.md).  Odd K sub-steps accumulate into a second
        // register set, summed once after the loop (int32: exact, so still bit-identical; fp32: one more addition order).
        constexpr bool SPLIT_ACC = (TM * TN == 1) && (KS % 2 == 0) && (MM == SDNQ_MM_I8);  // (integer sums only: a second fp32 accumulator would make the result depend on the tile)
        typename MT::acc_t acc_odd[1];
        if constexpr (SPLIT_ACC) MT::zero(acc_odd[0]);
        auto load_stage = [&](int slot, auto setc) {
            constexpr int st = decltype(setc)::value;
            const uint8_t* sA = lds + slot * STAGE_BYTES;
            const uint8_t* sB = sA + BM * BK;
#ifdef SDNQ_ABL_NOREAD
            if (slot >= 0) return;
#endif
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
                for (int i = 0; i < TN; ++i) fb[st][ks][i] = FragOps<MM>::template loadb<BKW>(sB, wn * WN + i * MS + frow, ks, fgrp, hk.swz);
#pragma unroll
                for (int j = 0; j < TM; ++j) fa[st][ks][j] = FragOps<MM>::template load<BK>(sA, wm * WM + j * MS + frow, ks, fgrp, hk.swz);
            }
        };
        // one fragment read of a stage, numbered ks-major: r -> (ks, weight sub-tile i | activation sub-tile j)
        constexpr int NRG = TM + TN, NRT = KS * NRG;
#ifndef SDNQ_OV_NMR_NUM
#define SDNQ_OV_NMR_NUM 3
#endif
        constexpr int NMR = (NM * SDNQ_OV_NMR_NUM) / 4 > 0 ? (NM * SDNQ_OV_NMR_NUM) / 4 : 1;
        auto read_one = [&](const uint8_t* sA, const uint8_t* sB, auto setc, auto rc) {
            constexpr int st = decltype(setc)::value, r = decltype(rc)::value, ks = r / NRG, e = r % NRG;
#ifdef SDNQ_ABL_NOREAD
            if (sA != nullptr) return;
#endif
            if constexpr (e < TN) fb[st][ks][e] = FragOps<MM>::template loadb<BKW>(sB, wn * WN + e * MS + frow, ks, fgrp, hk.swz);
            else fa[st][ks][e - TN] = FragOps<MM>::template load<BK>(sA, wm * WM + (e - TN) * MS + frow, ks, fgrp, hk.swz);
        };
        // the MFMAs of fragment set `st`; dealt out between them: the fragment reads of the NEXT stage (ring slot `rslot`, into the other
        // set) and the DMA pieces of logical stage `js` (into ring slot `slot`).  A wave that issued all its reads in one burst after the
        // barrier sat in the LDS issue queue behind the other seven waves' bursts until nearly all 64-96 KB had been served and only then
        // reached its first MFMA (reads + MFMAs measured as their SUM, tools/micro/build_abl.sh: nodma 64 us vs mmaonly 43.5)
        auto mma_issue = [&](auto setc, int js, int slot, int rslot) {
            constexpr int st = decltype(setc)::value;
            const uint8_t* sA = lds + rslot * STAGE_BYTES;
            const uint8_t* sB = sA + BM * BK;
            static_for_up<NM>([&](auto mc) {
                constexpr int mi = decltype(mc)::value, ks = mi / (TN * TM), i = (mi / TM) % TN, j = mi % TM;
#ifndef SDNQ_ABL_NOMMA
                if constexpr (SPLIT_ACC && (ks & 1)) FragOps<MM>::mma(acc_odd[0], FragOps<MM>::prep(fb[st][ks][i], wrow[i], wflip), fa[st][ks][j]);
                else FragOps<MM>::mma(acc[i][j], FragOps<MM>::prep(fb[st][ks][i], wrow[i], wflip), fa[st][ks][j]);
#endif
                __builtin_amdgcn_sched_barrier(0);
                static_for_up<NRT>([&](auto rc) {
                    constexpr int r = decltype(rc)::value;
                    // reads are dealt over the first NMR MFMAs only: the last ones retire under the stage's remaining MFMAs, so the
                    // lgkmcnt(0) in front of the next barrier does not wait for LDS latency
                    if constexpr (mi < NMR && r >= (mi * NRT) / NMR && r < ((mi + 1) * NRT) / NMR) read_one(sA, sB, std::integral_constant<int, st ^ 1>{}, rc);
                });
                __builtin_amdgcn_sched_barrier(0);
                static_for_up<PPW>([&](auto pc) {
                    constexpr int pi = decltype(pc)::value;
                    constexpr int after = ((pi + 1) * NM) / (PPW + 1) - 1 < 0 ? 0 : ((pi + 1) * NM) / (PPW + 1) - 1;
                    if constexpr (after == mi) {
#ifndef SDNQ_ABL_NODMA
                        issue_range(js, slot, std::integral_constant<int, pi>{}, std::integral_constant<int, pi + 1>{});
#endif
                        __builtin_amdgcn_sched_barrier(0);
                    }
                });
            });
        };
        // prologue: stages 0 .. NS-2 are in flight (above); stage 0 landed -> the last free slot gets stage NS-1, stage 0 -> set 0
        wait_ahead();
        __builtin_amdgcn_s_barrier();
        if (true) TRACE(2);
        issue(AHEAD);
        load_stage(0, std::integral_constant<int, 0>{});
        auto half = [&](auto setc, int kt) {
            constexpr int st = decltype(setc)::value;
            // stage kt + 1 has landed (this wave's pieces; the barrier makes it everybody's), the NS - 2 younger stages stay in flight;
            // this wave's reads of stage kt have retired, so after the barrier the slot of stage kt belongs to the DMA of stage kt + NS
#ifndef SDNQ_ABL_NODMA
            wait_ahead();
#endif
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#ifndef SDNQ_ABL_NOBAR
            __builtin_amdgcn_s_barrier();
#endif
            __builtin_amdgcn_sched_barrier(0);
            const int slot_free = slot_c;
            slot_c = (slot_c + 1 == NS) ? 0 : slot_c + 1;
            mma_issue(setc, kt + NS, slot_free, slot_c);  // (reads past the end of K: a re-fetched stage nobody multiplies)
        };
#pragma nounroll
        for (int kt = 0; kt < nk; kt += 2) {
            half(std::integral_constant<int, 0>{}, kt);
            if (kt + 1 < nk) half(std::integral_constant<int, 1>{}, kt + 1);
        }
        if constexpr (SPLIT_ACC) {
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[0][0][e] += acc_odd[0][e];
        }
    } else if constexpr (LD == LD_OG) {
        // the overlapped ring for wave tiles of several MFMA tiles: fragments are read ONE K sub-step ahead (two register sets of TM + TN
        // fragments instead of two whole stages: a 32x160 wave tile would need 192 fragment registers), everything else as LD_OV.  One loop
        // trip = {counted vmcnt, lgkmcnt(0), barrier} then the LAST sub-step of stage kt (its reads fetch sub-step 0 of stage kt + 1, which
        // the barrier just published) and sub-steps 0 .. KS-2 of stage kt + 1; the DMA pieces of stage kt + NS (ring slot of stage kt, free
        // since every wave's reads of it retired in front of the barrier) are dealt out over the trip's MFMAs.
        static_assert(!is_w8a16<MM> && MS == 32 && (BK / MT::KB) % 2 == 0 && (BK / MT::KB) >= 2, "group-ahead ring: an even number of K sub-steps per stage");
        typedef typename FragOps<MM>::frag_t frag_t;
        typedef typename FragOps<MM>::fragb_t fragb_t;
        constexpr int KS = BK / MT::KB, NMG = TN * TM, NRG = TM + TN, NMT = KS * NMG;
        frag_t fa[2][TM];
        fragb_t fb[2][TN];
        auto read_one = [&](const uint8_t* sA, const uint8_t* sB, auto setc, auto ksc, auto ec) {
            constexpr int st = decltype(setc)::value, ks = decltype(ksc)::value, e = decltype(ec)::value;
#ifdef SDNQ_ABL_NOREAD
            if (sA != nullptr) return;
#endif
            if constexpr (e < TN) fb[st][e] = FragOps<MM>::template loadb<BKW>(sB, wn * WN + e * MS + frow, ks, fgrp, hk.swz);
            else fa[st][e - TN] = FragOps<MM>::template load<BK>(sA, wm * WM + (e - TN) * MS + frow, ks, fgrp, hk.swz);
        };
        // MFMA group of set SC; between its MFMAs (FEED): the reads of sub-step KSN (ring slot rslot) into the other set and this group's
        // share of the DMA pieces; G = position of the group in the trip (0 .. KS-1), which selects the pieces
        auto group = [&](auto scc, auto ksnc, auto gc, auto feedc, int js, int dslot, int rslot) {
            constexpr int SC = decltype(scc)::value, KSN = decltype(ksnc)::value, G = decltype(gc)::value;
            constexpr bool FEED = decltype(feedc)::value;
            const uint8_t* sA = lds + rslot * STAGE_BYTES;
            const uint8_t* sB = sA + BM * BK;
            static_for_up<NMG>([&](auto mc) {
                constexpr int mi = decltype(mc)::value, i = mi / TM, j = mi % TM, gmi = G * NMG + mi;
#ifndef SDNQ_ABL_NOMMA
                FragOps<MM>::mma(acc[i][j], FragOps<MM>::prep(fb[SC][i], wrow[i], wflip), fa[SC][j]);
#endif
                if constexpr (FEED) {
                    __builtin_amdgcn_sched_barrier(0);
                    static_for_up<NRG>([&](auto ec) {
                        constexpr int e = decltype(ec)::value;
                        // a fragment register of the other set is free: its last MFMA was in the previous group
                        if constexpr (e >= (mi * NRG) / NMG && e < ((mi + 1) * NRG) / NMG) read_one(sA, sB, std::integral_constant<int, SC ^ 1>{}, ksnc, ec);
                    });
                    __builtin_amdgcn_sched_barrier(0);
                    static_for_up<PPW>([&](auto pc) {
                        constexpr int pi = decltype(pc)::value;
                        constexpr int after = ((pi + 1) * NMT) / (PPW + 1) - 1 < 0 ? 0 : ((pi + 1) * NMT) / (PPW + 1) - 1;
                        if constexpr (after == gmi) {
#ifndef SDNQ_ABL_NODMA
                            issue_range(js, dslot, std::integral_constant<int, pi>{}, std::integral_constant<int, pi + 1>{});
#endif
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    });
                }
            });
        };
        // trip kt = -1 has no last sub-step of a stage -1 to multiply: its first group runs on ZERO fragments (adds nothing) instead of
        // sitting behind a branch per MFMA
#pragma unroll
        for (int i = 0; i < TN; ++i) fb[1][i] = fragb_t{};
#pragma unroll
        for (int j = 0; j < TM; ++j) fa[1][j] = frag_t{};
        int slot_d = AHEAD;  // ring slot of stage kt (trip kt = -1: the one slot the prologue left empty)
        constexpr std::integral_constant<bool, true> feed{};
#pragma nounroll
        for (int kt = -1; kt < nk - 1; ++kt) {
#ifndef SDNQ_ABL_NODMA
            wait_ahead();  // stage kt + 1 has landed (this wave's pieces); NS - 2 younger stages stay in flight
#endif
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's reads of stage kt have retired
            __builtin_amdgcn_sched_barrier(0);
#ifndef SDNQ_ABL_NOBAR
            __builtin_amdgcn_s_barrier();
#endif
            __builtin_amdgcn_sched_barrier(0);
            if (kt < 0) TRACE(2);
            const int rslot = slot_c;  // stage kt + 1
            group(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, feed, kt + NS, slot_d, rslot);
            static_for_up<KS - 1>([&](auto kc) {
                constexpr int ks = decltype(kc)::value;
                group(std::integral_constant<int, ks & 1>{}, std::integral_constant<int, ks + 1>{}, std::integral_constant<int, ks + 1>{}, feed, kt + NS, slot_d, rslot);
            });
            slot_d = slot_c;
            slot_c = (slot_c + 1 == NS) ? 0 : slot_c + 1;
        }
        // the last sub-step of the last stage: nothing left to read or fetch
        group(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, std::integral_constant<bool, false>{}, 0, 0, 0);
    } else if constexpr (LD == LD_DMA) {
        // K loop: wait only for stage kt with a COUNTED vmcnt (the AHEAD-1 younger stages stay in flight across the
        // single raw barrier), refill the ring slot stage kt-1 occupied (an unconsumed re-fetch past the end of K), run the MFMAs.
#pragma nounroll
        for (int kt = 0; kt < nk; ++kt) {
            wait_ahead();
            __builtin_amdgcn_s_barrier();
            if (kt == 0) TRACE(2);
            issue(kt + AHEAD);
            compute();
        }
    }
    TRACE(3);
    wait_vmcnt<0>();  // the trailing (unconsumed) DMAs target ring slots the epilogue is about to reuse
    __syncthreads();
    TRACE(4);

    // ---- low-rank (SVD) term on the matrix cores: lr[n][m] = sum_r up[n][r] * t[m][r] -----------------------------
    // same 32x32 MFMA shape as the main product, so every lane gets the low-rank value of exactly the outputs it owns;
    // operands are 16-byte rows of t [M][R] / svd_up [N][R] straight from global (R/16 MFMAs per sub-tile).
    // (computed per 32x32 sub-tile right before that sub-tile is staged: only 16 extra accumulator registers are live at a
    // time -- holding all TN x TM low-rank tiles next to the main accumulators spilled 104 VGPRs in the 256x256 kernel)
    bool lr_mfma = false, lr_lds = false;
    if constexpr (is_lr<EPI>) {
        lr_mfma = p.lr_t != nullptr && (p.rank % 16) == 0 && p.bias_dtype != SDNQ_F32;
        // rank 32 (the default): the tile's rows of t and svd_up -- 64 bytes each -- are fetched ONCE into the (now idle) ring by
        // LDS-DMA and the sub-tile fragments come from LDS.  Fetching every sub-tile's fragments from global right before its MFMAs
        // exposed one L2 round trip per sub-tile, 8 per wave: ~26 us per FLUX GEMM (round 2 profile: 47 ms of GEMMs vs 36 ms
        // without the low-rank term).  16 rows per DMA; 16-byte chunk ^= (row >> 2) & 3 keeps the fragment reads conflict-free.
        lr_lds = lr_mfma && p.rank == 32;
        if (lr_lds) {
            for (int pc = wave; pc < (BM + BN) / 16; pc += NW) {
                const bool is_t = pc < BM / 16;
                const int row = (is_t ? pc : pc - BM / 16) * 16 + (lane >> 2);
                const int c = (lane & 3) ^ ((row >> 2) & 3);
                int64_t g = (is_t ? m0 : n0) + row;
                const int64_t lim = is_t ? hk.M : hk.N;
                if (g >= lim) g = lim - 1;
                const uint8_t* src = (const uint8_t*)(is_t ? p.lr_t : p.lr_up) + g * 64 + c * 16;
                __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(lds + pc * 1024), 16, 0, 0);
            }
            wait_vmcnt<0>();
            __syncthreads();
        }
    }

    // ---- epilogue ---------------------------------------------------------------------------------
    constexpr bool STAGED_EPILOGUE = is_lr<EPI>;
    if constexpr (STAGED_EPILOGUE) {
    // ======== LDS-staged epilogue with a compact runtime loop: low-rank / zero-point terms, and every 256-row tile ====
    // The low-rank arithmetic is long; fully unrolled over the accumulator registers (as the epilogue below is) it becomes
    // ~8k instructions of straight-line code that every wave executes once at instruction-fetch speed (measured: 2.5x slower
    // GEMM).  The 256-row tiles (128 accumulator registers per lane) also measured 2 % faster this way.  So the raw
    // accumulators (and the low-rank tile) are staged as 32-bit values and a small loop finishes them.
    if constexpr (is_lr<EPI> && OUT_T != SDNQ_F32) {
        if (EPI == EPI_LRFAST || (lr_lds && p.zp == nullptr && p.a_zp == nullptr && p.bias_dtype == OUT_T)) {
            // ======== SVD-only layers (no zero-point terms; FLUX int8 + SVD): finish in the MFMA register layout ========
            // Every lane owns outputs m = wm*WM + j*32 + (lane & 31), n = wn*WN + i*32 + (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5).  Per
            // sub-tile: two MFMAs on the staged factor tiles give the low-rank values of exactly those outputs; bias2d = cast_svd(bias +
            // low-rank); out = cast(fma(f32(acc) * sa, sb, bias2d)) (linear_int8.py:57-62, kernel_wrappers.py:132-136).  Only the FINAL
            // 16-bit values go through LDS (chunks of up to 128 rows) to leave as 16-byte row pieces.  The general staged loop below
            // (raw accumulators + a bias2d plane through LDS, chunk by chunk) cost 45 us per 256x256 tile: as four unrolled chunk
            // copies it was 15.7 K lines of ISA executed once at instruction-fetch speed, as a runtime loop it spilled.
            constexpr int CH2 = CHR, ECH2 = ECHR;
            constexpr int O_ROW = BN * OUT_B + 16;
            uint8_t* ostage = lds + LR_BYTES;
            static_assert(LR_BYTES + CH2 * O_ROW <= EPI_BYTES, "register-layout low-rank epilogue fits the staged epilogue's LDS budget");
            const bool hb = p.bias != nullptr;
            // the svd dtype is the OUTPUT dtype here (the launcher sends other combinations to the general path): a compile-time constant.
            // As a run-time flag every bias2d rounding sat behind two scalar branches (s_and / s_cbranch around the f16 and around the
            // bf16 convert, per ELEMENT: ~6 scalar + 4 vector instructions where 2 vector ones do) -- the epilogue of a 256x256 tile
            // was 24 K cycles against 13 K of the plain one (round 3 trace)
            constexpr bool is_bf = (OUT_T == SDNQ_BF16);
            TRACE(5);
#pragma nounroll
            for (int ch = 0; ch < ECH2; ++ch) {
                if (ch > 0) __syncthreads();  // previous chunk copied out before its staging area is overwritten
                // an unknown zero in every LDS address of the chunk body: without it the per-sub-tile address arithmetic (32 channel
                // offsets x 3 scalings) is hoisted out of this run-time loop and, next to 128 live accumulators, spilled 46-59 registers
                int opq = 0;
                if constexpr (ECH2 > 1) asm volatile("" : "+s"(opq));
                const uint8_t* ldsq = lds + opq;
                uint8_t* ostq = ostage + opq;
                const float* sbq = (const float*)((const uint8_t*)s_sb + opq);
                const float* biasq = (const float*)((const uint8_t*)s_bias + opq);
                // Two chunks: chunk ch takes HALF of the row blocks of EVERY wave (j in [ch TM/2, (ch + 1) TM/2)), so all eight waves work in
                // both chunks.  (Chunks of whole wave rows -- rows 0-127, then 128-255 -- left half of the workgroup idle in each: the
                // epilogue of a 256x256 tile took 35 K cycles against 13 K of the plain one, tools/trace_gemm.py --lowrank, round 3.)
                {
#pragma unroll
                    for (int j = 0; j < TM; ++j) {
                        if (ECH2 > 1 && (j * ECH2) / TM != ch) continue;  // wave-uniform; j is unrolled, so every copy runs in one chunk
                        const int tr = wm * WM + j * 32 + (lane & 31);  // tile row = this lane's output row
                        // its row in the chunk's staging area: [wave row][row blocks of this chunk][32]
                        const int sr = ECH2 == 1 ? tr : wm * (WM / ECH2) + (j % (TM / ECH2)) * 32 + (lane & 31);
                        int64_t gm = m0 + tr;
                        if (gm >= hk.M) gm = hk.M - 1;
                        const float sa = p.sa[gm];
                        const uint8_t* lt = ldsq + tr * 64;
                        const int tsw = (tr >> 2) & 3;
                        const v4i ft0 = *(const v4i*)(lt + (((lane >> 5) ^ tsw) << 4)), ft1 = *(const v4i*)(lt + (((2 + (lane >> 5)) ^ tsw) << 4));
#pragma unroll
                        for (int i = 0; i < TN; ++i) {
                            const int ur = wn * WN + i * 32 + (lane & 31);
                            const uint8_t* lu = ldsq + (BM + ur) * 64;
                            const int usw = (ur >> 2) & 3;
                            const v4i fu0 = *(const v4i*)(lu + (((lane >> 5) ^ usw) << 4)), fu1 = *(const v4i*)(lu + (((2 + (lane >> 5)) ^ usw) << 4));
                            v16f lrt;
#pragma unroll
                            for (int e = 0; e < 16; ++e) lrt[e] = 0.0f;
                            if (is_bf) {
                                lrt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(v8bf, fu0), __builtin_bit_cast(v8bf, ft0), lrt, 0, 0, 0);
                                lrt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(v8bf, fu1), __builtin_bit_cast(v8bf, ft1), lrt, 0, 0, 0);
                            } else {
                                lrt = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(v8h, fu0), __builtin_bit_cast(v8h, ft0), lrt, 0, 0, 0);
                                lrt = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(v8h, fu1), __builtin_bit_cast(v8h, ft1), lrt, 0, 0, 0);
                            }
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const int nl0 = wn * WN + i * 32 + 8 * q + 4 * fgrp;
                                const v4f sb4 = *(const v4f*)(sbq + nl0), b4 = *(const v4f*)(biasq + nl0);
                                float r[4], lv[4], b2[4];
#pragma unroll
                                for (int e = 0; e < 4; ++e) lv[e] = hb ? lrt[4 * q + e] + b4[e] : lrt[4 * q + e];
                                if constexpr (is_bf) {  // cast_svd on pairs: one packed convert per two values, then the halves back to f32
                                    const u32 p01 = pack2<SDNQ_BF16>(lv[0], lv[1]), p23 = pack2<SDNQ_BF16>(lv[2], lv[3]);
                                    b2[0] = __uint_as_float(p01 << 16); b2[1] = __uint_as_float(p01 & 0xffff0000u);
                                    b2[2] = __uint_as_float(p23 << 16); b2[3] = __uint_as_float(p23 & 0xffff0000u);
                                } else {
#pragma unroll
                                    for (int e = 0; e < 4; ++e) b2[e] = FT<SDNQ_F16>::round(lv[e]);
                                }
#pragma unroll
                                for (int e = 0; e < 4; ++e) {
                                    const float vv = acc_times_sa<LP>(MT::tof(acc[i][j], 4 * q + e), sa);
                                    r[e] = fmaf(vv, sb4[e], b2[e]);
                                }
                                *(v2i*)(ostq + sr * O_ROW + nl0 * OUT_B) = (v2i){(int)pack2<OUT_T>(r[0], r[1]), (int)pack2<OUT_T>(r[2], r[3])};
                            }
                            __builtin_amdgcn_sched_barrier(0);  // one sub-tile at a time (register pressure next to 128 accumulators)
                        }
                    }
                }
                __syncthreads();
                constexpr int PPR2 = BN * OUT_B / 16, EPP2 = 16 / OUT_B;
#pragma nounroll
                for (int v = tid; v < CH2 * PPR2; v += NT) {
                    const int r = v / PPR2, c = v % PPR2;
                    // staging row -> tile row (the inverse of `sr` above)
                    const int trow = ECH2 == 1 ? r : (r / (WM / ECH2)) * WM + ch * (WM / ECH2) + r % (WM / ECH2);
                    const int64_t gm = m0 + trow, gn0 = n0 + c * EPP2;
                    if (gm >= hk.M || c * EPP2 >= tv.n_lim) continue;  // N % 8 == 0: a piece never straddles N
                    store16(out_piece(p, tv, gm, gn0, c * EPP2, OUT_B), *(const uint4*)(ostage + r * O_ROW + c * 16));
                }
            }
            TRACE(6);
            return;
        }
    }
    if constexpr (EPI != EPI_LRFAST) {
    constexpr int ACC_ROW = BN * 4 + 16;
    // (1) raw accumulators -> LDS [BM][BN] 32-bit (one 16-byte store per run of 4 consecutive output channels):
    //     acc[i][j][reg]: n = wn*WN + i*32 + (reg&3) + 8*(reg>>2) + 4*(lane>>5),  m = wm*WM + j*32 + (lane&31)
    uint8_t* stage = lds + LR_BYTES;
    uint8_t* stage2 = stage + CH * ACC_ROW;  // EPI_LOWRANK: bias2d plane, 16-bit
    const bool has_bias_lr = p.bias != nullptr;
    // RUNTIME loop over the chunks: unrolled (it used to be a static_for) the LOWRANK kernel carried four copies of the staging and
    // of the compact loop -- 15.7 K lines of ISA executed once, at instruction-fetch speed: 45 us of epilogue per 256x256 tile against
    // a 47 us main loop at K = 3072 (tools/trace_gemm.py --lowrank, round 2)
#pragma nounroll
    for (int ch = 0; ch < ECH; ++ch) {

    if (ch > 0) __syncthreads();  // previous chunk fully stored before its staging area is overwritten
    // an unknown zero in every LDS address of the chunk body (see the register-layout loop above: hoisted out of this run-time loop,
    // the per-sub-tile address arithmetic spilled next to the 128 live accumulators of the 256-row tiles)
    int opq = 0;
    if constexpr (ECH > 1) asm volatile("" : "+s"(opq));
    const uint8_t* ldsq = lds + opq;
    uint8_t* stageq = stage + opq;
    uint8_t* stage2q = stage2 + opq;
    const float* s_sbq = (const float*)((const uint8_t*)s_sb + opq);
    const float* s_biasq = (const float*)((const uint8_t*)s_bias + opq);
    const float* s_zpq = (const float*)((const uint8_t*)s_zp + opq);
    const float* s_wcsq = (const float*)((const uint8_t*)s_wcs + opq);
    (void)ldsq; (void)stage2q; (void)s_zpq; (void)s_wcsq; (void)s_biasq; (void)s_sbq;

#pragma unroll
    for (int j = 0; j < TM; ++j)
#pragma unroll
        for (int i = 0; i < TN; ++i) {
            if (ECH > 1 && (wm * WM + j * 32) / CH != ch) continue;  // wave-uniform: this 32-row block is in another chunk
            v16f lrt;
            if constexpr (is_lr<EPI>) {
                if (lr_mfma) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) lrt[e] = 0.0f;
                    int64_t gn = n0 + wn * WN + i * 32 + (lane & 31), gm = m0 + wm * WM + j * 32 + (lane & 31);
                    if (gn >= hk.N) gn = hk.N - 1;
                    if (gm >= hk.M) gm = hk.M - 1;
                    const uint16_t* up = (const uint16_t*)p.lr_up + gn * p.rank + (lane >> 5) * 8;
                    const uint16_t* tt = (const uint16_t*)p.lr_t + gm * p.rank + (lane >> 5) * 8;
                    // staged tiles (rank 32): row r of t at lds + r * 64, of svd_up at lds + (BM + r) * 64, chunk-swizzled
                    const int tr = wm * WM + j * 32 + (lane & 31), ur = wn * WN + i * 32 + (lane & 31);
                    const uint8_t* lt = ldsq + tr * 64, *lu = ldsq + (BM + ur) * 64;
                    const int tsw = (tr >> 2) & 3, usw = (ur >> 2) & 3;
                    // (fetching every sub-tile's t / svd_up fragments in one burst before the DMA drain -- 48 more live VGPRs --
                    // spilled 104 registers in the 256x256 kernel and gained nothing: 72.8 vs 69.7 ms per FLUX step, round 2)
                    for (int kr = 0; kr < p.rank; kr += 16) {
                        v4i fu, ft;
                        if (lr_lds) {
                            const int c = (kr >> 3) + (lane >> 5);
                            fu = *(const v4i*)(lu + ((c ^ usw) << 4));
                            ft = *(const v4i*)(lt + ((c ^ tsw) << 4));
                        } else {
                            fu = *(const v4i*)(up + kr);
                            ft = *(const v4i*)(tt + kr);
                        }
                        if (p.bias_dtype == SDNQ_BF16)
                            lrt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(v8bf, fu), __builtin_bit_cast(v8bf, ft), lrt, 0, 0, 0);
                        else
                            lrt = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(v8h, fu), __builtin_bit_cast(v8h, ft), lrt, 0, 0, 0);
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int ml = wm * WM + j * 32 + frow - ch * CH;
                const int nl0 = wn * WN + i * 32 + 8 * q + 4 * fgrp;
                if constexpr (MM == SDNQ_MM_I8)
                    *(v4i*)(stageq + ml * ACC_ROW + nl0 * 4) = (v4i){acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
                else
                    *(v4f*)(stageq + ml * ACC_ROW + nl0 * 4) = (v4f){acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
                if constexpr (is_lr<EPI>) {
                    if (lr_mfma) {
                        // bias2d = cast_svd(f32(bias[n]) + low-rank) (addmm in the svd dtype, linear_int8.py:57-62; no bias: s_biasq = 0)
                        const v4f b4 = *(const v4f*)(s_biasq + nl0);
                        u32 h[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float v = has_bias_lr ? lrt[4 * q + e] + b4[e] : lrt[4 * q + e];
                            h[e] = (p.bias_dtype == SDNQ_BF16) ? (u32)f32_to_bf16_bits(v) : (u32)f32_to_f16_bits(v);
                        }
                        *(v2i*)(stage2q + ml * B2_ROW + nl0 * 2) = (v2i){(int)(h[0] | (h[1] << 16)), (int)(h[2] | (h[3] << 16))};
                    }
                }
            }
        }
    __syncthreads();
    TRACE(5);
    // (2) one compact loop: 8 consecutive channels of one row per thread -> scale, bias, cast, 16/32-byte store
    constexpr int G8 = BN / 8;
    const bool has_bias = p.bias != nullptr;
    if constexpr (EPI <= EPI_BIAS1D && OUT_T != SDNQ_F32) {
        if (p.out_hw > 0) {
            // channel-major store for the conv forwards: 8 consecutive output positions of ONE channel per thread (they are
            // contiguous in the [B][N][HW] image: HW % 8 == 0 and tiles start on multiples of 64), same fma as below
#pragma nounroll
            for (int v = tid; v < (CH / 8) * BN; v += NT) {
                const int n = v / (CH / 8), r8 = (v % (CH / 8)) * 8;
                const int64_t gm = m0 + ch * CH + r8, gn = n0 + n;
                if (gm >= hk.M || gn >= hk.N) continue;
                const float sbn = is_float_mm<MM> ? 1.0f : s_sbq[n];
                const float bn = (EPI == EPI_BIAS1D) ? s_biasq[n] : 0.0f;
                float o[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float a;
                    if constexpr (MM == SDNQ_MM_I8) a = (float)*(const int*)(stageq + (r8 + e) * ACC_ROW + n * 4);
                    else a = *(const float*)(stageq + (r8 + e) * ACC_ROW + n * 4);
                    if constexpr (is_float_mm<MM>) {
                        o[e] = (EPI == EPI_BIAS1D) ? a + bn : a;
                    } else {
                        const float vv = a * p.sa[gm + e];
                        o[e] = (EPI == EPI_BIAS1D) ? fmaf(vv, sbn, bn) : vv * sbn;
                    }
                }
                const int64_t img = gm / p.out_hw, px = gm - img * p.out_hw;
                *(uint4*)((uint8_t*)p.out + ((img * p.ldc + gn) * p.out_hw + px) * OUT_B) = Vec16<OUT_T>::pack(o);
            }
            continue;  // next chunk
        }
    }
    const bool lr_fast = is_lr<EPI> && lr_mfma && p.zp == nullptr && p.a_zp == nullptr;
#pragma nounroll
    for (int v = tid; v < CH * G8; v += NT) {
        const int r = v / G8, c8 = (v % G8) * 8;  // r: row inside the chunk
        const int64_t gm = m0 + ch * CH + r, gn0 = n0 + c8;
        if (gm >= hk.M || c8 >= tv.n_lim) continue;  // N % 8 == 0: a group of 8 never straddles N
        const float sa = is_float_mm<MM> ? 1.0f : p.sa[gm];
        float zsum = 0.0f, azp = 0.0f;
        if constexpr (is_lr<EPI>) {
            if (p.zp_rowsum) {  // sum(int32).to(scale dtype).mul_(input_scale), linear_int8.py:66 (LP: both steps rounded to bf16)
                zsum = (float)p.zp_rowsum[gm];
                if constexpr (LP) zsum = FT<SDNQ_BF16>::round(FT<SDNQ_BF16>::round(zsum) * sa);
                else zsum *= sa;
            }
            if (p.a_zp) azp = p.a_zp[gm];
        }
        float o[8];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            float a4[4];
            if constexpr (MM == SDNQ_MM_I8) {
                const v4i t = *(const v4i*)(stageq + r * ACC_ROW + (c8 + 4 * h) * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) a4[e] = (float)t[e];  // int32 -> f32 (RNE above 2^24)
            } else {
                const v4f t = *(const v4f*)(stageq + r * ACC_ROW + (c8 + 4 * h) * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) a4[e] = t[e];
            }
            if constexpr (is_float_mm<MM>) {  // F.linear: f32 accumulate, bias added in f32, one rounding
#pragma unroll
                for (int e = 0; e < 4; ++e) o[4 * h + e] = (EPI == EPI_BIAS1D) ? a4[e] + s_biasq[c8 + 4 * h + e] : a4[e];
                continue;
            }
            const v4f sb4 = *(const v4f*)(s_sbq + c8 + 4 * h);
            v4f lr4 = {0.0f, 0.0f, 0.0f, 0.0f};  // bias2d values (already cast to the svd dtype) of the staged low-rank plane
            if constexpr (is_lr<EPI>) {
                if (lr_mfma) {
                    const v2i b2 = *(const v2i*)(stage2q + r * B2_ROW + (c8 + 4 * h) * 2);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const uint16_t bits = (uint16_t)(((u32)b2[e >> 1]) >> (16 * (e & 1)));
                        lr4[e] = (p.bias_dtype == SDNQ_BF16) ? bf16_bits_to_f32(bits) : f16_bits_to_f32(bits);
                    }
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float vv = acc_times_sa<LP>(a4[e], sa);
                float res;
                if constexpr (EPI == EPI_NONE) {
                    res = vv * sb4[e];
                } else if constexpr (EPI == EPI_BIAS1D) {
                    res = fmaf(vv, sb4[e], s_biasq[c8 + 4 * h + e]);
                } else if constexpr (EPI == EPI_BIAS2D) {
                    res = fmaf(vv, sb4[e], ldf_rt(p.bias, gm * p.ld_bias + gn0 + 4 * h + e, p.bias_dtype));
                } else {
                    // bias2d = cast_svd(f32(bias[n]) + sum_r t[m][r]*up[n][r]) (linear_int8.py:57-62), then the zero-point
                    // term f32(rowsum)*sa*zp[n] + bias2d (linear_int8.py:65-69), all f32 into the single-rounding fma
                    const int cn = c8 + 4 * h + e;
                    if (lr_fast) {  // SVD only (cfg5): staged bias2d, no zero-point terms
                        res = fmaf(vv, sb4[e], lr4[e]);
                    } else {
                        float bv = s_biasq[cn];
                        bool has = has_bias;
                        if (p.lr_t) {
                            if (lr_mfma) {
                                bv = lr4[e];  // staged: cast_svd(bias + low-rank)
                            } else {  // f32 factors or a rank that is not a multiple of 16: plain fma chain
                                float sacc = 0.0f;
                                for (int rr = 0; rr < p.rank; ++rr)
                                    sacc = fmaf(ldf_rt(p.lr_t, gm * p.rank + rr, p.bias_dtype), ldf_rt(p.lr_up, (n0 + cn) * p.rank + rr, p.bias_dtype), sacc);
                                bv = round_rt(has ? sacc + bv : sacc, p.bias_dtype);
                            }
                            has = true;
                        }
                        float zb = 0.0f;
                        bool hasz = false;
                        if (p.zp) {
                            zb = zsum * s_zpq[cn];
                            if constexpr (LP) zb = FT<SDNQ_BF16>::round(zb);  // .mul(zero_point) on bf16 tensors
                            hasz = true;
                        }
                        if (p.a_zp) {  // uint8 matmul: + colsum(w)*ws*xzp  + K * (xzp * wzp)   (linear_uint8.py:61-66)
                            if constexpr (LP) {  // the same chain on bfloat16 tensors: every torch op rounds once (wcs arrives rounded twice)
                                const auto rb = [](float x) { return FT<SDNQ_BF16>::round(x); };
                                const float t2 = rb(s_wcsq[cn] * azp);
                                zb = hasz ? rb(zb + t2) : t2;
                                if (p.zp) {
                                    if (p.zp_k < 0) zb = rb(zb + rb(rb(azp * (float)(-p.zp_k)) * s_zpq[cn]));
                                    else zb = rb(fmaf(rb(azp * s_zpq[cn]), (float)(p.zp_k ? p.zp_k : hk.K), zb));
                                }
                            } else {
                            const float t2 = s_wcsq[cn] * azp;
                            zb = hasz ? zb + t2 : t2;
                            if (p.zp) {
                                if (p.zp_k < 0)  // conv form (conv_uint8.py:66): input_zero_point.mul_(K) in place, .mul(zero_point), a plain add_: three roundings
                                    zb = __fadd_rn(zb, __fmul_rn(__fmul_rn(azp, (float)(-p.zp_k)), s_zpq[cn]));
                                else  // linear form (linear_uint8.py:66): add_(mul(xzp, wzp), alpha=K) is ONE fused multiply-add on the CPU
                                    zb = fmaf(azp * s_zpq[cn], (float)(p.zp_k ? p.zp_k : hk.K), zb);
                            }
                            }
                            hasz = true;
                        }
                        if (hasz) {
                            bv = has ? zb + bv : zb;  // zero_bias.add_(bias), linear_int8.py:67-68
                            if constexpr (LP) bv = FT<SDNQ_BF16>::round(bv);
                            has = true;
                        }
                        res = has ? fmaf(vv, sb4[e], bv) : vv * sb4[e];
                    }
                }
                o[4 * h + e] = res;
            }
        }
        uint8_t* dst = out_piece(p, tv, gm, gn0, c8, OUT_B);
        if constexpr (OUT_T == SDNQ_F32) {
            store16(dst, Vec16<SDNQ_F32>::pack(o));
            store16(dst + 16, Vec16<SDNQ_F32>::pack(o + 4));
        } else {
            store16(dst, Vec16<OUT_T>::pack(o));
        }
    }
    }
    }  // EPI != EPI_LRFAST
    } else {
    // ======== simple epilogues of the 64-row tiles: arithmetic in the MFMA register layout (+2.6 % on the SDXL step's GEMMs) ==
    // Every lane owns output position m = wm*WM + j*32 + (lane&31) and channels n = wn*WN + i*32 + (reg&3) + 8*(reg>>2) +
    // 4*(lane>>5); per-channel vectors come from LDS; only the FINAL values (OUT_B bytes each) are staged through LDS
    // [CH rows][BN] to leave as 16-byte row pieces (or, for the conv forwards, channel-major runs of 8 positions): half the LDS
    // traffic of staging raw 32-bit accumulators, and ~6 instructions per output, so unrolling it stays small.
    constexpr int OUT_ROW = BN * OUT_B + 16;  // staging row pitch in bytes
    uint8_t* stage = lds;
    // 256-row tiles leave in two chunks of 128 rows.  A RUNTIME loop with one wave-uniform test per chunk: the arithmetic of a wave's
    // sub-tiles is emitted once (unrolled per chunk -- or staged as raw accumulators with a compact loop, as these tiles used to
    // be -- the epilogue is thousands of instructions executed once at instruction-fetch speed).
#pragma nounroll
    for (int ch = 0; ch < ECHR; ++ch) {
    if (ch > 0) __syncthreads();  // previous chunk fully stored before its staging area is overwritten
    if (ECHR == 1 || (wm * WM) / CHR == ch) {
#pragma unroll
    for (int j = 0; j < TM; ++j) {
        const int ml = wm * WM + j * MS + frow - ch * CHR;        // row inside the chunk
        int64_t gm = m0 + wm * WM + j * MS + frow;
        const bool m_ok = gm < hk.M;
        if (!m_ok) gm = hk.M - 1;
        const float sa = is_float_mm<MM> ? 1.0f : p.sa[gm];
#pragma unroll
        for (int i = 0; i < TN; ++i) {
#pragma unroll
            for (int q = 0; q < (MS == 32 ? 4 : 1); ++q) {
                // 4 consecutive channels: 32x32 tiles hold runs (reg & 3) + 8 (reg >> 2) + 4 fgrp, 16x16 tiles the one run 4 fgrp + reg
                const int nl0 = wn * WN + i * MS + (MS == 32 ? 8 * q + 4 * fgrp : 4 * fgrp);
                float o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int cn = nl0 + e;
                    const float a = MT::tof(acc[i][j], 4 * q + e);  // int32 -> f32 (RNE above 2^24) | f32
                    float res;
                    if constexpr (is_float_mm<MM>) {  // F.linear: f32 accumulate, bias added in f32, one rounding
                        res = (EPI == EPI_BIAS1D) ? a + s_bias[cn] : a;
                    } else {
                        const float vv = acc_times_sa<LP>(a, sa);
                        const float sbn = s_sb[cn];
                        if constexpr (EPI == EPI_NONE) {
                            res = vv * sbn;
                        } else if constexpr (EPI == EPI_BIAS1D) {
                            res = fmaf(vv, sbn, s_bias[cn]);
                        } else {
                            int64_t gn = n0 + cn;
                            if (gn >= hk.N) gn = hk.N - 1;
                            res = fmaf(vv, sbn, ldf_rt(p.bias, gm * p.ld_bias + gn, p.bias_dtype));  // EPI_BIAS2D
                        }
                    }
                    o[e] = res;
                }
                uint8_t* dst = stage + ml * OUT_ROW + nl0 * OUT_B;
                if constexpr (OUT_T == SDNQ_F32) {
                    *(uint4*)dst = Vec16<SDNQ_F32>::pack(o);
                } else {
                    const uint32_t lo = pack2<OUT_T>(o[0], o[1]), hi = pack2<OUT_T>(o[2], o[3]);
                    *(uint2*)dst = make_uint2(lo, hi);
                }
            }
        }
    }
    }  // this wave's rows are in chunk ch
    __syncthreads();
    TRACE(5);
    if constexpr (EPI <= EPI_BIAS1D && OUT_T != SDNQ_F32) {
        if (p.out_hw > 0) {
            // channel-major store for the conv forwards: 8 consecutive output positions of ONE channel per thread (they are
            // contiguous in the [B][N][HW] image: HW % 8 == 0 and tiles start on multiples of 64)
#pragma nounroll
            for (int v = tid; v < (CHR / 8) * BN; v += NT) {
                const int n = v / (CHR / 8), r8 = (v % (CHR / 8)) * 8;
                const int64_t gm = m0 + ch * CHR + r8, gn = n0 + n;
                if (gm >= hk.M || gn >= hk.N) continue;
                uint16_t h8[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) h8[e] = *(const uint16_t*)(stage + (r8 + e) * OUT_ROW + n * 2);
                const int64_t img = gm / p.out_hw, px = gm - img * p.out_hw;
                *(uint4*)((uint8_t*)p.out + ((img * p.ldc + gn) * p.out_hw + px) * 2) = *(const uint4*)h8;
            }
            continue;  // next chunk
        }
    }
    // plain copy: 16-byte pieces of the output rows
    constexpr int PPR = BN * OUT_B / 16;  // pieces per row
    constexpr int EPP = 16 / OUT_B;       // output elements per piece
#pragma nounroll
    for (int v = tid; v < CHR * PPR; v += NT) {
        const int r = v / PPR, c = v % PPR;
        const int64_t gm = m0 + ch * CHR + r, gn0 = n0 + c * EPP;
        if (gm >= hk.M || c * EPP >= tv.n_lim) continue;  // N % 8 == 0: a piece never straddles N
        store16(out_piece(p, tv, gm, gn0, c * EPP, OUT_B), *(const uint4*)(stage + r * OUT_ROW + c * 16));
    }
    }
    }
    TRACE(6);
}

// The next GEMM launch of this thread carries these ranges as prefetch work (consumed by that launch whether or not it had room for it)
// (`device`: the device that was current when the hint was given -- the one its pointers live on; a launch on another device drops it)
struct PrefetchHint { const void* ptr[4]; int64_t bytes[4]; int device; };
thread_local PrefetchHint g_pf_hint = {};
inline int current_device() { int dev = 0; return hipGetDevice(&dev) == hipSuccess ? dev : -1; }
// the pending hint, if it belongs to the device this thread is launching on; a hint of another device is discarded
inline bool pf_hint_pending() {
    if (!(g_pf_hint.ptr[0] || g_pf_hint.ptr[1] || g_pf_hint.ptr[2] || g_pf_hint.ptr[3])) return false;
    if (g_pf_hint.device != current_device()) { g_pf_hint = PrefetchHint{}; return false; }
    return true;
}
inline int cu_count() {  // of the CURRENT device (a process may drive different parts / partitions)
    static std::atomic<int> cus[64];
    const int dev = current_device();
    if (dev < 0 || dev >= 64) return 256;
    int v = cus[dev].load(std::memory_order_relaxed);
    if (v == 0) {
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        cus[dev].store(v, std::memory_order_relaxed);
    }
    return v;
}

// sdnq_hip_scaled_mm_tile: a dry run of the tile heuristics -- the launcher that WOULD run records its tile here and returns
struct TileProbe { int bm, bn, threads; };
thread_local TileProbe* g_tile_probe = nullptr;

template <int MM, int OUT_T, int EPI, int BM, int BN, int WM, int WN, int NS, int LD = LD_DMA, int BK = BKB, bool LP = false>
int launch_one(GemmParams p, hipStream_t s) {
    static_assert(!LP || (OUT_T == SDNQ_BF16 && !is_float_mm<MM>), "LP: the bf16-scale epilogue of the quantized matmuls");
    constexpr int NW = (BM / WM) * (BN / WN);
    if (g_tile_probe) { *g_tile_probe = TileProbe{BM, BN, NW * 64}; return SDNQ_OK; }
    constexpr int MAIN = (LD == LD_HT) ? HT_SLOTS * HT_BYTES : NS * (BM * BK + BN * (is_w8a16<MM> ? BK / 2 : BK));
    constexpr int CHS = BM > 128 ? 64 : BM, CHR = epi_chunk_rows<EPI, BM, BN, FT<OUT_T>::bytes>();  // rows per epilogue chunk (as in gemm_kernel)
    constexpr int EPI_GEN = (BM + BN) * 64 + CHS * (BN * 4 + 16) + CHS * (BN * 2 + 16);
    constexpr int EPI_REG = (is_lr<EPI> ? (BM + BN) * 64 : 0) + CHR * (BN * FT<OUT_T>::bytes + 16);
    constexpr int EPIB = (is_lr<EPI> && (EPI_GEN > EPI_REG || OUT_T == SDNQ_F32)) ? EPI_GEN : EPI_REG;
    constexpr int LDS_BYTES = (MAIN > EPIB ? MAIN : EPIB) + ((is_lr<EPI> || is_w8a16<MM>) ? 4 : 2) * BN * 4;  // ring | staging, then the per-channel vectors
    static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
    auto kern = gemm_kernel<MM, OUT_T, EPI, BM, BN, WM, WN, NS, LD, BK, LP>;
    // (the attribute belongs to the function ON ONE DEVICE: a process that drives several GPUs sets it once per device, not once)
    static std::atomic<uint64_t> attr_devices{0};
    if (LDS_BYTES > 64 * 1024) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) return SDNQ_ERR_LAUNCH;
        const uint64_t bit = 1ull << (dev & 63);
        if (!(attr_devices.load(std::memory_order_acquire) & bit)) {
            if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess) return SDNQ_ERR_LAUNCH;
            attr_devices.fetch_or(bit, std::memory_order_release);
        }
    }
    if (p.lda == 0) p.lda = p.K;
    if (p.ldb == 0) p.ldb = is_w8a16<MM> ? p.K / 2 : p.K;
    if (p.ldc == 0) p.ldc = p.N;
    p.tiles_m = (int)((p.M + BM - 1) / BM);
    p.tiles_n = (int)((p.N + BN - 1) / BN);
    {
        static const int gm_env = [] { const char* e = getenv("SDNQ_HIP_GROUP_M"); return e ? atoi(e) : 0; }();  // tuning aid
        // near-square patch of ~32 concurrent tiles per XCD: rows*BM ~ cols*BN
        int gm = gm_env > 0 ? gm_env : (BM >= BN ? 6 : 8);
        if (gm > p.tiles_m) gm = p.tiles_m;
        if (gm > 255) gm = 255;  // travels in 8 bits of hk_flags
        p.group_m = gm;
        static const int swz_env = [] { const char* e = getenv("SDNQ_HIP_SWZ"); return e ? atoi(e) : 7; }();
        p.swz = swz_env;
        auto magic = [](uint64_t d) -> uint32_t { return d <= 1 ? 0u : (uint32_t)(((1ull << 32) + d - 1) / d); };
        static const bool fast_env = [] { const char* e = getenv("SDNQ_HIP_FASTMAP"); return !e || atoi(e) != 0; }();  // A/B aid
        const uint64_t nwg = (uint64_t)p.tiles_m * (uint64_t)p.tiles_n, per_group = (uint64_t)gm * (uint64_t)p.tiles_n;
        const uint64_t tail = (uint64_t)(p.tiles_m % gm);
        p.fastmap = fast_env && nwg * (per_group > (uint64_t)gm ? per_group : (uint64_t)gm) <= (1ull << 32);
        p.mg_per_group = magic(per_group); p.mg_group_m = magic((uint64_t)gm); p.mg_tail = magic(tail ? tail : 1);
        p.fastunit = fast_env && p.units != nullptr && p.unit_n > 0 && (uint64_t)p.N * (uint64_t)p.unit_n <= (1ull << 32);
        p.mg_unit = magic(p.unit_n > 0 ? (uint64_t)p.unit_n : 1);
    }
    if (p.M > 0x7fffffffll || p.N > 0x7fffffffll || p.K > 0x7fffffffll || p.lda > 0x7fffffffll || p.ldb > 0x7fffffffll) return SDNQ_ERR_SHAPE;
    const uint32_t hk_flags = (uint32_t)(p.group_m & 0xff) | ((uint32_t)(p.swz & 0xff) << 8) | ((uint32_t)(p.fastmap != 0) << 16) |
                              ((uint32_t)(p.fastunit != 0) << 17) | ((uint32_t)(p.units != nullptr) << 18);
    // prefetch workgroups ride along where the launch leaves workgroup slots free (its last round does not fill the chip): they run
    // beside the tiles on CUs that would idle, so the hint costs the launch nothing; a launch without room drops the hint
    int pf_wgs = 0;
    if (pf_hint_pending()) {
        static std::atomic<int> occ{0};
        int o = occ.load(std::memory_order_relaxed);
        if (o == 0) {
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&o, (const void*)kern, NW * 64, LDS_BYTES) != hipSuccess || o <= 0) o = -1;
            occ.store(o, std::memory_order_relaxed);
        }
        const int64_t slots = o > 0 ? (int64_t)o * cu_count() : 0, tiles = (int64_t)p.tiles_m * p.tiles_n;
        // hosts are the ONE-ROUND launches of the bs = 1 steps: workgroups appended to a multi-round launch start in its last round and
        // stretch the tail, and beside the power-limited 256x256 tiles of a FLUX-size GEMM they cost more than cold weights do there
        // (flux_int4_had 34.4 -> 34.8 ms, flux_int8_svd 40.3 -> 40.9 with every launch hosting; profiles/r05_prefetch_across_layers.txt)
        static const double host_max_mac = [] { const char* e = getenv("SDNQ_HIP_PREFETCH_HOST_MAX_GMAC"); return (e ? atof(e) : 20.0) * 1e9; }();
        const bool small = (double)p.M * (double)p.N * (double)p.K <= host_max_mac;
        const int64_t room = slots > 0 && tiles < slots && small ? slots - tiles : 0;
        int64_t lines = 0;
        for (int r = 0; r < 4; ++r) {
            p.pf_ptr[r] = nullptr; p.pf_lines[r] = 0;
            if (!g_pf_hint.ptr[r] || g_pf_hint.bytes[r] <= 0) continue;
            const uintptr_t a0 = (uintptr_t)g_pf_hint.ptr[r] & ~(uintptr_t)127;
            const int64_t n = (int64_t)(((uintptr_t)g_pf_hint.ptr[r] + (uintptr_t)g_pf_hint.bytes[r] + 127 - a0) / 128);
            if (n > 0x7fffffffll) continue;
            p.pf_ptr[r] = (const uint8_t*)a0; p.pf_lines[r] = (int)n;
            lines += n;
        }
        static const int pf_max = [] { const char* e = getenv("SDNQ_HIP_PREFETCH_WGS"); return e ? atoi(e) : 96; }();  // tuning aid
        int64_t want = (lines + NW * 64 * 4 - 1) / (NW * 64 * 4);  // ~4 lines per thread
        if (want > pf_max) want = pf_max;
        pf_wgs = (int)(want < room ? want : room);
        g_pf_hint = PrefetchHint{};
    }
    hipLaunchKernelGGL(kern, dim3(p.tiles_m * p.tiles_n + pf_wgs), dim3(NW * 64), LDS_BYTES, s, p.a, p.b, (int)p.lda, (int)p.ldb, (int)p.M, (int)p.N, (int)p.K,
                       p.tiles_m, p.tiles_n, hk_flags, p.mg_per_group, p.mg_group_m, p);
    SDNQ_CHECK_LAUNCH();
    return SDNQ_OK;
}

// Tile choice. The chip has 256 CUs and the LDS-DMA latency is ~1 us, so what matters for diffusion-size GEMMs
// (1-30 GOP, a few hundred tiles) is (a) enough workgroups to touch every CU and (b) as many bytes in flight per CU
// as the 160 KB LDS allows: every configuration uses a ring deep enough to fill most of the LDS of its CU.
// Tile override for tuning sweeps: environment SDNQ_HIP_TILE at first use, or sdnq_hip_set_tile_override() at run time.
std::atomic<int> g_forced_tile{-2};
inline int forced_tile() {
    int f = g_forced_tile.load(std::memory_order_relaxed);
    if (f == -2) {
        const char* e = getenv("SDNQ_HIP_TILE");
        f = e ? atoi(e) : -1;
        g_forced_tile.store(f, std::memory_order_relaxed);
    }
    return f;
}

// Shape-keyed override for IN-STEP tuning (tools/tune_tiles_in_step.py): SDNQ_HIP_TILE_MAP="MxNxK=tile,MxNxK=tile,..." forces a tile
// for exactly those problems and leaves every other launch of the step to the heuristics -- a tile is judged by what it does to the
// step, not by a kernel replayed alone.
inline int forced_tile_for(const GemmParams& p) {
    struct Ent { int64_t m, n, k; int tile; };
    static const std::vector<Ent> map = [] {
        std::vector<Ent> v;
        const char* e = getenv("SDNQ_HIP_TILE_MAP");
        while (e && *e) {
            Ent t{};
            int used = 0;
            if (sscanf(e, "%ldx%ldx%ld=%d%n", &t.m, &t.n, &t.k, &t.tile, &used) == 4) v.push_back(t);
            else break;
            e += used;
            if (*e == ',') ++e;
        }
        return v;
    }();
    for (const Ent& t : map)
        if (t.m == p.M && t.n == p.N && t.k == p.K) return t.tile;
    return forced_tile();
}

// the half-tile ring addresses a tile's rows with 32-bit byte offsets from the tile's first row
inline bool ht_ok(const GemmParams& p) {
    const int64_t lda = p.lda ? p.lda : p.K, ldb = p.ldb ? p.ldb : p.K;
    return (p.K % 128) == 0 && 256 * lda + p.K < (int64_t)1 << 31 && 256 * ldb + p.K < (int64_t)1 << 31;
}

// what the K-split tile (gemm_ks.hip) takes: one plain [M][N] output, no unit table
inline bool ks_ok(const GemmParams& p) {
    return p.units == nullptr && p.seg_n == 0 && p.out_hw == 0 && (p.bias == nullptr || p.bias_ndim <= 1) &&
           sdnq_internal_ks_eligible(p.M, p.N, p.K, p.lda ? p.lda : p.K, p.ldb ? p.ldb : p.K);
}

// the ping-pong configurations exist for the quantized matmuls with the plain epilogues and 16-bit outputs (the model paths)
template <int MM, int OUT_T, int EPI> constexpr bool PP_OK = !is_float_mm<MM> && EPI <= EPI_BIAS1D && OUT_T != SDNQ_F32;

template <int MM, int OUT_T, int EPI>
int launch_tiles(const GemmParams& p, hipStream_t s) {
    auto tiles = [&](int bm, int bn) { return ((p.M + bm - 1) / bm) * ((p.N + bn - 1) / bn); };
    const int force = forced_tile_for(p);  // tuning aid
    if constexpr (EPI != EPI_BIAS2D && EPI != EPI_LOWRANK) {
        if (force == 0) return launch_one<MM, OUT_T, EPI, 256, 256, 128, 64, 4, LD_PIPE, 64>(p, s);
    } else if (force == 0) return launch_one<MM, OUT_T, EPI, 256, 128, 64, 64, 3, LD_PIPE, 64>(p, s);
    if (force == 1) return launch_one<MM, OUT_T, EPI, 64, 128, 32, 32, 3, LD_DMA>(p, s);
    if (force == 2) return launch_one<MM, OUT_T, EPI, 64, 64, 32, 32, 4, LD_PIPE>(p, s);
    if (force == 3) return launch_one<MM, OUT_T, EPI, 256, 128, 64, 64, 3, LD_PIPE, 64>(p, s);
    if constexpr (PP_OK<MM, OUT_T, EPI>) {
        if (force == 4) return launch_one<MM, OUT_T, EPI, 256, 256, 128, 64, 4, LD_PP, 64>(p, s);
        if (force == 5) return launch_one<MM, OUT_T, EPI, 256, 128, 64, 64, 3, LD_PP, 64>(p, s);
        if (force == 6) return launch_one<MM, OUT_T, EPI, 128, 256, 64, 64, 4, LD_PP, 64>(p, s);
        if (force == 7) return launch_one<MM, OUT_T, EPI, 128, 128, 64, 32, 3, LD_PP, 128>(p, s);
        if (force == 8) return launch_one<MM, OUT_T, EPI, 128, 128, 64, 32, 4, LD_PP, 64>(p, s);
        if (force == 9) return launch_one<MM, OUT_T, EPI, 64, 128, 32, 32, 3, LD_PP, 128>(p, s);
        if (force == 10) return launch_one<MM, OUT_T, EPI, 128, 128, 64, 32, 4, LD_PIPE, 64>(p, s);
        if (force == 11) return launch_one<MM, OUT_T, EPI, 128, 256, 64, 64, 3, LD_PP, 64>(p, s);
        if (force == 12) return launch_one<MM, OUT_T, EPI, 256, 160, 32, 160, 4, LD_PIPE, 64>(p, s);
        if (force == 13) return launch_one<MM, OUT_T, EPI, 256, 160, 32, 160, 3, LD_PIPE, 64>(p, s);
        if (force == 14) return launch_one<MM, OUT_T, EPI, 128, 320, 32, 160, 3, LD_PIPE, 64>(p, s);
        if (force == 15) return launch_one<MM, OUT_T, EPI, 256, 160, 32, 160, 4, LD_DMA, 64>(p, s);
        if (force == 16) return launch_one<MM, OUT_T, EPI, 128, 320, 32, 160, 4, LD_PIPE, 64>(p, s);
        if (force == 17) return launch_one<MM, OUT_T, EPI, 128, 128, 64, 32, 3, LD_PIPE, 128>(p, s);
        if (force == 18) return launch_one<MM, OUT_T, EPI, 256, 256, 128, 64, 4, LD_8P, 64>(p, s);
        if (force == 19) return launch_one<MM, OUT_T, EPI, 256, 128, 64, 64, 3, LD_8P, 128>(p, s);
        if (force == 20 && (p.K % 128) == 0 && ht_ok(p)) return launch_one<MM, OUT_T, EPI, 256, 256, 128, 64, 2, LD_HT, 128>(p, s);
        if (force == 20) return launch_one<MM, OUT_T, EPI, 256, 256, 128, 64, 4, LD_PIPE, 64>(p, s);
        if (force == 21) return launch_one<MM, OUT_T, EPI, 64, 128, 32, 32, 3, LD_OV, 128>(p, s);
        if (force == 22) return launch_one<MM, OUT_T, EPI, 64, 128, 32, 32, 4, LD_OV, 128>(p, s);
        if (force == 23) return launch_one<MM, OUT_T, EPI, 128, 128, 64, 32, 3, LD_OV, 128>(p, s);
        if (force == 24) return launch_one<MM, OUT_T, EPI, 256, 128, 64, 64, 3, LD_OV, 128>(p, s);
        if (force == 25) return launch_one<MM, OUT_T, EPI, 256, 160, 32, 160, 3, LD_OG, 128>(p, s);
        if (force == 26) return launch_one<MM, OUT_T, EPI, 128, 128, 64, 32, 3, LD_OG, 128>(p, s);
        if (force == 27) return launch_one<MM, OUT_T, EPI, 256, 128, 64, 64, 3, LD_OG, 128>(p, s);
        // 28: 64x80 tiles, eight waves in two K groups that share one 8-deep ring, int32 partial sums reduced through LDS (gemm_ks.hip, round 6);
        //     problems it does not take (fp8, grouped / linked / conv outputs, K tails) fall through to the heuristics
        if constexpr (MM == SDNQ_MM_I8) {
            if (force == 28 && ks_ok(p) && g_tile_probe) { *g_tile_probe = TileProbe{64, 80, 512}; return SDNQ_OK; }
            if (force == 28 && ks_ok(p))
                return sdnq_internal_scaled_mm_ks(p.a, p.b, p.sa, p.sb, EPI == EPI_BIAS1D ? p.bias : nullptr, p.bias_dtype, p.out, OUT_T, p.M, p.N, p.K, p.lda, p.ldb, p.ldc, s);
        }
        // (round 5, measured and NOT instantiated: 64x80 tiles on v_mfma_i32_16x16x64_i8 -- four waves of 16x80, 256 workgroups for 1024 x 1280
        //  outputs, 25 % fewer LDS-fill bytes per CU -- on LD_DMA and on LD_OV, 3 / 4 ring slots: replayed alone 7.67-8.09 us against 7.09 at
        //  K = 1280, in the step +0.36..+0.53 ms over the 180 launches of 1024 x 1280 x 1280, +0.11..+0.21 at K = 5120; profiles/r05_overlapped_ring_lab.txt)
        // (round 4 lab, not instantiated: 128x160 tiles -- N = 320 in two exact columns, 256 tiles at M = 16384 -- equal the 64x128 tile on
        //  the conv step, 64x320 tiles are slower; profiles/r04_conv_tiles_in_step.txt)
        // (round 4, profiles/r04_ring_depth_in_step.txt: 5- and 6-deep rings for the 64x128 tile -- the one-workgroup-per-CU problems of the
        //  SDXL step, 160 tiles on 256 CUs -- judged on the step: 1024 x 1280 x 1280 +0.11 / +0.13 ms, 4096 x 640 x 640 +0.15 / +0.16 ms,
        //  1024 x 1280 x 5120 +-0.00: more bytes in flight did not shorten that loop -- its 720 cycles per stage are the SUM of DMA issue,
        //  fragment reads and MFMAs of a lock-step stage (round 5, DESIGN.md section 6: the fill path alone sustains 47 B/clk/CU), not a service
        //  rate -- and the deeper prologue only delays the first MFMA.  Not instantiated in the library.)
        // (64x80 tiles on v_mfma_i32_16x16x64_i8 -- MM_I8_16, instantiated by tools/micro/gemm_lab.hip only -- cut 1024 x 1280 outputs
        //  into exactly 256 workgroups with 25 % fewer LDS-fill bytes per CU, and measured SLOWER than 160 tiles of 64x128: 9.5 vs 7.9 us
        //  at K = 1280, 23.5 vs 19.0 us at K = 5120: four waves of 16x80 read six fragments per five 16-cycle MFMAs)
    }
    // measured on MI355X (tools/bench_gemm.py, profiles/r01_gemm_tile_sweep.txt). Every kernel launch starts with cold
    // L2s (data comes from MALL/HBM at ~2 us loaded latency) and the L2->LDS fill rate per CU is ~30 B/clk, so:
    //  * large problems: 256x256 tiles (8 waves of 128x64, 64-byte K stages, 4-deep ring) -- twice the MACs per byte
    //    staged through LDS of a 128x128 tile: 2.0-2.1 POP/s at 8192^3 / 16384x8192x4096 vs 1.4-1.5;
    //  * diffusion-size GEMMs (1-30 GOP): 64x128 tiles, two waves per SIMD and TWO co-resident workgroups per CU
    //    (3-deep ring, 72 KB LDS each);
    //  * few-row GEMMs (M <= 128, e.g. the 77-token text projections): 64x64 tiles.
    // Deeper rings for the 64x128 tiles (5-6 stages, one workgroup per CU) measured 5-30 % slower than 3 stages x 2 workgroups.
    // Software-pipelined fragment reads (LD_PIPE) measured +3..9 % on the 256x256 tiles and +8 % on the 64x64 ones, -1.5 % on
    // the SDXL step for the 64x128 tiles (one MFMA per sub-step leaves nothing to hide behind), so those keep LD_DMA.
    auto fits = [&](int bn) { return p.units == nullptr || (p.unit_n % bn) == 0; };  // grouped launch: a tile stays inside one unit
    // (round 2, tools/micro/gemm_lab.hip + tools/sweep_gemm.py, profiles/r02_gemm_tile_sweep.txt) short-K problems do not amortise
    // the 256x256 tile's prologue / epilogue: 4096 x 5120 x 640 ran 39 us on it vs 29 us on 256x128
    // (160 rather than 200 tiles since the register-layout epilogue: 4096 x 3072 x 3072 -- 192 tiles -- 56.5 us vs 57.7-62.6 on 256x128,
    //  4096 x 3072 x 12288 160 vs 198 us; profiles/r02_gemm_tile_sweep.txt)
    // (a 2-D bias -- the operator seam's rare form -- never runs on the 256x256 tiles: next to 128 accumulators its per-element global
    //  bias loads spilled 780-920 registers; tools/check_spills.py keeps the library free of scratch)
    if constexpr (EPI == EPI_LOWRANK) {
        // low-rank layers reach the 256x256 tiles through EPI_LRFAST only (see the enum)
        if constexpr (OUT_T != SDNQ_F32) {
            if (tiles(256, 256) >= 160 && p.K >= 2048 && fits(256) && p.lr_t != nullptr && p.rank == 32 && p.bias_dtype == OUT_T &&
                p.zp == nullptr && p.a_zp == nullptr) {
                if (ht_ok(p)) return launch_one<MM, OUT_T, EPI_LRFAST, 256, 256, 128, 64, 2, LD_HT, 128>(p, s);
                return launch_one<MM, OUT_T, EPI_LRFAST, 256, 256, 128, 64, 4, LD_PIPE, 64>(p, s);
            }
        }
    } else if constexpr (EPI != EPI_BIAS2D) {
        if (p.M > 128 && tiles(256, 256) >= 160 && p.K >= 2048 && fits(256)) {  // (few rows against many channels: the 128-row stream below)
            // round 3: the half-tile ring (LD_HT: 128-byte rows, buffer-load DMAs with no vector-ALU address work, fine ping-pong) is
            // 15-22 % faster than the 64-byte-stage software pipeline on every FLUX / large shape (profiles/r03_gemm_ht.txt)
            if (ht_ok(p)) return launch_one<MM, OUT_T, EPI, 256, 256, 128, 64, 2, LD_HT, 128>(p, s);
            return launch_one<MM, OUT_T, EPI, 256, 256, 128, 64, 4, LD_PIPE, 64>(p, s);
        }
    }
    //  * tall problems that cannot fill the chip with 256x256 tiles (conv GEMMs 16384 x 320 x 2880..8640, 4096 x 5120 x 640):
    //    256x128 tiles, 8 waves of 64x64, two co-resident workgroups per CU -- +10..26 % over 64x128 there, slower elsewhere
    //    (round 4, judged on the conv step: with 150..229 such tiles -- the N = 320 convs, 192 tiles on 256 CUs -- the chip is a quarter idle
    //     and 64x128 tiles win: 16384 x 320 x 2880 -0.08 ms over its 7 launches, x 640 / x 5760 / x 8640 -0.01..-0.03 each;
    //     profiles/r04_conv_tiles_in_step.txt.  From 230 tiles on -- 4096 x 1920: 240, 4096 x 5120: 640 -- the tall tile stays.)
    static const int tall_min = [] { const char* e = getenv("SDNQ_HIP_TALL_MIN_TILES"); return e ? atoi(e) : 230; }();  // tuning aid
    if (p.M >= 2048 && tiles(256, 128) >= tall_min && fits(128)) return launch_one<MM, OUT_T, EPI, 256, 128, 64, 64, 3, LD_PIPE, 64>(p, s);
    if constexpr (PP_OK<MM, OUT_T, EPI>) {
        //  * a few hundred rows against a wide N and a long K (the text stream of FLUX: 512 x 9216 / 12288 x 3072): ONE round of 256x128
        //    tiles on the fine-grained ping-pong -- 22.8 vs 28.9 us and 24.9 vs 30.7 us against 64x128 tiles (profiles/r03_gemm_tile_resweep.txt);
        //    with fewer tiles (N = 3072: 48) or more than one round the small tiles win
        if (p.M < 2048 && p.K >= 2048 && tiles(256, 128) >= 128 && tiles(256, 128) <= 256 && fits(128))
            return launch_one<MM, OUT_T, EPI, 256, 128, 64, 64, 3, LD_8P, 128>(p, s);
        //  * wide-N, few-row problems (the GEGLU projection 1024 x 10240 x 1280): 256x160 tiles cut N = 10240 into exactly 256
        //    workgroups of 8 waves (wave tile 32x160), 43 % of the LDS-fill bytes of 64x128 tiles: 27.6 -> 22.4 us
        //    (round 5: on the group-ahead overlapped ring with 128-byte stage rows -- LD_OG -- 21.3 -> 20.2 us replayed alone and
        //     -0.08 ms over the 60 launches of the SDXL step once the weights arrive from the Infinity Cache, profiles/r05_*)
        if ((p.N % 160) == 0 && tiles(256, 160) >= 192 && tiles(256, 160) <= 512 && fits(160))
            return launch_one<MM, OUT_T, EPI, 256, 160, 32, 160, 3, LD_OG, 128>(p, s);
        //  * one round of 128x128 tiles (160..256 of them) with enough K to matter (1024 x 3840 x 1280: 13.4 -> 11.1 us,
        //    4096 x 640 x 2560: 16.2 -> 15.4 us): two thirds of the LDS-fill bytes per CU of 64x128 tiles
        if (tiles(128, 128) >= 160 && tiles(128, 128) <= 256 && p.K >= 1024 && fits(128))
            return launch_one<MM, OUT_T, EPI, 128, 128, 64, 32, 3, LD_PIPE, 128>(p, s);
    }
    if constexpr (PP_OK<MM, OUT_T, EPI> && MM == SDNQ_MM_I8) {
        //  * one round of 64x80 tiles where 64x128 tiles leave CUs idle (1024 x 1280 outputs: 256 workgroups instead of 160, 25 % fewer
        //    LDS-fill bytes per CU): the K-split tile of gemm_ks.hip (round 6; profiles/r06_ksplit_*)
        if (force < 0 && ks_ok(p) && sdnq_internal_ks_preferred(p.M, p.N, p.K) && g_tile_probe) { *g_tile_probe = TileProbe{64, 80, 512}; return SDNQ_OK; }
        if (force < 0 && ks_ok(p) && sdnq_internal_ks_preferred(p.M, p.N, p.K))
            return sdnq_internal_scaled_mm_ks(p.a, p.b, p.sa, p.sb, EPI == EPI_BIAS1D ? p.bias : nullptr, p.bias_dtype, p.out, OUT_T, p.M, p.N, p.K, p.lda, p.ldb, p.ldc, s);
    }
    if (p.M > 128 && fits(128)) return launch_one<MM, OUT_T, EPI, 64, 128, 32, 32, 3, LD_DMA>(p, s);
    if constexpr (PP_OK<MM, OUT_T, EPI>) {
        //  * 65..128 rows against MANY weight rows (all cross-attention k / v projections of a UNet in one grouped launch: 77 text tokens x
        //    2048 -> 120 x 1280 channels, 315 MB of weights read once): a pure weight stream -- one 128-row tile reads every weight row once
        //    where two 64-row tiles read it twice through LDS, and half as many workgroups pay a prologue: 104 -> 88 us (3.0 -> 3.6 TB/s,
        //    tools/kv_group_lab.py; the 20 x 640 group, 100 such tiles, stays: 19.8 vs 22 us)
        if (p.M > 64 && tiles(128, 128) >= 512 && p.K >= 1024 && fits(128))
            return launch_one<MM, OUT_T, EPI, 128, 128, 64, 32, 3, LD_PIPE, 128>(p, s);
    }
    return launch_one<MM, OUT_T, EPI, 64, 64, 32, 32, 4, LD_PIPE>(p, s);
}

// development override SDNQ_HIP_TILE with a grouped launch: refuse tiles that do not divide the unit
inline bool force_tile_unfit(const GemmParams& p) {
    const int force = forced_tile_for(p);
    if (force < 0 || p.units == nullptr) return false;
    static const int bn_of[] = {256, 128, 64, 128, 256, 128, 256, 128, 128, 128, 128, 256, 160, 160, 320, 160, 320, 128, 256, 128, 256, 128, 128, 128, 128, 160, 128, 128, 128};  // (28 falls back to the heuristics on grouped launches)
    return force < (int)(sizeof(bn_of) / sizeof(int)) ? (p.unit_n % bn_of[force]) != 0 : false;
}

template <int MM, int EPI>
int dispatch_out(const GemmParams& p, int out_dtype, hipStream_t s) {
    switch (out_dtype) {
        case SDNQ_BF16: return launch_tiles<MM, SDNQ_BF16, EPI>(p, s);
        case SDNQ_F16: return launch_tiles<MM, SDNQ_F16, EPI>(p, s);
        case SDNQ_F32: return launch_tiles<MM, SDNQ_F32, EPI>(p, s);
        default: return SDNQ_ERR_DTYPE;
    }
}

template <int MM>
int dispatch_epi(const GemmParams& p, int epi, int out_dtype, hipStream_t s) {
    switch (epi) {
        case EPI_NONE: return dispatch_out<MM, EPI_NONE>(p, out_dtype, s);
        case EPI_BIAS1D: return dispatch_out<MM, EPI_BIAS1D>(p, out_dtype, s);
        case EPI_BIAS2D: return dispatch_out<MM, EPI_BIAS2D>(p, out_dtype, s);
        default: return dispatch_out<MM, EPI_LOWRANK>(p, out_dtype, s);
    }
}

int check_common(int mm_dtype, const void* a, const void* b, const float* sa, const float* sb, void* out, int out_dtype,
                 int64_t m, int64_t n, int64_t k) {
    if (!a || !b || !sa || !sb || !out) return SDNQ_ERR_NULL;
    if (mm_dtype != SDNQ_MM_I8 && mm_dtype != SDNQ_MM_FP8) return SDNQ_ERR_DTYPE;
    if (out_dtype < 0 || out_dtype > 2) return SDNQ_ERR_DTYPE;
    if (m <= 0 || n <= 0 || k <= 0 || (k % 16) != 0 || (n % 8) != 0) return SDNQ_ERR_SHAPE;
    if (((uintptr_t)a % 16) || ((uintptr_t)b % 16) || ((uintptr_t)out % 16)) return SDNQ_ERR_ALIGN;
    return SDNQ_OK;
}

}  // namespace

#ifndef SDNQ_LAB  // tools/micro/gemm_lab.hip includes this file for the kernel template only
// The pending weight-prefetch hint handed to ANOTHER translation unit's launcher (gemm_aq.hip: the activation-quantizing GEMM hosts
// prefetch workgroups exactly as launch_one does): line ranges of the hint + how many workgroups of `threads` threads to append, given the
// launch's free workgroup slots.  Consumes the hint.  Internal to the library (not part of the C ABI).
int sdnq_internal_take_prefetch(int64_t room, int threads, const uint8_t* pf_ptr[4], int pf_lines[4]) {
    for (int r = 0; r < 4; ++r) { pf_ptr[r] = nullptr; pf_lines[r] = 0; }
    if (!pf_hint_pending()) return 0;
    int64_t lines = 0;
    for (int r = 0; r < 4; ++r) {
        if (!g_pf_hint.ptr[r] || g_pf_hint.bytes[r] <= 0) continue;
        const uintptr_t a0 = (uintptr_t)g_pf_hint.ptr[r] & ~(uintptr_t)127;
        const int64_t n = (int64_t)(((uintptr_t)g_pf_hint.ptr[r] + (uintptr_t)g_pf_hint.bytes[r] + 127 - a0) / 128);
        if (n > 0x7fffffffll) continue;
        pf_ptr[r] = (const uint8_t*)a0; pf_lines[r] = (int)n;
        lines += n;
    }
    static const int pf_max = [] { const char* e = getenv("SDNQ_HIP_PREFETCH_WGS"); return e ? atoi(e) : 96; }();
    int64_t want = (lines + (int64_t)threads * 4 - 1) / ((int64_t)threads * 4);
    if (want > pf_max) want = pf_max;
    g_pf_hint = PrefetchHint{};
    if (room < 0) room = 0;
    return (int)(want < room ? want : room);
}

extern "C" void sdnq_hip_set_tile_override(int tile_id) { g_forced_tile.store(tile_id < 0 ? -1 : tile_id, std::memory_order_relaxed); }

extern "C" int sdnq_hip_prefetch_hint(const void* p0, int64_t b0, const void* p1, int64_t b1, const void* p2, int64_t b2, const void* p3, int64_t b3) {
    const void* ps[4] = {p0, p1, p2, p3};
    const int64_t bs[4] = {b0, b1, b2, b3};
    for (int r = 0; r < 4; ++r) {
        if (ps[r] && bs[r] < 0) return SDNQ_ERR_SHAPE;
        g_pf_hint.ptr[r] = bs[r] > 0 ? ps[r] : nullptr;
        g_pf_hint.bytes[r] = ps[r] ? bs[r] : 0;
    }
    g_pf_hint.device = current_device();
    return SDNQ_OK;
}

#ifdef SDNQ_TRACE
extern "C" int sdnq_hip_debug_trace(unsigned long long* host, int n_words) {
    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(g_trace), sizeof(unsigned long long) * n_words) != hipSuccess) return -7;
    void* dptr = nullptr;
    if (hipGetSymbolAddress(&dptr, HIP_SYMBOL(g_trace)) != hipSuccess) return -7;
    return hipMemset(dptr, 0, sizeof(unsigned long long) * 4096 * 8) == hipSuccess ? 0 : -7;
}
extern "C" int sdnq_hip_debug_trace_shape(int m, int n, int k) {
    const int h[4] = {m, n, k, 0};
    return hipMemcpyToSymbol(HIP_SYMBOL(g_trace_shape), h, sizeof(h)) == hipSuccess ? 0 : -7;
}
#endif

// dequantize_fp32=False, bfloat16 scales: one plain tile configuration per epilogue (a compatibility mode, not a tuned one)
template <int MM, int EPI>
int launch_lp(const GemmParams& p, hipStream_t s) {
    if (p.M > 128) return launch_one<MM, SDNQ_BF16, EPI, 64, 128, 32, 32, 3, LD_DMA, BKB, true>(p, s);
    return launch_one<MM, SDNQ_BF16, EPI, 64, 64, 32, 32, 4, LD_PIPE, BKB, true>(p, s);
}

extern "C" int sdnq_hip_scaled_mm_lp_zp(int mm_dtype, const void* a, const void* b, const float* sa, const float* sb, const void* bias,
                                        int bias_ndim, int64_t ld_bias, const void* t, const void* svd_up, int rank,
                                        const int32_t* zp_rowsum, const float* zp, void* out, int64_t m, int64_t n, int64_t k,
                                        sdnq_stream_t stream) {
    int st = check_common(mm_dtype, a, b, sa, sb, out, SDNQ_BF16, m, n, k);
    if (st != SDNQ_OK) return st;
    if ((zp_rowsum == nullptr) != (zp == nullptr)) return SDNQ_ERR_NULL;
    if (zp && bias_ndim == 2) return SDNQ_ERR_SHAPE;
    if (bias_ndim < 0 || bias_ndim > 2 || (bias_ndim != 0 && !bias)) return SDNQ_ERR_NULL;
    if ((t == nullptr) != (svd_up == nullptr)) return SDNQ_ERR_NULL;
    if (t && (rank <= 0 || bias_ndim == 2)) return SDNQ_ERR_SHAPE;
    if (t && (((uintptr_t)t % 16) || ((uintptr_t)svd_up % 16))) return SDNQ_ERR_ALIGN;
    GemmParams p{};
    p.a = (const uint8_t*)a; p.b = (const uint8_t*)b; p.sa = sa; p.sb = sb; p.bias = bias; p.out = out;
    p.lr_t = t; p.lr_up = svd_up; p.rank = rank; p.zp_rowsum = zp_rowsum; p.zp = zp;
    p.M = m; p.N = n; p.K = k; p.ld_bias = ld_bias; p.bias_ndim = bias_ndim; p.bias_dtype = SDNQ_BF16;
    hipStream_t s = (hipStream_t)stream;
#define LPD(MMV)                                                              \
    do {                                                                      \
        if (t || zp) return launch_lp<MMV, EPI_LOWRANK>(p, s);                \
        if (bias_ndim == 0) return launch_lp<MMV, EPI_NONE>(p, s);            \
        if (bias_ndim == 1) return launch_lp<MMV, EPI_BIAS1D>(p, s);          \
        return launch_lp<MMV, EPI_BIAS2D>(p, s);                              \
    } while (0)
    if (mm_dtype == SDNQ_MM_I8) LPD(SDNQ_MM_I8);
    LPD(SDNQ_MM_FP8);
#undef LPD
}

extern "C" int sdnq_hip_scaled_mm_lp_uzp(const void* a, const void* b, const float* sa, const float* sb, const void* bias,
                                         const int32_t* zp_rowsum, const float* zp, const float* a_zp, const float* w_colsum_scaled,
                                         int64_t zp_k, void* out, int64_t m, int64_t n, int64_t k, sdnq_stream_t stream) {
    int st = check_common(SDNQ_MM_I8, a, b, sa, sb, out, SDNQ_BF16, m, n, k);
    if (st != SDNQ_OK) return st;
    if (!a_zp || !w_colsum_scaled) return SDNQ_ERR_NULL;
    if ((zp_rowsum == nullptr) != (zp == nullptr)) return SDNQ_ERR_NULL;
    GemmParams p{};
    p.a = (const uint8_t*)a; p.b = (const uint8_t*)b; p.sa = sa; p.sb = sb; p.bias = bias; p.out = out;
    p.zp_rowsum = zp_rowsum; p.zp = zp; p.a_zp = a_zp; p.wcs = w_colsum_scaled; p.zp_k = zp_k;
    p.M = m; p.N = n; p.K = k; p.bias_ndim = bias ? 1 : 0; p.bias_dtype = SDNQ_BF16;
    return launch_lp<SDNQ_MM_I8, EPI_LOWRANK>(p, (hipStream_t)stream);
}

extern "C" int sdnq_hip_scaled_mm_lp_uzp_svd(const void* a, const void* b, const float* sa, const float* sb, const void* bias,
                                             const int32_t* zp_rowsum, const float* zp, const float* a_zp, const float* w_colsum_scaled,
                                             int64_t zp_k, const void* t, const void* svd_up, int rank, void* out, int64_t m, int64_t n,
                                             int64_t k, sdnq_stream_t stream) {
    int st = check_common(SDNQ_MM_I8, a, b, sa, sb, out, SDNQ_BF16, m, n, k);
    if (st != SDNQ_OK) return st;
    if (!a_zp || !w_colsum_scaled || !t || !svd_up) return SDNQ_ERR_NULL;
    if ((zp_rowsum == nullptr) != (zp == nullptr)) return SDNQ_ERR_NULL;
    if (rank <= 0) return SDNQ_ERR_SHAPE;
    if (((uintptr_t)t % 16) || ((uintptr_t)svd_up % 16)) return SDNQ_ERR_ALIGN;
    GemmParams p{};
    p.a = (const uint8_t*)a; p.b = (const uint8_t*)b; p.sa = sa; p.sb = sb; p.bias = bias; p.out = out;
    p.lr_t = t; p.lr_up = svd_up; p.rank = rank;
    p.zp_rowsum = zp_rowsum; p.zp = zp; p.a_zp = a_zp; p.wcs = w_colsum_scaled; p.zp_k = zp_k;
    p.M = m; p.N = n; p.K = k; p.bias_ndim = bias ? 1 : 0; p.bias_dtype = SDNQ_BF16;
    return launch_lp<SDNQ_MM_I8, EPI_LOWRANK>(p, (hipStream_t)stream);
}

extern "C" int sdnq_hip_scaled_mm_lp(int mm_dtype, const void* a, const void* b, const float* sa, const float* sb, const void* bias,
                                     int bias_ndim, int64_t ld_bias, const void* t, const void* svd_up, int rank, void* out,
                                     int64_t m, int64_t n, int64_t k, sdnq_stream_t stream) {
    return sdnq_hip_scaled_mm_lp_zp(mm_dtype, a, b, sa, sb, bias, bias_ndim, ld_bias, t, svd_up, rank, nullptr, nullptr, out, m, n, k, stream);
}

// tile choice of the float16 scaled matmul (the reference's compatibility mode for GPUs without int8 / fp8 matrix cores, linear_fp16.py):
// the four tile shapes the float GEMMs of launch_tiles use, by the same rules (K of the rules in BYTES)
template <int OUT_T, int EPI>
int launch_f16s(const GemmParams& p, hipStream_t s) {
    auto tiles = [&](int bm, int bn) { return ((p.M + bm - 1) / bm) * ((p.N + bn - 1) / bn); };
    if constexpr (EPI == EPI_BIAS2D) {  // (the [M][N] bias of a layer with SVD factors: never on the 256-row tiles, see launch_tiles)
        if (p.M > 128) return launch_one<MM_F16S, OUT_T, EPI, 64, 128, 32, 32, 3, LD_DMA>(p, s);
        return launch_one<MM_F16S, OUT_T, EPI, 64, 64, 32, 32, 4, LD_PIPE>(p, s);
    } else
    if (p.M > 128 && tiles(256, 256) >= 160 && p.K >= 2048) {
        if (ht_ok(p)) return launch_one<MM_F16S, OUT_T, EPI, 256, 256, 128, 64, 2, LD_HT, 128>(p, s);
        return launch_one<MM_F16S, OUT_T, EPI, 256, 256, 128, 64, 4, LD_PIPE, 64>(p, s);
    }
    if (p.M >= 2048 && tiles(256, 128) >= 230) return launch_one<MM_F16S, OUT_T, EPI, 256, 128, 64, 64, 3, LD_PIPE, 64>(p, s);
    if (p.M > 128) return launch_one<MM_F16S, OUT_T, EPI, 64, 128, 32, 32, 3, LD_DMA>(p, s);
    return launch_one<MM_F16S, OUT_T, EPI, 64, 64, 32, 32, 4, LD_PIPE>(p, s);
}

extern "C" int sdnq_hip_scaled_mm_f16(const void* a, const void* b, const float* sa, const float* sb, const void* bias, int bias_dtype, int bias_ndim,
                                      int64_t ld_bias, void* out, int out_dtype, int64_t m, int64_t n, int64_t k, sdnq_stream_t stream) {
    if (!a || !b || !sa || !sb || !out) return SDNQ_ERR_NULL;
    if (bias_ndim < 0 || bias_ndim > 2 || (bias_ndim != 0 && !bias) || (bias_ndim == 2 && ld_bias < n)) return SDNQ_ERR_SHAPE;
    if (bias_ndim == 0) bias = nullptr;
    if (out_dtype < 0 || out_dtype > 2) return SDNQ_ERR_DTYPE;
    if (bias && (bias_dtype < 0 || bias_dtype > 2)) return SDNQ_ERR_DTYPE;
    if (m <= 0 || n <= 0 || k <= 0 || (k % 8) != 0 || (n % 8) != 0) return SDNQ_ERR_SHAPE;
    if (((uintptr_t)a % 16) || ((uintptr_t)b % 16) || ((uintptr_t)out % 16)) return SDNQ_ERR_ALIGN;
    GemmParams p{};
    p.a = (const uint8_t*)a; p.b = (const uint8_t*)b; p.sa = sa; p.sb = sb; p.bias = bias; p.out = out;
    p.M = m; p.N = n; p.K = k * 2; p.bias_ndim = bias ? bias_ndim : 0; p.ld_bias = ld_bias; p.bias_dtype = bias ? bias_dtype : out_dtype;
    hipStream_t s = (hipStream_t)stream;
#define F16S_EPI(OT) (p.bias_ndim == 2 ? launch_f16s<OT, EPI_BIAS2D>(p, s) : (bias ? launch_f16s<OT, EPI_BIAS1D>(p, s) : launch_f16s<OT, EPI_NONE>(p, s)))
    if (out_dtype == SDNQ_BF16) return F16S_EPI(SDNQ_BF16);
    if (out_dtype == SDNQ_F16) return F16S_EPI(SDNQ_F16);
    return F16S_EPI(SDNQ_F32);
#undef F16S_EPI
}

// internal (used by sdnq_hip_linear_float in dequant.hip): out[M][N] = cast(x[M][K] . w[N][K]^T + bias), all of `dtype`
int sdnq_float_gemm(const void* x, const void* w, const void* bias, int dtype, void* out, int64_t m, int64_t n, int64_t k,
                    int64_t ldx, hipStream_t s, void* const* outs = nullptr, int n_outs = 0, int64_t seg_n = 0, int64_t ldc = 0) {
    const int eb = (dtype == SDNQ_F32) ? 4 : 2;
    GemmParams p{};
    p.ldc = ldc;
    p.a = (const uint8_t*)x; p.b = (const uint8_t*)w; p.bias = bias; p.out = out;
    if (outs) {  // several output tensors, one per stacked layer (sdnq_hip_linear_float_multi)
        for (int i = 0; i < n_outs; ++i) p.out_seg[i] = outs[i];
        p.seg_n = seg_n;
    }
    p.M = m; p.N = n; p.K = k * eb; p.lda = ldx * eb; p.ldb = k * eb; p.bias_ndim = bias ? 1 : 0; p.bias_dtype = dtype;
#define FG(MMV, T) (bias ? launch_tiles<MMV, T, EPI_BIAS1D>(p, s) : launch_tiles<MMV, T, EPI_NONE>(p, s))
    if (dtype == SDNQ_BF16) return FG(MM_BF16, SDNQ_BF16);
    if (dtype == SDNQ_F16) return FG(MM_F16, SDNQ_F16);
    return FG(MM_F32, SDNQ_F32);
#undef FG
}

// tile choice of the fused dequantize GEMM: wave tiles with two activation sub-tiles per weight sub-tile (the 22-VALU conversion of
// a weight fragment is shared by two MFMAs), 128-byte activation rows / 64-byte weight rows per stage
template <int MM, int OUT_T, int EPI>
int launch_tiles_w8(const GemmParams& p, hipStream_t s) {
    auto tiles = [&](int bm, int bn) { return ((p.M + bm - 1) / bm) * ((p.N + bn - 1) / bn); };
    // measured (tools/sweep_w8a16.py, profiles/r02_w8a16_sweep.txt): what matters is how many MFMAs share one converted weight
    // fragment -- 128x128 tiles of four waves with 128x32 wave tiles (one conversion per FOUR MFMAs) beat every 64-row wave tile
    // once there are enough tiles (4096^3: 178 us vs 211-224; 1024 x 10240 x 1280: 47 vs 50-54); 64x128 tiles (conversion per two
    // MFMAs) for the small problems, where tiles are few
    const int force = forced_tile_for(p);  // tuning / test aid: 0 256x128, 1 64x128, 2 128x128 (64x32 waves), 3 64x64, 4 128x128 (128x32 waves); K of the map in BYTES (2 x elements)
    const bool fit128 = p.units == nullptr || (p.unit_n % 128) == 0;
    if (force == 0 && fit128) return launch_one<MM, OUT_T, EPI, 256, 128, 64, 64, 3, LD_PIPE, 128>(p, s);
    if (force == 2 && fit128) return launch_one<MM, OUT_T, EPI, 128, 128, 64, 32, 3, LD_PIPE, 128>(p, s);
    // (round 5: the overlapped ring LD_OV on this kernel's 64x128 tile -- DMA issue and the next stage's fragment reads dealt out between
    //  the MFMAs and conversions, 3 / 4 ring slots -- compiled, bit-identical, and judged on the step: -0.09 ... +0.11 ms per problem, all
    //  problems together +0.03 / +0.11 ms of 12.67 (profiles/r05_fused_rowquant_gemm.txt item 9); not instantiated)
    if (force == 3 || (force < 0 && p.M <= 64) || !fit128) return launch_one<MM, OUT_T, EPI, 64, 64, 64, 32, 4, LD_DMA, 128>(p, s);
    // (round 3 re-sweep on the buffer-load loaders, profiles/r03_w8a16_sweep.txt: 160-240 tiles of 128x128 lose to 64x128 -- 4096 x 640 x 640
    //  12.7 vs 11.4 us, 1024 x 3840 x 1280 20.2 vs 19.2 -- from 480 tiles up they win: 4096 x 1920 x 640 17.2 vs 21.0)
    static const int t128_min = [] { const char* e = getenv("SDNQ_HIP_W8_T128_MIN"); return e ? atoi(e) : 320; }();  // tuning aid
    // (judged on the STEP, tools/tune_tiles_in_step.py + profiles/r03_tiles_in_step_dequant.txt: the GEGLU projection 1024 x 10240 x 1280 --
    //  640 tiles of 128x128 -- is 0.32 ms per step faster on 64x128 tiles, 12.75 against 13.07 ms; the 4096-row problems are indifferent:
    //  128-row wave tiles only for M >= 2048)
    if (force == 4 || (force < 0 && tiles(128, 128) >= t128_min && p.M >= 2048)) return launch_one<MM, OUT_T, EPI, 128, 128, 128, 32, 3, LD_DMA, 128>(p, s);
    // (128x64 tiles of two 128x32 waves -- the conversion shared by four MFMAs at twice the workgroups -- and a 4-deep ring for the
    //  128x128 tile measured 20-60 % slower than the above on every SDXL shape: too few waves to hide the DMA / LDS latency)
    return launch_one<MM, OUT_T, EPI, 64, 128, 64, 32, 3, LD_DMA, 128>(p, s);
}

extern "C" int sdnq_hip_linear_w8a16(const void* x, int x_dtype, const void* w, const float* scale, const float* zero_point,
                                     const void* bias, void* out, int64_t m, int64_t n, int64_t k, int64_t ldx, sdnq_stream_t stream) {
    if (!x || !w || !scale || !out) return SDNQ_ERR_NULL;
    if (x_dtype != SDNQ_BF16 && x_dtype != SDNQ_F16) return SDNQ_ERR_DTYPE;
    if (m <= 0 || n <= 0 || k <= 0 || (k % 16) != 0 || (n % 8) != 0 || ldx < k) return SDNQ_ERR_SHAPE;
    if (((uintptr_t)x % 16) || ((uintptr_t)w % 16) || ((uintptr_t)out % 16) || ((ldx * 2) % 16)) return SDNQ_ERR_ALIGN;
    GemmParams p{};
    p.a = (const uint8_t*)x; p.b = (const uint8_t*)w; p.sb = scale; p.zp = zero_point; p.bias = bias; p.out = out;
    p.M = m; p.N = n; p.K = k * 2; p.lda = ldx * 2; p.ldb = k; p.bias_ndim = bias ? 1 : 0; p.bias_dtype = x_dtype;
    hipStream_t s = (hipStream_t)stream;
#define W8(MMV, T) (bias ? launch_tiles_w8<MMV, T, EPI_BIAS1D>(p, s) : launch_tiles_w8<MMV, T, EPI_NONE>(p, s))
    if (zero_point) {  // unsigned codes
        if (x_dtype == SDNQ_BF16) return W8(MM_W8BF16U, SDNQ_BF16);
        return W8(MM_W8F16U, SDNQ_F16);
    }
    if (x_dtype == SDNQ_BF16) return W8(MM_W8BF16, SDNQ_BF16);
    return W8(MM_W8F16, SDNQ_F16);
#undef W8
}

extern "C" int sdnq_hip_linear_w8a16_grouped(const void* x, int x_dtype, const SdnqGemmUnit* units, int64_t n_units, int64_t unit_n,
                                             int has_bias, void* out, int64_t m, int64_t k, int64_t ldx, sdnq_stream_t stream) {
    if (!x || !units || !out) return SDNQ_ERR_NULL;
    if (x_dtype != SDNQ_BF16 && x_dtype != SDNQ_F16) return SDNQ_ERR_DTYPE;
    if (m <= 0 || n_units <= 0 || unit_n <= 0 || (unit_n % 64) != 0 || k <= 0 || (k % 16) != 0 || ldx < k) return SDNQ_ERR_SHAPE;
    if (((uintptr_t)x % 16) || ((uintptr_t)out % 16) || ((ldx * 2) % 16)) return SDNQ_ERR_ALIGN;
    GemmParams p{};
    p.a = (const uint8_t*)x; p.out = out; p.units = units; p.unit_n = unit_n;
    p.M = m; p.N = n_units * unit_n; p.K = k * 2; p.lda = ldx * 2; p.ldb = k; p.bias_ndim = has_bias ? 1 : 0; p.bias_dtype = x_dtype;
    hipStream_t s = (hipStream_t)stream;
#define W8(MMV, T) (has_bias ? launch_tiles_w8<MMV, T, EPI_BIAS1D>(p, s) : launch_tiles_w8<MMV, T, EPI_NONE>(p, s))
    if (x_dtype == SDNQ_BF16) return W8(MM_W8BF16, SDNQ_BF16);
    return W8(MM_W8F16, SDNQ_F16);
#undef W8
}

extern "C" int sdnq_hip_linear_float_multi(const void* x, const void* wd, const void* bias, int dtype, void* const* outs, int n_outs,
                                           int64_t seg_n, int64_t m, int64_t n, int64_t k, int64_t ldx, sdnq_stream_t stream) {
    if (!x || !wd || !outs) return SDNQ_ERR_NULL;
    if (dtype < 0 || dtype > 2) return SDNQ_ERR_DTYPE;
    if (n_outs < 1 || n_outs > 4 || seg_n <= 0 || seg_n % 8 || seg_n * n_outs != n) return SDNQ_ERR_SHAPE;
    const int eb = (dtype == SDNQ_F32) ? 4 : 2;
    if (m <= 32 || k <= 0 || ldx < k || ((k * eb) % 16) != 0) return SDNQ_ERR_SHAPE;  // few rows: the per-layer path
    if (((uintptr_t)x % 16) || ((uintptr_t)wd % 16) || ((ldx * eb) % 16)) return SDNQ_ERR_ALIGN;
    for (int i = 0; i < n_outs; ++i)
        if (!outs[i] || (uintptr_t)outs[i] % 16) return outs[i] ? SDNQ_ERR_ALIGN : SDNQ_ERR_NULL;
    return sdnq_float_gemm(x, wd, bias, dtype, outs[0], m, n, k, ldx, (hipStream_t)stream, outs, n_outs, seg_n);
}

extern "C" int sdnq_hip_scaled_mm(int mm_dtype, const void* a, const void* b, const float* sa, const float* sb,
                                  const void* bias, int bias_dtype, int bias_ndim, int64_t ld_bias, void* out,
                                  int out_dtype, int64_t m, int64_t n, int64_t k, sdnq_stream_t stream) {
    int st = check_common(mm_dtype, a, b, sa, sb, out, out_dtype, m, n, k);
    if (st != SDNQ_OK) return st;
    if (bias_ndim < 0 || bias_ndim > 2) return SDNQ_ERR_SHAPE;
    if (bias_ndim != 0 && !bias) return SDNQ_ERR_NULL;
    if (bias_ndim == 0) { bias = nullptr; bias_dtype = out_dtype; }
    if (bias_dtype < 0 || bias_dtype > 2) return SDNQ_ERR_DTYPE;
    if (bias_ndim == 2 && ld_bias < n) return SDNQ_ERR_SHAPE;
    GemmParams p{};
    p.a = (const uint8_t*)a; p.b = (const uint8_t*)b; p.sa = sa; p.sb = sb; p.bias = bias; p.out = out;
    p.M = m; p.N = n; p.K = k; p.ld_bias = ld_bias; p.bias_ndim = bias_ndim; p.bias_dtype = bias_dtype;
    hipStream_t s = (hipStream_t)stream;
    if (mm_dtype == SDNQ_MM_I8) return dispatch_epi<SDNQ_MM_I8>(p, bias_ndim, out_dtype, s);
    return dispatch_epi<SDNQ_MM_FP8>(p, bias_ndim, out_dtype, s);
}

extern "C" int sdnq_hip_scaled_mm_tile(int mm_dtype, int out_dtype, int has_bias, int64_t m, int64_t n, int64_t k, int* bm, int* bn, int* threads,
                                       int64_t* workgroups) {
    if (mm_dtype != SDNQ_MM_I8 && mm_dtype != SDNQ_MM_FP8) return SDNQ_ERR_DTYPE;
    if (out_dtype < 0 || out_dtype > 2) return SDNQ_ERR_DTYPE;
    if (m <= 0 || n <= 0 || k <= 0 || (k % 16) != 0 || (n % 8) != 0) return SDNQ_ERR_SHAPE;
    GemmParams p{};
    p.M = m; p.N = n; p.K = k; p.bias_ndim = has_bias ? 1 : 0; p.bias_dtype = out_dtype;
    TileProbe probe{0, 0, 0};
    g_tile_probe = &probe;  // (no pointer of `p` is read on the host: the dry run returns in front of the launch)
    const int st = mm_dtype == SDNQ_MM_I8 ? dispatch_epi<SDNQ_MM_I8>(p, p.bias_ndim, out_dtype, nullptr) : dispatch_epi<SDNQ_MM_FP8>(p, p.bias_ndim, out_dtype, nullptr);
    g_tile_probe = nullptr;
    if (st != SDNQ_OK) return st;
    if (probe.bm == 0) return SDNQ_ERR_UNSUPPORTED;
    if (bm) *bm = probe.bm;
    if (bn) *bn = probe.bn;
    if (threads) *threads = probe.threads;
    if (workgroups) *workgroups = ((m + probe.bm - 1) / probe.bm) * ((n + probe.bn - 1) / probe.bn);
    return SDNQ_OK;
}

extern "C" int sdnq_hip_scaled_mm_multi(int mm_dtype, const void* a, const void* b, const float* sa, const float* sb, const void* bias,
                                        int bias_dtype, void* const* outs, int n_outs, int64_t seg_n, int out_dtype, int64_t m,
                                        int64_t n, int64_t k, sdnq_stream_t stream) {
    if (!outs || n_outs < 1 || n_outs > 4) return SDNQ_ERR_SHAPE;
    for (int i = 0; i < n_outs; ++i)
        if (!outs[i] || (uintptr_t)outs[i] % 16) return outs[i] ? SDNQ_ERR_ALIGN : SDNQ_ERR_NULL;
    if (seg_n <= 0 || seg_n % 8 || seg_n * n_outs != n) return SDNQ_ERR_SHAPE;
    int st = check_common(mm_dtype, a, b, sa, sb, outs[0], out_dtype, m, n, k);
    if (st != SDNQ_OK) return st;
    if (bias && (bias_dtype < 0 || bias_dtype > 2)) return SDNQ_ERR_DTYPE;
    GemmParams p{};
    p.a = (const uint8_t*)a; p.b = (const uint8_t*)b; p.sa = sa; p.sb = sb; p.bias = bias; p.out = outs[0];
    for (int i = 0; i < n_outs; ++i) p.out_seg[i] = outs[i];
    p.seg_n = seg_n;
    p.M = m; p.N = n; p.K = k; p.bias_ndim = bias ? 1 : 0; p.bias_dtype = bias ? bias_dtype : out_dtype;
    hipStream_t s = (hipStream_t)stream;
    if (mm_dtype == SDNQ_MM_I8) return dispatch_epi<SDNQ_MM_I8>(p, p.bias_ndim, out_dtype, s);
    return dispatch_epi<SDNQ_MM_FP8>(p, p.bias_ndim, out_dtype, s);
}

extern "C" int sdnq_hip_scaled_mm_grouped(int mm_dtype, const void* a, const float* sa, const SdnqGemmUnit* units, int64_t n_units,
                                          int64_t unit_n, int bias_dtype, void* out, int out_dtype, int64_t m, int64_t k,
                                          sdnq_stream_t stream) {
    if (!units) return SDNQ_ERR_NULL;
    if (n_units <= 0 || unit_n <= 0 || (unit_n % 64) != 0) return SDNQ_ERR_SHAPE;  // the smallest tile is 64 channels wide
    // `units` is device memory: its b / sb pointers cannot be checked here; check_common sees a / sa / out and the shapes
    int st = check_common(mm_dtype, a, a, sa, sa, out, out_dtype, m, n_units * unit_n, k);
    if (st != SDNQ_OK) return st;
    if (bias_dtype > 2) return SDNQ_ERR_DTYPE;
    GemmParams p{};
    p.a = (const uint8_t*)a; p.sa = sa; p.out = out; p.units = units; p.unit_n = unit_n;
    p.M = m; p.N = n_units * unit_n; p.K = k; p.bias_ndim = bias_dtype >= 0 ? 1 : 0; p.bias_dtype = bias_dtype >= 0 ? bias_dtype : out_dtype;
    hipStream_t s = (hipStream_t)stream;
    if (force_tile_unfit(p)) return SDNQ_ERR_SHAPE;
    if (mm_dtype == SDNQ_MM_I8) return dispatch_epi<SDNQ_MM_I8>(p, p.bias_ndim, out_dtype, s);
    return dispatch_epi<SDNQ_MM_FP8>(p, p.bias_ndim, out_dtype, s);
}

extern "C" int sdnq_hip_scaled_mm_nchw(int mm_dtype, const void* a, const void* b, const float* sa, const float* sb, const void* bias,
                                       int bias_dtype, void* out, int out_dtype, int64_t m, int64_t n, int64_t k, int64_t hw,
                                       sdnq_stream_t stream) {
    int st = check_common(mm_dtype, a, b, sa, sb, out, out_dtype, m, n, k);
    if (st != SDNQ_OK) return st;
    if (out_dtype == SDNQ_F32) return SDNQ_ERR_UNSUPPORTED;
    if (hw <= 0 || (hw % 8) != 0 || (m % hw) != 0) return SDNQ_ERR_SHAPE;
    if (!bias) bias_dtype = out_dtype;
    if (bias_dtype < 0 || bias_dtype > 2) return SDNQ_ERR_DTYPE;
    GemmParams p{};
    p.a = (const uint8_t*)a; p.b = (const uint8_t*)b; p.sa = sa; p.sb = sb; p.bias = bias; p.out = out;
    p.M = m; p.N = n; p.K = k; p.bias_ndim = bias ? 1 : 0; p.bias_dtype = bias_dtype; p.out_hw = hw;
    hipStream_t s = (hipStream_t)stream;
    if (mm_dtype == SDNQ_MM_I8) return dispatch_epi<SDNQ_MM_I8>(p, p.bias_ndim, out_dtype, s);
    return dispatch_epi<SDNQ_MM_FP8>(p, p.bias_ndim, out_dtype, s);
}

// sdnq_hip_scaled_mm / sdnq_hip_scaled_mm_nchw on VIEWS: a is [M][lda] with K valid columns, out is [M][ldc] with N valid columns
// (hw == 0) or the channel slice [B][ldc channels][hw] starting at `out` (hw > 0).  Used for the per-group matmuls of grouped convs
// (conv_int8.py:73-79, conv_fp8.py:56-60: int_mm / scaled_mm per group on column slices of the quantized unfolded input).
extern "C" int sdnq_hip_scaled_mm_strided(int mm_dtype, const void* a, int64_t lda, const void* b, const float* sa, const float* sb,
                                          const void* bias, int bias_dtype, void* out, int64_t ldc, int out_dtype, int64_t m, int64_t n,
                                          int64_t k, int64_t hw, sdnq_stream_t stream) {
    int st = check_common(mm_dtype, a, b, sa, sb, out, out_dtype, m, n, k);
    if (st != SDNQ_OK) return st;
    if (lda < k || (lda % 16) != 0 || ldc < n) return SDNQ_ERR_SHAPE;
    if (hw < 0 || (hw > 0 && ((hw % 8) != 0 || (m % hw) != 0))) return SDNQ_ERR_SHAPE;
    if (hw > 0 && out_dtype == SDNQ_F32) return SDNQ_ERR_UNSUPPORTED;
    if (hw == 0 && (ldc * (out_dtype == SDNQ_F32 ? 4 : 2)) % 16 != 0) return SDNQ_ERR_ALIGN;
    if (!bias) bias_dtype = out_dtype;
    if (bias_dtype < 0 || bias_dtype > 2) return SDNQ_ERR_DTYPE;
    GemmParams p{};
    p.a = (const uint8_t*)a; p.b = (const uint8_t*)b; p.sa = sa; p.sb = sb; p.bias = bias; p.out = out;
    p.M = m; p.N = n; p.K = k; p.lda = lda; p.ldc = ldc; p.bias_ndim = bias ? 1 : 0; p.bias_dtype = bias_dtype; p.out_hw = hw;
    hipStream_t s = (hipStream_t)stream;
    if (mm_dtype == SDNQ_MM_I8) return dispatch_epi<SDNQ_MM_I8>(p, p.bias_ndim, out_dtype, s);
    return dispatch_epi<SDNQ_MM_FP8>(p, p.bias_ndim, out_dtype, s);
}

// sdnq_hip_scaled_mm_lowrank on VIEWS (one group of a grouped conv whose epilogue carries zero-point / activation-zero-point terms,
// conv_int8.py:65-79, conv_uint8.py:58-79): a is [M][lda] with K valid columns, out [M][ldc] with N valid columns; zp_k = the K of the
// reference's K * (xzp * wzp) term (the WHOLE unfolded row, all groups), 0: k; negative: -K with the conv forwards' rounding order.
extern "C" int sdnq_hip_scaled_mm_lowrank_strided(int mm_dtype, const void* a, int64_t lda, const void* b, const float* sa, const float* sb,
                                                  const void* bias, int bias_dtype, const int32_t* zp_rowsum, const float* zp, const float* a_zp,
                                                  const float* w_colsum_scaled, int64_t zp_k, void* out, int64_t ldc, int out_dtype, int64_t m,
                                                  int64_t n, int64_t k, sdnq_stream_t stream) {
    int st = check_common(mm_dtype, a, b, sa, sb, out, out_dtype, m, n, k);
    if (st != SDNQ_OK) return st;
    if (lda < k || (lda % 16) != 0 || ldc < n) return SDNQ_ERR_SHAPE;
    if ((ldc * (out_dtype == SDNQ_F32 ? 4 : 2)) % 16 != 0) return SDNQ_ERR_ALIGN;
    if ((zp_rowsum == nullptr) != (zp == nullptr)) return SDNQ_ERR_NULL;
    if ((a_zp == nullptr) != (w_colsum_scaled == nullptr)) return SDNQ_ERR_NULL;
    const int bt = bias ? bias_dtype : out_dtype;
    if (bt < 0 || bt > 2) return SDNQ_ERR_DTYPE;
    GemmParams p{};
    p.a = (const uint8_t*)a; p.b = (const uint8_t*)b; p.sa = sa; p.sb = sb; p.bias = bias; p.out = out;
    p.zp_rowsum = zp_rowsum; p.zp = zp; p.a_zp = a_zp; p.wcs = w_colsum_scaled; p.zp_k = zp_k;
    p.M = m; p.N = n; p.K = k; p.lda = lda; p.ldc = ldc; p.bias_ndim = bias ? 1 : 0; p.bias_dtype = bt;
    hipStream_t s = (hipStream_t)stream;
    if (mm_dtype == SDNQ_MM_I8) return dispatch_epi<SDNQ_MM_I8>(p, EPI_LOWRANK, out_dtype, s);
    return dispatch_epi<SDNQ_MM_FP8>(p, EPI_LOWRANK, out_dtype, s);
}

extern "C" int sdnq_hip_scaled_mm_lowrank(int mm_dtype, const void* a, const void* b, const float* sa, const float* sb,
                                          const void* bias, int bias_dtype, const void* t, const void* svd_up,
                                          int svd_dtype, int rank, const int32_t* zp_rowsum, const float* zp, const float* a_zp,
                                          const float* w_colsum_scaled, void* out, int out_dtype, int64_t m, int64_t n, int64_t k,
                                          sdnq_stream_t stream) {
    int st = check_common(mm_dtype, a, b, sa, sb, out, out_dtype, m, n, k);
    if (st != SDNQ_OK) return st;
    if ((t == nullptr) != (svd_up == nullptr)) return SDNQ_ERR_NULL;
    if ((zp_rowsum == nullptr) != (zp == nullptr)) return SDNQ_ERR_NULL;
    if ((a_zp == nullptr) != (w_colsum_scaled == nullptr)) return SDNQ_ERR_NULL;
    if (t && (rank <= 0 || rank > 1024)) return SDNQ_ERR_SHAPE;
    // the [M][N] bias of the reference lives in the svd dtype (addmm in svd_down.dtype, linear_int8.py:60);
    // a 1-D bias is cast to it first, so bias/t/up share one element type here.
    int bt = t ? svd_dtype : (bias ? bias_dtype : out_dtype);
    if (bias && t && bias_dtype != svd_dtype) return SDNQ_ERR_DTYPE;
    if (bt < 0 || bt > 2) return SDNQ_ERR_DTYPE;
    GemmParams p{};
    p.a = (const uint8_t*)a; p.b = (const uint8_t*)b; p.sa = sa; p.sb = sb; p.bias = bias; p.out = out;
    p.lr_t = t; p.lr_up = svd_up; p.rank = rank; p.zp_rowsum = zp_rowsum; p.zp = zp; p.a_zp = a_zp; p.wcs = w_colsum_scaled;
    p.M = m; p.N = n; p.K = k; p.bias_ndim = bias ? 1 : 0; p.bias_dtype = bt;
    hipStream_t s = (hipStream_t)stream;
    if (mm_dtype == SDNQ_MM_I8) return dispatch_epi<SDNQ_MM_I8>(p, EPI_LOWRANK, out_dtype, s);
    return dispatch_epi<SDNQ_MM_FP8>(p, EPI_LOWRANK, out_dtype, s);
}
#endif  // SDNQ_LAB
