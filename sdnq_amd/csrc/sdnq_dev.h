// Device-side helpers shared by the gfx950 kernels (wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/sdnq_hip.h"

typedef int v2i __attribute__((ext_vector_type(2)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));

// ---- vector-memory instructions beside MFMAs (round 3, tools/micro/dma_mfma_lab.hip, profiles/r03_dma_beside_mfma.txt) ---------------
// A global / flat memory instruction whose address is a 64-bit VGPR pair (`global_load_dwordx4 v, v[a:b], off`, the same for stores and
// for global_load_lds) waits ~1000-2000 cycles at issue while the OTHER wave of its SIMD streams MFMAs (65-86 cycles alone); the forms
// with a scalar base and a 32-bit per-lane offset -- `global_load v, v_off, s[base]` and every buffer instruction -- are not affected.
// hipcc picks the 64-bit form for almost every pointer expression, so kernels whose waves share a SIMD with matrix work address
// memory through buffer descriptors: wave-uniform base in SGPRs, 32-bit byte offset per lane, optional scalar offset.
// The descriptor type and its builtins exist in the DEVICE pass only; the host pass also parses kernel bodies, and a generic lambda
// that uses them there makes it drop the kernel's launch stub without a diagnostic -- so the host pass sees inert stand-ins.
typedef __attribute__((address_space(3))) void* sdnq_lds_ptr_t;
#if defined(__HIP_DEVICE_COMPILE__)
#define SDNQ_MAKE_RSRC(ptr) __builtin_amdgcn_make_buffer_rsrc((void*)(ptr), 0, 0x7fffffff, 0x00020000)
// the same with a byte extent: a lane whose offset reaches past it reads zeros instead of memory that may not be mapped
#define SDNQ_MAKE_RSRC_N(ptr, nbytes) __builtin_amdgcn_make_buffer_rsrc((void*)(ptr), 0, (int)(nbytes), 0x00020000)
#define SDNQ_DMA16(rs, dst, voff, soff) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (sdnq_lds_ptr_t)(dst), 16, voff, soff, 0, 0)
#define SDNQ_DMA4(rs, dst, voff, soff) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (sdnq_lds_ptr_t)(dst), 4, voff, soff, 0, 0)
#define SDNQ_BUF_LOAD16(rs, voff, soff) __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0))
#define SDNQ_BUF_LOAD8(rs, voff, soff) __builtin_bit_cast(v2i, __builtin_amdgcn_raw_buffer_load_b64(rs, voff, soff, 0))
#define SDNQ_BUF_LOAD4(rs, voff, soff) ((int)__builtin_amdgcn_raw_buffer_load_b32(rs, voff, soff, 0))
#define SDNQ_BUF_LOAD2(rs, voff, soff) ((int)__builtin_amdgcn_raw_buffer_load_b16(rs, voff, soff, 0))
#define SDNQ_BUF_STORE16(rs, v, voff, soff) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, v), rs, voff, soff, 0)
#define SDNQ_BUF_STORE16_NT(rs, v, voff, soff) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, v), rs, voff, soff, 2)
#else
#define SDNQ_MAKE_RSRC(ptr) ((const void*)(ptr))
#define SDNQ_MAKE_RSRC_N(ptr, nbytes) ((void)(nbytes), (const void*)(ptr))
#define SDNQ_DMA16(rs, dst, voff, soff) ((void)(rs), (void)(dst), (void)(voff), (void)(soff))
#define SDNQ_DMA4(rs, dst, voff, soff) ((void)(rs), (void)(dst), (void)(voff), (void)(soff))
#define SDNQ_BUF_LOAD16(rs, voff, soff) ((void)(rs), (void)(voff), (void)(soff), (v4i){0, 0, 0, 0})
#define SDNQ_BUF_LOAD8(rs, voff, soff) ((void)(rs), (void)(voff), (void)(soff), (v2i){0, 0})
#define SDNQ_BUF_LOAD4(rs, voff, soff) ((void)(rs), (void)(voff), (void)(soff), 0)
#define SDNQ_BUF_LOAD2(rs, voff, soff) ((void)(rs), (void)(voff), (void)(soff), 0)
#define SDNQ_BUF_STORE16(rs, v, voff, soff) ((void)(rs), (void)(v), (void)(voff), (void)(soff))
#define SDNQ_BUF_STORE16_NT(rs, v, voff, soff) ((void)(rs), (void)(v), (void)(voff), (void)(soff))
#endif

typedef float v16f __attribute__((ext_vector_type(16)));
typedef short v8s __attribute__((ext_vector_type(8)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
typedef __bf16 v2bf __attribute__((ext_vector_type(2)));
typedef _Float16 v2h __attribute__((ext_vector_type(2)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef unsigned int u32;

#define SDNQ_WAVE 64

// ---- float format conversions -----------------------------------------------------------------
__device__ __forceinline__ float bf16_bits_to_f32(uint16_t b) { return __uint_as_float(((u32)b) << 16); }
__device__ __forceinline__ uint16_t f32_to_bf16_bits(float f) {  // RNE, v_cvt_pk_bf16_f32 on gfx950
    __bf16 h = (__bf16)f;
    return __builtin_bit_cast(uint16_t, h);
}
__device__ __forceinline__ float f16_bits_to_f32(uint16_t b) { return (float)__builtin_bit_cast(_Float16, b); }
__device__ __forceinline__ uint16_t f32_to_f16_bits(float f) {
    _Float16 h = (_Float16)f;
    return __builtin_bit_cast(uint16_t, h);
}

// element type traits: T_ID is SdnqFloat
template <int T_ID> struct FT;
template <> struct FT<SDNQ_F32> {
    typedef float store_t;
    static constexpr int bytes = 4;
    static __device__ __forceinline__ float load(const void* p, int64_t i) { return ((const float*)p)[i]; }
    static __device__ __forceinline__ void store(void* p, int64_t i, float v) { ((float*)p)[i] = v; }
    static __device__ __forceinline__ float round(float v) { return v; }
    static __device__ __forceinline__ uint32_t bits(float v) { return __float_as_uint(v); }
};
template <> struct FT<SDNQ_BF16> {
    typedef uint16_t store_t;
    static constexpr int bytes = 2;
    static __device__ __forceinline__ float load(const void* p, int64_t i) { return bf16_bits_to_f32(((const uint16_t*)p)[i]); }
    static __device__ __forceinline__ void store(void* p, int64_t i, float v) { ((uint16_t*)p)[i] = f32_to_bf16_bits(v); }
    static __device__ __forceinline__ float round(float v) { return bf16_bits_to_f32(f32_to_bf16_bits(v)); }
    static __device__ __forceinline__ uint16_t bits(float v) { return f32_to_bf16_bits(v); }
};
template <> struct FT<SDNQ_F16> {
    typedef uint16_t store_t;
    static constexpr int bytes = 2;
    static __device__ __forceinline__ float load(const void* p, int64_t i) { return f16_bits_to_f32(((const uint16_t*)p)[i]); }
    static __device__ __forceinline__ void store(void* p, int64_t i, float v) { ((uint16_t*)p)[i] = f32_to_f16_bits(v); }
    static __device__ __forceinline__ float round(float v) { return f16_bits_to_f32(f32_to_f16_bits(v)); }
    static __device__ __forceinline__ uint16_t bits(float v) { return f32_to_f16_bits(v); }
};

// round to a dtype chosen at run time (wave-uniform `dt`)
__device__ __forceinline__ float round_rt(float v, int dt) {
    return dt == SDNQ_F32 ? v : (dt == SDNQ_BF16 ? FT<SDNQ_BF16>::round(v) : FT<SDNQ_F16>::round(v));
}

// two floats -> one dword of 16-bit elements (low half = a), round-to-nearest-even: ONE v_cvt_pk_bf16_f32 on gfx950 when written as a
// vector conversion (converting the halves separately and OR-ing them costs a convert per element plus the merge)
template <int T_ID> __device__ __forceinline__ u32 pack2(float a, float b) {
    if constexpr (T_ID == SDNQ_BF16) {
        const v2bf p = {(__bf16)a, (__bf16)b};
        return __builtin_bit_cast(u32, p);
    } else {  // f16: element-wise converts (the packed vector conversion differed from RNE on one value in 10^5: kept scalar)
        return (u32)f32_to_f16_bits(a) | ((u32)f32_to_f16_bits(b) << 16);
    }
}

// 16-byte vector of elements -> 8 (16-bit) or 4 (f32) floats
template <int T_ID> struct Vec16;
template <> struct Vec16<SDNQ_F32> {
    static constexpr int n = 4;
    static __device__ __forceinline__ void unpack(const uint4& v, float* f) {
        f[0] = __uint_as_float(v.x); f[1] = __uint_as_float(v.y); f[2] = __uint_as_float(v.z); f[3] = __uint_as_float(v.w);
    }
    static __device__ __forceinline__ uint4 pack(const float* f) {
        return make_uint4(__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3]));
    }
};
template <> struct Vec16<SDNQ_BF16> {
    static constexpr int n = 8;
    static __device__ __forceinline__ void unpack(const uint4& v, float* f) {
        const u32 w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f[2 * i] = __uint_as_float(w[i] << 16);
            f[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
        }
    }
    static __device__ __forceinline__ uint4 pack(const float* f) {
        u32 w[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) w[i] = pack2<SDNQ_BF16>(f[2 * i], f[2 * i + 1]);
        return make_uint4(w[0], w[1], w[2], w[3]);
    }
};
template <> struct Vec16<SDNQ_F16> {
    static constexpr int n = 8;
    static __device__ __forceinline__ void unpack(const uint4& v, float* f) {
        const u32 w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f[2 * i] = f16_bits_to_f32((uint16_t)(w[i] & 0xffffu));
            f[2 * i + 1] = f16_bits_to_f32((uint16_t)(w[i] >> 16));
        }
    }
    static __device__ __forceinline__ uint4 pack(const float* f) {
        u32 w[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) w[i] = pack2<SDNQ_F16>(f[2 * i], f[2 * i + 1]);
        return make_uint4(w[0], w[1], w[2], w[3]);
    }
};

// x / scale for MANY x and one scale, correctly rounded, in 3 VALU instead of the ~10 of an IEEE division sequence (row quantization of
// long rows is VALU-bound, not HBM-bound: 64 quotients per lane at K = 15360).  Markstein's correction: with y = RN(1 / s) (one IEEE
// division per row), q0 = RN(x y), r = x - s q0 (exact in one fma), RN(q0 + r y) = RN(x / s) -- provided nothing over- or underflows
// and the significand of s is not all ones (then RN(1 / s) is too coarse).  `fast` (wave-uniform) says the row's scale is inside
// 2^-60 .. 2^60 with an ordinary significand; otherwise the IEEE division is used.  |x| <= 127.5 s by construction, so q0 is at most
// 128 in magnitude and r, r y are far from overflow; quotients so small that they underflow round to 0 either way.  Checked against
// IEEE division on 2 x 10^7 random and near-half-integer cases (none differ) and by the bit-exact row quantization tests.
struct RowDiv {
    float scale, rcp;
    bool fast;
    __device__ __forceinline__ void set(float s) {
        scale = s;
        const u32 b = __float_as_uint(s);
        const int e = (int)((b >> 23) & 0xffu);
        fast = e > 127 - 60 && e < 127 + 60 && (b & 0x7fffffu) != 0x7fffffu;
        rcp = 1.0f / s;
    }
    __device__ __forceinline__ float fastdiv(float x) const {
        const float q0 = x * rcp;
        return fmaf(fmaf(-scale, q0, x), rcp, q0);
    }
};

// a 64-bit integer division is ~150 instructions on this ISA, a 32-bit one ~30 -- and the index quotients of these kernels (heads, rows,
// groups) fit 32 bits in every call but the > 2^31-element ones; the attention prepare kernel ran four of them per thread around 60
// instructions of work, the weight dequantizer three around 16 weights
__device__ __forceinline__ void divmod(int64_t a, int64_t b, int64_t& q, int64_t& r) {
    if ((((uint64_t)a | (uint64_t)b) >> 32) == 0) {
        const uint32_t x = (uint32_t)a, y = (uint32_t)b;
        q = x / y; r = x % y;
    } else {
        q = a / b; r = a % b;
    }
}
typedef float pv2f __attribute__((ext_vector_type(2)));
// The quantizer launches of a bs = 1 step are latency chains of ONE wave per SIMD (~10 cycles per dependent vector instruction measured: the
// per-token quantization of a 128-query attention tile cost 2 us at 13 instructions per element), so the per-element work is packed fp32 math:
//   x / scale, correctly rounded, for two values in three packed instructions (RowDiv::fastdiv, sdnq_dev.h) ...
__device__ __forceinline__ pv2f fastdiv2(pv2f x, const RowDiv& d) {
    const pv2f r2 = {d.rcp, d.rcp}, ns = {-d.scale, -d.scale};
    const pv2f q0 = x * r2;
    return __builtin_elementwise_fma(__builtin_elementwise_fma(ns, q0, x), r2, q0);
}
//   ... and rint + int8 packing without a conversion: |q| <= 128, so q + 1.5 * 2^23 is exact up to the round-to-nearest-even that rint would do
//   and leaves rint(q) as a two's complement byte in the low mantissa bits; two byte-permutes and one shift-or per four values
__device__ __forceinline__ u32 pack4_rne_i8(pv2f a, pv2f b) {
    const pv2f magic = {12582912.0f, 12582912.0f};
    a += magic;
    b += magic;
    const u32 t0 = __builtin_amdgcn_perm(__float_as_uint(a[1]), __float_as_uint(a[0]), 0x0c0c0400u);
    const u32 t1 = __builtin_amdgcn_perm(__float_as_uint(b[1]), __float_as_uint(b[0]), 0x0c0c0400u);
    return t0 | (t1 << 16);
}

// ---- fp8 e4m3fn (OCP) -------------------------------------------------------------------------
// float -> e4m3fn, round-to-nearest-even, input already clamped to [-448, 448] (quant_utils.py:298),
// so no overflow handling is needed; matches torch's .to(torch.float8_e4m3fn) on that range.
__device__ __forceinline__ uint8_t f32_to_e4m3fn(float f) {
    u32 u = __float_as_uint(f);
    const u32 sign = (u >> 24) & 0x80u;
    u &= 0x7fffffffu;
    if (u >= 0x43f00000u) return (uint8_t)(sign | 0x7f);  // >= 480 or NaN -> NaN (torch does not saturate)
    u32 r;
    if (u < 0x3c800000u) {  // |f| < 2^-6: e4m3 subnormal, quantum 2^-9 (x512 is exact, rint is RNE)
        r = (u32)__builtin_rintf(__uint_as_float(u) * 512.0f);
    } else {
        u += 0x7ffffu + ((u >> 20) & 1u);  // RNE at bit 20; a mantissa carry bumps the exponent
        r = (u >> 20) - (120u << 3);       // rebias 127 -> 7
    }
    return (uint8_t)(sign | r);
}
// four floats ALREADY CLAMPED to [-448, 448] and NaN-free -> four e4m3fn bytes (lowest byte = a): gfx950's v_cvt_pk_fp8_f32 (OCP
// e4m3fn, RNE) equals f32_to_e4m3fn on every float of that range (tools/micro/fp8_cvt_probe.hip: all 2.3e9 of them), 2 instructions
// instead of ~50
__device__ __forceinline__ u32 pack4_e4m3fn_clamped(float a, float b, float c, float d) {
    int w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
    return (u32)w;
}
__device__ __forceinline__ uint8_t f32_to_e4m3fn_clamped(float a) {  // one value, same precondition
    return (uint8_t)(__builtin_amdgcn_cvt_pk_fp8_f32(a, 0.0f, 0, false) & 0xff);
}
__device__ __forceinline__ float e4m3fn_to_f32(uint8_t b) {
    const u32 sign = ((u32)b & 0x80u) << 24;
    const u32 e = (b >> 3) & 0xfu, m = b & 7u;
    float v;
    if (e == 0) v = (float)m * 0.001953125f;                      // m * 2^-9
    else if (e == 15 && m == 7) v = __uint_as_float(0x7fc00000u);  // NaN
    else v = __uint_as_float(((e + 120u) << 23) | (m << 20));
    return __uint_as_float(__float_as_uint(v) | sign);
}
__device__ __forceinline__ float e5m2_to_f32(uint8_t b) {  // == fp16 with the low byte zero
    return f16_bits_to_f32((uint16_t)((u32)b << 8));
}

// ---- reductions -------------------------------------------------------------------------------
// All-lane reductions of a wave as xor butterflies.  The exchanges for lane ^ 1, 2, 4, 8 are DPP row operations of the VALU,
// lane ^ 16 is ds_swizzle (bit mode), lane ^ 32 one v_permlane32_swap: no ds_bpermute round trips through the LDS crossbar
// (six of them per reduction made up a noticeable part of the ~5 us a row-quantization launch lives).
__device__ __forceinline__ int lane_xor_i32(int x, const int mask) {  // value of x in lane (lane ^ mask); mask is a constant power of two <= 16
    switch (mask) {
        case 1: return __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xf, 0xf, false);   // quad_perm [1,0,3,2]
        case 2: return __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xf, 0xf, false);   // quad_perm [2,3,0,1]
        case 4:  // lane ^ 7 (row_half_mirror) then ^ 3 (quad_perm [3,2,1,0])
            return __builtin_amdgcn_update_dpp(0, __builtin_amdgcn_update_dpp(0, x, 0x141, 0xf, 0xf, false), 0x1B, 0xf, 0xf, false);
        case 8: return __builtin_amdgcn_update_dpp(0, x, 0x128, 0xf, 0xf, false);  // row_ror:8
        default: return __builtin_amdgcn_ds_swizzle(x, 0x401F);                    // 16: bit mode, xor 0x10
    }
}
template <typename OP>
__device__ __forceinline__ int wave_allreduce_i32(int x, OP op) {
#pragma unroll
    for (int m = 1; m <= 16; m <<= 1) x = op(x, lane_xor_i32(x, m));
    const auto sw = __builtin_amdgcn_permlane32_swap((u32)x, (u32)x, false, false);  // {own half's value, other half's value}
    return op((int)sw[0], (int)sw[1]);
}
__device__ __forceinline__ float wave_max(float v) {
    return __int_as_float(wave_allreduce_i32(__float_as_int(v), [](int a, int b) { return __float_as_int(fmaxf(__int_as_float(a), __int_as_float(b))); }));
}
__device__ __forceinline__ float wave_min(float v) {
    return __int_as_float(wave_allreduce_i32(__float_as_int(v), [](int a, int b) { return __float_as_int(fminf(__int_as_float(a), __int_as_float(b))); }));
}
__device__ __forceinline__ int wave_sum_i32(int v) {
    return wave_allreduce_i32(v, [](int a, int b) { return a + b; });
}

// Host-side launch helpers implemented per translation unit.
#define SDNQ_CHECK_LAUNCH()                              \
    do {                                                 \
        if (hipGetLastError() != hipSuccess) return SDNQ_ERR_LAUNCH; \
    } while (0)

// Kernel arguments fetched in ONE batch at kernel entry.  Left to itself the compiler issues the s_load of each argument (or field of a
// by-value parameter struct) where it is first used, so a prologue waits for several DEPENDENT round trips to a scalar cache that is cold
// at every launch of the bs = 1 steps (~600-800 cycles each; the GEMM: entry -> first LDS-DMA 3176 -> 1451 cycles, SDXL step 8.43 ->
// 8.15 ms).  An empty asm with the values as scalar inputs makes them live here: the loads go out together behind one s_waitcnt.
// (device pass only; at most 30 operands per statement)
#define SDNQ_KA1(a) "s"(a)
#define SDNQ_KERNARGS_NOW(...) asm volatile("" ::__VA_ARGS__)
