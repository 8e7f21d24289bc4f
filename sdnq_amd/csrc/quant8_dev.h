// The per-element arithmetic of the activation row quantizers (int8 / fp8 codes of 8 consecutive elements of a row whose scale is known),
// shared by the row-quantization kernels (rowquant.hip) and the GEMM that quantizes its own activation rows in LDS (gemm_aq.hip): ONE
// implementation, so the two routes cannot differ by a bit.
#pragma once
#include "sdnq_dev.h"

// The branch-free body of quant8's fast path: symmetric rows whose scale is `d.fast` (finite, ordinary significand, 2^-60 .. 2^60), float32
// arithmetic.  int8: packed division, rounding and byte packing in ~3 instructions per element instead of ~11 (sdnq_dev.h); |x| <= amax
// keeps every quotient inside +-127.5.  fp8: no NaN to flush; x = +-amax can still land one ulp above 448, hence the clamp.
template <int MM>
__device__ __forceinline__ uint2 quant8_fast(const float (&v)[8], const RowDiv& d) {
    if constexpr (MM == SDNQ_MM_I8) {
        const u32 w0 = pack4_rne_i8(fastdiv2((pv2f){v[0], v[1]}, d), fastdiv2((pv2f){v[2], v[3]}, d));
        const u32 w1 = pack4_rne_i8(fastdiv2((pv2f){v[4], v[5]}, d), fastdiv2((pv2f){v[6], v[7]}, d));
        return make_uint2(w0, w1);
    } else {
        float c[8];
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const pv2f q = fastdiv2((pv2f){v[2 * h], v[2 * h + 1]}, d);
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                // the sign of the numerator ORed onto the quotient: a no-op unless the quotient is the +0 the correction term makes of
                // -0.0 / scale, whose fp8 code is 0x80 (round 4, tools/fuzz_ops.py)
                const float qs = __uint_as_float(__float_as_uint(q[e]) | (__float_as_uint(v[2 * h + e]) & 0x80000000u));
                c[2 * h + e] = __builtin_amdgcn_fmed3f(qs, -448.0f, 448.0f);
            }
        }
        return make_uint2(pack4_e4m3fn_clamped(c[0], c[1], c[2], c[3]), pack4_e4m3fn_clamped(c[4], c[5], c[6], c[7]));
    }
}

// LP_T: SDNQ_F32 = the reference's default float32 arithmetic; SDNQ_BF16 / SDNQ_F16 = the quotient is rounded to that dtype
// before round-half-even / the fp8 cast (torch.div on 16-bit tensors, dequantize_fp32=False: linear_int8.py:15-22)
template <int MM, int LP_T = SDNQ_F32>
__device__ __forceinline__ uint2 quant8(const float (&v)[8], const RowDiv& d, int& isum, float zp = 0.0f, bool asym = false) {
    const float scale = d.scale;
    if constexpr (MM == SDNQ_MM_I8 && LP_T == SDNQ_F32) {
        if (d.fast && !asym) {  // wave-uniform
            // the symmetric int8 row of the w8a8 step: packed division, rounding and byte packing in ~3 instructions per element instead of
            // ~11 (sdnq_dev.h); `fast` excludes scale 0 / inf / nan, and |x| <= amax keeps every quotient inside +-127.5
            const uint2 w = quant8_fast<MM>(v, d);
            isum = __builtin_amdgcn_sdot4((int)w.y, 0x01010101, __builtin_amdgcn_sdot4((int)w.x, 0x01010101, isum, false), false);
            return w;
        }
    }
    if constexpr (MM == SDNQ_MM_FP8 && LP_T == SDNQ_F32) {
        if (d.fast && !asym) {  // wave-uniform: finite ordinary scale, so no NaN to flush; x = +-amax can still land one ulp above 448
            return quant8_fast<MM>(v, d);
        }
    }
    float qv[8];
    if (LP_T == SDNQ_F32 && d.fast) {  // wave-uniform
#pragma unroll
        for (int e = 0; e < 8; ++e) qv[e] = d.fastdiv(asym ? v[e] - zp : v[e]);
    } else {
        // (16-bit arithmetic: torch.sub(x, zero_point) rounds to the dtype before .div_(scale) does, quant_utils.py:282)
#pragma unroll
        for (int e = 0; e < 8; ++e) qv[e] = (asym ? (LP_T == SDNQ_F32 ? v[e] - zp : FT<LP_T>::round(v[e] - zp)) : v[e]) / scale;
    }
    if constexpr (MM == SDNQ_MM_FP8) {  // nan_to_num, clamp (+-inf fall to it), hardware conversion of four values at a time
        float c[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            // -0.0 / scale is -0.0 and its fp8 code 0x80: the three-instruction division returns +0 there (its correction term
            // x - scale * q0 is +0, and -0 + +0 = +0), so a zero numerator passes through (round 4, tools/fuzz_ops.py: f16 activations
            // that underflow to -0.0; the int8 codes have no signed zero)
            if (LP_T == SDNQ_F32 && d.fast && v[e] == 0.0f) qv[e] = v[e];
            float q = FT<LP_T>::round(qv[e]);
            if (q != q) q = 0.0f;
            c[e] = fminf(fmaxf(q, -448.0f), 448.0f);
        }
        return make_uint2(pack4_e4m3fn_clamped(c[0], c[1], c[2], c[3]), pack4_e4m3fn_clamped(c[4], c[5], c[6], c[7]));
    }
    u32 w0 = 0, w1 = 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        u32 byte;
        if constexpr (MM == SDNQ_MM_I8) {
            // x/0 -> NaN -> int8 cast gives 0 in the reference; define it explicitly
            // asymmetric (quantize_uint_mm, quant_utils.py:277-286): (x - zero_point) / scale
            float q;
            if constexpr (LP_T == SDNQ_F32) {
                q = (scale == 0.0f) ? 0.0f : __builtin_rintf(qv[e]);
            } else {  // a 16-bit scale can underflow to 0 under a nonzero row: x / 0 = +-inf -> the clamp, 0 / 0 = NaN -> 0
                q = __builtin_rintf(FT<LP_T>::round(qv[e]));
                if (q != q) q = 0.0f;
            }
            q = fminf(fmaxf(q, -128.0f), 127.0f);
            const int qi = (int)q;
            isum += qi;
            byte = (u32)qi & 0xffu;
        } else {
            float q = FT<LP_T>::round(qv[e]);
            if (q != q) q = 0.0f;  // nan_to_num; +-inf fall to the clamp
            q = fminf(fmaxf(q, -448.0f), 448.0f);
            byte = f32_to_e4m3fn(q);
        }
        if (e < 4) w0 |= byte << (8 * e);
        else w1 |= byte << (8 * (e - 4));
    }
    return make_uint2(w0, w1);
}

