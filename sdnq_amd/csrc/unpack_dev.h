// Device-side decoders for the reference's quantized weight storage formats.
//
//   packed (u)int1..7   : 8 elements <-> `bits` uint8 words      (packed_int/unpack.py:233-372)
//   packed (u)int9..15  : 16 elements <-> `bits` int16 words     (packed_int/unpack.py:7-229)
//   signed packed ints are stored as value - min, i.e. + 2^(bits-1) (packed_int/__init__.py:77-88)
//   custom eXmY floats  : sign | exponent | mantissa codes, bias 2^(e-1)-1, every exponent code finite,
//                         subnormals supported; "fnu" types have no sign bit (packed_float.py:86-132)
// The unit of work is 16 consecutive elements (one thread), which is a whole number of codec groups
// for every format.
#pragma once
#include "sdnq_dev.h"

// 8 elements from `bits` (1..7) bytes
__host__ __device__ __forceinline__ void unpack8_u8(const uint8_t* __restrict__ p, int bits, u32 (&e)[8]) {
    switch (bits) {
        case 1: {
            const u32 w = p[0];
#pragma unroll
            for (int j = 0; j < 8; ++j) e[j] = (w >> j) & 1u;
            break;
        }
        case 2: {
            const u32 w0 = p[0], w1 = p[1];
#pragma unroll
            for (int j = 0; j < 4; ++j) { e[j] = (w0 >> (2 * j)) & 3u; e[4 + j] = (w1 >> (2 * j)) & 3u; }
            break;
        }
        case 3: {
            const u32 w0 = p[0], w1 = p[1], w2 = p[2];
            e[0] = w0 & 7u; e[1] = w1 & 7u; e[2] = w2 & 7u;
            e[3] = (w0 >> 3) & 7u; e[4] = (w1 >> 3) & 7u; e[5] = (w2 >> 3) & 7u;
            e[6] = ((w0 >> 6) | ((w2 >> 4) & 4u)) & 7u;
            e[7] = ((w1 >> 6) | ((w2 >> 5) & 4u)) & 7u;
            break;
        }
        case 4: {
#pragma unroll
            for (int j = 0; j < 4; ++j) { const u32 w = p[j]; e[2 * j] = w & 15u; e[2 * j + 1] = w >> 4; }
            break;
        }
        case 5: {
            const u32 w0 = p[0], w1 = p[1], w2 = p[2], w3 = p[3], w4 = p[4];
            e[0] = w0 & 31u; e[1] = w1 & 31u; e[2] = w2 & 31u; e[3] = w3 & 31u; e[4] = w4 & 31u;
            e[5] = (w0 >> 5) | ((w3 >> 2) & 24u);
            e[6] = (w1 >> 5) | ((w4 >> 2) & 24u);
            e[7] = (w2 >> 5) | ((w3 >> 3) & 16u) | ((w4 >> 4) & 8u);
            break;
        }
        case 6: {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const u32 w0 = p[3 * h], w1 = p[3 * h + 1], w2 = p[3 * h + 2];
                e[4 * h] = w0 & 63u; e[4 * h + 1] = w1 & 63u; e[4 * h + 2] = w2 & 63u;
                e[4 * h + 3] = ((w0 >> 2) & 48u) | ((w1 >> 4) & 12u) | (w2 >> 6);
            }
            break;
        }
        default: {  // 7
            u32 w[7];
#pragma unroll
            for (int j = 0; j < 7; ++j) { w[j] = p[j]; e[j] = w[j] & 127u; }
            e[7] = ((w[0] >> 1) & 64u) | ((w[1] >> 2) & 32u) | ((w[2] >> 3) & 16u) | ((w[3] >> 4) & 8u) |
                   ((w[4] >> 5) & 4u) | ((w[5] >> 6) & 2u) | (w[6] >> 7);
            break;
        }
    }
}

// 16 elements from `bits` (9..15) 16-bit words
__host__ __device__ __forceinline__ void unpack16_i16(const uint16_t* __restrict__ p, int bits, u32 (&e)[16]) {
    switch (bits) {
        case 9: {
            const u32 w8 = p[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const u32 w = p[j];
                e[j] = w & 511u;
                // stack(w8<<7, <<5, <<3, <<1, >>1, >>3, >>5, >>7) & 384
                const u32 hi = (j < 4) ? (w8 << (7 - 2 * j)) : (w8 >> (2 * j - 7));
                e[8 + j] = ((w >> 9) & 127u) | (hi & 384u);
            }
            break;
        }
        case 10: {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const u32 w0 = p[5 * h], w1 = p[5 * h + 1], w2 = p[5 * h + 2], w3 = p[5 * h + 3], w4 = p[5 * h + 4];
                u32* o = e + 8 * h;
                o[0] = w0 & 1023u; o[1] = w1 & 1023u; o[2] = w2 & 1023u; o[3] = w3 & 1023u; o[4] = w4 & 1023u;
                o[5] = ((w0 >> 10) & 63u) | ((w3 >> 4) & 960u);
                o[6] = ((w1 >> 10) & 63u) | ((w4 >> 4) & 960u);
                o[7] = ((w2 >> 10) & 63u) | ((w3 >> 6) & 768u) | ((w4 >> 8) & 192u);
            }
            break;
        }
        case 11: {
            u32 w[11];
#pragma unroll
            for (int j = 0; j < 11; ++j) w[j] = p[j];
#pragma unroll
            for (int j = 0; j < 8; ++j) e[j] = w[j] & 2047u;
            // high parts (mask 2016 = bits 5..10): cat(w[8:11] << 5, w[8:11] >> 1, ((w[8:10] >> 7) & 480) | (stack(w10>>3, w10>>5) & 1536))
            u32 hi[8];
            hi[0] = w[8] << 5; hi[1] = w[9] << 5; hi[2] = w[10] << 5;
            hi[3] = w[8] >> 1; hi[4] = w[9] >> 1; hi[5] = w[10] >> 1;
            hi[6] = ((w[8] >> 7) & 480u) | ((w[10] >> 3) & 1536u);
            hi[7] = ((w[9] >> 7) & 480u) | ((w[10] >> 5) & 1536u);
#pragma unroll
            for (int j = 0; j < 8; ++j) e[8 + j] = ((w[j] >> 11) & 31u) | (hi[j] & 2016u);
            break;
        }
        case 12: {
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                const u32 w0 = p[3 * h], w1 = p[3 * h + 1], w2 = p[3 * h + 2];
                e[4 * h] = w0 & 4095u; e[4 * h + 1] = w1 & 4095u; e[4 * h + 2] = w2 & 4095u;
                e[4 * h + 3] = ((w0 >> 4) & 3840u) | ((w1 >> 8) & 240u) | ((w2 >> 12) & 15u);
            }
            break;
        }
        case 13: {
            u32 w[13];
#pragma unroll
            for (int j = 0; j < 13; ++j) { w[j] = p[j]; e[j] = w[j] & 8191u; }
#pragma unroll
            for (int j = 0; j < 3; ++j)
                e[13 + j] = ((w[j] >> 13) & 7u) | ((w[3 + j] >> 10) & 56u) | ((w[6 + j] >> 7) & 448u) |
                            ((w[9 + j] >> 4) & 3584u) | ((w[12] >> (1 + j)) & 4096u);
            break;
        }
        case 14: {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                u32 w[7];
#pragma unroll
                for (int j = 0; j < 7; ++j) { w[j] = p[7 * h + j]; e[8 * h + j] = w[j] & 16383u; }
                e[8 * h + 7] = ((w[0] >> 2) & 12288u) | ((w[1] >> 4) & 3072u) | ((w[2] >> 6) & 768u) | ((w[3] >> 8) & 192u) |
                               ((w[4] >> 10) & 48u) | ((w[5] >> 12) & 12u) | ((w[6] >> 14) & 3u);
            }
            break;
        }
        default: {  // 15
            u32 acc = 0;
#pragma unroll
            for (int j = 0; j < 15; ++j) {
                const u32 w = p[j];
                e[j] = w & 32767u;
                acc |= (w >> (j + 1)) & (16384u >> j);
            }
            e[15] = acc;
            break;
        }
    }
}

// eXmY code -> f32, restating unpack_float's bit surgery (packed_float.py:86-132) arithmetically:
// normal: (-1)^s * 2^(E - bias) * (1 + m/2^M), E in [1, 2^e - 1];  E == 0: (-1)^s * 2^(1-bias) * m/2^M.
__device__ __forceinline__ float decode_exmy(u32 code, int ebits, int mbits, bool is_unsigned) {
    const u32 mant = code & ((1u << mbits) - 1u);
    const u32 expo = (code >> mbits) & ((1u << ebits) - 1u);
    const u32 sign = is_unsigned ? 0u : ((code >> (ebits + mbits)) & 1u);
    const int bias = (1 << (ebits - 1)) - 1;
    float v;
    if (expo == 0) {
        // subnormal: m * 2^(1 - bias - M)
        v = (float)mant * __uint_as_float((u32)(127 + 1 - bias - mbits) << 23);
    } else {
        v = __uint_as_float(((expo + 127u - (u32)bias) << 23) | (mant << (23 - mbits)));
    }
    // both zero codes decode to +0.0 in the reference (its final where(x & mask, x, 0), packed_float.py:121-122)
    return (sign && v != 0.0f) ? -v : v;
}

struct WeightFmt {
    int storage, kind, bits, ebits, mbits, native_float;
};

// 16 consecutive elements starting at element index e0 (multiple of 16) -> numeric values before scaling
__device__ __forceinline__ void load16_values(const void* __restrict__ w, int64_t e0, const WeightFmt& f, float (&v)[16]) {
    u32 c[16];
    if (f.storage == SDNQ_ST_PACKED_U8) {
        const uint8_t* p = (const uint8_t*)w + (e0 >> 3) * f.bits;
        u32 lo[8], hi[8];
        if (f.bits == 4) {  // fast path: 8 bytes, one aligned 8-byte load
            const uint2 q = *(const uint2*)p;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                lo[j] = (q.x >> (4 * j)) & 15u;
                hi[j] = (q.y >> (4 * j)) & 15u;
            }
        } else {
            unpack8_u8(p, f.bits, lo);
            unpack8_u8(p + f.bits, f.bits, hi);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) { c[j] = lo[j]; c[8 + j] = hi[j]; }
    } else if (f.storage == SDNQ_ST_PACKED_I16) {
        unpack16_i16((const uint16_t*)w + (e0 >> 4) * f.bits, f.bits, c);
    } else if (f.storage == SDNQ_ST_RAW8) {
        const uint4 q = *(const uint4*)((const uint8_t*)w + e0);
        const u32 ww[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int j = 0; j < 16; ++j) c[j] = (ww[j >> 2] >> (8 * (j & 3))) & 0xffu;
    } else {
        const uint4 q0 = *(const uint4*)((const uint16_t*)w + e0);
        const uint4 q1 = *(const uint4*)((const uint16_t*)w + e0 + 8);
        const u32 ww[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
#pragma unroll
        for (int j = 0; j < 16; ++j) c[j] = (ww[j >> 1] >> (16 * (j & 1))) & 0xffffu;
    }
    const bool packed = (f.storage == SDNQ_ST_PACKED_U8 || f.storage == SDNQ_ST_PACKED_I16);
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        float x;
        if (f.kind == SDNQ_KIND_INT) {
            int iv;
            if (packed) iv = (int)c[j] - (1 << (f.bits - 1));                   // stored as value - min
            else if (f.bits <= 8) iv = (int)(int8_t)c[j];
            else iv = (int)(int16_t)c[j];
            x = (float)iv;
        } else if (f.kind == SDNQ_KIND_UINT) {
            x = (float)c[j];
        } else if (f.native_float) {
            if (f.bits == 8) x = (f.ebits == 4) ? e4m3fn_to_f32((uint8_t)c[j]) : e5m2_to_f32((uint8_t)c[j]);
            else x = (f.ebits == 5) ? f16_bits_to_f32((uint16_t)c[j]) : bf16_bits_to_f32((uint16_t)c[j]);
        } else {
            x = decode_exmy(c[j], f.ebits, f.mbits, f.kind == SDNQ_KIND_UFLOAT);
        }
        v[j] = x;
    }
}
