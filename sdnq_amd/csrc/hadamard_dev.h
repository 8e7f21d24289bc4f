// In-register fast Hadamard transform over a wave (64 lanes x 8 consecutive elements = 512 per pass).
//
// Reference: rotate_hadamard (quant_utils.py:194-209) = x.view(.., K/g, g) @ H_g with
//   H_g = kron^p(H4) * g^-1/2 when g is a power of 4   (build_hadamard_n4, quant_utils.py:157-165)
//   H_g = kron^p(H2) * g^-1/2 otherwise (power of 2)   (build_hadamard_n2, quant_utils.py:145-153)
//   H4 = [[1,1,1,-1],[1,1,-1,1],[1,-1,1,1],[-1,1,1,1]],  H2 = [[1,1],[1,-1]]
// kron structure => y = apply the 4-point (or 2-point) butterfly along every base-4 (base-2) digit
// of the in-group index.  For H4: y(i) = S - 2*x(3-i) with S the sum of the four digit partners, and
// the digit-(3-i) partner is the element with BOTH digit bits flipped.
// Element index inside the wave pass: idx = lane*8 + e  -> bits 0..2 = e (registers), bits 3..8 = lane.
#pragma once
#include "sdnq_dev.h"

// Value of `v` held by lane (lane ^ mask), mask a compile-time constant after unrolling.  ds_bpermute (what __shfl_xor compiles
// to) costs an LDS-crossbar pass per call and made the FWHT twice as slow as the rest of the row quantization; the xor patterns of
// the butterflies map onto DPP row operations of the VALU instead (gfx9 DPP: quad_perm for bits 0-1, row_half_mirror / row_mirror
// composed with quad_perm for bit 2, row_ror:8 for bit 3) and onto ds_swizzle's xor mode for bit 4; only bit 5 (group 512) still
// takes the crossbar.  Measured at 4608 x 15360, group 256: 198 us (bpermute) -> 174 us; gfx950's v_permlane16_swap + select for
// bit 4 was slower (191 us) than ds_swizzle.
__device__ __forceinline__ float lane_xor(float v, const int mask) {
    int x = __float_as_int(v);
    if (mask & 32) x = __shfl_xor(x, 32, 64);
    if (mask & 16) x = __builtin_amdgcn_ds_swizzle(x, 0x401F);                    // bitmask mode: and 0x1f, or 0, xor 0x10
    if (mask & 8) x = __builtin_amdgcn_update_dpp(0, x, 0x128, 0xf, 0xf, false);  // row_ror:8  == lane ^ 8 inside a 16-lane row
    switch (mask & 7) {
        case 1: x = __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xf, 0xf, false); break;  // quad_perm [1,0,3,2]
        case 2: x = __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xf, 0xf, false); break;  // quad_perm [2,3,0,1]
        case 3: x = __builtin_amdgcn_update_dpp(0, x, 0x1B, 0xf, 0xf, false); break;  // quad_perm [3,2,1,0]
        case 4:  // lane ^ 7 (row_half_mirror) then ^ 3
            x = __builtin_amdgcn_update_dpp(0, x, 0x141, 0xf, 0xf, false);
            x = __builtin_amdgcn_update_dpp(0, x, 0x1B, 0xf, 0xf, false);
            break;
        case 5:
            x = __builtin_amdgcn_update_dpp(0, x, 0x141, 0xf, 0xf, false);
            x = __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xf, 0xf, false);
            break;
        case 6:
            x = __builtin_amdgcn_update_dpp(0, x, 0x141, 0xf, 0xf, false);
            x = __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xf, 0xf, false);
            break;
        case 7: x = __builtin_amdgcn_update_dpp(0, x, 0x141, 0xf, 0xf, false); break;
        default: break;
    }
    return __int_as_float(x);
}

// scale constants c(g) = |H_g[0][0]| as the reference materialises them (H.div_(n**0.5) in the
// activation dtype); powers of 4 are exact powers of two. Values for the Sylvester sizes were read
// from the reference (tests/golden/hadamard.npz for f32; bf16/f16 are the dtype-rounded values).
__device__ __forceinline__ float hadamard_scale(int log2g, int dtype) {
    if ((log2g & 1) == 0) return __uint_as_float((u32)(127 - log2g / 2) << 23);
    // 1/sqrt(2^L), L odd
    switch (dtype) {
        case SDNQ_BF16:
            switch (log2g) {
                case 1: return __uint_as_float(0x3f350000u);
                case 3: return __uint_as_float(0x3eb50000u);
                case 5: return __uint_as_float(0x3e350000u);
                case 7: return __uint_as_float(0x3db50000u);
                default: return __uint_as_float(0x3d350000u);  // 9
            }
        case SDNQ_F16:
            switch (log2g) {
                case 1: return 0.70703125f;        // 0x39a8
                case 3: return 0.353515625f;       // 0x35a8
                case 5: return 0.1767578125f;      // 0x31a8
                case 7: return 0.08837890625f;     // 0x2da8
                default: return 0.044189453125f;   // 0x29a8
            }
        default:
            switch (log2g) {
                case 1: return __uint_as_float(0x3f3504f3u);
                case 3: return __uint_as_float(0x3eb504f3u);
                case 5: return __uint_as_float(0x3e3504f3u);
                case 7: return __uint_as_float(0x3db504f3u);
                default: return __uint_as_float(0x3d3504f3u);
            }
    }
}

__device__ __forceinline__ void had4_inlane(float& a, float& b, float& c, float& d) {
    const float s = (a + b) + (c + d);
    const float na = s - 2.0f * d, nb = s - 2.0f * c, nc = s - 2.0f * b, nd = s - 2.0f * a;
    a = na; b = nb; c = nc; d = nd;
}

// v[8]: this lane's 8 consecutive elements. log2g in [2, 9]. Unscaled butterflies, then * scale.
__device__ __forceinline__ void wave_hadamard(float (&v)[8], int log2g, float scale) {
    const int lane = threadIdx.x & 63;
    if ((log2g & 1) == 0) {
        // ---- kron powers of H4: digits (0,1) (2,3) (4,5) (6,7)
        had4_inlane(v[0], v[1], v[2], v[3]);
        had4_inlane(v[4], v[5], v[6], v[7]);
        if (log2g >= 4) {  // digit (bit2 = register, bit3 = lane bit 0)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float a = v[e], b = v[e + 4];
                const float xa = lane_xor(a, 1), xb = lane_xor(b, 1);
                const float s = (a + b) + (xa + xb);
                // partner with both bits flipped: for register e -> other lane's e+4, for e+4 -> other lane's e
                v[e] = s - 2.0f * xb;
                v[e + 4] = s - 2.0f * xa;
            }
        }
#pragma unroll
        for (int d = 6; d <= 8; d += 2) {  // digits made of lane bits (1,2) then (3,4)
            if (log2g >= d) {
                const int ma = 1 << (d - 5), mb = 1 << (d - 4);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float p1 = lane_xor(v[e], ma), p2 = lane_xor(v[e], mb), p3 = lane_xor(v[e], ma | mb);
                    const float s = (v[e] + p1) + (p2 + p3);
                    v[e] = s - 2.0f * p3;
                }
            }
        }
    } else {
        // ---- Sylvester: single-bit butterflies on bits 0..log2g-1
#pragma unroll
        for (int bit = 0; bit < 3; ++bit) {
            const int st = 1 << bit;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                if ((e & st) == 0) {
                    const float a = v[e], b = v[e + st];
                    v[e] = a + b;
                    v[e + st] = a - b;
                }
            }
        }
#pragma unroll
        for (int bit = 3; bit < 9; ++bit) {
            if (log2g > bit) {
                const int m = 1 << (bit - 3);
                const bool hi = (lane & m) != 0;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float p = lane_xor(v[e], m);
                    v[e] = hi ? (p - v[e]) : (v[e] + p);
                }
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] *= scale;
}

// Same transform for a lane that owns 16 consecutive elements (idx = lane*16 + e: bits 0..3 = registers, bits 4..8 =
// lane bits 0..4), used where the unit of work is a 16-element codec run (fused skinny linear on Hadamard layers).
__device__ __forceinline__ void wave_hadamard16(float (&v)[16], int log2g, float scale) {
    const int lane = threadIdx.x & 63;
    if ((log2g & 1) == 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) had4_inlane(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);  // digit (0,1)
        if (log2g >= 4) {
#pragma unroll
            for (int e = 0; e < 4; ++e) had4_inlane(v[e], v[e + 4], v[e + 8], v[e + 12]);             // digit (2,3)
        }
#pragma unroll
        for (int d = 6; d <= 8; d += 2) {  // digits (4,5) = lane bits (0,1), (6,7) = lane bits (2,3)
            if (log2g >= d) {
                const int ma = 1 << (d - 6), mb = 1 << (d - 5);
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float p1 = lane_xor(v[e], ma), p2 = lane_xor(v[e], mb), p3 = lane_xor(v[e], ma | mb);
                    const float s = (v[e] + p1) + (p2 + p3);
                    v[e] = s - 2.0f * p3;
                }
            }
        }
    } else {
#pragma unroll
        for (int bit = 0; bit < 4; ++bit) {
            const int st = 1 << bit;
            if (log2g > bit) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    if ((e & st) == 0) {
                        const float a = v[e], b = v[e + st];
                        v[e] = a + b;
                        v[e + st] = a - b;
                    }
                }
            }
        }
#pragma unroll
        for (int bit = 4; bit < 9; ++bit) {
            if (log2g > bit) {
                const int m = 1 << (bit - 4);
                const bool hi = (lane & m) != 0;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float p = lane_xor(v[e], m);
                    v[e] = hi ? (p - v[e]) : (v[e] + p);
                }
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) v[e] *= scale;
}

// ---- group-256 rotation on the matrix cores (see rowquant_had256_kernel in rowquant.hip for the derivation) ------------------------
typedef short v4s __attribute__((ext_vector_type(4)));
typedef _Float16 v4h __attribute__((ext_vector_type(4)));
// the constant operand of both stages: H16[k][n] / 4 for k = 4 (lane >> 4) + e, n = lane & 15;  H16[a][b] = H4[a >> 2][b >> 2] H4[a & 3][b & 3],
// H4[i][j] = -1 on the anti-diagonal
__device__ __forceinline__ void had16_operand(int lane, float (&hf)[4]) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int k = 4 * (lane >> 4) + e, n = lane & 15;
        const bool neg = (((k >> 2) + (n >> 2)) == 3) != (((k & 3) + (n & 3)) == 3);
        hf[e] = neg ? -0.25f : 0.25f;
    }
}
// one group: raw = this lane's 4 consecutive elements X[lane & 15][4 (lane >> 4) .. +3] -> the rotated values of the same 4 positions (fp32)
template <int T_ID>
__device__ __forceinline__ v4f had256_group(const uint2& raw, const float (&hf)[4]) {
    v4f d1 = {0.0f, 0.0f, 0.0f, 0.0f};
    if constexpr (T_ID == SDNQ_BF16) {
        const v4s hb = {(short)f32_to_bf16_bits(hf[0]), (short)f32_to_bf16_bits(hf[1]), (short)f32_to_bf16_bits(hf[2]), (short)f32_to_bf16_bits(hf[3])};
        d1 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(v4s, raw), hb, d1, 0, 0, 0);
    } else {
        const v4h hh = {(_Float16)hf[0], (_Float16)hf[1], (_Float16)hf[2], (_Float16)hf[3]};
        d1 = __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(v4h, raw), hh, d1, 0, 0, 0);
    }
    v4f y = {0.0f, 0.0f, 0.0f, 0.0f};
#ifndef SDNQ_HAD256_F32_STAGE2
    if constexpr (T_ID == SDNQ_BF16) {
        // Second stage on the bf16 matrix rate (round 4).  The fp32 intermediate is split EXACTLY into three bfloat16 terms by
        // truncation -- d1 = h1 + h2 + h3 with 8 significant bits each (24 in all) -- and multiplied with the same +-1/4 operand: every
        // product is exact and the fp32 accumulation adds the same real numbers as the four v_mfma_f32_16x16x4_f32 did (those cost
        // 4 x 32 matrix cycles per group; three 16x16x16 bf16 MFMAs cost 3 x 16).  The accumulator registers are
        // still, unmoved, the A operand: lane l holds rows k = 4 (l >> 4) + e of column l & 15, which is what both bf16 forms want.
        const v4s hb = {(short)f32_to_bf16_bits(hf[0]), (short)f32_to_bf16_bits(hf[1]), (short)f32_to_bf16_bits(hf[2]), (short)f32_to_bf16_bits(hf[3])};
        u32 b1[4], r1[4], r2[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            b1[e] = __float_as_uint(d1[e]);
            const float f1 = d1[e] - __uint_as_float(b1[e] & 0xffff0000u);  // exact: the low 16 bits of the significand
            r1[e] = __float_as_uint(f1);
            r2[e] = __float_as_uint(f1 - __uint_as_float(r1[e] & 0xffff0000u));  // exact: at most 8 significant bits are left
        }
        // v_perm: the HIGH halves (= the truncated bf16) of two dwords side by side
        const u32 a0 = __builtin_amdgcn_perm(b1[1], b1[0], 0x07060302u), a1 = __builtin_amdgcn_perm(b1[3], b1[2], 0x07060302u);
        const u32 a2 = __builtin_amdgcn_perm(r1[1], r1[0], 0x07060302u), a3 = __builtin_amdgcn_perm(r1[3], r1[2], 0x07060302u);
        const u32 a4 = __builtin_amdgcn_perm(r2[1], r2[0], 0x07060302u), a5 = __builtin_amdgcn_perm(r2[3], r2[2], 0x07060302u);
        // three 16x16x16 MFMAs, smallest term first (the form stage 1 uses; the 16x16x32 form on [h1 | h2] measured wrong results in
        // accumulator registers 0 and 1 -- its operand layout is not the one assumed -- and is not worth another 15 cycles)
        y = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(v4s, make_uint2(a4, a5)), hb, y, 0, 0, 0);
        y = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(v4s, make_uint2(a2, a3)), hb, y, 0, 0, 0);
        y = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(v4s, make_uint2(a0, a1)), hb, y, 0, 0, 0);
        return y;
    }
#endif
#pragma unroll
    for (int sidx = 0; sidx < 4; ++sidx) y = __builtin_amdgcn_mfma_f32_16x16x4f32(d1[sidx], hf[sidx], y, 0, 0, 0);
    return y;
}

