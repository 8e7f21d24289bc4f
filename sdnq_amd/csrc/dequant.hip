// Weight-side kernels for gfx950: dequantize, re-quantize-for-matmul, float linear.
//
//   sdnq_hip_dequant   <- SDNQDequantizer.__call__ / dequantize_weight   (dequantizer.py:135-162, 389-429)
//                         dequantize_symmetric :52-84, dequantize_asymmetric :15-48
//   sdnq_hip_requant   <- re_quantize_matmul -> re_quantize_int_mm / re_quantize_fp_mm
//                         (dequantizer.py:204-239, 166-174, 190-201; quantize_int_mm quant_utils.py:265-273)
//   sdnq_hip_linear_float <- torch.nn.functional.linear on the dequantized weight
//                         (layers/linear/forward.py:25-26; M<32 branch linear_int8.py:102-103)
//
// All three are HBM-streaming kernels: a thread owns 16 consecutive elements of one weight row (a
// whole number of codec groups for every packed format), so packed reads are contiguous per lane
// and outputs are 16/32/64-byte lane-contiguous vectors.
#include "hadamard_dev.h"
#include "sdnq_dev.h"
#include "unpack_dev.h"

namespace {

struct DeqParams {
    const void* w;
    const float* scale;
    const float* zp;
    const void* svd_up;    // [N][R]
    const void* svd_down;  // [R][K]
    int64_t N, K;
    int group_size, G, rank;
    int P, SG;  // conv weights: kernel positions per channel (1 for Linear) and scales per output row (G * P)
    int sdt;    // SdnqWeight.scale_dtype: 16-bit -> the product below is rounded to it (dequantize_fp32=False)
    WeightFmt fmt;
};

// the scalar fields of a by-value DeqParams in one batch of kernarg loads (SDNQ_KERNARGS_NOW, sdnq_dev.h)
#define SDNQ_DEQ_ARGS_NOW(p)                                                                                                              \
    SDNQ_KERNARGS_NOW("s"((p).w), "s"((p).scale), "s"((p).zp), "s"((p).svd_up), "s"((p).svd_down), "s"((p).N), "s"((p).K), "s"((p).group_size), \
                      "s"((p).G), "s"((p).rank), "s"((p).P), "s"((p).SG), "s"((p).sdt))

// dequantize 16 elements (row n, columns k0..k0+15) to fp32: f32(w)*s or fma(f32(w), s, zp)
__device__ __forceinline__ void dequant16(const DeqParams& p, int64_t n, int64_t k0, float (&v)[16]) {
    load16_values(p.w, n * p.K + k0, p.fmt, v);
    const float* srow = p.scale + n * p.SG;
    const float* zrow = p.zp ? p.zp + n * p.SG : nullptr;
    if (p.P > 1) {
        // conv weight [C_out][C_in][positions] quantized along C_in (quantizer.py:120-123, 205-209): one scale per
        // (output channel, channel group, kernel position); flattened k = c * P + pos
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int k = (int)(k0 + j), c = k / p.P;
            const int g = (c / p.group_size) * p.P + (k - c * p.P);
            v[j] = zrow ? fmaf(v[j], srow[g], zrow[g]) : v[j] * srow[g];
        }
    } else if ((p.group_size & 15) == 0) {  // one group covers the whole 16-run (wave-uniform branch)
        int64_t g64, grem;
        divmod(k0, p.group_size, g64, grem);
        const int g = (int)g64;
        const float s = srow[g];
        if (zrow) {
            const float z = zrow[g];
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] = fmaf(v[j], s, z);  // torch.addcmul == single-rounding fma
        } else {
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] = v[j] * s;
        }
    } else {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int g = (int)((k0 + j) / p.group_size);
            v[j] = zrow ? fmaf(v[j], srow[g], zrow[g]) : v[j] * srow[g];
        }
    }
    if (p.sdt != SDNQ_F32) {
        // scale / zero_point stored in the model dtype: weight.to(scale.dtype).mul_(scale) / addcmul on 16-bit tensors compute in
        // fp32 and round ONCE to that dtype (dequantizer.py:27, 63); w * s is exact in fp32, so this is that one rounding
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = round_rt(v[j], p.sdt);
    }
}

template <int OUT_T, int SVD_T>
__global__ __launch_bounds__(256) void dequant_kernel(const DeqParams p, void* __restrict__ out) {
    const int64_t units_per_row = p.K / 16;
    const int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (u >= p.N * units_per_row) return;
    int64_t n, k0;
    divmod(u, units_per_row, n, k0);  // (32-bit whenever it fits, sdnq_dev.h)
    k0 *= 16;
    float v[16];
    dequant16(p, n, k0, v);
    if (p.svd_up) {
        // result.to(svd dtype).addmm_(svd_up, svd_down): one rounding of W, fp32 accumulate, one rounding of the sum
        float acc[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) { v[j] = FT<SVD_T>::round(v[j]); acc[j] = 0.0f; }
        for (int r = 0; r < p.rank; ++r) {
            const float up = FT<SVD_T>::load(p.svd_up, n * p.rank + r);
            float dn[16];  // svd_down[r][k0 .. k0+15]: 16-byte vector loads (k0 % 16 == 0, K % 16 == 0)
            if constexpr (SVD_T == SDNQ_F32) {
#pragma unroll
                for (int q = 0; q < 4; ++q) Vec16<SDNQ_F32>::unpack(*(const uint4*)((const float*)p.svd_down + (int64_t)r * p.K + k0 + 4 * q), dn + 4 * q);
            } else {
                Vec16<SVD_T>::unpack(*(const uint4*)((const uint16_t*)p.svd_down + (int64_t)r * p.K + k0), dn);
                Vec16<SVD_T>::unpack(*(const uint4*)((const uint16_t*)p.svd_down + (int64_t)r * p.K + k0 + 8), dn + 8);
            }
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[j] = fmaf(up, dn[j], acc[j]);
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = FT<SVD_T>::round(v[j] + acc[j]);
    }
    uint8_t* o = (uint8_t*)out + (n * p.K + k0) * FT<OUT_T>::bytes;
    if constexpr (OUT_T == SDNQ_F32) {
#pragma unroll
        for (int q = 0; q < 4; ++q) *(uint4*)(o + 16 * q) = Vec16<SDNQ_F32>::pack(v + 4 * q);
    } else {
        *(uint4*)o = Vec16<OUT_T>::pack(v);
        *(uint4*)(o + 16) = Vec16<OUT_T>::pack(v + 8);
    }
}

// one wave per weight row: phase 1 amax of the fp32 dequant, phase 2 quantize (recompute, L2-hot)
// ASYM = re_quantize_uint_mm (dequantizer.py:178-187): int8 codes of (w - zero_point) / scale with the row's min / max range
template <int MM, bool ASYM = false>
__global__ __launch_bounds__(256) void requant_kernel(const DeqParams p, uint8_t* __restrict__ wq, float* __restrict__ ws,
                                                      float* __restrict__ wzp) {
    const int lane = threadIdx.x & 63;
    const int64_t n = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= p.N) return;
    const int64_t npass = (p.K + 1023) / 1024;
    float amax = 0.0f, vmin = 3.4e38f, vmax = -3.4e38f;
    for (int64_t ps = 0; ps < npass; ++ps) {
        const int64_t k0 = ps * 1024 + lane * 16;
        if (k0 < p.K) {
            float v[16];
            dequant16(p, n, k0, v);
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                if constexpr (ASYM) { vmin = fminf(vmin, v[j]); vmax = fmaxf(vmax, v[j]); }
                else amax = fmaxf(amax, fabsf(v[j]));
            }
        }
    }
    float scale, zpv = 0.0f;
    if constexpr (ASYM) {
        vmin = wave_min(vmin);
        vmax = wave_max(vmax);
        // get_scale_asymmetric (quant_utils.py:10-19) with the int8 range; zero_point.sub_(scale, alpha=-128): 128 * scale is exact.
        // A 16-bit scale dtype (dequantize_fp32=False) rounds sub_, div_ and the alpha-sub once each (round_rt is the identity for f32)
        scale = round_rt(round_rt(vmax - vmin, p.sdt) / 255.0f, p.sdt);
        zpv = round_rt(fmaf(128.0f, scale, vmin), p.sdt);
        if (lane == 0) wzp[n] = zpv;
    } else {
        amax = wave_max(amax);
        const float qmax = (MM == SDNQ_MM_I8) ? 127.0f : 448.0f;
        scale = round_rt(amax / qmax, p.sdt);  // 16-bit scale_dtype: quantize_int_mm / quantize_fp_mm run in that dtype
    }
    if (lane == 0) ws[n] = scale;
    for (int64_t ps = 0; ps < npass; ++ps) {
        const int64_t k0 = ps * 1024 + lane * 16;
        if (k0 < p.K) {
            float v[16];
            dequant16(p, n, k0, v);
            u32 o[4] = {0, 0, 0, 0};
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                u32 byte;
                if constexpr (MM == SDNQ_MM_I8) {
                    float q = __builtin_rintf(round_rt((ASYM ? round_rt(v[j] - zpv, p.sdt) : v[j]) / scale, p.sdt));
                    if (q != q) q = 0.0f;  // 0/0 of a constant row: NaN.to(int8) is 0 in the reference
                    q = fminf(fmaxf(q, -128.0f), 127.0f);
                    byte = (u32)(int)q & 0xffu;
                } else {
                    float q = round_rt(v[j] / scale, p.sdt);
                    if (q != q) q = 0.0f;
                    q = fminf(fmaxf(q, -448.0f), 448.0f);
                    byte = f32_to_e4m3fn_clamped(q);
                }
                o[j >> 2] |= byte << (8 * (j & 3));
            }
            *(uint4*)(wq + n * p.K + k0) = make_uint4(o[0], o[1], o[2], o[3]);
        }
    }
}

// Re-quantization of 4-BIT weights through a 16-entry table per (row, group) (round 3; verdict item "LUT re-quantization").
// A 4-bit code has 16 values, so within one quantization group of one row the re-quantized int8 / fp8 byte takes at most 16
// distinct values: instead of one exact division per ELEMENT (64 per group of 64), the four lanes that hold a group compute four
// table entries each -- with the very expression of requant_kernel, so the bytes are identical by construction -- swap them with
// three DPP quad moves, and expand their 16 codes through v_perm byte look-ups.  `ws_known`: the row scales were computed before
// (they depend only on the static weights) and are read from `ws` instead of being derived in a first pass over the row: the
// per-call path of SDNQ_HIP_CACHE_WEIGHTS=0 then reads the codes once.  Needs packed 4-bit storage, group_size % 64 == 0, P == 1.
template <int MM, int NP>
__global__ __launch_bounds__(256) void requant_lut4_kernel(const DeqParams p, uint8_t* __restrict__ wq, float* __restrict__ ws, int ws_known, u32* __restrict__ lut) {
    SDNQ_DEQ_ARGS_NOW(p);
    SDNQ_KERNARGS_NOW("s"(wq), "s"(ws), "s"(ws_known), "s"(lut));
    // NP = passes of 1024 elements per row (K <= 1024 NP), compile time: every load of the row -- codes and group scales of all
    // passes -- is issued before the first use (unconditionally, from clamped addresses: a load under a condition gets its own
    // vmcnt(0)), so a row costs one memory round trip instead of one per pass
    const int lane = threadIdx.x & 63;
    const int64_t n = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= p.N) return;
    const float* srow = p.scale + n * p.SG;
    const float* zrow = p.zp ? p.zp + n * p.SG : nullptr;
    const uint8_t* crow = (const uint8_t*)p.w + ((n * p.K) >> 1);
    uint2 cw[NP];
    float sg[NP], zg[NP];
#pragma unroll
    for (int ps = 0; ps < NP; ++ps) {
        int64_t k0 = (int64_t)ps * 1024 + lane * 16;
        if (k0 >= p.K) k0 = p.K - 16;  // out-of-range lanes read the row's last run and store nothing
        const int g = (int)(k0 / p.group_size);
        cw[ps] = *(const uint2*)(crow + (k0 >> 1));
        sg[ps] = srow[g];
        zg[ps] = zrow ? zrow[g] : 0.0f;
    }
    const float knownscale = ws_known ? ws[n] : 0.0f;
    auto value_of = [&](u32 code, float s, float z) {
        float x;
        if (p.fmt.kind == SDNQ_KIND_INT) x = (float)((int)code - 8);
        else if (p.fmt.kind == SDNQ_KIND_UINT) x = (float)code;
        else x = decode_exmy(code, p.fmt.ebits, p.fmt.mbits, p.fmt.kind == SDNQ_KIND_UFLOAT);
        float v = zrow ? fmaf(x, s, z) : x * s;  // dequant16
        if (p.sdt != SDNQ_F32) v = round_rt(v, p.sdt);
        return v;
    };
    const int quad = lane & 3;
    float scale = knownscale;
    if (!ws_known) {
        // amax over the row = max over (group, code present) of |value|: the four lanes of a group each test four codes against
        // the codes that actually occur in their 16 elements
        float amax = 0.0f;
#pragma unroll
        for (int ps = 0; ps < NP; ++ps) {
            const bool live = (int64_t)ps * 1024 + lane * 16 < p.K;
            u32 present = 0;  // bit c set: code c occurs among this lane's 16 elements
#pragma unroll
            for (int j = 0; j < 8; ++j) present |= (1u << ((cw[ps].x >> (4 * j)) & 15u)) | (1u << ((cw[ps].y >> (4 * j)) & 15u));
            if (!live) present = 0;
            // union over the quad (the table entries are split by code, the occurrences by lane)
            present |= (u32)__builtin_amdgcn_update_dpp(0, (int)present, 0xB1, 0xf, 0xf, false);
            present |= (u32)__builtin_amdgcn_update_dpp(0, (int)present, 0x4E, 0xf, 0xf, false);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const u32 code = 4u * quad + e;
                const float v = fabsf(value_of(code, sg[ps], zg[ps]));
                if ((present >> code) & 1u) amax = fmaxf(amax, v);
            }
        }
        amax = wave_max(amax);
        const float qmax = (MM == SDNQ_MM_I8) ? 127.0f : 448.0f;
        scale = round_rt(amax / qmax, p.sdt);
        if (lane == 0) ws[n] = scale;
    }
#pragma unroll
    for (int ps = 0; ps < NP; ++ps) {
        const int64_t k0 = (int64_t)ps * 1024 + lane * 16;
        // this lane's four table entries: codes 4 * quad .. 4 * quad + 3 (the expression of requant_kernel's second pass)
        u32 mine = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float v = value_of(4u * quad + e, sg[ps], zg[ps]);
            u32 byte;
            if constexpr (MM == SDNQ_MM_I8) {
                float q = __builtin_rintf(round_rt(v / scale, p.sdt));
                if (q != q) q = 0.0f;  // 0/0 of a constant row: NaN.to(int8) is 0 in the reference
                q = fminf(fmaxf(q, -128.0f), 127.0f);
                byte = (u32)(int)q & 0xffu;
            } else {
                float q = round_rt(v / scale, p.sdt);
                if (q != q) q = 0.0f;
                q = fminf(fmaxf(q, -448.0f), 448.0f);
                byte = f32_to_e4m3fn_clamped(q);
            }
            mine |= byte << (8 * e);
        }
        // sdnq_hip_lut4_build (round 6: the GEMM that expands the codes itself, gemm_w4.hip): the tables ARE the result -- lane l's four
        // entries are dword (l & 3) of the table of (row n, columns 64 (l >> 2) ..), i.e. lut[n][K / 64][16 bytes] written lane-linearly
        if (lut != nullptr) {
            if (k0 < p.K) lut[(n * p.K + k0) >> 4] = mine;
            continue;
        }
        // the quad's four dwords = the 16-entry table (entry c in byte c & 3 of dword c >> 2)
        const u32 t0 = (u32)__builtin_amdgcn_update_dpp(0, (int)mine, 0x00, 0xf, 0xf, false);  // quad_perm [0,0,0,0]
        const u32 t1 = (u32)__builtin_amdgcn_update_dpp(0, (int)mine, 0x55, 0xf, 0xf, false);  // [1,1,1,1]
        const u32 t2 = (u32)__builtin_amdgcn_update_dpp(0, (int)mine, 0xAA, 0xf, 0xf, false);  // [2,2,2,2]
        const u32 t3 = (u32)__builtin_amdgcn_update_dpp(0, (int)mine, 0xFF, 0xf, 0xf, false);  // [3,3,3,3]
        u32 o[4];
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const u32 half = ((h < 2 ? cw[ps].x : cw[ps].y) >> (16 * (h & 1))) & 0xffffu;  // 4 codes: element j = nibble j
            const u32 a = __builtin_amdgcn_perm(0u, half, 0x01010000u);            // bytes [b0, b0, b1, b1]
            const u32 sel = (a & 0x000f000fu) | ((a >> 4) & 0x0f000f00u);          // one code per byte
            const u32 lo = __builtin_amdgcn_perm(t1, t0, sel & 0x07070707u);       // entries 0..7 (selector k = byte k of {t1:t0})
            const u32 hi = __builtin_amdgcn_perm(t3, t2, sel & 0x07070707u);       // entries 8..15
            const u32 m = ((sel >> 3) & 0x01010101u) * 0xffu;                      // 0xff where the code is >= 8
            o[h] = (hi & m) | (lo & ~m);
        }
        if (k0 < p.K) *(uint4*)(wq + n * p.K + k0) = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

// float16 matmul operand straight from stored FLOAT codes (round 6, linear_fp16.py:27-31: `unpack_float(...).to(float16)` for packed
// formats, `weight.to(float16)` for native fp8): the decoded value rounded to float16 (exact for every format of <= 11 significand bits)
__global__ __launch_bounds__(256) void unpack_mm_f16_kernel(const DeqParams p, uint16_t* __restrict__ wq) {
    const int64_t units = p.N * p.K / 16;
    const int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (u >= units) return;
    float v[16];
    load16_values(p.w, u * 16, p.fmt, v);
    *(uint4*)(wq + u * 16) = Vec16<SDNQ_F16>::pack(v);
    *(uint4*)(wq + u * 16 + 8) = Vec16<SDNQ_F16>::pack(v + 8);
}

// matmul operand straight from the stored codes (no scaling): see sdnq_hip_unpack_mm in the header
template <int MM>
__global__ __launch_bounds__(256) void unpack_mm_kernel(const DeqParams p, uint8_t* __restrict__ wq) {
    const int64_t units = p.N * p.K / 16;
    const int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (u >= units) return;
    float v[16];
    load16_values(p.w, u * 16, p.fmt, v);
    u32 o[4] = {0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        u32 byte;
        if constexpr (MM == SDNQ_MM_I8) {
            int iv = (int)v[j];  // exact: |v| < 256
            if (p.fmt.kind == SDNQ_KIND_UINT && p.fmt.storage == SDNQ_ST_RAW8) iv ^= 0x80;  // uint8 -> int8 (linear_int8.py:46)
            byte = (u32)iv & 0xffu;
        } else {
            byte = f32_to_e4m3fn(v[j]);
        }
        o[j >> 2] |= byte << (8 * (j & 3));
    }
    *(uint4*)(wq + u * 16) = make_uint4(o[0], o[1], o[2], o[3]);
}

// out[m][n] = cast( sum_k x[m][k] * w[n][k] + bias[n] ), fp32 accumulate.
// One wave per output channel n and a chunk of MC activation rows; lanes stride K in 16-byte vectors.
template <int T_ID, int MC>
__global__ __launch_bounds__(256) void linear_float_kernel(const void* __restrict__ x, const void* __restrict__ w,
                                                           const void* __restrict__ bias, void* __restrict__ out, int64_t M,
                                                           int64_t N, int64_t K, int64_t ldx, int64_t ldc) {
    constexpr int VN = Vec16<T_ID>::n;
    const int lane = threadIdx.x & 63;
    const int64_t n = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t m0 = (int64_t)blockIdx.y * MC;
    if (n >= N) return;
    float acc[MC];
#pragma unroll
    for (int i = 0; i < MC; ++i) acc[i] = 0.0f;
    const uint8_t* wrow = (const uint8_t*)w + n * K * FT<T_ID>::bytes;
    for (int64_t k = (int64_t)lane * VN; k < K; k += 64 * VN) {
        float wv[VN];
        Vec16<T_ID>::unpack(*(const uint4*)(wrow + k * FT<T_ID>::bytes), wv);
#pragma unroll
        for (int i = 0; i < MC; ++i) {
            const int64_t m = (m0 + i < M) ? m0 + i : M - 1;
            float xv[VN];
            Vec16<T_ID>::unpack(*(const uint4*)((const uint8_t*)x + (m * ldx + k) * FT<T_ID>::bytes), xv);
#pragma unroll
            for (int e = 0; e < VN; ++e) acc[i] = fmaf(xv[e], wv[e], acc[i]);
        }
    }
#pragma unroll
    for (int i = 0; i < MC; ++i) {
        float s = acc[i];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        if (lane == 0 && m0 + i < M) {
            if (bias) s += FT<T_ID>::load(bias, n);
            FT<T_ID>::store(out, (m0 + i) * ldc + n, s);
        }
    }
}

// Fused skinny linear (M < 32): out[m][n] = cast( sum_k x[m][k] * round_T(dequant(W)[n][k]) + bias[n] ).
// Streams the QUANTIZED weight exactly once (bits/8 bytes per element instead of writing and re-reading a 2-byte
// dequantized copy): one wave per output channel, a lane decodes 16 consecutive elements per pass with the same
// arithmetic as sdnq_hip_dequant (f32(w)*s | fma, one rounding to the activation dtype T -- the reference rounds the
// dequantized weight to result_dtype before F.linear, dequantizer.py:82-83), fp32 accumulate, wave reduction.
// MROWS activation rows per launch column (grid.y walks M); x slices are re-read per channel from L1/L2.
// log2had != 0: the stored weight is Hadamard-rotated; the rounded dequantized run is un-rotated in registers (FWHT across
// the wave, 1024 elements per pass, groups never straddle a pass) and rounded to T again, exactly the order of the
// reference (dequantize -> .to(result_dtype) -> rotate_hadamard in result_dtype, dequantizer.py:82-87).
template <int T_ID, int MROWS>
__global__ __launch_bounds__(256) void linear_skinny_kernel(const DeqParams p, const void* __restrict__ x, const void* __restrict__ bias,
                                                            void* __restrict__ out, int64_t M, int64_t ldx, int log2had) {
    SDNQ_DEQ_ARGS_NOW(p);
    const int lane = threadIdx.x & 63;
    const int64_t n = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t m0 = (int64_t)blockIdx.y * MROWS;
    if (n >= p.N) return;
    float acc[MROWS];
#pragma unroll
    for (int i = 0; i < MROWS; ++i) acc[i] = 0.0f;
    const float hscale = log2had ? hadamard_scale(log2had, T_ID) : 1.0f;
    for (int64_t kb = 0; kb < p.K; kb += 1024) {  // wave-uniform trip count: the FWHT shuffles need every lane
        const int64_t k0 = kb + (int64_t)lane * 16;
        const bool live = k0 < p.K;
        float w[16];
        if (live) {
            dequant16(p, n, k0, w);
        } else {
#pragma unroll
            for (int j = 0; j < 16; ++j) w[j] = 0.0f;
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) w[j] = FT<T_ID>::round(w[j]);
        if (log2had) {
            wave_hadamard16(w, log2had, hscale);
#pragma unroll
            for (int j = 0; j < 16; ++j) w[j] = FT<T_ID>::round(w[j]);
        }
        if (!live) continue;
#pragma unroll
        for (int i = 0; i < MROWS; ++i) {
            const int64_t m = (m0 + i < M) ? m0 + i : M - 1;
            float xv[16];
            if constexpr (T_ID == SDNQ_F32) {
#pragma unroll
                for (int q = 0; q < 4; ++q) Vec16<SDNQ_F32>::unpack(*(const uint4*)((const float*)x + m * ldx + k0 + 4 * q), xv + 4 * q);
            } else {
                Vec16<T_ID>::unpack(*(const uint4*)((const uint16_t*)x + m * ldx + k0), xv);
                Vec16<T_ID>::unpack(*(const uint4*)((const uint16_t*)x + m * ldx + k0 + 8), xv + 8);
            }
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[i] = fmaf(xv[j], w[j], acc[i]);
        }
    }
#pragma unroll
    for (int i = 0; i < MROWS; ++i) {
        float sum = acc[i];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
        if (lane == 0 && m0 + i < M) {
            if (bias) sum += FT<T_ID>::load(bias, n);
            FT<T_ID>::store(out, (m0 + i) * p.N + n, sum);
        }
    }
}

// Fast few-row linear for the two storage formats that matter at M <= 4 (raw 8-bit integers and 4-bit packed integers, signed
// or unsigned, any group size that is a multiple of 16): the generic kernel above decodes inside its K loop, so a wave has ONE
// 0.5-1 KiB weight load in flight and runs at 0.7-1.2 TB/s; here the loads of up to four 1024-element chunks of the row are
// issued before anything is decoded (3-4 KiB in flight per wave, ~20 waves per CU), everything else -- f32(w)*s | fma, rounding
// to T, optional FWHT un-rotation, fp32 accumulation, wave reduction -- is the same arithmetic in the same order.
template <int T_ID, int BITS, int MROWS>
__global__ __launch_bounds__(256) void linear_skinny_fast_kernel(const DeqParams p, const void* __restrict__ x, const void* __restrict__ bias,
                                                                 void* __restrict__ out, int64_t M, int64_t ldx, int log2had) {
    SDNQ_DEQ_ARGS_NOW(p);
    const int lane = threadIdx.x & 63;
    const int64_t n = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= p.N) return;
    float acc[MROWS];
#pragma unroll
    for (int i = 0; i < MROWS; ++i) acc[i] = 0.0f;
    const float hscale = log2had ? hadamard_scale(log2had, T_ID) : 1.0f;
    const uint8_t* wrow = (const uint8_t*)p.w + (BITS == 8 ? n * p.K : n * p.K / 2);
    const float* srow = p.scale + n * p.G;
    const float* zrow = p.zp ? p.zp + n * p.G : nullptr;
    const bool is_signed = p.fmt.kind == SDNQ_KIND_INT;
    constexpr bool PFX = MROWS == 1 && T_ID != SDNQ_F32;  // single activation row: its pieces are fetched with the weights
    for (int64_t kb = 0; kb < p.K; kb += 4096) {
        // every load of this 4096-element stretch first -- codes, scales / zero points, (one-row case) activations -- all
        // UNCONDITIONAL with clamped addresses: a load under a per-lane condition gets its own s_waitcnt vmcnt(0), and a scale
        // fetched next to its use adds a dependent round trip per chunk (round 2: 45 us for FLUX's 18432 x 3072 int4 layers, 7 us of
        // weight traffic)
        uint4 raw[4], xr[PFX ? 4 : 1][2];
        float scv[4], zpv[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int64_t k0 = kb + c * 1024 + (int64_t)lane * 16;
            const int64_t ks = k0 < p.K ? k0 : 0;
            if constexpr (BITS == 8) raw[c] = *(const uint4*)(wrow + ks);
            else { const uint2 q = *(const uint2*)(wrow + ks / 2); raw[c] = make_uint4(q.x, q.y, 0, 0); }
            const int g = (int)(ks / p.group_size);  // group_size % 16 == 0: one group per 16-run
            scv[c] = srow[g];
            zpv[c] = zrow ? zrow[g] : 0.0f;
            if constexpr (PFX) {
                xr[c][0] = *(const uint4*)((const uint16_t*)x + ks);
                xr[c][1] = *(const uint4*)((const uint16_t*)x + ks + 8);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (kb + c * 1024 >= p.K) break;  // wave-uniform
            const int64_t k0 = kb + c * 1024 + (int64_t)lane * 16;
            const bool live = k0 < p.K;
            float w[16];
            const u32 ww[4] = {raw[c].x, raw[c].y, raw[c].z, raw[c].w};
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                int code;
                if constexpr (BITS == 8) {
                    const u32 b = (ww[j >> 2] >> (8 * (j & 3))) & 0xffu;
                    code = is_signed ? (int)(int8_t)b : (int)b;
                } else {
                    const u32 b = (ww[j >> 3] >> (4 * (j & 7))) & 15u;
                    code = is_signed ? (int)b - 8 : (int)b;  // packed signed ints are stored as value - min
                }
                w[j] = (float)code;
            }
            if (live) {
                const float sc = scv[c];
                if (zrow) {
                    const float z = zpv[c];
#pragma unroll
                    for (int j = 0; j < 16; ++j) w[j] = fmaf(w[j], sc, z);
                } else {
#pragma unroll
                    for (int j = 0; j < 16; ++j) w[j] = w[j] * sc;
                }
            } else {
#pragma unroll
                for (int j = 0; j < 16; ++j) w[j] = 0.0f;
            }
#pragma unroll
            for (int j = 0; j < 16; ++j) w[j] = FT<T_ID>::round(w[j]);
            if (log2had) {
                wave_hadamard16(w, log2had, hscale);
#pragma unroll
                for (int j = 0; j < 16; ++j) w[j] = FT<T_ID>::round(w[j]);
            }
            if (!live) continue;
#pragma unroll
            for (int i = 0; i < MROWS; ++i) {
                const int64_t m = (i < M) ? i : M - 1;
                float xv[16];
                if constexpr (PFX) {
                    Vec16<T_ID>::unpack(xr[c][0], xv);
                    Vec16<T_ID>::unpack(xr[c][1], xv + 8);
                } else if constexpr (T_ID == SDNQ_F32) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) Vec16<SDNQ_F32>::unpack(*(const uint4*)((const float*)x + m * ldx + k0 + 4 * q), xv + 4 * q);
                } else {
                    Vec16<T_ID>::unpack(*(const uint4*)((const uint16_t*)x + m * ldx + k0), xv);
                    Vec16<T_ID>::unpack(*(const uint4*)((const uint16_t*)x + m * ldx + k0 + 8), xv + 8);
                }
#pragma unroll
                for (int j = 0; j < 16; ++j) acc[i] = fmaf(xv[j], w[j], acc[i]);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < MROWS; ++i) {
        float sum = acc[i];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
        if (lane == 0 && i < M) {
            if (bias) sum += FT<T_ID>::load(bias, n);
            FT<T_ID>::store(out, (int64_t)i * p.N + n, sum);
        }
    }
}

// The few-row linear of HADAMARD layers with the default rotation group 256 (FLUX int4 + Hadamard adaLN projections): the weight row is
// un-rotated on the matrix cores.  linear_skinny_fast_kernel's FWHT (16 elements per lane: two radix-4 stages across lanes = 96 DPP /
// ds_swizzle moves per 16 elements) took 37 of the 59 us of an 18432 x 3072 int4 layer.  Here a wave owns one output channel and
// walks its row group by group in the MFMA layout of hadamard_dev.h: lane l holds the 4 consecutive columns 16 (l & 15) + 4 (l >> 4)
// .. +3 of the group -- 2 bytes of int4 codes / 4 bytes of int8 codes per lane, a whole group = one contiguous 128 / 256 bytes per
// wave-load -- dequantizes them (f32(q) * s | fma, rounded to T: dequantizer.py:27, 63), rotates (five MFMAs, rounded to T:
// dequantizer.py:82-87) and multiplies with the same 4 columns of x.  All loads of up to 16 groups are issued before the first use.
template <int T_ID, int BITS, int MROWS>
__global__ __launch_bounds__(256) void linear_skinny_had256_kernel(const DeqParams p, const void* __restrict__ x, const void* __restrict__ bias,
                                                                   void* __restrict__ out, int64_t M, int64_t ldx) {
    SDNQ_DEQ_ARGS_NOW(p);
    static_assert(T_ID == SDNQ_BF16 || T_ID == SDNQ_F16, "16-bit activations");
    constexpr int NG = 16;
    const int lane = threadIdx.x & 63;
    // the wave's output channel is wave-uniform: pinned to a scalar register, so that the buffer descriptors derived from it live in SGPRs
    // (left as a function of threadIdx.x they were VGPRs, and each of the 48 buffer loads sat in a readfirstlane waterfall loop)
    const int64_t n = (int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (n >= p.N) return;
    float acc[MROWS];
#pragma unroll
    for (int i = 0; i < MROWS; ++i) acc[i] = 0.0f;
    const int eoff = 16 * (lane & 15) + 4 * (lane >> 4);
    const uint8_t* wrow = (const uint8_t*)p.w + (BITS == 8 ? n * p.K : n * p.K / 2) + (BITS == 8 ? eoff : eoff / 2);
    const float* srow = p.scale + n * p.G;
    const float* zrow = p.zp ? p.zp + n * p.G : nullptr;
    const bool is_signed = p.fmt.kind == SDNQ_KIND_INT;
    // codes -> numbers without branches: int8 two's complement: (byte ^ 0x80) - 128;  uint8: byte;  packed signed nibble: code - 8
    const u32 flip8 = (is_signed && BITS == 8) ? 0x80808080u : 0u;
    const float qsub = is_signed ? (BITS == 8 ? 128.0f : 8.0f) : 0.0f;
    float hf[4];
    had16_operand(lane, hf);
    const int ngroups = (int)(p.K / 256);
    auto rsW = SDNQ_MAKE_RSRC((const uint8_t*)p.w + (BITS == 8 ? n * p.K : n * p.K / 2));
    auto rsS = SDNQ_MAKE_RSRC(srow);
    auto rsZ = SDNQ_MAKE_RSRC(zrow ? zrow : srow);
    auto rsX = SDNQ_MAKE_RSRC(x);
    for (int g0 = 0; g0 < ngroups; g0 += NG) {
        u32 code[NG];
        float sc[NG], zp[NG];
        uint2 xr[NG][MROWS];
#pragma unroll
        for (int g = 0; g < NG; ++g) {  // unconditional, clamped: a group past the end re-reads group 0 and is dropped
            // (buffer loads -- wave-uniform base, 32-bit lane offset, group offset in the scalar operand: a load with a 64-bit VGPR
            //  address waits ~1000 cycles at issue while another wave of the SIMD runs the rotation's MFMAs, sdnq_dev.h)
            const int kk = (g0 + g < ngroups ? g0 + g : 0) * 256;
            if constexpr (BITS == 8) code[g] = (u32)SDNQ_BUF_LOAD4(rsW, eoff, kk);
            else code[g] = (u32)SDNQ_BUF_LOAD2(rsW, eoff / 2, kk / 2);
            const int gi = (kk + eoff) / p.group_size;  // group_size % 4 == 0: the 4 columns share one scale group
            sc[g] = __builtin_bit_cast(float, SDNQ_BUF_LOAD4(rsS, gi * 4, 0));
            zp[g] = zrow ? __builtin_bit_cast(float, SDNQ_BUF_LOAD4(rsZ, gi * 4, 0)) : 0.0f;
#pragma unroll
            for (int i = 0; i < MROWS; ++i) {
                const v2i t = SDNQ_BUF_LOAD8(rsX, (int)((i < M ? i : 0) * ldx + eoff) * 2, kk * 2);
                xr[g][i] = make_uint2((u32)t[0], (u32)t[1]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (g0 + g < ngroups) {  // wave-uniform
            // VALU diet (the kernel is VALU + MFMA bound, not HBM bound): codes -> floats with one extract + one convert each, the two
            // roundings to T as PACKED converts whose results feed the MFMA / the dot product directly, x . w as packed dot products
            float w[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float q;
                if constexpr (BITS == 8) q = (float)(((code[g] ^ flip8) >> (8 * e)) & 0xffu) - qsub;   // v_cvt_f32_ubyteN
                else q = (float)((code[g] >> (4 * e)) & 15u) - qsub;
                w[e] = zrow ? fmaf(q, sc[g], zp[g]) : q * sc[g];
            }
            const uint2 wp = make_uint2(pack2<T_ID>(w[0], w[1]), pack2<T_ID>(w[2], w[3]));  // the rounding to T (dequantizer.py:27, 63)
            const v4f y = had256_group<T_ID>(wp, hf);
#pragma unroll
            for (int i = 0; i < MROWS; ++i) {
                const u32 y0 = pack2<T_ID>(y[0], y[1]), y1 = pack2<T_ID>(y[2], y[3]);  // the rounding to T after the rotation (:82-87)
                if constexpr (T_ID == SDNQ_BF16) {
                    acc[i] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(v2bf, xr[g][i].x), __builtin_bit_cast(v2bf, y0), acc[i], false);
                    acc[i] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(v2bf, xr[g][i].y), __builtin_bit_cast(v2bf, y1), acc[i], false);
                } else {
                    acc[i] = __builtin_amdgcn_fdot2(__builtin_bit_cast(v2h, xr[g][i].x), __builtin_bit_cast(v2h, y0), acc[i], false);
                    acc[i] = __builtin_amdgcn_fdot2(__builtin_bit_cast(v2h, xr[g][i].y), __builtin_bit_cast(v2h, y1), acc[i], false);
                }
            }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < MROWS; ++i) {
        float sum = acc[i];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
        if (lane == 0 && i < M) {
            if (bias) sum += FT<T_ID>::load(bias, n);
            FT<T_ID>::store(out, (int64_t)i * p.N + n, sum);
        }
    }
}

// Few-row linear on an int8 row-wise weight WITH SVD factors (the M < 32 branch of an SVD layer, e.g. FLUX adaLN projections):
// y = x . W^T + b with W = round(round(q * s) + svd_up . svd_down) exactly as sdnq_hip_dequant forms it (dequantizer.py:79-83),
// but the rank-R product is done on the matrix cores tile by tile and W never exists in memory.  One workgroup = 32 output
// channels; its 4 waves split K in blocks of 32.  Per block: D[k][n] = down_t[k][:] . up[n][:] (R/16 MFMAs, operands are
// 16-byte rows of down_t [K][R] and svd_up [N][R]); lane (n = lane & 31, half = lane >> 5) then owns k = (reg & 3) + 8 (reg >> 2)
// + 4 half of that tile, decodes the matching 4 x 4 int8 codes of row n, forms W and multiplies by x (f32 copy in LDS).
// HBM-bound on the codes: N*K bytes (the dequantize + GEMV pair it replaces moves 5 N*K bytes and is VALU-bound on the rank loop).
template <bool IS_BF16, int MR, int BITS>
__global__ __launch_bounds__(256) void skinny_svd_kernel(const DeqParams p, const uint16_t* __restrict__ down_t, const void* __restrict__ x,
                                                         const void* __restrict__ bias, void* __restrict__ out, int64_t M, int64_t ldx) {
    constexpr int T_ID = IS_BF16 ? SDNQ_BF16 : SDNQ_F16;
    extern __shared__ __attribute__((aligned(16))) float xs[];  // [MR][K]
    __shared__ float red[4][MR][32];
    const int tid = threadIdx.x, lane = tid & 63, nl = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int K = (int)p.K, R = p.rank;
    for (int i = tid; i < MR * K; i += 256) {
        const int m = i / K, k = i - m * K;
        xs[i] = (m < M) ? FT<T_ID>::load(x, (int64_t)m * ldx + k) : 0.0f;
    }
    __syncthreads();
    int64_t gn = (int64_t)blockIdx.x * 32 + nl;
    const bool n_ok = gn < p.N;
    if (!n_ok) gn = p.N - 1;
    const float* srow = p.scale + gn * p.G;
    const float* zrow = p.zp ? p.zp + gn * p.G : nullptr;
    const bool is_signed = p.fmt.kind == SDNQ_KIND_INT;
    const uint16_t* up = (const uint16_t*)p.svd_up + gn * R + hi * 8;
    const uint8_t* wrow = (const uint8_t*)p.w + (BITS == 8 ? gn * K : gn * K / 2);
    float acc[MR];
#pragma unroll
    for (int m = 0; m < MR; ++m) acc[m] = 0.0f;
    for (int k0 = wave * 32; k0 < K; k0 += 128) {
        v16f ud;
#pragma unroll
        for (int e = 0; e < 16; ++e) ud[e] = 0.0f;
        const uint16_t* dn = down_t + (int64_t)(k0 + nl) * R + hi * 8;
        for (int kr = 0; kr < R; kr += 16) {
            const uint4 fd = *(const uint4*)(dn + kr), fu = *(const uint4*)(up + kr);
            if constexpr (IS_BF16) ud = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(v8bf, fd), __builtin_bit_cast(v8bf, fu), ud, 0, 0, 0);
            else ud = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(v8h, fd), __builtin_bit_cast(v8h, fu), ud, 0, 0, 0);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int kb = k0 + 8 * g + 4 * hi;  // 4 consecutive columns: one scale group (group_size % 4 == 0)
            const float s = srow[kb / p.group_size];
            const float z = zrow ? zrow[kb / p.group_size] : 0.0f;
            u32 w4;
            if constexpr (BITS == 8) w4 = *(const u32*)(wrow + kb);
            else w4 = *(const uint16_t*)(wrow + kb / 2);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float q;
                if constexpr (BITS == 8) q = is_signed ? (float)(int)(int8_t)(w4 >> (8 * e)) : (float)((w4 >> (8 * e)) & 0xffu);
                else q = is_signed ? (float)((int)((w4 >> (4 * e)) & 15u) - 8) : (float)((w4 >> (4 * e)) & 15u);  // packed signed: value - min
                float wv = FT<T_ID>::round(zrow ? fmaf(q, s, z) : q * s);  // dequantize -> .to(svd dtype)
                wv = FT<T_ID>::round(wv + ud[4 * g + e]);                // addmm_(svd_up, svd_down): one rounding of the sum
#pragma unroll
                for (int m = 0; m < MR; ++m) acc[m] = fmaf(xs[m * K + kb + e], wv, acc[m]);
            }
        }
    }
#pragma unroll
    for (int m = 0; m < MR; ++m) {
        acc[m] += __shfl_xor(acc[m], 32, 64);
        if (hi == 0) red[wave][m][nl] = acc[m];
    }
    __syncthreads();
    if (tid < 32 * MR) {
        const int m = tid / 32, n = tid % 32;
        const int64_t on = (int64_t)blockIdx.x * 32 + n;
        if (m < M && on < p.N) {
            float sum = (red[0][m][n] + red[1][m][n]) + (red[2][m][n] + red[3][m][n]);
            if (bias) sum += FT<T_ID>::load(bias, on);
            FT<T_ID>::store(out, (int64_t)m * p.N + on, sum);
        }
    }
}

// The same few-row SVD linear for the default rank R = 32, fed by LDS-DMA.  skinny_svd_kernel above loads 4 bytes per lane per load
// and waits for every 32-k block's loads before using them: 24 dependent memory round trips per wave, 76 us for FLUX's 18432 x 3072
// modulation layers against 14 us of weight traffic.  Here every wave owns a private ring of D stages (one 32-k block each: the
// block's codes, 32 rows x 32 bytes, and its 32 rows of down_t, 64 bytes each = 3 LDS-DMAs of 1 KB), D - 1 blocks in flight, no
// workgroup barrier in the loop (a wave only reads what it fetched itself).  The MFMA's k rows are fed in a PERMUTED order --
// A-operand row i carries k = 16 ((i >> 2) & 1) + 4 (i >> 3) + (i & 3) -- so that the 16 accumulator registers of lane (n = lane &
// 31, half = lane >> 5) are the 16 CONSECUTIVE columns 16 half .. 16 half + 15 of row n: one 16-byte LDS read fetches their codes.
// LDS swizzles (applied on the global side of the DMA, LDS stays lane-linear): codes: 16-byte half ^= (row >> 3) & 1; down_t:
// 16-byte chunk ^= (row >> 2) & 3.
template <bool IS_BF16, int MR, int BITS>
__global__ __launch_bounds__(256) void skinny_svd32_kernel(const DeqParams p, const uint16_t* __restrict__ down_t, const void* __restrict__ x,
                                                           const void* __restrict__ bias, void* __restrict__ out, int64_t M, int64_t ldx) {
    SDNQ_DEQ_ARGS_NOW(p);
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    constexpr int T_ID = IS_BF16 ? SDNQ_BF16 : SDNQ_F16;
    constexpr int D = 4, STG = 3072, R = 32;
    extern __shared__ __attribute__((aligned(1024))) uint8_t smem[];  // [4 waves][D][STG] rings, xs [MR][K] f32, scales / zero points [32][G] f32 each
    __shared__ float red[4][MR][32];
    const int tid = threadIdx.x, lane = tid & 63, nl = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int K = (int)p.K;
    uint8_t* ring = smem + wave * (D * STG);
    uint16_t* xs = (uint16_t*)(smem + 4 * D * STG);  // [MR][K] activations, 16-bit
    const int64_t n0 = (int64_t)blockIdx.x * 32;
    // ---- DMA roles
    const uint8_t* wsrc;  // this lane's code bytes of block 0
    if constexpr (BITS == 8) {
        const int r = lane >> 1, lh = (lane & 1) ^ ((r >> 3) & 1);
        int64_t g = n0 + r;
        if (g >= p.N) g = p.N - 1;
        wsrc = (const uint8_t*)p.w + g * K + 16 * lh;
    } else {  // 32 rows x 16 bytes = half a DMA: the upper 32 lanes fetch the same bytes again (their LDS kilobyte half is not read)
        int64_t g = n0 + (lane & 31);
        if (g >= p.N) g = p.N - 1;
        wsrc = (const uint8_t*)p.w + g * (K / 2);
    }
    const int drow = lane >> 2, dchunk = lane & 3;  // down_t piece a: row 16 a + drow, physical chunk dchunk
    const uint16_t* dsrc0 = down_t + (int64_t)drow * R + ((dchunk ^ ((drow >> 2) & 3)) << 3);
    const uint16_t* dsrc1 = down_t + (int64_t)(16 + drow) * R + ((dchunk ^ (((16 + drow) >> 2) & 3)) << 3);
    const int nblk = K / 32, nit = (nblk + 3) / 4;
    auto issue = [&](int it) {  // block 4 it + wave; past the end of K: the last block again (dropped by `live` below)
        int b = 4 * it + wave;
        if (b >= nblk) b = nblk - 1;
        uint8_t* base = ring + (it % D) * STG;
        const int k0 = b * 32;
        __builtin_amdgcn_global_load_lds((gptr_t)(wsrc + (BITS == 8 ? k0 : k0 / 2)), (lptr_t)base, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)(dsrc0 + (int64_t)k0 * R), (lptr_t)(base + 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)(dsrc1 + (int64_t)k0 * R), (lptr_t)(base + 2048), 16, 0, 0);
    };
#pragma unroll
    for (int s0 = 0; s0 < D - 1; ++s0) issue(s0);
    // ---- x in LDS as it is (16-bit elements; rows past M are zero) while the first blocks are in flight: 16-byte pieces, four loads
    // per thread in flight (an element-at-a-time loop waits one memory round trip per element: 12 of them for K = 3072)
    {
        const int kc = K / 8, total = MR * kc;  // 16-byte pieces
        for (int c0 = tid; c0 < total; c0 += 4 * 256) {
            uint4 v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c = c0 + j * 256 < total ? c0 + j * 256 : 0;
                const int m = c / kc, k8 = c - m * kc;
                v[j] = *(const uint4*)((const uint16_t*)x + (int64_t)(m < M ? m : 0) * ldx + k8 * 8);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c = c0 + j * 256;
                if (c < total) *(uint4*)(xs + c * 8) = (c / kc < M) ? v[j] : make_uint4(0, 0, 0, 0);
            }
        }
    }
    // scales / zero points of the 32 rows in LDS: a global load inside the K loop would make the compiler drain the DMA ring
    // (s_waitcnt vmcnt(0)) at its first use
    const int G = p.G;
    float* s_sc = (float*)(xs + MR * K);
    float* s_zp = s_sc + 32 * G;
    for (int i = tid; i < 32 * G; i += 256) {
        int64_t g = n0 + i / G;
        if (g >= p.N) g = p.N - 1;
        s_sc[i] = p.scale[g * G + i % G];
        s_zp[i] = p.zp ? p.zp[g * G + i % G] : 0.0f;  // fma(q, s, +0) == q * s
    }
    int64_t gn = n0 + nl;
    if (gn >= p.N) gn = p.N - 1;
    // codes -> numbers without branches: int8 two's complement: (byte ^ 0x80) - 128;  uint8: byte;  packed signed nibble: code - 8
    const bool is_signed = p.fmt.kind == SDNQ_KIND_INT;
    const u32 flip = (is_signed && BITS == 8) ? 0x80808080u : 0u;
    const float qsub = is_signed ? (BITS == 8 ? 128.0f : 8.0f) : 0.0f;
    const float inv_group = 1.0f / (float)p.group_size;
    const uint16_t* up = (const uint16_t*)p.svd_up + gn * R + hi * 8;
    const v4i fu0 = *(const v4i*)up, fu1 = *(const v4i*)(up + 16);
    // fragment reads: A row of this lane = the permuted k row; code bytes of row nl
    const int krow = 16 * ((nl >> 2) & 1) + 4 * (nl >> 3) + (nl & 3);
    const int a_off0 = 1024 + krow * 64 + (((0 + hi) ^ ((krow >> 2) & 3)) << 4);
    const int a_off1 = 1024 + krow * 64 + (((2 + hi) ^ ((krow >> 2) & 3)) << 4);
    const int w_off = BITS == 8 ? nl * 32 + ((hi ^ ((nl >> 3) & 1)) << 4) : nl * 16 + hi * 8;
    float acc[MR];
#pragma unroll
    for (int m = 0; m < MR; ++m) acc[m] = 0.0f;
    __syncthreads();  // xs complete
    for (int it = 0; it < nit; ++it) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 2) * 3) : "memory");  // this wave's block `it` has landed
        issue(it + D - 1);  // refills the slot read in the previous iteration (its reads fed arithmetic already)
        const uint8_t* base = ring + (it % D) * STG;
        if (4 * it + wave >= nblk) continue;  // wave-uniform: a block past the end of K (its DMAs re-fetched the last block)
        const int kb = (4 * it + wave) * 32 + 16 * hi;  // this lane's 16 consecutive columns: one scale group (group_size % 16 == 0)
        v16f ud;
#pragma unroll
        for (int e = 0; e < 16; ++e) ud[e] = 0.0f;
        const v4i fd0 = *(const v4i*)(base + a_off0), fd1 = *(const v4i*)(base + a_off1);
        if constexpr (IS_BF16) {
            ud = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(v8bf, fd0), __builtin_bit_cast(v8bf, fu0), ud, 0, 0, 0);
            ud = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(v8bf, fd1), __builtin_bit_cast(v8bf, fu1), ud, 0, 0, 0);
        } else {
            ud = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(v8h, fd0), __builtin_bit_cast(v8h, fu0), ud, 0, 0, 0);
            ud = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(v8h, fd1), __builtin_bit_cast(v8h, fu1), ud, 0, 0, 0);
        }
        u32 ww[4];
        if constexpr (BITS == 8) { const v4i t4 = *(const v4i*)(base + w_off); ww[0] = t4[0]; ww[1] = t4[1]; ww[2] = t4[2]; ww[3] = t4[3]; }
        else { const v2i t2 = *(const v2i*)(base + w_off); ww[0] = t2[0]; ww[1] = t2[1]; ww[2] = 0; ww[3] = 0; }
        const int gi = (int)(((float)kb + 0.5f) * inv_group);  // kb / group_size (exact: both are multiples of 16, K < 2^20)
        const float sc = s_sc[nl * G + gi], zc = s_zp[nl * G + gi];
        // VALU diet (the kernel is bound by the per-element arithmetic, ~14 instructions before): codes -> floats with one
        // v_cvt_f32_ubyteN each, both roundings to T as PACKED converts of a column pair, the products as packed dot products
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            u32 xq[MR][2];  // columns (4 g, 4 g + 1) and (4 g + 2, 4 g + 3) of every activation row, as stored
#pragma unroll
            for (int m = 0; m < MR; ++m) {
                const v2i t2 = *(const v2i*)(xs + m * K + kb + 4 * g);
                xq[m][0] = (u32)t2[0];
                xq[m][1] = (u32)t2[1];
            }
            u32 cw;  // the 4 codes of columns 4 g .. 4 g + 3, one per byte
            if constexpr (BITS == 8) {
                cw = ww[g] ^ flip;
            } else {
                const u32 n4 = (ww[g >> 1] >> (16 * (g & 1))) & 0xffffu;  // 4 nibbles -> 4 bytes
                cw = (n4 & 0xfu) | ((n4 & 0xf0u) << 4) | ((n4 & 0xf00u) << 8) | ((n4 & 0xf000u) << 12);
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const float q0 = (float)((cw >> (16 * h)) & 0xffu) - qsub, q1 = (float)((cw >> (16 * h + 8)) & 0xffu) - qsub;
                const u32 pw = pack2<T_ID>(fmaf(q0, sc, zc), fmaf(q1, sc, zc));  // dequantize -> .to(svd dtype)
                float r0, r1;
                if constexpr (IS_BF16) { r0 = __uint_as_float(pw << 16); r1 = __uint_as_float(pw & 0xffff0000u); }
                else { r0 = f16_bits_to_f32((uint16_t)pw); r1 = f16_bits_to_f32((uint16_t)(pw >> 16)); }
                const u32 ps = pack2<T_ID>(r0 + ud[4 * g + 2 * h], r1 + ud[4 * g + 2 * h + 1]);  // addmm_: one rounding of the sum
#pragma unroll
                for (int m = 0; m < MR; ++m) {
                    if constexpr (IS_BF16) acc[m] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(v2bf, xq[m][h]), __builtin_bit_cast(v2bf, ps), acc[m], false);
                    else acc[m] = __builtin_amdgcn_fdot2(__builtin_bit_cast(v2h, xq[m][h]), __builtin_bit_cast(v2h, ps), acc[m], false);
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // trailing refills
#pragma unroll
    for (int m = 0; m < MR; ++m) {
        acc[m] += __shfl_xor(acc[m], 32, 64);
        if (hi == 0) red[wave][m][nl] = acc[m];
    }
    __syncthreads();
    if (tid < 32 * MR) {
        const int m = tid / 32, n = tid % 32;
        const int64_t on = n0 + n;
        if (m < M && on < p.N) {
            float sum = (red[0][m][n] + red[1][m][n]) + (red[2][m][n] + red[3][m][n]);
            if (bias) sum += FT<T_ID>::load(bias, on);
            FT<T_ID>::store(out, (int64_t)m * p.N + on, sum);
        }
    }
}

// t[M][R] = cast( x[M][K] . down[R][K]^T ) on the matrix cores (bf16 / f16): the inner torch.mm of the SVD branch
// (linear_int8.py:60).  HBM-bound on x: 2*M*K bytes (28 MB for a FLUX activation).
// One workgroup (4 waves) = 16 activation rows x all of K x 32 factor rows, walked in stages of 128 k through an LDS ring:
//   HBM / L2 -> LDS by LDS-DMA (global_load_lds_dwordx4: no registers, so the ring depth -- not the compiler's s_waitcnt model --
//   decides how many stages are in flight): a piece = 4 rows x 256 bytes, lane l -> row l / 16, 16-byte chunk l % 16, XOR-swizzled
//   on the global side so that LDS stays lane-linear; 4 activation + 8 factor pieces per stage, 3 per wave;
//   LDS -> v_mfma_f32_16x16x32 fragments: wave w owns k-step w of every stage (A = factor rows, B = activation rows); the 16 rows of
//   a fragment read hit 16 different bank groups thanks to the swizzle.
// History (rounds 1-2): the fragments used to be loaded straight from global memory in the MFMA layout -- lane l = row (l & 15) at
// a 6 KB row stride = 64 different cache lines per load instruction -- and the L1 tag rate, not HBM, bounded the kernel: 20-24 us
// per call at 4608 x 3072 whatever the software pipelining, 12 us even for 77 rows.  A register-staged coalesced variant lost to
// the compiler's pessimistic s_waitcnt on loop-carried loads (47 us).
__device__ const uint4 g_lr_zero16 = {0u, 0u, 0u, 0u};  // source of chunks past the end of K
// RT = activation row tiles of 16 per workgroup.  What bounds the kernel is the LDS-DMA rate of a CU (a stage moves 4 KB of activations
// per row tile and ALWAYS 8 KB of factor rows), so the launcher picks RT by the largest number of bytes a CU has to move: 4608 rows are
// 288 workgroups of one tile -- 32 CUs get two, 2 x 288 KB per K = 3072 -- or 144 workgroups of two tiles, 384 KB each: 8.5 -> 6 us.
template <bool IS_BF16, int RT>
__global__ __launch_bounds__(256) void lowrank_down_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ down,
                                                           uint16_t* __restrict__ t, int64_t M, int64_t K, int64_t ldx, int R) {
    SDNQ_KERNARGS_NOW("s"(x), "s"(down), "s"(t), "s"(M), "s"(K), "s"(ldx), "s"(R));
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    constexpr int NS = 4, XS = RT * 16 * 256, STAGE = XS + 32 * 256;  // 12 / 16 KB per stage; 48 / 64 KB ring
    constexpr int NDMA = RT + 2;  // DMAs per wave and stage
    __shared__ __attribute__((aligned(1024))) uint8_t lds[NS * STAGE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t m0 = (int64_t)blockIdx.x * (16 * RT);
    const int n_tiles = (R + 31) / 32;
    const int64_t nst = (K + 127) / 128;
    // DMA role of this lane: row 4 * wave + lane / 16 of every activation tile and of each half of the factor tile; physical chunk
    // lane % 16 holds logical chunk (lane % 16) ^ row
    const int drow = wave * 4 + (lane >> 4);
    const int lchunk = (lane & 15) ^ drow;
    // fragment role: row lane & 15, logical chunk 4 * wave + lane / 16 of the stage
    const int frow = lane & 15;
    const int foff = frow * 256 + (((wave * 4 + (lane >> 4)) ^ frow) << 4);
    const uint16_t* sx[RT];
#pragma unroll
    for (int r = 0; r < RT; ++r) {
        int64_t gm = m0 + r * 16 + drow;
        if (gm >= M) gm = M - 1;
        sx[r] = x + gm * ldx + lchunk * 8;
    }
    for (int nt = 0; nt < n_tiles; ++nt) {
        const int gn0 = nt * 32 + drow, gn1 = gn0 + 16;
        const uint16_t* sd0 = down + (int64_t)(gn0 < R ? gn0 : 0) * K + lchunk * 8;  // rows past R: valid memory, never stored
        const uint16_t* sd1 = down + (int64_t)(gn1 < R ? gn1 : 0) * K + lchunk * 8;
        auto issue = [&](int64_t st) {  // stages past the end of K are all-zero DMAs: the counted vmcnt stays a constant
            uint8_t* base = lds + (st % NS) * STAGE;
            const int64_t k0 = st * 128;
            const bool ok = k0 + lchunk * 8 < K;
            const uintptr_t z = (uintptr_t)&g_lr_zero16;  // (integer selects: a pointer ternary became three divergent branches)
#pragma unroll
            for (int r = 0; r < RT; ++r) {
                const uintptr_t px = ok ? (uintptr_t)(sx[r] + k0) : z;
                __builtin_amdgcn_global_load_lds((gptr_t)px, (lptr_t)(base + r * 4096 + wave * 1024), 16, 0, 0);
            }
            const uintptr_t p0 = ok ? (uintptr_t)(sd0 + k0) : z, p1 = ok ? (uintptr_t)(sd1 + k0) : z;
            __builtin_amdgcn_global_load_lds((gptr_t)p0, (lptr_t)(base + XS + wave * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr_t)p1, (lptr_t)(base + XS + 4096 + wave * 1024), 16, 0, 0);
        };
        v4f acc[RT][2];
#pragma unroll
        for (int r = 0; r < RT; ++r) { acc[r][0] = (v4f){0.0f, 0.0f, 0.0f, 0.0f}; acc[r][1] = (v4f){0.0f, 0.0f, 0.0f, 0.0f}; }
#pragma unroll
        for (int s0 = 0; s0 < NS - 1; ++s0) issue(s0);
        for (int64_t st = 0; st < nst; ++st) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * NDMA) : "memory");  // this wave's pieces of stage st have landed
            // ... and everybody's; every wave is also done reading stage st - 1 (its fragments fed MFMAs already), whose slot is
            // refilled next.  Raw s_barrier: __syncthreads() would drain the DMAs in flight (s_waitcnt vmcnt(0)).
            __builtin_amdgcn_s_barrier();
            issue(st + NS - 1);
            const uint8_t* base = lds + (st % NS) * STAGE;
            // (ext-vector loads: an LDS read typed as the HIP uint4 struct makes the compiler drain the LDS-DMAs first, vmcnt(0))
            v4i fx[RT];
#pragma unroll
            for (int r = 0; r < RT; ++r) fx[r] = *(const v4i*)(base + r * 4096 + foff);
            const v4i f0 = *(const v4i*)(base + XS + foff);
            const v4i f1 = *(const v4i*)(base + XS + 4096 + foff);
#pragma unroll
            for (int r = 0; r < RT; ++r) {
                if constexpr (IS_BF16) {
                    acc[r][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, f0), __builtin_bit_cast(v8bf, fx[r]), acc[r][0], 0, 0, 0);
                    acc[r][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, f1), __builtin_bit_cast(v8bf, fx[r]), acc[r][1], 0, 0, 0);
                } else {
                    acc[r][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(v8h, f0), __builtin_bit_cast(v8h, fx[r]), acc[r][0], 0, 0, 0);
                    acc[r][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(v8h, f1), __builtin_bit_cast(v8h, fx[r]), acc[r][1], 0, 0, 0);
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the trailing zero DMAs target the ring the partial sums reuse
        __syncthreads();
        float* part = (float*)lds;  // [4 waves][RT row tiles][2 rank halves][4 regs][64 lanes]
        constexpr int WSTRIDE = RT * 2 * 4 * 64;
#pragma unroll
        for (int r = 0; r < RT; ++r)
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int e = 0; e < 4; ++e) part[wave * WSTRIDE + ((r * 2 + h) * 4 + e) * 64 + lane] = acc[r][h][e];
        __syncthreads();
        // accumulator layout: lane l of (row tile r, rank half h) holds n = 16 h + 4 (l >> 4) + e, m = 16 r + (l & 15).  Consecutive
        // threads take consecutive n of one row (64-byte runs of t)
#pragma unroll
        for (int o = tid; o < RT * 512; o += 256) {
            const int n = o & 31, m = o >> 5;
            const int r = m >> 4, ml = m & 15, h = n >> 4, q = n & 15, l = (q >> 2) * 16 + ml, e = q & 3;
            const int idx = ((r * 2 + h) * 4 + e) * 64 + l;
            const float sum = (part[idx] + part[WSTRIDE + idx]) + (part[2 * WSTRIDE + idx] + part[3 * WSTRIDE + idx]);
            const int gn = nt * 32 + n;
            if (m0 + m < M && gn < R) t[(m0 + m) * R + gn] = IS_BF16 ? f32_to_bf16_bits(sum) : f32_to_f16_bits(sum);
        }
        __syncthreads();  // partial sums consumed before the next n-tile's DMAs overwrite them
    }
}

int fill_params(const SdnqWeight* w, DeqParams& p) {
    if (!w || !w->weight || !w->scale) return SDNQ_ERR_NULL;
    const int pos = w->positions > 1 ? w->positions : 1;
    if (w->n <= 0 || w->k <= 0 || w->group_size <= 0 || (w->k % pos) != 0 || ((w->k / pos) % w->group_size) != 0) return SDNQ_ERR_SHAPE;
    if ((w->k % 16) != 0) return SDNQ_ERR_SHAPE;
    if (w->storage < 0 || w->storage > 3 || w->kind < 0 || w->kind > 3) return SDNQ_ERR_DTYPE;
    if (w->bits < 1 || w->bits > 16) return SDNQ_ERR_DTYPE;
    if (w->storage == SDNQ_ST_PACKED_U8 && w->bits > 7) return SDNQ_ERR_DTYPE;
    if (w->storage == SDNQ_ST_PACKED_I16 && (w->bits < 9 || w->bits > 15)) return SDNQ_ERR_DTYPE;
    if (w->storage == SDNQ_ST_RAW8 && w->bits != 8) return SDNQ_ERR_DTYPE;
    if (w->storage == SDNQ_ST_RAW16 && w->bits != 16) return SDNQ_ERR_DTYPE;
    if ((w->kind == SDNQ_KIND_UINT || w->kind == SDNQ_KIND_UFLOAT) && !w->zero_point) return SDNQ_ERR_NULL;
    if ((w->kind == SDNQ_KIND_FLOAT || w->kind == SDNQ_KIND_UFLOAT) && !w->native_float) {
        const int sign = (w->kind == SDNQ_KIND_FLOAT) ? 1 : 0;
        if (w->exponent < 1 || w->exponent > 7 || w->mantissa < 0 || sign + w->exponent + w->mantissa != w->bits) return SDNQ_ERR_DTYPE;
    }
    if ((uintptr_t)w->weight % 16) return SDNQ_ERR_ALIGN;
    if ((w->svd_up == nullptr) != (w->svd_down == nullptr)) return SDNQ_ERR_NULL;
    if (w->svd_up && (w->svd_rank <= 0 || w->svd_dtype < 0 || w->svd_dtype > 2)) return SDNQ_ERR_SHAPE;
    p.w = w->weight; p.scale = w->scale; p.zp = w->zero_point; p.svd_up = w->svd_up; p.svd_down = w->svd_down;
    p.N = w->n; p.K = w->k; p.group_size = w->group_size; p.G = (w->k / pos) / w->group_size; p.rank = w->svd_rank;
    p.P = pos; p.SG = p.G * pos;
    if (w->scale_dtype < 0 || w->scale_dtype > 2) return SDNQ_ERR_DTYPE;
    p.sdt = w->scale_dtype;
    p.fmt = WeightFmt{w->storage, w->kind, w->bits, w->exponent, w->mantissa, w->native_float};
    return SDNQ_OK;
}

}  // namespace

extern "C" int sdnq_hip_dequant(const SdnqWeight* w, int hadamard_group, void* out, int out_dtype, sdnq_stream_t stream) {
    DeqParams p{};
    int st = fill_params(w, p);
    if (st != SDNQ_OK) return st;
    if (!out) return SDNQ_ERR_NULL;
    if (out_dtype < 0 || out_dtype > 2) return SDNQ_ERR_DTYPE;
    if ((uintptr_t)out % 16) return SDNQ_ERR_ALIGN;
    hipStream_t s = (hipStream_t)stream;
    const int64_t units = p.N * (p.K / 16);
    dim3 grid((unsigned)((units + 255) / 256)), block(256);
    const int svd_t = p.svd_up ? w->svd_dtype : out_dtype;
#define DQ_CASE(O, S) \
    if (out_dtype == O && svd_t == S) hipLaunchKernelGGL((dequant_kernel<O, S>), grid, block, 0, s, p, out);
    DQ_CASE(SDNQ_F32, SDNQ_F32)
    else DQ_CASE(SDNQ_F32, SDNQ_BF16)
    else DQ_CASE(SDNQ_F32, SDNQ_F16)
    else DQ_CASE(SDNQ_BF16, SDNQ_BF16)
    else DQ_CASE(SDNQ_BF16, SDNQ_F32)
    else DQ_CASE(SDNQ_F16, SDNQ_F16)
    else DQ_CASE(SDNQ_F16, SDNQ_F32)
    else return SDNQ_ERR_DTYPE;
#undef DQ_CASE
    SDNQ_CHECK_LAUNCH();
    if (hadamard_group != 0) return sdnq_hip_hadamard(out, out_dtype, p.N, p.K, p.K, hadamard_group, out, p.K, stream);
    return SDNQ_OK;
}

// 4-bit packed weights in groups of a multiple of 64 go through the table kernel (bit-identical, ~2-3x fewer vector instructions);
// `ws_known` (row scales already in ws) is honoured by it and ignored -- the scales are simply recomputed -- by the general kernel
static int launch_requant(const DeqParams& p, int mm_dtype, void* wq, float* ws, int ws_known, hipStream_t s, void* lut = nullptr) {
    dim3 grid((unsigned)((p.N + 3) / 4)), block(256);
    // test / tuning aid, read per call (a re-quantization is a whole-weight pass: a getenv is nothing beside it) so that one
    // process can A/B the table kernel against the general one (tests/test_gpu_parity.py)
    const char* lut_env = getenv("SDNQ_HIP_REQUANT_LUT");
    const bool no_lut = lut_env && atoi(lut_env) == 0;
    const bool lut_ok = p.fmt.storage == SDNQ_ST_PACKED_U8 && p.fmt.bits == 4 && p.P == 1 && (p.group_size % 64) == 0 && (p.K % 64) == 0 && !p.fmt.native_float;
    const bool use_lut = lut_ok && (!no_lut || lut != nullptr);
    const int np = (int)((p.K + 1023) / 1024);
    if (lut != nullptr && !(lut_ok && np <= 16)) return SDNQ_ERR_UNSUPPORTED;  // tables exist for what the table kernel handles
#define LUT_NP(MMV, NPV) hipLaunchKernelGGL((requant_lut4_kernel<MMV, NPV>), grid, block, 0, s, p, (uint8_t*)wq, ws, ws_known, (u32*)lut)
#define LUT_CASES(MMV)                                                                                   \
    switch (np) {                                                                                        \
        case 1: LUT_NP(MMV, 1); break;   case 2: LUT_NP(MMV, 2); break;   case 3: LUT_NP(MMV, 3); break;   \
        case 4: LUT_NP(MMV, 4); break;   case 5: LUT_NP(MMV, 5); break;   case 6: LUT_NP(MMV, 6); break;   \
        case 7: case 8: LUT_NP(MMV, 8); break;                                                           \
        case 9: case 10: case 11: case 12: LUT_NP(MMV, 12); break;                                       \
        default: LUT_NP(MMV, 16); break;                                                                 \
    }
    if (mm_dtype == SDNQ_MM_I8) {
        if (use_lut && np <= 16) { LUT_CASES(SDNQ_MM_I8) }
        else hipLaunchKernelGGL((requant_kernel<SDNQ_MM_I8>), grid, block, 0, s, p, (uint8_t*)wq, ws, (float*)nullptr);
    } else if (mm_dtype == SDNQ_MM_FP8) {
        if (use_lut && np <= 16) { LUT_CASES(SDNQ_MM_FP8) }
        else hipLaunchKernelGGL((requant_kernel<SDNQ_MM_FP8>), grid, block, 0, s, p, (uint8_t*)wq, ws, (float*)nullptr);
    } else {
        return SDNQ_ERR_DTYPE;
    }
#undef LUT_CASES
#undef LUT_NP
    SDNQ_CHECK_LAUNCH();
    return SDNQ_OK;
}

extern "C" int sdnq_hip_requant(const SdnqWeight* w, int mm_dtype, void* wq, float* ws, sdnq_stream_t stream) {
    DeqParams p{};
    int st = fill_params(w, p);
    if (st != SDNQ_OK) return st;
    if (!wq || !ws) return SDNQ_ERR_NULL;
    if ((uintptr_t)wq % 16) return SDNQ_ERR_ALIGN;
    p.svd_up = nullptr; p.svd_down = nullptr;  // re_quantize_matmul never receives the SVD factors (linear_int8.py:105)
    hipStream_t s = (hipStream_t)stream;
    return launch_requant(p, mm_dtype, wq, ws, 0, s);
}

extern "C" int sdnq_hip_requant_ws(const SdnqWeight* w, int mm_dtype, void* wq, float* ws, int ws_known, sdnq_stream_t stream) {
    DeqParams p{};
    int st = fill_params(w, p);
    if (st != SDNQ_OK) return st;
    if (!wq || !ws) return SDNQ_ERR_NULL;
    if ((uintptr_t)wq % 16) return SDNQ_ERR_ALIGN;
    p.svd_up = nullptr; p.svd_down = nullptr;
    return launch_requant(p, mm_dtype, wq, ws, ws_known, (hipStream_t)stream);
}

extern "C" int sdnq_hip_lut4_build(const SdnqWeight* w, int mm_dtype, float* ws, int ws_known, void* lut, sdnq_stream_t stream) {
    DeqParams p{};
    int st = fill_params(w, p);
    if (st != SDNQ_OK) return st;
    if (!ws || !lut) return SDNQ_ERR_NULL;
    if ((uintptr_t)lut % 16) return SDNQ_ERR_ALIGN;
    p.svd_up = nullptr; p.svd_down = nullptr;  // as sdnq_hip_requant
    return launch_requant(p, mm_dtype, nullptr, ws, ws_known, (hipStream_t)stream, lut);
}

extern "C" int sdnq_hip_requant_asym(const SdnqWeight* w, void* wq, float* ws, float* wzp, sdnq_stream_t stream) {
    DeqParams p{};
    int st = fill_params(w, p);
    if (st != SDNQ_OK) return st;
    if (!wq || !ws || !wzp) return SDNQ_ERR_NULL;
    if ((uintptr_t)wq % 16) return SDNQ_ERR_ALIGN;
    p.svd_up = nullptr; p.svd_down = nullptr;  // as sdnq_hip_requant (linear_uint8.py:110)
    // (16-bit scales: get_scale_asymmetric and the quotient round in the scale dtype, see requant_kernel)
    dim3 grid((unsigned)((p.N + 3) / 4)), block(256);
    hipLaunchKernelGGL((requant_kernel<SDNQ_MM_I8, true>), grid, block, 0, (hipStream_t)stream, p, (uint8_t*)wq, ws, wzp);
    SDNQ_CHECK_LAUNCH();
    return SDNQ_OK;
}

extern "C" int sdnq_hip_unpack_mm(const SdnqWeight* w, int mm_dtype, void* wq, sdnq_stream_t stream) {
    DeqParams p{};
    int st = fill_params(w, p);
    if (st != SDNQ_OK) return st;
    if (!wq) return SDNQ_ERR_NULL;
    if ((uintptr_t)wq % 16) return SDNQ_ERR_ALIGN;
    if (mm_dtype == SDNQ_MM_I8) {
        if ((p.fmt.kind != SDNQ_KIND_INT && p.fmt.kind != SDNQ_KIND_UINT) || p.fmt.bits > 8) return SDNQ_ERR_DTYPE;
    } else if (mm_dtype == SDNQ_MM_FP8) {
        if (p.fmt.kind != SDNQ_KIND_FLOAT || p.fmt.bits > 8) return SDNQ_ERR_DTYPE;
    } else if (mm_dtype == SDNQ_MM_F16) {
        if (p.fmt.kind != SDNQ_KIND_FLOAT || p.fmt.bits > 16) return SDNQ_ERR_DTYPE;
    } else {
        return SDNQ_ERR_DTYPE;
    }
    hipStream_t s = (hipStream_t)stream;
    const int64_t units = p.N * p.K / 16;
    dim3 grid((unsigned)((units + 255) / 256)), block(256);
    if (mm_dtype == SDNQ_MM_F16) {
        hipLaunchKernelGGL(unpack_mm_f16_kernel, grid, block, 0, s, p, (uint16_t*)wq);
        SDNQ_CHECK_LAUNCH();
        return SDNQ_OK;
    }
    if (mm_dtype == SDNQ_MM_I8) hipLaunchKernelGGL((unpack_mm_kernel<SDNQ_MM_I8>), grid, block, 0, s, p, (uint8_t*)wq);
    else hipLaunchKernelGGL((unpack_mm_kernel<SDNQ_MM_FP8>), grid, block, 0, s, p, (uint8_t*)wq);
    SDNQ_CHECK_LAUNCH();
    return SDNQ_OK;
}

int sdnq_float_gemm(const void* x, const void* w, const void* bias, int dtype, void* out, int64_t m, int64_t n, int64_t k,
                    int64_t ldx, hipStream_t s, void* const* outs = nullptr, int n_outs = 0, int64_t seg_n = 0, int64_t ldc = 0);  // gemm.hip

extern "C" int sdnq_hip_linear_float_strided(const void* x, const void* wd, const void* bias, int dtype, void* out, int64_t m,
                                             int64_t n, int64_t k, int64_t ldx, int64_t ldc, sdnq_stream_t stream) {
    if (!x || !wd || !out) return SDNQ_ERR_NULL;
    if (dtype < 0 || dtype > 2) return SDNQ_ERR_DTYPE;
    const int eb = (dtype == SDNQ_F32) ? 4 : 2;
    if (m <= 0 || n <= 0 || k <= 0 || ldx < k || ldc < n || ((k * eb) % 16) != 0) return SDNQ_ERR_SHAPE;
    if (((uintptr_t)x % 16) || ((uintptr_t)wd % 16) || ((ldx * eb) % 16)) return SDNQ_ERR_ALIGN;
    hipStream_t s = (hipStream_t)stream;
    // more than a few rows: the MFMA GEMM of gemm.hip (bf16 / f16 / f32 matrix cores); its stores are 8 channels wide
    if (m > 32 && (n % 8) == 0 && ((uintptr_t)out % 16) == 0 && ((ldc * eb) % 16) == 0)
        return sdnq_float_gemm(x, wd, bias, dtype, out, m, n, k, ldx, s, nullptr, 0, 0, ldc);
    constexpr int MC = 8;
    dim3 grid((unsigned)((n + 3) / 4), (unsigned)((m + MC - 1) / MC)), block(256);
    switch (dtype) {
        case SDNQ_F32: hipLaunchKernelGGL((linear_float_kernel<SDNQ_F32, MC>), grid, block, 0, s, x, wd, bias, out, m, n, k, ldx, ldc); break;
        case SDNQ_BF16: hipLaunchKernelGGL((linear_float_kernel<SDNQ_BF16, MC>), grid, block, 0, s, x, wd, bias, out, m, n, k, ldx, ldc); break;
        default: hipLaunchKernelGGL((linear_float_kernel<SDNQ_F16, MC>), grid, block, 0, s, x, wd, bias, out, m, n, k, ldx, ldc); break;
    }
    SDNQ_CHECK_LAUNCH();
    return SDNQ_OK;
}

extern "C" int sdnq_hip_linear_float(const void* x, const void* wd, const void* bias, int dtype, void* out, int64_t m,
                                     int64_t n, int64_t k, int64_t ldx, sdnq_stream_t stream) {
    return sdnq_hip_linear_float_strided(x, wd, bias, dtype, out, m, n, k, ldx, n, stream);
}

extern "C" int sdnq_hip_lowrank_down(const void* x, int x_dtype, int64_t m, int64_t k, int64_t ldx, const void* svd_down,
                                     int svd_dtype, int rank, void* t, sdnq_stream_t stream) {
    // t = mm(x.to(svd dtype), svd_down): x already lives in the activation dtype; the reference casts x to
    // svd_down.dtype first (linear_int8.py:60) -- both are the model dtype, so require equality.
    if (x_dtype != svd_dtype) return SDNQ_ERR_DTYPE;
    if (x_dtype != SDNQ_F32 && (k % 16) == 0 && rank > 0 && x && svd_down && t && ((uintptr_t)x % 16) == 0 &&
        ((uintptr_t)svd_down % 16) == 0 && ((ldx * 2) % 16) == 0) {
        hipStream_t s = (hipStream_t)stream;
        // row tiles per workgroup: whichever leaves the busiest CU fewer bytes to move (see the kernel; 3 : 4 = bytes per workgroup and stage)
        static const int rt_env = [] { const char* e = getenv("SDNQ_HIP_LRD_RT"); return e ? atoi(e) : 0; }();  // tuning aid: 1 / 2
        const int64_t wg1 = (m + 15) / 16, wg2 = (m + 31) / 32;
        const int64_t cus = 256;
        const bool two = rt_env ? rt_env == 2 : ((wg2 + cus - 1) / cus) * 4 < ((wg1 + cus - 1) / cus) * 3;
        dim3 grid((unsigned)(two ? wg2 : wg1)), block(256);
#define LRD(BF, RTV) hipLaunchKernelGGL((lowrank_down_kernel<BF, RTV>), grid, block, 0, s, (const uint16_t*)x, (const uint16_t*)svd_down, (uint16_t*)t, m, k, ldx, rank)
        if (x_dtype == SDNQ_BF16) { if (two) LRD(true, 2); else LRD(true, 1); }
        else { if (two) LRD(false, 2); else LRD(false, 1); }
#undef LRD
        SDNQ_CHECK_LAUNCH();
        return SDNQ_OK;
    }
    return sdnq_hip_linear_float(x, svd_down, nullptr, x_dtype, t, m, rank, k, ldx, stream);
}

extern "C" int sdnq_hip_linear_skinny_svd(const SdnqWeight* w, const void* svd_down_t, const void* x, const void* bias, int dtype,
                                          void* out, int64_t m, int64_t ldx, sdnq_stream_t stream) {
    DeqParams p{};
    int st = fill_params(w, p);
    if (st != SDNQ_OK) return st;
    if (!x || !out || !svd_down_t || !w->svd_up) return SDNQ_ERR_NULL;
    if (dtype != SDNQ_BF16 && dtype != SDNQ_F16) return SDNQ_ERR_DTYPE;
    if (w->svd_dtype != dtype) return SDNQ_ERR_DTYPE;
    if (p.sdt != SDNQ_F32 && p.sdt != dtype) return SDNQ_ERR_DTYPE;  // 16-bit scales: q * s is rounded to the scale dtype = svd dtype here
    const bool int_fmt = p.fmt.kind == SDNQ_KIND_INT || p.fmt.kind == SDNQ_KIND_UINT;
    const bool raw8 = p.fmt.storage == SDNQ_ST_RAW8 && int_fmt, pk4 = p.fmt.storage == SDNQ_ST_PACKED_U8 && p.fmt.bits == 4 && int_fmt;
    if (!(raw8 || pk4) || (p.group_size % 4) != 0 || p.P != 1) return SDNQ_ERR_UNSUPPORTED;
    if (m <= 0 || m > 4 || ldx < p.K || (p.K % 32) != 0 || p.rank <= 0 || (p.rank % 16) != 0) return SDNQ_ERR_SHAPE;
    if (((uintptr_t)svd_down_t % 16) || ((uintptr_t)w->svd_up % 16)) return SDNQ_ERR_ALIGN;
    const size_t lds = (size_t)(m <= 1 ? 1 : (m <= 2 ? 2 : 4)) * p.K * sizeof(float);
    if (lds > 150 * 1024) return SDNQ_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    dim3 grid((unsigned)((p.N + 31) / 32)), block(256);
    // skinny_svd32_kernel: x as 16-bit elements + the four private DMA rings + the rows' scales / zero points
    const size_t lds32 = lds / 2 + 4 * 4 * 3072 + (size_t)32 * p.G * 8;
    if (p.rank == 32 && lds32 <= 150 * 1024 && (p.group_size % 16) == 0 && p.G <= 64 && (p.K % 32) == 0 && (raw8 || (p.K % 64) == 0) &&
        ((uintptr_t)x % 16) == 0 && ((ldx * 2) % 16) == 0) {
#define S32_LAUNCH(B, MR)                                                                                                    \
    do {                                                                                                                     \
        auto kern = raw8 ? skinny_svd32_kernel<B, MR, 8> : skinny_svd32_kernel<B, MR, 4>;                                    \
        if (lds32 > 64 * 1024 && hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess) \
            return SDNQ_ERR_LAUNCH;                                                                                          \
        hipLaunchKernelGGL(kern, grid, block, lds32, s, p, (const uint16_t*)svd_down_t, x, bias, out, m, ldx);               \
    } while (0)
#define S32_DISPATCH(B)            \
    do {                           \
        if (m <= 1) S32_LAUNCH(B, 1); \
        else if (m <= 2) S32_LAUNCH(B, 2); \
        else S32_LAUNCH(B, 4);     \
    } while (0)
        if (dtype == SDNQ_BF16) S32_DISPATCH(true);
        else S32_DISPATCH(false);
#undef S32_DISPATCH
#undef S32_LAUNCH
        SDNQ_CHECK_LAUNCH();
        return SDNQ_OK;
    }
#define SS_LAUNCH(B, MR)                                                                                                     \
    do {                                                                                                                     \
        auto kern = raw8 ? skinny_svd_kernel<B, MR, 8> : skinny_svd_kernel<B, MR, 4>;                                        \
        if (lds > 64 * 1024 && hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess) \
            return SDNQ_ERR_LAUNCH;                                                                                          \
        hipLaunchKernelGGL(kern, grid, block, lds, s, p, (const uint16_t*)svd_down_t, x, bias, out, m, ldx);                 \
    } while (0)
#define SS_DISPATCH(B)            \
    do {                          \
        if (m <= 1) SS_LAUNCH(B, 1); \
        else if (m <= 2) SS_LAUNCH(B, 2); \
        else SS_LAUNCH(B, 4);     \
    } while (0)
    if (dtype == SDNQ_BF16) SS_DISPATCH(true);
    else SS_DISPATCH(false);
#undef SS_DISPATCH
#undef SS_LAUNCH
    SDNQ_CHECK_LAUNCH();
    return SDNQ_OK;
}

extern "C" int sdnq_hip_linear_skinny(const SdnqWeight* w, int hadamard_group, const void* x, const void* bias, int dtype, void* out,
                                      int64_t m, int64_t ldx, sdnq_stream_t stream) {
    DeqParams p{};
    int st = fill_params(w, p);
    if (st != SDNQ_OK) return st;
    if (!x || !out) return SDNQ_ERR_NULL;
    if (dtype < 0 || dtype > 2) return SDNQ_ERR_DTYPE;
    if (m <= 0 || m > 64 || ldx < p.K) return SDNQ_ERR_SHAPE;
    if (w->svd_up) return SDNQ_ERR_UNSUPPORTED;  // the SVD term needs the dequantize-then-GEMM path
    int log2had = 0;
    if (hadamard_group != 0) {
        if (hadamard_group < 4 || hadamard_group > 512 || (hadamard_group & (hadamard_group - 1)) || (p.K % hadamard_group)) return SDNQ_ERR_SHAPE;
        while ((1 << log2had) < hadamard_group) ++log2had;
    }
    const int eb = (dtype == SDNQ_F32) ? 4 : 2;
    if (((uintptr_t)x % 16) || ((ldx * eb) % 16)) return SDNQ_ERR_ALIGN;
    hipStream_t s = (hipStream_t)stream;
    {
        const bool int_fmt = p.fmt.kind == SDNQ_KIND_INT || p.fmt.kind == SDNQ_KIND_UINT;
        const bool raw8 = p.fmt.storage == SDNQ_ST_RAW8 && int_fmt, pk4 = p.fmt.storage == SDNQ_ST_PACKED_U8 && p.fmt.bits == 4 && int_fmt;
        static const int had_mfma = [] { const char* e = getenv("SDNQ_HIP_HADAMARD_MFMA"); return e ? atoi(e) : 1; }();
        if ((raw8 || pk4) && m <= 4 && p.P == 1 && hadamard_group == 256 && dtype != SDNQ_F32 && (p.group_size % 4) == 0 && had_mfma &&
            (p.sdt == SDNQ_F32 || p.sdt == dtype) && ((uintptr_t)x % 8) == 0 && ((ldx * 2) % 8) == 0) {
            dim3 grid((unsigned)((p.N + 3) / 4)), block(256);
#define SH_LAUNCH(T, B, MR) hipLaunchKernelGGL((linear_skinny_had256_kernel<T, B, MR>), grid, block, 0, s, p, x, bias, out, m, ldx)
#define SH_M(T, B) do { if (m == 1) SH_LAUNCH(T, B, 1); else if (m == 2) SH_LAUNCH(T, B, 2); else SH_LAUNCH(T, B, 4); } while (0)
#define SH_T(B) do { if (dtype == SDNQ_BF16) SH_M(SDNQ_BF16, B); else SH_M(SDNQ_F16, B); } while (0)
            if (raw8) SH_T(8);
            else SH_T(4);
#undef SH_T
#undef SH_M
#undef SH_LAUNCH
            SDNQ_CHECK_LAUNCH();
            return SDNQ_OK;
        }
        // (the fast kernel rounds q * s straight to the activation dtype: with 16-bit scales that is the scale dtype's rounding too)
        if ((raw8 || pk4) && m <= 4 && p.P == 1 && (p.group_size % 16) == 0 && (p.K % 16) == 0 && (p.sdt == SDNQ_F32 || p.sdt == dtype)) {
            dim3 grid((unsigned)((p.N + 3) / 4)), block(256);
#define SF_LAUNCH(T, B, MR) hipLaunchKernelGGL((linear_skinny_fast_kernel<T, B, MR>), grid, block, 0, s, p, x, bias, out, m, ldx, log2had)
#define SF_M(T, B)                       \
    do {                                 \
        if (m == 1) SF_LAUNCH(T, B, 1);  \
        else if (m == 2) SF_LAUNCH(T, B, 2); \
        else SF_LAUNCH(T, B, 4);         \
    } while (0)
#define SF_T(B)                                      \
    do {                                             \
        if (dtype == SDNQ_F32) SF_M(SDNQ_F32, B);    \
        else if (dtype == SDNQ_BF16) SF_M(SDNQ_BF16, B); \
        else SF_M(SDNQ_F16, B);                      \
    } while (0)
            if (raw8) SF_T(8);
            else SF_T(4);
#undef SF_T
#undef SF_M
#undef SF_LAUNCH
            SDNQ_CHECK_LAUNCH();
            return SDNQ_OK;
        }
    }
#define SK_LAUNCH(T, MR)                                                                                      \
    hipLaunchKernelGGL((linear_skinny_kernel<T, MR>), dim3((unsigned)((p.N + 3) / 4), (unsigned)((m + MR - 1) / MR)), \
                       dim3(256), 0, s, p, x, bias, out, m, ldx, log2had)
#define SK_DISPATCH(T)                 \
    do {                               \
        if (m == 1) SK_LAUNCH(T, 1);   \
        else if (m <= 2) SK_LAUNCH(T, 2); \
        else if (m <= 4) SK_LAUNCH(T, 4); \
        else SK_LAUNCH(T, 8);          \
    } while (0)
    switch (dtype) {
        case SDNQ_F32: SK_DISPATCH(SDNQ_F32); break;
        case SDNQ_BF16: SK_DISPATCH(SDNQ_BF16); break;
        default: SK_DISPATCH(SDNQ_F16); break;
    }
#undef SK_DISPATCH
#undef SK_LAUNCH
    SDNQ_CHECK_LAUNCH();
    return SDNQ_OK;
}
