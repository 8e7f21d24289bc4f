// Load-time weight quantizer for gfx950: float [N][K] weight -> codes in the reference's storage
// format + f32 group scales (+ zero points).
//
//   pass 1  quant_stats_kernel : per (row, group) min / max / amax -> scale, zero_point   (quant_utils.py:10-24)
//   pass 2  quant_pack_kernel  : q = w / s  or  (w - zp) / s ; ints: round-half-even, clamp; floats: nan_to_num, clamp,
//                                convert (quant_utils.py:28-56) ; then the reference's packing:
//                                signed ints stored as value - min (packed_int/__init__.py:77-80), bit-interleaved group
//                                codecs (packed_int/pack.py), eXmY float encoder with its own rounding rule
//                                (packed_float.py:27-82).
//
// The bit placement of every packed format is NOT written out a second time: the host derives it by probing the decoders
// of unpack_dev.h one word bit at a time (each word bit feeds exactly one element bit), so packer and unpacker cannot
// disagree; the goldens captured from the reference pin both.
//
// HBM-bound, run once per layer: N*K*(src bytes) read twice (2nd pass mostly L2-hot per row block) + N*K*bits/8 written.
#include <hip/hip_runtime.h>

#include "../../include/sdnq_hip.h"
#include "sdnq_dev.h"
#include "unpack_dev.h"

namespace {

struct PackTable {
    // word bit i (bit i&7 of byte i>>3, or bit i&15 of 16-bit word i>>4) <- bit eb[i] of element el[i] (0xff: unused)
    uint8_t el[240];
    uint8_t eb[240];
    // element a word holds at shift 0 (0xff: none).  The reference ORs those in UNMASKED (packed_int/pack.py, e.g. :115
    // `packed[:, :8] | (packed[:, 8:] << 11)`): a code that does not fit in `bits` bits -- uint9..15 have max = 2^bits in
    // the dtype table and the largest element of every asymmetric group quantizes to exactly that -- leaks its high bits
    // into the word.  Reproduced so that the bytes equal the reference's.
    uint8_t base[16];
    int nbits;
};

struct QuantParams {
    const void* src;
    int64_t ld, N, K;
    int group_size, G;
    int P, SG;  // conv: kernel positions per channel (1 for Linear), scales per output row = G * P
    void* q;
    float* scale;
    float* zp;
    WeightFmt fmt;
    float qmin, qmax;
};

template <int LANES>
__device__ __forceinline__ float sub_max(float v) {
#pragma unroll
    for (int o = LANES / 2; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
template <int LANES>
__device__ __forceinline__ float sub_min(float v) {
#pragma unroll
    for (int o = LANES / 2; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
    return v;
}

// LANES consecutive lanes own one (row, group)
template <int SRC_T, int LANES>
__global__ __launch_bounds__(256) void quant_stats_kernel(const QuantParams p) {
    const int64_t gid = ((int64_t)blockIdx.x * 256 + threadIdx.x) / LANES;
    const int l = threadIdx.x % LANES;
    const bool live = gid < p.N * p.SG;
    const int64_t n = live ? gid / p.SG : 0;
    const int sg = live ? (int)(gid % p.SG) : 0;
    // Linear: group g = sg, elements contiguous.  Conv (P > 1): sg = (channel group, position); the group's elements are the
    // group's channels at that kernel position: k = (cg * group_size + j) * P + pos
    const int64_t base = n * p.ld + (int64_t)(sg / p.P) * p.group_size * p.P + (sg % p.P);
    float lo = __builtin_inff(), hi = -__builtin_inff(), amax = 0.0f;
    if (live) {
        for (int j = l; j < p.group_size; j += LANES) {
            const float v = FT<SRC_T>::load(p.src, base + (int64_t)j * p.P);
            lo = fminf(lo, v);
            hi = fmaxf(hi, v);
            amax = fmaxf(amax, fabsf(v));
        }
    }
    lo = sub_min<LANES>(lo);
    hi = sub_max<LANES>(hi);
    amax = sub_max<LANES>(amax);
    if (!live || l != 0) return;
    const bool asym = (p.fmt.kind == SDNQ_KIND_UINT || p.fmt.kind == SDNQ_KIND_UFLOAT);
    if (asym) {
        const float s = (hi - lo) / (p.qmax - p.qmin);
        p.scale[gid] = s;
        p.zp[gid] = (p.qmin == 0.0f) ? lo : lo - s * p.qmin;
    } else {
        p.scale[gid] = amax / p.qmax;
    }
}

__device__ __forceinline__ uint8_t f32_to_e5m2(float f) {  // RNE, |f| <= 57344 (clamped by the caller)
    u32 u = __float_as_uint(f);
    const u32 sign = (u >> 24) & 0x80u;
    u &= 0x7fffffffu;
    u32 r;
    if (u < 0x38800000u) {  // |f| < 2^-14: subnormal, quantum 2^-16
        r = (u32)__builtin_rintf(__uint_as_float(u) * 65536.0f);
    } else {
        u += 0xfffffu + ((u >> 21) & 1u);
        r = (u >> 21) - (112u << 2);
    }
    return (uint8_t)(sign | r);
}

// f32 (already clamped to the format's range) -> eXmY code; restates packed_float.py:27-73 step by step, including its
// rounding rule (round up only if the top four dropped bits exceed one half) and the exponent re-bias by bit surgery.
__device__ __forceinline__ u32 encode_exmy(float x, int ebits, int mbits, bool is_unsigned) {
    const int drop = 23 - mbits;
    u32 bits = __float_as_uint(x);
    const u32 low = (drop > 4) ? ((1u << (drop - 4)) - 1u) : 0u;
    const u32 top4 = bits & (((1u << drop) - 1u) & ~low);
    if (top4 > (1u << (drop - 1))) bits += (1u << drop);
    const u32 sign = bits >> 31;
    const int bias = (1 << (ebits - 1)) - 1;
    const float min_normal = __uint_as_float((u32)(127 + 1 - bias) << 23);
    const float mag = fabsf(__uint_as_float(bits));
    if (mag < min_normal) {
        // integer mantissa on the 2^(1-bias-M) grid; a carry out of the mantissa lands in exponent bit 0
        const float grid = __uint_as_float((u32)(127 + mbits - 1 + bias) << 23);  // 2^M / min_normal
        bits = ((u32)(int)__builtin_rintf(mag * grid)) << drop;
    }
    const u32 exp8 = (bits >> 23) & 0xffu;
    const u32 mant = (bits >> drop) & ((1u << mbits) - 1u);
    const u32 new_exp = ((exp8 >> 7) << (ebits - 1)) | (exp8 & ((1u << (ebits - 1)) - 1u));
    u32 code = (new_exp << mbits) | mant;
    if (!is_unsigned) code |= sign << (ebits + mbits);
    return code & ((1u << (ebits + mbits + (is_unsigned ? 0 : 1))) - 1u);
}

// one thread = 16 consecutive elements of one row (K % 16 == 0), codes staged in LDS for table-driven bit placement
template <int SRC_T>
__global__ __launch_bounds__(256) void quant_pack_kernel(const QuantParams p, const PackTable t) {
    __shared__ u32 codes[256][17];
    const int64_t units_per_row = p.K / 16;
    const int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (u >= p.N * units_per_row) return;
    const int64_t n = u / units_per_row, k0 = (u % units_per_row) * 16;
    const WeightFmt f = p.fmt;
    const bool asym = (f.kind == SDNQ_KIND_UINT || f.kind == SDNQ_KIND_UFLOAT);
    const bool is_int = (f.kind == SDNQ_KIND_INT || f.kind == SDNQ_KIND_UINT);
    const bool packed = (f.storage == SDNQ_ST_PACKED_U8 || f.storage == SDNQ_ST_PACKED_I16);
    u32* c = codes[threadIdx.x];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const float w = FT<SRC_T>::load(p.src, n * p.ld + k0 + j);
        const int kk = (int)(k0 + j), cc = kk / p.P;
        const int64_t gi = n * p.SG + (p.P > 1 ? (cc / p.group_size) * p.P + (kk - cc * p.P) : kk / p.group_size);
        const float s = p.scale[gi];
        float q = asym ? (w - p.zp[gi]) / s : w / s;
        u32 code;
        if (is_int) {
            // 0/0 of an all-zero group: round and clamp keep the NaN, the integer cast of the reference turns it into 0
            const bool is_nan = q != q;
            q = fminf(fmaxf(__builtin_rintf(q), p.qmin), p.qmax);
            int iv = is_nan ? 0 : (int)q;
            if (f.kind == SDNQ_KIND_INT && packed) iv -= (int)p.qmin;  // stored as value - min
            code = (u32)iv;
        } else {
            if (q != q) q = 0.0f;                                      // nan_to_num_
            else if (q == __builtin_inff()) q = 3.4028234663852886e38f;
            else if (q == -__builtin_inff()) q = -3.4028234663852886e38f;
            q = fminf(fmaxf(q, p.qmin), p.qmax);
            if (f.native_float) {
                if (f.bits == 8) code = (f.ebits == 4) ? f32_to_e4m3fn(q) : f32_to_e5m2(q);
                else code = (f.ebits == 5) ? f32_to_f16_bits(q) : f32_to_bf16_bits(q);
            } else {
                code = encode_exmy(q, f.ebits, f.mbits, f.kind == SDNQ_KIND_UFLOAT);
            }
        }
        c[j] = code;
    }
    const int64_t e0 = n * p.K + k0;
    if (f.storage == SDNQ_ST_RAW8) {
        uint8_t* o = (uint8_t*)p.q + e0;
        u32 ww[4] = {0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < 16; ++j) ww[j >> 2] |= (c[j] & 0xffu) << (8 * (j & 3));
        *(uint4*)o = make_uint4(ww[0], ww[1], ww[2], ww[3]);
    } else if (f.storage == SDNQ_ST_RAW16) {
        uint16_t* o = (uint16_t*)p.q + e0;
#pragma unroll
        for (int j = 0; j < 16; ++j) o[j] = (uint16_t)c[j];
    } else if (f.storage == SDNQ_ST_PACKED_U8) {
        uint8_t* o = (uint8_t*)p.q + (e0 >> 3) * f.bits;
        for (int h = 0; h < 2; ++h) {
            for (int b = 0; b < f.bits; ++b) {
                u32 byte = 0;
                for (int i = 0; i < 8; ++i) {
                    const int wi = b * 8 + i;
                    if (t.el[wi] != 0xff) byte |= ((c[8 * h + t.el[wi]] >> t.eb[wi]) & 1u) << i;
                }
                if (t.base[b] != 0xff) byte |= c[8 * h + t.base[b]] & 0xffu & ~((1u << f.bits) - 1u);
                o[h * f.bits + b] = (uint8_t)byte;
            }
        }
    } else {
        uint16_t* o = (uint16_t*)p.q + (e0 >> 4) * f.bits;
        for (int b = 0; b < f.bits; ++b) {
            u32 word = 0;
            for (int i = 0; i < 16; ++i) {
                const int wi = b * 16 + i;
                if (t.el[wi] != 0xff) word |= ((c[t.el[wi]] >> t.eb[wi]) & 1u) << i;
            }
            if (t.base[b] != 0xff) word |= c[t.base[b]] & 0xffffu & ~((1u << f.bits) - 1u);
            o[b] = (uint16_t)word;
        }
    }
}

int build_table(int storage, int bits, PackTable& t) {
    for (int i = 0; i < 240; ++i) { t.el[i] = 0xff; t.eb[i] = 0; }
    t.nbits = 0;
    if (storage == SDNQ_ST_PACKED_U8) {
        t.nbits = 8 * bits;
        for (int i = 0; i < t.nbits; ++i) {
            uint8_t w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            w[i >> 3] = (uint8_t)(1u << (i & 7));
            u32 e[8];
            unpack8_u8(w, bits, e);
            for (int j = 0; j < 8; ++j) {
                const u32 v = e[j] & ((1u << bits) - 1u);
                if (v) { t.el[i] = (uint8_t)j; t.eb[i] = (uint8_t)__builtin_ctz(v); break; }
            }
        }
    } else if (storage == SDNQ_ST_PACKED_I16) {
        t.nbits = 16 * bits;
        for (int i = 0; i < t.nbits; ++i) {
            uint16_t w[16] = {0};
            w[i >> 4] = (uint16_t)(1u << (i & 15));
            u32 e[16];
            unpack16_i16(w, bits, e);
            for (int j = 0; j < 16; ++j) {
                const u32 v = e[j] & ((1u << bits) - 1u);
                if (v) { t.el[i] = (uint8_t)j; t.eb[i] = (uint8_t)__builtin_ctz(v); break; }
            }
        }
    }
    const int wbits = (storage == SDNQ_ST_PACKED_U8) ? 8 : 16;
    for (int w = 0; w < 16; ++w) t.base[w] = 0xff;
    for (int w = 0; w < bits && t.nbits; ++w) {
        const int e = t.el[w * wbits];
        bool is_base = e != 0xff;
        for (int b = 0; b < bits && is_base; ++b) is_base = (t.el[w * wbits + b] == e && t.eb[w * wbits + b] == b);
        if (is_base) t.base[w] = (uint8_t)e;
    }
    // every element bit must be fed by exactly one word bit
    if (t.nbits) {
        int seen[16] = {0};
        for (int i = 0; i < t.nbits; ++i)
            if (t.el[i] != 0xff) seen[t.el[i]] |= 1 << t.eb[i];
        const int elems = (storage == SDNQ_ST_PACKED_U8) ? 8 : 16;
        for (int j = 0; j < elems; ++j)
            if (seen[j] != (1 << bits) - 1) return SDNQ_ERR_UNSUPPORTED;
    }
    return SDNQ_OK;
}

template <int SRC_T>
int launch(const QuantParams& p, const PackTable& t, hipStream_t s) {
    const int64_t groups = p.N * p.SG;
    const int gs = p.group_size;
#define STATS(L)                                                                                              \
    hipLaunchKernelGGL((quant_stats_kernel<SRC_T, L>), dim3((unsigned)((groups * L + 255) / 256)), dim3(256), 0, s, p)
    if (gs >= 512) STATS(64);
    else if (gs >= 128) STATS(16);
    else if (gs >= 32) STATS(4);
    else STATS(1);
#undef STATS
    SDNQ_CHECK_LAUNCH();
    const int64_t units = p.N * (p.K / 16);
    hipLaunchKernelGGL((quant_pack_kernel<SRC_T>), dim3((unsigned)((units + 255) / 256)), dim3(256), 0, s, p, t);
    SDNQ_CHECK_LAUNCH();
    return SDNQ_OK;
}

}  // namespace

extern "C" int sdnq_hip_quantize_weight(const void* src, int src_dtype, int64_t ld_src, const SdnqWeight* w, float qmin,
                                        float qmax, sdnq_stream_t stream) {
    if (!src || !w || !w->weight || !w->scale) return SDNQ_ERR_NULL;
    if (src_dtype < 0 || src_dtype > 2) return SDNQ_ERR_DTYPE;
    const int pos = w->positions > 1 ? w->positions : 1;
    if (w->n <= 0 || w->k <= 0 || w->group_size <= 0 || (w->k % pos) != 0 || ((w->k / pos) % w->group_size) != 0 || (w->k % 16) != 0) return SDNQ_ERR_SHAPE;
    if (w->storage < 0 || w->storage > 3 || w->kind < 0 || w->kind > 3 || w->bits < 1 || w->bits > 16) return SDNQ_ERR_DTYPE;
    if (w->storage == SDNQ_ST_PACKED_U8 && w->bits > 7) return SDNQ_ERR_DTYPE;
    if (w->storage == SDNQ_ST_PACKED_I16 && (w->bits < 9 || w->bits > 15)) return SDNQ_ERR_DTYPE;
    if (w->storage == SDNQ_ST_RAW8 && w->bits != 8) return SDNQ_ERR_DTYPE;
    if (w->storage == SDNQ_ST_RAW16 && w->bits != 16) return SDNQ_ERR_DTYPE;
    const bool asym = (w->kind == SDNQ_KIND_UINT || w->kind == SDNQ_KIND_UFLOAT);
    const bool is_float = (w->kind == SDNQ_KIND_FLOAT || w->kind == SDNQ_KIND_UFLOAT);
    if (asym && !w->zero_point) return SDNQ_ERR_NULL;
    if (is_float && !w->native_float) {
        const int sign = (w->kind == SDNQ_KIND_FLOAT) ? 1 : 0;
        if (w->exponent < 1 || w->exponent > 7 || w->mantissa < 0 || sign + w->exponent + w->mantissa != w->bits) return SDNQ_ERR_DTYPE;
    }
    if (is_float && w->native_float) {
        const bool ok = (w->bits == 8 && ((w->exponent == 4 && w->mantissa == 3) || (w->exponent == 5 && w->mantissa == 2))) ||
                        (w->bits == 16 && ((w->exponent == 5 && w->mantissa == 10) || (w->exponent == 8 && w->mantissa == 7)));
        if (!ok) return SDNQ_ERR_UNSUPPORTED;
    }
    if (!(qmax > qmin)) return SDNQ_ERR_SHAPE;
    if ((uintptr_t)w->weight % 16) return SDNQ_ERR_ALIGN;
    QuantParams p{};
    p.src = src; p.ld = ld_src; p.N = w->n; p.K = w->k; p.group_size = w->group_size; p.G = (w->k / pos) / w->group_size; p.P = pos; p.SG = p.G * pos;
    p.q = const_cast<void*>(w->weight); p.scale = const_cast<float*>(w->scale); p.zp = const_cast<float*>(w->zero_point);
    p.fmt = WeightFmt{w->storage, w->kind, w->bits, w->exponent, w->mantissa, w->native_float};
    p.qmin = qmin; p.qmax = qmax;
    PackTable t;
    int st = build_table(w->storage, w->bits, t);
    if (st != SDNQ_OK) return st;
    hipStream_t s = (hipStream_t)stream;
    if (src_dtype == SDNQ_F32) return launch<SDNQ_F32>(p, t, s);
    if (src_dtype == SDNQ_BF16) return launch<SDNQ_BF16>(p, t, s);
    return launch<SDNQ_F16>(p, t, s);
}
