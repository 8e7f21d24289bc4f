// Quantized attention forward for gfx950: int8 Q.K^T on the matrix cores, softmax in fp32, P.V in the
// value dtype (bf16 / f16 MFMA) -- the default configuration of the reference's `sdnq_triton_atten`
// (kernels/triton_atten.py:540-618: matmul_dtype="int8", pv_matmul_dtype=None, smooth_k=True).
//
//   sdnq_hip_attn_prepare <- quantize_attn (triton_atten.py:443-487): K minus its per-channel token mean (smooth_k), per-token
//                            symmetric int8 of Q and K (quantize_int_mm, quant_utils.py:265-273), plus V^T for the PV operand
//   sdnq_hip_attn_fwd     <- sdnq_attn_kernel (triton_atten.py:143-335) with qk_is_quantized=1, pv_is_quantized=0
//
// Register layout of the forward kernel (one wave = 32 queries, no LDS, no barriers):
//   S^T = K.Q^T with K as the first MFMA operand: lane l owns query (l & 31); its 16 accumulator registers are 16 keys of the
//   32-key block, the other 16 live in lane l ^ 32.  Row statistics are therefore in-lane reductions plus ONE lane exchange.
//   The K fragment rows are taken in a permuted order (bits 2 and 3 of the row index swapped) so that registers 8c..8c+7 of
//   lane group g hold the CONTIGUOUS keys 16c + 8g .. + 7: exactly the K-slice that lane feeds to PV MFMA c.  P never moves
//   between lanes.  sdnq_hip_attn_prepare stores quantized K and V already in MFMA-fragment order (1-KiB tiles), so every
//   fragment load of the loop is one fully coalesced 16-bytes-per-lane access.
//   O^T = V^T.P^T: query stays on (l & 31), so alpha / 1/l are per-lane scalars.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <type_traits>

#include "../../include/sdnq_hip.h"
#include "sdnq_dev.h"
#include "hadamard_dev.h"
#include "quant8_dev.h"

namespace {

// element strides of a [batch][heads][tokens][head_dim] tensor whose head_dim is contiguous (e.g. the transposed view of a
// [batch][tokens][heads * head_dim] projection output); `heads` splits a linear batch*heads index
struct Strides {
    int64_t b, h, n, heads;
    __device__ __forceinline__ int64_t at(int64_t head_lin, int64_t tok) const {
        int64_t zb, hh;
        divmod(head_lin, heads, zb, hh);
        return zb * b + hh * h + tok * n;
    }
};

// ---- K channel sums over the tokens of one (batch, head), split over KMEAN_SPLITS workgroups ----------------------------
// part[head][split][d] = sum of the split's tokens; the consumer adds the splits in a fixed order (deterministic mean).
constexpr int KMEAN_SPLITS = 32;

template <int T_ID>
__device__ __forceinline__ void attn_kmean_block(const void* __restrict__ k, const Strides ks, float* __restrict__ part, int64_t kn, int d, int d_src,
                                                 int64_t block, float* red /* 256 * 8 floats of LDS */) {
    const int lpr = d / 8, rpp = 256 / lpr;  // lanes per token row, rows per pass
    const int tid = threadIdx.x, lsh = d == 64 ? 3 : 4, c8 = (tid & (lpr - 1)) * 8, r0 = tid >> lsh;  // d is 64 or 128
    const int64_t head = block / KMEAN_SPLITS, split = block % KMEAN_SPLITS;  // (KMEAN_SPLITS is a power of two: shifts)
    const int64_t per = (kn + KMEAN_SPLITS - 1) / KMEAN_SPLITS, lo = split * per, hi = lo + per < kn ? lo + per : kn;
    const uint16_t* khead = (const uint16_t*)k + ks.at(head, 0);
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int64_t r = lo + r0; r < hi; r += rpp) {
        float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (c8 < d_src) Vec16<T_ID>::unpack(*(const uint4*)(khead + r * ks.n + c8), v);  // channels past d_src: zero padding
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += v[e];
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[tid * 8 + e] = acc[e];
    __syncthreads();
    if (tid < d) {
        float s = 0.0f;
        const int lane_of = tid / 8, e = tid % 8;
        for (int r = 0; r < rpp; ++r) s += red[(r * lpr + lane_of) * 8 + e];
        part[block * d + tid] = s;
    }
}

// ---- channel means of ONE head's K by one 256-thread workgroup (short key sequences: no separate channel-sum section) --------
// Thread tid owns channels [8 (tid % lpr), + 8) of the rows tid / lpr, + 256 / lpr, ...: `acc` = its sums in row order.  xw: 4 * 128 floats,
// smean: 128 floats of LDS.  One summation order for every caller (the prepare kernel's inline form and the single-launch attention), so
// both routes produce the same K codes.
__device__ __forceinline__ void attn_means_reduce(float (&acc)[8], int64_t kn, int d, float* xw, float* smean) {
    const int lpr = d / 8, tid = threadIdx.x, c8 = (tid & (lpr - 1)) * 8;  // d is 64 or 128
    // the lanes of a wave that hold the same channels are lpr apart (lane exchange on the VALU / swizzle paths, hadamard_dev.h)
    if (lpr == 8) {
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += lane_xor(acc[e], 8);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] += lane_xor(acc[e], 16);
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] += lane_xor(acc[e], 32);
    if ((tid & 63) < lpr) {
#pragma unroll
        for (int e = 0; e < 8; ++e) xw[(tid >> 6) * 128 + c8 + e] = acc[e];
    }
    __syncthreads();
    if (tid < d) smean[tid] = ((xw[tid] + xw[128 + tid]) + (xw[256 + tid] + xw[384 + tid])) / (float)kn;  // k.mean(dim=2), triton_atten.py:459
    __syncthreads();
}
template <int T_ID>
__device__ __forceinline__ void attn_head_means(const void* __restrict__ k, const Strides kst, int64_t head, int64_t kn, int d, int d_src, float* xw, float* smean) {
    const int lpr = d / 8, lsh = d == 64 ? 3 : 4, rpp = 256 >> lsh, tid = threadIdx.x, c8 = (tid & (lpr - 1)) * 8;  // d is 64 or 128
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const uint16_t* khead = (const uint16_t*)k + kst.at(head, 0);
    for (int64_t r = tid >> lsh; r < kn; r += rpp) {
        float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (c8 < d_src) Vec16<T_ID>::unpack(*(const uint4*)(khead + r * kst.n + c8), v);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += v[e];
    }
    attn_means_reduce(acc, kn, d, xw, smean);
}

// per-token symmetric int8 of one row spread over lpr lanes (8 channels each): codes of this lane's 8 channels, the row's scale
__device__ __forceinline__ float attn_quant8(const float (&v)[8], int lpr, u32 (&o)[2]) {
    float amax = 0.0f;
#pragma unroll
    for (int e = 0; e < 8; ++e) amax = fmaxf(amax, fabsf(v[e]));
    amax = fmaxf(amax, lane_xor(amax, 1));
    amax = fmaxf(amax, lane_xor(amax, 2));
    amax = fmaxf(amax, lane_xor(amax, 4));
    if (lpr == 16) amax = fmaxf(amax, lane_xor(amax, 8));
    const float scale = amax / 127.0f;
    RowDiv rd;  // the correctly rounded 3-instruction division (sdnq_dev.h) when every row of the wave has an ordinary scale
    rd.set(scale);
    if (__all(rd.fast)) {
#pragma unroll
        for (int w = 0; w < 2; ++w)
            o[w] = pack4_rne_i8(fastdiv2((pv2f){v[4 * w], v[4 * w + 1]}, rd), fastdiv2((pv2f){v[4 * w + 2], v[4 * w + 3]}, rd));
        return scale;
    }
    // a zero row (scale 0: 0 / 0 must become code 0), a non-finite one or an extreme scale somewhere in the wave: the plain sequence
    o[0] = o[1] = 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        float q = __builtin_rintf(v[e] / scale);
        if (q != q) q = 0.0f;
        q = fminf(fmaxf(q, -128.0f), 127.0f);
        o[e >> 2] |= ((u32)(int)q & 0xffu) << (8 * (e & 3));
    }
    return scale;
}
// byte offset of the 8 codes (key n, channels [c8, c8 + 8)) inside one head's K operand in MFMA-fragment order: one 1-KiB tile per (32-key
// block, 32-channel step), lane (g, rho) holds the 16 bytes [32 kk + 16 g, +16) of key pi(rho) (pi = swap bits 2 and 3, see
// attn_fwd_kernel) -> every fragment load of the forward kernel is one fully coalesced 1-KiB access
__device__ __forceinline__ int64_t attn_kfrag_offset(int64_t n, int c8, int d) {
    const int kk = c8 >> 5, g = (c8 >> 4) & 1, half = (c8 >> 3) & 1, nl = (int)(n & 31);
    const int rho = (nl & 0x13) | ((nl & 4) << 1) | ((nl & 8) >> 1);
    return ((n / 32) * (d / 32) + kk) * 1024 + (g * 32 + rho) * 16 + half * 8;
}

// ---- per-token int8 quantization of [heads][n_src][d] (d / 8 lanes per token), optional mean subtraction ------------------
// The destination has n_dst >= n_src token slots per head (K: rounded up to the 32-key block; the extra tokens get zero codes
// and a zero scale and are masked in the forward kernel).  `mean`: this head's channel means (LDS) or nullptr.
template <int T_ID>
__device__ __forceinline__ void attn_quant_block(const void* __restrict__ x, const Strides xst, const float* mean, int8_t* __restrict__ xq, float* __restrict__ xs,
                                                 int64_t heads, int64_t n_src, int64_t n_dst, int d, bool frag_major, int64_t block,
                                                 int log2g, int d_src) {
    const int lpr = d / 8, lsh = d == 64 ? 3 : 4;  // d is 64 or 128
    const int64_t t = block * 256 + threadIdx.x;
    const int64_t row = t >> lsh;
    const int c8 = (int)(t & (lpr - 1)) * 8;
    const bool live = row < heads * n_dst;
    int64_t head = 0, n = 0;
    if (live) divmod(row, n_dst, head, n);
    const bool real = live && n < n_src;
    float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (real && c8 < d_src) Vec16<T_ID>::unpack(*(const uint4*)((const uint16_t*)x + xst.at(head, n) + c8), v);  // 8 elements per 16-byte load; head dims
    // that are not 64 / 128 are zero-padded to the next one (get_attn_inputs pads to the next power of two, triton_atten.py:514-519)
    if (mean != nullptr && real && c8 < d_src) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] -= mean[c8 + e];  // k.to(float32).sub_(mean), triton_atten.py:459-463
    }
    if (log2g != 0) {
        // apply_hadamard(q) / rotate_hadamard(k.to(hadamard.dtype)) (triton_atten.py:464-467): x.view(.., D/g, g) @ H_g in the
        // tensor dtype; this lane's 8 channels are elements lane*8 + e of the wave, groups of g <= D are aligned segments
        if (mean != nullptr) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = FT<T_ID>::round(v[e]);
        }
        wave_hadamard(v, log2g, hadamard_scale(log2g, T_ID));
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = FT<T_ID>::round(v[e]);
    }
    u32 o[2];
    const float scale = attn_quant8(v, lpr, o);
    if (live) {
        if (frag_major) {  // K operand in MFMA-fragment order (n_dst is a multiple of 32)
            *(uint2*)(xq + head * n_dst * d + attn_kfrag_offset(n, c8, d)) = make_uint2(o[0], o[1]);
        } else {
            *(uint2*)(xq + row * d + c8) = make_uint2(o[0], o[1]);
        }
        if (c8 == 0) xs[row] = scale;
    }
}

// ---- V [heads][kn][d] -> PV operand in MFMA-fragment order (knp = kn rounded up to 32, zero padded) -----------------------
// one 1-KiB tile per (32-key block kb, 32-channel block dd, 16-key step c): lane (g, ql) holds the 8 keys
// kb*32 + 16c + 8g + 0..7 of channel 32dd + ql, i.e. exactly the first operand of PV MFMA (dd, c) of attn_fwd_kernel.
template <int PITCH>
__device__ __forceinline__ void attn_vt_block(const uint16_t* __restrict__ v, const Strides vst, uint16_t* __restrict__ vt, int64_t kn, int64_t knp, int d,
                                              int64_t head, int64_t kb, uint16_t (*tile)[PITCH], int d_src) {
    const int64_t key0 = kb * 32;
    const int lpr = d / 8, kkn = d / 32;
    uint16_t* dst = vt + (head * (knp / 32) + kb) * (int64_t)(kkn * 2 * 512);
    const int lsh = d == 64 ? 3 : 4;  // d is 64 or 128
    const uint16_t* vhead = v + vst.at(head, 0);
    for (int t = threadIdx.x; t < 32 * lpr; t += 256) {
        const int kr = t >> lsh, c8 = (t & (lpr - 1)) * 8;
        uint4 val = make_uint4(0, 0, 0, 0);
        if (key0 + kr < kn && c8 < d_src) val = *(const uint4*)(vhead + (key0 + kr) * vst.n + c8);
        const uint16_t* h = (const uint16_t*)&val;
#pragma unroll
        for (int e = 0; e < 8; ++e) tile[kr][c8 + e] = h[e];
    }
    __syncthreads();
    for (int t = threadIdx.x; t < kkn * 2 * 64; t += 256) {
        const int lane = t & 63, c = (t >> 6) & 1, dd = t >> 7;
        const int dch = 32 * dd + (lane & 31), k8 = 16 * c + 8 * (lane >> 5);
        u32 w[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) w[e] = (u32)tile[k8 + 2 * e][dch] | ((u32)tile[k8 + 2 * e + 1][dch] << 16);
        *(uint4*)(dst + (int64_t)t * 8) = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

struct PrepParams {
    const void *q, *k, *v;
    int8_t *qq, *kq;
    float *qs, *ks;
    uint16_t* vt;
    Strides qst, kst, vst;
    const float* kpart;  // [kheads][KMEAN_SPLITS][d] channel sums (smooth_k) or nullptr
    float* kpart_out;    // the same table, written by the channel-sum section of a launch
    int64_t qheads, kheads, qn, kn, knp, nqb, nkb, nvb, nmb;  // workgroups per section: Q rows, K rows, V tiles, K channel sums
    int d, d_src, log2g;  // d: head dim padded to 64 / 128, d_src: the tensors' head dim; log2g: log2 of the Hadamard group (0 = none)
    bool smooth_inline;  // K means computed inside the K workgroups (short key sequences)
};

// one launch, four kinds of workgroup: [0, nqb) quantize Q, the next nkb quantize K (a workgroup never straddles heads: knp is a
// multiple of the 32 / 16 tokens it covers), the next nvb lay out V, the last nmb sum K's channels over a token split.  Any count may be 0:
// with smooth_k over a long key sequence the K rows wait for the channel sums, so the host launches {Q, V, sums} and then {K}
// (round 4; before, the sums were a launch of their own -- 4.7 us of latency for 5 MB -- and everything else waited for it).
template <int T_ID>
__global__ __launch_bounds__(256) void attn_prepare_kernel(const PrepParams p) {
    SDNQ_KERNARGS_NOW("s"(p.q), "s"(p.k), "s"(p.v), "s"(p.qq), "s"(p.kq), "s"(p.qs), "s"(p.ks), "s"(p.vt), "s"(p.kpart), "s"(p.kpart_out), "s"(p.qheads), "s"(p.kheads), "s"(p.qn), "s"(p.kn),
                      "s"(p.knp), "s"(p.nqb), "s"(p.nkb), "s"(p.nvb), "s"(p.nmb), "s"(p.d), "s"(p.d_src), "s"(p.log2g));
    __shared__ float smean[128];
    __shared__ __attribute__((aligned(16))) uint16_t tile[32][128 + 2];
    const int64_t b = blockIdx.x;
    if (b < p.nqb) {
        attn_quant_block<T_ID>(p.q, p.qst, nullptr, p.qq, p.qs, p.qheads, p.qn, p.qn, p.d, false, b, p.log2g, p.d_src);
    } else if (b < p.nqb + p.nkb) {
        const int64_t kb = b - p.nqb;
        const float* mean = nullptr;
        if (p.smooth_inline) {
            // short key sequences (cross-attention onto text tokens): every K workgroup sums its head's channels itself
            // instead of a separate attn_kmean_kernel launch
            int64_t head, rem;
            divmod(kb * (256 / (p.d / 8)), p.knp, head, rem);
            attn_head_means<T_ID>(p.k, p.kst, head, p.kn, p.d, p.d_src, (float*)&tile[0][0], smean);
            mean = smean;
        } else if (p.kpart != nullptr) {
            int64_t head, rem;
            divmod(kb * (256 / (p.d / 8)), p.knp, head, rem);
            if ((int)threadIdx.x < p.d) {
                const float* pp = p.kpart + head * KMEAN_SPLITS * p.d + threadIdx.x;
                float s = 0.0f;
                for (int i = 0; i < KMEAN_SPLITS; ++i) s += pp[i * p.d];
                smean[threadIdx.x] = s / (float)p.kn;  // k.mean(dim=2), triton_atten.py:459
            }
            __syncthreads();
            mean = smean;
        }
        attn_quant_block<T_ID>(p.k, p.kst, mean, p.kq, p.ks, p.kheads, p.kn, p.knp, p.d, true, kb, p.log2g, p.d_src);
    } else if (b < p.nqb + p.nkb + p.nvb) {
        int64_t vhead, vblk;
        divmod(b - p.nqb - p.nkb, p.knp >> 5, vhead, vblk);
        attn_vt_block((const uint16_t*)p.v, p.vst, p.vt, p.kn, p.knp, p.d, vhead, vblk, tile, p.d_src);
    } else {
        attn_kmean_block<T_ID>(p.k, p.kst, p.kpart_out, p.kn, p.d, p.d_src, b - p.nqb - p.nkb - p.nvb, (float*)&tile[0][0]);  // 256 * 8 floats <= sizeof(tile)
    }
}

struct AttnParams {
    const int8_t* qq; const float* qs; const int8_t* kq; const float* ks; const uint16_t* vt;
    void* out;
    int64_t qh, kh, qn, kn, knp;
    int qblocks, split;
    int shared_kv;  // the 4 waves of a workgroup (4 query tiles of one head) stream K / V through LDS once instead of 4 times
    float log2_sm_scale;
    Strides ost;       // output strides (elements)
    int d_out;         // channels the output tensor has (<= D: padded head dims)
    const void* mask;  // attention mask [*, *, q, key] (key stride 1) or nullptr
    int mask_dtype;    // -1: int8 / bool (0 = masked out), else SdnqFloat of an additive mask
    int64_t ms_z, ms_h, ms_q;  // element strides (0 for broadcast dimensions)
    // Q in the tensor dtype, quantized per token by the wave that owns the 32 queries (sdnq_hip_attn_fwd_q16); nullptr: qq / qs hold it.
    // A query is read by one tile only, so quantizing it ahead costs a 16-bit read, an 8-bit write and an 8-bit read where this is one 16-bit read.
    const void* q_src;
    Strides qst;
    int d_src;
    // K and V in the tensor dtype too (sdnq_hip_attn, at most 128 keys: cross-attention onto text tokens): every workgroup builds its head's
    // quantized K, the V operand and the key scales in LDS -- the whole attention is ONE launch
    const void* k_src;
    const void* v_src;
    Strides kst, vst;
    int smooth;
};

__device__ __forceinline__ float ldf_mask(const void* p, int64_t i, int dt) {
    return dt == SDNQ_F32 ? ((const float*)p)[i] : (dt == SDNQ_BF16 ? bf16_bits_to_f32(((const uint16_t*)p)[i]) : f16_bits_to_f32(((const uint16_t*)p)[i]));
}

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

typedef float v2f __attribute__((ext_vector_type(2)));
typedef __bf16 v2bf __attribute__((ext_vector_type(2)));
typedef _Float16 v2h __attribute__((ext_vector_type(2)));

// ---- forward: one wave = 32 queries of one head; 4 waves per workgroup -------------------------------------------------
// The loop is VALU-bound (16 scores per lane and 32-key block against 2 + 4 MFMAs at D = 64), so the score arithmetic is kept
// to cvt, one packed multiply (k_scale * q_scale * log2(e) * sm_scale folded per key), max3, packed subtract, exp2, packed add
// and a packed convert; masks exist only in the one block that needs them (key tail / causal diagonal), the running sum stays
// split over the two lanes of a query until the end, and O is rescaled only when some row maximum of the wave moved.
// Software pipeline: the K fragments of block kb+1 are already in registers when block kb starts, so S(kb+1) is on the matrix
// pipe while the VALU works on the scores of kb; V / k_scale of kb+1 and K of kb+2 are in flight meanwhile.
template <int V_T, int OUT_T, int D, bool CAUSAL, bool HAS_MASK>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const AttnParams p) {
    // one batch of kernarg loads (SDNQ_KERNARGS_NOW, sdnq_dev.h)
    SDNQ_KERNARGS_NOW("s"(p.qq), "s"(p.qs), "s"(p.kq), "s"(p.ks), "s"(p.vt), "s"(p.out), "s"(p.qh), "s"(p.kh), "s"(p.qn), "s"(p.kn), "s"(p.knp), "s"(p.qblocks), "s"(p.split),
                      "s"(p.shared_kv), "s"(p.log2_sm_scale), "s"(p.ost.b), "s"(p.ost.h), "s"(p.ost.n), "s"(p.ost.heads), "s"(p.d_out), "s"(p.mask), "s"(p.mask_dtype),
                      "s"(p.ms_z), "s"(p.ms_h), "s"(p.ms_q), "s"(p.q_src), "s"(p.qst.b), "s"(p.qst.h), "s"(p.qst.n), "s"(p.qst.heads), "s"(p.d_src), "s"(p.k_src), "s"(p.v_src));
    constexpr int KK = D / 32;  // int8 MFMA K steps of Q.K^T; also the 32-channel blocks of O
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave index in SGPRs: block indices stay scalar
    const int ql = lane & 31, g = lane >> 5;
    // workgroup b runs on XCD b % 8 (private L2 each): give every XCD a CONTIGUOUS range of the (head, query block) sequence, so
    // that the query blocks of one head -- which all stream the same K / V -- share one L2 instead of filling all eight
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x, q = nwg / 8, r = nwg % 8, xcd = bid % 8, j = bid / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    }
    const int64_t head_lin = bid / p.qblocks;  // z * QH + h
    const int qblk = bid % p.qblocks;
    // split == 1: the 4 waves of a workgroup take 4 query tiles.  split == S in {2, 4} (few tiles for the 1024 SIMDs of the chip):
    // 4 / S query tiles per workgroup, waves S*t .. S*t + S-1 each take 1/S of the key blocks of tile t and merge through LDS.
    const int part = p.split > 1 ? (wave & (p.split - 1)) : 0;
    const int wtile = p.split == 4 ? 0 : (p.split == 2 ? wave >> 1 : wave);  // query tile of this wave inside the workgroup
    const int64_t q0 = ((int64_t)qblk * (4 / p.split) + wtile) * 32;
    const bool active = q0 < p.qn;  // wave-uniform
    const bool inline_kv = p.k_src != nullptr;  // (then split == 1 and no shared_kv streaming)
    if (!active && p.split == 1 && !p.shared_kv && !inline_kv) return;  // (shared / inline K / V: every wave is needed for the loads and barriers)
    int64_t z, h;
    divmod(head_lin, p.qh, z, h);
    const int64_t mz = z, mh = h;  // attention-mask batch / head index (strides are 0 where the mask broadcasts)
    int64_t kvh, kvr;
    divmod(h * p.kh, p.qh, kvh, kvr);
    const int64_t kv_lin = z * p.kh + kvh;  // offset_k of triton_atten.py:212 (grouped-query mapping)

    const int64_t qi = q0 + ql, qrow = qi < p.qn ? qi : p.qn - 1;
    // inline K / V: thread tid owns channels [8 (tid % LPR), + 8) of the keys tid / LPR + i * RPP.  ALL loads of K and V are issued here, in
    // front of the Q loads, so the kernel has ONE memory round trip in front of its arithmetic
    constexpr int LPR = D / 8, RPP = 256 / LPR, NR = 128 / RPP;
    const int ic8 = ((int)threadIdx.x % LPR) * 8, ir0 = (int)threadIdx.x / LPR;
    uint4 kraw[NR], vraw[NR];
    if (inline_kv) {
        const uint16_t* kh = (const uint16_t*)p.k_src + p.kst.at(kv_lin, 0) + ic8;
        const uint16_t* vh = (const uint16_t*)p.v_src + p.vst.at(kv_lin, 0) + ic8;
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int64_t r = ir0 + i * RPP;
            kraw[i] = vraw[i] = make_uint4(0, 0, 0, 0);
            if (ic8 < p.d_src && r < p.kn) {
                kraw[i] = *(const uint4*)(kh + r * p.kst.n);
                vraw[i] = *(const uint4*)(vh + r * p.vst.n);
            }
        }
    }
    v4i qf[KK];
    float qsl;
    if (p.q_src != nullptr) {
        // per-token symmetric int8 of this lane's query (quantize_int_mm, quant_utils.py:265-273; the arithmetic of attn_quant8): the lane
        // holds channels [32 kk + 16 g, + 16) of every K step, its partner lane ^ 32 the other half of the row
        const uint16_t* qp = (const uint16_t*)p.q_src + p.qst.at(head_lin, qrow);
        uint4 raw[KK][2];
#pragma unroll
        for (int kk = 0; kk < KK; ++kk)
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                const int c8 = 32 * kk + 16 * g + 8 * hf;
                raw[kk][hf] = make_uint4(0, 0, 0, 0);
                if (c8 < p.d_src) raw[kk][hf] = *(const uint4*)(qp + c8);  // channels past the tensors' head dim: zero padding
            }
        // |x| of bf16 / f16 values orders like the unsigned integer of its low 15 bits: the row maximum on packed 16-bit integers
        typedef unsigned short us2 __attribute__((ext_vector_type(2)));
        us2 mx = {0, 0};
#pragma unroll
        for (int kk = 0; kk < KK; ++kk)
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                const u32 w[4] = {raw[kk][hf].x, raw[kk][hf].y, raw[kk][hf].z, raw[kk][hf].w};
#pragma unroll
                for (int i = 0; i < 4; ++i) mx = __builtin_elementwise_max(mx, __builtin_bit_cast(us2, w[i] & 0x7fff7fffu));
            }
        u32 ab = mx[0] > mx[1] ? mx[0] : mx[1];
        {
            const auto sw = __builtin_amdgcn_permlane32_swap(ab, ab, false, false);
            ab = sw[0] > sw[1] ? sw[0] : sw[1];
        }
        const float amax = V_T == SDNQ_BF16 ? __uint_as_float(ab << 16) : f16_bits_to_f32((uint16_t)ab);
        const float scale = amax / 127.0f;
        RowDiv dv;
        dv.set(scale);
        if (__all(dv.fast)) {
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) {
                float v[16];
                Vec16<V_T>::unpack(raw[kk][0], v);
                Vec16<V_T>::unpack(raw[kk][1], v + 8);
                u32 o4[4];
#pragma unroll
                for (int w = 0; w < 4; ++w)
                    o4[w] = pack4_rne_i8(fastdiv2((pv2f){v[4 * w], v[4 * w + 1]}, dv), fastdiv2((pv2f){v[4 * w + 2], v[4 * w + 3]}, dv));
                qf[kk] = (v4i){(int)o4[0], (int)o4[1], (int)o4[2], (int)o4[3]};
            }
        } else {  // a zero row (0 / 0 -> code 0), a non-finite one or an extreme scale somewhere in the wave
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) {
                float v[16];
                Vec16<V_T>::unpack(raw[kk][0], v);
                Vec16<V_T>::unpack(raw[kk][1], v + 8);
                u32 o4[4] = {0, 0, 0, 0};
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    float qv = __builtin_rintf(v[e] / scale);
                    if (qv != qv) qv = 0.0f;
                    qv = fminf(fmaxf(qv, -128.0f), 127.0f);
                    o4[e >> 2] |= ((u32)(int)qv & 0xffu) << (8 * (e & 3));
                }
                qf[kk] = (v4i){(int)o4[0], (int)o4[1], (int)o4[2], (int)o4[3]};
            }
        }
        qsl = scale * p.log2_sm_scale;
    } else {
        auto rsQ = SDNQ_MAKE_RSRC(p.qq + head_lin * p.qn * D);  // (buffer form: see the K / V loads below)
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) qf[kk] = SDNQ_BUF_LOAD16(rsQ, (int)qrow * D + 16 * g, 32 * kk);
        // ((acc * q_scale) * k_scale) * log2_sm_scale of triton_atten.py:278 as acc * (k_scale * (q_scale * log2_sm_scale))
        qsl = p.qs[head_lin * p.qn + qrow] * p.log2_sm_scale;
    }

    v16f o[KK];
#pragma unroll
    for (int dd = 0; dd < KK; ++dd)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dd][r] = 0.0f;
    float m_i = -__builtin_inff();
    v2f l2 = {0.0f, 0.0f};  // this lane's half of the row sum (16 of every 32 keys), two partial sums

    // K and V arrive in MFMA-fragment order (sdnq_hip_attn_prepare): 1-KiB tiles, lane l reads bytes [16 l, 16 l + 16)
    // Every load of the loop goes through a buffer descriptor (head base in SGPRs, constant 16-byte-per-lane offset, block offset in
    // the scalar operand): a global load with a 64-bit VGPR address waits ~1000 cycles at issue while another wave of the SIMD
    // streams MFMAs (sdnq_dev.h; the loop has 10 vector-memory instructions per key block).
    auto rsK = SDNQ_MAKE_RSRC(p.kq + kv_lin * p.knp * D);
    auto rsV = SDNQ_MAKE_RSRC(p.vt + kv_lin * p.knp * D);
    auto rsS = SDNQ_MAKE_RSRC(p.ks + kv_lin * p.knp);
    const int lofs = lane * 16, sofs = 32 * g;  // byte offsets of this lane inside a fragment tile / a block's 32 k_scales

    struct Blk { v4i v[KK][2]; v4f ks[4]; };
    auto load_k = [&](int kb, v4i (&kf)[KK]) {
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) kf[kk] = SDNQ_BUF_LOAD16(rsK, lofs, kb * (KK * 1024) + kk * 1024);
    };
    auto load_vs = [&](int kb, Blk& b) {
#pragma unroll
        for (int dd = 0; dd < KK; ++dd)
#pragma unroll
            for (int c = 0; c < 2; ++c) b.v[dd][c] = SDNQ_BUF_LOAD16(rsV, lofs, kb * (KK * 2048) + (dd * 2 + c) * 1024);
        // registers 8c..8c+7 <-> keys key0 + 16c + 8g + 0..7 (k_scale rows are padded to knp, so this never leaves the row)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            b.ks[2 * c] = __builtin_bit_cast(v4f, SDNQ_BUF_LOAD16(rsS, sofs, kb * 128 + 64 * c));
            b.ks[2 * c + 1] = __builtin_bit_cast(v4f, SDNQ_BUF_LOAD16(rsS, sofs, kb * 128 + 64 * c + 16));
        }
    };
    auto qk_mfma = [&](const v4i (&kf)[KK]) {
        v16i s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0;
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) s = __builtin_amdgcn_mfma_i32_32x32x32_i8(kf[kk], qf[kk], s, 0, 0, 0);
        return s;
    };
    auto softmax_pv = [&](const v16i& s, const Blk& b, int64_t key0, auto maskedc) {
        constexpr bool MASKED = decltype(maskedc)::value;
        // t = acc * k_scale; the per-query factor qsl >= 0 commutes with the row maximum, so it is applied inside the fma that
        // also subtracts the maximum: p = exp2(t * qsl - m)
        v2f t[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const v4f k4 = b.ks[j >> 1];
            t[j] = (v2f){(float)s[2 * j], (float)s[2 * j + 1]} * (v2f){k4[2 * (j & 1)], k4[2 * (j & 1) + 1]};
        }
        float alpha;
        v2f psum = {0.0f, 0.0f};
        if constexpr (MASKED) {
            // slow path (key tail, causal diagonal, attention mask): fully scaled scores first, then the masks, and the -inf-safe
            // updates of triton_atten.py:299-301 (a row that has seen no visible key yet keeps m = -inf, alpha = 1, p = 0)
#pragma unroll
            for (int j = 0; j < 8; ++j) t[j] *= qsl;
            const char* mrow = nullptr;
            if constexpr (HAS_MASK) {
                const int64_t mq = qi < p.qn ? qi : p.qn - 1;
                mrow = (const char*)p.mask + (mz * p.ms_z + mh * p.ms_h + mq * p.ms_q) * (p.mask_dtype == -1 ? 1 : (p.mask_dtype == SDNQ_F32 ? 4 : 2));
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t key = key0 + 16 * (r >> 3) + 8 * g + (r & 7);
                bool ok = key < p.kn;              // triton_atten.py:295-296
                if (CAUSAL) ok = ok && key <= qi;  // :287-288
                float add = 0.0f;
                if constexpr (HAS_MASK) {
                    if (ok) {
                        if (p.mask_dtype == -1) ok = ((const int8_t*)mrow)[key] != 0;                 // :290-291
                        else add = ldf_mask(mrow, key, p.mask_dtype);                                  // :292-293 (added as is)
                    }
                }
                t[r >> 1][r & 1] = ok ? t[r >> 1][r & 1] + add : -__builtin_inff();
            }
            float m_blk = fmaxf(t[0][0], t[0][1]);
#pragma unroll
            for (int j = 1; j < 8; ++j) m_blk = fmaxf(fmaxf(m_blk, t[j][0]), t[j][1]);
            {
                const u32 mb = __float_as_uint(m_blk);
                const auto sw = __builtin_amdgcn_permlane32_swap(mb, mb, false, false);
                m_blk = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
            }
            const float m_new = fmaxf(m_i, m_blk);
            const bool dead = m_new == -__builtin_inff();
            alpha = dead ? 1.0f : __builtin_amdgcn_exp2f(m_i - m_new);
            m_i = m_new;
            const float m_use = dead ? 0.0f : m_new;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                t[j] -= m_use;
                t[j] = (v2f){__builtin_amdgcn_exp2f(t[j][0]), __builtin_amdgcn_exp2f(t[j][1])};
                psum += t[j];
            }
        } else {
            float m_blk = fmaxf(t[0][0], t[0][1]);
#pragma unroll
            for (int j = 1; j < 8; ++j) m_blk = fmaxf(fmaxf(m_blk, t[j][0]), t[j][1]);
            {   // the other 16 keys of this query live in lane ^ 32: one v_permlane32_swap (VALU) instead of a trip through LDS
                const u32 mb = __float_as_uint(m_blk);
                const auto sw = __builtin_amdgcn_permlane32_swap(mb, mb, false, false);
                m_blk = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
            }
            // lazy reference maximum: m_i follows the row maximum only when it grew by more than 2^8 for some query of the wave
            // (first block: from -inf).  Otherwise p = 2^(s - m_i) <= 2^8 -- harmless in fp32 sums and in the bf16 / f16 P operand
            // -- and alpha = 1: no exp2, no rescale of O.  o / l is the same quotient either way (the reference rescales every
            // block, triton_atten.py:303-309; only fp32 rounding order differs).
            const float m_cand = m_blk * qsl;  // finite: every key of a plain block is visible
            alpha = 1.0f;
            if (__builtin_amdgcn_ballot_w64(m_cand > m_i + 8.0f) != 0) {
                const float m_new = fmaxf(m_i, m_cand);
                alpha = __builtin_amdgcn_exp2f(m_i - m_new);
                m_i = m_new;
            }
            const v2f qsl2 = {qsl, qsl}, mneg2 = {-m_i, -m_i};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                t[j] = __builtin_elementwise_fma(t[j], qsl2, mneg2);
                t[j] = (v2f){__builtin_amdgcn_exp2f(t[j][0]), __builtin_amdgcn_exp2f(t[j][1])};
                psum += t[j];
            }
        }
        l2 = l2 * alpha + psum;  // l_i = fma(l_i, alpha, sum(p)), :308
        if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {
#pragma unroll
            for (int dd = 0; dd < KK; ++dd) o[dd] *= alpha;
        }
        // P in the value dtype (p.to(v.dtype), :332), packed as the second MFMA operand: 8 keys per lane group and K step
        v4i pf[2];
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                if constexpr (V_T == SDNQ_BF16) pf[c][w] = __builtin_bit_cast(int, __builtin_convertvector(t[4 * c + w], v2bf));
                else pf[c][w] = __builtin_bit_cast(int, __builtin_convertvector(t[4 * c + w], v2h));
            }
        // key step outermost: consecutive MFMAs write different accumulators (back-to-back MFMAs on one accumulator wait for each
        // other's latency)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int dd = 0; dd < KK; ++dd) {
                if constexpr (V_T == SDNQ_BF16)
                    o[dd] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(v8bf, b.v[dd][c]), __builtin_bit_cast(v8bf, pf[c]), o[dd], 0, 0, 0);
                else
                    o[dd] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(v8h, b.v[dd][c]), __builtin_bit_cast(v8h, pf[c]), o[dd], 0, 0, 0);
            }
    };

    // blocks [0, n_plain) need no mask; at most ONE more block does (the key tail, or the causal diagonal block key0 == q0)
    int nkb = (int)((p.kn + 31) / 32), n_plain = (int)(p.kn / 32);
    if (CAUSAL) {
        const int lim = (int)(q0 / 32) + 1;  // blocks past the last query of this wave are fully masked (triton_atten.py:255)
        nkb = nkb < lim ? nkb : lim;
        n_plain = n_plain < lim - 1 ? n_plain : lim - 1;
    }
    // this wave's share: plain blocks [lo, hi) and slow-path blocks [mlo, mhi) (the one tail / diagonal block goes to the last
    // share; with an attention mask EVERY block takes the slow path and the halves split them)
    int mlo = n_plain, mhi = nkb;
    if (HAS_MASK) { n_plain = 0; mlo = 0; }
    int lo = 0, hi = n_plain;
    if (p.split > 1) {
        const int per = (n_plain + p.split - 1) / p.split;  // plain blocks per part
        lo = part * per < n_plain ? part * per : n_plain;
        hi = lo + per < n_plain ? lo + per : n_plain;
        if (HAS_MASK) {
            const int mper = (nkb + p.split - 1) / p.split;
            mlo = part * mper < nkb ? part * mper : nkb;
            mhi = mlo + mper < nkb ? mlo + mper : nkb;
        } else if (part != p.split - 1) {
            mhi = mlo;
        }
    }
    // ---- shared K / V (split == 1, no masks): the workgroup's 4 waves work on 4 query tiles of ONE head, so every key block is
    // fetched from L2 once per workgroup -- LDS-DMA into a double-buffered stage of two blocks -- and read from LDS by all four.
    // Per CU the loop otherwise pulls ~25 B/clk from L2, the same per-CU fill plateau the GEMM kernels hit (a timing-only build
    // whose loads all hit one block in L1 was 22 % faster).
    constexpr int STG_K = 2 * KK * 1024, STG_V = 4 * KK * 1024, STG_BYTES = STG_K + STG_V + 256;
    constexpr int COMB_BYTES = 3 * (KK * 16 + 2) * 64 * 4;
    // inline K / V (<= 128 keys = 4 blocks): [K tiles 4 KK KiB][V tiles 8 KK KiB][128 key scales][4 x 128 partial channel sums][128 means]
    constexpr int IN_V = 4 * KK * 1024, IN_S = 12 * KK * 1024, IN_XW = IN_S + 512, IN_MEAN = IN_XW + 2048, IN_BYTES = IN_MEAN + 512;
    constexpr int STREAM_BYTES = 2 * STG_BYTES > COMB_BYTES ? 2 * STG_BYTES : COMB_BYTES;
    __shared__ __attribute__((aligned(16))) uint8_t smem[STREAM_BYTES > IN_BYTES ? STREAM_BYTES : IN_BYTES];
    if (inline_kv) {
        // from the registers loaded at the top: V scattered into its operand tiles, the channel means, K quantized
        const int c8 = ic8, r0 = ir0;
        const bool col = c8 < p.d_src;
        // V operand: key r, channel ch -> tile (r / 32, ch / 32, (r % 32) / 16), lane (r % 16) / 8 * 32 + ch % 32, element r % 8
        // (one 1-KiB tile per (32-key block, 32-channel block, 16-key step), see attn_vt_block); padding keys / channels are zeros
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int r = r0 + i * RPP;
            if (r < (int)p.knp) {
                const uint16_t* h = (const uint16_t*)&vraw[i];
                uint16_t* dst = (uint16_t*)(smem + IN_V) + (r / 32) * (KK * 2 * 512) + ((r & 31) >> 4) * 512 + (((r & 15) >> 3) * 32) * 8 + (r & 7);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int ch = c8 + e;
                    dst[(ch >> 5) * 1024 + (ch & 31) * 8] = h[e];
                }
            }
        }
        float kv[NR][8];
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            Vec16<V_T>::unpack(kraw[i], kv[i]);  // (zeros where nothing was loaded)
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += kv[i][e];
        }
        if (p.smooth) {
            attn_means_reduce(acc, p.kn, D, (float*)(smem + IN_XW), (float*)(smem + IN_MEAN));
            const float* mean = (const float*)(smem + IN_MEAN) + c8;
#pragma unroll
            for (int i = 0; i < NR; ++i) {
                if (col && r0 + i * RPP < p.kn) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) kv[i][e] -= mean[e];  // k.to(float32).sub_(mean), triton_atten.py:459-463
                }
            }
        }
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int r = r0 + i * RPP;
            u32 o2[2];
            const float scale = attn_quant8(kv[i], LPR, o2);
            if (r < (int)p.knp) {
                *(uint2*)(smem + attn_kfrag_offset(r, c8, D)) = make_uint2(o2[0], o2[1]);
                if (c8 == 0) ((float*)(smem + IN_S))[r] = scale;
            }
        }
        __syncthreads();
        if (!active) return;
        auto lds_block = [&](int kb, v4i (&kf)[KK], Blk& b) {
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) kf[kk] = *(const v4i*)(smem + (kb * KK + kk) * 1024 + lane * 16);
#pragma unroll
            for (int dd = 0; dd < KK; ++dd)
#pragma unroll
                for (int c = 0; c < 2; ++c) b.v[dd][c] = *(const v4i*)(smem + IN_V + ((kb * KK + dd) * 2 + c) * 1024 + lane * 16);
            const float* ksl = (const float*)(smem + IN_S) + kb * 32 + 8 * g;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                b.ks[2 * c] = *(const v4f*)(ksl + 16 * c);
                b.ks[2 * c + 1] = *(const v4f*)(ksl + 16 * c + 4);
            }
        };
#pragma nounroll
        for (int kb = lo; kb < hi; ++kb) {
            v4i kf[KK];
            Blk b;
            lds_block(kb, kf, b);
            const v16i sc = qk_mfma(kf);
            softmax_pv(sc, b, (int64_t)kb * 32, std::false_type{});
        }
#pragma nounroll
        for (int kb = mlo; kb < mhi; ++kb) {
            v4i kf[KK];
            Blk b;
            lds_block(kb, kf, b);
            const v16i sc = qk_mfma(kf);
            softmax_pv(sc, b, (int64_t)kb * 32, std::true_type{});
        }
        lo = hi; mlo = mhi;  // nothing left for the streaming loops below
    }
    // (head_dim 128 only: at 64 sharing never paid -- tools/bench_attention.py, round 2 -- and its two-blocks-at-once register sets kept the
    //  head_dim 64 kernel at 160 VGPRs)
    if (D == 128 && !CAUSAL && !HAS_MASK && p.shared_kv) {
        const int n_st = n_plain / 2;  // full two-block stages; the same for every wave (no causal limit)
        auto dma_stage = [&](int st, int buf) {
            uint8_t* dst = smem + buf * STG_BYTES;
#pragma unroll
            for (int i = 0; i < 6 * KK / 4; ++i) {  // 1-KiB tiles w, w + 4, ...: first the 2 KK tiles of K, then the 4 KK of V
                const int tile = wave + 4 * i;  // wave-uniform
                if (tile < 2 * KK) SDNQ_DMA16(rsK, dst + tile * 1024, lofs, st * STG_K + tile * 1024);
                else SDNQ_DMA16(rsV, dst + tile * 1024, lofs, st * STG_V + (tile - 2 * KK) * 1024);
            }
            if (wave == 0) SDNQ_DMA4(rsS, dst + STG_K + STG_V, lane * 4, st * 256);  // 64 k_scales
        };
        auto lds_block = [&](int buf, int blk, v4i (&kf)[KK], Blk& b) {
            const uint8_t* base = smem + buf * STG_BYTES;
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) kf[kk] = *(const v4i*)(base + (blk * KK + kk) * 1024 + lane * 16);
#pragma unroll
            for (int dd = 0; dd < KK; ++dd)
#pragma unroll
                for (int c = 0; c < 2; ++c) b.v[dd][c] = *(const v4i*)(base + STG_K + ((blk * KK + dd) * 2 + c) * 1024 + lane * 16);
            const float* ksl = (const float*)(base + STG_K + STG_V) + blk * 32 + 8 * g;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                b.ks[2 * c] = *(const v4f*)(ksl + 16 * c);
                b.ks[2 * c + 1] = *(const v4f*)(ksl + 16 * c + 4);
            }
        };
        if (n_st > 0) {
            dma_stage(0, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
#pragma nounroll
            for (int st = 0; st < n_st; ++st) {
                if (st + 1 < n_st) dma_stage(st + 1, (st + 1) & 1);
                if (active) {
                    v4i kf0[KK], kf1[KK];
                    Blk b0, b1;
                    lds_block(st & 1, 0, kf0, b0);
                    lds_block(st & 1, 1, kf1, b1);
                    const v16i s0 = qk_mfma(kf0), s1 = qk_mfma(kf1);
                    softmax_pv(s0, b0, (int64_t)st * 64, std::false_type{});
                    softmax_pv(s1, b1, (int64_t)st * 64 + 32, std::false_type{});
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's share of stage st + 1 has landed
                __syncthreads();                                   // ... everybody's has, and everybody is done reading stage st
            }
            lo = 2 * n_st;  // an odd last plain block and the key tail go through the per-wave code below
        }
    }
    if (!active) { hi = lo; mhi = mlo; }
    if (p.shared_kv && !active) return;
    if (D == 64) {
        // head_dim 64: one block per trip, no software pipeline.  Round 4, judged on the step and per shape: the two-block pipeline below
        // (162 VGPRs against 132) is no faster at 10 x 4096^2 (70.9 us both) and slower on the short calls (20 x 1024^2: 15.2 -> 13.9 us,
        // the 77-key shapes -0.3..0.6 us; sdxl_attn_int8 2.904 -> 2.872 ms): its prologue / epilogue blocks are a larger share of 8-block
        // key parts than of the 144-block rows of FLUX, which keep it
#pragma nounroll
        for (int kb = lo; kb < hi; ++kb) {
            v4i kf[KK];
            Blk b;
            load_k(kb, kf);
            load_vs(kb, b);
            const v16i sc = qk_mfma(kf);
            softmax_pv(sc, b, (int64_t)kb * 32, std::false_type{});
        }
        lo = hi;
    }
    if (lo < hi) {
        const int last = hi - 1;
        v4i kfA[KK], kfB[KK];
        Blk bA, bB;
        v16i sA, sB;
        load_k(lo, kfA);
        load_vs(lo, bA);
        load_k(lo + 1 < last ? lo + 1 : last, kfB);
        sA = qk_mfma(kfA);
        // two blocks per trip so that the A / B register sets swap roles without moves
#pragma nounroll
        for (int kb = lo; kb < hi; kb += 2) {
            // block kb: scores in sA, V / scales in bA; K(kb+1) in kfB
            sB = qk_mfma(kfB);
            load_vs(kb + 1 < last ? kb + 1 : last, bB);
            load_k(kb + 2 < last ? kb + 2 : last, kfA);
            softmax_pv(sA, bA, (int64_t)kb * 32, std::false_type{});
            if (kb + 1 >= hi) break;
            // block kb+1: scores in sB, V / scales in bB; K(kb+2) in kfA
            sA = qk_mfma(kfA);
            load_vs(kb + 2 < last ? kb + 2 : last, bA);
            load_k(kb + 3 < last ? kb + 3 : last, kfB);
            softmax_pv(sB, bB, (int64_t)(kb + 1) * 32, std::false_type{});
        }
    }
#pragma nounroll
    for (int kb = mlo; kb < mhi; ++kb) {
        v4i kf[KK];
        Blk b;
        load_k(kb, kf);
        load_vs(kb, b);
        const v16i s = qk_mfma(kf);
        softmax_pv(s, b, (int64_t)kb * 32, std::true_type{});
    }
    float l_i = l2[0] + l2[1];
    if (p.split > 1) {
        // merge the key parts of a query tile: o = sum_i o_i * 2^(m_i - m), same for the row sums
        float (*comb)[KK * 16 + 2][64] = (float (*)[KK * 16 + 2][64])smem;  // split 2: one slot per tile (2 tiles); split 4: three slots of the one tile
        if (part != 0 && active) {
            float (*cb)[64] = comb[p.split == 2 ? wtile : part - 1];
#pragma unroll
            for (int dd = 0; dd < KK; ++dd)
#pragma unroll
                for (int r = 0; r < 16; ++r) cb[dd * 16 + r][lane] = o[dd][r];
            cb[KK * 16][lane] = m_i;
            cb[KK * 16 + 1][lane] = l_i;
        }
        __syncthreads();
        if (part != 0 || !active) return;
#pragma nounroll
        for (int i = 1; i < p.split; ++i) {
            float (*cb)[64] = comb[p.split == 2 ? wtile : i - 1];
            const float m1 = cb[KK * 16][lane], l1 = cb[KK * 16 + 1][lane];
            float m = fmaxf(m_i, m1);
            if (m == -__builtin_inff()) m = 0.0f;  // no visible key in either part (attention mask): both weights become 0
            const float a0 = __builtin_amdgcn_exp2f(m_i - m), a1 = __builtin_amdgcn_exp2f(m1 - m);
            l_i = l_i * a0 + l1 * a1;
#pragma unroll
            for (int dd = 0; dd < KK; ++dd)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[dd][r] = o[dd][r] * a0 + cb[dd * 16 + r][lane] * a1;
            m_i = fmaxf(m_i, m1);
        }
    }
    if (qi >= p.qn) return;
    l_i += __shfl_xor(l_i, 32);
    const float inv = l_i > 0.0f ? 1.0f / l_i : 0.0f;  // acc *= fdiv(1.0, l_i), :336; a row with no visible key is 0 (l stays 1, acc 0 there)
    char* orow = (char*)p.out + p.ost.at(head_lin, qi) * FT<OUT_T>::bytes;
#pragma unroll
    for (int dd = 0; dd < KK; ++dd)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int dcol = 32 * dd + 8 * t + 4 * g;  // registers 4t..4t+3 are 4 consecutive channels
            if (dcol >= p.d_out) continue;
            float f[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) f[u] = o[dd][4 * t + u] * inv;
            if constexpr (OUT_T == SDNQ_F32) {
                *(float4*)(orow + dcol * 4) = make_float4(f[0], f[1], f[2], f[3]);
            } else {
                const u32 lo = FT<OUT_T>::bits(f[0]) | ((u32)FT<OUT_T>::bits(f[1]) << 16);
                const u32 hi = FT<OUT_T>::bits(f[2]) | ((u32)FT<OUT_T>::bits(f[3]) << 16);
                *(uint2*)(orow + dcol * 2) = make_uint2(lo, hi);
            }
        }
}

template <int V_T, int OUT_T, int D>
int launch_fwd(const AttnParams& p, int causal, int64_t blocks, hipStream_t s) {
    const dim3 grid((unsigned)blocks), block(256);
    if (p.mask) {
        if (causal) hipLaunchKernelGGL((attn_fwd_kernel<V_T, OUT_T, D, true, true>), grid, block, 0, s, p);
        else hipLaunchKernelGGL((attn_fwd_kernel<V_T, OUT_T, D, false, true>), grid, block, 0, s, p);
    } else {
        if (causal) hipLaunchKernelGGL((attn_fwd_kernel<V_T, OUT_T, D, true, false>), grid, block, 0, s, p);
        else hipLaunchKernelGGL((attn_fwd_kernel<V_T, OUT_T, D, false, false>), grid, block, 0, s, p);
    }
    SDNQ_CHECK_LAUNCH();
    return SDNQ_OK;
}

template <int V_T, int OUT_T>
int launch_fwd_d(const AttnParams& p, int d, int causal, int64_t blocks, hipStream_t s) {
    if (d == 64) return launch_fwd<V_T, OUT_T, 64>(p, causal, blocks, s);
    if (d == 128) return launch_fwd<V_T, OUT_T, 128>(p, causal, blocks, s);
    return SDNQ_ERR_UNSUPPORTED;
}

bool shape_ok(int64_t batch, int64_t qh, int64_t kh, int64_t qn, int64_t kn, int64_t d) {
    return batch > 0 && qh > 0 && kh > 0 && qn > 0 && kn > 0 && d > 0 && qh % kh == 0;
}

}  // namespace

extern "C" int sdnq_hip_attn_prepare(const void* q, const void* k, const void* v, int dtype, int64_t batch, int64_t q_heads,
                                     int64_t kv_heads, int64_t q_len, int64_t kv_len, int64_t head_dim, int smooth_k,
                                     int hadamard_group, const int64_t* q_strides, const int64_t* k_strides,
                                     const int64_t* v_strides, void* qq, float* qs, void* kq, float* ks, void* vt, float* kmean,
                                     sdnq_stream_t stream) {
    const bool with_q = qq != nullptr || qs != nullptr;  // both null: Q is quantized by the forward kernel (sdnq_hip_attn_fwd_q16)
    if (!k || !v || !kq || !ks || !vt || (smooth_k && !kmean) || (with_q && (!q || !qq || !qs))) return SDNQ_ERR_NULL;
    if (!shape_ok(batch, q_heads, kv_heads, q_len, kv_len, head_dim)) return SDNQ_ERR_SHAPE;
    if (head_dim < 8 || head_dim > 128 || head_dim % 8) return SDNQ_ERR_UNSUPPORTED;
    if (dtype != SDNQ_BF16 && dtype != SDNQ_F16) return SDNQ_ERR_UNSUPPORTED;  // PV runs in the value dtype on the matrix cores
    const int64_t head_dim_src = head_dim;
    head_dim = head_dim <= 64 ? 64 : 128;  // the kernels' head dim; the extra channels are zeros
    int log2g = 0;
    if (hadamard_group != 0) {
        if (hadamard_group < 4 || hadamard_group > head_dim || (hadamard_group & (hadamard_group - 1)) || head_dim % hadamard_group) return SDNQ_ERR_SHAPE;
        while ((1 << log2g) < hadamard_group) ++log2g;
    }
    if (((with_q ? (uintptr_t)q | (uintptr_t)qq : 0) | (uintptr_t)k | (uintptr_t)v | (uintptr_t)kq | (uintptr_t)vt) % 16) return SDNQ_ERR_ALIGN;
    hipStream_t s = (hipStream_t)stream;
    const int d = (int)head_dim, lpr = d / 8;
    const int64_t kheads = batch * kv_heads;
    PrepParams p{};
    p.q = q; p.k = k; p.v = v; p.qq = (int8_t*)qq; p.kq = (int8_t*)kq; p.qs = qs; p.ks = ks; p.vt = (uint16_t*)vt;
    const bool inline_mean = smooth_k && kv_len <= 256;
    p.smooth_inline = inline_mean;
    p.kpart = (smooth_k && !inline_mean) ? kmean : nullptr;
    auto strides_of = [&](const int64_t* st, int64_t heads, int64_t len, Strides& out) {
        out.heads = heads;
        if (st) { out.b = st[0]; out.h = st[1]; out.n = st[2]; } else { out.b = heads * len * head_dim_src; out.h = len * head_dim_src; out.n = head_dim_src; }
        return out.b % 8 == 0 && out.h % 8 == 0 && out.n % 8 == 0;  // 16-byte rows
    };
    if ((with_q && !strides_of(q_strides, q_heads, q_len, p.qst)) || !strides_of(k_strides, kv_heads, kv_len, p.kst) ||
        !strides_of(v_strides, kv_heads, kv_len, p.vst))
        return SDNQ_ERR_ALIGN;
    p.qheads = batch * q_heads; p.kheads = kheads; p.qn = q_len; p.kn = kv_len; p.knp = (kv_len + 31) / 32 * 32; p.d = d; p.d_src = (int)head_dim_src;
    p.log2g = log2g;
    p.nqb = with_q ? (p.qheads * q_len * lpr + 255) / 256 : 0;
    p.nkb = kheads * p.knp * lpr / 256;  // exact: knp * lpr is a multiple of 256
    p.nvb = kheads * (p.knp / 32);
    p.nmb = 0;
    p.kpart_out = kmean;
    auto launch = [&](const PrepParams& pp) {
        const int64_t blocks = pp.nqb + pp.nkb + pp.nvb + pp.nmb;
        if (blocks == 0) return;
        if (dtype == SDNQ_BF16) hipLaunchKernelGGL((attn_prepare_kernel<SDNQ_BF16>), dim3((unsigned)blocks), dim3(256), 0, s, pp);
        else hipLaunchKernelGGL((attn_prepare_kernel<SDNQ_F16>), dim3((unsigned)blocks), dim3(256), 0, s, pp);
    };
    if (p.kpart) {
        PrepParams first = p, second = p;
        first.nkb = 0; first.nmb = kheads * KMEAN_SPLITS;  // everything that does not need the means, next to the channel sums
        second.nqb = 0; second.nvb = 0;                     // then the K rows
        launch(first);
        launch(second);
    } else {
        launch(p);
    }
    SDNQ_CHECK_LAUNCH();
    return SDNQ_OK;
}

namespace {
struct RawKV { const void *k, *v; const int64_t *k_strides, *v_strides; int smooth; };  // K / V in the value dtype (at most 128 keys)

// q16 != nullptr: Q in the value dtype with element strides q_strides (null: contiguous [batch][heads][q_len][head_dim]); else qq / qs.
// raw != nullptr: K / V in the value dtype as well (kq / ks / vt unused), needs q16.
int attn_fwd_impl(const void* qq, const float* qs, const void* q16, const int64_t* q_strides, const RawKV* raw, const void* kq, const float* ks, const void* vt, int v_dtype,
                  float sm_scale, int is_causal, const void* mask, int mask_dtype, int64_t mask_stride_b,
                  int64_t mask_stride_h, int64_t mask_stride_q, void* out, int out_dtype,
                  const int64_t* out_strides, int64_t batch,
                  int64_t q_heads, int64_t kv_heads, int64_t q_len, int64_t kv_len, int64_t head_dim,
                  sdnq_stream_t stream) {
    if ((!q16 && (!qq || !qs)) || (!raw && (!kq || !ks || !vt)) || (raw && (!raw->k || !raw->v || !q16)) || !out) return SDNQ_ERR_NULL;
    if (mask && mask_dtype != -1 && mask_dtype != SDNQ_F32 && mask_dtype != SDNQ_BF16 && mask_dtype != SDNQ_F16) return SDNQ_ERR_DTYPE;
    if (!shape_ok(batch, q_heads, kv_heads, q_len, kv_len, head_dim)) return SDNQ_ERR_SHAPE;
    if (head_dim < 8 || head_dim > 128 || head_dim % 8) return SDNQ_ERR_UNSUPPORTED;
    const int64_t head_dim_src = head_dim;
    head_dim = head_dim <= 64 ? 64 : 128;  // as in sdnq_hip_attn_prepare: qq / kq / vt hold the padded head dim
    if (((uintptr_t)qq | (uintptr_t)kq | (uintptr_t)vt | (uintptr_t)out) % 8 || (uintptr_t)q16 % 16) return SDNQ_ERR_ALIGN;
    AttnParams p{};
    if (q16) {
        p.q_src = q16; p.qst.heads = q_heads; p.d_src = (int)head_dim_src;
        if (q_strides) { p.qst.b = q_strides[0]; p.qst.h = q_strides[1]; p.qst.n = q_strides[2]; }
        else { p.qst.b = q_heads * q_len * head_dim_src; p.qst.h = q_len * head_dim_src; p.qst.n = head_dim_src; }
        if (p.qst.b % 8 || p.qst.h % 8 || p.qst.n % 8) return SDNQ_ERR_ALIGN;  // 16-byte rows
    }
    if (raw) {
        if (kv_len > 128) return SDNQ_ERR_UNSUPPORTED;
        if (((uintptr_t)raw->k | (uintptr_t)raw->v) % 16) return SDNQ_ERR_ALIGN;
        auto set = [&](const int64_t* st, Strides& o) {
            o.heads = kv_heads;
            if (st) { o.b = st[0]; o.h = st[1]; o.n = st[2]; } else { o.b = kv_heads * kv_len * head_dim_src; o.h = kv_len * head_dim_src; o.n = head_dim_src; }
            return o.b % 8 == 0 && o.h % 8 == 0 && o.n % 8 == 0;
        };
        if (!set(raw->k_strides, p.kst) || !set(raw->v_strides, p.vst)) return SDNQ_ERR_ALIGN;
        p.k_src = raw->k; p.v_src = raw->v; p.smooth = raw->smooth; p.d_src = (int)head_dim_src;
    }
    p.qq = (const int8_t*)qq; p.qs = qs; p.kq = (const int8_t*)kq; p.ks = ks; p.vt = (const uint16_t*)vt; p.out = out;
    p.qh = q_heads; p.kh = kv_heads; p.qn = q_len; p.kn = kv_len; p.knp = (kv_len + 31) / 32 * 32;
    // few query tiles for the 1024 SIMDs (e.g. SDXL at batch 1: 1280): split every tile's keys over two waves
    static const int force_split = [] { const char* e = getenv("SDNQ_HIP_ATTN_SPLIT"); return e ? atoi(e) : 0; }();  // tuning aid
    const int64_t tiles = batch * q_heads * ((q_len + 31) / 32);
    // measured (tools/bench_attention.py): splitting pays at head_dim 64 (SDXL 10 x 4096^2: 90 -> 78 us); at head_dim 128 the
    // unsplit kernel with K / V shared through LDS is faster (FLUX 24 x 4608^2: 333 -> 309 us), at 64 sharing does not pay
    const bool want_shared = head_dim == 128 && kv_len >= 2048 && !is_causal && !mask;
    // round 4, judged on the step (sdxl_attn_int8, one box: round-3 rule -- 2 parts for 1024 < tiles < 4096 and >= 2048 keys -- 3.35-3.42 ms;
    // target 2048: 3.20; 4096: 3.155; 8192: 3.164; 4 parts forced everywhere: 3.235): the smallest number of parts that puts at least
    // `split_target` waves on the chip, every part keeping >= 4 key blocks (the 77-key cross-attention stays whole)
    static const int split_target = [] { const char* e = getenv("SDNQ_HIP_ATTN_SPLIT_TARGET"); return e ? atoi(e) : 4096; }();  // tuning aid
    int auto_split = 1;
    if (!want_shared) {
        const int64_t kblocks = (kv_len + 31) / 32;
        while (auto_split < 4 && tiles * auto_split < split_target && kblocks >= 4 * (auto_split * 2)) auto_split *= 2;
    }
    p.split = raw ? 1 : (force_split ? force_split : auto_split);
    if (p.split != 1 && p.split != 2 && p.split != 4) return SDNQ_ERR_SHAPE;
    static const int force_shared = [] { const char* e = getenv("SDNQ_HIP_ATTN_SHARED"); return e ? atoi(e) : -1; }();  // tuning aid
    p.shared_kv = (!raw && p.split == 1 && !is_causal && !mask && kv_len >= 64) ? (force_shared < 0 ? (want_shared ? 1 : 0) : force_shared) : 0;
    const int tiles_per_wg = 4 / p.split;
    p.qblocks = (int)((q_len + 32 * tiles_per_wg - 1) / (32 * tiles_per_wg));
    p.log2_sm_scale = sm_scale * 1.4426950408889634f;  // triton_atten.py:203
    p.ost.heads = q_heads;
    if (out_strides) { p.ost.b = out_strides[0]; p.ost.h = out_strides[1]; p.ost.n = out_strides[2]; }
    else { p.ost.b = q_heads * q_len * head_dim_src; p.ost.h = q_len * head_dim_src; p.ost.n = head_dim_src; }
    p.d_out = (int)head_dim_src;
    if (p.ost.b % 4 || p.ost.h % 4 || p.ost.n % 4) return SDNQ_ERR_ALIGN;
    p.mask = mask; p.mask_dtype = mask_dtype; p.ms_z = mask_stride_b; p.ms_h = mask_stride_h; p.ms_q = mask_stride_q;
    const int64_t blocks = batch * q_heads * p.qblocks;
    hipStream_t s = (hipStream_t)stream;
    const int d = (int)head_dim;
#define ATTN_CASE(VT, OT) if (v_dtype == VT && out_dtype == OT) return launch_fwd_d<VT, OT>(p, d, is_causal, blocks, s)
    ATTN_CASE(SDNQ_BF16, SDNQ_BF16);
    ATTN_CASE(SDNQ_BF16, SDNQ_F32);
    ATTN_CASE(SDNQ_F16, SDNQ_F16);
    ATTN_CASE(SDNQ_F16, SDNQ_F32);
#undef ATTN_CASE
    return SDNQ_ERR_DTYPE;
}
}  // namespace

extern "C" int sdnq_hip_attn_fwd(const void* qq, const float* qs, const void* kq, const float* ks, const void* vt, int v_dtype,
                                 float sm_scale, int is_causal, const void* mask, int mask_dtype, int64_t mask_stride_b,
                                 int64_t mask_stride_h, int64_t mask_stride_q, void* out, int out_dtype,
                                 const int64_t* out_strides, int64_t batch,
                                 int64_t q_heads, int64_t kv_heads, int64_t q_len, int64_t kv_len, int64_t head_dim,
                                 sdnq_stream_t stream) {
    if (!qq || !qs) return SDNQ_ERR_NULL;
    return attn_fwd_impl(qq, qs, nullptr, nullptr, nullptr, kq, ks, vt, v_dtype, sm_scale, is_causal, mask, mask_dtype, mask_stride_b, mask_stride_h, mask_stride_q, out,
                         out_dtype, out_strides, batch, q_heads, kv_heads, q_len, kv_len, head_dim, stream);
}

extern "C" int sdnq_hip_attn_fwd_q16(const void* q, const int64_t* q_strides, const void* kq, const float* ks, const void* vt, int v_dtype,
                                     float sm_scale, int is_causal, const void* mask, int mask_dtype, int64_t mask_stride_b,
                                     int64_t mask_stride_h, int64_t mask_stride_q, void* out, int out_dtype,
                                     const int64_t* out_strides, int64_t batch,
                                     int64_t q_heads, int64_t kv_heads, int64_t q_len, int64_t kv_len, int64_t head_dim,
                                     sdnq_stream_t stream) {
    if (!q) return SDNQ_ERR_NULL;
    return attn_fwd_impl(nullptr, nullptr, q, q_strides, nullptr, kq, ks, vt, v_dtype, sm_scale, is_causal, mask, mask_dtype, mask_stride_b, mask_stride_h, mask_stride_q, out,
                         out_dtype, out_strides, batch, q_heads, kv_heads, q_len, kv_len, head_dim, stream);
}

// ---- the whole attention as one entry point (sdnq_triton_atten, triton_atten.py:540-618) -------------------------------------------
namespace {
constexpr int64_t ATTN_SINGLE_MAX_KEYS = 128;
inline int64_t up256(int64_t n) { return (n + 255) / 256 * 256; }
struct AttnWs { int64_t qq, qs, kq, ks, vt, kmean, total; };
// workspace slices (bytes, 256-byte aligned): quantized Q only when a Hadamard rotation keeps it in the prepare pass
AttnWs attn_ws(int64_t batch, int64_t q_heads, int64_t kv_heads, int64_t q_len, int64_t kv_len, int64_t head_dim, int hadamard_group) {
    AttnWs w{};
    const int64_t d = head_dim <= 64 ? 64 : 128, knp = (kv_len + 31) / 32 * 32;
    const bool with_q = hadamard_group != 0;
    if (!with_q && kv_len <= ATTN_SINGLE_MAX_KEYS) return w;  // single launch: nothing leaves the kernel
    int64_t off = 0;
    w.qq = off; off += with_q ? up256(batch * q_heads * q_len * d) : 0;
    w.qs = off; off += with_q ? up256(batch * q_heads * q_len * 4) : 0;
    w.kq = off; off += up256(batch * kv_heads * knp * d);
    w.ks = off; off += up256(batch * kv_heads * knp * 4);
    w.vt = off; off += up256(batch * kv_heads * knp * d * 2);
    w.kmean = off; off += up256(batch * kv_heads * KMEAN_SPLITS * d * 4);
    w.total = off;
    return w;
}
}  // namespace

extern "C" int64_t sdnq_hip_attn_workspace_bytes(int64_t batch, int64_t q_heads, int64_t kv_heads, int64_t q_len, int64_t kv_len, int64_t head_dim,
                                                 int hadamard_group) {
    if (!shape_ok(batch, q_heads, kv_heads, q_len, kv_len, head_dim) || head_dim > 128) return SDNQ_ERR_SHAPE;
    return attn_ws(batch, q_heads, kv_heads, q_len, kv_len, head_dim, hadamard_group).total;
}

extern "C" int sdnq_hip_attn(const void* q, const void* k, const void* v, int dtype, int64_t batch, int64_t q_heads, int64_t kv_heads,
                             int64_t q_len, int64_t kv_len, int64_t head_dim, const int64_t* q_strides, const int64_t* k_strides,
                             const int64_t* v_strides, int smooth_k, int hadamard_group, float sm_scale, int is_causal, const void* mask,
                             int mask_dtype, int64_t mask_stride_b, int64_t mask_stride_h, int64_t mask_stride_q, void* out, int out_dtype,
                             const int64_t* out_strides, void* workspace, int64_t workspace_bytes, sdnq_stream_t stream) {
    if (!q || !k || !v || !out) return SDNQ_ERR_NULL;
    if (!shape_ok(batch, q_heads, kv_heads, q_len, kv_len, head_dim)) return SDNQ_ERR_SHAPE;
    if (head_dim < 8 || head_dim > 128 || head_dim % 8) return SDNQ_ERR_UNSUPPORTED;
    if (dtype != SDNQ_BF16 && dtype != SDNQ_F16) return SDNQ_ERR_UNSUPPORTED;
    const AttnWs w = attn_ws(batch, q_heads, kv_heads, q_len, kv_len, head_dim, hadamard_group);
    if (w.total == 0) {  // <= 128 keys, no rotation: Q, K and V are quantized / laid out inside the one forward launch
        const RawKV raw{k, v, k_strides, v_strides, smooth_k};
        return attn_fwd_impl(nullptr, nullptr, q, q_strides, &raw, nullptr, nullptr, nullptr, dtype, sm_scale, is_causal, mask, mask_dtype, mask_stride_b,
                             mask_stride_h, mask_stride_q, out, out_dtype, out_strides, batch, q_heads, kv_heads, q_len, kv_len, head_dim, stream);
    }
    if (!workspace) return SDNQ_ERR_NULL;
    if (workspace_bytes < w.total) return SDNQ_ERR_SHAPE;
    if ((uintptr_t)workspace % 256) return SDNQ_ERR_ALIGN;
    char* ws = (char*)workspace;
    const bool with_q = hadamard_group != 0;
    const int rc = sdnq_hip_attn_prepare(q, k, v, dtype, batch, q_heads, kv_heads, q_len, kv_len, head_dim, smooth_k, hadamard_group, q_strides, k_strides,
                                         v_strides, with_q ? ws + w.qq : nullptr, with_q ? (float*)(ws + w.qs) : nullptr, ws + w.kq, (float*)(ws + w.ks),
                                         ws + w.vt, (float*)(ws + w.kmean), stream);
    if (rc != SDNQ_OK) return rc;
    return attn_fwd_impl(with_q ? ws + w.qq : nullptr, with_q ? (const float*)(ws + w.qs) : nullptr, with_q ? nullptr : q, q_strides, nullptr, ws + w.kq,
                         (const float*)(ws + w.ks), ws + w.vt, dtype, sm_scale, is_causal, mask, mask_dtype, mask_stride_b, mask_stride_h, mask_stride_q, out,
                         out_dtype, out_strides, batch, q_heads, kv_heads, q_len, kv_len, head_dim, stream);
}

// =====================================================================================================================================
// Round 6: the other matmul formats of the reference's attention (triton_atten.py:443-487 quantize_attn, :273-284 Q.K^T, :303-323 P.V):
//   Q.K^T on fp8 (e4m3) codes, per-token scale amax / 448 (quantize_fp_mm, quant_utils.py:290-299);
//   P.V with V quantized per token (int8 / fp8 / float16 codes) and P quantized per (query, 32-key block):
//       p *= v_scale;  p_scale = max_k(p) / qmax  (1 where <= 2e-38);  int8: floor(fma(p, 1 / p_scale, 0.5)),  fp8 / f16: (p * (1 / p_scale)).to(fmt)
//       acc = fma(dot(p_q, v_q), p_scale, acc)
//   The key block of the P quantization is the reference's autotuned BLOCK_SIZE_N; this kernel works in blocks of 32 keys (= the fixtures').
// A separate, plain kernel (one wave = 32 queries, fragments straight from memory, every block on the reference's exact -inf-safe update,
// causal / mask / output dtype as run-time switches): the default configuration above keeps its tuned kernel untouched.  Same fragment
// layouts: K codes as attn_kfrag_offset places them (int8 and e4m3 bytes alike -- the fp8 MFMA pairs the same byte of both operands, so
// any byte order common to K and Q is a valid K order of the dot product); 8-bit V codes in 1-KiB tiles per (32-key block, 32-channel
// block): lane (g, ql) holds the 16 bytes j = 0..15 <-> key 16 (j >> 3) + 8 g + (j & 7) of channel 32 dd + ql -- the order this lane's 16
// probabilities come out of the score MFMA in.
namespace {

enum { PVQ_NONE = 0, PVQ_I8 = 1, PVQ_FP8 = 2, PVQ_F16 = 3 };

struct VarPrepParams {
    const void *q, *k, *v;
    uint8_t *qq, *kq;
    float *qs, *ks, *vs;
    void* vt;
    const float* kmean;  // [kheads][d] channel means (smooth_k) or nullptr
    Strides qst, kst, vst;
    int64_t qheads, kheads, qn, kn, knp, nqb, nkb, nvb;
    int d, d_src, log2g, qk_fp8, pvq;
};

template <int T_ID>
__global__ __launch_bounds__(256) void attn_var_kmean_kernel(const void* k, const Strides kst, float* kmean, int64_t kn, int d, int d_src) {
    __shared__ float xw[512];
    __shared__ float smean[128];
    attn_head_means<T_ID>(k, kst, blockIdx.x, kn, d, d_src, xw, smean);
    if ((int)threadIdx.x < d) kmean[(int64_t)blockIdx.x * d + threadIdx.x] = smean[threadIdx.x];
}

// one token row spread over lpr lanes (8 channels each) -> codes of this lane's 8 channels + the row's scale.  FMT 0: int8 (attn_quant8),
// 1: e4m3 (quant8<fp8>: x / scale, nan_to_num, clamp, round to nearest even), 2: float16 (same with +-65504)
template <int FMT>
__device__ __forceinline__ float attn_quant_row(const float (&v)[8], int lpr, u32 (&o)[4]) {
    if constexpr (FMT == 0) {
        u32 o2[2];
        const float s = attn_quant8(v, lpr, o2);
        o[0] = o2[0]; o[1] = o2[1]; o[2] = o[3] = 0;
        return s;
    } else {
        float amax = 0.0f;
#pragma unroll
        for (int e = 0; e < 8; ++e) amax = fmaxf(amax, fabsf(v[e]));
        amax = fmaxf(amax, lane_xor(amax, 1));
        amax = fmaxf(amax, lane_xor(amax, 2));
        amax = fmaxf(amax, lane_xor(amax, 4));
        if (lpr == 16) amax = fmaxf(amax, lane_xor(amax, 8));
        const float scale = amax / (FMT == 1 ? 448.0f : 65504.0f);
        if constexpr (FMT == 1) {
            RowDiv rd;
            rd.set(scale);
            int isum = 0;
            const uint2 w = quant8<SDNQ_MM_FP8>(v, rd, isum);
            o[0] = w.x; o[1] = w.y; o[2] = o[3] = 0;
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float qv = v[e] / scale;
                if (qv != qv) qv = 0.0f;                                   // nan_to_num_
                qv = fminf(fmaxf(qv, -65504.0f), 65504.0f);               // clamp_ (+-inf fall to it)
                const u32 h = f32_to_f16_bits(qv);
                if (e & 1) o[e >> 1] |= h << 16; else o[e >> 1] = h;
            }
        }
        return scale;
    }
}

template <int T_ID>
__global__ __launch_bounds__(256) void attn_var_prepare_kernel(const VarPrepParams p) {
    __shared__ __attribute__((aligned(16))) uint16_t tile[32][128 + 2];
    const int64_t b = blockIdx.x;
    const int d = p.d, lpr = d / 8, lsh = d == 64 ? 3 : 4;
    if (p.pvq == PVQ_NONE && b >= p.nqb + p.nkb) {  // V in the value dtype: the operand layout of the default configuration
        int64_t vhead, vblk;
        divmod(b - p.nqb - p.nkb, p.knp >> 5, vhead, vblk);
        attn_vt_block((const uint16_t*)p.v, p.vst, (uint16_t*)p.vt, p.kn, p.knp, d, vhead, vblk, tile, p.d_src);
        return;
    }
    // ---- token rows: section 0 = Q, 1 = K (minus the channel means, MFMA-fragment order), 2 = V (quantized P.V)
    int sec = 0;
    int64_t blk = b;
    if (b >= p.nqb + p.nkb) { sec = 2; blk = b - p.nqb - p.nkb; }
    else if (b >= p.nqb) { sec = 1; blk = b - p.nqb; }
    const void* x = sec == 0 ? p.q : (sec == 1 ? p.k : p.v);
    const Strides xst = sec == 0 ? p.qst : (sec == 1 ? p.kst : p.vst);
    const int64_t heads = sec == 0 ? p.qheads : p.kheads, n_src = sec == 0 ? p.qn : p.kn, n_dst = sec == 0 ? p.qn : p.knp;
    const int64_t t = blk * 256 + threadIdx.x;
    const int64_t row = t >> lsh;
    const int c8 = (int)(t & (lpr - 1)) * 8;
    const bool live = row < heads * n_dst;
    int64_t head = 0, n = 0;
    if (live) divmod(row, n_dst, head, n);
    const bool real = live && n < n_src;
    float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (real && c8 < p.d_src) Vec16<T_ID>::unpack(*(const uint4*)((const uint16_t*)x + xst.at(head, n) + c8), v);
    const bool smooth = sec == 1 && p.kmean != nullptr;
    if (smooth && real && c8 < p.d_src) {
        const float* mean = p.kmean + head * d + c8;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] -= mean[e];  // k.to(float32).sub_(mean), triton_atten.py:459-463
    }
    if (p.log2g != 0) {  // apply_hadamard(q) / rotate_hadamard(k.to(hadamard.dtype)) / rotate_hadamard(v.to(hadamard.dtype)), :464-467, :479-480
        if (smooth) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = FT<T_ID>::round(v[e]);
        }
        wave_hadamard(v, p.log2g, hadamard_scale(p.log2g, T_ID));
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = FT<T_ID>::round(v[e]);
    }
    const int fmt = sec == 2 ? (p.pvq == PVQ_I8 ? 0 : (p.pvq == PVQ_FP8 ? 1 : 2)) : (p.qk_fp8 ? 1 : 0);  // workgroup-uniform
    u32 o[4];
    float scale;
    if (fmt == 0) scale = attn_quant_row<0>(v, lpr, o);
    else if (fmt == 1) scale = attn_quant_row<1>(v, lpr, o);
    else scale = attn_quant_row<2>(v, lpr, o);
    if (!live) return;
    if (sec == 0) {
        *(uint2*)(p.qq + row * d + c8) = make_uint2(o[0], o[1]);
        if (c8 == 0) p.qs[row] = scale;
    } else if (sec == 1) {
        *(uint2*)(p.kq + head * n_dst * d + attn_kfrag_offset(n, c8, d)) = make_uint2(o[0], o[1]);
        if (c8 == 0) p.ks[row] = scale;
    } else {
        const int nl = (int)(n & 31), vg = (nl >> 3) & 1;
        const int64_t kb = n >> 5;
        if (p.pvq == PVQ_F16) {  // 16-bit operand tiles: (kb, dd, c = nl >> 4), lane (g, ql), element nl & 7
            uint16_t* base = (uint16_t*)p.vt + (head * (n_dst >> 5) + kb) * (int64_t)(d / 32 * 2 * 512) + (nl >> 4) * 512 + (vg * 32) * 8 + (nl & 7);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int ch = c8 + e;
                base[(ch >> 5) * 1024 + (ch & 31) * 8] = (uint16_t)(o[e >> 1] >> (16 * (e & 1)));
            }
        } else {  // 8-bit operand tiles: (kb, dd), lane (g, ql), byte j = 8 (nl >> 4) + (nl & 7)
            uint8_t* base = (uint8_t*)p.vt + (head * (n_dst >> 5) + kb) * (int64_t)(d / 32 * 1024) + (vg * 32) * 16 + ((nl >> 4) << 3) + (nl & 7);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int ch = c8 + e;
                base[(ch >> 5) * 1024 + (ch & 31) * 16] = (uint8_t)(o[e >> 2] >> (8 * (e & 3)));
            }
        }
        if (c8 == 0) p.vs[row] = scale;
    }
}

struct VarParams {
    const uint8_t* qq; const float* qs; const uint8_t* kq; const float* ks; const void* vt; const float* vs;
    void* out;
    int64_t qh, kh, qn, kn, knp;
    int qblocks, causal, out_dtype;
    float log2_sm_scale;
    Strides ost;
    int d_out;
    const void* mask;
    int mask_dtype;
    int64_t ms_z, ms_h, ms_q;
};

template <int QK_FP8, int PVQ, int V_T, int D>
__global__ __launch_bounds__(256) void attn_fwd_var_kernel(const VarParams p) {
    constexpr int KK = D / 32;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ql = lane & 31, g = lane >> 5;
    const int64_t head_lin = blockIdx.x / p.qblocks;  // z * QH + h
    const int qblk = blockIdx.x % p.qblocks;
    const int64_t q0 = ((int64_t)qblk * 4 + wave) * 32;
    if (q0 >= p.qn) return;
    int64_t z, h;
    divmod(head_lin, p.qh, z, h);
    int64_t kvh, kvr;
    divmod(h * p.kh, p.qh, kvh, kvr);
    const int64_t kv_lin = z * p.kh + kvh;  // offset_k of triton_atten.py:212
    const int64_t qi = q0 + ql, qrow = qi < p.qn ? qi : p.qn - 1;
    v4i qf[KK];
    {
        const uint8_t* qp = p.qq + (head_lin * p.qn + qrow) * D + 16 * g;
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) qf[kk] = *(const v4i*)(qp + 32 * kk);
    }
    const float qsl = p.qs[head_lin * p.qn + qrow] * p.log2_sm_scale;
    v16f o[KK];
#pragma unroll
    for (int dd = 0; dd < KK; ++dd)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dd][r] = 0.0f;
    float m_i = -__builtin_inff();
    v2f l2 = {0.0f, 0.0f};
    // every load of the loop goes through a buffer descriptor (head base in SGPRs, constant per-lane offset, block offset in the scalar
    // operand; see attn_fwd_kernel), and the fragments of block kb + 1 are requested before the arithmetic of block kb starts
    auto rsK = SDNQ_MAKE_RSRC(p.kq + kv_lin * p.knp * D);
    auto rsS = SDNQ_MAKE_RSRC(p.ks + kv_lin * p.knp);
    auto rsVS = SDNQ_MAKE_RSRC((PVQ != PVQ_NONE ? p.vs : p.ks) + kv_lin * p.knp);
    constexpr int VTILE = (PVQ == PVQ_I8 || PVQ == PVQ_FP8) ? 1024 : 2048;  // bytes of V operand per (key block, 32-channel block)
    constexpr int NV = VTILE / 1024;
    auto rsV = SDNQ_MAKE_RSRC((const uint8_t*)p.vt + kv_lin * (p.knp / 32) * (int64_t)(KK * VTILE));
    const int lofs = lane * 16, sofs = 32 * g;
    struct Blk { v4i kf[KK]; v4i vf[KK][NV]; v4f ks4[4]; v4f vs4[4]; };
    auto load_blk = [&](int kb, Blk& b) {
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) b.kf[kk] = SDNQ_BUF_LOAD16(rsK, lofs, kb * (KK * 1024) + kk * 1024);
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            b.ks4[2 * c] = __builtin_bit_cast(v4f, SDNQ_BUF_LOAD16(rsS, sofs, kb * 128 + 64 * c));
            b.ks4[2 * c + 1] = __builtin_bit_cast(v4f, SDNQ_BUF_LOAD16(rsS, sofs, kb * 128 + 64 * c + 16));
            if constexpr (PVQ != PVQ_NONE) {
                b.vs4[2 * c] = __builtin_bit_cast(v4f, SDNQ_BUF_LOAD16(rsVS, sofs, kb * 128 + 64 * c));
                b.vs4[2 * c + 1] = __builtin_bit_cast(v4f, SDNQ_BUF_LOAD16(rsVS, sofs, kb * 128 + 64 * c + 16));
            }
        }
#pragma unroll
        for (int dd = 0; dd < KK; ++dd)
#pragma unroll
            for (int c = 0; c < NV; ++c) b.vf[dd][c] = SDNQ_BUF_LOAD16(rsV, lofs, kb * (KK * VTILE) + (dd * NV + c) * 1024);
    };
    const char* mrow = nullptr;
    if (p.mask != nullptr)
        mrow = (const char*)p.mask + (z * p.ms_z + h * p.ms_h + qrow * p.ms_q) * (p.mask_dtype == -1 ? 1 : (p.mask_dtype == SDNQ_F32 ? 4 : 2));
    int nkb = (int)((p.kn + 31) / 32);
    if (p.causal) {
        const int lim = (int)(q0 / 32) + 1;  // blocks past the last query of this wave are fully masked (triton_atten.py:255)
        nkb = nkb < lim ? nkb : lim;
    }
    auto compute = [&](const Blk& b, int kb) {
        const int64_t key0 = (int64_t)kb * 32;
        const v4i (&kf)[KK] = b.kf;
        const v4f (&ks4)[4] = b.ks4;
        const v4f (&vs4)[4] = b.vs4;
        // wave-uniform: does this block need any of the three masks?
        const bool need_mask = key0 + 32 > p.kn || (p.causal && key0 + 31 > q0) || mrow != nullptr;
        // ---- scores: lane holds keys key0 + 16 (r >> 3) + 8 g + (r & 7), r = 0..15, of query q0 + ql
        float sf[16];
        if constexpr (!QK_FP8) {
            v16i s;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = 0;
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) s = __builtin_amdgcn_mfma_i32_32x32x32_i8(kf[kk], qf[kk], s, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 16; ++r) sf[r] = (float)s[r];
        } else {
            v16f s;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = 0.0f;
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) {
                const long ka = ((long)(u32)kf[kk][1] << 32) | (u32)kf[kk][0], kb2 = ((long)(u32)kf[kk][3] << 32) | (u32)kf[kk][2];
                const long qa = ((long)(u32)qf[kk][1] << 32) | (u32)qf[kk][0], qb = ((long)(u32)qf[kk][3] << 32) | (u32)qf[kk][2];
                s = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(ka, qa, s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(kb2, qb, s, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) sf[r] = s[r];
        }
        float t[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float kscale = ks4[r >> 2][r & 3];
            float tv = sf[r] * kscale * qsl;   // (acc * k_scale) * (q_scale * log2_sm_scale): the factor order of the default kernel
            const int64_t key = key0 + 16 * (r >> 3) + 8 * g + (r & 7);
            bool ok = true;
            float add = 0.0f;
            if (need_mask) {
                ok = key < p.kn;                           // triton_atten.py:295-296
                if (p.causal) ok = ok && key <= qi;        // :287-288
                if (mrow != nullptr && ok) {
                    if (p.mask_dtype == -1) ok = ((const int8_t*)mrow)[key] != 0;   // :290-291
                    else add = ldf_mask(mrow, key, p.mask_dtype);                   // :292-293
                }
            }
            t[r] = ok ? tv + add : -__builtin_inff();
        }
        float m_blk = t[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) m_blk = fmaxf(m_blk, t[r]);
        {
            const u32 mb = __float_as_uint(m_blk);
            const auto sw = __builtin_amdgcn_permlane32_swap(mb, mb, false, false);
            m_blk = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
        }
        const float m_new = fmaxf(m_i, m_blk);
        const bool dead = m_new == -__builtin_inff();   // no visible key so far: alpha = 1, p = 0 (:299-301)
        const float alpha = dead ? 1.0f : __builtin_amdgcn_exp2f(m_i - m_new);
        m_i = m_new;
        const float m_use = dead ? 0.0f : m_new;
        v2f psum = {0.0f, 0.0f};
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            t[r] = __builtin_amdgcn_exp2f(t[r] - m_use);
            psum[r & 1] += t[r];
        }
        l2 = l2 * alpha + psum;  // l_i = fma(l_i, alpha, sum(p)), :308
#pragma unroll
        for (int dd = 0; dd < KK; ++dd) o[dd] *= alpha;
        // ---- P.V
        if constexpr (PVQ == PVQ_NONE) {
            v4i pf[2];
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int w = 0; w < 4; ++w) pf[c][w] = (int)pack2<V_T>(t[8 * c + 2 * w], t[8 * c + 2 * w + 1]);   // p.to(v.dtype), :332
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int dd = 0; dd < KK; ++dd) {
                    const v4i vf = b.vf[dd][c];
                    if constexpr (V_T == SDNQ_BF16)
                        o[dd] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(v8bf, vf), __builtin_bit_cast(v8bf, pf[c]), o[dd], 0, 0, 0);
                    else
                        o[dd] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(v8h, vf), __builtin_bit_cast(v8h, pf[c]), o[dd], 0, 0, 0);
                }
        } else {
            // p *= v_scale; p_scale = max(p, 1) / qmax, 1 where tiny (:311-319)
            float pmax = 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                t[r] *= vs4[r >> 2][r & 3];
                pmax = fmaxf(pmax, t[r]);
            }
            {
                const u32 mb = __float_as_uint(pmax);
                const auto sw = __builtin_amdgcn_permlane32_swap(mb, mb, false, false);
                pmax = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
            }
            float ps = pmax * (PVQ == PVQ_I8 ? (float)(1.0 / 127.0) : (PVQ == PVQ_FP8 ? (float)(1.0 / 448.0) : (float)(1.0 / 65504.0)));
            if (ps <= 2e-38f) ps = 1.0f;
            const float inv = 1.0f / ps;  // tl.fdiv(1.0, p_scale)
            if constexpr (PVQ == PVQ_I8) {
                v4i pq;
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    u32 word = 0;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float qv = __builtin_floorf(__builtin_fmaf(t[4 * w + e], inv, 0.5f));  // <= 127
                        word |= ((u32)(int)qv & 0xffu) << (8 * e);
                    }
                    pq[w] = (int)word;
                }
#pragma unroll
                for (int dd = 0; dd < KK; ++dd) {
                    const v4i vf = b.vf[dd][0];
                    v16i acc;
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[r] = 0;
                    acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(vf, pq, acc, 0, 0, 0);
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[dd][r] = __builtin_fmaf((float)acc[r], ps, o[dd][r]);  // :315
                }
            } else if constexpr (PVQ == PVQ_FP8) {
                float c[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) c[r] = fminf(t[r] * inv, 448.0f);  // (max * (1 / (max / 448)) can land an ulp above 448)
                const u32 w0 = pack4_e4m3fn_clamped(c[0], c[1], c[2], c[3]), w1 = pack4_e4m3fn_clamped(c[4], c[5], c[6], c[7]);
                const u32 w2 = pack4_e4m3fn_clamped(c[8], c[9], c[10], c[11]), w3 = pack4_e4m3fn_clamped(c[12], c[13], c[14], c[15]);
                const long pa = ((long)w1 << 32) | w0, pb = ((long)w3 << 32) | w2;
#pragma unroll
                for (int dd = 0; dd < KK; ++dd) {
                    const v4i vf = b.vf[dd][0];
                    const long va = ((long)(u32)vf[1] << 32) | (u32)vf[0], vb = ((long)(u32)vf[3] << 32) | (u32)vf[2];
                    v16f acc;
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(va, pa, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(vb, pb, acc, 0, 0, 0);
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[dd][r] = __builtin_fmaf(acc[r], ps, o[dd][r]);  // :323
                }
            } else {
                v4i pf[2];
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int w = 0; w < 4; ++w)
                        pf[c][w] = (int)((u32)f32_to_f16_bits(t[8 * c + 2 * w] * inv) | ((u32)f32_to_f16_bits(t[8 * c + 2 * w + 1] * inv) << 16));
#pragma unroll
                for (int dd = 0; dd < KK; ++dd) {
                    v16f acc;
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        const v4i vf = b.vf[dd][c];
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(v8h, vf), __builtin_bit_cast(v8h, pf[c]), acc, 0, 0, 0);
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[dd][r] = __builtin_fmaf(acc[r], ps, o[dd][r]);
                }
            }
        }
    };
    if constexpr (D == 128) {
        // (measured, tools/attn_variants_lab.py: with two register sets of 128-channel fragments the quantized-P.V kernels lose occupancy --
        //  24 x 4608^2 x 128 int8 / int8 907 us against 721 without the prefetch; at head_dim 64 the prefetch wins, 292 -> 259 us)
#pragma nounroll
        for (int kb = 0; kb < nkb; ++kb) {
            Blk b;
            load_blk(kb, b);
            compute(b, kb);
        }
    } else if (nkb > 0) {
        Blk bA, bB;
        load_blk(0, bA);
#pragma nounroll
        for (int kb = 0; kb < nkb; kb += 2) {  // two blocks per trip: the register sets swap roles without moves
            if (kb + 1 < nkb) load_blk(kb + 1, bB);
            compute(bA, kb);
            if (kb + 1 >= nkb) break;
            if (kb + 2 < nkb) load_blk(kb + 2, bA);
            compute(bB, kb + 1);
        }
    }
    if (qi >= p.qn) return;
    float l_i = l2[0] + l2[1];
    l_i += __shfl_xor(l_i, 32);
    const float inv = l_i > 0.0f ? 1.0f / l_i : 0.0f;  // acc *= fdiv(1.0, l_i), :336; a row with no visible key is 0
    const int ob = p.out_dtype == SDNQ_F32 ? 4 : 2;
    char* orow = (char*)p.out + p.ost.at(head_lin, qi) * ob;
#pragma unroll
    for (int dd = 0; dd < KK; ++dd)
#pragma unroll
        for (int t4 = 0; t4 < 4; ++t4) {
            const int dcol = 32 * dd + 8 * t4 + 4 * g;  // registers 4t..4t+3 are 4 consecutive channels
            if (dcol >= p.d_out) continue;
            float f[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) f[u] = o[dd][4 * t4 + u] * inv;
            if (p.out_dtype == SDNQ_F32) {
                *(float4*)(orow + dcol * 4) = make_float4(f[0], f[1], f[2], f[3]);
            } else if (p.out_dtype == SDNQ_BF16) {
                *(uint2*)(orow + dcol * 2) = make_uint2(pack2<SDNQ_BF16>(f[0], f[1]), pack2<SDNQ_BF16>(f[2], f[3]));
            } else {
                *(uint2*)(orow + dcol * 2) = make_uint2(pack2<SDNQ_F16>(f[0], f[1]), pack2<SDNQ_F16>(f[2], f[3]));
            }
        }
}

template <int QK_FP8, int PVQ, int V_T>
int launch_var(const VarParams& p, int d, int64_t blocks, hipStream_t s) {
    const dim3 grid((unsigned)blocks), block(256);
    if (d == 64) hipLaunchKernelGGL((attn_fwd_var_kernel<QK_FP8, PVQ, V_T, 64>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((attn_fwd_var_kernel<QK_FP8, PVQ, V_T, 128>), grid, block, 0, s, p);
    SDNQ_CHECK_LAUNCH();
    return SDNQ_OK;
}

int pvq_of(int pv_dtype) { return pv_dtype < 0 ? PVQ_NONE : (pv_dtype == SDNQ_MM_I8 ? PVQ_I8 : (pv_dtype == SDNQ_MM_FP8 ? PVQ_FP8 : (pv_dtype == SDNQ_MM_F16 ? PVQ_F16 : -1))); }

}  // namespace

extern "C" int sdnq_hip_attn_prepare_ex(const void* q, const void* k, const void* v, int dtype, int64_t batch, int64_t q_heads, int64_t kv_heads,
                                        int64_t q_len, int64_t kv_len, int64_t head_dim, int smooth_k, int hadamard_group, const int64_t* q_strides,
                                        const int64_t* k_strides, const int64_t* v_strides, int qk_dtype, int pv_dtype, void* qq, float* qs, void* kq,
                                        float* ks, void* vt, float* vs, float* kmean, sdnq_stream_t stream) {
    const int pvq = pvq_of(pv_dtype);
    if (!q || !k || !v || !qq || !qs || !kq || !ks || !vt || (smooth_k && !kmean) || (pvq != PVQ_NONE && !vs)) return SDNQ_ERR_NULL;
    if ((qk_dtype != SDNQ_MM_I8 && qk_dtype != SDNQ_MM_FP8) || pvq < 0) return SDNQ_ERR_DTYPE;
    if (!shape_ok(batch, q_heads, kv_heads, q_len, kv_len, head_dim)) return SDNQ_ERR_SHAPE;
    if (head_dim < 8 || head_dim > 128 || head_dim % 8) return SDNQ_ERR_UNSUPPORTED;
    if (dtype != SDNQ_BF16 && dtype != SDNQ_F16) return SDNQ_ERR_UNSUPPORTED;
    const int64_t head_dim_src = head_dim;
    head_dim = head_dim <= 64 ? 64 : 128;
    int log2g = 0;
    if (hadamard_group != 0) {
        if (hadamard_group < 4 || hadamard_group > head_dim || (hadamard_group & (hadamard_group - 1)) || head_dim % hadamard_group) return SDNQ_ERR_SHAPE;
        while ((1 << log2g) < hadamard_group) ++log2g;
    }
    if (((uintptr_t)q | (uintptr_t)qq | (uintptr_t)k | (uintptr_t)v | (uintptr_t)kq | (uintptr_t)vt) % 16) return SDNQ_ERR_ALIGN;
    hipStream_t s = (hipStream_t)stream;
    const int d = (int)head_dim, lpr = d / 8;
    VarPrepParams p{};
    p.q = q; p.k = k; p.v = v; p.qq = (uint8_t*)qq; p.kq = (uint8_t*)kq; p.qs = qs; p.ks = ks; p.vs = vs; p.vt = vt;
    auto strides_of = [&](const int64_t* st, int64_t heads, int64_t len, Strides& out) {
        out.heads = heads;
        if (st) { out.b = st[0]; out.h = st[1]; out.n = st[2]; } else { out.b = heads * len * head_dim_src; out.h = len * head_dim_src; out.n = head_dim_src; }
        return out.b % 8 == 0 && out.h % 8 == 0 && out.n % 8 == 0;  // 16-byte rows
    };
    if (!strides_of(q_strides, q_heads, q_len, p.qst) || !strides_of(k_strides, kv_heads, kv_len, p.kst) || !strides_of(v_strides, kv_heads, kv_len, p.vst))
        return SDNQ_ERR_ALIGN;
    p.qheads = batch * q_heads; p.kheads = batch * kv_heads; p.qn = q_len; p.kn = kv_len; p.knp = (kv_len + 31) / 32 * 32;
    p.d = d; p.d_src = (int)head_dim_src; p.log2g = log2g; p.qk_fp8 = qk_dtype == SDNQ_MM_FP8; p.pvq = pvq;
    p.kmean = smooth_k ? kmean : nullptr;
    p.nqb = (p.qheads * q_len * lpr + 255) / 256;
    p.nkb = p.kheads * p.knp * lpr / 256;  // exact: knp * lpr is a multiple of 256
    p.nvb = pvq == PVQ_NONE ? p.kheads * (p.knp / 32) : p.nkb;
    if (smooth_k) {
        if (dtype == SDNQ_BF16) hipLaunchKernelGGL((attn_var_kmean_kernel<SDNQ_BF16>), dim3((unsigned)p.kheads), dim3(256), 0, s, k, p.kst, kmean, kv_len, d, (int)head_dim_src);
        else hipLaunchKernelGGL((attn_var_kmean_kernel<SDNQ_F16>), dim3((unsigned)p.kheads), dim3(256), 0, s, k, p.kst, kmean, kv_len, d, (int)head_dim_src);
    }
    const int64_t blocks = p.nqb + p.nkb + p.nvb;
    if (dtype == SDNQ_BF16) hipLaunchKernelGGL((attn_var_prepare_kernel<SDNQ_BF16>), dim3((unsigned)blocks), dim3(256), 0, s, p);
    else hipLaunchKernelGGL((attn_var_prepare_kernel<SDNQ_F16>), dim3((unsigned)blocks), dim3(256), 0, s, p);
    SDNQ_CHECK_LAUNCH();
    return SDNQ_OK;
}

extern "C" int sdnq_hip_attn_fwd_ex(const void* qq, const float* qs, const void* kq, const float* ks, const void* vt, const float* vs, int v_dtype,
                                    int qk_dtype, int pv_dtype, float sm_scale, int is_causal, const void* mask, int mask_dtype, int64_t mask_stride_b,
                                    int64_t mask_stride_h, int64_t mask_stride_q, void* out, int out_dtype, const int64_t* out_strides, int64_t batch,
                                    int64_t q_heads, int64_t kv_heads, int64_t q_len, int64_t kv_len, int64_t head_dim, sdnq_stream_t stream) {
    const int pvq = pvq_of(pv_dtype);
    if (!qq || !qs || !kq || !ks || !vt || !out || (pvq != PVQ_NONE && !vs)) return SDNQ_ERR_NULL;
    if ((qk_dtype != SDNQ_MM_I8 && qk_dtype != SDNQ_MM_FP8) || pvq < 0) return SDNQ_ERR_DTYPE;
    if (out_dtype != SDNQ_F32 && out_dtype != SDNQ_BF16 && out_dtype != SDNQ_F16) return SDNQ_ERR_DTYPE;
    if (pvq == PVQ_NONE && v_dtype != SDNQ_BF16 && v_dtype != SDNQ_F16) return SDNQ_ERR_DTYPE;
    if (mask && mask_dtype != -1 && mask_dtype != SDNQ_F32 && mask_dtype != SDNQ_BF16 && mask_dtype != SDNQ_F16) return SDNQ_ERR_DTYPE;
    if (!shape_ok(batch, q_heads, kv_heads, q_len, kv_len, head_dim)) return SDNQ_ERR_SHAPE;
    if (head_dim < 8 || head_dim > 128 || head_dim % 8) return SDNQ_ERR_UNSUPPORTED;
    const int64_t head_dim_src = head_dim;
    head_dim = head_dim <= 64 ? 64 : 128;
    if (((uintptr_t)qq | (uintptr_t)kq | (uintptr_t)vt) % 16 || (uintptr_t)out % 8) return SDNQ_ERR_ALIGN;
    VarParams p{};
    p.qq = (const uint8_t*)qq; p.qs = qs; p.kq = (const uint8_t*)kq; p.ks = ks; p.vt = vt; p.vs = vs; p.out = out;
    p.qh = q_heads; p.kh = kv_heads; p.qn = q_len; p.kn = kv_len; p.knp = (kv_len + 31) / 32 * 32;
    p.qblocks = (int)((q_len + 127) / 128);
    p.causal = is_causal ? 1 : 0;
    p.out_dtype = out_dtype;
    p.log2_sm_scale = sm_scale * 1.4426950408889634f;  // triton_atten.py:203
    p.ost.heads = q_heads;
    if (out_strides) { p.ost.b = out_strides[0]; p.ost.h = out_strides[1]; p.ost.n = out_strides[2]; }
    else { p.ost.b = q_heads * q_len * head_dim_src; p.ost.h = q_len * head_dim_src; p.ost.n = head_dim_src; }
    p.d_out = (int)head_dim_src;
    if (p.ost.b % 4 || p.ost.h % 4 || p.ost.n % 4) return SDNQ_ERR_ALIGN;
    p.mask = mask; p.mask_dtype = mask_dtype; p.ms_z = mask_stride_b; p.ms_h = mask_stride_h; p.ms_q = mask_stride_q;
    const int64_t blocks = batch * q_heads * p.qblocks;
    hipStream_t s = (hipStream_t)stream;
    const int d = (int)head_dim;
    const bool f8 = qk_dtype == SDNQ_MM_FP8;
    switch (pvq) {
        case PVQ_NONE:
            if (v_dtype == SDNQ_BF16) return f8 ? launch_var<1, PVQ_NONE, SDNQ_BF16>(p, d, blocks, s) : launch_var<0, PVQ_NONE, SDNQ_BF16>(p, d, blocks, s);
            return f8 ? launch_var<1, PVQ_NONE, SDNQ_F16>(p, d, blocks, s) : launch_var<0, PVQ_NONE, SDNQ_F16>(p, d, blocks, s);
        case PVQ_I8: return f8 ? launch_var<1, PVQ_I8, SDNQ_F16>(p, d, blocks, s) : launch_var<0, PVQ_I8, SDNQ_F16>(p, d, blocks, s);
        case PVQ_FP8: return f8 ? launch_var<1, PVQ_FP8, SDNQ_F16>(p, d, blocks, s) : launch_var<0, PVQ_FP8, SDNQ_F16>(p, d, blocks, s);
        default: return f8 ? launch_var<1, PVQ_F16, SDNQ_F16>(p, d, blocks, s) : launch_var<0, PVQ_F16, SDNQ_F16>(p, d, blocks, s);
    }
}
