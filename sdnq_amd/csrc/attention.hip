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
//   The K fragment rows are fetched in a permuted order (bits 2 and 3 of the row index swapped) so that registers 8c..8c+7 of
//   lane group g hold the CONTIGUOUS keys 16c + 8g .. + 7: exactly the K-slice that lane feeds to PV MFMA c.  P never moves
//   between lanes and the V^T fragments are plain 16-byte loads.
//   O^T = V^T.P^T: query stays on (l & 31), so alpha / 1/l are per-lane scalars.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/sdnq_hip.h"
#include "sdnq_dev.h"

namespace {

// ---- K channel means over the tokens of one (batch, head) ------------------------------------------------------------
template <int T_ID>
__global__ __launch_bounds__(256) void attn_kmean_kernel(const void* __restrict__ k, float* __restrict__ mean, int64_t kn, int d) {
    __shared__ float red[256 * 8];
    const int lpr = d / 8, rpp = 256 / lpr;  // lanes per token row, rows per pass
    const int tid = threadIdx.x, c8 = (tid % lpr) * 8, r0 = tid / lpr;
    const char* base = (const char*)k + (int64_t)blockIdx.x * kn * d * FT<T_ID>::bytes;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int64_t r = r0; r < kn; r += rpp) {
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += FT<T_ID>::load(base, r * d + c8 + e);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[tid * 8 + e] = acc[e];
    __syncthreads();
    if (tid < d) {
        float s = 0.0f;
        const int lane_of = tid / 8, e = tid % 8;
        for (int r = 0; r < rpp; ++r) s += red[(r * lpr + lane_of) * 8 + e];
        mean[(int64_t)blockIdx.x * d + tid] = s / (float)kn;
    }
}

// ---- per-token int8 quantization of [rows][d] (d / 8 lanes per row), optional mean subtraction ---------------------------
template <int T_ID>
__global__ __launch_bounds__(256) void attn_quant_kernel(const void* __restrict__ x, const float* __restrict__ mean, int8_t* __restrict__ xq,
                                                         float* __restrict__ xs, int64_t rows, int64_t rows_per_head, int d) {
    const int lpr = d / 8;
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t row = t / lpr;
    const int c8 = (int)(t % lpr) * 8;
    const bool live = row < rows;
    const int64_t rr = live ? row : rows - 1;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = FT<T_ID>::load(x, rr * d + c8 + e);
    if (mean != nullptr) {
        const float* mu = mean + (rr / rows_per_head) * d + c8;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] -= mu[e];  // k.to(float32).sub_(mean), triton_atten.py:459-463
    }
    float amax = 0.0f;
#pragma unroll
    for (int e = 0; e < 8; ++e) amax = fmaxf(amax, fabsf(v[e]));
    for (int m = 1; m < lpr; m <<= 1) amax = fmaxf(amax, __shfl_xor(amax, m));
    const float scale = amax / 127.0f;
    u32 o[2] = {0, 0};
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        float q = __builtin_rintf(v[e] / scale);
        if (q != q) q = 0.0f;
        q = fminf(fmaxf(q, -128.0f), 127.0f);
        o[e >> 2] |= ((u32)(int)q & 0xffu) << (8 * (e & 3));
    }
    if (live) {
        *(uint2*)(xq + row * d + c8) = make_uint2(o[0], o[1]);
        if (c8 == 0) xs[row] = scale;
    }
}

// ---- V [heads][kn][d] -> V^T [heads][d][knp] (knp = kn rounded up to 32, zero padded) ------------------------------------
__global__ __launch_bounds__(256) void attn_vt_kernel(const uint16_t* __restrict__ v, uint16_t* __restrict__ vt, int64_t kn, int64_t knp, int d) {
    __shared__ uint16_t tile[32][128 + 2];
    const int64_t head = blockIdx.y, key0 = (int64_t)blockIdx.x * 32;
    const uint16_t* src = v + head * kn * d;
    uint16_t* dst = vt + head * d * knp;
    const int lpr = d / 8;
    for (int t = threadIdx.x; t < 32 * lpr; t += 256) {
        const int kr = t / lpr, c8 = (t % lpr) * 8;
        uint4 val = make_uint4(0, 0, 0, 0);
        if (key0 + kr < kn) val = *(const uint4*)(src + (key0 + kr) * d + c8);
        const uint16_t* h = (const uint16_t*)&val;
#pragma unroll
        for (int e = 0; e < 8; ++e) tile[kr][c8 + e] = h[e];
    }
    __syncthreads();
    for (int t = threadIdx.x; t < d * 4; t += 256) {
        const int dd = t / 4, k8 = (t % 4) * 8;
        u32 w[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) w[e] = (u32)tile[k8 + 2 * e][dd] | ((u32)tile[k8 + 2 * e + 1][dd] << 16);
        *(uint4*)(dst + (int64_t)dd * knp + key0 + k8) = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

struct AttnParams {
    const int8_t* qq; const float* qs; const int8_t* kq; const float* ks; const uint16_t* vt;
    void* out;
    int64_t qh, kh, qn, kn, knp;
    int qblocks;
    float log2_sm_scale;
};

// ---- forward: one wave = 32 queries of one head; 4 waves per workgroup -------------------------------------------------
template <int V_T, int OUT_T, int D, bool CAUSAL>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const AttnParams p) {
    constexpr int KK = D / 32;  // int8 MFMA K steps of Q.K^T; also the 32-channel blocks of O
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ql = lane & 31, g = lane >> 5;
    const int64_t head_lin = blockIdx.x / p.qblocks;  // z * QH + h
    const int qblk = blockIdx.x % p.qblocks;
    const int64_t q0 = (int64_t)qblk * 128 + wave * 32;
    if (q0 >= p.qn) return;  // wave-uniform
    const int64_t z = head_lin / p.qh, h = head_lin % p.qh;
    const int64_t kv_lin = z * p.kh + (h * p.kh) / p.qh;  // offset_k of triton_atten.py:212 (grouped-query mapping)

    const int64_t qi = q0 + ql, qrow = qi < p.qn ? qi : p.qn - 1;
    const int8_t* qbase = p.qq + (head_lin * p.qn + qrow) * D + 16 * g;
    v4i qf[KK];
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) qf[kk] = *(const v4i*)(qbase + 32 * kk);
    const float qscale = p.qs[head_lin * p.qn + qrow];

    v16f o[KK];
#pragma unroll
    for (int dd = 0; dd < KK; ++dd)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dd][r] = 0.0f;
    float m_i = -__builtin_inff(), l_i = 1.0f;  // triton_atten.py:231-232

    const int8_t* kbase = p.kq + kv_lin * p.kn * D + 16 * g;
    const float* ksbase = p.ks + kv_lin * p.kn;
    const uint16_t* vbase = p.vt + (kv_lin * D + ql) * p.knp + 8 * g;
    const int prow = (ql & 0x13) | ((ql & 4) << 1) | ((ql & 8) >> 1);  // K fragment row permutation: swap bits 2 and 3

    int64_t nkb = (p.kn + 31) / 32;
    if (CAUSAL) {
        const int64_t lim = (q0 + 31) / 32 + 1;  // blocks past the last query of this wave are fully masked (triton_atten.py:255)
        nkb = nkb < lim ? nkb : lim;
    }
#pragma nounroll
    for (int64_t kb = 0; kb < nkb; ++kb) {
        const int64_t key0 = kb * 32;
        int64_t kr = key0 + prow;
        if (kr >= p.kn) kr = p.kn - 1;
        v4i kf[KK];
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) kf[kk] = *(const v4i*)(kbase + kr * D + 32 * kk);
        v4i vf[KK][2];
#pragma unroll
        for (int dd = 0; dd < KK; ++dd)
#pragma unroll
            for (int c = 0; c < 2; ++c) vf[dd][c] = *(const v4i*)(vbase + (int64_t)dd * 32 * p.knp + key0 + 16 * c);
        float ksc[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int64_t key = key0 + 16 * (r >> 3) + 8 * g + (r & 7);
            ksc[r] = ksbase[key < p.kn ? key : p.kn - 1];
        }
        v16i s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0;
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) s = __builtin_amdgcn_mfma_i32_32x32x32_i8(kf[kk], qf[kk], s, 0, 0, 0);

        float qk[16];
        float m_blk = -__builtin_inff();
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int64_t key = key0 + 16 * (r >> 3) + 8 * g + (r & 7);
            float val = (((float)s[r] * qscale) * ksc[r]) * p.log2_sm_scale;  // triton_atten.py:278
            bool ok = key < p.kn;                                                // :295-296
            if (CAUSAL) ok = ok && key <= qi;                                    // :287-288
            qk[r] = ok ? val : -__builtin_inff();
            m_blk = fmaxf(m_blk, qk[r]);
        }
        m_blk = fmaxf(m_blk, __shfl_xor(m_blk, 32));
        const float m_new = fmaxf(m_i, m_blk);          // finite from block 0 on: key 0 is valid for every query
        const float alpha = __builtin_amdgcn_exp2f(m_i - m_new);
        float psum = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            qk[r] = __builtin_amdgcn_exp2f(qk[r] - m_new);
            psum += qk[r];
        }
        psum += __shfl_xor(psum, 32);
        l_i = fmaf(l_i, alpha, psum);  // :308
        m_i = m_new;
#pragma unroll
        for (int dd = 0; dd < KK; ++dd)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[dd][r] *= alpha;
        // P in the value dtype (p.to(v.dtype), :332), packed as the second MFMA operand: 8 keys per lane group and K step
        v4i pf[2];
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const u32 lo = FT<V_T>::bits(qk[8 * c + 2 * w]), hi = FT<V_T>::bits(qk[8 * c + 2 * w + 1]);
                pf[c][w] = (int)(lo | (hi << 16));
            }
#pragma unroll
        for (int dd = 0; dd < KK; ++dd)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                if constexpr (V_T == SDNQ_BF16)
                    o[dd] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(v8bf, vf[dd][c]), __builtin_bit_cast(v8bf, pf[c]), o[dd], 0, 0, 0);
                else
                    o[dd] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(v8h, vf[dd][c]), __builtin_bit_cast(v8h, pf[c]), o[dd], 0, 0, 0);
            }
    }
    if (qi >= p.qn) return;
    const float inv = 1.0f / l_i;  // acc *= fdiv(1.0, l_i), :336
    char* orow = (char*)p.out + (head_lin * p.qn + qi) * D * FT<OUT_T>::bytes;
#pragma unroll
    for (int dd = 0; dd < KK; ++dd)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int dcol = 32 * dd + 8 * t + 4 * g;  // registers 4t..4t+3 are 4 consecutive channels
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
    if (causal) hipLaunchKernelGGL((attn_fwd_kernel<V_T, OUT_T, D, true>), dim3((unsigned)blocks), dim3(256), 0, s, p);
    else hipLaunchKernelGGL((attn_fwd_kernel<V_T, OUT_T, D, false>), dim3((unsigned)blocks), dim3(256), 0, s, p);
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
                                     int64_t kv_heads, int64_t q_len, int64_t kv_len, int64_t head_dim, int smooth_k, void* qq,
                                     float* qs, void* kq, float* ks, void* vt, float* kmean, sdnq_stream_t stream) {
    if (!q || !k || !v || !qq || !qs || !kq || !ks || !vt || (smooth_k && !kmean)) return SDNQ_ERR_NULL;
    if (!shape_ok(batch, q_heads, kv_heads, q_len, kv_len, head_dim)) return SDNQ_ERR_SHAPE;
    if (head_dim != 64 && head_dim != 128) return SDNQ_ERR_UNSUPPORTED;
    if (dtype != SDNQ_BF16 && dtype != SDNQ_F16) return SDNQ_ERR_UNSUPPORTED;  // PV runs in the value dtype on the matrix cores
    if (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)qq | (uintptr_t)kq | (uintptr_t)vt) % 16) return SDNQ_ERR_ALIGN;
    hipStream_t s = (hipStream_t)stream;
    const int d = (int)head_dim, lpr = d / 8;
    const int64_t kheads = batch * kv_heads, qrows = batch * q_heads * q_len, krows = kheads * kv_len;
    const int64_t knp = (kv_len + 31) / 32 * 32;
#define ATTN_T(T)                                                                                                                  \
    do {                                                                                                                           \
        if (smooth_k) hipLaunchKernelGGL((attn_kmean_kernel<T>), dim3((unsigned)kheads), dim3(256), 0, s, k, kmean, kv_len, d);    \
        hipLaunchKernelGGL((attn_quant_kernel<T>), dim3((unsigned)((qrows * lpr + 255) / 256)), dim3(256), 0, s, q,                \
                           (const float*)nullptr, (int8_t*)qq, qs, qrows, q_len, d);                                               \
        hipLaunchKernelGGL((attn_quant_kernel<T>), dim3((unsigned)((krows * lpr + 255) / 256)), dim3(256), 0, s, k,                \
                           smooth_k ? (const float*)kmean : (const float*)nullptr, (int8_t*)kq, ks, krows, kv_len, d);             \
    } while (0)
    if (dtype == SDNQ_BF16) ATTN_T(SDNQ_BF16);
    else ATTN_T(SDNQ_F16);
#undef ATTN_T
    hipLaunchKernelGGL(attn_vt_kernel, dim3((unsigned)(knp / 32), (unsigned)kheads), dim3(256), 0, s, (const uint16_t*)v, (uint16_t*)vt,
                       kv_len, knp, d);
    SDNQ_CHECK_LAUNCH();
    return SDNQ_OK;
}

extern "C" int sdnq_hip_attn_fwd(const void* qq, const float* qs, const void* kq, const float* ks, const void* vt, int v_dtype,
                                 float sm_scale, int is_causal, void* out, int out_dtype, int64_t batch, int64_t q_heads,
                                 int64_t kv_heads, int64_t q_len, int64_t kv_len, int64_t head_dim, sdnq_stream_t stream) {
    if (!qq || !qs || !kq || !ks || !vt || !out) return SDNQ_ERR_NULL;
    if (!shape_ok(batch, q_heads, kv_heads, q_len, kv_len, head_dim)) return SDNQ_ERR_SHAPE;
    if (((uintptr_t)qq | (uintptr_t)kq | (uintptr_t)vt | (uintptr_t)out) % 16) return SDNQ_ERR_ALIGN;
    AttnParams p{};
    p.qq = (const int8_t*)qq; p.qs = qs; p.kq = (const int8_t*)kq; p.ks = ks; p.vt = (const uint16_t*)vt; p.out = out;
    p.qh = q_heads; p.kh = kv_heads; p.qn = q_len; p.kn = kv_len; p.knp = (kv_len + 31) / 32 * 32;
    p.qblocks = (int)((q_len + 127) / 128);
    p.log2_sm_scale = sm_scale * 1.4426950408889634f;  // triton_atten.py:203
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
