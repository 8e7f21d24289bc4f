// Convolution as GEMM on gfx950: the explicit im2col the reference's conv forwards feed to the same
// quantized matmul as the Linear layers.
//
//   process_conv_input (layers/conv/forward.py:30-76): F.unfold(x, kernel, padding, stride, dilation).transpose(1, 2)
//   -> rows m = (b, h_out, w_out), columns k = (c_in, i, j), zero padding.  Conv1d is the H = 1 case (:66-67).
//
// One workgroup copies a 64 (m) x 64 (k) tile: global reads are coalesced along w_out (consecutive m of one (c, i, j)),
// the tile is transposed through LDS and leaves as 16-byte row segments of the [M][K] matrix.  HBM-bound:
// reads ~ input * (kernel positions hit in L2), writes M*K*e bytes.
#include <hip/hip_runtime.h>

#include "../../include/sdnq_hip.h"
#include <type_traits>

#include "sdnq_dev.h"

namespace {

struct Im2colParams {
    const void* x;
    void* out;
    int B, C, H, W, KH, KW, SH, SW, PH, PW, DH, DW, HO, WO;
    int64_t M, K;
};

// ---- fused activation quantization of the conv matmul: thread = (output position m, input channel c) --------------------
// A thread walks the KH x KW patch of its channel: adjacent lanes are adjacent w_out, so every load is a coalesced run of the
// image row, there is no integer division in the loops, and the P = KH*KW values are P CONSECUTIVE columns k = c*P .. c*P+P-1
// of row m.  Reads hit L2 / MALL after the first touch (each pixel is visited once per kernel position).

// The row amax of the unfolded activation is a WINDOW maximum of the per-pixel channel amax: with
// A[b][h][w] = max_c |x[b][c][h][w]|, max_k |x_unfold[m][k]| = max over the KH x KW taps of A (zero padding adds nothing).
// So x is read ONCE (not once per tap) to build A, and the quantizing kernel derives its 64 row scales from <= P floats each.
// A is accumulated as the bit pattern of a non-negative float with atomicMax (zeroed by the host).  Workgroup = 32 groups of 8
// consecutive pixels (one 16-byte load per channel, 512 contiguous bytes per wave-half) x 8 channel lanes; blockIdx.y splits the
// channels so that small images still fill the chip; partial maxima are combined in LDS, one atomic per pixel per workgroup.
__global__ __launch_bounds__(256) void zero_words_kernel(unsigned int* __restrict__ p, int64_t n) {
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i + 3 < n) *(uint4*)(p + i) = make_uint4(0, 0, 0, 0);
    else for (int64_t j = i; j < n; ++j) p[j] = 0;
}

template <int T_ID>
__global__ __launch_bounds__(256) void conv_pixel_amax_kernel(const void* __restrict__ x, int64_t pixels, int channels, int64_t total,
                                                              int cpb, unsigned int* __restrict__ amap) {
    __shared__ float red[8][32][9];
    const int tid = threadIdx.x, pg = tid & 31, cl = tid >> 5;
    const int64_t q = (int64_t)blockIdx.x * 32 + pg;  // group of 8 consecutive pixels of one image (pixels % 8 == 0)
    const bool ok = q * 8 < total;
    int64_t b = 0, px = 0;
    if (ok) divmod(q * 8, pixels, b, px);  // (32-bit whenever it fits, sdnq_dev.h)
    const int c_begin = blockIdx.y * cpb, c_end = (c_begin + cpb < channels) ? c_begin + cpb : channels;
    float a[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) a[e] = 0.0f;
    if (ok) {
#pragma unroll 4
        for (int c = c_begin + cl; c < c_end; c += 8) {
            float v[8];
            const int64_t off = (b * channels + c) * pixels + px;
            if constexpr (T_ID == SDNQ_F32) {
                Vec16<SDNQ_F32>::unpack(*(const uint4*)((const float*)x + off), v);
                Vec16<SDNQ_F32>::unpack(*(const uint4*)((const float*)x + off + 4), v + 4);
            } else {
                Vec16<T_ID>::unpack(*(const uint4*)((const uint16_t*)x + off), v);
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) a[e] = fmaxf(a[e], fabsf(v[e]));
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[cl][pg][e] = a[e];
    __syncthreads();
    // 256 threads -> the 256 pixels of the workgroup
    const int pg2 = tid >> 3, e2 = tid & 7;
    const int64_t q2 = (int64_t)blockIdx.x * 32 + pg2;
    if (q2 * 8 < total) {
        float r = red[0][pg2][e2];
#pragma unroll
        for (int l = 1; l < 8; ++l) r = fmaxf(r, red[l][pg2][e2]);
        atomicMax(amap + q2 * 8 + e2, __float_as_uint(r));
    }
}

// xq[m][c*P + pos] = quantize(x patch) with xs[m] = amax_ws[m] / qmax -- the arithmetic of rowquant.hip (IEEE division,
// round-half-even, clamp; 0/0 -> 0).  64 rows x CT channels per workgroup (CT*P % 16 == 0), bytes staged in LDS rows of
// CT*P (+4 pad) bytes, written out as 16-byte pieces of the [M][K] rows.  Workgroups with blockIdx.y == 0 also write xs.
template <int T_ID, int MM, int CT>
__global__ __launch_bounds__(256) void conv_quant_kernel(const Im2colParams p, const unsigned int* amap, float qmax,
                                                         uint8_t* __restrict__ xq, float* __restrict__ xs, unsigned int* ticket, int64_t map_words) {
    SDNQ_KERNARGS_NOW("s"(p.x), "s"(p.out), "s"(p.B), "s"(p.C), "s"(p.H), "s"(p.W), "s"(p.KH), "s"(p.KW), "s"(p.SH), "s"(p.SW), "s"(p.PH), "s"(p.PW), "s"(p.DH),
                      "s"(p.DW), "s"(p.HO), "s"(p.WO), "s"(p.M), "s"(p.K));  // one batch of kernarg loads (sdnq_dev.h)
    extern __shared__ __attribute__((aligned(16))) uint8_t tile[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int P = p.KH * p.KW;
    const int pitch = CT * P + 4;  // words per row odd -> the 64 lanes of a byte store hit distinct banks
    const int64_t m0 = (int64_t)blockIdx.x * 64, m = m0 + lane;
    const bool mok = m < p.M;
    const int c_base = blockIdx.y * CT;
    if (mok) {
        int64_t t64, wo64, b64, ho64;  // pixel -> (image, row, column): two 32-bit divisions (sdnq_dev.h: divmod), not four 64-bit ones
        divmod(m, p.WO, t64, wo64);
        divmod(t64, p.HO, b64, ho64);
        const int wo = (int)wo64, ho = (int)ho64, b = (int)b64;
        const int64_t img = (int64_t)p.H * p.W;
        float amax = 0.0f;  // window maximum of the channel-amax map = amax of this unfolded row
        for (int i = 0; i < p.KH; ++i) {
            const int h = ho * p.SH - p.PH + i * p.DH;
            if (h < 0 || h >= p.H) continue;
            for (int j = 0; j < p.KW; ++j) {
                const int ww = wo * p.SW - p.PW + j * p.DW;
                if (ww >= 0 && ww < p.W) amax = fmaxf(amax, __uint_as_float(amap[(int64_t)b * img + (int64_t)h * p.W + ww]));
            }
        }
        const float scale = amax / qmax;
        if (blockIdx.y == 0 && w == 0) xs[m] = scale;
        RowDiv rd;  // correctly rounded x / scale in 3 VALU (sdnq_dev.h); the scale differs per lane here, so the fast form is used
        rd.set(scale);  // when EVERY lane's scale qualifies
        const bool all_fast = __all(rd.fast);
        constexpr int CPT = CT / 4;  // channels per thread
        const int c0 = c_base + w * CPT;
        uint8_t* row = tile + lane * pitch + (w * CPT) * P;
        for (int i = 0; i < p.KH; ++i) {
            const int h = ho * p.SH - p.PH + i * p.DH;
            for (int j = 0; j < p.KW; ++j) {
                const int ww = wo * p.SW - p.PW + j * p.DW;
                const bool inb = h >= 0 && h < p.H && ww >= 0 && ww < p.W;
                // UNCONDITIONAL loads from a clamped (always valid) address, zeroed afterwards: a load under a condition gets its
                // own s_waitcnt vmcnt(0), and the CPT x KH x KW loads of a thread then run one memory round trip after the other
                const int hc = h < 0 ? 0 : (h >= p.H ? p.H - 1 : h), wc = ww < 0 ? 0 : (ww >= p.W ? p.W - 1 : ww);
                const int64_t base = ((int64_t)b * p.C) * img + (int64_t)hc * p.W + wc;
                float v[CPT];
#pragma unroll
                for (int u = 0; u < CPT; ++u) {
                    const int cc = c0 + u < p.C ? c0 + u : p.C - 1;
                    v[u] = FT<T_ID>::load(p.x, base + (int64_t)cc * img);
                }
#pragma unroll
                for (int u = 0; u < CPT; ++u) v[u] = (inb && c0 + u < p.C) ? v[u] : 0.0f;
                if constexpr (MM == SDNQ_MM_I8) {
                    if (all_fast) {  // wave-uniform: every scale of the wave is an ordinary number (not 0), |x| <= amax keeps |x / scale| <= 127.5
                        // rint and the int8 cast without conversions: q + 1.5 * 2^23 rounds to nearest even like rint and leaves the two's
                        // complement byte in the low mantissa bits, which the byte store takes as is (sdnq_dev.h: pack4_rne_i8)
#pragma unroll
                        for (int u = 0; u < CPT; ++u) row[u * P + i * p.KW + j] = (uint8_t)__float_as_uint(rd.fastdiv(v[u]) + 12582912.0f);
                        continue;
                    }
                }
                if (all_fast) {  // wave-uniform
#pragma unroll
                    for (int u = 0; u < CPT; ++u) {
                        const float x0 = v[u];
                        v[u] = rd.fastdiv(x0);
                        if constexpr (MM == SDNQ_MM_FP8) v[u] = (x0 == 0.0f) ? x0 : v[u];  // -0.0 keeps its sign (fp8 code 0x80), see rowquant.hip: quant8
                    }
                } else {
#pragma unroll
                    for (int u = 0; u < CPT; ++u) v[u] = v[u] / scale;
                }
#pragma unroll
                for (int u = 0; u < CPT; ++u) {
                    uint8_t byte;
                    const float qd = v[u];
                    if constexpr (MM == SDNQ_MM_I8) {
                        float q = (scale == 0.0f) ? 0.0f : __builtin_rintf(qd);
                        q = fminf(fmaxf(q, -128.0f), 127.0f);
                        byte = (uint8_t)((int)q & 0xff);
                    } else {
                        float q = qd;
                        if (q != q) q = 0.0f;
                        q = fminf(fmaxf(q, -448.0f), 448.0f);
                        byte = f32_to_e4m3fn_clamped(q);
                    }
                    row[u * P + i * p.KW + j] = byte;
                }
            }
        }
    }
    __syncthreads();
    if (ticket != nullptr) {
        // self-cleaning amax map (sdnq_hip_im2col_rowquant_z): every workgroup has read its window maxima by now; the LAST one to say so
        // zeroes the map and the ticket for the next call -- the separate zeroing launch (4.9 us x 49 convs of an SDXL step) goes away
        // two levels of tickets: thousands of workgroups on ONE counter serialize at ~25 ns per atomic (measured: +65 us per convolution);
        // the channel tiles of a row strip meet on that strip's counter (one cache line each, 256 of them), the strips on the global one
        __shared__ unsigned int s_last;
        // (no fence: an agent-scope release writes the XCD's whole L2 back -- +65 us per convolution when every workgroup did one.  None is
        //  needed: the map reads below the barrier above have RETURNED -- their values went into the scales this workgroup already used --
        //  and the device-scope atomics order the tickets; the zero stores leave this XCD's L2 with the end-of-kernel release like any output)
        if (tid == 0) {
            unsigned int* rowc = ticket + 32 + (blockIdx.x & 255u) * 32;
            const unsigned int per_row = gridDim.y * ((gridDim.x - (blockIdx.x & 255u) + 255u) / 256u);  // workgroups that share this strip counter
            unsigned int last = 0;
            if (atomicAdd(rowc, 1u) == per_row - 1) {
                *rowc = 0;
                const unsigned int strips = gridDim.x < 256u ? gridDim.x : 256u;
                last = atomicAdd(ticket, 1u) == strips - 1 ? 1u : 0u;
            }
            s_last = last;
        }
        __syncthreads();
        if (s_last) {
            unsigned int* mapw = const_cast<unsigned int*>(amap);
            for (int64_t i = (int64_t)tid * 4; i < map_words; i += 1024) {
                if (i + 3 < map_words) *(uint4*)(mapw + i) = make_uint4(0, 0, 0, 0);
                else for (int64_t j = i; j < map_words; ++j) mapw[j] = 0;
            }
            if (tid == 0) *ticket = 0;
        }
    }
    const int c_valid = (p.C - c_base) < CT ? (p.C - c_base) : CT;
    const int cpr = c_valid * P / 16;  // 16-byte pieces per row (C*P % 16 == 0 and CT*P % 16 == 0)
    for (int ch = tid; ch < 64 * cpr; ch += 256) {
        const int r = ch / cpr, idx = ch - r * cpr;
        if (m0 + r >= p.M) continue;
        const u32* src = (const u32*)(tile + r * pitch + idx * 16);
        *(uint4*)(xq + (m0 + r) * p.K + (int64_t)c_base * P + idx * 16) = make_uint4(src[0], src[1], src[2], src[3]);
    }
}

// explicit copy (float [M][K] matrix for the float / SVD / zero-point branches); T = element type, 2 or 4 bytes
template <typename T>
__global__ __launch_bounds__(256) void im2col_kernel(const Im2colParams p) {
    SDNQ_KERNARGS_NOW("s"(p.x), "s"(p.out), "s"(p.B), "s"(p.C), "s"(p.H), "s"(p.W), "s"(p.KH), "s"(p.KW), "s"(p.SH), "s"(p.SW), "s"(p.PH), "s"(p.PW), "s"(p.DH),
                      "s"(p.DW), "s"(p.HO), "s"(p.WO), "s"(p.M), "s"(p.K));
    typedef T O;
    constexpr int EPC = 16 / sizeof(O);  // elements per 16-byte chunk
    __shared__ O tile[64][64 + EPC];     // [k][m]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int64_t m0 = (int64_t)blockIdx.x * 64, k0 = (int64_t)blockIdx.y * 64;
    const int64_t m = m0 + lane;
    const bool mok = m < p.M;
    int64_t t64, wo64, b64, ho64;
    divmod(m, p.WO, t64, wo64);
    divmod(t64, p.HO, b64, ho64);
    const int wo = (int)wo64, ho = (int)ho64, b = (int)b64;
    const T* xb = (const T*)p.x + (int64_t)b * p.C * p.H * p.W;
    const int P = p.KH * p.KW;
#pragma unroll 4
    for (int kk = w; kk < 64; kk += 4) {
        const int64_t k = k0 + kk;
        T v = 0;
        if (mok && k < p.K) {
            const int c = (int)(k / P), r = (int)(k - (int64_t)c * P), i = r / p.KW, j = r - i * p.KW;
            const int h = ho * p.SH - p.PH + i * p.DH, ww = wo * p.SW - p.PW + j * p.DW;
            if (h >= 0 && h < p.H && ww >= 0 && ww < p.W) v = xb[((int64_t)c * p.H + h) * p.W + ww];
        }
        tile[kk][lane] = v;
    }
    __syncthreads();
    constexpr int CPR = 64 / EPC;  // 16-byte chunks per tile row
    for (int ch = tid; ch < 64 * CPR; ch += 256) {
        const int r = ch / CPR, c16 = ch % CPR;
        const int64_t gm = m0 + r, gk = k0 + c16 * EPC;
        if (gm >= p.M || gk >= p.K) continue;  // K % EPC == 0: a chunk never straddles K
        O tmp[EPC];
#pragma unroll
        for (int e = 0; e < EPC; ++e) tmp[e] = tile[c16 * EPC + e][r];
        *(uint4*)((O*)p.out + gm * p.K + gk) = *(const uint4*)tmp;
    }
}

int fill_geometry(Im2colParams& p, const void* x, void* out, int batch, int channels, int height, int width, int kh, int kw,
                  int stride_h, int stride_w, int pad_h, int pad_w, int dil_h, int dil_w) {
    if (!x || !out) return SDNQ_ERR_NULL;
    if (batch <= 0 || channels <= 0 || height <= 0 || width <= 0 || kh <= 0 || kw <= 0 || stride_h <= 0 || stride_w <= 0 ||
        pad_h < 0 || pad_w < 0 || dil_h <= 0 || dil_w <= 0)
        return SDNQ_ERR_SHAPE;
    p.x = x; p.out = out; p.B = batch; p.C = channels; p.H = height; p.W = width; p.KH = kh; p.KW = kw;
    p.SH = stride_h; p.SW = stride_w; p.PH = pad_h; p.PW = pad_w; p.DH = dil_h; p.DW = dil_w;
    p.HO = (height + 2 * pad_h - dil_h * (kh - 1) - 1) / stride_h + 1;
    p.WO = (width + 2 * pad_w - dil_w * (kw - 1) - 1) / stride_w + 1;
    if (p.HO <= 0 || p.WO <= 0) return SDNQ_ERR_SHAPE;
    p.M = (int64_t)batch * p.HO * p.WO;
    p.K = (int64_t)channels * kh * kw;
    if ((uintptr_t)out % 16) return SDNQ_ERR_ALIGN;
    return SDNQ_OK;
}

}  // namespace

namespace {
// self_clean: amax_ws = [header: global ticket, then 256 strip tickets one cache line apart][batch * height * width words], ALL zero on
// entry, left all zero by the launch
constexpr int SDNQ_CONV_WS_HEADER_WORDS = 32 + 256 * 32;
int im2col_rowquant_impl(const void* x, int dtype, int batch, int channels, int height, int width, int kh, int kw,
                         int stride_h, int stride_w, int pad_h, int pad_w, int dil_h, int dil_w, int mm_dtype,
                         void* xq, float* xs, void* amax_ws, bool self_clean, sdnq_stream_t stream) {
    if (dtype < 0 || dtype > 2) return SDNQ_ERR_DTYPE;
    if (mm_dtype != SDNQ_MM_I8 && mm_dtype != SDNQ_MM_FP8) return SDNQ_ERR_DTYPE;
    if (!xs || !amax_ws) return SDNQ_ERR_NULL;
    Im2colParams p{};
    int st = fill_geometry(p, x, xq, batch, channels, height, width, kh, kw, stride_h, stride_w, pad_h, pad_w, dil_h, dil_w);
    if (st != SDNQ_OK) return st;
    const int P = kh * kw;
    if (p.K % 16) return SDNQ_ERR_SHAPE;
    if (P > 25) return SDNQ_ERR_UNSUPPORTED;  // LDS tile of 64 x 16*P bytes; larger kernels take im2col + rowquant
    hipStream_t s = (hipStream_t)stream;
    const int64_t pixels = (int64_t)height * width, total = (int64_t)batch * pixels;
    if (pixels % 8) return SDNQ_ERR_UNSUPPORTED;  // the channel-amax pass reads 8 pixels per 16-byte load
    if ((uintptr_t)x % 16 || (uintptr_t)amax_ws % 16) return SDNQ_ERR_ALIGN;
    // zero the amax map with a plain kernel: hipMemsetAsync goes through the runtime's generic fill kernel (4.6 us per call
    // in the SDXL conv step, profiles/r01_bench_sdxl_conv_kernel_stats.csv)
    unsigned int* ticket = nullptr;
    if (self_clean) {
        ticket = (unsigned int*)amax_ws;
        amax_ws = (unsigned int*)amax_ws + SDNQ_CONV_WS_HEADER_WORDS;
    } else {
        hipLaunchKernelGGL(zero_words_kernel, dim3((unsigned)((total + 1023) / 1024)), dim3(256), 0, s, (unsigned int*)amax_ws, total);
    }
    const float qmax = (mm_dtype == SDNQ_MM_I8) ? 127.0f : 448.0f;
    const unsigned mb = (unsigned)((p.M + 63) / 64);
    const int ct = (P <= 9) ? 32 : 16;
    const unsigned nx = (unsigned)((total / 8 + 31) / 32);
    unsigned ny = (1024 + nx - 1) / nx;  // enough workgroups for 256 CUs, at least 8 channels (one per channel lane) each
    if (ny > (unsigned)(channels + 7) / 8) ny = (unsigned)(channels + 7) / 8;
    if (ny < 1) ny = 1;
    const int cpb = (int)(((channels + ny - 1) / ny + 7) / 8 * 8);
    ny = (unsigned)((channels + cpb - 1) / cpb);
    dim3 g1(nx, ny), g2(mb, (unsigned)((channels + ct - 1) / ct)), block(256);
    const size_t lds = (size_t)64 * (ct * P + 4);
#define CQ2(TID, MMV)                                                                                                      \
    do {                                                                                                                   \
        if (ct == 32) hipLaunchKernelGGL((conv_quant_kernel<TID, MMV, 32>), g2, block, lds, s, p, (const unsigned int*)amax_ws, qmax, (uint8_t*)xq, xs, ticket, total); \
        else hipLaunchKernelGGL((conv_quant_kernel<TID, MMV, 16>), g2, block, lds, s, p, (const unsigned int*)amax_ws, qmax, (uint8_t*)xq, xs, ticket, total); \
    } while (0)
#define CQ(TID)                                                                                      \
    do {                                                                                             \
        hipLaunchKernelGGL((conv_pixel_amax_kernel<TID>), g1, block, 0, s, x, pixels, channels, total, cpb, (unsigned int*)amax_ws); \
        if (mm_dtype == SDNQ_MM_I8) CQ2(TID, SDNQ_MM_I8);                                            \
        else CQ2(TID, SDNQ_MM_FP8);                                                                  \
    } while (0)
    if (dtype == SDNQ_F32) CQ(SDNQ_F32);
    else if (dtype == SDNQ_BF16) CQ(SDNQ_BF16);
    else CQ(SDNQ_F16);
#undef CQ
#undef CQ2
    SDNQ_CHECK_LAUNCH();
    return SDNQ_OK;
}
}  // namespace

extern "C" int sdnq_hip_im2col_rowquant(const void* x, int dtype, int batch, int channels, int height, int width, int kh, int kw,
                                        int stride_h, int stride_w, int pad_h, int pad_w, int dil_h, int dil_w, int mm_dtype,
                                        void* xq, float* xs, void* amax_ws, sdnq_stream_t stream) {
    return im2col_rowquant_impl(x, dtype, batch, channels, height, width, kh, kw, stride_h, stride_w, pad_h, pad_w, dil_h, dil_w, mm_dtype, xq, xs,
                                amax_ws, false, stream);
}

extern "C" int sdnq_hip_im2col_rowquant_z(const void* x, int dtype, int batch, int channels, int height, int width, int kh, int kw,
                                          int stride_h, int stride_w, int pad_h, int pad_w, int dil_h, int dil_w, int mm_dtype,
                                          void* xq, float* xs, void* zeroed_ws, sdnq_stream_t stream) {
    return im2col_rowquant_impl(x, dtype, batch, channels, height, width, kh, kw, stride_h, stride_w, pad_h, pad_w, dil_h, dil_w, mm_dtype, xq, xs,
                                zeroed_ws, true, stream);
}

extern "C" int sdnq_hip_im2col(const void* x, int dtype, int batch, int channels, int height, int width, int kh, int kw, int stride_h,
                               int stride_w, int pad_h, int pad_w, int dil_h, int dil_w, void* out, sdnq_stream_t stream) {
    if (dtype < 0 || dtype > 2) return SDNQ_ERR_DTYPE;
    Im2colParams p{};
    int st = fill_geometry(p, x, out, batch, channels, height, width, kh, kw, stride_h, stride_w, pad_h, pad_w, dil_h, dil_w);
    if (st != SDNQ_OK) return st;
    const int eb = (dtype == SDNQ_F32) ? 4 : 2;
    if ((p.K * eb) % 16) return SDNQ_ERR_SHAPE;
    dim3 grid((unsigned)((p.M + 63) / 64), (unsigned)((p.K + 63) / 64)), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (eb == 4) hipLaunchKernelGGL((im2col_kernel<uint32_t>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((im2col_kernel<uint16_t>), grid, block, 0, s, p);
    SDNQ_CHECK_LAUNCH();
    return SDNQ_OK;
}
