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
#include "sdnq_dev.h"

namespace {

struct Im2colParams {
    const void* x;
    void* out;
    int B, C, H, W, KH, KW, SH, SW, PH, PW, DH, DW, HO, WO;
    int64_t M, K;
};

template <typename T>
__global__ __launch_bounds__(256) void im2col_kernel(const Im2colParams p) {
    constexpr int EPC = 16 / sizeof(T);  // elements per 16-byte chunk
    __shared__ T tile[64][64 + EPC];     // [k][m]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int64_t m0 = (int64_t)blockIdx.x * 64, k0 = (int64_t)blockIdx.y * 64;
    const int64_t m = m0 + lane;
    const bool mok = m < p.M;
    const int wo = (int)(m % p.WO), ho = (int)((m / p.WO) % p.HO), b = (int)(m / ((int64_t)p.WO * p.HO));
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
        T tmp[EPC];
#pragma unroll
        for (int e = 0; e < EPC; ++e) tmp[e] = tile[c16 * EPC + e][r];
        *(uint4*)((T*)p.out + gm * p.K + gk) = *(const uint4*)tmp;
    }
}

}  // namespace

extern "C" int sdnq_hip_im2col(const void* x, int dtype, int batch, int channels, int height, int width, int kh, int kw, int stride_h,
                               int stride_w, int pad_h, int pad_w, int dil_h, int dil_w, void* out, sdnq_stream_t stream) {
    if (!x || !out) return SDNQ_ERR_NULL;
    if (dtype < 0 || dtype > 2) return SDNQ_ERR_DTYPE;
    if (batch <= 0 || channels <= 0 || height <= 0 || width <= 0 || kh <= 0 || kw <= 0 || stride_h <= 0 || stride_w <= 0 ||
        pad_h < 0 || pad_w < 0 || dil_h <= 0 || dil_w <= 0)
        return SDNQ_ERR_SHAPE;
    Im2colParams p{};
    p.x = x; p.out = out; p.B = batch; p.C = channels; p.H = height; p.W = width; p.KH = kh; p.KW = kw;
    p.SH = stride_h; p.SW = stride_w; p.PH = pad_h; p.PW = pad_w; p.DH = dil_h; p.DW = dil_w;
    p.HO = (height + 2 * pad_h - dil_h * (kh - 1) - 1) / stride_h + 1;
    p.WO = (width + 2 * pad_w - dil_w * (kw - 1) - 1) / stride_w + 1;
    if (p.HO <= 0 || p.WO <= 0) return SDNQ_ERR_SHAPE;
    p.M = (int64_t)batch * p.HO * p.WO;
    p.K = (int64_t)channels * kh * kw;
    const int eb = (dtype == SDNQ_F32) ? 4 : 2;
    if ((p.K * eb) % 16) return SDNQ_ERR_SHAPE;
    if ((uintptr_t)out % 16) return SDNQ_ERR_ALIGN;
    dim3 grid((unsigned)((p.M + 63) / 64), (unsigned)((p.K + 63) / 64)), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (eb == 4) hipLaunchKernelGGL((im2col_kernel<uint32_t>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((im2col_kernel<uint16_t>), grid, block, 0, s, p);
    SDNQ_CHECK_LAUNCH();
    return SDNQ_OK;
}
