// Scaled int8 / fp8 matmul for gfx950 (CDNA4):  out = cast(fma(f32(A.B^T) * sa[m], sb[n], bias))
//
// Replaces the reference's Triton kernel sdnq_scaled_mm_kernel (kernels/triton_scaled_mm.py:127-236,
// wrapper :239-275) and its eager twins int_scaled_mm_torch / fp8_scaled_mm_torch
// (kernel_wrappers.py:132-144).  Same contract, different machine mapping:
//   * both operands are K-contiguous in HBM (a [M][K]; b physical [N][K] == the reference's
//     b[K,N] with strides (1,K)), so every MFMA fragment is one 16-byte (int8) / 32-byte (fp8)
//     K-contiguous vector per lane -- no transposes anywhere;
//   * int8 -> v_mfma_i32_32x32x32_i8 (exact int32 accumulate, so any K order is bit-identical);
//     fp8  -> v_mfma_scale_f32_32x32x64_f8f6f4 with neutral E8M0 scales (the only full-rate fp8 MFMA);
//   * MFMA computes the TRANSPOSED tile (A-operand = weight rows n, B-operand = activation rows m):
//     a lane then owns ONE activation row m and 16 output channels in runs of 4, so the epilogue
//     packs 4 results per 8-byte LDS store and the tile leaves the CU as full 16-byte row segments;
//   * LDS tile rows are 128 B (BK = 128 bytes of K) with the 16-byte chunk index XOR-swizzled by
//     (row >> 1) & 7: conflict-free ds_read_b128 for the 32-row fragment pattern;
//   * epilogue: fma(f32(acc) * sa[m], sb[n], bias) -- single-rounding FMA like tl.fma
//     (triton_scaled_mm.py:225) and CPU addcmul; int32 -> f32 conversion is RNE above 2^24.
//   * optional fused low-rank (SVD) bias: bias2d[m][n] = cast_svd(f32(bias[n]) + sum_r t[m][r] * up[n][r])
//     (linear_int8.py:57-62) and zero-point term f32(rowsum[m]) * sa[m] * zp[n] (linear_int8.py:65-69)
//     are produced in the epilogue instead of materialising an [M][N] bias in HBM.
#include "sdnq_dev.h"

namespace {

constexpr int BKB = 128;  // bytes of K per LDS stage row

struct GemmParams {
    const uint8_t* a;   // [M][K]
    const uint8_t* b;   // [N][K]
    const float* sa;    // [M]
    const float* sb;    // [N]
    const void* bias;   // [N] or [M][ld_bias] or null
    void* out;          // [M][N]
    const void* lr_t;   // [M][R]  low-rank activations (svd dtype) or null
    const void* lr_up;  // [N][R]
    const int32_t* zp_rowsum;  // [M] or null
    const float* zp;           // [N] or null
    int64_t M, N, K;
    int64_t ld_bias;
    int bias_ndim;
    int rank;
    int tiles_m, tiles_n;
};

template <int MM> struct MmaTraits;
template <> struct MmaTraits<SDNQ_MM_I8> {
    typedef v16i acc_t;
    static constexpr int KB = 32;  // bytes of K per MFMA per operand row
    static __device__ __forceinline__ void zero(acc_t& c) {
#pragma unroll
        for (int i = 0; i < 16; ++i) c[i] = 0;
    }
    static __device__ __forceinline__ float tof(const acc_t& c, int i) { return (float)c[i]; }
};
template <> struct MmaTraits<SDNQ_MM_FP8> {
    typedef v16f acc_t;
    static constexpr int KB = 64;
    static __device__ __forceinline__ void zero(acc_t& c) {
#pragma unroll
        for (int i = 0; i < 16; ++i) c[i] = 0.0f;
    }
    static __device__ __forceinline__ float tof(const acc_t& c, int i) { return c[i]; }
};

// LDS byte offset of 16-byte chunk c (0..7) of tile row r; rows are 128 B, chunk XOR-swizzled.
__device__ __forceinline__ int lds_off(int r, int c) { return r * BKB + ((c ^ ((r >> 1) & 7)) << 4); }

template <int T_ID> __device__ __forceinline__ float ldf(const void* p, int64_t i) { return FT<T_ID>::load(p, i); }

// BM x BN block tile, WM x WN wave tile (multiples of 32), 256 threads.
template <int MM, int OUT_T, int BIAS_T, int BM, int BN, int WM, int WN, bool LOWRANK>
__global__ __launch_bounds__(256) void gemm_kernel(const GemmParams p) {
    typedef MmaTraits<MM> MT;
    constexpr int WAVES_M = BM / WM, WAVES_N = BN / WN;
    static_assert(WAVES_M * WAVES_N == 4, "4 waves per block");
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int A_CHUNKS = BM * 8 / 256, B_CHUNKS = BN * 8 / 256;  // 16-byte chunks per thread per stage
    constexpr int OUT_B = FT<OUT_T>::bytes;
    constexpr int STAGE_ROW = BN * OUT_B + 16;  // epilogue staging row stride (bytes)
    constexpr int MAIN_BYTES = 2 * (BM + BN) * BKB;
    constexpr int EPI_BYTES = BM * STAGE_ROW;
    constexpr int LDS_BYTES = MAIN_BYTES > EPI_BYTES ? MAIN_BYTES : EPI_BYTES;
    __shared__ __attribute__((aligned(16))) uint8_t lds[LDS_BYTES];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;

    // XCD-aware tile order: consecutive ids on one XCD walk the n-tiles of one m-strip, so the strip of
    // A (the larger operand at diffusion shapes) stays in that XCD's L2. Block b runs on XCD b % 8.
    const int nwg = p.tiles_m * p.tiles_n;
    int bid = blockIdx.x;
    {
        const int q = nwg / 8, r = nwg % 8, xcd = bid % 8, j = bid / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    }
    const int tile_m = bid / p.tiles_n, tile_n = bid % p.tiles_n;
    const int64_t m0 = (int64_t)tile_m * BM, n0 = (int64_t)tile_n * BN;
    const int64_t K = p.K;

    // global->LDS staging assignment: chunk id = i*256 + tid -> row = id/8, chunk = id%8
    const uint8_t* ga[A_CHUNKS];
    const uint8_t* gb[B_CHUNKS];
    int la[A_CHUNKS], lb[B_CHUNKS];
    const int ck = tid & 7;
#pragma unroll
    for (int i = 0; i < A_CHUNKS; ++i) {
        const int r = (i * 256 + tid) >> 3;
        int64_t gm = m0 + r;
        if (gm >= p.M) gm = p.M - 1;  // clamp: rows past M are computed on valid memory and never stored
        ga[i] = p.a + gm * K + ck * 16;
        la[i] = lds_off(r, ck);
    }
#pragma unroll
    for (int i = 0; i < B_CHUNKS; ++i) {
        const int r = (i * 256 + tid) >> 3;
        int64_t gn = n0 + r;
        if (gn >= p.N) gn = p.N - 1;
        gb[i] = p.b + gn * K + ck * 16;
        lb[i] = BM * BKB + lds_off(r, ck);
    }

    typename MT::acc_t acc[TN][TM];  // [n-subtile][m-subtile]; MFMA A-operand = weights (n), B-operand = activations (m)
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) MT::zero(acc[i][j]);

    const int nk = (int)((K + BKB - 1) / BKB);
    uint4 ra[A_CHUNKS], rb[B_CHUNKS];
    auto gload = [&](int kt) {
        const int64_t k0 = (int64_t)kt * BKB;
        const bool ok = (k0 + ck * 16) < K;  // K % 16 == 0, so a chunk is fully in or fully out
#pragma unroll
        for (int i = 0; i < A_CHUNKS; ++i) ra[i] = ok ? *(const uint4*)(ga[i] + k0) : make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < B_CHUNKS; ++i) rb[i] = ok ? *(const uint4*)(gb[i] + k0) : make_uint4(0, 0, 0, 0);
    };
    auto lstore = [&](int buf) {
        uint8_t* base = lds + buf * (BM + BN) * BKB;
#pragma unroll
        for (int i = 0; i < A_CHUNKS; ++i) *(uint4*)(base + la[i]) = ra[i];
#pragma unroll
        for (int i = 0; i < B_CHUNKS; ++i) *(uint4*)(base + lb[i]) = rb[i];
    };

    gload(0);
    lstore(0);
    __syncthreads();

    const int frow = lane & 31, fgrp = lane >> 5;
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) gload(kt + 1);
        const uint8_t* sA = lds + buf * (BM + BN) * BKB;
        const uint8_t* sB = sA + BM * BKB;
#pragma unroll
        for (int ks = 0; ks < BKB / MT::KB; ++ks) {
            if constexpr (MM == SDNQ_MM_I8) {
                v4i fa[TM], fb[TN];
#pragma unroll
                for (int j = 0; j < TM; ++j) fa[j] = *(const v4i*)(sA + lds_off(wm * WM + j * 32 + frow, ks * 2 + fgrp));
#pragma unroll
                for (int i = 0; i < TN; ++i) fb[i] = *(const v4i*)(sB + lds_off(wn * WN + i * 32 + frow, ks * 2 + fgrp));
#pragma unroll
                for (int i = 0; i < TN; ++i)
#pragma unroll
                    for (int j = 0; j < TM; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fb[i], fa[j], acc[i][j], 0, 0, 0);
            } else {
                v8i fa[TM], fb[TN];
#pragma unroll
                for (int j = 0; j < TM; ++j) {
                    const int r = wm * WM + j * 32 + frow;
                    const v4i lo = *(const v4i*)(sA + lds_off(r, ks * 4 + fgrp * 2));
                    const v4i hi = *(const v4i*)(sA + lds_off(r, ks * 4 + fgrp * 2 + 1));
                    fa[j] = (v8i){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                }
#pragma unroll
                for (int i = 0; i < TN; ++i) {
                    const int r = wn * WN + i * 32 + frow;
                    const v4i lo = *(const v4i*)(sB + lds_off(r, ks * 4 + fgrp * 2));
                    const v4i hi = *(const v4i*)(sB + lds_off(r, ks * 4 + fgrp * 2 + 1));
                    fb[i] = (v8i){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                }
#pragma unroll
                for (int i = 0; i < TN; ++i)
#pragma unroll
                    for (int j = 0; j < TM; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fb[i], fa[j], acc[i][j], 0, 0, 0,
                                                                                    0x7f7f7f7f, 0, 0x7f7f7f7f);
            }
        }
        if (kt + 1 < nk) lstore(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue ---------------------------------------------------------------------------------
    // acc[i][j][reg]: n = n0 + wn*WN + i*32 + (reg&3) + 8*(reg>>2) + 4*(lane>>5),  m = m0 + wm*WM + j*32 + (lane&31)
    uint8_t* stage = lds;  // [BM][STAGE_ROW]; main-loop buffers are dead (last iteration ended with a barrier)
#pragma unroll
    for (int j = 0; j < TM; ++j) {
        const int ml = wm * WM + j * 32 + frow;
        int64_t gm = m0 + ml;
        if (gm >= p.M) gm = p.M - 1;
        const float sa = p.sa[gm];
        float zsum = 0.0f;
        if constexpr (LOWRANK) {
            if (p.zp_rowsum) zsum = (float)p.zp_rowsum[gm] * sa;  // .to(f32).mul_(input_scale), linear_int8.py:66
        }
#pragma unroll
        for (int i = 0; i < TN; ++i) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int nl = wn * WN + i * 32 + e + 8 * q + 4 * fgrp;
                    int64_t gn = n0 + nl;
                    if (gn >= p.N) gn = p.N - 1;
                    const float v = MT::tof(acc[i][j], q * 4 + e) * sa;
                    const float sb = p.sb[gn];
                    if constexpr (LOWRANK) {
                        // bias2d = cast_svd(f32(bias[n]) + sum_r t[m][r]*up[n][r]); then + zp term; f32 into the fma
                        float bv = 0.0f;
                        bool has = false;
                        if (p.lr_t) {
                            float s = 0.0f;
                            for (int r = 0; r < p.rank; ++r)
                                s = fmaf(ldf<BIAS_T>(p.lr_t, gm * p.rank + r), ldf<BIAS_T>(p.lr_up, gn * p.rank + r), s);
                            if (p.bias) s += ldf<BIAS_T>(p.bias, gn);
                            bv = FT<BIAS_T>::round(s);
                            has = true;
                        } else if (p.bias) {
                            bv = ldf<BIAS_T>(p.bias, gn);
                            has = true;
                        }
                        if (p.zp) {
                            const float zb = zsum * p.zp[gn];
                            bv = has ? zb + bv : zb;  // zero_bias.add_(bias), linear_int8.py:67-68
                            has = true;
                        }
                        o[e] = has ? fmaf(v, sb, bv) : v * sb;
                    } else {
                        if (p.bias_ndim == 1) o[e] = fmaf(v, sb, ldf<BIAS_T>(p.bias, gn));
                        else if (p.bias_ndim == 2) o[e] = fmaf(v, sb, ldf<BIAS_T>(p.bias, gm * p.ld_bias + gn));
                        else o[e] = v * sb;
                    }
                }
                const int nl0 = wn * WN + i * 32 + 8 * q + 4 * fgrp;
                uint8_t* dst = stage + ml * STAGE_ROW + nl0 * OUT_B;
                if constexpr (OUT_T == SDNQ_F32) {
                    *(uint4*)dst = Vec16<SDNQ_F32>::pack(o);
                } else if constexpr (OUT_T == SDNQ_BF16) {
                    *(uint2*)dst = make_uint2((u32)f32_to_bf16_bits(o[0]) | ((u32)f32_to_bf16_bits(o[1]) << 16),
                                              (u32)f32_to_bf16_bits(o[2]) | ((u32)f32_to_bf16_bits(o[3]) << 16));
                } else {
                    *(uint2*)dst = make_uint2((u32)f32_to_f16_bits(o[0]) | ((u32)f32_to_f16_bits(o[1]) << 16),
                                              (u32)f32_to_f16_bits(o[2]) | ((u32)f32_to_f16_bits(o[3]) << 16));
                }
            }
        }
    }
    __syncthreads();
    // coalesced tile store: 16-byte vectors along n
    constexpr int VEC_PER_ROW = BN * OUT_B / 16;
    constexpr int ELEMS_PER_VEC = 16 / OUT_B;
    for (int v = tid; v < BM * VEC_PER_ROW; v += 256) {
        const int r = v / VEC_PER_ROW, c = v % VEC_PER_ROW;
        const int64_t gm = m0 + r, gn = n0 + (int64_t)c * ELEMS_PER_VEC;
        if (gm < p.M && gn < p.N) {  // N % 16 == 0 (utils.py:96-97) so a 16-byte vector never straddles N
            const uint4 val = *(const uint4*)(stage + r * STAGE_ROW + c * 16);
            *(uint4*)((uint8_t*)p.out + (gm * p.N + gn) * OUT_B) = val;
        }
    }
}

template <int MM, int OUT_T, int BIAS_T, bool LOWRANK>
int launch_tiles(const GemmParams& p0, hipStream_t s) {
    GemmParams p = p0;
    // tile choice: fill >= ~1 wave of CUs (256) when the problem allows it
    const int64_t t128 = ((p.M + 127) / 128) * ((p.N + 127) / 128);
    const int64_t t64x128 = ((p.M + 63) / 64) * ((p.N + 127) / 128);
    if (t128 >= 256 || t64x128 < 64) {
        p.tiles_m = (int)((p.M + 127) / 128);
        p.tiles_n = (int)((p.N + 127) / 128);
        hipLaunchKernelGGL((gemm_kernel<MM, OUT_T, BIAS_T, 128, 128, 64, 64, LOWRANK>), dim3(p.tiles_m * p.tiles_n), dim3(256), 0, s, p);
    } else if (t64x128 >= 256 || ((p.M + 63) / 64) * ((p.N + 63) / 64) < 64) {
        p.tiles_m = (int)((p.M + 63) / 64);
        p.tiles_n = (int)((p.N + 127) / 128);
        hipLaunchKernelGGL((gemm_kernel<MM, OUT_T, BIAS_T, 64, 128, 32, 64, LOWRANK>), dim3(p.tiles_m * p.tiles_n), dim3(256), 0, s, p);
    } else {
        p.tiles_m = (int)((p.M + 63) / 64);
        p.tiles_n = (int)((p.N + 63) / 64);
        hipLaunchKernelGGL((gemm_kernel<MM, OUT_T, BIAS_T, 64, 64, 32, 32, LOWRANK>), dim3(p.tiles_m * p.tiles_n), dim3(256), 0, s, p);
    }
    SDNQ_CHECK_LAUNCH();
    return SDNQ_OK;
}

template <int MM, bool LOWRANK>
int dispatch_types(const GemmParams& p, int out_dtype, int bias_dtype, hipStream_t s) {
#define GEMM_CASE(O, B) \
    if (out_dtype == O && bias_dtype == B) return launch_tiles<MM, O, B, LOWRANK>(p, s);
    GEMM_CASE(SDNQ_BF16, SDNQ_BF16)
    GEMM_CASE(SDNQ_BF16, SDNQ_F32)
    GEMM_CASE(SDNQ_F16, SDNQ_F16)
    GEMM_CASE(SDNQ_F16, SDNQ_F32)
    GEMM_CASE(SDNQ_F32, SDNQ_F32)
    GEMM_CASE(SDNQ_F32, SDNQ_BF16)
    GEMM_CASE(SDNQ_F32, SDNQ_F16)
#undef GEMM_CASE
    return SDNQ_ERR_DTYPE;
}

int check_common(int mm_dtype, const void* a, const void* b, const float* sa, const float* sb, void* out, int out_dtype,
                 int64_t m, int64_t n, int64_t k) {
    if (!a || !b || !sa || !sb || !out) return SDNQ_ERR_NULL;
    if (mm_dtype != SDNQ_MM_I8 && mm_dtype != SDNQ_MM_FP8) return SDNQ_ERR_DTYPE;
    if (out_dtype < 0 || out_dtype > 2) return SDNQ_ERR_DTYPE;
    if (m <= 0 || n <= 0 || k <= 0 || (k % 16) != 0 || (n % 8) != 0) return SDNQ_ERR_SHAPE;
    if (((uintptr_t)a % 16) || ((uintptr_t)b % 16) || ((uintptr_t)out % 16)) return SDNQ_ERR_ALIGN;
    return SDNQ_OK;
}

}  // namespace

extern "C" int sdnq_hip_scaled_mm(int mm_dtype, const void* a, const void* b, const float* sa, const float* sb,
                                  const void* bias, int bias_dtype, int bias_ndim, int64_t ld_bias, void* out,
                                  int out_dtype, int64_t m, int64_t n, int64_t k, sdnq_stream_t stream) {
    int st = check_common(mm_dtype, a, b, sa, sb, out, out_dtype, m, n, k);
    if (st != SDNQ_OK) return st;
    if (bias_ndim < 0 || bias_ndim > 2) return SDNQ_ERR_SHAPE;
    if (bias_ndim != 0 && !bias) return SDNQ_ERR_NULL;
    if (bias_ndim == 0) { bias = nullptr; bias_dtype = (out_dtype == SDNQ_F16) ? SDNQ_F16 : (out_dtype == SDNQ_BF16 ? SDNQ_BF16 : SDNQ_F32); }
    if (bias_dtype < 0 || bias_dtype > 2) return SDNQ_ERR_DTYPE;
    if (bias_ndim == 2 && ld_bias < n) return SDNQ_ERR_SHAPE;
    if ((n % 16) != 0 && out_dtype != SDNQ_F32) return SDNQ_ERR_SHAPE;
    GemmParams p{};
    p.a = (const uint8_t*)a; p.b = (const uint8_t*)b; p.sa = sa; p.sb = sb; p.bias = bias; p.out = out;
    p.M = m; p.N = n; p.K = k; p.ld_bias = ld_bias; p.bias_ndim = bias_ndim;
    hipStream_t s = (hipStream_t)stream;
    if (mm_dtype == SDNQ_MM_I8) return dispatch_types<SDNQ_MM_I8, false>(p, out_dtype, bias_dtype, s);
    return dispatch_types<SDNQ_MM_FP8, false>(p, out_dtype, bias_dtype, s);
}

extern "C" int sdnq_hip_scaled_mm_lowrank(int mm_dtype, const void* a, const void* b, const float* sa, const float* sb,
                                          const void* bias, int bias_dtype, const void* t, const void* svd_up,
                                          int svd_dtype, int rank, const int32_t* zp_rowsum, const float* zp, void* out,
                                          int out_dtype, int64_t m, int64_t n, int64_t k, sdnq_stream_t stream) {
    int st = check_common(mm_dtype, a, b, sa, sb, out, out_dtype, m, n, k);
    if (st != SDNQ_OK) return st;
    if ((t == nullptr) != (svd_up == nullptr)) return SDNQ_ERR_NULL;
    if ((zp_rowsum == nullptr) != (zp == nullptr)) return SDNQ_ERR_NULL;
    if (t && (rank <= 0 || rank > 1024)) return SDNQ_ERR_SHAPE;
    if ((n % 16) != 0 && out_dtype != SDNQ_F32) return SDNQ_ERR_SHAPE;
    // the [M][N] bias of the reference lives in the svd dtype (addmm in svd_down.dtype, linear_int8.py:60);
    // a 1-D bias is cast to it first, so bias/t/up share one element type here.
    int bt = t ? svd_dtype : (bias ? bias_dtype : out_dtype);
    if (bias && t && bias_dtype != svd_dtype) return SDNQ_ERR_DTYPE;
    if (bt < 0 || bt > 2) return SDNQ_ERR_DTYPE;
    GemmParams p{};
    p.a = (const uint8_t*)a; p.b = (const uint8_t*)b; p.sa = sa; p.sb = sb; p.bias = bias; p.out = out;
    p.lr_t = t; p.lr_up = svd_up; p.rank = rank; p.zp_rowsum = zp_rowsum; p.zp = zp;
    p.M = m; p.N = n; p.K = k; p.bias_ndim = bias ? 1 : 0;
    hipStream_t s = (hipStream_t)stream;
    if (mm_dtype == SDNQ_MM_I8) return dispatch_types<SDNQ_MM_I8, true>(p, out_dtype, bt, s);
    return dispatch_types<SDNQ_MM_FP8, true>(p, out_dtype, bt, s);
}
