/*
 * sdnq_hip.h -- C ABI of the MI355X (gfx950 / CDNA4) dequantize-and-matmul hot path.
 *
 * The reference (Disty0/sdnq 0.2.5) is pure Python and has NO FFI; this header is the seam a
 * maintainer would bind (ctypes stub in INTEGRATION.md).  Every entry point replaces one piece
 * of the reference's per-call chain.  Citations are file:line under /root/reference/src/sdnq/.
 *
 * Contract (all entry points):
 *   - plain pointers + sizes only; device pointers unless stated; no ownership transfer;
 *   - stream-ordered and non-blocking on `stream` (a hipStream_t passed as void*);
 *   - no allocation: outputs and workspace are caller-provided;
 *   - returns SDNQ_OK (0) or a negative SdnqStatus; never throws across the ABI;
 *   - thread-safe: no mutable global state on any data path.  The one process-wide knob, sdnq_hip_set_tile_override (a tuning /
 *     test aid that pins the GEMM tile configuration; atomic, default "off"), changes which kernel computes a result, never the
 *     result: every configuration is bit-identical for int8 and within the stated tolerance for fp8 / float.
 *
 * Memory layouts ("physical" = what is in HBM):
 *   weight   physical [N][K] with K contiguous (the reference's transposed qmm layout, logical
 *            [K,N] strides (1,K), quantizer.py:239-244, is the same bytes); packed sub-byte
 *            formats hold the flattened [N][K] element order in the reference's group codecs
 *            (packed_int/pack.py) -- byte-identical to the reference state_dict.
 *   scale / zero_point   float32, one per (row n, group g): [N][G]   (G = K / group_size)
 *   svd_up   physical [N][R] (R contiguous)     svd_down   physical [R][K] (K contiguous)
 *   activations x [M][K] row-major with row stride ldx; outputs [M][N] row-major.
 */
#ifndef SDNQ_HIP_H
#define SDNQ_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SDNQ_HIP_ABI_VERSION 1

typedef void* sdnq_stream_t; /* hipStream_t */

typedef enum SdnqStatus {
    SDNQ_OK = 0,
    SDNQ_ERR_NULL = -1,        /* required pointer is NULL */
    SDNQ_ERR_DTYPE = -2,       /* unknown / unsupported dtype code */
    SDNQ_ERR_SHAPE = -3,       /* bad M/N/K/group (reference needs N,K >= 32 and %16 == 0 for qmm, utils.py:93-98) */
    SDNQ_ERR_ALIGN = -4,       /* pointer or leading dimension not 16-byte aligned */
    SDNQ_ERR_UNSUPPORTED = -5, /* valid in the reference but not built yet */
    SDNQ_ERR_ARCH = -6,        /* device is not gfx950 */
    SDNQ_ERR_LAUNCH = -7,      /* hipLaunch failed (see hipGetLastError) */
    SDNQ_ERR_WORKSPACE = -8    /* workspace too small */
} SdnqStatus;

/* element types of activations / outputs / bias / svd factors */
typedef enum SdnqFloat { SDNQ_F32 = 0, SDNQ_BF16 = 1, SDNQ_F16 = 2 } SdnqFloat;

/* matmul operand types (dtype_dict rows "int8", "float8_e4m3fn"; common.py:20,65) */
typedef enum SdnqMM { SDNQ_MM_I8 = 0, SDNQ_MM_FP8 = 1, SDNQ_MM_F16 = 2 /* float16 operands: sdnq_hip_unpack_mm, sdnq_hip_rowquant_f16, sdnq_hip_scaled_mm_f16 only */ } SdnqMM;

/* how the quantized weight elements are held in HBM */
typedef enum SdnqStorage {
    SDNQ_ST_PACKED_U8 = 0,  /* (u)int1..7, float2..7: reference group codecs in uint8 words (packed_int/unpack.py:233-372) */
    SDNQ_ST_PACKED_I16 = 1, /* (u)int9..15, float9..15: group codecs in int16 words (packed_int/unpack.py:7-229) */
    SDNQ_ST_RAW8 = 2,       /* int8 / uint8 / custom float8 codes / native float8_e4m3fn / float8_e5m2 */
    SDNQ_ST_RAW16 = 3       /* int16 / uint16 / custom float16 codes / native float16 */
} SdnqStorage;

/* numeric class of the stored code (dtype_dict is_integer / is_unsigned; common.py:16-267) */
typedef enum SdnqKind {
    SDNQ_KIND_INT = 0,   /* signed int: stored as value - min (packed) or two's complement (raw) */
    SDNQ_KIND_UINT = 1,  /* unsigned int, asymmetric (zero_point required) */
    SDNQ_KIND_FLOAT = 2, /* signed eXmY "fn" float code (packed_float.py:86-132) */
    SDNQ_KIND_UFLOAT = 3 /* unsigned eXmY "fnu" float code, asymmetric (zero_point required) */
} SdnqKind;

/* A quantized Linear weight exactly as the reference state_dict holds it. */
typedef struct SdnqWeight {
    const void* weight;      /* packed / raw codes, see SdnqStorage */
    const float* scale;      /* [N][G]; always float32 in memory, see scale_dtype */
    const float* zero_point; /* [N][G] or NULL */
    const void* svd_up;      /* [N][R] or NULL */
    const void* svd_down;    /* [R][K] or NULL */
    int32_t n;               /* output channels N */
    int32_t k;               /* input channels K */
    int32_t group_size;      /* elements per scale group along K; == k for row-wise */
    int32_t svd_rank;        /* R (0 if no SVD) */
    int32_t svd_dtype;       /* SdnqFloat of svd_up / svd_down */
    int32_t storage;         /* SdnqStorage */
    int32_t kind;            /* SdnqKind */
    int32_t bits;            /* 1..16 */
    int32_t exponent;        /* float kinds: exponent bits */
    int32_t mantissa;        /* float kinds: mantissa bits */
    int32_t native_float;    /* 1: codes are IEEE/OCP native (float8_e4m3fn, float8_e5m2, float16) */
    int32_t positions;       /* 0 / 1: Linear.  P > 1: conv weight [N][C_in][P] quantized along C_in (quantizer.py:120-123,
                                205-209): k = C_in * P, group_size counts CHANNELS, scale / zero_point are
                                [N][C_in / group_size][P] (one per output channel, channel group and kernel position) */
    int32_t scale_dtype;     /* SdnqFloat the layer STORES scale / zero_point in: SDNQ_F32 (dequantize_fp32=True, the default)
                                or the model dtype (dequantize_fp32=False, quantizer.py:147-156).  The arrays handed over are
                                the float32 upcast (exact) either way; with a 16-bit scale_dtype the kernels reproduce the
                                reference's arithmetic on 16-bit tensors: w * scale (+ zero_point) is rounded to scale_dtype
                                once (dequantizer.py:27, 63) before anything else uses it, and the re-quantizer's row scale
                                and quotient are rounded to it as well (dequantizer.py:219-239 with dtype=scale.dtype) */
} SdnqWeight;

/* ---- library ---------------------------------------------------------------------------- */
int sdnq_hip_version(void);
const char* sdnq_hip_strerror(int status);
/* 1 if device `ordinal` is gfx950 (arch gate; the reference parses gcnArchName the same way, sdnext.py:101-105) */
int sdnq_hip_device_supported(int ordinal);

/* ---- a8/a9: row-wise activation quantization ----------------------------------------------
 * replaces quantize_int_mm_input (layers/linear/linear_int8.py:15-22 -> quant_utils.py:265-273)
 * and quantize_fp_mm_input (layers/linear/linear_fp8.py:15-22 -> quant_utils.py:290-299), with the
 * optional Hadamard rotation of the activation fused in front (linear_int8.py:55-56,
 * quant_utils.py:194-209).  xs[m] = amax_k|x| / qmax ; xq = cast(clamp(round_half_even(x / xs))).
 * x: [M][K] of x_dtype, row stride ldx elements. xq: [M][K] int8 or fp8-e4m3fn bytes. xs: [M] f32.
 * rowsum: optional [M] int32 = sum_k xq (zero-point bias, linear_int8.py:65-69); NULL to skip.
 * xrot: optional [M][K] of x_dtype receiving the rotated activation (needed by the SVD branch).  For rotated rows longer
 * than 5120 elements it doubles as the kernel's parking space (rotate once, quantize from the copy): pass it whenever
 * K > 5120 and hadamard_group != 0 -- without it the rotation is simply computed twice.
 * hadamard_group: 0 = no rotation, else power of two in [4, 512] dividing K.
 * prefetch / prefetch_bytes: optional software prefetch (may be NULL / 0): extra workgroups of the same launch read
 * this range (the weight operand of the matmul that follows) so that it is resident in the last-level cache when
 * the GEMM starts; pure hint, no effect on results.
 * xzp: NULL for the symmetric quantization above; non-NULL ([M] f32) selects the ASYMMETRIC int8 quantization of the
 * uint8 matmul (quantize_uint_mm_input, layers/linear/linear_uint8.py:15-23 -> quant_utils.py:277-286):
 * xs = (max - min) / 255, xzp = min + 128 * xs, xq = clamp(round_half_even((x - xzp) / xs), -128, 127). */
int sdnq_hip_rowquant(const void* x, int x_dtype, int64_t m, int64_t k, int64_t ldx, int mm_dtype,
                      int hadamard_group, void* xq, float* xs, int32_t* rowsum, void* xrot,
                      const void* prefetch, int64_t prefetch_bytes, float* xzp, sdnq_stream_t stream);

/* dequantize_fp32=False form of sdnq_hip_rowquant: the layer's scale is stored in the model dtype (quantizer.py:147-156) and the
 * reference then quantizes the activation IN that dtype (`input.to(dtype=scale.dtype)`, linear_int8.py:15-22 / linear_fp8.py:13-20;
 * torch ops on 16-bit tensors compute in fp32 and round each result once):
 *     xs[m] = round_T(amax_k|x| / qmax),   xq = cast(clamp(round_half_even(round_T(x / xs))))      (fp8: nan_to_num, clamp, cast)
 * x: [M][K] of x_dtype = T (SDNQ_BF16 or SDNQ_F16 only).  xs receives the T-representable scale as float32 -- for float16 that IS the
 * reference's promotion "fp16 will overflow" (linear_int8.py:20-21), for bfloat16 the value the bf16 epilogue of
 * sdnq_hip_scaled_mm_lp reads.  rowsum / xrot / hadamard_group as in sdnq_hip_rowquant; symmetric quantization only. */
int sdnq_hip_rowquant_lp(const void* x, int x_dtype, int64_t m, int64_t k, int64_t ldx, int mm_dtype, int hadamard_group,
                         void* xq, float* xs, int32_t* rowsum, void* xrot, sdnq_stream_t stream);

/* ---- a15/a16: scaled matmul (the operator seam) --------------------------------------------
 * replaces int_scaled_mm_func / fp8_scaled_mm_func (kernel_wrappers.py:193-204) and the Triton op
 * sdnq::scaled_mm (kernels/triton_scaled_mm.py:127-275):
 *     out[m][n] = cast( fma( f32(sum_k a[m][k]*b[n][k]) * sa[m], sb[n], bias ) )     (bias present)
 *     out[m][n] = cast( (f32(acc) * sa[m]) * sb[n] )                                 (no bias)
 * a: [M][K] int8/fp8, b: physical [N][K] int8/fp8 (the reference's b[K,N] strides (1,K)).
 * bias_ndim 0 (none), 1 ([N]) or 2 ([M][N], row stride ld_bias). int8: int32 accumulate (exact);
 * fp8: fp32 accumulate. K % 16 == 0. */
int sdnq_hip_scaled_mm(int mm_dtype, const void* a, const void* b, const float* sa, const float* sb,
                       const void* bias, int bias_dtype, int bias_ndim, int64_t ld_bias, void* out,
                       int out_dtype, int64_t m, int64_t n, int64_t k, sdnq_stream_t stream);

/* Which tile sdnq_hip_scaled_mm would run this problem on (a dry run of the launcher's shape rules -- the counterpart of the config
 * pruning by shape in kernels/triton_scaled_mm.py:36-58; nothing is launched, no device is touched): bm x bn outputs per workgroup of
 * `threads` threads, `workgroups` of them.  Any output pointer may be null.  Used by bench.py's per-shape roofline table (CUs a launch
 * occupies) and by tuning tools. */
int sdnq_hip_scaled_mm_tile(int mm_dtype, int out_dtype, int has_bias, int64_t m, int64_t n, int64_t k, int* bm, int* bn, int* threads,
                            int64_t* workgroups);

/* ---- the float16 quantized matmul (quantized_matmul_dtype = "float16"; round 6) ------------------------------------------------
 * replaces quantize_fp_mm_input(..., matmul_dtype="float16") (layers/linear/linear_fp8.py:15-22 -> quant_utils.py:290-299, called
 * from linear_fp16.py:46) and fp_scaled_mm_func (kernel_wrappers.py:207-211; Triton kernels/triton_scaled_mm.py with float16 operands):
 * sdnq_hip_rowquant_f16: per row, in float32, xs[m] = amax|x| / 65504, xq[m][k] = float16(clamp(nan_to_num(x / xs[m]), +-65504)).
 * sdnq_hip_scaled_mm_f16: out[m][n] = cast(fma(f32(sum_k a[m][k] * b[n][k]) * sa[m], sb[n], bias[n])), a [M][K], b [N][K] float16,
 * fp32 accumulation on the f16 matrix cores (the reference's CPU route instead pre-scales both operands by 1 / sqrt(65536 K) and rounds
 * them to float16 again, kernel_wrappers.py:115-129: results agree to the float16 rounding of the operands).  K % 8 == 0, N % 8 == 0.
 * bias: NULL (bias_ndim 0), [N] (1) or [M][ld_bias] (2: the low-rank term of a layer with SVD factors, addmm(bias, x . svd_down, svd_up) of
 * linear_fp16.py:38-43, computed by the caller in the factors' dtype), of bias_dtype. */
int sdnq_hip_rowquant_f16(const void* x, int x_dtype, int64_t m, int64_t k, int64_t ldx, void* xq, float* xs, sdnq_stream_t stream);
int sdnq_hip_scaled_mm_f16(const void* a, const void* b, const float* sa, const float* sb, const void* bias, int bias_dtype, int bias_ndim,
                           int64_t ld_bias, void* out, int out_dtype, int64_t m, int64_t n, int64_t k, sdnq_stream_t stream);

/* ---- N1: the re-quantization of 4-bit weights fused into the matmul (round 6) -----------------------------------------------
 * replaces, as ONE launch per call, what sdnq_hip_requant + sdnq_hip_scaled_mm compute for 4-bit packed weights
 * (dequantizer.py:166-239 re_quantize_matmul + layers/linear/linear_int8.py:104-107): the stored codes are the weight operand, the
 * re-quantized int8 value of a code comes from a 16-entry table per (row, 64 columns).
 *
 * sdnq_hip_lut4_build: the tables of a weight -- lut[n][K / 64][16] bytes (entry c = the int8 byte sdnq_hip_requant writes for code c
 * in that block of that row; 16-byte aligned) -- and the row scales ws[N] (ws_known != 0: read from ws instead of derived).  Built
 * once per layer (they depend only on static parameters).  4-bit packed storage, positions == 1, group_size % 64 == 0, K % 64 == 0,
 * K <= 16384; anything else: SDNQ_ERR_UNSUPPORTED.  mm_dtype: SDNQ_MM_I8 or SDNQ_MM_FP8 (the tables then hold e4m3 bytes).
 *
 * sdnq_hip_scaled_mm_w4: out[m][n] = cast(fma(f32(sum_k a[m][k] * table(codes[n][k])) * sa[m], sb[n], bias[n])) -- bit for bit
 * sdnq_hip_scaled_mm on the operand sdnq_hip_requant would have written.  a: [M][K] int8 (row stride lda bytes, 0: K); codes: the
 * stored packed tensor [N][K / 2]; int8 matmul, 16-bit outputs, 1-D bias or none, K % 128 == 0.  ..._supported: 1 where this route is
 * built AND expected to win (few-row problems: every row block of 64 rows repeats the expansion). */
int sdnq_hip_lut4_build(const SdnqWeight* w, int mm_dtype, float* ws, int ws_known, void* lut, sdnq_stream_t stream);
int sdnq_hip_scaled_mm_w4(const void* a, const void* codes, const void* lut, const float* sa, const float* sb, const void* bias, int bias_dtype,
                          void* out, int out_dtype, int64_t m, int64_t n, int64_t k, int64_t lda, sdnq_stream_t stream);
int sdnq_hip_scaled_mm_w4_supported(int mm_dtype, int out_dtype, int64_t m, int64_t n, int64_t k);

/* dequantize_fp32=False with BFLOAT16 scales: int_scaled_mm_torch / fp8_scaled_mm_torch on bf16 tensors (kernel_wrappers.py:132-144),
 *     t = bf16(acc);  t = bf16(t * sa[m]);  out = bf16(t * sb[n])   or   bf16(fma(t, sb[n], bias))     (fp32 op-math per step)
 * sa / sb: float32 arrays holding bf16-representable values (sdnq_hip_rowquant_lp's xs; the layer's upcast scale).  bias: NULL,
 * [N] (bias_ndim 1) or [M][ld_bias] (bias_ndim 2), bfloat16.  t / svd_up (both or neither, bf16, [M][rank] / [N][rank]): the
 * low-rank bias  bf16(f32(bias[n]) + sum_r t[m][r] * svd_up[n][r])  of the SVD layers (linear_int8.py:57-62), bias_ndim <= 1 then.
 * out: [M][N] bfloat16.  (float16 scales need no such form: the activation scale is promoted to float32, which makes the whole
 * epilogue the float32 one of sdnq_hip_scaled_mm.) */
int sdnq_hip_scaled_mm_lp(int mm_dtype, const void* a, const void* b, const float* sa, const float* sb, const void* bias,
                          int bias_ndim, int64_t ld_bias, const void* t, const void* svd_up, int rank, void* out,
                          int64_t m, int64_t n, int64_t k, sdnq_stream_t stream);

/* sdnq_hip_scaled_mm_lp plus the zero-point term of unsigned weights on bfloat16 tensors (linear_int8.py:65-69 with dequantize_fp32=False):
 *     zero_bias = bf16(bf16(bf16(f32(rowsum[m])) * sa[m]) * zp[n]);  bias' = bf16(zero_bias + bias)  (bias: the [N] bias or the low-rank bias)
 * zp_rowsum [M] int32 (sdnq_hip_rowquant_lp's rowsum) and zp [N] f32 holding bf16-representable values: both or neither; bias_ndim <= 1. */
int sdnq_hip_scaled_mm_lp_zp(int mm_dtype, const void* a, const void* b, const float* sa, const float* sb, const void* bias,
                             int bias_ndim, int64_t ld_bias, const void* t, const void* svd_up, int rank, const int32_t* zp_rowsum,
                             const float* zp, void* out, int64_t m, int64_t n, int64_t k, sdnq_stream_t stream);

/* The uint8 (asymmetric-activation) matmul on BFLOAT16 scales -- linear_uint8.py:15-23, 57-102 with dequantize_fp32=False; every torch op
 * of the chain rounds its float32 result to bfloat16 once.
 * sdnq_hip_rowquant_lp_asym <- quantize_uint_mm_input(input, dtype=scale.dtype) (linear_uint8.py:15-23, quant_utils.py:10-19, 277-286):
 *     xs = round_T(round_T(max - min) / 255);  xzp = round_T(min + 128 xs);  xq = int8(clamp(rint(round_T(round_T(x - xzp) / xs))))
 *   x [M][K] of x_dtype = T (bf16 / f16); xs / xzp [M] float32 holding T-representable values; rowsum / xrot / hadamard_group as in
 *   sdnq_hip_rowquant.
 * sdnq_hip_scaled_mm_lp_uzp <- get_uint8_matmul_inputs' zero_bias (:61-68) + int_scaled_mm_torch (kernel_wrappers.py:132-144) on bf16 tensors:
 *     t1 = bf16(bf16(bf16(f32(rowsum[m])) * sa[m]) * zp[n])                  (zp_rowsum / zp: both or neither -- weights with a zero point)
 *     t2 = bf16(w_colsum_scaled[n] * a_zp[m]),  w_colsum_scaled[n] = bf16(bf16(f32(sum_k b[n][k])) * sb[n]) precomputed by the caller
 *     zb = bf16(t1 + t2);  zb = bf16(fma(bf16(a_zp[m] * zp[n]), zp_k ? zp_k : K, zb))   (zp_k < 0: the conv order, see sdnq_hip_scaled_mm_zp)
 *     zb = bf16(zb + bias[n]);   out = bf16(fma(bf16(bf16(acc) * sa[m]), sb[n], zb))
 *   a / b int8 [M][K] / [N][K]; sa, sb, zp, a_zp, w_colsum_scaled float32 arrays holding bf16-representable values; bias NULL or [N] bf16;
 *   out [M][N] bf16. */
int sdnq_hip_rowquant_lp_asym(const void* x, int x_dtype, int64_t m, int64_t k, int64_t ldx, int hadamard_group, void* xq, float* xs,
                              float* xzp, int32_t* rowsum, void* xrot, sdnq_stream_t stream);
int sdnq_hip_scaled_mm_lp_uzp(const void* a, const void* b, const float* sa, const float* sb, const void* bias, const int32_t* zp_rowsum,
                              const float* zp, const float* a_zp, const float* w_colsum_scaled, int64_t zp_k, void* out, int64_t m,
                              int64_t n, int64_t k, sdnq_stream_t stream);
/* ... of a layer with SVD factors (linear_uint8.py:57-62 on bfloat16 tensors): t [M][R] = bf16(x . svd_down) (sdnq_hip_lowrank_down) and
 * svd_up [N][R], both bfloat16; the last bias step becomes zb = bf16(zb + bf16(f32(bias[n]) + sum_r t[m][r] svd_up[n][r])) -- the
 * addmm of :60, one rounding -- everything else as sdnq_hip_scaled_mm_lp_uzp. */
int sdnq_hip_scaled_mm_lp_uzp_svd(const void* a, const void* b, const float* sa, const float* sb, const void* bias,
                                  const int32_t* zp_rowsum, const float* zp, const float* a_zp, const float* w_colsum_scaled, int64_t zp_k,
                                  const void* t, const void* svd_up, int rank, void* out, int64_t m, int64_t n, int64_t k,
                                  sdnq_stream_t stream);

/* the same scaled matmul over the STACKED weights of layers that consume one activation (to_q / to_k / to_v of an attention block),
 * each layer's columns stored in its own contiguous tensor: b [n_outs * seg_n][K], sb / bias [n_outs * seg_n], outs[i] is
 * [M][seg_n] (seg_n % 8 == 0, n_outs <= 4, n == n_outs * seg_n).  One launch and one pass over the quantized activation instead
 * of n_outs; every output element is the value sdnq_hip_scaled_mm gives for its layer (int_scaled_mm_func per layer,
 * kernel_wrappers.py:193-204). */
int sdnq_hip_scaled_mm_multi(int mm_dtype, const void* a, const void* b, const float* sa, const float* sb, const void* bias,
                             int bias_dtype, void* const* outs, int n_outs, int64_t seg_n, int out_dtype, int64_t m, int64_t n,
                             int64_t k, sdnq_stream_t stream);

/* ---- a3/a4 fused: dequantize + float GEMM in one launch (the reference's DEFAULT mode, M > 32) -------------------------------
 * replaces SDNQDequantizer.__call__ -> dequantize_symmetric / dequantize_asymmetric (dequantizer.py:52-84, 20-48) followed by
 * torch.nn.functional.linear (layers/linear/forward.py:25-26) for ROW-WISE 8-bit weights (int8: zero_point NULL; uint8: zero_point
 * [N]): the weight stays 1 byte per element in HBM and LDS, every lane converts its weight row's codes to the activation dtype
 * between LDS and the matrix core -- W[n][k] = cast(f32(w) * scale[n]) resp. cast(fma(f32(w), scale[n], zero_point[n])), the very
 * value sdnq_hip_dequant produces -- and out = cast(x . W^T + bias) with fp32 accumulation.
 * x [M][K] bf16 / f16 (row stride ldx), w [N][K] int8 / uint8, scale / zero_point [N] f32, bias [N] of x_dtype or NULL,
 * out [M][N] of x_dtype.  K % 16 == 0, N % 8 == 0. */
int sdnq_hip_linear_w8a16(const void* x, int x_dtype, const void* w, const float* scale, const float* zero_point, const void* bias,
                          void* out, int64_t m, int64_t n, int64_t k, int64_t ldx, sdnq_stream_t stream);

/* the same fused dequantize + float GEMM for SEVERAL signed-int8 row-wise layers that consume one activation (linked attention
 * projections in the dequantize + F.linear mode), through the unit table of sdnq_hip_scaled_mm_grouped below (SdnqGemmUnit.b =
 * int8 weight rows, .sb = their scales, .bias of x_dtype): one launch, each layer's output its own [M][n_seg] matrix inside `out`. */
struct SdnqGemmUnit;
int sdnq_hip_linear_w8a16_grouped(const void* x, int x_dtype, const struct SdnqGemmUnit* units, int64_t n_units, int64_t unit_n,
                                  int has_bias, void* out, int64_t m, int64_t k, int64_t ldx, sdnq_stream_t stream);

/* ---- grouped scaled matmul: many layers that consume ONE activation, no stacked weight copy ------------------------------
 * The same arithmetic as sdnq_hip_scaled_mm, per layer (int_scaled_mm_func per layer, kernel_wrappers.py:193-204), for layers
 * whose inputs are the very same tensor: to_q / to_k / to_v of a self-attention block, or every cross-attention to_k / to_v of a
 * UNet step (all 140 of SDXL read the one encoder_hidden_states).  One launch, one pass over the quantized activation, and the
 * weights are read where the layers' own parameters live: the output channels of all layers are cut into UNITS of `unit_n`
 * channels (unit_n divides every layer's N; unit_n % 64 == 0) and `units` -- a DEVICE-resident table, one entry per unit, built
 * once per group -- says where each unit's weight rows, scales and bias are.
 * Output: ONE buffer `out` of M * (n_units * unit_n) elements; the layer that starts at channel n_start and has n_seg channels
 * owns the contiguous [M][n_seg] matrix at element offset M * n_start.
 * bias_dtype < 0: no layer has a bias (SdnqGemmUnit.bias ignored), else every layer has one of that dtype. */
typedef struct SdnqGemmUnit {
    const void* b;     /* weight rows [unit_n][K] of this unit (physical [N][K] layout, K contiguous) */
    const float* sb;   /* [unit_n] weight scales */
    const void* bias;  /* [unit_n] bias elements or NULL */
    int64_t n_start;   /* first output channel of the LAYER this unit belongs to, in the concatenated channel order */
    int32_t n_seg;     /* channels of that layer */
    int32_t n_loc;     /* first channel of this unit inside its layer */
} SdnqGemmUnit;
int sdnq_hip_scaled_mm_grouped(int mm_dtype, const void* a, const float* sa, const SdnqGemmUnit* units, int64_t n_units,
                               int64_t unit_n, int bias_dtype, void* out, int out_dtype, int64_t m, int64_t k,
                               sdnq_stream_t stream);

/* Tuning hook (development / benchmarking only, process-wide): force the GEMM tile configuration of every following scaled
 * matmul launch (ids: sdnq_amd/csrc/gemm.hip, launch_tiles); < 0 restores the built-in shape heuristics.  Never needed for
 * correct results -- every configuration produces identical outputs. */
void sdnq_hip_set_tile_override(int tile_id);

/* the float GEMM of sdnq_hip_linear_float over the STACKED dequantized weights of layers that consume one activation, each
 * layer's columns in its own contiguous tensor (the dequantize + F.linear mode of linked attention projections, layers/linear/
 * forward.py:25-26 per layer): wd [n_outs * seg_n][K], bias NULL or [n_outs * seg_n], outs[i] [M][seg_n]; M > 32. */
int sdnq_hip_linear_float_multi(const void* x, const void* wd, const void* bias, int dtype, void* const* outs, int n_outs,
                                int64_t seg_n, int64_t m, int64_t n, int64_t k, int64_t ldx, sdnq_stream_t stream);

/* ---- a4/a5/a6: dequantize ---------------------------------------------------------------------
 * replaces SDNQDequantizer.__call__ -> dequantize_weight (dequantizer.py:135-162, 389-429):
 * unpack -> f32(w)*scale | fma(f32(w),scale,zp) -> [+ svd_up@svd_down in svd dtype] -> cast ->
 * [Hadamard rotation in out dtype].  out: [N][K] row-major of out_dtype. */
int sdnq_hip_dequant(const SdnqWeight* w, int hadamard_group, void* out, int out_dtype, sdnq_stream_t stream);

/* ---- a7: re-quantize for matmul ---------------------------------------------------------------
 * replaces re_quantize_matmul (dequantizer.py:204-239): fp32 dequant (Hadamard NOT undone), then a
 * per-output-row symmetric quantization to the matmul dtype. wq: physical [N][K]; ws: [N] f32. */
int sdnq_hip_requant(const SdnqWeight* w, int mm_dtype, void* wq, float* ws, sdnq_stream_t stream);

/* sdnq_hip_requant with the option of KNOWN row scales: ws_known != 0 means ws[N] already holds the per-row scales (they depend
 * only on the static weights; a caller that re-quantizes on every forward -- the reference's behaviour, dequantizer.py:204-239 --
 * keeps these N floats and skips the pass that derives them).  Values are identical to sdnq_hip_requant.  4-bit packed weights in
 * groups of a multiple of 64 take a table path (16 possible bytes per (row, group): four exact divisions per lane instead of
 * sixteen); other formats recompute the scales and ignore ws_known. */
int sdnq_hip_requant_ws(const SdnqWeight* w, int mm_dtype, void* wq, float* ws, int ws_known, sdnq_stream_t stream);

/* asymmetric form for quantized_matmul_dtype "uint8": replaces re_quantize_uint_mm (dequantizer.py:178-187) ->
 * quantize_uint_mm (quant_utils.py:277-286): scale = (max - min) / 255, zero_point = min + 128 * scale per output row,
 * wq = int8 codes of (w - zero_point) / scale.  wq: physical [N][K] int8; ws, wzp: [N] f32. */
int sdnq_hip_requant_asym(const SdnqWeight* w, void* wq, float* ws, float* wzp, sdnq_stream_t stream);

/* ---- a12/a13 (weight half): matmul operand WITHOUT re-quantization ----------------------------
 * replaces the per-call unpack in get_int8_matmul_inputs / get_fp8_matmul_inputs
 * (layers/linear/linear_int8.py:38-50, linear_fp8.py:36-38) for row-wise weights whose codes already fit
 * the matmul dtype:
 *   int8 mm: packed signed -> value; packed unsigned -> raw code (caller keeps zero_point);
 *            raw uint8 -> code ^ 0x80 (caller adds 128*scale to the zero point); raw int8 -> copy
 *   fp8  mm: packed custom float -> e4m3fn(decoded value); native float8_e4m3fn -> copy
 *   f16  mm (round 6; linear_fp16.py:27-31): any float format of <= 16 bits -> the decoded value rounded to float16, wq [N][K] x 2 bytes
 * wq: physical [N][K] bytes. Scales are untouched (row-wise, already [N]). */
int sdnq_hip_unpack_mm(const SdnqWeight* w, int mm_dtype, void* wq, sdnq_stream_t stream);

/* ---- a11: Hadamard rotation -------------------------------------------------------------------
 * replaces rotate_hadamard (quant_utils.py:194-209): y.view(rows, K/g, g) @ H_g, rounded to dtype.
 * H_g = kron powers of the reference's H4 (g a power of 4) or Sylvester H2 (other powers of 2),
 * scaled g^-1/2 (quant_utils.py:145-175).  In-place allowed (y == x). */
int sdnq_hip_hadamard(const void* x, int dtype, int64_t rows, int64_t k, int64_t ldx, int hadamard_group,
                      void* y, int64_t ldy, sdnq_stream_t stream);

/* ---- a12: SVD low-rank prologue ---------------------------------------------------------------
 * t[M][R] = cast_svd_dtype( x[M][K] @ svd_down^T ), the inner torch.mm of
 * addmm(bias, mm(x, svd_down), svd_up) (linear_int8.py:57-62).  The outer product with svd_up and
 * the bias add are fused into sdnq_hip_scaled_mm_lowrank's epilogue. */
int sdnq_hip_lowrank_down(const void* x, int x_dtype, int64_t m, int64_t k, int64_t ldx, const void* svd_down,
                          int svd_dtype, int rank, void* t, sdnq_stream_t stream);

/* scaled matmul whose bias is  cast_svd( f32(bias[n]) + sum_r t[m][r]*svd_up[n][r] )  [+ zero-point terms]:
 *   zp_rowsum/zp (both or neither): adds f32(rowsum[m]) * sa[m] * zp[n]  (linear_int8.py:65-69);
 *   a_zp/w_colsum_scaled (both or neither; the uint8 matmul, linear_uint8.py:61-66): adds
 *       w_colsum_scaled[n] * a_zp[m]  +  K * (a_zp[m] * zp[n])      with w_colsum_scaled[n] = f32(sum_k b[n][k]) * sb[n]. */
int sdnq_hip_scaled_mm_lowrank(int mm_dtype, const void* a, const void* b, const float* sa, const float* sb,
                               const void* bias, int bias_dtype, const void* t, const void* svd_up, int svd_dtype,
                               int rank, const int32_t* zp_rowsum, const float* zp, const float* a_zp,
                               const float* w_colsum_scaled, void* out, int out_dtype,
                               int64_t m, int64_t n, int64_t k, sdnq_stream_t stream);

/* ---- a3: float path  F.linear(x, dequant(W), bias) --------------------------------------------
 * replaces quantized_linear_forward (layers/linear/forward.py:25-26) and the M<32 branch of the
 * quantized forwards (linear_int8.py:102-103): out = x @ Wd^T + bias with Wd already dequantized to
 * the activation dtype ([N][K]), fp32 accumulate, one rounding to the output dtype. */
int sdnq_hip_linear_float(const void* x, const void* wd, const void* bias, int dtype, void* out, int64_t m,
                          int64_t n, int64_t k, int64_t ldx, sdnq_stream_t stream);

/* fused skinny variant (M <= 64, meant for the M < 32 branch): streams the quantized weight ONCE, dequantizes in
 * registers to the activation dtype (same arithmetic and rounding as sdnq_hip_dequant) and accumulates in fp32.
 * No SVD (SDNQ_ERR_UNSUPPORTED if w->svd_up is set).  hadamard_group != 0: the stored weight is rotated; each rounded
 * 16-element run is un-rotated in registers and rounded again (dequantizer.py:82-87 order), so x is passed as is. */
int sdnq_hip_linear_skinny(const SdnqWeight* w, int hadamard_group, const void* x, const void* bias, int dtype, void* out,
                           int64_t m, int64_t ldx, sdnq_stream_t stream);

/* ---- 8(f) rank 1: load-time weight quantizer + packers ------------------------------------------
 * replaces quantize_weight (quant_utils.py:28-56: get_scale_symmetric :23-24, get_scale_asymmetric :10-19) followed by
 * pack_int (packed_int/__init__.py:77-80 + packed_int/pack.py) or pack_float (packed_float.py:27-82) for one float
 * weight src [N][K] (row stride ld_src elements, dtype src_dtype = SdnqFloat), grouped along K by w->group_size.
 * `w` describes the OUTPUT: w->weight (codes in w->storage / kind / bits / exponent / mantissa, element order [N][K] --
 * the same bytes the reference stores, including its transposed [K,N]-strides-(1,K) matmul layout), w->scale [N][G] f32,
 * w->zero_point [N][G] f32 (unsigned kinds; NULL otherwise).  svd_* fields are ignored (the caller subtracts the
 * low-rank term first, as apply_svdquant does, quant_utils.py:124-141).  qmin / qmax = dtype_dict[...]["min"/"max"]
 * (common.py:16-267).  Symmetric: s = amax|w| / qmax, q = w / s.  Asymmetric: s = (max - min) / (qmax - qmin),
 * zp = min, q = (w - zp) / s.  Integers: round-half-even, clamp, stored as value - qmin when packed.  Floats:
 * nan_to_num, clamp, then the native conversion (fp8 e4m3fn / e5m2, fp16, bf16) or the reference's eXmY encoder. */
int sdnq_hip_quantize_weight(const void* src, int src_dtype, int64_t ld_src, const SdnqWeight* w, float qmin,
                             float qmax, sdnq_stream_t stream);

/* ---- weight prefetch (round 5) -------------------------------------------------------------------
 * Pulls [ptr, ptr + bytes) into the memory-side cache (256-MiB Infinity Cache of MI355X): one 4-byte read per 128-byte line, nothing
 * stored.  Meant for the static weights of the layers that run NEXT, launched on a side stream while the current layer computes -- the
 * reference has no counterpart (its GEMMs read their weights cold, like any launch-per-layer pipeline); a wrong guess costs bandwidth,
 * never correctness.  workgroups <= 0: 32. */
int sdnq_hip_prefetch(const void* ptr, int64_t bytes, int workgroups, sdnq_stream_t stream);
/* The same work WITHOUT a launch of its own: the next scaled-matmul launch of the calling thread (any sdnq_hip_scaled_mm* / sdnq_hip_linear*
 * entry point) appends workgroups that pull up to four ranges -- the weights of the launches that run next and after next -- into the
 * memory-side cache while its tiles compute, when that launch leaves workgroup slots free (a launch whose tiles fill every slot drops
 * the hint: that is why more than one launch ahead is named).  NULL / 0 for an unused range.  Thread-local; consumed by ONE launch; the
 * ranges must stay mapped until that launch has run (stream order).  SDNQ_HIP_PREFETCH_WGS caps the workgroups (default 96). */
int sdnq_hip_prefetch_hint(const void* p0, int64_t b0, const void* p1, int64_t b1, const void* p2, int64_t b2, const void* p3, int64_t b3);

/* ---- 8(e): tensor-parallel glue ------------------------------------------------------------------
 * Re-assembly of a column-sharded Linear's output after the RCCL all-gather (the reference has no inference parallelism, SURVEY 2.1;
 * north_star: "large Linear layers are optionally column-sharded across the 8 GPUs of one node with RCCL all-gather over xGMI").
 * gathered: [world][m_rows][wmax] -- rank r's slab y_r (its N / world output channels of rows [m0, m0 + m_rows)), padded to the
 * widest slab; out: row-major [m][N] with N = starts[world]; out[m0 + i][starts[r] + c] = gathered[r][i][c] for c < starts[r+1] -
 * starts[r].  starts: HOST array of world + 1 channel offsets (multiples of 16 / elem_bytes... of 8 elements at least; the
 * reference's N % 16 rule gives multiples of 16).  One HBM-bound pass; m0 / m_rows let a pipelined caller un-shard one M chunk
 * while the next chunk's gather is in flight. */
#define SDNQ_MAX_TP_RANKS 64
int sdnq_hip_unshard_columns(const void* gathered, void* out, int elem_bytes, int64_t m0, int64_t m_rows, int64_t m,
                             int64_t wmax, int world, const int64_t* starts, sdnq_stream_t stream);

/* Copy-free gather of a column-sharded Linear over PEER-MAPPED memory (round 4; SURVEY 8e: "have the GEMM write directly into a
 * symmetric [M,N] buffer").  Every rank of the node owns an arena that all ranks have mapped through hipIpc handles (exchanged once, by
 * the host side: sdnq_amd/parallel.py); a rank's output matrix [M][ldc] of gather number `seq` lives at a byte offset of ITS arena
 * that it announces with sdnq_hip_push_post (a 64-bit word -- seq in the top 24 bits, offset / 256 below -- stored into slot `rank` of
 * every rank's `post` array).  sdnq_hip_push_columns then copies this rank's slab y [rows][w] (row stride ldy elements) into columns
 * [col0, col0 + w), rows [row0, row0 + rows) of EVERY rank's matrix (local stores for its own, P2P stores over xGMI for the peers'),
 * stores `seq` into slot `rank` of every rank's `done` array and returns (stream-ordered) only when slot r of its OWN done array
 * carries `seq` for every r: this rank's matrix is then complete.  No staging buffer, no collective, no re-assembly pass.
 * arena / post / done: HOST arrays of `world` device pointers as mapped in THIS process (entry `rank` = the local one); post / done
 * are u64 [world] arrays IN SIGNAL MEMORY (below); ticket: device u32, zero; status: i32 in device or host-coherent memory, set to 1 when a rendezvous spin exceeded timeout_ms (the host
 * side raises instead of hanging the GPU).  Not capturable into a hipGraph together with its peers' launches in a fixed order only if
 * the arena offsets are the same at replay -- the host side runs it eagerly. */
#define SDNQ_MAX_PUSH_RANKS 16
/* Signal memory for the words above (round 5).  post[] / done[] are written by REMOTE kernels over xGMI while a local kernel spins on
 * them: only fine-grained / uncached allocations guarantee that such a write becomes visible inside a running kernel, so the control
 * words must not live in ordinary device memory (the bulk arena may: it is read only after the kernel that waited for `done`).
 *   sdnq_hip_signal_alloc(bytes, kind, &ptr, &granted): kind 0 = device memory, hipExtMallocWithFlags(hipDeviceMallocUncached), else
 *     hipDeviceMallocFinegrained (granted = SDNQ_SIGNAL_UNCACHED / _FINEGRAINED), zero-filled; kind 1 = pinned, mapped, coherent HOST
 *     memory (granted = SDNQ_SIGNAL_HOST_COHERENT) whose address is valid on the device too -- the `status` word, polled by the host
 *     without a synchronization.   sdnq_hip_signal_free(ptr, kind).
 *   sdnq_hip_ipc_export(ptr, handle64) / sdnq_hip_ipc_import(handle64, &ptr) / sdnq_hip_ipc_close(ptr): hipIpc handle of a kind-0
 *     block as 64 opaque bytes (the ranks exchange them through any host channel) and the peer's mapping of it. */
#define SDNQ_SIGNAL_UNCACHED 1
#define SDNQ_SIGNAL_FINEGRAINED 2
#define SDNQ_SIGNAL_HOST_COHERENT 3
int sdnq_hip_signal_alloc(int64_t bytes, int kind, void** ptr, int* granted);
int sdnq_hip_signal_free(void* ptr, int kind);
int sdnq_hip_ipc_export(const void* ptr, void* handle64);
int sdnq_hip_ipc_import(const void* handle64, void** ptr);
int sdnq_hip_ipc_close(void* ptr);
int sdnq_hip_push_post(void* const* post, int world, int rank, uint64_t seq, uint64_t arena_offset, sdnq_stream_t stream);
int sdnq_hip_push_columns(const void* y, int elem_bytes, int64_t rows, int64_t w, int64_t ldy, void* const* arena, void* const* post,
                          void* const* done, int world, int rank, uint64_t seq, int64_t ldc, int64_t col0, int64_t row0, void* ticket,
                          void* status, int timeout_ms, sdnq_stream_t stream);

/* ---- 8(f) rank 3: convolution as GEMM -----------------------------------------------------------
 * replaces the F.unfold(...).transpose(1, 2) of process_conv_input (layers/conv/forward.py:30-76) for Conv1d (height = 1,
 * kh = 1) and Conv2d inputs x [batch][channels][height][width] of `dtype`: out [M][K] with rows m = (b, h_out, w_out),
 * columns k = (c, i, j), zero padding; K * sizeof(dtype) must be a multiple of 16.  The result feeds sdnq_hip_rowquant /
 * sdnq_hip_scaled_mm (conv_int8_matmul, layers/conv/conv_int8.py:18-91) or sdnq_hip_linear_float exactly like a Linear
 * activation; the [M][C_out] product is the NHWC image of the convolution. */
int sdnq_hip_im2col(const void* x, int dtype, int batch, int channels, int height, int width, int kh, int kw, int stride_h,
                    int stride_w, int pad_h, int pad_w, int dil_h, int dil_w, void* out, sdnq_stream_t stream);

/* sdnq_hip_rowquant (no rowsum / rotated copy / prefetch / asymmetric mode) followed by sdnq_hip_scaled_mm (bias: NULL or [N])
 * in one call: the plain w8a8 Linear of int8_matmul / fp8_matmul (linear_int8.py:75-97, linear_fp8.py:58-78).  Two launches on
 * `stream`; xq [M][K] and xs [M] are outputs that stay valid (sibling layers reuse them).  hadamard_group as in sdnq_hip_rowquant
 * (rows longer than 5120 elements are then rotated twice, see there). */
int sdnq_hip_linear_w8a8(int mm_dtype, const void* x, int x_dtype, int64_t m, int64_t k, int64_t ldx, int hadamard_group,
                         void* xq, float* xs, const void* b, const float* sb, const void* bias, int bias_dtype, void* out,
                         int out_dtype, int64_t n, sdnq_stream_t stream);

/* *id = the id of the stream capture `stream` is part of, 0 when it is not capturing (nothing in the reference: its Triton launches keep no
 * per-stream device state).  Used by hosts that keep such state: a captured launch must address a buffer of ITS capture. */
int sdnq_hip_stream_capture_id(sdnq_stream_t stream, unsigned long long* id);

/* The plain w8a8 Linear (int8_matmul / fp8_matmul: linear_int8.py:15-22, 64, 75-97 -> kernels/triton_scaled_mm.py:194-232; linear_fp8.py
 * the same) as ONE launch: every GEMM workgroup row-quantizes its own 64 activation rows into LDS (amax, scale = amax / qmax, codes with the
 * arithmetic of sdnq_hip_rowquant) and streams only the weight operand -- no quantized copy of the activation exists in HBM.  Results are
 * bit-identical to sdnq_hip_linear_w8a8 (csrc/gemm_aq.hip).  x [M][K] bf16 / f16 with row stride ldx, b [N][K] codes, sb [N], bias NULL or
 * [N] of bias_dtype, out [M][N] of x's dtype.  Built for K % 128 == 0, K <= 1280 (the rows stay resident in LDS); other shapes return
 * SDNQ_ERR_UNSUPPORTED -- callers ask sdnq_hip_linear_w8a8_fused_supported first: 1 where this route is built AND expected to win (one
 * round of workgroups, few column tiles per row block: the projections of a bs = 1 step; SDNQ_HIP_FUSED_ROWQUANT=0 turns it off). */
int sdnq_hip_linear_w8a8_fused(int mm_dtype, const void* x, int x_dtype, int64_t m, int64_t k, int64_t ldx, const void* b, const float* sb,
                               const void* bias, int bias_dtype, void* out, int out_dtype, int64_t n, sdnq_stream_t stream);
int sdnq_hip_linear_w8a8_fused_supported(int mm_dtype, int x_dtype, int out_dtype, int64_t m, int64_t n, int64_t k);

/* ---- the whole quantized-matmul forward of one layer behind ONE call -------------------------------------------------
 * int8_matmul / fp8_matmul / uint8_matmul (layers/linear/linear_int8.py:75-97, linear_fp8.py:58-78, linear_uint8.py:80-102) for every
 * layer form: row quantization of the activation (Hadamard-rotated when the layer is, asymmetric for the uint8 matmul), the low-rank
 * (SVD) product t = x . svd_down^T, and the scaled matmul whose epilogue adds bias, low-rank and zero-point terms -- the sequence
 * sdnq_hip_rowquant -> [sdnq_hip_lowrank_down] -> sdnq_hip_scaled_mm | sdnq_hip_scaled_mm_lowrank, with identical results.
 * A POD: plain pointers and sizes, no ownership.  Set struct_size = sizeof(SdnqLinearArgs) (the library refuses a struct it does not
 * know).  The weight side is the matmul operand as the kernels take it: wq [N][K] int8 / fp8 codes and ws [N] row scales -- the
 * stored tensors of a row-wise int8 / fp8 layer, or the output of sdnq_hip_requant / sdnq_hip_unpack_mm for group-wise / packed ones.
 * Intermediates: xq [M][K] bytes, xs [M] f32, rowsum [M] i32 (with zp), xrot [M][K] of x_dtype (SVD on a Hadamard layer), xzp [M] f32
 * (asymmetric) and t [M][svd_rank] of svd_dtype.  Each of xq / xs / rowsum / xrot / xzp may be supplied by the caller (to keep them,
 * e.g. for sibling layers that consume the same activation) or left NULL, in which case it lives in `workspace`; with x_prequantized
 * the supplied ones are INPUTS and the row quantization is skipped.  sdnq_hip_linear_workspace_bytes tells how much workspace the
 * call needs for the pointers left NULL (256-byte aligned device memory, used by one stream at a time). */
typedef struct SdnqLinearArgs {
    int32_t struct_size;
    int32_t mm_dtype;        /* SDNQ_MM_I8 | SDNQ_MM_FP8 */
    int32_t x_dtype;         /* SdnqFloat of x (and of xrot) */
    int32_t out_dtype;       /* SdnqFloat of out */
    int32_t bias_dtype;      /* SdnqFloat of bias (ignored without one) */
    int32_t svd_dtype;       /* SdnqFloat of svd_down / svd_up / t */
    int32_t hadamard_group;  /* 0: the layer is not rotated */
    int32_t svd_rank;        /* 0: no low-rank term */
    int32_t asymmetric;      /* 1: asymmetric activations (the uint8 matmul); int8 matmul dtype only */
    int32_t x_prequantized;  /* 1: xq / xs (and rowsum / xrot / xzp where the layer needs them) are inputs */
    int64_t m, n, k, ldx;    /* x is [M][ldx] with K valid columns */
    const void* x;
    void* out;               /* [M][N] */
    const void* wq;          /* [N][K] */
    const float* ws;         /* [N] */
    const void* bias;        /* [N] or NULL */
    const void* svd_down;    /* physical [R][K] or NULL */
    const void* svd_up;      /* physical [N][R] or NULL */
    const float* zp;         /* [N] weight zero-point term (unsigned weights, linear_int8.py:65-69) or NULL */
    const float* w_colsum_scaled; /* [N] f32(sum_k wq[n][k]) * ws[n]: required with asymmetric */
    void* xq;
    float* xs;
    int32_t* rowsum;
    void* xrot;
    float* xzp;
    void* workspace;
    int64_t workspace_bytes;
} SdnqLinearArgs;
int sdnq_hip_linear(const SdnqLinearArgs* args, sdnq_stream_t stream);
int sdnq_hip_linear_workspace_bytes(const SdnqLinearArgs* args, int64_t* bytes);

/* the scaled matmul of the conv forwards with the channel-major store fused into the epilogue: out is the conv output
 * [B][N][hw] (NCHW / NCL), rows m = b * hw + pixel -- replaces int_scaled_mm_func(...).view(mm_output_shape) followed by
 * .permute(0, 3, 1, 2).contiguous() (conv_int8.py:71, 81-88).  bias: NULL or [N] of bias_dtype; hw % 8 == 0, m % hw == 0,
 * out_dtype bf16 / f16. */
int sdnq_hip_scaled_mm_nchw(int mm_dtype, const void* a, const void* b, const float* sa, const float* sb, const void* bias,
                            int bias_dtype, void* out, int out_dtype, int64_t m, int64_t n, int64_t k, int64_t hw,
                            sdnq_stream_t stream);

/* sdnq_hip_scaled_mm / sdnq_hip_scaled_mm_nchw on VIEWS, for the per-group matmuls of grouped convs (conv_int8.py:73-79, conv_fp8.py:56-60:
 * `int_mm_func(input[:, i], weight[:, i])` per group on column slices of the quantized unfolded input, results concatenated along the
 * channels).  a: [M][lda] with K valid columns (lda % 16 == 0); out: hw == 0 -> [M][ldc] with the N results in its first N columns
 * (pass out + first channel); hw > 0 -> the image [B][ldc channels][hw], `out` pointing at the group's first channel plane.  bias NULL or [N]. */
int sdnq_hip_scaled_mm_strided(int mm_dtype, const void* a, int64_t lda, const void* b, const float* sa, const float* sb,
                               const void* bias, int bias_dtype, void* out, int64_t ldc, int out_dtype, int64_t m, int64_t n,
                               int64_t k, int64_t hw, sdnq_stream_t stream);

/* sdnq_hip_scaled_mm_lowrank's zero-point terms on VIEWS: one group of a grouped conv whose weights are unsigned (zero-point term
 * f32(rowsum) * sa * zp[n], conv_int8.py:65-69 -- rowsum taken over the WHOLE unfolded row, as the reference does) or whose matmul is
 * the uint8 one (activation zero point: + colsum(w) * ws * xzp + K * (xzp * wzp), conv_uint8.py:58-66 -- K again the whole row:
 * zp_k).  zp_k > 0: linear form, fma(xzp * wzp, K, .) as torch's add_(., alpha=K) computes it (linear_uint8.py:66); zp_k < 0: K = -zp_k in
 * the conv forwards' order, (xzp * K) * wzp added without fusion (conv_uint8.py:66: input_zero_point.mul_(K) in place first); 0: this
 * launch's k, linear form.  a: [M][lda] with K valid columns; out: [M][ldc] with N valid columns; everything per channel (sb, bias, zp,
 * w_colsum_scaled) points at this group's first channel.  Also the entry point of the UNGROUPED uint8 conv matmul (lda = k, ldc = n). */
int sdnq_hip_scaled_mm_lowrank_strided(int mm_dtype, const void* a, int64_t lda, const void* b, const float* sa, const float* sb,
                                       const void* bias, int bias_dtype, const int32_t* zp_rowsum, const float* zp, const float* a_zp,
                                       const float* w_colsum_scaled, int64_t zp_k, void* out, int64_t ldc, int out_dtype, int64_t m,
                                       int64_t n, int64_t k, sdnq_stream_t stream);

/* sdnq_hip_linear_float with an output row stride (ldc >= n elements): the float matmul of one conv group on views. */
int sdnq_hip_linear_float_strided(const void* x, const void* wd, const void* bias, int dtype, void* out, int64_t m,
                                  int64_t n, int64_t k, int64_t ldx, int64_t ldc, sdnq_stream_t stream);

/* fused variant for the quantized-matmul conv forwards: row scales xs[m] = amax_k |x_unfold[m][k]| / qmax straight from the
 * image, then the unfold writes the QUANTIZED operand xq [M][K] (int8 or fp8-e4m3fn bytes) -- the bf16 [M][K] matrix of
 * process_conv_input + quantize_int_mm_input / quantize_fp_mm_input (conv_int8.py:31, 64; quant_utils.py:265-273, 290-299)
 * is never materialised; values are identical to sdnq_hip_im2col followed by sdnq_hip_rowquant.  K % 16 == 0,
 * kh * kw <= 25 and height * width % 8 == 0 (SDNQ_ERR_UNSUPPORTED otherwise: use im2col + rowquant).  amax_ws:
 * caller-provided workspace of batch * height * width 32-bit words (the per-pixel channel amax map; zeroed here). */
int sdnq_hip_im2col_rowquant(const void* x, int dtype, int batch, int channels, int height, int width, int kh, int kw,
                             int stride_h, int stride_w, int pad_h, int pad_w, int dil_h, int dil_w, int mm_dtype,
                             void* xq, float* xs, void* amax_ws, sdnq_stream_t stream);

/* sdnq_hip_im2col_rowquant with a SELF-CLEANING workspace: zeroed_ws = 8224 + batch * height * width 32-bit words that are ALL ZERO on entry
 * (the first 8224: ticket counters) and are all zero again when the call's last kernel has run -- the last workgroup of the quantizing kernel
 * zeroes the amax map -- so a caller that keeps one such buffer per stream pays the zeroing launch once, not per convolution (4.9 us x 49
 * convs of an SDXL step).  One call at a time per buffer (stream order); a buffer left dirty by a failed call must be zeroed again. */
int sdnq_hip_im2col_rowquant_z(const void* x, int dtype, int batch, int channels, int height, int width, int kh, int kw,
                               int stride_h, int stride_w, int pad_h, int pad_w, int dil_h, int dil_w, int mm_dtype,
                               void* xq, float* xs, void* zeroed_ws, sdnq_stream_t stream);

/* the same few-row linear for 8-bit raw / 4-bit packed integer layers (signed or unsigned, row- or group-wise, group % 4 == 0)
 * WITH SVD factors (w->svd_up [N][R] set): W = round(dequant(q) + svd_up . svd_down) is formed tile by tile with the rank-R product on the matrix cores and never stored (replaces the
 * dequantize incl. addmm_ + F.linear pair of the M < 32 branch, linear_int8.py:102-103 + dequantizer.py:79-83).
 * svd_down_t: [K][R] (the transposed factor, i.e. the reference's stored matmul layout of svd_down), dtype = bf16 / f16
 * = svd dtype; m <= 4, K % 32 == 0, R % 16 == 0; other layouts: SDNQ_ERR_UNSUPPORTED (use dequant + linear_float). */
int sdnq_hip_linear_skinny_svd(const SdnqWeight* w, const void* svd_down_t, const void* x, const void* bias, int dtype,
                               void* out, int64_t m, int64_t ldx, sdnq_stream_t stream);

/* ---- quantized attention (forward) ------------------------------------------------------------
 * replaces sdnq_triton_atten (kernels/triton_atten.py:540-618) in its default configuration: matmul_dtype "int8" for
 * Q.K^T, pv_matmul_dtype None (P.V in the value dtype), smooth_k, optional Hadamard rotation of Q and K (hadamard_group: 0 or a
 * power of two in [4, head_dim] dividing head_dim; apply_hadamard / rotate_hadamard, triton_atten.py:464-467), optional attention mask,
 * optional causal masking,
 * grouped-query head mapping (kv head = h * kv_heads / q_heads, triton_atten.py:212-213).
 * Tensors are [batch][heads][len][head_dim] with head_dim contiguous; q / k / v / out may be strided views (x_strides = element
 * strides {batch, head, token}, each a multiple of 8; NULL = contiguous), e.g. the transposed view of a [batch][len][heads *
 * head_dim] projection output, which the reference would first copy (`.contiguous()`, triton_atten.py:469-470).  head_dim 64
 * or 128; dtype bf16 / f16.
 *
 * sdnq_hip_attn_prepare <- quantize_attn (triton_atten.py:443-487): kmean [batch*kv_heads][32][head_dim] f32 (workspace for
 *   the channel sums of 32 token splits; K minus its token mean when smooth_k), qq / kq int8 codes + qs / ks f32 per-token scales (quantize_int_mm, quant_utils.py:265-273;
 *   qs [batch*q_heads][q_len], ks [batch*kv_heads][kv_len rounded up to 32], padding never read as a value),
 *   vt = V transposed to [batch*kv_heads][head_dim][kv_len rounded up to 32] (zero padded) in the value dtype.
 *   qq == NULL and qs == NULL: K and V only (q is not read) -- for sdnq_hip_attn_fwd_q16.  Launches: one, or two when smooth_k meets more
 *   than 256 keys ({V layout, [Q rows,] channel sums of K} then {K rows}).
 * sdnq_hip_attn_fwd <- sdnq_attn_kernel (triton_atten.py:143-335): out [batch][q_heads][q_len][head_dim] of out_dtype
 *   (the value dtype or f32).  mask: NULL, or the attention mask of get_attn_inputs (triton_atten.py:520-527) addressed as
 *   mask[b * mask_stride_b + h * mask_stride_h + q * mask_stride_q + key] (element strides, 0 where it broadcasts, keys
 *   contiguous); mask_dtype -1 = int8 / bool (0 = masked out, :290-291), else SdnqFloat = additive mask, added to the base-2
 *   logits as is (:292-293).  A query with no visible key returns 0 (l_i stays 1, :232).
 * sdnq_hip_attn_fwd_q16: the same forward with Q still in the value dtype (q [batch][q_heads][q_len][head_dim], element strides q_strides
 *   or NULL = contiguous): each wave quantizes its 32 queries per token (the quantize_int_mm arithmetic of sdnq_hip_attn_prepare, same
 *   codes and scales, so the output is bit-identical to prepare + fwd) -- a query is read by one tile only, so the separate pass over Q
 *   (16-bit read, 8-bit write, 8-bit read) becomes one 16-bit read.  Not with a Hadamard rotation (prepare rotates Q). */
int sdnq_hip_attn_prepare(const void* q, const void* k, const void* v, int dtype, int64_t batch, int64_t q_heads,
                          int64_t kv_heads, int64_t q_len, int64_t kv_len, int64_t head_dim, int smooth_k, int hadamard_group,
                          const int64_t* q_strides, const int64_t* k_strides, const int64_t* v_strides, void* qq, float* qs,
                          void* kq, float* ks, void* vt, float* kmean, sdnq_stream_t stream);
int sdnq_hip_attn_fwd(const void* qq, const float* qs, const void* kq, const float* ks, const void* vt, int v_dtype,
                      float sm_scale, int is_causal, const void* mask, int mask_dtype, int64_t mask_stride_b,
                      int64_t mask_stride_h, int64_t mask_stride_q, void* out, int out_dtype, const int64_t* out_strides,
                      int64_t batch, int64_t q_heads, int64_t kv_heads, int64_t q_len, int64_t kv_len, int64_t head_dim,
                      sdnq_stream_t stream);
int sdnq_hip_attn_fwd_q16(const void* q, const int64_t* q_strides, const void* kq, const float* ks, const void* vt, int v_dtype,
                          float sm_scale, int is_causal, const void* mask, int mask_dtype, int64_t mask_stride_b,
                          int64_t mask_stride_h, int64_t mask_stride_q, void* out, int out_dtype, const int64_t* out_strides,
                          int64_t batch, int64_t q_heads, int64_t kv_heads, int64_t q_len, int64_t kv_len, int64_t head_dim,
                          sdnq_stream_t stream);

/* The other matmul formats of the reference's attention (round 6; quantize_attn triton_atten.py:443-487 with matmul_dtype / pv_matmul_dtype,
 * sdnq_attn_kernel :273-284 and :303-323):
 *   qk_dtype  SDNQ_MM_I8 | SDNQ_MM_FP8: Q.K^T on int8 codes or on e4m3 codes (quantize_fp_mm, quant_utils.py:290-299: scale = amax / 448);
 *   pv_dtype  -1: P.V in the value dtype | SDNQ_MM_I8 | SDNQ_MM_FP8 | SDNQ_MM_F16: V quantized per token (vs [batch*kv_heads][kv_len
 *             rounded up to 32], rotated first under hadamard_group: the CALLER rotates the output back, triton_atten.py:609-612), P scaled
 *             by v_scale and quantized per (query, 32-key block): p_scale = max / qmax (1 where <= 2e-38), int8 floor(fma(p, 1 / p_scale, 0.5)),
 *             fp8 / float16 round-to-nearest-even; acc = fma(dot(p_q, v_q), p_scale, acc).  The 32-key block is the reference's
 *             BLOCK_SIZE_N (autotuned there; the fixtures are made with 32).
 * sdnq_hip_attn_prepare_ex always quantizes Q too (qq / qs required); kq and the 8-bit vt are MFMA-fragment ordered like sdnq_hip_attn_prepare's
 * (vt: [batch*kv_heads][blocks][head_dim/32][64 lanes][16 bytes], or the 16-bit layout for pv_dtype -1 / SDNQ_MM_F16); kmean: [batch*kv_heads][head_dim
 * padded to 64 / 128] floats of workspace when smooth_k.  sdnq_hip_attn_fwd_ex: out_dtype any SdnqFloat; v_dtype matters for pv_dtype -1 only. */
int sdnq_hip_attn_prepare_ex(const void* q, const void* k, const void* v, int dtype, int64_t batch, int64_t q_heads, int64_t kv_heads,
                             int64_t q_len, int64_t kv_len, int64_t head_dim, int smooth_k, int hadamard_group, const int64_t* q_strides,
                             const int64_t* k_strides, const int64_t* v_strides, int qk_dtype, int pv_dtype, void* qq, float* qs, void* kq,
                             float* ks, void* vt, float* vs, float* kmean, sdnq_stream_t stream);
int sdnq_hip_attn_fwd_ex(const void* qq, const float* qs, const void* kq, const float* ks, const void* vt, const float* vs, int v_dtype,
                         int qk_dtype, int pv_dtype, float sm_scale, int is_causal, const void* mask, int mask_dtype, int64_t mask_stride_b,
                         int64_t mask_stride_h, int64_t mask_stride_q, void* out, int out_dtype, const int64_t* out_strides, int64_t batch,
                         int64_t q_heads, int64_t kv_heads, int64_t q_len, int64_t kv_len, int64_t head_dim, sdnq_stream_t stream);

/* sdnq_hip_attn <- sdnq_triton_atten (triton_atten.py:540-618) as ONE call: quantize_attn + sdnq_atten_fwd with the arguments of the three
 * entry points above (q / k / v of `dtype` = bf16 / f16 with element strides or NULL, smooth_k, hadamard_group, sm_scale, is_causal, mask, out).
 * Route, chosen here:
 *   kv_len <= 128 and no rotation (cross-attention onto text tokens): a SINGLE launch -- every workgroup builds its head's smoothed,
 *     quantized K, the V operand and the key scales in LDS and quantizes its own queries; no workspace (workspace_bytes() returns 0);
 *   otherwise sdnq_hip_attn_prepare (K, V; Q too under a rotation) into `workspace` + sdnq_hip_attn_fwd(_q16).
 * Results equal the three-call sequence bit for bit.  workspace: sdnq_hip_attn_workspace_bytes(...) bytes (< 0: SdnqStatus), 256-byte
 * aligned, owned by the caller until the stream has run the call. */
int64_t sdnq_hip_attn_workspace_bytes(int64_t batch, int64_t q_heads, int64_t kv_heads, int64_t q_len, int64_t kv_len, int64_t head_dim,
                                      int hadamard_group);
int sdnq_hip_attn(const void* q, const void* k, const void* v, int dtype, int64_t batch, int64_t q_heads, int64_t kv_heads, int64_t q_len,
                  int64_t kv_len, int64_t head_dim, const int64_t* q_strides, const int64_t* k_strides, const int64_t* v_strides,
                  int smooth_k, int hadamard_group, float sm_scale, int is_causal, const void* mask, int mask_dtype,
                  int64_t mask_stride_b, int64_t mask_stride_h, int64_t mask_stride_q, void* out, int out_dtype,
                  const int64_t* out_strides, void* workspace, int64_t workspace_bytes, sdnq_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SDNQ_HIP_H */
