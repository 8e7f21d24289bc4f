// Row-wise activation quantization (+ fused Hadamard rotation) for gfx950.
//
// Reference chain replaced (per call, several eager/Inductor kernels there; ONE launch here):
//   rotate_hadamard(x)                          quant_utils.py:194-209   (linear_int8.py:55-56)
//   x.flatten(0,-2).to(float32)                 linear_int8.py:16-18
//   scale = amax(|x|, -1) / 127   (or / 448)    quant_utils.py:23-24, 268 / 293
//   q = clamp(round(x / scale), -128, 127).to(int8)                       quant_utils.py:269-272
//   q = clamp(nan_to_num(x / scale), -448, 448).to(float8_e4m3fn)         quant_utils.py:298
//   rowsum = sum(q, -1, dtype=int32)            linear_int8.py:66
//
// HBM-bound: reads 2 B (bf16) and writes 1 B per element + 4 B per row.
// Layout: one wave (64 lanes) per activation row, 8 consecutive elements per lane per pass, so
// every global access is a 16-byte (bf16/f16) or 2x16-byte (f32) lane-contiguous vector and a
// 256/512-wide Hadamard group lives inside one pass (cross-lane butterflies via DPP/bpermute).
// Two phases over the row: (1) amax, (2) quantize; the second read hits L2.
#include <cstdlib>

#include "hadamard_dev.h"
#include "quant8_dev.h"
#include "sdnq_dev.h"

namespace {

// 8 consecutive elements of a row as floats; lanes past the end of the row (`ok` false) get zeros.  The load itself is
// UNCONDITIONAL (a lane past the end re-reads the start of its row): a load under `if (ok)` makes the compiler close every
// pass with its own s_waitcnt vmcnt(0), so the passes of a row were fetched one HBM round trip after the other (round 1:
// 12.2 us for 1024 x 5120 rows, 8.9 us with the loads in flight together).
template <int T_ID>
__device__ __forceinline__ void load8_raw(const void* row, int64_t idx, bool ok, uint4& a, uint4& b) {
    const int64_t i = ok ? idx : 0;
    if constexpr (T_ID == SDNQ_F32) {
        a = *(const uint4*)((const float*)row + i);
        b = *(const uint4*)((const float*)row + i + 4);
    } else {
        a = *(const uint4*)((const uint16_t*)row + i);
        b = a;
    }
}
template <int T_ID>
__device__ __forceinline__ void unpack8(const uint4& a, const uint4& b, bool ok, float (&v)[8]) {
    if constexpr (T_ID == SDNQ_F32) {
        Vec16<SDNQ_F32>::unpack(a, v);
        Vec16<SDNQ_F32>::unpack(b, v + 4);
    } else {
        Vec16<T_ID>::unpack(a, v);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = ok ? v[e] : 0.0f;
}
template <int T_ID>
__device__ __forceinline__ void load8(const void* row, int64_t idx, bool ok, float (&v)[8]) {
    uint4 a, b;
    load8_raw<T_ID>(row, idx, ok, a, b);
    unpack8<T_ID>(a, b, ok, v);
}

template <int T_ID>
__device__ __forceinline__ void store8(void* row, int64_t idx, const float (&v)[8]) {
    if constexpr (T_ID == SDNQ_F32) {
        *(uint4*)((float*)row + idx) = Vec16<SDNQ_F32>::pack(v);
        *(uint4*)((float*)row + idx + 4) = Vec16<SDNQ_F32>::pack(v + 4);
    } else {
        *(uint4*)((uint16_t*)row + idx) = Vec16<T_ID>::pack(v);
    }
}

// T_ID: activation dtype; MM: SdnqMM; HAD: rotate first; NP: 512-element passes of the row held in registers
// (K <= NP*512: the row is read from HBM exactly once, all loads in flight together); NP == 0: two-phase fallback
// for very long rows (second read is L2-hot).
// WPR: waves per row (1, 2 or 4; register-resident path only).  With few rows (M <= 2048: one wave per SIMD at best) a long row
// is the whole latency of the launch -- 80 elements per lane at K = 5120 -- so the row is cut into WPR contiguous parts, one wave
// each, and the partial amax / min / max / row sums meet in LDS (two barriers): 1024 x 5120 rows 9.4 -> ... us per launch.
// LP: dequantize_fp32=False -- the layer's scale is stored in the activation dtype and the reference quantizes the activation in
// THAT dtype (`input.to(dtype=scale.dtype)`, linear_int8.py:15-22): scale = round_T(amax / qmax), q = rint(round_T(x / scale)).
// Built for the two-phase path only (NP == 0): a compatibility mode, not the tuned one.
template <int T_ID, int MM, bool HAD, int NP, int WPR = 1, bool LP = false>
// Argument order: what the row loads need first -- 14 dwords, the part a kernarg-preloading build (SDNQ_PRELOAD_ROWQUANT, build.sh)
// delivers in SGPRs with the wave -- then the rest, fetched in one batch (SDNQ_KERNARGS_NOW, sdnq_dev.h).
__global__ __launch_bounds__(256) void rowquant_kernel(const void* __restrict__ x, int64_t M, int64_t K, int64_t ldx, int row_blocks,
                                                       int log2g, uint8_t* __restrict__ xq, float* __restrict__ xs,
                                                       int32_t* __restrict__ rowsum, void* __restrict__ xrot,
                                                       const uint4* __restrict__ pf, int64_t pf_vecs, float* __restrict__ xzp) {
    // this kernel is 443 of the 905 launches of the SDXL step
    SDNQ_KERNARGS_NOW("s"(x), "s"(M), "s"(K), "s"(ldx), "s"(row_blocks), "s"(log2g), "s"(xq), "s"(xs));
#ifndef SDNQ_PRELOAD_ROWQUANT
    SDNQ_KERNARGS_NOW("s"(rowsum), "s"(xrot), "s"(pf), "s"(pf_vecs), "s"(xzp));
#endif
    if ((int)blockIdx.x >= row_blocks) {
        // software prefetch of the following GEMM's weight operand: these extra workgroups just stream it once so it
        // sits in the last-level cache (MALL) / L2 when the GEMM's LDS-DMA asks for it. 8 loads in flight per lane.
        const int64_t base = ((int64_t)blockIdx.x - row_blocks) * (256 * 8) + threadIdx.x;
        u32 acc = 0;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int64_t i = base + (int64_t)u * 256;
            if (i < pf_vecs) { const uint4 v = pf[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
        }
        if (acc == 0x9e3779b9u && pf_vecs < 0) xs[0] = 0.0f;  // never true: keeps the loads alive
        return;
    }
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    int64_t m = (int64_t)blockIdx.x * (4 / WPR) + wv / WPR;
    const int part = wv % WPR;            // which part of the row this wave owns (WPR > 1)
    const int64_t k_part = (int64_t)part * NP * 512;  // first element of that part
    const bool row_ok = m < M;
    if (WPR == 1 && !row_ok) return;  // whole wave exits together (wave-uniform); with WPR > 1 every wave must reach the barriers
    if (!row_ok) m = M - 1;
    __shared__ float s_stat[4][4];
    __shared__ int s_isum[4];
    const void* row = (const char*)x + m * ldx * FT<T_ID>::bytes;
    const float hscale = HAD ? hadamard_scale(log2g, T_ID) : 1.0f;
    const float qmax = (MM == SDNQ_MM_I8) ? 127.0f : 448.0f;
    uint8_t* qrow = xq + m * K;
    int isum = 0;

    if constexpr (NP > 0) {
        float v[NP][8];
        {
            uint4 ra[NP], rb[NP];  // every pass of the row in flight before the first use
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                const int64_t idx = k_part + (int64_t)p * 512 + lane * 8;
                load8_raw<T_ID>(row, idx, idx < K, ra[p], rb[p]);
            }
#ifdef SDNQ_PRELOAD_ROWQUANT
            // the arguments that did not arrive preloaded: one batch of scalar loads, issued behind the row's loads (their round
            // trip hides under the row's)
            __builtin_amdgcn_sched_barrier(0);
            SDNQ_KERNARGS_NOW("s"(rowsum), "s"(xrot), "s"(xzp));
#endif
#pragma unroll
            for (int p = 0; p < NP; ++p) unpack8<T_ID>(ra[p], rb[p], k_part + (int64_t)p * 512 + lane * 8 < K, v[p]);
        }
        const bool asym = xzp != nullptr;  // asymmetric int8 activations of the uint8 matmul (linear_uint8.py:15-23)
        float amax = 0.0f, vmin = 3.4e38f, vmax = -3.4e38f;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            if constexpr (HAD) {
                if (k_part + (int64_t)p * 512 < K) {  // wave-uniform
                    wave_hadamard(v[p], log2g, hscale);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[p][e] = FT<T_ID>::round(v[p][e]);
                }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) amax = fmaxf(amax, fabsf(v[p][e]));
        }
        if (asym) {  // (wave-uniform; the symmetric rows of the w8a8 step do not pay for the extrema)
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                if (k_part + (int64_t)p * 512 + lane * 8 < K) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) { vmin = fminf(vmin, v[p][e]); vmax = fmaxf(vmax, v[p][e]); }
                }
            }
        }
        float scale, zpv = 0.0f;
        if constexpr (WPR > 1) {  // the parts of a row meet in LDS
            amax = wave_max(amax);
            if (asym) { vmin = wave_min(vmin); vmax = wave_max(vmax); }
            if (lane == 0) { s_stat[wv][0] = amax; s_stat[wv][1] = vmin; s_stat[wv][2] = vmax; }
            __syncthreads();
            const int w0 = wv - part;
#pragma unroll
            for (int q = 0; q < WPR; ++q) {
                amax = fmaxf(amax, s_stat[w0 + q][0]);
                vmin = fminf(vmin, s_stat[w0 + q][1]);
                vmax = fmaxf(vmax, s_stat[w0 + q][2]);
            }
        }
        const bool writer = row_ok && lane == 0 && part == 0;
        if (asym) {
            if constexpr (WPR == 1) { vmin = wave_min(vmin); vmax = wave_max(vmax); }
            scale = (vmax - vmin) / 255.0f;          // get_scale_asymmetric, quant_utils.py:10-19 with the int8 range
            zpv = fmaf(128.0f, scale, vmin);         // zero_point.sub_(scale, alpha=-128); 128*scale is exact
            if (writer) xzp[m] = zpv;
        } else {
            if constexpr (WPR == 1) amax = wave_max(amax);
            scale = amax / qmax;  // IEEE division (get_scale_symmetric, quant_utils.py:23-24)
        }
        if (writer) xs[m] = scale;
        RowDiv rd;
        rd.set(scale);
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int64_t idx = k_part + (int64_t)p * 512 + lane * 8;
            if (idx < K && row_ok) {
                if constexpr (HAD) {
                    if (xrot != nullptr) store8<T_ID>((char*)xrot + m * K * FT<T_ID>::bytes, idx, v[p]);
                }
                {
                    const uint2 qv = quant8<MM>(v[p], rd, isum, zpv, asym);
                    // write-through (sc0 sc1): the codes are read by the following GEMM from OTHER XCDs; left dirty in this XCD's L2 they cost
                    // that GEMM a remote write-back per first touch (row quantization + GEMM pair 12.04 -> 11.19 us, SDXL step 8.08 -> 8.04 ms)
#if !defined(SDNQ_RQ_STORE_PLAIN)
                    typedef int v2i_t __attribute__((ext_vector_type(2)));
                    const v2i_t w2 = {(int)qv.x, (int)qv.y};
                    asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(qrow + idx), "v"(w2) : "memory");
#else
                    *(uint2*)(qrow + idx) = qv;
#endif
                }
            }
        }
    } else {
        const int64_t npass = (K + 511) / 512;
        const bool asym = xzp != nullptr;
        // ---- phase 1: row amax / min / max (after rotation + rounding to the activation dtype)
        float amax = 0.0f, vmin = 3.4e38f, vmax = -3.4e38f;
        for (int64_t p = 0; p < npass; ++p) {
            const int64_t idx = p * 512 + lane * 8;
            float v[8];
            load8<T_ID>(row, idx, idx < K, v);
            if constexpr (HAD) {
                wave_hadamard(v, log2g, hscale);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = FT<T_ID>::round(v[e]);
                // with a rotated-copy buffer the FWHT -- the expensive part of a rotated row -- runs only once: the rounded row
                // is parked there (L2-resident, 2-4 bytes per element) and phase 2 reads it back
                if (xrot != nullptr && idx < K) store8<T_ID>((char*)xrot + m * K * FT<T_ID>::bytes, idx, v);
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                amax = fmaxf(amax, fabsf(v[e]));
                if (idx < K) { vmin = fminf(vmin, v[e]); vmax = fmaxf(vmax, v[e]); }
            }
        }
        float scale, zpv = 0.0f;
        if (asym) {  // quantize_uint_mm_input, as in the register-resident path above
            vmin = wave_min(vmin);
            vmax = wave_max(vmax);
            if constexpr (LP) {  // get_scale_asymmetric on 16-bit tensors (quant_utils.py:10-19): sub_, div_ and the alpha-sub each round once
                scale = FT<T_ID>::round(FT<T_ID>::round(vmax - vmin) / 255.0f);
                zpv = FT<T_ID>::round(fmaf(128.0f, scale, vmin));
            } else {
                scale = (vmax - vmin) / 255.0f;
                zpv = fmaf(128.0f, scale, vmin);
            }
            if (lane == 0) xzp[m] = zpv;
        } else {
            amax = wave_max(amax);
            scale = amax / qmax;
            if constexpr (LP) scale = FT<T_ID>::round(scale);
        }
        if (lane == 0) xs[m] = scale;
        RowDiv rd;
        rd.set(scale);
        // ---- phase 2: quantize
        for (int64_t p = 0; p < npass; ++p) {
            const int64_t idx = p * 512 + lane * 8;
            const bool ok = idx < K;
            float v[8];
            if (HAD && xrot != nullptr) {  // wave-uniform
                load8<T_ID>((const char*)xrot + m * K * FT<T_ID>::bytes, idx, ok, v);  // this lane's own stores of phase 1
            } else {
                load8<T_ID>(row, idx, ok, v);
                if constexpr (HAD) {
                    wave_hadamard(v, log2g, hscale);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = FT<T_ID>::round(v[e]);
                }
            }
            int isum_p = 0;
            const uint2 w = quant8<MM, LP ? T_ID : SDNQ_F32>(v, rd, isum_p, zpv, asym);
            if (ok) { *(uint2*)(qrow + idx) = w; isum += isum_p; }
        }
    }
    if (rowsum != nullptr) {
        isum = wave_sum_i32(isum);
        if constexpr (WPR > 1) {
            if (lane == 0) s_isum[wv] = isum;
            __syncthreads();
            if (part == 0) {
                isum = 0;
#pragma unroll
                for (int q = 0; q < WPR; ++q) isum += s_isum[wv + q];
            }
        }
        if (row_ok && lane == 0 && part == 0) rowsum[m] = isum;
    }
}

// 4-byte store of quantized codes, write-through (see rowquant_kernel: the consumer GEMM reads them from other XCDs)
__device__ __forceinline__ void store_codes4(uint8_t* dst, u32 w) {
#if !defined(SDNQ_RQ_STORE_PLAIN)
    asm volatile("global_store_dword %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(dst), "v"(w) : "memory");
#else
    *(u32*)dst = w;
#endif
}

// ---- group-256 Hadamard rotation on the matrix cores -------------------------------------------------------------------------------
// H_256 = kron^4(H4) / 16 = (H16 / 4) (x) (H16 / 4) with H16 = H4 (x) H4 (quant_utils.py:157-165), so for a group seen as a 16 x 16
// row-major matrix X (element 16 r + c):  rotate(x) = vec( (H16/4) . X . (H16/4) )  -- two 16 x 16 x 16 matrix products instead of
// four radix-4 butterfly stages over registers and lanes (the FWHT of wave_hadamard costs ~13 VALU per element, most of them DPP /
// ds_swizzle moves: 4608 x 15360 rows took 130 us rotated vs 61 us plain).
//   stage 1: D1 = X . (H16/4)      v_mfma_f32_16x16x16 (bf16 / f16): A = X in the A layout -- lane l holds X[l & 15][4 (l >> 4) .. +3], one
//            8-byte load, a whole group per wave-load (512 contiguous bytes) -- B = the constant matrix, generated in registers;
//   stage 2: Y^T = D1^T . (H16/4)  4 x v_mfma_f32_16x16x4_f32: stage 1's accumulator registers (lane l: D1[4 (l >> 4) + s][l & 15]) ARE the
//            A operand of D1^T, full fp32, no rounding in between; the output lands in the layout the input came in: lane l holds
//            Y[l & 15][4 (l >> 4) .. +3], so codes / the rotated copy leave as 4- / 8-byte pieces of 256- / 512-byte contiguous runs.
// Products are exact (x . +-1/4), sums are fp32, ONE rounding to the activation dtype at the end, as rotate_hadamard's matmul.
// One workgroup = 4 waves; WPR = 1: a row per wave (K <= 4096), WPR = 4: the row split over the four waves (K <= 16384).
// standalone rotation with the same arithmetic (sdnq_hip_hadamard, group 256, 16-bit): one wave per row, a group per step
template <int T_ID>
__global__ __launch_bounds__(256) void hadamard256_kernel(const void* __restrict__ x, int64_t rows, int64_t K, int64_t ldx, void* __restrict__ y, int64_t ldy) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const int eoff = 16 * (lane & 15) + 4 * (lane >> 4);
    const uint16_t* src = (const uint16_t*)x + r * ldx + eoff;
    uint16_t* dst = (uint16_t*)y + r * ldy + eoff;
    float hf[4];
    had16_operand(lane, hf);
    const int ngroups = (int)(K / 256);
    for (int g0 = 0; g0 < ngroups; g0 += 4) {  // four groups' loads in flight
        uint2 raw[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) raw[u] = *(const uint2*)(src + (int64_t)(g0 + u < ngroups ? g0 + u : g0) * 256);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (g0 + u < ngroups) {
                const v4f v = had256_group<T_ID>(raw[u], hf);
                const uint32_t lo = (uint32_t)FT<T_ID>::bits(v[0]) | ((uint32_t)FT<T_ID>::bits(v[1]) << 16);
                const uint32_t hi = (uint32_t)FT<T_ID>::bits(v[2]) | ((uint32_t)FT<T_ID>::bits(v[3]) << 16);
                *(uint2*)(dst + (int64_t)(g0 + u) * 256) = make_uint2(lo, hi);
            }
        }
    }
}

// element `hi` (0 / 1) of a dword of two 16-bit activations, as a float
template <int T_ID> __device__ __forceinline__ float unpk(u32 w, int hi) {
    if constexpr (T_ID == SDNQ_BF16) return __uint_as_float(hi ? (w & 0xffff0000u) : (w << 16));
    else return f16_bits_to_f32((uint16_t)(hi ? (w >> 16) : (w & 0xffffu)));
}

template <int T_ID, int MM, int NG, int WPR>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu((NG >= 16 || (NG >= 12 && WPR > 1)) ? 4 : 5))) void rowquant_had256_kernel(const void* __restrict__ x, int64_t M, int64_t K, int64_t ldx, uint8_t* __restrict__ xq,
                                                              float* __restrict__ xs, int32_t* __restrict__ rowsum, void* __restrict__ xrot) {
    static_assert(T_ID == SDNQ_BF16 || T_ID == SDNQ_F16, "16-bit activations");
    SDNQ_KERNARGS_NOW("s"(x), "s"(M), "s"(K), "s"(ldx), "s"(xq), "s"(xs), "s"(rowsum));  // the 14 dwords a preloading build delivers
#ifndef SDNQ_PRELOAD_ROWQUANT
    SDNQ_KERNARGS_NOW("s"(xrot));
#endif
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int64_t m = (int64_t)blockIdx.x * (4 / WPR) + wv / WPR;
    const int part = wv % WPR;
    const bool row_ok = m < M;
    if (WPR == 1 && !row_ok) return;
    if (!row_ok) m = M - 1;
    __shared__ float s_amax[4];
    __shared__ int s_isum[4];
    const int ngroups = (int)(K / 256);
    const int g0 = part * NG;  // first group of this wave
    const int eoff = 16 * (lane & 15) + 4 * (lane >> 4);  // this lane's 4 consecutive elements inside a group
    const uint16_t* row = (const uint16_t*)x + m * ldx + eoff;
    float hf[4];
    had16_operand(lane, hf);
    uint2 raw[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) {  // all loads in flight; a group past the end of the row re-reads the first one and is dropped
        const int gg = (g0 + g < ngroups) ? g0 + g : 0;
        raw[g] = *(const uint2*)(row + (int64_t)gg * 256);
    }
    // the rotated row is kept as it will be stored -- rounded to the activation dtype, two elements per register -- not as floats:
    // 2 registers per group instead of 4 (+ the 2 of the raw load, dead by then), so a 12-group row (K = 3072) fits 8 waves per SIMD
    // instead of 3 and a 4608-row launch is one round of workgroups instead of two
    u32 pk[NG][2];
    float amax = 0.0f;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const v4f y = had256_group<T_ID>(raw[g], hf);
        const bool live = g0 + g < ngroups;  // wave-uniform
        pk[g][0] = live ? pack2<T_ID>(y[0], y[1]) : 0u;
        pk[g][1] = live ? pack2<T_ID>(y[2], y[3]) : 0u;
#pragma unroll
        for (int e = 0; e < 4; ++e) amax = fmaxf(amax, fabsf(unpk<T_ID>(pk[g][e >> 1], e & 1)));
    }
    amax = wave_max(amax);
    if constexpr (WPR > 1) {
        if (lane == 0) s_amax[wv] = amax;
        __syncthreads();
        amax = fmaxf(fmaxf(s_amax[0], s_amax[1]), fmaxf(s_amax[2], s_amax[3]));
    }
    const float qmax = (MM == SDNQ_MM_I8) ? 127.0f : 448.0f;
    const float scale = amax / qmax;  // IEEE division (get_scale_symmetric, quant_utils.py:23-24)
    if (row_ok && lane == 0 && part == 0) xs[m] = scale;
    RowDiv rd;
    rd.set(scale);
    uint8_t* qrow = xq + m * K + eoff;
    int isum = 0;
    if (xrot != nullptr) {  // the rotated copy (needed by the SVD branch)
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (g0 + g < ngroups && row_ok)
                *(uint2*)((uint16_t*)xrot + m * K + eoff + (int64_t)(g0 + g) * 256) = make_uint2(pk[g][0], pk[g][1]);
        }
    }
#ifndef SDNQ_HAD256_NO_LEAN  // (A/B builds)
    if (MM == SDNQ_MM_I8 && rd.fast) {  // wave-uniform
        // the lean codes of the plain row quantizer (quant8_fast: packed correctly rounded division, rint + int8 cast as one packed add,
        // byte permutes, v_dot4 row sums -- ~4 instructions per element where the general loop below spends ~10; bit-identical by
        // construction).  Round 4 built this as a second path INSIDE the general loop and spilled; as its own loop nothing does.
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (g0 + g < ngroups && row_ok) {
                // (opaque copies: left visible, the unpacking of all NG groups is hoisted above the branch as common to both loops --
                //  64 more live registers; the 16-group instantiation then spilled 28)
                u32 p0 = pk[g][0], p1 = pk[g][1];
                asm volatile("" : "+v"(p0), "+v"(p1));
                const pv2f a = {unpk<T_ID>(p0, 0), unpk<T_ID>(p0, 1)}, b = {unpk<T_ID>(p1, 0), unpk<T_ID>(p1, 1)};
                const u32 w = pack4_rne_i8(fastdiv2(a, rd), fastdiv2(b, rd));
                isum = __builtin_amdgcn_sdot4((int)w, 0x01010101, isum, false);
                store_codes4(qrow + (int64_t)(g0 + g) * 256, w);
            }
        }
    } else
#endif
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        if (g0 + g < ngroups && row_ok) {
            const int64_t go = (int64_t)(g0 + g) * 256;
            float qd[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float ve = unpk<T_ID>(pk[g][e >> 1], e & 1);
                qd[e] = rd.fast ? rd.fastdiv(ve) : ve / scale;  // rd.fast is wave-uniform
            }
            if constexpr (MM == SDNQ_MM_FP8) {
                float c[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float q = qd[e];
                    if (q != q) q = 0.0f;
                    c[e] = fminf(fmaxf(q, -448.0f), 448.0f);
                }
                store_codes4(qrow + go, pack4_e4m3fn_clamped(c[0], c[1], c[2], c[3]));
            } else {
                u32 w = 0;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float q = (scale == 0.0f) ? 0.0f : __builtin_rintf(qd[e]);
                    q = fminf(fmaxf(q, -128.0f), 127.0f);
                    const int qi = (int)q;
                    isum += qi;
                    w |= ((u32)qi & 0xffu) << (8 * e);
                }
                store_codes4(qrow + go, w);
            }
        }
    }
    if (rowsum != nullptr) {
        isum = wave_sum_i32(isum);
        if constexpr (WPR > 1) {
            if (lane == 0) s_isum[wv] = isum;
            __syncthreads();
            isum = (s_isum[0] + s_isum[1]) + (s_isum[2] + s_isum[3]);
        }
        if (row_ok && lane == 0 && part == 0) rowsum[m] = isum;
    }
}

// standalone rotation: y = rotate(x), rounded to dtype
template <int T_ID>
__global__ __launch_bounds__(256) void hadamard_kernel(const void* __restrict__ x, int64_t rows, int64_t K, int64_t ldx,
                                                       int log2g, void* __restrict__ y, int64_t ldy) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const void* row = (const char*)x + r * ldx * FT<T_ID>::bytes;
    void* orow = (char*)y + r * ldy * FT<T_ID>::bytes;
    const float hscale = hadamard_scale(log2g, T_ID);
    const int64_t npass = (K + 511) / 512;
    for (int64_t p = 0; p < npass; ++p) {
        const int64_t idx = p * 512 + lane * 8;
        const bool ok = idx < K;
        float v[8];
        load8<T_ID>(row, idx, ok, v);
        wave_hadamard(v, log2g, hscale);
        if (ok) store8<T_ID>(orow, idx, v);
    }
}

int ilog2(int64_t v) {
    int l = 0;
    while ((1LL << l) < v) ++l;
    return l;
}

}  // namespace

extern "C" int sdnq_hip_rowquant(const void* x, int x_dtype, int64_t m, int64_t k, int64_t ldx, int mm_dtype,
                                 int hadamard_group, void* xq, float* xs, int32_t* rowsum, void* xrot,
                                 const void* prefetch, int64_t prefetch_bytes, float* xzp, sdnq_stream_t stream) {
    if (!x || !xq || !xs) return SDNQ_ERR_NULL;
    if (xzp && mm_dtype != SDNQ_MM_I8) return SDNQ_ERR_UNSUPPORTED;  // asymmetric activations exist for the int8 operand only
    if (m <= 0 || k <= 0 || (k % 8) != 0 || ldx < k) return SDNQ_ERR_SHAPE;
    if (mm_dtype != SDNQ_MM_I8 && mm_dtype != SDNQ_MM_FP8) return SDNQ_ERR_DTYPE;
    if (x_dtype < 0 || x_dtype > 2) return SDNQ_ERR_DTYPE;
    const int eb = (x_dtype == SDNQ_F32) ? 4 : 2;
    if (((uintptr_t)x % 16) || ((ldx * eb) % 16) || ((uintptr_t)xq % 8)) return SDNQ_ERR_ALIGN;
    int log2g = 0;
    if (hadamard_group != 0) {
        log2g = ilog2(hadamard_group);
        if ((1 << log2g) != hadamard_group || hadamard_group < 4 || hadamard_group > 512 || (k % hadamard_group) != 0)
            return SDNQ_ERR_SHAPE;
        if (hadamard_group < 8 && false) return SDNQ_ERR_UNSUPPORTED;
        if (xrot && ((uintptr_t)xrot % 16)) return SDNQ_ERR_ALIGN;
    }
    if (rowsum && mm_dtype != SDNQ_MM_I8) return SDNQ_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    static const int had_mfma = [] { const char* e = getenv("SDNQ_HIP_HADAMARD_MFMA"); return e ? atoi(e) : 1; }();  // 0: always the FWHT (A/B aid)
    if (hadamard_group == 256 && x_dtype != SDNQ_F32 && !xzp && k <= 16384 && had_mfma && ((uintptr_t)x % 8) == 0 && ((ldx * 2) % 8) == 0 &&
        ((uintptr_t)xq % 4) == 0 && (!xrot || ((uintptr_t)xrot % 8) == 0)) {
        // the default group size, 16-bit activations: rotation on the matrix cores (rowquant_had256_kernel)
        const int groups = (int)(k / 256);
        static const int wide_min = [] { const char* e = getenv("SDNQ_HIP_RQH_WIDE_MIN"); return e ? atoi(e) : 16; }();  // tuning aid
        // the row split over four waves: long rows, and FEW rows of >= 8 groups (512 x 3072: 4.9 -> 3.8 us; at 4608 rows the one-wave
        // form is the faster one, 15.0 vs 15.9 us).  Every wave must keep at least one group.
        const bool wide = groups > wide_min || (m <= 1024 && groups >= 8 && ((groups + 3) / 4) * 3 < groups);
        const int per = wide ? (groups + 3) / 4 : groups;
        dim3 grid((unsigned)(wide ? m : (m + 3) / 4)), block(256);
#define RQH(T, MMV, NGV, W) hipLaunchKernelGGL((rowquant_had256_kernel<T, MMV, NGV, W>), grid, block, 0, s, x, m, k, ldx, (uint8_t*)xq, xs, rowsum, xrot)
#define RQH_NG(T, MMV, W) do { if (per <= 4) RQH(T, MMV, 4, W); else if (per <= 8) RQH(T, MMV, 8, W); else if (per <= 12) RQH(T, MMV, 12, W); else RQH(T, MMV, 16, W); } while (0)
#define RQH_W(T, MMV) do { if (wide) RQH_NG(T, MMV, 4); else RQH_NG(T, MMV, 1); } while (0)
#define RQH_MM(T) do { if (mm_dtype == SDNQ_MM_I8) RQH_W(T, SDNQ_MM_I8); else RQH_W(T, SDNQ_MM_FP8); } while (0)
        if (x_dtype == SDNQ_BF16) RQH_MM(SDNQ_BF16);
        else RQH_MM(SDNQ_F16);
        SDNQ_CHECK_LAUNCH();
        return SDNQ_OK;
    }
    const int np = (int)((k + 511) / 512);
    // few long rows: two waves per row (see rowquant_kernel); the prefetch blocks are not combined with it
    static const int split_env = [] { const char* e = getenv("SDNQ_HIP_RQ_SPLIT"); return e ? atoi(e) : 0; }();  // tuning aid: 1, 2, 4
    const bool can_split = m <= 2048 && np >= 3 && !(prefetch && prefetch_bytes > 0);
    // very long rows (5120 < K <= 16384: FLUX's 12288 / 15360-wide activations): four waves per row hold a quarter each in registers
    // -- ONE pass over HBM instead of the two-phase fallback's two (4608 x 15360: 66 us per launch on the two-phase path)
    const bool long4 = np > 10 && np <= 32 && !(prefetch && prefetch_bytes > 0) && split_env != 1;
    // Two waves per row (1024 x 1280 rows: 3.2 us against 3.6 replayed alone) LOSE inside the step -- judged there, same box, two runs
    // each: SDXL int8 step 8.097 / 8.099 ms with them, 7.967 / 7.966 without, fp8 step 8.16 -> 8.11, FLUX int8 + SVD neutral
    // (profiles/r04_rowquant_split_in_step.txt): twice the workgroups to launch and drain in front of a GEMM that waits for the last of
    // them.  Four waves per row for K >= 3584 still pay (all splits off: 8.005).  SDNQ_HIP_RQ_SPLIT2=1 brings the two-wave rows back.
    static const int split2_env = [] { const char* e = getenv("SDNQ_HIP_RQ_SPLIT2"); return e ? atoi(e) : 0; }();
    static const int split4_np = [] { const char* e = getenv("SDNQ_HIP_RQ_SPLIT4_NP"); return e ? atoi(e) : 7; }();    // tuning aid: four-wave rows from this many 512-element passes
    const bool split4 = long4 || (can_split && np <= 12 && (split_env == 4 || (split_env == 0 && np >= split4_np)));
    const bool split2 = can_split && !split4 && np <= 10 && split_env != 1 && split2_env != 0;
    const int row_blocks = split4 ? (int)m : (split2 ? (int)((m + 1) / 2) : (int)((m + 3) / 4));
    const uint4* pf = (const uint4*)prefetch;
    int64_t pf_vecs = 0;
    if (prefetch && prefetch_bytes > 0 && ((uintptr_t)prefetch % 16) == 0) {
        const int64_t cap = (int64_t)32 << 20;  // at most 32 MiB: beyond that the prefetch outlives this kernel
        pf_vecs = (prefetch_bytes < cap ? prefetch_bytes : cap) / 16;
    }
    const int pf_blocks = (int)((pf_vecs + 256 * 8 - 1) / (256 * 8));
    dim3 grid((unsigned)(row_blocks + pf_blocks)), block(256);
#define RQ_LAUNCH(T, MMV, H, NPV) \
    hipLaunchKernelGGL((rowquant_kernel<T, MMV, H, NPV>), grid, block, 0, s, x, m, k, ldx, row_blocks, log2g, (uint8_t*)xq, xs, rowsum, xrot, pf, pf_vecs, xzp)
#define RQ_LAUNCH2(T, MMV, H, NPV) \
    hipLaunchKernelGGL((rowquant_kernel<T, MMV, H, NPV, 2>), grid, block, 0, s, x, m, k, ldx, row_blocks, log2g, (uint8_t*)xq, xs, rowsum, xrot, pf, pf_vecs, xzp)
#define RQ_LAUNCH4(T, MMV, H, NPV) \
    hipLaunchKernelGGL((rowquant_kernel<T, MMV, H, NPV, 4>), grid, block, 0, s, x, m, k, ldx, row_blocks, log2g, (uint8_t*)xq, xs, rowsum, xrot, pf, pf_vecs, xzp)
#define RQ_DISPATCH_NP(T, MMV, H)               \
    do {                                        \
        if (split4 && np <= 4) RQ_LAUNCH4(T, MMV, H, 1);      \
        else if (split4 && np <= 8) RQ_LAUNCH4(T, MMV, H, 2); \
        else if (split4 && np <= 12) RQ_LAUNCH4(T, MMV, H, 3); \
        else if (split4) RQ_LAUNCH4(T, MMV, H, 8);            \
        else if (split2 && np <= 4) RQ_LAUNCH2(T, MMV, H, 2);      \
        else if (split2 && np <= 6) RQ_LAUNCH2(T, MMV, H, 3); \
        else if (split2) RQ_LAUNCH2(T, MMV, H, 5);            \
        else if (np <= 2) RQ_LAUNCH(T, MMV, H, 2);   \
        else if (np <= 3) RQ_LAUNCH(T, MMV, H, 3); \
        else if (np <= 5) RQ_LAUNCH(T, MMV, H, 5); \
        else if (np <= 10) RQ_LAUNCH(T, MMV, H, 10); \
        else RQ_LAUNCH(T, MMV, H, 0);            \
    } while (0)
#define RQ_DISPATCH_H(T, MMV)                      \
    do {                                           \
        if (log2g) RQ_DISPATCH_NP(T, MMV, true);    \
        else RQ_DISPATCH_NP(T, MMV, false);         \
    } while (0)
#define RQ_DISPATCH_MM(T)                                         \
    do {                                                          \
        if (mm_dtype == SDNQ_MM_I8) RQ_DISPATCH_H(T, SDNQ_MM_I8); \
        else RQ_DISPATCH_H(T, SDNQ_MM_FP8);                       \
    } while (0)
    switch (x_dtype) {
        case SDNQ_F32: RQ_DISPATCH_MM(SDNQ_F32); break;
        case SDNQ_BF16: RQ_DISPATCH_MM(SDNQ_BF16); break;
        default: RQ_DISPATCH_MM(SDNQ_F16); break;
    }
    SDNQ_CHECK_LAUNCH();
    return SDNQ_OK;
}

static int rowquant_lp_impl(const void* x, int x_dtype, int64_t m, int64_t k, int64_t ldx, int mm_dtype, int hadamard_group,
                            void* xq, float* xs, float* xzp, int32_t* rowsum, void* xrot, sdnq_stream_t stream) {
    if (!x || !xq || !xs) return SDNQ_ERR_NULL;
    if (m <= 0 || k <= 0 || (k % 8) != 0 || ldx < k) return SDNQ_ERR_SHAPE;
    if (mm_dtype != SDNQ_MM_I8 && mm_dtype != SDNQ_MM_FP8) return SDNQ_ERR_DTYPE;
    if (x_dtype != SDNQ_BF16 && x_dtype != SDNQ_F16) return SDNQ_ERR_DTYPE;  // float32 scales: sdnq_hip_rowquant
    if (((uintptr_t)x % 16) || ((ldx * 2) % 16) || ((uintptr_t)xq % 8)) return SDNQ_ERR_ALIGN;
    int log2g = 0;
    if (hadamard_group != 0) {
        log2g = ilog2(hadamard_group);
        if ((1 << log2g) != hadamard_group || hadamard_group < 4 || hadamard_group > 512 || (k % hadamard_group) != 0)
            return SDNQ_ERR_SHAPE;
        if (xrot && ((uintptr_t)xrot % 16)) return SDNQ_ERR_ALIGN;
    }
    if (rowsum && mm_dtype != SDNQ_MM_I8) return SDNQ_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    const int row_blocks = (int)((m + 3) / 4);
    dim3 grid((unsigned)row_blocks), block(256);
#define RQLP(T, MMV, H) \
    hipLaunchKernelGGL((rowquant_kernel<T, MMV, H, 0, 1, true>), grid, block, 0, s, x, m, k, ldx, row_blocks, log2g, (uint8_t*)xq, xs, rowsum, xrot, \
                       (const uint4*)nullptr, (int64_t)0, xzp)
#define RQLP_H(T, MMV) do { if (log2g) RQLP(T, MMV, true); else RQLP(T, MMV, false); } while (0)
#define RQLP_MM(T) do { if (mm_dtype == SDNQ_MM_I8) RQLP_H(T, SDNQ_MM_I8); else RQLP_H(T, SDNQ_MM_FP8); } while (0)
    if (x_dtype == SDNQ_BF16) RQLP_MM(SDNQ_BF16);
    else RQLP_MM(SDNQ_F16);
    SDNQ_CHECK_LAUNCH();
    return SDNQ_OK;
}

extern "C" int sdnq_hip_rowquant_lp(const void* x, int x_dtype, int64_t m, int64_t k, int64_t ldx, int mm_dtype, int hadamard_group,
                                    void* xq, float* xs, int32_t* rowsum, void* xrot, sdnq_stream_t stream) {
    return rowquant_lp_impl(x, x_dtype, m, k, ldx, mm_dtype, hadamard_group, xq, xs, nullptr, rowsum, xrot, stream);
}

extern "C" int sdnq_hip_rowquant_lp_asym(const void* x, int x_dtype, int64_t m, int64_t k, int64_t ldx, int hadamard_group, void* xq,
                                         float* xs, float* xzp, int32_t* rowsum, void* xrot, sdnq_stream_t stream) {
    if (!xzp) return SDNQ_ERR_NULL;
    return rowquant_lp_impl(x, x_dtype, m, k, ldx, SDNQ_MM_I8, hadamard_group, xq, xs, xzp, rowsum, xrot, stream);
}

// Activation row quantization of the float16 matmul (round 6): quantize_fp_mm_input(input, dtype=scale.dtype, matmul_dtype="float16")
// (layers/linear/linear_fp8.py:15-22 -> quantize_fp_mm, quant_utils.py:290-299, called from linear_fp16.py:46): in float32,
// scale = amax(|x|) / 65504, q = clamp(nan_to_num(x / scale), +-65504) rounded to float16.  One wave per row, two passes over the row
// (the second is L2-hot): a compatibility path, not a tuned one.
template <int T_ID>
__global__ __launch_bounds__(256) void rowquant_f16_kernel(const void* __restrict__ x, int64_t M, int64_t K, int64_t ldx, uint16_t* __restrict__ xq,
                                                           float* __restrict__ xs) {
    const int lane = threadIdx.x & 63;
    const int64_t m = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    const void* row = (const char*)x + m * ldx * FT<T_ID>::bytes;
    float amax = 0.0f;
    bool has_nan = false;
    for (int64_t k0 = (int64_t)lane * 8; k0 < K; k0 += 512) {
        float v[8];
        load8<T_ID>(row, k0, true, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            amax = fmaxf(amax, fabsf(v[e]));  // (fmaxf drops a NaN operand: torch.amax propagates it)
            has_nan |= v[e] != v[e];
        }
    }
    amax = wave_max(amax);
    if (__builtin_amdgcn_ballot_w64(has_nan) != 0) amax = __uint_as_float(0x7fc00000u);
    const float scale = amax / 65504.0f;
    if (lane == 0) xs[m] = scale;
    for (int64_t k0 = (int64_t)lane * 8; k0 < K; k0 += 512) {
        float v[8], q[8];
        load8<T_ID>(row, k0, true, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float t = v[e] / scale;
            if (t != t) t = 0.0f;  // nan_to_num (0 / 0 of an all-zero row, a NaN row); +-inf fall to the clamp
            q[e] = fminf(fmaxf(t, -65504.0f), 65504.0f);
        }
        *(uint4*)(xq + m * K + k0) = Vec16<SDNQ_F16>::pack(q);
    }
}

extern "C" int sdnq_hip_rowquant_f16(const void* x, int x_dtype, int64_t m, int64_t k, int64_t ldx, void* xq, float* xs, sdnq_stream_t stream) {
    if (!x || !xq || !xs) return SDNQ_ERR_NULL;
    if (x_dtype < 0 || x_dtype > 2) return SDNQ_ERR_DTYPE;
    if (m <= 0 || k <= 0 || (k % 8) != 0 || ldx < k) return SDNQ_ERR_SHAPE;
    const int eb = (x_dtype == SDNQ_F32) ? 4 : 2;
    if (((uintptr_t)x % 16) || ((uintptr_t)xq % 16) || ((ldx * eb) % 16)) return SDNQ_ERR_ALIGN;
    hipStream_t s = (hipStream_t)stream;
    dim3 grid((unsigned)((m + 3) / 4)), block(256);
    switch (x_dtype) {
        case SDNQ_F32: hipLaunchKernelGGL((rowquant_f16_kernel<SDNQ_F32>), grid, block, 0, s, x, m, k, ldx, (uint16_t*)xq, xs); break;
        case SDNQ_BF16: hipLaunchKernelGGL((rowquant_f16_kernel<SDNQ_BF16>), grid, block, 0, s, x, m, k, ldx, (uint16_t*)xq, xs); break;
        default: hipLaunchKernelGGL((rowquant_f16_kernel<SDNQ_F16>), grid, block, 0, s, x, m, k, ldx, (uint16_t*)xq, xs); break;
    }
    SDNQ_CHECK_LAUNCH();
    return SDNQ_OK;
}

extern "C" int sdnq_hip_hadamard(const void* x, int dtype, int64_t rows, int64_t k, int64_t ldx, int hadamard_group,
                                 void* y, int64_t ldy, sdnq_stream_t stream) {
    if (!x || !y) return SDNQ_ERR_NULL;
    if (rows <= 0 || k <= 0 || (k % 8) != 0 || ldx < k || ldy < k) return SDNQ_ERR_SHAPE;
    if (dtype < 0 || dtype > 2) return SDNQ_ERR_DTYPE;
    const int log2g = ilog2(hadamard_group);
    if ((1 << log2g) != hadamard_group || hadamard_group < 4 || hadamard_group > 512 || (k % hadamard_group) != 0)
        return SDNQ_ERR_SHAPE;
    const int eb = (dtype == SDNQ_F32) ? 4 : 2;
    if (((uintptr_t)x % 16) || ((uintptr_t)y % 16) || ((ldx * eb) % 16) || ((ldy * eb) % 16)) return SDNQ_ERR_ALIGN;
    hipStream_t s = (hipStream_t)stream;
    dim3 grid((unsigned)((rows + 3) / 4)), block(256);
    static const int had_mfma = [] { const char* e = getenv("SDNQ_HIP_HADAMARD_MFMA"); return e ? atoi(e) : 1; }();
    if (hadamard_group == 256 && dtype != SDNQ_F32 && had_mfma) {  // the matrix-core rotation of rowquant_had256_kernel, bit for bit
        if (dtype == SDNQ_BF16) hipLaunchKernelGGL((hadamard256_kernel<SDNQ_BF16>), grid, block, 0, s, x, rows, k, ldx, y, ldy);
        else hipLaunchKernelGGL((hadamard256_kernel<SDNQ_F16>), grid, block, 0, s, x, rows, k, ldx, y, ldy);
        SDNQ_CHECK_LAUNCH();
        return SDNQ_OK;
    }
    switch (dtype) {
        case SDNQ_F32: hipLaunchKernelGGL((hadamard_kernel<SDNQ_F32>), grid, block, 0, s, x, rows, k, ldx, log2g, y, ldy); break;
        case SDNQ_BF16: hipLaunchKernelGGL((hadamard_kernel<SDNQ_BF16>), grid, block, 0, s, x, rows, k, ldx, log2g, y, ldy); break;
        default: hipLaunchKernelGGL((hadamard_kernel<SDNQ_F16>), grid, block, 0, s, x, rows, k, ldx, log2g, y, ldy); break;
    }
    SDNQ_CHECK_LAUNCH();
    return SDNQ_OK;
}
