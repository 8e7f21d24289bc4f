// The w8a8 Linear as ONE launch: the GEMM workgroup row-quantizes its own activation rows into LDS (north_star: "activations
// row-quantized on the fly in LDS") and streams only the weight operand through its LDS-DMA ring.
//
// Reference chain replaced (linear_int8.py:15-22, 64 -> kernels/triton_scaled_mm.py:194-232; linear_fp8.py the same with e4m3fn codes):
//   x.to(float32); scale = amax(|x|, -1) / 127; q = clamp(round(x / scale), -128, 127).to(int8)      one Inductor kernel per call
//   out = (f32(q @ Wq^T) * scale) * ws [+ bias]                                                        the Triton scaled-mm kernel
// Here (rounds 1-4): sdnq_hip_rowquant + sdnq_hip_scaled_mm, two launches -- at the bs = 1 sizes of an SDXL step the first is 5 us of
// launch boundary, cold first bytes and a write-through of codes that the second launch waits another 1.5 us to read back.
//
// Why this form and not the ones measured before (profiles/r02_fused_rowquant_prologue.txt, r02_sync_lab.txt):
//   * the row scale needs the WHOLE row, so a workgroup quantizes whole rows: BM = 64 rows x K <= 1280 codes = 80 KB stay RESIDENT in
//     LDS (in the ring's own stage layout: 128-byte rows, XOR-swizzled chunks), next to a 4-deep ring of weight stages (64 KB);
//   * every workgroup of a row block repeats the quantization (tiles_n of them: 10 for N = 1280) -- affordable only since round 4's
//     lean arithmetic (quant8_dev.h: ~3 instructions per element instead of ~20; round 2 measured this form at 17.9 us with the
//     IEEE-division sequence);
//   * the K loop then fills LDS with weight rows only (2/3 of the bytes per stage of the 64x128 tile), and nothing is exchanged
//     between workgroups: no flags, no grid barrier, no deadlock to argue about.
// Bit-identical to the two-launch route by construction (the same quant8, the same MFMA, the same epilogue expression) and by
// tests/test_gemm_aq.py.
//
// Quantization layout: wave w owns tile rows 8w .. 8w+7 in two passes of 4 rows; a row is read by a QUARTER wave (16 lanes x 16 bytes
// = 128 elements per load instruction = one K stage), so the row amax is a reduction inside one 16-lane DPP row (four DPP exchanges,
// no LDS) and lane l's j-th chunk (8 elements) IS bytes 8 (l & 15) .. +8 of stage j's row: one ds_write_b64.
#include <atomic>
#include <cstdlib>
#include <type_traits>

#include "quant8_dev.h"
#include "sdnq_dev.h"

int sdnq_internal_take_prefetch(int64_t room, int threads, const uint8_t* pf_ptr[4], int pf_lines[4]);  // gemm.hip

namespace {

constexpr int BM = 64, BN = 128, BK = 128, NW = 8, NT = NW * 64;
constexpr int A_STAGE = BM * BK, B_STAGE = BN * BK;

struct AqParams {
    const float* sb;      // [N] weight row scales
    const void* bias;     // [N] or null
    void* out;            // [M][ldc]
    int64_t ldc;
    int bias_dtype;
    const uint8_t* pf_ptr[4];  // weight prefetch hosted by this launch (sdnq_hip_prefetch_hint)
    int pf_lines[4];
    unsigned long long* trace;  // lab: per-workgroup phase stamps (8 per workgroup), or null
#ifdef SDNQ_AQ_LAB
    int lab;  // lab build (timing only, results invalid): 1 no loop DMAs, 2 no fragment reads, 4 no MFMAs, 8 no barrier, 16 no quantization arithmetic
#endif
};

typedef unsigned short us2 __attribute__((ext_vector_type(2)));

template <int MM> struct AqMma;
template <> struct AqMma<SDNQ_MM_I8> {
    typedef v16i acc_t;
    typedef v4i frag_t;
    static constexpr int KB = 32;  // K bytes per MFMA
    static __device__ __forceinline__ frag_t load(const uint8_t* s, int r, int ks, int fgrp) {
        return *(const v4i*)(s + r * 128 + (((ks * 2 + fgrp) ^ ((r >> 1) & 7)) << 4));
    }
    static __device__ __forceinline__ void mma(acc_t& c, const frag_t& w, const frag_t& x) { c = __builtin_amdgcn_mfma_i32_32x32x32_i8(w, x, c, 0, 0, 0); }
    static __device__ __forceinline__ float tof(const acc_t& c, int i) { return (float)c[i]; }
};
template <> struct AqMma<SDNQ_MM_FP8> {
    typedef v16f acc_t;
    typedef v8i frag_t;
    static constexpr int KB = 64;
    static __device__ __forceinline__ frag_t load(const uint8_t* s, int r, int ks, int fgrp) {
        const int sw = (r >> 1) & 7;
        const v4i lo = *(const v4i*)(s + r * 128 + (((ks * 4 + fgrp * 2) ^ sw) << 4));
        const v4i hi = *(const v4i*)(s + r * 128 + (((ks * 4 + fgrp * 2 + 1) ^ sw) << 4));
        return (v8i){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    }
    static __device__ __forceinline__ void mma(acc_t& c, const frag_t& w, const frag_t& x) {
        c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(w, x, c, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    }
    static __device__ __forceinline__ float tof(const acc_t& c, int i) { return c[i]; }
};

template <int N, int I = 0, typename F> __device__ __forceinline__ void aq_static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        aq_static_for<N, I + 1>(f);
    }
}
template <int N> __device__ __forceinline__ void aq_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

#define AQ_TRACE(slot)                                                                                                   \
    do {                                                                                                                 \
        if (p.trace != nullptr && threadIdx.x == 0 && blockIdx.x < 1024) p.trace[blockIdx.x * 8 + (slot)] = __builtin_amdgcn_s_memtime(); \
    } while (0)

// X_T: activation dtype (bf16 / f16) = output dtype; NJ: K stages held (K <= 128 NJ); NSB: weight ring depth
template <int X_T, int MM, bool HAS_BIAS, int NJ, int NSB>
__global__ __launch_bounds__(NT) void linear_aq_kernel(const uint16_t* __restrict__ x, const uint8_t* __restrict__ w, int ldx, int ldb, int M, int N,
                                                       int K, int tiles_m, int tiles_n, int group_m, AqParams p_) {
    SDNQ_KERNARGS_NOW("s"(x), "s"(w), "s"(ldx), "s"(ldb), "s"(M), "s"(N), "s"(K), "s"(tiles_m), "s"(tiles_n), "s"(group_m));
    const AqParams& p = p_;
    typedef AqMma<MM> MT;
    extern __shared__ __attribute__((aligned(1024))) uint8_t lds[];
    uint8_t* const ldsA = lds;                       // [NJ][64 rows][128 B] quantized activation rows, resident
    uint8_t* const ldsB = lds + NJ * A_STAGE;        // [NSB][128 rows][128 B] weight ring
    float* const s_sb = (float*)(ldsB + NSB * B_STAGE);
    float* const s_bias = s_sb + BN;
    float* const s_xs = s_bias + BN;                 // [64] activation row scales

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nwg = tiles_m * tiles_n;
    int bid = blockIdx.x;
    if (bid >= nwg) {  // hosted weight prefetch: one dword of every 128-byte line of the next layers' weights (gemm.hip, launch_one)
        const int t = (bid - nwg) * NT + tid, stride = ((int)gridDim.x - nwg) * NT;
#pragma nounroll
        for (int r = 0; r < 4; ++r) {
            const uint8_t* base = p.pf_ptr[r];
            const int lines = p.pf_lines[r];
            for (int i = t; i < lines; i += stride) {
                int v;
                asm volatile("global_load_dword %0, %1, off" : "=v"(v) : "v"(base + (int64_t)i * 128) : "memory");
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }
    const unsigned long long t_entry = __builtin_amdgcn_s_memtime();  // (stored with stamp 1: nothing waits for the trace pointer before the loads go out)
    {   // block b runs on XCD b % 8 (private L2 each): give every XCD a contiguous range of the tile sequence
        const int q = nwg / 8, r = nwg % 8, xcd = bid % 8, j = bid / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    }
    // ... and walk the sequence in groups of `group_m` row blocks, m fastest (gemm.hip's order): the ~20 tiles an XCD holds then share few
    // weight blocks (the operand the K loop streams: its DMAs mostly hit in L2) and read more distinct activation rows, all of them
    // requested at once up front.  n fastest -- every column tile of two row blocks per XCD -- ran the K loop at 850 cycles per stage
    // against 730 for the two-operand loop of gemm.hip: half of its weight pieces missed L2.
    int tile_m, tile_n;
    {
        const int per_group = group_m * tiles_n;
        const int gid = bid / per_group, first_m = gid * group_m;
        const int gsz = (tiles_m - first_m) < group_m ? (tiles_m - first_m) : group_m;
        const int in_g = bid - gid * per_group;
        tile_n = in_g / gsz;
        tile_m = first_m + in_g - tile_n * gsz;
    }
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int nk = K / BK;  // (launcher: K % 128 == 0, K <= 128 NJ)
    const int m_rows = (M - m0) < BM ? (M - m0) : BM, n_lim = (N - n0) < BN ? (N - n0) : BN;

    // ---- activation rows: every load of the wave's 8 rows in flight before anything else ---------------------------------------------
    const auto rsX = SDNQ_MAKE_RSRC_N((const uint8_t*)x + (int64_t)m0 * ldx * 2, ((int64_t)(m_rows - 1) * ldx + K) * 2);
    v4i xr[2][NJ];
    int qrow[2];
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
        qrow[ps] = wave * 8 + ps * 4 + (lane >> 4);
        const int rc = qrow[ps] < m_rows ? qrow[ps] : m_rows - 1;  // rows past M: computed on valid memory, never stored
        const int vo = rc * ldx * 2 + (lane & 15) * 16;
#pragma unroll
        for (int j = 0; j < NJ; ++j) xr[ps][j] = SDNQ_BUF_LOAD16(rsX, vo, j < nk ? j * 256 : 0x40000000);  // stages past K: out of range = zeros
    }
    // ---- weight ring prologue: piece = 8 rows x 128 B, lane l -> row l / 8, physical chunk l % 8 (source chunk = swizzle-inverse);
    // wave w owns pieces w and w + 8 of every stage; constant per-lane offsets, the K advance in the scalar operand
    const auto rsB = SDNQ_MAKE_RSRC_N(w + (int64_t)n0 * ldb, (int64_t)(n_lim - 1) * ldb + K);
    int voB[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int r = (wave + u * NW) * 8 + (lane >> 3);
        const int rc = r < n_lim ? r : n_lim - 1;
        voB[u] = rc * ldb + (((lane & 7) ^ ((r >> 1) & 7)) << 4);
    }
    auto issueB = [&](int st, int slot) {  // stages past the end re-fetch stage 0 (never consumed; keeps the counted vmcnt a constant)
        const int s = st < nk ? st : 0;
#pragma unroll
        for (int u = 0; u < 2; ++u) SDNQ_DMA16(rsB, ldsB + slot * B_STAGE + (wave + u * NW) * 1024, voB[u], s * BK);
    };
    constexpr int AHEAD = NSB - 1;
#pragma unroll
    for (int s = 0; s < AHEAD; ++s) issueB(s, s);
    __builtin_amdgcn_sched_barrier(0);
    SDNQ_KERNARGS_NOW("s"(p_.sb), "s"(p_.bias), "s"(p_.out), "s"(p_.ldc), "s"(p_.bias_dtype), "s"(p_.trace));
    if (p.trace != nullptr && tid == 0 && blockIdx.x < 1024) p.trace[blockIdx.x * 8] = t_entry;
    AQ_TRACE(1);
    // per-channel epilogue vectors: requested now (behind the rows and the ring prologue), parked in LDS after the quantization
    float ev_sb = 0.0f;
    u32 ev_bias = 0;  // raw bits: converted when parked (a conversion here would wait for every load in flight)
    if (tid < BN) {
        const int64_t gi = n0 + (tid < n_lim ? tid : n_lim - 1);
        ev_sb = p.sb[gi];
        if constexpr (HAS_BIAS) {
            if (p.bias_dtype == SDNQ_F32) ev_bias = ((const u32*)p.bias)[gi];
            else ev_bias = ((const uint16_t*)p.bias)[gi];
        }
    }

    // ---- row quantization: amax inside the 16-lane row group, IEEE scale, lean codes -> the resident LDS image -------------------------
    constexpr float QMAX = (MM == SDNQ_MM_I8) ? 127.0f : 448.0f;
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
        // loads return in order: pass 0 has landed when at most the NJ loads of pass 1 and this wave's 2 AHEAD ring pieces are outstanding
        if (ps == 0) aq_wait_vmcnt<NJ + 2 * AHEAD>();
        else aq_wait_vmcnt<2 * AHEAD>();
        // |x| of 16-bit floats orders like the unsigned integer of its low 15 bits: packed integer max, two elements per instruction
        us2 mx = {0, 0};
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int d = 0; d < 4; ++d)
                mx = __builtin_elementwise_max(mx, __builtin_bit_cast(us2, (u32)xr[ps][j][d] & 0x7fff7fffu));
        u32 m32 = __builtin_bit_cast(u32, mx);
#pragma unroll
        for (int s = 1; s <= 8; s <<= 1) {
            const u32 o = (u32)lane_xor_i32((int)m32, s);
            m32 = __builtin_bit_cast(u32, __builtin_elementwise_max(__builtin_bit_cast(us2, m32), __builtin_bit_cast(us2, o)));
        }
        const u32 mb = (m32 & 0xffffu) > (m32 >> 16) ? (m32 & 0xffffu) : (m32 >> 16);
        const float amax = X_T == SDNQ_BF16 ? __uint_as_float(mb << 16) : f16_bits_to_f32((uint16_t)mb);
        const float scale = amax / QMAX;  // get_scale_symmetric, quant_utils.py:23-24
        RowDiv rd;
        rd.set(scale);
        if ((lane & 15) == 0) s_xs[qrow[ps]] = scale;
        const int r = qrow[ps];
        uint8_t* dst = ldsA + r * 128 + ((lane & 1) << 3) + ((((lane & 15) >> 1) ^ ((r >> 1) & 7)) << 4);
        // one test per pass, not per chunk: every row of the wave on the lean path (scale finite and ordinary -- anything but an all-zero
        // / inf / nan row), else the general path for all four rows (same codes: quant8 takes the same lean branch lane by lane)
        const bool all_fast = __builtin_amdgcn_ballot_w64(!rd.fast) == 0;
#ifdef SDNQ_AQ_LAB
        if (p.lab & 16) continue;
#endif
        if (all_fast) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                float v[8];
                const v4i t = xr[ps][j];
                Vec16<X_T>::unpack(make_uint4((u32)t[0], (u32)t[1], (u32)t[2], (u32)t[3]), v);
                const uint2 q = quant8_fast<MM>(v, rd);
                *(v2i*)(dst + j * A_STAGE) = (v2i){(int)q.x, (int)q.y};  // (ext-vector store: a HIP-struct store drains the LDS-DMAs first)
            }
        } else {  // (unrolled as well: a run-time index into the row registers would put them in scratch memory)
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                float v[8];
                const v4i t = xr[ps][j];
                Vec16<X_T>::unpack(make_uint4((u32)t[0], (u32)t[1], (u32)t[2], (u32)t[3]), v);
                int isum = 0;
                const uint2 q = quant8<MM>(v, rd, isum);
                *(v2i*)(dst + j * A_STAGE) = (v2i){(int)q.x, (int)q.y};
            }
        }
    }
    if (tid < BN) {
        s_sb[tid] = ev_sb;
        if constexpr (HAS_BIAS)
            s_bias[tid] = p.bias_dtype == SDNQ_F32 ? __uint_as_float(ev_bias) : (p.bias_dtype == SDNQ_BF16 ? __uint_as_float(ev_bias << 16) : f16_bits_to_f32((uint16_t)ev_bias));
    }
    AQ_TRACE(2);

    // ---- K loop: weight stages through the ring, activation fragments from the resident image ----------------------------------------
    typename MT::acc_t acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0;
    const int wm = wave >> 2, wn = wave & 3;
    const int frow = lane & 31, fgrp = lane >> 5;
    constexpr int KS = BK / MT::KB;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's codes and scales are in LDS before the first barrier lets anyone read them
    // Software pipeline one STAGE deep (the LD_OV form of gemm.hip, profiles/r05_overlapped_ring_lab.txt): the fragments of stage kt sit in
    // registers when its barrier falls, so the MFMAs start at once; dealt out between them are the fragment reads of stage kt + 1 (which that
    // barrier published) and the DMA pieces of stage kt + NSB into the slot stage kt has just vacated -- NSB - 1 stages in flight behind the
    // landed one.  Two register sets, so the loop is unrolled by two.
    typename MT::frag_t fa[2][KS], fb[2][KS];
    auto read_stage = [&](auto setc, auto ksc, int st, int slot) {
        constexpr int sx = decltype(setc)::value, ks = decltype(ksc)::value;
        fa[sx][ks] = MT::load(ldsA + (st < nk ? st : 0) * A_STAGE, wm * 32 + frow, ks, fgrp);
        fb[sx][ks] = MT::load(ldsB + slot * B_STAGE, wn * 32 + frow, ks, fgrp);
    };
    aq_wait_vmcnt<(AHEAD - 1) * 2>();  // stage 0 (this wave's pieces; the barrier makes it everybody's)
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    issueB(AHEAD, AHEAD);              // the ring's last free slot
    aq_static_for<KS>([&](auto ksc) { read_stage(std::integral_constant<int, 0>{}, ksc, 0, 0); });
    int slot_c = 0;  // ring slot of the stage whose fragments are in registers
#ifdef SDNQ_AQ_LAB
    const int lab = __builtin_amdgcn_readfirstlane(p.lab);
#else
    constexpr int lab = 0;  // (run-time switches inside the K loop cost it 35 %: a lab build only)
#endif
    auto half = [&](auto setc, int kt) {
        constexpr int sx = decltype(setc)::value;
        aq_wait_vmcnt<(NSB - 2) * 2>();                      // this wave's pieces of stage kt + 1 have landed
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // ... and its reads of stage kt have retired: the slot may be refilled
        __builtin_amdgcn_sched_barrier(0);
        if (!(lab & 8)) __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        const int slot_free = slot_c;
        slot_c = (slot_c + 1 == NSB) ? 0 : slot_c + 1;
        // MFMA ks of stage kt, then read ks of stage kt + 1 (the last sub-step's reads retire under the next barrier's wait; KS <= 4)
        aq_static_for<KS>([&](auto ksc) {
            constexpr int ks = decltype(ksc)::value;
            if (!(lab & 4)) MT::mma(acc, fb[sx][ks], fa[sx][ks]);
            __builtin_amdgcn_sched_barrier(0);
            if (!(lab & 2)) read_stage(std::integral_constant<int, sx ^ 1>{}, ksc, kt + 1, slot_c);
            if constexpr (ks == 0) { if (!(lab & 1)) issueB(kt + NSB, slot_free); }
            __builtin_amdgcn_sched_barrier(0);
        });
    };
#pragma nounroll
    for (int kt = 0; kt < nk; kt += 2) {
        half(std::integral_constant<int, 0>{}, kt);
        if (kt + 1 < nk) half(std::integral_constant<int, 1>{}, kt + 1);
    }
    AQ_TRACE(3);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the trailing filler DMAs target the ring; the staging area below is the A image
    __syncthreads();                                  // every wave is done with the A image before it becomes the output staging area
    AQ_TRACE(4);

    // ---- epilogue in the MFMA register layout: lane owns row wm*32 + (lane & 31), channels wn*32 + (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5);
    // out = cast(fma(f32(acc) * xs, ws, bias)) (kernel_wrappers.py:132-144); final 16-bit values leave through LDS as 16-byte row pieces
    constexpr int OUT_ROW = BN * 2 + 16;
    static_assert(BM * OUT_ROW <= NJ * A_STAGE + NSB * B_STAGE, "output staging fits the operand area");
    {
        const int ml = wm * 32 + frow;
        const float sa = s_xs[ml];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int nl0 = wn * 32 + 8 * q + 4 * fgrp;
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float vv = MT::tof(acc, 4 * q + e) * sa;
                if constexpr (HAS_BIAS) o[e] = fmaf(vv, s_sb[nl0 + e], s_bias[nl0 + e]);
                else o[e] = vv * s_sb[nl0 + e];
            }
            *(uint2*)(lds + ml * OUT_ROW + nl0 * 2) = make_uint2(pack2<X_T>(o[0], o[1]), pack2<X_T>(o[2], o[3]));
        }
    }
    __syncthreads();
    AQ_TRACE(5);
    constexpr int PPR = BN * 2 / 16;  // 16-byte pieces per output row
#pragma unroll
    for (int v = tid; v < BM * PPR; v += NT) {
        const int r = v / PPR, c = v % PPR;
        if (r >= m_rows || c * 8 >= n_lim) continue;  // N % 8 == 0: a piece never straddles N
        const uint4 val = *(const uint4*)(lds + r * OUT_ROW + c * 16);
        __builtin_nontemporal_store((v4i){(int)val.x, (int)val.y, (int)val.z, (int)val.w},
                                    (v4i*)((uint8_t*)p.out + ((int64_t)(m0 + r) * p.ldc + n0 + c * 8) * 2));
    }
    AQ_TRACE(6);
}

// which problems take the one-launch route.  Costs that grow with it: tiles_n workgroups repeat the quantization of a row block (VALU
// time on the critical path of every tile), and a tile keeps its rows resident (K <= 1280).  Wins where the row-quantization launch
// is a large part of the pair: the one-round projections of the bs = 1 steps.
inline int64_t aq_env(const char* name, int64_t dflt) {
    const char* e = getenv(name);
    return e ? atoll(e) : dflt;
}

inline int aq_cu_count() {  // of the CURRENT device (a process may drive different parts / partitions)
    static std::atomic<int> cus[64];
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    v = cus[dev].load(std::memory_order_relaxed);
    if (v == 0) {
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        cus[dev].store(v, std::memory_order_relaxed);
    }
    return v;
}

std::atomic<unsigned long long*> g_aq_trace{nullptr};

template <int X_T, int MM, bool HAS_BIAS, int NJ, int NSB>
int launch_aq(const void* x, const void* w, int64_t ldx, int64_t m, int64_t n, int64_t k, AqParams p, hipStream_t s) {
    constexpr int LDS_BYTES = NJ * A_STAGE + NSB * B_STAGE + (2 * BN + BM) * 4;
    static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
    auto kern = linear_aq_kernel<X_T, MM, HAS_BIAS, NJ, NSB>;
    // (the attribute belongs to the function ON ONE DEVICE: a process that drives several GPUs sets it once per device, not once)
    static std::atomic<uint64_t> attr_devices{0};
    if (LDS_BYTES > 64 * 1024) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) return SDNQ_ERR_LAUNCH;
        const uint64_t bit = 1ull << (dev & 63);
        if (!(attr_devices.load(std::memory_order_acquire) & bit)) {
            if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess) return SDNQ_ERR_LAUNCH;
            attr_devices.fetch_or(bit, std::memory_order_release);
        }
    }
    const int tiles_m = (int)((m + BM - 1) / BM), tiles_n = (int)((n + BN - 1) / BN);
    const int64_t tiles = (int64_t)tiles_m * tiles_n;
    constexpr int WG_PER_CU = (160 * 1024) / LDS_BYTES;
    p.trace = g_aq_trace.load(std::memory_order_relaxed);
#ifdef SDNQ_AQ_LAB
    { const char* e = getenv("SDNQ_HIP_AQ_LAB"); p.lab = e ? atoi(e) : 0; }
#endif
    static const int gm_env = (int)aq_env("SDNQ_HIP_FUSED_ROWQUANT_GROUP_M", 8);  // tuning aid (1 = n fastest)
    const int group_m = gm_env < 1 ? 1 : (gm_env > tiles_m ? tiles_m : gm_env);
    const int pf_wgs = sdnq_internal_take_prefetch((int64_t)WG_PER_CU * aq_cu_count() - tiles, NT, p.pf_ptr, p.pf_lines);
    hipLaunchKernelGGL(kern, dim3((unsigned)(tiles + pf_wgs)), dim3(NT), LDS_BYTES, s, (const uint16_t*)x, (const uint8_t*)w, (int)ldx, (int)k, (int)m,
                       (int)n, (int)k, tiles_m, tiles_n, group_m, p);
    SDNQ_CHECK_LAUNCH();
    return SDNQ_OK;
}

}  // namespace

extern "C" int sdnq_hip_linear_w8a8_fused_supported(int mm_dtype, int x_dtype, int out_dtype, int64_t m, int64_t n, int64_t k) {
    static const int64_t on = aq_env("SDNQ_HIP_FUSED_ROWQUANT", 1), max_tn = aq_env("SDNQ_HIP_FUSED_ROWQUANT_MAX_TILES_N", 12),
                         min_m = aq_env("SDNQ_HIP_FUSED_ROWQUANT_MIN_M", 33), min_k = aq_env("SDNQ_HIP_FUSED_ROWQUANT_MIN_K", 128),
                         max_k = aq_env("SDNQ_HIP_FUSED_ROWQUANT_MAX_K", 1280);
    // fp8: built and bit-identical, off by default -- its quantization is costlier per element (sign fix-up, clamp, two converts per four
    // codes) and the SDXL fp8 step lost 2 % with it (7.08 -> 7.23 ms, profiles/r05_fused_rowquant_gemm.txt)
    static const int64_t fp8_on = aq_env("SDNQ_HIP_FUSED_ROWQUANT_FP8", 0);
    if (!on) return 0;
    if (mm_dtype != SDNQ_MM_I8 && !(mm_dtype == SDNQ_MM_FP8 && fp8_on)) return 0;
    if ((x_dtype != SDNQ_BF16 && x_dtype != SDNQ_F16) || out_dtype != x_dtype) return 0;
    if (m < min_m || n <= 0 || (n % 8) != 0 || k <= 0 || (k % 128) != 0 || k > 1280) return 0;
    const int64_t tiles_m = (m + BM - 1) / BM, tiles_n = (n + BN - 1) / BN;
    if (tiles_n > max_tn || k < min_k || k > max_k) return 0;
    if (tiles_m * tiles_n > (int64_t)aq_cu_count()) return 0;  // one round, one workgroup per CU
    return 1;
}

extern "C" int sdnq_hip_linear_w8a8_fused(int mm_dtype, const void* x, int x_dtype, int64_t m, int64_t k, int64_t ldx, const void* b,
                                          const float* sb, const void* bias, int bias_dtype, void* out, int out_dtype, int64_t n,
                                          sdnq_stream_t stream) {
    if (!x || !b || !sb || !out) return SDNQ_ERR_NULL;
    if (mm_dtype != SDNQ_MM_I8 && mm_dtype != SDNQ_MM_FP8) return SDNQ_ERR_DTYPE;
    if ((x_dtype != SDNQ_BF16 && x_dtype != SDNQ_F16) || out_dtype != x_dtype) return SDNQ_ERR_UNSUPPORTED;
    if (bias && (bias_dtype < 0 || bias_dtype > 2)) return SDNQ_ERR_DTYPE;
    if (m <= 0 || n <= 0 || k <= 0 || ldx < k || (n % 8) != 0) return SDNQ_ERR_SHAPE;
    if ((k % 128) != 0 || k > 1280) return SDNQ_ERR_UNSUPPORTED;
    if (m > 0x7fffffffll / 2 || n > 0x7fffffffll || ldx * 2 * 64 > 0x7fffffffll || k * 128 > 0x7fffffffll) return SDNQ_ERR_SHAPE;
    if (((uintptr_t)x % 16) || ((uintptr_t)b % 16) || ((uintptr_t)out % 16) || ((ldx * 2) % 16)) return SDNQ_ERR_ALIGN;
    AqParams p{};
    p.sb = sb; p.bias = bias; p.out = out; p.ldc = n; p.bias_dtype = bias_dtype;
    hipStream_t s = (hipStream_t)stream;
#define AQ_NJ(XT, MMV, HB) (k <= 640 ? launch_aq<XT, MMV, HB, 5, 4>(x, b, ldx, m, n, k, p, s) : launch_aq<XT, MMV, HB, 10, 4>(x, b, ldx, m, n, k, p, s))
#define AQ_B(XT, MMV) (bias ? AQ_NJ(XT, MMV, true) : AQ_NJ(XT, MMV, false))
#define AQ_X(MMV) (x_dtype == SDNQ_BF16 ? AQ_B(SDNQ_BF16, MMV) : AQ_B(SDNQ_F16, MMV))
    return mm_dtype == SDNQ_MM_I8 ? AQ_X(SDNQ_MM_I8) : AQ_X(SDNQ_MM_FP8);
#undef AQ_X
#undef AQ_B
#undef AQ_NJ
}

// lab: phase stamps of the one-launch Linear (device buffer of 1024 x 8 uint64, or null to stop).  A C++ symbol internal to the library
// (tools/aq_lab.py binds its mangled name), not part of the C ABI of include/sdnq_hip.h
void sdnq_internal_aq_trace(unsigned long long* device_buf) { g_aq_trace.store(device_buf, std::memory_order_relaxed); }
