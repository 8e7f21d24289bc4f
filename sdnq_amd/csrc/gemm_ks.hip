// Scaled int8 matmul on 64 x 80 output tiles with an IN-WORKGROUP K split (round 6):  out = cast(fma(f32(A.B^T) * sa[m], sb[n], bias))
//
// The same contract as gemm.hip's gemm_kernel<MM_I8> (reference: kernels/triton_scaled_mm.py:127-236, tile loop :194-205; eager twin
// kernel_wrappers.py:132-144) for the problems of the bs = 1 diffusion steps whose outputs do not fill the chip with 64 x 128 tiles:
// 1024 x 1280 outputs are 160 such tiles on 256 CUs; 64 x 80 tiles cut them into exactly 256 workgroups with 25 % fewer operand bytes
// per CU (184 instead of 245 KB at K = 1280).  Four waves per workgroup could not feed that tile (rounds 4 / 5: 9.5 vs 7.9 us), so the
// workgroup has EIGHT waves in two K groups: waves 0-3 and 4-7 both own the whole 64 x 80 tile (wave w & 3 = 16 rows x 80 channels on
// v_mfma_i32_16x16x64_i8) and take ALTERNATE 128-byte K stages of one shared ring.  The int32 partial sums of the two groups meet
// through LDS after the loop (exact integer addition: the result is bit-identical to every other tile configuration) -- nothing is
// exchanged through memory, no tickets, no second launch.
//
// What else differs from gemm_kernel, all of it aimed at the fixed cost of a short launch and at cold operands:
//   * the ring is 8 stages deep (147 KB: one workgroup per CU by construction) and the prologue requests up to SIX stages before anything
//     else happens -- at K = 1280 that is 60 % of the operand bytes of the tile in flight behind one memory latency;
//   * per stage the WEIGHT pieces are issued first (the weight operand never depends on the producer of the activation);
//   * no filler DMAs past the end of K (the counted waits are computed at run time instead): 2 of 12 stages of traffic less at K = 1280;
//   * the per-channel / per-row epilogue vectors (sb, bias, sa) arrive by LDS-DMA too, queued BEHIND the prologue stages: nobody
//     waits for them before the epilogue, and no wave ever executes `s_waitcnt vmcnt(0)` in front of the K loop;
//   * the launch fills every CU, so there is no room for prefetch workgroups: the weight prefetch of the NEXT launches
//     (sdnq_hip_prefetch_hint) is done by the tile workgroups themselves -- four 4-byte LDS-DMA loads per thread into a scratch
//     area, issued after this tile's last stage so that no counted wait ever sits behind an HBM miss.
//
// Where it stands (profiles/r06_ksplit_lab.txt, r06_fill_lab.txt): bit-identical; on operands that come from HBM it is 10-20 % faster than
// the 64 x 128 tile (1024 x 1280 x 5120: 19.0 vs 23.6 us, x 2560: 11.6 vs 12.6 -- six stages in flight hide the latency the 3-stage
// ring exposes); on warm operands it is equal at K = 5120 and 9 % SLOWER at K = 1280 (7.6 vs 7.0 us), and in the SDXL step it loses
// (see sdnq_internal_ks_preferred below), so the heuristics do not pick it: tile id 28, SDNQ_HIP_KSPLIT=1.  The premise -- a CU's own
// fill path bounds the launch, so fewer bytes per CU on more CUs win -- does not hold on 256 CUs: tools/micro/fill_lab.hip (pure
// fetch of this kernel's access pattern, nothing consumed) sustains 28.7 B/clk/CU with 8 issuing waves and 34 with 16 (15.5 / 18.3 TB/s
// chip-wide L2 -> LDS), where 160 workgroups of 64 x 128 get 36 B/clk each: the launch is bound by what the eight L2s deliver to all
// CUs together, and 64 x 80 tiles need 47 MB of that where 64 x 128 tiles need 39 MB.  Two other schedules of the same tile were built
// and measured slower (same file): the K groups alternating fetch and multiply steps (four issuing waves: 24 B/clk), and sixteen
// waves with eight dedicated fetch waves (the first stages then land behind the whole prologue: +2 us).
#include <atomic>
#include <cstdlib>
#include <type_traits>

#include "sdnq_dev.h"

int sdnq_internal_take_prefetch(int64_t room, int threads, const uint8_t* pf_ptr[4], int pf_lines[4]);  // gemm.hip

namespace {

constexpr int BM = 64, BN = 80, BK = 128, NW = 8, NT = NW * 64, NS = 8;
constexpr int A_STAGE = BM * BK, B_STAGE = BN * BK, STAGE = A_STAGE + B_STAGE;  // 8 + 10 KiB
constexpr int RING = NS * STAGE;
constexpr int VEC_OFF = RING;  // 256-byte areas: [sb 0-63][sb 64-79][bias ...][bias (f32) 64-79][sa 0-63][-][scratch]
constexpr int LDS_BYTES = RING + 7 * 256;
static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");

struct KsParams {
    const float* sa;      // [M] activation row scales
    const float* sb;      // [N] weight row scales
    const void* bias;     // [N] or null
    void* out;            // [M][ldc]
    int64_t ldc;
    int bias_dtype;
    const uint8_t* pf_ptr[4];  // weight prefetch carried by this launch (128-byte-aligned bases)
    int pf_lines[4];
    unsigned long long* trace;  // lab: per-workgroup phase stamps (8 per workgroup), or null
    int pro;                    // stages the prologue requests (2 .. 6); the first iterations top the ring up, three stages at a time
    int pf_early;               // the prefetch loads go out right behind the prologue (else: behind the tile's last stage)
};

template <int N> __device__ __forceinline__ void ks_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// s_waitcnt vmcnt(n) with the largest immediate of a sorted candidate table that does not exceed `allow` (wave-uniform; waiting for
// more than necessary is always safe): a balanced tree of scalar compares, four levels for sixteen candidates
struct KsWait24 { static constexpr int n = 24; static constexpr int c[24] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23}; };
template <typename T, int LO, int HI> __device__ __forceinline__ void ks_wait_tree(int allow) {
    if constexpr (LO == HI) {
        ks_wait_vmcnt<T::c[LO]>();
    } else {
        constexpr int MID = (LO + HI + 1) / 2;
        if (allow >= T::c[MID]) ks_wait_tree<T, MID, HI>(allow);
        else ks_wait_tree<T, LO, MID - 1>(allow);
    }
}

#define KS_TRACE(slot)                                                                                                   \
    do {                                                                                                                 \
        if (p.trace != nullptr && threadIdx.x == 0 && blockIdx.x < 1024) p.trace[blockIdx.x * 8 + (slot)] = __builtin_amdgcn_s_memtime(); \
    } while (0)

// OUT_T: bf16 / f16 output
template <int OUT_T, bool HAS_BIAS>
__global__ __launch_bounds__(NT) void gemm_ks_kernel(const uint8_t* __restrict__ a, const uint8_t* __restrict__ b, int lda, int ldb, int M, int N, int K,
                                                     int tiles_m, int tiles_n, int group_m, KsParams p_) {
    SDNQ_KERNARGS_NOW("s"(a), "s"(b), "s"(lda), "s"(ldb), "s"(M), "s"(N), "s"(K), "s"(tiles_m), "s"(tiles_n), "s"(group_m), "s"(p_.pro), "s"(p_.pf_early));
    const KsParams& p = p_;
    extern __shared__ __attribute__((aligned(1024))) uint8_t lds[];
    const unsigned long long t_entry = __builtin_amdgcn_s_memtime();
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nwg = tiles_m * tiles_n;
    int bid = blockIdx.x;
    {   // block b runs on XCD b % 8 (private L2 each): every XCD walks a contiguous range of the tile sequence ...
        const int q = nwg / 8, r = nwg % 8, xcd = bid % 8, j = bid / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    }
    int tile_m, tile_n;
    {   // ... in groups of `group_m` row blocks, m fastest (gemm.hip's order): the 32 tiles of an XCD share 8 row blocks x 4 weight blocks
        const int per_group = group_m * tiles_n;
        const int gid = bid / per_group, first_m = gid * group_m;
        const int gsz = (tiles_m - first_m) < group_m ? (tiles_m - first_m) : group_m;
        const int in_g = bid - gid * per_group;
        tile_n = in_g / gsz;
        tile_m = first_m + in_g - tile_n * gsz;
    }
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int nk = K / BK;  // (launcher: K % 128 == 0)
    const int m_rows = (M - m0) < BM ? (M - m0) : BM, n_lim = (N - n0) < BN ? (N - n0) : BN;

    // ---- LDS-DMA assignment: a piece = 8 tile rows x 128 B (lane l -> row l / 8, physical chunk l % 8; the source chunk is the
    // swizzle-inverse so that LDS stays lane-linear).  A stage is 8 activation + 10 weight pieces: wave w owns activation piece w,
    // weight piece w and -- waves 0 and 1 -- weight piece 8 + w.  Constant per-lane offsets, the K advance in the scalar operand.
    // Rows past the tile's valid rows are clamped (computed on valid memory, never stored).
    const auto rsA = SDNQ_MAKE_RSRC_N(a + (int64_t)m0 * lda, (int64_t)(m_rows - 1) * lda + K);
    const auto rsB = SDNQ_MAKE_RSRC_N(b + (int64_t)n0 * ldb, (int64_t)(n_lim - 1) * ldb + K);
    const bool big = wave < 2;  // wave-uniform: three pieces per stage instead of two
    int voA, voB0, voB1;
    {
        const int r = wave * 8 + (lane >> 3), sw = ((lane & 7) ^ ((r >> 1) & 7)) << 4;  // (r + 64 has the same swizzle: 64 is a multiple of 16)
        voA = (r < m_rows ? r : m_rows - 1) * lda + sw;
        voB0 = (r < n_lim ? r : n_lim - 1) * ldb + sw;
        voB1 = (r + 64 < n_lim ? r + 64 : n_lim - 1) * ldb + sw;
    }
    auto issue_stage = [&](int st) {  // K stage st into ring slot st % NS: weight pieces first
        uint8_t* slot = lds + (st & (NS - 1)) * STAGE;
        SDNQ_DMA16(rsB, slot + A_STAGE + wave * 1024, voB0, st * BK);
        if (big) SDNQ_DMA16(rsB, slot + A_STAGE + (8 + wave) * 1024, voB1, st * BK);
        SDNQ_DMA16(rsA, slot + wave * 1024, voA, st * BK);
    };
    const int npro = nk < p_.pro ? nk : p_.pro;
    int issued = 0;  // stages requested so far (wave-uniform)
#pragma nounroll
    for (; issued < npro; ++issued) issue_stage(issued);
    __builtin_amdgcn_sched_barrier(0);
    SDNQ_KERNARGS_NOW("s"(p_.sa), "s"(p_.sb), "s"(p_.bias), "s"(p_.out), "s"(p_.ldc), "s"(p_.bias_dtype), "s"(p_.trace));
    // ---- epilogue vectors by LDS-DMA (4 bytes per lane, lane-linear dwords in LDS), ONE instruction per wave so that every wave's
    // queue has the same shape: wave 0 / 1: sb[0..63] / sb[64..79]; wave 2 (/ 3): the bias elements as raw bits (16-bit: 128 of them
    // fit one piece; f32: 64 + 16); wave 4: sa[0..63]; the others: an empty descriptor (no memory access) into a scratch area
    {
        const int bb = p.bias_dtype == SDNQ_F32 ? 4 : 2;  // bytes per bias element
        const uint8_t* src = a;                           // (a valid base for the empty descriptors)
        int nbytes = 0, area = 6;
        if (wave == 0) { src = (const uint8_t*)(p.sb + n0); nbytes = (n_lim < 64 ? n_lim : 64) * 4; area = 0; }
        else if (wave == 1 && n_lim > 64) { src = (const uint8_t*)(p.sb + n0 + 64); nbytes = (n_lim - 64) * 4; area = 1; }
        else if (wave == 2 && HAS_BIAS) { src = (const uint8_t*)p.bias + (int64_t)n0 * bb; nbytes = (bb == 2 ? n_lim : (n_lim < 64 ? n_lim : 64)) * bb; area = 2; }
        else if (wave == 3 && HAS_BIAS && bb == 4 && n_lim > 64) { src = (const uint8_t*)p.bias + (int64_t)(n0 + 64) * 4; nbytes = (n_lim - 64) * 4; area = 3; }
        else if (wave == 4) { src = (const uint8_t*)(p.sa + m0); nbytes = m_rows * 4; area = 4; }
        const auto rsV = SDNQ_MAKE_RSRC_N(src, nbytes);
        SDNQ_DMA4(rsV, lds + VEC_OFF + area * 256, lane * 4, 0);
    }
    // the weights of the NEXT launches, one dword per 128-byte line, nothing kept (an LDS-DMA into the scratch area, not a register
    // load: a destination register would have to stay reserved until the data is back, and the compiler waits for it wherever it moves
    // that register).  Past the range: no memory access.  Four operations in every wave's queue.
    auto prefetch_next = [&]() {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int64_t nb = p.pf_ptr[q] ? (int64_t)p.pf_lines[q] * 128 : 0;
            const auto rsP = SDNQ_MAKE_RSRC_N(p.pf_ptr[q] ? p.pf_ptr[q] : a, nb < 0x7fffffffll ? nb : 0x7fffffffll);
            SDNQ_DMA4(rsP, lds + VEC_OFF + 6 * 256, ((int)blockIdx.x * NT + tid) * 128, 0);
        }
    };
    const bool pf_early = p_.pf_early != 0;
    if (pf_early) prefetch_next();
    if (p.trace != nullptr && tid == 0 && blockIdx.x < 1024) p.trace[blockIdx.x * 8] = t_entry;
    KS_TRACE(1);

    // ---- K loop: iteration t = stages 2t (K group 0) and 2t + 1 (K group 1); ONE barrier per two stages ---------------------------------
    const int grp = wave >> 2, w4 = wave & 3;  // K group; rows 16 w4 .. +16 of the tile
    v4i acc[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) acc[i] = (v4i){0, 0, 0, 0};
    // fragment addresses: row (lane & 15) of a 16-row block, 16-byte chunk ks * 4 + (lane >> 4), XOR-swizzled by (row >> 1) & 7 -- the
    // same for every 16-row block (16 is a multiple of 16), so the blocks differ by immediates
    const int frow = lane & 15, fch = lane >> 4;
    const int fo0 = frow * 128 + (((0 + fch) ^ ((frow >> 1) & 7)) << 4), fo1 = frow * 128 + (((4 + fch) ^ ((frow >> 1) & 7)) << 4);
    const int PPW = big ? 3 : 2;
    bool pf_done = pf_early;  // (early: the four loads sit with the vector piece, behind the prologue's stages)
    const int niter = (nk + 1) >> 1;
#pragma nounroll
    for (int t = 0; t < niter; ++t) {
        // stages 2t, 2t + 1 have landed (this wave's pieces; the barrier makes it everybody's).  Operations younger than the last piece
        // of stage 2t + 1 in this wave's queue: the stages behind it (up to 4), the vector piece (behind the prologue), the prefetch loads
        int r = issued - 2 * t - 2;
        r = r < 0 ? 0 : r;
        ks_wait_tree<KsWait24, 0, 23>(r * PPW + (2 * t + 1 < npro ? (pf_early ? 5 : 1) : 0) + (pf_done && !pf_early ? 4 : 0));
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (t == 0) KS_TRACE(2);
        const int st = 2 * t + grp;
        v4i xa0, xa1, wf0[5], wf1[5];
        const bool mine = st < nk;  // (odd stage count: group 1 has no stage in the last iteration)
        if (mine) {
            const uint8_t* sA = lds + (st & (NS - 1)) * STAGE + w4 * 2048;
            const uint8_t* sB = lds + (st & (NS - 1)) * STAGE + A_STAGE;
            xa0 = *(const v4i*)(sA + fo0);
#pragma unroll
            for (int i = 0; i < 5; ++i) wf0[i] = *(const v4i*)(sB + i * 2048 + fo0);
            xa1 = *(const v4i*)(sA + fo1);
#pragma unroll
            for (int i = 0; i < 5; ++i) wf1[i] = *(const v4i*)(sB + i * 2048 + fo1);
        }
        __builtin_amdgcn_sched_barrier(0);
        // refill: the slots of the stages below 2t are free (every wave's reads of them retired in front of the barrier above), so stages
        // below 2t + 8 may be requested; at most three per iteration (a short prologue is topped up over the first iterations)
        {
            const int target = 2 * t + NS < nk ? 2 * t + NS : nk;
#pragma nounroll
            for (int c = 0; c < 3 && issued < target; ++c, ++issued) issue_stage(issued);
        }
        if (!pf_done && issued >= nk) { prefetch_next(); pf_done = true; }
        __builtin_amdgcn_sched_barrier(0);
        if (mine) {
#pragma unroll
            for (int i = 0; i < 5; ++i) acc[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wf0[i], xa0, acc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 5; ++i) acc[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wf1[i], xa1, acc[i], 0, 0, 0);
        }
    }
    KS_TRACE(3);
    // (the vector piece is older than every stage the loop waited for from iteration 3 on; a shorter loop waits here: everything but
    //  the four prefetch loads)
    if (niter <= 3) { if (pf_early) ks_wait_vmcnt<0>(); else ks_wait_vmcnt<4>(); }
    // ---- the two K groups meet: group 0 finishes channel blocks 0-2, group 1 blocks 3-4; each hands the other its partial sums of the
    // blocks it does not finish (lane-linear 16-byte stores: [wave of the group][block][lane])
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // every wave's fragment reads are done, every stage and every vector piece has landed: the ring is free
    __builtin_amdgcn_sched_barrier(0);
    uint8_t* const xch = lds;                       // [2 groups][4 waves][3 blocks][64 lanes] x 16 B = 24 KiB
    uint8_t* const ostage = lds + 32 * 1024;        // output staging [64 rows][OUT_ROW]
    constexpr int OUT_ROW = BN * 2 + 16;
    const float* s_sb = (const float*)(lds + VEC_OFF);
    const float* s_sa = (const float*)(lds + VEC_OFF + 1024);
    float* s_biasf = (float*)(lds + VEC_OFF + 1280);  // [80] f32 (areas 5-6: the scratch area's DMAs landed long ago)
    if (grp == 0) {
#pragma unroll
        for (int i = 3; i < 5; ++i) *(v4i*)(xch + ((w4 * 3 + (i - 3)) * 64 + lane) * 16) = acc[i];
    } else {
#pragma unroll
        for (int i = 0; i < 3; ++i) *(v4i*)(xch + 12288 + ((w4 * 3 + i) * 64 + lane) * 16) = acc[i];
    }
    if constexpr (HAS_BIAS) {  // the raw bias bits -> f32, once (a run-time dtype switch per output element is 5 branches per value)
        if (tid < BN) {
            const uint8_t* raw = lds + VEC_OFF + 512;
            float bv;
            if (p.bias_dtype == SDNQ_F32) bv = ((const float*)raw)[tid];
            else if (p.bias_dtype == SDNQ_BF16) bv = bf16_bits_to_f32(((const uint16_t*)raw)[tid]);
            else bv = f16_bits_to_f32(((const uint16_t*)raw)[tid]);
            s_biasf[tid] = bv;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    KS_TRACE(4);
    // epilogue in the MFMA register layout: lane owns row 16 w4 + (lane & 15), channels 16 i + 4 (lane >> 4) + 0..3;
    // out = cast(fma(f32(acc) * sa, sb, bias)) (kernel_wrappers.py:132-144); the final 16-bit values leave through LDS as 16-byte row pieces
    {
        const int ml = w4 * 16 + frow;
        const float sa = s_sa[ml];
        const int i0 = grp == 0 ? 0 : 3, ni = grp == 0 ? 3 : 2;
#pragma unroll
        for (int ii = 0; ii < 3; ++ii) {
            if (ii >= ni) break;
            const int i = i0 + ii;
            const v4i oth = *(const v4i*)(xch + (grp == 0 ? 12288 : 0) + ((w4 * 3 + ii) * 64 + lane) * 16);
            const v4i own = grp == 0 ? acc[ii] : acc[3 + (ii < 2 ? ii : 0)];
            const int nl0 = i * 16 + 4 * fch;
            const v4f sb4 = *(const v4f*)(s_sb + nl0);
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float vv = (float)(own[e] + oth[e]) * sa;  // int32 -> f32 (RNE above 2^24)
                if constexpr (HAS_BIAS) o[e] = fmaf(vv, sb4[e], s_biasf[nl0 + e]);
                else o[e] = vv * sb4[e];
            }
            // (ext-vector store: a HIP-struct store makes the compiler drain every outstanding vector-memory operation first)
            *(v2i*)(ostage + ml * OUT_ROW + nl0 * 2) = (v2i){(int)pack2<OUT_T>(o[0], o[1]), (int)pack2<OUT_T>(o[2], o[3])};
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    KS_TRACE(5);
    constexpr int PPR = BN * 2 / 16;  // 16-byte pieces per output row
#pragma unroll
    for (int v = tid; v < BM * PPR; v += NT) {
        const int r = v / PPR, c = v - r * PPR;
        if (r >= m_rows || c * 8 >= n_lim) continue;  // N % 8 == 0: a piece never straddles N
        const v4i val = *(const v4i*)(ostage + r * OUT_ROW + c * 16);
        __builtin_nontemporal_store(val,
                                    (v4i*)((uint8_t*)p.out + ((int64_t)(m0 + r) * p.ldc + n0 + c * 8) * 2));
    }
    KS_TRACE(6);
}

inline int ks_cu_count() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    static std::atomic<int> cus[64];
    int v = cus[dev].load(std::memory_order_relaxed);
    if (v == 0) {
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        cus[dev].store(v, std::memory_order_relaxed);
    }
    return v;
}

std::atomic<unsigned long long*> g_ks_trace{nullptr};

template <int OUT_T, bool HAS_BIAS>
int launch_ks(const void* a, const void* b, int64_t lda, int64_t ldb, int64_t m, int64_t n, int64_t k, KsParams p, hipStream_t s) {
    auto kern = gemm_ks_kernel<OUT_T, HAS_BIAS>;
    static std::atomic<uint64_t> attr_devices{0};  // (the attribute belongs to the function ON ONE DEVICE)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return SDNQ_ERR_LAUNCH;
    const uint64_t bit = 1ull << (dev & 63);
    if (!(attr_devices.load(std::memory_order_acquire) & bit)) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess) return SDNQ_ERR_LAUNCH;
        attr_devices.fetch_or(bit, std::memory_order_release);
    }
    const int tiles_m = (int)((m + BM - 1) / BM), tiles_n = (int)((n + BN - 1) / BN);
    const int64_t tiles = (int64_t)tiles_m * tiles_n;
    p.trace = g_ks_trace.load(std::memory_order_relaxed);
    static const int gm_env = [] { const char* e = getenv("SDNQ_HIP_KS_GROUP_M"); return e ? atoi(e) : 8; }();  // tuning aid
    const int group_m = gm_env < 1 ? 1 : (gm_env > tiles_m ? tiles_m : gm_env);
    static const int pro_env = [] { const char* e = getenv("SDNQ_HIP_KS_PRO"); const int v = e ? atoi(e) : 6; return v < 2 ? 2 : (v > 6 ? 6 : v); }();  // tuning aid
    p.pro = pro_env;
    static const int pf_env = [] { const char* e = getenv("SDNQ_HIP_KS_PF"); return e ? atoi(e) : 1; }();  // 0 off, 1 behind the last stage, 2 behind the prologue
    p.pf_early = pf_env == 2;
    // the pending prefetch hint rides inside the tile workgroups: thread t of the launch touches line t of each range
    sdnq_internal_take_prefetch(1 << 20, NT, p.pf_ptr, p.pf_lines);
    for (int r = 0; r < 4; ++r) {
        if (pf_env == 0 || tiles * NT * 128 > 0x7fffffffll) p.pf_lines[r] = 0;  // (32-bit line offsets)
        else if ((int64_t)p.pf_lines[r] > tiles * NT) p.pf_lines[r] = (int)(tiles * NT);
    }  // (what the launch can touch: 16 MB per range at 256 tiles)
    hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(NT), LDS_BYTES, s, (const uint8_t*)a, (const uint8_t*)b, (int)lda, (int)ldb, (int)m, (int)n, (int)k,
                       tiles_m, tiles_n, group_m, p);
    SDNQ_CHECK_LAUNCH();
    return SDNQ_OK;
}

}  // namespace

// Internal to the library (called by gemm.hip's launch_tiles; not part of the C ABI).
// Which problems may run on the K-split tile: int8, 16-bit output, plain epilogues, whole 128-byte K stages, 32-bit tile offsets.
bool sdnq_internal_ks_eligible(int64_t m, int64_t n, int64_t k, int64_t lda, int64_t ldb) {
    if (m <= 0 || n <= 0 || (n % 8) != 0 || k < BK || (k % BK) != 0) return false;
    if (m > 0x7fffffffll || n > 0x7fffffffll || 64 * lda + k > 0x7fffffffll || 80 * ldb + k > 0x7fffffffll) return false;
    return true;
}

// heuristic: the problems this tile exists for -- one round of 64 x 80 tiles that uses more CUs than one round of 64 x 128 tiles would.
// OFF by default (SDNQ_HIP_KSPLIT=1 turns it on): judged on the SDXL step it loses to the 64 x 128 tile wherever the cross-layer weight
// prefetch works (+0.19 ms over the 70 launches of 1024 x 1280 x 5120: that tile leaves 96 CUs to the prefetch workgroups of the NEXT
// launches, this one fills the chip and its in-tile prefetch burst costs more than it brings); it wins when weights come from HBM
// (same step without the prefetch: 7.85 -> 7.61 ms).  profiles/r06_ksplit_lab.txt
bool sdnq_internal_ks_preferred(int64_t m, int64_t n, int64_t k) {
    static const int on = [] { const char* e = getenv("SDNQ_HIP_KSPLIT"); return e ? atoi(e) : 0; }();
    if (!on) return false;
    const int64_t cus = ks_cu_count();
    const int64_t t80 = ((m + 63) / 64) * ((n + 79) / 80), t128 = ((m + 63) / 64) * ((n + 127) / 128);
    return (n % 80) == 0 && t80 <= cus && t80 * 5 >= cus * 4 && t128 < t80 && k >= 512;
}

int sdnq_internal_scaled_mm_ks(const void* a, const void* b, const float* sa, const float* sb, const void* bias, int bias_dtype, void* out,
                               int out_dtype, int64_t m, int64_t n, int64_t k, int64_t lda, int64_t ldb, int64_t ldc, hipStream_t s) {
    KsParams p{};
    p.sa = sa; p.sb = sb; p.bias = bias; p.out = out; p.ldc = ldc ? ldc : n; p.bias_dtype = bias_dtype;
    if (lda == 0) lda = k;
    if (ldb == 0) ldb = k;
    if (out_dtype == SDNQ_BF16) return bias ? launch_ks<SDNQ_BF16, true>(a, b, lda, ldb, m, n, k, p, s) : launch_ks<SDNQ_BF16, false>(a, b, lda, ldb, m, n, k, p, s);
    if (out_dtype == SDNQ_F16) return bias ? launch_ks<SDNQ_F16, true>(a, b, lda, ldb, m, n, k, p, s) : launch_ks<SDNQ_F16, false>(a, b, lda, ldb, m, n, k, p, s);
    return SDNQ_ERR_DTYPE;
}

void sdnq_internal_ks_trace(unsigned long long* device_buf) { g_ks_trace.store(device_buf, std::memory_order_relaxed); }
