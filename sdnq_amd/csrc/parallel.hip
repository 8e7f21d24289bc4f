// Tensor-parallel glue on gfx950: the re-assembly of a column-sharded Linear's output after the RCCL all-gather.
//
// Rank r computes y_r [M][w_r] (its N / W output channels); ncclAllGather moves ONE contiguous buffer per rank, so the gathered
// result is rank-major: g [W][M][wmax] (slabs padded to the widest one when N / 16 does not divide evenly).  The consumer wants
// row-major y [M][N] with N = sum w_r: out[m][start_r + c] = g[r][m][c].  The reference has no inference parallelism;
// round 2 did this with torch permute / reshape (even shards) or zeros + cat (uneven) -- several launches and, uneven, several
// passes.  Here: one HBM-bound pass, 16 bytes per lane, reads and writes coalesced along the channels of a rank.
#include <hip/hip_runtime.h>
#include <string.h>

#include "../../include/sdnq_hip.h"

#include "sdnq_dev.h"

namespace {

struct UnshardParams {
    const uint8_t* g;   // [W][M][wmax] elements
    uint8_t* out;       // [M][N]
    int64_t m, n_bytes, wmax_bytes;  // row sizes in BYTES
    int64_t m0, m_rows;              // rows [m0, m0 + m_rows) of the slabs are valid in g (M-chunked gathers pass one chunk at a time)
    int world;
    int64_t start_bytes[SDNQ_MAX_TP_RANKS + 1];  // byte offset of every rank's first channel inside an output row; [world] = n_bytes
};

// one thread = one 16-byte piece of an output row
__global__ __launch_bounds__(256) void unshard_columns_kernel(const UnshardParams p) {
    const int64_t pieces = p.n_bytes >> 4;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= p.m_rows * pieces) return;
    const int64_t r_m = idx / pieces, cb = (idx - r_m * pieces) << 4;  // row inside the chunk, byte column
    int r = 0;
#pragma unroll 1
    while (r + 1 < p.world && cb >= p.start_bytes[r + 1]) ++r;  // <= 63 steps; shard bounds are multiples of 16 channels
    const uint4 v = *(const uint4*)(p.g + ((int64_t)r * p.m_rows + r_m) * p.wmax_bytes + (cb - p.start_bytes[r]));
    *(uint4*)(p.out + (p.m0 + r_m) * p.n_bytes + cb) = v;
}

// ---- copy-free gather over peer-mapped memory (round 4) -----------------------------------------------------------------------------
// Every rank owns an output matrix [M][N] inside an IPC-shared arena that all ranks of the node have mapped (hipIpc handles, exchanged
// once).  Rank r PUSHES its slab y_r [rows][w] straight into columns [col0, col0 + w) of every rank's matrix -- its own with local
// stores, the peers' with P2P stores over xGMI -- so the gather needs no staging buffer, no collective call and no re-assembly pass:
// the one HBM read of y_r feeds W writes.  Rendezvous through two small peer-mapped u64 arrays per rank (each rank's arrays live in
// ITS OWN arena, written remotely, read locally):
//   post[s]  : rank s says "sequence number q of mine goes to byte offset o of MY arena" (one word: q in the top 24 bits, o / 256 below);
//              the receiver chooses where it receives, so the ranks' allocators never have to agree
//   done[s]  : rank s says "my slab of sequence q has landed in your matrix"
// push_columns_kernel: every workgroup reads the W posts (spinning until they carry q -- they were posted before the layer's matmul),
// copies its pieces to the W destinations, fences at system scope and takes a ticket; the LAST workgroup stores done[rank] = q into
// every rank's array (release, system scope) and then waits until its own done[*] carry q: when the kernel ends, this rank's output
// matrix is complete and every peer has been told about this rank's slab.  Spins are bounded by the wall clock (status word != 0 on
// timeout: the host raises instead of hanging the GPU).
struct PushParams {
    const uint8_t* y;        // [rows][ldy_bytes]
    int64_t rows, w_bytes, ldy_bytes;
    int64_t ldc_bytes, col0_bytes, row0;   // destination geometry: row-major [*][ldc], first column / first row of the slab
    uint8_t* arena[SDNQ_MAX_PUSH_RANKS];         // every rank's arena base as mapped HERE
    unsigned long long* post[SDNQ_MAX_PUSH_RANKS];   // rank p's post array (u64 [world]) as mapped here
    unsigned long long* done[SDNQ_MAX_PUSH_RANKS];   // rank p's done array
    int world, rank;
    unsigned long long seq;   // 24-bit sequence number of this gather
    unsigned int* ticket;     // device-local counter (zero before and after the launch)
    int* status;              // device-local: set to 1 on a rendezvous timeout
    long long timeout_ticks;  // wall_clock64 ticks (100 MHz)
};

__device__ __forceinline__ unsigned long long ld_sys(const unsigned long long* p) {
    return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ __launch_bounds__(256) void push_columns_kernel(const PushParams p) {
    __shared__ unsigned long long s_off[SDNQ_MAX_PUSH_RANKS];
    __shared__ int s_fail, s_last;
    const int tid = threadIdx.x;
    if (tid == 0) { s_fail = 0; s_last = 0; }
    __syncthreads();
    const long long t0 = wall_clock64();
    if (tid < p.world) {  // where does rank `tid` want sequence p.seq?  (it stored the word into slot `tid` of MY post array: a local read)
        unsigned long long v;
        for (;;) {
            v = ld_sys(p.post[p.rank] + tid);
            if ((v >> 40) == p.seq) break;
            if (wall_clock64() - t0 > p.timeout_ticks) { s_fail = 1; break; }
            __builtin_amdgcn_s_sleep(8);
        }
        s_off[tid] = (v & 0xffffffffffull) << 8;
    }
    __syncthreads();
    if (s_fail) {
        if (tid == 0) __hip_atomic_store(p.status, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);  // (host-coherent memory: the host polls it)
    } else {
        const int64_t pieces = p.w_bytes >> 4;
        const int64_t total = p.rows * pieces;
        for (int64_t idx = (int64_t)blockIdx.x * 256 + tid; idx < total; idx += (int64_t)gridDim.x * 256) {
            const int64_t r = idx / pieces, cb = (idx - r * pieces) << 4;
            const uint4 v = *(const uint4*)(p.y + r * p.ldy_bytes + cb);
            const int64_t dofs = (p.row0 + r) * p.ldc_bytes + p.col0_bytes + cb;
#pragma unroll 1
            for (int d = 0; d < p.world; ++d) {
                const int q = (p.rank + d) % p.world;  // own matrix first, then the peers round-robin (every link busy at once)
                *(uint4*)(p.arena[q] + s_off[q] + dofs) = v;
            }
        }
    }
    __threadfence_system();  // this workgroup's stores are visible system-wide before its ticket
    __syncthreads();
    if (tid == 0) s_last = (atomicAdd(p.ticket, 1u) == gridDim.x - 1);
    __syncthreads();
    if (!s_last) return;
    if (tid == 0) *p.ticket = 0;  // ready for the next launch (stream-ordered)
    __threadfence_system();
    if (tid < p.world) {
        if (!s_fail) __hip_atomic_store(p.done[tid] + p.rank, p.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        for (;;) {  // every peer's slab has landed here
            if (ld_sys(p.done[p.rank] + tid) == p.seq) break;
            if (wall_clock64() - t0 > p.timeout_ticks) { __hip_atomic_store(p.status, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); break; }
            __builtin_amdgcn_s_sleep(8);
        }
    }
}

struct PostParams {
    unsigned long long* post[SDNQ_MAX_PUSH_RANKS];
    int world, rank;
    unsigned long long word;
};

__global__ void post_kernel(const PostParams p) {
    const int t = threadIdx.x;
    if (t < p.world) __hip_atomic_store(p.post[t] + p.rank, p.word, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

}  // namespace

// ---- signal memory of the peer gather (round 5) -------------------------------------------------------------------------------------
// The post[] / done[] words are written by REMOTE kernels (P2P stores over xGMI) while a LOCAL kernel spins on them.  In-kernel
// visibility of such writes is only guaranteed for fine-grained / uncached allocations -- ordinary (coarse-grained) device memory is
// made coherent at kernel boundaries, so a spin on it may read a stale line for as long as the kernel runs (two processes on ONE GPU share
// a coherence point and never see that).  The control words therefore live in their own small allocation made with
// hipExtMallocWithFlags(hipDeviceMallocUncached) -- hipDeviceMallocFinegrained where the runtime refuses that -- exported and opened
// with hipIpc like the bulk arena (which stays coarse-grained: it is only read after the kernel that waited for `done`).  kind 1 is
// host memory (pinned, mapped, coherent) for the status word the HOST polls without synchronizing.
extern "C" int sdnq_hip_signal_alloc(int64_t bytes, int kind, void** ptr, int* granted) {
    if (!ptr || !granted) return SDNQ_ERR_NULL;
    if (bytes <= 0 || (kind != 0 && kind != 1)) return SDNQ_ERR_SHAPE;
    *ptr = nullptr; *granted = 0;
    if (kind == 1) {
        if (hipHostMalloc(ptr, (size_t)bytes, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) { (void)hipGetLastError(); return SDNQ_ERR_LAUNCH; }
        memset(*ptr, 0, (size_t)bytes);
        *granted = SDNQ_SIGNAL_HOST_COHERENT;
        return SDNQ_OK;
    }
    if (hipExtMallocWithFlags(ptr, (size_t)bytes, hipDeviceMallocUncached) == hipSuccess) *granted = SDNQ_SIGNAL_UNCACHED;
    else {
        (void)hipGetLastError();
        if (hipExtMallocWithFlags(ptr, (size_t)bytes, hipDeviceMallocFinegrained) == hipSuccess) *granted = SDNQ_SIGNAL_FINEGRAINED;
        else { (void)hipGetLastError(); return SDNQ_ERR_LAUNCH; }
    }
    if (hipMemset(*ptr, 0, (size_t)bytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess) { (void)hipFree(*ptr); *ptr = nullptr; return SDNQ_ERR_LAUNCH; }
    return SDNQ_OK;
}

extern "C" int sdnq_hip_signal_free(void* ptr, int kind) {
    if (!ptr) return SDNQ_OK;
    return (kind == 1 ? hipHostFree(ptr) : hipFree(ptr)) == hipSuccess ? SDNQ_OK : SDNQ_ERR_LAUNCH;
}

extern "C" int sdnq_hip_ipc_export(const void* ptr, void* handle64) {
    if (!ptr || !handle64) return SDNQ_ERR_NULL;
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "the handle travels as 64 bytes");
    return hipIpcGetMemHandle((hipIpcMemHandle_t*)handle64, (void*)ptr) == hipSuccess ? SDNQ_OK : ((void)hipGetLastError(), SDNQ_ERR_LAUNCH);
}

extern "C" int sdnq_hip_ipc_import(const void* handle64, void** ptr) {
    if (!handle64 || !ptr) return SDNQ_ERR_NULL;
    hipIpcMemHandle_t h;
    memcpy(&h, handle64, sizeof(h));
    return hipIpcOpenMemHandle(ptr, h, hipIpcMemLazyEnablePeerAccess) == hipSuccess ? SDNQ_OK : ((void)hipGetLastError(), SDNQ_ERR_LAUNCH);
}

extern "C" int sdnq_hip_ipc_close(void* ptr) {
    if (!ptr) return SDNQ_OK;
    return hipIpcCloseMemHandle(ptr) == hipSuccess ? SDNQ_OK : ((void)hipGetLastError(), SDNQ_ERR_LAUNCH);
}

extern "C" int sdnq_hip_push_post(void* const* post, int world, int rank, uint64_t seq, uint64_t arena_offset, sdnq_stream_t stream) {
    if (!post) return SDNQ_ERR_NULL;
    if (world < 1 || world > SDNQ_MAX_PUSH_RANKS || rank < 0 || rank >= world) return SDNQ_ERR_SHAPE;
    if ((arena_offset & 255) || (arena_offset >> 48) || (seq >> 24)) return SDNQ_ERR_ALIGN;
    PostParams p{};
    for (int r = 0; r < world; ++r) {
        if (!post[r]) return SDNQ_ERR_NULL;
        p.post[r] = (unsigned long long*)post[r];
    }
    p.world = world; p.rank = rank; p.word = (seq << 40) | (arena_offset >> 8);
    hipLaunchKernelGGL(post_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, p);
    SDNQ_CHECK_LAUNCH();
    return SDNQ_OK;
}

extern "C" int sdnq_hip_push_columns(const void* y, int elem_bytes, int64_t rows, int64_t w, int64_t ldy, void* const* arena,
                                     void* const* post, void* const* done, int world, int rank, uint64_t seq, int64_t ldc,
                                     int64_t col0, int64_t row0, void* ticket, void* status, int timeout_ms, sdnq_stream_t stream) {
    if (!y || !arena || !post || !done || !ticket || !status) return SDNQ_ERR_NULL;
    if (world < 1 || world > SDNQ_MAX_PUSH_RANKS || rank < 0 || rank >= world || (elem_bytes != 2 && elem_bytes != 4)) return SDNQ_ERR_SHAPE;
    if (rows <= 0 || w <= 0 || ldy < w || ldc < col0 + w || col0 < 0 || row0 < 0 || (seq >> 24)) return SDNQ_ERR_SHAPE;
    if (((uintptr_t)y % 16) || ((w * elem_bytes) % 16) || ((ldy * elem_bytes) % 16) || ((ldc * elem_bytes) % 16) || ((col0 * elem_bytes) % 16))
        return SDNQ_ERR_ALIGN;
    PushParams p{};
    p.y = (const uint8_t*)y; p.rows = rows; p.w_bytes = w * elem_bytes; p.ldy_bytes = ldy * elem_bytes;
    p.ldc_bytes = ldc * elem_bytes; p.col0_bytes = col0 * elem_bytes; p.row0 = row0;
    for (int r = 0; r < world; ++r) {
        if (!arena[r] || !post[r] || !done[r]) return SDNQ_ERR_NULL;
        if ((uintptr_t)arena[r] % 256) return SDNQ_ERR_ALIGN;
        p.arena[r] = (uint8_t*)arena[r]; p.post[r] = (unsigned long long*)post[r]; p.done[r] = (unsigned long long*)done[r];
    }
    p.world = world; p.rank = rank; p.seq = seq; p.ticket = (unsigned int*)ticket; p.status = (int*)status;
    p.timeout_ticks = (long long)(timeout_ms > 0 ? timeout_ms : 2000) * 100000ll;  // wall_clock64: 100 MHz
    const int64_t total = rows * (p.w_bytes >> 4);
    int64_t blocks = (total + 255) / 256;
    if (blocks > 1024) blocks = 1024;  // grid-stride: enough workgroups to keep every link and HBM channel busy
    hipLaunchKernelGGL(push_columns_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p);
    SDNQ_CHECK_LAUNCH();
    return SDNQ_OK;
}

extern "C" int sdnq_hip_unshard_columns(const void* gathered, void* out, int elem_bytes, int64_t m0, int64_t m_rows, int64_t m,
                                        int64_t wmax, int world, const int64_t* starts, sdnq_stream_t stream) {
    if (!gathered || !out || !starts) return SDNQ_ERR_NULL;
    if (world < 1 || world > SDNQ_MAX_TP_RANKS || (elem_bytes != 2 && elem_bytes != 4)) return SDNQ_ERR_SHAPE;
    if (m0 < 0 || m_rows <= 0 || m0 + m_rows > m || wmax <= 0) return SDNQ_ERR_SHAPE;
    if (((uintptr_t)gathered % 16) || ((uintptr_t)out % 16)) return SDNQ_ERR_ALIGN;
    UnshardParams p{};
    p.g = (const uint8_t*)gathered; p.out = (uint8_t*)out; p.m = m; p.m0 = m0; p.m_rows = m_rows; p.world = world;
    p.wmax_bytes = wmax * elem_bytes;
    for (int r = 0; r <= world; ++r) {
        if (starts[r] < 0 || (r > 0 && (starts[r] <= starts[r - 1] || starts[r] - starts[r - 1] > wmax))) return SDNQ_ERR_SHAPE;
        if ((starts[r] * elem_bytes) % 16) return SDNQ_ERR_ALIGN;  // a 16-byte piece never straddles two ranks
        p.start_bytes[r] = starts[r] * elem_bytes;
    }
    if (starts[0] != 0 || (p.wmax_bytes % 16)) return SDNQ_ERR_SHAPE;
    p.n_bytes = p.start_bytes[world];
    const int64_t total = m_rows * (p.n_bytes >> 4);
    hipLaunchKernelGGL(unshard_columns_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p);
    SDNQ_CHECK_LAUNCH();
    return SDNQ_OK;
}
