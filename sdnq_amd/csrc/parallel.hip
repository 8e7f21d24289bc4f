// Tensor-parallel glue on gfx950: the re-assembly of a column-sharded Linear's output after the RCCL all-gather.
//
// Rank r computes y_r [M][w_r] (its N / W output channels); ncclAllGather moves ONE contiguous buffer per rank, so the gathered
// result is rank-major: g [W][M][wmax] (slabs padded to the widest one when N / 16 does not divide evenly).  The consumer wants
// row-major y [M][N] with N = sum w_r: out[m][start_r + c] = g[r][m][c].  The reference has no inference parallelism;
// round 2 did this with torch permute / reshape (even shards) or zeros + cat (uneven) -- several launches and, uneven, several
// passes.  Here: one HBM-bound pass, 16 bytes per lane, reads and writes coalesced along the channels of a rank.
#include <hip/hip_runtime.h>

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

}  // namespace

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
