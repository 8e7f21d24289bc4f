"""Column-sharded (tensor-parallel) quantized Linear over the GPUs of one node: RCCL all-gather over xGMI.

The reference has no inference parallelism; this is the new capability BASELINE.json's north_star
asks for.  The path shards on OUTPUT CHANNELS: weight rows, scale[n], zero_point[n], bias[n],
svd_up[n,:] and the re-quantized row scale are all per-output-channel, while x, its row scale, the Hadamard
rotation and x @ svd_down depend only on the replicated activation.  So rank r quantizes the full activation
(cheap, M*K) and multiplies it with its N/W slab of the weight; ONE collective per layer re-assembles the output:

    y_r [M, N/W]  --ncclAllGather-->  [W, M, N/W]  --strided copy-->  y [M, N]

One process per GPU; ``torch.distributed`` backend "nccl" is RCCL on ROCm.  The message per rank is 2*M*N/W bytes
(bf16); xGMI is point-to-point (7 links x ~153 GB/s per GPU), so a fully connected all-gather is bound by one link:
t ~ 2*M*N/W / 153 GB/s + launch latency -- at SDXL bs=1 sizes that is longer than the sharded GEMM itself, which is
why the default multi-GPU mode of bench.py is independent replicas and TP is opt-in.

Slicing happens on the FLOAT weight before quantization (N/W must stay a multiple of 16, utils.py:96-97), so every
shard is an ordinary SDNQLinear with the reference's state_dict layout; quantizing per shard is numerically identical
to slicing a quantized full layer because all quantization statistics are per output row (Hadamard rotates along K;
the SVD split is the one exception: it is computed per shard, which changes the low-rank factors but not the contract).
"""
from __future__ import annotations

import torch

from .quantizer import SDNQConfig, sdnq_quantize_layer


def shard_bounds(n: int, rank: int, world: int, multiple: int = 16) -> tuple[int, int]:
    """Even split of N output channels in units of `multiple` (the reference needs N % 16 == 0 for quantized matmul)."""
    units = n // multiple
    if n % multiple or units < world:
        raise ValueError(f"N={n} cannot be column-sharded {world} ways in multiples of {multiple}")
    base, extra = divmod(units, world)
    start = (rank * base + min(rank, extra)) * multiple
    size = (base + (1 if rank < extra else 0)) * multiple
    return start, start + size


class ColumnShardedLinear(torch.nn.Module):
    """Holds this rank's slab of a Linear; forward = local forward + all-gather along the channel axis."""

    def __init__(self, local: torch.nn.Module, n_total: int, rank: int, world: int, group=None):
        super().__init__()
        self.local = local
        self.n_total, self.rank, self.world, self.group = n_total, rank, world, group
        self.bounds = [shard_bounds(n_total, r, world) for r in range(world)]
        self.even = len({b - a for a, b in self.bounds}) == 1

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        import torch.distributed as dist
        y_local = self.local(x)
        lead = y_local.shape[:-1]
        y2 = y_local.reshape(-1, y_local.shape[-1]).contiguous()
        m = y2.shape[0]
        if self.even:
            gathered = torch.empty((self.world * m, y2.shape[1]), device=y2.device, dtype=y2.dtype)  # rank-major concat
            dist.all_gather_into_tensor(gathered, y2, group=self.group)
            out = gathered.view(self.world, m, y2.shape[1]).permute(1, 0, 2).reshape(m, self.n_total)
        else:  # uneven shards: pad every slab to the widest one (collectives want equal messages), gather, trim
            wmax = max(b - a for a, b in self.bounds)
            padded = torch.zeros((m, wmax), device=y2.device, dtype=y2.dtype)
            padded[:, : y2.shape[1]] = y2
            gathered = torch.empty((self.world * m, wmax), device=y2.device, dtype=y2.dtype)
            dist.all_gather_into_tensor(gathered, padded, group=self.group)
            g3 = gathered.view(self.world, m, wmax)
            out = torch.cat([g3[r, :, : b - a] for r, (a, b) in enumerate(self.bounds)], dim=-1)
        return out.view(*lead, self.n_total)


@torch.no_grad()
def column_shard_linear(linear: torch.nn.Linear, config: SDNQConfig, rank: int, world: int, group=None,
                        quantize=sdnq_quantize_layer) -> ColumnShardedLinear:
    """Slice a float nn.Linear on its output channels, quantize the slab, wrap it. `quantize` is injectable so the
    CPU (gloo) tests can exercise the sharding + collective plumbing without a GPU."""
    n = linear.out_features
    a, b = shard_bounds(n, rank, world)
    slab = torch.nn.Linear(linear.in_features, b - a, bias=linear.bias is not None, device=linear.weight.device,
                           dtype=linear.weight.dtype)
    slab.weight.copy_(linear.weight[a:b])
    if linear.bias is not None:
        slab.bias.copy_(linear.bias[a:b])
    local, _ = quantize(slab, config)
    return ColumnShardedLinear(local, n, rank, world, group)
