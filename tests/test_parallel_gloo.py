"""N>1 path on CPU: world_size-2 gloo processes exercise the column sharding + all-gather plumbing.

The local compute is injected (the CPU oracle stands in for the HIP forward -- tests may use the oracle as the
checker/stand-in); what is under test is shard_bounds, the slab slicing, the collective and the re-assembly.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from sdnq_amd.parallel import ColumnShardedLinear, column_shard_linear, shard_bounds


def test_shard_bounds_cover_and_align():
    for n in (640, 1280, 5120, 10240, 48 * 16, 3072, 18432):
        for world in (1, 2, 4, 8):
            if n // 16 < world:
                continue
            spans = [shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            for (a, b), (c, d) in zip(spans, spans[1:]):
                assert b == c
            assert all((b - a) % 16 == 0 and b > a for a, b in spans)
    with pytest.raises(ValueError):
        shard_bounds(100, 0, 2)       # not a multiple of 16
    with pytest.raises(ValueError):
        shard_bounds(64, 0, 8)        # fewer 16-channel units than ranks


class _OracleShard(torch.nn.Module):
    """Stand-in for the quantized slab: int8 row-wise w8a8 forward computed by the CPU oracle."""

    def __init__(self, lin):
        super().__init__()
        from oracle import oracle as O
        w = lin.weight.detach().float().numpy()
        self.scale = (np.abs(w).max(-1) / 127).astype(np.float32)
        self.wq = np.clip(np.rint(w / self.scale[:, None]), -128, 127).astype(np.int8)
        self.bias = None if lin.bias is None else O.round_dtype(lin.bias.detach().float().numpy(), "bf16")
        self.O = O

    def forward(self, x):
        O = self.O
        x2 = x.float().numpy().reshape(-1, x.shape[-1])
        xq, xs, _ = O.rowquant(x2, "int8")
        y = O.scaled_mm("int8", xq, self.wq, xs, self.scale, self.bias, "bf16")
        return torch.from_numpy(y).to(torch.bfloat16).view(*x.shape[:-1], -1)


def _worker(rank, world, port, n, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(0)
        lin = torch.nn.Linear(128, n, bias=True).to(torch.bfloat16)
        x = torch.randn(2, 20, 128).to(torch.bfloat16)
        sharded = column_shard_linear(lin, None, rank, world, quantize=lambda slab, cfg: (_OracleShard(slab), cfg))
        assert isinstance(sharded, ColumnShardedLinear)
        y = sharded(x)
        full = _OracleShard(lin)(x)
        ok = torch.equal(y, full) and tuple(y.shape) == (2, 20, n)
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n", [64, 80])   # 80 -> uneven shards (48 + 32): exercises the list all_gather branch
def test_column_sharded_linear_world2_gloo(n):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    results = dict(q.get(timeout=10) for _ in range(2))
    assert results == {0: True, 1: True}
