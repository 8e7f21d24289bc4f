"""GPU side of the tensor-parallel path: the slabs of a pre-quantized layer computed one after the other on ONE GPU and concatenated
must equal the unsharded layer bit for bit (every storage format / forward), and -- when the box has two GPUs -- the same over two
RCCL ranks with the all-gather."""
import os
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu

from tests.test_parallel_gloo import _SHARD_CFGS  # noqa: E402


@pytest.mark.parametrize("cfg", _SHARD_CFGS, ids=lambda c: "-".join(f"{k[:6]}={v}" for k, v in c.items()))
@pytest.mark.parametrize("m", [1, 200])
def test_sequential_shards_equal_unsharded_layer(cfg, m, gpu_device):
    import sdnq_amd
    from sdnq_amd.parallel import shard_bounds, shard_quantized_module
    torch.manual_seed(0)
    n, k, world = 1280, 640, 4
    lin = torch.nn.Linear(k, n, bias=True).to(torch.bfloat16).to(gpu_device)
    mod, _ = sdnq_amd.sdnq_quantize_layer(lin, sdnq_amd.SDNQConfig(**cfg))
    x = torch.randn(m, k, device=gpu_device, dtype=torch.bfloat16)
    want = mod(x)
    parts = [shard_quantized_module(mod, *shard_bounds(n, r, world))(x) for r in range(world)]
    got = torch.cat(parts, dim=-1)
    assert got.shape == want.shape and torch.equal(got, want), (cfg, m, int((got != want).sum()))


def _rccl_worker(rank, world, port, q):
    import torch.distributed as dist
    import sdnq_amd
    from sdnq_amd.parallel import column_shard_module
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    try:
        ok = True
        for cfg in (_SHARD_CFGS[0], _SHARD_CFGS[1], _SHARD_CFGS[3]):
            torch.manual_seed(0)
            lin = torch.nn.Linear(640, 1280 + 16, bias=True).to(torch.bfloat16).to(dev)  # 81 units of 16: uneven shards
            mod, _ = sdnq_amd.sdnq_quantize_layer(lin, sdnq_amd.SDNQConfig(**cfg))
            x = torch.randn(2, 100, 640, generator=torch.Generator().manual_seed(1)).to(torch.bfloat16).to(dev)
            y = column_shard_module(mod, rank, world)(x)
            ok = ok and torch.equal(y, mod(x))
        torch.cuda.synchronize()
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_column_shard_module_two_rccl_ranks(gpu_device):
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (RCCL over xGMI); the gloo test covers the plumbing, the sequential test the arithmetic")
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rccl_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    assert dict(q.get(timeout=10) for _ in range(2)) == {0: True, 1: True}
