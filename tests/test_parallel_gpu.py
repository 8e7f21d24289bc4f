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
            x3 = torch.randn(700, 640, generator=torch.Generator().manual_seed(2)).to(torch.bfloat16).to(dev)
            ok = ok and torch.equal(column_shard_module(mod, rank, world, chunks=2)(x3), mod(x3))  # gather of chunk 0 under the matmul of chunk 1
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


def test_sequential_shards_at_cfg5_geometry(gpu_device):
    """BASELINE configs[4]: FLUX.1-dev proj_mlp, int8 + SVD rank 32, TP = 8 -- N = 12288, K = 3072, bias, 4096 + 512 tokens.  The
    eight slabs of 1536 channels (their own tile choices: 18 x 12 tiles of 256 x 128 instead of the 864 tiles of the whole layer)
    computed one after the other and concatenated must equal the unsharded layer bit for bit, for the w8a8 GEMM rows and for the
    M = 1 (adaLN-style) branch."""
    import sdnq_amd
    from sdnq_amd.parallel import shard_bounds, shard_quantized_module
    torch.manual_seed(0)
    n, k, world = 12288, 3072, 8
    lin = torch.nn.Linear(k, n, bias=True).to(torch.bfloat16).to(gpu_device)
    mod, _ = sdnq_amd.sdnq_quantize_layer(lin, sdnq_amd.SDNQConfig(weights_dtype="int8", group_size=-1, use_quantized_matmul=True, use_svd=True,
                                                                   svd_rank=32))
    slabs = [shard_quantized_module(mod, *shard_bounds(n, r, world)) for r in range(world)]
    assert all(s.weight.shape[1 if s.sdnq_dequantizer.weight_is_transposed else 0] == 1536 for s in slabs)
    for m in (4608, 1):
        x = torch.randn(m, k, device=gpu_device, dtype=torch.bfloat16)
        x[:, 11] *= 20
        want = mod(x)
        got = torch.cat([s(x) for s in slabs], dim=-1)
        assert torch.equal(got, want), (m, int((got != want).sum()))


@pytest.mark.parametrize("world,n,m0,rows,m", [(8, 12288, 0, 4608, 4608), (4, 1280 + 16, 64, 100, 300), (3, 96, 0, 1, 1), (2, 4096, 1024, 1024, 2048)])
@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float32])
def test_unshard_columns_kernel(world, n, m0, rows, m, dt, gpu_device):
    """sdnq_hip_unshard_columns against the torch re-assembly it replaces: even and uneven shard widths (padded slabs), a row
    window of the output (the M-chunked pipeline), 16- and 32-bit elements; rows outside the window stay untouched."""
    from sdnq_amd import ops
    from sdnq_amd.parallel import shard_bounds
    bounds = [shard_bounds(n, r, world) for r in range(world)]
    wmax = max(b - a for a, b in bounds)
    g = torch.randn(world, rows, wmax, device=gpu_device).to(dt)
    out = torch.full((m, n), -7.0, device=gpu_device, dtype=dt)
    ops.unshard_columns(g, out, [a for a, _ in bounds] + [n], m0)
    want = torch.full((m, n), -7.0, device=gpu_device, dtype=dt)
    want[m0:m0 + rows] = torch.cat([g[r, :, : b - a] for r, (a, b) in enumerate(bounds)], dim=-1)
    assert torch.equal(out, want)


def _rccl_single_rank_worker(port, q):
    import torch.distributed as dist
    import sdnq_amd
    from sdnq_amd.parallel import column_shard_module
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        torch.manual_seed(0)
        lin = torch.nn.Linear(640, 1280, bias=True).to(torch.bfloat16).to(dev)
        mod, _ = sdnq_amd.sdnq_quantize_layer(lin, sdnq_amd.SDNQConfig(weights_dtype="int8", group_size=-1, use_quantized_matmul=True))
        x = torch.randn(2, 1000, 640, generator=torch.Generator().manual_seed(1)).to(torch.bfloat16).to(dev)
        want = mod(x)
        ok = True
        for chunks in (1, 3):  # the pipelined (side-stream) gather is bit-identical to the plain one
            y = column_shard_module(mod, 0, 1, chunks=chunks)(x)
            torch.cuda.synchronize()
            ok = ok and torch.equal(y, want)
        q.put(bool(ok))
    finally:
        dist.destroy_process_group()


def test_column_shard_module_one_rccl_rank_plain_and_pipelined(gpu_device):
    """The RCCL path end to end on the one GPU a test box has: all-gather (world 1) + the un-shard kernel, plain and M-chunked with
    the gather on a side stream, against the unsharded layer."""
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_single_rank_worker, args=(port, q))
    p.start()
    p.join(300)
    assert p.exitcode == 0
    assert q.get(timeout=10) is True


def _peer_worker(rank, world, port, q, device_of_rank):
    """One rank of the copy-free gather test: gloo for the handle exchange (RCCL refuses two ranks on one device), the data path is
    sdnq_hip_push_post / sdnq_hip_push_columns over IPC-mapped arenas."""
    import torch.distributed as dist
    import sdnq_amd
    from sdnq_amd.parallel import PeerArena, column_shard_module
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    d = device_of_rank[rank]
    torch.cuda.set_device(d)
    dev = torch.device("cuda", d)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        arena = PeerArena(rank, world, device=dev, arena_bytes=64 << 20, timeout_ms=20000)
        ok, why = True, []
        # the words remote kernels write and local kernels spin on live in signal memory (fine-grained / uncached), outside the arena
        lo, hi = arena.buf.data_ptr(), arena.buf.data_ptr() + arena.buf.numel()
        if arena.ctrl_kind not in ("uncached", "finegrained") or any(lo <= int(p or 0) < hi for p in list(arena._post) + list(arena._done)):
            ok = False
            why.append(("control words", arena.ctrl_kind))
        arena.poll()
        # (1) the raw gather, uneven slabs: rank r contributes columns filled with r + 1 (+ row index)
        bounds = [(0, 48), (48, 80)] if world == 2 else [(16 * r, 16 * (r + 1)) for r in range(world)]
        n_total = bounds[-1][1]
        for rows in (1, 77, 1000):
            a, b = bounds[rank]
            y = (torch.arange(rows, device=dev, dtype=torch.float32)[:, None] * 0.5 + (rank + 1)).to(torch.bfloat16).expand(rows, b - a).contiguous()
            out = arena.gather(y, n_total, a)
            arena.check()
            want = torch.cat([(torch.arange(rows, device=dev, dtype=torch.float32)[:, None] * 0.5 + (r + 1)).to(torch.bfloat16).expand(rows, bb - aa)
                              for r, (aa, bb) in enumerate(bounds)], dim=1)
            if not torch.equal(out, want):
                ok = False
                why.append(("raw", rows))
        # (2) column-sharded quantized layers through the arena == the unsharded layer, bit for bit
        for cfg in (_SHARD_CFGS[0], _SHARD_CFGS[1], _SHARD_CFGS[3]):
            torch.manual_seed(0)
            lin = torch.nn.Linear(640, 1280 + 16, bias=True).to(torch.bfloat16).to(dev)  # 81 units of 16: uneven shards
            mod, _ = sdnq_amd.sdnq_quantize_layer(lin, sdnq_amd.SDNQConfig(**cfg))
            sh = column_shard_module(mod, rank, world, peer=arena)
            outs = []
            for seed, shape in ((1, (2, 100, 640)), (2, (700, 640)), (3, (1, 640))):
                x = torch.randn(*shape, generator=torch.Generator().manual_seed(seed)).to(torch.bfloat16).to(dev)
                y = sh(x)
                outs.append((y, mod(x)))  # keep every output alive: the ring must not reuse their memory
            arena.check()
            for i, (y, want) in enumerate(outs):
                if y.shape != want.shape or not torch.equal(y, want):
                    ok = False
                    why.append((str(cfg)[:40], i))
        # (3) the ring: dead outputs are recycled, live ones never -- run more bytes through than the arena holds
        keep = arena.gather(torch.full((64, bounds[rank][1] - bounds[rank][0]), 3.0, device=dev, dtype=torch.bfloat16), n_total, bounds[rank][0])
        for it in range(40):  # 40 x ~3.3 MB through a 64 MB arena
            tmp = arena.gather(torch.full((20000, bounds[rank][1] - bounds[rank][0]), float(it), device=dev, dtype=torch.bfloat16), n_total, bounds[rank][0])
            del tmp
        arena.check()
        if not bool((keep == 3.0).all()):
            ok = False
            why.append("ring overwrote a live tensor")
        # (4) a rank that does not show up: the waiting rank's status word (host-coherent memory) makes its NEXT gather raise
        lonely = PeerArena(rank, world, device=dev, arena_bytes=1 << 20, timeout_ms=300)
        if rank == 0:
            lonely.gather(torch.ones((8, bounds[0][1] - bounds[0][0]), device=dev, dtype=torch.bfloat16), n_total, bounds[0][0])
            torch.cuda.synchronize(dev)
            try:
                lonely.gather(torch.ones((8, bounds[0][1] - bounds[0][0]), device=dev, dtype=torch.bfloat16), n_total, bounds[0][0])
                ok = False
                why.append("a timed-out rendezvous went unnoticed")
            except RuntimeError as e:
                if "did not arrive" not in str(e):
                    raise
        dist.barrier()
        q.put((rank, bool(ok), why))
    except Exception as e:  # noqa: BLE001
        q.put((rank, False, [repr(e)]))
        raise
    finally:
        dist.destroy_process_group()


def _run_peer_test(world, device_of_rank):
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_peer_worker, args=(r, world, port, q, device_of_rank)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
    alive = [p for p in procs if p.is_alive()]
    for p in alive:
        p.kill()
    assert not alive, "a peer-gather rank hung"
    res = [q.get(timeout=10) for _ in range(world)]
    assert all(r[1] for r in res), res


def test_copy_free_peer_gather_two_processes_one_gpu(gpu_device):
    """The copy-free gather (PeerArena: IPC-mapped arenas, P2P stores, mailbox rendezvous) between TWO PROCESSES that share the one
    GPU of this box -- IPC handles work intra-device, so the whole protocol (handle exchange, posts, pushes into the peer's arena,
    completion flags, the ring allocator's liveness rule) runs for real; only the xGMI hop is missing.  Bit-identical to the unsharded
    layer for three storage formats, uneven shards, M = 1 / 200 / 700."""
    _run_peer_test(2, [0, 0])


def test_copy_free_peer_gather_two_gpus(gpu_device):
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (P2P stores over xGMI); the one-GPU two-process test runs the same protocol")
    _run_peer_test(2, [0, 1])
