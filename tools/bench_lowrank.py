"""Time sdnq_hip_lowrank_down (t = x . svd_down^T) on FLUX / SDXL shapes and check it against torch.  GPU box only."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sdnq_amd import ops

dev = torch.device("cuda:0")
for (m, k, r) in [(4608, 3072, 32), (4608, 12288, 32), (4608, 15360, 32), (4096, 3072, 32), (512, 3072, 32), (1024, 1280, 32), (4096, 640, 32), (77, 2048, 32), (4608, 3072, 16), (4608, 3072, 64), (33, 48, 8)]:
    g = torch.Generator().manual_seed(m + k)
    x = torch.randn(m, k, generator=g).to(torch.bfloat16).to(dev)
    d = (torch.randn(r, k, generator=g) * 0.05).to(torch.bfloat16).to(dev)
    t = ops.lowrank_down(x, d)
    ref = (x.float() @ d.float().t())
    err = (t.float() - ref).abs().max().item() / ref.abs().max().item()
    side = torch.cuda.Stream()
    n = 50
    with torch.cuda.stream(side):
        for _ in range(3):
            ops.lowrank_down(x, d)
        side.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):  # launch cost excluded: n back-to-back launches in one graph
            for _ in range(n):
                ops.lowrank_down(x, d)
        graph.replay()
        side.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(side)
        for _ in range(4):
            graph.replay()
        e1.record(side)
        side.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (4 * n)
    print(f"lowrank_down {m:5d} x {k:5d} r={r:2d}: {us:7.2f} us  {2 * m * k / us / 1e6:6.2f} TB/s of x   rel err {err:.2e}")


def bench_skinny_svd(n, k, r=32, m=1):
    """sdnq_hip_linear_skinny_svd (the M < 32 branch of an int8 + SVD layer) against dequantize + F.linear on the same tensors."""
    g = torch.Generator().manual_seed(n + k)
    w = torch.randint(-127, 128, (n, k), generator=g, dtype=torch.int8).to(dev)
    sc = (torch.rand(n, 1, generator=g) * 0.01 + 0.001).to(dev)
    up = (torch.randn(n, r, generator=g) * 0.05).to(torch.bfloat16).to(dev)
    down = (torch.randn(r, k, generator=g) * 0.05).to(torch.bfloat16).to(dev)
    x = torch.randn(m, k, generator=g).to(torch.bfloat16).to(dev)
    qw = ops.make_quant_weight("int8", w, sc, None, up, down, n, k, k, transposed=False, svd_transposed=False)
    down_t = down.t().contiguous()
    y = ops.linear_skinny_svd(qw, down_t, x, None)
    wd = ops.dequant(qw, torch.bfloat16, 0)
    ref = x.float() @ wd.float().t()
    err = (y.float() - ref).abs().max().item() / ref.abs().max().item()
    side = torch.cuda.Stream()
    reps = 50
    with torch.cuda.stream(side):
        ops.linear_skinny_svd(qw, down_t, x, None)
        side.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            for _ in range(reps):
                ops.linear_skinny_svd(qw, down_t, x, None)
        graph.replay()
        side.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(side)
        for _ in range(4):
            graph.replay()
        e1.record(side)
        side.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (4 * reps)
    print(f"skinny_svd  m={m} {n:6d} x {k:5d} r={r}: {us:7.2f} us  {n * k / us / 1e6:6.2f} TB/s of codes   rel err vs dequant+matmul {err:.2e}")


for (n, k, m) in [(18432, 3072, 1), (9216, 3072, 1), (6144, 3072, 1), (3072, 3072, 2), (1280, 1280, 4), (64, 256, 1)]:
    bench_skinny_svd(n, k, 32, m)
bench_skinny_svd(3072, 3072, 16, 1)
