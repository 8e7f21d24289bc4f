"""Time the scaled matmul with and without the low-rank (SVD) epilogue on FLUX shapes (graph replay, GPU box only)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sdnq_amd import ops

dev = torch.device("cuda:0")


def timeit(fn, reps=20):
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        fn()
        side.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            for _ in range(reps):
                fn()
        graph.replay()
        side.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(side)
        for _ in range(3):
            graph.replay()
        e1.record(side)
        side.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (3 * reps)


for (m, n, k) in [(4608, 3072, 3072), (4608, 12288, 3072), (4608, 3072, 12288), (4608, 3072, 15360), (4096, 9216, 3072), (1024, 1280, 1280)]:
    g = torch.Generator().manual_seed(1)
    a = torch.randint(-127, 128, (m, k), generator=g, dtype=torch.int8).to(dev)
    b = torch.randint(-127, 128, (n, k), generator=g, dtype=torch.int8).to(dev)
    sa = (torch.rand(m, 1, generator=g) * 0.01).to(dev)
    sb = (torch.rand(n, generator=g) * 0.01).to(dev)
    bias = torch.randn(n, generator=g).to(torch.bfloat16).to(dev)
    t = torch.randn(m, 32, generator=g).to(torch.bfloat16).to(dev)
    up = torch.randn(n, 32, generator=g).to(torch.bfloat16).to(dev)
    t0 = timeit(lambda: ops.scaled_mm(ops.MM_I8, a, b, sa, sb, bias, torch.bfloat16))
    t1 = timeit(lambda: ops.scaled_mm_lowrank(ops.MM_I8, a, b, sa, sb, bias, t, up, None, None, torch.bfloat16))
    print(f"{m} x {n} x {k}: plain {t0:8.2f} us ({2 * m * n * k / t0 / 1e9:6.1f} TOP/s)   low-rank {t1:8.2f} us   +{t1 - t0:6.2f} us")
