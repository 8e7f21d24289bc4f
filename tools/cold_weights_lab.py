#!/usr/bin/env python3
"""Development aid (round 5): what a GEMM launch costs when its weights come from the L2, from the 256-MiB Infinity Cache or from HBM.
One graph of `n` launches of one problem, each on its own weight matrix out of a pool of `pool` matrices; replayed; us per launch."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdnq_amd import ops
dev = torch.device("cuda:0")
shapes = [tuple(int(v) for v in t.split(",")) for t in (sys.argv[1] if len(sys.argv) > 1 else "1024,1280,1280;1024,1280,5120;4096,640,640").split(";")]
for (m, n, k) in shapes:
    x = torch.randn(m, k, device=dev, dtype=torch.bfloat16)
    xq, xs, _, _ = ops.rowquant(x, ops.MM_I8)
    sb = torch.rand(n, device=dev) * 0.01
    bias = torch.randn(n, device=dev, dtype=torch.bfloat16)
    line = f"M={m} N={n} K={k} ({n * k / 1e6:.2f} MB of weights per launch):"
    for pool_mb in (0, 96, 1536):
        pool = max(1, int(pool_mb * 1e6 / (n * k)))
        ws = [torch.randint(-128, 128, (n, k), dtype=torch.int8, device=dev) for _ in range(pool)]
        nl = max(pool, 64)
        s = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(s):
            ops.scaled_mm(ops.MM_I8, xq, ws[0], xs, sb, bias, torch.bfloat16); s.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                for i in range(nl):
                    ops.scaled_mm(ops.MM_I8, xq, ws[i % pool], xs, sb, bias, torch.bfloat16)
            g.replay(); s.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s)
            for _ in range(3):
                g.replay()
            e1.record(s); s.synchronize()
        line += f"  pool {pool_mb:4d} MB ({pool} matrices): {e0.elapsed_time(e1) / (3 * nl) * 1e3:7.2f} us"
        del ws, g
    print(line, flush=True)
