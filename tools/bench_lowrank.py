#!/usr/bin/env python3
"""scaled_mm vs scaled_mm_lowrank (SVD epilogue) at FLUX shapes. usage: bench_lowrank.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdnq_amd import ops
dev = torch.device("cuda:0")
def t(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for (m, n, k) in ((4608, 3072, 3072), (4608, 12288, 3072), (4608, 3072, 15360), (4096, 3072, 3072), (512, 3072, 3072)):
    x = torch.randn(m, k, device=dev, dtype=torch.bfloat16)
    b = torch.randint(-128, 128, (n, k), dtype=torch.int8, device=dev)
    sb = torch.rand(n, device=dev) * 0.01
    bias = torch.randn(n, device=dev, dtype=torch.bfloat16)
    up = (torch.randn(n, 32, device=dev) * 0.1).to(torch.bfloat16)
    down = (torch.randn(32, k, device=dev) * 0.1).to(torch.bfloat16)
    xq, xs, _, _ = ops.rowquant(x, ops.MM_I8)
    tt = ops.lowrank_down(x, down)
    a = t(lambda: ops.scaled_mm(ops.MM_I8, xq, b, xs, sb, bias, torch.bfloat16))
    c = t(lambda: ops.scaled_mm_lowrank(ops.MM_I8, xq, b, xs, sb, bias, tt, up, None, None, torch.bfloat16))
    d = t(lambda: ops.lowrank_down(x, down))
    print(f"M={m} N={n} K={k}: plain {a:8.1f} us   lowrank {c:8.1f} us   lowrank_down {d:7.1f} us")
