#!/usr/bin/env python3
"""Kernel micro-benchmark (development aid): per-shape time of rowquant and scaled_mm, graph-replayed back-to-back
launches timed with HIP events on the launch stream. Usage: python tools/bench_gemm.py [int8|fp8] [shape-set]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdnq_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
mm = ops.MM_FP8 if (len(sys.argv) > 1 and sys.argv[1] == "fp8") else ops.MM_I8
SDXL = [(4096, 640, 640), (4096, 5120, 640), (4096, 640, 2560), (1024, 1280, 1280), (1024, 10240, 1280), (1024, 1280, 5120),
        (77, 640, 2048), (77, 1280, 2048)]
BIG = [(16384, 8192, 4096), (4608, 3072, 3072), (4608, 12288, 3072), (4608, 3072, 15360), (8192, 8192, 8192)]
CONV = [(16384, 320, 2880), (16384, 320, 8640), (4096, 640, 5760), (4096, 640, 17280), (1024, 1280, 11520), (1024, 1280, 23040), (4096, 1280, 11520)]
shapes = CONV if (len(sys.argv) > 2 and sys.argv[2] == "conv") else SDXL + (BIG if (len(sys.argv) > 2 and sys.argv[2] == "all") else BIG[:1])


def timed(fn, reps=20):
    s = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(s):
        fn()
        s.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps):
                fn()
        g.replay()
        s.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(3):
            g.replay()
        e1.record(s)
        s.synchronize()
    return e0.elapsed_time(e1) / (3 * reps) * 1e3  # us per call


for (m, n, k) in shapes:
    x = torch.randn(m, k, device=dev, dtype=torch.bfloat16)
    if mm == ops.MM_I8:
        b = torch.randint(-128, 128, (n, k), dtype=torch.int8, device=dev)
    else:
        b = (torch.randn(n, k, device=dev) * 30).to(torch.float8_e4m3fn)
    sb = torch.rand(n, device=dev) * 0.01
    bias = torch.randn(n, device=dev, dtype=torch.bfloat16)
    xq, xs, _, _ = ops.rowquant(x, mm)
    tq = timed(lambda: ops.rowquant(x, mm))
    tg = timed(lambda: ops.scaled_mm(mm, xq, b, xs, sb, bias, torch.bfloat16))
    print(f"M={m:6d} N={n:6d} K={k:6d}: rowquant {tq:7.2f} us ({(3 * m * k) / tq / 1e3:7.1f} GB/s)   gemm {tg:8.2f} us  {2 * m * n * k / tg / 1e6:7.1f} TOP/s", flush=True)
