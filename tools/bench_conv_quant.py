#!/usr/bin/env python3
"""Kernel micro-benchmark (development aid): the fused unfold + row quantization of the conv matmul (sdnq_hip_im2col_rowquant_z) per SDXL
conv geometry, graph-replayed, against the bytes it must write (the [M][K] operand) and read.  usage: tools/bench_conv_quant.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdnq_amd import ops
dev = torch.device("cuda:0")
GEO = [(320, 128, 3, 1, 1), (640, 128, 3, 1, 1), (960, 128, 3, 1, 1), (320, 128, 3, 2, 1), (640, 64, 3, 1, 1), (1280, 64, 3, 1, 1), (1920, 64, 3, 1, 1),
       (1280, 32, 3, 1, 1), (2560, 32, 3, 1, 1), (640, 64, 1, 1, 0), (2560, 32, 1, 1, 0)]


def timed(fn, reps=10):
    s = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(s):
        fn(); s.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps):
                fn()
        g.replay(); s.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(3):
            g.replay()
        e1.record(s); s.synchronize()
    return e0.elapsed_time(e1) / (3 * reps) * 1e3


for (c, hw, k, st, pd) in GEO:
    x = torch.randn(1, c, hw, hw, device=dev, dtype=torch.bfloat16)
    t = timed(lambda: ops.im2col_rowquant(x, (k, k), (st, st), (pd, pd), (1, 1), ops.MM_I8))
    ho = (hw + 2 * pd - k) // st + 1
    m, kk = ho * ho, c * k * k
    wr, rd = m * kk, c * hw * hw * 2
    print(f"C={c:5d} {hw:3d}x{hw:<3d} k={k} s={st}: {t:7.1f} us   writes {wr / 1e6:6.1f} MB ({wr / t / 1e6:5.2f} TB/s)   reads {rd / 1e6:5.1f} MB once", flush=True)
