#!/usr/bin/env python3
"""A/B of GEMM tile configurations on given shapes (development aid): graph-replayed launches, HIP events.
usage: tile_ab.py "M,N,K;M,N,K" "-1,20,3,1" """
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdnq_amd import _lib, ops
dev = torch.device("cuda:0")
shapes = [tuple(int(v) for v in s.split(",")) for s in sys.argv[1].split(";")]
tiles = [int(v) for v in sys.argv[2].split(",")]
lib = _lib.load()


def timed(fn, reps=20):
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


for (m, n, k) in shapes:
    x = torch.randn(m, k, device=dev, dtype=torch.bfloat16)
    b = torch.randint(-128, 128, (n, k), dtype=torch.int8, device=dev)
    sb = torch.rand(n, device=dev) * 0.01
    bias = torch.randn(n, device=dev, dtype=torch.bfloat16)
    xq, xs, _, _ = ops.rowquant(x, ops.MM_I8)
    ref = None
    line = f"M={m:6d} N={n:6d} K={k:6d}:"
    for t in tiles:
        lib.sdnq_hip_set_tile_override(t)
        y = ops.scaled_mm(ops.MM_I8, xq, b, xs, sb, bias, torch.bfloat16)
        if ref is None:
            ref = y
        ok = torch.equal(y.view(torch.int16), ref.view(torch.int16))
        us = timed(lambda: ops.scaled_mm(ops.MM_I8, xq, b, xs, sb, bias, torch.bfloat16))
        line += f"  tile {t:3d}: {us:8.2f} us{'' if ok else ' MISMATCH'}"
    lib.sdnq_hip_set_tile_override(-1)
    print(line, flush=True)
