#!/usr/bin/env python3
"""Fused dequantize GEMM (sdnq_hip_linear_w8a16) vs dequantize + float GEMM, per shape and tile (development aid)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdnq_amd import _lib, ops
dev = torch.device("cuda:0")
lib = _lib.load()
SHAPES = [(4096, 640, 640), (4096, 1920, 640), (4096, 5120, 640), (4096, 640, 2560), (1024, 1280, 1280), (1024, 3840, 1280), (1024, 10240, 1280),
          (1024, 1280, 5120), (77, 1280, 2048), (4096, 4096, 4096)]


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


print("# us per call; unfused = sdnq_hip_dequant + bf16 GEMM; fused tiles: 0 256x128 | 1 64x128 | 2 128x128 | 3 64x64")
for (m, n, k) in SHAPES:
    x = torch.randn(m, k, device=dev, dtype=torch.bfloat16)
    w = torch.randint(-128, 128, (n, k), dtype=torch.int8, device=dev)
    sc = torch.rand(n, device=dev) * 0.01 + 1e-4
    bias = torch.randn(n, device=dev, dtype=torch.bfloat16)
    wd = (w.float() * sc[:, None]).to(torch.bfloat16)
    lib.sdnq_hip_set_tile_override(-1)
    t_gemm = timed(lambda: ops.linear_float(x, wd, bias))
    ref = ops.linear_float(x, wd, bias)
    line = f"M={m:5d} N={n:6d} K={k:5d}: bf16 GEMM alone {t_gemm:7.2f} (+ dequant launch) | fused default {timed(lambda: ops.linear_w8a16(x, w, sc, None, bias)):7.2f} |"
    for t in range(5):
        lib.sdnq_hip_set_tile_override(t)
        out = ops.linear_w8a16(x, w, sc, None, bias)
        ok = torch.equal(out, ref)
        line += f" {t}:{timed(lambda: ops.linear_w8a16(x, w, sc, None, bias)):7.2f}{'' if ok else '~'}"
    lib.sdnq_hip_set_tile_override(-1)
    print(line, flush=True)
