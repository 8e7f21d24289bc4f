#!/usr/bin/env python3
"""HBM-side kernels on packed weights: dequantize, re-quantize, few-row fused linear -- algorithmic GB/s at FLUX / SDXL layer sizes
(graph-replayed launches, HIP events).  usage: bench_dequant.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sdnq_amd
from sdnq_amd import linear as L, ops
dev = torch.device("cuda:0")

def timed(fn, reps=20):
    s = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(s):
        fn(); s.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps): fn()
        g.replay(); s.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(3): g.replay()
        e1.record(s); s.synchronize()
    return e0.elapsed_time(e1) / (3 * reps) * 1e3

print("kernel                      dtype      N      K  group    us    GB/s (algorithmic bytes)")
for (wd, gs, bits, n, k) in (("int4", 64, 4, 12288, 3072), ("int4", 64, 4, 3072, 15360), ("uint4", 32, 4, 12288, 3072), ("int8", -1, 8, 12288, 3072),
                             ("float4_e2m1fn", 32, 4, 12288, 3072), ("int6", -1, 6, 12288, 3072), ("int3", 32, 3, 12288, 3072), ("int8", -1, 8, 1280, 1280)):
    lin = torch.nn.Linear(k, n, bias=True, device=dev, dtype=torch.bfloat16)
    mod, _ = sdnq_amd.sdnq_quantize_layer(lin, sdnq_amd.SDNQConfig(weights_dtype=wd, group_size=gs, use_quantized_matmul=False))
    st = L._state(mod)
    g = (k // gs) if gs > 0 else 1
    wbytes = n * k * bits / 8 + n * g * 4 * (2 if mod.zero_point is not None else 1)
    t = timed(lambda: ops.dequant(st.qw, torch.bfloat16))
    print(f"dequant -> bf16             {wd:14s} {n:6d} {k:6d} {gs:4d} {t:8.1f} {(wbytes + n * k * 2) / t / 1e3:8.1f}")
    t = timed(lambda: ops.requant(st.qw, ops.MM_I8))
    print(f"requant -> int8 + scale     {wd:14s} {n:6d} {k:6d} {gs:4d} {t:8.1f} {(2 * wbytes + n * k + 4 * n) / t / 1e3:8.1f}   (weights read twice)")
    x = torch.randn(1, k, device=dev, dtype=torch.bfloat16)
    t = timed(lambda: ops.linear_skinny(st.qw, x, mod.bias))
    print(f"linear_skinny M=1           {wd:14s} {n:6d} {k:6d} {gs:4d} {t:8.1f} {(wbytes + 2 * k + 4 * n) / t / 1e3:8.1f}")
