#!/usr/bin/env python3
"""Kernel micro-benchmark (development aid): quantized attention forward, graph-replayed launches timed with HIP events.
Usage: python tools/bench_attention.py [bf16|f16]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdnq_amd import attention as A  # noqa: E402

dev = torch.device("cuda:0")
dt = torch.float16 if (len(sys.argv) > 1 and sys.argv[1] == "f16") else torch.bfloat16
# (batch, q_heads, kv_heads, q_len, kv_len, head_dim): SDXL self-attention at 1024 px (64x64 and 32x32 latents), SDXL cross-attention
# (77 text tokens), FLUX.1 joint attention (4096 image + 512 text tokens, 24 heads of 128)
SHAPES = [(1, 10, 10, 4096, 4096, 64), (1, 20, 20, 1024, 1024, 64), (1, 10, 10, 4096, 77, 64), (1, 20, 20, 1024, 77, 64), (1, 24, 24, 4608, 4608, 128),
          (2, 10, 10, 4096, 4096, 64)]


def timed(fn, reps=10):
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


for (z, qh, kh, qn, kn, d) in SHAPES:
    q = torch.randn(z, qh, qn, d, device=dev, dtype=dt)
    k = torch.randn(z, kh, kn, d, device=dev, dtype=dt)
    v = torch.randn(z, kh, kn, d, device=dev, dtype=dt)
    ops = 4.0 * z * qh * qn * kn * d
    parts = A.quantize_attn(q, k, v)
    tp = timed(lambda: A.quantize_attn(q, k, v))
    tf = timed(lambda: A.atten_fwd(*parts, kn, d ** -0.5, False, dt))
    parts16 = A.quantize_attn(q, k, v, with_query=False)
    tq = timed(lambda: A.atten_fwd(*parts16, kn, d ** -0.5, False, dt))  # Q quantized by the forward kernel, K / V prepared
    ts = timed(lambda: torch.nn.functional.scaled_dot_product_attention(q, k, v))
    tc = timed(lambda: A.sdnq_hip_atten(q, k, v))  # what a caller gets: the fused single launch for short key sequences
    print(f"Z={z} H={qh} QN={qn:5d} KN={kn:5d} D={d:3d}: prepare {tp:8.1f} us   fwd {tf:9.1f} us ({ops / tf / 1e6:7.1f} TOP/s)   fwd_q16 {tq:9.1f} us   "
          f"sdnq_hip_atten {tc:9.1f} us   torch sdpa ({str(dt)[6:]}) {ts:9.1f} us", flush=True)
