#!/usr/bin/env python3
"""Timing of the float branch (dequantize -> linear_float) at BASELINE configs[0] sizes. usage: bench_float.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdnq_amd import ops
dev = torch.device("cuda:0")
def t(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for dt in (torch.bfloat16, torch.float16, torch.float32):
    for (m, k, n) in ((64, 4096, 4096), (4096, 4096, 4096), (1024, 1280, 1280), (4096, 640, 5120), (16384, 4096, 8192)):
        x = torch.randn(m, k, device=dev).to(dt); w = (torch.randn(n, k, device=dev) * 0.02).to(dt); b = torch.randn(n, device=dev).to(dt)
        us = t(lambda: ops.linear_float(x, w, b))
        print(f"{str(dt):15s} M={m:6d} K={k:6d} N={n:6d}: {us:9.2f} us  {2*m*k*n/us/1e6:8.1f} TFLOP/s")
