#!/usr/bin/env python3
"""Row-quantization micro-benchmark (with / without Hadamard) at FLUX / SDXL activation shapes. usage: bench_rowquant.py"""
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
for (m, k) in ((4608, 3072), (4096, 3072), (512, 3072), (4608, 12288), (4608, 15360), (1024, 1280), (4096, 2560)):
    x = torch.randn(m, k, device=dev, dtype=torch.bfloat16)
    for had in (0, 256):
        for xrot in (False, True):
            if xrot and not had: continue
            us = t(lambda: ops.rowquant(x, ops.MM_I8, had, want_xrot=xrot))
            byts = m * k * (3 + (2 if xrot else 0))
            print(f"M={m:5d} K={k:6d} had={had:3d} xrot={int(xrot)}: {us:8.2f} us  {byts/us/1e6:7.2f} TB/s")
