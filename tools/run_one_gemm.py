#!/usr/bin/env python3
"""Run one scaled_mm shape a few times (for rocprofv3 counter collection). usage: run_one_gemm.py M N K [reps]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdnq_amd import ops
m, n, k = (int(v) for v in sys.argv[1:4])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
dev = torch.device("cuda:0")
x = torch.randn(m, k, device=dev, dtype=torch.bfloat16)
b = torch.randint(-128, 128, (n, k), dtype=torch.int8, device=dev)
sb = torch.rand(n, device=dev) * 0.01
bias = torch.randn(n, device=dev, dtype=torch.bfloat16)
xq, xs, _, _ = ops.rowquant(x, ops.MM_I8)
for _ in range(reps):
    ops.scaled_mm(ops.MM_I8, xq, b, xs, sb, bias, torch.bfloat16)
torch.cuda.synchronize()
