#!/usr/bin/env python3
"""Run one scaled_mm shape a few times (for rocprofv3 counter collection / clock probes).
usage: run_one_gemm.py M N K [reps] [tile-id|-1] [uniform|gauss|zeros]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdnq_amd import _lib, ops
m, n, k = (int(v) for v in sys.argv[1:4])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
tile = int(sys.argv[5]) if len(sys.argv) > 5 else -1
data = sys.argv[6] if len(sys.argv) > 6 else "uniform"
dev = torch.device("cuda:0")
if data == "uniform":
    a = torch.randint(-128, 128, (m, k), dtype=torch.int8, device=dev)
    b = torch.randint(-128, 128, (n, k), dtype=torch.int8, device=dev)
elif data == "gauss":  # what row-wise absmax quantization of Gaussian tensors gives: sigma ~ 127 / 4
    a = (torch.randn(m, k, device=dev) * 30).round().clamp(-127, 127).to(torch.int8)
    b = (torch.randn(n, k, device=dev) * 30).round().clamp(-127, 127).to(torch.int8)
else:
    a = torch.zeros((m, k), dtype=torch.int8, device=dev)
    b = torch.zeros((n, k), dtype=torch.int8, device=dev)
sa = torch.rand(m, device=dev) * 0.01
sb = torch.rand(n, device=dev) * 0.01
bias = torch.randn(n, device=dev, dtype=torch.bfloat16)
_lib.load().sdnq_hip_set_tile_override(tile)
ops.scaled_mm(ops.MM_I8, a, b, sa, sb, bias, torch.bfloat16)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    ops.scaled_mm(ops.MM_I8, a, b, sa, sb, bias, torch.bfloat16)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / reps * 1e3
print(f"M={m} N={n} K={k} tile={tile} data={data}: {us:.1f} us/launch, {2 * m * n * k / us / 1e6:.1f} TOP/s")
