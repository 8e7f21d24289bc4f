#!/usr/bin/env python3
"""One pass of a workload's M>=32 layers through the raw ops (rowquant + scaled_mm), with distinct weights per layer and
the model's activation sharing -- the smallest process that launches exactly the step's hot kernels, for rocprofv3
counter collection (--pmc serialises dispatches at ~50 ms each, so bench.py's layer construction is far too slow there).
usage: pmc_shapes.py [sdxl|flux] [passes]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdnq_amd import ops, shapes
wl = sys.argv[1] if len(sys.argv) > 1 else "sdxl"
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 1
seq = shapes.sdxl_unet_layer_sequence() if wl == "sdxl" else shapes.flux_dev_layer_sequence()
dev = torch.device("cuda:0")
layers, inputs = [], {}
for (name, m, k, n, has_bias, key) in seq:
    if m < 32:
        continue
    if key not in inputs:
        inputs[key] = torch.randn(m, k, device=dev, dtype=torch.bfloat16)
    w = torch.randint(-128, 128, (n, k), dtype=torch.int8, device=dev)
    layers.append((inputs[key], w, key, n, has_bias))
sb = torch.rand(16384, device=dev) * 0.01
bias = torch.randn(16384, device=dev, dtype=torch.bfloat16)
torch.cuda.synchronize()
for _ in range(passes):
    last_key, q = None, None
    for (x, w, key, n, has_bias) in layers:
        if key != last_key:
            q = ops.rowquant(x, ops.MM_I8)
            last_key = key
        ops.scaled_mm(ops.MM_I8, q[0], w, q[1], sb[:n], bias[:n] if has_bias else None, torch.bfloat16)
torch.cuda.synchronize()
print(f"{wl}: {len(layers)} GEMM launches/pass, {passes} passes")
