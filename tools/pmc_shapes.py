#!/usr/bin/env python3
"""One pass of a workload's M>=32 layers through the raw ops (rowquant + scaled_mm), with distinct weights per layer and
the model's activation sharing -- the smallest process that launches exactly the step's hot kernels, for rocprofv3
counter collection (--pmc serialises dispatches at ~50 ms each, so bench.py's layer construction is far too slow there).
usage: pmc_shapes.py [sdxl|flux] [passes] [linked]   (linked: attention projections that share their input run as one
sdnq_hip_scaled_mm_multi launch over the stacked weights, as sdnq_amd.accelerate / bench.py do by default)"""
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
linked = len(sys.argv) > 3 and sys.argv[3] == "linked"
names = [nm for (nm, m, k, n, hb, key) in seq if m >= 32]
groups, i = [], 0
while i < len(layers):  # (first layer index, member count, stacked weight)
    j = i + 1
    is_proj = lambda nm: any(t in nm for t in (".to_q", ".to_k", ".to_v"))
    while linked and j < len(layers) and layers[j][2] == layers[i][2] and j - i < 3 and is_proj(names[i]) and is_proj(names[j]) \
            and layers[j][3] == layers[i][3] and layers[j][4] == layers[i][4]:
        j += 1
    groups.append((i, j - i, torch.cat([layers[t][1] for t in range(i, j)], dim=0).contiguous() if j - i > 1 else layers[i][1]))
    i = j
torch.cuda.synchronize()
for _ in range(passes):
    last_key, q = None, None
    for (i, g, w) in groups:
        x, _, key, n, has_bias = layers[i]
        if key != last_key:
            q = ops.rowquant(x, ops.MM_I8)
            last_key = key
        if g == 1:
            ops.scaled_mm(ops.MM_I8, q[0], w, q[1], sb[:n], bias[:n] if has_bias else None, torch.bfloat16)
        else:
            ops.scaled_mm_multi(ops.MM_I8, q[0], w, q[1], sb[:g * n], bias[:g * n] if has_bias else None, torch.bfloat16, g)
torch.cuda.synchronize()
print(f"{wl}: {len(groups)} GEMM launches/pass, {passes} passes" + (" (linked projections)" if linked else ""))
