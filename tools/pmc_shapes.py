#!/usr/bin/env python3
"""One pass of a workload's M>=32 layers through the raw ops (rowquant + scaled_mm), with distinct weights per layer and
the model's activation sharing -- the smallest process that launches exactly the step's hot kernels, for rocprofv3
counter collection (--pmc serialises dispatches at ~50 ms each, so bench.py's layer construction is far too slow there).
usage: pmc_shapes.py [sdxl|sdxl_fp8|sdxl_dequant|linear|flux|flux_svd] [passes] [linked]   (sdxl_dequant: the default float mode -- bf16 activations,
int8 weight bytes dequantized inside the GEMM, sdnq_hip_linear_w8a16(_grouped); linear: the reference's micro-benchmark layer 16384 x 4096 -> 8192,
eight layers with weights of their own; sdxl_fp8: e4m3 operands on the fp8 MFMA; flux: the int8 GEMMs of
FLUX.1-dev -- what flux_int4_had runs after its weights were re-quantized; flux_svd: the same with the rank-32 low-rank epilogue of
flux_int8_svd; linked: attention projections that share their input run as ONE grouped launch
-- sdnq_hip_scaled_mm_grouped over the layers' own weights -- exactly as sdnq_amd.accelerate / bench.py do by default: q/k/v of a
self-attention block together, and ALL cross-attention k/v projections (one text tensor) as one model-wide group)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdnq_amd import ops, shapes
wl = sys.argv[1] if len(sys.argv) > 1 else "sdxl"
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 1
linked = len(sys.argv) > 3 and sys.argv[3] == "linked"
seq = shapes.sdxl_unet_layer_sequence() if wl.startswith("sdxl") else shapes.flux_dev_layer_sequence()
if wl == "linear":
    seq = [(f"layer{i}", 16384, 4096, 8192, True, f"x{i}") for i in range(8)]
MM = ops.MM_FP8 if wl == "sdxl_fp8" else ops.MM_I8
SVD = wl == "flux_svd"
FLOAT = wl == "sdxl_dequant"
dev = torch.device("cuda:0")
layers, inputs = [], {}
for (name, m, k, n, has_bias, key) in seq:
    if m < 32:
        continue
    if key not in inputs:
        inputs[key] = torch.randn(m, k, device=dev, dtype=torch.bfloat16)
    w = torch.randint(-128, 128, (n, k), dtype=torch.int8, device=dev)
    if MM == ops.MM_FP8:
        w = (torch.randn(n, k, device=dev) * 60).clamp(-448, 448).to(torch.float8_e4m3fn)
    layers.append((name, inputs[key], w, key, n, has_bias))
lr_t = {}  # flux_svd: t = x . svd_down^T per activation [M, 32] and one svd_up [N, 32] per width
lr_up = {}
sb = torch.rand(16384, device=dev) * 0.01
bias = torch.randn(16384, device=dev, dtype=torch.bfloat16)
is_proj = lambda nm: any(t in nm for t in (".to_q", ".to_k", ".to_v"))
# launch plan in execution order: ("single", layer) or ("group", first layer index, GemmGroup)
by_key = {}
if linked:
    for i, l in enumerate(layers):
        if is_proj(l[0]):
            by_key.setdefault(l[3], []).append(i)
group_of, plan = {}, []
for key, idxs in by_key.items():
    if len(idxs) > 1:
        members = [(layers[i][2], sb[:layers[i][4]].contiguous(), bias[:layers[i][4]].contiguous() if layers[i][5] else None) for i in idxs]
        g = ops.GemmGroup(members)
        for i in idxs:
            group_of[i] = (idxs[0], g)
for i, l in enumerate(layers):
    if i in group_of:
        if group_of[i][0] == i:
            plan.append(("group", i, group_of[i][1]))
    else:
        plan.append(("single", i, None))
torch.cuda.synchronize()
for _ in range(passes):
    quant = {}
    for (kind, i, g) in plan:
        name, x, w, key, n, has_bias = layers[i]
        if FLOAT:  # (no activation quantization in this mode)
            if kind == "single":
                ops.linear_w8a16(x, w, sb[:n], None, bias[:n] if has_bias else None)
            else:
                ops.linear_w8a16_grouped(x, g)
            continue
        if key not in quant:
            quant[key] = ops.rowquant(x, MM)
        q = quant[key]
        if kind == "single" and SVD:
            if key not in lr_t:
                lr_t[key] = torch.randn(x.shape[0], 32, device=dev, dtype=torch.bfloat16)
            if n not in lr_up:
                lr_up[n] = torch.randn(n, 32, device=dev, dtype=torch.bfloat16)
            ops.scaled_mm_lowrank(MM, q[0], w, q[1], sb[:n], bias[:n] if has_bias else None, lr_t[key], lr_up[n], None, None, torch.bfloat16)
        elif kind == "single":
            ops.scaled_mm(MM, q[0], w, q[1], sb[:n], bias[:n] if has_bias else None, torch.bfloat16)
        else:
            ops.scaled_mm_grouped(MM, q[0], q[1], g, torch.bfloat16)
torch.cuda.synchronize()
print(f"{wl}: {len(plan)} GEMM launches/pass, {len({l[3] for l in layers})} row quantizations/pass, {passes} passes" + (" (linked projections)" if linked else ""))
