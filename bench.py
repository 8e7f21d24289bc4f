#!/usr/bin/env python3
"""Headline benchmark: quantized-matmul GOP/s + SDXL-UNet Linear step latency, int8 w8a8, bs=1, on N MI355X.

A "step" is ONE pass of the hot path over every quantized Linear of one SDXL-UNet denoising step
(BASELINE.json configs[1]: weights_dtype=int8, group_size=-1, use_quantized_matmul=True, bf16 activations):
722 int8 GEMM layers (M >= 32: fused row-quantize + MFMA scaled-mm) + 19 M=1 embedding layers (dequant +
float GEMM branch), synthetic weights (randn*0.02, quantized by this package) and synthetic bf16 activations,
all resident in HBM before the timed region.  Attention/conv/norm are NOT part of the step (the reference
only replaces Linear layers by default; quant_conv=False).  The step is replayed from a hipGraph so the
number measures the GPU, not the Python launch loop (the reference's own harness uses torch.compile).

    value       = algorithmic ops per step (reference formula 2*M*K*N + M*N*[bias]) * steps / time, whole job
    ms_per_step = SDXL-UNet Linear step latency;  tokens/s = 16384 latent tokens / step latency
    roofline    = the int8 MFMA scaled-mm kernel alone: algorithmic ops of all its launches in one step divided by
                  their summed duration, measured with HIP events on the launch stream, vs the dense int8 MFMA peak
                  (per workload: the matmul that actually runs -- int8 / fp8 scaled-mm, or the bf16 fused dequantize GEMM of
                  the reference's default mode -- against the dense peak of ITS dtype)
    cpu_baseline= the reference's CPU-eager path restated with the torch CPU operators it issues (row-quantize, torch._int_mm,
                  addcmul epilogue) on the host's physical cores, on a bounded sample of the same layer list; the C + OpenMP
                  oracle ("port_c_oracle") beside it; for the headline also BASELINE configs[0] (cfg1) on CPU and GPU
                  (rank 0, N=1 only)

N > 1: one rank per GPU.  The driver launches the ranks via torch.distributed.run; started plainly (`python bench.py --gpus N`, no
WORLD_SIZE in the environment) this script re-executes itself under torch.distributed.run with N ranks, and exits non-zero when
fewer than N GPUs are visible -- it never silently runs one.  Default = one independent latent per GPU (replicas, weak scaling,
no data-path collective).  ``--tp`` column-shards every Linear across the ranks and all-gathers the outputs over RCCL/xGMI
(strong scaling; see DESIGN.md for why that is link/latency-bound at bs=1).  ``--dry-run`` exercises the multi-rank plumbing
(rendezvous, barriers, max-over-ranks, the one JSON line) on CPU over gloo with NO device work: the line says so and is not a
measurement.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E ~8 TB/s
INT8_MFMA_PEAK_TOPS = 5033.0  # dense: 1024 MAC/clk/SIMD x 4 SIMD x 256 CU x 2.4 GHz x 2 (MI355X_MICROARCH.md: i8 = 2x bf16 rate)
FP8_MFMA_PEAK_TFLOPS = 5033.0
BF16_MFMA_PEAK_TFLOPS = 2516.6  # dense bf16 / f16: 512 MAC/clk/SIMD (MI355X_MICROARCH.md: ~2.5 PFLOP/s)
XGMI_LINK_GBPS = 153.0  # one xGMI link, one direction (guide); 7 links per GPU, point to point
# measured ceilings of the same instructions in a register-resident loop on this part (tools/mfma_peak.hip,
# profiles/r02_mfma_peak.txt): the issue rate is the datasheet's (one 32x32x32 i8 MFMA per 32 clocks per SIMD) but the part clocks
# to its power budget -- 2.0-2.3 GHz with small-integer operands (4.1-4.8 POP/s, the round-1 probe), 1.63-1.74 GHz with
# full-entropy operand bytes (3.3-3.5 POP/s), which is what a GEMM on quantized tensors feeds the pipe
INT8_MFMA_MEASURED_TOPS = 3490.0
BF16_MFMA_MEASURED_TFLOPS = 1945.0
HBM_PEAK_GBS = 8000.0


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--workload", default="sdxl_int8", choices=["sdxl_int8", "sdxl_fp8", "sdxl_int4", "flux_int4_had", "flux_int8_svd", "linear_int8", "sdxl_conv_int8", "sdxl_int8_dequant",
                            "sdxl_attn_int8", "flux_attn_int8", "sdxl_unet_all"])
    p.add_argument("--tp", action="store_true", help="column-shard every Linear across ranks + RCCL all-gather")
    p.add_argument("--tp-chunks", type=int, default=2, help="with --tp: M chunks per layer (gather of chunk i on a side stream under the matmul of chunk i + 1; 1 = plain)")
    p.add_argument("--tp-gather", choices=["rccl", "peer"], default="rccl",
                   help="with --tp: RCCL all-gather + re-assembly pass (default), or the copy-free gather -- every rank pushes its slab into every "
                        "rank's [M, N] output over peer-mapped memory (sdnq_amd.parallel.PeerArena, sdnq_hip_push_columns)")
    p.add_argument("--tp-report", action="store_true",
                   help="attach the column-sharded (strong-scaling) numbers of the same workload, RCCL and copy-free gather, to the default line; "
                        "ON by default when --gpus > 1 without --tp (SDNQ_BENCH_TP_REPORT=0 turns it off)")
    p.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay")
    p.add_argument("--launch", choices=["graph", "eager", "compile", "capture"], default=None,
                   help="how a step is launched: one captured hipGraph (default), eager Python (= --no-graph), or torch.compile(mode='reduce-overhead') "
                        "over the layer list (every SDNQ layer one sdnq_hip::layer_forward op; Inductor's own CUDA-graph trees do the replay)")
    p.add_argument("--activation-pool", type=int, default=0,
                   help="sensitivity variant: distinct activations of one shape rotate through this many buffers (0 = one buffer per activation, "
                        "every read cold from HBM; a model's activations were just written by their producer)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-seconds", type=float, default=10.0)
    p.add_argument("--fuse-projections", action="store_true",
                   help="variant: q/k/v (and cross-attention k/v) projections fused into one layer each, like diffusers' fuse_projections()")
    p.add_argument("--no-link-projections", action="store_true",
                   help="do not link attention projections that share their input (sdnq_amd.accelerate links them by default)")
    p.add_argument("--layers-scale", type=float, default=1.0, help="debug: fraction of each layer's repeat count")
    p.add_argument("--profile-markers", action="store_true",
                   help="bracket the TIMED steps with two marker kernels (torch.cuda._sleep: `spin_kernel`) so that tools/prof_window.py can cut "
                        "exactly those steps out of a rocprofv3 kernel trace (build, warm-up, roofline replays and RNG kernels excluded)")
    p.add_argument("--dry-run", action="store_true",
                   help="multi-rank plumbing only, on CPU over gloo, NO device work (for tests): the JSON line is marked as such")
    return p.parse_args()


def workload_config(name: str):
    from sdnq_amd import shapes
    if name == "sdxl_int8":
        return shapes.sdxl_unet_layer_sequence(), dict(weights_dtype="int8", group_size=-1, use_quantized_matmul=True), "int8", 16384
    if name == "sdxl_fp8":
        return shapes.sdxl_unet_layer_sequence(), dict(weights_dtype="fp8", quantized_matmul_dtype="fp8", group_size=-1,
                                                use_quantized_matmul=True), "fp8", 16384
    if name == "sdxl_int4":  # N1's question at SDXL sizes: 4-bit group-wise weights on the int8 matmul (re_quantize_for_matmul: dequantizer.py:166-239)
        return shapes.sdxl_unet_layer_sequence(), dict(weights_dtype="uint4", use_quantized_matmul=True), "int8", 16384
    if name == "flux_int4_had":
        return shapes.flux_dev_layer_sequence(), dict(weights_dtype="int4", use_hadamard=True, hadamard_group_size=256,
                                               use_quantized_matmul=True), "int8", 4608
    if name == "flux_int8_svd":
        return shapes.flux_dev_layer_sequence(), dict(weights_dtype="int8", group_size=-1, use_svd=True, svd_rank=32,
                                               use_quantized_matmul=True), "int8", 4608
    if name == "sdxl_int8_dequant":  # the reference's DEFAULT mode (use_quantized_matmul=False): dequantize + bf16 matrix-core GEMM per call
        return shapes.sdxl_unet_layer_sequence(), dict(weights_dtype="int8", group_size=-1, use_quantized_matmul=False), "int8", 16384
    if name == "sdxl_conv_int8":  # the UNet's Conv2d layers through the same int8 matmul (use_quantized_matmul_conv)
        return shapes.sdxl_unet_convs(), dict(weights_dtype="int8", group_size=-1, quant_conv=True, use_quantized_matmul_conv=True), "int8", 16384
    if name == "linear_int8":  # the reference's own micro-benchmark shape (scripts/benchmark_sdnq_inference_matmul.py)
        return [("bench.linear", 16384, 4096, 8192, True, 1)], dict(weights_dtype="int8", group_size=-1, use_quantized_matmul=True), "int8", 16384
    raise ValueError(name)


def expand_layers(shape_list, scale=1.0):
    """-> execution-ordered [(name, M, K, N, bias, input_key)].  Aggregated lists (name, M, K, N, bias, repeat) are expanded
    with one distinct activation tensor per layer instance."""
    seq = []
    for i, e in enumerate(shape_list):
        if isinstance(e[5], str):
            seq.append(e)
        else:
            for r in range(max(1, int(round(e[5] * scale)))):
                seq.append((e[0], e[1], e[2], e[3], e[4], f"{e[0]}#{r}"))
    if scale != 1.0 and shape_list and isinstance(shape_list[0][5], str):
        seq = seq[: max(1, int(round(len(seq) * scale)))]
    return seq


def build_layers(shape_list, cfg_kwargs, device, scale=1.0, tp_rank=0, tp_world=1, seed=0, tp_chunks=1, activation_pool=0, tp_peer=None):
    """-> list of (name, module, x, M, K, N, bias). One module per layer instance (distinct weights, like a real UNet /
    DiT); one activation tensor per distinct input_key: layers that consume the same tensor in the real model (q/k/v
    projections, every cross-attention k/v) get the SAME tensor object here, all others get their own."""
    import sdnq_amd
    g = torch.Generator(device=device).manual_seed(seed)
    layers, inputs, pools = [], {}, {}
    for (name, m, k, n, has_bias, key) in expand_layers(shape_list, scale):
        if key not in inputs:
            if activation_pool > 0:
                # sensitivity variant (--activation-pool P): distinct activations of one shape rotate through P buffers, each a SEPARATE
                # tensor object over shared storage (identity-keyed reuse still sees distinct tensors).  In a model an activation was
                # written by the producing kernel microseconds before the Linear reads it (L2 / MALL resident); the default -- one
                # buffer per activation, 1.2 GB cycled per step -- reads every one of them cold from HBM.
                pool = pools.setdefault((m, k), [])
                if len(pool) < activation_pool:
                    pool.append([torch.randn(m, k, device=device, dtype=torch.bfloat16, generator=g), 0])
                slot = pool[sum(c for _, c in pool) % activation_pool] if len(pool) == activation_pool else pool[-1]
                slot[1] += 1
                inputs[key] = slot[0].view(m, k)
            else:
                inputs[key] = torch.randn(m, k, device=device, dtype=torch.bfloat16, generator=g)
        x = inputs[key]
        assert x.shape == (m, k), (name, key, x.shape, m, k)
        lin = torch.nn.Linear(k, n, bias=has_bias, device=device, dtype=torch.bfloat16)
        with torch.no_grad():
            lin.weight.copy_(torch.randn(n, k, device=device, generator=g) * 0.02)
            if has_bias:
                lin.bias.copy_(torch.randn(n, device=device, generator=g) * 0.1)
        cfg = sdnq_amd.SDNQConfig(**cfg_kwargs)
        mod, _ = sdnq_amd.sdnq_quantize_layer(lin, cfg)
        if tp_world > 1 and m >= 32:
            # tensor parallel: every rank holds the same quantized "checkpoint" layer and takes its slab of output channels
            # (views of the stored tensors, sdnq_amd.parallel.column_shard_module); M = 1 embedding layers stay replicated
            from sdnq_amd.parallel import column_shard_module
            mod = column_shard_module(mod, tp_rank, tp_world, chunks=tp_chunks, peer=tp_peer)
        layers.append((name, mod, x, m, k, n, has_bias))
    return layers


def build_conv_layers(conv_list, cfg_kwargs, device, seed=0):
    """-> list of (name, module, x [1,C,H,W], M, K, N, bias) for sdxl_unet_convs entries (distinct weights and inputs per layer)."""
    import sdnq_amd
    from sdnq_amd import shapes
    g = torch.Generator(device=device).manual_seed(seed)
    layers = []
    for e in conv_list:
        name, cin, h, w, cout, k, stride, pad = e
        conv = torch.nn.Conv2d(cin, cout, k, stride=stride, padding=pad, bias=True, device=device, dtype=torch.bfloat16)
        with torch.no_grad():
            conv.weight.copy_(torch.randn(conv.weight.shape, device=device, generator=g) * 0.02)
            conv.bias.copy_(torch.randn(cout, device=device, generator=g) * 0.1)
        mod, _ = sdnq_amd.sdnq_quantize_layer(conv, sdnq_amd.SDNQConfig(**cfg_kwargs))
        x = torch.randn(1, cin, h, w, device=device, dtype=torch.bfloat16, generator=g)
        m, kk, n = shapes.conv_gemm_dims(e)
        layers.append((name, mod, x, m, kk, n, True))
    return layers


def fuse_shared_input_layers(layers):
    """--fuse-projections: what sdnq_amd.fuse_projections does to a diffusers model -- consecutive layers that consume the SAME
    tensor (self-attention q/k/v, cross-attention k/v) become one layer with the concatenated output channels."""
    from sdnq_amd import loader
    out, i = [], 0
    while i < len(layers):
        j = i + 1
        is_proj = lambda nm: any(t in nm for t in (".to_q", ".to_k", ".to_v", ".qkv"))  # noqa: E731  (attention projections only)
        while j < len(layers) and layers[j][2] is layers[i][2] and j - i < 3 and is_proj(layers[i][0]) and is_proj(layers[j][0]):
            j += 1
        group = layers[i:j]
        if len(group) > 1 and loader._fusable([g[1] for g in group]):
            name, _, x, m, k, _, has_bias = group[0]
            out.append((name + "+%d" % (len(group) - 1), loader._concat_linears([g[1] for g in group]), x, m, k, sum(g[5] for g in group), has_bias))
        else:
            out.extend(group)
        i = j
    return out


def link_shared_input_layers(layers):
    """What sdnq_amd.accelerate(model) does to a diffusers model (loader.link_projections): the attention projections that consume
    the SAME tensor become a ProjectionGroup -- to_q / to_k / to_v of each self-attention block, and ALL cross-attention to_k /
    to_v of the model (every one of them reads the one encoder_hidden_states tensor) as one model-wide group.  The modules and
    their outputs stay what they were; the first member called in a step runs one grouped scaled matmul for all of them."""
    import sdnq_amd
    is_proj = lambda nm: any(t in nm for t in (".to_q", ".to_k", ".to_v"))  # noqa: E731
    by_input = {}
    for l in layers:
        if is_proj(l[0]):
            by_input.setdefault(id(l[2]), []).append(l[1])
    return sum(1 for mods in by_input.values() if len(mods) > 1 and sdnq_amd.link_layers(mods))


@torch.no_grad()
def run_step(layers):
    """One pass over every layer (under no_grad, the state every inference pipeline runs its model in -- the reference decorates each forward
    with its inference context).  The activation-quantization cache is emptied first: within a step a tensor consumed by
    several layers is quantized once (sdnq_amd/linear.py:_ActivationCache), but nothing is carried across steps."""
    from sdnq_amd import linear as L
    L.clear_activation_cache()
    out = None
    for (_, mod, x, *_rest) in layers:
        out = mod(x)
    L.join_weight_pipeline()  # per-call mode: a weight prefetch nobody consumed re-joins the stream (no-op otherwise)
    return out


def resident_weight_bytes(layers):
    """Bytes the quantized layers keep resident in HBM: stored parameters + whatever the forwards cached on the module (the int8 / fp8
    matmul operand of a re-quantized layer, row scales) + the per-call mode's two scratch buffers."""
    from sdnq_amd import linear as L
    seen, stored, cached = set(), 0, 0

    def add(t):
        if t is None or not isinstance(t, torch.Tensor):
            return 0
        key = (t.untyped_storage().data_ptr(), t.storage_offset(), t.numel())
        if key in seen:
            return 0
        seen.add(key)
        return t.numel() * t.element_size()
    for (_, mod, *_r) in layers:
        inner = getattr(mod, "local", mod)
        for name in ("weight", "scale", "zero_point", "svd_up", "svd_down", "bias"):
            stored += add(getattr(inner, name, None))
        st = inner.__dict__.get("_sdnq_hip_state")
        if st is not None:
            for name in ("mm_weight", "mm_scale", "mm_zp", "mm_wcs", "wd", "svd_down_t"):
                cached += add(getattr(st, name, None))
            if isinstance(getattr(st, "lut", None), tuple):  # the fused 4-bit route's tables + row scales (per-call mode)
                cached += add(st.lut[0]) + add(st.lut[1])
    return {"stored_parameters": stored, "cached_matmul_operands": cached, "per_call_scratch": L._weight_pipeline.scratch_bytes(),
            "total": stored + cached + L._weight_pipeline.scratch_bytes()}


def time_gemm_kernel(layers, mm_name, device):
    """Dominant-kernel roofline: every M>=32 layer's scaled-mm launch alone, back to back on one stream, HIP events."""
    from sdnq_amd import linear as L
    from sdnq_amd import ops
    mm = ops.MM_I8 if mm_name == "int8" else ops.MM_FP8
    calls, total_ops, total_bytes, seen_groups = [], 0, 0, set()
    unit_w = []  # per launch: the weight tensors it reads
    if not L.CACHE_WEIGHTS and L.PIPELINE_WEIGHTS:
        return None  # per-call mode: every layer's operand lives in one of two scratch buffers, they cannot all be held at once
    for (_, mod, x, m, k, n, has_bias) in layers:
        if m < 32 or not hasattr(mod, "sdnq_dequantizer"):
            continue
        dq = mod.sdnq_dequantizer
        has_svd = bool(dq.svd_rank) and getattr(mod, "svd_up", None) is not None
        group = mod.__dict__.get("_sdnq_group") if L.LINK_PROJECTIONS else None
        group = group[0] if group is not None else None
        if group is not None:  # linked projections: ONE launch for the members, exactly as in the step
            if id(group) in seen_groups:
                continue
            seen_groups.add(id(group))
            if not group._operands(mm):
                raise RuntimeError("linked group without a unit table")
            xq, xs, _, _ = ops.rowquant(x, mm, 0)
            calls.append((xq, group.gemm, xs, None, None, len(group.mods)))
            unit_w.append(list(group.pf_tensors))
            for gm in group.mods:
                gn, gb = gm.sdnq_dequantizer.out_features, gm.bias is not None
                total_ops += 2 * m * k * gn + (m * gn if gb else 0)
                total_bytes += gn * k + 2 * m * gn + 4 * gn + (2 * gn if gb else 0)
            total_bytes += m * k + 4 * m
            continue
        st = L._state(mod)
        wq, ws, zp = L._prepare_mm_weights(mod, st, mm)
        if x.ndim == 4:  # conv layer: the GEMM sees the unfolded input
            from sdnq_amd import conv as C
            x = C._unfold(mod, x)[0]
        had = dq.hadamard_group_size if dq.use_hadamard else 0
        xq, xs, rowsum, xrot = ops.rowquant(x, mm, had, want_rowsum=zp is not None, want_xrot=has_svd and bool(had))
        if has_svd or zp is not None:  # the scaled matmul with the low-rank / zero-point epilogue, its t = x . svd_down^T precomputed
            t = ops.lowrank_down(xrot if xrot is not None else x.reshape(-1, k), st.svd_down) if has_svd else None
            calls.append((xq, wq, xs, ws, mod.bias, ("lowrank", t, st.svd_up if has_svd else None, rowsum, zp)))
            unit_w.append([wq])
            r = int(st.svd_up.shape[1]) if has_svd else 0
            total_ops += 2 * m * n * r
            total_bytes += 2 * r * (m + n)
        else:
            calls.append((xq, wq, xs, ws, mod.bias, 1))
            unit_w.append([wq])
        total_ops += 2 * m * k * n + (m * n if has_bias else 0)
        total_bytes += m * k + n * k + 2 * m * n + 4 * (m + n) + (2 * n if has_bias else 0)  # xq + Wq + y(bf16) + xs + ws + bias
    if not calls:
        return None

    # the launches carry the weight-prefetch hints they carry in the step (linear._PrefetchChain: the next two launch units' weights)
    ranges = [L._LaunchUnit(ts).ranges if ts else () for ts in unit_w]
    lib = ops._lib.load()

    def launch_all():
        for i, (xq, wq, xs, ws, bias, g) in enumerate(calls):
            if L.PREFETCH_NEXT:
                rs = (ranges[i + 1] if i + 1 < len(calls) else ()) + (ranges[i + 2] if i + 2 < len(calls) else ())
                if rs:
                    rs = (tuple(rs) + ((0, 0),) * 4)[:4]
                    lib.sdnq_hip_prefetch_hint(rs[0][0], rs[0][1], rs[1][0], rs[1][1], rs[2][0], rs[2][1], rs[3][0], rs[3][1])
            if g == 1:
                ops.scaled_mm(mm, xq, wq, xs, ws, bias, torch.bfloat16)
            elif isinstance(g, tuple):
                ops.scaled_mm_lowrank(mm, xq, wq, xs, ws, bias, g[1], g[2], g[3], g[4], torch.bfloat16)
            else:
                ops.scaled_mm_grouped(mm, xq, xs, wq, torch.bfloat16)
    launch_all()
    torch.cuda.synchronize()
    s = torch.cuda.Stream(device=device)
    reps = 5
    with torch.cuda.stream(s):
        graph = torch.cuda.CUDAGraph()
        launch_all()
        s.synchronize()
        with torch.cuda.graph(graph, stream=s):
            launch_all()
        graph.replay()
        s.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(reps):
            graph.replay()
        e1.record(s)
        s.synchronize()
    dur_s = e0.elapsed_time(e1) / 1e3 / reps

    # per shape class (round 6, verdict item 5b): the launches of one M x N x K replayed alone, back to back, so that the shape furthest
    # below the roof is named by the bench line itself; tile / workgroups from the library's own dry run (sdnq_hip_scaled_mm_tile)
    import ctypes
    classes = {}
    for i, c in enumerate(calls):
        xq, wq, xs, ws, bias, g = c
        m_, k_ = int(xq.shape[0]), int(xq.shape[1])
        if isinstance(g, tuple):
            n_, kind = int(ws.shape[0]), "low-rank / zero-point epilogue"
        elif g == 1:
            n_, kind = int(ws.shape[0]), "plain"
        else:  # (wq is the ops.GemmGroup of the linked projections)
            n_, kind = int(wq.n_total), f"grouped launch of {g} layers"
        classes.setdefault((m_, n_, k_, kind), []).append(i)
    table = []
    cus = torch.cuda.get_device_properties(device).multi_processor_count
    peak = INT8_MFMA_PEAK_TOPS if mm_name == "int8" else FP8_MFMA_PEAK_TFLOPS
    for (m_, n_, k_, kind), idxs in sorted(classes.items(), key=lambda kv: -len(kv[1]) * kv[0][0] * kv[0][1] * kv[0][2]):
        def launch_class():
            for i in idxs:
                xq, wq, xs, ws, bias, g = calls[i]
                if g == 1:
                    ops.scaled_mm(mm, xq, wq, xs, ws, bias, torch.bfloat16)
                elif isinstance(g, tuple):
                    ops.scaled_mm_lowrank(mm, xq, wq, xs, ws, bias, g[1], g[2], g[3], g[4], torch.bfloat16)
                else:
                    ops.scaled_mm_grouped(mm, xq, xs, wq, torch.bfloat16)
        rep_c = max(1, 64 // len(idxs))
        with torch.cuda.stream(s):
            launch_class()
            s.synchronize()
            gcls = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gcls, stream=s):
                for _ in range(rep_c):
                    launch_class()
            gcls.replay()
            s.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s)
            for _ in range(3):
                gcls.replay()
            e1.record(s)
            s.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / (3 * rep_c * len(idxs))
        bm, bn, thr, wgs = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int64(0)
        tile = None
        if kind == "plain":
            if lib.sdnq_hip_scaled_mm_tile(mm, 1, 1, m_, n_, k_, ctypes.byref(bm), ctypes.byref(bn), ctypes.byref(thr), ctypes.byref(wgs)) == 0:
                tile = {"tile": f"{bm.value}x{bn.value}", "threads": thr.value, "workgroups": wgs.value, "cus_used": min(cus, wgs.value)}
        ops_l = 2.0 * m_ * n_ * k_
        table.append({"shape": f"{m_}x{n_}x{k_}", "kind": kind, "launches_per_step": len(idxs), "avg_us": round(us, 2),
                      "frac": round(ops_l / (us * 1e-6) / 1e12 / peak, 4) if n_ else None, **(tile or {})})
    return {"launches": len(calls), "ops": total_ops, "bytes": total_bytes, "seconds": dur_s, "per_shape": table}


def time_float_kernel(layers, device):
    """Dominant-kernel roofline of the DEFAULT mode (use_quantized_matmul=False): every M >= 32 layer is ONE launch of the fused
    dequantize GEMM (int8 weight bytes converted between LDS and the bf16 MFMA; no activation pre-pass exists), linked projections
    one grouped launch -- so the kernel's launches are the step without its M < 32 layers.  Graph-replayed, HIP events on the
    launch stream."""
    from sdnq_amd import linear as L
    big = [l for l in layers if l[3] >= 32 and hasattr(l[1], "sdnq_dequantizer")]
    if not big:
        return None
    total_ops = sum(2 * m * k * n + (m * n if b else 0) for (_, _, _, m, k, n, b) in big)
    # x (bf16) + W (1 B / weight) + y (bf16) + row scales + bias; a shared activation is counted once per launch that reads it
    total_bytes, launches, seen_groups = 0, 0, set()
    for (_, mod, x, m, k, n, b) in big:
        group = mod.__dict__.get("_sdnq_group") if L.LINK_PROJECTIONS else None
        gid = id(group[0]) if group is not None else None
        total_bytes += n * k + 2 * m * n + 4 * n + (2 * n if b else 0)
        if gid is None or gid not in seen_groups:
            launches += 1
            total_bytes += 2 * m * k
            if gid is not None:
                seen_groups.add(gid)
    run_step(big)
    torch.cuda.synchronize()
    s = torch.cuda.Stream(device=device)
    reps = 5
    with torch.cuda.stream(s):
        run_step(big)
        s.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=s):
            run_step(big)
        graph.replay()
        s.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(reps):
            graph.replay()
        e1.record(s)
        s.synchronize()
    return {"launches": launches, "ops": total_ops, "bytes": total_bytes, "seconds": e0.elapsed_time(e1) / 1e3 / reps}


def time_all_gathers(layers, device, reps=3):
    """--tp: the all-gathers of one step ALONE (same message sizes, same process group, no matmuls), back to back on the current
    stream -> seconds per step and bytes RECEIVED per rank: what the xGMI links deliver, to be read against 153 GB/s per link."""
    import torch.distributed as dist
    msgs = []
    for (_, mod, x, m, k, n, b) in layers:
        if type(mod).__name__ != "ColumnShardedLinear":
            continue
        wmax = max(bb - aa for aa, bb in mod.bounds)
        msgs.append((m, wmax, mod.world))
    if not msgs:
        return None
    bufs = {}
    for (m, w, world) in set(msgs):
        bufs[(m, w, world)] = (torch.empty((m, w), device=device, dtype=torch.bfloat16), torch.empty((world * m, w), device=device, dtype=torch.bfloat16))

    def run():
        for key in msgs:
            send, recv = bufs[key]
            dist.all_gather_into_tensor(recv, send)
    run()
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        run()
    torch.cuda.synchronize()
    sec = (time.perf_counter() - t0) / reps
    recv_bytes = sum(2 * m * w * (world - 1) for (m, w, world) in msgs)
    return {"gathers_per_step": len(msgs), "seconds_per_step": sec, "bytes_received_per_rank_per_step": recv_bytes,
            "largest_message_bytes": max(2 * m * w for (m, w, _) in msgs)}


def time_peer_gathers(layers, peer, device, reps=3):
    """The copy-free gathers of one step ALONE (PeerArena.gather on slabs of the step's sizes, no matmuls): seconds per step."""
    import torch.distributed as dist
    msgs = []
    for (_, mod, x, m, k, n, b) in layers:
        if type(mod).__name__ == "ColumnShardedLinear":
            a, bnd = mod.bounds[mod.rank]
            msgs.append((m, bnd - a, mod.n_total, a, mod.world))
    if not msgs:
        return None
    slabs = {(m, w): torch.zeros((m, w), device=device, dtype=torch.bfloat16) for (m, w, _, _, _) in msgs}

    def run():
        for (m, w, n_total, a, _) in msgs:
            peer.gather(slabs[(m, w)], n_total, a)
    run()
    peer.check()
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        run()
    torch.cuda.synchronize()
    sec = (time.perf_counter() - t0) / reps
    peer.check()
    return {"gathers_per_step": len(msgs), "seconds_per_step": sec,
            "bytes_received_per_rank_per_step": sum(2 * m * (n_total - w) for (m, w, n_total, _, _) in msgs),
            "largest_message_bytes": max(2 * m * w for (m, w, _, _, _) in msgs)}


def tp_report(args, device, world, rank, shape_list, cfg_kwargs, ops_per_step):
    """The SAME workload column-sharded over the ranks (strong scaling), both gathers, a few steps each -- attached to the default
    (replica) line of a multi-GPU run so that one driver run per N yields the replica number AND the sharded numbers north_star names.
    Every rank takes part; every mode is guarded (a failing mode reports its error, the replica line is printed regardless)."""
    import torch.distributed as dist
    out = {"ranks": world, "scaling": "strong", "workload": args.workload, "xgmi_link_gbps": XGMI_LINK_GBPS, "xgmi_links_used": max(1, world - 1),
           "note": "ms_per_step: max over ranks of the whole step (row quantization replicated, GEMMs on N / W channels, one gather per layer); "
                   "gather_*: the step's gathers run alone, bytes RECEIVED per rank against (W - 1) links x 153 GB/s"}
    steps = max(2, min(args.steps, 5))
    for mode in ("rccl", "peer"):
        res = {}
        try:
            peer = None
            if mode == "peer":
                from sdnq_amd.parallel import PeerArena
                peer = PeerArena.try_create(rank, world, device=device)
                if peer is None:
                    out[mode] = {"unavailable": "no peer access between every pair of ranks' devices (or more than one host): the RCCL gather is the path"}
                    continue
                res["control_words"] = peer.ctrl_kind
            layers = build_layers(shape_list, cfg_kwargs, device, scale=args.layers_scale, tp_rank=rank, tp_world=world, seed=0,
                                  tp_chunks=args.tp_chunks if mode == "rccl" else 1, tp_peer=peer)
            for _ in range(2):
                run_step(layers)
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                run_step(layers)
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()
            t = torch.tensor([time.perf_counter() - t0], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            if peer is not None:
                peer.check()
            ms = float(t.item()) / steps * 1e3
            res.update({"ms_per_step": round(ms, 4), "steps": steps, "launch": "eager", "value_gops": round(ops_per_step / (ms / 1e3) / 1e9, 1),
                        "sharded_layers": sum(1 for l in layers if type(l[1]).__name__ == "ColumnShardedLinear")})
            g = time_all_gathers(layers, device) if mode == "rccl" else time_peer_gathers(layers, peer, device)
            if g:
                gbps = g["bytes_received_per_rank_per_step"] / g["seconds_per_step"] / 1e9 if g["bytes_received_per_rank_per_step"] else 0.0
                res.update({"gathers_per_step": g["gathers_per_step"], "gather_bytes_received_per_rank_per_step": g["bytes_received_per_rank_per_step"],
                            "gathers_alone_ms_per_step": round(g["seconds_per_step"] * 1e3, 4), "gather_achieved_gbps_per_rank": round(gbps, 1),
                            "gather_frac_of_links": round(gbps / (XGMI_LINK_GBPS * max(1, world - 1)), 4)})
            del layers
            torch.cuda.empty_cache()
        except Exception as e:  # noqa: BLE001
            res["error"] = repr(e)[:400]
        out[mode] = res
    return out


def cpu_baseline(shape_list, mm_name, budget_s):
    """The oracle (a C + OpenMP port of the reference's CPU-eager semantics) timed on a bounded sample of the layer list."""
    import numpy as np
    from oracle import oracle as O
    L = O.lib()
    cores = L.orc_num_threads()
    rng = np.random.default_rng(0)
    done_ops, t_total, sample = 0, 0.0, []
    # distinct GEMM shapes of the step, weighted like the step (layer counts), cycled until the time budget is used
    seen = sorted({(e[1], e[2], e[3], e[4]) for e in shape_list if e[1] >= 32}, key=lambda s: s[0] * s[1] * s[2])
    data = {}
    for (m, k, n, b) in seen:
        x = O.round_dtype(rng.standard_normal((m, k), dtype=np.float32), "bf16")
        w = rng.integers(-127, 128, size=(n, k), dtype=np.int8)
        ws = (rng.random(n, dtype=np.float32) * 0.01 + 1e-4).astype(np.float32)
        bias = O.round_dtype(rng.standard_normal(n, dtype=np.float32), "bf16") if b else None
        data[(m, k, n, b)] = (x, w, ws, bias)
    passes = 0
    while t_total < budget_s:
        for key in seen:
            m, k, n, b = key
            x, w, ws, bias = data[key]
            t0 = time.perf_counter()
            xq, xs, _ = O.rowquant(x, "int8" if mm_name == "int8" else "fp8")
            if mm_name == "int8":
                O.scaled_mm("int8", xq, w, xs, ws, bias, "bf16")
            else:
                O.scaled_mm("fp8", xq, w.view(np.uint8) & 0x7e, xs, ws, bias, "bf16")
            t_total += time.perf_counter() - t0
            done_ops += 2 * m * k * n + (m * n if b else 0)
        passes += 1
    sample = [f"{m}x{k}x{n}" for (m, k, n, b) in seen]
    return {"value": round(done_ops / t_total / 1e9, 2), "unit": "GOP/s", "cores": cores, "kind": "port",
            "sample": f"row-quantize + {mm_name} scaled-mm over the step's {len(sample)} distinct GEMM shapes (MxKxN " + ",".join(sample) + f"), {passes} passes, {t_total:.1f}s"}


def _physical_cores() -> int:
    """Physical cores of this host (SMT siblings counted once), from the kernel's topology files; os.cpu_count() if unreadable."""
    seen = set()
    try:
        base = "/sys/devices/system/cpu"
        for d in os.listdir(base):
            if d.startswith("cpu") and d[3:].isdigit():
                f = os.path.join(base, d, "topology", "thread_siblings_list")
                if os.path.exists(f):
                    seen.add(open(f).read().strip())
    except OSError:
        pass
    return len(seen) if seen else (os.cpu_count() or 1)


def cpu_baseline_torch_eager(shape_list, mm_name, budget_s):
    """The reference's CPU-EAGER path of the quantized matmul restated with the torch CPU operators it issues -- quantize_int_mm_input
    (linear_int8.py:15-22 -> quant_utils.py:265-273: amax, divide, round, clamp, cast), torch._int_mm (kernel_wrappers.py:107) and the
    addcmul epilogue (kernel_wrappers.py:132-136) -- on a bounded sample of the step's layer list, torch threads pinned to the
    PHYSICAL core count.  This is what a user of the reference gets on this host's CPU; the C oracle ("port") is 10-15x slower at
    the matmul (a plain triple loop, the checker's job is to be obviously right)."""
    import torch
    phys = _physical_cores()
    old = torch.get_num_threads()
    torch.set_num_threads(phys)
    try:
        g = torch.Generator().manual_seed(0)
        seen = sorted({(e[1], e[2], e[3], e[4]) for e in shape_list if e[1] >= 32}, key=lambda s: s[0] * s[1] * s[2])
        data = {}
        for (m, k, n, b) in seen:
            x = torch.randn(m, k, generator=g).to(torch.bfloat16)
            w = torch.randint(-127, 128, (n, k), dtype=torch.int8, generator=g)
            ws = torch.rand(1, n, generator=g) * 0.01 + 1e-4
            bias = torch.randn(n, generator=g).to(torch.bfloat16) if b else None
            data[(m, k, n, b)] = (x, w.t(), ws, bias)  # weight as the reference holds it: logical [K, N], strides (1, K)
        done_ops, t_total, passes = 0, 0.0, 0
        while t_total < budget_s:
            for key in seen:
                m, k, n, b = key
                x, wt, ws, bias = data[key]
                t0 = time.perf_counter()
                xf = x.to(torch.float32)
                xs = torch.amax(xf.abs(), dim=-1, keepdims=True).div_(127)
                xq = torch.div(xf, xs).round_().clamp_(-128, 127).to(torch.int8)
                acc = torch._int_mm(xq, wt)
                y = acc.to(torch.float32).mul_(xs)
                y = torch.addcmul(bias, y, ws) if bias is not None else y.mul_(ws)
                y = y.to(torch.bfloat16)
                t_total += time.perf_counter() - t0
                done_ops += 2 * m * k * n + (m * n if b else 0)
            passes += 1
    finally:
        torch.set_num_threads(old)
    return {"value": round(done_ops / t_total / 1e9, 2), "unit": "GOP/s", "cores": phys, "threads": phys, "host_logical_cpus": os.cpu_count(),
            "kind": "port", "impl": "torch_eager",
            "sample": f"torch CPU restatement of the reference's eager path (row-quantize, torch._int_mm, addcmul epilogue) over the step's {len(seen)} "
                      f"distinct GEMM shapes, {passes} passes, {t_total:.1f}s, torch threads = physical cores ({phys} of {os.cpu_count()} logical CPUs)"}


def _cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline_cfg1(budget_s: float = 12.0):
    """SURVEY 8(d) / BASELINE configs[0]: ONE 4096 x 4096 Linear, int8 row-wise, use_quantized_matmul=False, fp32 -- the
    reference's CPU-eager path is dequantize (f32(w) * scale) + fp32 matmul -- timed on this box's host cores at M in {1, 64, 4096},
    with every core and with one core.  Two restatements of that path:
      * "port": the C + OpenMP oracle (oracle/sdnq_oracle.c: orc_dequant_f32 + orc_linear_float), the checker of the parity tests;
      * "torch_eager": the same two steps as the torch CPU ops the reference's eager path issues (w.to(f32).mul(scale), F.linear).
    Bounded: a case that would exceed its share of the budget is timed on a slice of the rows and says so."""
    import numpy as np
    from oracle import oracle as O
    L = O.lib()
    n = k = 4096
    rng = np.random.default_rng(0)
    w = rng.integers(-127, 128, size=(n, k)).astype(np.float32)  # int8 values as the oracle's dequantizer takes them
    sc = (rng.random(n, dtype=np.float32) * 0.01 + 1e-4).astype(np.float32)
    bias = rng.standard_normal(n, dtype=np.float32)
    all_cores = L.orc_num_threads()
    share = budget_s / 10.0
    out = {"layer": "4096x4096 int8 row-wise, use_quantized_matmul=False, fp32 (dequantize + fp32 matmul per call)",
           "cpu_model": _cpu_model(), "host_cores": os.cpu_count(), "torch_version": torch.__version__, "port": {}, "torch_eager": {}}

    def timed(fn, rows_done, rows_total):
        t0 = time.perf_counter()
        fn()
        dt = time.perf_counter() - t0
        reps = 1
        while dt < 0.05 and reps < 64:  # short cases: repeat for a stable figure
            t0 = time.perf_counter()
            for _ in range(reps * 2):
                fn()
            dt = (time.perf_counter() - t0) / (reps * 2)
            reps *= 2
            if dt * reps > share:
                break
        return dt * rows_total / rows_done

    W = np.empty((n, k), dtype=np.float32)
    for threads, tname in ((all_cores, "all_cores"), (1, "one_core")):
        L.orc_set_num_threads(threads)
        t_deq = timed(lambda: L.orc_dequant_f32(O._p(w), O._p(sc), None, n, k, k, O._p(W)), 1, 1)
        for m in (1, 64, 4096):
            rows = m if (threads > 1 or m <= 64) else 64  # one core, M = 4096: a 64-row slice of the matmul, scaled
            x = rng.standard_normal((rows, k), dtype=np.float32)
            y = np.empty((rows, n), dtype=np.float32)
            t_mm = timed(lambda: L.orc_linear_float(O._p(x), O._p(W), O._p(bias), rows, n, k, 0, O._p(y)), rows, m)
            out["port"][f"M{m}_{tname}"] = {"ms": round((t_deq + t_mm) * 1e3, 3), "dequant_ms": round(t_deq * 1e3, 3), "matmul_ms": round(t_mm * 1e3, 3),
                                           "threads": threads, "rows_timed": rows}
    L.orc_set_num_threads(all_cores)
    wt = torch.from_numpy(w).to(torch.int8)
    st = torch.from_numpy(sc).reshape(n, 1)
    bt = torch.from_numpy(bias)
    old_threads = torch.get_num_threads()
    try:
        for threads, tname in ((old_threads, "all_cores"), (1, "one_core")):
            torch.set_num_threads(threads)
            for m in (1, 64, 4096):
                rows = m if (threads > 1 or m <= 64) else 256
                x = torch.randn(rows, k)
                t_deq = timed(lambda: wt.to(torch.float32).mul_(st), 1, 1)
                Wt = wt.to(torch.float32).mul_(st)
                t_mm = timed(lambda: torch.nn.functional.linear(x, Wt, bt), rows, m)
                out["torch_eager"][f"M{m}_{tname}"] = {"ms": round((t_deq + t_mm) * 1e3, 3), "dequant_ms": round(t_deq * 1e3, 3),
                                                      "matmul_ms": round(t_mm * 1e3, 3), "threads": threads, "rows_timed": rows}
    finally:
        torch.set_num_threads(old_threads)
    ops = 2 * 4096 * n * k
    out["value_gops_M4096_all_cores_port"] = round(ops / (out["port"]["M4096_all_cores"]["ms"] / 1e3) / 1e9, 2)
    out["value_gops_M4096_all_cores_torch_eager"] = round(ops / (out["torch_eager"]["M4096_all_cores"]["ms"] / 1e3) / 1e9, 2)
    return out


def cfg1_gpu(device):
    """The GPU twin of cpu_baseline.cfg1 (BASELINE configs[0]: one 4096 x 4096 int8 row-wise Linear, use_quantized_matmul=False): the
    layer's forward on this GPU at M in {1, 64, 4096}, fp32 and bf16, graph-replayed launches timed with events -> ms per call."""
    import sdnq_amd
    out = {}
    for dt, name in ((torch.float32, "f32"), (torch.bfloat16, "bf16")):
        torch.manual_seed(0)
        lin = torch.nn.Linear(4096, 4096, bias=True).to(dt).to(device)
        mod, _ = sdnq_amd.sdnq_quantize_layer(lin, sdnq_amd.SDNQConfig(weights_dtype="int8", group_size=-1, use_quantized_matmul=False))
        for m in (1, 64, 4096):
            x = torch.randn(m, 4096, device=device, dtype=dt)
            st = torch.cuda.Stream(device=device)
            with torch.cuda.stream(st), torch.no_grad():
                mod(x)
                st.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=st):
                    for _ in range(10):
                        mod(x)
                g.replay()
                st.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(st)
                for _ in range(3):
                    g.replay()
                e1.record(st)
                st.synchronize()
            ms = e0.elapsed_time(e1) / 30
            out[f"M{m}_{name}"] = {"ms": round(ms, 4), "gflops": round(2 * m * 4096 * 4096 / ms / 1e6, 1)}
    return out


# Q.K^T runs on the int8 matrix pipe and P.V on the bf16 one, half of the operations each: ops / (ops/2/int8 + ops/2/bf16)
ATTN_MFMA_PEAK_TOPS = round(2.0 / (1.0 / INT8_MFMA_PEAK_TOPS + 1.0 / (INT8_MFMA_PEAK_TOPS / 2.0)), 1)


def attention_bench(args, device, distributed, world, rank):
    """the quantized attention calls of one denoising step (int8 Q.K^T, bf16 P.V), each = prepare
    (smooth-K, per-token int8 of Q and K, V layout) + forward kernel.  Same timing contract as the Linear workloads."""
    from sdnq_amd import attention as A
    from sdnq_amd import shapes
    calls = shapes.sdxl_unet_attentions() if args.workload == "sdxl_attn_int8" else shapes.flux_dev_attentions()
    g = torch.Generator(device=device).manual_seed(rank)
    tensors = {}
    for (name, h, qn, kn, d, rep) in calls:
        tensors[name] = tuple(torch.randn(1, h, n, d, device=device, dtype=torch.bfloat16, generator=g) for n in (qn, kn, kn))
    ops_per_step = shapes.ops_of_attentions(calls)
    n_calls = sum(c[5] for c in calls)

    def run_step():
        for (name, h, qn, kn, d, rep) in calls:
            q, k, v = tensors[name]
            for _ in range(rep):
                A.sdnq_hip_atten(q, k, v)

    def capture(fn):
        side = torch.cuda.Stream(device=device)
        with torch.cuda.stream(side):
            fn()
            side.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=side):
                fn()
        torch.cuda.synchronize()
        return gr, side

    run_step()
    torch.cuda.synchronize()
    graph = None if args.no_graph else capture(run_step)[0]
    step = graph.replay if graph is not None else run_step
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if distributed:
        import torch.distributed as dist
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if distributed:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = elapsed / args.steps * 1e3
    replicas = world if distributed else 1
    result = {
        "metric": f"quantized-attention GOP/s ({args.workload})", "value": round(ops_per_step * args.steps * replicas / elapsed / 1e9, 1),
        "unit": "GOP/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "i8 (Q.K^T) + bf16 (P.V)", "data": "synthetic",
        "config": {"workload": f"{args.workload}: {n_calls} attention calls of one denoising step, bs=1 ("
                               + ", ".join(f"{rep} x {h} heads {qn}x{kn}x{d}" for (_, h, qn, kn, d, rep) in calls) + ")",
                   "parallelism": f"{world} independent replicas" if distributed else "single GPU",
                   "launch": "eager" if graph is None else "hipGraph replay", "activations": "bf16", "smooth_k": True,
                   "matmul_dtype": "int8", "pv_matmul_dtype": None, "ops_per_step": ops_per_step},
        "step_latency_ms": round(ms_per_step, 4),
    }
    if rank == 0:
        # the forward kernel alone (operands prepared once), graph-replayed, HIP events on the launch stream
        prepared = {name: A.quantize_attn(*tensors[name]) for name in tensors}

        def fwd_only():
            for (name, h, qn, kn, d, rep) in calls:
                parts = prepared[name]
                for _ in range(rep):
                    A.atten_fwd(*parts, kn, d ** -0.5, False, torch.bfloat16)

        gr, side = capture(fwd_only)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(side):
            gr.replay()
            e0.record(side)
            for _ in range(5):
                gr.replay()
            e1.record(side)
            side.synchronize()
        sec = e0.elapsed_time(e1) / 5 / 1e3
        ach = ops_per_step / sec / 1e12
        result["roofline"] = {"bound": "mfma", "achieved": round(ach, 1), "peak": ATTN_MFMA_PEAK_TOPS, "unit": "TOP/s",
                              "frac": round(ach / ATTN_MFMA_PEAK_TOPS, 4), "traffic": None,
                              "peak_note": "half of the operations on the int8 matrix pipe, half on the bf16 one",
                              "kernel": "attn_fwd_kernel", "launches_per_step": n_calls, "avg_launch_us": round(sec / n_calls * 1e6, 3)}
        if world == 1 and not args.no_cpu_baseline:
            try:
                result["cpu_baseline"] = cpu_baseline_attention(calls, args.cpu_seconds)
            except Exception as e:  # noqa: BLE001
                result["cpu_baseline_error"] = repr(e)
        print(json.dumps(result))


def unet_all_bench(args, device, distributed, world, rank):
    """Every quantizable operation of one SDXL-UNet denoising step at bs=1 in ONE hipGraph: the 741 Linear layers (int8 w8a8), the
    49 Conv2d layers as int8 GEMMs and the 140 attention calls (int8 Q.K^T) -- the three workloads sdxl_int8, sdxl_conv_int8 and
    sdxl_attn_int8 back to back.  Reported beside the headline, not instead of it."""
    from sdnq_amd import attention as A
    from sdnq_amd import shapes
    lin_cfg = dict(weights_dtype="int8", group_size=-1, use_quantized_matmul=True)
    linears = build_layers(shapes.sdxl_unet_layer_sequence(), lin_cfg, device, seed=rank)
    linked = 0 if args.no_link_projections else link_shared_input_layers(linears)  # as sdnq_amd.accelerate(model) does
    convs = build_conv_layers(shapes.sdxl_unet_convs(), dict(weights_dtype="int8", group_size=-1, quant_conv=True, use_quantized_matmul_conv=True),
                              device, seed=rank)
    calls = shapes.sdxl_unet_attentions()
    g = torch.Generator(device=device).manual_seed(100 + rank)
    qkv = {name: tuple(torch.randn(1, h, n, d, device=device, dtype=torch.bfloat16, generator=g) for n in (qn, kn, kn))
           for (name, h, qn, kn, d, rep) in calls}
    ops = {"linear": sum(2 * m * k * n + (m * n if b else 0) for (_, _, _, m, k, n, b) in linears),
           "conv": sum(2 * m * k * n + m * n for (_, _, _, m, k, n, _) in convs), "attention": shapes.ops_of_attentions(calls)}

    def attn_step():
        for (name, h, qn, kn, d, rep) in calls:
            for _ in range(rep):
                A.sdnq_hip_atten(*qkv[name])

    parts = {"linear": lambda: run_step(linears), "conv": lambda: run_step(convs), "attention": attn_step}

    def full_step():
        for fn in parts.values():
            fn()

    def capture(fn):
        side = torch.cuda.Stream(device=device)
        with torch.cuda.stream(side):
            fn()
            side.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=side):
                fn()
        torch.cuda.synchronize()
        return gr

    for _ in range(2):
        full_step()
    torch.cuda.synchronize()
    graph = capture(full_step)
    for _ in range(args.warmup):
        graph.replay()
    torch.cuda.synchronize()
    if distributed:
        import torch.distributed as dist
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        graph.replay()
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if distributed:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms = elapsed / args.steps * 1e3
    total_ops = sum(ops.values())
    replicas = world if distributed else 1
    result = {"metric": "quantized-op GOP/s (SDXL UNet step: Linear + Conv2d + attention, bs=1)",
              "value": round(total_ops * args.steps * replicas / elapsed / 1e9, 1), "unit": "GOP/s", "n_gpus": world, "steps": args.steps,
              "warmup": args.warmup, "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
              "dtype": "i8 (+ bf16 P.V)", "data": "synthetic",
              "config": {"workload": f"sdxl_unet_all: {len(linears)} Linear + {len(convs)} Conv2d + {sum(c[5] for c in calls)} attention calls of one "
                                     "denoising step, bs=1, one hipGraph", "parallelism": f"{world} independent replicas" if distributed else "single GPU",
                         "launch": "hipGraph replay", "activations": "bf16", "linked_projection_groups": linked, "ops_per_step": total_ops,
                         "ops_by_part": ops},
              "step_latency_ms": round(ms, 4)}
    if rank == 0:
        part_ms = {}
        for name, fn in parts.items():  # each part alone, graph-replayed, HIP events
            gr = capture(fn)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            gr.replay()
            e0.record()
            for _ in range(5):
                gr.replay()
            e1.record()
            torch.cuda.synchronize()
            part_ms[name] = round(e0.elapsed_time(e1) / 5, 4)
        result["parts_ms"] = part_ms
        print(json.dumps(result))


def cpu_baseline_attention(calls, budget_s):
    """The oracle's restatement of the reference attention (numpy) on a bounded sample: 2 heads of 1024 x 1024 tokens."""
    import numpy as np
    from oracle import oracle as O
    d = calls[0][4]
    rng = np.random.default_rng(0)
    q, k, v = (O.round_dtype(rng.standard_normal((1, 2, 1024, d)).astype(np.float32), "bf16") for _ in range(3))
    ops = 4 * 2 * 1024 * 1024 * d
    O.attention(q, k, v, "bf16")
    t0, n = time.perf_counter(), 0
    while time.perf_counter() - t0 < budget_s or n == 0:
        O.attention(q, k, v, "bf16")
        n += 1
    sec = time.perf_counter() - t0
    return {"value": round(ops * n / sec / 1e9, 2), "unit": "GOP/s", "cores": os.cpu_count(), "kind": "port",
            "sample": f"oracle (numpy) quantized attention, 2 heads x 1024 x 1024 x {d}, {n} passes, {sec:.1f}s (numpy / BLAS default threading)"}


def _free_port() -> int:
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def spawn_ranks(args) -> int:
    """`python bench.py --gpus N` started WITHOUT a launcher: re-execute this script under torch.distributed.run with N ranks on
    this node (one per GPU, rendezvous on 127.0.0.1) and pass its exit code on.  Fewer than N visible GPUs is an error, not a
    1-GPU run."""
    import subprocess
    if not args.dry_run:
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < args.gpus:
            print(f"bench.py: --gpus {args.gpus} asked for, {have} GPU(s) visible: refusing to run fewer ranks than asked", file=sys.stderr)
            return 2
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL across processes needs it on this driver
    return subprocess.call(cmd, env=env)


def dry_run(args, world, rank):
    """The multi-rank skeleton of the benchmark with NO device work (CPU, gloo): rendezvous, warm-up, barrier-bracketed timed region,
    MAX over ranks, one JSON line from rank 0.  For tests of the launch contract only -- the line is marked as not a measurement."""
    import torch.distributed as dist
    distributed = world > 1 or "RANK" in os.environ
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
    for _ in range(args.warmup):
        time.sleep(0.001)
    if distributed:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.001 * (1 + rank))  # ranks differ: the MAX over ranks must come out
    if distributed:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank == 0:
        tp = args.tp and distributed
        print(json.dumps({"metric": "DRY RUN (no device work): launch-contract plumbing only", "value": 0.0, "unit": "GOP/s", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
                          "higher_is_better": True, "scaling": "strong" if tp else "weak", "vs_baseline": None, "dtype": "none",
                          "data": "dry-run (no device work, gloo on CPU) -- NOT a measurement",
                          "config": {"workload": args.workload, "parallelism": (f"tp{world}" if tp else f"{world} independent replicas")},
                          "dry_run": True}))
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


def main():
    args = parse()
    launched = "WORLD_SIZE" in os.environ or "RANK" in os.environ
    if args.gpus > 1 and not launched:
        raise SystemExit(spawn_ranks(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1 or ("RANK" in os.environ and "MASTER_ADDR" in os.environ)  # under torch.distributed.run, also with 1 rank
    if args.gpus != world and (distributed or args.gpus > 1):
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.dry_run:
        return dry_run(args, world, rank)
    if not torch.cuda.is_available() or torch.cuda.device_count() <= local_rank:
        raise SystemExit(f"bench.py: rank {rank} needs GPU {local_rank}, {torch.cuda.device_count() if torch.cuda.is_available() else 0} visible "
                         "(the HIP path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=device)

    from sdnq_amd import _lib, shapes
    from sdnq_amd import linear as L
    if not _lib.load().sdnq_hip_device_supported(local_rank):
        raise SystemExit("device is not gfx950: the HIP kernels of this repo target MI355X only")

    if args.workload == "sdxl_unet_all":
        unet_all_bench(args, device, distributed, world, rank)
        if distributed:
            dist.barrier()
            dist.destroy_process_group()
        return
    if args.workload.endswith("_attn_int8"):
        attention_bench(args, device, distributed, world, rank)
        if distributed:
            dist.barrier()
            dist.destroy_process_group()
        return

    shape_list, cfg_kwargs, mm_name, tokens = workload_config(args.workload)
    tp = args.tp and distributed
    is_conv = args.workload.startswith("sdxl_conv")
    tp_peer = None
    if is_conv:
        layers = build_conv_layers(shape_list, cfg_kwargs, device, seed=rank)
    else:
        if tp and args.tp_gather == "peer":
            from sdnq_amd.parallel import PeerArena
            tp_peer = PeerArena(rank, world, device=device)
        layers = build_layers(shape_list, cfg_kwargs, device, scale=args.layers_scale, tp_rank=rank if tp else 0,
                              tp_world=world if tp else 1, seed=0 if tp else rank, tp_chunks=args.tp_chunks, activation_pool=args.activation_pool,
                              tp_peer=tp_peer)
    if args.fuse_projections and not is_conv and not tp:
        layers = fuse_shared_input_layers(layers)
    linked = 0
    if not args.no_link_projections and not args.fuse_projections and not is_conv and not tp:
        linked = link_shared_input_layers(layers)
    ops_per_step = sum(2 * m * k * n + (m * n if b else 0) for (_, _, _, m, k, n, b) in layers)

    # eager warm-up (builds the per-module weight caches; in eager mode also: the layers learn which of them share their input and get
    # their fast-path plans, sdnq_amd/linear.py UNSHARED_AFTER), then capture the whole step
    for _ in range(5 if args.launch == "eager" or args.no_graph else 2):
        run_step(layers)
    torch.cuda.synchronize()
    graph = None
    one_launch = None
    if args.launch == "eager":
        args.no_graph = True
    compiled = None
    captured = None
    if args.launch == "capture" and not tp:
        # the PUBLIC form of the hand-written capture below: sdnq_amd.capture(model, example_input) -- what a pipeline user calls after
        # accelerate(model).  The step's layers as one module; the first activation is the example input (every layer keeps its own tensor)
        import sdnq_amd

        class StepModule(torch.nn.Module):
            def __init__(self, layers):
                super().__init__()
                self.mods = torch.nn.ModuleList([l[1] for l in layers])
                self.xs = [l[2] for l in layers]

            def forward(self, x0):
                out = None
                for mod, x in zip(self.mods, self.xs):
                    out = mod(x)
                return out

        captured = sdnq_amd.capture(StepModule(layers), layers[0][2], warmup=2)
        captured_x0 = layers[0][2]
        args.no_graph = True
    if args.launch == "compile" and not tp:
        # what `torch.compile(pipeline.unet, mode="reduce-overhead")` does to the Linear layers of an unmodified pipeline: Dynamo traces the
        # module calls (SDNQLayer.forward emits one sdnq_hip::layer_forward op per layer), Inductor wraps the result in CUDA-graph trees
        class Step(torch.nn.Module):
            def __init__(self, layers):
                super().__init__()
                self.mods = torch.nn.ModuleList([l[1] for l in layers])
                # the step's activations are buffers of the module (static addresses: the CUDA-graph trees do not copy them into
                # private input buffers every replay, as they would 446 positional inputs); shared tensors stay ONE buffer
                self.slot, seen = [], {}
                for l in layers:
                    if id(l[2]) not in seen:
                        seen[id(l[2])] = len(seen)
                        self.register_buffer(f"x{seen[id(l[2])]}", l[2], persistent=False)
                    self.slot.append(seen[id(l[2])])

            def forward(self):
                xs = [getattr(self, f"x{i}") for i in self.slot]
                return [mod(x) for mod, x in zip(self.mods, xs)]  # every output is a graph output: nothing is dead code

        from sdnq_amd import torch_ops
        for l in layers:
            if hasattr(l[1], "sdnq_dequantizer"):
                torch_ops.layer_handle(l[1])
        if not args.no_link_projections:
            # layers whose layer_matmul nodes share one quantized activation become one grouped launch (the compiled-graph form of
            # the eager path's linked projections; sdnq_amd.torch_ops.MergeLayerMatmuls)
            torch_ops.enable_compile_grouping()
        step_mod = Step(layers)
        t_c = time.perf_counter()
        compiled = torch.compile(step_mod, mode="reduce-overhead", fullgraph=True)
        with torch.no_grad():
            for _ in range(3):
                compiled()
        torch.cuda.synchronize()
        compile_s = time.perf_counter() - t_c
        args.no_graph = True
    if not args.no_graph and not tp:
        side = torch.cuda.Stream(device=device)
        with torch.cuda.stream(side):
            run_step(layers)
            side.synchronize()
            graph = torch.cuda.CUDAGraph()
            from sdnq_amd import ops as _ops
            _ops.reset_fused_calls()
            with torch.cuda.graph(graph, stream=side):
                run_step(layers)
            one_launch = _ops.fused_call_count()
        torch.cuda.synchronize()

    def step():
        if captured is not None:
            captured(captured_x0)
        elif compiled is not None:
            with torch.no_grad():
                compiled()
        elif graph is not None:
            graph.replay()
        else:
            run_step(layers)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if args.profile_markers:
        torch.cuda._sleep(200000)  # marker kernel in front of the timed window (outside the timed region)
        torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if args.profile_markers:
        torch.cuda._sleep(200000)  # ... and behind it
        torch.cuda.synchronize()
    if distributed:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if tp_peer is not None:
        tp_peer.check()  # a rendezvous that timed out (a rank out of step) is an error, not a number
    ms_per_step = elapsed / args.steps * 1e3
    replicas = 1 if tp or not distributed else world
    total_ops = ops_per_step * args.steps * replicas
    value = total_ops / elapsed / 1e9  # GOP/s, whole job

    result = {
        "metric": "quantized-matmul GOP/s (SDXL UNet int8 Linear step, bs=1)" if args.workload == "sdxl_int8"
        else f"quantized-matmul GOP/s ({args.workload})",
        "value": round(value, 1), "unit": "GOP/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "strong" if tp else "weak",
        "vs_baseline": None, "dtype": "i8" if mm_name == "int8" else "fp8", "data": "synthetic",
        "config": {"workload": f"{args.workload}: {len(layers)} quantized {'Conv2d' if is_conv else 'Linear'} layers of one denoising step, bs=1 "
                               f"({sum(1 for l in layers if l[3] >= 32)} w8a8 GEMMs + {sum(1 for l in layers if l[3] < 32)} M=1 layers)",
                   "parallelism": (f"tp{world} column-shard + RCCL all-gather" if tp else (f"{world} independent replicas" if distributed else "single GPU")),
                   "launch": ("torch.compile(mode='reduce-overhead'): every plain layer rowquant + layer_matmul operators, layers on one quantized activation merged into grouped launches by the post-grad pass"
                              if compiled is not None else ("sdnq_amd.capture(model, x): hipGraph replay through the public API" if captured is not None
                                                         else ("eager" if graph is None else "hipGraph replay"))), "activations": "bf16",
                   **({"compile_seconds": round(compile_s, 1), "compile_grouping": dict(__import__("sdnq_amd").torch_ops.merge_stats)} if compiled is not None else {}),
                   "distinct_activation_tensors": len({id(l[2]) for l in layers}), "activation_quant_cache": L.CACHE_ACTIVATIONS > 0,
                   **({"activation_pool": args.activation_pool, "activation_buffers": len({l[2].data_ptr() for l in layers})} if args.activation_pool else {}),
                   "requantized_weight_cache": L.CACHE_WEIGHTS, "fused_projections": bool(args.fuse_projections),
                   "resident_weight_bytes": resident_weight_bytes(layers),
                   **({"per_call_weight_pipeline": {"enabled": L.PIPELINE_WEIGHTS, **L._weight_pipeline.stats}} if not L.CACHE_WEIGHTS else {}),
                   "linked_projection_groups": linked,
                   # layers whose row quantization runs INSIDE their GEMM launch (sdnq_hip_linear_w8a8_fused, csrc/gemm_aq.hip); None: not a graph run
                   "one_launch_linears": one_launch,
                   "ops_per_step": ops_per_step, **{k: v for k, v in cfg_kwargs.items()}},
        **({"tp": {"ranks": world, "rank_devices": [f"cuda:{r}" for r in range(world)], "rccl_version": list(torch.cuda.nccl.version()),
                   "sharded_layers": sum(1 for l in layers if type(l[1]).__name__ == "ColumnShardedLinear"),
                   "collective": ("copy-free: sdnq_hip_push_columns -- P2P stores of [M, N/W] bf16 into every rank's [M, N] output over IPC-mapped "
                                  "arenas, mailbox rendezvous, no RCCL call per layer" if args.tp_gather == "peer"
                                  else "all_gather_into_tensor of [M, N/W] bf16 per layer + sdnq_hip_unshard_columns"),
                   "gather": args.tp_gather, "m_chunks": args.tp_chunks if args.tp_gather == "rccl" else 1}} if tp else {}),
        "tokens_per_s": round(tokens * replicas / (ms_per_step / 1e3), 1),
        "step_latency_ms": round(ms_per_step, 4),
    }

    gathers = None
    if tp:
        try:
            gathers = time_all_gathers(layers, device)  # a collective: every rank takes part
        except Exception as e:  # noqa: BLE001
            result["tp"]["gather_timing_error"] = repr(e)
        if gathers:
            link_gbps = gathers["bytes_received_per_rank_per_step"] / gathers["seconds_per_step"] / 1e9
            result["tp"].update({
                "gathers_per_step": gathers["gathers_per_step"], "gather_bytes_received_per_rank_per_step": gathers["bytes_received_per_rank_per_step"],
                "largest_gather_message_bytes": gathers["largest_message_bytes"],
                "gathers_alone_ms_per_step": round(gathers["seconds_per_step"] * 1e3, 4), "gather_achieved_gbps_per_rank": round(link_gbps, 1),
                "xgmi_link_gbps": XGMI_LINK_GBPS, "xgmi_links_used": world - 1,
                "gather_frac_of_links": round(link_gbps / (XGMI_LINK_GBPS * max(1, world - 1)), 4),
                "note": "bytes received per rank / time of the step's all-gathers run alone; a fully connected all-gather receives from "
                        "W - 1 peers over W - 1 links at once, each bound by 153 GB/s"})

    want_tp_report = distributed and not tp and not is_conv and (args.tp_report or (world > 1 and os.environ.get("SDNQ_BENCH_TP_REPORT", "1") != "0"))

    float_mode = not cfg_kwargs.get("use_quantized_matmul", cfg_kwargs.get("use_quantized_matmul_conv", False))
    if float_mode:
        result["dtype"] = "bf16"
        result["config"]["workload"] = (f"{args.workload}: {len(layers)} quantized Linear layers of one denoising step, bs=1, the reference's DEFAULT mode "
                                        f"({sum(1 for l in layers if l[3] >= 32)} int8-weight x bf16-activation GEMMs, weights dequantized inside the kernel, "
                                        f"+ {sum(1 for l in layers if l[3] < 32)} M=1 layers)")
    if rank == 0:
        try:
            gk = time_float_kernel(layers, device) if float_mode else time_gemm_kernel(layers, mm_name, device)
        except Exception as e:  # noqa: BLE001
            gk, result["roofline_error"] = None, repr(e)
        traffic, traffic_src, traffic_meta, traffic_stale = None, None, None, None
        # the PMC profile of the SAME launch set (tools/pmc_step.sh <tag> <set>): one tracked file per workload, newest round first
        pmc_set = {"sdxl_int8": "sdxl", "sdxl_fp8": "sdxl_fp8", "flux_int4_had": "flux", "flux_int8_svd": "flux_svd", "sdxl_int8_dequant": "sdxl_dequant",
                   "linear_int8": "linear"}.get(args.workload)
        prof_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
        cands = ([f"r06_pmc_gemm_traffic_{pmc_set}.json"] if pmc_set else []) + (["r05_pmc_gemm_traffic_linked.json"] if args.workload == "sdxl_int8" and linked else []) \
            + (["r01_pmc_gemm_traffic.json"] if args.workload == "sdxl_int8" and not linked else [])
        pmc_name = next((c for c in cands if os.path.exists(os.path.join(prof_dir, c))), cands[0] if cands else "none")
        pmc = os.path.join(prof_dir, pmc_name)
        if pmc_set and os.path.exists(pmc) and not args.fuse_projections:
            # HBM bytes per launch of the same 722 launches, from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes
            # (tools/pmc_shapes.py + tools/pmc_traffic.py; counters cannot be read inside this process)
            with open(pmc) as f:
                pj = json.load(f)
            traffic = round(pj["_all_gemm_kernel"]["hbm_bytes_per_launch"])
            traffic_src = f"profiles/{pmc_name} (rocprofv3 --pmc FETCH_SIZE, WRITE_SIZE; bytes per launch)"
            traffic_meta = pj.get("_provenance", "collected in an earlier profiling session (see `git log -- profiles/" + pmc_name + "`), not by this run")
            # the constant describes the library it was collected on: say so when that is not the library this run loaded
            traffic_stale = not (isinstance(traffic_meta, dict) and traffic_meta.get("library_srchash") == (_lib.source_hash() or "")[:16])
        if gk:
            ach = gk["ops"] / gk["seconds"] / 1e12
            if float_mode:
                peak, unit, kernel = BF16_MFMA_PEAK_TFLOPS, "TFLOP/s", "gemm_kernel<MM_W8BF16> (bf16 MFMA; int8 weight bytes dequantized between LDS and the MFMA)"
                peak_meas, peak_meas_src = BF16_MFMA_MEASURED_TFLOPS, "profiles/r02_mfma_peak.txt (bf16 register-resident MFMA loop, random operands)"
            elif mm_name == "int8":
                peak, unit, kernel = INT8_MFMA_PEAK_TOPS, "TOP/s", "gemm_kernel<MM_I8> (int8 MFMA scaled-mm)"
                peak_meas, peak_meas_src = INT8_MFMA_MEASURED_TOPS, "profiles/r02_mfma_peak.txt (register-resident MFMA loop on random operand bytes, tools/mfma_peak.hip)"
            else:
                peak, unit, kernel = FP8_MFMA_PEAK_TFLOPS, "TFLOP/s", "gemm_kernel<MM_FP8> (fp8 e4m3 MFMA scaled-mm)"
                peak_meas, peak_meas_src = None, None
            result["roofline"] = {"bound": "mfma", "achieved": round(ach, 1), "peak": peak, "unit": unit,
                                  "frac": round(ach / peak, 4), "traffic": traffic, "traffic_source": traffic_src,
                                  # the PMC figure is a constant read from a tracked profile (counters cannot be collected inside a timed run)
                                  "traffic_measured_in_run": False if traffic is not None else None, "traffic_provenance": traffic_meta,
                                  "traffic_stale": traffic_stale if traffic is not None else None,
                                  "algorithmic_bytes_per_launch": round(gk["bytes"] / gk["launches"]),
                                  "hbm_achieved_gbps": round(gk["bytes"] / gk["seconds"] / 1e9, 1), "hbm_peak_gbps": HBM_PEAK_GBPS,
                                  # every M x N x K class of the step replayed alone (launches, avg us, fraction of the peak, tile, CUs used)
                                  **({"per_shape": gk["per_shape"]} if gk.get("per_shape") else {}),
                                  "hbm_frac": round(gk["bytes"] / gk["seconds"] / 1e9 / HBM_PEAK_GBPS, 4), "kernel": kernel,
                                  "launches_per_step": gk["launches"], "avg_launch_us": round(gk["seconds"] / gk["launches"] * 1e6, 3),
                                  "peak_measured": peak_meas, "frac_of_measured_peak": round(ach / peak_meas, 4) if peak_meas else None,
                                  "peak_measured_source": peak_meas_src,
                                  "traffic_over_algorithmic": round(traffic / (gk["bytes"] / gk["launches"]), 3) if traffic else None}
        if world == 1 and graph is not None and not is_conv:
            # beside the headline (outside its timed region): the same step launched EAGERLY from Python, layer call by layer call -- what
            # `accelerate(model)` gives a pipeline that does not capture anything (round 6: the per-layer plans of csrc/fastpath.cpp)
            try:
                for _ in range(8):
                    run_step(layers)
                torch.cuda.synchronize()
                t_e = time.perf_counter()
                for _ in range(20):
                    run_step(layers)
                torch.cuda.synchronize()
                result["config"]["eager_ms_per_step"] = round((time.perf_counter() - t_e) / 20 * 1e3, 4)
                result["config"]["eager_fast_path"] = L._FP is not None
            except Exception as e:  # noqa: BLE001
                result["config"]["eager_error"] = repr(e)
        if world == 1 and not args.no_cpu_baseline and not is_conv:
            try:
                port = cpu_baseline(shape_list, mm_name, args.cpu_seconds / 2)
                result["cpu_baseline"] = port
                if mm_name == "int8":
                    # headline CPU figure = the torch-eager restatement (the reference's CPU path IS torch); the C oracle stays beside it
                    try:
                        te = cpu_baseline_torch_eager(shape_list, mm_name, args.cpu_seconds / 2)
                        te["port_c_oracle"] = {"value": port["value"], "unit": port["unit"], "cores": port["cores"], "sample": port["sample"]}
                        result["cpu_baseline"] = te
                    except Exception as e:  # noqa: BLE001  (torch._int_mm missing on this host's build: keep the C port)
                        result["cpu_baseline"]["torch_eager_error"] = repr(e)
                if args.workload == "sdxl_int8":  # the reference's own CPU-runnable case (BASELINE configs[0])
                    result["cpu_baseline"]["cfg1"] = cpu_baseline_cfg1(args.cpu_seconds)
                    result["cpu_baseline"]["cfg1"]["gpu"] = cfg1_gpu(device)  # the same layer on this GPU, side by side
                    result["cpu_baseline"]["cpu_model"] = result["cpu_baseline"]["cfg1"]["cpu_model"]
                    result["cpu_baseline"]["torch_version"] = torch.__version__
            except Exception as e:  # noqa: BLE001
                result["cpu_baseline_error"] = repr(e)
    import threading as _threading
    done, printed = _threading.Event(), [False]
    if want_tp_report:
        # LAST, after everything the replica line is made of (a collective: every rank runs it; the other ranks wait for rank 0 at its first
        # barrier), and under a watchdog: the sharded modes have never met real xGMI links, and a rank stuck in a collective cannot be
        # interrupted from Python -- after SDNQ_BENCH_TP_TIMEOUT seconds (default 240) every rank's watchdog prints (rank 0) the replica
        # line with the timeout named in `tp` and leaves the process, so the driver always gets its line and the launcher a clean exit
        import threading

        def _watchdog():
            if done.wait(float(os.environ.get("SDNQ_BENCH_TP_TIMEOUT", "240"))):
                return
            if rank == 0 and not printed[0]:
                result["tp"] = {"error": "the sharded report did not finish in time (watchdog); the replica numbers above are complete"}
                print(json.dumps(result), flush=True)
            os._exit(0)

        threading.Thread(target=_watchdog, daemon=True).start()
        try:
            result["tp"] = tp_report(args, device, world, rank, shape_list, cfg_kwargs, ops_per_step)
        except Exception as e:  # noqa: BLE001
            result["tp"] = {"error": repr(e)[:400]}
    if rank == 0:
        print(json.dumps(result), flush=True)
        printed[0] = True
    if distributed:
        dist.barrier()  # (still under the watchdog: a rank that left the sharded report early waits here for one that is stuck in it)
        dist.destroy_process_group()
    done.set()


if __name__ == "__main__":
    main()
