#!/usr/bin/env python3
"""Lab of the fused 4-bit GEMM (csrc/gemm_w4.hip) against the two routes it competes with, graph-replayed on distinct weights:
cached int8 operand (scaled_mm alone), per-call re-quantization (requant + scaled_mm), fused (scaled_mm_w4).  us per layer."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sdnq_amd  # noqa: E402
from sdnq_amd import linear as L, ops  # noqa: E402

dev = torch.device("cuda:0")
shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or [(1024, 1280, 1280), (1024, 1280, 5120), (1024, 10240, 1280), (4096, 640, 640), (1024, 3840, 1280), (2048, 1280, 1280)]


def timed(fn, n_launch, reps=5):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for i in range(n_launch):
            fn(i)
        s.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for i in range(n_launch):
                fn(i)
    torch.cuda.synchronize()
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / n_launch)
    return best


for (m, n, k) in shapes:
    P = 16
    mods = []
    for i in range(P):
        torch.manual_seed(i)
        lin = torch.nn.Linear(k, n, bias=True).to(torch.bfloat16)
        mod, _ = sdnq_amd.sdnq_quantize_layer(lin, sdnq_amd.SDNQConfig(weights_dtype="uint4", group_size=64, use_quantized_matmul=True))
        mods.append(mod.to(dev))
    sts = [L._state(mo) for mo in mods]
    pre = [ops.requant(st.qw, ops.MM_I8) for st in sts]
    luts = [ops.lut4_build(st.qw, ops.MM_I8) for st in sts]
    x = (torch.randn(m, k, device=dev)).to(torch.bfloat16)
    xq, xs, _, _ = ops.rowquant(x, ops.MM_I8)
    bias = mods[0].bias
    scratch = torch.empty(n * k, dtype=torch.uint8, device=dev)
    a = timed(lambda i: ops.scaled_mm(ops.MM_I8, xq, pre[i % P][0], xs, pre[i % P][1], bias, torch.bfloat16), 32)
    b = timed(lambda i: ops.scaled_mm(ops.MM_I8, xq, ops.requant(sts[i % P].qw, ops.MM_I8, pre[i % P][1], out=scratch)[0], xs, pre[i % P][1], bias, torch.bfloat16), 32)
    c = timed(lambda i: ops.scaled_mm_w4(xq, sts[i % P].qw.keep[0], luts[i % P][0], xs, luts[i % P][1], bias, torch.bfloat16), 32)
    ok = torch.equal(ops.scaled_mm_w4(xq, sts[0].qw.keep[0], luts[0][0], xs, luts[0][1], bias, torch.bfloat16).view(torch.int16),
                     ops.scaled_mm(ops.MM_I8, xq, pre[0][0], xs, pre[0][1], bias, torch.bfloat16).view(torch.int16))
    print(f"{m}x{n}x{k}: int8 operand resident {a:7.2f} us | re-quantize + GEMM {b:7.2f} us | fused 4-bit GEMM {c:7.2f} us   {'bit-identical' if ok else 'MISMATCH'}", flush=True)
