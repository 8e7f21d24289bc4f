#!/usr/bin/env python3
"""Lab of the float16 matmul forward's two launches (csrc/rowquant.hip rowquant_f16, csrc/gemm.hip scaled_mm_f16), graph-replayed on
distinct weights, beside torch.mm on the same float16 operands (hipBLASLt; no scales, no bias: a floor for the library GEMM).  us."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sdnq_amd  # noqa: E402
from sdnq_amd import linear as L, ops  # noqa: E402

dev = torch.device("cuda:0")
shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or [(1024, 1280, 1280), (1024, 10240, 1280), (4096, 640, 640), (4096, 5120, 640), (16384, 8192, 4096), (8192, 8192, 8192)]


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
    P = 4 if n * k > 2 ** 25 else 16
    w16, ws = [], []
    for i in range(P):
        torch.manual_seed(i)
        lin = torch.nn.Linear(k, n, bias=True).to(torch.bfloat16)
        mod, _ = sdnq_amd.sdnq_quantize_layer(lin, sdnq_amd.SDNQConfig(weights_dtype="fp8", quantized_matmul_dtype="float16", group_size=-1, use_quantized_matmul=True))
        mod = mod.to(dev)
        st = L._state(mod)
        w16.append(ops.unpack_mm_f16(st.qw)); ws.append(st.qw.keep[1]); bias = mod.bias
    x = torch.randn(m, k, device=dev).to(torch.bfloat16)
    xq, xs = ops.rowquant_f16(x)
    rq = timed(lambda i: ops.rowquant_f16(x), 16)
    mm = timed(lambda i: ops.scaled_mm_f16(xq, w16[i % P], xs, ws[i % P], bias, torch.bfloat16), 16)
    wt = [w.t() for w in w16]
    lib = timed(lambda i: torch.mm(xq, wt[i % P]), 16)
    print(f"{m}x{n}x{k}: rowquant_f16 {rq:7.2f} us | scaled_mm_f16 {mm:8.2f} us = {2 * m * n * k / mm / 1e6:7.1f} TFLOP/s | torch.mm f16 {lib:8.2f} us = {2 * m * n * k / lib / 1e6:7.1f} TFLOP/s", flush=True)
