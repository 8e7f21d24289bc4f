#!/usr/bin/env python3
"""Development aid: what alternating kernels costs over running each back to back (graph-replayed chains, HIP events).
chains: rowquant x N | gemm x N | (rowquant, gemm) x N on ONE layer's buffers (dependent) | the same on N distinct layers (cold weights)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdnq_amd import ops
dev = torch.device("cuda:0")
mm = ops.MM_I8
N_REP = 40


def timed(fn):
    s = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(s):
        fn(); s.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            fn()
        g.replay(); s.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(5):
            g.replay()
        e1.record(s); s.synchronize()
    return e0.elapsed_time(e1) / 5 * 1e3  # us per chain


for (m, n, k) in [(1024, 1280, 1280), (1024, 1280, 5120), (4096, 640, 640)]:
    xs_ = [torch.randn(m, k, device=dev, dtype=torch.bfloat16) for _ in range(N_REP)]
    ws_ = [torch.randint(-128, 128, (n, k), dtype=torch.int8, device=dev) for _ in range(N_REP)]
    sb = torch.rand(n, device=dev) * 0.01
    bias = torch.randn(n, device=dev, dtype=torch.bfloat16)
    xq0, xsc0, _, _ = ops.rowquant(xs_[0], mm)
    t_rq = timed(lambda: [ops.rowquant(xs_[0], mm) for _ in range(N_REP)]) / N_REP
    t_rq_cold = timed(lambda: [ops.rowquant(xs_[i], mm) for i in range(N_REP)]) / N_REP
    t_mm = timed(lambda: [ops.scaled_mm(mm, xq0, ws_[0], xsc0, sb, bias, torch.bfloat16) for _ in range(N_REP)]) / N_REP
    t_mm_cold = timed(lambda: [ops.scaled_mm(mm, xq0, ws_[i], xsc0, sb, bias, torch.bfloat16) for i in range(N_REP)]) / N_REP

    def pair_same():
        for _ in range(N_REP):
            xq, xsc, _, _ = ops.rowquant(xs_[0], mm)
            ops.scaled_mm(mm, xq, ws_[0], xsc, sb, bias, torch.bfloat16)

    def pair_cold():
        for i in range(N_REP):
            xq, xsc, _, _ = ops.rowquant(xs_[i], mm)
            ops.scaled_mm(mm, xq, ws_[i], xsc, sb, bias, torch.bfloat16)

    def pair_indep():  # alternating kernels WITHOUT the data dependency (the matmul reads a fixed quantized activation)
        for i in range(N_REP):
            ops.rowquant(xs_[i], mm)
            ops.scaled_mm(mm, xq0, ws_[i], xsc0, sb, bias, torch.bfloat16)

    t_ps, t_pc, t_pi = timed(pair_same) / N_REP, timed(pair_cold) / N_REP, timed(pair_indep) / N_REP
    print(f"M={m} N={n} K={k}: rowquant {t_rq:.2f} (distinct inputs {t_rq_cold:.2f})  gemm {t_mm:.2f} (distinct weights {t_mm_cold:.2f})  sum {t_rq + t_mm:.2f} / {t_rq_cold + t_mm_cold:.2f} | "
          f"pair same buffers {t_ps:.2f}  pair distinct layers {t_pc:.2f}  pair distinct layers, no data dependency {t_pi:.2f}", flush=True)
