#!/usr/bin/env python3
"""The attention formats side by side (graph-replayed, whole call: prepare + forward): the default configuration on its tuned kernels against
fp8 Q.K^T and the quantized P.V formats on the plain kernel of csrc/attention.hip (round 6), and torch's SDPA.  us per call."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdnq_amd import attention as A  # noqa: E402

dev = torch.device("cuda:0")
shapes = [(1, 10, 4096, 4096, 64), (1, 20, 1024, 1024, 64), (1, 24, 4608, 4608, 128), (1, 10, 4096, 77, 64)]


def timed(fn, n=8, reps=5):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn(); s.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(n):
                fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / n)
    return best


for (z, h, qn, kn, d) in shapes:
    torch.manual_seed(0)
    q, k, v = (torch.randn(z, h, n, d, device=dev).to(torch.bfloat16) for n in (qn, kn, kn))
    row = [f"{z}x{h}x{qn}x{kn}x{d}:"]
    ref = torch.nn.functional.scaled_dot_product_attention(q.float(), k.float(), v.float())
    for mm, pv in (("int8", None), ("fp8", None), ("int8", "int8"), ("fp8", "fp8"), ("int8", "float16")):
        t = timed(lambda: A.sdnq_hip_atten(q, k, v, matmul_dtype=mm, pv_matmul_dtype=pv))
        err = float((A.sdnq_hip_atten(q, k, v, matmul_dtype=mm, pv_matmul_dtype=pv).float() - ref).norm() / ref.norm())
        row.append(f"{mm}/{pv or 'value dtype'} {t:7.1f} us (rel L2 vs fp32 {err:.1e})")
    row.append(f"torch SDPA bf16 {timed(lambda: torch.nn.functional.scaled_dot_product_attention(q, k, v)):7.1f} us")
    print(" | ".join(row), flush=True)
