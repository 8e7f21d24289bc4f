#!/usr/bin/env python3
"""Random-shape sweep of the ONE-launch w8a8 Linear (sdnq_hip_linear_w8a8_fused, csrc/gemm_aq.hip) against the two-launch route
(sdnq_hip_linear_w8a8): int8 and fp8 codes, bf16 / f16 activations, ragged M and N, every K stage count 1..10, with and without bias,
row-strided inputs, degenerate rows (all zero, one huge element, denormal-scale rows), eager and inside a captured hipGraph.  The two
routes must agree bit for bit.  usage: tools/fuzz_fused.py [seed] [cases]"""
import os
import random
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdnq_amd import ops  # noqa: E402


def run(seed: int, cases: int, verbose: bool = True):
    rng = random.Random(seed)
    dev = torch.device("cuda:0")
    bad = []
    for c in range(cases):
        g = torch.Generator().manual_seed(seed * 1000 + c)
        mm = rng.choice([ops.MM_I8, ops.MM_I8, ops.MM_FP8])
        dt = rng.choice([torch.bfloat16, torch.float16])
        k = 128 * rng.randint(1, 10)
        m = rng.choice([33, 64, 65, 100, 257, 777, 1024, rng.randint(33, 1500)])
        n = 8 * rng.randint(1, 200)
        x = torch.randn(m, k, generator=g) * torch.exp(2.0 * torch.randn(m, 1, generator=g))
        r = rng.randrange(m)
        x[r] = 0                                             # scale 0: the general path of the quantizer
        x[rng.randrange(m), rng.randrange(k)] = 3.0e4        # one element owns the row's scale
        x[rng.randrange(m)] *= 1e-30                         # rows whose scale leaves the fast range (f16: underflow to zero rows)
        x = x.to(dt)
        ldx = k + rng.choice([0, 0, 8, 264])
        wide = torch.zeros(m, ldx, dtype=dt)
        wide[:, :k] = x
        xv = wide.to(dev)[:, :k]
        if mm == ops.MM_I8:
            b = torch.randint(-128, 128, (n, k), dtype=torch.int8, generator=g).to(dev)
        else:
            b = (torch.randn(n, k, generator=g) * 40).clamp(-448, 448).to(torch.float8_e4m3fn).to(dev)
        sb = (torch.rand(n, generator=g) * 0.02 + 1e-4).to(dev)
        bias = torch.randn(n, generator=g).to(dt).to(dev) if rng.random() < 0.6 else None
        want = ops.linear_w8a8(mm, xv.contiguous(), b, sb, bias, dt)[0]
        got = ops.linear_w8a8_fused(mm, xv, b, sb, bias, dt)
        if rng.random() < 0.3:  # the same launch replayed from a graph
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                ops.linear_w8a8_fused(mm, xv, b, sb, bias, dt)
                s.synchronize()
                gr = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gr, stream=s):
                    got2 = ops.linear_w8a8_fused(mm, xv, b, sb, bias, dt)
            gr.replay()
            torch.cuda.synchronize()
            if not torch.equal(got2.view(torch.int16), want.view(torch.int16)):
                bad.append(("graph", mm, dt, m, n, k, ldx, bias is not None))
        torch.cuda.synchronize()
        if not torch.equal(got.view(torch.int16), want.view(torch.int16)):
            bad.append((mm, dt, m, n, k, ldx, bias is not None, int((got.view(torch.int16) != want.view(torch.int16)).sum())))
        if verbose:
            print(c, mm, dt, m, n, k, ldx, bias is not None, "ok" if not bad else bad[-1], flush=True)
    return bad


if __name__ == "__main__":
    out = run(int(sys.argv[1]) if len(sys.argv) > 1 else 1, int(sys.argv[2]) if len(sys.argv) > 2 else 200)
    print("mismatches:", out)
    sys.exit(1 if out else 0)
