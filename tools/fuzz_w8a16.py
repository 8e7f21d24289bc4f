#!/usr/bin/env python3
"""Random-shape sweep of the fused dequantize GEMM (signed codes, and unsigned codes with a zero point) against sdnq_hip_dequant's
values + the float GEMM (bit-exact: same weight values, same accumulation order).  `run(seed, iters)` is also driven, bounded, by
tests/test_fuzz_gpu.py under -m gpu."""
import os, sys, random
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(seed: int = 0, iters: int = 80, verbose: bool = True) -> list:
    from sdnq_amd import ops
    dev = torch.device("cuda:0")
    rng = random.Random(seed)
    bad = []
    for it in range(iters):
        m = rng.choice([33, 47, 64, 65, 100, 128, 129, 255, 300, 512, 1000, 1024, 2048])
        n = 8 * rng.randint(1, 200)
        k = 16 * rng.randint(1, 100)
        dt = rng.choice([torch.bfloat16, torch.float16])
        unsigned = rng.random() < 0.4
        x = (torch.randn(m, k) * rng.choice([0.1, 1.0, 30.0])).to(dt).to(dev)
        sc = (torch.rand(n) * 0.02 + 1e-5).to(dev)
        bias = torch.randn(n).to(dt).to(dev) if rng.random() < 0.6 else None
        if unsigned:
            w = torch.randint(0, 256, (n, k), dtype=torch.uint8, device=dev)
            zp = (-(torch.rand(n) * 255).round() * sc.cpu()).to(dev)
            # fma(u, s, zp): one rounding (the product of an 8-bit and a 24-bit significand is exact in float64, the sum nearly always is)
            wd = (w.double() * sc.double()[:, None] + zp.double()[:, None]).float().to(dt)
        else:
            w = torch.randint(-128, 128, (n, k), dtype=torch.int8, device=dev)
            zp = None
            wd = (w.float() * sc[:, None]).to(dt)
        ref = ops.linear_float(x, wd, bias)
        out = ops.linear_w8a16(x, w, sc, zp, bias)
        if not torch.equal(out, ref):
            bad.append((m, n, k, str(dt), unsigned, bias is not None, int((out != ref).sum().item())))
            if verbose:
                print("MISMATCH", *bad[-1])
    if verbose:
        print("w8a16 fuzz done, mismatches:", len(bad), "of", iters)
    return bad


if __name__ == "__main__":
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 0, int(sys.argv[2]) if len(sys.argv) > 2 else 80) else 0)
