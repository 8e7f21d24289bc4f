#!/usr/bin/env python3
"""Random-shape sweep of the w8a8 Linear forward against the oracle (bit-exact for int8 without Hadamard / SVD).
`run(seed, iters)` is also driven, bounded, by tests/test_fuzz_gpu.py under -m gpu."""
import os, sys, random
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(seed: int = 0, iters: int = 60, verbose: bool = True) -> list:
    import sdnq_amd
    from tests.modules_util import oracle_from_module
    from oracle import oracle as O
    dev = torch.device("cuda:0")
    rng = random.Random(seed)
    bad = []
    for it in range(iters):
        m = rng.choice([32, 33, 47, 63, 64, 65, 100, 127, 129, 255, 257, 300, 511, 1000])
        n = 16 * rng.randint(2, 40)
        k = 16 * rng.randint(2, 48)
        wd = rng.choice(["int8", "int8", "uint8", "int6", "uint4", "int4"])
        gs = -1 if wd in ("int8", "uint8") else rng.choice([-1, 16, 32]) if k % 32 == 0 else -1
        bias = rng.random() < 0.7
        dt = rng.choice([torch.bfloat16, torch.float16])
        tag = "bf16" if dt == torch.bfloat16 else "f16"
        lin = torch.nn.Linear(k, n, bias=bias)
        cfg = sdnq_amd.SDNQConfig(weights_dtype=wd, group_size=gs, use_quantized_matmul=True, quantized_matmul_dtype="int8")
        mod, _ = sdnq_amd.sdnq_quantize_layer(lin.to(dt).to(dev), cfg)
        x = torch.randn(m, k).to(dt)
        y = mod(x.to(dev)).float().cpu().numpy()
        ref = O.forward(oracle_from_module(mod), x.float().numpy(), tag)
        if not np.array_equal(y, ref):
            bad.append((m, n, k, wd, gs, bias, tag, int((y != ref).sum()), float(np.abs(y - ref).max())))
            if verbose:
                print("MISMATCH", *bad[-1])
    if verbose:
        print("fuzz done, mismatches:", len(bad), "of", iters)
    return bad


if __name__ == "__main__":
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 0, int(sys.argv[2]) if len(sys.argv) > 2 else 60) else 0)
