#!/usr/bin/env python3
"""Random LARGE problems through the tile heuristics (256x256 half-tile ring, 256x128, 256x160, 128x128 ...) against the same problem on
forced 64x128 tiles -- the configuration the small-shape oracle tests pin -- bit-exact for int8 (integer accumulation is order-free),
within the float tolerance for fp8; plain, bias, low-rank (SVD) and zero-point epilogues; ragged M / N / K.
`run(seed, iters)` is also driven, bounded, by tests/test_fuzz_gpu.py under -m gpu."""
import os, sys, random
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(seed: int = 0, iters: int = 30, verbose: bool = True) -> list:
    from sdnq_amd import _lib, ops
    lib = _lib.load()
    dev = torch.device("cuda:0")
    rng = random.Random(seed)
    g = torch.Generator(device=dev).manual_seed(seed)
    bad = []
    for it in range(iters):
        m = rng.choice([1024, 2048, 4096, 4608, 2049, 3000, 5000, 777, 16384])
        n = 8 * rng.randint(40, 1600)
        k = rng.choice([16 * rng.randint(8, 400), 128 * rng.randint(2, 48)])
        if m * n * k > 3e11:
            k = max(128, int(3e11 / (m * n)) // 128 * 128)
        name = rng.choice(["int8", "int8", "int8", "fp8"])
        mm = ops.MM_I8 if name == "int8" else ops.MM_FP8
        form = rng.choice(["plain", "bias", "bias", "svd", "zp"])
        if name == "int8":
            a = torch.randint(-128, 128, (m, k), dtype=torch.int8, device=dev, generator=g)
            b = torch.randint(-128, 128, (n, k), dtype=torch.int8, device=dev, generator=g)
        else:
            a = (torch.randn(m, k, device=dev, generator=g) * 50).clamp(-448, 448).to(torch.float8_e4m3fn)
            b = (torch.randn(n, k, device=dev, generator=g) * 50).clamp(-448, 448).to(torch.float8_e4m3fn)
        sa = torch.rand(m, device=dev, generator=g) * 0.02 + 1e-4
        sb = torch.rand(n, device=dev, generator=g) * 0.02 + 1e-4
        bias = torch.randn(n, device=dev, generator=g).to(torch.bfloat16) if form != "plain" else None

        def call():
            if form == "svd":
                t = (torch.randn(m, 32, device=dev, generator=torch.Generator(device=dev).manual_seed(it)) * 0.3).to(torch.bfloat16)
                up = (torch.randn(n, 32, device=dev, generator=torch.Generator(device=dev).manual_seed(it + 1)) * 0.3).to(torch.bfloat16)
                return ops.scaled_mm_lowrank(mm, a, b, sa, sb, bias, t, up, None, None, torch.bfloat16)
            if form == "zp" and name == "int8":
                rowsum = a.to(torch.int32).sum(dim=1).to(torch.int32)
                zp = torch.randn(n, device=dev, generator=torch.Generator(device=dev).manual_seed(it + 2)) * 0.1
                return ops.scaled_mm_lowrank(mm, a, b, sa, sb, bias, None, None, rowsum, zp, torch.bfloat16)
            return ops.scaled_mm(mm, a, b, sa, sb, bias, torch.bfloat16)
        lib.sdnq_hip_set_tile_override(-1)
        got = call()
        lib.sdnq_hip_set_tile_override(1)
        try:
            want = call()
        finally:
            lib.sdnq_hip_set_tile_override(-1)
        torch.cuda.synchronize()
        if name == "int8":
            ok = torch.equal(got.view(torch.int16), want.view(torch.int16))
            err = int((got != want).sum().item())
        else:
            scale = float(want.float().abs().max()) or 1.0
            err = float((got.float() - want.float()).abs().max()) / scale
            ok = err <= 2 * 2.0 ** -8
        if not ok:
            bad.append((name, form, m, n, k, err))
            if verbose:
                print("MISMATCH", *bad[-1], flush=True)
        elif verbose and os.environ.get("FUZZ_TRACE"):
            print("ok", name, form, m, n, k, flush=True)
    if verbose:
        print(f"tile fuzz done: {len(bad)} mismatches in {iters} problems", flush=True)
    return bad


if __name__ == "__main__":
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 0, int(sys.argv[2]) if len(sys.argv) > 2 else 30) else 0)
