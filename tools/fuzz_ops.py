#!/usr/bin/env python3
"""Random sweep at the OPERATOR level against the oracle: (1) scaled_mm int8 / fp8 over odd shapes, every bias form and output dtype,
(2) row quantization int8 / fp8 / asymmetric over odd K and row strides, (3) dequantize + re-quantize of every storage dtype x group
size x shape, (4) quantized attention over heads / lengths / head_dim / causal.  Bit-exact wherever the arithmetic is order-free.
`run(seed, iters)` is also driven, bounded, by tests/test_fuzz_gpu.py under -m gpu."""
import os, sys, random
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def bits(t):
    t = t.detach().contiguous().cpu()
    return (t.view(torch.uint8) if t.element_size() == 1 else t.view(torch.int16 if t.element_size() == 2 else torch.int32)).numpy()


def f32(t):
    return t.detach().float().cpu().numpy()


def run(seed: int = 0, iters: int = 60, verbose: bool = True, what=("mm", "rowquant", "weights", "attention")) -> list:
    import sdnq_amd
    from sdnq_amd import ops, common
    from tests.modules_util import oracle_from_module
    from oracle import oracle as O
    dev = torch.device("cuda:0")
    rng = random.Random(seed)
    g = torch.Generator().manual_seed(seed)
    bad = []

    def report(*a):
        bad.append(a)
        if verbose:
            print("MISMATCH", *a, flush=True)

    for it in range(iters):
        if os.environ.get("FUZZ_TRACE"):
            print("iteration", it, flush=True)
        # ---- (1) scaled_mm
        if "mm" in what:
            m = rng.choice([1, 7, 32, 33, 63, 64, 65, 127, 129, 200, 255, 256, 257, 300, 511, 513, 1000, 1025, 2049])
            n = 8 * rng.randint(1, 170)
            k = 16 * rng.randint(1, 90)
            name = rng.choice(["int8", "int8", "fp8"])
            mm = ops.MM_I8 if name == "int8" else ops.MM_FP8
            out_dt = rng.choice([torch.bfloat16, torch.bfloat16, torch.float16, torch.float32])
            tag = {torch.bfloat16: "bf16", torch.float16: "f16", torch.float32: "f32"}[out_dt]
            if name == "int8":
                a = torch.randint(-128, 128, (m, k), dtype=torch.int8, generator=g)
                b = torch.randint(-128, 128, (n, k), dtype=torch.int8, generator=g)
                a_np, b_np = a.numpy(), b.numpy()
            else:
                a = (torch.randn(m, k, generator=g) * 60).clamp(-448, 448).to(torch.float8_e4m3fn)
                b = (torch.randn(n, k, generator=g) * 60).clamp(-448, 448).to(torch.float8_e4m3fn)
                a_np, b_np = a.view(torch.uint8).numpy(), b.view(torch.uint8).numpy()
            sa, sb = torch.rand(m, generator=g) * 0.02 + 1e-4, torch.rand(n, generator=g) * 0.02 + 1e-4
            bm = rng.choice(["none", "1d", "1d", "2d"])
            bias = None if bm == "none" else (torch.randn(n, generator=g).to(out_dt) if bm == "1d" else torch.randn(m, n, generator=g))
            out = ops.scaled_mm(mm, a.to(dev), b.to(dev), sa.to(dev), sb.to(dev), None if bias is None else bias.to(dev), out_dt)
            ref = O.scaled_mm(name, a_np, b_np, sa.numpy(), sb.numpy(), None if bias is None else bias.float().numpy(), tag)
            got = f32(out)
            if name == "int8":
                if not np.array_equal(got, ref):
                    report("scaled_mm int8", m, n, k, tag, bm, int((got != ref).sum()))
            else:
                scale = float(np.abs(ref).max()) or 1.0
                lim = {"bf16": 2 * 2.0 ** -8, "f16": 2 * 2.0 ** -11, "f32": 1e-4}[tag]
                if float(np.abs(got - ref).max()) / scale > lim:
                    report("scaled_mm fp8", m, n, k, tag, bm, float(np.abs(got - ref).max()) / scale)
        # ---- (2) row quantization
        if "rowquant" in what:
            m = rng.choice([1, 5, 33, 100, 257, 1024])
            k = 8 * rng.randint(1, 700)
            dt = rng.choice([torch.bfloat16, torch.float16, torch.float32])
            pad = rng.choice([0, 0, 8, 64])
            xfull = (torch.randn(m, k + pad, generator=g) * rng.choice([0.01, 1.0, 300.0])).to(dt)
            if (xfull.stride(0) * xfull.element_size()) % 16:
                pad = 0
                xfull = xfull[:, :k].contiguous()
            x = xfull[:, :k]
            x[min(2, m - 1)] = 0
            xd = xfull.to(dev)[:, :k]
            mode = rng.choice(["int8", "fp8", "asym"])
            if mode == "asym":
                res = ops.rowquant(xd, ops.MM_I8, 0, want_rowsum=True, asymmetric=True)
                q, s, z = O.rowquant_asym(x.float().numpy())
                ok = np.array_equal(bits(res[0]), q.view(np.uint8)) and np.array_equal(res[1].cpu().numpy().reshape(-1), s.reshape(-1)) \
                    and np.array_equal(res[4].cpu().numpy().reshape(-1), z.reshape(-1))
            else:
                res = ops.rowquant(xd, ops.MM_I8 if mode == "int8" else ops.MM_FP8, 0, want_rowsum=mode == "int8")
                q, s, rs = O.rowquant(x.float().numpy(), mode)
                ok = np.array_equal(bits(res[0]), q.view(np.uint8)) and np.array_equal(res[1].cpu().numpy().reshape(-1), s.reshape(-1))
                if mode == "int8":
                    ok = ok and np.array_equal(res[2].cpu().numpy(), rs)
            if not ok:
                qa, qb = bits(res[0]), q.view(np.uint8)
                rows = np.nonzero((qa != qb).any(axis=1))[0]
                sdiff = int((res[1].cpu().numpy().reshape(-1) != s.reshape(-1)).sum())
                ex = []
                for r in rows[:2]:
                    c = int(np.nonzero(qa[r] != qb[r])[0][0])
                    ex.append((int(r), c, float(x[r, c]), float(s.reshape(-1)[r]), int(qa[r, c]), int(qb[r, c])))
                report("rowquant", mode, m, k, str(dt), "row stride", xfull.stride(0), "rows with wrong codes", len(rows), "wrong scales", sdiff,
                       "(row, col, x, scale, got, want):", ex)
        # ---- (3) storage dtypes: quantize (HIP) -> dequantize / re-quantize (HIP) vs the oracle's decode of the same bytes
        if "weights" in what:
            wd = rng.choice([w for w, e in common.dtype_dict.items() if isinstance(e.get("num_bits"), int) and e["num_bits"] <= 16
                             and w not in ("bool", "int1", "uint1", "float8_e8m0fnu", "float8_e4m3fnuz", "float8_e5m2fnuz") and not w.startswith("fp")
                             and w not in ("float16", "bfloat16", "int16", "uint16", "float16_e5m10fn")])
            k = 16 * rng.randint(2, 40)
            n = 16 * rng.randint(1, 12)
            gs = rng.choice([-1, 0, 16, 32, 64])
            if gs > 0 and k % gs:
                gs = -1
            dt = rng.choice([torch.bfloat16, torch.float16, torch.float32])
            tag = {torch.bfloat16: "bf16", torch.float16: "f16", torch.float32: "f32"}[dt]
            lin = torch.nn.Linear(k, n, bias=False)
            lin.weight.data[:, rng.randrange(k)] *= 9
            try:
                mod, _ = sdnq_amd.sdnq_quantize_layer(lin.to(dt).to(dev), sdnq_amd.SDNQConfig(weights_dtype=wd, group_size=gs, use_quantized_matmul=False))
                dq = mod.sdnq_dequantizer
                w = dq(mod.weight, mod.scale, zero_point=mod.zero_point)
                want = oracle_from_module(mod).dequantize(tag)
                if not np.array_equal(f32(w).reshape(n, k), want.reshape(n, k)):
                    report("dequant", wd, gs, n, k, tag, int((f32(w).reshape(n, k) != want.reshape(n, k)).sum()))
            except NotImplementedError:
                pass
            except (ValueError, AssertionError, KeyError) as e:  # the fuzzer's own plumbing (module -> oracle) for an exotic format
                if verbose:
                    print("weights: not checked:", wd, gs, n, k, tag, repr(e)[:100], flush=True)
        # ---- (4) quantized attention
        if "attention" in what and it % 4 == 0:
            from sdnq_amd import attention as A
            h = rng.choice([1, 2, 5])
            kvh = h if rng.random() < 0.7 or h == 1 else 1
            qn = rng.choice([1, 31, 64, 77, 200, 513])
            kn = rng.choice([1, 17, 77, 128, 333, 1024])
            d = rng.choice([40, 64, 80, 128])
            causal = rng.random() < 0.3 and qn == kn
            dt = rng.choice([torch.bfloat16, torch.float16])
            tag = "bf16" if dt == torch.bfloat16 else "f16"
            q_, k_, v_ = (torch.randn(1, hh, nn, d, generator=g).to(dt) for hh, nn in ((h, qn), (kvh, kn), (kvh, kn)))
            try:
                out = A.sdnq_hip_atten(q_.to(dev), k_.to(dev), v_.to(dev), is_causal=causal, enable_gqa=kvh != h)
                ref = O.attention(q_.float().numpy(), k_.float().numpy(), v_.float().numpy(), tag, is_causal=causal)
                got = f32(out)
                scale = float(np.abs(ref).max()) or 1.0
                # two implementations that each round P and the output to the value dtype: up to ~2 ulp(dtype) of max|out| apart
                if got.shape != ref.shape or float(np.abs(got - ref).max()) / scale > (1e-2 if tag == "bf16" else 2e-3):
                    report("attention", h, kvh, qn, kn, d, causal, tag, float(np.abs(got - ref).max()) / scale if got.shape == ref.shape else "shape")
            except NotImplementedError:
                pass
    if verbose:
        print(f"operator fuzz done: {len(bad)} mismatches in {iters} iterations", flush=True)
    return bad


if __name__ == "__main__":
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 0, int(sys.argv[2]) if len(sys.argv) > 2 else 60) else 0)
