#!/usr/bin/env python3
"""Random sweep over the CONFIGURATION space of the Linear forwards -- storage dtype x group size x matmul dtype (int8 / fp8 / uint8 /
float mode) x Hadamard x SVD x scale dtype x activation dtype x odd shapes -- against the oracle, with the tolerance rules of
tests/test_gpu_parity.py::test_module_forward_vs_golden_and_oracle (bit-exact where the arithmetic is order-free).
`run(seed, iters)` is also driven, bounded, by tests/test_fuzz_gpu.py under -m gpu."""
import os, sys, random
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

WEIGHTS = ["int8", "uint8", "int4", "uint4", "int6", "uint7", "int5", "uint3", "int2", "float8_e4m3fn", "float4_e2m1fn", "float6_e3m2fn", "int12"]


def close(got, ref, tag, hadamard, lp_float=False, f16mm=False):
    scale = float(np.abs(ref).max()) or 1.0
    # 16-bit scales on a float matmul: the epilogue rounds the accumulator, its product with the activation scale and the result to
    # bf16 (kernel_wrappers.py:139-144 on bf16 tensors) -- fp32 accumulation-order noise can flip each of the three: 3 ulp instead of 2
    lim = {"bf16": 2 * 2.0 ** -8, "f16": 2 * 2.0 ** -11, "f32": 2e-5}[tag] * (2.0 if hadamard else 1.0) * (1.5 if lp_float else 1.0)
    lim2 = {"bf16": 2e-3, "f16": 5e-4, "f32": 2e-5}[tag]
    if f16mm:  # the float16 matmul: the oracle restates the reference's CPU route, which rounds both operands and the dot product to float16
        lim, lim2 = max(lim, 8 * 2.0 ** -11 * (2.0 if hadamard else 1.0)), max(lim2, 2e-3)  # (tests/test_gpu_parity.py: assert_close_float f16mm)
    err = float(np.abs(got - ref).max()) / scale
    l2 = float(np.linalg.norm(got - ref) / (np.linalg.norm(ref) or 1.0))
    return err <= lim and l2 <= lim2, err, l2


def diagnose(mod, x, tag):
    """Where does a mismatching layer first differ from the oracle: re-quantized weights, quantized activations, or only the output?"""
    from sdnq_amd import linear as L, ops
    from tests.modules_util import oracle_from_module
    from oracle import oracle as O
    d = mod.sdnq_dequantizer
    om = oracle_from_module(mod)
    st = L._state(mod)
    mmd = str(d.quantized_matmul_dtype)
    mm = ops.MM_I8 if mmd in ("int8", "uint8") else ops.MM_FP8
    if d.re_quantize_for_matmul and mmd != "uint8":
        wq, ws, _ = L._prepare_mm_weights(mod, st, mm)
        rq, rs = om.re_quantize_matmul()[:2]
        a = wq.view(torch.uint8).cpu().numpy().reshape(om.N, om.K)
        b = np.ascontiguousarray(rq).view(np.uint8).reshape(om.N, om.K)
        print("   requant: codes differing", int((a != b).sum()), "of", a.size, " scales differing", int((ws.cpu().numpy().reshape(-1) != np.asarray(rs, np.float32).reshape(-1)).sum()))
    _, inter = O.forward(om, x.float().numpy(), tag, want_intermediates=True)
    if "xq" in inter:
        lp = mod.scale.dtype != torch.float32
        had = d.hadamard_group_size if d.use_hadamard else 0
        res = (ops.rowquant_lp if lp else ops.rowquant)(x.to(mod.weight.device).reshape(-1, om.K), mm, had)
        a = res[0].view(torch.uint8).cpu().numpy()
        b = np.asarray(inter["xq"]).view(np.uint8).reshape(a.shape)
        print("   activations: codes differing", int((a != b).sum()), "of", a.size, " scales differing", int((res[1].cpu().numpy().reshape(-1) != np.asarray(inter["xs"], np.float32).reshape(-1)).sum()))


def run(seed: int = 0, iters: int = 60, verbose: bool = True) -> list:
    import sdnq_amd
    from sdnq_amd import support
    from tests.modules_util import oracle_from_module
    from oracle import oracle as O
    dev = torch.device("cuda:0")
    rng = random.Random(seed)
    bad, done, skipped = [], 0, 0
    for it in range(iters):
        wd = rng.choice(WEIGHTS)
        k = 16 * rng.randint(2, 48)
        n = 16 * rng.randint(2, 24)
        m = rng.choice([1, 3, 31, 32, 33, 64, 100, 257, 640])
        gs = rng.choice([-1, -1, 0, 16, 32, 64]) if wd not in ("int8", "uint8", "float8_e4m3fn") else rng.choice([-1, -1, 32])
        if gs > 0 and k % gs:
            gs = -1
        qmm = rng.random() < 0.75
        mmd = rng.choice([None, None, "int8", "float8_e4m3fn", "uint8", "float16"]) if qmm else None
        had = rng.random() < 0.25 and k % 64 == 0
        svd = rng.random() < 0.2
        lp = rng.random() < 0.2
        dt = rng.choice([torch.bfloat16, torch.bfloat16, torch.float16]) if (lp or rng.random() < 0.8) else torch.float32
        if dt == torch.float32 and (qmm or had):
            dt = torch.bfloat16
        tag = {torch.bfloat16: "bf16", torch.float16: "f16", torch.float32: "f32"}[dt]
        kw = dict(weights_dtype=wd, group_size=gs, use_quantized_matmul=qmm, use_hadamard=had, dequantize_fp32=not lp)
        if mmd:
            kw["quantized_matmul_dtype"] = mmd
        if svd:
            kw.update(use_svd=True, svd_rank=rng.choice([16, 32]))
        torch.manual_seed(seed * 1000 + it)
        lin = torch.nn.Linear(k, n, bias=rng.random() < 0.7)
        lin.weight.data[:, rng.randrange(k)] *= 7
        try:
            mod, _ = sdnq_amd.sdnq_quantize_layer(lin.to(dt).to(dev), sdnq_amd.SDNQConfig(**kw))
        except (NotImplementedError, ValueError, AssertionError):
            skipped += 1
            continue
        if not hasattr(mod, "sdnq_dequantizer") or support.unsupported_reason(mod) is not None:
            skipped += 1
            continue
        d = mod.sdnq_dequantizer
        x = torch.randn(m, k) * rng.choice([0.1, 1.0, 20.0])
        x[:, rng.randrange(k)] *= 15
        if m >= 3:
            x[2] = 0
        x = x.to(dt)
        if os.environ.get("FUZZ_TRACE"):  # the last line printed names the case that took the process down
            print("case", it, kw, "m n k", m, n, k, tag, "bias", lin.bias is not None, flush=True)
        try:
            y = mod(x.to(dev)).float().cpu().numpy()
            ref = O.forward(oracle_from_module(mod), x.float().numpy(), tag)
        except (NotImplementedError, AssertionError) as e:  # a configuration the HIP path or the oracle's restatement does not cover
            skipped += 1
            if verbose and isinstance(e, AssertionError):
                print("oracle does not cover:", kw, str(e)[:80])
            continue
        done += 1
        is_qmm = d.use_quantized_matmul and m >= 32
        exact = is_qmm and str(d.quantized_matmul_dtype) in ("int8", "uint8") and not d.use_hadamard and getattr(mod, "svd_up", None) is None
        if exact:
            ok, err, l2 = np.array_equal(y, ref), float(np.abs(y - ref).max()), 0.0
        else:
            ok, err, l2 = close(y, ref, tag, bool(d.use_hadamard), lp_float=lp and is_qmm and str(d.quantized_matmul_dtype) not in ("int8", "uint8"),
                                f16mm=is_qmm and str(d.quantized_matmul_dtype) == "float16")
        if not ok:
            bad.append((kw, m, n, k, tag, "exact" if exact else "close", err, l2))
            if verbose:
                print("MISMATCH", *bad[-1])
                try:
                    diagnose(mod, x, tag)
                except Exception as e:  # noqa: BLE001
                    print("   (diagnosis failed:", repr(e)[:120], ")")
    if verbose:
        print(f"mode fuzz done: {len(bad)} mismatches of {done} layer calls ({skipped} configurations not applicable)")
    return bad


if __name__ == "__main__":
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 0, int(sys.argv[2]) if len(sys.argv) > 2 else 60) else 0)
