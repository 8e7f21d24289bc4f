#!/usr/bin/env python3
"""BUILD CONTAINER ONLY (imports the reference, which does not travel): random sweep of the ORACLE against the reference's own CPU-eager
forwards, beyond what the committed fixtures pin -- random storage dtype x group size x matmul dtype x Hadamard x SVD x scale dtype x
shapes (larger than the fixtures), Linear and conv.  Bit-exact where the arithmetic is order-free (int8 / uint8 matmuls without
Hadamard / SVD), the parity tests' float tolerance elsewhere.  usage: tools/fuzz_oracle_vs_reference.py [seed] [iterations]"""
import os, sys, random
sys.argv_saved, sys.argv = sys.argv, ["x"]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
sys.path.insert(0, ROOT)
import make_golden as G  # noqa: E402
import numpy as np  # noqa: E402
import torch  # noqa: E402
from oracle import oracle as O  # noqa: E402
from tests.modules_util import oracle_from_module  # noqa: E402

WEIGHTS = ["int8", "uint8", "int4", "uint4", "int6", "uint7", "int5", "uint3", "float8_e4m3fn", "float4_e2m1fn", "float6_e3m2fn", "int12"]


def run(seed=0, iters=100, verbose=True):
    rng = random.Random(seed)
    bad, done, skipped = [], 0, 0
    for it in range(iters):
        wd = rng.choice(WEIGHTS)
        conv = rng.random() < 0.25
        dt = rng.choice([torch.bfloat16, torch.bfloat16, torch.float16, torch.float32])
        tag = {torch.bfloat16: "bf16", torch.float16: "f16", torch.float32: "f32"}[dt]
        gs = rng.choice([-1, -1, 0, 16, 32, 64])
        qmm = rng.random() < 0.75
        had = rng.random() < 0.2
        svd = rng.random() < 0.15 and not conv
        lp = rng.random() < 0.2 and dt != torch.float32
        kw = dict(weights_dtype=wd, group_size=gs, dequantize_fp32=not lp, use_hadamard=had)
        if rng.random() < 0.4:
            kw["quantized_matmul_dtype"] = rng.choice(["int8", "float8_e4m3fn", "uint8"])
        if svd:
            kw.update(use_svd=True, svd_rank=rng.choice([16, 32]))
        torch.manual_seed(seed * 10000 + it)
        if conv:
            groups = rng.choice([1, 1, 2])
            cin, cout = groups * 16 * rng.randint(2, 4), groups * 16 * rng.randint(2, 4)
            layer = torch.nn.Conv2d(cin, cout, rng.choice([1, 3]), padding=rng.choice([0, 1]), stride=rng.choice([1, 2]), groups=groups, bias=rng.random() < 0.7)
            kw.update(quant_conv=True, use_quantized_matmul_conv=qmm)
            x = torch.randn(rng.choice([1, 2]), cin, rng.randint(6, 20), rng.randint(6, 20))
            x[:, 1] *= 12
        else:
            k, n = 16 * rng.randint(2, 40), 16 * rng.randint(2, 24)
            layer = torch.nn.Linear(k, n, bias=rng.random() < 0.7)
            kw.update(use_quantized_matmul=qmm)
            x = torch.randn(rng.choice([1, 5, 33, 64, 257, 640]), k) * rng.choice([0.1, 1.0, 20.0])
            x[:, rng.randrange(k)] *= 15
        with torch.no_grad():
            layer.weight.view(layer.weight.shape[0], -1)[:, 3] *= 7
        layer = layer.to(dt)
        x = x.to(dt)
        try:
            ref_layer = G.sdnq_quantize_layer(layer, G.SDNQConfig(**kw))[0]
            if not hasattr(ref_layer, "sdnq_dequantizer"):
                continue
            with torch.no_grad():
                y_ref = ref_layer(x).float().numpy()
        except Exception:  # noqa: BLE001  (a configuration the reference itself cannot run on the CPU)
            skipped += 1
            continue
        d = ref_layer.sdnq_dequantizer
        try:
            from sdnq_amd.loader import adopt_dequantizer
            ref_layer.sdnq_dequantizer = adopt_dequantizer(d)  # the same 22 fields as this package's record (adds the geometry properties)
            om = oracle_from_module(ref_layer)
            if conv:
                meta = {"nd": 2, "kernel_size": list(ref_layer.kernel_size), "stride": list(ref_layer.stride), "padding": list(ref_layer.padding),
                        "dilation": list(ref_layer.dilation), "padding_mode": ref_layer.padding_mode, "groups": ref_layer.groups}
                y = O.conv_forward(om, x.float().numpy(), meta, tag)
            else:
                y = O.forward(om, x.float().numpy(), tag)
        except (AssertionError, NotImplementedError, KeyError, TypeError, ValueError) as e:
            skipped += 1
            if verbose:
                print("oracle does not cover:", kw, "conv" if conv else "linear", repr(e)[:90], flush=True)
            continue
        done += 1
        rows = x.numel() / x.shape[2] if conv else x.shape[0]
        is_qmm = d.use_quantized_matmul and rows >= 32
        exact = is_qmm and str(d.quantized_matmul_dtype) in ("int8", "uint8") and not d.use_hadamard and getattr(ref_layer, "svd_up", None) is None \
            and ref_layer.scale.dtype == torch.float32
        if y.shape != y_ref.shape:
            bad.append((kw, "shape", y.shape, y_ref.shape))
        elif exact:
            if not np.array_equal(y, y_ref):
                bad.append((kw, "conv" if conv else "linear", tuple(x.shape), tuple(layer.weight.shape), tag, "exact", int((y != y_ref).sum()), y.size))
        else:
            scale = float(np.abs(y_ref).max()) or 1.0
            lim = {"bf16": 2 * 2.0 ** -8, "f16": 2 * 2.0 ** -11, "f32": 2e-5}[tag] * (2.0 if d.use_hadamard else 1.0) * (1.5 if ref_layer.scale.dtype != torch.float32 else 1.0)
            err = float(np.abs(y - y_ref).max()) / scale
            if d.use_hadamard and is_qmm:
                # the rotation's summation order may move a quantized activation by one code step on rare elements: rel-L2 <= 2e-3
                l2 = float(np.linalg.norm(y - y_ref) / (np.linalg.norm(y_ref) or 1.0))
                lim, err = 2e-3, l2
            if err > lim:
                bad.append((kw, ("conv groups %d" % layer.groups) if conv else "linear", tuple(x.shape), tuple(layer.weight.shape), tag, "close", err, lim))
        if bad and bad[-1][0] is kw and verbose:
            print("MISMATCH", *bad[-1], flush=True)
    if verbose:
        print(f"oracle vs the reference: {len(bad)} mismatches in {done} forwards ({skipped} not applicable)", flush=True)
    return bad


if __name__ == "__main__":
    a = sys.argv_saved
    sys.exit(1 if run(int(a[1]) if len(a) > 1 else 0, int(a[2]) if len(a) > 2 else 100) else 0)
