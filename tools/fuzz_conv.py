#!/usr/bin/env python3
"""Random-geometry sweep of the int8 Conv2d / Conv3d forward (use_quantized_matmul_conv) against the oracle (bit-exact).
`run(seed, iters)` is also driven, bounded, by tests/test_fuzz_gpu.py under -m gpu."""
import os, sys, random
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(seed: int = 0, iters: int = 40, verbose: bool = True, variety: bool = False) -> list:
    import sdnq_amd
    from tests.modules_util import oracle_from_module
    from oracle import oracle as O
    dev = torch.device("cuda:0")
    rng = random.Random(seed)
    bad = []
    done = 0
    for it in range(iters):
        groups = rng.choice([1, 1, 2, 4])
        # per group: input channels * taps a multiple of 16 and >= 32 channels where the quantized matmul is to be taken
        cin, cout = groups * 16 * rng.randint(2, 4), groups * 16 * rng.randint(2, 4)
        ks = rng.choice([1, 3, 3, (3, 1), (1, 3), 5])
        stride = rng.choice([1, 1, 2, (2, 1)])
        pad = rng.choice([0, 1, 2])
        dil = rng.choice([1, 1, 2])
        h, w = rng.randint(5, 20), rng.randint(5, 20)
        b = rng.choice([1, 2])
        dt = rng.choice([torch.bfloat16, torch.float16])
        tag = "bf16" if dt == torch.bfloat16 else "f16"
        nd = rng.choice([2, 2, 3])
        if nd == 3:
            ks = rng.choice([1, 3, (3, 1, 1), (1, 3, 3), (2, 3, 3)])
            stride = rng.choice([1, 1, 2, (1, 2, 2), (2, 1, 1)])
            pad = rng.choice([0, 1, (1, 0, 0), (0, 1, 1)])
            depth, h, w = rng.randint(2, 6), rng.randint(4, 10), rng.randint(4, 10)
            conv = torch.nn.Conv3d(cin, cout, ks, stride=stride, padding=pad, dilation=dil, groups=groups, bias=rng.random() < 0.7)
            shape = (b, cin, depth, h, w)
        else:
            conv = torch.nn.Conv2d(cin, cout, ks, stride=stride, padding=pad, dilation=dil, groups=groups, bias=rng.random() < 0.7)
            shape = (b, cin, h, w)
        try:
            y_shape = conv(torch.zeros(shape)).shape
        except RuntimeError:
            continue
        if min(y_shape[2:]) < 1:
            continue
        # configuration: mostly the int8 matmul on int8 weights (bit-exact), sometimes unsigned weights (zero-point terms), the uint8
        # (asymmetric-activation) matmul, fp8, re-quantized 4-bit weights or the float mode
        # (round 5: Hadamard-rotated weights -- grouped convs included --, 16-bit scales on the int8 / uint8 matmuls)
        cfgname = rng.choice(["int8", "int8", "int8", "uint8w", "uint8mm", "fp8", "int4g16", "float", "int8had", "floathad", "int8lp", "uint8mmlp"]) if variety else "int8"
        cfg = {"int8": dict(weights_dtype="int8", use_quantized_matmul_conv=True),
               "uint8w": dict(weights_dtype="uint8", quantized_matmul_dtype="int8", use_quantized_matmul_conv=True),
               "uint8mm": dict(weights_dtype="uint8", use_quantized_matmul_conv=True),
               "fp8": dict(weights_dtype="fp8", quantized_matmul_dtype="fp8", use_quantized_matmul_conv=True),
               "int4g16": dict(weights_dtype="int4", group_size=16, use_quantized_matmul_conv=True),
               "float": dict(weights_dtype="uint4"),
               "int8had": dict(weights_dtype="int8", use_quantized_matmul_conv=True, use_hadamard=True),
               "floathad": dict(weights_dtype="uint4", group_size=16, use_hadamard=True),
               "int8lp": dict(weights_dtype="int8", use_quantized_matmul_conv=True, dequantize_fp32=False),
               "uint8mmlp": dict(weights_dtype="uint8", use_quantized_matmul_conv=True, dequantize_fp32=False)}[cfgname]
        try:
            mod, _ = sdnq_amd.sdnq_quantize_layer(conv.to(dt).to(dev), sdnq_amd.SDNQConfig(quant_conv=True, **cfg))
        except (NotImplementedError, ValueError):
            continue
        from sdnq_amd import support
        if not hasattr(mod, "sdnq_dequantizer") or support.unsupported_reason(mod) is not None:
            continue
        x = torch.randn(shape).to(dt)
        y = mod(x.to(dev)).float().cpu().numpy()
        meta = {"nd": nd, "kernel_size": list(mod.kernel_size), "stride": list(mod.stride), "padding": list(mod.padding), "dilation": list(mod.dilation),
                "padding_mode": mod.padding_mode, "groups": groups}
        try:
            ref = O.conv_forward(oracle_from_module(mod), x.float().numpy(), meta, tag)
        except (NotImplementedError, AssertionError):  # a form the oracle does not restate
            continue
        done += 1
        d = mod.sdnq_dequantizer
        exact = d.use_quantized_matmul and str(d.quantized_matmul_dtype) in ("int8", "uint8") and x.numel() / x.shape[2] >= 32 and not d.use_hadamard
        if not exact and y.shape == ref.shape:  # float / fp8 matmuls: the float tolerance of the parity tests
            scale = float(np.abs(ref).max()) or 1.0
            if float(np.abs(y - ref).max()) / scale <= {"bf16": 2 * 2.0 ** -8, "f16": 2 * 2.0 ** -11}[tag]:
                continue
        if y.shape != ref.shape or not np.array_equal(y, ref):
            bad.append((cfgname, nd, groups, mod.forward_func.__name__, cin, cout, ks, stride, pad, dil, h, w, b, tag, y.shape, ref.shape,
                        int((y != ref).sum()) if y.shape == ref.shape else -1))
            if verbose:
                print("MISMATCH", *bad[-1])
    if verbose:
        print("conv fuzz done, mismatches:", len(bad), "of", done, "geometries")
    return bad


if __name__ == "__main__":
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 0, int(sys.argv[2]) if len(sys.argv) > 2 else 40, variety="--variety" in sys.argv) else 0)
