#!/usr/bin/env python3
"""BUILD CONTAINER ONLY (imports the reference from /root/reference, which does not travel): random sweep of the host-side load-time
quantizer (sdnq_amd.quantizer on CPU tensors: the torch mirror that the HIP quantizer is tested against on the GPU) against the
REFERENCE's sdnq_quantize_layer -- same float layer, same config -> same state dict bit for bit and same dequantizer record.
usage: tools/fuzz_quantizer_vs_reference.py [seed] [iterations]"""
import os, sys, random
sys.argv_saved, sys.argv = sys.argv, ["x"]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
sys.path.insert(0, ROOT)
import make_golden as G  # installs the fake diffusers, imports the reference as `sdnq`  # noqa: E402
import torch  # noqa: E402
import sdnq_amd  # noqa: E402

WEIGHTS = ["int8", "uint8", "int4", "uint4", "int6", "uint7", "int5", "uint3", "int2", "uint2", "int3", "uint5", "uint6", "int7", "float8_e4m3fn",
           "float8_e5m2", "float4_e2m1fn", "float6_e3m2fn", "float5_e2m2fn", "float7_e3m3fn", "int12", "uint11", "float12_e4m7fn"]


def bits(t):
    t = t.detach().contiguous()
    if t.dtype == torch.bool:
        t = t.to(torch.uint8)
    return t.view(torch.uint8).reshape(-1) if t.element_size() > 0 else t


def run(seed=0, iters=200, verbose=True):
    rng = random.Random(seed)
    bad, done = [], 0
    fields = ["weights_dtype", "quantized_matmul_dtype", "group_size", "hadamard_group_size", "svd_rank", "use_quantized_matmul", "re_quantize_for_matmul",
              "use_hadamard", "is_packed", "is_unsigned", "is_integer", "layer_class_name"]
    for it in range(iters):
        wd = rng.choice(WEIGHTS)
        conv = rng.random() < 0.3
        dt = rng.choice([torch.float32, torch.bfloat16, torch.float16])
        gs = rng.choice([-1, 0, 0, 16, 32, 64])
        kw = dict(weights_dtype=wd, group_size=gs, dequantize_fp32=rng.random() < 0.7)
        torch.manual_seed(seed * 10000 + it)
        if conv:
            cin, cout, groups = 16 * rng.randint(1, 4), 16 * rng.randint(1, 4), rng.choice([1, 1, 2])
            cin, cout = cin * groups, cout * groups
            layer = rng.choice([torch.nn.Conv2d(cin, cout, rng.choice([1, 3]), groups=groups), torch.nn.Conv1d(cin, cout, 3, groups=groups)])
            kw.update(quant_conv=True, use_quantized_matmul_conv=rng.random() < 0.6)
        else:
            layer = torch.nn.Linear(16 * rng.randint(2, 24), 16 * rng.randint(2, 12), bias=rng.random() < 0.7)
            kw.update(use_quantized_matmul=rng.random() < 0.6)
        if rng.random() < 0.3:
            kw["quantized_matmul_dtype"] = rng.choice(["int8", "float8_e4m3fn", "uint8"])
        layer = layer.to(dt)
        with torch.no_grad():
            layer.weight.mul_(rng.choice([0.02, 1.0, 50.0]))
            layer.weight.view(layer.weight.shape[0], -1)[:, rng.randrange(layer.weight[0].numel())] *= 9
        import copy
        try:
            ref = G.sdnq_quantize_layer(copy.deepcopy(layer), G.SDNQConfig(**kw))[0]
        except Exception as e:  # noqa: BLE001  (a configuration the reference itself rejects)
            continue
        try:
            mine = sdnq_amd.sdnq_quantize_layer(copy.deepcopy(layer), sdnq_amd.SDNQConfig(**kw))[0]
        except NotImplementedError:
            continue
        done += 1
        why = []
        rq, mq = hasattr(ref, "sdnq_dequantizer"), hasattr(mine, "sdnq_dequantizer")
        if rq != mq:
            why.append(("quantized at all", rq, mq))
        elif rq:
            for f in fields:
                a, b = getattr(ref.sdnq_dequantizer, f), getattr(mine.sdnq_dequantizer, f)
                if f == "quantized_matmul_dtype":
                    a, b = {"fp8": "float8_e4m3fn"}.get(a, a), {"fp8": "float8_e4m3fn"}.get(b, b)
                if a != b:
                    why.append((f, a, b))
            if list(ref.sdnq_dequantizer.quantized_weight_shape) != list(mine.sdnq_dequantizer.quantized_weight_shape):
                why.append(("quantized_weight_shape", list(ref.sdnq_dequantizer.quantized_weight_shape), list(mine.sdnq_dequantizer.quantized_weight_shape)))
            if ref.forward_func.__name__ != mine.forward_func.__name__:
                why.append(("forward_func", ref.forward_func.__name__, mine.forward_func.__name__))
            for key in ("weight", "scale", "zero_point", "bias"):
                a, b = getattr(ref, key, None), getattr(mine, key, None)
                if (a is None) != (b is None):
                    why.append((key, "None-ness"))
                elif a is not None:
                    if tuple(a.shape) != tuple(b.shape) or a.dtype != b.dtype or tuple(a.stride()) != tuple(b.stride()):
                        why.append((key, "layout", tuple(a.shape), a.dtype, tuple(a.stride()), tuple(b.shape), b.dtype, tuple(b.stride())))
                    elif not torch.equal(bits(a), bits(b)):
                        why.append((key, "values", int((bits(a) != bits(b)).sum())))
        if why:
            bad.append((kw, type(layer).__name__, tuple(layer.weight.shape), str(dt), why))
            if verbose:
                print("MISMATCH", *bad[-1], flush=True)
    if verbose:
        print(f"quantizer fuzz vs the reference: {len(bad)} mismatches in {done} layers", flush=True)
    return bad


if __name__ == "__main__":
    a = sys.argv_saved
    sys.exit(1 if run(int(a[1]) if len(a) > 1 else 0, int(a[2]) if len(a) > 2 else 200) else 0)
