#!/usr/bin/env python3
"""Stateful sweep of the host-side reuse machinery (identity-keyed activation cache, linked projection groups, per-module weight state):
a random sequence of layer calls on a small pool of SHARED tensors, in-place edits of those tensors and of layer parameters,
invalidations, step boundaries and inference-mode calls.  Every result must equal the same layer on a fresh CLONE of the input
(a tensor no cache has ever seen) bit for bit.  `run(seed, steps)` is also driven, bounded, by tests/test_fuzz_gpu.py under -m gpu."""
import os, sys, random
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


class Attn(torch.nn.Module):
    def __init__(self, d):
        super().__init__()
        self.to_q, self.to_k, self.to_v = (torch.nn.Linear(d, d, bias=False) for _ in range(3))
        self.to_out = torch.nn.Linear(d, d)


class Net(torch.nn.Module):
    def __init__(self, d=256):
        super().__init__()
        self.attn1, self.attn2 = Attn(d), Attn(d)
        self.ff1, self.ff2 = torch.nn.Linear(d, 2 * d), torch.nn.Linear(2 * d, d)


def run(seed: int = 0, steps: int = 300, verbose: bool = True) -> list:
    import sdnq_amd
    from sdnq_amd import linear as L
    dev = torch.device("cuda:0")
    rng = random.Random(seed)
    torch.manual_seed(seed)
    d = 256
    cfgs = [dict(weights_dtype="int8", group_size=-1, use_quantized_matmul=True), dict(weights_dtype="int8", group_size=-1, use_quantized_matmul=False),
            dict(weights_dtype="int4", use_quantized_matmul=True), dict(weights_dtype="uint8", group_size=-1, use_quantized_matmul=True, quantized_matmul_dtype="int8")]
    net = Net(d).to(torch.bfloat16).to(dev)
    net, _ = sdnq_amd.apply_sdnq_to_module(net, sdnq_amd.SDNQConfig(minimum_allowed_numel=1024, minimum_allowed_channel_size=32, **rng.choice(cfgs)))
    sdnq_amd.accelerate(net)
    layers = [(n, m) for n, m in net.named_modules() if hasattr(m, "sdnq_dequantizer")]
    pool = {d: [torch.randn(rng.choice([48, 200]), d, device=dev, dtype=torch.bfloat16) for _ in range(3)],
            2 * d: [torch.randn(64, 2 * d, device=dev, dtype=torch.bfloat16) for _ in range(2)]}
    bad = []
    for step in range(steps):
        op = rng.random()
        if op < 0.70:
            name, mod = rng.choice(layers)
            k = mod.sdnq_dequantizer.in_features
            t = rng.choice(pool[k])
            view = rng.random() < 0.15
            x = t[: t.shape[0] // 2] if view else t
            if rng.random() < 0.1:
                with torch.inference_mode():
                    y = mod(x)
                    want = mod(x.clone())
            else:
                with torch.no_grad():
                    y = mod(x)
                    want = mod(x.clone())
            if not torch.equal(y, want):
                bad.append((step, name, "call", tuple(x.shape), int((y != want).sum().item())))
                if verbose:
                    print("MISMATCH", *bad[-1], flush=True)
        elif op < 0.82:  # in-place edit of a shared activation (version bump): every derived quantity must be recomputed
            t = rng.choice(rng.choice(list(pool.values())))
            with torch.no_grad():
                if rng.random() < 0.5:
                    t.mul_(rng.choice([0.5, 2.0, -1.0]))
                else:
                    t[rng.randrange(t.shape[0])] = torch.randn(t.shape[1], device=dev, dtype=t.dtype)
        elif op < 0.88:  # raw write behind autograd's back + explicit invalidation
            t = rng.choice(rng.choice(list(pool.values())))
            t.data.view(torch.int16).bitwise_xor_(torch.tensor(rng.choice([0, -32768]), dtype=torch.int16, device=dev))  # flip every sign bit, or not
            sdnq_amd.invalidate(t)
        elif op < 0.93:  # a parameter changes in place: scale or bias
            name, mod = rng.choice(layers)
            with torch.no_grad():
                if mod.bias is not None and rng.random() < 0.5:
                    mod.bias.add_(0.25)
                else:
                    mod.scale.mul_(2.0)
        elif op < 0.97:
            L.clear_activation_cache()  # a step boundary
        else:
            sdnq_amd.invalidate(None)
    torch.cuda.synchronize()
    if verbose:
        fp = L._FP
        print(f"host-state fuzz done: {len(bad)} mismatches in {steps} operations"
              + (f" ({fp.plan_calls()} layer calls carried by fast-path plans)" if fp is not None else ""), flush=True)
    return bad


if __name__ == "__main__":
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 0, int(sys.argv[2]) if len(sys.argv) > 2 else 300) else 0)
