#!/usr/bin/env python3
"""Random sweep of the attention ROUTES against each other: sdnq_hip_attn (single launch up to 128 keys, else K / V prepare + forward kernel that
quantizes its own queries) must equal the three-call sequence sdnq_hip_attn_prepare (with Q) + sdnq_hip_attn_fwd BIT FOR BIT -- random batch,
grouped heads, lengths around the 32-key blocks and the 128-key limit, padded head dims, causal, bool / additive masks, strided query views,
zero rows.  usage: tools/fuzz_attention_routes.py [seed] [iterations]"""
import os, sys, random
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(seed=0, iters=80, verbose=True):
    from sdnq_amd import attention as A
    dev = torch.device("cuda:0")
    rng = random.Random(seed)
    g = torch.Generator().manual_seed(seed)
    bad = []
    for it in range(iters):
        dt = rng.choice([torch.bfloat16, torch.float16])
        z = rng.choice([1, 1, 2, 3])
        kh = rng.choice([1, 2, 3, 5])
        qh = kh * rng.choice([1, 1, 2, 4])
        d = rng.choice([40, 64, 64, 72, 128, 128, 8])
        kn = rng.choice([1, 31, 32, 33, 64, 77, 96, 127, 128, 129, 160, 300, 1024 + rng.randint(0, 40)])
        qn = rng.choice([1, 5, 32, 77, 128, 130, 333, 1024])
        causal = rng.random() < 0.25
        mk = rng.choice([None, None, "bool", "f32", "bf16"])
        smooth = rng.random() < 0.8
        q = torch.randn(z, qh, qn, d, generator=g).to(dt)
        k = (torch.randn(z, kh, kn, d, generator=g) + torch.randn(1, kh, 1, d, generator=g)).to(dt)
        v = torch.randn(z, kh, kn, d, generator=g).to(dt)
        if qn > 2:
            q[:, :, qn // 2] = 0
        if kn > 4 and rng.random() < 0.3:
            k[:, :, 1] = 0
        mask = None
        if mk == "bool":
            mask = torch.rand(rng.choice([1, z]), 1, qn, kn, generator=g) > 0.3
        elif mk is not None:
            mask = (torch.randn(1, rng.choice([1, qh]), 1, kn, generator=g) * 2).to(torch.float32 if mk == "f32" else torch.bfloat16)
        q, k, v = q.to(dev), k.to(dev), v.to(dev)
        if d % 8 == 0 and qh > 1 and rng.random() < 0.5:
            q = q.transpose(1, 2).contiguous().transpose(1, 2)
        mask = None if mask is None else mask.to(dev)
        try:
            one = A.sdnq_hip_atten(q, k, v, attn_mask=mask, is_causal=causal, smooth_k=smooth)
            qq, qs, kq, ks, vt = A.quantize_attn(q, k, v, smooth_k=smooth)
            m = A.prepare_mask(mask, qn, kn) if mask is not None else None
            three = A.atten_fwd(qq, qs, kq, ks, vt, kn, d ** -0.5, causal, dt, m, head_dim=d)
        except NotImplementedError:
            continue
        torch.cuda.synchronize()
        if not torch.equal(one.contiguous(), three.contiguous()):
            bad.append((it, str(dt), z, qh, kh, qn, kn, d, causal, mk, smooth, int((one.contiguous() != three.contiguous()).sum())))
            if verbose:
                print("MISMATCH", *bad[-1], flush=True)
    if verbose:
        print(f"attention route fuzz done: {len(bad)} mismatches in {iters} cases")
    return bad


if __name__ == "__main__":
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 0, int(sys.argv[2]) if len(sys.argv) > 2 else 80) else 0)
