#!/usr/bin/env python3
"""Development aid: the group-256 matrix-core rotation (sdnq_hip_hadamard) against a float64 x @ H256 reference: error statistics and
where inside a group the wrong elements sit."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdnq_amd import ops
from oracle import oracle as O
dev = torch.device("cuda:0")
H = torch.from_numpy(np.asarray(O.hadamard_matrix(256), dtype=np.float64)).to(dev)
for scale in (3.0, 0.02, 1e-4, 300.0):
    torch.manual_seed(1)
    x = (torch.randn(64, 512, device=dev) * scale).to(torch.bfloat16)
    y = ops.hadamard(x, 256).double()
    ref = (x.double().view(64, 2, 256) @ H).view(64, 512)
    err = (y - ref).abs()
    ulp = ref.abs().clamp_min(1e-30) * 2.0 ** -8
    bad = (err > ulp)
    print(f"scale {scale}: max err {err.max().item():.3e}  max |ref| {ref.abs().max().item():.3e}  elements beyond 1 bf16 ulp: {int(bad.sum())} of {bad.numel()}")
    if bad.any():
        idx = bad.nonzero()[:8]
        print("   first bad (row, col, got, ref):", [(int(i), int(j), float(y[i, j]), float(ref[i, j])) for i, j in idx])
        print("   bad columns mod 16 histogram:", torch.bincount((bad.nonzero()[:, 1] % 16), minlength=16).tolist())
        print("   bad columns // 16 % 16 histogram:", torch.bincount(((bad.nonzero()[:, 1] // 16) % 16), minlength=16).tolist())
