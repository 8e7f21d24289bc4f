#!/usr/bin/env python3
"""Development aid: a forced tile id (launch_tiles' `force` list) against forced 64x128 tiles on a few ragged int8 problems, bit-exact.
usage: tools/micro/check_tile.py <tile id> [M N K ...]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sdnq_amd import _lib, ops
lib = _lib.load(); dev = torch.device("cuda:0"); tile = int(sys.argv[1])
probs = [(16384, 320, 2880), (16384, 320, 8640), (4096, 640, 5760), (1000, 328, 208), (129, 160, 64), (5000, 1920, 1296)]
g = torch.Generator(device=dev).manual_seed(0)
bad = 0
for (m, n, k) in probs:
    a = torch.randint(-128, 128, (m, k), dtype=torch.int8, device=dev, generator=g); b = torch.randint(-128, 128, (n, k), dtype=torch.int8, device=dev, generator=g)
    sa = torch.rand(m, device=dev, generator=g) * 0.02 + 1e-4; sb = torch.rand(n, device=dev, generator=g) * 0.02 + 1e-4
    bias = torch.randn(n, device=dev, generator=g).to(torch.bfloat16)
    for bb in (None, bias):
        lib.sdnq_hip_set_tile_override(tile); got = ops.scaled_mm(ops.MM_I8, a, b, sa, sb, bb, torch.bfloat16)
        lib.sdnq_hip_set_tile_override(1); want = ops.scaled_mm(ops.MM_I8, a, b, sa, sb, bb, torch.bfloat16)
        lib.sdnq_hip_set_tile_override(-1); torch.cuda.synchronize()
        if not torch.equal(got.view(torch.int16), want.view(torch.int16)):
            bad += 1; print("MISMATCH", tile, m, n, k, bb is not None, int((got != want).sum()))
print("tile", tile, "mismatches", bad)
