#!/usr/bin/env python3
"""Lab of the K-split 64x80 tile (csrc/gemm_ks.hip, tile id 28) against the heuristics' tile: bit equality, graph-replayed timings on
re-read and on distinct (cold) operands, and the phase stamps of the kernel.  usage: python tools/ks_lab.py [MxNxK ...]"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdnq_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda:0")
lib = _lib.load()
shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:] if "x" in a] or [(1024, 1280, 1280), (1024, 1280, 5120), (1024, 1280, 640), (1024, 1280, 2560), (1000, 1200, 1152)]
TILES = [int(v) for v in os.environ.get("KS_TILES", "1,21,28").split(",")]
ks_trace = getattr(ctypes.CDLL(_lib.LIB_PATH), "_Z22sdnq_internal_ks_tracePy")
ks_trace.argtypes = [ctypes.c_void_p]
ks_trace.restype = None


def timed(fn, n_launch, reps=5):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for i in range(n_launch):
            fn(i)
        s.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for i in range(n_launch):
                fn(i)
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / n_launch)
    return best


for (m, n, k) in shapes:
    for pool_n, tag in ((1, "re-read"), (48, "distinct operands")):
        g = torch.Generator(device=dev).manual_seed(m + n + k)
        xq = [torch.randint(-128, 128, (m, k), dtype=torch.int8, device=dev, generator=g) for _ in range(pool_n)]
        bs = [torch.randint(-128, 128, (n, k), dtype=torch.int8, device=dev, generator=g) for _ in range(pool_n)]
        xs = torch.rand(m, device=dev) * 0.02 + 1e-4
        sb = torch.rand(n, device=dev) * 0.02 + 1e-4
        bias = torch.randn(n, device=dev).to(torch.bfloat16)
        line = f"{m}x{n}x{k} {tag:18s}:"
        ref = None
        for t in TILES:
            lib.sdnq_hip_set_tile_override(t)
            y = ops.scaled_mm(ops.MM_I8, xq[0], bs[0], xs, sb, bias, torch.bfloat16)
            y2 = ops.scaled_mm(ops.MM_I8, xq[0], bs[0], xs, sb, None, torch.float16)
            if ref is None:
                ref = (y, y2)
            ok = torch.equal(y.view(torch.int16), ref[0].view(torch.int16)) and torch.equal(y2.view(torch.int16), ref[1].view(torch.int16))
            us = timed(lambda i: ops.scaled_mm(ops.MM_I8, xq[i % pool_n], bs[i % pool_n], xs, sb, bias, torch.bfloat16), max(pool_n, 32))
            line += f"  tile {t}: {us:7.2f} us{'' if ok else ' MISMATCH'}"
        lib.sdnq_hip_set_tile_override(-1)
        print(line, flush=True)
    # phase stamps (s_memtime ticks of 10 ns): entry, prologue issued, stage 0+1 landed, K loop done, partial sums exchanged, staged, stored
    buf = torch.zeros(1024 * 8, dtype=torch.int64, device=dev)
    ks_trace(buf.data_ptr())
    lib.sdnq_hip_set_tile_override(28)
    for _ in range(3):
        buf.zero_()
        ops.scaled_mm(ops.MM_I8, xq[0], bs[0], xs, sb, bias, torch.bfloat16)
        torch.cuda.synchronize()
    ks_trace(None)
    lib.sdnq_hip_set_tile_override(-1)
    t = buf.view(1024, 8).cpu()
    t = t[t[:, 0] > 0]
    if len(t):
        d = (t[:, 1:7] - t[:, 0:6]).float()
        print(f"   ks phases ({len(t)} workgroups; 10-ns ticks; mean / max): " + "  ".join(f"{a:.0f}/{b:.0f}" for a, b in zip(d.mean(0).tolist(), d.max(0).values.tolist()))
              + f"   span {(t[:, 6].max() - t[:, 0].min()).item() / 100:.2f} us", flush=True)
