#!/usr/bin/env python3
"""Tile-configuration sweep of the int8 / fp8 scaled matmul (development aid): every configuration id of
sdnq_amd/csrc/gemm.hip:launch_tiles on a list of shapes, each checked BIT-EXACTLY against the default configuration's output and
timed as graph-replayed back-to-back launches with HIP events on the launch stream.
usage: python tools/sweep_gemm.py [int8|fp8] [sdxl|big|flux|all] [ids=0,1,2,...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdnq_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda:0")
mm = ops.MM_FP8 if (len(sys.argv) > 1 and sys.argv[1] == "fp8") else ops.MM_I8
which = sys.argv[2] if len(sys.argv) > 2 else "sdxl"
ids = [int(v) for v in sys.argv[3].split("=")[1].split(",")] if len(sys.argv) > 3 else list(range(17))
SDXL = [(4096, 640, 640), (4096, 1920, 640), (4096, 5120, 640), (4096, 640, 2560), (1024, 1280, 1280), (1024, 3840, 1280),
        (1024, 10240, 1280), (1024, 1280, 5120), (77, 1280, 2048)]
BIG = [(8192, 8192, 8192), (16384, 8192, 4096)]
FLUX = [(4608, 3072, 3072), (4608, 9216, 3072), (4608, 12288, 3072), (4608, 3072, 15360), (512, 3072, 3072)]
if os.environ.get("SHAPES"):  # SHAPES="m,n,k;m,n,k"
    shapes = [tuple(int(v) for v in t.split(",")) for t in os.environ["SHAPES"].split(";")]
else:
    shapes = {"sdxl": SDXL, "big": BIG, "flux": FLUX, "all": SDXL + BIG + FLUX}[which]
lib = _lib.load()


def timed(fn, reps):
    s = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(s):
        fn()
        s.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps):
                fn()
        g.replay()
        s.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(3):
            g.replay()
        e1.record(s)
        s.synchronize()
    return e0.elapsed_time(e1) / (3 * reps) * 1e3  # us per call


print(f"# {'int8' if mm == ops.MM_I8 else 'fp8'} scaled-mm, bf16 out, bias; us per launch (graph replay, incl. ~1.6 us boundary) / TOP/s; '!' = output differs")
print("# ids: 0 256x256 pipe | 1 64x128 dma | 2 64x64 pipe | 3 256x128 pipe | 4 256x256 PP | 5 256x128 PP ns3 | 6 128x256 PP ns4 | "
      "7 128x128 PP bk128 ns3 | 8 128x128 PP bk64 ns4 | 9 64x128 PP | 10 128x128 pipe | 11 128x256 PP ns3 | 12 256x160 pipe ns4 | 13 256x160 pipe ns3 | 14 128x320 pipe ns3 | 15 256x160 dma ns4 | 16 128x320 pipe ns4")
for (m, n, k) in shapes:
    x = torch.randn(m, k, device=dev, dtype=torch.bfloat16)
    if mm == ops.MM_I8:
        b = torch.randint(-128, 128, (n, k), dtype=torch.int8, device=dev)
    else:
        b = (torch.randn(n, k, device=dev) * 30).to(torch.float8_e4m3fn)
    sb = torch.rand(n, device=dev) * 0.01
    bias = torch.randn(n, device=dev, dtype=torch.bfloat16)
    xq, xs, _, _ = ops.rowquant(x, mm)
    lib.sdnq_hip_set_tile_override(-1)
    ref = ops.scaled_mm(mm, xq, b, xs, sb, bias, torch.bfloat16)
    reps = 20 if 2 * m * n * k < 1e11 else 4
    t0 = timed(lambda: ops.scaled_mm(mm, xq, b, xs, sb, bias, torch.bfloat16), reps)
    line = f"M={m:6d} N={n:6d} K={k:6d}: default {t0:8.2f} us {2 * m * n * k / t0 / 1e6:7.1f} |"
    for tid in ids:
        lib.sdnq_hip_set_tile_override(tid)
        try:
            out = ops.scaled_mm(mm, xq, b, xs, sb, bias, torch.bfloat16)
            torch.cuda.synchronize()
            ok = torch.equal(out.view(torch.int16), ref.view(torch.int16))
            t = timed(lambda: ops.scaled_mm(mm, xq, b, xs, sb, bias, torch.bfloat16), reps)
            line += f" {tid}:{t:7.2f}{'' if ok else '!'}"
        except Exception as e:  # noqa: BLE001
            line += f" {tid}:ERR({type(e).__name__})"
    lib.sdnq_hip_set_tile_override(-1)
    print(line, flush=True)
