"""Lab: the grouped cross-attention k/v launch of the SDXL step (77 text tokens x 2048 -> 120 x 1280 / 20 x 640 channels, every layer its own
weights: 315 / 26 MB of int8 read once) on forced tiles; graph-replayed, each replay on cold weights (a second group of the same size is
run in between so nothing stays in the 256-MiB Infinity Cache).  usage: python tools/kv_group_lab.py"""
import sys

import torch

sys.path.insert(0, ".")
from sdnq_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda:0")
lib = _lib.load()
m, k = 77, 2048
x = torch.randn(m, k, device=dev).to(torch.bfloat16)
xq, xs = ops.rowquant(x, ops.MM_I8, 0)[:2]


def group(n_layers, n):
    return ops.GemmGroup([(torch.randint(-128, 128, (n, k), dtype=torch.int8, device=dev), torch.rand(n, device=dev) * 0.01 + 1e-4, None) for _ in range(n_layers)])


for (n_layers, n) in ((120, 1280), (20, 640)):
    ga, gb = group(n_layers, n), group(n_layers, n)
    mb = n_layers * n * k / 1e6
    line = f"{n_layers} x {n} ({mb:.0f} MB): "
    ref = None
    for tile in (-1, 2, 1, 17, 7, 10, 3, 23, 26):
        lib.sdnq_hip_set_tile_override(tile)
        try:
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                outs = ops.scaled_mm_grouped(ops.MM_I8, xq, xs, ga, torch.bfloat16)
                s.synchronize()
                if ref is None:
                    ref = [o.clone() for o in outs]
                ok = all(torch.equal(a, b) for a, b in zip(outs, ref))
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=s):
                    ops.scaled_mm_grouped(ops.MM_I8, xq, xs, ga, torch.bfloat16)
            best = 1e9
            for _ in range(5):
                ops.scaled_mm_grouped(ops.MM_I8, xq, xs, gb, torch.bfloat16)  # evict
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                g.replay()
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) * 1e3)
            line += f" tile {tile}: {best:.1f} us ({mb / best:.2f} TB/s){'' if ok else ' MISMATCH'} |"
        except Exception as e:  # noqa: BLE001
            line += f" tile {tile}: refused ({str(e)[:40]}) |"
    lib.sdnq_hip_set_tile_override(-1)
    print(line, flush=True)
