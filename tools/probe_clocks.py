#!/usr/bin/env python3
"""Development probe: launch overhead / clock behaviour on the MI355X box."""
import os, subprocess, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdnq_amd import ops
dev = torch.device("cuda:0")

def sh(cmd):
    try:
        print(subprocess.run(cmd, shell=True, capture_output=True, text=True, timeout=60).stdout[-1500:])
    except Exception as e:
        print("ERR", e)

sh("rocm-smi --showclocks --showperflevel --showpower 2>&1 | head -40")

def graph_time(fn, reps, replays=5):
    s = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(s):
        fn(); s.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps): fn()
        g.replay(); s.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(replays): g.replay()
        e1.record(s); s.synchronize()
    return e0.elapsed_time(e1) / (replays * reps) * 1e3

x = torch.zeros(1024, device=dev)
print("tiny torch kernel (x.add_(1)) in graph: %.2f us/launch" % graph_time(lambda: x.add_(1), 200))
m, n, k = 1024, 1280, 1280
xa = torch.randn(m, k, device=dev, dtype=torch.bfloat16)
b = torch.randint(-128, 128, (n, k), dtype=torch.int8, device=dev)
sb = torch.rand(n, device=dev) * 0.01
bias = torch.randn(n, device=dev, dtype=torch.bfloat16)
xq, xs, _, _ = ops.rowquant(xa, ops.MM_I8)
f = lambda: ops.scaled_mm(ops.MM_I8, xq, b, xs, sb, bias, torch.bfloat16)
for reps in (20, 200, 2000):
    print(f"gemm 1024x1280x1280 graph reps={reps}: {graph_time(f, reps):.2f} us/launch")
# sustained heavy load then re-measure
big_a = torch.randint(-128, 128, (8192, 4096), dtype=torch.int8, device=dev)
big_b = torch.randint(-128, 128, (8192, 4096), dtype=torch.int8, device=dev)
sa8 = torch.rand(8192, device=dev); sb8 = torch.rand(8192, device=dev)
t0 = time.time()
while time.time() - t0 < 1.0:
    for _ in range(20): ops.scaled_mm(ops.MM_I8, big_a, big_b, sa8, sb8, None, torch.bfloat16)
    torch.cuda.synchronize()
sh("rocm-smi --showclocks 2>&1 | grep -iE 'sclk|mclk|fclk' | head")
print(f"after 1s heavy load: gemm small reps=200: {graph_time(f, 200):.2f} us/launch")
t = graph_time(lambda: ops.scaled_mm(ops.MM_I8, big_a, big_b, sa8, sb8, None, torch.bfloat16), 10)
print(f"big 8192x8192x4096: {t:.1f} us  {2*8192*8192*4096/t/1e6:.0f} TOP/s")
