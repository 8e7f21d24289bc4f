#!/usr/bin/env python3
"""Host-side cost of one eager SDXL Linear step (no hipGraph): cProfile of the Python/ctypes path."""
import cProfile, os, pstats, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from sdnq_amd import shapes
dev = torch.device("cuda:0")
layers = bench.build_layers(shapes.sdxl_unet_layer_sequence(), dict(weights_dtype="int8", group_size=-1, use_quantized_matmul=True), dev)
for _ in range(3):
    bench.run_step(layers)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    bench.run_step(layers)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host launch time per step {(t1 - t0) / 5 * 1e3:.2f} ms; incl. GPU drain {(t2 - t0) / 5 * 1e3:.2f} ms; {len(layers)} layers")
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    bench.run_step(layers)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
