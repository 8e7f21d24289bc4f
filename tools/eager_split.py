#!/usr/bin/env python3
"""Is the eager SDXL Linear step bound by the host (Python + HIP launch calls) or by the GPU queue (gaps between dependent dispatches of
one stream)?  Per step: the time until the LAST launch call has returned (no synchronize: host issue time, with the queue drained
before), the time to completion, and the same step replayed as one hipGraph."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

dev = torch.device("cuda:0")
torch.cuda.set_device(0)
shape_list, cfg_kwargs, mm_name, tokens = bench.workload_config(sys.argv[1] if len(sys.argv) > 1 else "sdxl_int8")
layers = bench.build_layers(shape_list, cfg_kwargs, dev)
bench.link_shared_input_layers(layers)
for _ in range(3):
    bench.run_step(layers)
torch.cuda.synchronize()
issue, total = [], []
for _ in range(20):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    bench.run_step(layers)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    issue.append((t1 - t0) * 1e3); total.append((t2 - t0) * 1e3)
issue.sort(); total.sort()
print(f"eager step: host issue {issue[len(issue) // 2]:.2f} ms (min {issue[0]:.2f}) | to completion {total[len(total) // 2]:.2f} ms (min {total[0]:.2f}) | {len(layers)} layer calls")
# back to back (what bench.py times): the queue never drains
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20):
    bench.run_step(layers)
torch.cuda.synchronize()
print(f"eager, 20 steps back to back: {(time.perf_counter() - t0) / 20 * 1e3:.2f} ms per step")
# host-only cost: the same Python with the GPU work replaced by nothing is not available; instead: kernels per step
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    bench.run_step(layers); side.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        bench.run_step(layers)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20):
    g.replay()
torch.cuda.synchronize()
print(f"hipGraph replay: {(time.perf_counter() - t0) / 20 * 1e3:.2f} ms per step")
