#!/usr/bin/env python3
"""Development aid: phase stamps of the K-split tile (gemm_ks.hip) as it runs INSIDE a bench.py step.  The trace pointer is a kernel
argument, so it is baked into the captured graph: every ks launch of the step writes its stamps and the LAST one of the step stays.
    SDNQ_HIP_KSPLIT=0 SDNQ_HIP_TILE_MAP=1024x1280x5120=28 python tools/ks_trace_in_step.py   (exactly one shape on tile 28 at a time)"""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from sdnq_amd import _lib  # noqa: E402

workload = sys.argv[1] if len(sys.argv) > 1 else "sdxl_int8"
dev = torch.device("cuda:0")
_lib.load()
ks_trace = getattr(ctypes.CDLL(_lib.LIB_PATH), "_Z22sdnq_internal_ks_tracePy")
ks_trace.argtypes = [ctypes.c_void_p]
ks_trace.restype = None
shape_list, cfg_kwargs, mm_name, tokens = bench.workload_config(workload)[:4]
layers = bench.build_layers(shape_list, cfg_kwargs, dev)
bench.link_shared_input_layers(layers)
for _ in range(2):
    bench.run_step(layers)
torch.cuda.synchronize()
buf = torch.zeros(1024 * 8, dtype=torch.int64, device=dev)
ks_trace(buf.data_ptr())
side = torch.cuda.Stream(device=dev)
with torch.cuda.stream(side):
    bench.run_step(layers); side.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        bench.run_step(layers)
torch.cuda.synchronize()
for _ in range(3):
    graph.replay()
torch.cuda.synchronize()
buf.zero_()
graph.replay()
torch.cuda.synchronize()
ks_trace(None)
t = buf.view(1024, 8).cpu()
t = t[t[:, 0] > 0]
if not len(t):
    print("no ks launch in the step")
else:
    d = (t[:, 1:7] - t[:, 0:6]).float()
    print(f"ks phases IN STEP ({len(t)} workgroups; shader cycles; mean / max): entry->issued, ->stages 0+1 landed, K loop, exchange, epilogue, stored")
    print("   " + "  ".join(f"{a:.0f}/{b:.0f}" for a, b in zip(d.mean(0).tolist(), d.max(0).values.tolist()))
          + f"   span {(t[:, 6].max() - t[:, 0].min()).item()} cycles; first-entry -> last-entry {(t[:, 0].max() - t[:, 0].min()).item()}")
