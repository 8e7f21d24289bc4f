#!/usr/bin/env python3
"""Development aid: phase timeline of ONE GEMM problem as it runs INSIDE a bench.py step (cold weights from HBM, its row quantizer in
front of it) next to the same launch replayed alone.  Needs the -DSDNQ_TRACE build (tools/build_trace.sh):
    SDNQ_HIP_LIB=$PWD/build/libsdnq_hip_trace.so python tools/trace_in_step.py [workload] "M,N,K;M,N,K"   [SDNQ_HIP_TILE_MAP=... to force tiles]"""
import ctypes, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from sdnq_amd import _lib  # noqa: E402

workload = sys.argv[1] if len(sys.argv) > 1 else "sdxl_int8"
shapes = [tuple(int(v) for v in t.split(",")) for t in (sys.argv[2] if len(sys.argv) > 2 else "1024,1280,1280;1024,1280,5120;4096,640,640").split(";")]
dev = torch.device("cuda:0")
lib = _lib.load()
lib.sdnq_hip_debug_trace.argtypes = [ctypes.c_void_p, ctypes.c_int]
lib.sdnq_hip_debug_trace_shape.argtypes = [ctypes.c_int] * 3
shape_list, cfg_kwargs, mm_name, tokens = bench.workload_config(workload)[:4]
layers = bench.build_layers(shape_list, cfg_kwargs, dev)
bench.link_shared_input_layers(layers)  # (SDNQ_BENCH_PREFETCH_HINT=1 in the environment: bench.run_step sets the weight-prefetch hints)
for _ in range(2):
    bench.run_step(layers)
torch.cuda.synchronize()
side = torch.cuda.Stream(device=dev)
with torch.cuda.stream(side):
    bench.run_step(layers); side.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        bench.run_step(layers)
torch.cuda.synchronize()
names = ["entry", "issued", "stage0", "steady_end", "mainloop_end", "epi_compute", "stored"]
buf = np.zeros(4096 * 8, dtype=np.uint64)
for (m, n, k) in shapes:
    lib.sdnq_hip_debug_trace_shape(m, n, k)
    for _ in range(3):
        graph.replay()
    torch.cuda.synchronize()
    lib.sdnq_hip_debug_trace(buf.ctypes.data_as(ctypes.c_void_p), buf.size)  # read + clear
    graph.replay()
    torch.cuda.synchronize()
    lib.sdnq_hip_debug_trace(buf.ctypes.data_as(ctypes.c_void_p), buf.size)
    t = buf.reshape(4096, 8).astype(np.int64)
    nwg = int((t[:, 0] != 0).sum())
    if nwg == 0:
        print(f"M={m} N={n} K={k}: no such launch in the step"); continue
    t = t[:nwg]
    t0 = t[:, 0].min()
    d = np.diff(t[:, :7], axis=1)
    print(f"M={m} N={n} K={k} IN STEP: workgroups {nwg}; kernel span {(t[:, 6].max() - t0)} ticks; first-entry -> last-entry {t[:, 0].max() - t0}")
    print("   phase           " + "  ".join(f"{nm:>12s}" for nm in names))
    print("   per-WG mean     " + " " * 14 + "  ".join(f"{v:12.0f}" for v in d.mean(0)))
    print("   per-WG max      " + " " * 14 + "  ".join(f"{v:12.0f}" for v in d.max(0)))
lib.sdnq_hip_debug_trace_shape(0, 0, 0)
