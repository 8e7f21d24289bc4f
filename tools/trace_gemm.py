#!/usr/bin/env python3
"""Development aid: per-workgroup phase timeline of the GEMM kernel (needs the -DSDNQ_TRACE build: tools/build_trace.sh)."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdnq_amd import _lib, ops
dev = torch.device("cuda:0")
lib = _lib.load()
lib.sdnq_hip_debug_trace.argtypes = [ctypes.c_void_p, ctypes.c_int]
shapes = [(4096, 640, 640), (1024, 1280, 1280), (1024, 1280, 5120), (77, 640, 2048), (4096, 5120, 640)]
names = ["entry", "issued", "stage0", "steady_end", "mainloop_end", "epi_compute", "stored"]
for (m, n, k) in shapes:
    x = torch.randn(m, k, device=dev, dtype=torch.bfloat16)
    b = torch.randint(-128, 128, (n, k), dtype=torch.int8, device=dev)
    sb = torch.rand(n, device=dev) * 0.01
    bias = torch.randn(n, device=dev, dtype=torch.bfloat16)
    xq, xs, _, _ = ops.rowquant(x, ops.MM_I8)
    buf = np.zeros(4096 * 8, dtype=np.uint64)
    for _ in range(5):
        ops.scaled_mm(ops.MM_I8, xq, b, xs, sb, bias, torch.bfloat16)
    torch.cuda.synchronize()
    lib.sdnq_hip_debug_trace(buf.ctypes.data_as(ctypes.c_void_p), buf.size)  # read + clear
    ops.scaled_mm(ops.MM_I8, xq, b, xs, sb, bias, torch.bfloat16)
    torch.cuda.synchronize()
    lib.sdnq_hip_debug_trace(buf.ctypes.data_as(ctypes.c_void_p), buf.size)
    t = buf.reshape(4096, 8).astype(np.int64)
    nwg = int((t[:, 0] != 0).sum())
    t = t[:nwg]
    t0 = t[:, 0].min()
    rel = (t[:, :7] - t0)
    print(f"M={m} N={n} K={k}: workgroups traced {nwg}; clock ticks relative to first workgroup entry (100 MHz ticks? see below)")
    print("   phase           " + "  ".join(f"{nm:>12s}" for nm in names))
    d = np.diff(t[:, :7], axis=1)
    print("   per-WG deltas   " + " " * 14 + "  ".join(f"{v:12.0f}" for v in d.mean(0)))
