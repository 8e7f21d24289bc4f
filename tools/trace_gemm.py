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
LOWRANK = "--lowrank" in sys.argv  # the SVD epilogue on FLUX shapes, next to the plain epilogue
W8A16 = "--w8a16" in sys.argv      # the fused dequantize GEMM next to the int8 one
if LOWRANK:
    shapes = [(4608, 3072, 3072), (4608, 12288, 3072)]
if os.environ.get("SHAPES"):  # SHAPES="m,n,k;m,n,k"
    shapes = [tuple(int(v) for v in t.split(",")) for t in os.environ["SHAPES"].split(";")]
names = ["entry", "issued", "stage0", "steady_end", "mainloop_end", "epi_compute", "stored"]
for (m, n, k) in shapes:
    x = torch.randn(m, k, device=dev, dtype=torch.bfloat16)
    b = torch.randint(-128, 128, (n, k), dtype=torch.int8, device=dev)
    sb = torch.rand(n, device=dev) * 0.01
    bias = torch.randn(n, device=dev, dtype=torch.bfloat16)
    xq, xs, _, _ = ops.rowquant(x, ops.MM_I8)
    t_lr = torch.randn(m, 32, device=dev, dtype=torch.bfloat16)
    up = torch.randn(n, 32, device=dev, dtype=torch.bfloat16)
    for mode in (("plain", "lowrank") if LOWRANK else (("plain", "w8a16") if W8A16 else ("plain",))):
        def run():
            if mode == "plain":
                ops.scaled_mm(ops.MM_I8, xq, b, xs, sb, bias, torch.bfloat16)
            elif mode == "w8a16":
                ops.linear_w8a16(x, b, sb, None, bias)
            else:
                ops.scaled_mm_lowrank(ops.MM_I8, xq, b, xs, sb, bias, t_lr, up, None, None, torch.bfloat16)
        buf = np.zeros(4096 * 8, dtype=np.uint64)
        for _ in range(5):
            run()
        torch.cuda.synchronize()
        lib.sdnq_hip_debug_trace(buf.ctypes.data_as(ctypes.c_void_p), buf.size)  # read + clear
        run()
        torch.cuda.synchronize()
        lib.sdnq_hip_debug_trace(buf.ctypes.data_as(ctypes.c_void_p), buf.size)
        t = buf.reshape(4096, 8).astype(np.int64)
        nwg = int((t[:, 0] != 0).sum())
        t = t[:nwg]
        t0 = t[:, 0].min()
        print(f"M={m} N={n} K={k} [{mode}]: workgroups traced {nwg}; s_memtime ticks (10 ns); kernel span {(t[:, 6].max() - t0) / 100:.1f} us")
        print("   phase           " + "  ".join(f"{nm:>12s}" for nm in names))
        d = np.diff(t[:, :7], axis=1)
        print("   per-WG deltas   " + " " * 14 + "  ".join(f"{v:12.0f}" for v in d.mean(0)))
        print("   per-WG max      " + " " * 14 + "  ".join(f"{v:12.0f}" for v in d.max(0)))
