"""Lab of the one-launch w8a8 Linear (csrc/gemm_aq.hip) against the two-launch route: graph-replayed pairs on cold / warm operands and the
phase stamps of the fused kernel.  usage: python tools/aq_lab.py [MxNxK ...]"""
import ctypes
import sys

import torch

sys.path.insert(0, ".")
from sdnq_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda:0")
lib = _lib.load()
shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or [(1024, 1280, 1280), (1024, 1280, 640), (4096, 640, 640), (1024, 640, 1280), (2048, 1280, 1280), (1024, 3840, 1280)]


def timed(fn, pool, reps=5):
    """`pool` launches in ONE graph (each on its own operands), replayed `reps` times: us per launch."""
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for i in range(len(pool)):
            fn(i)
        s.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for i in range(len(pool)):
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
        best = min(best, e0.elapsed_time(e1) * 1e3 / len(pool))
    return best


for (m, n, k) in shapes:
    for pool_n, tag in ((1, "re-read"), (64, "distinct operands")):
        xs_ = [torch.randn(m, k, device=dev).to(torch.bfloat16) for _ in range(pool_n)]
        bs = [torch.randint(-128, 128, (n, k), dtype=torch.int8, device=dev) for _ in range(pool_n)]
        sb = torch.rand(n, device=dev) * 0.02 + 1e-4
        bias = torch.randn(n, device=dev).to(torch.bfloat16)
        reps = max(pool_n, 32)
        idx = list(range(reps))
        two = timed(lambda i: ops.linear_w8a8(ops.MM_I8, xs_[i % pool_n], bs[i % pool_n], sb, bias, torch.bfloat16), idx)
        sup = lib.sdnq_hip_linear_w8a8_fused_supported(0, 1, 1, m, n, k)
        try:
            one = timed(lambda i: ops.linear_w8a8_fused(ops.MM_I8, xs_[i % pool_n], bs[i % pool_n], sb, bias, torch.bfloat16), idx)
        except Exception as e:  # noqa: BLE001
            one = float("nan")
            print("  fused route refused:", e)
        print(f"{m}x{n}x{k} {tag:18s}: two launches {two:7.2f} us | one launch {one:7.2f} us   (supported() = {sup})", flush=True)
    # phase stamps (shader clock, 100 MHz memtime ticks -> reported in ticks): entry, loads+ring issued, quantized, K loop done, synced, staged, stored
    try:
        buf = torch.zeros(1024 * 8, dtype=torch.int64, device=dev)
        aq_trace = getattr(ctypes.CDLL(_lib.LIB_PATH), "_Z22sdnq_internal_aq_tracePy")
        aq_trace.argtypes = [ctypes.c_void_p]
        aq_trace.restype = None
        aq_trace(buf.data_ptr())
        x = torch.randn(m, k, device=dev).to(torch.bfloat16)
        b = torch.randint(-128, 128, (n, k), dtype=torch.int8, device=dev)
        for _ in range(3):
            buf.zero_()
            ops.linear_w8a8_fused(ops.MM_I8, x, b, sb, bias, torch.bfloat16)
            torch.cuda.synchronize()
        aq_trace(None)
        t = buf.view(1024, 8).cpu()
        t = t[t[:, 0] > 0]
        d = (t[:, 1:7] - t[:, 0:6]).double()
        names = ["entry->issued", "issued->quantized", "K loop", "drain+sync", "epilogue math", "stores"]
        print("   phases (memtime ticks, mean over", t.shape[0], "workgroups):", ", ".join(f"{nm} {v:.0f}" for nm, v in zip(names, d.mean(0).tolist())),
              "| total", f"{(t[:, 6] - t[:, 0]).double().mean():.0f}", flush=True)
    except Exception as e:  # noqa: BLE001
        print("   trace failed:", e)
