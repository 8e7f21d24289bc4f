import os, sys, torch
sys.path.insert(0, os.getcwd())
from sdnq_amd import ops
dev = torch.device("cuda:0")
def timed(fn, reps=30):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn(); s.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps): fn()
        g.replay(); s.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(3): g.replay()
        e1.record(s); s.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (3 * reps)
for (m, k) in [(4608, 3072), (512, 3072), (4608, 15360), (4608, 12288)]:
    x = torch.randn(m, k, device=dev, dtype=torch.bfloat16)
    t = timed(lambda: ops.rowquant(x, ops.MM_I8, 256))
    print(f"rowquant_had256 {m}x{k}: {t:.2f} us  {3*m*k/t/1e6:.2f} TB/s")
