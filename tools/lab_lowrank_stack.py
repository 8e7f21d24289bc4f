import os, sys, torch
sys.path.insert(0, os.getcwd())
from sdnq_amd import ops
sys.path.insert(0, "tools")
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
for (m, k) in [(4608, 3072), (4096, 3072), (512, 3072)]:
    x = torch.randn(m, k, device=dev, dtype=torch.bfloat16)
    for r in (32, 64, 96, 128):
        d = torch.randn(r, k, device=dev, dtype=torch.bfloat16)
        ds = [d[i*32:(i+1)*32].contiguous() for i in range(r // 32)]
        t1 = timed(lambda: ops.lowrank_down(x, d))
        t2 = timed(lambda: [ops.lowrank_down(x, e) for e in ds])
        print(f"{m}x{k} rank {r}: stacked {t1:.2f} us, {r//32} separate {t2:.2f} us")
