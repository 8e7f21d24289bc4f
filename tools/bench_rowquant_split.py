import sys, os
sys.path.insert(0, "/root/repo")
import torch
from sdnq_amd import ops
dev = torch.device("cuda:0")
def timed(fn, reps=50):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn(); s.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps): fn()
        g.replay(); s.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(5): g.replay()
        e1.record(s); s.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (5 * reps)
for (m, k) in [(1024, 1280), (4096, 640), (1024, 5120), (4096, 320), (4096, 2560), (1024, 640)]:
    x = torch.randn(m, k, device=dev).to(torch.bfloat16)
    print(f"rowquant {m} x {k} split={os.environ.get('SDNQ_HIP_RQ_SPLIT','auto')}: {timed(lambda: ops.rowquant(x, ops.MM_I8, 0)):6.2f} us")
