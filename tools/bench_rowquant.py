import sys, os
sys.path.insert(0, "/root/repo")
import torch
from sdnq_amd import ops
dev = torch.device("cuda:0")
def timed(fn, reps=20):
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
for (m, k) in [(16384, 4096), (4608, 15360), (4608, 12288), (4096, 12288), (512, 12288), (4608, 3072), (4608, 6144), (1024, 10240)]:
    x = torch.randn(m, k, device=dev).to(torch.bfloat16)
    for had in (0, 256):
        os.environ.pop("SDNQ_HIP_RQ_SPLIT", None)
        t = timed(lambda: ops.rowquant(x, ops.MM_I8, had))
        print(f"rowquant {m} x {k} had={had}: {t:7.2f} us  {(3 * m * k) / t / 1e6:5.2f} TB/s")
