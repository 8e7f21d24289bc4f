"""Time sdnq_hip_lowrank_down (t = x . svd_down^T, rank 32) at the FLUX activation shapes, graph-replayed.  SDNQ_HIP_LIB selects the library."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
for (m, k) in [(4608, 3072), (512, 3072), (4096, 3072), (4608, 15360), (4608, 12288), (4096, 1280)]:
    x = torch.randn(m, k, device=dev, dtype=torch.bfloat16)
    d = torch.randn(32, k, device=dev, dtype=torch.bfloat16)
    ref = (x.float() @ d.float().t())
    got = ops.lowrank_down(x, d).float()
    err = ((got - ref).abs().max() / ref.abs().max()).item()
    t = timed(lambda: ops.lowrank_down(x, d))
    print(f"lowrank_down {m}x{k}: {t:.2f} us  {2*m*k/t/1e6:.2f} TB/s  rel.err {err:.1e}")
