import torch
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
x = torch.randn(4608, 15360, device=dev).to(torch.bfloat16)
y = torch.empty_like(x)
q = torch.empty(4608, 15360, device=dev, dtype=torch.int8)
t = timed(lambda: y.copy_(x)); print(f"copy bf16 141MB->141MB: {t:.1f} us  {2*x.numel()*2/t/1e6:.2f} TB/s")
t = timed(lambda: q.copy_(x)); print(f"convert bf16->int8 141MB->71MB: {t:.1f} us  {x.numel()*3/t/1e6:.2f} TB/s")
t = timed(lambda: x.abs().amax(dim=-1)); print(f"abs+amax rows (read 141MB twice-ish): {t:.1f} us")
t = timed(lambda: torch.amax(x, dim=-1)); print(f"amax rows (read 141MB): {t:.1f} us  {x.numel()*2/t/1e6:.2f} TB/s")
