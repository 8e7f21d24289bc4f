"""Time the M = 1 forward (linear_skinny) of quantized layers on FLUX shapes: int4 + Hadamard, int8.  GPU box only."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sdnq_amd

dev = torch.device("cuda:0")


def timed(fn, reps=30):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn(); s.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps):
                fn()
        g.replay(); s.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(3):
            g.replay()
        e1.record(s); s.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (3 * reps)


for name, cfg in [("int4+had256", dict(weights_dtype="int4", use_hadamard=True, use_quantized_matmul=True)),
                  ("int4", dict(weights_dtype="int4", use_quantized_matmul=True)),
                  ("int8", dict(weights_dtype="int8", use_quantized_matmul=True)),
                  ("uint4", dict(weights_dtype="uint4", use_quantized_matmul=True))]:
    for (n, k) in [(18432, 3072), (9216, 3072)]:
        torch.manual_seed(0)
        lin = torch.nn.Linear(k, n, device=dev, dtype=torch.bfloat16)
        layer = sdnq_amd.sdnq_quantize_layer(lin, sdnq_amd.SDNQConfig(**cfg))[0]
        x = torch.randn(1, k, device=dev, dtype=torch.bfloat16)
        with torch.no_grad():
            t = timed(lambda: layer(x))
        bits = 4 if "int4" in name else 8
        print(f"M=1 {name:12s} {n:6d} x {k:5d}: {t:7.2f} us   {n * k * bits / 8 / t / 1e6:5.2f} TB/s of codes")
