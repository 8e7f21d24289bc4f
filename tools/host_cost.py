#!/usr/bin/env python3
"""Where the host time of ONE eager SDNQ Linear call goes (development aid): the module call, the forward function, the ops wrapper, the
binding call alone and the allocations alone, each timed over many calls with the GPU queue drained between batches (so the figures
are host issue cost, not back-pressure of the launch queue)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sdnq_amd  # noqa: E402
from sdnq_amd import linear as L  # noqa: E402
from sdnq_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
lin = torch.nn.Linear(1280, 1280, bias=True).to(torch.bfloat16).to(dev)
mod, _ = sdnq_amd.sdnq_quantize_layer(lin, sdnq_amd.SDNQConfig(weights_dtype="int8", group_size=-1, use_quantized_matmul=True))
xs_in = [torch.randn(1024, 1280, device=dev, dtype=torch.bfloat16) for _ in range(64)]


def timed(fn, n=64, rounds=8):
    best = 1e9
    for _ in range(rounds):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            fn(xs_in[i])
        best = min(best, (time.perf_counter() - t0) / n)
        torch.cuda.synchronize()
        L.clear_activation_cache()
    return best * 1e6


st = L._state(mod)
wq, ws, _ = L._prepare_mm_weights(mod, st, ops.MM_I8)
bias = mod.bias
print(f"module call            {timed(lambda x: mod(x)):7.2f} us")
print(f"forward_func(mod, x)   {timed(lambda x: mod.forward_func(mod, x)):7.2f} us")
print(f"ops.linear_w8a8        {timed(lambda x: ops.linear_w8a8(ops.MM_I8, x, wq, ws, bias, torch.bfloat16, 0)):7.2f} us")
print(f"3 x torch.empty        {timed(lambda x: (torch.empty((1024, 1280), device=dev, dtype=torch.bfloat16), torch.empty((1024, 1280), device=dev, dtype=torch.int8), torch.empty((1024,), device=dev, dtype=torch.float32))):7.2f} us")
print(f"_state(mod)            {timed(lambda x: L._state(mod)):7.2f} us")
print(f"ops._stream(x)         {timed(lambda x: ops._stream(x)):7.2f} us")
print(f"tensor_key(x)          {timed(lambda x: L.tensor_key(x)):7.2f} us")
print(f"x.reshape(-1, k)       {timed(lambda x: x.reshape(-1, 1280)):7.2f} us")
y = torch.empty((1024, 1280), device=dev, dtype=torch.bfloat16)
xq = torch.empty((1024, 1280), device=dev, dtype=torch.int8)
xsb = torch.empty((1024,), device=dev, dtype=torch.float32)
lib = ops._lib.load()
s = ops._stream(xs_in[0])
