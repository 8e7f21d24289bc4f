"""Development aid: host-side cost of one eager SDNQLinear call (a layer small enough that the GPU is never the bound), with a cProfile breakdown."""
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
import sdnq_amd
dev = torch.device("cuda:0")
lin = torch.nn.Linear(1280, 1280).to(torch.bfloat16).to(dev)
mod, _ = sdnq_amd.sdnq_quantize_layer(lin, sdnq_amd.SDNQConfig(weights_dtype="int8", group_size=-1, use_quantized_matmul=True))
sdnq_amd.accelerate(mod)
xs = [torch.randn(64, 1280, device=dev, dtype=torch.bfloat16) for _ in range(8)]
N = 3000
with torch.no_grad():
    for i in range(50): mod(xs[i % 8])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(N): y = mod(xs[i % 8])
    t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"per call: issue {1e6*(t1-t0)/N:.2f} us, incl. drain {1e6*(t2-t0)/N:.2f} us")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
with torch.no_grad():
    for i in range(N): y = mod(xs[i % 8])
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(14)
