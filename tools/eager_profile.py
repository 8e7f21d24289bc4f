#!/usr/bin/env python3
"""cProfile of the eager SDXL Linear step's host side (development aid: relative weights; the profiler itself inflates Python frames)."""
import cProfile, os, pstats, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

dev = torch.device("cuda:0")
torch.cuda.set_device(0)
shape_list, cfg_kwargs, mm_name, tokens = bench.workload_config(sys.argv[1] if len(sys.argv) > 1 else "sdxl_int8")
layers = bench.build_layers(shape_list, cfg_kwargs, dev)
bench.link_shared_input_layers(layers)
for _ in range(3):
    bench.run_step(layers)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    bench.run_step(layers)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
