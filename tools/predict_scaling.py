#!/usr/bin/env python3
"""PREDICTED 1/2/4/8-GPU numbers for DESIGN.md section 8 (no multi-GPU node was available to any round): the measured 1-GPU step +
the link model of sdnq_amd/parallel.py.  Replicas: N independent latents, no collective.  TP (column shards): per rank the row
quantization / low-rank down-projection / M = 1 layers stay whole (every rank quantizes the full activation), the GEMMs shrink to
1 / W (floored at a per-launch floor), and every sharded layer gathers 2 M N / W bytes per peer over W - 1 links in parallel.
usage: tools/predict_scaling.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdnq_amd import shapes

LINK = 153e9          # bytes/s per xGMI link and direction
RCCL_CALL = 20e-6     # one all_gather_into_tensor call of a few MB on one node (latency part; an assumption until measured)
PUSH_CALL = 3e-6      # post + rendezvous of the copy-free gather (measured intra-device: one small kernel + flag round trip)
HBM_COPY = 4e12       # bytes/s of the re-assembly pass (read + write counted)
GEMM_FLOOR = 8e-6     # a GEMM launch does not get shorter than this (section 6)

# measured on one MI355X (profiles/r03_bench_flux_int8_svd_kernel_stats.csv): per step
flux = dict(step=41.5e-3, gemm=32.0e-3)   # scaled-mm launches / everything else (row quantization, lowrank_down, M = 1 layers)
sdxl = dict(step=7.95e-3, gemm=5.99e-3)


def tp(seq, meas, world, call, reassemble):
    layers = [(m, k, n) for (_, m, k, n, _, _) in seq if m >= 32]
    n_gemm = len(layers)
    gemm = max(meas["gemm"] / world, n_gemm * GEMM_FLOOR) if world > 1 else meas["gemm"]
    other = meas["step"] - meas["gemm"]
    gather = 0.0
    for (m, k, n) in layers:
        if world == 1:
            continue
        per_link = 2.0 * m * n / world           # bytes each peer sends me (and I send each peer), one link each
        gather += call + per_link / LINK
        if reassemble:
            gather += 2 * 2.0 * m * n / HBM_COPY
    return other + gemm + gather, gather


for name, seq, meas, ops in (("flux_int8_svd", shapes.flux_dev_layer_sequence(), flux, None), ("sdxl_int8", shapes.sdxl_unet_layer_sequence(), sdxl, 4.355e12)):
    print(name)
    for w in (1, 2, 4, 8):
        r, gr = tp(seq, meas, w, RCCL_CALL, True)
        p, gp = tp(seq, meas, w, PUSH_CALL, False)
        line = f"  N={w}: TP rccl {r * 1e3:6.1f} ms (gathers {gr * 1e3:5.1f})   TP peer {p * 1e3:6.1f} ms (gathers {gp * 1e3:5.1f})"
        if ops:
            line += f"   replicas {w * ops / meas['step'] / 1e12:7.0f} TOP/s whole job"
        print(line)
