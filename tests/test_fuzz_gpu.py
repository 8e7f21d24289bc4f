"""Bounded slices (a few seconds each) of the random-shape sweeps in tools/fuzz_*.py under -m gpu: the driver's GPU run executes
them, so a shape class the hand-picked parity cases miss still has a chance of being seen.  Every sweep is bit-exact against its
checker (the oracle, or sdnq_hip_dequant's values + the float GEMM)."""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [101, 202])
def test_fuzz_linear_w8a8_bit_exact(seed, gpu_device):
    import fuzz_linear
    assert fuzz_linear.run(seed, 24, verbose=False) == []


@pytest.mark.parametrize("seed", [303, 404])
def test_fuzz_conv_int8_bit_exact(seed, gpu_device):
    import fuzz_conv
    assert fuzz_conv.run(seed, 24, verbose=False) == []
    assert fuzz_conv.run(seed + 1, 30, verbose=False, variety=True) == []  # unsigned weights, uint8 / fp8 matmuls, 4-bit, float mode, groups


@pytest.mark.parametrize("seed", [505, 606])
def test_fuzz_fused_dequantize_gemm_bit_exact(seed, gpu_device):
    import fuzz_w8a16
    assert fuzz_w8a16.run(seed, 30, verbose=False) == []


@pytest.mark.parametrize("seed", [707, 808])
def test_fuzz_one_launch_linear_equals_the_two_launch_route(seed, gpu_device):
    """Random shapes / dtypes / strides / degenerate rows through sdnq_hip_linear_w8a8_fused (the GEMM that row-quantizes its own
    activation rows in LDS), eager and replayed from a hipGraph, bit for bit against sdnq_hip_linear_w8a8 (tools/fuzz_fused.py)."""
    import fuzz_fused
    assert fuzz_fused.run(seed, 40, verbose=False) == []


@pytest.mark.parametrize("seed", [11, 12, 13])
def test_fuzz_configuration_space_vs_oracle(seed, gpu_device):
    """Storage dtype x group size x matmul dtype x Hadamard x SVD x scale dtype x odd shapes (tools/fuzz_modes.py).  Round 4: this
    sweep found a GPU memory fault at K < one LDS stage, epilogue terms the compiler had fused where torch rounds twice, and an
    oracle line that rounded twice where torch fuses."""
    import fuzz_modes
    assert fuzz_modes.run(seed, 60, verbose=False) == []


@pytest.mark.parametrize("seed", [21, 22])
def test_fuzz_operators_vs_oracle(seed, gpu_device):
    """scaled_mm (odd shapes, every bias form / output dtype), row quantization (odd K, padded rows, asymmetric), dequantize of every
    storage dtype, quantized attention (tools/fuzz_ops.py).  Round 4: this sweep found the lost sign of a -0.0 activation in the fp8
    row quantizer."""
    import fuzz_ops
    assert fuzz_ops.run(seed, 40, verbose=False) == []


def test_fuzz_large_problems_every_tile_agrees_with_the_small_tiles(gpu_device):
    """Large ragged problems through the tile heuristics (256x256 half-tile ring, 256x128, 256x160, 128x128) == the same problem on
    forced 64x128 tiles, bit for bit for int8, incl. the low-rank and zero-point epilogues (tools/fuzz_tiles.py)."""
    import fuzz_tiles
    assert fuzz_tiles.run(31, 16, verbose=False) == []


@pytest.mark.parametrize("seed", [41, 42])
def test_fuzz_host_side_reuse_never_serves_stale_results(seed, gpu_device):
    """Random call / in-place edit / invalidate / step-boundary sequences over shared tensors: the activation cache, the linked
    projection groups and the per-module weight state always equal a computation on a fresh clone (tools/fuzz_host_state.py)."""
    import fuzz_host_state
    assert fuzz_host_state.run(seed, 250, verbose=False) == []


@pytest.mark.parametrize("seed", [51, 52])
def test_fuzz_attention_routes_are_bit_identical(seed, gpu_device):
    """sdnq_hip_attn (one launch up to 128 keys; else K / V prepare + a forward kernel that quantizes its own queries) == the three-call
    sequence with the separate pass over Q, bit for bit: random grouped heads, lengths around the 32-key blocks and the 128-key limit,
    padded head dims, causal, bool / additive masks, strided queries, zero rows (tools/fuzz_attention_routes.py)."""
    import fuzz_attention_routes
    assert fuzz_attention_routes.run(seed, 40, verbose=False) == []
