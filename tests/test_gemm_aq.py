"""GPU parity of the ONE-launch w8a8 Linear (sdnq_hip_linear_w8a8_fused, csrc/gemm_aq.hip: the GEMM workgroup row-quantizes its own
activation rows in LDS -- linear_int8.py:15-22, 64 + kernels/triton_scaled_mm.py:194-232 behind one launch).

The contract: the same bits as the two-launch route (sdnq_hip_linear_w8a8 = sdnq_hip_rowquant + sdnq_hip_scaled_mm) and as the CPU
oracle (int8: bit-exact; fp8: the two routes bit-identical, <= 2 ulp of the output dtype against the oracle's float64-free sum),
on ragged M / N, both K stage counts, both 16-bit dtypes, with and without bias, rows of zeros, rows with ties and a strided input.
"""
import numpy as np
import pytest
import torch

from oracle import oracle as O
from tests.modules_util import to_f32_numpy

pytestmark = pytest.mark.gpu

from sdnq_amd import _lib, ops  # noqa: E402

SHAPES = [(64, 128, 128), (100, 136, 256), (1024, 1280, 1280), (77, 640, 640), (333, 1288, 1152), (200, 8, 384), (4096, 320, 640),
          (257, 264, 1280), (1000, 1280, 640)]


def _inputs(m, n, k, dtype, seed, dev):
    g = torch.Generator().manual_seed(seed)
    x = (torch.randn(m, k, generator=g) * torch.exp(torch.randn(m, 1, generator=g))).to(dtype)
    if m > 4:
        x[1] = 0                                   # an all-zero row: scale 0, codes 0 (the general path of the quantizer)
        x[2] = (torch.randint(-127, 128, (k,), generator=g).float() * 0.5).to(dtype)  # half-integers under scale amax / 127: ties
        x[2, 0] = 63.5
        x[3, : k // 2] = 0
    b = torch.randint(-128, 128, (n, k), dtype=torch.int8, generator=g)
    sb = torch.rand(n, generator=g) * 0.02 + 1e-4
    bias = torch.randn(n, generator=g).to(dtype)
    return x.to(dev), b.to(dev), sb.to(dev), bias.to(dev)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("shape", SHAPES)
def test_fused_int8_equals_two_launch_route_and_oracle(shape, dtype, gpu_device):
    m, n, k = shape
    x, b, sb, bias = _inputs(m, n, k, dtype, m + 7 * n + k, gpu_device)
    for with_bias in (True, False):
        bb = bias if with_bias else None
        y2, xq, xs = ops.linear_w8a8(ops.MM_I8, x, b, sb, bb, dtype)
        y1 = ops.linear_w8a8_fused(ops.MM_I8, x, b, sb, bb, dtype)
        torch.cuda.synchronize()
        assert torch.equal(y1.view(torch.int16), y2.view(torch.int16)), (shape, dtype, with_bias, int((y1.view(torch.int16) != y2.view(torch.int16)).sum()))
        # against the oracle: row quantization (linear_int8.py:15-22) then the scaled matmul
        xq_o, xs_o, _ = O.rowquant(x.float().cpu().numpy(), "int8")
        assert np.array_equal(xq.cpu().numpy(), xq_o) and np.array_equal(xs.cpu().numpy().reshape(-1), xs_o.reshape(-1))
        ref = O.scaled_mm("int8", xq_o, b.cpu().numpy(), xs_o.reshape(-1), sb.cpu().numpy(), bb.float().cpu().numpy() if with_bias else None,
                          "bf16" if dtype == torch.bfloat16 else "f16")
        assert np.array_equal(to_f32_numpy(y1), ref), (shape, dtype, with_bias)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("shape", [(64, 128, 128), (333, 1288, 1152), (1024, 1280, 1280), (1000, 640, 640)])
def test_fused_fp8_equals_two_launch_route(shape, dtype, gpu_device):
    m, n, k = shape
    x, _, sb, bias = _inputs(m, n, k, dtype, 3 * m + n + k, gpu_device)
    g = torch.Generator().manual_seed(k)
    b = (torch.randn(n, k, generator=g) * 50).clamp(-448, 448).to(torch.float8_e4m3fn).to(gpu_device)
    for with_bias in (True, False):
        bb = bias if with_bias else None
        y2, _, _ = ops.linear_w8a8(ops.MM_FP8, x, b, sb, bb, dtype)
        y1 = ops.linear_w8a8_fused(ops.MM_FP8, x, b, sb, bb, dtype)
        torch.cuda.synchronize()
        # the same codes through the same MFMA in the same K order: bit-identical
        assert torch.equal(y1.view(torch.int16), y2.view(torch.int16)), (shape, dtype, with_bias, int((y1.view(torch.int16) != y2.view(torch.int16)).sum()))


def test_fused_strided_rows_and_repeatability(gpu_device):
    """A row-strided activation view (ldx > K) and six back-to-back runs (a mis-ordered LDS-DMA / barrier shows up as run-to-run noise)."""
    m, n, k = 1024, 1280, 1280
    x, b, sb, bias = _inputs(m, n, k, torch.bfloat16, 5, gpu_device)
    wide = torch.zeros(m, k + 256, dtype=torch.bfloat16, device=gpu_device)
    wide[:, :k] = x
    xv = wide[:, :k]
    ref, _, _ = ops.linear_w8a8(ops.MM_I8, x, b, sb, bias, torch.bfloat16)
    for _ in range(6):
        y = ops.linear_w8a8_fused(ops.MM_I8, xv, b, sb, bias, torch.bfloat16)
        assert torch.equal(y.view(torch.int16), ref.view(torch.int16))


def test_fused_route_of_the_module_forward(gpu_device):
    """An SDNQLinear whose input is its own takes the one-launch route after two steps in which nobody used its parked quantized
    activation; outputs never change.  (`..._supported` answers per shape; the forward asks it.)"""
    import sdnq_amd
    from sdnq_amd import linear as L
    torch.manual_seed(0)
    lin = torch.nn.Linear(1280, 1280, bias=True).to(torch.bfloat16)
    layer, _ = sdnq_amd.sdnq_quantize_layer(lin, sdnq_amd.SDNQConfig(weights_dtype="int8", group_size=-1, use_quantized_matmul=True))
    layer = layer.to(gpu_device)
    assert _lib.load().sdnq_hip_linear_w8a8_fused_supported(0, 1, 1, 1024, 1280, 1280) == 1
    assert _lib.load().sdnq_hip_linear_w8a8_fused_supported(0, 1, 1, 1024, 1280, 5120) == 0   # rows do not fit LDS
    assert _lib.load().sdnq_hip_linear_w8a8_fused_supported(0, 1, 1, 1024, 10240, 1280) == 0  # too many column tiles per row block
    outs = []
    calls = []
    real = ops.linear_w8a8_fused
    try:
        ops.linear_w8a8_fused = lambda *a, **kw: (calls.append(1), real(*a, **kw))[1]
        x = torch.randn(1024, 1280, device=gpu_device).to(torch.bfloat16)
        for step in range(5):
            L.clear_activation_cache()
            outs.append(layer(x.clone()))
        torch.cuda.synchronize()
    finally:
        ops.linear_w8a8_fused = real
    assert calls, "the module never took the one-launch route"
    for o in outs[1:]:
        assert torch.equal(o.view(torch.int16), outs[0].view(torch.int16))
