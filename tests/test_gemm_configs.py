"""GPU parity of the scaled-matmul tile configurations and of the grouped (unit-table) launch.

Every configuration of sdnq_amd/csrc/gemm.hip:launch_tiles -- including the ping-pong (LD_PP) schedules -- must give the bits
the CPU oracle gives (int8: exact integer accumulation, so the schedule cannot change a bit), on ragged shapes that exercise
the M / N / K tails, and the grouped launch must equal one sdnq_hip_scaled_mm per layer.
"""
import numpy as np
import pytest
import torch

from oracle import oracle as O
from tests.modules_util import to_f32_numpy

pytestmark = pytest.mark.gpu

from sdnq_amd import _lib, ops  # noqa: E402

TILES = list(range(29))


@pytest.fixture()
def tile_override():
    lib = _lib.load()
    yield lib.sdnq_hip_set_tile_override
    lib.sdnq_hip_set_tile_override(-1)


@pytest.mark.parametrize("tile", TILES)
def test_every_tile_configuration_bit_exact_vs_oracle(tile, gpu_device, tile_override):
    # (K % 128 == 0 shapes: the half-tile ring of configuration 20 needs whole 128-byte K tiles and falls back otherwise)
    for (m, n, k) in ((300, 392, 528), (513, 1288, 208), (64, 64, 64), (1031, 264, 1296), (300, 392, 512), (1031, 520, 1280), (257, 264, 128)):
        g = torch.Generator().manual_seed(m + 3 * n + tile)
        a = torch.randint(-128, 128, (m, k), dtype=torch.int8, generator=g)
        b = torch.randint(-128, 128, (n, k), dtype=torch.int8, generator=g)
        sa = torch.rand(m, generator=g) * 0.02 + 1e-4
        sb = torch.rand(n, generator=g) * 0.02 + 1e-4
        bias = torch.randn(n, generator=g).to(torch.bfloat16)
        for with_bias in (True, False):
            tile_override(tile)
            out = ops.scaled_mm(ops.MM_I8, a.to(gpu_device), b.to(gpu_device), sa.to(gpu_device), sb.to(gpu_device),
                                bias.to(gpu_device) if with_bias else None, torch.bfloat16)
            torch.cuda.synchronize()
            ref = O.scaled_mm("int8", a.numpy(), b.numpy(), sa.numpy(), sb.numpy(), bias.float().numpy() if with_bias else None, "bf16")
            got = to_f32_numpy(out)
            assert np.array_equal(got, ref), (tile, (m, n, k), with_bias, int((got != ref).sum()))


@pytest.mark.parametrize("tile", TILES)
def test_every_tile_configuration_repeatable_at_model_size(tile, gpu_device, tile_override):
    """Race screen: a mis-ordered LDS-DMA / fragment read shows up as run-to-run differences or as a difference from the
    single-buffer-safe default; 1024 x 3840 x 1280 and a long-K shape, 6 runs each, every output element compared."""
    for (m, n, k) in ((1024, 3840, 1280), (1024, 1280, 5120), (4096, 1920, 640)):
        g = torch.Generator(device=gpu_device).manual_seed(tile * 31 + n)
        a = torch.randint(-128, 128, (m, k), dtype=torch.int8, device=gpu_device, generator=g)
        b = torch.randint(-128, 128, (n, k), dtype=torch.int8, device=gpu_device, generator=g)
        sa = torch.rand(m, device=gpu_device, generator=g) * 0.02 + 1e-4
        sb = torch.rand(n, device=gpu_device, generator=g) * 0.02 + 1e-4
        bias = torch.randn(n, device=gpu_device, generator=g).to(torch.bfloat16)
        tile_override(-1)
        ref = ops.scaled_mm(ops.MM_I8, a, b, sa, sb, bias, torch.bfloat16)
        # the default configuration against the CPU oracle on the first and last 64 rows
        for rows in (slice(0, 64), slice(m - 64, m)):
            want = O.scaled_mm("int8", a[rows].cpu().numpy(), b.cpu().numpy(), sa[rows].cpu().numpy(), sb.cpu().numpy(),
                               bias.float().cpu().numpy(), "bf16")
            assert np.array_equal(to_f32_numpy(ref[rows]), want)
        tile_override(tile)
        for _ in range(6):
            out = ops.scaled_mm(ops.MM_I8, a, b, sa, sb, bias, torch.bfloat16)
            assert torch.equal(out.view(torch.int16), ref.view(torch.int16)), (tile, (m, n, k))


@pytest.mark.parametrize("tile", [0, 4, 6, 7])
def test_fp8_tile_configurations_match_default(tile, gpu_device, tile_override):
    m, n, k = 520, 1288, 1344
    g = torch.Generator().manual_seed(5 + tile)
    a = (torch.randn(m, k, generator=g) * 60).clamp(-448, 448).to(torch.float8_e4m3fn).to(gpu_device)
    b = (torch.randn(n, k, generator=g) * 60).clamp(-448, 448).to(torch.float8_e4m3fn).to(gpu_device)
    sa = (torch.rand(m, generator=g) * 0.02 + 1e-4).to(gpu_device)
    sb = (torch.rand(n, generator=g) * 0.02 + 1e-4).to(gpu_device)
    bias = torch.randn(n, generator=g).to(torch.bfloat16).to(gpu_device)
    tile_override(-1)
    ref = ops.scaled_mm(ops.MM_FP8, a, b, sa, sb, bias, torch.bfloat16).float()
    tile_override(tile)
    out = ops.scaled_mm(ops.MM_FP8, a, b, sa, sb, bias, torch.bfloat16).float()
    # fp32 accumulation order differs between stage widths: 2 bf16 ulp of the output scale
    assert float((out - ref).abs().max()) <= 2 * 2.0 ** -8 * float(ref.abs().max())


@pytest.mark.parametrize("m", [77, 200, 1024])
@pytest.mark.parametrize("with_bias", [False, True])
def test_grouped_matmul_equals_one_matmul_per_layer(m, with_bias, gpu_device):
    k = 2048 if m == 77 else 640
    widths = [640, 1280, 640, 1920, 1280]
    g = torch.Generator(device=gpu_device).manual_seed(m)
    a = torch.randint(-128, 128, (m, k), dtype=torch.int8, device=gpu_device, generator=g)
    sa = torch.rand(m, device=gpu_device, generator=g) * 0.02 + 1e-4
    members = []
    for n in widths:
        w = torch.randint(-128, 128, (n, k), dtype=torch.int8, device=gpu_device, generator=g)
        ws = torch.rand(n, device=gpu_device, generator=g) * 0.02 + 1e-4
        bias = torch.randn(n, device=gpu_device, generator=g).to(torch.bfloat16) if with_bias else None
        members.append((w, ws, bias))
    group = ops.GemmGroup(members)
    assert group.unit_n == 640 and group.n_units == sum(widths) // 640
    outs = ops.scaled_mm_grouped(ops.MM_I8, a, sa, group, torch.bfloat16)
    for (w, ws, bias), out in zip(members, outs):
        want = ops.scaled_mm(ops.MM_I8, a, w, sa, ws, bias, torch.bfloat16)
        assert out.is_contiguous() and out.shape == want.shape
        assert torch.equal(out.view(torch.int16), want.view(torch.int16))
    # one layer against the CPU oracle as well
    w, ws, bias = members[1]
    ref = O.scaled_mm("int8", a.cpu().numpy(), w.cpu().numpy(), sa.cpu().numpy(), ws.cpu().numpy(),
                      None if bias is None else bias.float().cpu().numpy(), "bf16")
    assert np.array_equal(to_f32_numpy(outs[1]), ref)


def test_grouped_matmul_rejects_bad_groups(gpu_device):
    w = torch.zeros((96, 64), dtype=torch.int8, device=gpu_device)
    ws = torch.ones(96, device=gpu_device)
    with pytest.raises(_lib.SdnqHipError):
        ops.GemmGroup([(w, ws, None)])  # 96 channels: no 64-wide unit


@pytest.mark.parametrize("wdt", ["int8", "uint8"])
@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
def test_fused_dequant_gemm_equals_dequant_then_linear(wdt, dt, gpu_device, tile_override):
    """sdnq_hip_linear_w8a16 (weights converted between LDS and the MFMA) against the two-launch path it replaces (sdnq_hip_dequant
    + sdnq_hip_linear_float) on every tile configuration and on ragged shapes -- the weight values are identical by construction
    and the k order of the fp32 accumulation is the same, so the outputs are expected bit-identical -- and against the CPU oracle."""
    import sdnq_amd
    from sdnq_amd import linear as L
    from tests.modules_util import oracle_from_module
    from tests.test_gpu_parity import assert_close_float
    tag = "bf16" if dt == torch.bfloat16 else "f16"
    for (m, n, k, bias) in ((200, 392, 528, True), (1031, 1288, 1296, False), (4096, 640, 640, True), (33, 64, 64, True)):
        torch.manual_seed(m + n)
        lin = torch.nn.Linear(k, n, bias=bias).to(dt).to(gpu_device)
        if wdt == "uint8":
            with torch.no_grad():
                lin.weight.add_(0.02)
        mod, _ = sdnq_amd.sdnq_quantize_layer(lin, sdnq_amd.SDNQConfig(weights_dtype=wdt, group_size=-1, use_quantized_matmul=False))
        assert mod.forward_func.__name__ == "quantized_linear_forward" and (mod.zero_point is not None) == (wdt == "uint8")
        x = torch.randn(m, k, device=gpu_device, dtype=dt)
        old, old_max = L.FUSED_DEQUANT_GEMM, L.FUSED_DEQUANT_GEMM_MAX_FLOP
        L.FUSED_DEQUANT_GEMM_MAX_FLOP = 1e18  # every shape through the fused kernel
        try:
            L.FUSED_DEQUANT_GEMM = False
            tile_override(-1)
            want = mod(x)
            L.FUSED_DEQUANT_GEMM = True
            for tile in (-1, 0, 1, 2, 3, 4):
                tile_override(tile)
                got = mod(x)
                assert got.shape == want.shape
                diff = int((got != want).sum())
                assert diff == 0 or float((got.float() - want.float()).abs().max()) <= 2 * (2.0 ** (-8 if tag == "bf16" else -11)) * float(want.float().abs().max()), (wdt, tag, (m, n, k), tile, diff)
        finally:
            L.FUSED_DEQUANT_GEMM, L.FUSED_DEQUANT_GEMM_MAX_FLOP = old, old_max
            tile_override(-1)
        if m <= 1100:
            ref = O.forward(oracle_from_module(mod), x.float().cpu().numpy(), tag)
            assert_close_float(to_f32_numpy(got), ref, tag, (wdt, tag, (m, n, k)))
