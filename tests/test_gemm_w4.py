"""N1 (north_star: "the packed-weight unpack ... fused into the GEMM"): the quantized matmul on STORED 4-bit codes (csrc/gemm_w4.hip).

sdnq_hip_scaled_mm_w4 must give, bit for bit, what the two-step route gives -- sdnq_hip_requant (the reference's re_quantize_matmul,
dequantizer.py:166-239, itself pinned on the `*_group64_*` golden fixtures) followed by sdnq_hip_scaled_mm -- and the tables of
sdnq_hip_lut4_build must BE that re-quantization: expanding them with the stored codes reproduces sdnq_hip_requant's operand."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

CFGS = [
    dict(weights_dtype="uint4", group_size=64),
    dict(weights_dtype="int4", group_size=64),
    dict(weights_dtype="uint4", group_size=128),
    dict(weights_dtype="float4_e2m1fn", group_size=64),
    dict(weights_dtype="uint4", group_size=64, use_hadamard=True, hadamard_group_size=64),
]


def make_layer(cfg, n, k, bias, dtype, device, seed=0):
    import sdnq_amd
    torch.manual_seed(seed)
    lin = torch.nn.Linear(k, n, bias=bias).to(dtype)
    with torch.no_grad():
        lin.weight.mul_(torch.rand(n, 1).to(dtype) * 3 + 0.2)  # rows of different ranges
    mod, _ = sdnq_amd.sdnq_quantize_layer(lin, sdnq_amd.SDNQConfig(use_quantized_matmul=True, **cfg))
    return mod.to(device)


def unpack_nibbles(codes: torch.Tensor, n: int, k: int) -> torch.Tensor:
    b = codes.reshape(-1).view(torch.uint8).reshape(n, k // 2).to(torch.int64)
    out = torch.empty((n, k), dtype=torch.int64, device=codes.device)
    out[:, 0::2] = b & 15
    out[:, 1::2] = b >> 4
    return out


@pytest.mark.parametrize("cfg", CFGS, ids=lambda c: "-".join(str(v) for v in c.values()))
def test_tables_are_the_requantization(cfg, gpu_device):
    from sdnq_amd import linear as L, ops
    n, k = 264, 640
    mod = make_layer(cfg, n, k, True, torch.bfloat16, gpu_device)
    assert mod.sdnq_dequantizer.re_quantize_for_matmul
    st = L._state(mod)
    wq, ws = ops.requant(st.qw, ops.MM_I8)
    lut, ws2 = ops.lut4_build(st.qw, ops.MM_I8)
    assert torch.equal(ws, ws2)
    lut3, _ = ops.lut4_build(st.qw, ops.MM_I8, ws)  # row scales known: the same tables
    assert torch.equal(lut, lut3)
    codes = unpack_nibbles(st.qw.keep[0], n, k)                                   # [N, K] values 0..15
    tabs = lut.view(torch.int8).reshape(n, k // 64, 16).to(torch.int64)          # [N, K / 64, 16]
    expanded = torch.gather(tabs.repeat_interleave(64, dim=1), 2, codes.unsqueeze(-1)).squeeze(-1)
    assert torch.equal(expanded.to(torch.int8), wq)


@pytest.mark.parametrize("cfg", CFGS[:3], ids=lambda c: "-".join(str(v) for v in c.values()))
@pytest.mark.parametrize("m,n,k", [(1024, 1280, 1280), (300, 264, 640), (65, 136, 128), (2048, 640, 2560), (130, 1288, 384)])
def test_fused_matmul_equals_requant_then_scaled_mm(cfg, m, n, k, gpu_device):
    from sdnq_amd import linear as L, ops
    for dtype, bias in ((torch.bfloat16, True), (torch.float16, False)):
        mod = make_layer(cfg, n, k, bias, dtype, gpu_device, seed=m + n)
        st = L._state(mod)
        wq, ws = ops.requant(st.qw, ops.MM_I8)
        lut, _ = ops.lut4_build(st.qw, ops.MM_I8)
        g = torch.Generator(device=gpu_device).manual_seed(k)
        x = (torch.randn(m, k, device=gpu_device, generator=g) * 2).to(dtype)
        xq, xs, _, _ = ops.rowquant(x, ops.MM_I8)
        want = ops.scaled_mm(ops.MM_I8, xq, wq, xs, ws, mod.bias, dtype)
        got = ops.scaled_mm_w4(xq, st.qw.keep[0], lut, xs, ws, mod.bias, dtype)
        assert torch.equal(got.view(torch.int16), want.view(torch.int16)), (cfg, m, n, k, dtype, int((got != want).sum()))
        # strided activation rows (a view of a wider buffer) and repeatability
        wide = torch.zeros(m, k + 64, dtype=torch.int8, device=gpu_device)
        wide[:, :k] = xq
        for _ in range(3):
            again = ops.scaled_mm_w4(wide[:, :k], st.qw.keep[0], lut, xs, ws, mod.bias, dtype)
            assert torch.equal(again.view(torch.int16), want.view(torch.int16))


@pytest.mark.parametrize("cfg", [CFGS[0], CFGS[4]], ids=["uint4-g64", "uint4-g64-hadamard"])
def test_per_call_mode_takes_the_fused_route_and_equals_the_cached_mode(cfg, gpu_device, monkeypatch):
    """Module level: with SDNQ_HIP_CACHE_WEIGHTS=0 a few-row 4-bit layer keeps codes + tables (no int8 copy, no re-quantization launch)
    and computes the bits of the cached mode and of the oracle; eager and replayed from a hipGraph."""
    from sdnq_amd import linear as L
    from oracle import oracle as O
    from tests.modules_util import oracle_from_module, to_f32_numpy
    m, n, k = 320, 384, 1280
    mod = make_layer(cfg, n, k, True, torch.bfloat16, gpu_device, seed=5)
    x = (torch.randn(m, k, device=gpu_device) * 1.5).to(torch.bfloat16)
    L.clear_activation_cache()
    cached = mod(x)
    ref = O.forward(oracle_from_module(mod), x.float().cpu().numpy(), "bf16")
    if not cfg.get("use_hadamard"):
        assert np.array_equal(to_f32_numpy(cached), ref)
    monkeypatch.setattr(L, "CACHE_WEIGHTS", False)
    mod.__dict__.pop("_sdnq_hip_state", None)
    L.clear_activation_cache()
    y = mod(x)
    st = L._state(mod)
    assert isinstance(st.lut, tuple) and st.mm_weight is None, "the layer did not take the fused 4-bit route"
    assert torch.equal(y.view(torch.int16), cached.view(torch.int16))
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        L.clear_activation_cache()
        mod(x)
        s.synchronize()
        g = torch.cuda.CUDAGraph()
        L.clear_activation_cache()
        with torch.cuda.graph(g, stream=s):
            yg = mod(x)
        L.clear_activation_cache()
        g.replay()
        s.synchronize()
    assert torch.equal(yg.view(torch.int16), cached.view(torch.int16))
    # many rows: the stand-alone re-quantization kernel keeps the problem (sdnq_hip_scaled_mm_w4_supported says no)
    xb = (torch.randn(2304, k, device=gpu_device)).to(torch.bfloat16)
    monkeypatch.setattr(L, "CACHE_WEIGHTS", True)
    mod.__dict__.pop("_sdnq_hip_state", None)
    L.clear_activation_cache()
    want = mod(xb)
    monkeypatch.setattr(L, "CACHE_WEIGHTS", False)
    mod.__dict__.pop("_sdnq_hip_state", None)
    L.clear_activation_cache()
    assert torch.equal(mod(xb).view(torch.int16), want.view(torch.int16))


def test_unsupported_weights_are_refused_by_the_table_builder(gpu_device):
    from sdnq_amd import _lib, linear as L, ops
    mod = make_layer(dict(weights_dtype="uint4", group_size=32), 128, 256, False, torch.bfloat16, gpu_device)  # groups of 32: no table per 64 columns
    with pytest.raises(_lib.SdnqHipError):
        ops.lut4_build(L._state(mod).qw, ops.MM_I8)
    mod8 = make_layer(dict(weights_dtype="int6", group_size=64), 128, 256, False, torch.bfloat16, gpu_device)
    with pytest.raises(_lib.SdnqHipError):
        ops.lut4_build(L._state(mod8).qw, ops.MM_I8)
    lib = _lib.load()
    assert lib.sdnq_hip_scaled_mm_w4_supported(0, 1, 1024, 1280, 1280) == 1
    assert lib.sdnq_hip_scaled_mm_w4_supported(0, 1, 4608, 3072, 3072) == 0   # many rows: the expansion would be repeated 72 times
    assert lib.sdnq_hip_scaled_mm_w4_supported(0, 1, 1024, 1280, 5120) == 0 and lib.sdnq_hip_scaled_mm_w4_supported(0, 1, 1024, 10240, 1280) == 0  # measured slower
    assert lib.sdnq_hip_scaled_mm_w4_supported(0, 1, 16, 1280, 1280) == 0 and lib.sdnq_hip_scaled_mm_w4_supported(1, 1, 1024, 1280, 1280) == 0
    assert lib.sdnq_hip_scaled_mm_w4_supported(0, 0, 1024, 1280, 1280) == 0 and lib.sdnq_hip_scaled_mm_w4_supported(0, 1, 1024, 1280, 1312) == 0
