"""Quantized attention forward: the oracle restatement against fixtures produced by the reference's own
Triton kernel (run through Triton's interpreter, tests/golden/make_golden_attention.py), and the HIP path against both."""
import numpy as np
import pytest

from oracle import oracle as O
from tests.golden_util import AttnCase, attn_case_names


def _quant_agreement(got_q, got_s, ref_q, ref_s, inexact, hadamard=False):
    """int8 codes / scales of Q are bit-exact; K after smooth_k depends on the summation order of the token mean (fp32), so a
    code may move by one step where x/scale sits on a rounding boundary.  With a Hadamard rotation the rotated value itself is
    rounded to the tensor dtype after a differently ordered fp32 sum (FWHT vs matrix product): one dtype ulp on some elements,
    i.e. a few codes move by one step and a row scale can move by one ulp of the dtype (Hadamard tolerance)."""
    if got_q.dtype == np.float16:
        got_q = got_q.view(np.uint16)  # (fixtures hold 16-bit floats as their bit patterns)
    if not inexact and not hadamard:
        assert np.array_equal(got_q, ref_q) and np.array_equal(got_s, ref_s)
        return
    assert np.allclose(got_s, ref_s, rtol=2e-3 if hadamard else 1e-5, atol=0)
    if got_q.dtype == np.uint8:  # e4m3 codes: sign + magnitude, monotonic in the low 7 bits -> a signed step count
        step = lambda c: np.where(c & 0x80, -(c & 0x7f).astype(np.int32), (c & 0x7f).astype(np.int32))  # noqa: E731
        got_q, ref_q = step(got_q), step(ref_q)
    elif got_q.dtype in (np.float16, np.uint16):  # float16 operand values (pv_matmul_dtype="float16"): steps of the 11-bit mantissa
        a, b = got_q.view(np.float16).astype(np.float32), ref_q.view(np.float16).astype(np.float32)
        assert np.all(np.abs(a - b) <= 2.0 ** -10 * np.abs(b) + 1e-30) and np.mean(a != b) < (2e-2 if hadamard else 2e-3)
        return
    diff = np.abs(got_q.astype(np.int32) - ref_q.astype(np.int32))
    assert diff.max() <= 1 and np.mean(diff != 0) < (2e-2 if hadamard else 2e-3)


def _variant(kw):
    """(matmul_dtype, pv_matmul_dtype) of a fixture + the tolerances its output is compared at.  The default configuration keeps the limits of
    rounds 3-5.  fp8 Q.K^T: the scores are sums of exact products, equal up to float32 summation order -> the same limits.  A QUANTIZED P (8 bits per
    (query, 32-key block); 3 mantissa bits for e4m3) turns every exp2 ulp that crosses a rounding boundary into one code step of one
    probability: |step| / p_scale-range = 1/127 (int8) or 2^-4 (e4m3) of ONE of ~KN terms of a row -> the output limits are those of the
    format, stated here: int8 4e-3 max / 1.5e-3 L2, e4m3 2e-2 / 6e-3, float16 as the unquantized path."""
    mm, pv = kw.get("matmul_dtype", "int8"), kw.get("pv_matmul_dtype")
    lim = {None: None, "int8": (4e-3, 1.5e-3), "fp8": (2e-2, 6e-3), "float16": None}[pv]
    return mm, pv, lim


@pytest.mark.parametrize("name", attn_case_names())
def test_oracle_attention_vs_reference_kernel(name):
    c = AttnCase(name)
    kw = c.kwargs
    hg = c.meta.get("hadamard_group", 0)
    mm, pv, vlim = _variant(kw)
    out, inter = O.attention(c.f32("q"), c.f32("k"), c.f32("v"), c.tag, is_causal=kw.get("is_causal", False), scale=kw.get("scale"),
                             smooth_k=kw.get("smooth_k", True), block_n=c.meta["block_n"], want_intermediates=True, hadamard_group=hg,
                             mask=c.mask_array(), matmul_dtype=mm, pv_matmul_dtype=pv)
    _quant_agreement(inter["q_q"], inter["q_scale"], c.raw("q_q"), c.raw("q_scale"), False, hadamard=bool(hg))
    _quant_agreement(inter["k_q"], inter["k_scale"], c.raw("k_q"), c.raw("k_scale"), kw.get("smooth_k", True), hadamard=bool(hg))
    if pv is not None:  # V per token (rotated first under use_hadamard): elementwise, bit-exact without a rotation
        _quant_agreement(inter["v_q"], inter["v_scale"], c.raw("v_q"), c.raw("v_scale"), False, hadamard=bool(hg))
    ref = c.f32("out")
    assert out.shape == ref.shape
    # same arithmetic as the kernel up to exp2 / reduction-order rounding, then one f16 rounding of the output.  bf16_* fixtures hold the
    # reference kernel's float32 output on the bfloat16 path's quantized operands (P and the output unrounded, make_golden_attention.py):
    # the bfloat16 restatement rounds both to 8 bits of mantissa
    lim, lim2 = (1.2e-2, 4e-3) if c.tag == "bf16" else (2e-3, 5e-4)  # (bf16 fixtures hold the float32 output: the restatement rounds it)
    if vlim is not None:
        lim, lim2 = max(lim, vlim[0]), max(lim2, vlim[1])
    err = np.abs(out - ref).max() / np.abs(ref).max()
    assert err <= lim, (name, err)
    assert np.linalg.norm(out - ref) / np.linalg.norm(ref) <= lim2, name


def test_oracle_attention_block_size_only_changes_rounding():
    c = AttnCase("f16_d64_long")
    a = O.attention(c.f32("q"), c.f32("k"), c.f32("v"), c.tag, block_n=32)
    b = O.attention(c.f32("q"), c.f32("k"), c.f32("v"), c.tag, block_n=128)
    assert np.abs(a - b).max() / np.abs(a).max() <= 2e-3


def _is_variant(name):
    kw = AttnCase(name).kwargs
    return kw.get("matmul_dtype", "int8") != "int8" or kw.get("pv_matmul_dtype") is not None


@pytest.mark.gpu
@pytest.mark.parametrize("name", [n for n in attn_case_names() if not _is_variant(n)])
def test_hip_attention_vs_reference_kernel_and_oracle(name, gpu_device):
    import torch
    from sdnq_amd import attention as A
    c = AttnCase(name)
    kw = c.kwargs
    q, k, v = (c.torch_tensor(t, gpu_device) for t in ("q", "k", "v"))
    hg = c.meta.get("hadamard_group", 0)
    qq, qs, kq, ks, vt = A.quantize_attn(q, k, v, smooth_k=kw.get("smooth_k", True), hadamard_group=hg)
    d = q.shape[-1]  # head dims below 64 / 128 are zero-padded by the prepare kernel
    _quant_agreement(qq[..., :d].cpu().numpy(), qs.cpu().numpy(), c.raw("q_q"), c.raw("q_scale"), False, hadamard=bool(hg))
    kn = k.shape[2]
    k_rows, v_rows = A.unpack_k_fragments(kq), A.unpack_v_fragments(vt)  # MFMA-fragment order -> [Z, KH, KNp, D]
    _quant_agreement(k_rows[:, :, :kn, :d].cpu().numpy(), ks[..., :kn].cpu().numpy(), c.raw("k_q"), c.raw("k_scale"), kw.get("smooth_k", True),
                     hadamard=bool(hg))
    assert torch.equal(v_rows[:, :, :kn, :d], v) and not v_rows[:, :, kn:].any() and not k_rows[:, :, kn:].any() and not ks[..., kn:].any()
    assert not qq[..., d:].any() and not k_rows[..., d:].any() and not v_rows[..., d:].any()
    out = A.sdnq_hip_atten(q, k, v, attn_mask=c.torch_tensor("mask", gpu_device) if c.has("mask") else None, **kw)
    assert out.dtype == q.dtype and out.shape == q.shape
    got = out.float().cpu().numpy()
    for ref, what in ((c.f32("out"), "reference kernel"),
                      (O.attention(c.f32("q"), c.f32("k"), c.f32("v"), c.tag, is_causal=kw.get("is_causal", False), scale=kw.get("scale"),
                                   smooth_k=kw.get("smooth_k", True), hadamard_group=hg, mask=c.mask_array()), "oracle")):
        lim, lim2 = (1.2e-2, 4e-3) if c.tag == "bf16" else (3e-3, 1e-3)  # bf16: P and the output carry 8 bits of mantissa
        err = np.abs(got - ref).max() / np.abs(ref).max()
        assert err <= lim, (name, what, err)
        assert np.linalg.norm(got - ref) / np.linalg.norm(ref) <= lim2, (name, what)


@pytest.mark.gpu
@pytest.mark.parametrize("name", [n for n in attn_case_names() if _is_variant(n)])
def test_hip_attention_variants_vs_reference_kernel_and_oracle(name, gpu_device):
    """fp8 Q.K^T and the quantized P.V formats (round 6; sdnq_hip_attn_prepare_ex / sdnq_hip_attn_fwd_ex): the quantized operands against the
    reference's (Q / V codes and scales bit for bit, K up to the summation order of its token mean), the output against the reference kernel's
    and the oracle's at the limits of `_variant`."""
    import torch
    from sdnq_amd import attention as A
    c = AttnCase(name)
    kw = c.kwargs
    mm, pv, vlim = _variant(kw)
    q, k, v = (c.torch_tensor(t, gpu_device) for t in ("q", "k", "v"))
    hg = c.meta.get("hadamard_group", 0)
    qq, qs, kq, ks, vt, vs = A.quantize_attn_ex(q, k, v, smooth_k=kw.get("smooth_k", True), hadamard_group=hg, matmul_dtype=mm, pv_matmul_dtype=pv)
    d, kn = q.shape[-1], k.shape[2]
    as_ref = (lambda t: t.cpu().numpy().view(np.int8)) if mm == "int8" else (lambda t: t.cpu().numpy())
    _quant_agreement(as_ref(qq[..., :d]), qs.cpu().numpy(), c.raw("q_q"), c.raw("q_scale"), False, hadamard=bool(hg))
    k_rows = A.unpack_k_fragments(kq)
    _quant_agreement(as_ref(k_rows[:, :, :kn, :d]), ks[..., :kn].cpu().numpy(), c.raw("k_q"), c.raw("k_scale"), kw.get("smooth_k", True), hadamard=bool(hg))
    assert not k_rows[:, :, kn:].any() and not ks[..., kn:].any()
    if pv is not None:
        v_rows = A.unpack_v8_fragments(vt) if pv in ("int8", "fp8") else A.unpack_v_fragments(vt)
        got_v = v_rows[:, :, :kn, :d].cpu().numpy()
        got_v = got_v.view(np.int8) if pv == "int8" else (got_v.view(np.uint16) if pv == "float16" else got_v)
        _quant_agreement(got_v, vs[..., :kn].cpu().numpy(), c.raw("v_q"), c.raw("v_scale"), False, hadamard=bool(hg))
        assert not v_rows[:, :, kn:].any() and not vs[..., kn:].any()
    else:
        assert vs is None and torch.equal(A.unpack_v_fragments(vt)[:, :, :kn, :d], v)
    out = A.sdnq_hip_atten(q, k, v, attn_mask=c.torch_tensor("mask", gpu_device) if c.has("mask") else None, **kw)
    assert out.dtype == q.dtype and out.shape == q.shape
    got = out.float().cpu().numpy()
    orc = O.attention(c.f32("q"), c.f32("k"), c.f32("v"), c.tag, is_causal=kw.get("is_causal", False), scale=kw.get("scale"),
                      smooth_k=kw.get("smooth_k", True), hadamard_group=hg, mask=c.mask_array(), matmul_dtype=mm, pv_matmul_dtype=pv)
    for ref, what in ((c.f32("out"), "reference kernel"), (orc, "oracle")):
        lim, lim2 = (1.2e-2, 4e-3) if c.tag == "bf16" else (3e-3, 1e-3)
        if vlim is not None:
            lim, lim2 = max(lim, vlim[0]), max(lim2, vlim[1])
        err = np.abs(got - ref).max() / np.abs(ref).max()
        assert err <= lim, (name, what, err)
        assert np.linalg.norm(got - ref) / np.linalg.norm(ref) <= lim2, (name, what)
    # float32 output and a token-major query view (what an attention processor passes)
    out32 = A.sdnq_hip_atten(q.transpose(1, 2).contiguous().transpose(1, 2), k, v, attn_mask=c.torch_tensor("mask", gpu_device) if c.has("mask") else None,
                             out_dtype=torch.float32, **kw)
    assert out32.dtype == torch.float32
    lim = max(1.2e-2 if c.tag == "bf16" else 3e-3, vlim[0] if vlim else 0)
    assert np.abs(out32.cpu().numpy() - got).max() / np.abs(got).max() <= lim



@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["bf16", "f16"])
@pytest.mark.parametrize("shape", [(1, 3, 3, 77, 77, 64, False), (2, 4, 2, 200, 333, 128, False), (1, 2, 2, 130, 130, 64, True)])
def test_hip_attention_vs_oracle_random(dtype, shape, gpu_device):
    import torch
    from sdnq_amd import attention as A
    z, qh, kh, qn, kn, d, causal = shape
    tdt = {"bf16": torch.bfloat16, "f16": torch.float16}[dtype]
    g = torch.Generator().manual_seed(qn * 7 + d)
    q = torch.randn(z, qh, qn, d, generator=g).to(tdt)
    k = (torch.randn(z, kh, kn, d, generator=g) + 2.0 * torch.randn(1, kh, 1, d, generator=g)).to(tdt)
    v = torch.randn(z, kh, kn, d, generator=g).to(tdt)
    out = A.sdnq_hip_atten(q.to(gpu_device), k.to(gpu_device), v.to(gpu_device), is_causal=causal)
    ref = O.attention(q.float().numpy(), k.float().numpy(), v.float().numpy(), dtype, is_causal=causal)
    got = out.float().cpu().numpy()
    lim = 3e-3 if dtype == "f16" else 1.2e-2  # P and the output are rounded to the value dtype
    assert np.abs(got - ref).max() / np.abs(ref).max() <= lim
    assert np.linalg.norm(got - ref) / np.linalg.norm(ref) <= (1e-3 if dtype == "f16" else 4e-3)
    # and the quantized result stays close to exact fp32 attention (what the int8 path is an approximation of)
    exact = torch.nn.functional.scaled_dot_product_attention(q.float(), k.float().repeat_interleave(qh // kh, 1),
                                                             v.float().repeat_interleave(qh // kh, 1), is_causal=causal).numpy()
    assert np.linalg.norm(got - exact) / np.linalg.norm(exact) <= 5e-2


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["bool", "bf16", "f32"])
def test_hip_attention_masks_vs_oracle(kind, gpu_device):
    """Attention masks on the split-key path (2 waves per query tile) and with broadcast dimensions, bf16, against the oracle;
    queries with no visible key return 0 like the reference."""
    import torch
    from sdnq_amd import attention as A
    z, h, qn, kn, d = 1, 18, 2048, 2100, 64  # 1152 query tiles and kv_len >= 2048: the launcher picks the split-key variant
    g = torch.Generator().manual_seed(11)
    q = torch.randn(z, h, qn, d, generator=g).bfloat16()
    k = torch.randn(z, h, kn, d, generator=g).bfloat16()
    v = torch.randn(z, h, kn, d, generator=g).bfloat16()
    if kind == "bool":
        mask = torch.rand(1, 1, qn, kn, generator=g) > 0.5
        mask[..., 5, :] = False
        mask[..., :, 64:1100] = False  # whole key blocks masked, including everything one key half sees for some tiles
    else:
        mask = torch.randn(h, 1, kn, generator=g) * 3.0
        mask[torch.rand(h, 1, kn, generator=g) < 0.3] = float("-inf")
        mask[..., 7] = 0.0
        mask = mask.to(torch.bfloat16 if kind == "bf16" else torch.float32)
    out = A.sdnq_hip_atten(q.to(gpu_device), k.to(gpu_device), v.to(gpu_device), attn_mask=mask.to(gpu_device)).float().cpu().numpy()
    ref = O.attention(q.float().numpy(), k.float().numpy(), v.float().numpy(), "bf16",
                      mask=mask.numpy() if kind == "bool" else mask.float().numpy())
    assert np.isfinite(out).all()
    if kind == "bool":
        assert not out[:, :, 5].any()
    assert np.abs(out - ref).max() / np.abs(ref).max() <= 1.2e-2
    assert np.linalg.norm(out - ref) / np.linalg.norm(ref) <= 4e-3


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["bf16", "f16"])
def test_hip_attention_query_quantized_in_the_forward_kernel_is_bit_identical(dtype, gpu_device):
    """sdnq_hip_attn_fwd_q16 (the forward kernel quantizes its 32 queries per wave) against sdnq_hip_attn_prepare's Q pass followed by
    sdnq_hip_attn_fwd: same codes and scales, so EQUAL outputs -- on every launch variant (whole / 2 / 4 key parts, shared K / V at
    head_dim 128, causal, masked), padded head dims, strided query views, all-zero query rows and a ragged last query tile."""
    import torch
    from sdnq_amd import attention as A
    tdt = {"bf16": torch.bfloat16, "f16": torch.float16}[dtype]
    g = torch.Generator().manual_seed(5)
    #        z, qh, kh, qn,   kn,   d,   causal, mask
    cases = [(1, 3, 3, 77, 77, 64, False, False), (1, 10, 10, 1024, 77, 64, False, False),   # cross-attention: whole
             (1, 20, 20, 1024, 1024, 64, False, False),                                       # 640 tiles: 4 key parts
             (1, 18, 18, 2048, 2100, 64, False, True),                                        # 2 parts, masked
             (1, 4, 2, 2100, 2100, 128, False, False),                                        # shared K / V through LDS, grouped heads
             (2, 4, 2, 200, 333, 128, True, False), (1, 2, 1, 130, 130, 40, False, False), (1, 5, 5, 1, 500, 72, False, False),
             # at most 128 keys: sdnq_hip_attn is ONE launch (K / V quantized and laid out in LDS by every workgroup)
             (2, 4, 2, 200, 100, 128, True, False), (1, 2, 1, 130, 128, 40, False, True), (1, 3, 3, 50, 33, 64, False, False),
             (1, 6, 3, 300, 1, 64, False, False), (1, 2, 2, 64, 96, 128, False, True), (1, 20, 20, 1024, 77, 64, False, False)]
    for (z, qh, kh, qn, kn, d, causal, masked) in cases:
        q = torch.randn(z, qh, qn, d, generator=g).to(tdt)
        q[:, :, qn // 2] = 0  # a row whose scale is 0 (0 / 0 -> code 0)
        k = (torch.randn(z, kh, kn, d, generator=g) + torch.randn(1, kh, 1, d, generator=g)).to(tdt)
        v = torch.randn(z, kh, kn, d, generator=g).to(tdt)
        mask = (torch.rand(1, 1, qn, kn, generator=g) > 0.3).to(gpu_device) if masked else None
        q, k, v = q.to(gpu_device), k.to(gpu_device), v.to(gpu_device)
        views = [q]
        if d % 8 == 0 and qh > 1:  # the [Z, N, H*D] projection output seen as [Z, H, N, D]
            views.append(q.transpose(1, 2).contiguous().transpose(1, 2))
        for qv in views:
            qq, qs, kq, ks, vt = A.quantize_attn(qv, k, v)
            m = A.prepare_mask(mask, qn, kn) if masked else None
            two_pass = A.atten_fwd(qq, qs, kq, ks, vt, kn, d ** -0.5, causal, tdt, m, head_dim=d)
            q16, none, kq2, ks2, vt2 = A.quantize_attn(qv, k, v, with_query=False)
            assert none is None and torch.equal(kq2, kq) and torch.equal(ks2, ks) and torch.equal(vt2, vt)
            fused = A.atten_fwd(q16, None, kq2, ks2, vt2, kn, d ** -0.5, causal, tdt, m, head_dim=d)
            assert torch.equal(fused, two_pass), (z, qh, kh, qn, kn, d, causal, masked, tuple(qv.stride()))
            assert torch.equal(A.sdnq_hip_atten(qv, k, v, attn_mask=mask, is_causal=causal).contiguous(), two_pass.contiguous())
            if kn <= 128:  # and without K smoothing
                parts = A.quantize_attn(qv, k, v, smooth_k=False)
                plain = A.atten_fwd(*parts, kn, d ** -0.5, causal, tdt, m, head_dim=d)
                assert torch.equal(A.sdnq_hip_atten(qv, k, v, attn_mask=mask, is_causal=causal, smooth_k=False).contiguous(), plain.contiguous())
    with pytest.raises(ValueError, match="Hadamard"):
        A.quantize_attn(q, k, v, hadamard_group=8, with_query=False)


@pytest.mark.gpu
def test_hip_attention_ragged_shape_sweep(gpu_device):
    """Edge shapes: single query / single key, lengths one below / above the 32-wide blocks, causal with q_len != kv_len, grouped
    heads down to one KV head, f32 output -- every combination against the oracle."""
    import itertools
    import torch
    from sdnq_amd import attention as A
    g = torch.Generator().manual_seed(77)
    n = 0
    for qn, kn, d, causal, (qh, kh) in itertools.product((1, 31, 33, 100), (1, 5, 32, 63, 65, 129), (64, 128), (False, True), ((4, 1), (3, 3))):
        q = torch.randn(1, qh, qn, d, generator=g).half()
        k = (torch.randn(1, kh, kn, d, generator=g) + 1.0).half()
        v = torch.randn(1, kh, kn, d, generator=g).half()
        out = A.sdnq_hip_atten(q.to(gpu_device), k.to(gpu_device), v.to(gpu_device), is_causal=causal, out_dtype=torch.float32)
        assert out.dtype == torch.float32
        ref = O.attention(q.float().numpy(), k.float().numpy(), v.float().numpy(), "f16", is_causal=causal, out_tag="f32")
        got = out.cpu().numpy()
        assert np.isfinite(got).all(), (qn, kn, d, causal, qh, kh)
        assert np.abs(got - ref).max() <= 3e-3 * max(np.abs(ref).max(), 1e-3), (qn, kn, d, causal, qh, kh)
        n += 1
    assert n == 192


@pytest.mark.gpu
def test_hip_attention_strided_views_match_contiguous(gpu_device):
    """q / k / v as the transposed views of [Z, N, H*D] projection outputs (what attention processors pass) are read in place and
    give bit-identical results to their contiguous copies; the output then has the query's memory layout."""
    import torch
    from sdnq_amd import attention as A
    g = torch.Generator().manual_seed(3)
    z, h, kh, qn, kn, d = 2, 6, 3, 150, 333, 64
    q = torch.randn(z, qn, h * d, generator=g).bfloat16().to(gpu_device).view(z, qn, h, d).transpose(1, 2)
    kv = torch.randn(z, kn, 2 * kh * d, generator=g).bfloat16().to(gpu_device)  # a fused to_kv output: k and v interleaved per token
    k = kv[..., : kh * d].view(z, kn, kh, d).transpose(1, 2)
    v = kv[..., kh * d:].view(z, kn, kh, d).transpose(1, 2)
    assert not q.is_contiguous() and not k.is_contiguous()
    out = A.sdnq_hip_atten(q, k, v, is_causal=False)
    ref = A.sdnq_hip_atten(q.contiguous(), k.contiguous(), v.contiguous(), is_causal=False)
    assert torch.equal(out, ref)
    assert out.shape == (z, h, qn, d) and out.transpose(1, 2).is_contiguous() and ref.is_contiguous()


@pytest.mark.gpu
def test_hip_attention_full_size_properties(gpu_device):
    """SDXL self-attention size (4096 tokens, 10 heads of 64): rows of P sum to one, so attention over constant V returns
    that constant; and the output is invariant to a permutation of the key/value tokens."""
    import torch
    from sdnq_amd import attention as A
    g = torch.Generator().manual_seed(5)
    q = torch.randn(1, 10, 4096, 64, generator=g).bfloat16().to(gpu_device)
    k = torch.randn(1, 10, 4096, 64, generator=g).bfloat16().to(gpu_device)
    v = torch.randn(1, 10, 4096, 64, generator=g).bfloat16().to(gpu_device)
    const = torch.full_like(v, 0.75)
    out = A.sdnq_hip_atten(q, k, const).float()
    assert (out - 0.75).abs().max() <= 0.75 * 2 ** -7
    perm = torch.randperm(4096, generator=g).to(gpu_device)
    a = A.sdnq_hip_atten(q, k, v).float()
    b = A.sdnq_hip_atten(q, k[:, :, perm], v[:, :, perm]).float()
    assert (a - b).norm() / a.norm() <= 4e-3
    exact = torch.nn.functional.scaled_dot_product_attention(q.float(), k.float(), v.float())
    assert (a - exact).norm() / exact.norm() <= 5e-2


@pytest.mark.gpu
def test_hip_attention_beyond_2_31_elements(gpu_device):
    """Maximum sizes: q / k / v with more than 2^31 elements each (4 x 32 heads x 131 080 tokens x 128) and a key tail: every index
    must be 64-bit.  Sampled query tiles of the first and the last (batch, head) against exact fp32 attention of that head."""
    import torch
    from sdnq_amd import attention as A
    z, h, n, d = 4, 32, 131080, 128
    g = torch.Generator(device=gpu_device).manual_seed(1)
    q = torch.randn(z, h, n, d, device=gpu_device, dtype=torch.bfloat16, generator=g)
    k = torch.randn(z, h, n, d, device=gpu_device, dtype=torch.bfloat16, generator=g)
    v = torch.randn(z, h, n, d, device=gpu_device, dtype=torch.bfloat16, generator=g)
    assert q.numel() > 2 ** 31
    out = A.sdnq_hip_atten(q, k, v)
    for (zi, hi) in ((0, 0), (z - 1, h - 1)):
        for q0 in (0, 65536, n - 40):
            qs = q[zi, hi, q0:q0 + 40].float()
            p = torch.softmax(qs @ k[zi, hi].float().t() * d ** -0.5, dim=-1)
            exact = p @ v[zi, hi].float()
            got = out[zi, hi, q0:q0 + 40].float()
            assert torch.isfinite(got).all()
            assert (got - exact).norm() / exact.norm() <= 6e-2, (zi, hi, q0)


def test_attention_rejects_unbuilt_options():
    import torch
    from sdnq_amd import attention as A
    q = torch.zeros(1, 1, 32, 64, dtype=torch.bfloat16)
    for kw in (dict(use_fp16_accum=True), dict(return_backward=True), dict(matmul_dtype="float16"), dict(matmul_dtype="int4"),
               dict(pv_matmul_dtype="float8_e5m2"), dict(do_quantize=False), dict(matmul_dtype="none")):
        with pytest.raises(NotImplementedError):
            A.sdnq_hip_atten(q, q, q, **kw)
    from sdnq_amd._lib import SdnqHipError
    for kw in ({}, dict(matmul_dtype="float8_e4m3fn"), dict(pv_matmul_dtype="int8"), dict(matmul_dtype="fp8", pv_matmul_dtype="fp8")):
        with pytest.raises(SdnqHipError):
            A.sdnq_hip_atten(q, q, q, **kw)  # built formats, CPU tensors: no fallback
    # the reference's spellings of the formats (triton_atten.py:452-455)
    assert A._mm_name("auto") == A._mm_name("enabled") == A._mm_name("uint8") == "int8" and A._mm_name("float8_e4m3fn") == "fp8"
    assert A._mm_name("auto", pv=True) is None and A._mm_name(None, pv=True) is None and A._mm_name("uint8", pv=True) == "int8"
