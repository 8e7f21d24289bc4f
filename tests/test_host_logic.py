"""Host-side logic that needs no GPU: workload shape lists and the activation-quantization cache."""
import collections

import torch

from sdnq_amd import shapes as S
from sdnq_amd.linear import _ActivationCache


def _agg(seq):
    return dict(collections.Counter((e[0],) + tuple(e[1:5]) for e in seq))


def test_sdxl_sequence_matches_aggregate():
    seq = S.sdxl_unet_layer_sequence()
    assert len(seq) == 741
    assert S.ops_of_sequence(seq) == S.ops_of(S.sdxl_unet_linears()) == 4355400515520
    # 3-way shared self-attention input per layer, one text tensor for all 140 cross-attention k/v projections
    by_key = collections.Counter(e[5] for e in seq)
    assert by_key["text"] == 140
    assert sum(1 for k, c in by_key.items() if k.endswith(".h1") and c == 3) == 70
    for e in seq:  # every consumer of one tensor sees the same (M, K)
        assert {(x[1], x[2]) for x in seq if x[5] == e[5]} == {(e[1], e[2])}


def test_flux_sequence_matches_aggregate():
    seq = S.flux_dev_layer_sequence()
    want = {(e[0],) + tuple(e[1:5]): e[5] for e in S.flux_dev_linears()}
    assert _agg(seq) == want
    assert S.ops_of_sequence(seq) == S.ops_of(S.flux_dev_linears())


def test_activation_cache_identity_version_and_lru():
    c = _ActivationCache(2)
    a, b, d = torch.zeros(4, 8), torch.zeros(4, 8), torch.zeros(4, 8)
    p = (0, 0, False, False, False)
    assert c.get(a, p) is None
    c.put(a, p, "qa")
    assert c.get(a, p) == "qa"
    assert c.get(b, p) is None  # equal values, different tensor object
    assert c.get(a, (1,) + p[1:]) is None  # different quantization parameters
    a.add_(1)  # in-place update bumps _version -> stale
    assert c.get(a, p) is None
    c.put(a, p, "qa2"), c.put(b, p, "qb")
    assert c.get(a, p) == "qa2"  # refreshes a
    c.put(d, p, "qd")  # evicts b (least recently used)
    assert c.get(b, p) is None and c.get(a, p) == "qa2" and c.get(d, p) == "qd"
    c.clear()
    assert c.get(a, p) is None
    # key = storage address + view geometry + version; invalidate() drops everything that aliases a tensor's storage
    base = torch.zeros(8, 8)
    v1, v2 = base[:4], base[4:]
    c.put(v1, p, "v1"), c.put(v2, p, "v2")
    assert c.get(v1, p) == "v1" and c.get(v2, p) == "v2"
    c.invalidate(base)
    assert c.get(v1, p) is None and c.get(v2, p) is None
    # byte bound: an entry that pins more than the limit evicts the older ones (the newest always stays)
    small = _ActivationCache(8, max_bytes=1024)
    big = torch.zeros(1024)
    small.put(a, p, ("x",)), small.put(big, p, ("y",))
    assert small.get(a, p) is None and small.get(big, p) == ("y",)
    # inference tensors carry no version counter: never cached, never an error
    with torch.inference_mode():
        t = torch.zeros(4, 8)
        c.put(t, p, "no")
        assert c.get(t, p) is None


def test_fuse_projections_layout_on_cpu():
    """to_qkv / to_kv carry the concatenated matmul-layout bytes, per-channel scales and biases of their parts."""
    import sdnq_amd

    class Attn(torch.nn.Module):
        def __init__(self, qd, kd, inner, bias):
            super().__init__()
            self.to_q = torch.nn.Linear(qd, inner, bias=bias)
            self.to_k = torch.nn.Linear(kd, inner, bias=bias)
            self.to_v = torch.nn.Linear(kd, inner, bias=bias)

    torch.manual_seed(0)
    model = torch.nn.ModuleDict({"a": Attn(64, 64, 96, True), "b": Attn(64, 128, 96, False), "c": Attn(64, 64, 96, True)})
    model, _ = sdnq_amd.apply_sdnq_to_module(model, sdnq_amd.SDNQConfig(weights_dtype="int8", group_size=-1, use_quantized_matmul=True,
                                                                        modules_to_not_use_matmul=[".c.to_v"], minimum_allowed_numel=1024,
                                                                        minimum_allowed_channel_size=32))
    assert sdnq_amd.fuse_projections(model) == 2  # block c mixes a matmul and a non-matmul layer: left alone
    a, b, c = model["a"], model["b"], model["c"]
    assert not hasattr(c, "to_qkv") and not getattr(c, "fused_projections", False)
    f = a.to_qkv
    assert tuple(f.weight.shape) == (64, 288) and f.weight.stride() == (1, 64) and tuple(f.scale.shape) == (1, 288)
    assert f.sdnq_dequantizer.out_features == 288 and f.sdnq_dequantizer.in_features == 64 and tuple(f.bias.shape) == (288,)
    for i, part in enumerate((a.to_q, a.to_k, a.to_v)):
        assert torch.equal(f.weight[:, 96 * i:96 * (i + 1)], part.weight) and torch.equal(f.scale[:, 96 * i:96 * (i + 1)], part.scale)
        assert torch.equal(f.bias[96 * i:96 * (i + 1)], part.bias)
    assert tuple(b.to_kv.weight.shape) == (128, 192) and b.to_kv.bias is None


def test_link_projections_wiring():
    """loader.link_projections: which layers of an attention block become one ProjectionGroup (host logic only, no kernels)."""
    import torch
    import sdnq_amd

    class Attn(torch.nn.Module):
        def __init__(self, c, cross):
            super().__init__()
            self.to_q = torch.nn.Linear(c, c, bias=False)
            self.to_k = torch.nn.Linear(cross or c, c, bias=False)
            self.to_v = torch.nn.Linear(cross or c, c, bias=False)

    def quantized(blk, **cfg):
        for name in ("to_q", "to_k", "to_v"):
            setattr(blk, name, sdnq_amd.sdnq_quantize_layer(getattr(blk, name).to(torch.bfloat16), sdnq_amd.SDNQConfig(**cfg))[0])
        return blk

    row = dict(weights_dtype="int8", group_size=-1, use_quantized_matmul=True)
    blk = quantized(Attn(64, 0), **row)
    assert sdnq_amd.link_projections(blk) == 1
    g = blk.to_q.__dict__["_sdnq_group"][0]
    assert g is blk.to_k.__dict__["_sdnq_group"][0] is blk.to_v.__dict__["_sdnq_group"][0] and len(g.mods) == 3
    cross = quantized(Attn(64, 96), **row)
    assert sdnq_amd.link_projections(cross) == 1
    assert "_sdnq_group" not in cross.to_q.__dict__ and len(cross.to_k.__dict__["_sdnq_group"][0].mods) == 2
    # group-wise sub-byte weights with a quantized matmul are re-quantized per layer: not linked
    assert sdnq_amd.link_projections(quantized(Attn(64, 0), weights_dtype="uint4", use_quantized_matmul=True)) == 0
    # dequantize-mode layers (the reference's default) link in float mode, whatever the weight format
    deq = quantized(Attn(64, 0), weights_dtype="uint4", use_quantized_matmul=False)
    assert sdnq_amd.link_projections(deq) == 1 and deq.to_q.__dict__["_sdnq_group"][0].float_mode
    assert not blk.to_q.__dict__["_sdnq_group"][0].float_mode
    # members whose parameters are not all resident on one GPU (group / sequential offload, multi-device device_map: here simply
    # the CPU): the group steps aside instead of raising -- the members then run alone, as the reference's layers do
    from sdnq_amd import ops
    assert g._operands(ops.MM_I8) is False and g.gemm is None
    # accelerate() re-links from scratch and honours the switch
    from sdnq_amd import linear as L
    old = L.LINK_PROJECTIONS
    try:
        L.LINK_PROJECTIONS = False
        sdnq_amd.accelerate(blk)
        assert "_sdnq_group" not in blk.to_q.__dict__
        L.LINK_PROJECTIONS = True
        sdnq_amd.accelerate(blk)
        assert len(blk.to_q.__dict__["_sdnq_group"][0].mods) == 3
    finally:
        L.LINK_PROJECTIONS = old


def test_activation_cache_teaches_producers_whether_their_entry_was_used():
    """_ActivationCache._retire: an entry that leaves the cache unused bumps its producer's `_sdnq_unshared`; one that was hit resets it
    for good (CPU: the bookkeeping only)."""
    import torch
    from sdnq_amd.linear import _ActivationCache

    class M:
        pass

    a, b = M(), M()
    c = _ActivationCache(size=2, max_bytes=1 << 30)
    t1, t2, t3 = torch.zeros(4), torch.zeros(4), torch.zeros(4)
    c.put(t1, "p", ("r1",), producer=a)
    c.put(t2, "p", ("r2",), producer=b)
    assert c.get(t2, "p") == ("r2",)
    c.put(t3, "p", ("r3",), producer=a)       # evicts t1's entry, never hit
    assert a.__dict__["_sdnq_unshared"] == 1 and "_sdnq_unshared" not in b.__dict__
    c.clear()                                  # t2's entry was hit, t3's was not
    assert b.__dict__["_sdnq_unshared"] < 0 and a.__dict__["_sdnq_unshared"] == 2
    c.put(t1, "p", ("r1",), producer=b)
    c.invalidate(t1)
    assert b.__dict__["_sdnq_unshared"] < 0   # stays negative: one use by another layer settles it


def test_gpu_skip_allow_list_is_short_and_literal():
    """tests/conftest.py turns any skip of a -m gpu test on a GPU box into a failure unless its reason is allow-listed."""
    from tests import conftest as C
    assert C.skip_is_allowed("uint1: host-side packer only")
    assert C.skip_is_allowed("needs two GPUs (RCCL over xGMI); the gloo test covers the plumbing")
    assert C.skip_is_allowed("inductor backend unavailable here: RuntimeError")
    assert not C.skip_is_allowed("configuration does not re-quantize")
    assert not C.skip_is_allowed("large shapes once, in bf16")
    assert len(C.ALLOWED_GPU_SKIPS) <= 4


def test_gather_pipeline_never_leaves_a_chunk_below_the_small_batch_threshold():
    from sdnq_amd.parallel import chunk_rows
    for m in (256, 257, 1000, 2305, 4096, 4608, 4609, 33, 64, 65):
        for chunks in (1, 2, 3, 4, 8, 9, 16, 72):
            r = chunk_rows(m, chunks)
            assert r[0][0] == 0 and r[-1][1] == m and all(a[1] == b[0] for a, b in zip(r, r[1:]))
            assert all(b - a >= min(32, m) for a, b in r), (m, chunks, r)
            assert all(a % 32 == 0 for a, _ in r)
    assert chunk_rows(2305, 9) == [(0, 288), (288, 576), (576, 864), (864, 1152), (1152, 1440), (1440, 1728), (1728, 2016), (2016, 2305)]


def test_accelerate_never_breaks_a_working_model():
    """accelerate() decides support per module BEFORE re-pointing (support.unsupported_reason): a layer in a configuration the HIP
    path does not build keeps the forward it came with -- here a stand-in for the reference's working forward -- and is listed in
    ONE warning and in the result's `.skipped`; supported layers are re-pointed.  The forwards use the same predicate."""
    import warnings
    import pytest
    import sdnq_amd
    from sdnq_amd import linear, support

    def ref_forward(self, x):  # what a reference-built module would carry: a working forward of its own
        return torch.nn.functional.linear(x, torch.zeros(self.sdnq_dequantizer.out_features, x.shape[-1], dtype=x.dtype), None) + 7

    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(64, 64), torch.nn.Linear(64, 64), torch.nn.Conv2d(64, 64, 3, padding=1, groups=2)).to(torch.bfloat16)
    model[0] = sdnq_amd.sdnq_quantize_layer(model[0], sdnq_amd.SDNQConfig(weights_dtype="int8", use_quantized_matmul=True))[0]
    model[1] = sdnq_amd.sdnq_quantize_layer(model[1], sdnq_amd.SDNQConfig(weights_dtype="int8", use_quantized_matmul=True))[0]
    model[2] = sdnq_amd.sdnq_quantize_layer(model[2], sdnq_amd.SDNQConfig(weights_dtype="int8", quant_conv=True, use_quantized_matmul_conv=True,
                                                                          use_svd=True, svd_rank=8))[0]
    # layer 1: the reference's 16-bit float matmul (linear_fp16.py) with 16-bit scales -- not built here (with float32 scales it is, since
    # round 6); layer 2: a grouped conv with SVD factors -- not built
    model[1].sdnq_dequantizer.quantized_matmul_dtype = "float16"
    model[1].scale = torch.nn.Parameter(model[1].scale.data.to(torch.bfloat16), requires_grad=False)
    for i in (0, 1, 2):
        model[i].forward_func = ref_forward
    assert support.unsupported_reason(model[0]) is None
    assert "float16" in support.unsupported_reason(model[1]) and "SVD" in support.unsupported_reason(model[2])
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        res = sdnq_amd.accelerate(model)
    assert len(w) == 1 and "keep the forward they came with" in str(w[0].message)
    n, skipped = res
    assert res == 1 and n == 1 and res.accelerated == 1 and [s[0] for s in skipped] == ["1", "2"]
    assert model[0].forward_func is not ref_forward and model[1].forward_func is ref_forward and model[2].forward_func is ref_forward
    # the skipped layer still computes (its own forward); before this round accelerate() re-pointed it at a forward that raises
    y = model[1](torch.ones(3, 64, dtype=torch.bfloat16))
    assert y.shape == (3, 64) and float(y[0, 0]) == 7.0
    # a layer that reaches a HIP forward in an unbuilt configuration fails loudly, with the predicate's sentence
    model[1].forward_func = linear.quantized_linear_forward_int8_matmul
    with pytest.raises(NotImplementedError, match="float16"):
        linear._state(model[1])
    # nothing to warn about on a fully supported model
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        assert sdnq_amd.accelerate(torch.nn.Sequential(model[0])) == 1
    assert not w


def test_accelerate_reads_foreign_records_of_grouped_convs_and_options_leave_foreign_forwards_alone():
    """Advisor (round 4): the predicate runs on the dequantizer a model CAME with -- a foreign dataclass without this package's
    in_features / out_features properties -- so the grouped-conv branch must read original_shape; and apply_sdnq_options_to_model
    must not re-point a layer accelerate() left on a foreign forward because its configuration is not built."""
    import dataclasses
    import types
    import warnings
    import sdnq_amd
    from sdnq_amd import support

    def ref_forward(self, x):
        return x

    torch.manual_seed(0)
    conv = torch.nn.Conv2d(64, 64, 3, padding=1, groups=2).to(torch.bfloat16)
    q = sdnq_amd.sdnq_quantize_layer(conv, sdnq_amd.SDNQConfig(weights_dtype="int8", quant_conv=True, use_quantized_matmul_conv=True))[0]
    foreign = types.SimpleNamespace(**{f.name: getattr(q.sdnq_dequantizer, f.name) for f in dataclasses.fields(q.sdnq_dequantizer)})
    assert not hasattr(foreign, "in_features")
    q.sdnq_dequantizer = foreign
    q.forward_func = ref_forward
    assert support.unsupported_reason(q) is None  # (raised AttributeError before)
    lin = sdnq_amd.sdnq_quantize_layer(torch.nn.Linear(64, 64).to(torch.bfloat16), sdnq_amd.SDNQConfig(weights_dtype="int8", use_quantized_matmul=True))[0]
    lin.sdnq_dequantizer.quantized_matmul_dtype = "float16"  # the 16-bit float matmul on 16-bit scales: not built -> stays on its own forward
    lin.scale = torch.nn.Parameter(lin.scale.data.to(torch.bfloat16), requires_grad=False)
    lin.forward_func = ref_forward
    model = torch.nn.Sequential(q, lin)
    with warnings.catch_warnings(record=True):
        warnings.simplefilter("always")
        res = sdnq_amd.accelerate(model)
    assert res.accelerated == 1 and [n for n, _ in res.skipped] == ["1"]
    assert q.forward_func is not ref_forward and lin.forward_func is ref_forward
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        sdnq_amd.apply_sdnq_options_to_model(model, dtype=torch.float16)
    assert lin.forward_func is ref_forward and lin.sdnq_dequantizer.result_dtype == torch.bfloat16  # untouched
    assert any("left untouched" in str(x.message) for x in w)
    assert q.sdnq_dequantizer.result_dtype == torch.float16  # the accelerated layer took the option
    # a record the predicate cannot read at all is a reason to skip, not an exception out of accelerate()
    broken = sdnq_amd.sdnq_quantize_layer(torch.nn.Linear(64, 64).to(torch.bfloat16), sdnq_amd.SDNQConfig(weights_dtype="int8"))[0]
    broken.sdnq_dequantizer = types.SimpleNamespace(layer_class_name="Linear")
    broken.forward_func = ref_forward
    with warnings.catch_warnings(record=True):
        warnings.simplefilter("always")
        res = sdnq_amd.accelerate(torch.nn.Sequential(broken))
    assert res.accelerated == 0 and len(res.skipped) == 1 and broken.forward_func is ref_forward


def test_support_predicate_on_the_uint8_matmul_with_16_bit_scales():
    """The uint8 matmul of dequantize_fp32=False layers: built for bfloat16 Linear layers (round 4; with SVD factors since round 5),
    conv layers too), named as unsupported -- with the reason -- for float16 scales."""
    import sdnq_amd
    from sdnq_amd import support

    def layer(dtype, conv=False, **kw):
        torch.manual_seed(1)
        base = torch.nn.Conv2d(32, 32, 3, padding=1) if conv else torch.nn.Linear(64, 64)
        cfg = dict(weights_dtype="uint8", group_size=-1, dequantize_fp32=False)
        cfg.update(dict(quant_conv=True, use_quantized_matmul_conv=True) if conv else dict(use_quantized_matmul=True))
        cfg.update(kw)
        return sdnq_amd.sdnq_quantize_layer(base.to(dtype), sdnq_amd.SDNQConfig(**cfg))[0]

    assert support.unsupported_reason(layer(torch.bfloat16)) is None
    assert support.unsupported_reason(layer(torch.bfloat16, weights_dtype="uint4", quantized_matmul_dtype="uint8", group_size=32)) is None
    assert "float16" in support.unsupported_reason(layer(torch.float16))
    assert support.unsupported_reason(layer(torch.bfloat16, conv=True)) is None  # (built in round 5; grouped convs keep a reason)
    assert support.unsupported_reason(layer(torch.bfloat16, use_svd=True, svd_rank=16)) is None


def test_float16_matmul_support_rules():
    """The float16 matmul forward (linear_fp16.py; round 6): built for Linear layers with float32 scales -- stored float codes or weights
    re-quantized to float16 codes, Hadamard rotation, SVD factors; conv layers and 16-bit scales keep a reason."""
    import sdnq_amd
    from sdnq_amd import support
    import pytest
    from sdnq_amd.linear import quantized_linear_forward_fp16_matmul

    def layer(conv=False, **kw):
        torch.manual_seed(2)
        base = torch.nn.Conv2d(32, 32, 3, padding=1) if conv else torch.nn.Linear(64, 64)
        cfg = dict(weights_dtype="fp8", quantized_matmul_dtype="float16", group_size=-1)
        cfg.update(dict(quant_conv=True, use_quantized_matmul_conv=True) if conv else dict(use_quantized_matmul=True))
        cfg.update(kw)
        return sdnq_amd.sdnq_quantize_layer(base.to(torch.bfloat16), sdnq_amd.SDNQConfig(**cfg))[0]

    q = layer()
    assert support.unsupported_reason(q) is None and q.forward_func is quantized_linear_forward_fp16_matmul
    assert support.unsupported_reason(layer(weights_dtype="float6_e3m2fn")) is None
    assert support.unsupported_reason(layer(weights_dtype="int8")) is None
    assert support.unsupported_reason(layer(weights_dtype="int4", group_size=32)) is None
    assert support.unsupported_reason(layer(weights_dtype="uint4", group_size=32, use_hadamard=True, hadamard_group_size=32)) is None
    with pytest.raises(NotImplementedError, match="float16"):  # the conv forwards in float16 are not built: loud at quantize time
        layer(conv=True)
    assert "16-bit scales" in support.unsupported_reason(layer(dequantize_fp32=False))
    assert support.unsupported_reason(layer(use_svd=True, svd_rank=8)) is None


def test_support_predicate_on_grouped_convs_with_16_bit_scales():
    """Grouped conv matmul of dequantize_fp32=False layers (round 5): built for bfloat16 scales without a weight zero point; float16
    scales (the reference rounds acc * input_scale to float16 first) and unsigned weights keep their reason."""
    import sdnq_amd
    from sdnq_amd import support

    def layer(dtype, **kw):
        torch.manual_seed(2)
        cfg = dict(weights_dtype="int8", quant_conv=True, use_quantized_matmul_conv=True, dequantize_fp32=False)
        cfg.update(kw)
        return sdnq_amd.sdnq_quantize_layer(torch.nn.Conv2d(64, 64, 3, padding=1, groups=2).to(dtype), sdnq_amd.SDNQConfig(**cfg))[0]

    assert support.unsupported_reason(layer(torch.bfloat16)) is None
    assert "float16" in support.unsupported_reason(layer(torch.float16))
    assert "zero point" in support.unsupported_reason(layer(torch.bfloat16, weights_dtype="uint8", quantized_matmul_dtype="int8"))


def test_peer_arena_ring_steps_over_live_ranges_and_recycles_dead_ones():
    """PeerArena's allocator (the copy-free gather's receive ring), without a GPU: a range is reused only when no tensor made from it
    is alive; live ranges are stepped over; a ring full of live tensors raises."""
    import pytest
    from sdnq_amd import parallel as P

    class FakeBuf:
        def data_ptr(self):
            return 1 << 20
    a = P.PeerArena.__new__(P.PeerArena)
    a.size, a.CTRL, a.head, a.live, a.buf = 10240, 256, 256, [], FakeBuf()
    off0, keep = a._alloc(2000)
    assert off0 == 256
    seen = set()
    for _ in range(20):
        off, r = a._alloc(3000)
        assert off >= 256 + 2048 and off + 3072 <= 256 + 10240  # never on the live range, never past the end
        seen.add(off)
        del r
    assert len(seen) == 2  # the dead ranges are recycled
    held = [a._alloc(3000)[1] for _ in range(2)]
    with pytest.raises(RuntimeError, match="full of live output tensors"):
        for _ in range(5):
            held.append(a._alloc(3000)[1])
    del held, keep
    assert a._alloc(9000)[0] == 256  # everything died: the whole ring is free again


def test_prefetch_chain_learns_the_launch_order_and_names_two_units_ahead(monkeypatch):
    """linear._PrefetchChain (no GPU: the C call is captured): after one pass over units A -> B -> C -> D the second pass hands
    B + C at A's launch, C + D at B's, ...; a unit whose weights exceed the cap contributes nothing; a changed order is re-learned."""
    import torch
    from sdnq_amd import linear as L
    calls = []

    class FakeLib:
        def sdnq_hip_prefetch_hint(self, *a):
            calls.append(a)
            return 0

    monkeypatch.setattr(L.ops._lib, "load", lambda: FakeLib())
    monkeypatch.setattr(L, "_FP", None)  # the Python form of the chain (the C++ form: test_prefetch_chain_in_the_fast_path_module below)
    chain = L._PrefetchChain()
    ws = [torch.zeros(64 * (i + 1), dtype=torch.int8) for i in range(4)]
    units = [L._LaunchUnit((w,)) for w in ws]
    for u in units:
        chain.launch(u)
    assert calls == []  # first step: nothing known yet
    chain.reset()
    for u in units:
        chain.launch(u)
    rng = lambda i: (ws[i].data_ptr(), ws[i].numel())  # noqa: E731
    assert calls[0][:4] == (*rng(1), *rng(2)) and calls[0][4:] == (0, 0, 0, 0)
    assert calls[1][:4] == (*rng(2), *rng(3))
    assert calls[2][:2] == rng(3) and calls[2][2:] == (0, 0, 0, 0, 0, 0)
    assert len(calls) == 3  # the last unit has no successor
    # another order this step: the links follow it
    calls.clear()
    chain.reset()
    for i in (0, 2, 1, 3):
        chain.launch(units[i])
    chain.reset()
    calls.clear()
    for i in (0, 2, 1, 3):
        chain.launch(units[i])
    assert calls[0][:4] == (*rng(2), *rng(1))
    # a grouped launch of three members names three ranges; over the cap: none
    g = L._LaunchUnit(tuple(ws[:3]))
    assert len(g.ranges) == 3
    monkeypatch.setattr(L, "PREFETCH_NEXT_MAX_BYTES", 100)
    assert L._LaunchUnit((ws[3],)).ranges == ()
    # a dead unit (its module was deleted) ends the chain
    calls.clear()
    a, b = L._LaunchUnit((ws[0],)), L._LaunchUnit((ws[1],))
    monkeypatch.setattr(L, "PREFETCH_NEXT_MAX_BYTES", 1 << 30)
    a, b = L._LaunchUnit((ws[0],)), L._LaunchUnit((ws[1],))
    chain.reset(); chain.launch(a); chain.launch(b)
    del b
    chain.reset(); chain.launch(a)
    assert calls == []


class TinyNet(torch.nn.Module):
    """The skeleton of tests/golden/checkpoint_tiny (a model DEFINITION, like a diffusers class: the checkpoint holds only tensors + json)."""

    def __init__(self, d_in=64, d_hidden=128, d_mid=96, d_out=64, n_cls=10):
        super().__init__()
        self.proj_in = torch.nn.Linear(d_in, d_hidden)
        self.mid = torch.nn.Linear(d_hidden, d_mid, bias=False)
        self.proj_out = torch.nn.Linear(d_mid, d_out)
        self.norm = torch.nn.LayerNorm(d_out)
        self.head = torch.nn.Linear(d_out, n_cls)

    def forward(self, x):
        h = torch.nn.functional.silu(self.proj_in(x))
        h = torch.nn.functional.silu(self.mid(h))
        h = self.norm(self.proj_out(h))
        return self.head(h)


def test_prefetch_chain_in_the_fast_path_module():
    """The same chain as csrc/fastpath.cpp keeps it (units made with the module present link THERE: the plans and linear.py share one
    chain per thread): same hand-overs, read back through _fastpath.last_hint()."""
    import torch
    from sdnq_amd import linear as L
    fp = L._FP
    if fp is None:
        pytest.skip("sdnq_amd._fastpath is not built")
    chain = L._PrefetchChain()
    ws = [torch.zeros(64 * (i + 1), dtype=torch.int8) for i in range(4)]
    units = [L._LaunchUnit((w,)) for w in ws]
    assert all(u.c is not None for u in units)
    rng = lambda i: (ws[i].data_ptr(), ws[i].numel())  # noqa: E731
    chain.reset()
    n0 = fp.last_hint()[0]
    for u in units:
        chain.launch(u)
    assert fp.last_hint()[0] == n0  # first step: nothing known yet
    chain.reset()
    seen = []
    for u in units:
        chain.launch(u)
        seen.append(fp.last_hint())
    assert seen[0][1:5] == (*rng(1), *rng(2)) and seen[0][5:] == (0, 0, 0, 0)
    assert seen[1][1:5] == (*rng(2), *rng(3))
    assert seen[2][1:3] == rng(3) and seen[2][3:] == (0, 0, 0, 0, 0, 0)
    assert seen[3] == seen[2] and seen[3][0] == n0 + 3  # the last unit has no successor: no hand-over
    for _ in range(2):  # another order: re-learned in one step
        chain.reset()
        for i in (0, 2, 1, 3):
            chain.launch(units[i])
            if i == 0:
                first = fp.last_hint()
    assert first[1:5] == (*rng(2), *rng(1))
    # a dead unit ends the chain; a unit on another device is never named
    a, b = L._LaunchUnit((ws[0],)), L._LaunchUnit((ws[1],))
    chain.reset(); chain.launch(a); chain.launch(b)
    del b
    n1 = fp.last_hint()[0]
    chain.reset(); chain.launch(a)
    assert fp.last_hint()[0] == n1
    c0, c1 = fp.Unit((rng(0),), 0), fp.Unit((rng(1),), 1)
    fp.chain_reset(); c0.launch(); c1.launch()
    fp.chain_reset(); c0.launch()
    assert fp.last_hint()[0] == n1


def test_save_sdnq_model_writes_what_the_reference_wrote(tmp_path):
    """sdnq_amd.save_sdnq_model (reference loader.py:46-79): the reference-written checkpoint, loaded here and saved again, is the same
    checkpoint -- every tensor bit for bit under the same key (the direct-matmul weight back in its contiguous logical [K, N] form), the
    same quantization_config.json content -- through both writers: the built-in one and a model's own `save_pretrained`."""
    import json
    import os
    from safetensors.torch import load_file, save_file
    import sdnq_amd
    src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "checkpoint_tiny")
    model = sdnq_amd.load_sdnq_model(src, model_cls=TinyNet, device="cpu")
    before = {k: (v.stride(), v.data_ptr()) for k, v in model.state_dict().items()}
    want = load_file(os.path.join(src, "model.safetensors"))
    want_cfg = json.load(open(os.path.join(src, "quantization_config.json")))

    def check(path):
        got = load_file(os.path.join(path, "model.safetensors"))
        assert sorted(got) == sorted(want)
        for k in want:
            assert got[k].dtype == want[k].dtype and got[k].shape == want[k].shape, k
            assert torch.equal(got[k].view(torch.uint8), want[k].view(torch.uint8)), k
        cfg = json.load(open(os.path.join(path, "quantization_config.json")))
        for k, v in want_cfg.items():
            if k not in ("quantization_device", "return_device", "non_blocking", "add_skip_keys", "modules_to_not_convert"):
                assert cfg[k] == v, (k, cfg[k], v)
        assert set(want_cfg["modules_to_not_convert"]) <= set(cfg["modules_to_not_convert"])

    sdnq_amd.save_sdnq_model(model, str(tmp_path / "a"))
    check(str(tmp_path / "a"))
    # a model with its own save_pretrained (what diffusers' ModelMixin does: the state_dict as it is -- safetensors refuses strided tensors)
    def save_pretrained(path, max_shard_size=None):
        os.makedirs(path, exist_ok=True)
        save_file(dict(model.state_dict()), os.path.join(path, "model.safetensors"))
    model.save_pretrained = save_pretrained
    sdnq_amd.save_sdnq_model(model, str(tmp_path / "b"))
    check(str(tmp_path / "b"))
    # the model itself is untouched: same storage, same strides (the matmul operand stays in its physical [N][K] layout)
    assert {k: (v.stride(), v.data_ptr()) for k, v in model.state_dict().items()} == before
    # ... and what was written loads again
    again = sdnq_amd.load_sdnq_model(str(tmp_path / "a"), model_cls=TinyNet, device="cpu")
    for (k, a), (_, b) in zip(model.state_dict().items(), again.state_dict().items()):
        assert a.stride() == b.stride() and a.dtype == b.dtype and torch.equal(a, b), k


def test_load_sdnq_model_rebuilds_the_layers_of_a_reference_checkpoint():
    """sdnq_amd.load_sdnq_model on the checkpoint the REFERENCE wrote (tests/golden/make_golden_checkpoint.py): every layer's record
    equals what the reference's own loader derived (stored next to the checkpoint), the per-module overrides of the config are
    honoured (modules_dtype_dict, modules_quant_config, modules_to_not_convert), direct-matmul operands sit in the physical
    [N][K] layout, nothing is left on the meta device.  CPU: structure only (the forwards need the GPU)."""
    import json
    import os
    import numpy as np
    import sdnq_amd
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "checkpoint_tiny")
    model = sdnq_amd.load_sdnq_model(path, model_cls=TinyNet, device="cpu")
    meta = json.loads(bytes(np.load(os.path.join(path, "io.npz"))["meta_json"]).decode())
    got = {n: m for n, m in model.named_modules() if hasattr(m, "sdnq_dequantizer")}
    assert sorted(got) == sorted(meta) == ["mid", "proj_in", "proj_out"]
    assert type(model.head) is torch.nn.Linear and model.head.weight.dtype == torch.bfloat16
    for name, want in meta.items():
        dq = got[name].sdnq_dequantizer
        assert dq.weights_dtype == want["weights_dtype"] and dq.group_size == want["group_size"], name
        assert bool(dq.use_quantized_matmul) == want["use_quantized_matmul"] and bool(dq.re_quantize_for_matmul) == want["re_quantize_for_matmul"], name
        assert list(dq.quantized_weight_shape) == want["quantized_weight_shape"], name
        assert (got[name].svd_up is not None) == want["has_svd"], name
        assert got[name].forward_func.__module__.startswith("sdnq_amd")
    assert not any(p.is_meta for p in model.parameters())
    w = model.proj_in.weight  # logical [K, N] = [64, 128] over physical [N][K] bytes
    assert tuple(w.shape) == (64, 128) and w.stride() == (1, 64) and not w.requires_grad
    assert model.proj_out.svd_up.shape == (16, 64) and model.proj_out.svd_up.stride() == (1, 16)
    assert model.quantization_config.modules_dtype_dict == {"uint4": ["mid"]}
    # the same through a skeleton instance, and the missing-tensor error
    with torch.device("meta"):
        skel = TinyNet()
    m2 = sdnq_amd.load_sdnq_model(path, model=skel, device="cpu")
    assert torch.equal(m2.mid.weight, model.mid.weight)

    class Bigger(TinyNet):
        def __init__(self, **kw):
            super().__init__(**kw)
            self.extra = torch.nn.LayerNorm(64)

    import pytest
    with pytest.raises(RuntimeError, match="not in the checkpoint"):
        sdnq_amd.load_sdnq_model(path, model_cls=Bigger, device="cpu")
    # ... and the other direction (the reference loads strictly): a tensor of the checkpoint the rebuilt model has no slot for is an error, not dropped
    import shutil
    import tempfile
    from safetensors.torch import load_file, save_file
    with tempfile.TemporaryDirectory() as tmp:
        dst = os.path.join(tmp, "ckpt")
        shutil.copytree(path, dst)
        tensors = load_file(os.path.join(dst, "model.safetensors"))
        tensors["mid.codebook"] = torch.zeros(16)
        save_file(tensors, os.path.join(dst, "model.safetensors"))
        with pytest.raises(RuntimeError, match="no place in the rebuilt model.*mid.codebook"):
            sdnq_amd.load_sdnq_model(dst, model_cls=TinyNet, device="cpu")


def test_module_lists_match_like_the_reference():
    from sdnq_amd.quantizer import check_param_name_in, _minimum_dtype
    assert check_param_name_in("blocks.0.attn.to_q.weight", ["to_q"]) == "to_q"
    assert check_param_name_in("blocks.0.attn.to_q.weight", ["to_"]) is None            # a component, not a substring
    assert check_param_name_in("blocks.0.attn.to_q.weight", [".blocks.0"]) == ".blocks.0"  # leading dot: a prefix
    assert check_param_name_in("blocks.0.attn.to_q.weight", [".attn"]) is None
    assert check_param_name_in("blocks.0.attn.to_q.weight", ["blocks.*.to_q.weight"]) == "blocks.*.to_q.weight"
    assert check_param_name_in("head.weight", ["head.weight", "x"]) == "head.weight"
    assert _minimum_dtype("uint4", "a.ff.weight", {"minimum_6bit": ["ff"]}) == "int6"
    assert _minimum_dtype("int8", "a.ff.weight", {"minimum_6bit": ["ff"]}) == "int8"
    assert _minimum_dtype("int8", "a.ff.weight", {"uint4": ["ff"]}) == "uint4"
