"""torch.library registration (sdnq_amd/torch_ops.py): schemas, fake implementations, and Dynamo tracing of an SDNQ transformer
block without graph breaks (CPU: trace only; GPU: compiled run equals eager)."""
import pytest
import torch

import sdnq_amd
from sdnq_amd import torch_ops  # noqa: F401


class Block(torch.nn.Module):
    def __init__(self, d=128, heads=4):
        super().__init__()
        self.heads = heads
        self.norm1, self.norm2 = torch.nn.LayerNorm(d), torch.nn.LayerNorm(d)
        self.to_q, self.to_k, self.to_v = (torch.nn.Linear(d, d, bias=False) for _ in range(3))
        self.to_out = torch.nn.Linear(d, d)
        self.ff1, self.ff2 = torch.nn.Linear(d, 4 * d), torch.nn.Linear(4 * d, d)

    def forward(self, x):
        h = self.norm1(x)
        b, t, d = h.shape
        q, k, v = (f(h).view(b, t, self.heads, d // self.heads).transpose(1, 2) for f in (self.to_q, self.to_k, self.to_v))
        a = torch.nn.functional.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(b, t, d)
        x = x + self.to_out(a)
        return x + self.ff2(torch.nn.functional.gelu(self.ff1(self.norm2(x))))


def _quantized_block(device, **cfg):
    torch.manual_seed(0)
    blk = Block().to(torch.bfloat16).to(device)
    blk, _ = sdnq_amd.apply_sdnq_to_module(blk, sdnq_amd.SDNQConfig(minimum_allowed_numel=1024, minimum_allowed_channel_size=32, **cfg))
    return blk


def test_operator_schemas_match_the_reference_seam():
    s = str(torch.ops.sdnq_hip.scaled_mm.default._schema)
    # sdnq::scaled_mm(a, b, scale_a, scale_b, bias=None, out_dtype=float32) -> Tensor  (kernels/triton_scaled_mm.py:239-248)
    assert s.startswith("sdnq_hip::scaled_mm(Tensor a, Tensor b, Tensor scale_a, Tensor scale_b, Tensor? bias=None, ScalarType out_dtype=") and s.endswith("-> Tensor")
    for name in ("rowquant", "linear_w8a8", "dequant", "layer_forward"):
        op = getattr(torch.ops.sdnq_hip, name).default
        assert not op._schema.is_mutable, name  # mutates_args = {} like the reference's op


def test_fake_implementations_give_shapes_and_dtypes():
    from torch._subclasses.fake_tensor import FakeTensorMode
    with FakeTensorMode():
        a, b = torch.empty(48, 64, dtype=torch.int8), torch.empty(64, 32, dtype=torch.int8)
        y = torch.ops.sdnq_hip.scaled_mm(a, b, torch.empty(48, 1), torch.empty(1, 32), None, torch.bfloat16)
        assert y.shape == (48, 32) and y.dtype == torch.bfloat16
        xq, xs = torch.ops.sdnq_hip.rowquant(torch.empty(2, 5, 64, dtype=torch.bfloat16), "fp8", 0)
        assert xq.shape == (10, 64) and xq.dtype == torch.float8_e4m3fn and xs.shape == (10, 1) and xs.dtype == torch.float32
        y = torch.ops.sdnq_hip.linear_w8a8(torch.empty(2, 5, 64, dtype=torch.float16), torch.empty(96, 64, dtype=torch.int8), torch.empty(96), None, "int8", 0)
        assert y.shape == (2, 5, 96) and y.dtype == torch.float16
        w = torch.ops.sdnq_hip.dequant(torch.empty(96 * 32, dtype=torch.uint8), torch.empty(96, 1, 1), None, None, None, "int4", 96, 64, 64, False, False,
                                       0, torch.bfloat16)
        assert w.shape == (96, 64) and w.dtype == torch.bfloat16
        with pytest.raises(RuntimeError):
            torch.ops.sdnq_hip.scaled_mm(a, torch.empty(60, 32, dtype=torch.int8), torch.empty(48, 1), torch.empty(1, 32), None, torch.bfloat16)


def test_dynamo_traces_a_block_without_graph_breaks_cpu():
    """fullgraph tracing on CPU (no kernels run: export only traces with fake tensors).  Plain w8a8 layers trace as
    sdnq_hip::rowquant + sdnq_hip::layer_matmul (so that layers sharing an input can share its row quantization: merge_layer_matmuls), dequantize-mode layers as one sdnq_hip::layer_forward; nothing else of this package is in the graph."""
    blk = _quantized_block("cpu", weights_dtype="int8", group_size=-1, use_quantized_matmul=True)
    assert sum(isinstance(m, sdnq_amd.SDNQLinear) for m in blk.modules()) == 6
    x = torch.randn(2, 40, 128).to(torch.bfloat16)
    gm, _ = torch._dynamo.export(blk)(x)  # raises on any graph break

    def calls_of(gm, op):
        return [n for n in gm.graph.nodes if n.op == "call_function" and n.target in (op, op.default)]

    mm = calls_of(gm, torch.ops.sdnq_hip.layer_matmul)
    assert len(mm) == 6 and len(calls_of(gm, torch.ops.sdnq_hip.rowquant)) == 6 and not calls_of(gm, torch.ops.sdnq_hip.layer_forward)
    handles = {m.__dict__["_sdnq_hip_handle"] for m in blk.modules() if isinstance(m, sdnq_amd.SDNQLinear)}
    assert handles == {n.args[2] for n in mm}
    # to_q / to_k / to_v quantize the SAME graph value: identical (pure) operator calls, which merge_layer_matmuls merges
    rq_inputs = [n.args[0] for n in calls_of(gm, torch.ops.sdnq_hip.rowquant)]
    assert len(set(rq_inputs)) == 4
    deq = _quantized_block("cpu", weights_dtype="uint4", use_quantized_matmul=False)
    gm2, _ = torch._dynamo.export(deq)(x)
    assert len(calls_of(gm2, torch.ops.sdnq_hip.layer_forward)) == 6 and not calls_of(gm2, torch.ops.sdnq_hip.layer_matmul)


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [dict(weights_dtype="int8", group_size=-1, use_quantized_matmul=True),
                                 dict(weights_dtype="int4", use_quantized_matmul=True),
                                 dict(weights_dtype="uint4", use_quantized_matmul=False)], ids=["int8-qmm", "int4-requant", "uint4-dequant"])
def test_compiled_block_equals_eager(cfg, gpu_device):
    """torch.compile(block, fullgraph=True) of a small SDNQ transformer block runs (every quantized Linear one sdnq_hip::layer_forward
    op, linked projections and weight caches at work behind it) and gives the eager result: bit-identical through the aot_eager
    backend (same kernels in the same order), within bf16 rounding of the fused pointwise code through Inductor."""
    blk = _quantized_block(gpu_device, **cfg)
    sdnq_amd.accelerate(blk)
    # plain w8a8 layers trace as rowquant + layer_matmul (the post-grad pass then shares one row quantization among q / k / v);
    # everything else as one layer_forward operator
    plans = {name: m.__dict__.get("_sdnq_hip_plan") for name, m in blk.named_modules() if hasattr(m, "sdnq_dequantizer")}
    assert all((p is not None) == bool(cfg.get("use_quantized_matmul")) for p in plans.values()), plans
    x = torch.randn(2, 77, 128, device=gpu_device, dtype=torch.bfloat16)
    with torch.no_grad():
        want = blk(x)
        got = torch.compile(blk, fullgraph=True, backend="aot_eager")(x)
        assert torch.equal(got, want)
        try:
            ind = torch.compile(blk, fullgraph=True)(x)
        except Exception as e:  # noqa: BLE001  (no Triton code generation available on the box)
            pytest.skip(f"inductor backend unavailable here: {type(e).__name__}")
        assert (ind.float() - want.float()).abs().max() <= 0.05 * want.float().abs().max()
    # the operator seam on device, against the plain call
    a = torch.randint(-128, 128, (64, 128), dtype=torch.int8, device=gpu_device)
    b = torch.randint(-128, 128, (128, 96), dtype=torch.int8, device=gpu_device)
    sa, sb = torch.rand(64, 1, device=gpu_device), torch.rand(1, 96, device=gpu_device)
    assert torch.equal(torch.ops.sdnq_hip.scaled_mm(a, b, sa, sb, None, torch.bfloat16), sdnq_amd.int_scaled_mm_func(a, b, sa, sb, None, torch.bfloat16))


@pytest.mark.gpu
def test_layer_forward_op_never_reuses_by_tensor_identity(gpu_device):
    """Inductor recycles dead buffers in place and writes through raw pointers: the same tensor object, address, geometry and
    version can hold NEW values when the next layer is called (round-2 advisor finding).  The `sdnq_hip::layer_forward` operator
    therefore must not serve a quantized activation (or a parked sibling output) keyed on tensor identity.  Emulated here without
    Inductor: the input of linked q / k / v projections is overwritten in place with its version counter preserved."""
    blk = _quantized_block(gpu_device, weights_dtype="int8", group_size=-1, use_quantized_matmul=True)
    sdnq_amd.accelerate(blk)
    x = torch.randn(1, 64, 128, device=gpu_device, dtype=torch.bfloat16)
    x2 = torch.randn(1, 64, 128, device=gpu_device, dtype=torch.bfloat16)
    with torch.no_grad():
        want_q = blk.to_q(x.clone())
        want_k2 = blk.to_k(x2.clone())
        sdnq_amd.invalidate()
        buf = x.clone()
        q = torch.ops.sdnq_hip.layer_forward(buf, blk.to_q._sdnq_hip_handle)
        with torch.autograd._unsafe_preserve_version_counter(buf):
            buf.copy_(x2)  # "buf7 = buf1  # reuse": new contents, same object / address / geometry / version
        k2 = torch.ops.sdnq_hip.layer_forward(buf, blk.to_k._sdnq_hip_handle)
        assert torch.equal(q, want_q)
        assert torch.equal(k2, want_k2)  # stale with identity-keyed reuse: to_k would be served from x's quantized copy / group outputs


def test_copies_of_a_layer_get_their_own_operator_handle():
    """copy.deepcopy / pickle must not carry the original's operator handle, projection group or kernel-ready tensor cache
    (round-2 advisor finding: a deep-copied model ran the ORIGINAL module's weights under torch.compile)."""
    import copy
    import pickle
    lin = torch.nn.Linear(64, 32).to(torch.bfloat16)
    layer, _ = sdnq_amd.sdnq_quantize_layer(lin, sdnq_amd.SDNQConfig(weights_dtype="int8", group_size=-1, use_quantized_matmul=True))
    layer.__dict__["_sdnq_hip_state"] = object()
    h = layer._sdnq_hip_handle
    for clone in (copy.deepcopy(layer), pickle.loads(pickle.dumps(layer))):
        assert clone._sdnq_hip_handle != h
        assert torch_ops._layer(clone._sdnq_hip_handle) is clone and torch_ops._layer(h) is layer
        assert "_sdnq_hip_state" not in clone.__dict__ and "_sdnq_group" not in clone.__dict__
        assert torch.equal(clone.weight, layer.weight) and clone.weight is not layer.weight
    # a handle copied by hand (clone.__dict__.update) is re-issued on the next registration
    other = copy.copy(layer)
    other.__dict__["_sdnq_hip_handle"] = h
    assert torch_ops.layer_handle(other) != h and torch_ops._layer(h) is layer


def test_merge_pass_links_projections_in_a_traced_graph_cpu():
    """`merge_layer_matmuls` (the Inductor post-grad pass behind enable_compile_grouping): after common-subexpression elimination
    to_q / to_k / to_v are layer_matmul nodes on the SAME (xq, xs) values and become ONE layer_matmul_group node whose flat result
    the three outputs are sliced from; layers with their own input stay as they are.  CPU: graph surgery + fake-tensor shapes only."""
    blk = _quantized_block("cpu", weights_dtype="int8", group_size=-1, use_quantized_matmul=True)
    x = torch.randn(2, 40, 128).to(torch.bfloat16)
    gm, _ = torch._dynamo.export(blk, assume_static_by_default=True)(x)
    for n in gm.graph.nodes:
        if "val" not in n.meta and "example_value" in n.meta:
            n.meta["val"] = n.meta["example_value"]
    g = gm.graph  # six rowquant nodes: the pass itself merges the row quantizations of one value (Inductor's passes do not)

    def count(op):
        return sum(1 for n in g.nodes if n.op == "call_function" and n.target in (op, op.default))

    assert count(torch.ops.sdnq_hip.rowquant) == 6 and count(torch.ops.sdnq_hip.layer_matmul) == 6
    removed = torch_ops.merge_layer_matmuls(g)
    assert removed == 2 and count(torch.ops.sdnq_hip.rowquant) == 4
    assert count(torch.ops.sdnq_hip.layer_matmul) == 3 and count(torch.ops.sdnq_hip.layer_matmul_group) == 1
    grp = next(n for n in g.nodes if n.op == "call_function" and n.target is torch.ops.sdnq_hip.layer_matmul_group.default)
    assert [torch_ops._layer(h) for h in grp.args[2]] == [blk.to_q, blk.to_k, blk.to_v]
    assert grp.meta["val"].shape == (80 * 3 * 128,) and grp.meta["val"].dtype == torch.bfloat16
    views = [u for s in grp.users for u in s.users]
    assert sorted(tuple(v.meta["val"].shape) for v in views) == [(80, 128)] * 3
    g.lint()
    gm.recompile()
    # a second application finds nothing left to merge
    assert torch_ops.merge_layer_matmuls(g) == 0


@pytest.mark.gpu
def test_compiled_block_with_grouping_equals_eager(gpu_device):
    """enable_compile_grouping(): torch.compile (Inductor) of the block runs to_q / to_k / to_v as ONE grouped launch and still equals
    the eager block within the rounding of Inductor's fused pointwise code; the operator alone is bit-identical to the members."""
    blk = _quantized_block(gpu_device, weights_dtype="int8", group_size=-1, use_quantized_matmul=True)
    sdnq_amd.accelerate(blk)
    x = torch.randn(2, 77, 128, device=gpu_device, dtype=torch.bfloat16)
    with torch.no_grad():
        h = blk.norm1(x)
        xq, xs = torch.ops.sdnq_hip.rowquant(h, "int8", 0)
        hs = [m._sdnq_hip_handle for m in (blk.to_q, blk.to_k, blk.to_v)]
        flat = torch.ops.sdnq_hip.layer_matmul_group(xq, xs, hs, torch.bfloat16)
        m = xq.shape[0]
        for i, mod in enumerate((blk.to_q, blk.to_k, blk.to_v)):
            alone = torch.ops.sdnq_hip.layer_matmul(xq, xs, mod._sdnq_hip_handle, torch.bfloat16)
            assert torch.equal(flat[m * 128 * i:m * 128 * (i + 1)].view(m, 128), alone)
        want = blk(x)
        torch_ops.enable_compile_grouping()
        before = dict(torch_ops.merge_stats)
        # a compile served from Dynamo's / Inductor's caches (an earlier test of this process compiled the same block) runs no
        # post-grad pass, and the counter below would not move: start from a clean slate
        torch._dynamo.reset()
        import torch._inductor.config as _icfg
        try:
            with _icfg.patch(fx_graph_cache=False):
                got = torch.compile(blk, fullgraph=True)(x)
        except Exception as e:  # noqa: BLE001  (no Triton code generation available on the box)
            pytest.skip(f"inductor backend unavailable here: {type(e).__name__}")
        assert torch_ops.merge_stats["launches_removed"] - before["launches_removed"] == 2
        assert (got.float() - want.float()).abs().max() <= 0.05 * want.float().abs().max()


def test_options_refresh_the_compile_plan_cpu():
    """apply_sdnq_options_to_model may switch the matmul mode / scale dtype / matmul dtype of a layer; the operator plan that
    SDNQLayer.forward follows under torch.compile must follow (a stale plan makes the compiled model compute the OLD mode)."""
    from sdnq_amd import loader
    blk = _quantized_block(torch.device("cpu"), weights_dtype="int8", group_size=-1, use_quantized_matmul=True)
    mods = [m for m in blk.modules() if hasattr(m, "sdnq_dequantizer")]
    assert mods and all(m.__dict__.get("_sdnq_hip_plan") == ("q", "int8", 0) for m in mods)
    loader.apply_sdnq_options_to_model(blk, use_quantized_matmul=False)
    assert all(m.__dict__.get("_sdnq_hip_plan") is None for m in mods)          # float mode: one layer_forward operator
    loader.apply_sdnq_options_to_model(blk, use_quantized_matmul=True)
    assert all(m.__dict__.get("_sdnq_hip_plan") == ("q", "int8", 0) for m in mods)
    loader.apply_sdnq_options_to_model(blk, dequantize_fp32=False)                # bf16 scales: the _lp forwards, not the fp32 epilogue
    assert all(m.scale.dtype == torch.bfloat16 and m.__dict__.get("_sdnq_hip_plan") is None for m in mods)
    loader.apply_sdnq_options_to_model(blk, dequantize_fp32=True, quantized_matmul_dtype="float8_e4m3fn")
    assert all(m.__dict__.get("_sdnq_hip_plan") is not None and m.__dict__["_sdnq_hip_plan"][1] == "fp8" for m in mods)


@pytest.mark.gpu
def test_compile_after_options_follows_the_new_mode(gpu_device):
    """The compiled model after apply_sdnq_options_to_model equals the eager model in the NEW mode (aot_eager: same kernels)."""
    from sdnq_amd import loader
    blk = _quantized_block(gpu_device, weights_dtype="int8", group_size=-1, use_quantized_matmul=True)
    sdnq_amd.accelerate(blk)
    x = torch.randn(2, 77, 128, device=gpu_device, dtype=torch.bfloat16)
    with torch.no_grad():
        qmm = blk(x)
        loader.apply_sdnq_options_to_model(blk, use_quantized_matmul=False)
        want = blk(x)
        assert not torch.equal(want, qmm)  # the float mode really computes something else
        torch._dynamo.reset()
        got = torch.compile(blk, fullgraph=True, backend="aot_eager")(x)
        assert torch.equal(got, want)


class ConvBlock(torch.nn.Module):
    """A ResNet-style block of a UNet: two 3x3 convs, a strided down-sampling conv, a 1x1 skip, a Conv1d and a grouped conv."""

    def __init__(self, c=32):
        super().__init__()
        self.conv1, self.conv2 = torch.nn.Conv2d(c, 2 * c, 3, padding=1), torch.nn.Conv2d(2 * c, 2 * c, 3, padding=1)
        self.skip = torch.nn.Conv2d(c, 2 * c, 1)
        self.down = torch.nn.Conv2d(2 * c, 2 * c, 3, stride=2, padding=1)
        self.grouped = torch.nn.Conv2d(2 * c, 2 * c, 3, padding=1, groups=2)
        self.c1d = torch.nn.Conv1d(2 * c, c, 3, padding=2, dilation=2)

    def forward(self, x):
        h = self.conv2(torch.nn.functional.silu(self.conv1(x))) + self.skip(x)
        h = self.grouped(self.down(h))
        return self.c1d(h.flatten(2))


def _quantized_conv_block(device):
    torch.manual_seed(0)
    blk = ConvBlock().to(torch.bfloat16).to(device)
    blk, _ = sdnq_amd.apply_sdnq_to_module(blk, sdnq_amd.SDNQConfig(weights_dtype="int8", quant_conv=True, use_quantized_matmul_conv=True,
                                                                    minimum_allowed_numel=256, minimum_allowed_channel_size=16))
    return blk


def test_dynamo_traces_conv_layers_without_graph_breaks_cpu():
    """Round 4: quantized Conv1d / Conv2d / Conv3d layers trace as ONE sdnq_hip::layer_forward operator each (fake implementation =
    the conv output geometry), so a compiled UNet has no graph break at its quantized convs."""
    blk = _quantized_conv_block("cpu")
    convs = [m for m in blk.modules() if hasattr(m, "sdnq_dequantizer")]
    assert len(convs) == 6 and all(m.__dict__.get("_sdnq_hip_handle") is not None for m in convs)
    x = torch.randn(2, 32, 12, 10).to(torch.bfloat16)
    gm, _ = torch._dynamo.export(blk)(x)  # raises on any graph break
    calls = [n for n in gm.graph.nodes if n.op == "call_function" and n.target in (torch.ops.sdnq_hip.layer_forward, torch.ops.sdnq_hip.layer_forward.default)]
    assert len(calls) == 6
    # the fake implementation's shapes are the real convs' shapes
    want = ConvBlock().to(torch.bfloat16)(x).shape
    meta = calls[-1].meta.get("val", calls[-1].meta.get("example_value"))
    assert tuple(meta.shape) == tuple(want) == (2, 32, 30)
    conv3 = sdnq_amd.sdnq_quantize_layer(torch.nn.Conv3d(16, 32, (3, 3, 1), stride=(1, 2, 1), padding=(1, 0, 0)).to(torch.bfloat16),
                                         sdnq_amd.SDNQConfig(weights_dtype="int8", quant_conv=True, use_quantized_matmul_conv=True))[0]
    fake = torch.ops.sdnq_hip.layer_forward.default
    from torch._subclasses.fake_tensor import FakeTensorMode
    with FakeTensorMode():
        y = fake(torch.empty(2, 16, 5, 9, 4, dtype=torch.bfloat16), conv3.__dict__["_sdnq_hip_handle"])
    assert tuple(y.shape) == (2, 32, 5, 4, 4)


@pytest.mark.gpu
def test_compiled_conv_block_equals_eager(gpu_device):
    """torch.compile(fullgraph=True) of a block of quantized convs (plain, strided, 1x1, grouped, Conv1d with dilation): the
    aot_eager result is bit-identical to the eager one (same kernels behind the operator)."""
    blk = _quantized_conv_block(gpu_device)
    sdnq_amd.accelerate(blk)
    x = torch.randn(2, 32, 24, 16, device=gpu_device, dtype=torch.bfloat16)
    with torch.no_grad():
        want = blk(x)
        torch._dynamo.reset()
        got = torch.compile(blk, fullgraph=True, backend="aot_eager")(x)
    assert got.shape == want.shape and torch.equal(got, want)
