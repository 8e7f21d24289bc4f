"""The C++ fast path of the eager Linear forward (sdnq_amd/csrc/fastpath.cpp, round 6): a plan carries later calls of a layer through one
C++ call.  It restates decisions of sdnq_amd/linear.py, which stays the complete forward: every output must be BIT-identical with the
plans switched off, a plan must notice every change the Python forward notices (parameter object / storage / version, module switches,
group membership), and must decline what it does not carry (few rows, stream capture on the scratch route, other dtypes)."""
import numpy as np
import pytest
import torch


def _bits(t: torch.Tensor) -> np.ndarray:
    return t.detach().contiguous().view(torch.int16).cpu().numpy()


def _layer(k, n, device, dtype=torch.bfloat16, bias=True, seed=0, **cfg):
    import sdnq_amd
    torch.manual_seed(seed)
    lin = torch.nn.Linear(k, n, bias=bias).to(dtype)
    kw = dict(weights_dtype="int8", group_size=-1, use_quantized_matmul=True)
    kw.update(cfg)
    return sdnq_amd.sdnq_quantize_layer(lin, sdnq_amd.SDNQConfig(**kw))[0].to(device)


def _fp():
    from sdnq_amd import linear as L
    if L._FP is None:
        pytest.skip("sdnq_amd._fastpath is not built")
    return L._FP


def test_module_types_without_gpu():
    """CPU: the claim state of a group and the prefetch chain units behave like the Python forms they replace."""
    fp = _fp()
    g = fp.Group(3)
    assert g.pending() == 0 and g.peek() is None and g.wasted == 0
    x = torch.zeros(4, 8)
    outs = [torch.full((4, 2), float(i)) for i in range(3)]
    g.publish(x, outs)
    assert g.pending() == 3 and g.peek()[0] is x and g.peek()[2] == {0, 1, 2}
    assert g.claim(1, x.view(2, 2, 8)) is None          # another tensor object
    y = g.claim(1, x)
    assert y.shape == (4, 2) and float(y.sum()) == 8.0 and g.pending() == 2
    assert g.claim(1, x) is None                          # handed out once
    x.add_(1)                                             # the input changed (version counter): nothing is served any more
    assert g.claim(0, x) is None
    g.publish(x, outs)
    g.wasted = 5
    assert g.claim(0, x) is not None and g.claim(2, x.clone()) is None
    y = g.claim(2, x.view(4, 8))                          # x.view(4, 8) is a NEW object: no claim
    assert y is None
    assert g.claim(1, x) is not None and g.claim(2, x) is not None
    assert g.peek() is None and g.wasted == 0             # everything claimed: nothing is held, the miss count is reset
    with torch.inference_mode():
        xi = torch.zeros(4, 8)
    g.publish(xi, outs)                                   # inference tensors carry no version counter: never served
    assert g.peek() is None
    u = fp.Unit(((4096, 100), (8192, 50)), 0)
    u.launch(); u.launch()
    with pytest.raises(ValueError):
        fp.Unit(tuple((4096 * i, 1) for i in range(1, 6)), 0)
    with pytest.raises(TypeError):
        fp.Plan(("weight",), (None, None), 0, 8, 8)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_plans_are_bit_identical_and_carry_the_calls(dtype, gpu_device, monkeypatch):
    from sdnq_amd import linear as L, ops
    fp = _fp()
    L.clear_activation_cache()
    specs = [(1280, 1280, {}), (5120, 1280, {}), (640, 2560, dict(weights_dtype="fp8", quantized_matmul_dtype="fp8")), (1280, 640, dict(bias=False))]
    mods = [_layer(k, n, gpu_device, dtype=dtype, seed=i, **cfg) for i, (k, n, cfg) in enumerate(specs)]
    torch.manual_seed(7)
    xs = {k: [torch.randn(*shape, k, device=gpu_device).to(dtype) for shape in [(1024,), (2, 77), (3, 1, 40), (8,), (4096,)]] for k in (1280, 5120, 640)}
    strided = {k: torch.randn(256, 2 * k, device=gpu_device).to(dtype)[:, :k] for k in xs}   # row stride 2K: taken as it is
    odd = {k: torch.randn(k, 64, device=gpu_device).to(dtype).t() for k in xs}               # column-major: made contiguous first

    def run():
        out = []
        for (k, n, _), mod in zip(specs, mods):
            for x in xs[k] + [strided[k], odd[k]]:
                out.append(mod(x))
        torch.cuda.synchronize()
        return out

    with torch.no_grad():
        for _ in range(L.UNSHARED_AFTER + 2):   # the layers learn that nobody shares their input, then take (and plan) the fast routes
            L.clear_activation_cache()
            run()
        assert all("_sdnq_plan" in m.__dict__ for m in mods)
        fp.reset_counters(); ops.reset_fused_calls()
        L.clear_activation_cache()
        got = run()
        carried = fp.plan_calls()
        assert carried == len(mods) * 6, carried   # every call with >= 32 rows (the 8-row call takes the Python forward's float branch)
        assert fp.fused_calls() >= 2               # 1024 x 1280 x 1280 and 1024 x 640 x 1280 are the one-launch route's shapes
        monkeypatch.setattr(L, "FAST_PLANS", False)   # an upper-case switch: every plan is stale from here, none is made
        fp.reset_counters()
        L.clear_activation_cache()
        want = run()
        assert fp.plan_calls() == 0 and not any("_sdnq_plan" in m.__dict__ for m in mods)
    for a, b in zip(got, want):
        assert a.shape == b.shape and a.dtype == b.dtype and np.array_equal(_bits(a), _bits(b))


@pytest.mark.gpu
def test_plan_goes_stale_with_its_parameters(gpu_device):
    from sdnq_amd import linear as L
    fp = _fp()
    mod = _layer(1280, 1280, gpu_device)
    ref = _layer(1280, 1280, gpu_device)
    x = torch.randn(1024, 1280, device=gpu_device).to(torch.bfloat16)
    with torch.no_grad():
        for _ in range(L.UNSHARED_AFTER + 2):
            L.clear_activation_cache(); mod(x)
        assert "_sdnq_plan" in mod.__dict__
        # a new bias OBJECT
        nb = torch.randn(1280, device=gpu_device).to(torch.bfloat16)
        mod.bias = torch.nn.Parameter(nb.clone(), requires_grad=False)
        ref.bias = torch.nn.Parameter(nb.clone(), requires_grad=False)
        fp.reset_counters()
        y = mod(x)
        assert fp.plan_calls() == 0, "the plan of the old bias carried the call"
        assert np.array_equal(_bits(y), _bits(ref(x)))
        for _ in range(3):
            mod(x)
        assert fp.plan_calls() >= 1   # a fresh plan took over
        # the scale changed IN PLACE (same object and address, new version): the Python forward rebuilds its state, so must the plan's caller
        mod.scale.mul_(2.0)
        ref.scale.mul_(2.0)
        fp.reset_counters()
        y = mod(x)
        assert fp.plan_calls() == 0
        assert np.array_equal(_bits(y), _bits(ref(x)))
        # the weight moved to new storage
        mod.weight.data = mod.weight.data.clone()
        fp.reset_counters()
        y2 = mod(x)
        assert fp.plan_calls() == 0 and np.array_equal(_bits(y2), _bits(y))
        # few rows / another dtype / a CPU tensor: declined, not stale
        for _ in range(3):
            mod(x)
        fp.reset_counters()
        mod(x[:8]); mod(x.float())
        assert fp.plan_calls() == 0 and "_sdnq_plan" in mod.__dict__
        mod(x)
        assert fp.plan_calls() == 1
        # the layer is re-pointed at another forward (apply_sdnq_options_to_model): the int8 plan must not carry the float forward's calls
        import sdnq_amd
        model = torch.nn.Sequential(mod)
        sdnq_amd.apply_sdnq_options_to_model(model, use_quantized_matmul=False)
        assert not mod.sdnq_dequantizer.use_quantized_matmul
        fp.reset_counters()
        y_float = model(x)
        assert fp.plan_calls() == 0 and "_sdnq_plan" not in mod.__dict__
        want = torch.nn.functional.linear(x.float(), mod.sdnq_dequantizer(mod.weight, mod.scale, skip_quantized_matmul=True).float(), mod.bias.float())
        assert float((y_float.float() - want).abs().max()) <= 2.0 ** -7 * float(want.abs().max())   # bf16 activations x dequantized weight: not the int8 matmul
        sdnq_amd.apply_sdnq_options_to_model(model, use_quantized_matmul=True)
        for _ in range(4):
            L.clear_activation_cache(); y_back = model(x)
        assert np.array_equal(_bits(y_back), _bits(ref(x)))


@pytest.mark.gpu
def test_group_members_claim_through_their_plans(gpu_device, monkeypatch):
    from sdnq_amd import linear as L, loader
    fp = _fp()
    solo = [_layer(1280, 1280, gpu_device, seed=i) for i in range(3)]
    linked = [_layer(1280, 1280, gpu_device, seed=i) for i in range(3)]
    assert loader.link_layers(linked)
    group = linked[0].__dict__["_sdnq_group"][0]
    xs = [torch.randn(2, 512, 1280, device=gpu_device).to(torch.bfloat16) for _ in range(4)]
    with torch.no_grad():
        monkeypatch.setattr(L, "LINK_PROJECTIONS", True)
        for x in xs[:2]:
            [m(x) for m in linked]
        assert all("_sdnq_plan" in m.__dict__ for m in linked)
        fp.reset_counters()
        for x in xs:
            got = [m(x) for m in linked]
            want = [m(x) for m in solo]
            for a, b in zip(got, want):
                assert a.shape == b.shape and np.array_equal(_bits(a), _bits(b))
            assert group.last is None and group.wasted == 0
        assert fp.plan_calls() >= 2 * len(xs)   # two of the three members of every step picked theirs up in C++
        # the members stop sharing their input: the group notices (unclaimed outputs twice in a row) and dissolves -- plans included
        for _ in range(3):
            for m in linked:
                m(torch.randn(64, 1280, device=gpu_device).to(torch.bfloat16))
        assert all("_sdnq_group" not in m.__dict__ for m in linked)
        x = xs[0]
        for a, b in zip([m(x) for m in linked], [m(x) for m in solo]):
            assert np.array_equal(_bits(a), _bits(b))
        # identity reuse switched off on this thread (what a compiled graph's operator does): no claim through a plan either
        again = [_layer(1280, 1280, gpu_device, seed=i) for i in range(3)]
        assert loader.link_layers(again)
        for xx in xs[:2]:
            [m(xx) for m in again]
        fp.reset_counters()
        with L.identity_reuse_disabled():
            got = [m(x) for m in again]
        assert fp.plan_calls() == 0
        for a, b in zip(got, [m(x) for m in solo]):
            assert np.array_equal(_bits(a), _bits(b))


@pytest.mark.gpu
def test_plans_under_stream_capture(gpu_device):
    """The scratch route declines while its stream is being captured (the Python forward takes a scratch tensor from the graph's pool); the
    one-launch route is captured as it is.  Replays equal the eager outputs."""
    from sdnq_amd import linear as L
    _fp()
    a = _layer(1280, 1280, gpu_device, seed=1)    # one-launch route
    b = _layer(5120, 1280, gpu_device, seed=2)    # row quantizer + GEMM on the stream's scratch
    xa = torch.randn(1024, 1280, device=gpu_device).to(torch.bfloat16)
    xb = torch.randn(1024, 5120, device=gpu_device).to(torch.bfloat16)
    with torch.no_grad():
        for _ in range(L.UNSHARED_AFTER + 2):
            L.clear_activation_cache(); a(xa); b(xb)
        assert "_sdnq_plan" in a.__dict__ and "_sdnq_plan" in b.__dict__
        want = (a(xa).clone(), b(xb).clone())
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            a(xa); b(xb)
            side.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side):
                ya, yb = a(xa), b(xb)
        torch.cuda.synchronize()
        for _ in range(3):
            ya.zero_(); yb.zero_()
            g.replay()
            torch.cuda.synchronize()
            assert np.array_equal(_bits(ya), _bits(want[0])) and np.array_equal(_bits(yb), _bits(want[1]))
