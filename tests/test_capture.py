"""sdnq_amd.capture: a model step as one hipGraph replay (round 6; what bench.py's headline does by hand, as a public API), and the
per-thread host state of the forwards (two pipelines on two threads of one process).

The replayed step must give the bits the eager step gives -- the same kernels on the same operands -- for every input written into the
static buffers, with layers that share their input (activation cache, linked projections), float operators between the SDNQ layers, and
must refuse to replay after the model's parameters moved."""
import threading

import pytest
import torch

pytestmark = pytest.mark.gpu


class Block(torch.nn.Module):
    """to_q / to_k / to_v on one tensor (the activation-sharing pattern of an attention block), a float operator, two own-input layers."""

    def __init__(self, c=640, h=1280, dtype=torch.bfloat16):
        super().__init__()
        self.norm = torch.nn.LayerNorm(c)
        self.to_q, self.to_k, self.to_v = (torch.nn.Linear(c, c, bias=False) for _ in range(3))
        self.to_out = torch.nn.Linear(c, c)
        self.ff1, self.ff2 = torch.nn.Linear(c, h), torch.nn.Linear(h, c)
        self.to(dtype)

    def forward(self, x):
        h = self.norm(x)
        q, k, v = self.to_q(h), self.to_k(h), self.to_v(h)
        a = torch.softmax((q * k).float(), dim=-1).to(x.dtype) * v
        x = x + self.to_out(a)
        return x + self.ff2(torch.nn.functional.gelu(self.ff1(x)))


def make_model(device, seed=0, weights_dtype="int8", **cfg):
    import sdnq_amd
    torch.manual_seed(seed)
    model = Block().eval()
    conf = sdnq_amd.SDNQConfig(weights_dtype=weights_dtype, group_size=cfg.pop("group_size", -1), use_quantized_matmul=True, **cfg)
    model = sdnq_amd.sdnq_post_load_quant(model, quantization_config=conf, torch_dtype=torch.bfloat16).to(device)
    sdnq_amd.accelerate(model)
    return model


def eager(model, x):
    from sdnq_amd import linear as L
    L.clear_activation_cache()
    with torch.no_grad():
        return model(x)


@pytest.mark.parametrize("weights_dtype,extra", [("int8", {}), ("uint4", {"group_size": 64}), ("int8", {"use_svd": True, "svd_rank": 32})])
def test_captured_step_equals_the_eager_step(weights_dtype, extra, gpu_device):
    import sdnq_amd
    model = make_model(gpu_device, 1, weights_dtype, **extra)
    g = torch.Generator(device=gpu_device).manual_seed(5)
    x0 = torch.randn(256, 640, device=gpu_device, generator=g).to(torch.bfloat16)
    step = sdnq_amd.capture(model, x0)
    assert isinstance(step, sdnq_amd.CapturedModel)
    for i in range(4):
        x = (torch.randn(256, 640, device=gpu_device, generator=g) * (1 + i)).to(torch.bfloat16)
        want = eager(model, x)
        got = step(x)
        assert torch.equal(got.view(torch.int16), want.view(torch.int16)), i
    assert step.replays == 4
    # outputs are the static tensors unless copies are asked for
    y1 = step(x0)
    y2 = step(x0 * 2)
    assert y1.data_ptr() == y2.data_ptr()
    cloned = sdnq_amd.capture(model, x0, clone_outputs=True)
    a = cloned(x0)
    b = cloned(x0 * 2)
    assert a.data_ptr() != b.data_ptr() and torch.equal(a.view(torch.int16), eager(model, x0).view(torch.int16))


def test_capture_refuses_other_shapes_and_moved_parameters(gpu_device):
    import sdnq_amd
    model = make_model(gpu_device, 2)
    x0 = torch.randn(128, 640, device=gpu_device).to(torch.bfloat16)
    step = sdnq_amd.capture(model, x0)
    with pytest.raises(ValueError):
        step(torch.randn(64, 640, device=gpu_device).to(torch.bfloat16))
    with pytest.raises(ValueError):
        step(x0.float())
    # a parameter is replaced (what load_state_dict(assign=True) / model.to() do): the graph's pointers are stale
    model.ff1.bias = torch.nn.Parameter(model.ff1.bias.detach().clone(), requires_grad=False)
    with pytest.raises(RuntimeError):
        step(x0)
    # recapture=True: both situations capture again instead
    step2 = sdnq_amd.capture(model, x0, recapture=True)
    x1 = torch.randn(64, 640, device=gpu_device).to(torch.bfloat16)
    assert torch.equal(step2(x1).view(torch.int16), eager(model, x1).view(torch.int16))
    model.ff2.bias = torch.nn.Parameter(model.ff2.bias.detach().clone() * 2, requires_grad=False)
    assert torch.equal(step2(x1).view(torch.int16), eager(model, x1).view(torch.int16))
    with pytest.raises(ValueError):
        sdnq_amd.capture(model, x0.cpu())


def test_capture_leaves_nothing_behind_for_eager_steps(gpu_device):
    """The activation cache and the projection groups hold nothing of the capture afterwards: an eager step right after a replay is the
    eager step it would have been."""
    import sdnq_amd
    from sdnq_amd import linear as L
    model = make_model(gpu_device, 3)
    x0 = torch.randn(192, 640, device=gpu_device).to(torch.bfloat16)
    want = eager(model, x0)
    step = sdnq_amd.capture(model, x0)
    assert len(L._act_cache.entries) == 0
    step(x0)
    with torch.no_grad():
        again = model(x0)  # (no clear in between: what is cached now was put there by THIS eager step)
    assert torch.equal(again.view(torch.int16), want.view(torch.int16))


def test_two_threads_two_models_two_streams(gpu_device):
    """Round-5 verdict item 9: the forwards' host state (activation cache, identity-reuse switch, weight pipeline, prefetch chain) is
    per thread.  Two threads, each with its own model on its own stream, run interleaved eager steps -- with inputs that share one
    storage address across the threads' caches being impossible by construction -- and every output equals the single-threaded one."""
    import sdnq_amd  # noqa: F401
    from sdnq_amd import linear as L
    models = [make_model(gpu_device, 10 + i) for i in range(2)]
    xs = [[(torch.randn(96 + 32 * i, 640, device=gpu_device) * (j + 1)).to(torch.bfloat16) for j in range(6)] for i in range(2)]
    want = [[eager(models[i], x) for x in xs[i]] for i in range(2)]
    torch.cuda.synchronize()
    errors, caches = [], [None, None]
    barrier = threading.Barrier(2)

    def work(i):
        try:
            stream = torch.cuda.Stream(device=gpu_device)
            barrier.wait()
            with torch.cuda.stream(stream), torch.no_grad():
                for rep in range(25):
                    for j, x in enumerate(xs[i]):
                        if (rep + j) % 3 == 0:
                            L.clear_activation_cache()
                        if (rep + j) % 5 == 0:
                            with L.identity_reuse_disabled():
                                y = models[i](x)
                        else:
                            y = models[i](x)
                        stream.synchronize()
                        if not torch.equal(y.view(torch.int16), want[i][j].view(torch.int16)):
                            errors.append((i, rep, j))
            caches[i] = L._act_cache._get()
        except Exception as e:  # noqa: BLE001
            errors.append((i, repr(e)))

    threads = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert errors == []
    assert caches[0] is not None and caches[0] is not caches[1] and caches[0] is not L._act_cache._get()
