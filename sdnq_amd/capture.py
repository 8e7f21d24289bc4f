"""``sdnq_amd.capture(model, *example_inputs)``: a model step as ONE hipGraph replay.

At bs = 1 an SDXL-UNet step is ~900 kernel launches of 3-20 us each: launched eagerly from Python the step is host-bound (12.2 ms
where the same launches replayed from a graph take 6.8 ms, profiles/r05_bench_sdxl_int8_eager.json) -- ~16 us of interpreter time per
layer (linear.py: keys, cache look-ups, allocations, ctypes marshalling) in front of a 7-us kernel.  The reference leaves this to
``torch.compile`` (layers/__init__.py:29-30 only dispatches ``forward_func``); a diffusers user who calls ``accelerate(model)`` and runs
the pipeline eagerly would never see the headline number.  This wraps what ``bench.py`` does by hand -- warm-up steps on a side stream
(the cross-layer weight prefetch learns the launch order, linked projections form), then ``torch.cuda.graph`` capture -- with the checks
the host-side reuse machinery needs:

* inputs live in STATIC buffers (device tensors are copied in before every replay; shapes, dtypes and non-tensor arguments must equal
  the captured ones -- anything else raises, or re-captures with ``recapture=True``);
* the graph holds raw pointers to every parameter: a fingerprint of the model's parameter storages is checked before every replay
  (``model.to(...)``, ``load_state_dict(assign=True)``, ``apply_sdnq_options_to_model`` after capture -> ``RuntimeError``, never stale weights);
* the activation cache / projection groups are cleared around the capture, so nothing captured aliases a tensor of an eager step;
* outputs are the graph's static output tensors: valid until the next replay (``clone_outputs=True`` hands out copies).

Works for any module whose forward is capturable (no host synchronisation, static shapes): the float operators between the SDNQ
layers (attention, norms) are captured as they are.
"""
from __future__ import annotations

import torch

from . import linear as _L


def _flatten(obj, out):
    """Tensors of a nested (tuple / list / dict) argument structure, in a fixed order; returns a hashable description of the rest."""
    if isinstance(obj, torch.Tensor):
        out.append(obj)
        return ("T", tuple(obj.shape), obj.dtype, str(obj.device))
    if isinstance(obj, (tuple, list)):
        return (type(obj).__name__, tuple(_flatten(o, out) for o in obj))
    if isinstance(obj, dict):
        return ("dict", tuple((k, _flatten(obj[k], out)) for k in sorted(obj)))
    try:
        hash(obj)
        return ("V", obj)
    except TypeError:
        return ("V", repr(obj))


def _rebuild(obj, it):
    if isinstance(obj, torch.Tensor):
        return next(it)
    if isinstance(obj, (tuple, list)):
        return type(obj)(_rebuild(o, it) for o in obj)
    if isinstance(obj, dict):
        return {k: _rebuild(obj[k], it) for k in obj}
    return obj


def _fingerprint(model: torch.nn.Module):
    """What the captured kernels' pointers depend on: every parameter / buffer storage and the cached kernel operands of the SDNQ layers."""
    fp = []
    for t in list(model.parameters()) + list(model.buffers()):
        fp.append((t.data_ptr(), t.numel(), t.dtype))
    for mod in model.modules():
        st = mod.__dict__.get("_sdnq_hip_state")
        if st is not None:
            for name in ("mm_weight", "mm_scale"):
                t = getattr(st, name, None)
                if isinstance(t, torch.Tensor):
                    fp.append((t.data_ptr(), t.numel(), t.dtype))
    return hash(tuple(fp))


class CapturedModel:
    """Callable returned by :func:`capture`.  ``graph`` is the ``torch.cuda.CUDAGraph``; ``replays`` counts calls."""

    def __init__(self, model: torch.nn.Module, args, kwargs, warmup: int = 3, clone_outputs: bool = False, recapture: bool = False, pool=None):
        if not torch.cuda.is_available():
            raise _L.ops._lib.SdnqHipError("sdnq_amd.capture needs a GPU (no CPU fallback)")
        self.model, self.warmup, self.clone_outputs, self.recapture, self.pool = model, max(int(warmup), 1), clone_outputs, recapture, pool
        self.replays = 0
        self.graph = None
        self._capture(args, kwargs)

    def _capture(self, args, kwargs):
        tensors = []
        self._spec = _flatten((args, kwargs), tensors)
        if not tensors or not all(t.is_cuda for t in tensors):
            raise ValueError("sdnq_amd.capture: every tensor argument must live on the GPU (the graph reads static device buffers)")
        self._device = tensors[0].device
        self._static_in = [t.detach().clone() for t in tensors]
        s_args, s_kwargs = _rebuild((args, kwargs), iter(self._static_in))
        self._stream = torch.cuda.Stream(device=self._device)
        self._stream.wait_stream(torch.cuda.current_stream(self._device))
        was_training = self.model.training
        self.model.eval()
        with torch.no_grad(), torch.cuda.stream(self._stream):
            # eager steps on the capture stream: the layers build their kernel state, the activation-sharing pattern and the launch order
            # of the step are learnt (ProjectionGroups form, the weight prefetch names each launch's successors)
            for _ in range(self.warmup):
                _L.clear_activation_cache()
                self.model(*s_args, **s_kwargs)
                _L.join_weight_pipeline()
            self._stream.synchronize()
            _L.clear_activation_cache()
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph, stream=self._stream, **({"pool": self.pool} if self.pool is not None else {})):
                out = self.model(*s_args, **s_kwargs)
                _L.join_weight_pipeline()
            # nothing of the capture may be served to an eager step later (the entries alias the graph's private memory pool)
            _L.clear_activation_cache()
        torch.cuda.current_stream(self._device).wait_stream(self._stream)
        self.model.train(was_training)
        self._static_out = out
        self._fp = _fingerprint(self.model)

    def __call__(self, *args, **kwargs):
        tensors = []
        spec = _flatten((args, kwargs), tensors)
        if spec != self._spec:
            if not self.recapture:
                raise ValueError("sdnq_amd.capture: the arguments differ from the captured ones in shape, dtype, device or a non-tensor value "
                                 "(capture again, or pass recapture=True)")
            self._capture(args, kwargs)
            tensors = []
            _flatten((args, kwargs), tensors)
        if _fingerprint(self.model) != self._fp:
            if not self.recapture:
                raise RuntimeError("sdnq_amd.capture: the model's parameters moved since the capture (model.to / load_state_dict / "
                                   "apply_sdnq_options_to_model?): the graph would read freed memory -- capture again")
            self._capture(args, kwargs)
        cur = torch.cuda.current_stream(self._device)
        with torch.no_grad():
            for dst, src in zip(self._static_in, tensors):
                if dst.data_ptr() != src.data_ptr():
                    dst.copy_(src, non_blocking=True)
        self.graph.replay()  # (enqueued on the current stream, behind the copies)
        self.replays += 1
        del cur
        if self.clone_outputs:
            outs = []
            _flatten(self._static_out, outs)
            return _rebuild(self._static_out, iter([o.clone() for o in outs]))
        return self._static_out


def capture(model: torch.nn.Module, *example_args, warmup: int = 3, clone_outputs: bool = False, recapture: bool = False, pool=None,
            **example_kwargs) -> CapturedModel:
    """Capture ``model(*example_args, **example_kwargs)`` into a hipGraph and return a callable that replays it (see module docstring).

    ``model`` is usually the result of ``accelerate(model)`` / ``load_sdnq_model``; the example tensors define the static shapes."""
    return CapturedModel(model, example_args, example_kwargs, warmup=warmup, clone_outputs=clone_outputs, recapture=recapture, pool=pool)
