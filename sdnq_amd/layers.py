"""Module wrappers of the drop-in boundary.

Same surface as the reference's ``layers/__init__.py`` (SDNQLayer :6-33, SDNQLinear :36-37,
get_sdnq_wrapper_class :72-93): the wrapper adopts the original layer's ``__dict__`` (so it shares
``_parameters``), remembers ``original_class`` and routes ``forward`` through ``self.forward_func(self, x)``
-- the seam this build plugs its HIP forwards into.  Attributes ``weight, scale, zero_point, svd_up,
svd_down, bias, sdnq_dequantizer`` keep the reference names and state_dict layout.
"""
from __future__ import annotations

from collections.abc import Callable

import torch


class SDNQLayer(torch.nn.Module):
    def __init__(self, original_layer: torch.nn.Module, forward_func: Callable):
        torch.nn.Module.__init__(self)
        skip = {"forward", "forward_func", "original_class", "state_dict", "load_state_dict"}
        for key, value in original_layer.__dict__.items():
            if key not in skip:
                setattr(self, key, value)
        self.original_class = original_layer.__class__
        self.forward_func = forward_func
        if _traceable(self):
            from . import torch_ops
            torch_ops.layer_handle(self)

    # per-object runtime state that must not travel with a copy: the operator handle names THIS module, the projection group and the
    # kernel-ready tensor cache point at the original's siblings / parameters
    _RUNTIME_KEYS = ("_sdnq_hip_handle", "_sdnq_hip_plan", "_sdnq_group", "_sdnq_hip_state", "_sdnq_unshared", "_sdnq_compile_groups", "_sdnq_plan", "_sdnq_plan_declined")

    def __deepcopy__(self, memo):
        import copy
        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        for key, value in self.__dict__.items():
            if key not in self._RUNTIME_KEYS:
                new.__dict__[key] = copy.deepcopy(value, memo)
        if _traceable(new):
            from . import torch_ops
            torch_ops.layer_handle(new)
        return new

    def __getstate__(self):
        return {k: v for k, v in self.__dict__.items() if k not in self._RUNTIME_KEYS}

    def __setstate__(self, state):
        torch.nn.Module.__setstate__(self, state)
        if _traceable(self):
            from . import torch_ops
            torch_ops.layer_handle(self)

    @property
    def dtype(self) -> torch.dtype:
        return self.sdnq_dequantizer.result_dtype if hasattr(self, "sdnq_dequantizer") else self.weight.dtype

    def dequantize(self):
        """Back to the original float layer (reference layers/__init__.py:19-27)."""
        if hasattr(self, "sdnq_dequantizer"):
            dq = self.sdnq_dequantizer
            w = dq(self.weight, self.scale, zero_point=self.zero_point, svd_up=self.svd_up, svd_down=self.svd_down,
                   skip_quantized_matmul=dq.use_quantized_matmul)
            self.weight = torch.nn.Parameter(w, requires_grad=True)
            del self.sdnq_dequantizer, self.scale, self.zero_point, self.svd_up, self.svd_down
            self.__dict__.pop("_sdnq_hip_state", None)
            self.__dict__.pop("_sdnq_plan", None)
            group = self.__dict__.pop("_sdnq_group", None)
            if group is not None:  # linked siblings (attention projections sharing this layer's input) run alone from now on
                group[0].dissolve()
        self.__class__ = self.original_class
        del self.original_class, self.forward_func
        return self

    def forward(self, *args, **kwargs) -> torch.Tensor:
        if torch.compiler.is_compiling() and len(args) == 1 and not kwargs:
            # under torch.compile: ONE opaque operator per layer (sdnq_amd/torch_ops.py) instead of a graph break at the ctypes calls
            handle = getattr(self, "_sdnq_hip_handle", None)
            if handle is not None:
                x = args[0]
                plan = getattr(self, "_sdnq_hip_plan", None)
                if plan is not None and x.numel() // x.shape[-1] >= 32:
                    # rowquant + matmul as TWO operators: layers that consume one tensor then share its row quantization through
                    # the graph's common-subexpression elimination (torch_ops.layer_matmul)
                    xq, xs = torch.ops.sdnq_hip.rowquant(x, plan[1], plan[2])
                    y = torch.ops.sdnq_hip.layer_matmul(xq, xs, handle, x.dtype)
                    return y.view(*x.shape[:-1], y.shape[-1])
                return torch.ops.sdnq_hip.layer_forward(x, handle)
        return self.forward_func(self, *args, **kwargs)

    def __repr__(self) -> str:
        return (f"{self.__class__.__name__}(original_class={self.original_class} forward_func={self.forward_func} "
                f"sdnq_dequantizer={getattr(self, 'sdnq_dequantizer', None)})")


def _traceable(layer) -> bool:
    """Layers that trace as sdnq_hip:: operators under torch.compile: Linear (one or two operators, torch_ops.layer_plan) and the conv
    layers (one opaque layer_forward operator each -- round 4: a compiled UNet has no graph breaks at its 49 quantized convs)."""
    dq = layer.__dict__.get("sdnq_dequantizer")
    return dq is not None and dq.layer_class_name in ("Linear", "SDNQLinear", "Conv1d", "Conv2d", "Conv3d", "SDNQConv1d", "SDNQConv2d", "SDNQConv3d")


class SDNQLinear(SDNQLayer, torch.nn.Linear):
    original_class: torch.nn.Linear


class SDNQConv1d(SDNQLayer, torch.nn.Conv1d):
    original_class: torch.nn.Conv1d


class SDNQConv2d(SDNQLayer, torch.nn.Conv2d):
    original_class: torch.nn.Conv2d


class SDNQConv3d(SDNQLayer, torch.nn.Conv3d):
    original_class: torch.nn.Conv3d


torch.serialization.add_safe_globals([SDNQLayer, SDNQLinear, SDNQConv1d, SDNQConv2d, SDNQConv3d])


def get_sdnq_wrapper_class(original_layer: torch.nn.Module, forward_func: Callable) -> SDNQLayer:
    name = original_layer.__class__.__name__
    if name == "Linear":
        return SDNQLinear(original_layer, forward_func)
    if name == "Conv1d":
        return SDNQConv1d(original_layer, forward_func)
    if name == "Conv2d":
        return SDNQConv2d(original_layer, forward_func)
    if name == "Conv3d":
        return SDNQConv3d(original_layer, forward_func)
    # transposed conv / embedding wrappers are not built
    return SDNQLayer(original_layer, forward_func)
