"""``torch.library`` registration of the HIP kernels: traceable operators with fake (meta) implementations.

The reference registers its GEMM as ``sdnq::scaled_mm`` (kernels/triton_scaled_mm.py:239-248, a ``triton_op`` with
``mutates_args={}``) so that ``torch.compile`` can keep it in the graph; the ctypes calls of this build are opaque to Dynamo
(``data_ptr()``, Python-side caches), so without registration every quantized layer is a graph break.  Registered here, all over
the same C ABI (``include/sdnq_hip.h``) and with the same signatures / asserts as the operator seam:

    sdnq_hip::scaled_mm(a, b, scale_a, scale_b, bias=None, out_dtype=float32) -> Tensor      (schema of sdnq::scaled_mm)
    sdnq_hip::rowquant(x, matmul_dtype, hadamard_group=0) -> (xq, xs)                         (quantize_*_mm_input)
    sdnq_hip::linear_w8a8(x, wq, ws, bias=None, matmul_dtype="int8", hadamard_group=0) -> Tensor   (rowquant + scaled_mm)
    sdnq_hip::dequant(weight, scale, zero_point, svd_up, svd_down, weights_dtype, n, k, group_size, transposed,
                      svd_transposed, hadamard_group, out_dtype) -> Tensor [N, K]            (SDNQDequantizer.__call__)
    sdnq_hip::layer_forward(input, handle) -> Tensor                                           (SDNQLayer.forward of a whole layer)

``layer_forward`` is what makes ``torch.compile(model, fullgraph=True)`` work on an SDNQ model: under compilation
``SDNQLayer.forward`` emits ONE opaque op per layer (the module is found through an integer handle, a Dynamo constant), whose
implementation is the ordinary eager forward with its weight-side caches, but WITHOUT the identity-keyed reuse of activations
(activation cache, linked projections): Inductor recycles buffers in place, so "same tensor object, same version" proves nothing
inside a compiled graph.  Its fake implementation only needs the layer's out_features.
"""
from __future__ import annotations

import weakref
from typing import Optional, Tuple

import torch
from torch.library import custom_op

from . import _lib, ops

_MM = {"int8": ops.MM_I8, "fp8": ops.MM_FP8, "float8_e4m3fn": ops.MM_FP8}
_MM_TORCH = {"int8": torch.int8, "fp8": torch.float8_e4m3fn, "float8_e4m3fn": torch.float8_e4m3fn}


# ---- the operator seam ------------------------------------------------------------------------------------------------------
@custom_op("sdnq_hip::scaled_mm", mutates_args=())
def scaled_mm(a: torch.Tensor, b: torch.Tensor, scale_a: torch.Tensor, scale_b: torch.Tensor, bias: Optional[torch.Tensor] = None,
              out_dtype: torch.dtype = torch.float32) -> torch.Tensor:
    from .kernel_wrappers import _scaled_mm
    if a.dtype == torch.int8 and b.dtype == torch.int8:
        return _scaled_mm(ops.MM_I8, a, b, scale_a, scale_b, bias, out_dtype)
    if a.dtype == torch.float8_e4m3fn and b.dtype == torch.float8_e4m3fn:
        return _scaled_mm(ops.MM_FP8, a, b, scale_a, scale_b, bias, out_dtype)
    raise _lib.SdnqHipError("sdnq_hip::scaled_mm expects int8 or float8_e4m3fn operands of one type")


@scaled_mm.register_fake
def _(a, b, scale_a, scale_b, bias=None, out_dtype=torch.float32):
    torch._check(a.shape[1] == b.shape[0], lambda: "Incompatible dimensions")
    return a.new_empty((a.shape[0], b.shape[1]), dtype=out_dtype)


@custom_op("sdnq_hip::rowquant", mutates_args=())
def rowquant(x: torch.Tensor, matmul_dtype: str, hadamard_group: int = 0) -> Tuple[torch.Tensor, torch.Tensor]:
    x2 = x.reshape(-1, x.shape[-1])
    if x2.stride(-1) != 1 or (x2.stride(0) * x2.element_size()) % 16:
        x2 = x2.contiguous()
    xq, xs, _, _ = ops.rowquant(x2, _MM[matmul_dtype], hadamard_group)
    return xq, xs


@rowquant.register_fake
def _(x, matmul_dtype, hadamard_group=0):
    m = x.numel() // x.shape[-1]
    return x.new_empty((m, x.shape[-1]), dtype=_MM_TORCH[matmul_dtype]), x.new_empty((m, 1), dtype=torch.float32)


@custom_op("sdnq_hip::linear_w8a8", mutates_args=())
def linear_w8a8(x: torch.Tensor, wq: torch.Tensor, ws: torch.Tensor, bias: Optional[torch.Tensor] = None, matmul_dtype: str = "int8",
                hadamard_group: int = 0) -> torch.Tensor:
    """y = scaled_mm(rowquant(x), wq^T) with wq the physical [N, K] operand and ws [N] its row scales."""
    k = x.shape[-1]
    x2 = x.reshape(-1, k)
    if x2.stride(-1) != 1 or (x2.stride(0) * x2.element_size()) % 16:
        x2 = x2.contiguous()
    y, _, _ = ops.linear_w8a8(_MM[matmul_dtype], x2, wq, ws.reshape(-1), bias, x.dtype, hadamard_group)
    return y.view(*x.shape[:-1], wq.shape[0])


@linear_w8a8.register_fake
def _(x, wq, ws, bias=None, matmul_dtype="int8", hadamard_group=0):
    return x.new_empty((*x.shape[:-1], wq.shape[0]))


@custom_op("sdnq_hip::dequant", mutates_args=())
def dequant(weight: torch.Tensor, scale: torch.Tensor, zero_point: Optional[torch.Tensor], svd_up: Optional[torch.Tensor],
            svd_down: Optional[torch.Tensor], weights_dtype: str, n: int, k: int, group_size: int, transposed: bool,
            svd_transposed: bool, hadamard_group: int, out_dtype: torch.dtype) -> torch.Tensor:
    qw = ops.make_quant_weight(weights_dtype, weight, scale, zero_point, svd_up, svd_down, n, k, group_size if group_size > 0 else k,
                               transposed=transposed, svd_transposed=svd_transposed)
    return ops.dequant(qw, out_dtype, hadamard_group)


@dequant.register_fake
def _(weight, scale, zero_point, svd_up, svd_down, weights_dtype, n, k, group_size, transposed, svd_transposed, hadamard_group, out_dtype):
    return scale.new_empty((n, k), dtype=out_dtype)


# ---- whole layers -----------------------------------------------------------------------------------------------------------
_layers: dict[int, "weakref.ReferenceType"] = {}
_next_handle = [1]


def layer_handle(module: torch.nn.Module) -> int:
    """Integer handle of an SDNQ layer, assigned EAGERLY (SDNQLayer.__init__ / accelerate()): stable for the module's lifetime,
    stored as the plain attribute ``_sdnq_hip_handle`` -- a constant to Dynamo -- and registered here so that the operator (and
    its fake implementation, at trace time) can find the layer."""
    h = module.__dict__.get("_sdnq_hip_handle")
    if h is not None:
        ref = _layers.get(h)
        if ref is None or ref() is not module:  # a handle copied along with the module's __dict__ (copy / pickle): not this module's
            h = None
    if h is None:
        h = _next_handle[0]
        _next_handle[0] += 1
        module.__dict__["_sdnq_hip_handle"] = h
        _layers[h] = weakref.ref(module, lambda _r, h=h: _layers.pop(h, None))
    try:
        module.__dict__["_sdnq_hip_plan"] = layer_plan(module)
    except Exception:  # noqa: BLE001  (a foreign / half-built module: the whole-layer operator handles it)
        module.__dict__["_sdnq_hip_plan"] = None
    return h


def _layer(handle: int) -> torch.nn.Module:
    ref = _layers.get(handle)
    mod = None if ref is None else ref()
    if mod is None:
        raise _lib.SdnqHipError(f"sdnq_hip::layer_forward: no live layer behind handle {handle}")
    return mod


@custom_op("sdnq_hip::layer_forward", mutates_args=())
def layer_forward(input: torch.Tensor, handle: int) -> torch.Tensor:
    mod = _layer(handle)
    from .linear import identity_reuse_disabled
    with identity_reuse_disabled():  # no tensor-identity-keyed reuse inside a compiled graph (linear.py: Inductor recycles buffers in place)
        y = mod.forward_func(mod, input)
    return y.clone() if y._base is not None and y._base is input else y  # a custom op must not return an alias of its input


@custom_op("sdnq_hip::layer_matmul", mutates_args=())
def layer_matmul(xq: torch.Tensor, xs: torch.Tensor, handle: int, out_dtype: torch.dtype) -> torch.Tensor:
    """The matmul half of a plain w8a8 layer on an activation that `sdnq_hip::rowquant` already quantized:
    y [M, N] = scaled_mm(xq, Wq, xs, ws, bias).  Under torch.compile a layer whose forward is exactly rowquant + scaled_mm is traced
    as these two operators (``layer_plan``), so that the graph's own common-subexpression elimination merges the row quantization of
    layers that consume one tensor (to_q / to_k / to_v, every cross-attention to_k / to_v) -- the dataflow-level, and therefore safe,
    form of the eager path's identity-keyed activation cache."""
    from . import linear as L
    mod = _layer(handle)
    st = L._state(mod)
    mm = ops.MM_I8 if xq.dtype == torch.int8 else ops.MM_FP8
    wq, ws, zp = L._prepare_mm_weights(mod, st, mm)
    if zp is not None or st.svd_up is not None:
        raise _lib.SdnqHipError("sdnq_hip::layer_matmul: the layer changed to a form with zero-point / low-rank terms after tracing")
    return ops.scaled_mm(mm, xq, wq, xs, ws, L._attr(mod, "bias"), out_dtype)


@layer_matmul.register_fake
def _(xq, xs, handle, out_dtype):
    return xq.new_empty((xq.shape[0], _layer(handle).sdnq_dequantizer.out_features), dtype=out_dtype)


def layer_plan(module: torch.nn.Module):
    """("q", matmul dtype name, hadamard group) when the layer's forward at M >= 32 is exactly rowquant + scaled_mm (row-wise or
    re-quantized weights, no zero-point term, no SVD, fp32 scales), else None: decided from the module's static configuration."""
    from .common import dtype_dict
    dq = module.__dict__.get("sdnq_dequantizer")
    if dq is None or not dq.use_quantized_matmul or dq.is_conv or getattr(module, "svd_up", None) is not None:
        return None
    mmd = str(dq.quantized_matmul_dtype).replace("torch.", "")
    if mmd not in ("int8", "float8_e4m3fn", "fp8"):
        return None
    sc = module.__dict__.get("_parameters", {}).get("scale")
    if sc is None or sc.dtype != torch.float32:
        return None
    if dtype_dict[dq.weights_dtype]["is_unsigned"] and not dq.re_quantize_for_matmul:
        return None  # zero-point term in the epilogue
    had = dq.hadamard_group_size if dq.use_hadamard else 0
    if had and dq.in_features > 5120:
        return None
    return ("q", "int8" if mmd == "int8" else "fp8", int(had))


@layer_forward.register_fake
def _(input, handle):
    mod = _layer(handle)
    dq = mod.sdnq_dequantizer
    if dq.is_conv:
        raise NotImplementedError("sdnq_hip::layer_forward traces Linear layers; conv layers run eagerly (graph break)")
    return input.new_empty((*input.shape[:-1], dq.out_features))
