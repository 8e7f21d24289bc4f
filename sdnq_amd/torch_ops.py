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
from typing import List, Optional, Tuple

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
    as these two operators (``layer_plan``), so that the row quantization of layers that consume one tensor (to_q / to_k / to_v, every
    cross-attention to_k / to_v) is ONE graph value after `merge_layer_matmuls` (Inductor's own passes do not merge the copies) -- the
    dataflow-level, and therefore safe, form of the eager path's identity-keyed activation cache."""
    from . import linear as L
    mod = _layer(handle)
    st = L._state(mod)
    mm = ops.MM_I8 if xq.dtype == torch.int8 else ops.MM_FP8
    wq, ws, zp = L._prepare_mm_weights(mod, st, mm)
    if zp is not None or st.svd_up is not None:
        raise _lib.SdnqHipError("sdnq_hip::layer_matmul: the layer changed to a form with zero-point / low-rank terms after tracing")
    if L.PREFETCH_NEXT and st.mm_weight is wq:  # the weight prefetch across layers (linear._PrefetchChain), as on the eager / graph path
        L._pf_launch(st, (wq,))
    return ops.scaled_mm(mm, xq, wq, xs, ws, L._attr(mod, "bias"), out_dtype)


@layer_matmul.register_fake
def _(xq, xs, handle, out_dtype):
    return xq.new_empty((xq.shape[0], _layer(handle).sdnq_dequantizer.out_features), dtype=out_dtype)


# ---- linked projections inside a compiled graph ------------------------------------------------------------------------------
# Eagerly, layers that consume ONE tensor (to_q / to_k / to_v, every cross-attention to_k / to_v) share a launch through a run-time
# guess that is checked by tensor identity (linear.ProjectionGroup) -- which proves nothing inside a compiled graph.  There the same
# sharing is a DATAFLOW fact: after the graph's common-subexpression elimination those layers are `layer_matmul` nodes with the very
# same (xq, xs) arguments.  `MergeLayerMatmuls` (an Inductor post-grad pass, installed by `enable_compile_grouping()`) rewrites every
# such set into ONE `layer_matmul_group` node -- sdnq_hip_scaled_mm_grouped over the members' own weights, one pass over the quantized
# activation -- whose flat result the members' outputs are sliced from (views, free in Inductor).  Bit-identical to the members alone.


def _matmul_group(handles, mm):
    """(ProjectionGroup of the layers behind `handles`, ready for `mm`) or None when they cannot share a grouped launch."""
    from . import linear as L
    key = tuple(handles)
    # the group lives on its FIRST member (module.__dict__), not in a module-level table: a table would hold the members -- and their
    # quantized weights on the GPU -- alive for the life of the process (model-switching hosts; advisor, round 3).  Member -> group ->
    # members is a reference cycle the garbage collector frees together with the model.
    cache = _layer(key[0]).__dict__.setdefault("_sdnq_compile_groups", {})
    pg = cache.get(key)
    if pg is None or any(m is not _layers.get(h, lambda: None)() for m, h in zip(pg.mods, key)):
        pg = L.ProjectionGroup([_layer(h) for h in key])
        cache[key] = pg
    return pg if pg._operands(mm) else None


@custom_op("sdnq_hip::layer_matmul_group", mutates_args=())
def layer_matmul_group(xq: torch.Tensor, xs: torch.Tensor, handles: List[int], out_dtype: torch.dtype) -> torch.Tensor:
    """layer_matmul of several layers on ONE quantized activation: a flat [M * sum(N_i)] buffer in which layer i's [M, N_i] output is
    the contiguous block starting at element M * sum(N_j, j < i)."""
    from . import linear as L
    mm = ops.MM_I8 if xq.dtype == torch.int8 else ops.MM_FP8
    pg = _matmul_group(handles, mm)
    if pg is not None:
        if L.PREFETCH_NEXT and getattr(pg, "pf_tensors", None):
            L._pf_launch(pg, pg.pf_tensors)
        outs = ops.scaled_mm_grouped(mm, xq, xs, pg.gemm, out_dtype)
        return outs[0]._base if outs[0]._base is not None else torch.cat([o.reshape(-1) for o in outs])
    # the members stopped being groupable after tracing (a parameter moved / changed form): one launch each, same layout
    m = xq.shape[0]
    mods = [_layer(h) for h in handles]
    widths = [mod.sdnq_dequantizer.out_features for mod in mods]
    flat = torch.empty((m * sum(widths),), device=xq.device, dtype=out_dtype)
    start = 0
    for mod, n in zip(mods, widths):
        wq, ws, zp = L._prepare_mm_weights(mod, L._state(mod), mm)
        if zp is not None or L._state(mod).svd_up is not None:
            raise _lib.SdnqHipError("sdnq_hip::layer_matmul_group: a member changed to a form with zero-point / low-rank terms after tracing")
        ops.scaled_mm_into(mm, xq, wq.reshape(n, -1), xs, ws.reshape(-1), L._attr(mod, "bias"), flat[m * start:m * (start + n)].view(m, n), 0)
        start += n
    return flat


@layer_matmul_group.register_fake
def _(xq, xs, handles, out_dtype):
    return xq.new_empty((xq.shape[0] * sum(_layer(h).sdnq_dequantizer.out_features for h in handles),), dtype=out_dtype)


def _groupable(handles) -> bool:
    """Static test at pass time: one K, a common width divisor that is a multiple of 64, all or none with a bias."""
    import math
    mods = [_layer(h) for h in handles]
    dqs = [m.sdnq_dequantizer for m in mods]
    if len({dq.in_features for dq in dqs}) != 1:
        return False
    g = 0
    for dq in dqs:
        g = math.gcd(g, dq.out_features)
    if g % 64:
        return False
    has_bias = [getattr(m, "bias", None) is not None for m in mods]
    return all(has_bias) or not any(has_bias)


def merge_layer_matmuls(graph: "torch.fx.Graph") -> int:
    """Rewrite sets of `sdnq_hip::layer_matmul` nodes that share (xq, xs, out_dtype) into one `layer_matmul_group` node + views.
    Returns the number of launches removed.  Static shapes only (a symbolic row count leaves the graph untouched)."""
    targets = (torch.ops.sdnq_hip.layer_matmul.default, torch.ops.sdnq_hip.layer_matmul)  # post-grad graphs hold the overload
    sets: dict = {}
    names = ("xq", "xs", "handle", "out_dtype")
    import os
    debug = os.environ.get("SDNQ_HIP_DEBUG_PASS", "0") == "1"

    def operands(node):  # positional or keyword form (Inductor's passes may normalise a custom operator's call to keywords)
        vals = dict(zip(names, node.args))
        vals.update(node.kwargs)
        return vals if set(vals) == set(names) else None

    # (1) the row quantization of one value with one configuration is computed once: sdnq_hip::rowquant is a pure operator, but the
    #     graph passes in front of this one leave the per-layer copies in place (measured: to_q / to_k / to_v each kept its own)
    import operator
    rq_targets = (torch.ops.sdnq_hip.rowquant.default, torch.ops.sdnq_hip.rowquant)
    first_rq: dict = {}
    for node in list(graph.nodes):
        if node.op == "call_function" and node.target in rq_targets:
            key = (tuple(node.args), tuple(sorted(node.kwargs.items())))
            canon = first_rq.setdefault(key, node)
            if canon is not node:
                node.replace_all_uses_with(canon)
                graph.erase_node(node)
    for canon in first_rq.values():  # one getitem per output index
        items: dict = {}
        for user in list(canon.users):
            if user.op == "call_function" and user.target is operator.getitem:
                keep = items.setdefault(user.args[1], user)
                if keep is not user:
                    user.replace_all_uses_with(keep)
                    graph.erase_node(user)
    # (2) layer_matmul nodes on one (xq, xs) -> one grouped launch
    for node in graph.nodes:
        if node.op == "call_function" and node.target in targets:
            v = operands(node)
            if debug:
                print("[sdnq merge pass]", node.name, node.args, node.kwargs)
            if v is not None:
                sets.setdefault((v["xq"], v["xs"], v["out_dtype"]), []).append(node)
    removed = 0
    for (xq, xs, out_dtype), nodes in sets.items():
        if len(nodes) < 2 or not isinstance(xq, torch.fx.Node):
            continue
        val = xq.meta.get("val")
        if val is None or not isinstance(val.shape[0], int):
            if debug:
                print("[sdnq merge pass] no static fake value on", xq, type(val))
            continue
        handles = [int(operands(n)["handle"]) for n in nodes]
        try:
            if not _groupable(handles):
                continue
        except _lib.SdnqHipError:
            continue
        # build the group's device-resident unit table NOW (compile time, the ordinary allocator): built lazily inside the first run
        # it would be a live allocation in the CUDA-graph trees' private pool that is no output of the graph -- which they refuse
        if val.device.type == "cuda":
            try:
                if _matmul_group(handles, ops.MM_I8 if val.dtype == torch.int8 else ops.MM_FP8) is None:
                    continue
            except _lib.SdnqHipError:
                continue
        m = int(val.shape[0])
        widths = [_layer(h).sdnq_dequantizer.out_features for h in handles]
        fake_mode = getattr(val, "fake_mode", None)
        with graph.inserting_before(nodes[0]):
            flat = graph.call_function(torch.ops.sdnq_hip.layer_matmul_group.default, (xq, xs, handles, out_dtype))
            pieces, start = [], 0
            for n in widths:
                sl = graph.call_function(torch.ops.aten.slice.Tensor, (flat, 0, m * start, m * (start + n)))
                pieces.append((sl, graph.call_function(torch.ops.aten.view.default, (sl, [m, n]))))
                start += n
        if fake_mode is not None:  # Inductor lowers from the nodes' fake values
            with fake_mode:
                fv = torch.ops.sdnq_hip.layer_matmul_group.default(val, xs.meta["val"], handles, out_dtype)
                flat.meta["val"] = fv
                start = 0
                for (sl, vw), n in zip(pieces, widths):
                    sv = torch.ops.aten.slice.Tensor(fv, 0, m * start, m * (start + n))
                    sl.meta["val"] = sv
                    vw.meta["val"] = torch.ops.aten.view.default(sv, [m, n])
                    start += n
        for node, (_, vw) in zip(nodes, pieces):
            node.replace_all_uses_with(vw)
            graph.erase_node(node)
        removed += len(nodes) - 1
    if removed:
        graph.lint()
    return removed


merge_stats = {"graphs": 0, "launches_removed": 0}


def enable_compile_grouping() -> None:
    """Install `merge_layer_matmuls` as Inductor's post-grad custom pass (torch.compile of a model whose SDNQ layers trace as
    rowquant + layer_matmul).  Idempotent; an already installed foreign pass is chained, not replaced."""
    from torch._inductor import config
    from torch._inductor.custom_graph_pass import CustomGraphPass
    prev = config.post_grad_custom_post_pass
    if isinstance(prev, _MergePass):
        return

    config.post_grad_custom_post_pass = _MergePass(prev)


def _make_pass_class():
    from torch._inductor.custom_graph_pass import CustomGraphPass

    class MergePass(CustomGraphPass):
        def __init__(self, prev=None):
            self.prev = prev

        def __call__(self, graph):
            if self.prev is not None:
                self.prev(graph)
            merge_stats["graphs"] += 1
            merge_stats["launches_removed"] += merge_layer_matmuls(graph)

        def uuid(self):
            return None  # the graphs hold per-process layer handles: never served from Inductor's on-disk cache

    return MergePass


try:
    _MergePass = _make_pass_class()
except Exception:  # noqa: BLE001  (a torch build without Inductor)
    _MergePass = type("_MergePass", (), {})


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
        # Conv1d / Conv2d / Conv3d: output geometry from the module's own attributes (the arithmetic of torch.nn.functional.conv*d;
        # non-zero padding modes pad explicitly by `padding` first, which gives the same extents)
        nd = input.ndim - 2
        t = lambda v: (int(v),) * nd if isinstance(v, int) else tuple(int(e) for e in v)  # noqa: E731
        if isinstance(mod.padding, str):
            raise NotImplementedError("sdnq_hip::layer_forward: string padding modes are not built")
        ks, st, pd, dl = tuple(int(k) for k in dq.original_shape[2:]), t(mod.stride), t(mod.padding), t(mod.dilation)
        spatial = [(int(input.shape[2 + i]) + 2 * pd[i] - dl[i] * (ks[i] - 1) - 1) // st[i] + 1 for i in range(nd)]
        return input.new_empty((input.shape[0], dq.out_features, *spatial))
    return input.new_empty((*input.shape[:-1], dq.out_features))
