"""The Linear forwards of the drop-in boundary: ``forward_func(self: SDNQLinear, input) -> Tensor``.

One function per reference forward (same names, same control flow, same numerics contract):

    quantized_linear_forward               layers/linear/forward.py:25-26
    quantized_linear_forward_int8_matmul   layers/linear/linear_int8.py:101-125  (+ :26-97)
    quantized_linear_forward_fp8_matmul    layers/linear/linear_fp8.py:82-104    (+ :26-78)
    quantized_linear_forward_uint8_matmul  layers/linear/linear_uint8.py:106-131

Where the reference runs unpack / scale / Hadamard / SVD / row-quant / zero-point as a chain of eager or
Inductor kernels and then a Triton GEMM, this build issues at most three HIP launches per call:
``rowquant`` (Hadamard + amax + quantize [+ rowsum] fused), [``lowrank_down`` for SVD], and the MFMA
``scaled_mm`` whose epilogue applies scales, bias, the low-rank term and the zero-point term.

Weight-side work that depends only on static tensors (re-quantization of group-wise / packed weights to
int8/fp8, unpacking of 6/7-bit rows, SVD factor layout) is done ONCE and cached on the module -- the
reference redoes it every forward (dequantizer.py:204-239) but the result is identical because the
inputs never change.  ``SDNQ_HIP_CACHE_WEIGHTS=0`` restores the per-call behaviour.
"""
from __future__ import annotations

import os

import torch

from . import ops
from .common import dtype_dict

CACHE_WEIGHTS = os.environ.get("SDNQ_HIP_CACHE_WEIGHTS", "1").lower() not in {"0", "false", "no"}
PREFETCH_WEIGHTS = os.environ.get("SDNQ_HIP_PREFETCH_WEIGHTS", "0").lower() not in {"0", "false", "no"}  # measured: no gain
FUSED_SKINNY = os.environ.get("SDNQ_HIP_FUSED_SKINNY", "1").lower() not in {"0", "false", "no"}
CACHE_ACTIVATIONS = int(os.environ.get("SDNQ_HIP_CACHE_ACTIVATIONS", "12"))  # LRU entries; 0 disables


class _ActivationCache:
    """Quantized activations keyed on the identity of the input tensor.

    In a transformer block several Linear layers consume the SAME tensor object (attn.to_q / to_k / to_v all get
    `hidden_states`; every cross-attention to_k / to_v gets the same `encoder_hidden_states`).  The reference
    re-quantizes it for each of them (linear_int8.py:64); the result is identical every time, so this build keeps the
    last few (xq, xs, ...) tuples and reuses them when the very same tensor object (unchanged `_version`) comes back
    with the same quantization parameters.  Entries hold a strong reference to their input, so its storage cannot be
    recycled for a different tensor while the entry is alive.
    """

    def __init__(self, size: int | None = None):
        self.size = size   # None: the module-level CACHE_ACTIVATIONS, read at use time (the switch can be flipped after import)
        self.entries = []  # most recent last: (tensor, version, params, result)

    def get(self, t: torch.Tensor, params):
        for i in range(len(self.entries) - 1, -1, -1):
            e = self.entries[i]
            if e[0] is t and e[1] == t._version and e[2] == params:
                self.entries.append(self.entries.pop(i))
                return e[3]
        return None

    def put(self, t: torch.Tensor, params, result):
        self.entries.append((t, t._version, params, result))
        cap = max(CACHE_ACTIVATIONS, 0) if self.size is None else self.size
        while len(self.entries) > cap:
            self.entries.pop(0)

    def clear(self):
        self.entries.clear()


_act_cache = _ActivationCache()


def clear_activation_cache():
    _act_cache.clear()


def _rowquant_cached(input: torch.Tensor, k: int, mm: int, had: int, want_rowsum: bool, want_xrot: bool, prefetch, asymmetric=False):
    # the stream is part of the key: an entry produced on one stream is not ordered against work on another
    params = (mm, had, want_rowsum, want_xrot, asymmetric, ops._stream(input) if input.is_cuda else -1)
    if CACHE_ACTIVATIONS > 0:
        hit = _act_cache.get(input, params)
        if hit is not None:
            return hit
    x2 = input.reshape(-1, k)
    if x2.stride(-1) != 1 or (x2.stride(0) * x2.element_size()) % 16:
        x2 = x2.contiguous()
    res = ops.rowquant(x2, mm, had, want_rowsum=want_rowsum, want_xrot=want_xrot, prefetch=prefetch, asymmetric=asymmetric)
    res = (x2,) + tuple(res)
    if CACHE_ACTIVATIONS > 0:
        _act_cache.put(input, params, res)
    return res


class _State:
    """Per-module cache of kernel-ready tensors, keyed on the identity of the module's parameters."""
    __slots__ = ("key", "qw", "mm", "mm_weight", "mm_scale", "mm_zp", "mm_wcs", "svd_up", "svd_down", "svd_down_t", "wd", "bias")


_STATE_FIELDS = ("weight", "scale", "zero_point", "svd_up", "svd_down")


def _attr(mod, name):
    """mod.<name> without nn.Module.__getattr__ (parameters, then buffers, then plain attributes such as None)."""
    d = mod.__dict__
    t = d["_parameters"].get(name, d)
    if t is d:
        t = d.get(name, d)
        if t is d:
            t = getattr(mod, name, None)
    return t


def _signature(mod):
    out = []
    for name in _STATE_FIELDS:
        t = _attr(mod, name)
        out.append((name, t, None if t is None else t.data_ptr(), None if t is None else t._version))
    return tuple(out)


def _state(mod) -> _State:
    """Kernel-ready tensors of a module, rebuilt when a parameter object, its storage or its version changes (the check is on the
    per-forward path of eager models: a few dictionary lookups, no tuple building)."""
    st = mod.__dict__.get("_sdnq_hip_state")
    if st is not None:
        for name, ref, ptr, ver in st.key:
            t = _attr(mod, name)
            if t is not ref or (t is not None and (t.data_ptr() != ptr or t._version != ver)):
                break
        else:
            return st
    dq = mod.sdnq_dequantizer
    st = _State()
    st.key = _signature(mod)
    st.qw = dq.quant_weight(mod.weight, mod.scale, getattr(mod, "zero_point", None), getattr(mod, "svd_up", None),
                            getattr(mod, "svd_down", None))
    st.mm = None
    st.mm_weight = st.mm_scale = st.mm_zp = st.mm_wcs = None
    st.svd_up, st.svd_down = st.qw.keep[3], st.qw.keep[4]  # physical [N,R], [R,K]
    st.svd_down_t = None
    st.wd = None
    mod.__dict__["_sdnq_hip_state"] = st
    return st


def _float_forward(mod, input: torch.Tensor, st: _State) -> torch.Tensor:
    """F.linear(input, dequant(W), bias): dequantize to [N,K] in the result dtype (Hadamard un-rotated, SVD added),
    then a float GEMM with fp32 accumulation."""
    dq = mod.sdnq_dequantizer
    k, n = dq.in_features, dq.out_features
    if input.dtype != dq.result_dtype:
        raise RuntimeError(f"expected input dtype {dq.result_dtype} (the layer's result_dtype) but got {input.dtype}")
    m = input.numel() // input.shape[-1]
    if m == 0:  # empty batch: nothing to launch (F.linear returns an empty [.., N] tensor)
        if not input.is_cuda:
            raise ops._lib.SdnqHipError("sdnq_amd forwards need CUDA/HIP tensors (no CPU fallback)")
        return input.new_empty(*input.shape[:-1], n)
    if FUSED_SKINNY and m <= 32 and st.svd_up is None and k % 16 == 0:
        # few rows (time/AdaLN embeddings, the M < 32 branch): stream the quantized weight once instead of writing and
        # re-reading a dequantized copy; on Hadamard layers the kernel un-rotates each weight run in registers.
        x2 = input.reshape(-1, k)
        if x2.stride(-1) != 1 or (x2.stride(0) * x2.element_size()) % 16:
            x2 = x2.contiguous()
        return ops.linear_skinny(st.qw, x2, mod.bias, dq.hadamard_group_size if dq.use_hadamard else 0).view(*input.shape[:-1], n)
    if (FUSED_SKINNY and m <= 4 and st.svd_up is not None and not dq.use_hadamard and dq.weights_dtype in ("int8", "uint8", "int4", "uint4")
            and (dq.group_size <= 0 or dq.group_size % 4 == 0) and dq.kernel_positions == 1 and k % 32 == 0 and st.svd_up.shape[1] % 16 == 0 and input.dtype in (torch.bfloat16, torch.float16)
            and st.svd_up.dtype == input.dtype and m * k * 4 <= 150 * 1024):
        # int8 + SVD layer with a few rows: W = round(round(q s) + up.down) is formed on the fly (rank product on the matrix cores)
        if st.svd_down_t is None:
            st.svd_down_t = st.svd_down.t().contiguous()  # [K, R]
        x2 = input.reshape(-1, k)
        if x2.stride(-1) != 1:
            x2 = x2.contiguous()
        return ops.linear_skinny_svd(st.qw, st.svd_down_t, x2, mod.bias).view(*input.shape[:-1], n)
    group = mod.__dict__.get("_sdnq_group")
    if group is not None and group[0].float_mode and LINK_PROJECTIONS and m > 32 and input.is_cuda and st.wd is None and n % 8 == 0:
        return group[0].forward_float(mod, group[1], input)
    wd = st.wd
    if wd is None:
        wd = ops.dequant(st.qw, dq.result_dtype, dq.hadamard_group_size if dq.use_hadamard else 0)
        if CACHE_WEIGHTS and os.environ.get("SDNQ_HIP_CACHE_DEQUANT", "0") == "1":
            st.wd = wd
    x2 = input.reshape(-1, k)
    if x2.stride(-1) != 1:
        x2 = x2.contiguous()
    y = ops.linear_float(x2, wd, mod.bias)
    return y.view(*input.shape[:-1], n)


@torch.no_grad()
def quantized_linear_forward(self, input: torch.Tensor) -> torch.Tensor:
    return _float_forward(self, input, _state(self))


def _prepare_mm_weights(mod, st: _State, mm: int, asymmetric: bool = False):
    """Weight operand of the quantized matmul: (wq [N,K], ws [N], zp [N] | None).  `asymmetric` (the uint8 matmul) only changes
    the re-quantizer: min / max range and a zero point per output row."""
    dq = mod.sdnq_dequantizer
    key = (mm, asymmetric)
    if st.mm == key and st.mm_weight is not None:
        return st.mm_weight, st.mm_scale, st.mm_zp
    zp = None
    if dq.re_quantize_for_matmul and asymmetric:
        wq, ws, zp = ops.requant_asym(st.qw)  # linear_uint8.py:109-111
    elif dq.re_quantize_for_matmul:
        wq, ws = ops.requant(st.qw, mm)  # linear_int8.py:104-107; zero_point folded, none afterwards
    else:
        ws = st.qw.keep[1]  # row-wise scale [N]
        ent = dtype_dict[dq.weights_dtype]
        plain = (not ent["is_packed"]) and ((mm == ops.MM_I8 and dq.weights_dtype == "int8")
                                            or (mm == ops.MM_FP8 and ent["torch_dtype"] == torch.float8_e4m3fn))
        if plain:
            wq = st.qw.keep[0]  # already the physical [N,K] operand
            if wq.dtype == torch.uint8:
                wq = wq.view(torch.int8)
        else:
            wq = ops.unpack_mm(st.qw, mm)  # linear_int8.py:38-50 / linear_fp8.py:36-38
        if mm == ops.MM_I8 and ent["is_unsigned"]:
            zp = st.qw.keep[2]
            if not ent["is_packed"]:  # plain uint8: zero_point += 128 * scale (linear_int8.py:47-50)
                zp = torch.add(zp, ws, alpha=128) if zp is not None else ws * 128
    if CACHE_WEIGHTS:
        st.mm, st.mm_weight, st.mm_scale, st.mm_zp, st.mm_wcs = key, wq, ws, zp, None
    return wq, ws, zp


class ProjectionGroup:
    """Layers of one attention block that consume the SAME tensor (to_q / to_k / to_v of self-attention, to_k / to_v of
    cross-attention) and have equal shapes and a row-wise direct-matmul configuration (``loader._fusable``).  The first member
    called with a tensor runs ONE scaled matmul over the stacked weights (``sdnq_hip_scaled_mm_multi``: one launch and one pass
    over the quantized activation instead of one per layer) writing each member's output into its own contiguous tensor; the
    other members, called with the very same tensor object, just pick theirs up.  Every output element is bit-identical to what
    the member computes alone (each output channel keeps its own scale and bias).  The stacked operand is a copy of the members'
    weights (the members themselves, their parameters and the state_dict are untouched)."""

    def __init__(self, mods, float_mode: bool = False):
        self.mods = list(mods)
        self.float_mode = float_mode  # members run dequantize + F.linear (use_quantized_matmul=False) instead of the quantized matmul
        self.sig = None   # (matmul dtype, the members' weight / scale objects and versions the stacked operands were built from)
        self.wq = self.ws = self.bias = None
        self.last = None  # (input tensor, its version, stream, outputs, indices not handed out yet)

    def _operands(self, mm):
        # per-compute check on the eager path: the members' weight / scale objects and versions (a changed parameter rebuilds the
        # stacked operands); the full per-module signature is only evaluated when something changed
        refs = self.sig
        if refs is not None and refs[0] == mm:
            for m, (w, wv, sc, sv) in zip(self.mods, refs[1]):
                if _attr(m, "weight") is not w or w._version != wv or _attr(m, "scale") is not sc or sc._version != sv:
                    break
            else:
                return self.wq is not None
        states = [_state(m) for m in self.mods]
        parts = [_prepare_mm_weights(m, st, mm) for m, st in zip(self.mods, states)]
        self.sig = (mm, [(_attr(m, "weight"), _attr(m, "weight")._version, _attr(m, "scale"), _attr(m, "scale")._version) for m in self.mods])
        self.last = None
        if any(zp is not None for (_, _, zp) in parts):
            self.wq = None
            return False
        self.wq = torch.cat([wq.reshape(wq.shape[0], -1) for (wq, _, _) in parts], dim=0).contiguous()
        self.ws = torch.cat([ws.reshape(-1) for (_, ws, _) in parts], dim=0).contiguous()
        biases = [_attr(m, "bias") for m in self.mods]
        self.bias = None if biases[0] is None else torch.cat(biases, dim=0).contiguous()
        return True

    def forward(self, mod, idx: int, input: torch.Tensor, mm: int):
        stream = ops._stream(input)
        last = self.last
        if last is None or last[0] is not input or last[1] != input._version or last[2] != stream or idx not in last[4]:
            if not self._operands(mm):
                return None
            x2, xq, xs, _, _ = _rowquant_cached(input, input.shape[-1], mm, 0, False, False, None)
            outs = ops.scaled_mm_multi(mm, xq, self.wq, xs, self.ws, self.bias, input.dtype, len(self.mods))
            last = self.last = (input, input._version, stream, outs, set(range(len(self.mods))))
        y = last[3][idx].view(*input.shape[:-1], -1)
        last[4].discard(idx)
        if not last[4]:
            self.last = None  # every member has its output: hold on to nothing (the input and the outputs belong to the host again)
        return y


    def forward_float(self, mod, idx: int, input: torch.Tensor):
        """The dequantize + F.linear mode (use_quantized_matmul=False, M > 32): every member is dequantized into its slab of ONE
        [sum N][K] buffer (as many dequantize launches as before), then one float GEMM writes the members' outputs."""
        stream = ops._stream(input)
        last = self.last
        if last is None or last[0] is not input or last[1] != input._version or last[2] != stream or idx not in last[4]:
            dq = mod.sdnq_dequantizer
            k, n = dq.in_features, dq.out_features
            x2 = input.reshape(-1, k)
            if x2.stride(-1) != 1 or (x2.stride(0) * x2.element_size()) % 16:
                x2 = x2.contiguous()
            g = len(self.mods)
            wd = torch.empty((g * n, k), device=input.device, dtype=input.dtype)
            for i, m in enumerate(self.mods):
                d = m.sdnq_dequantizer
                ops.dequant(_state(m).qw, input.dtype, d.hadamard_group_size if d.use_hadamard else 0, out=wd[i * n:(i + 1) * n])
            biases = [_attr(m, "bias") for m in self.mods]
            bias = None if biases[0] is None else torch.cat(biases, dim=0)
            outs = ops.linear_float_multi(x2, wd, bias, g)
            last = self.last = (input, input._version, stream, outs, set(range(g)))
        y = last[3][idx].view(*input.shape[:-1], -1)
        last[4].discard(idx)
        if not last[4]:
            self.last = None
        return y


LINK_PROJECTIONS = os.environ.get("SDNQ_HIP_LINK_PROJECTIONS", "1").lower() not in {"0", "false", "no"}


def _quantized_matmul_forward(self, input: torch.Tensor, mm: int, small_batch_branch: bool = True) -> torch.Tensor:
    dq = self.sdnq_dequantizer
    st = _state(self)
    k, n = dq.in_features, dq.out_features
    m = input.numel() // input.shape[-1]
    if m == 0 or (small_batch_branch and m < 32):  # linear_int8.py:102-103: small batches take the dequant + float GEMM branch
        return _float_forward(self, input, st)
    group = self.__dict__.get("_sdnq_group")
    if group is not None and not group[0].float_mode and LINK_PROJECTIONS and input.is_cuda:
        y = group[0].forward(self, group[1], input, mm)
        if y is not None:
            return y
    wq, ws, zp = _prepare_mm_weights(self, st, mm)
    had = dq.hadamard_group_size if dq.use_hadamard else 0
    has_svd = st.svd_up is not None
    bias = _attr(self, "bias")
    if not has_svd and zp is None and (had == 0 or k <= 5120):
        # plain w8a8 layer: on a cache miss the row quantization and the GEMM go through ONE binding call (an eager model is
        # bound by the host-side cost per layer); the quantized activation still lands in the cache for sibling layers
        params = (mm, had, False, False, False, ops._stream(input) if input.is_cuda else -1)
        hit = _act_cache.get(input, params) if CACHE_ACTIVATIONS > 0 else None
        if hit is None:
            x2 = input.reshape(-1, k)
            if x2.stride(-1) != 1 or (x2.stride(0) * x2.element_size()) % 16:
                x2 = x2.contiguous()
            if not x2.is_cuda:
                raise ops._lib.SdnqHipError("sdnq_amd forwards need CUDA/HIP tensors (no CPU fallback)")
            y, xq, xs = ops.linear_w8a8(mm, x2, wq, ws, bias, input.dtype, had)
            if CACHE_ACTIVATIONS > 0:
                _act_cache.put(input, params, (x2, xq, xs, None, None))
            return y.view(*input.shape[:-1], n)
        x2, xq, xs, rowsum, xrot = hit
        return ops.scaled_mm(mm, xq, wq, xs, ws, bias, input.dtype).view(*input.shape[:-1], n)
    x2, xq, xs, rowsum, xrot = _rowquant_cached(input, k, mm, had, zp is not None, has_svd, wq if PREFETCH_WEIGHTS else None)
    if has_svd or zp is not None:
        t = None
        if has_svd:  # mm(x, svd_down) of addmm(bias, mm(x, svd_down), svd_up), linear_int8.py:57-62
            t = ops.lowrank_down(xrot if xrot is not None else x2, st.svd_down)
        y = ops.scaled_mm_lowrank(mm, xq, wq, xs, ws, bias, t, st.svd_up, rowsum, zp, input.dtype)
    else:
        y = ops.scaled_mm(mm, xq, wq, xs, ws, bias, input.dtype)
    return y.view(*input.shape[:-1], n)


@torch.no_grad()
def quantized_linear_forward_int8_matmul(self, input: torch.Tensor) -> torch.Tensor:
    return _quantized_matmul_forward(self, input, ops.MM_I8)


@torch.no_grad()
def quantized_linear_forward_fp8_matmul(self, input: torch.Tensor) -> torch.Tensor:
    return _quantized_matmul_forward(self, input, ops.MM_FP8)


def _uint8_matmul_forward(self, input: torch.Tensor, small_batch_branch: bool = True) -> torch.Tensor:
    """Asymmetric-activation int8 matmul (layers/linear/linear_uint8.py:106-131): activations get a per-row zero point,
    the three cross terms of (x - xzp)(w - wzp) are added in the GEMM epilogue instead of a materialised [M,N] bias."""
    dq = self.sdnq_dequantizer
    st = _state(self)
    k, n = dq.in_features, dq.out_features
    m = input.numel() // input.shape[-1]
    if m == 0 or (small_batch_branch and m < 32):
        return _float_forward(self, input, st)
    wq, ws, zp = _prepare_mm_weights(self, st, ops.MM_I8, asymmetric=True)
    wcs = st.mm_wcs
    if wcs is None:  # f32(sum_k wq[n][k]) * ws[n]: static per layer (linear_uint8.py:63 computes it every call)
        wcs = wq.to(torch.int32).sum(dim=1).to(torch.float32).mul_(ws)
        if CACHE_WEIGHTS:
            st.mm_wcs = wcs
    had = dq.hadamard_group_size if dq.use_hadamard else 0
    has_svd = st.svd_up is not None
    x2, xq, xs, rowsum, xrot, xzp = _rowquant_cached(input, k, ops.MM_I8, had, zp is not None, has_svd,
                                                     wq if PREFETCH_WEIGHTS else None, asymmetric=True)
    t = ops.lowrank_down(xrot if xrot is not None else x2, st.svd_down) if has_svd else None
    y = ops.scaled_mm_lowrank(ops.MM_I8, xq, wq, xs, ws, self.bias, t, st.svd_up, rowsum, zp, input.dtype, a_zp=xzp,
                              w_colsum_scaled=wcs)
    return y.view(*input.shape[:-1], n)


@torch.no_grad()
def quantized_linear_forward_uint8_matmul(self, input: torch.Tensor) -> torch.Tensor:
    return _uint8_matmul_forward(self, input)


@torch.no_grad()
def quantized_linear_forward_fp16_matmul(self, input: torch.Tensor) -> torch.Tensor:
    raise NotImplementedError("quantized_matmul_dtype='float16' is outside the MI355X hot path (SURVEY 8a note)")
