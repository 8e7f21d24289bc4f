"""Quantized attention forward on the HIP kernels.

Host-side mirror of the reference's ``sdnq_triton_atten`` (kernels/triton_atten.py:540-618): same argument names, same
defaults.  Built: the default configuration -- int8 Q.K^T (``matmul_dtype="int8"``), P.V in the value dtype
(``pv_matmul_dtype=None``), ``smooth_k``, optional ``is_causal``, grouped-query heads.  Reference-valid options that are not
built raise ``NotImplementedError`` naming the gap (attention masks, quantized P.V, Hadamard rotation, fp16 accumulation,
the backward outputs).
"""
from __future__ import annotations

import ctypes  # noqa: F401  (the binding itself lives in _lib)

import torch

from . import _lib, ops

_DISABLED = {None, "none", "no", "disabled"}


def quantize_attn(query: torch.Tensor, key: torch.Tensor, value: torch.Tensor, smooth_k: bool = True):
    """quantize_attn (triton_atten.py:443-487) for matmul_dtype="int8", pv_matmul_dtype=None.
    Returns (q_q int8, q_scale f32 [Z,H,QN], k_q int8, k_scale f32 [Z,KH,KN], v_t [Z,KH,D,KN rounded up to 32])."""
    if not query.is_cuda:
        raise _lib.SdnqHipError("sdnq_amd attention needs CUDA/HIP tensors (no CPU fallback)")
    z, qh, qn, d = query.shape
    _, kh, kn, _ = key.shape
    query, key, value = query.contiguous(), key.contiguous(), value.contiguous()
    dev = query.device
    knp = (kn + 31) // 32 * 32
    qq = torch.empty((z, qh, qn, d), device=dev, dtype=torch.int8)
    qs = torch.empty((z, qh, qn), device=dev, dtype=torch.float32)
    kq = torch.empty((z, kh, kn, d), device=dev, dtype=torch.int8)
    ks = torch.empty((z, kh, kn), device=dev, dtype=torch.float32)
    vt = torch.empty((z, kh, d, knp), device=dev, dtype=value.dtype)
    kmean = torch.empty((z, kh, d), device=dev, dtype=torch.float32)
    ops.check(_lib.load().sdnq_hip_attn_prepare(query.data_ptr(), key.data_ptr(), value.data_ptr(), ops.float_code(query.dtype),
                                                z, qh, kh, qn, kn, d, 1 if smooth_k else 0, qq.data_ptr(), qs.data_ptr(),
                                                kq.data_ptr(), ks.data_ptr(), vt.data_ptr(), kmean.data_ptr(),
                                                ops._stream(query)), "attn_prepare")
    return qq, qs, kq, ks, vt


def atten_fwd(qq, qs, kq, ks, vt, kn: int, sm_scale: float, is_causal: bool, out_dtype: torch.dtype) -> torch.Tensor:
    """sdnq_atten_fwd (triton_atten.py:338-385) on the quantized operands of ``quantize_attn``."""
    z, qh, qn, d = qq.shape
    kh = kq.shape[1]
    out = torch.empty((z, qh, qn, d), device=qq.device, dtype=out_dtype)
    ops.check(_lib.load().sdnq_hip_attn_fwd(qq.data_ptr(), qs.data_ptr(), kq.data_ptr(), ks.data_ptr(), vt.data_ptr(),
                                            ops.float_code(vt.dtype), float(sm_scale), 1 if is_causal else 0, out.data_ptr(),
                                            ops.float_code(out_dtype), z, qh, kh, qn, kn, d, ops._stream(qq)), "attn_fwd")
    return out


@torch.no_grad()
def sdnq_hip_atten(query: torch.Tensor, key: torch.Tensor, value: torch.Tensor, attn_mask: torch.Tensor | None = None,
                   dropout_p: float = 0.0, is_causal: bool = False, scale: float | None = None, enable_gqa: bool = False,
                   smooth_k: bool = True, use_hadamard: bool = False, hadamard_group_size: int = 256, matmul_dtype: str = "int8",
                   pv_matmul_dtype: str | None = None, do_quantize: bool = True, use_fp16_accum: bool = False,
                   out_dtype: torch.dtype | None = None, return_backward: bool = False) -> torch.Tensor:
    """Drop-in for ``sdnq_triton_atten(query, key, value, ...)`` (triton_atten.py:540-618); [Z, H, N, D] layout."""
    if attn_mask is not None:
        raise NotImplementedError("attention masks are not built for MI355X (is_causal is)")
    if return_backward:
        raise NotImplementedError("the backward outputs (lse) of the quantized attention are not built for MI355X")
    if use_hadamard:
        raise NotImplementedError("Hadamard-rotated quantized attention is not built for MI355X")
    if use_fp16_accum:
        raise NotImplementedError("use_fp16_accum is an RDNA work-around; the MI355X kernels accumulate in fp32")
    if matmul_dtype in {"auto", "enabled", "uint8"}:  # triton_atten.py:452-453
        matmul_dtype = "int8"
    if not do_quantize or matmul_dtype in _DISABLED or matmul_dtype != "int8":
        raise NotImplementedError(f"quantized attention with matmul_dtype={matmul_dtype!r} is not built (int8 Q.K^T is)")
    if pv_matmul_dtype not in _DISABLED | {"auto"}:
        raise NotImplementedError("quantized P.V (pv_matmul_dtype) is not built for MI355X; P.V runs in the value dtype")
    if query.ndim != 4 or key.ndim != 4 or value.ndim != 4:
        raise ValueError("query / key / value must be [batch, heads, tokens, head_dim]")
    d = query.shape[-1]
    if key.shape[-1] != d or value.shape[-1] != d or d not in (64, 128):
        raise NotImplementedError("head_dim must be 64 or 128 for query, key and value")
    if query.dtype not in (torch.bfloat16, torch.float16) or key.dtype != query.dtype or value.dtype != query.dtype:
        raise NotImplementedError("query / key / value must share one of bfloat16 / float16")
    if out_dtype is None:
        out_dtype = query.dtype
    sm_scale = d ** -0.5 if scale is None else scale  # triton_atten.py:512-513
    qq, qs, kq, ks, vt = quantize_attn(query, key, value, smooth_k=smooth_k)
    return atten_fwd(qq, qs, kq, ks, vt, key.shape[2], sm_scale, is_causal, out_dtype)
