"""Operator seam: ``int_scaled_mm_func`` / ``fp8_scaled_mm_func`` with the reference's signature
(kernel_wrappers.py:193-204; Triton op ``sdnq::scaled_mm`` kernels/triton_scaled_mm.py:239-275):

    out[M,N] = cast( fma( f32(a @ b) * scale_a, scale_b, bias ) )

``a`` is [M,K] row-major; ``b`` is the logical [K,N] operand.  On gfx950 the reference keeps ``b`` with strides
(1,K) (use_contiguous_int8_mm=False, kernel_wrappers.py:96-99) which is exactly the K-contiguous physical
[N,K] layout the MFMA kernel wants; a row-major [K,N] ``b`` is re-laid out like ``check_mats`` does
(layers/linear/forward.py:10-21).  Preconditions mirror the Triton wrapper's asserts (:249-255).
"""
from __future__ import annotations

import torch

from . import _lib, ops

# capability flags with the values the reference auto-selects on gfx950
is_fp8_mm_supported = True
use_tensorwise_fp8_matmul = True
use_contiguous_int8_mm = False
use_contiguous_fp16_mm = True
use_contiguous_fp8_mm = False
use_hip_mm = True


def _b_physical(b: torch.Tensor) -> torch.Tensor:
    bt = b.t()
    return bt if bt.is_contiguous() else bt.contiguous()


def _scaled_mm(mm: int, a, b, scale_a, scale_b, bias, out_dtype):
    assert a.shape[1] == b.shape[0], "Incompatible dimensions"
    assert a.is_contiguous(), "Matrix A must be contiguous"
    assert scale_a.is_contiguous(), "Matrix A scale must be contiguous"
    assert scale_b.is_contiguous(), "Matrix B scale must be contiguous"
    if bias is not None:
        assert bias.is_contiguous(), "Bias must be contiguous"
        assert bias.ndim in {1, 2}, "Bias must be 1D or 2D"
    m, n = a.shape[0], b.shape[1]
    sa = scale_a.to(torch.float32).reshape(-1)
    sb = scale_b.to(torch.float32).reshape(-1)
    if sa.numel() == 1:
        sa = sa.expand(m).contiguous()
    if sb.numel() == 1:
        sb = sb.expand(n).contiguous()
    return ops.scaled_mm(mm, a, _b_physical(b), sa, sb, bias, out_dtype)


@torch.no_grad()
def int_scaled_mm_func(a, b, scale_a, scale_b, bias=None, out_dtype: torch.dtype = torch.float32) -> torch.Tensor:
    if a.dtype != torch.int8 or b.dtype != torch.int8:
        raise _lib.SdnqHipError("int_scaled_mm_func expects int8 operands")
    return _scaled_mm(ops.MM_I8, a, b, scale_a, scale_b, bias, out_dtype)


@torch.no_grad()
def fp8_scaled_mm_func(a, b, scale_a, scale_b, bias=None, out_dtype: torch.dtype = torch.float32) -> torch.Tensor:
    if a.dtype != torch.float8_e4m3fn or b.dtype != torch.float8_e4m3fn:
        raise _lib.SdnqHipError("fp8_scaled_mm_func expects float8_e4m3fn operands")
    return _scaled_mm(ops.MM_FP8, a, b, scale_a, scale_b, bias, out_dtype)
