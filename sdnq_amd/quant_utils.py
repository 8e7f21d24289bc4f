"""Load-time quantization helpers (host side, torch ops) -- SURVEY 8(f) rank 1.

These run once when a float layer is quantized by this package (benchmarks, tests, users without
pre-quantized checkpoints); they are not on the per-forward hot path.  Semantics follow the reference's
quant_utils.py (get_scale_symmetric :23-24, get_scale_asymmetric :10-19, quantize_weight :28-56,
apply_svdquant :124-141, build_hadamard :145-175, get_hadamard_group_size :212-218).
"""
from __future__ import annotations

import torch

from .common import dtype_dict

_H4 = ((1, 1, 1, -1), (1, 1, -1, 1), (1, -1, 1, 1), (-1, 1, 1, 1))
_H2 = ((1, 1), (1, -1))
_HADAMARD_CACHE: dict = {}


def is_pow2(n: int) -> bool:
    return n > 0 and (n & (n - 1)) == 0


def is_pow4(n: int) -> bool:
    return is_pow2(n) and (n.bit_length() & 1) == 1


def get_hadamard(n: int, dtype: torch.dtype = torch.float32, device=None) -> torch.Tensor:
    """H_n = kron powers of H4 (n a power of 4) or of Sylvester's H2, divided by sqrt(n) in ``dtype``."""
    key = (n, dtype, str(device))
    h = _HADAMARD_CACHE.get(key)
    if h is None:
        if not is_pow2(n):
            raise RuntimeError(f"Hadamard Group Size must be a power of 2 but got {n}.")
        base = torch.tensor(_H4 if is_pow4(n) else _H2, dtype=dtype, device=device)
        h = base
        while h.shape[0] < n:
            h = torch.kron(h, base)
        h = h.div_(n ** 0.5)
        _HADAMARD_CACHE[key] = h
    return h


def get_hadamard_group_size(channel_size: int, group_size: int) -> tuple[bool, int]:
    g = 1
    while g < min(channel_size, group_size):
        g *= 2
    while channel_size % g != 0:
        g //= 2
    return g >= 4, g


def rotate_hadamard(weight: torch.Tensor, group_size: int) -> torch.Tensor:
    """weight.view(..., K/g, g) @ H_g  (load-time, on the float weight)."""
    h = get_hadamard(group_size, dtype=weight.dtype, device=weight.device)
    return torch.matmul(weight.unflatten(-1, (-1, group_size)), h).flatten(-2, -1)


def apply_hadamard(weight: torch.Tensor, group_size: int = 256, is_conv: bool = False):
    """quant_utils.py:222-236.  Conv weights ([C_out, C_in, *kernel], or already flattened for the direct matmul): the group size is
    chosen from dimension 1 and the groups run along the flattened (C_in, kernel) axis."""
    use, g = get_hadamard_group_size(weight.shape[1] if is_conv else weight.shape[-1], group_size)
    if use:
        if is_conv and weight.ndim > 2:
            weight = rotate_hadamard(weight.flatten(1, -1), g).unflatten(-1, tuple(weight.shape[1:]))
        else:
            weight = rotate_hadamard(weight, g)
    return weight, use, g


def apply_svdquant(weight: torch.Tensor, rank: int = 32, steps: int = 8, dtype: torch.dtype | None = None):
    """Split W = svd_up @ svd_down + residual with a randomized low-rank SVD; the residual is what gets quantized."""
    shape = weight.shape
    if weight.ndim > 2:  # conv weights: factor the flattened [C_out, C_in * kernel] matrix (quant_utils.py:126-129)
        weight = weight.flatten(1, -1)
    w = weight.to(torch.float32) if weight.dtype != torch.float64 else weight
    u, s, v = torch.svd_lowrank(w, q=rank, niter=steps)
    svd_up = u * s.unsqueeze(0)
    svd_down = v.t()
    if dtype is not None:
        svd_up, svd_down = svd_up.to(dtype), svd_down.to(dtype)
    residual = w - torch.mm(svd_up, svd_down)
    if len(shape) > 2:
        residual = residual.unflatten(-1, tuple(shape[1:]))
    return residual, svd_up, svd_down


def quantize_weight(weight: torch.Tensor, dim, weights_dtype: str, dtype: torch.dtype | None = None):
    """-> (quantized values in the dtype's torch_dtype, scale, zero_point | None)."""
    ent = dtype_dict[weights_dtype]
    w = weight if weight.dtype == torch.float64 else weight.to(torch.float32)
    if ent["is_unsigned"]:
        lo = torch.amin(w, dim=dim, keepdim=True)
        hi = torch.amax(w, dim=dim, keepdim=True)
        scale = (hi - lo) / (ent["max"] - ent["min"])
        zero_point = lo if ent["min"] == 0 else lo - scale * ent["min"]
        if dtype is not None:
            scale, zero_point = scale.to(dtype), zero_point.to(dtype)
        q = (w - zero_point) / scale
    else:
        scale = torch.amax(w.abs(), dim=dim, keepdim=True) / ent["max"]
        zero_point = None
        if dtype is not None:
            scale = scale.to(dtype)
        q = w / scale
    if ent["is_integer"]:
        q = q.round_()
    else:
        q = q.nan_to_num_()
    q = q.clamp_(ent["min"], ent["max"]).to(ent["torch_dtype"])
    return q, scale, zero_point
