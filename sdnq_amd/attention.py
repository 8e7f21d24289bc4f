"""Quantized attention forward on the HIP kernels.

Host-side mirror of the reference's ``sdnq_triton_atten`` (kernels/triton_atten.py:540-618): same argument names, same
defaults.  Built: the default configuration -- int8 Q.K^T (``matmul_dtype="int8"``), P.V in the value dtype
(``pv_matmul_dtype=None``), ``smooth_k``, optional ``use_hadamard``, ``is_causal`` and ``attn_mask`` (bool or additive), grouped-query heads -- on its
tuned kernels, and (round 6) the other matmul formats -- fp8 (e4m3) Q.K^T, P.V on int8 / fp8 / float16 codes with P quantized per (query,
32-key block) -- on a plain kernel of their own (``sdnq_hip_attn_prepare_ex`` / ``sdnq_hip_attn_fwd_ex``).  Reference-valid options that are not
built raise ``NotImplementedError`` naming the gap (fp16 accumulation -- an RDNA work-around --, the backward outputs).
"""
from __future__ import annotations

import ctypes
import math
import os

import torch

from . import _lib, ops

_DISABLED = {None, "none", "no", "disabled"}
_Q16 = os.environ.get("SDNQ_HIP_ATTN_Q16", "1") != "0"


def _rows16(t: torch.Tensor) -> torch.Tensor:
    """[Z,H,N,D] views whose head_dim is contiguous and whose other strides keep rows 16-byte aligned are used as they are (the
    transposed view of a [Z,N,H*D] projection output is the common case); anything else is copied like the reference does."""
    ok = t.stride(-1) == 1 and all(st % 8 == 0 for st in t.stride()[:-1]) and t.data_ptr() % 16 == 0
    return t if ok else t.contiguous()


def _strides(t: torch.Tensor):
    return (ctypes.c_int64 * 3)(*t.stride()[:3])


def _mm_name(matmul_dtype, pv: bool = False):
    """The reference's spellings (triton_atten.py:452-455) -> "int8" | "fp8" | "float16" | None (pv only: P.V in the value dtype)."""
    if pv and matmul_dtype in _DISABLED | {"auto"}:
        return None
    if matmul_dtype in ({"enabled", "uint8", "int8"} if pv else {"auto", "enabled", "uint8", "int8"}):
        return "int8"
    if matmul_dtype in {"fp8", "float8_e4m3fn"}:
        return "fp8"
    if pv and matmul_dtype == "float16":
        return "float16"
    raise NotImplementedError(f"quantized attention with {'pv_' if pv else ''}matmul_dtype={matmul_dtype!r} is not built "
                              f"(int8 and fp8{', float16' if pv else ''} are)")


_MM_CODE = {"int8": _lib.MM_I8, "fp8": _lib.MM_FP8, "float16": _lib.MM_F16, None: -1}


def quantize_attn_ex(query: torch.Tensor, key: torch.Tensor, value: torch.Tensor, smooth_k: bool = True, hadamard_group: int = 0,
                     matmul_dtype: str = "int8", pv_matmul_dtype: str | None = None):
    """quantize_attn (triton_atten.py:443-487) for any built (matmul_dtype, pv_matmul_dtype): (q_q, q_scale, k_q, k_scale, v_op, v_scale | None).
    q_q [Z,H,QN,D] and k_q (fragment order, see ``unpack_k_fragments``) are uint8 tensors holding int8 or e4m3 codes; v_op is the P.V operand:
    the value-dtype / float16-code tiles of ``unpack_v_fragments`` or, for int8 / fp8, uint8 tiles [Z,KH,B,D/32,64,16] (``unpack_v8_fragments``)."""
    mm, pv = _mm_name(matmul_dtype), _mm_name(pv_matmul_dtype, pv=True)
    if not query.is_cuda:
        raise _lib.SdnqHipError("sdnq_amd attention needs CUDA/HIP tensors (no CPU fallback)")
    z, qh, qn, d = query.shape
    _, kh, kn, _ = key.shape
    query, key, value = _rows16(query), _rows16(key), _rows16(value)
    dev = query.device
    knp = (kn + 31) // 32 * 32
    d_src, d = d, (64 if d <= 64 else 128)
    v_shape = ((z, kh, knp // 32, d // 32, 64, 16), torch.uint8) if pv in ("int8", "fp8") else \
        ((z, kh, knp // 32, d // 32, 2, 64, 8), torch.float16 if pv == "float16" else value.dtype)
    shapes = (((z, qh, qn, d), torch.uint8), ((z, qh, qn), torch.float32), ((z, kh, knp // 32, d // 32, 64, 16), torch.uint8),
              ((z, kh, knp), torch.float32), v_shape, ((z, kh, knp), torch.float32), ((z, kh, d), torch.float32))
    sizes = [-(-(math.prod(shp) * dt.itemsize) // 256) * 256 for shp, dt in shapes]
    pool = torch.empty((sum(sizes),), device=dev, dtype=torch.uint8)
    parts, off = [], 0
    for (shp, dt), nbytes in zip(shapes, sizes):
        parts.append(pool[off:off + math.prod(shp) * dt.itemsize].view(dt).view(shp))
        off += nbytes
    qq, qs, kq, ks, vt, vs, kmean = parts
    ops.check(_lib.load().sdnq_hip_attn_prepare_ex(query.data_ptr(), key.data_ptr(), value.data_ptr(), ops.float_code(query.dtype), z, qh, kh, qn, kn,
                                                   d_src, 1 if smooth_k else 0, hadamard_group, _strides(query), _strides(key), _strides(value),
                                                   _MM_CODE[mm], _MM_CODE[pv], qq.data_ptr(), qs.data_ptr(), kq.data_ptr(), ks.data_ptr(),
                                                   vt.data_ptr(), vs.data_ptr(), kmean.data_ptr(), ops._stream(query)), "attn_prepare_ex")
    return qq, qs, kq, ks, vt, (vs if pv is not None else None)


def unpack_v8_fragments(vf: torch.Tensor) -> torch.Tensor:
    """8-bit V operand tiles [Z,KH,B,D/32,64,16] -> codes [Z,KH,B*32,D] (lane = g*32 + ql holds, as byte j, key 16 (j >> 3) + 8 g + (j & 7) of
    channel 32dd + ql)."""
    z, kh, nb, kk = vf.shape[:4]
    t = vf.view(z, kh, nb, kk, 2, 32, 2, 8)  # [.., dd, g, ql, c, j8]
    return t.permute(0, 1, 2, 6, 4, 7, 3, 5).reshape(z, kh, nb * 32, kk * 32)  # key = 16c + 8g + j8, channel = 32dd + ql


def atten_fwd_ex(qq, qs, kq, ks, vt, vs, kn: int, sm_scale: float, is_causal: bool, out_dtype: torch.dtype, v_dtype: torch.dtype,
                 matmul_dtype: str = "int8", pv_matmul_dtype: str | None = None, attn_mask: torch.Tensor | None = None,
                 token_major: bool = False, head_dim: int | None = None) -> torch.Tensor:
    """sdnq_atten_fwd (triton_atten.py:338-385) on the operands of ``quantize_attn_ex``."""
    mm, pv = _mm_name(matmul_dtype), _mm_name(pv_matmul_dtype, pv=True)
    z, qh, qn, d = qq.shape
    kh = kq.shape[1]
    d = head_dim or d
    if token_major:
        out = torch.empty((z, qn, qh, d), device=qq.device, dtype=out_dtype).transpose(1, 2)
    else:
        out = torch.empty((z, qh, qn, d), device=qq.device, dtype=out_dtype)
    mptr, mdt, ms = None, 0, (0, 0, 0)
    if attn_mask is not None:
        mptr = attn_mask.data_ptr()
        mdt = -1 if attn_mask.dtype == torch.int8 else ops.float_code(attn_mask.dtype)
        ms = tuple(attn_mask.stride(i) if attn_mask.shape[i] != 1 else 0 for i in range(3))
    ops.check(_lib.load().sdnq_hip_attn_fwd_ex(qq.data_ptr(), qs.data_ptr(), kq.data_ptr(), ks.data_ptr(), vt.data_ptr(), None if vs is None else vs.data_ptr(),
                                               ops.float_code(v_dtype), _MM_CODE[mm], _MM_CODE[pv], float(sm_scale), 1 if is_causal else 0, mptr, mdt, *ms,
                                               out.data_ptr(), ops.float_code(out_dtype), _strides(out), z, qh, kh, qn, kn, d, ops._stream(qq)), "attn_fwd_ex")
    return out


def quantize_attn(query: torch.Tensor, key: torch.Tensor, value: torch.Tensor, smooth_k: bool = True, hadamard_group: int = 0,
                  with_query: bool = True):
    """quantize_attn (triton_atten.py:443-487) for matmul_dtype="int8", pv_matmul_dtype=None.
    Returns (q_q int8 [Z,H,QN,D], q_scale f32 [Z,H,QN], k_q, k_scale f32 [Z,KH,KNp], v_f) with KNp = KN rounded up to 32.
    k_q [Z,KH,KNp/32,D/32,64,16] int8 and v_f [Z,KH,KNp/32,D/32,2,64,8] are the K / V operands in MFMA-fragment order (see
    ``unpack_k_fragments`` / ``unpack_v_fragments``); tokens past KN are zero.
    ``with_query=False``: K and V only -- (query as the kernels will read it, None, k_q, k_scale, v_f) -- for ``atten_fwd``'s
    quantize-in-the-forward-kernel route (same codes and scales, one pass over Q less; not with a Hadamard rotation)."""
    if not with_query and hadamard_group:
        raise ValueError("the Hadamard rotation of Q runs in the prepare pass: with_query=False needs hadamard_group=0")
    if not query.is_cuda:
        raise _lib.SdnqHipError("sdnq_amd attention needs CUDA/HIP tensors (no CPU fallback)")
    z, qh, qn, d = query.shape
    _, kh, kn, _ = key.shape
    query, key, value = _rows16(query), _rows16(key), _rows16(value)  # strided views are read in place (no .contiguous() copy)
    dev = query.device
    knp = (kn + 31) // 32 * 32
    d_src, d = d, (64 if d <= 64 else 128)  # head dims below 64 / 128 are zero-padded inside the kernels
    # one allocation for the five operands + the K-mean workspace (an eager host pays per allocation), 256-byte aligned slices
    shapes = (((z, qh, qn if with_query else 0, d), torch.int8), ((z, qh, qn if with_query else 0), torch.float32), ((z, kh, knp // 32, d // 32, 64, 16), torch.int8),
              ((z, kh, knp), torch.float32), ((z, kh, knp // 32, d // 32, 2, 64, 8), value.dtype),
              ((z, kh, 32, d), torch.float32))  # last: channel sums of 32 token splits
    sizes = [-(-(math.prod(shp) * dt.itemsize) // 256) * 256 for shp, dt in shapes]
    pool = torch.empty((sum(sizes),), device=dev, dtype=torch.uint8)
    parts, off = [], 0
    for (shp, dt), nbytes in zip(shapes, sizes):
        parts.append(pool[off:off + math.prod(shp) * dt.itemsize].view(dt).view(shp))
        off += nbytes
    qq, qs, kq, ks, vt, kmean = parts
    ops.check(_lib.load().sdnq_hip_attn_prepare(query.data_ptr(), key.data_ptr(), value.data_ptr(), ops.float_code(query.dtype),
                                                z, qh, kh, qn, kn, d_src, 1 if smooth_k else 0, hadamard_group, _strides(query), _strides(key),
                                                _strides(value), qq.data_ptr() if with_query else None, qs.data_ptr() if with_query else None,
                                                kq.data_ptr(), ks.data_ptr(), vt.data_ptr(), kmean.data_ptr(),
                                                ops._stream(query)), "attn_prepare")
    if not with_query:
        return query, None, kq, ks, vt
    return qq, qs, kq, ks, vt


_PI = [(n & 0x13) | ((n & 4) << 1) | ((n & 8) >> 1) for n in range(32)]  # fragment row <-> key inside a 32-key block


def unpack_k_fragments(kq: torch.Tensor) -> torch.Tensor:
    """Fragment-order K codes [Z,KH,B,D/32,64,16] -> [Z,KH,B*32,D] (lane = g*32 + rho holds bytes [32kk+16g, +16) of key pi(rho))."""
    z, kh, nb, kk = kq.shape[:4]
    t = kq.view(z, kh, nb, kk, 2, 32, 16)[:, :, :, :, :, _PI]  # rho -> key order
    return t.permute(0, 1, 2, 5, 3, 4, 6).reshape(z, kh, nb * 32, kk * 32)


def unpack_v_fragments(vf: torch.Tensor) -> torch.Tensor:
    """Fragment-order V [Z,KH,B,D/32,2,64,8] -> [Z,KH,B*32,D] (lane = g*32 + ql holds keys 16c + 8g + 0..7 of channel 32dd + ql)."""
    z, kh, nb, kk = vf.shape[:4]
    t = vf.view(z, kh, nb, kk, 2, 2, 32, 8)  # [.., dd, c, g, ql, j]
    return t.permute(0, 1, 2, 4, 5, 7, 3, 6).reshape(z, kh, nb * 32, kk * 32)


def prepare_mask(attn_mask: torch.Tensor, qn: int, kn: int) -> torch.Tensor:
    """The mask normalisation of get_attn_inputs (triton_atten.py:520-527): bool -> int8, left-pad to 4-D, a trailing 1 expands
    to the key count, contiguous."""
    if attn_mask.dtype == torch.bool:
        attn_mask = attn_mask.to(dtype=torch.int8)
    while attn_mask.ndim < 4:
        attn_mask = attn_mask.unsqueeze(0)
    if attn_mask.shape[-1] == 1:
        attn_mask = attn_mask.expand(-1, -1, -1, kn)
    attn_mask = attn_mask.contiguous()
    if attn_mask.shape[-1] != kn or attn_mask.shape[-2] not in (1, qn):
        raise ValueError(f"attention mask of shape {tuple(attn_mask.shape)} does not match {qn} queries x {kn} keys")
    if attn_mask.dtype not in (torch.int8, torch.float32, torch.bfloat16, torch.float16):
        raise NotImplementedError(f"attention mask dtype {attn_mask.dtype} (bool / int8 / float32 / bfloat16 / float16 are built)")
    return attn_mask


def atten_fwd(qq, qs, kq, ks, vt, kn: int, sm_scale: float, is_causal: bool, out_dtype: torch.dtype,
              attn_mask: torch.Tensor | None = None, token_major: bool = False, head_dim: int | None = None) -> torch.Tensor:
    """sdnq_atten_fwd (triton_atten.py:338-385) on the quantized operands of ``quantize_attn``; ``attn_mask`` as ``prepare_mask``
    returns it (4-D, contiguous; size-1 dimensions broadcast, triton_atten.py:371-378).  ``qs is None``: ``qq`` is the query in the
    value dtype (``quantize_attn(with_query=False)``) and the forward kernel quantizes it."""
    z, qh, qn, d = qq.shape
    kh = kq.shape[1]
    d = head_dim or d  # qq holds the padded head dim; the output has the tensors' own
    if token_major:  # memory [Z, N, H, D], returned as its [Z, H, N, D] view: out.transpose(1, 2).reshape(Z, N, H*D) is then free
        out = torch.empty((z, qn, qh, d), device=qq.device, dtype=out_dtype).transpose(1, 2)
    else:
        out = torch.empty((z, qh, qn, d), device=qq.device, dtype=out_dtype)
    mptr, mdt, ms = None, 0, (0, 0, 0)
    if attn_mask is not None:
        mptr = attn_mask.data_ptr()
        mdt = -1 if attn_mask.dtype == torch.int8 else ops.float_code(attn_mask.dtype)
        ms = tuple(attn_mask.stride(i) if attn_mask.shape[i] != 1 else 0 for i in range(3))
    tail = (kq.data_ptr(), ks.data_ptr(), vt.data_ptr(), ops.float_code(vt.dtype), float(sm_scale), 1 if is_causal else 0, mptr, mdt, *ms,
            out.data_ptr(), ops.float_code(out_dtype), _strides(out), z, qh, kh, qn, kn, d, ops._stream(qq))
    if qs is None:
        if qq.dtype != vt.dtype:
            raise ValueError("query and value dtypes differ")
        ops.check(_lib.load().sdnq_hip_attn_fwd_q16(qq.data_ptr(), _strides(qq), *tail), "attn_fwd_q16")
    else:
        ops.check(_lib.load().sdnq_hip_attn_fwd(qq.data_ptr(), qs.data_ptr(), *tail), "attn_fwd")
    return out


@torch.no_grad()
def sdnq_hip_atten(query: torch.Tensor, key: torch.Tensor, value: torch.Tensor, attn_mask: torch.Tensor | None = None,
                   dropout_p: float = 0.0, is_causal: bool = False, scale: float | None = None, enable_gqa: bool = False,
                   smooth_k: bool = True, use_hadamard: bool = False, hadamard_group_size: int = 256, matmul_dtype: str = "int8",
                   pv_matmul_dtype: str | None = None, do_quantize: bool = True, use_fp16_accum: bool = False,
                   out_dtype: torch.dtype | None = None, return_backward: bool = False) -> torch.Tensor:
    """Drop-in for ``sdnq_triton_atten(query, key, value, ...)`` (triton_atten.py:540-618); [Z, H, N, D] layout."""
    if return_backward:
        raise NotImplementedError("the backward outputs (lse) of the quantized attention are not built for MI355X")
    if use_fp16_accum:
        raise NotImplementedError("use_fp16_accum is an RDNA work-around; the MI355X kernels accumulate in fp32")
    if not do_quantize or matmul_dtype in _DISABLED:
        raise NotImplementedError("the unquantized attention (do_quantize=False / matmul_dtype disabled) is torch's SDPA, not an SDNQ kernel")
    mm, pv = _mm_name(matmul_dtype), _mm_name(pv_matmul_dtype, pv=True)  # triton_atten.py:452-455; unknown formats raise
    if query.ndim != 4 or key.ndim != 4 or value.ndim != 4:
        raise ValueError("query / key / value must be [batch, heads, tokens, head_dim]")
    d = query.shape[-1]
    if key.shape[-1] != d or value.shape[-1] != d or d % 8 or not 8 <= d <= 128:
        raise NotImplementedError("query, key and value must share a head_dim that is a multiple of 8 and at most 128")
    if query.dtype not in (torch.bfloat16, torch.float16) or key.dtype != query.dtype or value.dtype != query.dtype:
        raise NotImplementedError("query / key / value must share one of bfloat16 / float16")
    if out_dtype is None:
        out_dtype = query.dtype
    sm_scale = d ** -0.5 if scale is None else scale  # triton_atten.py:512-513
    group = 0
    if use_hadamard:  # triton_atten.py:563-569: group from the (power-of-two) head dim, halved until it divides
        from .quant_utils import get_hadamard_group_size
        dp = 64 if d <= 64 else 128
        use_hadamard, group = get_hadamard_group_size(dp, min(hadamard_group_size, dp))
        group = group if use_hadamard else 0
    if not query.is_cuda or (attn_mask is not None and not attn_mask.is_cuda):
        raise _lib.SdnqHipError("sdnq_amd attention needs CUDA/HIP tensors (no CPU fallback)")
    if attn_mask is not None:
        attn_mask = prepare_mask(attn_mask, query.shape[2], key.shape[2])
    # like torch's SDPA, the output takes the memory layout of the query: a [Z,N,H,D]-backed query (the transposed view a
    # diffusers / transformers attention processor passes) gets a [Z,N,H,D]-backed output
    token_major = query.shape[1] > 1 and query.shape[2] > 1 and query.stride(2) > query.stride(1) and query.stride(-1) == 1
    if mm != "int8" or pv is not None:  # fp8 Q.K^T / quantized P.V (round 6): prepare + forward on the kernels of their own
        if pv is not None and group and d not in (64, 128):
            raise NotImplementedError("quantized P.V under a Hadamard rotation is built for head dims 64 and 128 (the output is rotated back over the padded head dim)")
        qq, qs, kq, ks, vt, vs = quantize_attn_ex(query, key, value, smooth_k=smooth_k, hadamard_group=group, matmul_dtype=mm, pv_matmul_dtype=pv)
        out = atten_fwd_ex(qq, qs, kq, ks, vt, vs, key.shape[2], sm_scale, is_causal, out_dtype, query.dtype, mm, pv, attn_mask,
                           token_major=token_major and not (pv is not None and group), head_dim=d)
        if pv is not None and group:  # V was rotated: the output comes back out of the rotated basis (triton_atten.py:609-612)
            out = ops.hadamard(out, group)
        return out
    if not _Q16:  # SDNQ_HIP_ATTN_Q16=0 (A/B aid): the three-step sequence with the separate pass over Q
        qq, qs, kq, ks, vt = quantize_attn(query, key, value, smooth_k=smooth_k, hadamard_group=group)
        return atten_fwd(qq, qs, kq, ks, vt, key.shape[2], sm_scale, is_causal, out_dtype, attn_mask, token_major=token_major, head_dim=d)
    # one C call (sdnq_hip_attn): a single launch for up to 128 keys, else K / V prepared into a workspace and Q quantized by the
    # forward kernel (under a rotation: in the prepare pass)
    z, qh, qn, _ = query.shape
    kh, kn = key.shape[1], key.shape[2]
    query, key, value = _rows16(query), _rows16(key), _rows16(value)
    lib = _lib.load()
    nbytes = lib.sdnq_hip_attn_workspace_bytes(z, qh, kh, qn, kn, d, group)
    if nbytes < 0:
        ops.check(int(nbytes), "attn_workspace_bytes")
    ws = torch.empty((nbytes,), device=query.device, dtype=torch.uint8) if nbytes else None
    if token_major:
        out = torch.empty((z, qn, qh, d), device=query.device, dtype=out_dtype).transpose(1, 2)
    else:
        out = torch.empty((z, qh, qn, d), device=query.device, dtype=out_dtype)
    mptr, mdt, ms = None, 0, (0, 0, 0)
    if attn_mask is not None:
        mptr = attn_mask.data_ptr()
        mdt = -1 if attn_mask.dtype == torch.int8 else ops.float_code(attn_mask.dtype)
        ms = tuple(attn_mask.stride(i) if attn_mask.shape[i] != 1 else 0 for i in range(3))
    ops.check(lib.sdnq_hip_attn(query.data_ptr(), key.data_ptr(), value.data_ptr(), ops.float_code(query.dtype), z, qh, kh, qn, kn, d,
                                _strides(query), _strides(key), _strides(value), 1 if smooth_k else 0, group, float(sm_scale),
                                1 if is_causal else 0, mptr, mdt, *ms, out.data_ptr(), ops.float_code(out_dtype), _strides(out),
                                ws.data_ptr() if ws is not None else None, nbytes, ops._stream(query)), "attn")
    return out
