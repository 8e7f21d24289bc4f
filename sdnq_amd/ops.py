"""Tensor-level wrappers over the C ABI (``include/sdnq_hip.h``).

PyTorch is only plumbing here: it owns device memory and the current HIP stream; every arithmetic
step runs in the hand-written gfx950 kernels of ``libsdnq_hip.so``.  All functions raise
``SdnqHipError`` on a non-zero status -- there is no eager fallback.
"""
from __future__ import annotations

import ctypes
import os
from dataclasses import dataclass

import torch

from . import _lib
from ._lib import BF16, F16, F32, MM_FP8, MM_I8, SdnqWeight, check
from .common import dtype_dict

_FLOAT_CODE = {torch.float32: F32, torch.bfloat16: BF16, torch.float16: F16}
_MM_TORCH = {MM_I8: torch.int8, MM_FP8: torch.float8_e4m3fn}


def float_code(dt: torch.dtype) -> int:
    try:
        return _FLOAT_CODE[dt]
    except KeyError:
        raise _lib.SdnqHipError(f"unsupported float dtype {dt}") from None


def mm_code(matmul_dtype: str) -> int:
    if matmul_dtype == "int8":
        return MM_I8
    if matmul_dtype in ("fp8", "float8_e4m3fn"):
        return MM_FP8
    raise _lib.SdnqHipError(f"unsupported quantized_matmul_dtype {matmul_dtype!r}")


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream(t: torch.Tensor) -> int:
    """Raw handle of torch's current HIP stream on t's device (the C accessor is ~10x cheaper than torch.cuda.current_stream,
    which dominated the host cost of an eager step)."""
    if _raw_stream is not None:
        return _raw_stream(t.device.index if t.device.index is not None else torch.cuda.current_device())
    return torch.cuda.current_stream(t.device).cuda_stream


def _ptr(t):
    return None if t is None else t.data_ptr()


def _require_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.SdnqHipError("sdnq_amd kernels need tensors on a gfx950 device (got a CPU tensor); "
                                    "there is no CPU fallback in the product path")


# ---------------------------------------------------------------------------------------------
# quantized weight descriptor
# ---------------------------------------------------------------------------------------------
@dataclass
class QuantWeight:
    """A quantized Linear weight in the physical layout the kernels read ([N][K], scales [N][G])."""
    desc: SdnqWeight
    n: int
    k: int
    group_size: int
    keep: tuple  # tensors kept alive for the raw pointers in `desc`
    scale_dtype: torch.dtype = torch.float32  # dtype the LAYER stores scale / zero_point in (dequantize_fp32=False: the model dtype)


def _storage_kind(weights_dtype: str):
    ent = dtype_dict[weights_dtype]
    bits = ent["num_bits"]
    native = 0
    if ent["is_integer"]:
        kind = _lib.KIND_UINT if ent["is_unsigned"] else _lib.KIND_INT
        if ent["is_packed"]:
            storage = _lib.ST_PACKED_U8 if bits < 8 else _lib.ST_PACKED_I16
        elif bits == 8:
            storage = _lib.ST_RAW8
        elif bits == 16:
            storage = _lib.ST_RAW16
        else:
            raise _lib.SdnqHipError(f"weights_dtype {weights_dtype} is not a storage format of the matmul path")
    else:
        kind = _lib.KIND_UFLOAT if ent["is_unsigned"] else _lib.KIND_FLOAT
        if ent["is_packed"]:
            storage = _lib.ST_RAW8 if bits == 8 else (_lib.ST_RAW16 if bits == 16 else (_lib.ST_PACKED_U8 if bits < 8 else _lib.ST_PACKED_I16))
        else:
            native = 1
            if bits == 8:
                storage = _lib.ST_RAW8
            elif bits == 16:
                storage = _lib.ST_RAW16
            else:
                raise _lib.SdnqHipError(f"weights_dtype {weights_dtype} is not a storage format of the matmul path")
    return storage, kind, bits, ent["exponent"], ent["mantissa"], native


def make_quant_weight(weights_dtype: str, weight: torch.Tensor, scale: torch.Tensor, zero_point, svd_up, svd_down,
                      n: int, k: int, group_size: int, transposed: bool, svd_transposed: bool | None = None,
                      positions: int = 1) -> QuantWeight:
    """Canonicalise module tensors (reference layouts) into the kernels' physical layout.

    transposed=False: weight is packed bytes / [N,K] / [N,G,g] (element order [N][K]); svd_up [N,R], svd_down [R,K].
    transposed=True : the qmm layout the reference's quantizer produces when use_quantized_matmul and not
                      re_quantize_for_matmul and not packed (quantizer.py:228-244): weight logical [K,N] with
                      strides (1,K) -- the same bytes as physical [N][K]; a *contiguous* [K,N] (as stored in
                      safetensors) is re-laid out once, like prepare_weight_for_matmul (quant_utils.py:240-249);
                      scale [1,N].
    svd_transposed  : svd_up [R,N], svd_down [K,R] -- the quantizer transposes the SVD factors whenever
                      use_quantized_matmul is on, independent of the weight layout (quantizer.py:164-167).
    positions       : P > 1 for conv weights [N, C_in, *kernel] quantized along C_in (quantizer.py:120-123, 205-209):
                      k = C_in * P, group_size counts channels, scale / zero_point hold N * (C_in / group_size) * P values.
    """
    if svd_transposed is None:
        svd_transposed = transposed
    _require_cuda(weight, scale)
    storage, kind, bits, ebits, mbits, native = _storage_kind(weights_dtype)
    if transposed:
        if tuple(weight.shape) != (k, n):
            raise _lib.SdnqHipError(f"transposed weight must be [K,N]=({k},{n}), got {tuple(weight.shape)}")
        w_phys = weight.t()
        if not w_phys.is_contiguous():
            w_phys = w_phys.contiguous()
    else:
        w_phys = weight if weight.is_contiguous() else weight.contiguous()
        if w_phys.dtype in (torch.int64, torch.bool):
            # 1-bit types: the reference's pack_uint1 on a bool tensor promotes to int64 words holding 8 bits each
            # (packed_int/pack.py:309-321); the kernels read uint8 words
            w_phys = w_phys.to(torch.uint8)
    scale_dtype = scale.dtype
    if scale_dtype not in (torch.float32, torch.bfloat16, torch.float16):
        raise _lib.SdnqHipError(f"scale must be float32, bfloat16 or float16, got {scale_dtype}")
    if scale_dtype != torch.float32 and bits > 8:
        raise _lib.SdnqHipError("16-bit scales with formats wider than 8 bits are not built (the codes are not exact in the scale dtype)")
    if zero_point is not None and zero_point.dtype != scale_dtype:
        raise _lib.SdnqHipError("scale and zero_point must share one dtype (loader.py:277-280 casts both)")
    g = (k // positions) // group_size * positions
    # dequantize_fp32=False keeps scale / zero_point in the model dtype (quantizer.py:147-156); the kernels read the exact float32
    # upcast and SdnqWeight.scale_dtype tells them where the reference's 16-bit tensors round
    sc = scale.to(torch.float32).contiguous().view(-1)
    if sc.numel() != n * g:
        raise _lib.SdnqHipError(f"scale has {sc.numel()} elements, expected N*G = {n * g}")
    zp = None
    if zero_point is not None:
        zp = zero_point.to(torch.float32).contiguous().view(-1)
        if zp.numel() != n * g:
            raise _lib.SdnqHipError("zero_point size mismatch")
    up = down = None
    rank, svd_dt = 0, 0
    if svd_up is not None:
        if svd_transposed:  # [R,N] , [K,R]
            up = svd_up.t().contiguous()
            down = svd_down.t().contiguous()
        else:  # [N,R] , [R,K]
            up = svd_up.contiguous()
            down = svd_down.contiguous()
        rank = up.shape[1]
        svd_dt = float_code(up.dtype)
    d = SdnqWeight(weight=_ptr(w_phys), scale=_ptr(sc), zero_point=_ptr(zp), svd_up=_ptr(up), svd_down=_ptr(down),
                   n=n, k=k, group_size=group_size, svd_rank=rank, svd_dtype=svd_dt, storage=storage, kind=kind,
                   bits=bits, exponent=ebits, mantissa=mbits, native_float=native, positions=positions,
                   scale_dtype=float_code(scale_dtype))
    return QuantWeight(desc=d, n=n, k=k, group_size=group_size, keep=(w_phys, sc, zp, up, down), scale_dtype=scale_dtype)


# ---------------------------------------------------------------------------------------------
# kernels
# ---------------------------------------------------------------------------------------------
def rowquant(x2d: torch.Tensor, mm: int, hadamard_group: int = 0, want_rowsum: bool = False, want_xrot: bool = False,
             prefetch: torch.Tensor | None = None, asymmetric: bool = False):
    """Row-quantize activations [M,K] -> (xq [M,K] int8|fp8, xs [M,1] f32, rowsum [M] i32|None, xrot|None).
    `prefetch`: tensor (the weight operand of the following matmul) to pull into the last-level cache meanwhile.
    `asymmetric`: int8 activations with a zero point (uint8 matmul); returns a 5th value xzp [M] f32."""
    _require_cuda(x2d)
    assert x2d.ndim == 2 and x2d.stride(1) == 1
    m, k = x2d.shape
    xq = torch.empty((m, k), device=x2d.device, dtype=_MM_TORCH[mm])
    xs = torch.empty((m, 1), device=x2d.device, dtype=torch.float32)
    rowsum = torch.empty((m,), device=x2d.device, dtype=torch.int32) if want_rowsum else None
    # rows beyond the kernel's register cache (K > 5120) are rotated once into this buffer and quantized from it, so it is
    # allocated for them even when the caller does not want the rotated copy
    # (group 256 on 16-bit activations up to K = 16384 rotates on the matrix cores, whole row in registers: no scratch copy)
    mfma_rot = hadamard_group == 256 and x2d.dtype != torch.float32 and k <= 16384 and not asymmetric
    need_rot = bool(hadamard_group) and (want_xrot or (k > 5120 and not mfma_rot))
    xrot = torch.empty((m, k), device=x2d.device, dtype=x2d.dtype) if need_rot else None
    xzp = torch.empty((m,), device=x2d.device, dtype=torch.float32) if asymmetric else None
    check(_lib.load().sdnq_hip_rowquant(x2d.data_ptr(), float_code(x2d.dtype), m, k, x2d.stride(0), mm, hadamard_group,
                                        xq.data_ptr(), xs.data_ptr(), _ptr(rowsum), _ptr(xrot), _ptr(prefetch),
                                        0 if prefetch is None else prefetch.numel() * prefetch.element_size(), _ptr(xzp),
                                        _stream(x2d)), "rowquant")
    if not want_xrot:
        xrot = None  # scratch only: released here (stream-ordered reuse by the caching allocator)
    if asymmetric:
        return xq, xs, rowsum, xrot, xzp
    return xq, xs, rowsum, xrot


def rowquant_lp(x2d: torch.Tensor, mm: int, hadamard_group: int = 0, want_rowsum: bool = False, want_xrot: bool = False):
    """sdnq_hip_rowquant_lp: the activation quantization carried out in x2d's own 16-bit dtype (dequantize_fp32=False layers).
    Returns (xq, xs [M,1] f32 holding dtype-representable values, rowsum | None, xrot | None)."""
    _require_cuda(x2d)
    assert x2d.ndim == 2 and x2d.stride(1) == 1 and x2d.dtype in (torch.bfloat16, torch.float16)
    m, k = x2d.shape
    xq = torch.empty((m, k), device=x2d.device, dtype=_MM_TORCH[mm])
    xs = torch.empty((m, 1), device=x2d.device, dtype=torch.float32)
    rowsum = torch.empty((m,), device=x2d.device, dtype=torch.int32) if want_rowsum else None
    xrot = torch.empty((m, k), device=x2d.device, dtype=x2d.dtype) if (hadamard_group and want_xrot) else None
    check(_lib.load().sdnq_hip_rowquant_lp(x2d.data_ptr(), float_code(x2d.dtype), m, k, x2d.stride(0), mm, hadamard_group,
                                           xq.data_ptr(), xs.data_ptr(), _ptr(rowsum), _ptr(xrot), _stream(x2d)), "rowquant_lp")
    return xq, xs, rowsum, xrot


def rowquant_lp_asym(x2d: torch.Tensor, hadamard_group: int = 0, want_rowsum: bool = False, want_xrot: bool = False):
    """sdnq_hip_rowquant_lp_asym: quantize_uint_mm_input in x2d's own 16-bit dtype (the uint8 matmul of dequantize_fp32=False layers).
    Returns (xq int8, xs [M,1] f32, xzp [M] f32 -- both holding dtype-representable values --, rowsum | None, xrot | None: the rotated
    activation of a Hadamard layer, for its SVD product)."""
    _require_cuda(x2d)
    assert x2d.ndim == 2 and x2d.stride(1) == 1 and x2d.dtype in (torch.bfloat16, torch.float16)
    m, k = x2d.shape
    xq = torch.empty((m, k), device=x2d.device, dtype=torch.int8)
    xs = torch.empty((m, 1), device=x2d.device, dtype=torch.float32)
    xzp = torch.empty((m,), device=x2d.device, dtype=torch.float32)
    rowsum = torch.empty((m,), device=x2d.device, dtype=torch.int32) if want_rowsum else None
    xrot = torch.empty((m, k), device=x2d.device, dtype=x2d.dtype) if (want_xrot and hadamard_group) else None
    check(_lib.load().sdnq_hip_rowquant_lp_asym(x2d.data_ptr(), float_code(x2d.dtype), m, k, x2d.stride(0), hadamard_group, xq.data_ptr(),
                                                xs.data_ptr(), xzp.data_ptr(), _ptr(rowsum), _ptr(xrot), _stream(x2d)), "rowquant_lp_asym")
    return xq, xs, xzp, rowsum, xrot


def scaled_mm_lp_uzp(a: torch.Tensor, b_phys: torch.Tensor, sa: torch.Tensor, sb: torch.Tensor, bias, rowsum, zp, a_zp: torch.Tensor,
                     w_colsum_scaled: torch.Tensor, zp_k: int = 0, t=None, svd_up=None) -> torch.Tensor:
    """sdnq_hip_scaled_mm_lp_uzp[_svd]: the uint8 matmul's epilogue on bfloat16 tensors (see the header); t [M,R] / svd_up [N,R] bf16 add the
    low-rank product to the bias (layers with SVD factors).  Returns [M,N] bf16."""
    _require_cuda(a, b_phys, sa, sb, bias, rowsum, zp, a_zp, w_colsum_scaled, t, svd_up)
    m, k = a.shape
    n = b_phys.shape[0]
    if bias is not None and bias.dtype != torch.bfloat16:
        raise _lib.SdnqHipError("scaled_mm_lp_uzp: bias must be bfloat16")
    out = torch.empty((m, n), device=a.device, dtype=torch.bfloat16)
    f32 = lambda t: None if t is None else t.reshape(-1).to(torch.float32).contiguous()  # noqa: E731
    sa_, sb_, zp_, azp_, wcs_ = f32(sa), f32(sb), f32(zp), f32(a_zp), f32(w_colsum_scaled)
    if t is not None:
        if t.dtype != torch.bfloat16 or svd_up.dtype != torch.bfloat16:
            raise _lib.SdnqHipError("scaled_mm_lp_uzp: low-rank factors must be bfloat16")
        t, svd_up = t.contiguous(), svd_up.contiguous()
        check(_lib.load().sdnq_hip_scaled_mm_lp_uzp_svd(a.data_ptr(), b_phys.data_ptr(), sa_.data_ptr(), sb_.data_ptr(), _ptr(None if bias is None else bias.contiguous()),
                                                        _ptr(rowsum), _ptr(zp_), azp_.data_ptr(), wcs_.data_ptr(), zp_k, t.data_ptr(), svd_up.data_ptr(),
                                                        t.shape[1], out.data_ptr(), m, n, k, _stream(a)), "scaled_mm_lp_uzp_svd")
        return out
    check(_lib.load().sdnq_hip_scaled_mm_lp_uzp(a.data_ptr(), b_phys.data_ptr(), sa_.data_ptr(), sb_.data_ptr(), _ptr(None if bias is None else bias.contiguous()),
                                                _ptr(rowsum), _ptr(zp_), azp_.data_ptr(), wcs_.data_ptr(), zp_k, out.data_ptr(), m, n, k, _stream(a)),
          "scaled_mm_lp_uzp")
    return out


def scaled_mm_lp(mm: int, a: torch.Tensor, b_phys: torch.Tensor, sa: torch.Tensor, sb: torch.Tensor, bias, t=None, svd_up=None,
                 rowsum=None, zp=None):
    """sdnq_hip_scaled_mm_lp[_zp]: the scaled matmul of bfloat16-scale layers (accumulator, activation-scale product and result each
    rounded to bf16); bias None | [N] | [M,N] bf16; t [M,R] / svd_up [N,R] bf16 add the low-rank bias; rowsum [M] i32 + zp [N] f32
    (bf16-representable) the zero-point term of unsigned weights.  Returns [M,N] bf16."""
    _require_cuda(a, b_phys, sa, sb, bias, t, svd_up, rowsum, zp)
    m, k = a.shape
    n = b_phys.shape[0]
    out = torch.empty((m, n), device=a.device, dtype=torch.bfloat16)
    bias_ndim, ld_bias = 0, 0
    if bias is not None:
        if bias.dtype != torch.bfloat16:
            raise _lib.SdnqHipError("scaled_mm_lp: bias must be bfloat16")
        bias = bias.contiguous()
        bias_ndim, ld_bias = bias.ndim, bias.shape[-1]
    rank = 0
    if t is not None:
        if t.dtype != torch.bfloat16 or svd_up.dtype != torch.bfloat16:
            raise _lib.SdnqHipError("scaled_mm_lp: low-rank factors must be bfloat16")
        t, svd_up = t.contiguous(), svd_up.contiguous()
        rank = t.shape[1]
    if zp is not None:
        zp = zp.reshape(-1).contiguous()
        assert zp.dtype == torch.float32 and rowsum is not None and rowsum.dtype == torch.int32
    check(_lib.load().sdnq_hip_scaled_mm_lp_zp(mm, a.data_ptr(), b_phys.data_ptr(), sa.data_ptr(), sb.data_ptr(), _ptr(bias), bias_ndim,
                                               ld_bias, _ptr(t), _ptr(svd_up), rank, _ptr(rowsum if zp is not None else None), _ptr(zp),
                                               out.data_ptr(), m, n, k, _stream(a)), "scaled_mm_lp")
    return out


def scaled_mm(mm: int, a: torch.Tensor, b_phys: torch.Tensor, sa: torch.Tensor, sb: torch.Tensor, bias, out_dtype: torch.dtype):
    """out[M,N] = cast(fma(f32(a @ b_phys^T) * sa, sb, bias)); b_phys is physical [N,K]."""
    _require_cuda(a, b_phys, sa, sb, bias)
    m, k = a.shape
    n = b_phys.shape[0]
    out = torch.empty((m, n), device=a.device, dtype=out_dtype)
    bias_ndim, ld_bias, bias_dt = 0, 0, 0
    if bias is not None:
        bias = bias.contiguous()
        bias_ndim = bias.ndim
        ld_bias = bias.shape[-1]
        bias_dt = float_code(bias.dtype)
    check(_lib.load().sdnq_hip_scaled_mm(mm, a.data_ptr(), b_phys.data_ptr(), sa.data_ptr(), sb.data_ptr(), _ptr(bias),
                                         bias_dt, bias_ndim, ld_bias, out.data_ptr(), float_code(out_dtype), m, n, k,
                                         _stream(a)), "scaled_mm")
    return out


def scaled_mm_multi(mm: int, a: torch.Tensor, b_phys: torch.Tensor, sa: torch.Tensor, sb: torch.Tensor, bias, out_dtype: torch.dtype,
                    n_outs: int):
    """One scaled matmul over the stacked weights of `n_outs` layers, each layer's output in its own contiguous [M, N / n_outs]
    tensor (sdnq_hip_scaled_mm_multi)."""
    _require_cuda(a, b_phys)
    m, k = a.shape
    n = b_phys.shape[0]
    seg = n // n_outs
    outs = [torch.empty((m, seg), device=a.device, dtype=out_dtype) for _ in range(n_outs)]
    ptrs = (ctypes.c_void_p * n_outs)(*[o.data_ptr() for o in outs])
    check(_lib.load().sdnq_hip_scaled_mm_multi(mm, a.data_ptr(), b_phys.data_ptr(), sa.data_ptr(), sb.data_ptr(), _ptr(bias),
                                               float_code(bias.dtype) if bias is not None else 0, ptrs, n_outs, seg, float_code(out_dtype),
                                               m, n, k, _stream(a)), "scaled_mm_multi")
    return outs


class GemmGroup:
    """Device-resident unit table of a grouped scaled matmul (sdnq_hip_scaled_mm_grouped): the output channels of several layers
    that consume ONE activation, cut into units of `unit_n` channels that point at the layers' own weight / scale / bias tensors
    (no stacked copy).  `members`: [(wq [N_i, K] int8 | fp8 physical, ws [N_i] f32, bias [N_i] | None)], all with one K."""

    def __init__(self, members):
        import math
        import numpy as np
        ns = [int(w.shape[0]) for (w, _, _) in members]
        unit = 0
        for n in ns:
            unit = math.gcd(unit, n)
        if unit % 64:
            raise _lib.SdnqHipError(f"grouped matmul needs layer widths with a common divisor that is a multiple of 64 (got {ns})")
        while unit > 2048 and unit % 128 == 0:  # keep units tile-sized multiples; any divisor of the gcd works
            unit //= 2
        has_bias = members[0][2] is not None
        if any((b is not None) != has_bias for (_, _, b) in members):
            raise _lib.SdnqHipError("grouped matmul: either every layer has a bias or none has")
        self.k = int(members[0][0].shape[1])
        self.keep = []  # the tensors the table points into
        rows = []
        n_start = 0
        for (w, ws, bias) in members:
            _require_cuda(w, ws, bias)
            n = int(w.shape[0])
            if w.shape[1] != self.k or not w.is_contiguous() or w.element_size() != 1:
                raise _lib.SdnqHipError("grouped matmul: weights must be contiguous [N, K] 1-byte operands of one K")
            ws = ws.reshape(-1)
            if ws.dtype != torch.float32 or not ws.is_contiguous() or ws.numel() != n:
                raise _lib.SdnqHipError("grouped matmul: scales must be contiguous float32 [N]")
            if bias is not None:
                bias = bias.contiguous()
                if bias.dtype != members[0][2].dtype:
                    raise _lib.SdnqHipError("grouped matmul: one bias dtype per group")
            self.keep.append((w, ws, bias))
            for loc in range(0, n, unit):
                rows.append((w.data_ptr() + loc * self.k, ws.data_ptr() + 4 * loc,
                             0 if bias is None else bias.data_ptr() + loc * bias.element_size(), n_start, n, loc))
            n_start += n
        dt = np.dtype([("b", "<u8"), ("sb", "<u8"), ("bias", "<u8"), ("n_start", "<i8"), ("n_seg", "<i4"), ("n_loc", "<i4")])
        assert dt.itemsize == ctypes.sizeof(_lib.SdnqGemmUnit)
        table = np.array(rows, dtype=dt)
        self.device = members[0][0].device
        self.table = torch.from_numpy(table.view(np.uint8).copy()).to(self.device)
        self.n_units, self.unit_n, self.n_total = len(rows), unit, n_start
        self.widths = ns
        self.bias_dtype = float_code(members[0][2].dtype) if has_bias else -1
        self.mm_torch = members[0][0].dtype


def scaled_mm_grouped(mm: int, a: torch.Tensor, sa: torch.Tensor, group: GemmGroup, out_dtype: torch.dtype):
    """One launch for every layer of `group` on the shared quantized activation a [M, K]; returns one contiguous [M, N_i] tensor
    per layer (views of ONE allocation), each bit-identical to scaled_mm on that layer alone."""
    _require_cuda(a, sa)
    m, k = a.shape
    if k != group.k:
        raise _lib.SdnqHipError(f"grouped matmul: activation has K={k}, the group K={group.k}")
    out = torch.empty((m * group.n_total,), device=a.device, dtype=out_dtype)
    check(_lib.load().sdnq_hip_scaled_mm_grouped(mm, a.data_ptr(), sa.data_ptr(), group.table.data_ptr(), group.n_units, group.unit_n,
                                                 group.bias_dtype, out.data_ptr(), float_code(out_dtype), m, k, _stream(a)),
          "scaled_mm_grouped")
    outs, start = [], 0
    for n in group.widths:
        outs.append(out[m * start:m * (start + n)].view(m, n))
        start += n
    return outs


def linear_float_multi(x2d: torch.Tensor, wd: torch.Tensor, bias, n_outs: int):
    """F.linear over the stacked dequantized weights of `n_outs` layers, one contiguous output per layer (M > 32)."""
    _require_cuda(x2d, wd)
    m, k = x2d.shape
    n = wd.shape[0]
    seg = n // n_outs
    outs = [torch.empty((m, seg), device=x2d.device, dtype=x2d.dtype) for _ in range(n_outs)]
    ptrs = (ctypes.c_void_p * n_outs)(*[o.data_ptr() for o in outs])
    check(_lib.load().sdnq_hip_linear_float_multi(x2d.data_ptr(), wd.data_ptr(), _ptr(bias), float_code(x2d.dtype), ptrs, n_outs, seg,
                                                  m, n, k, x2d.stride(0), _stream(x2d)), "linear_float_multi")
    return outs


def linear_w8a8(mm: int, x2d: torch.Tensor, b_phys: torch.Tensor, sb: torch.Tensor, bias, out_dtype: torch.dtype, hadamard_group: int = 0):
    """rowquant + scaled_mm through ONE binding call (two launches) -> (out [M,N], xq [M,K], xs [M,1])."""
    m, k = x2d.shape
    n = b_phys.shape[0]
    dev = x2d.device
    xq = torch.empty((m, k), device=dev, dtype=_MM_TORCH[mm])
    xs = torch.empty((m, 1), device=dev, dtype=torch.float32)
    out = torch.empty((m, n), device=dev, dtype=out_dtype)
    bias_dt = 0
    if bias is not None:
        bias_dt = float_code(bias.dtype)
    check(_lib.load().sdnq_hip_linear_w8a8(mm, x2d.data_ptr(), float_code(x2d.dtype), m, k, x2d.stride(0), hadamard_group, xq.data_ptr(),
                                           xs.data_ptr(), b_phys.data_ptr(), sb.data_ptr(), _ptr(bias), bias_dt, out.data_ptr(),
                                           float_code(out_dtype), n, _stream(x2d)), "linear_w8a8")
    return out, xq, xs


def linear_w8a8_ws(mm: int, x2d: torch.Tensor, b_phys: torch.Tensor, sb: torch.Tensor, bias, out_dtype: torch.dtype, hadamard_group: int = 0):
    """linear_w8a8 with the quantized activation in the stream's persistent scratch buffer (not returned, not kept): ONE allocation
    per call instead of three.  Same-stream launches are ordered, so the next layer overwriting the buffer is safe; a buffer replaced
    by a larger one stays valid for the work already queued (the caching allocator does not hand a freed block to another stream).
    While the stream is being captured into a graph the scratch is a fresh tensor of the graph's pool instead (the shared buffer
    may be replaced later)."""
    m, k = x2d.shape
    n = b_phys.shape[0]
    stream = _stream(x2d)
    xq_bytes = (m * k + 255) & ~255
    if torch.cuda.is_current_stream_capturing():
        scratch = torch.empty((xq_bytes + 4 * m + 512 + 255,), device=x2d.device, dtype=torch.uint8)
    else:
        scratch = _workspace(x2d.device, stream, xq_bytes + 4 * m + 512)  # the per-stream scratch buffer (below)
    base = (scratch.data_ptr() + 255) & ~255
    out = torch.empty((m, n), device=x2d.device, dtype=out_dtype)
    check(_lib.load().sdnq_hip_linear_w8a8(mm, x2d.data_ptr(), float_code(x2d.dtype), m, k, x2d.stride(0), hadamard_group, base,
                                           base + xq_bytes, b_phys.data_ptr(), sb.data_ptr(), _ptr(bias), 0 if bias is None else float_code(bias.dtype),
                                           out.data_ptr(), float_code(out_dtype), n, stream), "linear_w8a8")
    return out


def linear_w8a8_fused_supported(mm: int, x2d: torch.Tensor, n: int, out_dtype: torch.dtype) -> bool:
    """True where the ONE-launch w8a8 Linear (sdnq_hip_linear_w8a8_fused: the GEMM row-quantizes its own activation rows in LDS) is
    built and expected to win; the answer per (dtype, M, N, K) is memoized."""
    if x2d.dtype not in (torch.bfloat16, torch.float16) or out_dtype != x2d.dtype or not x2d.is_cuda or x2d.stride(0) * 128 >= (1 << 31):
        return False  # (a tile's 64 rows are addressed with 32-bit byte offsets)
    key = (mm, x2d.dtype, x2d.shape[0], n, x2d.shape[1], x2d.device.index)  # (per device: the answer depends on its CU count)
    r = _fused_ok.get(key)
    if r is None:
        with torch.cuda.device(x2d.device):
            r = bool(_lib.load().sdnq_hip_linear_w8a8_fused_supported(mm, float_code(x2d.dtype), float_code(out_dtype), x2d.shape[0], n, x2d.shape[1]))
        _fused_ok[key] = r
    return r


_fused_ok = {}
fused_calls = [0]  # how many Linear calls took the one-launch route through THIS module (fused_call_count() adds the fast path's)


def fused_call_count() -> int:
    """Linear calls that took the one-launch route since reset_fused_calls() (bench.py reports it per step)."""
    fp = _lib.fastpath()
    return fused_calls[0] + (fp.fused_calls() if fp is not None else 0)


def reset_fused_calls():
    fused_calls[0] = 0
    fp = _lib.fastpath()
    if fp is not None:
        fp.reset_counters()


def linear_w8a8_fused(mm: int, x2d: torch.Tensor, b_phys: torch.Tensor, sb: torch.Tensor, bias, out_dtype: torch.dtype) -> torch.Tensor:
    """The plain w8a8 Linear as one launch (no quantized copy of the activation is produced); bit-identical to linear_w8a8."""
    _require_cuda(x2d, b_phys, sb, bias)
    m, k = x2d.shape
    n = b_phys.shape[0]
    out = torch.empty((m, n), device=x2d.device, dtype=out_dtype)
    if bias is not None:
        bias = bias.contiguous()
    fused_calls[0] += 1
    check(_lib.load().sdnq_hip_linear_w8a8_fused(mm, x2d.data_ptr(), float_code(x2d.dtype), m, k, x2d.stride(0), b_phys.data_ptr(), sb.data_ptr(),
                                                 _ptr(bias), 0 if bias is None else float_code(bias.dtype), out.data_ptr(), float_code(out_dtype),
                                                 n, _stream(x2d)), "linear_w8a8_fused")
    return out


def scaled_mm_nchw(mm: int, a: torch.Tensor, b_phys: torch.Tensor, sa: torch.Tensor, sb: torch.Tensor, bias, out_dtype: torch.dtype,
                   batch: int, pixels: int) -> torch.Tensor:
    """Conv flavour of scaled_mm: rows m = (b, pixel); returns the channel-major image [batch, N, pixels] (the reference's
    .view(B, H, W, C).permute(0, 3, 1, 2).contiguous(), conv_int8.py:81-88, fused into the epilogue)."""
    _require_cuda(a, b_phys, sa, sb, bias)
    m, k = a.shape
    n = b_phys.shape[0]
    assert m == batch * pixels
    out = torch.empty((batch, n, pixels), device=a.device, dtype=out_dtype)
    bias_dt = 0
    if bias is not None:
        bias = bias.contiguous()
        bias_dt = float_code(bias.dtype)
    check(_lib.load().sdnq_hip_scaled_mm_nchw(mm, a.data_ptr(), b_phys.data_ptr(), sa.data_ptr(), sb.data_ptr(), _ptr(bias), bias_dt,
                                              out.data_ptr(), float_code(out_dtype), m, n, k, pixels, _stream(a)), "scaled_mm_nchw")
    return out


def scaled_mm_into(mm: int, a: torch.Tensor, b_phys: torch.Tensor, sa: torch.Tensor, sb: torch.Tensor, bias, out: torch.Tensor,
                   chan0: int, pixels: int = 0) -> None:
    """scaled_mm on views (sdnq_hip_scaled_mm_strided): `a` is a column slice [M, K] of a wider row-major matrix, the N = b_phys.shape[0]
    output channels go to channels chan0 .. chan0 + N of `out` -- [M, C] row-major (pixels == 0) or the conv image [B, C, pixels].
    One group of a grouped conv (conv_int8.py:73-79)."""
    _require_cuda(a, b_phys, sa, sb, bias, out)
    m, k = a.shape
    n = b_phys.shape[0]
    assert a.stride(1) == 1 and b_phys.is_contiguous() and out.is_contiguous()
    c_total = out.shape[1]
    esz = out.element_size()
    dst = out.data_ptr() + (chan0 * pixels * esz if pixels else chan0 * esz)
    bias_dt = 0
    if bias is not None:
        bias = bias.contiguous()
        bias_dt = float_code(bias.dtype)
    check(_lib.load().sdnq_hip_scaled_mm_strided(mm, a.data_ptr(), a.stride(0), b_phys.data_ptr(), sa.data_ptr(), sb.data_ptr(), _ptr(bias),
                                                 bias_dt, dst, c_total, float_code(out.dtype), m, n, k, pixels, _stream(a)), "scaled_mm_strided")


def scaled_mm_zp_into(mm: int, a: torch.Tensor, b_phys: torch.Tensor, sa: torch.Tensor, sb: torch.Tensor, bias, rowsum, zp, a_zp, w_colsum_scaled,
                      zp_k: int, out: torch.Tensor, chan0: int) -> None:
    """scaled_mm with the zero-point / activation-zero-point epilogue terms on views (sdnq_hip_scaled_mm_lowrank_strided): one group of a
    grouped conv with unsigned weights or the uint8 matmul (conv_int8.py:65-79, conv_uint8.py:58-79).  `rowsum` / `a_zp` are per ROW
    of the whole unfolded input, `sb / bias / zp / w_colsum_scaled` this group's channels, `zp_k` the whole row's K; the N results go to
    columns chan0 .. chan0 + N of the row-major `out` [M, C]."""
    _require_cuda(a, b_phys, sa, sb, bias, rowsum, zp, a_zp, w_colsum_scaled, out)
    m, k = a.shape
    n = b_phys.shape[0]
    assert a.stride(1) == 1 and b_phys.is_contiguous() and out.is_contiguous() and out.dim() == 2
    bias_dt = 0
    if bias is not None:
        bias = bias.contiguous()
        bias_dt = float_code(bias.dtype)
    check(_lib.load().sdnq_hip_scaled_mm_lowrank_strided(mm, a.data_ptr(), a.stride(0), b_phys.data_ptr(), sa.data_ptr(), sb.data_ptr(), _ptr(bias),
                                                         bias_dt, _ptr(rowsum), _ptr(zp), _ptr(a_zp), _ptr(w_colsum_scaled), int(zp_k),
                                                         out.data_ptr() + chan0 * out.element_size(), out.shape[1], float_code(out.dtype), m, n, k,
                                                         _stream(a)), "scaled_mm_lowrank_strided")


def linear_float_into(x2d: torch.Tensor, w: torch.Tensor, bias, out: torch.Tensor, chan0: int) -> None:
    """linear_float on views (sdnq_hip_linear_float_strided): x2d a column slice [M, K], the N outputs go to columns chan0 .. chan0 + N of
    the row-major `out` [M, C]."""
    _require_cuda(x2d, w, bias, out)
    m, k = x2d.shape
    n = w.shape[0]
    assert w.is_contiguous() and x2d.stride(1) == 1 and w.dtype == x2d.dtype == out.dtype and out.is_contiguous()
    if bias is not None and bias.dtype != x2d.dtype:
        bias = bias.to(x2d.dtype)
    check(_lib.load().sdnq_hip_linear_float_strided(x2d.data_ptr(), w.data_ptr(), _ptr(bias), float_code(x2d.dtype),
                                                    out.data_ptr() + chan0 * out.element_size(), m, n, k, x2d.stride(0), out.shape[1],
                                                    _stream(x2d)), "linear_float_strided")


def scaled_mm_lowrank(mm: int, a, b_phys, sa, sb, bias, t, svd_up_phys, rowsum, zp, out_dtype: torch.dtype, a_zp=None,
                      w_colsum_scaled=None):
    _require_cuda(a, b_phys)
    m, k = a.shape
    n = b_phys.shape[0]
    out = torch.empty((m, n), device=a.device, dtype=out_dtype)
    rank = 0 if t is None else t.shape[1]
    svd_dt = 0 if t is None else float_code(t.dtype)
    bias_dt = 0
    if bias is not None:
        if t is not None and bias.dtype != t.dtype:
            bias = bias.to(t.dtype)  # bias.to(dtype=svd_down.dtype), linear_int8.py:60
        bias = bias.contiguous()
        bias_dt = float_code(bias.dtype)
    check(_lib.load().sdnq_hip_scaled_mm_lowrank(mm, a.data_ptr(), b_phys.data_ptr(), sa.data_ptr(), sb.data_ptr(), _ptr(bias),
                                                 bias_dt, _ptr(t), _ptr(svd_up_phys), svd_dt, rank, _ptr(rowsum), _ptr(zp),
                                                 _ptr(a_zp), _ptr(w_colsum_scaled), out.data_ptr(), float_code(out_dtype), m, n, k, _stream(a)), "scaled_mm_lowrank")
    return out


def dequant(qw: QuantWeight, out_dtype: torch.dtype, hadamard_group: int = 0, use_svd: bool = True, out: torch.Tensor | None = None) -> torch.Tensor:
    dev = qw.keep[0].device
    if out is None:
        out = torch.empty((qw.n, qw.k), device=dev, dtype=out_dtype)
    d = qw.desc
    if not use_svd and d.svd_up:
        d = SdnqWeight.from_buffer_copy(bytes(d))
        d.svd_up = None
        d.svd_down = None
        d.svd_rank = 0
    check(_lib.load().sdnq_hip_dequant(ctypes.byref(d), hadamard_group, out.data_ptr(), float_code(out_dtype),
                                       torch.cuda.current_stream(dev).cuda_stream), "dequant")
    return out


def requant(qw: QuantWeight, mm: int, ws: torch.Tensor | None = None, out: torch.Tensor | None = None):
    """re_quantize_int_mm / re_quantize_fp_mm (dequantizer.py:166-174, 204-239): (wq [N, K] int8 | fp8, ws [N] f32).  `ws`: the row
    scales of an earlier call on the same weights (sdnq_hip_requant_ws: the pass that derives them is skipped where the kernel can).
    `out`: caller-owned byte buffer of at least N * K bytes (16-byte aligned) the operand is written into (the per-call mode's
    double-buffered scratch); the returned wq is a view of it.  Runs on torch's CURRENT stream."""
    dev = qw.keep[0].device
    if out is not None:
        assert out.is_cuda and out.element_size() == 1 and out.numel() >= qw.n * qw.k and out.data_ptr() % 16 == 0
        wq = out[: qw.n * qw.k].view(_MM_TORCH[mm]).view(qw.n, qw.k)
    else:
        wq = torch.empty((qw.n, qw.k), device=dev, dtype=_MM_TORCH[mm])
    known = ws is not None
    if not known:
        ws = torch.empty((qw.n,), device=dev, dtype=torch.float32)
    check(_lib.load().sdnq_hip_requant_ws(ctypes.byref(qw.desc), mm, wq.data_ptr(), ws.data_ptr(), 1 if known else 0,
                                          torch.cuda.current_stream(dev).cuda_stream), "requant")
    return wq, ws


def lut4_build(qw: QuantWeight, mm: int, ws: torch.Tensor | None = None):
    """The re-quantization tables of a 4-bit weight (sdnq_hip_lut4_build): (lut uint8 [N, K / 64, 16], ws [N] f32) -- entry c of a table is
    the byte `requant` writes for code c in that 64-column block of that row.  Built once per layer; the operand of scaled_mm_w4."""
    dev = qw.keep[0].device
    lut = torch.empty((qw.n, qw.k // 64, 16), device=dev, dtype=torch.uint8)
    known = ws is not None
    if not known:
        ws = torch.empty((qw.n,), device=dev, dtype=torch.float32)
    check(_lib.load().sdnq_hip_lut4_build(ctypes.byref(qw.desc), mm, ws.data_ptr(), 1 if known else 0, lut.data_ptr(),
                                          torch.cuda.current_stream(dev).cuda_stream), "lut4_build")
    return lut, ws


def scaled_mm_w4_supported(mm: int, m: int, n: int, k: int, out_dtype: torch.dtype) -> bool:
    return out_dtype in (torch.bfloat16, torch.float16) and bool(_lib.load().sdnq_hip_scaled_mm_w4_supported(mm, float_code(out_dtype), m, n, k))


def scaled_mm_w4(a: torch.Tensor, codes: torch.Tensor, lut: torch.Tensor, sa: torch.Tensor, sb: torch.Tensor, bias, out_dtype: torch.dtype):
    """The quantized matmul on the STORED 4-bit codes (sdnq_hip_scaled_mm_w4): bit for bit scaled_mm(a, requant(weight), ...).
    a [M, K] int8 (row stride >= K), codes: the packed weight tensor (N * K / 2 bytes), lut: lut4_build's tables."""
    _require_cuda(a, codes, lut, sa, sb, bias)
    m, k = a.shape
    n = lut.shape[0]
    if a.stride(-1) != 1 or codes.numel() * codes.element_size() != n * k // 2 or not codes.is_contiguous() or tuple(lut.shape) != (n, k // 64, 16):
        raise _lib.SdnqHipError("scaled_mm_w4: a [M, K] int8 (unit inner stride), contiguous packed codes of N * K / 2 bytes, lut [N, K / 64, 16]")
    out = torch.empty((m, n), device=a.device, dtype=out_dtype)
    check(_lib.load().sdnq_hip_scaled_mm_w4(a.data_ptr(), codes.data_ptr(), lut.data_ptr(), sa.data_ptr(), sb.data_ptr(), _ptr(bias),
                                            0 if bias is None else float_code(bias.dtype), out.data_ptr(), float_code(out_dtype), m, n, k,
                                            a.stride(0), _stream(a)), "scaled_mm_w4")
    return out


def rowquant_f16(x2d: torch.Tensor):
    """quantize_fp_mm_input(..., matmul_dtype="float16") (linear_fp8.py:15-22, quant_utils.py:290-299): (xq float16 [M, K], xs f32 [M])."""
    _require_cuda(x2d)
    m, k = x2d.shape
    if x2d.stride(-1) != 1:
        raise _lib.SdnqHipError("rowquant_f16: unit inner stride")
    xq = torch.empty((m, k), device=x2d.device, dtype=torch.float16)
    xs = torch.empty((m,), device=x2d.device, dtype=torch.float32)
    check(_lib.load().sdnq_hip_rowquant_f16(x2d.data_ptr(), float_code(x2d.dtype), m, k, x2d.stride(0), xq.data_ptr(), xs.data_ptr(), _stream(x2d)), "rowquant_f16")
    return xq, xs


def unpack_mm_f16(qw: QuantWeight) -> torch.Tensor:
    """Stored float codes -> float16 matmul operand [N, K] (linear_fp16.py:27-31)."""
    dev = qw.keep[0].device
    wq = torch.empty((qw.n, qw.k), device=dev, dtype=torch.float16)
    check(_lib.load().sdnq_hip_unpack_mm(ctypes.byref(qw.desc), _lib.MM_F16, wq.data_ptr(), torch.cuda.current_stream(dev).cuda_stream), "unpack_mm_f16")
    return wq


def scaled_mm_f16(a: torch.Tensor, b_phys: torch.Tensor, sa: torch.Tensor, sb: torch.Tensor, bias, out_dtype: torch.dtype) -> torch.Tensor:
    """fp_scaled_mm_func on float16 operands (kernel_wrappers.py:207-211): a [M, K] f16, b_phys [N, K] f16 -> [M, N]; bias None, [N] or the
    [M, N] low-rank term of a layer with SVD factors (linear_fp16.py:38-43)."""
    _require_cuda(a, b_phys, sa, sb, bias)
    m, k = a.shape
    n = b_phys.shape[0]
    if a.dtype != torch.float16 or b_phys.dtype != torch.float16 or not a.is_contiguous() or not b_phys.is_contiguous() or b_phys.shape[1] != k:
        raise _lib.SdnqHipError("scaled_mm_f16: contiguous float16 operands a [M, K], b [N, K]")
    nd, ldb = 0, 0
    if bias is not None:
        bias = bias.contiguous()
        nd = bias.dim()
        if nd == 2:
            if tuple(bias.shape) != (m, n):
                raise _lib.SdnqHipError("scaled_mm_f16: a 2-D bias must be [M, N]")
            ldb = bias.stride(0)
        elif nd != 1 or bias.numel() != n:
            raise _lib.SdnqHipError("scaled_mm_f16: bias must be [N] or [M, N]")
    out = torch.empty((m, n), device=a.device, dtype=out_dtype)
    check(_lib.load().sdnq_hip_scaled_mm_f16(a.data_ptr(), b_phys.data_ptr(), sa.data_ptr(), sb.data_ptr(), _ptr(bias), 0 if bias is None else float_code(bias.dtype),
                                             nd, ldb, out.data_ptr(), float_code(out_dtype), m, n, k, _stream(a)), "scaled_mm_f16")
    return out


def requant_asym(qw: QuantWeight):
    """re_quantize_uint_mm (dequantizer.py:178-187): (wq int8 [N,K], ws [N], zero_point [N])."""
    dev = qw.keep[0].device
    wq = torch.empty((qw.n, qw.k), device=dev, dtype=torch.int8)
    ws = torch.empty((qw.n,), device=dev, dtype=torch.float32)
    wzp = torch.empty((qw.n,), device=dev, dtype=torch.float32)
    check(_lib.load().sdnq_hip_requant_asym(ctypes.byref(qw.desc), wq.data_ptr(), ws.data_ptr(), wzp.data_ptr(),
                                            torch.cuda.current_stream(dev).cuda_stream), "requant_asym")
    return wq, ws, wzp


def unpack_mm(qw: QuantWeight, mm: int) -> torch.Tensor:
    """Stored codes -> matmul operand [N,K] without re-quantization (linear_int8.py:38-50, linear_fp8.py:36-38)."""
    dev = qw.keep[0].device
    wq = torch.empty((qw.n, qw.k), device=dev, dtype=_MM_TORCH[mm])
    check(_lib.load().sdnq_hip_unpack_mm(ctypes.byref(qw.desc), mm, wq.data_ptr(), torch.cuda.current_stream(dev).cuda_stream),
          "unpack_mm")
    return wq


def hadamard(x: torch.Tensor, group: int) -> torch.Tensor:
    _require_cuda(x)
    shape = x.shape
    x2 = x.reshape(-1, shape[-1])
    if x2.stride(1) != 1:
        x2 = x2.contiguous()
    y = torch.empty((x2.shape[0], x2.shape[1]), device=x.device, dtype=x.dtype)
    check(_lib.load().sdnq_hip_hadamard(x2.data_ptr(), float_code(x.dtype), x2.shape[0], x2.shape[1], x2.stride(0), group,
                                        y.data_ptr(), y.stride(0), _stream(x)), "hadamard")
    return y.view(shape)


def linear_float(x2d: torch.Tensor, w: torch.Tensor, bias) -> torch.Tensor:
    """x2d [M,K] @ w[N,K]^T + bias, all in one float dtype, fp32 accumulate."""
    _require_cuda(x2d, w, bias)
    m, k = x2d.shape
    n = w.shape[0]
    assert w.is_contiguous() and x2d.stride(1) == 1 and w.dtype == x2d.dtype
    if bias is not None and bias.dtype != x2d.dtype:
        bias = bias.to(x2d.dtype)
    out = torch.empty((m, n), device=x2d.device, dtype=x2d.dtype)
    check(_lib.load().sdnq_hip_linear_float(x2d.data_ptr(), w.data_ptr(), _ptr(bias), float_code(x2d.dtype), out.data_ptr(),
                                            m, n, k, x2d.stride(0), _stream(x2d)), "linear_float")
    return out


def linear_w8a16(x2d: torch.Tensor, w: torch.Tensor, scale: torch.Tensor, zero_point, bias) -> torch.Tensor:
    """Fused dequantize + float GEMM for row-wise 8-bit weights (sdnq_hip_linear_w8a16): x2d [M, K] bf16 / f16, w [N, K] int8 /
    uint8 physical, scale [N] f32, zero_point [N] f32 | None -> [M, N] of x2d's dtype."""
    _require_cuda(x2d, w, scale, zero_point, bias)
    m, k = x2d.shape
    n = w.shape[0]
    assert w.is_contiguous() and w.element_size() == 1 and x2d.stride(1) == 1 and scale.numel() == n
    if bias is not None and bias.dtype != x2d.dtype:
        bias = bias.to(x2d.dtype)
    out = torch.empty((m, n), device=x2d.device, dtype=x2d.dtype)
    check(_lib.load().sdnq_hip_linear_w8a16(x2d.data_ptr(), float_code(x2d.dtype), w.data_ptr(), scale.data_ptr(), _ptr(zero_point), _ptr(bias),
                                            out.data_ptr(), m, n, k, x2d.stride(0), _stream(x2d)), "linear_w8a16")
    return out


def linear_w8a16_grouped(x2d: torch.Tensor, group: GemmGroup):
    """linear_w8a16 for every (signed int8, row-wise) layer of `group` in one launch; one contiguous [M, N_i] tensor per layer."""
    _require_cuda(x2d)
    m, k = x2d.shape
    if k != group.k or group.mm_torch != torch.int8:
        raise _lib.SdnqHipError("grouped fused dequantize GEMM: int8 weights of the activation's K")
    if group.bias_dtype >= 0 and group.bias_dtype != float_code(x2d.dtype):
        raise _lib.SdnqHipError("grouped fused dequantize GEMM: bias must have the activation dtype")
    out = torch.empty((m * group.n_total,), device=x2d.device, dtype=x2d.dtype)
    check(_lib.load().sdnq_hip_linear_w8a16_grouped(x2d.data_ptr(), float_code(x2d.dtype), group.table.data_ptr(), group.n_units, group.unit_n,
                                                    1 if group.bias_dtype >= 0 else 0, out.data_ptr(), m, k, x2d.stride(0), _stream(x2d)),
          "linear_w8a16_grouped")
    outs, start = [], 0
    for n in group.widths:
        outs.append(out[m * start:m * (start + n)].view(m, n))
        start += n
    return outs


def linear_skinny(qw: QuantWeight, x2d: torch.Tensor, bias, hadamard_group: int = 0) -> torch.Tensor:
    """Fused dequant (+ Hadamard un-rotation of the weight in registers) + float linear for M < 32 rows: streams the
    quantized weight once (no SVD)."""
    _require_cuda(x2d, bias)
    m, k = x2d.shape
    assert k == qw.k and x2d.stride(1) == 1
    if bias is not None and bias.dtype != x2d.dtype:
        bias = bias.to(x2d.dtype)
    out = torch.empty((m, qw.n), device=x2d.device, dtype=x2d.dtype)
    check(_lib.load().sdnq_hip_linear_skinny(ctypes.byref(qw.desc), hadamard_group, x2d.data_ptr(), _ptr(bias), float_code(x2d.dtype), out.data_ptr(), m,
                                             x2d.stride(0), _stream(x2d)), "linear_skinny")
    return out


def linear_skinny_svd(qw: QuantWeight, svd_down_t: torch.Tensor, x2d: torch.Tensor, bias) -> torch.Tensor:
    """Few-row linear on an int8 row-wise weight with SVD factors; svd_down_t is [K, R] contiguous."""
    _require_cuda(x2d, bias, svd_down_t)
    m, k = x2d.shape
    assert k == qw.k and x2d.stride(1) == 1 and svd_down_t.is_contiguous()
    if bias is not None and bias.dtype != x2d.dtype:
        bias = bias.to(x2d.dtype)
    out = torch.empty((m, qw.n), device=x2d.device, dtype=x2d.dtype)
    check(_lib.load().sdnq_hip_linear_skinny_svd(ctypes.byref(qw.desc), svd_down_t.data_ptr(), x2d.data_ptr(), _ptr(bias),
                                                 float_code(x2d.dtype), out.data_ptr(), m, x2d.stride(0), _stream(x2d)), "linear_skinny_svd")
    return out


def lowrank_down(x2d: torch.Tensor, svd_down_phys: torch.Tensor) -> torch.Tensor:
    """t[M,R] = cast(x2d @ svd_down_phys^T); svd_down_phys is physical [R,K]."""
    m, k = x2d.shape
    r = svd_down_phys.shape[0]
    t = torch.empty((m, r), device=x2d.device, dtype=svd_down_phys.dtype)
    check(_lib.load().sdnq_hip_lowrank_down(x2d.data_ptr(), float_code(x2d.dtype), m, k, x2d.stride(0), svd_down_phys.data_ptr(),
                                            float_code(svd_down_phys.dtype), r, t.data_ptr(), _stream(x2d)), "lowrank_down")
    return t


_workspaces: dict = {}


def _workspace(dev: torch.device, stream: int, nbytes: int) -> torch.Tensor:
    """Scratch buffer of sdnq_hip_linear for (device, stream): grown on demand, reused by every layer on that stream (calls on one
    stream are ordered, so a buffer per stream is never used by two calls at once)."""
    key = (dev.index, stream)
    buf = _workspaces.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty((max(nbytes, 1 << 20) + 255,), device=dev, dtype=torch.uint8)
        _workspaces[key] = buf
    return buf


def linear_call(mm: int, x2d: torch.Tensor, wq: torch.Tensor, ws: torch.Tensor, bias, out_dtype: torch.dtype, hadamard_group: int = 0,
                svd_down=None, svd_up=None, zp=None, asymmetric: bool = False, w_colsum_scaled=None, pre=None):
    """The whole quantized-matmul forward of one layer through ONE C call (sdnq_hip_linear): row quantization (unless `pre` =
    (xq, xs, rowsum, xrot, xzp) of an earlier call on the same activation is handed in), low-rank product, scaled matmul with every
    epilogue term.  Returns (out [M, N], (xq, xs, rowsum, xrot, xzp)); the intermediates are torch tensors (they may be cached), only the
    low-rank product lives in the per-stream scratch buffer."""
    _require_cuda(x2d, wq, ws, bias, svd_down, svd_up, zp)
    m, k = x2d.shape
    n = wq.shape[0]
    dev = x2d.device
    a = _lib.SdnqLinearArgs()
    a.struct_size = ctypes.sizeof(_lib.SdnqLinearArgs)
    a.mm_dtype, a.x_dtype, a.out_dtype = mm, float_code(x2d.dtype), float_code(out_dtype)
    a.hadamard_group, a.asymmetric = int(hadamard_group), 1 if asymmetric else 0
    a.m, a.n, a.k, a.ldx = m, n, k, x2d.stride(0)
    out = torch.empty((m, n), device=dev, dtype=out_dtype)
    a.x, a.out, a.wq, a.ws = x2d.data_ptr(), out.data_ptr(), wq.data_ptr(), ws.data_ptr()
    if bias is not None:
        bias = bias.contiguous()
        a.bias, a.bias_dtype = bias.data_ptr(), float_code(bias.dtype)
    if svd_up is not None:
        a.svd_down, a.svd_up, a.svd_rank, a.svd_dtype = svd_down.data_ptr(), svd_up.data_ptr(), svd_up.shape[1], float_code(svd_up.dtype)
        if bias is not None and bias.dtype != svd_up.dtype:
            bias = bias.to(svd_up.dtype)  # the [M, N] bias of the reference lives in the svd dtype (linear_int8.py:60)
            a.bias, a.bias_dtype = bias.data_ptr(), float_code(bias.dtype)
    if zp is not None:
        a.zp = zp.data_ptr()
    if asymmetric:
        a.w_colsum_scaled = w_colsum_scaled.data_ptr()
    need_rowsum, need_xrot = zp is not None, svd_up is not None and hadamard_group != 0
    if pre is not None:
        xq, xs, rowsum, xrot, xzp = pre
        a.x_prequantized = 1
    else:
        xq = torch.empty((m, k), device=dev, dtype=_MM_TORCH[mm])
        xs = torch.empty((m, 1), device=dev, dtype=torch.float32)
        rowsum = torch.empty((m,), device=dev, dtype=torch.int32) if need_rowsum else None
        xrot = torch.empty((m, k), device=dev, dtype=x2d.dtype) if need_xrot else None
        xzp = torch.empty((m, 1), device=dev, dtype=torch.float32) if asymmetric else None
    a.xq, a.xs, a.rowsum, a.xrot, a.xzp = xq.data_ptr(), xs.data_ptr(), _ptr(rowsum), _ptr(xrot), _ptr(xzp)
    lib = _lib.load()
    need = ctypes.c_int64(0)
    check(lib.sdnq_hip_linear_workspace_bytes(ctypes.byref(a), ctypes.byref(need)), "linear_workspace_bytes")
    stream = _stream(x2d)
    if need.value > 0:
        # while the stream is being captured the scratch must belong to the graph's memory pool: the shared per-stream buffer may be
        # replaced (and freed) by a later, larger eager call while the captured graph still holds its address (advisor, round 3)
        w = (torch.empty((need.value + 255,), device=dev, dtype=torch.uint8) if torch.cuda.is_current_stream_capturing()
             else _workspace(dev, stream, need.value))
        base = (w.data_ptr() + 255) & ~255
        a.workspace, a.workspace_bytes = base, w.numel() - (base - w.data_ptr())
    check(lib.sdnq_hip_linear(ctypes.byref(a), stream), "linear")
    return out, (xq, xs, rowsum, xrot, xzp)


def unshard_columns(gathered: torch.Tensor, out: torch.Tensor, starts, m0: int = 0) -> torch.Tensor:
    """out[m0 + i][starts[r] + c] = gathered[r][i][c]: the row-major re-assembly of a column-sharded layer's all-gathered output
    (sdnq_hip_unshard_columns).  gathered [W, rows, wmax] (rank-major, slabs padded to the widest), out [M, N] with N = starts[W]."""
    _require_cuda(gathered, out)
    world, rows, wmax = gathered.shape
    arr = (ctypes.c_int64 * (world + 1))(*[int(v) for v in starts])
    check(_lib.load().sdnq_hip_unshard_columns(gathered.data_ptr(), out.data_ptr(), out.element_size(), m0, rows, out.shape[0], wmax,
                                               world, arr, _stream(out)), "unshard_columns")
    return out


def im2col(x: torch.Tensor, kernel, stride, padding, dilation) -> tuple[torch.Tensor, tuple]:
    """F.unfold(x, ...).transpose(1, 2) for x [B, C, H, W] -> ([B * H_out * W_out, C * kh * kw], (B, H_out, W_out));
    HIP replacement of process_conv_input's unfold (layers/conv/forward.py:75)."""
    _require_cuda(x)
    if x.ndim != 4:
        raise _lib.SdnqHipError("im2col expects [B, C, H, W]")
    x = x if x.is_contiguous() else x.contiguous()
    b, c, h, w = x.shape
    (kh, kw), (sh, sw), (ph, pw), (dh, dw) = kernel, stride, padding, dilation
    ho = (h + 2 * ph - dh * (kh - 1) - 1) // sh + 1
    wo = (w + 2 * pw - dw * (kw - 1) - 1) // sw + 1
    if ho <= 0 or wo <= 0:
        raise _lib.SdnqHipError(f"convolution output would be empty ({ho} x {wo})")
    out = torch.empty((b * ho * wo, c * kh * kw), device=x.device, dtype=x.dtype)
    check(_lib.load().sdnq_hip_im2col(x.data_ptr(), float_code(x.dtype), b, c, h, w, kh, kw, sh, sw, ph, pw, dh, dw, out.data_ptr(),
                                      _stream(x)), "im2col")
    return out, (b, ho, wo)


# per (device, stream): [zeroed int32 buffer (ticket + amax map) of sdnq_hip_im2col_rowquant_z, call-in-flight flag]
_amax_maps: dict = {}
_capture_amax_maps: dict = {}  # per (device, stream): (capture id, the map the convs of that capture address)
_AMAX_HEADER = 32 + 256 * 32  # ticket words in front of the map (csrc/conv.hip: SDNQ_CONV_WS_HEADER_WORDS)
SELF_CLEANING_AMAX = os.environ.get("SDNQ_HIP_CONV_SELF_CLEAN", "1") != "0"


def _zeroed_amax_map(x: torch.Tensor, words: int):
    """The stream's self-cleaning amax map, at least _AMAX_HEADER + `words` words, or None (then the caller takes the zero-per-call form): a new
    buffer cannot be made while the stream is capturing (its zero fill would be replayed, its memory would belong to the graph)."""
    if torch.cuda.is_current_stream_capturing():
        # never bake the stream's persistent map into a graph: a graph replayed on ANOTHER stream, beside eager convs on the capture stream,
        # would share one map and one set of tickets with no ordering between them (advisor, round 4).  A captured conv addresses a map of
        # ITS capture: made by the capture's first conv out of the graph's own pool -- its zero fill is part of the graph, ONE zeroing launch
        # per replay -- and kept all-zero between the graph's convs by the self-cleaning kernel (round 5: the zero-per-call form cost a
        # captured SDXL conv step 48 launches of 5 us, 8 % of it).
        cid = ctypes.c_uint64(0)
        check(_lib.load().sdnq_hip_stream_capture_id(_stream(x), ctypes.byref(cid)), "stream_capture_id")
        if cid.value == 0:
            return None
        key = (x.device.index, _stream(x))
        cap = _capture_amax_maps.get(key)
        need = _AMAX_HEADER + words
        if cap is None or cap[0] != cid.value or cap[1].numel() < need:
            cap = (cid.value, torch.zeros((max(need, _AMAX_HEADER + 65536),), device=x.device, dtype=torch.int32))
            _capture_amax_maps[key] = cap  # (the previous capture's map goes back to ITS graph's pool; that graph keeps addressing it)
        return [cap[1], False]
    key = (x.device.index, _stream(x))
    ent = _amax_maps.get(key)
    need = _AMAX_HEADER + words
    if ent is None or ent[0].numel() < need or ent[1]:
        if ent is not None and ent[0].numel() >= need:  # a call that failed midway may have left marks behind
            ent[0].zero_()
            ent[1] = False
        else:  # (an outgrown map is simply dropped: no graph holds its address, and the stream orders its last use before the free)
            ent = [torch.zeros((max(need, _AMAX_HEADER + 65536),), device=x.device, dtype=torch.int32), False]
            _amax_maps[key] = ent
    return ent


def im2col_rowquant(x: torch.Tensor, kernel, stride, padding, dilation, mm: int):
    """Fused unfold + row-wise activation quantization for the conv matmul forwards:
    -> (xq [M, K] int8 | fp8, xs [M, 1] f32, (B, H_out, W_out)); equals rowquant(im2col(x))."""
    _require_cuda(x)
    if x.ndim != 4:
        raise _lib.SdnqHipError("im2col_rowquant expects [B, C, H, W]")
    x = x if x.is_contiguous() else x.contiguous()
    b, c, h, w = x.shape
    (kh, kw), (sh, sw), (ph, pw), (dh, dw) = kernel, stride, padding, dilation
    ho = (h + 2 * ph - dh * (kh - 1) - 1) // sh + 1
    wo = (w + 2 * pw - dw * (kw - 1) - 1) // sw + 1
    if ho <= 0 or wo <= 0:
        raise _lib.SdnqHipError(f"convolution output would be empty ({ho} x {wo})")
    m, k = b * ho * wo, c * kh * kw
    xq = torch.empty((m, k), device=x.device, dtype=_MM_TORCH[mm])
    xs = torch.empty((m, 1), device=x.device, dtype=torch.float32)
    ent = _zeroed_amax_map(x, b * h * w) if SELF_CLEANING_AMAX else None
    if ent is not None:  # the stream's persistent map: zero on entry, zeroed again by the quantizing kernel's last workgroup
        ent[1] = True
        check(_lib.load().sdnq_hip_im2col_rowquant_z(x.data_ptr(), float_code(x.dtype), b, c, h, w, kh, kw, sh, sw, ph, pw, dh, dw, mm,
                                                     xq.data_ptr(), xs.data_ptr(), ent[0].data_ptr(), _stream(x)), "im2col_rowquant_z")
        ent[1] = False
        return xq, xs, (b, ho, wo)
    ws = torch.empty((b * h * w,), device=x.device, dtype=torch.int32)  # per-pixel channel-amax map (zeroed by the library)
    check(_lib.load().sdnq_hip_im2col_rowquant(x.data_ptr(), float_code(x.dtype), b, c, h, w, kh, kw, sh, sw, ph, pw, dh, dw, mm,
                                               xq.data_ptr(), xs.data_ptr(), ws.data_ptr(), _stream(x)), "im2col_rowquant")
    return xq, xs, (b, ho, wo)


def quantize_weight(weight2d: torch.Tensor, weights_dtype: str, group_size: int, positions: int = 1):
    """Float [N,K] weight -> (codes, scale [N,G] f32, zero_point [N,G] f32 | None) with the codes in the reference's storage
    (packed words shaped like the reference's packers return them, or [N,K] raw int8/uint8/int16/fp8/fp16...).
    HIP replacement of quantize_weight + pack_int / pack_float (quant_utils.py:28-56, packed_int/__init__.py:77-80,
    packed_float.py:27-82); `group_size` == K for row-wise.  positions = P > 1: conv weight flattened to [N, C_in * P],
    group_size counts channels, scales are [N, (C_in / group_size) * P]."""
    from .common import dtype_dict
    from . import packed as _packed
    _require_cuda(weight2d)
    if weight2d.dtype not in _FLOAT_CODE:
        raise _lib.SdnqHipError(f"quantize_weight: unsupported source dtype {weight2d.dtype}")
    w = weight2d if weight2d.stride(1) == 1 else weight2d.contiguous()
    n, k = w.shape
    ent = dtype_dict[weights_dtype]
    storage, kind, bits, ebits, mbits, native = _storage_kind(weights_dtype)
    g = (k // positions) // group_size * positions
    dev = w.device
    if storage == _lib.ST_PACKED_U8:
        raw = torch.empty((n * k // 8 * bits,), device=dev, dtype=torch.uint8)
    elif storage == _lib.ST_PACKED_I16:
        raw = torch.empty((n * k // 16 * bits,), device=dev, dtype=torch.int16)
    elif storage == _lib.ST_RAW8:
        raw = torch.empty((n, k), device=dev, dtype=torch.uint8)
    else:
        raw = torch.empty((n, k), device=dev, dtype=torch.int16)
    scale = torch.empty((n, g), device=dev, dtype=torch.float32)
    zp = torch.empty((n, g), device=dev, dtype=torch.float32) if ent["is_unsigned"] else None
    d = SdnqWeight(weight=raw.data_ptr(), scale=scale.data_ptr(), zero_point=_ptr(zp), svd_up=None, svd_down=None, n=n, k=k,
                   group_size=group_size, svd_rank=0, svd_dtype=0, storage=storage, kind=kind, bits=bits, exponent=ebits,
                   mantissa=mbits, native_float=native, positions=positions)
    check(_lib.load().sdnq_hip_quantize_weight(w.data_ptr(), float_code(w.dtype), w.stride(0), ctypes.byref(d), float(ent["min"]),
                                               float(ent["max"]), _stream(w)), "quantize_weight")
    if ent["is_packed"] and bits not in (8, 16):
        _g, words, _wb = _packed._GEOM[bits]
        codes = raw if words == 1 else raw.view(-1, words)
    elif ent["is_packed"]:  # custom float8 / float16 codes (pack_float returns uint8 / uint16)
        codes = raw.view(torch.uint16) if bits == 16 else raw
    else:
        codes = raw.view(ent["torch_dtype"]) if ent["torch_dtype"].itemsize == raw.element_size() else raw
    return codes, scale, zp
