"""Conv1d / Conv2d / Conv3d forwards of SDNQ-quantized layers as im2col + the Linear kernels.

Mirrors the reference's conv forwards (layers/conv/forward.py:80-81 ``quantized_conv_forward``,
layers/conv/conv_int8.py:94-123 ``quantized_conv_forward_int8_matmul``, conv_fp8.py): the input is unfolded to
``[B * H_out * W_out, C_in * kh * kw]`` (``process_conv_input``, forward.py:30-76) by ``sdnq_hip_im2col``; from there the
arithmetic is the Linear one -- row quantization, int8 / fp8 MFMA scaled matmul with the fused epilogue, or dequantize +
float GEMM -- and the ``[M, C_out]`` product is viewed back to NCHW (conv_int8.py:81-88).  The float branch uses this build's
own GEMM instead of the library convolution the reference calls (``_conv_forward``): same sum, fp32 accumulation.

Grouped convs (``groups > 1``; conv_int8.py:73-79, conv_fp8.py:56-60): the unfolded input keeps ALL input channels in a row (one
row scale over the whole row, as in the reference), and each group multiplies its column slice with its own output channels'
weight rows -- one launch per group on views (``sdnq_hip_scaled_mm_strided`` / ``sdnq_hip_linear_float_strided``), written straight
into its channel range of the output.  Built for the float forward, the plain int8 / fp8 matmuls and (round 4) the zero-point terms of
unsigned weights and of the uint8 matmul, computed from whole-row statistics exactly as the reference does
(``sdnq_hip_scaled_mm_lowrank_strided``); the reference's SVD term is not defined per group (its product has the wrong shape there)
and raises here.

Hadamard-rotated conv weights (quant_utils.py:222-236; conv_int8.py:52-53): the rotation groups run along the flattened (C_in, kernel)
axis, i.e. along the unfolded row -- the Linear kernels' Hadamard path as is (rotation fused into the row quantization; the float forward
un-rotates the weight rows).

Conv3d: see _geometry (depth taps gathered into the channel axis, then the 2-D path).

Not built (raise): fp16 matmul, Hadamard on grouped conv layers, Conv3d with unequal dilations.
"""
from __future__ import annotations

import torch

import os

from . import linear, ops

FUSED_CONV_QUANT = os.environ.get("SDNQ_HIP_FUSED_CONV_QUANT", "1").lower() not in {"0", "false", "no"}


CONV_PREFETCH = os.environ.get("SDNQ_HIP_CONV_PREFETCH", "1").lower() not in {"0", "false", "no"}


def _pair(v, n):
    return (int(v),) * n if isinstance(v, int) else tuple(int(e) for e in v)


def _geometry(self, input: torch.Tensor):
    """-> (input [B, C, H, W] (explicitly padded for non-zero padding modes), kernel, stride, padding, dilation, nd, depth_out).

    Conv1d is the H = 1 case.  Conv3d (forward.py:43-51, 59-73: padded explicitly, unfolded along depth, height and width, rows ordered
    (C_in, kd, kh, kw)) becomes a 2-D problem by gathering the kd depth taps of every output depth into the channel axis:
    Z[(b, do), (c, kd)] = x[b, c, do * sd + kd * dd] -- one strided copy (kd times the input, what the reference's own unfold
    materialises kd * kh * kw times) -- whose 2-D unfold has exactly the reference's row order; depth_out = D_out there, None otherwise."""
    if isinstance(self.padding, str):
        raise NotImplementedError("string padding modes ('same' / 'valid') are not supported by the reference's conv matmul either")
    nd = input.ndim - 2
    if nd not in (1, 2, 3):
        raise NotImplementedError(f"{input.ndim}-D conv input: only Conv1d / Conv2d / Conv3d are built")
    stride, padding, dilation = _pair(self.stride, nd), _pair(self.padding, nd), _pair(self.dilation, nd)
    kernel = tuple(int(k) for k in self.sdnq_dequantizer.original_shape[2:])
    if self.padding_mode != "zeros":  # forward.py:57-59: explicit padding first, then an unpadded unfold
        input = torch.nn.functional.pad(input, self._reversed_padding_repeated_twice, mode=self.padding_mode)
        padding = (0,) * nd
    if nd == 1:  # forward.py:24-27, 66-67: Conv1d is the H = 1 case
        input = input.unsqueeze(2)
        kernel, stride, padding, dilation = (1, kernel[0]), (1, stride[0]), (0, padding[0]), (1, dilation[0])
    depth_out = None
    if nd == 3:
        if not (dilation[0] == dilation[1] == dilation[2]):
            raise NotImplementedError("Conv3d with unequal dilations: the reference's unfold sizes every axis with dilation[0] (forward.py:62-64)")
        if padding[0]:  # height / width padding stays with the 2-D unfold
            input = torch.nn.functional.pad(input, (0, 0, 0, 0, padding[0], padding[0]))
        b, c, d, h, w = input.shape
        span = dilation[0] * (kernel[0] - 1) + 1
        depth_out = (d - span) // stride[0] + 1
        z = input.unfold(2, span, stride[0])  # [B, C, D_out, H, W, span]
        if dilation[0] > 1:
            z = z[..., ::dilation[0]]
        input = z.permute(0, 2, 1, 5, 3, 4).reshape(b * depth_out, c * kernel[0], h, w)  # (the copy)
        kernel, stride, padding, dilation = kernel[1:], stride[1:], padding[1:], dilation[1:]
    return input, kernel, stride, padding, dilation, nd, depth_out


def _folder(self, nd: int, b: int, ho: int, wo: int, depth_out=None):
    n = self.sdnq_dequantizer.out_features

    def fold(y2d: torch.Tensor) -> torch.Tensor:
        if nd == 1:
            return y2d.view(b, wo, n).transpose(1, 2).contiguous()  # conv_int8.py:81-82
        if nd == 3:
            return y2d.view(b // depth_out, depth_out, ho, wo, n).permute(0, 4, 1, 2, 3).contiguous()  # conv_int8.py:85-87
        return y2d.view(b, ho, wo, n).permute(0, 3, 1, 2).contiguous()  # conv_int8.py:83-84, 87
    return fold


def _unfold(self, input: torch.Tensor):
    """-> (x2d [M, K], fold) where fold(y2d [M, N]) gives the conv output in the reference's layout."""
    input, kernel, stride, padding, dilation, nd, depth_out = _geometry(self, input)
    x2d, (b, ho, wo) = ops.im2col(input, kernel, stride, padding, dilation)
    return x2d, _folder(self, nd, b, ho, wo, depth_out)


def _group_slices(self, k_total: int):
    """(K', N_g) of a grouped conv: columns g K' .. (g + 1) K' of the unfolded input meet output channels g N_g .. (g + 1) N_g."""
    g = int(self.groups)
    n = self.sdnq_dequantizer.out_features
    if k_total % g or n % g:
        raise RuntimeError(f"groups={g} does not divide {k_total} unfolded input columns / {n} output channels")
    return k_total // g, n // g


def _grouped_float_forward(self, x2d: torch.Tensor, fold):
    """F.conv*d(input, dequant(W), bias, groups=g) (layers/conv/forward.py:80-81): one float matmul per group on column views."""
    dq = self.sdnq_dequantizer
    st = linear._state(self)
    if x2d.dtype != dq.result_dtype:
        raise RuntimeError(f"expected input dtype {dq.result_dtype} (the layer's result_dtype) but got {x2d.dtype}")
    kg, ng = _group_slices(self, x2d.shape[1])
    wd = ops.dequant(st.qw, dq.result_dtype, dq.hadamard_group_size if dq.use_hadamard else 0)  # [N, K'], un-rotated
    out = torch.empty((x2d.shape[0], dq.out_features), device=x2d.device, dtype=x2d.dtype)
    aligned = (kg * x2d.element_size()) % 16 == 0
    for g in range(int(self.groups)):
        xg = x2d[:, g * kg:(g + 1) * kg]
        if not aligned:
            xg = xg.contiguous()
        bias = None if self.bias is None else self.bias[g * ng:(g + 1) * ng]
        ops.linear_float_into(xg, wd[g * ng:(g + 1) * ng], bias, out, g * ng)
    return fold(out)


@linear._no_grad
def quantized_conv_forward(self, input: torch.Tensor) -> torch.Tensor:
    x2d, fold = _unfold(self, input)
    if self.groups != 1:
        return _grouped_float_forward(self, x2d, fold)
    return fold(linear._float_forward(self, x2d, linear._state(self)))


def _conv_matmul_forward(self, input: torch.Tensor, mm: int) -> torch.Tensor:
    dq = self.sdnq_dequantizer
    if dq.is_packed and not dq.re_quantize_for_matmul:
        raise NotImplementedError("packed conv weights with a direct quantized matmul have no valid layout in the reference")
    if input.numel() / input.shape[2] < 32:  # conv_int8.py:96-97 (the reference's criterion, not the row count)
        return quantized_conv_forward(self, input)
    st = linear._state(self)
    wq, ws, zp = linear._prepare_mm_weights(self, st, mm)
    if self.groups != 1:
        return _grouped_matmul_forward(self, input, mm, st, wq, ws, zp)
    if FUSED_CONV_QUANT and st.svd_up is None and zp is None and st.qw.scale_dtype == torch.float32 and not dq.use_hadamard:
        # no SVD / zero-point terms: the float [M, K] matrix is never needed -- row scales straight from the image, then the
        # unfold writes the quantized operand (same values as im2col + rowquant)
        x4, kernel, stride, padding, dilation, nd, depth_out = _geometry(self, input)
        if kernel[0] * kernel[1] <= 25 and (x4.shape[2] * x4.shape[3]) % 8 == 0:
            xq, xs, (b, ho, wo) = ops.im2col_rowquant(x4, kernel, stride, padding, dilation, mm)
            if linear.PREFETCH_NEXT and st.mm_weight is wq and CONV_PREFETCH:  # the weight prefetch across layers (linear._PrefetchChain)
                linear._pf_launch(st, (wq,))
            # channel-major store fused into the GEMM epilogue; Conv3d: the rows of one image are its D_out * H_out * W_out positions
            imgs, px = (b, ho * wo) if nd != 3 else (b // depth_out, depth_out * ho * wo)
            if px % 8 == 0 and input.dtype != torch.float32:
                y = ops.scaled_mm_nchw(mm, xq, wq, xs, ws, self.bias, input.dtype, imgs, px)
                if nd == 3:
                    return y.view(imgs, -1, depth_out, ho, wo)
                return y.view(b, -1, wo) if nd == 1 else y.view(b, -1, ho, wo)
            return _folder(self, nd, b, ho, wo, depth_out)(ops.scaled_mm(mm, xq, wq, xs, ws, self.bias, input.dtype))
        x2d, (b, ho, wo) = ops.im2col(x4, kernel, stride, padding, dilation)
        fold = _folder(self, nd, b, ho, wo, depth_out)
    else:
        x2d, fold = _unfold(self, input)
    return fold(linear._quantized_matmul_forward(self, x2d, mm, small_batch_branch=False, cache_input=False))


def _grouped_matmul_forward(self, input: torch.Tensor, mm: int, st, wq, ws, zp) -> torch.Tensor:
    """conv_int8.py:73-79 / conv_fp8.py:56-60: the whole unfolded row is quantized with ONE scale, every group multiplies its column
    slice of the codes with its own weight rows, and the epilogue fma(acc * xs, ws, bias) is the ungrouped one."""
    if st.svd_up is not None:
        raise NotImplementedError("grouped conv with SVD: the reference's per-group matmul has no valid form for it (its SVD product does not "
                                  "match the grouped weight)")
    x4, kernel, stride, padding, dilation, nd, depth_out = _geometry(self, input)
    if st.qw.scale_dtype != torch.float32:
        return _grouped_lp_matmul_forward(self, x4, kernel, stride, padding, dilation, nd, depth_out, mm, st, wq, ws, zp)
    if zp is not None:
        return _grouped_zero_point_forward(self, x4, kernel, stride, padding, dilation, nd, depth_out, wq, ws, zp, asymmetric=False)
    # Hadamard-rotated weights (round 5; conv_int8.py:52-53): the WHOLE unfolded row is rotated in blocks of the rotation group -- it
    # divides C_in / groups (quant_utils.py:222-236), hence K', so no block straddles two conv groups
    had = self.sdnq_dequantizer.hadamard_group_size if self.sdnq_dequantizer.use_hadamard else 0
    if FUSED_CONV_QUANT and not had and kernel[0] * kernel[1] <= 25 and (x4.shape[2] * x4.shape[3]) % 8 == 0:
        xq, xs, (b, ho, wo) = ops.im2col_rowquant(x4, kernel, stride, padding, dilation, mm)
    else:
        x2d, (b, ho, wo) = ops.im2col(x4, kernel, stride, padding, dilation)
        xq, xs = ops.rowquant(x2d, mm, had)[:2]
    kg, ng = _group_slices(self, xq.shape[1])
    if had and kg % had:
        raise NotImplementedError(f"rotation group {had} does not divide the {kg} columns of a conv group")
    if kg % 16 or ng % 8:
        raise NotImplementedError(f"grouped conv matmul needs 16 | K per group and 8 | channels per group (got {kg}, {ng})")
    n = self.sdnq_dequantizer.out_features
    wq2, ws1 = wq.reshape(n, kg), ws.reshape(-1)
    imgs, pixels = (b, ho * wo) if nd != 3 else (b // depth_out, depth_out * ho * wo)  # Conv3d: an image = D_out * H_out * W_out rows
    nchw = pixels % 8 == 0 and input.dtype != torch.float32
    out = torch.empty((imgs, n, pixels) if nchw else (imgs * pixels, n), device=input.device, dtype=input.dtype)
    for g in range(int(self.groups)):
        bias = None if self.bias is None else self.bias[g * ng:(g + 1) * ng]
        ops.scaled_mm_into(mm, xq[:, g * kg:(g + 1) * kg], wq2[g * ng:(g + 1) * ng], xs, ws1[g * ng:(g + 1) * ng], bias, out, g * ng,
                           pixels if nchw else 0)
    if nchw:
        if nd == 3:
            return out.view(imgs, n, depth_out, ho, wo)
        return out.view(b, n, wo) if nd == 1 else out.view(b, n, ho, wo)
    return _folder(self, nd, b, ho, wo, depth_out)(out)


def _grouped_lp_matmul_forward(self, x4, kernel, stride, padding, dilation, nd, depth_out, mm: int, st, wq, ws, zp) -> torch.Tensor:
    """Grouped conv of a layer whose scale is stored in bfloat16 (dequantize_fp32=False; round 5): the whole unfolded row is quantized
    in bfloat16 (`quantize_int_mm_input(input, dtype=scale.dtype)`, conv_int8.py:64), `cat(int_mm per group).to(bf16).mul_(input_scale)`
    and addcmul(bias, ., scale) / .mul(scale) round once each (conv_int8.py:73-79, dequantizer.py:27, 63) -- per group exactly the
    bfloat16 epilogue of the ungrouped scaled matmul (sdnq_hip_scaled_mm_lp).  A compatibility mode: one launch per group on contiguous
    copies of the group's columns.  (float16 scales / zero-point terms: sdnq_amd.support names them unsupported.)"""
    if st.qw.scale_dtype != torch.bfloat16 or zp is not None or x4.dtype != torch.bfloat16:
        raise NotImplementedError("grouped conv matmul with 16-bit scales is built for bfloat16 layers without a weight zero point")
    x2d, (b, ho, wo) = ops.im2col(x4, kernel, stride, padding, dilation)
    xq, xs, _rowsum, _xrot = ops.rowquant_lp(x2d, mm, self.sdnq_dequantizer.hadamard_group_size if self.sdnq_dequantizer.use_hadamard else 0)
    kg, ng = _group_slices(self, xq.shape[1])
    if kg % 16 or ng % 8:
        raise NotImplementedError(f"grouped conv matmul needs 16 | K per group and 8 | channels per group (got {kg}, {ng})")
    n = self.sdnq_dequantizer.out_features
    wq2, ws1 = wq.reshape(n, kg), ws.reshape(-1)
    outs = []
    for g in range(int(self.groups)):
        sl = slice(g * ng, (g + 1) * ng)
        outs.append(ops.scaled_mm_lp(mm, xq[:, g * kg:(g + 1) * kg].contiguous(), wq2[sl], xs, ws1[sl], None if self.bias is None else self.bias[sl]))
    return _folder(self, nd, b, ho, wo, depth_out)(torch.cat(outs, dim=1))


def _grouped_zero_point_forward(self, x4, kernel, stride, padding, dilation, nd, depth_out, wq, ws, zp, asymmetric: bool, wcs=None):
    """Grouped conv whose epilogue carries zero-point terms (round 4).  As the reference computes it: the WHOLE unfolded row is
    quantized with one scale (and, for the uint8 matmul, one zero point), `zero_bias` is built from whole-row statistics --
    rowsum(xq) * xs * zp[n] (conv_int8.py:65-69) [+ colsum(w[n]) * ws[n] * xzp + K_row * (xzp * zp[n]), conv_uint8.py:58-66] [+ bias]
    -- and every group multiplies its column slice of the codes with its own weight rows; result = addcmul(zero_bias, acc * xs, ws)
    (conv_int8.py:73-79 / conv_uint8.py:70-79).  One launch per group on views, the general GEMM epilogue does the terms."""
    x2d, (b, ho, wo) = ops.im2col(x4, kernel, stride, padding, dilation)
    had = self.sdnq_dequantizer.hadamard_group_size if self.sdnq_dequantizer.use_hadamard else 0
    res = ops.rowquant(x2d, ops.MM_I8, had, want_rowsum=True, asymmetric=asymmetric)
    xq, xs, rowsum = res[0], res[1], res[2]
    xzp = res[4] if asymmetric else None
    kg, ng = _group_slices(self, xq.shape[1])
    if kg % 16 or ng % 8:
        raise NotImplementedError(f"grouped conv matmul needs 16 | K per group and 8 | channels per group (got {kg}, {ng})")
    n = self.sdnq_dequantizer.out_features
    wq2, ws1, zp1 = wq.reshape(n, kg), ws.reshape(-1), None if zp is None else zp.reshape(-1)  # (signed weights in the uint8 matmul: no weight zero point)
    wcs1 = None if wcs is None else wcs.reshape(-1)
    out = torch.empty((x2d.shape[0], n), device=x2d.device, dtype=x2d.dtype)
    for g in range(int(self.groups)):
        sl = slice(g * ng, (g + 1) * ng)
        # zp_k: the whole row's K; negative = the conv forwards' rounding order of the K * xzp * wzp term (conv_uint8.py:66)
        ops.scaled_mm_zp_into(ops.MM_I8, xq[:, g * kg:(g + 1) * kg], wq2[sl], xs, ws1[sl], None if self.bias is None else self.bias[sl],
                              None if zp1 is None else rowsum, None if zp1 is None else zp1[sl], xzp, None if wcs1 is None else wcs1[sl],
                              -xq.shape[1], out, g * ng)
    return _folder(self, nd, b, ho, wo, depth_out)(out)


@linear._no_grad
def quantized_conv_forward_int8_matmul(self, input: torch.Tensor) -> torch.Tensor:
    return _conv_matmul_forward(self, input, ops.MM_I8)


@linear._no_grad
def quantized_conv_forward_uint8_matmul(self, input: torch.Tensor) -> torch.Tensor:
    """layers/conv/conv_uint8.py:95-121: unfolded input through the asymmetric-activation int8 matmul."""
    if self.groups != 1:
        if input.numel() / input.shape[2] < 32:
            return quantized_conv_forward(self, input)
        st = linear._state(self)
        if st.svd_up is not None or st.qw.scale_dtype != torch.float32:
            raise NotImplementedError("grouped conv with SVD or 16-bit scales is not built")
        wq, ws, zp = linear._prepare_mm_weights(self, st, ops.MM_I8, asymmetric=True)
        kg = wq.numel() // self.sdnq_dequantizer.out_features
        wcs = wq.reshape(-1, kg).to(torch.int32).sum(dim=1).to(torch.float32).mul_(ws.reshape(-1))  # f32(colsum over the group's own K) * ws
        x4, kernel, stride, padding, dilation, nd, depth_out = _geometry(self, input)
        return _grouped_zero_point_forward(self, x4, kernel, stride, padding, dilation, nd, depth_out, wq, ws, zp, asymmetric=True, wcs=wcs)
    x2d, fold = _unfold(self, input)
    if input.numel() / input.shape[2] < 32:
        return fold(linear._float_forward(self, x2d, linear._state(self)))
    return fold(linear._uint8_matmul_forward(self, x2d, small_batch_branch=False, cache_input=False, conv_form=True))


@linear._no_grad
def quantized_conv_forward_fp8_matmul(self, input: torch.Tensor) -> torch.Tensor:
    return _conv_matmul_forward(self, input, ops.MM_FP8)
