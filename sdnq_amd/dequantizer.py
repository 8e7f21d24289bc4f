"""``SDNQDequantizer``: the static metadata record of a quantized layer + its dequantize entry points.

Field-for-field the reference's dataclass (dequantizer.py:279-387) so that ``torch.load`` / the loaders keep
working; ``__call__`` (dequantizer.py:389-429) and ``re_quantize_matmul`` (:351-387) run on the HIP kernels
(``sdnq_hip_dequant`` / ``sdnq_hip_requant``) instead of a chain of eager/Inductor ops.
"""
from __future__ import annotations

from dataclasses import dataclass

import torch

from . import ops
from .common import conv_types, dtype_dict, linear_types


@dataclass
class SDNQDequantizer:
    result_dtype: torch.dtype
    result_shape: torch.Size
    original_shape: torch.Size
    original_stride: list
    quantized_weight_shape: torch.Size
    weights_dtype: str
    quantized_matmul_dtype: str
    hadamard_group_size: int
    group_size: int
    svd_rank: int
    svd_steps: int
    codebook_steps: int
    use_quantized_matmul: bool
    re_quantize_for_matmul: bool
    use_stochastic_rounding: bool
    layer_class_name: str
    use_hadamard: bool
    use_codebook: bool
    is_packed: bool
    is_unsigned: bool
    is_integer: bool
    is_integer_matmul: bool

    def __init__(self, result_dtype, result_shape, original_shape, original_stride, quantized_weight_shape, weights_dtype,
                 quantized_matmul_dtype, hadamard_group_size, group_size, svd_rank, svd_steps, codebook_steps,
                 use_quantized_matmul, re_quantize_for_matmul, use_stochastic_rounding, use_hadamard, use_codebook,
                 layer_class_name):
        self.result_dtype = result_dtype
        self.result_shape = result_shape
        self.original_shape = original_shape
        self.original_stride = original_stride
        self.quantized_weight_shape = quantized_weight_shape
        self.weights_dtype = weights_dtype
        self.quantized_matmul_dtype = quantized_matmul_dtype
        self.hadamard_group_size = hadamard_group_size
        self.group_size = group_size
        self.svd_rank = svd_rank
        self.svd_steps = svd_steps
        self.codebook_steps = codebook_steps
        self.use_quantized_matmul = use_quantized_matmul
        self.re_quantize_for_matmul = re_quantize_for_matmul
        self.use_stochastic_rounding = use_stochastic_rounding
        self.use_hadamard = use_hadamard
        self.use_codebook = use_codebook
        self.layer_class_name = layer_class_name
        w, m = dtype_dict[weights_dtype], dtype_dict[quantized_matmul_dtype]
        self.num_bits, self.is_packed, self.is_integer, self.is_unsigned = w["num_bits"], w["is_packed"], w["is_integer"], w["is_unsigned"]
        self.num_bits_matmul, self.is_packed_matmul = m["num_bits"], m["is_packed"]
        self.is_integer_matmul, self.is_unsigned_matmul = m["is_integer"], m["is_unsigned"]

    # ---- geometry of the Linear this record describes -------------------------------------------
    @property
    def out_features(self) -> int:
        return int(self.original_shape[0])

    @property
    def in_features(self) -> int:
        """K of the matmul: in_features of a Linear, C_in * prod(kernel) of a conv weight (flatten(1, -1), quantizer.py:123)."""
        k = 1
        for d in self.original_shape[1:]:
            k *= int(d)
        return k

    @property
    def is_conv(self) -> bool:
        return self.layer_class_name in conv_types

    @property
    def kernel_positions(self) -> int:
        """P of the scale layout: conv weights that were NOT flattened before quantization carry one scale per
        (output channel, channel group, kernel position) (quantizer.py:120-123, 205-209); 1 otherwise."""
        if not self.is_conv or self.weight_is_transposed:
            return 1
        p = 1
        for d in self.original_shape[2:]:
            p *= int(d)
        return p

    @property
    def weight_is_transposed(self) -> bool:
        """The quantizer stores weight as logical [K,N] only for unpacked, non-re-quantized qmm layers (quantizer.py:228-244)."""
        return bool(self.use_quantized_matmul and not self.re_quantize_for_matmul and not self.is_packed)

    def quant_weight(self, weight, scale, zero_point=None, svd_up=None, svd_down=None) -> ops.QuantWeight:
        if self.use_codebook:
            raise NotImplementedError("use_codebook (Lloyd-Max LUT) is outside the MI355X hot path (SURVEY 8a note)")
        if self.layer_class_name not in linear_types and self.layer_class_name not in conv_types:
            raise NotImplementedError(f"{self.layer_class_name}: only Linear and Conv1d / Conv2d / Conv3d layers are built for MI355X")
        n, k, pos = self.out_features, self.in_features, self.kernel_positions
        group = self.group_size if self.group_size > 0 else k // pos
        return ops.make_quant_weight(self.weights_dtype, weight, scale, zero_point, svd_up, svd_down, n, k, group,
                                     transposed=self.weight_is_transposed, svd_transposed=bool(self.use_quantized_matmul),
                                     positions=pos)

    @torch.no_grad()
    def re_quantize_matmul(self, weight, scale, zero_point=None, svd_up=None, svd_down=None, hadamard=None,
                           non_hadamard: bool = True, skip_compile: bool = False):
        """fp32 dequant (Hadamard not undone) -> per-output-row quantization to the matmul dtype.
        Returns (weight [K,N] with strides (1,K), scale [1,N]) like the reference (dequantizer.py:166-174); for the uint8
        matmul dtype also the row zero points [1,N] (re_quantize_uint_mm, dequantizer.py:178-187)."""
        qw = self.quant_weight(weight, scale, zero_point)
        if self.quantized_matmul_dtype == "uint8":
            wq, ws, wzp = ops.requant_asym(qw)
            return wq.t(), ws.view(1, -1), wzp.view(1, -1)
        if self.quantized_matmul_dtype == "float16":  # re_quantize_fp_mm (dequantizer.py:190-200): float16 codes per output row, scale = amax / 65504
            if scale.dtype != torch.float32:
                raise NotImplementedError("re-quantization to float16 codes with 16-bit scales is not built")
            w16, ws = ops.rowquant_f16(ops.dequant(qw, torch.float32, 0))
            return w16.t(), ws.view(1, -1)
        wq, ws = ops.requant(qw, ops.mm_code(self.quantized_matmul_dtype))
        if scale.dtype in (torch.bfloat16, torch.float16):  # 16-bit scales: the row scale was computed in that dtype (exact cast)
            ws = ws.to(scale.dtype)
        return wq.t(), ws.view(1, -1)

    @torch.no_grad()
    def __call__(self, weight, scale, zero_point=None, svd_up=None, svd_down=None, hadamard=None,
                 skip_quantized_matmul: bool = False, non_hadamard: bool = False, skip_compile: bool = False,
                 dtype: torch.dtype | None = None) -> torch.Tensor:
        """Dequantize to [N,K] (or, for stored-transposed weights with skip_quantized_matmul=False, the [K,N] view the
        reference returns, dequantizer.py:28-29/65-66)."""
        if dtype is None:
            dtype = self.result_dtype
        qw = self.quant_weight(weight, scale, zero_point, svd_up, svd_down)
        had = self.hadamard_group_size if (self.use_hadamard and not non_hadamard) else 0
        w = ops.dequant(qw, dtype, had)
        if self.weight_is_transposed and not skip_quantized_matmul:
            return w.t()
        return w.view(self.original_shape) if tuple(self.original_shape) != tuple(w.shape) else w


torch.serialization.add_safe_globals([SDNQDequantizer])
