"""Forward-function dispatch: same predicate tree as the reference's ``get_forward_func``
(forward.py:6-57, Linear branch :39-57), re-pointed at the HIP-backed forwards."""
from __future__ import annotations

from collections.abc import Callable

from .common import conv_transpose_types, conv_types, dtype_dict, embedding_types


def get_forward_func(layer_class_name: str, quantized_matmul_dtype: str, use_quantized_matmul: bool) -> Callable:
    if layer_class_name in embedding_types or layer_class_name in conv_transpose_types:
        raise NotImplementedError(
            f"{layer_class_name}: only Linear and Conv1d / Conv2d / Conv3d layers are built for MI355X (quant_embedding and transposed "
            "convolutions are outside SURVEY 8)")
    if layer_class_name in conv_types:  # forward.py:10-28
        from . import conv
        if use_quantized_matmul:
            ent = dtype_dict[quantized_matmul_dtype]
            if ent["is_integer"]:
                return conv.quantized_conv_forward_uint8_matmul if ent["is_unsigned"] else conv.quantized_conv_forward_int8_matmul
            if not ent["is_integer"] and ent["num_bits"] == 8:
                return conv.quantized_conv_forward_fp8_matmul
            # (float16: the reference's own conv_fp16 forward fails on the layers its quantizer builds -- `result_shape` is None for them,
            #  layers/conv/conv_fp16.py:98 -> forward.py:39 -- so there is nothing to be in parity with; Linear layers have it)
            raise NotImplementedError(f"conv matmul in {quantized_matmul_dtype} is not built (int8, uint8 and fp8 are)")
        return conv.quantized_conv_forward
    from . import linear
    if use_quantized_matmul:
        ent = dtype_dict[quantized_matmul_dtype]
        if ent["is_integer"]:
            if ent["is_unsigned"]:
                return linear.quantized_linear_forward_uint8_matmul
            return linear.quantized_linear_forward_int8_matmul
        if ent["num_bits"] == 8:
            return linear.quantized_linear_forward_fp8_matmul
        return linear.quantized_linear_forward_fp16_matmul
    return linear.quantized_linear_forward
