"""ONE predicate for "does the MI355X path compute this layer": `unsupported_reason(module)`.

`accelerate()` asks it per module BEFORE re-pointing anything: a layer the HIP forwards do not build keeps the `forward_func` it
came with (a reference-built model keeps working on the reference's own forward for that layer -- the reference's idiom for a
missing kernel is fall back + log.warning, kernel_wrappers.py:80-88), and one `warnings.warn` lists what was left alone.  The
forwards ask the same function when they build a layer's kernel state (`linear._state`), so the two can never disagree: a layer
quantized by THIS package in an unbuilt configuration still fails loudly, with the same sentence.

Only STATIC facts of the module are judged here (configuration, stored dtypes, conv geometry); what depends on the call (input
dtype / device) stays in the forwards.
"""
from __future__ import annotations

import torch

from .common import conv_transpose_types, conv_types, dtype_dict, embedding_types, linear_types


def _scale_dtype(module):
    sc = getattr(module, "scale", None)
    return None if sc is None else sc.dtype


def unsupported_reason(module) -> str | None:
    """None when the HIP forwards compute `module` (an SDNQ-quantized layer); otherwise the sentence that says why not."""
    dq = getattr(module, "sdnq_dequantizer", None)
    if dq is None:
        return "not an SDNQ layer (no sdnq_dequantizer)"
    cls = getattr(dq, "layer_class_name", None)
    if cls in embedding_types or cls in conv_transpose_types:
        return f"{cls}: only Linear and Conv1d / Conv2d / Conv3d layers are built for MI355X (embeddings and transposed convolutions are outside SURVEY 8)"
    if cls not in linear_types and cls not in conv_types:
        return f"{cls}: unknown layer class"
    if getattr(dq, "use_codebook", False):
        return "use_codebook (Lloyd-Max LUT) is outside the MI355X hot path (SURVEY 8a note)"
    if dq.weights_dtype not in dtype_dict or dq.quantized_matmul_dtype not in dtype_dict:
        return f"unknown dtype {dq.weights_dtype} / {dq.quantized_matmul_dtype}"
    mm = dtype_dict[dq.quantized_matmul_dtype]
    qmm = bool(dq.use_quantized_matmul)
    sdt = _scale_dtype(module)
    lp = sdt is not None and sdt not in (torch.float32, torch.float64)
    if lp and sdt != dq.result_dtype:
        return (f"scale dtype {sdt} differs from the layer's result dtype {dq.result_dtype}: 16-bit scales are built for the layout "
                "apply_sdnq_options_to_model(dequantize_fp32=False) produces")
    if qmm and not mm["is_integer"] and mm["num_bits"] != 8:
        # the float16 matmul (linear_fp16.py; round 6) is built for Linear layers: stored float codes or weights re-quantized to float16 codes,
        # with Hadamard rotation and SVD factors
        if dq.quantized_matmul_dtype != "float16" or cls not in linear_types:
            return f"quantized_matmul_dtype='{dq.quantized_matmul_dtype}' on {cls} is outside the MI355X hot path (SURVEY 8a note)"
        if lp:
            return "the float16 matmul with 16-bit scales (dequantize_fp32=False) is not built"
    w = dtype_dict[dq.weights_dtype]
    if lp and w["num_bits"] > 8:
        return "16-bit scales with formats wider than 8 bits are not built (the codes are not exact in the scale dtype)"
    uint8_mm = qmm and mm["is_integer"] and mm["is_unsigned"]
    if uint8_mm and lp:
        if sdt != torch.bfloat16:
            return ("the uint8 matmul with float16 scales is not built: the reference's float16 column sums (linear_uint8.py:63) overflow "
                    "from K = 512 on")
        if cls not in linear_types and int(getattr(module, "groups", 1)) != 1:
            return "the grouped uint8 conv matmul with 16-bit scales (dequantize_fp32=False) is not built"
    if cls in linear_types:
        return None
    # ---- conv layers
    groups = int(getattr(module, "groups", 1))
    if isinstance(getattr(module, "padding", 0), str):
        return "string padding modes ('same' / 'valid') are not supported by the reference's conv matmul either"
    nd = len(dq.original_shape) - 2
    if nd == 3:
        d = getattr(module, "dilation", (1, 1, 1))
        d = (d,) * 3 if isinstance(d, int) else tuple(d)
        if not (d[0] == d[1] == d[2]):
            return "Conv3d with unequal dilations: the reference's unfold sizes every axis with dilation[0] (forward.py:62-64)"
    if qmm and dq.is_packed and not dq.re_quantize_for_matmul:
        return "packed conv weights with a direct quantized matmul have no valid layout in the reference"
    if uint8_mm and getattr(module, "svd_up", None) is not None:
        return "the uint8 conv matmul with SVD factors is not built (its K * xzp * wzp term is rounded in the conv forwards' own order)"
    if groups != 1:
        if qmm:
            # K per group and N from original_shape, NOT from this package's in_features / out_features properties: the predicate runs on
            # the dequantizer a model CAME with -- the reference's dataclass has no such properties -- before anything is adopted
            k_group = 1
            for d in tuple(dq.original_shape)[1:]:  # original_shape[1] is C_in / groups
                k_group *= int(d)
            k_total, n = k_group * groups, int(dq.original_shape[0])
            if k_total % groups or n % groups or (k_total // groups) % 16 or (n // groups) % 8:
                return f"grouped conv matmul needs 16 | K per group and 8 | channels per group (got {k_total // groups}, {n // groups})"
            if getattr(module, "svd_up", None) is not None:
                return "grouped conv with SVD: the reference's per-group matmul has no valid form for it (its SVD product does not match the grouped weight)"
            if lp and (sdt != torch.bfloat16 or uint8_mm
                       or (getattr(module, "zero_point", None) is not None and not dq.re_quantize_for_matmul)):
                return ("grouped conv matmul with 16-bit scales is built for bfloat16 scales on the int8 / fp8 matmul without a weight zero point (float16: the reference "
                        "casts acc * input_scale to float16 before the weight scale, dequantizer.py:27, 63 -- an epilogue of its own)")
    return None


def require(module) -> None:
    """The forwards' side of the contract: raise the predicate's sentence."""
    why = unsupported_reason(module)
    if why is not None:
        raise NotImplementedError(why)
