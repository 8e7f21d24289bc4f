"""sdnq_amd -- MI355X (gfx950 / CDNA4) native implementation of SDNQ's quantized-Linear hot path.

Drop-in surface (names as in Disty0/sdnq): SDNQConfig, SDNQLinear, SDNQDequantizer, get_forward_func,
sdnq_quantize_layer, apply_sdnq_to_module, apply_sdnq_options_to_model; plus ``accelerate(model)`` which re-points a
model built by the reference package at these kernels.  All arithmetic runs in hand-written HIP kernels behind the
C ABI of ``include/sdnq_hip.h`` (``sdnq_amd/libsdnq_hip.so``); there is no CPU or eager fallback.
"""
from .common import dtype_dict, sdnq_version
from .dequantizer import SDNQDequantizer
from .forward import get_forward_func
from .kernel_wrappers import fp8_scaled_mm_func, int_scaled_mm_func
from .layers import SDNQLayer, SDNQLinear, get_sdnq_wrapper_class
from .linear import invalidate
from .capture import CapturedModel, capture
from . import torch_ops  # registers the sdnq_hip::* operators with torch.library
from .loader import accelerate, apply_sdnq_options_to_model, fuse_projections, link_layers, link_projections, load_sdnq_model, post_process_model, save_sdnq_model
from .quantizer import (QuantizationMethod, SDNQConfig, apply_sdnq_to_module, sdnq_post_load_quant, sdnq_quantize_layer,
                        sdnq_quantize_layer_weight)

__version__ = sdnq_version


def __getattr__(name):  # the transformers / diffusers plugin imports `transformers` (1-2 s): loaded on first use
    if name in ("SDNQQuantizer", "hf_quantizer"):
        import importlib
        hf_quantizer = importlib.import_module(__name__ + ".hf_quantizer")
        return hf_quantizer if name == "hf_quantizer" else hf_quantizer.SDNQQuantizer
    raise AttributeError(f"module 'sdnq_amd' has no attribute {name!r}")


__all__ = [
    "CapturedModel", "QuantizationMethod", "SDNQConfig", "SDNQDequantizer", "SDNQLayer", "SDNQLinear", "accelerate", "capture", "fuse_projections", "link_layers", "link_projections",
    "apply_sdnq_options_to_model", "apply_sdnq_to_module", "load_sdnq_model", "save_sdnq_model", "post_process_model", "dtype_dict", "fp8_scaled_mm_func", "get_forward_func",
    "get_sdnq_wrapper_class", "int_scaled_mm_func", "invalidate", "sdnq_post_load_quant", "sdnq_quantize_layer",
    "sdnq_quantize_layer_weight",
]
