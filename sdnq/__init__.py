"""Import-name drop-in: ``import sdnq`` served by the MI355X build (``sdnq_amd``).

Glue code written against Disty0/sdnq does ``from sdnq import SDNQConfig``, ``from sdnq.loader import load_sdnq_model`` or relies on
``import sdnq`` registering the "sdnq" quantizer with transformers / diffusers (reference src/sdnq/__init__.py:1-16,
quantizer.py:1085-1101).  This package gives those module paths -- ``sdnq``, ``sdnq.quantizer``, ``.loader``, ``.common``, ``.layers``,
``.dequantizer``, ``.forward``, ``.kernel_wrappers`` -- as thin views of ``sdnq_amd``; no arithmetic lives here.  It is only importable
where the repo root is on ``sys.path`` and the reference package is not installed ahead of it.
"""
import sdnq_amd as _impl
import sdnq_amd.hf_quantizer as _hf

_PUBLIC = {
    "QuantizationMethod": _impl.QuantizationMethod,
    "SDNQConfig": _hf.SDNQConfig,           # the QuantizationConfigMixin form (what from_pretrained(quantization_config=...) takes)
    "SDNQQuantizer": _hf.SDNQQuantizer,
    "apply_sdnq_to_module": _impl.apply_sdnq_to_module,
    "load_sdnq_model": _impl.load_sdnq_model,
    "save_sdnq_model": _impl.save_sdnq_model,
    "sdnq_post_load_quant": _impl.sdnq_post_load_quant,
    "sdnq_quantize_layer": _impl.sdnq_quantize_layer,
}
globals().update(_PUBLIC)
__all__ = sorted(_PUBLIC)
__version__ = _impl.sdnq_version
is_mi355x_build = True  # (lets glue code tell this build from the reference)
