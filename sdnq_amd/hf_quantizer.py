"""transformers / diffusers quantizer plugin of this build: ``SDNQQuantizer`` + the config class the Auto* tables need.

Restates the five hooks the reference's plugin implements (quantizer.py:624-843; registration :1085-1101) on top of this build's own
loader pieces, so that ``AutoModel.from_pretrained(<SDNQ checkpoint>)`` and ``from_pretrained(..., quantization_config=SDNQConfig(...))``
work with only this package installed:

* a stored (pre-quantized) checkpoint: before the weights are read, every Linear (/ conv) the config does not exclude becomes an SDNQ
  layer with placeholders of the stored shapes and dtypes (``sdnq_post_load_quant(pre_quantized=True)``, what the reference does under
  ``init_empty_weights()``, quantizer.py:745-753); the framework then assigns ``weight`` / ``scale`` / ``zero_point`` / ``svd_up`` /
  ``svd_down`` tensor by tensor (they are ordinary parameters of the skeleton); afterwards the direct-matmul operands are re-laid out
  once (``post_process_model``, loader.py:199-217) and the layers are routed through the MI355X kernels (``accelerate``);
* a float checkpoint + ``quantization_config=``: the model is loaded as it is and quantized layer by layer afterwards
  (``sdnq_post_load_quant``; on a GPU by the HIP quantizer).  The reference quantizes each weight while it is being read
  (``create_quantized_param``, quantizer.py:669-730) -- same tensors, lower peak memory; not restated.

Importing this module imports ``transformers`` (1-2 s), so ``import sdnq_amd`` does not; ``sdnq_amd.SDNQQuantizer`` and the ``sdnq``
import-name package load it on demand.
"""
from __future__ import annotations

import os

import torch

from . import quantizer as _q
from .common import dtype_dict
from .quantizer import QuantizationMethod

try:
    from transformers.quantizers.base import HfQuantizer as _HfQuantizer
    from transformers.utils.quantization_config import QuantizationConfigMixin as _ConfigMixin
    HAVE_TRANSFORMERS = True
except ImportError:  # (the classes stay importable; registration is skipped)
    _HfQuantizer, _ConfigMixin, HAVE_TRANSFORMERS = object, object, False

try:
    from diffusers.quantizers.base import DiffusersQuantizer as _DiffusersQuantizer
    HAVE_DIFFUSERS = True
except ImportError:
    _DiffusersQuantizer, HAVE_DIFFUSERS = None, False


class SDNQConfig(_q.SDNQConfig, _ConfigMixin):
    """``sdnq_amd.quantizer.SDNQConfig`` as a ``QuantizationConfigMixin`` (what the Auto* tables of transformers / diffusers accept):
    the same keywords and defaults as the reference's (quantizer.py:846-1073); serialisation helpers come from the mixin."""

    @classmethod
    def from_dict(cls, config_dict: dict, return_unused_kwargs: bool = False, **kwargs):
        cfg = _q.SDNQConfig.from_dict.__func__(cls, config_dict, **kwargs)
        return (cfg, {}) if return_unused_kwargs else cfg

    def to_diff_dict(self) -> dict:  # (the mixin's version instantiates the class without arguments; every keyword here has a default)
        base = SDNQConfig().to_dict()
        return {k: v for k, v in self.to_dict().items() if k not in base or base[k] != v}


_BASES = ((_DiffusersQuantizer,) if HAVE_DIFFUSERS else ()) + (_HfQuantizer,)


class SDNQQuantizer(*_BASES):
    """Quantizer plugin for SDNQ checkpoints / on-the-fly SDNQ quantization (reference quantizer.py:624-843)."""

    requires_parameters_quantization = True
    use_keep_in_fp32_modules = True
    requires_calibration = False
    required_packages = None
    torch_dtype = None

    def __init__(self, quantization_config, **kwargs):
        if isinstance(quantization_config, dict):
            quantization_config = SDNQConfig.from_dict(quantization_config)
        self.quantization_config = quantization_config
        self.pre_quantized = kwargs.pop("pre_quantized", True)
        self.modules_to_not_convert = list(getattr(quantization_config, "modules_to_not_convert", None) or [])

    def __str__(self) -> str:
        return f"SDNQQuantizer(torch_dtype={self.torch_dtype}, pre_quantized={self.pre_quantized})"

    # ---- environment / dtype ---------------------------------------------------------------------------------------------------------
    def validate_environment(self, *args, **kwargs):
        if self.quantization_config.is_training:
            raise NotImplementedError("SDNQ training layers are outside the MI355X inference hot path")

    def update_torch_dtype(self, torch_dtype):
        self.torch_dtype = torch_dtype
        return torch_dtype

    def update_dtype(self, dtype):
        return self.update_torch_dtype(dtype)

    def adjust_target_dtype(self, target_dtype):
        return dtype_dict[self.quantization_config.weights_dtype]["target_dtype"]

    def adjust_max_memory(self, max_memory):
        return {key: val * 0.80 for key, val in max_memory.items()}

    def get_accelerator_warm_up_factor(self) -> int:
        return 32 // dtype_dict[self.quantization_config.weights_dtype]["num_bits"]

    def get_cuda_warm_up_factor(self) -> int:
        return self.get_accelerator_warm_up_factor()

    # ---- parameter hooks: every stored tensor is an ordinary parameter of the skeleton, nothing is converted while loading ---------------
    def check_if_quantized_param(self, model, param_value, param_name, *args, **kwargs) -> bool:
        return False

    def check_quantized_param(self, *args, **kwargs) -> bool:
        return False

    def param_needs_quantization(self, model, param_name, *args, **kwargs) -> bool:
        return False

    # ---- the two model hooks -----------------------------------------------------------------------------------------------------------
    def _process_model_before_weight_loading(self, model, device_map=None, keep_in_fp32_modules=None, **kwargs):
        cfg = self.quantization_config
        if self.pre_quantized:
            # the stored layers' records are functions of the config and of the layers' shapes: build them on the skeleton
            cfg.quantization_device = None
            cfg.return_device = None
            cfg.non_blocking = False
            cfg.add_skip_keys = False
            _q.sdnq_post_load_quant(model, torch_dtype=self.torch_dtype, pre_quantized=True, quantization_config=cfg)
        elif keep_in_fp32_modules:
            cfg.modules_to_not_convert = list(cfg.modules_to_not_convert or []) + list(keep_in_fp32_modules)
        return model

    def _process_model_after_weight_loading(self, model, **kwargs):
        from .loader import accelerate, post_process_model
        cfg = self.quantization_config
        if not self.pre_quantized:
            with torch.no_grad():
                _q.sdnq_post_load_quant(model, torch_dtype=self.torch_dtype, quantization_config=cfg)
        model.quantization_config = cfg
        model.quantization_method = QuantizationMethod.SDNQ
        if hasattr(model, "config"):
            try:
                model.config.quantization_config = cfg
            except Exception:  # noqa: BLE001
                pass
        model = post_process_model(model)
        if any(p.is_cuda for p in model.parameters()):
            accelerate(model)
        return model

    # ---- serialisation -------------------------------------------------------------------------------------------------------------------
    def get_state_dict_and_metadata(self, state_dict, **kwargs):
        if isinstance(state_dict, torch.nn.Module):  # transformers
            return None, {}
        return state_dict, {}  # diffusers

    def is_serializable(self, *args, **kwargs) -> bool:
        return not self.quantization_config.is_training

    @property
    def supports_safetensors_serialization(self) -> bool:
        return self.is_serializable()

    @property
    def is_trainable(self) -> bool:
        return bool(self.quantization_config.is_training)

    @property
    def is_qat_trainable(self) -> bool:
        return self.is_trainable

    @property
    def is_compileable(self) -> bool:
        return True

    def _dequantize(self, model, dtype=None):
        raise NotImplementedError("dequantizing a whole SDNQ model is not part of the MI355X hot path; load the float checkpoint instead")


def register(force: bool = False) -> list:
    """Put ``SDNQQuantizer`` / ``SDNQConfig`` into the Auto* tables under "sdnq" (reference quantizer.py:1085-1101: transformers unless
    SDNQ_REGISTER_TRANSFORMERS=0, diffusers only when SDNQ_REGISTER_DIFFUSERS=1 -- diffusers >= 0.40 ships its own entry).  An entry
    that is already there (the reference package imported first) is left alone unless `force`.  Returns the frameworks touched."""
    done = []
    off = {"0", "false", "no"}
    if HAVE_TRANSFORMERS and os.environ.get("SDNQ_REGISTER_TRANSFORMERS", "1").lower() not in off:
        import transformers.quantizers.auto as auto
        for name in ("sdnq", "sdnq_training"):
            if force or name not in auto.AUTO_QUANTIZER_MAPPING:
                auto.AUTO_QUANTIZER_MAPPING[name] = SDNQQuantizer
                auto.AUTO_QUANTIZATION_CONFIG_MAPPING[name] = SDNQConfig
        done.append("transformers")
    if HAVE_DIFFUSERS and os.environ.get("SDNQ_REGISTER_DIFFUSERS", "0").lower() not in off:
        import diffusers.quantizers.auto as dauto
        for name in ("sdnq", "sdnq_training"):
            if force or name not in dauto.AUTO_QUANTIZER_MAPPING:
                dauto.AUTO_QUANTIZER_MAPPING[name] = SDNQQuantizer
                dauto.AUTO_QUANTIZATION_CONFIG_MAPPING[name] = SDNQConfig
        done.append("diffusers")
    return done


register()
