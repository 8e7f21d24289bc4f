"""Config + load-time quantization of float Linear layers into the SDNQ state_dict layout.

API names are the reference's (quantizer.py: SDNQConfig :846, sdnq_quantize_layer_weight :66,
sdnq_quantize_layer :423, apply_sdnq_to_module :477, QuantizationMethod :60) so host code switches with an
import change; the module/tensor layout produced is byte-compatible with reference checkpoints
(SURVEY App. C), which tests/test_quantizer.py checks against the golden fixtures.  Only what feeds the Linear
hot path is implemented: no HF/diffusers quantizer plugin, no dynamic dtype search, no codebook, no
stochastic rounding, no conv/embedding (SURVEY 2, rows 12-16 marked out of scope).
"""
from __future__ import annotations

import os

from enum import Enum

import torch

from . import packed
from .common import conv_types, dtype_dict, linear_types, sdnq_version
from .dequantizer import SDNQDequantizer
from .forward import get_forward_func
from .layers import get_sdnq_wrapper_class
from .quant_utils import apply_hadamard, apply_svdquant, quantize_weight


class QuantizationMethod(str, Enum):
    SDNQ = "sdnq"
    SDNQ_TRAINING = "sdnq_training"


def get_quantized_matmul_dtype(weights_dtype: str, quantized_matmul_dtype: str | None = None) -> str:
    """Default matmul dtype per weight dtype (reference utils.py:203-214)."""
    if quantized_matmul_dtype is not None:
        return quantized_matmul_dtype
    ent = dtype_dict[weights_dtype]
    if ent["is_integer"]:
        return "uint8" if weights_dtype == "uint8" else "int8"
    return "float8_e4m3fn" if ent["num_bits"] < 16 else "float16"


def check_quantized_matmul_is_allowed(use_quantized_matmul: bool, output_channel_size: int, channel_size: int) -> bool:
    """reference utils.py:93-98."""
    return bool(use_quantized_matmul and output_channel_size >= 32 and channel_size >= 32
                and output_channel_size % 16 == 0 and channel_size % 16 == 0)


class SDNQConfig:
    """Quantization options; keyword names and defaults of the reference's SDNQConfig (quantizer.py:938-973)."""

    def __init__(self, weights_dtype: str = "int8", quantized_matmul_dtype: str | None = None, hadamard_group_size: int = 256,
                 group_size: int = 0, svd_rank: int = 32, svd_steps: int = 8, codebook_steps: int = 24,
                 dynamic_loss_threshold: float | None = None, use_svd: bool = False, use_hadamard: bool = False,
                 use_codebook: bool = False, use_grad_ckpt: bool = True, quant_conv: bool = False, quant_embedding: bool = False,
                 use_quantized_matmul: bool = False, use_quantized_matmul_conv: bool = False,
                 use_static_quantization: bool = True, use_dynamic_quantization: bool = False,
                 use_stochastic_rounding: bool = False, dequantize_fp32: bool = True, non_blocking: bool = False,
                 add_skip_keys: bool = True, minimum_allowed_numel: int = 16384, minimum_allowed_channel_size: int = 32,
                 modules_to_not_convert=None, modules_to_not_use_matmul=None, modules_dtype_dict=None,
                 modules_quant_config=None, quantization_device=None, return_device=None, sdnq_version: str | None = None,
                 is_training: bool = False, **kwargs):
        if weights_dtype not in dtype_dict:
            raise ValueError(f"SDNQ only support weight dtypes in {sorted(dtype_dict)} but found {weights_dtype}")
        if quantized_matmul_dtype is not None and quantized_matmul_dtype not in {"int8", "uint8", "fp8", "fp16", "float8_e4m3fn", "float16"}:
            raise ValueError(f"unsupported quantized_matmul_dtype {quantized_matmul_dtype}")
        for name in ("use_codebook", "use_dynamic_quantization", "use_stochastic_rounding", "quant_embedding",
                     "is_training"):
            if locals()[name]:
                raise NotImplementedError(f"SDNQConfig({name}=True) is outside the MI355X Linear hot path")
        self.weights_dtype = weights_dtype
        self.quantized_matmul_dtype = quantized_matmul_dtype
        self.hadamard_group_size = hadamard_group_size
        self.group_size = group_size
        self.svd_rank = svd_rank
        self.svd_steps = svd_steps
        self.codebook_steps = codebook_steps
        self.dynamic_loss_threshold = dynamic_loss_threshold
        self.use_svd = use_svd
        self.use_hadamard = use_hadamard
        self.use_codebook = use_codebook
        self.use_grad_ckpt = use_grad_ckpt
        self.quant_conv = quant_conv
        self.quant_embedding = quant_embedding
        self.use_quantized_matmul = use_quantized_matmul
        self.use_quantized_matmul_conv = use_quantized_matmul_conv
        self.use_static_quantization = use_static_quantization
        self.use_dynamic_quantization = use_dynamic_quantization
        self.use_stochastic_rounding = use_stochastic_rounding
        self.dequantize_fp32 = dequantize_fp32
        self.non_blocking = non_blocking
        self.add_skip_keys = add_skip_keys
        self.minimum_allowed_numel = minimum_allowed_numel
        self.minimum_allowed_channel_size = minimum_allowed_channel_size
        self.modules_to_not_convert = list(modules_to_not_convert or [])
        self.modules_to_not_use_matmul = list(modules_to_not_use_matmul or [])
        self.modules_dtype_dict = dict(modules_dtype_dict or {})
        self.modules_quant_config = dict(modules_quant_config or {})
        self.quantization_device = quantization_device
        self.return_device = return_device
        self.sdnq_version = globals()["sdnq_version"] if sdnq_version is None else sdnq_version
        self.is_training = is_training
        self.is_integer = dtype_dict[weights_dtype]["is_integer"]
        self.is_unsigned = dtype_dict[weights_dtype]["is_unsigned"]
        self.quant_method = QuantizationMethod.SDNQ

    def to_dict(self) -> dict:
        d = {k: v for k, v in self.__dict__.items()}
        d["quant_method"] = self.quant_method.value
        for k in ("quantization_device", "return_device"):
            d[k] = None if d[k] is None else str(d[k])
        return d

    @classmethod
    def from_dict(cls, config_dict: dict, **kwargs):
        cfg = dict(config_dict)
        for k in ("quant_method", "is_integer", "is_unsigned"):
            cfg.pop(k, None)
        cfg.update(kwargs)
        return cls(**cfg)


USE_HIP_QUANTIZER = os.environ.get("SDNQ_HIP_QUANTIZER", "1").lower() not in {"0", "false", "no"}
_HIP_QUANTIZER_SKIP = {"int1", "uint1", "bool", "float8_e8m0fnu", "float8_e4m3fnuz", "float8_e5m2fnuz"}


def _needs_requant(weights_dtype: str, matmul_dtype: str) -> bool:
    """Whether stored codes cannot be fed to the matmul as they are (reference quantizer.py:101-116)."""
    w, m = dtype_dict[weights_dtype], dtype_dict[matmul_dtype]
    if w["num_bits"] > m["num_bits"] or w["is_integer"] != m["is_integer"]:
        return True
    if w["is_unsigned"] and not m["is_integer"]:
        return True
    if w["is_packed"] and not w["is_integer"] and not m["is_integer"]:
        return w["num_bits"] >= m["num_bits"] or w["max"] > m["max"]
    return False


def _pick_group_size(group_size: int, channel_size: int, weights_dtype: str, is_linear: bool, has_svd: bool,
                     direct_matmul: bool) -> tuple[int, int]:
    """-> (group_size or -1, num_groups). Policy of reference quantizer.py:173-201."""
    if group_size == 0:
        if direct_matmul and dtype_dict[weights_dtype]["num_bits"] >= 6:
            return -1, 1
        p = 1 + dtype_dict[weights_dtype]["num_bits"] + (1 if is_linear else 0) + (1 if has_svd else 0)
        group_size = 2 ** p
    if group_size <= 0 or group_size >= channel_size:
        return -1, 1
    groups = channel_size // group_size
    while groups * group_size != channel_size:  # shrink the group count until it divides the channel size
        groups -= 1
        if groups <= 1:
            return -1, 1
        group_size = channel_size // groups
    return (int(group_size), int(groups)) if groups > 1 else (-1, 1)


@torch.no_grad()
def sdnq_quantize_layer_weight(weight: torch.Tensor, layer_class_name: str = "Linear", weights_dtype: str = "int8",
                               quantized_matmul_dtype: str | None = None, group_size: int = 0, hadamard_group_size: int = 256,
                               svd_rank: int = 32, svd_steps: int = 8, use_svd: bool = False, use_hadamard: bool = False,
                               use_quantized_matmul: bool = False, dequantize_fp32: bool = True,
                               torch_dtype: torch.dtype | None = None, **_unused):
    """Float [N,K] weight -> (SDNQDequantizer, {"weight","scale","zero_point","svd_up","svd_down"}).

    Order of operations as in the reference (quantizer.py:158-253): Hadamard -> SVD split -> grouping ->
    quantize -> (transpose for direct matmul) -> pack.
    """
    is_conv = layer_class_name in conv_types
    if layer_class_name not in linear_types and not is_conv:
        raise NotImplementedError(f"{layer_class_name}: only Linear and Conv1d / Conv2d / Conv3d layers are built for MI355X")
    weight = weight.detach()
    original_shape, original_stride = weight.shape, weight.stride()
    torch_dtype = weight.dtype if torch_dtype is None else torch_dtype
    n = weight.shape[0]
    channels = weight.shape[1] if is_conv else weight.shape[-1]  # the quantization axis (quantizer.py:121-123, 136-137)
    kpos = 1
    for d in weight.shape[2:]:
        kpos *= int(d)
    k = channels * kpos
    mm_dtype = get_quantized_matmul_dtype(weights_dtype, quantized_matmul_dtype)
    use_qmm = check_quantized_matmul_is_allowed(use_quantized_matmul, n, channels)
    requant = _needs_requant(weights_dtype, mm_dtype)
    ent = dtype_dict[weights_dtype]
    result_shape = None
    # conv weights feeding the matmul directly are flattened BEFORE quantization: one scale per output channel over all of
    # (C_in, kernel); every other conv layout keeps one scale per kernel position (quantizer.py:120-125)
    flat = is_conv and use_qmm and not requant and not ent["is_packed"]
    if flat:
        result_shape = weight.shape
        weight = weight.flatten(1, -1)

    if use_hadamard:
        weight, use_hadamard, hadamard_group_size = apply_hadamard(weight, hadamard_group_size, is_conv=is_conv)
    svd_up = svd_down = None
    if use_svd:
        weight, svd_up, svd_down = apply_svdquant(weight, rank=svd_rank, steps=svd_steps, dtype=torch_dtype)
        if use_qmm:  # the matmul branch consumes x @ svd_down then @ svd_up: store both transposed (:164-167)
            svd_up, svd_down = svd_up.t(), svd_down.t()

    group_size, groups = _pick_group_size(group_size, channels, weights_dtype, not is_conv, svd_up is not None,
                                          direct_matmul=use_qmm and not requant)
    dim = 1 if (is_conv and not flat) else -1
    if groups > 1:
        if flat:
            raise ValueError("group-wise scales cannot be combined with the flattened conv matmul layout (the reference fails too)")
        if result_shape is None:
            result_shape = weight.shape
        if is_conv:  # [N, C_in, *kernel] -> [N, groups, group_size, *kernel], reduce over group_size (quantizer.py:205-209)
            weight = weight.unflatten(1, (groups, group_size))
            dim = 2
        else:
            weight = weight.unflatten(-1, (groups, group_size))
    requant = requant or groups > 1
    transpose = use_qmm and not requant and not ent["is_packed"]
    positions = kpos if (is_conv and not flat) else 1
    # dequantize_fp32=False: scale / zero_point live in the model dtype and the weight is quantized against the ROUNDED scale
    # (quantizer.py:147-156 + quant_utils.py:33-43).  The float-matmul exclusion of the reference only bites without tensorwise
    # fp8 scaling, which gfx950 always uses (kernel_wrappers.use_tensorwise_fp8_matmul); 16-bit formats (max > 16384) keep fp32.
    scale_dtype = None
    if not dequantize_fp32 and ent["max"] <= 16384 and torch_dtype in (torch.bfloat16, torch.float16):
        scale_dtype = torch_dtype

    if weight.is_cuda and USE_HIP_QUANTIZER and weight.dtype in (torch.float32, torch.bfloat16, torch.float16) and k % 16 == 0 \
            and (ent["is_packed"] or ent["num_bits"] in (8, 16)) and weights_dtype not in _HIP_QUANTIZER_SKIP and scale_dtype is None:
        # GPU tensors: one HIP launch pair does scale/zero-point, quantize and pack (csrc/quantize.hip); the element order
        # [N][K] is the same for the plain, grouped, conv and transposed layouts, only the logical views differ
        from . import ops
        w2d = weight.reshape(n, k)
        unit = (group_size if groups > 1 else channels) if positions > 1 else (group_size if groups > 1 else k)
        q, scale, zero_point = ops.quantize_weight(w2d, weights_dtype, unit, positions=positions)
        quantized_weight_shape = torch.Size((k, n)) if transpose else weight.shape
        if positions > 1:
            sshape = (n, groups, 1, *weight.shape[3:]) if groups > 1 else (n, 1, *weight.shape[2:])
        elif groups > 1:
            sshape = (n, groups, 1)
        else:
            sshape = (1, n) if transpose else (n, 1)
        scale = scale.view(sshape)
        zero_point = None if zero_point is None else zero_point.view(sshape)
        if not ent["is_packed"]:
            q = q.t() if transpose else q.view(weight.shape)
        elif ent["num_bits"] in (8, 16):  # custom float8 / float16 codes keep the tensor shape (pack_float :75-80)
            q = q.view(quantized_weight_shape)
    else:
        q, scale, zero_point = quantize_weight(weight, dim, weights_dtype, dtype=scale_dtype)  # 16-bit scales: torch ops (load-time)
        if transpose:  # logical [K,N] with strides (1,K): the bytes stay [N][K] (prepare_weight_for_matmul on gfx950)
            q = q.t()
            scale = scale.t().contiguous()
            zero_point = None if zero_point is None else zero_point.t().contiguous()
        quantized_weight_shape = q.shape
        if ent["is_packed"]:
            q = packed.pack_int(q, weights_dtype) if ent["is_integer"] else packed.pack_float(q, weights_dtype)
        else:
            q = q.to(ent["torch_dtype"])

    dq = SDNQDequantizer(result_dtype=torch_dtype, result_shape=result_shape, original_shape=original_shape,
                         original_stride=original_stride, quantized_weight_shape=quantized_weight_shape,
                         weights_dtype=weights_dtype, quantized_matmul_dtype=mm_dtype, hadamard_group_size=hadamard_group_size,
                         group_size=group_size, svd_rank=svd_rank, svd_steps=svd_steps, codebook_steps=24,
                         use_quantized_matmul=use_qmm, re_quantize_for_matmul=requant, use_stochastic_rounding=False,
                         use_hadamard=bool(use_hadamard), use_codebook=False, layer_class_name=layer_class_name)
    return dq, {"weight": q, "scale": scale, "zero_point": zero_point, "svd_up": svd_up, "svd_down": svd_down}


def check_param_name_in(param_name: str, param_list) -> str | None:
    """Which entry of a module list names `param_name` (the matching rule of the reference's lists, utils.py:56-70): an entry that
    starts with "." is a prefix of the qualified name, otherwise it is the whole name, one of its dot-separated components, or a
    "*" pattern (".*" stands for a literal dot followed by anything)."""
    import re
    parts = param_name.split(".")
    for param in param_list:
        if not param:
            continue
        if param.startswith("."):
            if param_name.startswith(param[1:]):
                return param
            continue
        if param_name == param or param in parts or ("*" in param and re.match(param.replace(".*", "\\.*").replace("*", ".*"), param_name)):
            return param
    return None


def _minimum_dtype(weights_dtype: str, param_name: str, modules_dtype_dict: dict) -> str:
    """Per-module weight format (utils.py:125-147): a key is a dtype name, or "minimum_<N>bit(s)" / "minimum_uint<N>bits" -- then the
    module keeps the global format unless it is narrower than N bits."""
    for key, names in modules_dtype_dict.items():
        if check_param_name_in(param_name, names) is None:
            continue
        key = key.lower()
        if key.startswith("minimum") or key.endswith(("bit", "bits")):
            bits = key.removeprefix("minimum").removeprefix("-").removeprefix("_").removesuffix("bits").removesuffix("bit").removesuffix("-").removesuffix("_")
            unsigned = bits.startswith("uint")
            bits = bits.removeprefix("uint") if unsigned else bits.removeprefix("int")
            if dtype_dict[weights_dtype]["num_bits"] < int(bits):
                return ("uint" if (unsigned or int(bits) <= 4) else "int") + bits
        else:
            return key
    return weights_dtype


def _quant_kwargs(cfg: SDNQConfig, torch_dtype, param_name: str, layer_class_name: str = "Linear") -> dict:
    """Per-layer quantization arguments out of the model-wide config (utils.py:150-199): modules_quant_config overrides, the conv
    layers' own matmul switch, modules_dtype_dict, modules_to_not_use_matmul -- every list matched with `check_param_name_in`."""
    kw = dict(weights_dtype=cfg.weights_dtype, quantized_matmul_dtype=cfg.quantized_matmul_dtype, group_size=cfg.group_size,
              hadamard_group_size=cfg.hadamard_group_size, svd_rank=cfg.svd_rank, svd_steps=cfg.svd_steps,
              use_svd=cfg.use_svd, use_hadamard=cfg.use_hadamard, use_quantized_matmul=cfg.use_quantized_matmul,
              dequantize_fp32=cfg.dequantize_fp32, torch_dtype=torch_dtype)
    conv_mm = cfg.use_quantized_matmul_conv
    key = check_param_name_in(param_name, list(cfg.modules_quant_config.keys()))
    if key is not None:
        for k2, v2 in cfg.modules_quant_config[key].items():
            if k2 == "use_quantized_matmul_conv":
                conv_mm = v2
            elif k2 in kw:
                kw[k2] = v2
    if layer_class_name in conv_types:  # utils.py:188-189: convs follow their own matmul switch
        kw["use_quantized_matmul"] = conv_mm
    kw["weights_dtype"] = _minimum_dtype(kw["weights_dtype"], param_name, cfg.modules_dtype_dict)
    if check_param_name_in(param_name, cfg.modules_to_not_use_matmul) is not None:
        kw["use_quantized_matmul"] = False
    return kw


@torch.no_grad()
def sdnq_quantize_layer(layer: torch.nn.Module, quantization_config: SDNQConfig, torch_dtype: torch.dtype | None = None,
                        param_name: str = "", quant_kwargs: dict | None = None):
    """Quantize one Linear IN PLACE (the wrapper shares the layer's parameters) -> (SDNQLinear, config)."""
    if torch_dtype is None:
        torch_dtype = layer.weight.dtype
    name = layer.__class__.__name__
    if name not in linear_types and not (name in ("Conv1d", "Conv2d", "Conv3d") and quantization_config.quant_conv):  # quantizer.py:429-435
        quantization_config.modules_to_not_convert.append(param_name)
        return layer, quantization_config
    kw = quant_kwargs or _quant_kwargs(quantization_config, torch_dtype, param_name, name)
    layer.weight.requires_grad_(False)
    dev = layer.weight.device if quantization_config.return_device is None else quantization_config.return_device
    w = layer.weight if quantization_config.quantization_device is None else layer.weight.to(quantization_config.quantization_device)
    dq, tensors = sdnq_quantize_layer_weight(w, layer_class_name=name, **kw)
    layer.sdnq_dequantizer = dq
    layer = get_sdnq_wrapper_class(layer, get_forward_func(name, dq.quantized_matmul_dtype, dq.use_quantized_matmul))
    for key, value in tensors.items():  # (a meta skeleton -- load_sdnq_model -- keeps meta placeholders of the stored shapes and dtypes)
        setattr(layer, key, None if value is None else torch.nn.Parameter(value if value.is_meta else value.to(dev), requires_grad=False))
    if "_sdnq_hip_handle" in layer.__dict__:  # the tensors are in place now: decide how the layer traces under torch.compile
        from . import torch_ops
        torch_ops.layer_handle(layer)
    if kw["use_quantized_matmul"] and not dq.use_quantized_matmul and param_name not in quantization_config.modules_to_not_use_matmul:
        quantization_config.modules_to_not_use_matmul.append(param_name)
    return layer, quantization_config


@torch.no_grad()
def apply_sdnq_to_module(model: torch.nn.Module, quantization_config: SDNQConfig, torch_dtype: torch.dtype | None = None,
                         full_param_name: str = "", pre_quantized: bool = False):
    """Recursively replace eligible nn.Linear / conv children by SDNQ layers (reference quantizer.py:477-495).  pre_quantized: the
    model is the skeleton of a stored SDNQ checkpoint -- every Linear (and conv, with quant_conv) the config does not list in
    modules_to_not_convert WAS quantized, whatever its size (utils.py:73-91: the size rules only decide at quantization time)."""
    for child_name, child in list(model.named_children()):
        pname = f"{full_param_name}.{child_name}" if full_param_name else child_name
        cname = child.__class__.__name__
        if (cname == "Linear" or (cname in ("Conv1d", "Conv2d", "Conv3d") and quantization_config.quant_conv)) and getattr(child, "weight", None) is not None:
            wname = pname + ".weight"
            skip = check_param_name_in(wname, quantization_config.modules_to_not_convert) is not None
            big = pre_quantized or (child.weight.shape[-1 if cname == "Linear" else 1] >= quantization_config.minimum_allowed_channel_size
                                    and child.weight.numel() >= quantization_config.minimum_allowed_numel)
            if not skip and big and child.weight.dtype in (torch.float32, torch.float16, torch.bfloat16, torch.float64):
                child, quantization_config = sdnq_quantize_layer(child, quantization_config, torch_dtype=torch_dtype, param_name=wname)
                setattr(model, child_name, child)
            elif not skip:
                quantization_config.modules_to_not_convert.append(wname)
        else:
            apply_sdnq_to_module(child, quantization_config, torch_dtype=torch_dtype, full_param_name=pname, pre_quantized=pre_quantized)
    return model, quantization_config


def sdnq_post_load_quant(model: torch.nn.Module, weights_dtype: str = "int8", torch_dtype: torch.dtype | None = None,
                         quantization_config: SDNQConfig | None = None, pre_quantized: bool = False, **kwargs):
    """Quantize the Linear / conv layers of a loaded model in place (reference quantizer.py:498-600).  pre_quantized=True is the
    loader's use (loader.py:150): `model` is a skeleton (meta tensors), the layers become SDNQ layers with placeholders of the stored
    shapes, and nothing is computed."""
    cfg = quantization_config if quantization_config is not None else SDNQConfig(weights_dtype=weights_dtype, **kwargs)
    model, cfg = apply_sdnq_to_module(model, cfg, torch_dtype=torch_dtype, pre_quantized=pre_quantized)
    model.quantization_config = cfg
    model.quantization_method = QuantizationMethod.SDNQ
    return model
