"""Load-time quantizer (host side): layouts and values vs the reference-built goldens."""
import numpy as np
import pytest
import torch

import sdnq_amd
from sdnq_amd import quantizer
from tests.golden_util import Case


# cases whose quantization is deterministic given the float weight (no randomized SVD, no float GEMM before rounding)
EXACT = ["int8_rowwise_noqmm_f32", "int8_rowwise_qmm_bf16", "int8_rowwise_qmm_f16_nobias", "fp8_qmm_bf16", "uint4_qmm_bf16",
         "int6_rowwise_packed_qmm_bf16", "uint7_rowwise_packed_qmm_bf16", "uint8_int8mm_qmm_bf16", "uint8_uint8mm_qmm_bf16",
         "fp4_e2m1_fp8mm_qmm_bf16", "int5_group32_noqmm_bf16", "uint3_noqmm_f16", "int8_group64_uint8mm_qmm_bf16",
         "uint4_uint8mm_qmm_bf16",
         # dequantize_fp32=False: scale / zero_point in the model dtype, weight quantized against the rounded scale
         "int8_rowwise_qmm_bf16_lpscale", "int8_rowwise_qmm_f16_lpscale_nobias", "uint4_noqmm_bf16_lpscale", "fp8_qmm_bf16_lpscale"]


@pytest.mark.parametrize("name", EXACT)
def test_quantizer_reproduces_reference_state_dict(name):
    c = Case(name)
    w = c.torch_tensor("w_float")
    dq, tensors = quantizer.sdnq_quantize_layer_weight(w, layer_class_name="Linear", **c.meta["cfg"])
    d = c.deq
    for f in ("weights_dtype", "group_size", "use_quantized_matmul", "re_quantize_for_matmul", "use_hadamard", "is_packed"):
        assert getattr(dq, f) == d[f], (name, f, getattr(dq, f), d[f])
    assert dq.quantized_matmul_dtype in (d["quantized_matmul_dtype"], {"fp8": "float8_e4m3fn"}.get(d["quantized_matmul_dtype"]))
    assert list(dq.quantized_weight_shape) == d["quantized_weight_shape"]
    for key in ("weight", "scale", "zero_point"):
        ref = c.torch_tensor(key)
        mine = tensors[key]
        assert (ref is None) == (mine is None), (name, key)
        if ref is None:
            continue
        assert tuple(mine.shape) == tuple(ref.shape), (name, key, mine.shape, ref.shape)
        if key != "weight":
            assert mine.dtype == ref.dtype, (name, key, mine.dtype, ref.dtype)
        if key == "weight" and mine.ndim == 2 and dq.weight_is_transposed:
            assert mine.stride() == (1, mine.shape[0])
        a = mine.contiguous().view(torch.uint8) if mine.dtype in (torch.float8_e4m3fn, torch.int8) else mine.contiguous()
        b = ref.contiguous().view(torch.uint8) if ref.dtype in (torch.float8_e4m3fn, torch.int8) else ref.contiguous()
        assert torch.equal(a.to(b.dtype) if a.dtype != b.dtype else a, b), (name, key)


def test_hadamard_and_svd_configs_have_the_reference_layout():
    torch.manual_seed(0)
    lin = torch.nn.Linear(512, 64).to(torch.bfloat16)
    layer, cfg = sdnq_amd.sdnq_quantize_layer(lin, sdnq_amd.SDNQConfig(weights_dtype="int4", use_svd=True, svd_rank=16, use_hadamard=True,
                                                                       use_quantized_matmul=True))
    dq = layer.sdnq_dequantizer
    assert dq.re_quantize_for_matmul and dq.use_hadamard and dq.hadamard_group_size == 256 and dq.group_size == 128
    assert layer.weight.dtype == torch.uint8 and layer.weight.numel() == 64 * 512 // 2
    assert tuple(layer.scale.shape) == (64, 4, 1) and layer.zero_point is None
    assert tuple(layer.svd_up.shape) == (16, 64) and tuple(layer.svd_down.shape) == (512, 16)   # transposed for qmm
    assert layer.forward_func is sdnq_amd.linear.quantized_linear_forward_int8_matmul
    small = torch.nn.Linear(40, 24)
    l2, cfg2 = sdnq_amd.sdnq_quantize_layer(small, sdnq_amd.SDNQConfig(weights_dtype="int8", use_quantized_matmul=True), param_name="small.weight")
    assert not l2.sdnq_dequantizer.use_quantized_matmul and "small.weight" in cfg2.modules_to_not_use_matmul  # utils.py:93-98


def test_apply_to_module_and_config_roundtrip():
    model = torch.nn.Sequential(torch.nn.Linear(256, 256), torch.nn.ReLU(), torch.nn.Linear(256, 64), torch.nn.Linear(8, 8))
    cfg = sdnq_amd.SDNQConfig(weights_dtype="uint4", use_quantized_matmul=True, minimum_allowed_numel=1024)
    model, cfg = sdnq_amd.apply_sdnq_to_module(model, cfg)
    assert isinstance(model[0], sdnq_amd.SDNQLinear) and isinstance(model[2], sdnq_amd.SDNQLinear)
    assert not isinstance(model[3], sdnq_amd.SDNQLinear) and "3.weight" in cfg.modules_to_not_convert
    keys = set(model.state_dict().keys())
    assert {"0.weight", "0.scale", "0.zero_point", "0.bias"} <= keys
    cfg2 = sdnq_amd.SDNQConfig.from_dict(cfg.to_dict())
    assert cfg2.weights_dtype == "uint4" and cfg2.use_quantized_matmul and cfg2.to_dict()["quant_method"] == "sdnq"
    with pytest.raises(NotImplementedError):
        sdnq_amd.SDNQConfig(use_codebook=True)


def test_apply_options_scale_dtype_rules():
    """apply_sdnq_options_to_model(dequantize_fp32=...) re-types scale / zero_point like the reference (loader.py:262-283)."""
    def model():
        m = torch.nn.Sequential(torch.nn.Linear(64, 64), torch.nn.Linear(64, 64)).to(torch.bfloat16)
        m, _ = sdnq_amd.apply_sdnq_to_module(m, sdnq_amd.SDNQConfig(weights_dtype="uint4", minimum_allowed_numel=16))
        m[1] = sdnq_amd.sdnq_quantize_layer(torch.nn.Linear(64, 64).to(torch.bfloat16), sdnq_amd.SDNQConfig(weights_dtype="int12"))[0]
        return m
    m = model()
    assert m[0].scale.dtype == torch.float32 and m[0].zero_point.dtype == torch.float32
    sdnq_amd.apply_sdnq_options_to_model(m)  # nothing asked: float32 scales stay
    assert m[0].scale.dtype == torch.float32
    before = m[0].scale.detach().clone()
    sdnq_amd.apply_sdnq_options_to_model(m, dequantize_fp32=False)
    assert m[0].scale.dtype == torch.bfloat16 and m[0].zero_point.dtype == torch.bfloat16
    assert torch.equal(m[0].scale, before.to(torch.bfloat16))
    assert m[1].scale.dtype == torch.float32  # formats wider than 8 bits keep float32 scales
    sdnq_amd.apply_sdnq_options_to_model(m, dtype=torch.float16)  # nothing asked about fp32, 16-bit scale: follows the result dtype
    assert m[0].scale.dtype == torch.float16 and m[0].sdnq_dequantizer.result_dtype == torch.float16
    sdnq_amd.apply_sdnq_options_to_model(m, dequantize_fp32=True)
    assert m[0].scale.dtype == torch.float32 and m[0].zero_point.dtype == torch.float32
    # conv layers follow the same scale-dtype rule (the reference applies it to every SDNQ layer, loader.py:262-283; round-2 advisor)
    conv = sdnq_amd.sdnq_quantize_layer(torch.nn.Conv2d(64, 64, 3, padding=1, groups=2).to(torch.bfloat16),
                                        sdnq_amd.SDNQConfig(weights_dtype="int8", quant_conv=True, use_quantized_matmul_conv=True))[0]
    cm = torch.nn.Sequential(conv)
    assert cm[0].scale.dtype == torch.float32
    sdnq_amd.apply_sdnq_options_to_model(cm, dequantize_fp32=False)
    assert cm[0].scale.dtype == torch.bfloat16 and cm[0].forward_func.__name__ == "quantized_conv_forward_int8_matmul"
    sdnq_amd.apply_sdnq_options_to_model(cm, dequantize_fp32=True)
    assert cm[0].scale.dtype == torch.float32
    # quantizer: dequantize_fp32=False quantizes against the rounded scale; 16-bit formats and float32 models keep float32
    lin = torch.nn.Linear(64, 64).to(torch.bfloat16)
    l8 = sdnq_amd.sdnq_quantize_layer(lin, sdnq_amd.SDNQConfig(weights_dtype="int8", dequantize_fp32=False))[0]
    assert l8.scale.dtype == torch.bfloat16
    l16 = sdnq_amd.sdnq_quantize_layer(lin, sdnq_amd.SDNQConfig(weights_dtype="int16", dequantize_fp32=False))[0]
    assert l16.scale.dtype == torch.float32
    l32 = sdnq_amd.sdnq_quantize_layer(torch.nn.Linear(64, 64), sdnq_amd.SDNQConfig(weights_dtype="int8", dequantize_fp32=False))[0]
    assert l32.scale.dtype == torch.float32


def _dtype_entries():
    import json
    import os
    from tests.golden_util import GOLD
    with open(os.path.join(GOLD, "dequant_dtypes.json")) as f:
        return json.load(f)["dtypes"]


@pytest.mark.parametrize("key", sorted(_dtype_entries().keys()))
def test_host_quantizer_reproduces_every_storage_dtype(key):
    """All 84 dtype x group fixtures: codes (incl. the reference's unmasked-overflow quirk of uint9..15), scales, zero points."""
    import os
    from tests.golden_util import GOLD
    ent = _dtype_entries()[key]
    z = np.load(os.path.join(GOLD, "dequant_dtypes.npz"))
    dq, t = quantizer.sdnq_quantize_layer_weight(torch.from_numpy(z["w_float"]), weights_dtype=ent["deq"]["weights_dtype"],
                                                 group_size=ent["deq"]["group_size"], use_quantized_matmul=False)
    a = t["weight"].contiguous()
    a = (a.to(torch.uint8) if a.dtype in (torch.int64, torch.bool) else a).view(torch.uint8).numpy().reshape(-1)
    b = z[key + ".weight"]
    b = b.astype(np.uint8) if b.dtype == np.int64 else b.reshape(-1).view(np.uint8)
    assert a.shape == b.shape and np.array_equal(a, b), (key, "codes")
    assert list(t["weight"].shape) == ent["tensors"]["weight"]["shape"]
    assert np.array_equal(t["scale"].numpy().view(np.uint32), z[key + ".scale"].view(np.uint32)), (key, "scale")
    if key + ".zero_point" in z:
        assert np.array_equal(t["zero_point"].numpy().view(np.uint32), z[key + ".zero_point"].view(np.uint32)), (key, "zero_point")
    else:
        assert t["zero_point"] is None


from tests.golden_util import ConvCase, conv_case_names  # noqa: E402


def _conv_quant_kwargs(c):
    cfg = dict(c.meta["cfg"])
    cfg.pop("quant_conv", None)
    cfg["use_quantized_matmul"] = cfg.pop("use_quantized_matmul_conv", False)
    return cfg


def check_conv_state_dict(c, tensors, dq, name):
    d = c.deq
    for f in ("weights_dtype", "group_size", "use_quantized_matmul", "re_quantize_for_matmul", "is_packed", "use_hadamard"):
        assert getattr(dq, f) == d[f], (name, f, getattr(dq, f), d[f])
    if d["use_hadamard"]:
        assert dq.hadamard_group_size == d["hadamard_group_size"], (name, dq.hadamard_group_size)
    assert list(dq.quantized_weight_shape) == d["quantized_weight_shape"]
    assert (None if dq.result_shape is None else list(dq.result_shape)) == d["result_shape"]
    for key in ("weight", "scale", "zero_point"):
        ref, mine = c.torch_tensor(key), tensors[key]
        assert (ref is None) == (mine is None), (name, key)
        if ref is None:
            continue
        mine = mine.cpu()
        assert tuple(mine.shape) == tuple(ref.shape), (name, key, mine.shape, ref.shape)
        a = mine.contiguous().view(torch.uint8) if mine.dtype in (torch.float8_e4m3fn, torch.int8) else mine.contiguous()
        b = ref.contiguous().view(torch.uint8) if ref.dtype in (torch.float8_e4m3fn, torch.int8) else ref.contiguous()
        assert torch.equal(a.to(b.dtype) if a.dtype != b.dtype else a, b), (name, key)


@pytest.mark.parametrize("name", [n for n in conv_case_names() if "svd" not in n])
def test_host_quantizer_reproduces_conv_state_dict(name):
    """Conv1d / Conv2d weights: flattened direct-matmul layout, per-kernel-position scales, grouped, asymmetric."""
    c = ConvCase(name)
    dq, tensors = quantizer.sdnq_quantize_layer_weight(c.torch_tensor("w_float"), layer_class_name=c.deq["layer_class_name"],
                                                       **_conv_quant_kwargs(c))
    check_conv_state_dict(c, tensors, dq, name)
