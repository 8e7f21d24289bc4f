"""Build product-path modules (sdnq_amd.SDNQLinear) from golden fixtures or from scratch."""
import torch

from sdnq_amd.dequantizer import SDNQDequantizer
from sdnq_amd.forward import get_forward_func
from sdnq_amd.layers import SDNQLinear

TORCH_DT = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32, "bfloat16": torch.bfloat16,
            "float16": torch.float16, "float32": torch.float32}


def dequantizer_from_fields(d: dict) -> SDNQDequantizer:
    return SDNQDequantizer(
        result_dtype=TORCH_DT[d["result_dtype"]], result_shape=None if d["result_shape"] is None else torch.Size(d["result_shape"]),
        original_shape=torch.Size(d["original_shape"]), original_stride=list(torch.empty(d["original_shape"]).stride()),
        quantized_weight_shape=torch.Size(d["quantized_weight_shape"]), weights_dtype=d["weights_dtype"],
        quantized_matmul_dtype=d["quantized_matmul_dtype"], hadamard_group_size=d["hadamard_group_size"],
        group_size=d["group_size"], svd_rank=d["svd_rank"], svd_steps=8, codebook_steps=24,
        use_quantized_matmul=d["use_quantized_matmul"], re_quantize_for_matmul=d["re_quantize_for_matmul"],
        use_stochastic_rounding=False, use_hadamard=d["use_hadamard"], use_codebook=d["use_codebook"],
        layer_class_name=d["layer_class_name"])


def module_from_case(case, device) -> SDNQLinear:
    """An SDNQLinear holding exactly the tensors the reference's quantizer produced (same logical layouts/strides)."""
    dq = dequantizer_from_fields(case.deq)
    skeleton = torch.nn.Linear(8, 8, bias=False)  # parameters are replaced below
    skeleton.in_features, skeleton.out_features = case.K, case.N
    skeleton.sdnq_dequantizer = dq
    mod = SDNQLinear(skeleton, get_forward_func("Linear", dq.quantized_matmul_dtype, dq.use_quantized_matmul))
    for key in ("weight", "scale", "zero_point", "svd_up", "svd_down", "bias"):
        t = case.torch_tensor(key, device=device)
        setattr(mod, key, None if t is None else torch.nn.Parameter(t, requires_grad=False))
    return mod


def to_f32_numpy(t: torch.Tensor):
    return t.detach().float().cpu().numpy()


def oracle_from_module(mod):
    """OracleLinear mirroring a live SDNQLinear / SDNQConv module (tensors copied to numpy in their logical layouts)."""
    import numpy as np
    from oracle import oracle as O
    dq = mod.sdnq_dequantizer
    deq = {f: getattr(dq, f) for f in ("weights_dtype", "quantized_matmul_dtype", "hadamard_group_size", "group_size", "svd_rank",
                                       "use_quantized_matmul", "re_quantize_for_matmul", "use_hadamard", "use_codebook", "is_packed",
                                       "is_unsigned", "is_integer", "layer_class_name")}
    deq.update(result_dtype=str(dq.result_dtype).replace("torch.", ""), quantized_weight_shape=list(dq.quantized_weight_shape),
               original_shape=list(dq.original_shape), result_shape=None if dq.result_shape is None else list(dq.result_shape))

    def raw(t):
        if t is None:
            return None
        t = t.detach().cpu()
        if t.dtype in (torch.bfloat16, torch.float16):
            return t.float().numpy()
        if t.dtype in (torch.float8_e4m3fn, torch.float8_e5m2):
            return t.view(torch.uint8).numpy()
        if not t.is_contiguous():  # transposed matmul layout: logical [K, N] values
            return np.ascontiguousarray(t.numpy())
        return t.numpy()

    tag = {torch.bfloat16: "bf16", torch.float16: "f16", torch.float32: "f32"}
    svd_up = getattr(mod, "svd_up", None)
    return O.OracleLinear(deq, raw(mod.weight), raw(mod.scale), raw(getattr(mod, "zero_point", None)), raw(svd_up),
                          raw(getattr(mod, "svd_down", None)), raw(mod.bias), svd_tag=tag[svd_up.dtype] if svd_up is not None else "bf16",
                          bias_tag=tag[mod.bias.dtype] if mod.bias is not None else None, N=dq.out_features, K=dq.in_features,
                          scale_tag=tag[mod.scale.dtype])
