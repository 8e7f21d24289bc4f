"""Loading of the golden fixtures captured from the reference (tests/golden/make_golden.py)."""
import json
import os

import numpy as np

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TAGS = {"bf16": "bf16", "f16": "f16", "float32": "f32"}


def case_names():
    return sorted(f[5:-5] for f in os.listdir(GOLD) if f.startswith("case_") and f.endswith(".json"))


def conv_case_names():
    return sorted(f[5:-5] for f in os.listdir(GOLD) if f.startswith("conv_") and f.endswith(".json"))


class Case:
    def __init__(self, name, prefix="case"):
        self.name = name
        self.meta = json.load(open(os.path.join(GOLD, f"{prefix}_{name}.json")))
        self.z = np.load(os.path.join(GOLD, f"{prefix}_{name}.npz"))
        self.deq = self.meta["deq"]
        if "N" in self.meta:
            self.N, self.K = self.meta["N"], self.meta["K"]
        else:  # conv fixtures: [C_out, C_in, *kernel] flattened (quantizer.py:123)
            shp = self.deq["original_shape"]
            self.N, self.K = int(shp[0]), int(np.prod(shp[1:]))
        self.tag = self.meta["dtype"]  # bf16 | f16 | f32

    def info(self, key):
        return self.meta["tensors"].get(key)

    def has(self, key):
        i = self.info(key)
        return i is not None and i["dtype"] != "none"

    def raw(self, key):
        return self.z[key] if self.has(key) else None

    def f32(self, key):
        """float tensor -> float32 values (bf16/f16 bit patterns decoded)."""
        from oracle import oracle as O
        if not self.has(key):
            return None
        tag = self.info(key)["dtype"]
        arr = self.z[key]
        if tag in ("bf16", "f16"):
            return O.from_bits(arr, tag)
        return arr.astype(np.float32)

    def tensor_tag(self, key):
        return TAGS.get(self.info(key)["dtype"], self.info(key)["dtype"])

    def ms(self):
        return sorted(int(k[2:]) for k in self.meta["tensors"] if k.startswith("x_"))

    def oracle_module(self):
        from oracle import oracle as O
        svd_tag = self.tensor_tag("svd_up") if self.has("svd_up") else "bf16"
        scale_tag = self.tensor_tag("scale")  # "f32", or the model dtype with dequantize_fp32=False
        return O.OracleLinear(self.deq, self.raw("weight"), self.f32("scale"), self.f32("zero_point"), self.f32("svd_up"),
                              self.f32("svd_down"), self.f32("bias"), svd_tag=svd_tag,
                              bias_tag=self.tensor_tag("bias") if self.has("bias") else None, N=self.N, K=self.K,
                              scale_tag=scale_tag)

    # ---- torch views for the product path -------------------------------------------------------
    def torch_tensor(self, key, device=None):
        import torch
        if not self.has(key):
            return None
        info = self.info(key)
        t = torch.from_numpy(np.ascontiguousarray(self.z[key]))
        tag = info["dtype"]
        view = {"bf16": torch.bfloat16, "f16": torch.float16, "fp8e4m3": torch.float8_e4m3fn, "fp8e5m2": torch.float8_e5m2,
                "bool": torch.bool}.get(tag)
        if view is not None:
            t = t.view(view)
        st = info.get("stride")
        if st is not None and t.ndim == 2 and st == [1, t.shape[0]] and t.shape[0] != 1:
            t = t.t().contiguous().t()  # restore the reference's transposed (1, K) strides
        if device is not None:
            if st is not None and t.ndim == 2 and st == [1, t.shape[0]] and t.shape[0] != 1:
                t = t.t().contiguous().to(device).t()
            else:
                t = t.to(device)
        return t


class ConvCase(Case):
    """conv_<name>.{json,npz}: Conv1d / Conv2d layers quantized and run by the reference (make_golden.run_conv_case)."""

    def __init__(self, name):
        super().__init__(name, prefix="conv")
        self.conv = self.meta["conv"]

    def inputs(self):
        return list(self.meta["inputs"])

    def torch_module(self, device):
        """An SDNQConv1d / SDNQConv2d holding exactly the reference's tensors."""
        import torch
        from sdnq_amd.forward import get_forward_func
        from sdnq_amd.layers import get_sdnq_wrapper_class
        from tests.modules_util import dequantizer_from_fields
        c = self.conv
        ctor = {1: torch.nn.Conv1d, 2: torch.nn.Conv2d, 3: torch.nn.Conv3d}[c["nd"]]
        skel = ctor(c["in_channels"], c["out_channels"], tuple(c["kernel_size"]), stride=tuple(c["stride"]), padding=tuple(c["padding"]),
                    dilation=tuple(c["dilation"]), groups=c["groups"], bias=c["bias"], padding_mode=c["padding_mode"])
        dq = dequantizer_from_fields(self.deq)
        skel.sdnq_dequantizer = dq
        mod = get_sdnq_wrapper_class(skel, get_forward_func(dq.layer_class_name, dq.quantized_matmul_dtype, dq.use_quantized_matmul))
        for key in ("weight", "scale", "zero_point", "svd_up", "svd_down", "bias"):
            t = self.torch_tensor(key, device=device)
            setattr(mod, key, None if t is None else torch.nn.Parameter(t, requires_grad=False))
        return mod


def attn_case_names():
    return sorted(f[5:-5] for f in os.listdir(GOLD) if f.startswith("attn_") and f.endswith(".json"))


class AttnCase:
    """Quantized-attention fixtures: outputs of the reference's Triton kernel run through Triton's interpreter
    (tests/golden/make_golden_attention.py)."""

    def __init__(self, name):
        self.name = name
        self.meta = json.load(open(os.path.join(GOLD, f"attn_{name}.json")))
        self.z = np.load(os.path.join(GOLD, f"attn_{name}.npz"))
        self.tag = self.meta["dtype"]
        self.kwargs = self.meta["kwargs"]

    def f32(self, key):
        from oracle import oracle as O
        tag = self.meta["tensors"][key]["dtype"]
        return O.from_bits(self.z[key], tag) if tag in ("bf16", "f16") else self.z[key].astype(np.float32)

    def raw(self, key):
        return self.z[key]

    def has(self, key):
        return key in self.meta["tensors"]

    def mask_array(self):
        """The attention mask as the oracle takes it: bool array, or float32 values of the additive mask; None without one."""
        if not self.has("mask"):
            return None
        return self.z["mask"].astype(np.bool_) if self.meta["tensors"]["mask"]["dtype"] == "bool" else self.f32("mask")

    def torch_tensor(self, key, device=None):
        import torch
        t = torch.from_numpy(np.ascontiguousarray(self.z[key]))
        view = {"bf16": torch.bfloat16, "f16": torch.float16, "bool": torch.bool}.get(self.meta["tensors"][key]["dtype"])
        if view is not None:
            t = t.view(view)
        return t.to(device) if device is not None else t
