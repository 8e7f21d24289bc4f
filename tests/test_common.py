"""Host-side tables and dispatch predicates vs the reference fixture."""
import json
import os

import pytest
import torch

import sdnq_amd
from sdnq_amd import forward, linear, quantizer
from sdnq_amd.common import dtype_dict
from tests.golden_util import GOLD


def test_dtype_table_equals_reference_table():
    ref = json.load(open(os.path.join(GOLD, "dtype_table.json")))
    assert set(ref) == set(dtype_dict)
    for name, ent in ref.items():
        mine = dtype_dict[name]
        for k, v in ent.items():
            mv = mine[k]
            if not isinstance(mv, (int, float, bool, str)):
                mv = str(mv).replace("torch.", "")
            assert mv == v, (name, k, mv, v)


@pytest.mark.parametrize("mm,qmm,expect", [
    ("int8", True, linear.quantized_linear_forward_int8_matmul), ("uint8", True, linear.quantized_linear_forward_uint8_matmul),
    ("float8_e4m3fn", True, linear.quantized_linear_forward_fp8_matmul), ("fp8", True, linear.quantized_linear_forward_fp8_matmul),
    ("float16", True, linear.quantized_linear_forward_fp16_matmul), ("int8", False, linear.quantized_linear_forward)])
def test_get_forward_func_dispatch(mm, qmm, expect):
    assert forward.get_forward_func("Linear", mm, qmm) is expect


def test_matmul_predicates():
    assert quantizer.check_quantized_matmul_is_allowed(True, 64, 256)
    assert not quantizer.check_quantized_matmul_is_allowed(True, 24, 256)      # N < 32
    assert not quantizer.check_quantized_matmul_is_allowed(True, 64, 40)       # K % 16 != 0
    assert not quantizer.check_quantized_matmul_is_allowed(False, 64, 256)
    assert quantizer.get_quantized_matmul_dtype("int4") == "int8"
    assert quantizer.get_quantized_matmul_dtype("uint8") == "uint8"
    assert quantizer.get_quantized_matmul_dtype("float4_e2m1fn") == "float8_e4m3fn"
    assert quantizer.get_quantized_matmul_dtype("float16") == "float16"
    assert quantizer.get_quantized_matmul_dtype("int4", "fp8") == "fp8"


def test_product_path_has_no_cpu_fallback():
    lin = torch.nn.Linear(64, 64).to(torch.bfloat16)
    layer, _ = sdnq_amd.sdnq_quantize_layer(lin, sdnq_amd.SDNQConfig(weights_dtype="int8", group_size=-1, use_quantized_matmul=True))
    with pytest.raises(sdnq_amd._lib.SdnqHipError):
        layer(torch.randn(40, 64).to(torch.bfloat16))
    with pytest.raises(sdnq_amd._lib.SdnqHipError):
        layer(torch.randn(2, 64).to(torch.bfloat16))


def test_product_never_imports_oracle():
    import subprocess
    import sys
    code = "import sys, sdnq_amd, sdnq_amd.parallel, sdnq_amd.shapes; assert not any(m.startswith('oracle') for m in sys.modules), 'oracle leaked'"
    subprocess.run([sys.executable, "-c", code], check=True, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "sdnq_amd")
    for dirpath, _, files in os.walk(root):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".sh")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt and "sdnq_oracle" not in txt, f
