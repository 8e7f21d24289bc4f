"""The import-name drop-in (`import sdnq` served by sdnq_amd) and the transformers plugin of this build.

SURVEY 8(b) shape (i): glue that does `from sdnq import SDNQConfig` / relies on the "sdnq" entry of transformers' Auto* tables must reach
this build with no reference installed.  The fixture tests/golden/checkpoint_hf_tiny was written by the REFERENCE's own plugin
(tests/golden/make_golden_hf.py: a 2-layer LlamaForCausalLM quantized while loading, saved with save_pretrained); here it is loaded by
`AutoModelForCausalLM.from_pretrained` through THIS build's SDNQQuantizer (reference quantizer.py:624-843 restated in
sdnq_amd/hf_quantizer.py).  Every test runs in a fresh interpreter: the Auto* tables are process-global.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CKPT = os.path.join(ROOT, "tests", "golden", "checkpoint_hf_tiny")


def run_py(code: str, timeout=600):
    env = dict(os.environ)
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    return r.stdout


def test_import_name_package_exports_what_the_reference_exports():
    out = run_py("""
import json, sdnq, sdnq_amd
import sdnq.quantizer, sdnq.loader, sdnq.common, sdnq.layers, sdnq.dequantizer, sdnq.forward, sdnq.kernel_wrappers
assert sdnq.is_mi355x_build and 'reference' not in sdnq.__file__
assert sdnq.load_sdnq_model is sdnq_amd.load_sdnq_model and sdnq.loader.load_sdnq_model is sdnq_amd.load_sdnq_model
assert sdnq.layers.SDNQLinear is sdnq_amd.SDNQLinear and sdnq.dequantizer.SDNQDequantizer is sdnq_amd.SDNQDequantizer
assert sdnq.forward.get_forward_func is sdnq_amd.get_forward_func and sdnq.common.dtype_dict is sdnq_amd.dtype_dict
assert issubclass(sdnq.SDNQConfig, sdnq_amd.SDNQConfig) and sdnq.quantizer.SDNQQuantizer is sdnq.SDNQQuantizer
import transformers.quantizers.auto as auto
assert auto.AUTO_QUANTIZER_MAPPING['sdnq'] is sdnq.SDNQQuantizer and auto.AUTO_QUANTIZATION_CONFIG_MAPPING['sdnq'] is sdnq.SDNQConfig
cfg = sdnq.SDNQConfig(weights_dtype='uint4', use_quantized_matmul=True)
d = json.loads(cfg.to_json_string(use_diff=False))
assert d['quant_method'] == 'sdnq' and d['weights_dtype'] == 'uint4'
assert sdnq.SDNQConfig.from_dict(d).to_dict() == cfg.to_dict()
print(json.dumps(sorted(sdnq.__all__)))
""")
    # the reference's public names (src/sdnq/__init__.py:7-16)
    assert json.loads(out.strip().splitlines()[-1]) == sorted(["QuantizationMethod", "SDNQConfig", "SDNQQuantizer", "apply_sdnq_to_module", "load_sdnq_model",
                                                                "save_sdnq_model", "sdnq_post_load_quant", "sdnq_quantize_layer"])


def test_from_pretrained_builds_sdnq_layers_from_a_reference_written_checkpoint():
    """CPU: the skeleton conversion + tensor assignment (no kernel runs).  Every stored tensor arrives bit for bit in an sdnq_amd layer
    whose record (weights dtype, group size, matmul flags) is the one the reference's loader rebuilds."""
    out = run_py(f"""
import json, torch, transformers, sdnq, sdnq_amd
from safetensors import safe_open
m = transformers.AutoModelForCausalLM.from_pretrained({CKPT!r}, dtype=torch.float32)
layers = {{n: mod for n, mod in m.named_modules() if hasattr(mod, 'sdnq_dequantizer')}}
assert len(layers) == 14 and all(type(mod) is sdnq_amd.SDNQLinear for mod in layers.values()), sorted(layers)
assert type(m.lm_head) is torch.nn.Linear and m.quantization_method == sdnq.QuantizationMethod.SDNQ
sd = dict(m.state_dict())
with safe_open({CKPT + '/model.safetensors'!r}, 'pt') as f:
    keys = sorted(f.keys())
    for k in keys:
        t = f.get_tensor(k)
        assert k in sd and sd[k].dtype == t.dtype and sd[k].shape == t.shape and torch.equal(sd[k], t), k
assert sorted(sd) == keys
dq = layers['model.layers.0.mlp.down_proj'].sdnq_dequantizer
assert dq.weights_dtype == 'uint4' and dq.is_packed and layers['model.layers.0.mlp.down_proj'].weight.dtype == torch.uint8
dq = layers['model.layers.1.self_attn.q_proj'].sdnq_dequantizer
assert dq.weights_dtype == 'int8' and dq.use_quantized_matmul and not dq.re_quantize_for_matmul
print('ok', len(keys))
""")
    assert out.strip().splitlines()[-1].startswith("ok")


def test_quantization_config_argument_quantizes_a_float_checkpoint(tmp_path):
    """`from_pretrained(float checkpoint, quantization_config=SDNQConfig(...))` with this build only: the layers come out as the
    load-time quantizer of sdnq_amd makes them (the same tensors `sdnq_post_load_quant` gives the same float model)."""
    out = run_py(f"""
import torch, transformers, sdnq, sdnq_amd
torch.manual_seed(3)
cfg = transformers.LlamaConfig(hidden_size=64, intermediate_size=128, num_hidden_layers=1, num_attention_heads=4, num_key_value_heads=4, vocab_size=64,
                               max_position_embeddings=32, tie_word_embeddings=False)
model = transformers.LlamaForCausalLM(cfg).to(torch.float32)
model.save_pretrained({str(tmp_path)!r})
qcfg = sdnq.SDNQConfig(weights_dtype='int8', group_size=0, use_quantized_matmul=True, minimum_allowed_numel=4096, modules_to_not_convert=['lm_head'])
q = transformers.AutoModelForCausalLM.from_pretrained({str(tmp_path)!r}, quantization_config=qcfg, dtype=torch.float32)
ref = sdnq_amd.sdnq_post_load_quant(model, quantization_config=sdnq_amd.SDNQConfig(weights_dtype='int8', group_size=0, use_quantized_matmul=True,
                                    minimum_allowed_numel=4096, modules_to_not_convert=['lm_head']), torch_dtype=torch.float32)
a, b = dict(q.state_dict()), dict(ref.state_dict())
assert sorted(a) == sorted(b)
n = 0
for k in a:
    assert a[k].dtype == b[k].dtype and a[k].shape == b[k].shape and torch.equal(a[k].contiguous(), b[k].contiguous()), k
    n += a[k].dtype == torch.int8
assert n == 7 and type(q.model.layers[0].mlp.up_proj) is sdnq_amd.SDNQLinear and type(q.lm_head) is torch.nn.Linear
q.save_pretrained({str(tmp_path / 'q')!r})
import json
saved = json.load(open({str(tmp_path / 'q' / 'config.json')!r}))
assert saved['quantization_config']['quant_method'] == 'sdnq' and saved['quantization_config']['weights_dtype'] == 'int8'
back = transformers.AutoModelForCausalLM.from_pretrained({str(tmp_path / 'q')!r}, dtype=torch.float32)
c = dict(back.state_dict())
assert sorted(c) == sorted(a) and all(torch.equal(c[k].contiguous(), a[k].contiguous()) for k in a)
print('ok')
""")
    assert out.strip().splitlines()[-1] == "ok"


@pytest.mark.gpu
def test_loaded_model_reproduces_the_reference_logits_on_the_gpu():
    """The reference-written checkpoint, loaded by this build's plugin onto the GPU (every SDNQ layer on the HIP kernels), gives the
    logits the REFERENCE computed from it on the CPU (io.npz).  fp32 activations; the int8 matmuls are exact, what differs is the
    summation order of the float operators around them (attention, norms): relative L2 <= 2e-4."""
    out = run_py(f"""
import numpy as np, torch, transformers, sdnq, sdnq_amd
io = np.load({CKPT + '/io.npz'!r})
m = transformers.AutoModelForCausalLM.from_pretrained({CKPT!r}, dtype=torch.float32, device_map='cuda:0')
n_hip = sum(1 for mod in m.modules() if hasattr(mod, 'sdnq_dequantizer') and getattr(mod.forward_func, '__module__', '').startswith('sdnq_amd'))
assert n_hip == 14, n_hip
with torch.no_grad():
    y = m(input_ids=torch.from_numpy(io['input_ids']).cuda()).logits.float().cpu().numpy()
ref = io['logits']
rel = float(np.linalg.norm(y - ref) / np.linalg.norm(ref))
assert rel <= 2e-4, rel
print('ok', rel)
""")
    assert out.strip().splitlines()[-1].startswith("ok")
