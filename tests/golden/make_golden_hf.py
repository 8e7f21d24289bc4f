#!/usr/bin/env python3
"""Write a tiny transformers checkpoint quantized and saved by the REAL reference's transformers plugin -> tests/golden/checkpoint_hf_tiny/.

Build container only.  A 2-layer LlamaForCausalLM (hidden 64, 4 heads, vocab 128; random weights from a fixed seed) is written as a float
checkpoint, loaded through `AutoModelForCausalLM.from_pretrained(..., quantization_config=sdnq.SDNQConfig(...))` -- the reference's
`SDNQQuantizer` quantizes every eligible Linear while the weights are read -- and saved with `save_pretrained`: `config.json` (with the
`quantization_config` entry the plugin writes) + `model.safetensors`.  The re-loaded model's logits on one batch are stored next to it
(`io.npz`).  The fixture is DATA; tests/test_hf_plugin.py loads it with THIS build's plugin (`import sdnq` -> sdnq_amd) and no reference.

    python tests/golden/make_golden_hf.py [out_dir]
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as G  # noqa: E402,F401  (environment switches of the reference, the stand-in `diffusers`, `import sdnq` = the reference)
import numpy as np  # noqa: E402
import torch  # noqa: E402

OUT = sys.argv[1] if len(sys.argv) > 1 else os.path.join(HERE, "checkpoint_hf_tiny")


def main():
    import tempfile
    import transformers
    import sdnq
    assert "reference" in sdnq.__file__, sdnq.__file__
    torch.manual_seed(20250930)
    cfg = transformers.LlamaConfig(hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=4,
                                   vocab_size=128, max_position_embeddings=64, tie_word_embeddings=False)
    model = transformers.LlamaForCausalLM(cfg).to(torch.float32)
    with torch.no_grad():
        for p in model.parameters():
            if p.dim() == 2:
                p.copy_(torch.randn_like(p) * 0.08)
    with tempfile.TemporaryDirectory() as tmp:
        model.save_pretrained(tmp)
        qcfg = sdnq.SDNQConfig(weights_dtype="int8", group_size=0, use_quantized_matmul=True, minimum_allowed_numel=4096, minimum_allowed_channel_size=32,
                               modules_dtype_dict={"uint4": ["down_proj"]}, add_skip_keys=True)
        qmodel = transformers.AutoModelForCausalLM.from_pretrained(tmp, quantization_config=qcfg, dtype=torch.float32)
    os.makedirs(OUT, exist_ok=True)
    for f in os.listdir(OUT):
        os.remove(os.path.join(OUT, f))
    qmodel.save_pretrained(OUT)
    loaded = transformers.AutoModelForCausalLM.from_pretrained(OUT, dtype=torch.float32)
    ids = torch.randint(0, 128, (2, 24), generator=torch.Generator().manual_seed(7))
    with torch.no_grad():
        logits = loaded(input_ids=ids).logits
        logits_q = qmodel(input_ids=ids).logits
    assert torch.equal(logits, logits_q), "the re-loaded model differs from the model that was saved"
    kinds = {n: type(m).__name__ for n, m in loaded.named_modules() if hasattr(m, "sdnq_dequantizer")}
    np.savez_compressed(os.path.join(OUT, "io.npz"), input_ids=ids.numpy(), logits=logits.float().numpy(),
                        sdnq_layers=np.array(sorted(kinds), dtype=object).astype(str))
    print("wrote", OUT, sorted(os.listdir(OUT)), len(kinds), "SDNQ layers")


if __name__ == "__main__":
    main()
