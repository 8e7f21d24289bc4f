#!/usr/bin/env python3
"""Write a small pre-quantized SDNQ checkpoint with the REAL reference (Disty0/sdnq @ /root/reference) -> tests/golden/checkpoint_tiny/.

Runs ONLY in the build container.  A four-layer model is quantized by the reference's `sdnq_post_load_quant`, saved by the reference's
`save_sdnq_model` (its `quantization_config.json`; the tensors through the model's own `save_pretrained`, which -- like diffusers'
ModelMixin -- writes the state_dict to one safetensors file and the constructor arguments to `config.json`), RE-LOADED by the
reference's `load_sdnq_model`, and the re-loaded model's layer inputs / outputs on one batch are stored next to it (`io.npz`).
The fixture is DATA: tensors + two json files; `tests/test_gpu_parity.py::test_load_sdnq_model_*` defines the same four-layer
skeleton and loads it with `sdnq_amd.load_sdnq_model` -- no reference on the GPU box.

    python tests/golden/make_golden_checkpoint.py            # (re)write the fixture
    python tests/golden/make_golden_checkpoint.py <dir>      # write it somewhere else (the regeneration check of tests/test_oracle_golden.py)
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as G  # noqa: E402  (environment switches of the reference + the fake `diffusers` it needs to import)
import numpy as np  # noqa: E402
import torch  # noqa: E402

OUT = sys.argv[1] if (len(sys.argv) > 1 and not sys.argv[1].startswith("--")) else os.path.join(HERE, "checkpoint_tiny")


class TinyNet(torch.nn.Module):
    """proj_in 64 -> 128, mid 128 -> 96, proj_out 96 -> 64 (each followed by a LayerNorm-free SiLU), head 64 -> 10."""

    def __init__(self, d_in=64, d_hidden=128, d_mid=96, d_out=64, n_cls=10):
        super().__init__()
        self.cfg = dict(d_in=d_in, d_hidden=d_hidden, d_mid=d_mid, d_out=d_out, n_cls=n_cls)
        self.proj_in = torch.nn.Linear(d_in, d_hidden)
        self.mid = torch.nn.Linear(d_hidden, d_mid, bias=False)
        self.proj_out = torch.nn.Linear(d_mid, d_out)
        self.norm = torch.nn.LayerNorm(d_out)
        self.head = torch.nn.Linear(d_out, n_cls)

    def forward(self, x):
        h = torch.nn.functional.silu(self.proj_in(x))
        h = torch.nn.functional.silu(self.mid(h))
        h = self.norm(self.proj_out(h))
        return self.head(h)

    def save_pretrained(self, path, max_shard_size=None):  # what ModelMixin.save_pretrained leaves on disk
        from safetensors.torch import save_file
        os.makedirs(path, exist_ok=True)
        save_file({k: v.contiguous() for k, v in self.state_dict().items()}, os.path.join(path, "model.safetensors"))
        json.dump(dict(self.cfg), open(os.path.join(path, "config.json"), "w"), indent=1)  # (the constructor's arguments)


def main():
    # (make_golden.py installed the stand-in `diffusers` and imported the reference when it was imported above)
    # the two serialisation helpers diffusers' QuantizationConfigMixin gives every config (the stand-in of make_golden.py has none):
    # json.dumps(self.to_dict(), indent=2, sort_keys=True) + "\n" -- the file format of quantization_config.json
    import sdnq
    mixin = sdnq.SDNQConfig.__mro__[1]
    mixin.to_json_string = lambda self: json.dumps(self.to_dict(), indent=2, sort_keys=True) + "\n"
    mixin.to_json_file = lambda self, path: open(path, "w", encoding="utf-8").write(self.to_json_string())
    from sdnq import SDNQConfig, sdnq_post_load_quant
    from sdnq.loader import load_sdnq_model, save_sdnq_model
    torch.manual_seed(20250929)
    model = TinyNet().to(torch.bfloat16)
    with torch.no_grad():
        for p in model.parameters():
            if p.dim() == 2:
                p.copy_(torch.randn_like(p, dtype=torch.float32) * 0.08)
    cfg = SDNQConfig(weights_dtype="int8", group_size=0, use_quantized_matmul=True, minimum_allowed_numel=4096, minimum_allowed_channel_size=32,
                     modules_dtype_dict={"uint4": ["mid"]}, modules_to_not_convert=["head"],
                     modules_quant_config={"proj_out": {"use_svd": True, "svd_rank": 16}}, add_skip_keys=False)
    model = sdnq_post_load_quant(model, torch_dtype=torch.bfloat16, quantization_config=cfg)
    for f in os.listdir(OUT) if os.path.isdir(OUT) else []:
        os.remove(os.path.join(OUT, f))
    save_sdnq_model(model, OUT)
    loaded = load_sdnq_model(OUT, model_cls=TinyNet, device="cpu")
    x = (torch.randn(48, 64) * 1.5).to(torch.bfloat16)
    bits = lambda t: t.detach().contiguous().view(torch.uint16).numpy().copy()  # noqa: E731  (bf16 bit patterns)
    io = {"x": bits(x)}
    hooks = []
    for name in ("proj_in", "mid", "proj_out", "head"):
        def hook(mod, args, out, name=name):
            io[f"{name}.in"], io[f"{name}.out"] = bits(args[0]), bits(out)
        hooks.append(getattr(loaded, name).register_forward_hook(hook))
    with torch.no_grad():
        io["y"] = bits(loaded(x))
    meta = {n: {"weights_dtype": m.sdnq_dequantizer.weights_dtype, "group_size": int(m.sdnq_dequantizer.group_size),
                "use_quantized_matmul": bool(m.sdnq_dequantizer.use_quantized_matmul), "re_quantize_for_matmul": bool(m.sdnq_dequantizer.re_quantize_for_matmul),
                "quantized_weight_shape": list(m.sdnq_dequantizer.quantized_weight_shape), "has_svd": m.svd_up is not None}
            for n, m in loaded.named_modules() if hasattr(m, "sdnq_dequantizer")}
    io["meta_json"] = np.frombuffer(json.dumps(meta, sort_keys=True).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(OUT, "io.npz"), **io)
    print("wrote", OUT, sorted(os.listdir(OUT)), {k: v for k, v in meta.items()})


def verify_foreign(path):
    """`--verify <dir>`: a checkpoint written by ANOTHER writer (sdnq_amd.save_sdnq_model) loads in the reference's load_sdnq_model and
    the loaded model reproduces the stored layer outputs of the fixture (io.npz of checkpoint_tiny) bit for bit."""
    import sdnq  # noqa: F401  (imported by make_golden above)
    from sdnq.loader import load_sdnq_model
    loaded = load_sdnq_model(path, model_cls=TinyNet, device="cpu")
    io = np.load(os.path.join(HERE, "checkpoint_tiny", "io.npz"))
    x = torch.from_numpy(io["x"].copy()).view(torch.bfloat16)
    with torch.no_grad():
        y = loaded(x)
    got = y.detach().contiguous().view(torch.uint16).numpy()
    assert np.array_equal(got, io["y"]), int((got != io["y"]).sum())
    print("reference loaded", path, "and reproduced the fixture's outputs")


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--verify":
        verify_foreign(sys.argv[2])
    else:
        main()
