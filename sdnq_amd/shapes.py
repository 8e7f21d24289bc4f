"""Linear shape lists of the synthetic "model step" workloads (from the public model configs,
not from the reference, which ships no model code).

Each entry: (name, M tokens, K in_features, N out_features, has_bias, repeat).
``M`` is the number of activation rows the layer sees in ONE denoising step at batch size 1.
"""
from __future__ import annotations


def sdxl_unet_linears(latent: int = 128, text_tokens: int = 77):
    """SDXL-base UNet, bs=1, latent x latent (128 -> 1024^2 px). 11 Transformer2D modules:
    C=640 @ (latent/2)^2 tokens: 5 modules x 2 layers; C=1280 @ (latent/4)^2 tokens: 6 modules x 10 layers."""
    out = []
    for c, tokens, modules, layers in ((640, (latent // 2) ** 2, 5, 2), (1280, (latent // 4) ** 2, 6, 10)):
        nl = modules * layers
        out += [
            (f"c{c}.attn1.to_qkv", tokens, c, c, False, 3 * nl),
            (f"c{c}.attn1.to_out", tokens, c, c, True, nl),
            (f"c{c}.attn2.to_q", tokens, c, c, False, nl),
            (f"c{c}.attn2.to_kv", text_tokens, 2048, c, False, 2 * nl),
            (f"c{c}.attn2.to_out", tokens, c, c, True, nl),
            (f"c{c}.ff.proj_geglu", tokens, c, 8 * c, True, nl),
            (f"c{c}.ff.out", tokens, 4 * c, c, True, nl),
            (f"c{c}.proj_in_out", tokens, c, c, True, 2 * modules),
        ]
    # M = 1 layers (time / added-condition embeddings): take the M < 32 dequant + float GEMM branch
    out += [
        ("resnet.time_emb_proj.320", 1, 1280, 320, True, 5),
        ("resnet.time_emb_proj.640", 1, 1280, 640, True, 5),
        ("resnet.time_emb_proj.1280", 1, 1280, 1280, True, 7),
        ("add_embedding.linear_1", 1, 2816, 1280, True, 1),
        ("add_embedding.linear_2", 1, 1280, 1280, True, 1),
    ]
    return out


def sdxl_unet_layer_sequence(latent: int = 128, text_tokens: int = 77):
    """The same Linear layers as `sdxl_unet_linears`, in execution order, each with the identity of the activation tensor
    it consumes: (name, M, K, N, has_bias, input_key).  Layers with the same input_key read the SAME tensor object in
    diffusers' attention processors (to_q / to_k / to_v of self-attention share `hidden_states`; every cross-attention
    to_k / to_v reads the one `encoder_hidden_states`), which is what an activation-quantization cache can exploit."""
    seq = []
    lid = 0
    for c, tokens, modules, layers in ((640, (latent // 2) ** 2, 5, 2), (1280, (latent // 4) ** 2, 6, 10)):
        for mod in range(modules):
            seq.append((f"c{c}.proj_in", tokens, c, c, True, f"m{c}.{mod}.in"))
            for _ in range(layers):
                L = f"L{lid}"
                lid += 1
                seq += [
                    (f"c{c}.attn1.to_q", tokens, c, c, False, L + ".h1"), (f"c{c}.attn1.to_k", tokens, c, c, False, L + ".h1"),
                    (f"c{c}.attn1.to_v", tokens, c, c, False, L + ".h1"), (f"c{c}.attn1.to_out", tokens, c, c, True, L + ".a1"),
                    (f"c{c}.attn2.to_q", tokens, c, c, False, L + ".h2"), (f"c{c}.attn2.to_k", text_tokens, 2048, c, False, "text"),
                    (f"c{c}.attn2.to_v", text_tokens, 2048, c, False, "text"), (f"c{c}.attn2.to_out", tokens, c, c, True, L + ".a2"),
                    (f"c{c}.ff.proj_geglu", tokens, c, 8 * c, True, L + ".h3"), (f"c{c}.ff.out", tokens, 4 * c, c, True, L + ".g"),
                ]
            seq.append((f"c{c}.proj_out", tokens, c, c, True, f"m{c}.{mod}.out"))
    for i, (k, n, rep) in enumerate(((1280, 320, 5), (1280, 640, 5), (1280, 1280, 7))):
        seq += [(f"resnet.time_emb_proj.{n}", 1, k, n, True, "temb")] * rep
    seq += [("add_embedding.linear_1", 1, 2816, 1280, True, "add_in"), ("add_embedding.linear_2", 1, 1280, 1280, True, "add_mid")]
    return seq


def sdxl_unet_convs(latent: int = 128):
    """Conv2d layers of the SDXL-base UNet at bs=1 whose channel counts allow the quantized matmul (C_in, C_out >= 32; conv_in
    4->320 and conv_out 320->4 stay float): (name, C_in, H, W, C_out, kernel, stride, padding), in execution order.
    block_out_channels (320, 640, 1280), 2 resnets per down block, 3 per up block (public model config)."""
    out = []
    h0, h1, h2 = latent, latent // 2, latent // 4

    def resnet(tag, cin, cout, h):
        out.append((f"{tag}.conv1", cin, h, h, cout, 3, 1, 1))
        out.append((f"{tag}.conv2", cout, h, h, cout, 3, 1, 1))
        if cin != cout:
            out.append((f"{tag}.conv_shortcut", cin, h, h, cout, 1, 1, 0))

    resnet("down0.res0", 320, 320, h0); resnet("down0.res1", 320, 320, h0)
    out.append(("down0.downsample", 320, h0, h0, 320, 3, 2, 1))
    resnet("down1.res0", 320, 640, h1); resnet("down1.res1", 640, 640, h1)
    out.append(("down1.downsample", 640, h1, h1, 640, 3, 2, 1))
    resnet("down2.res0", 640, 1280, h2); resnet("down2.res1", 1280, 1280, h2)
    resnet("mid.res0", 1280, 1280, h2); resnet("mid.res1", 1280, 1280, h2)
    resnet("up0.res0", 2560, 1280, h2); resnet("up0.res1", 2560, 1280, h2); resnet("up0.res2", 1920, 1280, h2)
    out.append(("up0.upsample", 1280, h1, h1, 1280, 3, 1, 1))
    resnet("up1.res0", 1920, 640, h1); resnet("up1.res1", 1280, 640, h1); resnet("up1.res2", 960, 640, h1)
    out.append(("up1.upsample", 640, h0, h0, 640, 3, 1, 1))
    resnet("up2.res0", 960, 320, h0); resnet("up2.res1", 640, 320, h0); resnet("up2.res2", 640, 320, h0)
    return out


def conv_gemm_dims(e):
    """(M, K, N) of the im2col GEMM of one sdxl_unet_convs entry."""
    _, cin, h, w, cout, k, s, p = e
    ho, wo = (h + 2 * p - k) // s + 1, (w + 2 * p - k) // s + 1
    return ho * wo, cin * k * k, cout


def flux_dev_linears(img_tokens: int = 4096, txt_tokens: int = 512, d: int = 3072):
    """FLUX.1-dev transformer, 1024^2 px: 19 double blocks + 38 single blocks."""
    t_all = img_tokens + txt_tokens
    out = []
    for stream, tokens in (("img", img_tokens), ("txt", txt_tokens)):
        out += [
            (f"double.{stream}.qkv", tokens, d, d, True, 3 * 19),
            (f"double.{stream}.out", tokens, d, d, True, 19),
            (f"double.{stream}.ff.proj", tokens, d, 4 * d, True, 19),
            (f"double.{stream}.ff.out", tokens, 4 * d, d, True, 19),
            (f"double.{stream}.adaln", 1, d, 6 * d, True, 19),
        ]
    out += [
        ("single.qkv", t_all, d, d, True, 3 * 38),
        ("single.proj_mlp", t_all, d, 4 * d, True, 38),
        ("single.proj_out", t_all, 5 * d, d, True, 38),
        ("single.adaln", 1, d, 3 * d, True, 37),
    ]
    return out


def flux_dev_layer_sequence(img_tokens: int = 4096, txt_tokens: int = 512, d: int = 3072):
    """`flux_dev_linears` in execution order with activation identity (see `sdxl_unet_layer_sequence`): in the double
    blocks q/k/v of each stream read one normed tensor, in the single blocks q/k/v/proj_mlp read one normed tensor, and
    every adaLN modulation layer reads the one conditioning vector."""
    t_all = img_tokens + txt_tokens
    seq = []
    for i in range(19):
        for stream, tokens in (("img", img_tokens), ("txt", txt_tokens)):
            p = f"D{i}.{stream}"
            seq.append((f"double.{stream}.adaln", 1, d, 6 * d, True, "vec"))
            seq += [(f"double.{stream}.qkv", tokens, d, d, True, p + ".n1")] * 3
        for stream, tokens in (("img", img_tokens), ("txt", txt_tokens)):
            p = f"D{i}.{stream}"
            seq += [
                (f"double.{stream}.out", tokens, d, d, True, p + ".a"),
                (f"double.{stream}.ff.proj", tokens, d, 4 * d, True, p + ".n2"),
                (f"double.{stream}.ff.out", tokens, 4 * d, d, True, p + ".g"),
            ]
    for i in range(38):
        if i < 37:
            seq.append(("single.adaln", 1, d, 3 * d, True, "vec"))
        seq += [("single.qkv", t_all, d, d, True, f"S{i}.n")] * 3
        seq += [("single.proj_mlp", t_all, d, 4 * d, True, f"S{i}.n"), ("single.proj_out", t_all, 5 * d, d, True, f"S{i}.cat")]
    return seq


def ops_of_sequence(seq, min_m: int = 0) -> int:
    return sum(2 * m * k * n + (m * n if b else 0) for (_, m, k, n, b, _key) in seq if m >= min_m)


def ops_of(shapes, min_m: int = 0) -> int:
    """Algorithmic ops with the reference's formula 2*M*K*N + M*N*[bias] (scripts/benchmark_sdnq_inference_matmul.py:41-42)."""
    return sum(r * (2 * m * k * n + (m * n if b else 0)) for (_, m, k, n, b, r) in shapes if m >= min_m)


def bytes_of(shapes, weight_bits: float = 8.0, act_bytes: int = 2) -> dict:
    w = sum(r * (n * k * weight_bits / 8 + 4 * n + (act_bytes * n if b else 0)) for (_, m, k, n, b, r) in shapes)
    a = sum(r * (act_bytes * m * k + act_bytes * m * n) for (_, m, k, n, b, r) in shapes)
    return {"weights": w, "activations": a}


def sdxl_unet_attentions(latent: int = 128, text_tokens: int = 77, head_dim: int = 64):
    """Attention calls of one SDXL-base UNet step at bs=1: every transformer layer runs one self-attention over its image
    tokens and one cross-attention onto the text tokens.  Entries: (name, heads, q_len, kv_len, head_dim, repeat)."""
    out = []
    for c, tokens, modules, layers in ((640, (latent // 2) ** 2, 5, 2), (1280, (latent // 4) ** 2, 6, 10)):
        out += [(f"c{c}.attn1", c // head_dim, tokens, tokens, head_dim, modules * layers),
                (f"c{c}.attn2", c // head_dim, tokens, text_tokens, head_dim, modules * layers)]
    return out


def flux_dev_attentions(img_tokens: int = 4096, txt_tokens: int = 512, d: int = 3072, head_dim: int = 128):
    """FLUX.1-dev: 19 double-stream + 38 single-stream blocks, each one joint attention over text + image tokens."""
    return [("joint_attn", d // head_dim, img_tokens + txt_tokens, img_tokens + txt_tokens, head_dim, 19 + 38)]


def ops_of_attentions(calls, batch: int = 1) -> int:
    """2 * QN * KN * D for Q.K^T plus the same for P.V, per head."""
    return sum(4 * batch * h * qn * kn * d * rep for (_, h, qn, kn, d, rep) in calls)
