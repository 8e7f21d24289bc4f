"""Re-pointing already-built SDNQ models at the MI355X kernels.

``accelerate(model)`` is the least-assumption drop-in (SURVEY 8b ii): it walks a model that the *reference*
package built or loaded, finds modules carrying ``sdnq_dequantizer`` (the reference's own idiom, loader.py:204,239)
and replaces ``forward_func`` exactly like the reference's ``apply_sdnq_options_to_module`` does (loader.py:301).
Nothing else about the module changes: parameters, names and state_dict layout stay the reference's.

``apply_sdnq_options_to_model`` mirrors the reference entry point of the same name (loader.py:315) for the
options that matter on this path: toggling ``use_quantized_matmul`` and choosing the matmul dtype.
"""
from __future__ import annotations

import os

import torch

from .common import dtype_dict
from .dequantizer import SDNQDequantizer
from .forward import get_forward_func
from .quantizer import check_quantized_matmul_is_allowed

_DQ_FIELDS = ("result_dtype", "result_shape", "original_shape", "original_stride", "quantized_weight_shape", "weights_dtype",
              "quantized_matmul_dtype", "hadamard_group_size", "group_size", "svd_rank", "svd_steps", "codebook_steps",
              "use_quantized_matmul", "re_quantize_for_matmul", "use_stochastic_rounding", "use_hadamard", "use_codebook",
              "layer_class_name")


def adopt_dequantizer(dq) -> SDNQDequantizer:
    """Any object with the reference dataclass' fields (dequantizer.py:282-303) -> this package's record."""
    if isinstance(dq, SDNQDequantizer):
        return dq
    return SDNQDequantizer(**{f: getattr(dq, f) for f in _DQ_FIELDS})


def is_hot_path_linear(module: torch.nn.Module) -> bool:
    """An SDNQ Linear the HIP forwards compute (support.unsupported_reason is the one predicate)."""
    from .support import unsupported_reason
    dq = getattr(module, "sdnq_dequantizer", None)
    return dq is not None and getattr(dq, "layer_class_name", None) in ("Linear", "SDNQLinear") and unsupported_reason(module) is None


def is_hot_path_conv(module: torch.nn.Module) -> bool:
    from .support import unsupported_reason
    dq = getattr(module, "sdnq_dequantizer", None)
    return (dq is not None and getattr(dq, "layer_class_name", None) in ("Conv1d", "Conv2d", "Conv3d", "SDNQConv1d", "SDNQConv2d", "SDNQConv3d")
            and unsupported_reason(module) is None)


def _unlink(module: torch.nn.Module):
    """Take `module` out of its ProjectionGroup; the group dissolves (its other members run alone again)."""
    g = module.__dict__.pop("_sdnq_group", None)
    if g is not None:
        g[0].dissolve()


def _clear_step_state(_module=None, _args=None):
    """forward-pre-hook of the root model: nothing derived from activations outlives a step (quantized copies of the previous
    step's inputs, outputs of linked projections nobody claimed)."""
    if torch.compiler.is_compiling():
        return  # traced by Dynamo: nothing to clear at trace time (the caches are identity- and version-keyed, never required for correctness)
    from . import linear
    linear.invalidate(None)


class AccelerateResult(int):
    """What `accelerate()` returns: an int (the number of re-pointed modules, as before) that also carries `.accelerated` and
    `.skipped` -- [(qualified module name, reason)] of the SDNQ layers that were left on the forward they came with."""

    def __new__(cls, accelerated: int, skipped):
        r = super().__new__(cls, accelerated)
        r.accelerated, r.skipped = int(accelerated), list(skipped)
        return r

    def __iter__(self):  # `n, skipped = accelerate(model)`
        return iter((self.accelerated, self.skipped))


def _end_step(_module=None, _args=None, _output=None):
    """forward-hook of the root model: the per-call mode's weight prefetch re-joins the caller's stream (linear._WeightPipeline)."""
    if torch.compiler.is_compiling():
        return
    from . import linear
    linear.join_weight_pipeline()


@torch.no_grad()
def accelerate(model: torch.nn.Module) -> AccelerateResult:
    """Route every quantized Linear (and Conv1d / Conv2d / Conv3d, any ``groups``) of ``model`` that the HIP forwards compute
    through them.  NEVER turns a working model into a failing one: support is decided per module here, before anything is
    re-pointed (`support.unsupported_reason`, the predicate the forwards themselves use); an SDNQ layer in a configuration this
    package does not build keeps the ``forward_func`` it came with -- on a reference-built model that is the reference's own
    working forward (its idiom for a missing kernel: fall back + log.warning, kernel_wrappers.py:80-88) -- and ONE
    ``warnings.warn`` lists them.  Returns the number of re-pointed modules (an int with ``.accelerated`` / ``.skipped``,
    unpackable as ``(accelerated, skipped)``)."""
    from .support import unsupported_reason
    count, skipped = 0, []
    for name, module in model.named_modules():
        if getattr(module, "sdnq_dequantizer", None) is None:
            continue
        try:
            why = unsupported_reason(module)
        except Exception as e:  # noqa: BLE001  a foreign record the predicate cannot read is a reason to leave the layer alone, not to fail
            why = f"{type(e).__name__} while reading the layer's record: {e}"
        if why is None:
            try:
                dq = adopt_dequantizer(module.sdnq_dequantizer)
                fwd = get_forward_func(dq.layer_class_name, dq.quantized_matmul_dtype, dq.use_quantized_matmul)
            except (NotImplementedError, KeyError, AttributeError, TypeError) as e:  # a foreign record this package cannot read
                why = f"{type(e).__name__}: {e}"
        if why is not None:
            skipped.append((name or type(module).__name__, why))
            continue
        module.sdnq_dequantizer = dq
        module.forward_func = fwd
        module.__dict__.pop("_sdnq_hip_state", None)
        _unlink(module)
        from . import torch_ops
        torch_ops.layer_handle(module)  # torch.compile: the layer (Linear or conv) traces as sdnq_hip:: operators, no graph break
        count += 1
    if skipped:
        import warnings
        warnings.warn(f"sdnq_amd.accelerate: {len(skipped)} SDNQ layer(s) are not computed by the MI355X kernels and keep the forward they "
                      "came with: " + "; ".join(f"{n} ({w})" for n, w in skipped[:8]) + (" ..." if len(skipped) > 8 else ""), stacklevel=2)
    from . import linear
    if linear.LINK_PROJECTIONS:
        link_projections(model)
        if count and os.environ.get("SDNQ_HIP_COMPILE_GROUPING", "1").lower() not in {"0", "false", "no"}:
            # the compiled-graph form of the linked projections: an Inductor post-grad pass that merges layers on one quantized
            # activation into grouped launches (torch.compile of the SDXL step: 11.8 -> 8.9 ms); inert without torch.compile
            try:
                from . import torch_ops
                torch_ops.enable_compile_grouping()
            except Exception:  # noqa: BLE001  (a torch build without Inductor)
                pass
    if count and not getattr(model, "_sdnq_hip_step_hook", None):
        model._sdnq_hip_step_hook = model.register_forward_pre_hook(_clear_step_state)
        model._sdnq_hip_step_end_hook = model.register_forward_hook(_end_step)
    return AccelerateResult(count, skipped)


@torch.no_grad()
def link_layers(mods) -> bool:
    """Make `mods` (layers that consume the same tensor) a ``linear.ProjectionGroup`` if their configuration allows it: any number
    of row-wise direct-matmul layers of one input size whose widths share a divisor that is a multiple of 64 (one grouped launch,
    no weight copy), or up to four equally shaped layers in the dequantize + F.linear mode (one float GEMM)."""
    from .linear import ProjectionGroup
    mods = list(mods)
    if len(mods) < 2:
        return False
    float_mode = _float_linkable(mods)
    if not float_mode and not _fusable(mods):
        return False
    d0 = mods[0].sdnq_dequantizer
    if float_mode:
        if len(mods) > 4 or any(m.sdnq_dequantizer.out_features != d0.out_features for m in mods) or d0.out_features % 8:
            return False
    else:
        import math
        unit = 0
        for m in mods:
            unit = math.gcd(unit, m.sdnq_dequantizer.out_features)
        if unit % 64:
            return False
    for m in mods:
        _unlink(m)
    group = ProjectionGroup(mods, float_mode=float_mode)
    from .linear import drop_plans
    drop_plans(mods)  # (fast-path plans made while the layers ran alone)
    for i, m in enumerate(mods):
        m.__dict__["_sdnq_group"] = (group, i)
    return True


def _is_cross_attention(module, q, k) -> bool:
    """diffusers' Attention says so itself (is_cross_attention / cross_attention_dim); otherwise a key projection that reads a
    different width than the query projection can only be fed another tensor."""
    flag = getattr(module, "is_cross_attention", None)
    if flag is not None:
        return bool(flag)
    if getattr(module, "cross_attention_dim", None) is not None:
        return True
    return q is None or getattr(q, "in_features", None) != getattr(k, "in_features", None)


@torch.no_grad()
def link_projections(model: torch.nn.Module) -> int:
    """Link the attention projections of ``model`` that consume one tensor -- transparent to the host model: the modules, their
    names, parameters and outputs stay what they were.

    * self-attention blocks: ``to_q / to_k / to_v`` (and ``add_q_proj / add_k_proj / add_v_proj`` of the joint SD3 / FLUX blocks)
      become one ProjectionGroup: three GEMMs of a shared input run as one launch;
    * cross-attention blocks: ``to_q`` reads the image tokens and stays alone; ``to_k / to_v`` of EVERY cross-attention block with
      the same input width and configuration go into ONE model-wide group -- a UNet hands the same ``encoder_hidden_states``
      tensor to all of them (140 projections in SDXL), so the first one called in a step computes all of them in one launch.
    Whether the members really receive one tensor is checked at run time (``ProjectionGroup``): a group whose guess is wrong
    dissolves itself into the smaller groups it was made of (the model-wide group into per-block ``to_k / to_v`` pairs, a
    ``to_q / to_k / to_v`` triple into the ``to_k / to_v`` pair), and those, if wrong too, into single layers.  Returns the number
    of groups."""
    count = 0
    cross = {}  # (in_features, configuration) -> [to_k, to_v, to_k, to_v, ...] in module order
    for module in model.modules():
        for names in (("to_q", "to_k", "to_v"), ("add_q_proj", "add_k_proj", "add_v_proj")):
            q, k, v = (getattr(module, a, None) for a in names)
            if k is None or v is None or k is v or not is_hot_path_linear(k) or not is_hot_path_linear(v):
                continue
            if names[0] == "to_q" and _is_cross_attention(module, q, k):
                dk = k.sdnq_dequantizer
                key = (dk.in_features, dk.weights_dtype, dk.quantized_matmul_dtype, dk.use_quantized_matmul, dk.result_dtype, k.bias is None)
                cross.setdefault(key, []).extend([k, v])
                continue
            if q is not None and link_layers([q, k, v]):
                count += 1
                q.__dict__["_sdnq_group"][0].fallback = [[k, v]]  # a block whose query reads another tensor after all
            elif link_layers([k, v]):
                count += 1
    for mods in cross.values():
        if link_layers(mods):
            count += 1
            if len(mods) > 2:  # a wrong guess (per-block encoder states, skipped blocks) falls back to the per-block pairs
                mods[0].__dict__["_sdnq_group"][0].fallback = [mods[i:i + 2] for i in range(0, len(mods), 2)]
        else:  # e.g. the dequantize + F.linear mode (at most four equal layers per group): per-block pairs
            for i in range(0, len(mods), 2):
                count += bool(link_layers(mods[i:i + 2]))
    return count


def _float_linkable(mods) -> bool:
    """Layers in the dequantize + F.linear mode (use_quantized_matmul=False) of equal in_features and result dtype: any weight
    format works, the members are dequantized side by side and share one float GEMM."""
    dqs = [getattr(m, "sdnq_dequantizer", None) for m in mods]
    if any(d is None for d in dqs) or not all(is_hot_path_linear(m) for m in mods):
        return False
    d0 = dqs[0]
    if any(d.use_quantized_matmul or d.in_features != d0.in_features or d.result_dtype != d0.result_dtype for d in dqs):
        return False
    return len({m.bias is None for m in mods}) == 1


def _fusable(mods) -> bool:
    """Row-wise, directly-multipliable layers (one scale per output channel, weights stored in the matmul layout) of equal
    in_features and configuration: their concatenation along the output channels is itself such a layer, bit for bit."""
    dqs = [getattr(m, "sdnq_dequantizer", None) for m in mods]
    if any(d is None for d in dqs) or not all(is_hot_path_linear(m) for m in mods):
        return False
    d0 = dqs[0]
    same = ("weights_dtype", "quantized_matmul_dtype", "group_size", "use_quantized_matmul", "re_quantize_for_matmul", "use_hadamard",
            "result_dtype", "in_features")
    if not all(all(getattr(d, f) == getattr(d0, f) for f in same) for d in dqs):
        return False
    if not (d0.use_quantized_matmul and not d0.re_quantize_for_matmul and not d0.is_packed and not d0.use_hadamard):
        return False
    if any(getattr(m, "svd_up", None) is not None for m in mods):
        return False
    if any(m.scale.dtype != torch.float32 for m in mods):  # dequantize_fp32=False layers run their own (compatibility) path
        return False
    return len({m.bias is None for m in mods}) == 1


@torch.no_grad()
def _concat_linears(mods):
    from .layers import SDNQLinear
    d0 = mods[0].sdnq_dequantizer
    k = d0.in_features
    n = sum(m.sdnq_dequantizer.out_features for m in mods)
    skeleton = torch.nn.Linear(8, 8, bias=False)
    skeleton.in_features, skeleton.out_features = k, n
    dq = SDNQDequantizer(**{f: getattr(d0, f) for f in _DQ_FIELDS})
    dq.original_shape = torch.Size((n, k))
    dq.original_stride = [k, 1]
    dq.quantized_weight_shape = torch.Size((k, n))
    skeleton.sdnq_dequantizer = dq
    fused = SDNQLinear(skeleton, get_forward_func("Linear", dq.quantized_matmul_dtype, dq.use_quantized_matmul))
    phys = torch.cat([m.weight.t() for m in mods], dim=0).contiguous()  # [N][K] bytes of the matmul layout
    fused.weight = torch.nn.Parameter(phys.t(), requires_grad=False)    # logical [K, N], strides (1, K)
    fused.scale = torch.nn.Parameter(torch.cat([m.scale.reshape(1, -1) for m in mods], dim=1), requires_grad=False)
    zps = [getattr(m, "zero_point", None) for m in mods]
    fused.zero_point = None if zps[0] is None else torch.nn.Parameter(torch.cat([z.reshape(1, -1) for z in zps], dim=1), requires_grad=False)
    fused.svd_up = fused.svd_down = None
    fused.bias = None if mods[0].bias is None else torch.nn.Parameter(torch.cat([m.bias for m in mods], dim=0), requires_grad=False)
    return fused


@torch.no_grad()
def fuse_projections(model: torch.nn.Module) -> int:
    """diffusers' ``Attention.fuse_projections()`` for SDNQ layers: for every attention block whose ``to_q / to_k / to_v``
    (self-attention) or ``to_k / to_v`` (cross-attention) are row-wise direct-matmul SDNQLinear layers of one input size, add
    ``to_qkv`` / ``to_kv`` holding the concatenated quantized weights and set ``fused_projections = True`` (the attribute the
    fused attention processors read).  One activation quantization + one GEMM launch instead of three (two); outputs are
    bit-identical to the separate layers because every output channel keeps its own scale.  Returns the number of fused blocks."""
    count = 0
    for module in model.modules():
        q, k, v = (getattr(module, a, None) for a in ("to_q", "to_k", "to_v"))
        if k is None or v is None:
            continue
        if q is not None and _fusable([q, k, v]):
            module.to_qkv = _concat_linears([q, k, v])
            module.fused_projections = True
            count += 1
        elif _fusable([k, v]):
            module.to_kv = _concat_linears([k, v])
            module.fused_projections = True
            count += 1
    return count


def _map_key(key: str, key_mapping) -> str:
    """`_checkpoint_conversion_mapping` of transformers models: the first regular expression that matches renames the key (file_loader.py:6-13)."""
    import re
    if key_mapping:
        for pattern, replacement in key_mapping.items():
            key2, n = re.subn(pattern, replacement, key)
            if n > 0:
                return key2
    return key


@torch.no_grad()
def post_process_model(model: torch.nn.Module) -> torch.nn.Module:
    """What the reference does to a freshly loaded SDNQ model (loader.py:199-217): nothing requires a gradient, and the operands a
    matmul reads directly are re-laid out ONCE -- a checkpoint stores the direct-matmul weight as a contiguous logical [K, N]; the
    kernels (like the reference on gfx950, quant_utils.py:240-249) want the same logical tensor over physical [N][K] bytes, K
    contiguous; the SVD factors of such layers likewise.  The parameter itself is replaced, so no second copy stays resident."""
    P = lambda t: torch.nn.Parameter(t, requires_grad=False)  # noqa: E731
    for module in model.modules():
        dq = getattr(module, "sdnq_dequantizer", None)
        if dq is None:
            continue
        for name in ("weight", "scale", "zero_point", "svd_up", "svd_down", "bias"):
            t = getattr(module, name, None)
            if isinstance(t, torch.nn.Parameter):
                t.requires_grad_(False)
        if dq.use_quantized_matmul and not dq.re_quantize_for_matmul and module.weight.dim() == 2 and module.weight.is_contiguous() \
                and not getattr(dq, "is_packed", False):
            module.weight = P(module.weight.t().contiguous().t())  # logical [K, N], strides (1, K)
        if getattr(module, "svd_up", None) is not None and dq.use_quantized_matmul:
            # stored [R, N] / [K, R] (quantizer.py:164-167); the kernels read [N][R] / [R][K] rows
            if module.svd_up.is_contiguous():
                module.svd_up = P(module.svd_up.t().contiguous().t())
            if module.svd_down.is_contiguous():
                module.svd_down = P(module.svd_down.t().contiguous().t())
        module.__dict__.pop("_sdnq_hip_state", None)
    return model


@torch.no_grad()
def load_sdnq_model(model_path: str, model_cls=None, file_name: str | None = None, dtype: torch.dtype | None = None,
                    device: torch.device | str = "cuda", dequantize_fp32: bool | None = None, use_quantized_matmul: bool | None = None,
                    model_config: dict | None = None, quantization_config=None, model: torch.nn.Module | None = None) -> torch.nn.Module:
    """Load a pre-quantized SDNQ checkpoint -- the reference's on-disk format: `quantization_config.json` (or the `quantization_config`
    entry of `config.json`) + `*.safetensors` holding every layer's stored tensors (`weight` codes, `scale`, `zero_point`, `svd_up`,
    `svd_down`, `bias`) -- WITHOUT the reference package (reference loader.py:82-196, same arguments).

    The skeleton comes from `model` (an instance, usually built under `torch.device("meta")`), else from `model_cls` the way the
    reference builds it (`load_config` + `from_config` of diffusers models, `AutoConfig` of transformers models, else
    `model_cls(**model_config)`), on the meta device.  Its Linear / conv layers become SDNQ layers with placeholders of the stored
    shapes (`sdnq_post_load_quant(pre_quantized=True)`: every layer's record is a function of the config and the layer's shape),
    the tensors are read straight to `device` and assigned, direct-matmul operands are re-laid out once (`post_process_model`),
    the options are applied (`apply_sdnq_options_to_model`) and the layers are routed through the MI355X kernels (`accelerate`)."""
    import json
    from .quantizer import QuantizationMethod, SDNQConfig, sdnq_post_load_quant
    device = torch.device(device)
    config_path, qconfig_path = os.path.join(model_path, "config.json"), os.path.join(model_path, "quantization_config.json")
    if model_config is None:
        model_config = json.load(open(config_path, encoding="utf-8")) if os.path.exists(config_path) else {}
    if quantization_config is None:
        if os.path.exists(qconfig_path):
            quantization_config = json.load(open(qconfig_path, encoding="utf-8"))
        else:
            quantization_config = model_config.get("quantization_config", None)
            if quantization_config is None:
                raise ValueError(f"Cannot determine quantization_config for {model_path}, please provide quantization_config argument")
    if not isinstance(quantization_config, SDNQConfig):
        drop = ("quantization_device", "return_device", "non_blocking", "add_skip_keys", "use_dynamic_quantization", "use_stochastic_rounding",
                "is_training")  # (what the reference strips before it rebuilds the layers: utils.py:101-122)
        quantization_config = SDNQConfig.from_dict({k: v for k, v in dict(quantization_config).items() if k not in drop})
    quantization_config.add_skip_keys = False
    if model is None:
        if model_cls is None:
            class_name = model_config.get("_class_name", None) or model_config.get("architectures", None)
            if isinstance(class_name, list):
                class_name = class_name[0]
            for pkg in ("diffusers", "transformers"):
                if class_name is None or model_cls is not None:
                    break
                try:
                    model_cls = getattr(__import__(pkg), class_name, None)
                except ImportError:
                    continue
        if model_cls is None:
            raise ValueError(f"Cannot determine model class for {model_path}, please provide model_cls (or a model skeleton)")
        with torch.device("meta"):
            if hasattr(model_cls, "load_config") and hasattr(model_cls, "from_config"):
                config = model_cls.load_config(model_path)
                if hasattr(config, "pop"):
                    config.pop("quantization_config", None)
                model = model_cls.from_config(config)
            elif hasattr(model_cls, "_from_config"):
                import transformers
                config = transformers.AutoConfig.from_pretrained(model_path)
                if hasattr(config, "quantization_config"):
                    del config.quantization_config
                model = model_cls(config)
            else:
                cfg_kwargs = {k: v for k, v in model_config.items() if k != "quantization_config" and not k.startswith("_")}
                model = model_cls(**cfg_kwargs)
    model.eval()
    model = sdnq_post_load_quant(model, torch_dtype=dtype, pre_quantized=True, quantization_config=quantization_config)

    key_mapping = getattr(model, "_checkpoint_conversion_mapping", None)
    if file_name:
        files = [os.path.join(model_path, file_name)]
    else:
        files = sorted(os.path.join(model_path, f) for f in os.listdir(model_path) if f.endswith(".safetensors"))
    if not files:
        raise FileNotFoundError(f"no .safetensors file in {model_path}")
    from safetensors.torch import safe_open
    state_dict = {}
    for fn in files:
        with safe_open(fn, framework="pt", device=str(device)) as f:
            for key in f.keys():  # noqa: SIM118
                state_dict[_map_key(key, key_mapping)] = f.get_tensor(key)
    tied = getattr(model, "_tied_weights_keys", None)
    if isinstance(tied, dict):
        for key, value in tied.items():
            if value in state_dict and key not in state_dict:
                state_dict[key] = state_dict[value]
    missing, unexpected = model.load_state_dict(state_dict, strict=False, assign=True)
    del state_dict
    still_meta = [n for n, p in list(model.named_parameters()) + list(model.named_buffers()) if p.is_meta]
    if still_meta:
        raise RuntimeError(f"{model_path}: {len(still_meta)} tensor(s) of the model are not in the checkpoint ({', '.join(still_meta[:6])}"
                           f"{' ...' if len(still_meta) > 6 else ''}); unexpected keys: {list(unexpected)[:6]}")
    if unexpected:
        # the reference loads strictly (loader.py:215-227: load_state_dict of the rebuilt skeleton): a tensor the skeleton has no slot for -- a
        # `codebook` of a configuration this build does not construct, a layer the config did not mention -- must not be dropped silently
        raise RuntimeError(f"{model_path}: {len(unexpected)} tensor(s) of the checkpoint have no place in the rebuilt model "
                           f"({', '.join(list(unexpected)[:6])}{' ...' if len(unexpected) > 6 else ''})")
    model.quantization_config = quantization_config
    model.quantization_method = QuantizationMethod.SDNQ
    if hasattr(model, "config"):
        try:
            model.config.quantization_config = quantization_config
        except Exception:  # noqa: BLE001
            pass
    model = post_process_model(model)
    if dtype is not None:  # the float leaves follow the requested dtype too (loader.py:228-232)
        for module in model.modules():
            if getattr(module, "sdnq_dequantizer", None) is None and not list(module.children()):
                for name, p in list(module.named_parameters(recurse=False)):
                    if p.dtype in (torch.float16, torch.bfloat16) and p.dtype != dtype:
                        setattr(module, name, torch.nn.Parameter(p.to(dtype), requires_grad=False))
    if (dtype is not None) or (dequantize_fp32 is not None) or (use_quantized_matmul is not None):
        model = apply_sdnq_options_to_model(model, dtype=dtype, dequantize_fp32=dequantize_fp32, use_quantized_matmul=use_quantized_matmul)
    accelerate(model)
    return model


@torch.no_grad()
def save_sdnq_model(model: torch.nn.Module, model_path: str, max_shard_size: str = "5GB", is_pipeline: bool = False, sdnq_config=None) -> None:
    """Write `model` in the reference's on-disk format (reference loader.py:46-79, same arguments): the tensors as safetensors --
    through the model's own `save_pretrained` when it has one (diffusers / transformers models), else `model.safetensors` + the model's
    `config` (a mapping) as `config.json` -- and `quantization_config.json` (the given `sdnq_config`, else the model's own; for a pipeline
    one per quantized sub-module).  What is stored is the LOGICAL tensor of every layer as a contiguous array -- the direct-matmul weight
    as [K, N], SVD factors as [R, N] / [K, R]: the layout `save_pretrained` leaves behind in the reference and `post_process_model`
    (here and there) re-lays out at load time -- so the checkpoint loads in the reference and here alike.  The model is not modified."""
    import json
    from .quantizer import SDNQConfig
    os.makedirs(model_path, exist_ok=True)

    def write_config(cfg, path):  # diffusers' QuantizationConfigMixin.to_json_file: json.dumps(to_dict(), indent=2, sort_keys=True) + "\n"
        with open(path, "w", encoding="utf-8") as f:
            f.write(json.dumps(cfg.to_dict(), indent=2, sort_keys=True) + "\n")

    def config_of(module):
        q = getattr(module, "quantization_config", None)
        if isinstance(q, SDNQConfig):
            return q
        q = getattr(getattr(module, "config", None), "quantization_config", None)
        return q if isinstance(q, SDNQConfig) else None

    if hasattr(model, "save_pretrained"):
        # safetensors refuses strided tensors: the layers' parameters are presented contiguous for the duration of the save
        swapped = []
        try:
            for module in model.modules() if isinstance(model, torch.nn.Module) else []:
                if getattr(module, "sdnq_dequantizer", None) is None:
                    continue
                for name in ("weight", "svd_up", "svd_down"):
                    t = getattr(module, name, None)
                    if isinstance(t, torch.nn.Parameter) and not t.is_contiguous():
                        swapped.append((module, name, t))
                        setattr(module, name, torch.nn.Parameter(t.contiguous(), requires_grad=False))
            model.save_pretrained(model_path, max_shard_size=max_shard_size)
        finally:
            for module, name, t in swapped:
                setattr(module, name, t)
    else:
        from safetensors.torch import save_file
        save_file({k: v.detach().contiguous() for k, v in model.state_dict().items()}, os.path.join(model_path, "model.safetensors"))
        cfg = getattr(model, "config", None)
        if isinstance(cfg, dict) or hasattr(cfg, "items"):
            with open(os.path.join(model_path, "config.json"), "w", encoding="utf-8") as f:
                json.dump({k: v for k, v in dict(cfg).items() if k != "quantization_config"}, f, indent=1)
    qpath = os.path.join(model_path, "quantization_config.json")
    if sdnq_config is not None:
        write_config(sdnq_config, qpath)
    if is_pipeline:
        names = [n for n in getattr(model, "_internal_dict", {}).keys() if not n.startswith("_")] if hasattr(model, "_internal_dict") else \
            [n for n, _ in model.named_children()]
        for n in sorted(set(names)):
            sub = getattr(model, n, None)
            if isinstance(sub, torch.nn.Module) and config_of(sub) is not None and os.path.isdir(os.path.join(model_path, n)):
                write_config(config_of(sub), os.path.join(model_path, n, "quantization_config.json"))
    elif sdnq_config is None and config_of(model) is not None:
        write_config(config_of(model), qpath)


@torch.no_grad()
def apply_sdnq_options_to_model(model: torch.nn.Module, dtype: torch.dtype | None = None, dequantize_fp32: bool | None = None,
                                use_quantized_matmul: bool | None = None, quantized_matmul_dtype: str | None = None):
    skipped_foreign = []
    for module in model.modules():
        # every SDNQ Linear / conv layer, whether or not the HIP forwards compute its (old or new) configuration: the options only
        # re-type tensors and re-point forward_func; an unbuilt configuration then fails loudly at its forward (support.require)
        cls = getattr(getattr(module, "sdnq_dequantizer", None), "layer_class_name", None)
        if cls is None or getattr(module.sdnq_dequantizer, "use_codebook", False):
            continue
        conv = cls in ("Conv1d", "Conv2d", "Conv3d", "SDNQConv1d", "SDNQConv2d", "SDNQConv3d")
        if not (conv or cls in ("Linear", "SDNQLinear")):
            continue
        fwd_now = getattr(module, "forward_func", None)
        if fwd_now is not None and not str(getattr(fwd_now, "__module__", "")).startswith("sdnq_amd"):
            # a layer that runs on a FOREIGN forward (accelerate() left it on the reference's, because this package does not build
            # its configuration): re-pointing it here would turn a working layer into one that raises at its forward.  It stays
            # what it is -- options for such layers are the business of the package whose forward they run on.
            from .support import unsupported_reason
            try:
                why = unsupported_reason(module)
            except Exception as e:  # noqa: BLE001
                why = f"{type(e).__name__}: {e}"
            if why is not None:
                skipped_foreign.append(why)
                continue
        dq = adopt_dequantizer(module.sdnq_dequantizer)
        module.sdnq_dequantizer = dq
        if dtype is not None:
            dq.result_dtype = dtype
            for name in ("svd_up", "svd_down", "bias"):
                t = getattr(module, name, None)
                if t is not None and t.dtype != dtype:
                    setattr(module, name, torch.nn.Parameter(t.to(dtype), requires_grad=False))
        # scale / zero_point dtype (reference loader.py:262-283): float32 when asked for (or for > 8-bit formats), untouched when
        # already float32 and nothing was asked, else the layer's result dtype.  (The reference's float-matmul clause needs
        # row-wise fp8 scaling hardware; gfx950 runs tensorwise, kernel_wrappers.use_tensorwise_fp8_matmul.)
        sdt = module.scale.dtype
        wide = sdt in (torch.float32, torch.float64)
        if dequantize_fp32 or dtype_dict[dq.weights_dtype]["num_bits"] > 8:
            want_sdt = sdt if wide else (torch.float64 if dq.result_dtype == torch.float64 else torch.float32)
        elif dequantize_fp32 is None and wide:
            want_sdt = sdt
        else:
            want_sdt = dq.result_dtype
        if want_sdt != sdt:
            module.scale = torch.nn.Parameter(module.scale.to(want_sdt), requires_grad=False)
            if getattr(module, "zero_point", None) is not None:
                module.zero_point = torch.nn.Parameter(module.zero_point.to(want_sdt), requires_grad=False)
        if conv:
            # conv layers: result dtype and scale dtype only -- the reference leaves their matmul switch alone here
            # (`current_use_quantized_matmul = None` for everything that is not a Linear, loader.py:244-255)
            module.forward_func = get_forward_func(dq.layer_class_name, dq.quantized_matmul_dtype, dq.use_quantized_matmul)
            module.__dict__.pop("_sdnq_hip_state", None)
            _refresh_compile_plan(module)
            continue
        if use_quantized_matmul is not None and use_quantized_matmul != dq.use_quantized_matmul:
            n, k = dq.out_features, dq.in_features
            want = check_quantized_matmul_is_allowed(use_quantized_matmul, n, k)
            if want != dq.use_quantized_matmul:
                _relayout(module, dq, want)
        if quantized_matmul_dtype is not None:
            dq.quantized_matmul_dtype = quantized_matmul_dtype
        module.forward_func = get_forward_func("Linear", dq.quantized_matmul_dtype, dq.use_quantized_matmul)
        module.__dict__.pop("_sdnq_hip_state", None)
        _unlink(module)  # the layer's layout / forward may have changed: its group (if any) dissolves, siblings run alone
        _refresh_compile_plan(module)
    if skipped_foreign:
        import warnings
        warnings.warn(f"sdnq_amd.apply_sdnq_options_to_model: {len(skipped_foreign)} SDNQ layer(s) run on another package's forward in a "
                      f"configuration the MI355X kernels do not build and were left untouched ({skipped_foreign[0]})", stacklevel=2)
    return model


def _refresh_compile_plan(module) -> None:
    """The operator plan SDNQLayer.forward follows under torch.compile (`_sdnq_hip_plan`, torch_ops.layer_plan) is derived from the
    dequantizer's matmul switch / dtypes and the scale dtype: whatever changed those must re-derive it, or the compiled model keeps
    computing the OLD mode while the eager forward_func runs the new one (advisor, round 3)."""
    if "_sdnq_hip_handle" in module.__dict__:
        from . import torch_ops
        torch_ops.layer_handle(module)


def _relayout(module, dq: SDNQDequantizer, want_qmm: bool):
    """Switch a layer between the plain and the transposed (direct-matmul) layouts (reference loader.py:262-300)."""
    was_transposed = dq.weight_is_transposed
    dq.use_quantized_matmul = want_qmm
    now_transposed = dq.weight_is_transposed
    P = lambda t: torch.nn.Parameter(t, requires_grad=False)  # noqa: E731
    if was_transposed != now_transposed:
        module.weight = P(module.weight.t())  # same bytes, other logical view
        module.scale = P(module.scale.t().contiguous())
        if getattr(module, "zero_point", None) is not None:
            module.zero_point = P(module.zero_point.t().contiguous())
        dq.quantized_weight_shape = module.weight.shape
    if getattr(module, "svd_up", None) is not None:  # SVD factors follow use_quantized_matmul (quantizer.py:164-167)
        module.svd_up = P(module.svd_up.t())
        module.svd_down = P(module.svd_down.t())
