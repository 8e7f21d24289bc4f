#!/usr/bin/env python3
"""Golden vectors for the quantized attention forward, made by RUNNING the reference's own Triton kernel
(`sdnq_attn_kernel`, kernels/triton_atten.py:143-335) and its host code (`sdnq_triton_atten`, :540-618) on the CPU through
Triton's interpreter (TRITON_INTERPRET=1; there is no GPU in the build container).

What this harness has to supply because there is no GPU driver here (nothing of the reference is modified or stored):
  * the autotuner cannot benchmark without a device, so the kernel is launched with ONE fixed configuration
    (BLOCK_SIZE_M = BLOCK_SIZE_N = 32, recorded in the fixture's meta; with pv_matmul_dtype=None the result depends on the
    block size only through fp32 rounding order);
  * the interpreter does not convert Python floats to scalars, so `sm_scale` is wrapped as an fp32 scalar handle;
  * the interpreter has no working bfloat16 (numpy has none: Triton 3.6 accepts the tensors and returns garbage), so the f16_* cases are
    float16; token counts are multiples of 4 because the interpreter's tensor descriptors want 16-byte aligned bases for the per-token
    fp32 scale vectors;
  * the bf16_* cases (round 5; the dtype SDXL / FLUX run): the reference's host code `quantize_attn` -- plain torch -- runs on the REAL
    bfloat16 tensors (its dtype-dependent steps: the Hadamard rotation in the tensor dtype, the values themselves), so q_q / q_scale /
    k_q / k_scale of the fixture are the reference's bfloat16 path bit for bit; the KERNEL then runs on those quantized operands with V
    handed over as float32 (the same values: every bfloat16 is a float32), i.e. with P and the output NOT rounded to bfloat16 -- the
    fixture's `out` is float32 and a bfloat16 implementation is compared with it at bfloat16 tolerance (meta: "out_is").
  * the fp8 P.V cases (round 6): the interpreter's float32 -> float8 cast (`_convert_float`) TRUNCATES the mantissa (its "rtne" branch adds
    the cut-off bit without carrying into the exponent), where Triton's language semantics -- and every GPU lowering -- round `x.to(tl.float8e4nv)`
    to nearest even (`fp_downcast_rounding` defaults to "rtne").  For `p.to(v.dtype)` (triton_atten.py:319) the harness substitutes torch's
    float32 -> float8_e4m3fn conversion (round to nearest even) for that one cast of the interpreter; every other interpreter operation is untouched;
Fixtures are DATA only (inputs, the reference's quantized operands, outputs).  Run:  python tests/golden/make_golden_attention.py
"""
import json
import os
import sys
import types

os.environ["TRITON_INTERPRET"] = "1"
os.environ["SDNQ_USE_TORCH_COMPILE"] = "0"

import numpy as np  # noqa: E402
import torch  # noqa: E402
import triton.language as tl  # noqa: E402
from triton.runtime.interpreter import TensorHandle  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
pkg = types.ModuleType("sdnq")
pkg.__path__ = ["/root/reference/src/sdnq"]
sys.modules["sdnq"] = pkg
from sdnq.kernels import triton_atten as ta  # noqa: E402  (the reference)

BLOCK_M = BLOCK_N = 32


class FixedConfig:
    """Stands in for the autotuner: same kernel function, one fixed tile configuration."""

    def __init__(self, fn):
        self.fn = fn

    def __getitem__(self, grid):
        def launch(*args, **kw):
            g = grid({"BLOCK_SIZE_M": BLOCK_M, "BLOCK_SIZE_N": BLOCK_N}) if callable(grid) else grid
            args = [tl.tensor(TensorHandle(np.array([a], dtype=np.float32), tl.float32), tl.float32) if isinstance(a, float) else a
                    for a in args]
            return self.fn[g](*args, BLOCK_SIZE_M=BLOCK_M, BLOCK_SIZE_N=BLOCK_N, **kw)
        return launch


import triton.runtime.interpreter as _interp  # noqa: E402

_interp_convert_float = _interp._convert_float


def _convert_float_rne(input, input_dtype, output_dtype, rounding_mode):
    if input_dtype == tl.float32 and output_dtype == tl.float8e4nv:  # (see the module docstring)
        t = torch.from_numpy(np.ascontiguousarray(input).view(np.float32)).to(torch.float8_e4m3fn)
        return t.view(torch.uint8).numpy().reshape(input.shape)
    return _interp_convert_float(input, input_dtype, output_dtype, rounding_mode)


_interp._convert_float = _convert_float_rne

ta.wrap_triton = lambda k: k
ta.sdnq_attn_kernel = FixedConfig(ta.sdnq_attn_kernel.fn)

CASES = [
    dict(name="f16_d64_tail", z=1, qh=2, kh=2, qn=40, kn=52, d=64, kw={}),
    dict(name="f16_d64_causal", z=2, qh=2, kh=2, qn=64, kn=64, d=64, kw=dict(is_causal=True)),
    dict(name="f16_d64_causal_tail", z=1, qh=1, kh=1, qn=44, kn=44, d=64, kw=dict(is_causal=True)),
    dict(name="f16_d128_gqa", z=1, qh=4, kh=2, qn=36, kn=68, d=128, kw={}),
    dict(name="f16_d64_nosmooth_scale", z=1, qh=2, kh=1, qn=36, kn=100, d=64, kw=dict(smooth_k=False, scale=0.2)),
    dict(name="f16_d64_long", z=1, qh=1, kh=1, qn=132, kn=260, d=64, kw={}),
    dict(name="f16_d64_boolmask", z=2, qh=2, kh=2, qn=40, kn=80, d=64, kw={}, mask=dict(kind="bool", shape=(2, 1, 40, 80), dead_rows=(3, 17))),
    dict(name="f16_d64_floatmask_2d_causal", z=1, qh=2, kh=1, qn=48, kn=48, d=64, kw=dict(is_causal=True), mask=dict(kind="f16", shape=(48, 48))),
    dict(name="f16_d128_f32mask_keyonly", z=1, qh=2, kh=2, qn=36, kn=44, d=128, kw={}, mask=dict(kind="f32", shape=(1, 2, 1, 44))),
    dict(name="f16_d40_padded", z=1, qh=2, kh=2, qn=40, kn=56, d=40, kw={}),
    dict(name="f16_d80_padded_causal", z=1, qh=2, kh=1, qn=36, kn=36, d=80, kw=dict(is_causal=True)),
    dict(name="f16_d64_hadamard", z=1, qh=2, kh=2, qn=48, kn=72, d=64, kw=dict(use_hadamard=True)),
    dict(name="f16_d128_hadamard_g32", z=1, qh=2, kh=1, qn=36, kn=40, d=128, kw=dict(use_hadamard=True, hadamard_group_size=32, is_causal=True)),
    # bfloat16 (round 5): P and the output are rounded to 8 bits of mantissa, V is the bf16 operand of the second matmul
    dict(name="bf16_d64_tail", dtype="bf16", z=1, qh=2, kh=2, qn=40, kn=52, d=64, kw={}),
    dict(name="bf16_d128_gqa_causal", dtype="bf16", z=1, qh=4, kh=2, qn=44, kn=44, d=128, kw=dict(is_causal=True)),
    dict(name="bf16_d64_hadamard", dtype="bf16", z=1, qh=2, kh=2, qn=48, kn=72, d=64, kw=dict(use_hadamard=True)),
    dict(name="bf16_d64_boolmask", dtype="bf16", z=2, qh=2, kh=1, qn=40, kn=80, d=64, kw={}, mask=dict(kind="bool", shape=(2, 1, 40, 80), dead_rows=(5,))),
    # round 6: fp8 (e4m3) Q.K^T and the quantized P.V variants (triton_atten.py:303-323, 443-487).  The P quantization is per (query, key
    # BLOCK): the fixtures hold it for BLOCK_SIZE_N = 32 (meta "block_n"), the block the MI355X kernel works in
    dict(name="f16_d64_fp8qk_tail", z=1, qh=2, kh=2, qn=40, kn=52, d=64, kw=dict(matmul_dtype="fp8")),
    dict(name="f16_d128_fp8qk_gqa_causal", z=1, qh=4, kh=2, qn=44, kn=44, d=128, kw=dict(matmul_dtype="fp8", is_causal=True)),
    dict(name="f16_d64_pvint8_tail", z=1, qh=2, kh=2, qn=40, kn=84, d=64, kw=dict(pv_matmul_dtype="int8")),
    dict(name="f16_d128_pvint8_causal", z=1, qh=2, kh=1, qn=68, kn=68, d=128, kw=dict(pv_matmul_dtype="int8", is_causal=True)),
    dict(name="f16_d64_fp8qk_pvfp8", z=1, qh=2, kh=2, qn=36, kn=100, d=64, kw=dict(matmul_dtype="fp8", pv_matmul_dtype="fp8")),
    dict(name="f16_d64_pvfp8_boolmask", z=2, qh=2, kh=2, qn=40, kn=80, d=64, kw=dict(pv_matmul_dtype="fp8"), mask=dict(kind="bool", shape=(2, 1, 40, 80), dead_rows=(3,))),
    dict(name="bf16_d64_fp8qk_pvint8", dtype="bf16", z=1, qh=2, kh=2, qn=40, kn=52, d=64, kw=dict(matmul_dtype="fp8", pv_matmul_dtype="int8")),
    dict(name="f16_d64_pvint8_hadamard", z=1, qh=2, kh=2, qn=48, kn=72, d=64, kw=dict(pv_matmul_dtype="int8", use_hadamard=True)),
    dict(name="f16_d64_pvf16_tail", z=1, qh=2, kh=1, qn=36, kn=52, d=64, kw=dict(pv_matmul_dtype="float16")),
]


def bits(t):
    t = t.detach().cpu().contiguous()
    if t.dtype == torch.float16:
        return t.view(torch.uint16).numpy().copy(), "f16"
    if t.dtype == torch.bfloat16:
        return t.view(torch.uint16).numpy().copy(), "bf16"
    if t.dtype == torch.bool:
        return t.view(torch.uint8).numpy().copy(), "bool"
    if t.dtype == torch.float8_e4m3fn:
        return t.view(torch.uint8).numpy().copy(), "e4m3"
    return t.numpy().copy(), str(t.dtype).replace("torch.", "")


def run(case):
    g = torch.Generator().manual_seed(sum(map(ord, case["name"])))
    z, qh, kh, qn, kn, d = (case[k] for k in ("z", "qh", "kh", "qn", "kn", "d"))
    q = torch.randn(z, qh, qn, d, generator=g)
    k = torch.randn(z, kh, kn, d, generator=g) + 3.0 * torch.randn(1, kh, 1, d, generator=g)  # channel offsets: what smooth_k removes
    v = torch.randn(z, kh, kn, d, generator=g)
    q[..., 5] *= 6.0
    tdt = torch.bfloat16 if case.get("dtype", "f16") == "bf16" else torch.float16
    q, k, v = q.to(tdt), k.to(tdt), v.to(tdt)
    mask = None
    if "mask" in case:
        ms = case["mask"]
        if ms["kind"] == "bool":
            mask = torch.rand(ms["shape"], generator=g) > 0.35
            for r in ms.get("dead_rows", ()):
                mask[..., r, :] = False  # queries with no visible key: the reference returns 0 for them
            mask[..., 32:64] &= torch.rand(ms["shape"][:-1] + (1,), generator=g) > 0.5  # some fully masked 32-key blocks
        else:
            mask = torch.randn(ms["shape"], generator=g) * 2.0
            mask[torch.rand(ms["shape"], generator=g) < 0.2] = float("-inf")
            mask[..., 0] = 0.5  # every query keeps a visible key
            mask = mask.to(tdt if ms["kind"] == "f16" else torch.float32)
    bf16 = case.get("dtype", "f16") == "bf16"
    if bf16:
        real_quantize, captured = ta.quantize_attn, {}

        def on_bf16(q_, k_, v_, smooth_k=True, hadamard=None, **kw_):
            r = list(real_quantize(q_.to(torch.bfloat16), k_.to(torch.bfloat16), v_.to(torch.bfloat16), smooth_k=smooth_k,
                                   hadamard=None if hadamard is None else hadamard.to(torch.bfloat16), **kw_))
            captured["r"] = tuple(r)
            if r[5] is None:
                r[4] = r[4].float()  # V: the same values as float32, so that the interpreter can run the kernel
            return tuple(r)
        ta.quantize_attn = on_bf16
        try:
            out = ta.sdnq_triton_atten(q.float(), k.float(), v.float(), attn_mask=mask, **case["kw"])
        finally:
            ta.quantize_attn = real_quantize
    else:
        out = ta.sdnq_triton_atten(q, k, v, attn_mask=mask, **case["kw"])
    hadamard, hgroup = None, 0
    if case["kw"].get("use_hadamard"):  # the group / matrix choice of sdnq_triton_atten (triton_atten.py:563-569)
        hch = ta.next_power_of_2(d)
        ok, hgroup = ta.get_hadamard_group_size(hch, min(case["kw"].get("hadamard_group_size", 256), hch))
        hadamard = ta.get_hadamard(hgroup, dtype=q.dtype, device=q.device) if ok else None
    q_q, q_s, k_q, k_s, v_q, v_s, used_h, used_g = ta.quantize_attn(q, k, v, smooth_k=case["kw"].get("smooth_k", True), hadamard=hadamard,
                                                                    hadamard_group_size=hgroup or 256, matmul_dtype=case["kw"].get("matmul_dtype", "int8"),
                                                                    pv_matmul_dtype=case["kw"].get("pv_matmul_dtype"))
    assert (v_s is None and v_q.dtype == tdt) or case["kw"].get("pv_matmul_dtype")
    if bf16:  # what the kernel really consumed == what the reference's host code gives on the bfloat16 tensors
        c_ = captured["r"]
        assert torch.equal(c_[0], q_q) and torch.equal(c_[1], q_s) and torch.equal(c_[2], k_q) and torch.equal(c_[3], k_s) and torch.equal(c_[4], v_q)
    arrays, meta = {}, {"name": case["name"], "dtype": case.get("dtype", "f16"), "shape": dict(z=z, qh=qh, kh=kh, qn=qn, kn=kn, d=d), "kwargs": case["kw"],
                        "block_m": BLOCK_M, "block_n": BLOCK_N, "hadamard_group": int(used_g) if used_h else 0, "tensors": {},
                        **({"out_is": "float32: the reference kernel on the bfloat16 path's quantized operands with V as float32 (P and the output unrounded)"} if bf16 else {})}
    for key, t in (("q", q), ("k", k), ("v", v), ("out", out), ("q_q", q_q), ("q_scale", q_s), ("k_q", k_q), ("k_scale", k_s)) \
            + ((("mask", mask),) if mask is not None else ()) + ((("v_q", v_q), ("v_scale", v_s)) if v_s is not None else ()):
        arrays[key], tag = bits(t)
        meta["tensors"][key] = {"dtype": tag, "shape": list(t.shape)}
    np.savez_compressed(os.path.join(HERE, f"attn_{case['name']}.npz"), **arrays)
    with open(os.path.join(HERE, f"attn_{case['name']}.json"), "w") as f:
        json.dump(meta, f, indent=1)
    if mask is None:
        ref = torch.nn.functional.scaled_dot_product_attention(q.float(), k.float().repeat_interleave(qh // kh, 1), v.float().repeat_interleave(qh // kh, 1),
                                                               is_causal=case["kw"].get("is_causal", False), scale=case["kw"].get("scale"))
        print("wrote attn", case["name"], tuple(out.shape), "max |out - fp32 sdpa| =", float((out.float() - ref).abs().max()))
    else:
        print("wrote attn", case["name"], tuple(out.shape), "mask", mask.dtype, tuple(mask.shape), "finite:", bool(torch.isfinite(out.float()).all()))


if __name__ == "__main__":
    only = sys.argv[1:] or None
    for c in CASES:
        if only is None or c["name"] in only:
            run(c)
