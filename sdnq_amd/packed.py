"""Load-time codecs of the SDNQ storage formats (host side, torch ops; runs on CPU or GPU tensors).

The hot path only ever *unpacks*, and does so inside the HIP kernels (csrc/unpack_dev.h).  Packing is needed
when this package quantizes a float layer itself (quantizer.py) and in tests.  Formats are the reference's
(packed_int/pack.py, packed_float.py:27-82); the implementation here is table-driven: one generic
bit-scatter over the placement map of each format instead of one hand-written function per bit width.
"""
from __future__ import annotations

import functools

import torch

from .common import dtype_dict

# codec group geometry: bits -> (elements per group, words per group, word bits)
_GEOM = {1: (8, 1, 8), 2: (4, 1, 8), 3: (8, 3, 8), 4: (2, 1, 8), 5: (8, 5, 8), 6: (4, 3, 8), 7: (8, 7, 8),
         9: (16, 9, 16), 10: (8, 5, 16), 11: (16, 11, 16), 12: (4, 3, 16), 13: (16, 13, 16), 14: (8, 7, 16), 15: (16, 15, 16)}


def _place(bits: int, e: int, b: int):
    """(word, bit position) of bit ``b`` of element ``e`` of a codec group."""
    if bits in (1, 2, 4):
        return 0, bits * e + b
    if bits == 3:
        if e < 3: return e, b
        if e < 6: return e - 3, 3 + b
        return (e - 6, 6 + b) if b < 2 else (2, e)
    if bits == 5:
        if e < 5: return e, b
        if b < 3: return e - 5, 5 + b
        if e == 5: return 3, 2 + b
        if e == 6: return 4, 2 + b
        return (4, 7) if b == 3 else (3, 7)
    if bits == 6:
        return (e, b) if e < 3 else (2 - b // 2, 6 + (b & 1))
    if bits == 7:
        return (e, b) if e < 7 else (6 - b, 7)
    if bits == 9:
        if e < 8: return e, b
        return (e - 8, 9 + b) if b < 7 else (8, 2 * (e - 8) + b - 7)
    if bits == 10:
        if e < 5: return e, b
        if b < 6: return e - 5, 10 + b
        if e == 5: return 3, 4 + b
        if e == 6: return 4, 4 + b
        return (4, 8 + b) if b < 8 else (3, 6 + b)
    if bits == 11:
        if e < 8: return e, b
        if b < 5: return e - 8, 11 + b
        if e < 11: return e, b - 5
        if e < 14: return e - 3, 1 + b
        if b < 9: return e - 6, 7 + b
        return 10, (3 if e == 14 else 5) + b
    if bits == 12:
        return (e, b) if e < 3 else (2 - b // 4, 12 + (b & 3))
    if bits == 13:
        if e < 13: return e, b
        return (3 * (b // 3) + e - 13, 13 + b % 3) if b < 12 else (12, e)
    if bits == 14:
        return (e, b) if e < 7 else (6 - b // 2, 14 + (b & 1))
    if bits == 15:
        return (e, b) if e < 15 else (14 - b, 15)
    raise ValueError(bits)


@functools.lru_cache(maxsize=None)
def _placement(bits: int):
    g, w, _ = _GEOM[bits]
    return [(e, b) + _place(bits, e, b) for e in range(g) for b in range(bits)]


@functools.lru_cache(maxsize=None)
def _base_elements(bits: int):
    """[(word, element)] for the elements a word holds at shift 0.  The reference ORs those in UNMASKED
    (packed_int/pack.py, e.g. :115 `packed[:, :8] | (packed[:, 8:] << 11)`), so a code that does not fit in `bits` bits --
    uint9..15 have max = 2**bits in the dtype table (common.py), and the largest element of every asymmetric group
    quantizes to exactly that -- leaks its high bits into the word.  Reproduced for byte-identical checkpoints."""
    g, w, _ = _GEOM[bits]
    out = []
    for e in range(g):
        places = [_place(bits, e, b) for b in range(bits)]
        if all(p == (places[0][0], b) for b, p in enumerate(places)):
            out.append((places[0][0], e))
    return out


def pack_uint(codes: torch.Tensor, bits: int) -> torch.Tensor:
    """Unsigned codes (any int dtype, numel % group == 0) -> packed words, [numel/G, W] (1-D for 1/2/4 bits)."""
    g, w, wb = _GEOM[bits]
    c = codes.reshape(-1, g).to(torch.int32)
    out = torch.zeros((c.shape[0], w), dtype=torch.int32, device=codes.device)
    for e, b, word, pos in _placement(bits):
        out[:, word] |= ((c[:, e] >> b) & 1) << pos
    for word, e in _base_elements(bits):
        out[:, word] |= c[:, e] & (((1 << wb) - 1) & ~((1 << bits) - 1))
    if wb == 8:
        out = out.to(torch.uint8)
    else:
        out = out.to(torch.int16) if bits != 16 else out  # int16 container: values >= 2^15 wrap like the reference's int16 ops
    return out.reshape(-1) if w == 1 else out


def unpack_uint(packed: torch.Tensor, bits: int, shape) -> torch.Tensor:
    g, w, wb = _GEOM[bits]
    p = packed.reshape(-1, w).to(torch.int32)
    if wb == 16:
        p = p & 0xffff
    else:
        p = p & 0xff
    out = torch.zeros((p.shape[0], g), dtype=torch.int32, device=packed.device)
    for e, b, word, pos in _placement(bits):
        out[:, e] |= ((p[:, word] >> pos) & 1) << b
    return out.reshape(shape)


def pack_int(tensor: torch.Tensor, weights_dtype: str) -> torch.Tensor:
    """Signed values are stored as value - min (reference packed_int/__init__.py:77-80)."""
    ent = dtype_dict[weights_dtype]
    t = tensor.to(torch.int32)
    if not ent["is_unsigned"]:
        t = t - ent["min"]
    return pack_uint(t, ent["num_bits"])


def unpack_int(packed: torch.Tensor, weights_dtype: str, shape, dtype: torch.dtype | None = None) -> torch.Tensor:
    ent = dtype_dict[weights_dtype]
    t = unpack_uint(packed, ent["num_bits"], shape)
    if not ent["is_unsigned"]:
        t = t + ent["min"]
        return t.to(ent["torch_dtype"] if dtype is None else dtype)
    return t.to(torch.uint8 if ent["num_bits"] < 8 else torch.int16)


# ---- custom eXmY floats -------------------------------------------------------------------------
def float_to_code(x: torch.Tensor, weights_dtype: str) -> torch.Tensor:
    """float32 values (already clamped to the format's range) -> eXmY codes (int32).

    Restates the reference's encoder (packed_float.py:27-73) including its rounding rule: the mantissa is
    rounded UP only when the top four dropped bits exceed one half (ties and near-ties truncate), subnormals
    (|x| < 2^(1-bias)) are rounded half-to-even on the 2^(1-bias-M) grid.
    """
    ent = dtype_dict[weights_dtype]
    e, m, total = ent["exponent"], ent["mantissa"], ent["num_bits"]
    unsigned = ent["is_unsigned"]
    bias = (1 << (e - 1)) - 1
    drop = 23 - m
    bits = x.to(torch.float32).contiguous().view(torch.int32)
    top4 = bits & (((1 << drop) - 1) & ~((1 << max(drop - 4, 0)) - 1))
    bits = torch.where(top4 > (1 << (drop - 1)), bits + (1 << drop), bits)
    sign = (bits >> 31) & 1
    if e < 8:
        # subnormals: integer mantissa on the 2^(1-bias-M) grid, written into the f32 mantissa position (a carry out of
        # the mantissa lands in bit 0 of the f32 exponent field, i.e. becomes exponent code 1 when e > 1)
        min_normal = 2.0 ** (1 - bias)
        mag = bits.view(torch.float32).abs()
        sub_bits = torch.round(mag * ((1 << m) / min_normal)).to(torch.int32) << drop
        bits = torch.where(mag < min_normal, sub_bits, bits)
    exp8 = (bits >> 23) & 0xff
    mant = (bits >> drop) & ((1 << m) - 1)
    # exponent re-bias by bit surgery, as the reference does it: [msb of the f32 exponent][its low e-1 bits]
    # (equals exp8 - 127 + bias on the representable range; for e == 1 a subnormal carry is dropped, like the reference)
    new_exp = ((exp8 >> 7) << (e - 1)) | (exp8 & ((1 << (e - 1)) - 1))
    code = (new_exp << m) | mant
    if not unsigned:
        code = code | (sign << (e + m))
    return code & ((1 << total) - 1)


def pack_float(x: torch.Tensor, weights_dtype: str) -> torch.Tensor:
    ent = dtype_dict[weights_dtype]
    code = float_to_code(x, weights_dtype)
    if ent["num_bits"] == 8:
        return code.to(torch.uint8)
    if ent["num_bits"] == 16:
        return code.to(torch.uint16)
    return pack_uint(code, ent["num_bits"])


def decode_float(code: torch.Tensor, weights_dtype: str) -> torch.Tensor:
    ent = dtype_dict[weights_dtype]
    e, m = ent["exponent"], ent["mantissa"]
    bias = (1 << (e - 1)) - 1
    c = code.to(torch.int32)
    mant = (c & ((1 << m) - 1)).to(torch.float32)
    ex = (c >> m) & ((1 << e) - 1)
    normal = torch.ldexp(1.0 + mant / (1 << m), (ex - bias).to(torch.int32))
    sub = torch.ldexp(mant / (1 << m), torch.full_like(ex, 1 - bias))
    v = torch.where(ex == 0, sub, normal)
    if not ent["is_unsigned"]:
        neg = ((c >> (e + m)) & 1).bool() & (v != 0)
        v = torch.where(neg, -v, v)
    return v


def unpack_float(packed: torch.Tensor, weights_dtype: str, shape) -> torch.Tensor:
    ent = dtype_dict[weights_dtype]
    if ent["num_bits"] in (8, 16):
        code = packed.reshape(shape).to(torch.int32) & ((1 << ent["num_bits"]) - 1)
    else:
        code = unpack_uint(packed, ent["num_bits"], shape)
    return decode_float(code, weights_dtype)
