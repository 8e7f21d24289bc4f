"""CPU ORACLE of the SDNQ quantized-Linear hot path -- TEST INFRASTRUCTURE ONLY.

A restatement of the reference's algorithm (Disty0/sdnq 0.2.5; citations are file:line under
/root/reference/src/sdnq/) in numpy + plain C (``sdnq_oracle.c``, compiled on demand with gcc).
Imported only by ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg as the
CHECKER; the product path (``sdnq_amd``) never touches it.

Pinning: ``tests/test_oracle_golden.py`` checks this oracle against vectors captured from the imported
reference (``tests/golden/*.npz`` made by ``tests/golden/make_golden.py``): bit-exact for unpack / dequant /
activation quantization / int8 matmul, and within the tolerances SURVEY.md 8(c) states for the
floating-point GEMMs (Hadamard, SVD, fp8, bf16 linear), whose summation order is unspecified in the reference.

Conventions: float tensors are float32 ndarrays holding values representable in the tagged dtype
("f32" | "bf16" | "f16"); weights/scales are ndarrays in the reference's LOGICAL state_dict layouts.
"""
from __future__ import annotations

import ctypes
import hashlib
import os
import platform
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "sdnq_oracle.c")
_DT = {"f32": 0, "bf16": 1, "f16": 2, "float32": 0, "bfloat16": 1, "float16": 2}


# ------------------------------------------------------------------------------------------------
# build + load the C part
# ------------------------------------------------------------------------------------------------
def _cpu_tag() -> str:
    model = platform.machine()
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    model += line
                    break
                if line.startswith("flags"):
                    model += line
                    break
    except OSError:
        pass
    return hashlib.sha1((model + open(_SRC).read()).encode()).hexdigest()[:12]


def build_oracle() -> str:
    out_dir = os.path.join(_HERE, "_build")
    os.makedirs(out_dir, exist_ok=True)
    so = os.path.join(out_dir, f"libsdnq_oracle_{_cpu_tag()}.so")
    if not os.path.exists(so):
        tmp = so + f".{os.getpid()}.tmp"
        cmd = ["gcc", "-O2", "-ftree-vectorize", "-march=native", "-fopenmp", "-ffp-contract=off", "-shared", "-fPIC",
               "-o", tmp, _SRC, "-lm"]
        subprocess.run(cmd, check=True)
        os.replace(tmp, so)
        for other in os.listdir(out_dir):  # builds for another host CPU / an older source: stale, do not let them pile up or travel
            if other.startswith("libsdnq_oracle_") and other.endswith(".so") and os.path.join(out_dir, other) != so:
                try:
                    os.remove(os.path.join(out_dir, other))
                except OSError:
                    pass
    return so


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = ctypes.CDLL(build_oracle())
        c = ctypes
        vp, i32, i64, f32 = c.c_void_p, c.c_int, c.c_int64, c.c_float
        L.orc_round_bf16.restype = f32
        L.orc_round_bf16.argtypes = [f32]
        L.orc_round_f16.restype = f32
        L.orc_round_f16.argtypes = [f32]
        L.orc_e4m3fn_to_f32.restype = f32
        L.orc_e4m3fn_to_f32.argtypes = [c.c_uint8]
        L.orc_f32_to_e4m3fn.restype = c.c_uint8
        L.orc_f32_to_e4m3fn.argtypes = [f32]
        L.orc_decode_exmy.restype = f32
        L.orc_decode_exmy.argtypes = [c.c_uint32, i32, i32, i32]
        L.orc_unpack_uint.argtypes = [i32, vp, i64, vp]
        L.orc_pack_uint.argtypes = [i32, vp, i64, vp]
        L.orc_dequant_f32.argtypes = [vp, vp, vp, i64, i64, i64, vp]
        L.orc_svd_add.argtypes = [vp, vp, vp, i64, i64, i64, i32]
        L.orc_hadamard_matrix.argtypes = [i32, f32, vp]
        L.orc_hadamard_rotate.argtypes = [vp, i64, i64, i32, f32, i32, vp]
        L.orc_rowquant_i8.argtypes = [vp, i64, i64, vp, vp, vp]
        L.orc_rowquant_fp8.argtypes = [vp, i64, i64, vp, vp]
        L.orc_scaled_mm_i8.argtypes = [vp, vp, vp, vp, vp, i64, i64, i64, i64, i32, vp]
        L.orc_scaled_mm_fp8.argtypes = [vp, vp, vp, vp, vp, i64, i64, i64, i64, i32, vp]
        L.orc_linear_float.argtypes = [vp, vp, vp, i64, i64, i64, i32, vp]
        L.orc_lowrank_bias.argtypes = [vp, vp, vp, i64, i64, i64, i32, vp]
        L.orc_num_threads.restype = i32
        L.orc_set_num_threads.argtypes = [i32]
        _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


# ------------------------------------------------------------------------------------------------
# dtype helpers
# ------------------------------------------------------------------------------------------------
def round_dtype(a: np.ndarray, tag: str) -> np.ndarray:
    """float32 -> nearest value representable in `tag` (RNE), returned as float32."""
    a = _c(a, np.float32)
    code = _DT[tag]
    if code == 0:
        return a.copy()
    if code == 1:
        u = a.view(np.uint32).astype(np.uint64)
        nan = (u & 0x7fffffff) > 0x7f800000
        r = ((u + 0x7fff + ((u >> 16) & 1)) & 0xffff0000).astype(np.uint32)
        r[nan] = 0x7fc00000
        return r.view(np.float32).reshape(a.shape)
    return a.astype(np.float16).astype(np.float32)  # numpy float16 conversion is IEEE RNE


def from_bits(bits: np.ndarray, tag: str) -> np.ndarray:
    if tag in ("f32", "float32"):
        return _c(bits, np.float32)
    if tag in ("bf16", "bfloat16"):
        return (bits.astype(np.uint32) << 16).view(np.float32)
    if tag in ("f16", "float16"):
        return bits.view(np.float16).astype(np.float32)
    raise ValueError(tag)


def to_bits(a: np.ndarray, tag: str) -> np.ndarray:
    a = _c(a, np.float32)
    if tag in ("f32", "float32"):
        return a
    if tag in ("bf16", "bfloat16"):
        return (round_dtype(a, "bf16").view(np.uint32) >> 16).astype(np.uint16)
    return a.astype(np.float16).view(np.uint16)


def e4m3_decode(codes: np.ndarray) -> np.ndarray:
    L = lib()
    lut = np.array([L.orc_e4m3fn_to_f32(i) for i in range(256)], dtype=np.float32)
    return lut[codes.astype(np.uint8)]


def e5m2_decode(codes: np.ndarray) -> np.ndarray:
    return (codes.astype(np.uint16) << 8).view(np.float16).astype(np.float32)


# ------------------------------------------------------------------------------------------------
# a19: dtype table (restated from the format rules; checked against the reference's table fixture)
# ------------------------------------------------------------------------------------------------
def dtype_info(name: str) -> dict:
    import re
    alias = {"fp8": "float8_e4m3fn", "fp16": "float16", "bf16": "bfloat16", "fp32": "float32", "bool": "uint1", "int1": "uint1",
             "fp1": "float1_e1m0fnu"}
    name = alias.get(name, name)
    m = re.fullmatch(r"(u?)int(\d+)", name)
    if m:
        bits, uns = int(m.group(2)), bool(m.group(1))
        return dict(kind="uint" if uns else "int", bits=bits, packed=bits not in (8, 16, 32), exponent=0, mantissa=0,
                    native=False, minv=0 if uns else -(1 << (bits - 1)))
    if name == "float8_e4m3fn":
        return dict(kind="float", bits=8, packed=False, exponent=4, mantissa=3, native=True)
    if name == "float8_e5m2":
        return dict(kind="float", bits=8, packed=False, exponent=5, mantissa=2, native=True)
    if name == "float16":
        return dict(kind="float", bits=16, packed=False, exponent=5, mantissa=10, native=True)
    if name == "bfloat16":
        return dict(kind="float", bits=16, packed=False, exponent=8, mantissa=7, native=True)
    m = re.fullmatch(r"float(\d+)_e(\d+)m(\d+)fn(u?)(_sdnq)?", name)
    if m:
        bits, e, mm, uns = int(m.group(1)), int(m.group(2)), int(m.group(3)), bool(m.group(4))
        return dict(kind="ufloat" if uns else "float", bits=bits, packed=True, exponent=e, mantissa=mm, native=False)
    m = re.fullmatch(r"(u?)fp(\d+)", name)
    if m:
        bits, uns = int(m.group(2)), bool(m.group(1))
        e = {1: 1, 2: 1, 3: 1, 4: 2, 5: 2, 6: 3, 7: 3, 8: 4, 9: 4}.get(bits, 5)
        return dict(kind="ufloat" if uns else "float", bits=bits, packed=True, exponent=e, mantissa=bits - e - (0 if uns else 1), native=False)
    raise KeyError(name)


# ------------------------------------------------------------------------------------------------
# a5 / a6: unpack
# ------------------------------------------------------------------------------------------------
def unpack_codes(packed: np.ndarray, bits: int, numel: int) -> np.ndarray:
    """Reference group codecs (packed_int/unpack.py) -> unsigned codes, via the bit-placement tables."""
    if bits == 8:
        return packed.reshape(-1).view(np.uint8).astype(np.int32)[:numel]
    if bits == 16:
        return packed.reshape(-1).view(np.uint16).astype(np.int32)[:numel]
    src = np.ascontiguousarray(packed).reshape(-1)
    if src.dtype in (np.int64, np.bool_):
        # 1-bit types: the reference packs torch.bool inputs with bitwise ops that promote to int64, so the
        # state_dict holds one 8-bit word per int64 element (packed_int/pack.py:309-321)
        src = src.astype(np.uint8)
    if bits < 8:
        src = src.view(np.uint8)
    else:
        src = src.view(np.uint16)
    out = np.empty(numel, dtype=np.int32)
    rc = lib().orc_unpack_uint(bits, _p(src), numel, _p(out))
    assert rc == 0, rc
    return out


def pack_codes(codes: np.ndarray, bits: int) -> np.ndarray:
    codes = _c(codes.reshape(-1), np.int32)
    G = {1: 8, 2: 4, 3: 8, 4: 2, 5: 8, 6: 4, 7: 8, 9: 16, 10: 8, 11: 16, 12: 4, 13: 16, 14: 8, 15: 16}[bits]
    wb = 8 if bits < 8 else 16
    nwords = codes.size // G * (G * bits // wb)
    out = np.zeros(nwords, dtype=np.uint8 if wb == 8 else np.uint16)
    rc = lib().orc_pack_uint(bits, _p(codes), codes.size, _p(out))
    assert rc == 0, rc
    return out


def weight_values(weight: np.ndarray, weights_dtype: str, shape) -> np.ndarray:
    """Stored weight -> numeric values (float32) of logical shape `shape`, before scaling.
    unpack_int (packed_int/__init__.py:84-88: signed = unsigned code + min) / unpack_float (packed_float.py:86)."""
    info = dtype_info(weights_dtype)
    numel = int(np.prod(shape))
    if info["kind"] in ("int", "uint"):
        if info["packed"]:
            codes = unpack_codes(weight, info["bits"], numel)
            vals = codes + (info["minv"] if info["kind"] == "int" else 0)
        else:
            raw = weight.reshape(-1)
            if info["kind"] == "int":
                vals = raw.view({8: np.int8, 16: np.int16, 32: np.int32}[info["bits"]]).astype(np.int32)
            else:
                vals = raw.view({8: np.uint8, 16: np.uint16}[info["bits"]]).astype(np.int32)
        return vals.astype(np.float32).reshape(shape)
    # floats
    if info["native"]:
        raw = weight.reshape(-1)
        if info["bits"] == 8:
            vals = e4m3_decode(raw.view(np.uint8)) if info["exponent"] == 4 else e5m2_decode(raw.view(np.uint8))
        else:
            vals = from_bits(raw.view(np.uint16), "f16" if info["exponent"] == 5 else "bf16")
        return _c(vals, np.float32).reshape(shape)
    codes = unpack_codes(weight, info["bits"], numel)
    L = lib()
    uns = 1 if info["kind"] == "ufloat" else 0
    table = np.array([L.orc_decode_exmy(int(c), info["exponent"], info["mantissa"], uns) for c in range(1 << min(info["bits"], 16))],
                     dtype=np.float32)
    return table[codes].reshape(shape)


# ------------------------------------------------------------------------------------------------
# a11: Hadamard
# ------------------------------------------------------------------------------------------------
def hadamard_scale(g: int, tag: str) -> float:
    """|H_g[0][0]|: H.div_(n**0.5) evaluated in the activation dtype (quant_utils.py:151,163)."""
    root = np.float64(g) ** 0.5
    if _DT[tag] == 0:
        return float(np.float32(1.0) / np.float32(root))
    one = round_dtype(np.array([1.0], np.float32), tag)[0]
    r = round_dtype(np.array([root], np.float32), tag)[0]
    return float(round_dtype(np.array([np.float32(one) / np.float32(r)], np.float32), tag)[0])


def get_hadamard_group_size(channel_size: int, group_size: int):
    """quant_utils.py:212-218."""
    g = 1
    while g < min(channel_size, group_size):
        g *= 2
    while channel_size % g != 0:
        g //= 2
    return g >= 4, g


def hadamard_matrix(g: int, tag: str = "f32") -> np.ndarray:
    H = np.empty((g, g), dtype=np.float32)
    lib().orc_hadamard_matrix(g, ctypes.c_float(hadamard_scale(g, tag)), _p(H))
    return H


def rotate_hadamard(x: np.ndarray, g: int, tag: str) -> np.ndarray:
    x2 = _c(x.reshape(-1, x.shape[-1]), np.float32)
    y = np.empty_like(x2)
    lib().orc_hadamard_rotate(_p(x2), x2.shape[0], x2.shape[1], g, ctypes.c_float(hadamard_scale(g, tag)), _DT[tag], _p(y))
    return y.reshape(x.shape)


# ------------------------------------------------------------------------------------------------
# module container + a4 dequantize / a7 re-quantize
# ------------------------------------------------------------------------------------------------
class OracleLinear:
    """An SDNQ-quantized Linear as the reference holds it: logical-layout ndarrays + dequantizer fields."""

    def __init__(self, deq: dict, weight, scale, zero_point=None, svd_up=None, svd_down=None, bias=None, svd_tag="bf16",
                 bias_tag=None, N=None, K=None, scale_tag="f32"):
        self.deq = deq
        # dtype the layer's scale / zero_point are STORED in: "f32" (dequantize_fp32=True, the default) or the model dtype
        # (dequantize_fp32=False, quantizer.py:147-156); `scale` / `zero_point` are float32 arrays of its values either way.
        # A 16-bit scale makes the reference carry the whole dequantize / activation-quantization arithmetic in that dtype.
        self.scale_tag = scale_tag
        self.weight, self.scale, self.zero_point = weight, scale, zero_point
        self.svd_up, self.svd_down, self.bias = svd_up, svd_down, bias  # float32 arrays of dtype-representable values
        self.svd_tag = svd_tag
        self.N, self.K = N, K
        self.result_tag = {"bfloat16": "bf16", "float16": "f16", "float32": "f32"}[deq["result_dtype"]]
        self.bias_tag = bias_tag or self.result_tag
        if scale_tag != "f32" and dtype_info(deq["weights_dtype"])["bits"] > 8:
            # the reference carries the dequantize of a 9..16-bit code in the 16-bit scale dtype (the codes themselves are rounded there);
            # that arithmetic is not restated (tools/fuzz_oracle_vs_reference.py: int12 + bf16 scales differed from the reference), and
            # the HIP path names the configuration unsupported (sdnq_amd.support)
            raise NotImplementedError("16-bit scales with storage formats wider than 8 bits are not restated")

    # -- layout facts -------------------------------------------------------------------------
    @property
    def transposed(self):  # quantizer.py:228-244
        d = self.deq
        return bool(d["use_quantized_matmul"] and not d["re_quantize_for_matmul"] and not d["is_packed"])

    def _nk_values_scale(self):
        """numeric values [N,K] (f32), scale [N,G], zp [N,G] in row-major [N][K] order."""
        d = self.deq
        N, K = self.N, self.K
        qshape = d["quantized_weight_shape"]
        vals = weight_values(self.weight, d["weights_dtype"], qshape)
        if self.transposed:  # logical [K,N]
            vals = np.ascontiguousarray(vals.reshape(K, N).T)
        vals = vals.reshape(N, K)
        P = self.positions
        group = d["group_size"] if d["group_size"] > 0 else K // P
        G = (K // P) // group * P
        sc = _c(self.scale, np.float32).reshape(-1)
        assert sc.size == N * G, (sc.size, N, G)
        zp = None if self.zero_point is None else _c(self.zero_point, np.float32).reshape(-1)
        return vals, sc, zp, group

    @property
    def positions(self) -> int:
        """Conv weights quantized along C_in keep one scale per (output channel, channel group, kernel position)
        (quantizer.py:120-123, 205-209); the flattened direct-matmul layout and Linear layers have P = 1."""
        d = self.deq
        if not str(d.get("layer_class_name", "Linear")).endswith(("Conv1d", "Conv2d", "Conv3d")) or self.transposed:
            return 1
        return int(np.prod(d["original_shape"][2:]))

    def dequant_f32_nk(self) -> np.ndarray:
        """a4 core: W[n][k] = f32(w)*s (dequantizer.py:63) or fma(f32(w), s, zp) (:27)."""
        vals, sc, zp, group = self._nk_values_scale()
        N, K, P = self.N, self.K, self.positions
        out = np.empty((N, K), dtype=np.float32)
        if P == 1:
            lib().orc_dequant_f32(_p(_c(vals, np.float32)), _p(sc), _p(zp), N, K, group, _p(out))
            return out
        # conv: k = (c, pos) with scales [N][C/group][P].  Moving the position axis in front of the channel axis turns this
        # into the contiguous-group layout of the C routine (k' = pos * C + c, scales [N][P][C/group]); element-wise, so exact.
        C = K // P
        G = C // group
        v2 = _c(vals.reshape(N, C, P).transpose(0, 2, 1).reshape(N, K), np.float32)
        sc2 = _c(sc.reshape(N, G, P).transpose(0, 2, 1).reshape(-1), np.float32)
        zp2 = None if zp is None else _c(zp.reshape(N, G, P).transpose(0, 2, 1).reshape(-1), np.float32)
        lib().orc_dequant_f32(_p(v2), _p(sc2), _p(zp2), N, K, group, _p(out))
        return np.ascontiguousarray(out.reshape(N, P, C).transpose(0, 2, 1).reshape(N, K))

    def svd_nr_rk(self):
        """svd_up as [N,R], svd_down as [R,K] (the qmm layout stores them transposed, quantizer.py:164-167)."""
        if self.svd_up is None:
            return None, None
        if self.deq["use_quantized_matmul"]:
            return np.ascontiguousarray(self.svd_up.T), np.ascontiguousarray(self.svd_down.T)
        return _c(self.svd_up, np.float32), _c(self.svd_down, np.float32)

    def dequantize(self, tag=None, hadamard=True, use_svd=True) -> np.ndarray:
        """SDNQDequantizer.__call__ -> dequantize_weight (dequantizer.py:389-429, 135-162): [N,K] in `tag`."""
        tag = tag or self.result_tag
        W = self.dequant_f32_nk()
        if self.scale_tag != "f32":
            # weight.to(scale.dtype).mul_(scale) / addcmul(zero_point, weight.to(scale.dtype), scale) (dequantizer.py:63, 27):
            # 16-bit tensors, float32 op-math, ONE rounding to the scale dtype (the products are exact in float32)
            W = round_dtype(W, self.scale_tag)
        up, down = self.svd_nr_rk()
        if up is not None and use_svd:
            lib().orc_svd_add(_p(W), _p(up), _p(down), self.N, self.K, up.shape[1], _DT[self.svd_tag])
        W = round_dtype(W, tag)
        if hadamard and self.deq["use_hadamard"]:
            W = rotate_hadamard(W, self.deq["hadamard_group_size"], tag)
        return W

    def re_quantize_matmul(self):
        """re_quantize_matmul (dequantizer.py:204-239): fp32 dequant (no SVD, Hadamard not undone) -> per-row quant.
        Returns (wq [N,K] int8 | e4m3 codes uint8, ws [N])."""
        W = self.dequant_f32_nk()
        if self.deq["quantized_matmul_dtype"] == "float16":
            # re_quantize_fp_mm (dequantizer.py:190-200) -> quantize_fp_mm with matmul_dtype "float16" (quant_utils.py:290-299): scale = amax / 65504,
            # codes = float16(clamp(nan_to_num(w / scale), +-65504)); returns (float16 [N,K], scale [N])
            assert self.scale_tag == "f32", "float16 matmul: float32 scales"
            W = _c(W, np.float32)
            with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
                sc = (np.abs(W).max(axis=1, keepdims=True).astype(np.float32) / np.float32(65504.0)).astype(np.float32)
                wq = (W / sc).astype(np.float32)
                wq = np.nan_to_num(wq, nan=0.0, posinf=np.finfo(np.float32).max, neginf=np.finfo(np.float32).min)
            return np.clip(wq, np.float32(-65504.0), np.float32(65504.0)).astype(np.float16), sc.reshape(-1)
        if self.scale_tag != "f32":  # dequantize_weight(..., dtype=scale.dtype) then quantize_*_mm in that dtype (dequantizer.py:219-239)
            if self.deq["quantized_matmul_dtype"] == "uint8":  # re_quantize_uint_mm on the scale-dtype dequantization (dequantizer.py:178-187, 219-239)
                if self.scale_tag != "bf16":
                    raise NotImplementedError("uint8 matmul with float16 scales is not restated (the reference's float16 column sums overflow)")
                return rowquant_asym_lp(round_dtype(W, self.scale_tag), self.scale_tag)
            return rowquant_lp(round_dtype(W, self.scale_tag), self.deq["quantized_matmul_dtype"], self.scale_tag)[:2]
        if self.deq["quantized_matmul_dtype"] == "uint8":
            return rowquant_asym(W)  # re_quantize_uint_mm (dequantizer.py:178-187): (wq int8, ws, zero_point)
        return rowquant(W, self.deq["quantized_matmul_dtype"])[:2]


def rowquant_asym(x_f32: np.ndarray):
    """quantize_uint_mm with matmul_dtype "uint8" -> "int8" range (quant_utils.py:277-286, get_scale_asymmetric :10-19).
    Returns (q int8 [M,K], scale [M], zero_point [M])."""
    f = np.float32
    x = _c(x_f32, f)
    xmin, xmax = x.min(-1, keepdims=True).astype(f), x.max(-1, keepdims=True).astype(f)
    scale = ((xmax - xmin).astype(f) / f(255.0)).astype(f)
    zp = (xmin - f(-128.0) * scale).astype(f)  # zero_point.sub_(scale, alpha=min); 128*scale is exact
    with np.errstate(invalid="ignore", divide="ignore"):
        q = ((x - zp).astype(f) / scale).astype(f)
    q = np.clip(np.rint(q), -128, 127)
    q = np.where(np.isnan(q), 0, q).astype(np.int8)
    return q, scale.reshape(-1), zp.reshape(-1)


def rowquant(x_f32: np.ndarray, matmul_dtype: str):
    """quantize_int_mm / quantize_fp_mm on float32 rows (quant_utils.py:265-273, 290-299) -> (q, scale[M], rowsum|None)."""
    x = _c(x_f32, np.float32)
    M, K = x.shape
    s = np.empty(M, dtype=np.float32)
    if matmul_dtype == "int8":
        q = np.empty((M, K), dtype=np.int8)
        rs = np.empty(M, dtype=np.int32)
        lib().orc_rowquant_i8(_p(x), M, K, _p(q), _p(s), _p(rs))
        return q, s, rs
    q = np.empty((M, K), dtype=np.uint8)
    lib().orc_rowquant_fp8(_p(x), M, K, _p(q), _p(s))
    return q, s, None


def rowquant_lp(x_t: np.ndarray, matmul_dtype: str, tag: str):
    """quantize_int_mm / quantize_fp_mm on rows held in a 16-bit dtype `tag` (dequantize_fp32=False: linear_int8.py:15-22
    `input.to(dtype=scale.dtype)`, quant_utils.py:265-273, 290-299).  Every torch op on 16-bit tensors computes in float32 and
    rounds its result to the dtype once: scale = round(amax / qmax), quotient = round(x / scale), then round-half-even / clamp
    (int8) or nan_to_num / clamp / cast (fp8).  Returns (q, scale [M] as float32 values of `tag`, rowsum | None)."""
    f = np.float32
    x = round_dtype(_c(x_t, f), tag)
    qmax = f(127.0 if matmul_dtype == "int8" else 448.0)
    s = round_dtype((np.abs(x).max(-1, keepdims=True).astype(f) / qmax).astype(f), tag)
    with np.errstate(invalid="ignore", divide="ignore"):
        q = round_dtype((x / s).astype(f), tag)
    if matmul_dtype == "int8":
        q = np.clip(np.rint(q), -128, 127)
        q = np.where(np.isnan(q), 0, q).astype(np.int8)
        return q, s.reshape(-1), q.astype(np.int32).sum(-1).astype(np.int32)
    q = np.clip(np.nan_to_num(q, nan=0.0, posinf=3.4028235e38, neginf=-3.4028235e38), -448.0, 448.0).astype(f)
    codes = np.array([lib().orc_f32_to_e4m3fn(float(v)) for v in q.reshape(-1)], dtype=np.uint8).reshape(q.shape)
    return codes, s.reshape(-1), None


def rowquant_asym_lp(x_t: np.ndarray, tag: str):
    """quantize_uint_mm (quant_utils.py:277-286, get_scale_asymmetric :10-19) on rows held in a 16-bit dtype `tag`
    (linear_uint8.py:15-18 `input.to(dtype=scale.dtype)`): every torch op rounds its float32 result to the dtype once --
    scale = round(round(max - min) / 255), zero_point = round(min + 128 scale), q = rint(round(round(x - zero_point) / scale)).
    Returns (q int8 [M,K], scale [M], zero_point [M]) as float32 values of `tag`."""
    f = np.float32
    x = round_dtype(_c(x_t, f), tag)
    xmin, xmax = x.min(-1, keepdims=True).astype(f), x.max(-1, keepdims=True).astype(f)
    scale = round_dtype((xmax - xmin).astype(f), tag)          # scale.sub_(zero_point)
    scale = round_dtype((scale / f(255.0)).astype(f), tag)     # .div_(max - min)
    zp = round_dtype((xmin + f(128.0) * scale).astype(f), tag)  # zero_point.sub_(scale, alpha=-128): 128 * scale is exact in float32
    with np.errstate(invalid="ignore", divide="ignore"):
        q = round_dtype((x - zp).astype(f), tag)
        q = round_dtype((q / scale).astype(f), tag)
    q = np.clip(np.rint(q), -128, 127)
    q = np.where(np.isnan(q), 0, q).astype(np.int8)
    return q, scale.reshape(-1), zp.reshape(-1)


def scaled_mm_lp(matmul_dtype: str, a, b_nk, sa, sb, bias, tag: str) -> np.ndarray:
    """int_scaled_mm_torch / fp8_scaled_mm_torch with BFLOAT16 scales (kernel_wrappers.py:132-144): the accumulator is
    converted to the scale dtype FIRST (`int_mm_func(a, b, out_dtype=scale_a.dtype)`), `.mul_(scale_a)` rounds again, and the
    last step -- `.mul_(scale_b)` or addcmul(bias, ., scale_b) in float32 op-math -- rounds once more.  (float16 scales never
    get here: the activation scale is promoted to float32, linear_int8.py:20-21, which makes the whole chain float32.)"""
    f = np.float32
    if matmul_dtype == "int8":
        acc = (a.astype(np.int64) @ b_nk.astype(np.int64).T).astype(np.int32).astype(f)  # int32 -> float32 (RNE), then -> tag
    else:
        acc = (e4m3_decode(a).astype(np.float64) @ e4m3_decode(b_nk).astype(np.float64).T).astype(f)
    t = round_dtype(acc, tag)
    t = round_dtype((t * _c(sa, f).reshape(-1, 1)).astype(f), tag)
    sbv = _c(sb, f).reshape(1, -1)
    if bias is None:
        return round_dtype((t * sbv).astype(f), tag)
    bias = _c(bias, f)
    bias = bias.reshape(1, -1) if bias.ndim == 1 else bias
    # single-rounding fma in float32 as the reference's CPU addcmul compiles to (same as scaled_mm above)
    return round_dtype((t.astype(np.float64) * sbv.astype(np.float64) + bias.astype(np.float64)).astype(f), tag)


def scaled_mm(matmul_dtype: str, a, b_nk, sa, sb, bias, out_tag: str) -> np.ndarray:
    """int_scaled_mm_torch / fp8_scaled_mm_torch (kernel_wrappers.py:132-144); b_nk is [N,K]; bias None|[N]|[M,N] f32."""
    M, K = a.shape
    N = b_nk.shape[0]
    out = np.empty((M, N), dtype=np.float32)
    bias_ld = 0
    if bias is not None:
        bias = _c(bias, np.float32)
        bias_ld = N if bias.ndim == 2 else 0
    fn = lib().orc_scaled_mm_i8 if matmul_dtype == "int8" else lib().orc_scaled_mm_fp8
    a = _c(a, np.int8 if matmul_dtype == "int8" else np.uint8)
    b_nk = _c(b_nk, np.int8 if matmul_dtype == "int8" else np.uint8)
    fn(_p(a), _p(b_nk), _p(_c(sa, np.float32).reshape(-1)), _p(_c(sb, np.float32).reshape(-1)), _p(bias), bias_ld, M, N, K,
       _DT[out_tag], _p(out))
    return out


def linear_float(x, w_nk, bias, tag: str) -> np.ndarray:
    M, K = x.shape
    N = w_nk.shape[0]
    out = np.empty((M, N), dtype=np.float32)
    lib().orc_linear_float(_p(_c(x, np.float32)), _p(_c(w_nk, np.float32)), _p(None if bias is None else _c(bias, np.float32)),
                           M, N, K, _DT[tag], _p(out))
    return out


def lowrank_bias(t, up_nr, bias, tag: str) -> np.ndarray:
    M, R = t.shape
    N = up_nr.shape[0]
    out = np.empty((M, N), dtype=np.float32)
    lib().orc_lowrank_bias(_p(_c(t, np.float32)), _p(_c(up_nr, np.float32)), _p(None if bias is None else _c(bias, np.float32)),
                           M, N, R, _DT[tag], _p(out))
    return out


# ------------------------------------------------------------------------------------------------
# a1/a3/a12/a14: the forwards
# ------------------------------------------------------------------------------------------------
def im2col(x: np.ndarray, kernel, stride, padding, dilation) -> tuple[np.ndarray, tuple]:
    """F.unfold(x, ...).transpose(1, 2) (process_conv_input, layers/conv/forward.py:75): x [B,C,H,W] ->
    ([B*Ho*Wo, C*kh*kw], (B, Ho, Wo)); columns ordered (c, i, j), zero padding."""
    B, C, H, W = x.shape
    (kh, kw), (sh, sw), (ph, pw), (dh, dw) = kernel, stride, padding, dilation
    Ho = (H + 2 * ph - dh * (kh - 1) - 1) // sh + 1
    Wo = (W + 2 * pw - dw * (kw - 1) - 1) // sw + 1
    xp = np.zeros((B, C, H + 2 * ph, W + 2 * pw), dtype=x.dtype)
    xp[:, :, ph:ph + H, pw:pw + W] = x
    cols = np.empty((B, Ho, Wo, C, kh, kw), dtype=x.dtype)
    for i in range(kh):
        for j in range(kw):
            cols[:, :, :, :, i, j] = xp[:, :, i * dh:i * dh + (Ho - 1) * sh + 1:sh, j * dw:j * dw + (Wo - 1) * sw + 1:sw].transpose(0, 2, 3, 1)
    return np.ascontiguousarray(cols.reshape(B * Ho * Wo, C * kh * kw)), (B, Ho, Wo)


def _grouped_conv_rows(mod: OracleLinear, x2d: np.ndarray, groups: int, tag: str, small: bool) -> np.ndarray:
    """groups > 1 on the unfolded rows [M, groups * K']: the float forward is F.conv*d(..., groups) on the dequantized weight
    (layers/conv/forward.py:80-81); the int8 / fp8 matmuls quantize the WHOLE row with one scale and multiply per group
    (conv_int8.py:64, 73-79; conv_fp8.py:56-60: int_mm per column slice, cat, .mul_(input_scale), addcmul(bias, ., scale))."""
    d = mod.deq
    Kg, N = mod.K, mod.N
    Ng = N // groups
    assert x2d.shape[1] == groups * Kg and N % groups == 0
    x2d = _c(x2d, np.float32)
    outs = []
    if not d["use_quantized_matmul"] or small:
        W = mod.dequantize(mod.result_tag)  # (a rotated weight is un-rotated here, as for every float forward)
        for g in range(groups):
            b = None if mod.bias is None else mod.bias[g * Ng:(g + 1) * Ng]
            outs.append(linear_float(_c(x2d[:, g * Kg:(g + 1) * Kg], np.float32), _c(W[g * Ng:(g + 1) * Ng], np.float32), b, tag))
        return np.concatenate(outs, axis=1)
    if d["use_hadamard"]:
        # conv_int8.py:52-53: the WHOLE unfolded row is rotated in blocks of the rotation group; the group divides C_in / groups
        # (quant_utils.py:222-236), hence K', so no block straddles two conv groups and every group's columns meet the rotation its
        # weight rows got at quantization
        assert Kg % d["hadamard_group_size"] == 0
        x2d = _c(rotate_hadamard(x2d, d["hadamard_group_size"], tag), np.float32)
    mmd = d["quantized_matmul_dtype"]
    assert mmd in ("int8", "uint8", "float8_e4m3fn", "fp8") and mod.svd_up is None
    f = np.float32
    K_row = groups * Kg  # input.shape[-1] of the reference: the WHOLE unfolded row
    if mod.scale_tag != "f32":
        # 16-bit scales (dequantize_fp32=False) on the plain grouped matmul (conv_int8.py:64, 73-79): the row is quantized in the scale
        # dtype; `cat(int_mm per group).to(dtype=input_scale.dtype).mul_(input_scale)` then addcmul(bias, ., scale) / .mul(scale) --
        # on bfloat16 tensors every step rounds (= scaled_mm_lp per group).  float16 scales are not restated: the activation scale is
        # promoted to float32 (linear_int8.py:20-21) but dequantize_symmetric / _asymmetric cast `acc * input_scale` to float16 before the
        # weight scale is applied (dequantizer.py:27, 63) -- an epilogue of its own, and one that overflows
        assert mmd != "uint8", "the grouped uint8 matmul on 16-bit scales is not restated"
        assert mod.scale_tag == "bf16", "grouped conv matmul on float16 scales is not restated"
        mm = "int8" if mmd == "int8" else "fp8"
        wq, ws, zp = _mm_weights(mod, mm)
        assert zp is None, "grouped conv with a weight zero point on 16-bit scales is not restated"
        xq, xs, _rs = rowquant_lp(x2d, mm, mod.scale_tag)
        wq = wq.reshape(N, Kg)
        ws = _c(ws, f).reshape(-1)
        for g in range(groups):
            sl = slice(g * Ng, (g + 1) * Ng)
            b = None if mod.bias is None else mod.bias[sl]
            a_g, w_g = np.ascontiguousarray(xq[:, g * Kg:(g + 1) * Kg]), np.ascontiguousarray(wq[sl])
            outs.append(round_dtype(scaled_mm_lp(mm, a_g, w_g, xs, np.ascontiguousarray(ws[sl]), b, "bf16"), tag))
        return np.concatenate(outs, axis=1)
    if mmd == "uint8":
        # conv_uint8.py:58-79: asymmetric activations over the whole row; zero_bias from whole-row statistics, colsum over the group's own K
        if d["re_quantize_for_matmul"]:
            wq, ws, zp = mod.re_quantize_matmul()
        else:
            assert not d["is_packed"]
            vals, ws, zpv, _group = mod._nk_values_scale()
            wq = (vals.astype(np.int32).astype(np.uint8) ^ 0x80).view(np.int8)
            zp = (zpv + f(128.0) * ws).astype(f)
        xq, xs, xzp = rowquant_asym(x2d)
        xzp = xzp[:, None]
        rowsum = xq.astype(np.int32).sum(-1)
        colsum = wq.reshape(N, Kg).astype(np.int32).sum(-1)
        bias2d = ((rowsum.astype(f) * xs.reshape(-1)).astype(f)[:, None] * zp[None, :]).astype(f)
        bias2d = (bias2d + ((colsum.astype(f) * ws).astype(f)[None, :] * xzp).astype(f)).astype(f)
        # conv_uint8.py:66: input_zero_point.mul_(K) in place (one rounding), .mul(zero_point) (one rounding), then a plain add
        bias2d = (bias2d + ((xzp * f(K_row)).astype(f) * zp[None, :]).astype(f)).astype(f)
        if mod.bias is not None:
            bias2d = (bias2d + mod.bias.astype(f)[None, :]).astype(f)
        mm = "int8"
    else:
        mm = "int8" if mmd == "int8" else "fp8"
        wq, ws, zp = _mm_weights(mod, mm)
        xq, xs, rowsum = rowquant(x2d, mm)
        bias2d = None
        if zp is not None:  # conv_int8.py:65-69: the row sum runs over the whole unfolded row, every group gets the same zero_bias row term
            bias2d = ((rowsum.astype(f) * xs.reshape(-1)).astype(f)[:, None] * zp.astype(f)[None, :]).astype(f)
            if mod.bias is not None:
                bias2d = (bias2d + mod.bias.astype(f)[None, :]).astype(f)
    wq = wq.reshape(N, Kg)
    for g in range(groups):
        sl = slice(g * Ng, (g + 1) * Ng)
        b = (None if mod.bias is None else mod.bias[sl]) if bias2d is None else np.ascontiguousarray(bias2d[:, sl])
        outs.append(scaled_mm(mm, np.ascontiguousarray(xq[:, g * Kg:(g + 1) * Kg]), np.ascontiguousarray(wq[sl]), xs,
                              np.ascontiguousarray(ws[sl]), b, tag))
    return np.concatenate(outs, axis=1)


def conv_forward(mod: OracleLinear, x: np.ndarray, conv: dict, tag: str) -> np.ndarray:
    """SDNQConv1d / SDNQConv2d / SDNQConv3d forward (layers/conv/forward.py:80-81, conv_int8.py:94-123, conv_fp8.py): unfold, the Linear
    arithmetic on [M, K] rows, fold back to NCHW.  `conv` = {"nd", "kernel_size", "stride", "padding", "dilation",
    "padding_mode", "groups"} as in the fixtures' meta."""
    groups = int(conv["groups"])
    nd = conv["nd"]
    k, s, p, dl = (tuple(conv[f]) for f in ("kernel_size", "stride", "padding", "dilation"))
    small = x.size / x.shape[2] < 32  # conv_int8.py:96 (evaluated on the conv input, before unfolding)
    if conv["padding_mode"] != "zeros":  # forward.py:57-59
        pads = [(0, 0), (0, 0)] + [(int(q), int(q)) for q in p]
        x = np.pad(x, pads, mode={"reflect": "reflect", "replicate": "edge", "circular": "wrap"}[conv["padding_mode"]])
        p = (0,) * nd
    Do = None
    if nd == 1:
        x = x[:, :, None, :]
        k, s, p, dl = (1, k[0]), (1, s[0]), (0, p[0]), (1, dl[0])
    elif nd == 3:
        # forward.py:59-73: Conv3d is padded explicitly (zeros too), then unfolded along depth, height and width; a row is ordered
        # (C_in, kd, kh, kw).  Restated as: gather the kd depth taps of every output depth into the channel axis -- Z[(b, do), (c, kd)] =
        # x[b, c, do * sd + kd * dd] -- and unfold Z in 2-D, which yields exactly that row order.
        assert dl[0] == dl[1] == dl[2], "the reference's Conv3d unfold takes dilation[0] for every axis (forward.py:62-64)"
        x = np.pad(x, [(0, 0), (0, 0)] + [(int(q), int(q)) for q in p])
        B0, C, D = x.shape[:3]
        Do = (D - dl[0] * (k[0] - 1) - 1) // s[0] + 1
        taps = (np.arange(Do)[:, None] * s[0] + np.arange(k[0])[None, :] * dl[0])  # [Do, kd]
        z = x[:, :, taps]  # [B, C, Do, kd, H, W]
        x = np.ascontiguousarray(z.transpose(0, 2, 1, 3, 4, 5)).reshape(B0 * Do, C * k[0], x.shape[3], x.shape[4])
        k, s, p, dl = k[1:], s[1:], (0, 0), dl[1:]
    x2d, (B, Ho, Wo) = im2col(np.asarray(x, dtype=np.float32), k, s, p, dl)
    if groups == 1:
        y = forward(mod, x2d, tag, small_batch=small, conv_form=True)
    else:
        y = _grouped_conv_rows(mod, x2d, groups, tag, small)
    if nd == 3:  # conv_int8.py:85-86
        return np.ascontiguousarray(y.reshape(B // Do, Do, Ho, Wo, mod.N).transpose(0, 4, 1, 2, 3))
    if nd == 1:
        return np.ascontiguousarray(y.reshape(B, Wo, mod.N).transpose(0, 2, 1))
    return np.ascontiguousarray(y.reshape(B, Ho, Wo, mod.N).transpose(0, 3, 1, 2))


def info_is_plain_uint8(d: dict) -> bool:
    """A raw (unpacked) unsigned 8-bit weight: the case whose zero point the int8 forward shifts by 128 * scale (linear_int8.py:45-50)."""
    return (not d["is_packed"]) and dtype_info(d["weights_dtype"])["kind"] == "uint"


def _mm_weights(mod: OracleLinear, mm: str):
    """The matmul operand of the int8 / fp8 forwards: (wq [N, K] int8 | e4m3 bytes, ws [N], zero-point term | None)
    (linear_int8.py:38-50, 104-107; linear_fp8.py:36-38)."""
    d = mod.deq
    K, N = mod.K, mod.N
    zp = None
    if d["re_quantize_for_matmul"]:
        wq, ws = mod.re_quantize_matmul()  # linear_int8.py:104-107
    else:
        vals, sc, zpv, group = mod._nk_values_scale()
        assert group == K, "row-wise only without re-quantization"
        ws = sc
        info = dtype_info(d["weights_dtype"])
        if mm == "int8":
            if d["is_packed"]:
                # unpack_int(..., dtype=int8): signed -> values; unsigned -> raw codes viewed as int8 (linear_int8.py:38-44)
                wq = vals.astype(np.int32).astype(np.uint8).view(np.int8) if info["kind"] == "uint" else vals.astype(np.int8)
                zp = zpv
            elif info["kind"] == "uint":  # plain uint8: w ^ 0x80, zp += 128*scale (linear_int8.py:45-50)
                wq = (vals.astype(np.int32).astype(np.uint8) ^ 0x80).view(np.int8)
                zp = (zpv + np.float32(128.0) * sc).astype(np.float32) if zpv is not None else (sc * np.float32(128.0)).astype(np.float32)
            else:
                wq = vals.astype(np.int8)
        else:
            if d["is_packed"]:  # unpack_float(...).to(float8_e4m3fn) (linear_fp8.py:37)
                wq = np.array([lib().orc_f32_to_e4m3fn(float(v)) for v in vals.reshape(-1)], dtype=np.uint8).reshape(N, K)
            else:
                wq = np.ascontiguousarray(mod.weight.reshape(K, N).T).view(np.uint8) if mod.transposed else mod.weight.reshape(N, K).view(np.uint8)
    return wq, ws, zp


def forward(mod: OracleLinear, x: np.ndarray, tag: str, want_intermediates: bool = False, small_batch=None, conv_form: bool = False):
    """SDNQLinear.forward: dispatch of get_forward_func (forward.py:39-57) + the four Linear forwards.
    small_batch: override of the M < 32 branch predicate (the conv forwards evaluate it on the un-folded input)."""
    d = mod.deq
    K, N = mod.K, mod.N
    lead = x.shape[:-1]
    x2 = _c(x.reshape(-1, K), np.float32)
    M = x2.shape[0]
    inter = {}
    mmd = d["quantized_matmul_dtype"]
    if small_batch is None:
        small_batch = M < 32
    if not d["use_quantized_matmul"] or small_batch:
        # quantized_linear_forward (layers/linear/forward.py:25-26) and the M<32 branch (linear_int8.py:102-103)
        W = mod.dequantize(mod.result_tag)  # dtype defaults to result_dtype (dequantizer.py:402-403)
        y = linear_float(x2, W, mod.bias, tag)
        return (y.reshape(*lead, N), inter) if want_intermediates else y.reshape(*lead, N)

    if mmd == "uint8":
        y = _forward_uint8(mod, x2, tag, conv_form=conv_form)
        return (y.reshape(*lead, N), inter) if want_intermediates else y.reshape(*lead, N)
    if mmd == "float16":
        y = _forward_fp16(mod, x2, tag, inter)
        return (y.reshape(*lead, N), inter) if want_intermediates else y.reshape(*lead, N)
    mm = "int8" if mmd == "int8" else "fp8"

    wq, ws, zp = _mm_weights(mod, mm)
    inter["wq"], inter["ws"] = wq, ws

    if d["use_hadamard"]:
        x2 = rotate_hadamard(x2, d["hadamard_group_size"], tag)  # linear_int8.py:55-56
        inter["xrot"] = x2
    bias = mod.bias
    if mod.svd_up is not None:  # linear_int8.py:57-62
        up, down = mod.svd_nr_rk()
        t = linear_float(round_dtype(x2, mod.svd_tag), down, None, mod.svd_tag)
        b1 = None if bias is None else round_dtype(bias, mod.svd_tag)
        bias = lowrank_bias(t, up, b1, mod.svd_tag)
    lp = mod.scale_tag != "f32"
    if lp and zp is not None and info_is_plain_uint8(d):
        # `torch.add(zero_point, scale, alpha=128)` on 16-bit tensors (linear_int8.py:47-50): float32 op-math, rounded to the scale dtype
        zp = round_dtype(zp, mod.scale_tag)
    xq, xs, rowsum = rowquant_lp(x2, mm, mod.scale_tag) if lp else rowquant(x2, mm)  # linear_int8.py:64
    inter["xq"], inter["xs"] = xq, xs
    if lp and mod.scale_tag == "bf16":
        if zp is not None:
            # linear_int8.py:65-69 on bfloat16 tensors: sum(int32).to(bf16).mul_(input_scale).mul(zero_point) [.add_(bias)], every
            # step rounded to bfloat16
            zs = round_dtype(round_dtype(rowsum.astype(np.float32), "bf16") * xs.reshape(-1), "bf16")
            zero_bias = round_dtype(zs[:, None] * zp.astype(np.float32).reshape(1, -1), "bf16")
            if bias is not None:
                b2 = _c(bias, np.float32)
                zero_bias = round_dtype(zero_bias + (b2.reshape(1, -1) if b2.ndim == 1 else b2), "bf16")
            bias = zero_bias
        y = round_dtype(scaled_mm_lp(mm, xq, wq, xs, ws, bias, "bf16"), tag)
        return (y.reshape(*lead, N), inter) if want_intermediates else y.reshape(*lead, N)
    if zp is not None:  # linear_int8.py:65-69
        zero_bias = (rowsum.astype(np.float32) * xs).astype(np.float32)[:, None] * zp.astype(np.float32)[None, :]
        zero_bias = zero_bias.astype(np.float32)
        if bias is not None:
            zero_bias = (zero_bias + bias.astype(np.float32)).astype(np.float32)
        bias = zero_bias
    y = scaled_mm(mm, xq, wq, xs, ws, bias, tag)
    return (y.reshape(*lead, N), inter) if want_intermediates else y.reshape(*lead, N)


def _forward_fp16(mod: OracleLinear, x2: np.ndarray, tag: str, inter: dict) -> np.ndarray:
    """quantized_linear_forward_fp16_matmul (layers/linear/linear_fp16.py:16-110) as the reference's CPU route computes it
    (fp_scaled_mm_torch -> fp_mm_torch, kernel_wrappers.py:115-129, 153-157), row-wise float weights without SVD / Hadamard:
      weight = decoded codes .to(float16)  (linear_fp16.py:27-31);  input, input_scale = quantize_fp_mm_input(x, f32, "float16")
      fp_mm_torch: both operands x 1 / sqrt(65536 K) in float32, rounded to float16 AGAIN; torch.mm in float16 (the accumulated dot
      product is rounded to float16); x 65536 K in float32
      out = addcmul(bias, mm * input_scale, scale)  |  mm * input_scale * scale
    The float16 GEMM's summation order is the library's: the float32-accumulated dot product rounded once to float16 stands in for it."""
    d = mod.deq
    K, N = mod.K, mod.N
    assert mod.scale_tag == "f32", "fp16 matmul oracle: float32 scales"
    if d["re_quantize_for_matmul"]:
        w16, sc = mod.re_quantize_matmul()  # float32 dequantization (no SVD term, rotation not undone) -> float16 codes per output row
    else:
        vals, sc, zpv, group = mod._nk_values_scale()
        assert group == K and zpv is None
        w16 = _c(vals, np.float32).astype(np.float16)
    if d["use_hadamard"]:
        x2 = rotate_hadamard(x2, d["hadamard_group_size"], tag)  # linear_fp16.py:35-36
        inter["xrot"] = x2
    bias2d = None
    if mod.svd_up is not None:  # linear_fp16.py:37-43: addmm(bias.to(svd dtype), mm(input.to(svd dtype), svd_down), svd_up)
        up, down = mod.svd_nr_rk()
        t = linear_float(round_dtype(x2, mod.svd_tag), down, None, mod.svd_tag)
        b1 = None if mod.bias is None else round_dtype(mod.bias, mod.svd_tag)
        bias2d = lowrank_bias(t, up, b1, mod.svd_tag)
    x = _c(x2, np.float32)
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        amax = np.abs(x).max(axis=1, keepdims=True).astype(np.float32)
        xs = (amax / np.float32(65504.0)).astype(np.float32)
        q = (x / xs).astype(np.float32)
        q = np.nan_to_num(q, nan=0.0, posinf=np.finfo(np.float32).max, neginf=np.finfo(np.float32).min)
        xq = np.clip(q, np.float32(-65504.0), np.float32(65504.0)).astype(np.float16)
    inter["xq"], inter["xs"], inter["wq"], inter["ws"] = xq, xs.reshape(-1), w16, sc
    fp16_scale = np.float32(65536.0 * K)  # (b is float16 here, never float8: kernel_wrappers.py:116-119)
    in_scale = np.float32(1.0 / float(fp16_scale) ** 0.5)
    a = (xq.astype(np.float32) * in_scale).astype(np.float16)
    b = (w16.astype(np.float32) * in_scale).astype(np.float16)
    mm = (a.astype(np.float32) @ b.astype(np.float32).T).astype(np.float16)  # torch.mm on float16 tensors returns float16
    acc = (mm.astype(np.float32) * fp16_scale).astype(np.float32)
    vv = (acc * xs).astype(np.float32)
    sb = _c(sc, np.float32).reshape(1, -1)
    if bias2d is not None:
        y = (vv.astype(np.float64) * sb.astype(np.float64) + _c(bias2d, np.float32).astype(np.float64)).astype(np.float32)
    elif mod.bias is not None:  # addcmul: the product and the sum round once (float64 holds the float32 product exactly)
        y = (vv.astype(np.float64) * sb.astype(np.float64) + _c(mod.bias, np.float32).astype(np.float64).reshape(1, -1)).astype(np.float32)
    else:
        y = (vv * sb).astype(np.float32)
    return round_dtype(y, tag)


def _fma_scalar(a: np.ndarray, k, c: np.ndarray) -> np.ndarray:
    """fl32(a * k + c) with ONE rounding: the product of two float32 values is exact in float64, the float64 sum then carries at
    most one extra rounding far below float32 precision."""
    return (a.astype(np.float64) * np.float64(k) + c.astype(np.float64)).astype(np.float32)


def _forward_uint8(mod: OracleLinear, x2: np.ndarray, tag: str, conv_form: bool = False) -> np.ndarray:
    """quantized_linear_forward_uint8_matmul (layers/linear/linear_uint8.py:27-131): asymmetric int8 activations."""
    d = mod.deq
    K, N = mod.K, mod.N
    M = x2.shape[0]
    f = np.float32
    lp = mod.scale_tag != "f32"
    if lp and mod.scale_tag != "bf16":
        # float16 scales: `sum(weight).to(dtype=scale.dtype)` (linear_uint8.py:63) is a float16 column sum of up to 128 K -- inf from
        # K = 512 on -- and the activation scale is promoted to float32 around it (:20-22); not a mode anybody can use, not restated
        raise NotImplementedError("uint8 matmul with float16 scales is not restated")
    if lp:
        return _forward_uint8_bf16(mod, x2, tag, conv_form)
    if d["re_quantize_for_matmul"]:  # linear_uint8.py:109-111: int8 codes + per-row scale and zero point, no xor
        wq, sc, zp = mod.re_quantize_matmul()
    else:
        vals, sc, zpv, group = mod._nk_values_scale()
        if d["is_packed"]:
            # linear_uint8.py:38-44: packed row-wise codes are unpacked straight to int8 -- the codes AS THEY ARE (7 bits at most: they
            # fit), no xor, the zero point as stored (None for signed formats)
            wq, zp = vals.astype(np.int8), (None if zpv is None else zpv.astype(f))
        elif dtype_info(d["weights_dtype"])["kind"] == "uint":  # linear_uint8.py:45-50: w ^ 0x80, zero_point + 128 * scale
            wq = (vals.astype(np.int32).astype(np.uint8) ^ 0x80).view(np.int8)
            zp = (zpv + f(128.0) * sc).astype(f) if zpv is not None else (sc * f(128.0)).astype(f)
        else:  # signed row-wise int8 through the uint8 matmul: no weight zero point, only the activation's (linear_uint8.py:67-68)
            wq, zp = vals.astype(np.int8), None
    if d["use_hadamard"]:
        x2 = rotate_hadamard(x2, d["hadamard_group_size"], tag)
    bias = mod.bias
    if mod.svd_up is not None:
        up, down = mod.svd_nr_rk()
        t = linear_float(round_dtype(x2, mod.svd_tag), down, None, mod.svd_tag)
        bias = lowrank_bias(t, up, None if bias is None else round_dtype(bias, mod.svd_tag), mod.svd_tag)
    q, xs, xzp = rowquant_asym(x2)  # quantize_uint_mm_input (linear_uint8.py:15-23)
    xzp = xzp[:, None]
    rowsum = q.astype(np.int32).sum(-1)
    colsum = wq.astype(np.int32).sum(-1)  # sum over K of the weight, per output channel
    if zp is None:
        zero_bias = ((colsum.astype(f) * sc).astype(f)[None, :] * xzp).astype(f)
        if bias is not None:
            zero_bias = (zero_bias + bias.astype(f)).astype(f)
        return scaled_mm("int8", q, wq, xs, sc, zero_bias, tag)
    zero_bias = ((rowsum.astype(f) * xs).astype(f)[:, None] * zp[None, :]).astype(f)
    zero_bias = (zero_bias + ((colsum.astype(f) * sc).astype(f)[None, :] * xzp).astype(f)).astype(f)
    # zero_bias.add_(mul(xzp, zp), alpha=K) (linear_uint8.py:66): torch's CPU add-with-alpha is ONE fused multiply-add per element
    # (vec::fmadd), not a rounded product followed by a rounded sum -- found by tools/fuzz_modes.py in round 4 (57 of 245 760 outputs of a
    # 640 x 384 x 368 layer differed from the reference with two roundings; the small fixtures never showed it)
    if conv_form:  # conv_uint8.py:66: input_zero_point.mul_(K) in place, .mul(zero_point), a plain add_
        zero_bias = (zero_bias + ((xzp * f(K)).astype(f) * zp[None, :]).astype(f)).astype(f)
    else:
        zero_bias = _fma_scalar((xzp * zp[None, :]).astype(f), f(K), zero_bias)
    if bias is not None:
        zero_bias = (zero_bias + bias.astype(f)).astype(f)
    return scaled_mm("int8", q, wq, xs, sc, zero_bias, tag)


def _forward_uint8_bf16(mod: OracleLinear, x2: np.ndarray, tag: str, conv_form: bool = False) -> np.ndarray:
    """The uint8 matmul of a layer whose scale / zero point are stored in bfloat16 (dequantize_fp32=False): the chain of
    linear_uint8.py:27-102 on bfloat16 tensors -- every torch op computes in float32 and rounds its result to bfloat16 once."""
    d = mod.deq
    K = mod.K
    f = np.float32
    r = lambda a: round_dtype(np.asarray(a, dtype=f), "bf16")  # noqa: E731
    if tag != "bf16":
        raise NotImplementedError("bfloat16 scales with another activation dtype are not restated")
    if d["re_quantize_for_matmul"]:
        wq, sc, zp = mod.re_quantize_matmul()
    else:
        vals, sc, zpv, group = mod._nk_values_scale()
        sc = sc.reshape(-1)
        if d["is_packed"]:  # linear_uint8.py:38-44: the unpacked codes as they are, the stored zero point (no arithmetic, so no rounding)
            wq, zp = vals.astype(np.int8), (None if zpv is None else _c(zpv, f).reshape(-1))
        elif dtype_info(d["weights_dtype"])["kind"] == "uint":  # linear_uint8.py:45-50 on bfloat16 tensors
            wq = (vals.astype(np.int32).astype(np.uint8) ^ 0x80).view(np.int8)
            zp = r(zpv.reshape(-1) + f(128.0) * sc) if zpv is not None else r(sc * f(128.0))
        else:
            wq, zp = vals.astype(np.int8), None
    if d["use_hadamard"]:
        x2 = rotate_hadamard(x2, d["hadamard_group_size"], tag)
    bias = mod.bias
    if mod.svd_up is not None:
        up, down = mod.svd_nr_rk()
        t = linear_float(round_dtype(x2, mod.svd_tag), down, None, mod.svd_tag)
        bias = lowrank_bias(t, up, None if bias is None else round_dtype(bias, mod.svd_tag), mod.svd_tag)
    q, xs, xzp = rowquant_asym_lp(x2, "bf16")  # quantize_uint_mm_input(input, dtype=scale.dtype) (linear_uint8.py:15-23)
    rowsum = q.astype(np.int32).sum(-1)
    colsum = wq.astype(np.int32).sum(-1)
    wcs = r(r(colsum.astype(f)) * sc)                   # sum(weight, int32).to(scale.dtype).mul_(scale)
    t2 = r(wcs[None, :] * xzp[:, None])                 # .mul(input_zero_point)
    if zp is None:
        zero_bias = t2
    else:
        t1 = r(r(r(rowsum.astype(f)) * xs)[:, None] * zp[None, :])  # sum(input, int32).to(dtype).mul_(input_scale).mul(zero_point)
        zero_bias = r(t1 + t2)
        if conv_form:  # conv_uint8.py:66
            zero_bias = r(zero_bias + r(r(xzp * f(K))[:, None] * zp[None, :]))
        else:  # zero_bias.add_(mul(input_zero_point, zero_point), alpha=K): float32 fused multiply-add, then the bfloat16 rounding
            zero_bias = r(_fma_scalar(r(xzp[:, None] * zp[None, :]), f(K), zero_bias))
    if bias is not None:
        b2 = _c(bias, f)
        zero_bias = r(zero_bias + (b2.reshape(1, -1) if b2.ndim == 1 else b2))
    return round_dtype(scaled_mm_lp("int8", q, wq, xs, sc, zero_bias, "bf16"), tag)


# ---- quantized attention forward ---------------------------------------------------------------
def _mm_key(matmul_dtype):
    if matmul_dtype in ("auto", "enabled", "uint8", "int8"):  # triton_atten.py:452-455
        return "int8"
    if matmul_dtype in ("fp8", "float8_e4m3fn"):
        return "fp8"
    if matmul_dtype == "float16":
        return "float16"
    raise ValueError(f"attention oracle: matmul dtype {matmul_dtype!r}")


def _f32_to_e4m3(x: np.ndarray) -> np.ndarray:
    flat = _c(x, np.float32).reshape(-1)
    return np.array([lib().orc_f32_to_e4m3fn(float(v)) for v in flat], dtype=np.uint8).reshape(x.shape)


def _rowquant_attn(x2: np.ndarray, key: str):
    """quantize_int_mm / quantize_fp_mm (quant_utils.py:265-273, 290-299) of float32 rows -> (operand VALUES as float32 [exact for all three
    formats], scale [M], stored codes: int8 / e4m3 bytes / float16)."""
    f = np.float32
    if key in ("int8", "fp8"):
        q, s, _ = rowquant(x2, key)
        return (q.astype(f) if key == "int8" else e4m3_decode(q).astype(f)), s, q
    x = _c(x2, f)
    s = (np.abs(x).max(-1, keepdims=True).astype(f) / f(65504.0)).astype(f)
    with np.errstate(invalid="ignore", divide="ignore"):
        q = (x / s).astype(f)
    q = np.clip(np.nan_to_num(q, nan=0.0, posinf=3.4028235e38, neginf=-3.4028235e38), f(-65504.0), f(65504.0)).astype(np.float16)
    return q.astype(f), s.reshape(-1), q


def attention_quantize(q: np.ndarray, k: np.ndarray, smooth_k: bool = True, hadamard_group: int = 0, tag: str = "bf16", matmul_dtype: str = "int8",
                       v: np.ndarray | None = None, pv_matmul_dtype=None):
    """quantize_attn (kernels/triton_atten.py:443-487): K minus its token mean in fp32 (:457-463), then quantize_int_mm / quantize_fp_mm per
    token (quant_utils.py:265-273, 290-299); with pv_matmul_dtype the (rotated) V per token too (:478-483).
    q [Z,H,QN,D], k / v [Z,KH,KN,D] float32 values.
    Returns (q_q codes, q_scale [Z,H,QN], k_q codes, k_scale [Z,KH,KN]) -- int8 or e4m3 bytes -- and, when v is given, a dict with the
    float32 VALUES of the three operands (+ v codes / scales when V is quantized) as fifth element."""
    f = np.float32
    q, k = _c(q, f), _c(k, f)
    if smooth_k:
        k = (k - k.mean(axis=2, keepdims=True, dtype=f)).astype(f)
    if hadamard_group:  # apply_hadamard(q) / rotate_hadamard(k.to(hadamard.dtype)) in the tensor dtype (:464-467)
        q = rotate_hadamard(q, hadamard_group, tag)
        k = rotate_hadamard(round_dtype(k, tag), hadamard_group, tag)
    d = q.shape[-1]
    key = _mm_key(matmul_dtype)
    qv, qs, qq = _rowquant_attn(q.reshape(-1, d), key)
    kv, ks, kq = _rowquant_attn(k.reshape(-1, d), key)
    res = (qq.reshape(q.shape), qs.reshape(q.shape[:-1]), kq.reshape(k.shape), ks.reshape(k.shape[:-1]))
    if v is None:
        return res
    vals = dict(q=qv.reshape(q.shape), k=kv.reshape(k.shape), v=_c(v, f), v_scale=None, v_q=None)
    if pv_matmul_dtype not in (None, "auto", "none", "no", "disabled"):
        vv = _c(v, f)
        if hadamard_group:  # rotate_hadamard(v.to(dtype=hadamard.dtype)) (:479-480)
            vv = rotate_hadamard(round_dtype(vv, tag), hadamard_group, tag)
        pk = _mm_key("int8" if pv_matmul_dtype in ("enabled", "uint8") else pv_matmul_dtype)
        vq, vs, vc = _rowquant_attn(vv.reshape(-1, vv.shape[-1]), pk)
        vals.update(v=vq.reshape(vv.shape), v_scale=vs.reshape(vv.shape[:-1]), v_q=vc.reshape(vv.shape), pv=pk)
    return res + (vals,)


def attention(q: np.ndarray, k: np.ndarray, v: np.ndarray, tag: str, is_causal: bool = False, scale=None, smooth_k: bool = True,
              block_n: int = 32, out_tag: str | None = None, want_intermediates: bool = False, hadamard_group: int = 0,
              mask: np.ndarray | None = None, matmul_dtype: str = "int8", pv_matmul_dtype=None):
    """sdnq_triton_atten (kernels/triton_atten.py:540-618): int8 or fp8 (e4m3) Q.K^T, P.V in the value dtype or quantized (int8 / fp8 /
    float16, :303-323): the online-softmax loop of sdnq_attn_kernel (:143-335) over key blocks of `block_n` -- the quantized P is scaled
    per (query, key BLOCK), so block_n is part of the result there --, all queries of a head at once.
    q/k/v: float32 VALUES of `tag` tensors [Z,H,N,D]; returns float32 values rounded to out_tag (default tag).
    mask: None, a bool / int8 array (0 = masked out, :290-291) or a float array added to the base-2 logits as is (:292-293),
    broadcastable to [Z,H,QN,KN] after left-padding to 4-D (get_attn_inputs :520-527)."""
    f = np.float32
    Z, QH, QN, D = q.shape
    _, KH, KN, _ = k.shape
    sm_scale = f(D ** -0.5 if scale is None else scale)                       # :512-513
    log2_sm = f(sm_scale * f(1.4426950408889634))                            # :203
    qq, qs, kq, ks, vals = attention_quantize(q, k, smooth_k, hadamard_group, tag, matmul_dtype, v=v, pv_matmul_dtype=pv_matmul_dtype)
    pv = vals.get("pv")
    vop, vsc = vals["v"], vals["v_scale"]
    out = np.empty((Z, QH, QN, D), dtype=f)
    qidx = np.arange(QN)[:, None]
    mask_is_bool = False
    if mask is not None:
        mask_is_bool = mask.dtype in (np.bool_, np.int8)
        mask = mask.reshape((1,) * (4 - mask.ndim) + mask.shape)
        mask = np.broadcast_to(mask != 0 if mask_is_bool else mask.astype(f), (Z, QH, QN, KN))
    for z in range(Z):
        for h in range(QH):
            kh = (h * KH) // QH                                              # :212-213
            # int8: exact; fp8: every product is exact in float32, the sum of D of them in float64 rounds once (tl.dot accumulates in float32)
            S = (vals["q"][z, h].astype(np.float64) @ vals["k"][z, kh].astype(np.float64).T).astype(f)
            m = np.full(QN, -np.inf, dtype=f)
            l = np.ones(QN, dtype=f)
            acc = np.zeros((QN, D), dtype=f)
            for n0 in range(0, KN, block_n):
                n1 = min(n0 + block_n, KN)
                if is_causal and QN <= n0:
                    continue
                s = ((S[:, n0:n1] * qs[z, h][:, None]).astype(f) * ks[z, kh][None, n0:n1]).astype(f)
                s = (s * log2_sm).astype(f)                                  # :278 / :283
                if is_causal:
                    s = np.where(qidx >= np.arange(n0, n1)[None, :], s, f(-np.inf))  # :287-288
                if mask is not None:
                    mb = mask[z, h, :, n0:n1]
                    s = np.where(mb, s, f(-np.inf)) if mask_is_bool else (s + mb).astype(f)
                m_new = np.maximum(m, s.max(axis=1))
                dead = np.isneginf(m_new)  # no visible key so far: alpha = exp2(0), p = exp2(-inf - 0) (:299-301)
                with np.errstate(invalid="ignore"):
                    alpha = np.where(dead, f(1), np.exp2(m - m_new)).astype(f)
                    p = np.exp2(s - np.where(dead, f(0), m_new)[:, None]).astype(f)
                l = (l * alpha + p.sum(axis=1, dtype=f)).astype(f)           # :308
                acc = (acc * alpha[:, None]).astype(f)
                vb = _c(vop[z, kh, n0:n1], f)
                if pv is None:
                    acc = (acc + round_dtype(p, tag) @ vb).astype(f)         # p.to(v.dtype); fp32 accumulate (:332-333)
                else:                                                        # :311-327
                    p = (p * vsc[z, kh][None, n0:n1]).astype(f)
                    qmax = {"int8": 127.0, "fp8": 448.0, "float16": 65504.0}[pv]
                    ps = (p.max(axis=1, keepdims=True) * f(1.0 / qmax)).astype(f)
                    ps = np.where(ps <= f(2e-38), f(1.0), ps).astype(f)
                    inv = (f(1.0) / ps).astype(f)
                    if pv == "int8":
                        pq = np.floor((p.astype(np.float64) * inv.astype(np.float64) + 0.5).astype(f)).astype(f)  # floor(fma(p, 1 / p_scale, 0.5))
                    elif pv == "fp8":
                        pq = e4m3_decode(_f32_to_e4m3((p * inv).astype(f))).astype(f)
                    else:
                        pq = (p * inv).astype(f).astype(np.float16).astype(f)
                    dot = (pq.astype(np.float64) @ vb.astype(np.float64)).astype(f)  # int8: exact; fp8 / f16: one rounding of the float32 sum
                    acc = (dot.astype(np.float64) * ps.astype(np.float64) + acc.astype(np.float64)).astype(f)  # fma(dot, p_scale, acc)
                m = m_new
            out[z, h] = acc * (f(1.0) / l)[:, None]                          # :336
    if pv is not None and hadamard_group:  # the output comes back out of the rotated basis (triton_atten.py:609-612), in the OUTPUT dtype
        out = rotate_hadamard(round_dtype(out, out_tag or tag), hadamard_group, out_tag or tag)
    out = round_dtype(out, out_tag or tag)
    if want_intermediates:
        return out, dict(q_q=qq, q_scale=qs, k_q=kq, k_scale=ks, v_q=vals["v_q"], v_scale=vals["v_scale"])
    return out
