"""ctypes binding of the C-ABI library ``libsdnq_hip.so`` (include/sdnq_hip.h).

The shared object is built in-tree by ``sdnq_amd/csrc/build.sh`` (hipcc --offload-arch=gfx950) so
that it travels with the source snapshot; nothing here falls back to another implementation: if
the library is missing or the device is not gfx950 the product path raises.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SDNQ_HIP_LIB") or os.path.join(_HERE, "libsdnq_hip.so")  # override: development builds only
_CSRC = os.path.join(_HERE, "csrc")

# enums of include/sdnq_hip.h
F32, BF16, F16 = 0, 1, 2
MM_I8, MM_FP8, MM_F16 = 0, 1, 2
ST_PACKED_U8, ST_PACKED_I16, ST_RAW8, ST_RAW16 = 0, 1, 2, 3
KIND_INT, KIND_UINT, KIND_FLOAT, KIND_UFLOAT = 0, 1, 2, 3

EXPORTS = [
    "sdnq_hip_version", "sdnq_hip_strerror", "sdnq_hip_device_supported", "sdnq_hip_rowquant",
    "sdnq_hip_scaled_mm", "sdnq_hip_dequant", "sdnq_hip_requant", "sdnq_hip_unpack_mm", "sdnq_hip_hadamard",
    "sdnq_hip_lowrank_down", "sdnq_hip_scaled_mm_lowrank", "sdnq_hip_linear_float", "sdnq_hip_linear_skinny",
    "sdnq_hip_quantize_weight", "sdnq_hip_im2col", "sdnq_hip_im2col_rowquant", "sdnq_hip_scaled_mm_nchw",
    "sdnq_hip_linear_skinny_svd", "sdnq_hip_linear_w8a8", "sdnq_hip_requant_asym",
    "sdnq_hip_im2col_rowquant_z", "sdnq_hip_attn_prepare", "sdnq_hip_attn_fwd", "sdnq_hip_attn_fwd_q16", "sdnq_hip_attn_prepare_ex", "sdnq_hip_attn_fwd_ex", "sdnq_hip_attn", "sdnq_hip_attn_workspace_bytes", "sdnq_hip_scaled_mm_multi", "sdnq_hip_linear_float_multi",
    "sdnq_hip_scaled_mm_grouped", "sdnq_hip_set_tile_override", "sdnq_hip_linear_w8a16", "sdnq_hip_linear_w8a16_grouped",
    "sdnq_hip_rowquant_lp", "sdnq_hip_rowquant_lp_asym", "sdnq_hip_scaled_mm_lp", "sdnq_hip_scaled_mm_lp_uzp", "sdnq_hip_unshard_columns", "sdnq_hip_requant_ws", "sdnq_hip_linear", "sdnq_hip_linear_workspace_bytes",
    "sdnq_hip_scaled_mm_strided", "sdnq_hip_linear_float_strided", "sdnq_hip_scaled_mm_lp_zp",
    "sdnq_hip_push_post", "sdnq_hip_push_columns", "sdnq_hip_scaled_mm_lowrank_strided", "sdnq_hip_prefetch", "sdnq_hip_prefetch_hint",
    "sdnq_hip_signal_alloc", "sdnq_hip_signal_free", "sdnq_hip_ipc_export", "sdnq_hip_ipc_import", "sdnq_hip_ipc_close",
    "sdnq_hip_linear_w8a8_fused", "sdnq_hip_linear_w8a8_fused_supported", "sdnq_hip_scaled_mm_lp_uzp_svd", "sdnq_hip_stream_capture_id",
    "sdnq_hip_scaled_mm_tile", "sdnq_hip_lut4_build", "sdnq_hip_scaled_mm_w4", "sdnq_hip_scaled_mm_w4_supported",
    "sdnq_hip_rowquant_f16", "sdnq_hip_scaled_mm_f16",
]


class SdnqWeight(ctypes.Structure):
    _fields_ = [
        ("weight", ctypes.c_void_p), ("scale", ctypes.c_void_p), ("zero_point", ctypes.c_void_p),
        ("svd_up", ctypes.c_void_p), ("svd_down", ctypes.c_void_p),
        ("n", ctypes.c_int32), ("k", ctypes.c_int32), ("group_size", ctypes.c_int32), ("svd_rank", ctypes.c_int32),
        ("svd_dtype", ctypes.c_int32), ("storage", ctypes.c_int32), ("kind", ctypes.c_int32), ("bits", ctypes.c_int32),
        ("exponent", ctypes.c_int32), ("mantissa", ctypes.c_int32), ("native_float", ctypes.c_int32),
        ("positions", ctypes.c_int32), ("scale_dtype", ctypes.c_int32),
    ]


class SdnqLinearArgs(ctypes.Structure):
    """POD arguments of sdnq_hip_linear (include/sdnq_hip.h): the whole quantized-matmul forward of one layer behind one call."""
    _fields_ = [
        ("struct_size", ctypes.c_int32), ("mm_dtype", ctypes.c_int32), ("x_dtype", ctypes.c_int32), ("out_dtype", ctypes.c_int32),
        ("bias_dtype", ctypes.c_int32), ("svd_dtype", ctypes.c_int32), ("hadamard_group", ctypes.c_int32), ("svd_rank", ctypes.c_int32),
        ("asymmetric", ctypes.c_int32), ("x_prequantized", ctypes.c_int32),
        ("m", ctypes.c_int64), ("n", ctypes.c_int64), ("k", ctypes.c_int64), ("ldx", ctypes.c_int64),
        ("x", ctypes.c_void_p), ("out", ctypes.c_void_p), ("wq", ctypes.c_void_p), ("ws", ctypes.c_void_p), ("bias", ctypes.c_void_p),
        ("svd_down", ctypes.c_void_p), ("svd_up", ctypes.c_void_p), ("zp", ctypes.c_void_p), ("w_colsum_scaled", ctypes.c_void_p),
        ("xq", ctypes.c_void_p), ("xs", ctypes.c_void_p), ("rowsum", ctypes.c_void_p), ("xrot", ctypes.c_void_p), ("xzp", ctypes.c_void_p),
        ("workspace", ctypes.c_void_p), ("workspace_bytes", ctypes.c_int64),
    ]


class SdnqGemmUnit(ctypes.Structure):
    """One unit of output channels of a grouped scaled matmul (include/sdnq_hip.h); the table lives in DEVICE memory."""
    _fields_ = [("b", ctypes.c_void_p), ("sb", ctypes.c_void_p), ("bias", ctypes.c_void_p), ("n_start", ctypes.c_int64),
                ("n_seg", ctypes.c_int32), ("n_loc", ctypes.c_int32)]


class SdnqHipError(RuntimeError):
    pass


_lock = threading.Lock()
_lib = None


_SRCS = ("api", "rowquant", "gemm", "gemm_aq", "gemm_ks", "gemm_w4", "dequant", "quantize", "conv", "attention", "parallel")
_FLAGS = " --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-command-line-argument"


def source_hash() -> str:
    """SHA-256 over every HIP source, header and compiler flag, computed exactly as sdnq_amd/csrc/build.sh does: the library that
    is loaded is the one built from the sources in the tree iff this equals the contents of ``libsdnq_hip.so.srchash``."""
    import hashlib
    hdrs = sorted(f for f in os.listdir(_CSRC) if f.endswith(".h"))
    h = hashlib.sha256()
    for f in hdrs:
        h.update(open(os.path.join(_CSRC, f), "rb").read())
    h.update(open(os.path.join(_HERE, "..", "include", "sdnq_hip.h"), "rb").read())
    hdr_hash = h.hexdigest()
    flags = os.environ.get("SDNQ_EXTRA_FLAGS", "") + _FLAGS.replace("-ffp-contract=off", "-ffp-contract=" + os.environ.get("SDNQ_FP_CONTRACT", "off"))
    parts = ""
    for f in _SRCS:
        extra = "-mllvm -amdgpu-mfma-vgpr-form" if f == "attention" else ""
        if f == "rowquant" and os.environ.get("SDNQ_PRELOAD_ROWQUANT", "1") != "0":  # as build.sh
            extra = "-DSDNQ_PRELOAD_ROWQUANT -mllvm -amdgpu-kernarg-preload-count=14"
        if f == "gemm" and os.environ.get("SDNQ_PRELOAD_GEMM", "1") != "0":
            extra = "-DSDNQ_PRELOAD_GEMM -mllvm -amdgpu-kernarg-preload-count=14"
        if f in ("dequant", "conv", "gemm_aq", "gemm_ks", "gemm_w4"):
            extra = "-mllvm -amdgpu-kernarg-preload-count=14"
        g = hashlib.sha256((f"{hdr_hash} {flags} {extra}\n").encode())
        g.update(open(os.path.join(_CSRC, f + ".hip"), "rb").read())
        parts += f" {f}:{g.hexdigest()}"
    parts += " binding:" + hashlib.sha256(open(os.path.join(_CSRC, "binding.c"), "rb").read()).hexdigest()
    parts += " fastpath:" + hashlib.sha256(open(os.path.join(_CSRC, "fastpath.cpp"), "rb").read()
                                           + open(os.path.join(_HERE, "..", "include", "sdnq_hip.h"), "rb").read()).hexdigest()
    return hashlib.sha256((parts + "\n").encode()).hexdigest()


def lib_is_current() -> bool:
    try:
        host = [os.path.join(os.path.dirname(LIB_PATH), f) for f in ("_binding.so", "_fastpath.so")]  # built by the same script
        return os.path.exists(LIB_PATH) and all(os.path.exists(f) for f in host) and open(LIB_PATH + ".srchash").read().strip() == source_hash()
    except OSError:
        return False


def sources_newer_than_lib() -> bool:  # kept for callers of the old name
    return not lib_is_current()


def build(force: bool = False) -> str:
    """Compile every HIP source for gfx950 into sdnq_amd/libsdnq_hip.so (no GPU needed).  Content-addressed: nothing is rebuilt
    when the library's recorded source hash equals the hash of the tree; `force` recompiles every object."""
    if force or not lib_is_current():
        env = dict(os.environ, FORCE="1") if force else None
        subprocess.run(["bash", os.path.join(_CSRC, "build.sh"), LIB_PATH], check=True, env=env)
        if not lib_is_current():
            raise SdnqHipError("libsdnq_hip.so was built but its source hash does not match the tree (build.sh / _lib.source_hash out of sync)")
    return LIB_PATH


def _declare(lib):
    c = ctypes
    vp, i32, i64 = c.c_void_p, c.c_int, c.c_int64
    lib.sdnq_hip_version.restype = c.c_int
    lib.sdnq_hip_strerror.restype = c.c_char_p
    lib.sdnq_hip_strerror.argtypes = [c.c_int]
    lib.sdnq_hip_device_supported.argtypes = [c.c_int]
    lib.sdnq_hip_rowquant.argtypes = [vp, i32, i64, i64, i64, i32, i32, vp, vp, vp, vp, vp, i64, vp, vp]
    lib.sdnq_hip_scaled_mm.argtypes = [i32, vp, vp, vp, vp, vp, i32, i32, i64, vp, i32, i64, i64, i64, vp]
    lib.sdnq_hip_prefetch.argtypes = [vp, i64, i32, vp]
    lib.sdnq_hip_signal_alloc.argtypes = [i64, i32, c.POINTER(c.c_void_p), c.POINTER(c.c_int)]
    lib.sdnq_hip_signal_free.argtypes = [vp, i32]
    lib.sdnq_hip_ipc_export.argtypes = [vp, vp]
    lib.sdnq_hip_ipc_import.argtypes = [vp, c.POINTER(c.c_void_p)]
    lib.sdnq_hip_ipc_close.argtypes = [vp]
    lib.sdnq_hip_prefetch_hint.argtypes = [vp, i64, vp, i64, vp, i64, vp, i64]
    lib.sdnq_hip_unshard_columns.argtypes = [vp, vp, i32, i64, i64, i64, i64, i32, c.POINTER(c.c_int64), vp]
    lib.sdnq_hip_scaled_mm_lowrank_strided.argtypes = [i32, vp, i64, vp, vp, vp, vp, i32, vp, vp, vp, vp, i64, vp, i64, i32, i64, i64, i64, vp]
    pvp = c.POINTER(c.c_void_p)
    lib.sdnq_hip_push_post.argtypes = [pvp, i32, i32, c.c_uint64, c.c_uint64, vp]
    lib.sdnq_hip_push_columns.argtypes = [vp, i32, i64, i64, i64, pvp, pvp, pvp, i32, i32, c.c_uint64, i64, i64, i64, vp, vp, i32, vp]
    lib.sdnq_hip_dequant.argtypes = [c.POINTER(SdnqWeight), i32, vp, i32, vp]
    lib.sdnq_hip_requant.argtypes = [c.POINTER(SdnqWeight), i32, vp, vp, vp]
    lib.sdnq_hip_requant_ws.argtypes = [c.POINTER(SdnqWeight), i32, vp, vp, i32, vp]
    lib.sdnq_hip_linear.argtypes = [c.POINTER(SdnqLinearArgs), vp]
    lib.sdnq_hip_linear_workspace_bytes.argtypes = [c.POINTER(SdnqLinearArgs), c.POINTER(c.c_int64)]
    lib.sdnq_hip_requant_asym.argtypes = [c.POINTER(SdnqWeight), vp, vp, vp, vp]
    lib.sdnq_hip_unpack_mm.argtypes = [c.POINTER(SdnqWeight), i32, vp, vp]
    lib.sdnq_hip_hadamard.argtypes = [vp, i32, i64, i64, i64, i32, vp, i64, vp]
    lib.sdnq_hip_lowrank_down.argtypes = [vp, i32, i64, i64, i64, vp, i32, i32, vp, vp]
    lib.sdnq_hip_scaled_mm_lowrank.argtypes = [i32, vp, vp, vp, vp, vp, i32, vp, vp, i32, i32, vp, vp, vp, vp, vp, i32, i64, i64, i64, vp]
    lib.sdnq_hip_linear_float.argtypes = [vp, vp, vp, i32, vp, i64, i64, i64, i64, vp]
    lib.sdnq_hip_linear_float_strided.argtypes = [vp, vp, vp, i32, vp, i64, i64, i64, i64, i64, vp]
    lib.sdnq_hip_scaled_mm_strided.argtypes = [i32, vp, i64, vp, vp, vp, vp, i32, vp, i64, i32, i64, i64, i64, i64, vp]
    lib.sdnq_hip_linear_skinny.argtypes = [c.POINTER(SdnqWeight), i32, vp, vp, i32, vp, i64, i64, vp]
    lib.sdnq_hip_quantize_weight.argtypes = [vp, i32, i64, c.POINTER(SdnqWeight), c.c_float, c.c_float, vp]
    lib.sdnq_hip_linear_skinny_svd.argtypes = [c.POINTER(SdnqWeight), vp, vp, vp, i32, vp, i64, i64, vp]
    lib.sdnq_hip_linear_w8a8.argtypes = [i32, vp, i32, i64, i64, i64, i32, vp, vp, vp, vp, vp, i32, vp, i32, i64, vp]
    lib.sdnq_hip_stream_capture_id.argtypes = [vp, c.POINTER(c.c_uint64)]
    lib.sdnq_hip_linear_w8a8_fused.argtypes = [i32, vp, i32, i64, i64, i64, vp, vp, vp, i32, vp, i32, i64, vp]
    lib.sdnq_hip_linear_w8a8_fused_supported.argtypes = [i32, i32, i32, i64, i64, i64]
    lib.sdnq_hip_scaled_mm_multi.argtypes = [i32, vp, vp, vp, vp, vp, i32, vp, i32, i64, i32, i64, i64, i64, vp]
    lib.sdnq_hip_linear_w8a16_grouped.argtypes = [vp, i32, vp, i64, i64, i32, vp, i64, i64, i64, vp]
    lib.sdnq_hip_linear_w8a16.argtypes = [vp, i32, vp, vp, vp, vp, vp, i64, i64, i64, i64, vp]
    lib.sdnq_hip_rowquant_lp.argtypes = [vp, i32, i64, i64, i64, i32, i32, vp, vp, vp, vp, vp]
    lib.sdnq_hip_rowquant_lp_asym.argtypes = [vp, i32, i64, i64, i64, i32, vp, vp, vp, vp, vp, vp]
    lib.sdnq_hip_scaled_mm_lp_uzp.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, vp, i64, i64, i64, vp]
    lib.sdnq_hip_scaled_mm_lp_uzp_svd.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, vp, vp, i32, vp, i64, i64, i64, vp]
    lib.sdnq_hip_scaled_mm_lp.argtypes = [i32, vp, vp, vp, vp, vp, i32, i64, vp, vp, i32, vp, i64, i64, i64, vp]
    lib.sdnq_hip_scaled_mm_lp_zp.argtypes = [i32, vp, vp, vp, vp, vp, i32, i64, vp, vp, i32, vp, vp, vp, i64, i64, i64, vp]
    lib.sdnq_hip_set_tile_override.argtypes = [i32]
    lib.sdnq_hip_set_tile_override.restype = None
    lib.sdnq_hip_scaled_mm_grouped.argtypes = [i32, vp, vp, vp, i64, i64, i32, vp, i32, i64, i64, vp]
    lib.sdnq_hip_linear_float_multi.argtypes = [vp, vp, vp, i32, vp, i32, i64, i64, i64, i64, i64, vp]
    lib.sdnq_hip_scaled_mm_nchw.argtypes = [i32, vp, vp, vp, vp, vp, i32, vp, i32, i64, i64, i64, i64, vp]
    lib.sdnq_hip_im2col.argtypes = [vp, i32] + [i32] * 12 + [vp, vp]
    lib.sdnq_hip_im2col_rowquant.argtypes = [vp, i32] + [i32] * 12 + [i32, vp, vp, vp, vp]
    lib.sdnq_hip_im2col_rowquant_z.argtypes = [vp, i32] + [i32] * 12 + [i32, vp, vp, vp, vp]
    lib.sdnq_hip_attn_prepare.argtypes = [vp, vp, vp, i32] + [i64] * 6 + [i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.sdnq_hip_attn_fwd.argtypes = [vp, vp, vp, vp, vp, i32, c.c_float, i32, vp, i32, i64, i64, i64, vp, i32, vp] + [i64] * 6 + [vp]
    lib.sdnq_hip_attn_prepare_ex.argtypes = [vp, vp, vp, i32] + [i64] * 6 + [i32, i32, vp, vp, vp, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.sdnq_hip_attn_fwd_ex.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, i32, c.c_float, i32, vp, i32, i64, i64, i64, vp, i32, vp] + [i64] * 6 + [vp]
    lib.sdnq_hip_attn_fwd_q16.argtypes = [vp, vp, vp, vp, vp, i32, c.c_float, i32, vp, i32, i64, i64, i64, vp, i32, vp] + [i64] * 6 + [vp]
    lib.sdnq_hip_attn.argtypes = [vp, vp, vp, i32] + [i64] * 6 + [vp, vp, vp, i32, i32, c.c_float, i32, vp, i32, i64, i64, i64, vp, i32, vp, vp, i64, vp]
    lib.sdnq_hip_attn_workspace_bytes.argtypes = [i64] * 6 + [i32]
    lib.sdnq_hip_scaled_mm_tile.argtypes = [i32, i32, i32, i64, i64, i64, c.POINTER(c.c_int), c.POINTER(c.c_int), c.POINTER(c.c_int), c.POINTER(c.c_int64)]
    lib.sdnq_hip_lut4_build.argtypes = [c.POINTER(SdnqWeight), i32, vp, i32, vp, vp]
    lib.sdnq_hip_scaled_mm_w4.argtypes = [vp, vp, vp, vp, vp, vp, i32, vp, i32, i64, i64, i64, i64, vp]
    lib.sdnq_hip_scaled_mm_w4_supported.argtypes = [i32, i32, i64, i64, i64]
    lib.sdnq_hip_rowquant_f16.argtypes = [vp, i32, i64, i64, i64, vp, vp, vp]
    lib.sdnq_hip_scaled_mm_f16.argtypes = [vp, vp, vp, vp, vp, i32, i32, i64, vp, i32, i64, i64, i64, vp]
    for name in EXPORTS:
        if name not in ("sdnq_hip_strerror", "sdnq_hip_set_tile_override"):
            getattr(lib, name).restype = c.c_int
    lib.sdnq_hip_attn_workspace_bytes.restype = c.c_int64


def load():
    """Load the library (building it first if the sources are newer). Raises if unavailable."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                if not os.path.exists(LIB_PATH):
                    raise SdnqHipError(
                        f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                        "(hipcc --offload-arch=gfx950). There is no fallback path.")
                lib = ctypes.CDLL(LIB_PATH)
                _declare(lib)
                _lib = _with_typed_binding(lib)
    return _lib


USE_BINDING = os.environ.get("SDNQ_HIP_BINDING", "1").lower() not in {"0", "false", "no"}


class _BoundLib:
    """The ctypes handle with its hot entry points served by the typed CPython binding (csrc/binding.c: the same named symbols of
    the same library, ~0.5 us instead of ~10 us of argument conversion per call; an eager diffusion step makes ~900 calls).
    Every other entry point -- and every entry point when _binding.so is absent or SDNQ_HIP_BINDING=0 -- goes through ctypes."""

    def __init__(self, lib, binding):
        self._ctypes = lib
        for name in dir(binding):
            if name.startswith("sdnq_hip_"):
                setattr(self, name, getattr(binding, name))

    def __getattr__(self, name):  # not in the binding: the ctypes function
        return getattr(self._ctypes, name)


def _with_typed_binding(lib):
    if not USE_BINDING:
        return lib
    try:
        from . import _binding  # built by csrc/build.sh next to the library
    except ImportError:
        return lib  # binding only: every call still lands in libsdnq_hip.so, through ctypes
    _binding.init(LIB_PATH)
    return _BoundLib(lib, _binding)


USE_FAST_PLANS = os.environ.get("SDNQ_HIP_FAST_PLANS", "1").lower() not in {"0", "false", "no"}
_fastpath = None  # None: not looked for yet; False: unavailable / switched off


def fastpath():
    """The C++ fast path of the eager Linear forward (csrc/fastpath.cpp -> sdnq_amd/_fastpath.so), or None with SDNQ_HIP_FAST_PLANS=0 /
    when the module or the kernel library is not built: host logic only -- every launch it makes is a named entry point of
    libsdnq_hip.so, and sdnq_amd/linear.py is the complete forward without it."""
    global _fastpath
    if _fastpath is None:
        with _lock:
            if _fastpath is None:
                mod = False
                if USE_FAST_PLANS and os.path.exists(LIB_PATH):
                    try:
                        import torch  # noqa: F401  (the module links against torch's libraries: they must be loaded first)
                        from . import _fastpath as mod
                        mod.init(LIB_PATH)
                    except (ImportError, OSError) as e:
                        import warnings
                        warnings.warn(f"sdnq_amd._fastpath is unavailable ({e}); eager Linear calls take the Python forward")
                        mod = False
                _fastpath = mod
    return _fastpath or None


def check(status: int, what: str = ""):
    if status != 0:
        lib = load()
        try:  # a call that failed in front of its GEMM launch leaves its weight-prefetch hint pending: drop it
            lib.sdnq_hip_prefetch_hint(None, 0, None, 0, None, 0, None, 0)
        except Exception:  # noqa: BLE001
            pass
        msg = lib.sdnq_hip_strerror(status).decode()
        raise SdnqHipError(f"sdnq_hip {what} failed: {msg} (status {status})")
