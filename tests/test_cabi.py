"""The C-ABI shared library loads (no GPU needed) and exports every symbol include/sdnq_hip.h declares."""
import ctypes
import os
import re

from sdnq_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "sdnq_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(sdnq_hip_[a-z0-9_]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol():
    _lib.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    syms = declared_symbols()
    assert len(syms) >= 13
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/sdnq_hip.h but not exported"
    assert set(syms) == set(_lib.EXPORTS), (set(syms) ^ set(_lib.EXPORTS))


def test_version_and_strerror():
    lib = _lib.load()
    assert lib.sdnq_hip_version() == 1
    assert lib.sdnq_hip_strerror(0) == b"ok"
    for code in range(-8, 0):
        msg = lib.sdnq_hip_strerror(code)
        assert msg and msg != b"unknown status"
    assert lib.sdnq_hip_strerror(-99) == b"unknown status"


def test_argument_validation_without_gpu():
    """Validation runs before any launch, so error codes are observable on a CPU-only box."""
    lib = _lib.load()
    assert lib.sdnq_hip_rowquant(None, 1, 4, 64, 64, 0, 0, None, None, None, None, None, 0, None, None) == -1          # NULL
    buf = ctypes.create_string_buffer(4096)
    p = ctypes.addressof(buf)
    p += (-p) % 16
    assert lib.sdnq_hip_rowquant(p, 7, 4, 64, 64, 0, 0, p, p, None, None, None, 0, None, None) == -2                  # dtype
    assert lib.sdnq_hip_rowquant(p, 1, 4, 60, 60, 0, 0, p, p, None, None, None, 0, None, None) == -3                  # K % 8
    assert lib.sdnq_hip_rowquant(p + 2, 1, 4, 64, 64, 0, 0, p, p, None, None, None, 0, None, None) == -4              # alignment
    assert lib.sdnq_hip_rowquant(p, 1, 4, 64, 64, 0, 48, p, p, None, None, None, 0, None, None) == -3                 # Hadamard group not pow2
    assert lib.sdnq_hip_scaled_mm(0, p, p, p, p, None, 0, 0, 0, p, 1, 32, 32, 24, None) == -3          # K % 16
    assert lib.sdnq_hip_scaled_mm(5, p, p, p, p, None, 0, 0, 0, p, 1, 32, 32, 32, None) == -2          # mm dtype
    w = _lib.SdnqWeight(weight=p, scale=p, zero_point=None, svd_up=None, svd_down=None, n=16, k=64, group_size=48, svd_rank=0,
                        svd_dtype=0, storage=0, kind=0, bits=4, exponent=0, mantissa=0, native_float=0)
    assert lib.sdnq_hip_dequant(ctypes.byref(w), 0, p, 1, None) == -3                                   # K % group
    w.group_size = 64
    w.kind = 1
    assert lib.sdnq_hip_dequant(ctypes.byref(w), 0, p, 1, None) == -1                                   # uint without zero_point
    w.kind, w.bits, w.storage = 0, 9, 0
    assert lib.sdnq_hip_dequant(ctypes.byref(w), 0, p, 1, None) == -2                                   # 9 bits in uint8 words
    w.bits, w.scale_dtype = 4, 5
    assert lib.sdnq_hip_dequant(ctypes.byref(w), 0, p, 1, None) == -2                                   # unknown scale dtype
    # model-dtype-scale entry points (dequantize_fp32=False)
    assert lib.sdnq_hip_rowquant_lp(p, 0, 4, 64, 64, 0, 0, p, p, None, None, None) == -2               # float32 activations: sdnq_hip_rowquant
    assert lib.sdnq_hip_rowquant_lp(None, 1, 4, 64, 64, 0, 0, p, p, None, None, None) == -1
    assert lib.sdnq_hip_rowquant_lp(p, 1, 4, 64, 64, 1, 0, p, p, p, None, None) == -5                  # rowsum with fp8 codes
    assert lib.sdnq_hip_scaled_mm_lp(0, p, p, p, p, None, 1, 0, None, None, 0, p, 32, 32, 32, None) == -1   # bias_ndim without a bias
    assert lib.sdnq_hip_scaled_mm_lp(0, p, p, p, p, None, 0, 0, p, None, 32, p, 32, 32, 32, None) == -1     # t without svd_up
    assert lib.sdnq_hip_scaled_mm_lp(0, p, p, p, p, None, 0, 0, None, None, 0, p, 32, 32, 24, None) == -3   # K % 16
    assert lib.sdnq_hip_lowrank_down(p, 1, 4, 64, 64, p, 2, 32, p, None) == -2                         # activation / factor dtype mismatch
    assert lib.sdnq_hip_scaled_mm_lp_zp(0, p, p, p, p, None, 0, 0, None, None, 0, p, None, p, 32, 32, 32, None) == -1   # rowsum without zero point
    assert lib.sdnq_hip_scaled_mm_lp_zp(0, p, p, p, p, p, 2, 32, None, None, 0, p, p, p, 32, 32, 32, None) == -3        # zero point with a 2-D bias
    # matmuls on views (grouped convs)
    smm = lib.sdnq_hip_scaled_mm_strided
    assert smm(0, p, 16, p, p, p, None, 0, p, 32, 1, 32, 32, 32, 0, None) == -3                    # lda < K
    assert smm(0, p, 72, p, p, p, None, 0, p, 32, 1, 32, 32, 32, 0, None) == -3                    # lda % 16
    assert smm(0, p, 64, p, p, p, None, 0, p, 16, 1, 32, 32, 32, 0, None) == -3                    # ldc < N
    assert smm(0, p, 64, p, p, p, None, 0, p, 32, 1, 32, 32, 32, 12, None) == -3                   # hw % 8 / M % hw
    assert smm(0, p, 64, p, p, p, None, 0, p, 32, 0, 32, 32, 32, 16, None) == -5                   # float32 output with the NCHW store
    lfs = lib.sdnq_hip_linear_float_strided
    assert lfs(None, p, None, 1, p, 40, 32, 32, 32, 32, None) == -1
    assert lfs(p, p, None, 1, p, 40, 32, 32, 32, 16, None) == -3                                   # ldc < N
    assert lfs(p, p, None, 1, p, 40, 32, 32, 16, 32, None) == -3                                   # ldx < K


def test_prefetch_argument_validation_without_gpu():
    lib = _lib.load()
    buf = ctypes.create_string_buffer(4096)
    p = ctypes.addressof(buf)
    assert lib.sdnq_hip_prefetch(None, 128, 0, None) == -1                       # NULL
    assert lib.sdnq_hip_prefetch(p, 0, 0, None) == 0                             # nothing to fetch: no launch
    assert lib.sdnq_hip_prefetch_hint(p, -1, None, 0, None, 0, None, 0) == -3    # negative size
    assert lib.sdnq_hip_prefetch_hint(p, 256, None, 0, p, 128, None, 0) == 0
    assert lib.sdnq_hip_prefetch_hint(None, 0, None, 0, None, 0, None, 0) == 0   # cleared again (nothing may stay pending for a later launch)


def test_attention_argument_validation_without_gpu():
    """The attention entry points validate before launching as well."""
    lib = _lib.load()
    buf = ctypes.create_string_buffer(8192)
    p = ctypes.addressof(buf)
    p += (-p) % 16
    prep = lib.sdnq_hip_attn_prepare
    ok_tail = (None, None, None, p, p, p, p, p, p, None)  # strides (contiguous), qq, qs, kq, ks, vt, kmean, stream
    assert prep(None, p, p, 1, 1, 2, 2, 8, 8, 64, 1, 0, *ok_tail) == -1                       # NULL query
    assert prep(p, p, p, 0, 1, 2, 2, 8, 8, 64, 1, 0, *ok_tail) == -5                          # float32 inputs: not built
    assert prep(p, p, p, 1, 1, 3, 2, 8, 8, 64, 1, 0, *ok_tail) == -3                          # q_heads % kv_heads
    assert prep(p, p, p, 1, 1, 2, 2, 8, 8, 136, 1, 0, *ok_tail) == -5                         # head_dim > 128
    assert prep(p, p, p, 1, 1, 2, 2, 8, 8, 60, 1, 0, *ok_tail) == -5                          # head_dim % 8
    assert prep(p, p, p, 1, 1, 2, 2, 8, 8, 64, 1, 48, *ok_tail) == -3                         # Hadamard group not a power of two
    assert prep(p + 2, p, p, 1, 1, 2, 2, 8, 8, 64, 1, 0, *ok_tail) == -4                      # alignment
    bad = (ctypes.c_int64 * 3)(1024, 516, 68)                                                 # token stride not a multiple of 8
    assert prep(p, p, p, 1, 1, 2, 2, 8, 8, 64, 1, 0, bad, None, None, p, p, p, p, p, p, None) == -4
    fwd = lib.sdnq_hip_attn_fwd
    assert fwd(p, p, p, p, p, 1, 0.125, 0, None, 0, 0, 0, 0, None, 1, None, 1, 2, 2, 8, 8, 64, None) == -1    # NULL out
    assert fwd(p, p, p, p, p, 1, 0.125, 0, p, 5, 0, 0, 0, p, 1, None, 1, 2, 2, 8, 8, 64, None) == -2          # mask dtype
    assert fwd(p, p, p, p, p, 1, 0.125, 0, None, 0, 0, 0, 0, p, 1, None, 1, 2, 2, 8, 0, 64, None) == -3       # kv_len 0
    assert fwd(p, p, p, p, p, 0, 0.125, 0, None, 0, 0, 0, 0, p, 1, None, 1, 2, 2, 8, 8, 64, None) == -2       # f32 value operand
    assert fwd(None, p, p, p, p, 1, 0.125, 0, None, 0, 0, 0, 0, p, 1, None, 1, 2, 2, 8, 8, 64, None) == -1    # no quantized Q
    # K / V only (qq and qs both NULL, q not read) pairs with sdnq_hip_attn_fwd_q16; one of the two alone is an error
    assert prep(p, p, p, 1, 1, 2, 2, 8, 8, 64, 1, 0, None, None, None, p, None, p, p, p, p, None) == -1
    assert prep(None, p, None, 1, 1, 2, 2, 8, 8, 64, 1, 0, None, None, None, None, None, p, p, p, p, None) == -1   # NULL value
    fq = lib.sdnq_hip_attn_fwd_q16
    assert fq(None, None, p, p, p, 1, 0.125, 0, None, 0, 0, 0, 0, p, 1, None, 1, 2, 2, 8, 8, 64, None) == -1  # NULL query
    assert fq(p + 8, None, p, p, p, 1, 0.125, 0, None, 0, 0, 0, 0, p, 1, None, 1, 2, 2, 8, 8, 64, None) == -4  # query rows not 16-byte aligned
    assert fq(p, bad, p, p, p, 1, 0.125, 0, None, 0, 0, 0, 0, p, 1, None, 1, 2, 2, 8, 8, 64, None) == -4       # token stride
    assert fq(p, None, p, p, p, 1, 0.125, 0, None, 0, 0, 0, 0, p, 1, None, 1, 3, 2, 8, 8, 64, None) == -3      # q_heads % kv_heads
    # the one-call form: no workspace up to 128 keys without a rotation, else the prepare pass's operands
    wsb, att = lib.sdnq_hip_attn_workspace_bytes, lib.sdnq_hip_attn
    assert wsb(1, 2, 2, 4096, 77, 64, 0) == 0 and wsb(1, 2, 2, 4096, 128, 128, 0) == 0
    assert wsb(1, 2, 2, 4096, 77, 64, 64) > 2 * 4096 * 64            # a rotation keeps the separate pass (with Q)
    assert wsb(1, 2, 1, 64, 129, 64, 0) == 160 * 64 + 768 + 160 * 64 * 2 + 32 * 64 * 4   # kq + ks (640 -> 768) + vt + channel sums, one kv head
    assert wsb(1, 3, 2, 64, 129, 64, 0) == -3 and wsb(1, 2, 2, 64, 0, 64, 0) == -3
    tail = (None, None, None, 1, 0, 0.125, 0, None, 0, 0, 0, 0)      # strides, smooth_k, hadamard_group, sm_scale, causal, mask
    assert att(p, p, None, 1, 1, 2, 2, 8, 8, 64, *tail, p, 1, None, None, 0, None) == -1       # NULL value
    assert att(p, p, p, 1, 1, 2, 2, 8, 8, 64, *tail, None, 1, None, None, 0, None) == -1       # NULL out
    assert att(p, p, p, 0, 1, 2, 2, 8, 8, 64, *tail, p, 1, None, None, 0, None) == -5          # float32 inputs
    assert att(p, p, p, 1, 1, 2, 2, 8, 200, 64, *tail, p, 1, None, None, 0, None) == -1        # > 128 keys need the workspace
    assert att(p, p, p, 1, 1, 2, 2, 8, 200, 64, *tail, p, 1, None, p, 100, None) == -3         # ... of the advertised size
    assert att(p, p + 8, p, 1, 1, 2, 2, 8, 8, 64, *tail, p, 1, None, None, 0, None) == -4      # key rows not 16-byte aligned


def test_bf16_uint8_matmul_argument_validation_without_gpu():
    """sdnq_hip_rowquant_lp_asym / sdnq_hip_scaled_mm_lp_uzp (the uint8 matmul on bfloat16 scales) reject bad arguments before any launch."""
    import ctypes
    from sdnq_amd import _lib
    lib = _lib.load()
    buf = ctypes.create_string_buffer(8192)
    p = ctypes.addressof(buf)
    p += (-p) % 16
    rq = lib.sdnq_hip_rowquant_lp_asym
    assert rq(p, 1, 4, 64, 64, 0, p, p, None, None, None, None) == -1     # no zero-point output
    assert rq(None, 1, 4, 64, 64, 0, p, p, p, None, None, None) == -1     # NULL input
    assert rq(p, 0, 4, 64, 64, 0, p, p, p, None, None, None) == -2        # float32 rows: sdnq_hip_rowquant
    assert rq(p, 1, 4, 60, 64, 0, p, p, p, None, None, None) == -3        # K % 8
    assert rq(p, 1, 4, 64, 64, 48, p, p, p, None, None, None) == -3       # Hadamard group not a power of two
    mm = lib.sdnq_hip_scaled_mm_lp_uzp
    assert mm(p, p, p, p, None, None, None, None, p, 0, p, 4, 16, 64, None) == -1   # no activation zero points
    assert mm(p, p, p, p, None, p, None, p, p, 0, p, 4, 16, 64, None) == -1         # rowsum without a weight zero point
    assert mm(p, p, p, p, None, None, None, p, p, 0, None, 4, 16, 64, None) == -1   # NULL out
    assert mm(p, p, p, p, None, None, None, p, p, 0, p, 4, 16, 60, None) == -3      # K % 16


def test_linear_args_struct_matches_the_header():
    """The ctypes mirror of SdnqLinearArgs has the header's field order and the size the library checks (struct_size)."""
    import ctypes
    import re
    from sdnq_amd import _lib
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "sdnq_hip.h")).read()
    body = hdr[hdr.index("typedef struct SdnqLinearArgs {"):hdr.index("} SdnqLinearArgs;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for decl in body.split("{", 1)[1].split(";"):
        decl = decl.strip()
        if not decl:
            continue
        parts = decl.replace("*", " ").split()
        for nm in decl.split(",") if "," in decl else [decl]:
            names.append(nm.replace("*", " ").split()[-1])
        del parts
    assert names == [f[0] for f in _lib.SdnqLinearArgs._fields_], names
    assert ctypes.sizeof(_lib.SdnqLinearArgs) == 10 * 4 + 4 * 8 + 16 * 8


def test_one_launch_linear_predicate_and_argument_validation_without_gpu():
    """sdnq_hip_linear_w8a8_fused_supported is host logic (no launch): where the one-launch w8a8 Linear is built AND offered -- 16-bit
    activations, K % 128 == 0, K <= 1280, one round of 64 x 128 tiles, at most 12 column tiles per row block, int8 (fp8 behind its
    switch) -- and the entry point's argument checks return before anything is launched."""
    from sdnq_amd import _lib
    lib = _lib.load()
    sup = lib.sdnq_hip_linear_w8a8_fused_supported
    I8, FP8, F32, BF16, F16 = 0, 1, 0, 1, 2
    assert sup(I8, BF16, BF16, 1024, 1280, 1280) == 1 and sup(I8, F16, F16, 1024, 1280, 640) == 1
    assert sup(I8, BF16, BF16, 1024, 1280, 5120) == 0      # the rows do not fit LDS
    assert sup(I8, BF16, BF16, 1024, 1280, 1200) == 0      # K % 128
    assert sup(I8, BF16, BF16, 1024, 10240, 1280) == 0     # 80 column tiles per row block
    assert sup(I8, BF16, BF16, 2048, 1280, 1280) == 0      # 320 tiles: two rounds
    assert sup(I8, BF16, BF16, 16, 1280, 1280) == 0        # the M < 32 branch is not a matmul at all
    assert sup(I8, F32, F32, 1024, 1280, 1280) == 0 and sup(I8, BF16, F16, 1024, 1280, 1280) == 0
    assert sup(FP8, BF16, BF16, 1024, 1280, 1280) == 0     # built and bit-identical, off by default (SDNQ_HIP_FUSED_ROWQUANT_FP8)
    f = lib.sdnq_hip_linear_w8a8_fused
    p = 4096  # any non-null, 16-byte aligned address: every call below returns before a launch
    assert f(I8, None, BF16, 64, 128, 128, p, p, None, 0, p, BF16, 128, None) == -1      # NULL activation
    assert f(2, p, BF16, 64, 128, 128, p, p, None, 0, p, BF16, 128, None) == -2          # unknown matmul dtype
    assert f(I8, p, BF16, 64, 128, 64, p, p, None, 0, p, BF16, 128, None) == -3          # ldx < K
    assert f(I8, p, BF16, 64, 128, 128, p, p, None, 0, p, BF16, 132, None) == -3         # N % 8
    assert f(I8, p, BF16, 64, 192, 192, p, p, None, 0, p, BF16, 128, None) == -5         # K % 128: valid in the reference, not built here
    assert f(I8, p, BF16, 64, 2560, 2560, p, p, None, 0, p, BF16, 128, None) == -5       # K > 1280
    assert f(I8, p, F32, 64, 128, 128, p, p, None, 0, p, F32, 128, None) == -5           # float32 activations
    assert f(I8, p + 8, BF16, 64, 128, 128, p, p, None, 0, p, BF16, 128, None) == -4     # alignment


def test_scaled_mm_tile_is_a_dry_run_of_the_shape_rules():
    """sdnq_hip_scaled_mm_tile answers without a device: the tile, its threads and the workgroup count the launcher would use
    (the SDXL bs = 1 shapes of launch_tiles' comments), and refuses what sdnq_hip_scaled_mm refuses."""
    import ctypes
    lib = _lib.load()

    def tile(m, n, k, mm=0):
        bm, bn, thr, wgs = ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int64()
        st = lib.sdnq_hip_scaled_mm_tile(mm, 1, 1, m, n, k, ctypes.byref(bm), ctypes.byref(bn), ctypes.byref(thr), ctypes.byref(wgs))
        return st, bm.value, bn.value, thr.value, wgs.value
    assert tile(1024, 1280, 1280) == (0, 64, 128, 512, 160)      # 160 of 256 CUs: the launch the round-6 K-split tile was built against
    assert tile(1024, 10240, 1280) == (0, 256, 160, 512, 256)    # GEGLU: exactly one workgroup per CU
    assert tile(1024, 3840, 1280) == (0, 128, 128, 512, 240)
    assert tile(4608, 3072, 3072)[1:3] == (256, 256)             # FLUX: the half-tile ring
    assert tile(77, 640, 2048)[1:3] == (64, 64)
    assert tile(1024, 1280, 1280, mm=1)[0] == 0                  # fp8
    assert tile(1024, 1284, 1280)[0] != 0 and tile(0, 1280, 1280)[0] != 0 and tile(64, 64, 64, mm=7)[0] != 0
    assert lib.sdnq_hip_scaled_mm_tile(0, 1, 0, 1024, 1280, 1280, None, None, None, None) == 0  # every output pointer is optional


def test_every_export_has_declared_argument_types():
    """ctypes converts an undeclared Python int to a 32-bit C int: a device pointer or an int64 size would be truncated silently (round 6:
    a new entry point without argtypes sent truncated pointers to the GPU).  Every export must carry argtypes."""
    lib = _lib.load()
    raw = getattr(lib, "_ctypes", lib)
    missing = [name for name in _lib.EXPORTS if name not in ("sdnq_hip_version",) and getattr(raw, name).argtypes is None]
    assert missing == [], missing
