"""Load-time codecs (sdnq_amd/packed.py) vs known-answer vectors from the reference's codecs."""
import json
import os

import numpy as np
import pytest
import torch

from sdnq_amd import packed
from sdnq_amd.common import dtype_dict
from tests.golden_util import GOLD

Z = np.load(os.path.join(GOLD, "codecs.npz"))
META = json.load(open(os.path.join(GOLD, "codecs.json")))


@pytest.mark.parametrize("bits", list(range(1, 8)) + list(range(9, 16)))
def test_int_pack_unpack_known_answers(bits):
    vals = torch.from_numpy(Z[f"uint{bits}.values"])
    ref = Z[f"uint{bits}.packed"]
    mine = packed.pack_uint(vals, bits)
    refw = ref.astype(np.uint8) if ref.dtype == np.int64 else ref
    assert np.array_equal(mine.numpy().reshape(-1).astype(np.int64) & 0xffff, refw.reshape(-1).astype(np.int64) & 0xffff), bits
    back = packed.unpack_uint(torch.from_numpy(refw.copy()), bits, vals.shape)
    assert torch.equal(back, vals), bits


def test_signed_offset_roundtrip():
    for name in ("int2", "int4", "int6", "int7", "int9", "int12"):
        ent = dtype_dict[name]
        v = torch.randint(ent["min"], ent["max"] + 1, (480,), dtype=torch.int32)
        assert torch.equal(packed.unpack_int(packed.pack_int(v, name), name, v.shape, dtype=torch.int32), v), name


@pytest.mark.parametrize("name", [k for k in META if k.startswith("float")])
def test_float_encode_decode_known_answers(name):
    ent = dtype_dict[name]
    codes, dec = torch.from_numpy(Z[f"{name}.codes"]), Z[f"{name}.decoded"]
    mine = packed.decode_float(codes, name).numpy()
    assert np.array_equal(mine.view(np.uint32), dec.view(np.uint32)), name
    sweep = torch.from_numpy(Z[f"{name}.sweep_in"])
    refp = Z[f"{name}.sweep_packed"]
    if refp.dtype == np.int64:
        refp = refp.astype(np.uint8)
    minep = packed.pack_float(sweep, name).numpy()
    assert np.array_equal(minep.reshape(-1).astype(np.int64) & 0xffff, refp.reshape(-1).astype(np.int64) & 0xffff), name
    back = packed.unpack_float(torch.from_numpy(refp.copy()), name, (480,)).numpy()
    assert np.array_equal(back, Z[f"{name}.sweep_decoded"]), name
