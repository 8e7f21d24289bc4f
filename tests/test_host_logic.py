"""Host-side logic that needs no GPU: workload shape lists and the activation-quantization cache."""
import collections

import torch

from sdnq_amd import shapes as S
from sdnq_amd.linear import _ActivationCache


def _agg(seq):
    return dict(collections.Counter((e[0],) + tuple(e[1:5]) for e in seq))


def test_sdxl_sequence_matches_aggregate():
    seq = S.sdxl_unet_layer_sequence()
    assert len(seq) == 741
    assert S.ops_of_sequence(seq) == S.ops_of(S.sdxl_unet_linears()) == 4355400515520
    # 3-way shared self-attention input per layer, one text tensor for all 140 cross-attention k/v projections
    by_key = collections.Counter(e[5] for e in seq)
    assert by_key["text"] == 140
    assert sum(1 for k, c in by_key.items() if k.endswith(".h1") and c == 3) == 70
    for e in seq:  # every consumer of one tensor sees the same (M, K)
        assert {(x[1], x[2]) for x in seq if x[5] == e[5]} == {(e[1], e[2])}


def test_flux_sequence_matches_aggregate():
    seq = S.flux_dev_layer_sequence()
    want = {(e[0],) + tuple(e[1:5]): e[5] for e in S.flux_dev_linears()}
    assert _agg(seq) == want
    assert S.ops_of_sequence(seq) == S.ops_of(S.flux_dev_linears())


def test_activation_cache_identity_version_and_lru():
    c = _ActivationCache(2)
    a, b, d = torch.zeros(4, 8), torch.zeros(4, 8), torch.zeros(4, 8)
    p = (0, 0, False, False, False)
    assert c.get(a, p) is None
    c.put(a, p, "qa")
    assert c.get(a, p) == "qa"
    assert c.get(b, p) is None  # equal values, different tensor object
    assert c.get(a, (1,) + p[1:]) is None  # different quantization parameters
    a.add_(1)  # in-place update bumps _version -> stale
    assert c.get(a, p) is None
    c.put(a, p, "qa2"), c.put(b, p, "qb")
    assert c.get(a, p) == "qa2"  # refreshes a
    c.put(d, p, "qd")  # evicts b (least recently used)
    assert c.get(b, p) is None and c.get(a, p) == "qa2" and c.get(d, p) == "qd"
    c.clear()
    assert c.get(a, p) is None
