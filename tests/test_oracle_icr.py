"""Pin the TPS-ResNet-BiLSTM-Attn CPU oracle against goldens from the reference's own Model(opt)."""
import os

import numpy as np
import pytest

from marie_icr_amd.weights import CRNN_CHARSET, make_icr_state, state_checksum
from oracle import crnn_numpy
from oracle.icr_torch import TorchIcrOracle, attn_decode

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("tag", ["a", "b"])
def test_icr_oracle_matches_reference(tag):
    g = np.load(os.path.join(GOLD, f"icr_attn_{tag}.npz"))
    st = make_icr_state(int(g["weight_seed"]))
    assert state_checksum(st) == str(g["weight_sha256"])
    o = TorchIcrOracle(st)
    logits, stages = o.logits(crnn_numpy.normalize_u8(g["crops_u8"]), want_stages=True)
    np.testing.assert_allclose(stages["rectified"], g["rectified"], atol=1e-5, rtol=0)
    np.testing.assert_allclose(logits, g["logits"], atol=5e-4, rtol=0)
    idx, texts, conf = attn_decode(logits, CRNN_CHARSET)
    np.testing.assert_array_equal(idx, g["argmax"])
    assert texts == [str(s) for s in g["strings"]]
    np.testing.assert_allclose(conf, g["confidence"], rtol=2e-3, atol=1e-30)


def test_attn_decode_rule():
    # tokens: 2.. are characters; 1 = [s]; 0 = [GO]
    steps, nc = 6, 96
    logits = np.full((3, steps, nc), -8.0, np.float32)
    seqs = [[12, 13, 1, 14, 14, 14],      # "ab" then end
            [1, 12, 12, 12, 12, 12],      # immediate end -> ("", 0)
            [12, 0, 13, 1, 5, 5]]         # a [GO] b [s]: the cut position counts the 4 characters of "[GO]"
    for n, s in enumerate(seqs):
        for i, t in enumerate(s):
            logits[n, i, t] = 8.0
    idx, texts, conf = attn_decode(logits, CRNN_CHARSET)
    assert texts[0] == "AB" and texts[1] == "" and conf[1] == 0.0
    assert texts[2] == "A[GO]B"
    assert 0 < conf[0] < 1
