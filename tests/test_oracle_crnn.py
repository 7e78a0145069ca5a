"""Pin the CPU oracle (numpy + torch restatements) against the golden vectors the
reference's own ``Model(opt)`` produced (``oracle/gen_golden.py``)."""
import os

import numpy as np
import pytest

from marie_icr_amd.weights import CRNN_CHARSET, make_crnn_state, state_checksum
from oracle import crnn_numpy
from oracle.crnn_torch import TorchCrnnOracle, default_init_state

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _load(tag):
    g = np.load(os.path.join(GOLD, f"crnn_{tag}.npz"))
    if tag.startswith("default"):
        st = default_init_state(int(g["weight_seed"]))
    else:
        st = make_crnn_state(int(g["weight_seed"]))
    assert state_checksum(st) == str(g["weight_sha256"]), "weight set drifted from the fixture"
    return g, st


@pytest.mark.parametrize("tag", ["scaled_w256", "scaled_w100", "default_w256"])
def test_numpy_oracle_matches_reference(tag):
    g, st = _load(tag)
    logits, idx, texts, conf = crnn_numpy.recognize_crops_u8(g["crops_u8"], st, CRNN_CHARSET)
    assert logits.shape == g["logits"].shape
    # fp32 restatement vs fp32 reference: summation order differs only
    np.testing.assert_allclose(logits, g["logits"], atol=2e-4, rtol=0)
    np.testing.assert_array_equal(idx, g["argmax"])
    assert texts == [str(s) for s in g["strings"]]
    np.testing.assert_allclose(conf, g["confidence"], rtol=2e-3, atol=1e-30)


@pytest.mark.parametrize("tag", ["scaled_w256", "scaled_w100", "default_w256"])
def test_torch_oracle_matches_reference(tag):
    g, st = _load(tag)
    o = TorchCrnnOracle(st)
    logits = o.logits(crnn_numpy.normalize_u8(g["crops_u8"]))
    np.testing.assert_allclose(logits, g["logits"], atol=2e-4, rtol=0)
    idx, texts, conf = o.decode(logits, CRNN_CHARSET)
    np.testing.assert_array_equal(idx, g["argmax"])
    assert texts == [str(s) for s in g["strings"]]
    np.testing.assert_allclose(conf, g["confidence"], rtol=2e-3, atol=1e-30)


def test_ctc_collapse_rule():
    # blank=0 dropped, repeats merged, repeats separated by blank kept, upper-cased
    t = 6
    logits = np.full((1, t, 95), -5.0, np.float32)
    seq = [11, 11, 0, 11, 12, 12]  # 'a','a',blank,'a','b','b' -> "AAB"
    for i, s in enumerate(seq):
        logits[0, i, s] = 5.0
    idx, texts, conf = crnn_numpy.ctc_greedy(logits, CRNN_CHARSET)
    assert texts == ["AAB"]
    assert idx[0].tolist() == seq
    assert 0 < conf[0] < 1
