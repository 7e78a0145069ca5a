"""The geometry oracle against vectors produced by the reference's own overlap.py / line_processor.py
(oracle/gen_golden.py --geometry-only)."""
import os

import numpy as np
import pytest

from oracle import geometry_ref as gr

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "geometry.npz"))
CASES = list(range(int(G["n_cases"])))
# cases where numpy's default (unstable) argsort ordered equal-y boxes differently from input order on the machine that
# generated the goldens; everywhere else the two reference runs agree
TIE_CASES = [k for k in CASES if not np.array_equal(G[f"lines_{k}"], G[f"lines_platform_{k}"])]


def _xywh(b):
    bi = b.astype(np.int32)
    return np.stack([bi[:, 0], bi[:, 1], bi[:, 2] - bi[:, 0], bi[:, 3] - bi[:, 1]], 1)


@pytest.mark.parametrize("k", CASES)
def test_merge_boxes(k):
    got = gr.merge_boxes(G[f"boxes_{k}"])
    np.testing.assert_array_equal(got, G[f"merged_{k}"])


@pytest.mark.parametrize("k", CASES)
def test_line_merge_and_numbers(k):
    xywh = _xywh(G[f"boxes_{k}"])
    lines = gr.line_merge(xywh)
    np.testing.assert_array_equal(lines, G[f"lines_{k}"])          # reference code, equal-y ties kept in input order
    if k not in TIE_CASES:                                          # reference code, numpy's platform tie order
        np.testing.assert_array_equal(lines, G[f"lines_platform_{k}"])
    nums = [gr.find_line_number(lines, r) for r in xywh]
    np.testing.assert_array_equal(nums, G[f"linenum_{k}"])


def test_blocks():
    for k in CASES:
        xywh = _xywh(G[f"boxes_{k}"])
        blocks = [gr.merge_bboxes_as_block(xywh[: 1 + (j % len(xywh))]) for j in range(0, len(xywh), 7)]
        np.testing.assert_array_equal(np.asarray(blocks), G[f"blocks_{k}"])
