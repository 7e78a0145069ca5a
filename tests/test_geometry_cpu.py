"""Native geometry (csrc/geometry.hip through the C ABI; host code, runs without a GPU) against the reference-made
goldens and, for lines_from_bboxes, against the oracle's dense raster."""
import os

import numpy as np
import pytest

from marie_icr_amd import geometry as geo
from oracle import geometry_ref as gr

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "geometry.npz"))
CASES = list(range(int(G["n_cases"])))


def _xywh(b):
    bi = b.astype(np.int32)
    return np.stack([bi[:, 0], bi[:, 1], bi[:, 2] - bi[:, 0], bi[:, 3] - bi[:, 1]], 1)


@pytest.mark.parametrize("k", CASES)
def test_merge_boxes_golden(k):
    got = np.asarray(geo.merge_boxes([r for r in G[f"boxes_{k}"]], 0.08), np.float32)
    np.testing.assert_array_equal(got, G[f"merged_{k}"])


@pytest.mark.parametrize("k", CASES)
def test_line_merge_golden(k):
    xywh = _xywh(G[f"boxes_{k}"])
    lines = geo.line_merge(np.zeros((4, 4), np.uint8), xywh.tolist())
    np.testing.assert_array_equal(lines, G[f"lines_{k}"])
    np.testing.assert_array_equal(geo.find_line_numbers(lines, xywh), G[f"linenum_{k}"])


def test_empty_and_degenerate():
    assert geo.merge_boxes([]) == []
    assert geo.line_merge(np.zeros((2, 2)), []) == []
    assert geo.find_line_numbers([], [[1, 2, 3, 4]]) == [-1]
    assert len(geo.lines_from_bboxes(np.zeros((100, 100)), np.zeros((0, 4)))) == 0
    # a zero-height line never counts as an overlap; the nearest bottom edge wins instead
    assert geo.find_line_numbers([[0, 10, 50, 0], [0, 100, 50, 20]], [[5, 8, 10, 6]]) == [1]


@pytest.mark.parametrize("k", CASES)
def test_lines_from_bboxes_vs_dense_raster(k):
    b = G[f"merged_{k}"]
    for (h, w) in ((3300, 2550), (1700, 2400)):
        got = geo.lines_from_bboxes(np.zeros((h, w), np.uint8), b)
        np.testing.assert_array_equal(got, gr.lines_from_bboxes(b, h, w))


def test_lines_from_bboxes_borders_and_even_kernel():
    rng = np.random.default_rng(5)
    for w in (2560, 330, 100):          # k = 16 (even), 2, and the stride <= 1 branch (k = width // 2)
        n = 60
        x = rng.integers(-20, w - 5, n)
        y = rng.integers(-10, 400, n)
        b = np.stack([x, y, x + rng.integers(3, 120, n), y + rng.integers(3, 60, n)], 1).astype(np.float32)
        got = geo.lines_from_bboxes(np.zeros((420, w), np.uint8), b)
        np.testing.assert_array_equal(got, gr.lines_from_bboxes(b, 420, w))
