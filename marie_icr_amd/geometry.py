"""Word-box / line geometry of the DiT box processor — the reference's function names over libmarie_hip.so's native
implementations (``csrc/geometry.hip``).  These are O(N^2) Python loops in the reference; at ~400 word boxes a page they
cost more than the detector does on an MI355X, so they are native here.  Host code, no GPU needed.

reference: marie/utils/overlap.py:268-330 (merge_boxes), :186-204 (merge_bboxes_as_block);
marie/boxes/line_processor.py:15-44 (find_line_number), :105-171 (line_merge);
marie/boxes/dit/ulim_dit_box_processor.py:201-288 (lines_from_bboxes).
"""
from __future__ import annotations

import ctypes as C
from typing import List, Sequence

import numpy as np

from . import _lib
from ._lib import MarieHipError


def _check(rc: int, what: str):
    if rc != 0:
        raise MarieHipError(f"{what} failed with code {rc}")


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def merge_boxes(bboxes_xyxy, delta_x: float = 0.0, delta_y: float = 0.0) -> List[List[float]]:
    """Merge word boxes that overlap in x and whose centre-y lies within +-0.5 h of the anchor; at most three rounds.
    ``delta_x`` / ``delta_y`` are accepted and ignored, as in the reference.  Returns a list of [x1, y1, x2, y2]."""
    b = np.ascontiguousarray(np.asarray(bboxes_xyxy, np.float32).reshape(-1, 4))
    out = np.empty_like(b)
    n = C.c_int(0)
    _check(_lib.load().mhip_merge_boxes(_ptr(b), len(b), _ptr(out), C.byref(n)), "mhip_merge_boxes")
    return [list(r) for r in out[: n.value]]


def line_merge(image, bboxes: Sequence[Sequence[int]], enable_visualization: bool = False) -> np.ndarray:
    """Merge (x, y, w, h) boxes into lines by iterated vertical IoU; ``image`` is only asked for being present, as in
    the reference.  Returns an (n, 4) integer array sorted by y ([] for no boxes)."""
    if len(bboxes) == 0:
        return []
    if image is None:
        raise ValueError("Image is None or invalid.")
    b = np.ascontiguousarray(np.asarray(bboxes, np.int32).reshape(-1, 4))
    out = np.empty_like(b)
    n = C.c_int(0)
    _check(_lib.load().mhip_line_merge(_ptr(b), len(b), _ptr(out), C.byref(n)), "mhip_line_merge")
    return out[: n.value].astype(np.int64)


def find_line_numbers(lines, boxes_xywh) -> List[int]:
    """``find_line_number`` for every box of a page in one call."""
    ln = np.ascontiguousarray(np.asarray(lines, np.int32).reshape(-1, 4))
    bx = np.ascontiguousarray(np.asarray(boxes_xywh, np.int32).reshape(-1, 4))
    out = np.empty((len(bx),), np.int32)
    _check(_lib.load().mhip_find_line_numbers(_ptr(ln), len(ln), _ptr(bx), len(bx), _ptr(out)), "mhip_find_line_numbers")
    return out.tolist()


def lines_from_bboxes(image, bboxes) -> np.ndarray:
    """Line boxes (x, y, w, h) for the word boxes (xmin, ymin, xmax, ymax) of a page; only ``image.shape[:2]`` is used."""
    h, w = int(image.shape[0]), int(image.shape[1])
    b = np.ascontiguousarray(np.asarray(bboxes, np.float32).reshape(-1, 4))
    cap = max(len(b), 1)
    out = np.empty((cap, 4), np.int32)
    n = C.c_int(0)
    _check(_lib.load().mhip_lines_from_bboxes(_ptr(b), len(b), h, w, _ptr(out), cap, C.byref(n)), "mhip_lines_from_bboxes")
    return out[: n.value].astype(np.int64)
