"""Shared pieces of the box-processor surface: ``PSMode`` and the line-number helpers.

reference: marie/boxes/box_processor.py:129-163 (PSMode), marie/boxes/line_processor.py:15-44
(find_line_number), marie/utils/overlap.py:42-103 (find_overlap_vertical).  This is host logic in the
reference too (a handful of boxes per page); it is restated here because the reference package cannot
travel with this one.
"""
from __future__ import annotations

from enum import Enum
from typing import List, Sequence


class PSMode(Enum):
    """Page segmentation modes — same members/values as the reference enum."""

    WORD = "word"
    SPARSE = "sparse"
    LINE = "line"
    RAW_LINE = "raw_line"
    MULTI_LINE = "multiline"

    @staticmethod
    def from_value(value):
        if value is None:
            return PSMode.SPARSE
        for data in PSMode:
            if data.value == str(value).lower():
                return data
        return PSMode.SPARSE


def find_overlap_vertical(box: Sequence[int], data: Sequence[Sequence[int]]):
    """1-D vertical IoU of ``box`` (x, y, w, h) against every box of ``data``; identical boxes are skipped.
    reference: marie/utils/overlap.py:42-103."""
    overlaps, indexes, scores = [], [], []
    if len(data) == 0:
        return [], [], []
    x, y, w, h = box
    y1min, y1max = y, y + h
    for i, bb in enumerate(data):
        _x, _y, _w, _h = bb
        y2min, y2max = _y, _y + _h
        if h <= 0 or _h <= 0:
            continue
        if box[0] == bb[0] and box[1] == bb[1] and box[2] == bb[2] and box[3] == bb[3]:
            continue
        y_bottom, y_top = min(y1max, y2max), max(y1min, y2min)
        if y1min < y2max and y2min < y1max:
            inter = y_bottom - y_top
            iou = max(min(inter / float(h + _h - inter), 1.0), 0.0)
            scores.append(iou)
            overlaps.append(bb)
            indexes.append(i)
    return overlaps, indexes, scores


def find_line_number(lines: Sequence[Sequence[int]], box: Sequence[int]) -> int:
    """1-based index of the line with the best vertical IoU; with no overlap, the line whose bottom edge is nearest
    to the box centre; -1 when there are no lines at all.  reference: marie/boxes/line_processor.py:15-44."""
    line_number = -1
    overlaps, indexes, scores = find_overlap_vertical(box, lines)
    if len(indexes) == 1:
        line_number = indexes[0] + 1
    elif len(indexes) > 1:
        iou_best = 0
        for overlap, index, score in zip(overlaps, indexes, scores):
            if score > iou_best:
                iou_best = score
                line_number = index + 1
    if line_number == -1:
        min_y = 100000
        for i, line in enumerate(lines):
            line_y = line[1] + line[3]
            box_y = box[1] + box[3] // 2
            dy = abs(box_y - line_y)
            if dy < min_y:
                line_number = i + 1
                min_y = dy
    return line_number
