"""ORACLE (test infrastructure, not product code) — CPU restatement of the reference's word-box / line geometry.

PINNED: ``merge_boxes``, ``line_merge``, ``find_line_number`` and ``merge_bboxes_as_block`` are checked in
``tests/test_oracle_geometry.py`` against ``tests/golden/geometry.npz``, which ``oracle/gen_golden.py --geometry-only``
produced by running the reference's unmodified ``marie/utils/overlap.py`` and ``marie/boxes/line_processor.py``.
``lines_from_bboxes`` rasterises with numpy and labels with scipy where the reference calls OpenCV
(``cv2.rectangle`` / ``erode`` / ``dilate`` / ``connectedComponentsWithStats``; cv2 is not installed here): that stage is
PARITY UNPINNED against OpenCV itself; the band-analytic product code is checked against this dense raster.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this module.
"""
from __future__ import annotations

import numpy as np

F32 = np.float32


def _round6(v):
    """``round(np.float32, 6)`` as numpy evaluates it: multiply, rint, divide — all in float32."""
    return F32(np.rint(F32(v) * F32(1e6)) / F32(1e6))


def merge_boxes(xyxy: np.ndarray) -> np.ndarray:
    """reference: marie/utils/overlap.py:268-330 with find_overlap_horizontal(center_y_overlap=0.5) :106-183 and
    merge_bboxes_as_block :186-204.  float32 arithmetic throughout (numpy-2 scalar rules)."""
    b = np.asarray(xyxy, F32).reshape(-1, 4)
    cur = np.stack([b[:, 0], b[:, 1], b[:, 2] - b[:, 0], b[:, 3] - b[:, 1]], 1).astype(F32)
    last = len(cur)
    for _ in range(3):
        n = len(cur)
        x, y, w, h = cur[:, 0], cur[:, 1], cur[:, 2], cur[:, 3]
        xr = (x + w).astype(F32)
        cy = (y + np.floor_divide(h, F32(2))).astype(F32)
        lo = (cy - h * F32(0.5)).astype(F32)
        hi = (cy + h * F32(0.5)).astype(F32)
        same = (cur[:, None, :] == cur[None, :, :]).all(axis=2)
        hit = (x[:, None] < xr[None, :]) & (x[None, :] < xr[:, None]) & ~same
        hit &= ~((cy[None, :] < lo[:, None]) | (cy[None, :] > hi[:, None]))
        visited = np.zeros(n, bool)
        groups = []
        for i in range(n):
            if visited[i]:
                continue
            visited[i] = True
            members = np.flatnonzero(hit[i])
            visited[members] = True
            groups.append(np.concatenate([[i], members]))
        if len(groups) == n:
            break
        nxt = []
        for g in groups:
            p = cur[g]
            mx, my = p[:, 0].min(), p[:, 1].min()
            nxt.append([_round6(mx), _round6(my), _round6((p[:, 0] + p[:, 2]).max() - mx),
                        _round6((p[:, 1] + p[:, 3]).max() - my)])
        cur = np.asarray(nxt, F32)
        if last == len(cur):
            break
        last = len(cur)
    return np.stack([cur[:, 0], cur[:, 1], cur[:, 0] + cur[:, 2], cur[:, 1] + cur[:, 3]], 1).astype(F32)


def merge_bboxes_as_block(xywh) -> list:
    """reference: marie/utils/overlap.py:186-204."""
    b = np.asarray(xywh)
    mx, my = b[:, 0].min(), b[:, 1].min()
    return [round(k, 6) for k in (mx, my, (b[:, 0] + b[:, 2]).max() - mx, (b[:, 1] + b[:, 3]).max() - my)]


def _viou(b: np.ndarray):
    """Pairwise 1-D vertical IoU table + the 'counts as an overlap' mask of find_overlap_vertical
    (marie/utils/overlap.py:42-103): positive heights, not the identical box, open-interval intersection."""
    y, h = b[:, 1].astype(np.int64), b[:, 3].astype(np.int64)
    yb = y + h
    inter = np.minimum(yb[:, None], yb[None, :]) - np.maximum(y[:, None], y[None, :])
    ok = (y[:, None] < yb[None, :]) & (y[None, :] < yb[:, None]) & (h[:, None] > 0) & (h[None, :] > 0)
    ok &= ~(b[:, None, :] == b[None, :, :]).all(axis=2)
    with np.errstate(divide="ignore", invalid="ignore"):
        iou = inter / (h[:, None] + h[None, :] - inter).astype(np.float64)
    return ok, np.clip(iou, 0.0, 1.0)


def _line_merge_pass(b: np.ndarray, min_iou: float) -> np.ndarray:
    """reference: marie/boxes/line_processor.py:47-102.  Sort by y is STABLE here (the reference's default
    ``np.argsort`` leaves the order of equal keys to the platform's sort kernel)."""
    b = b[np.argsort(b[:, 1], kind="stable")]
    ok, iou = _viou(b)
    count = ok.sum(axis=1)
    visited = np.zeros(len(b), bool)
    out = []
    for i in range(len(b)):
        if visited[i]:
            continue
        visited[i] = True
        group = [i]
        for j in np.flatnonzero(ok[i]):
            if visited[j] or iou[i, j] < min_iou:
                continue
            if count[j] == count[i]:
                group.append(j)
                visited[j] = True
        p = b[group]
        mx = p[:, 0].min()
        out.append([mx, p[:, 1].min(), (p[:, 0] + p[:, 2]).max() - mx, p[:, 3].max()])
    return np.asarray(out, b.dtype)


def line_merge(xywh) -> np.ndarray:
    """reference: marie/boxes/line_processor.py:105-171."""
    b = np.asarray(xywh, np.int64).reshape(-1, 4)
    if len(b) == 0:
        return b
    still = 0
    for thr in (0.8, 0.7, 0.6, 0.5, 0.4, 0.37, 0.35):
        before = len(b)
        b = _line_merge_pass(b, thr)
        if len(b) == before:
            still += 1
            if still > 2:
                break
    x, y, w, h = b[:, 0], b[:, 1], b[:, 2], b[:, 3]
    inside = (x[None, :] > x[:, None]) & ((x + w)[None, :] < (x + w)[:, None]) & (y[None, :] > y[:, None]) & \
        ((y + h)[None, :] < (y + h)[:, None])
    np.fill_diagonal(inside, False)
    b = b[~inside.any(axis=0)]
    return b[np.argsort(b[:, 1], kind="stable")]


def find_line_number(lines, box) -> int:
    """reference: marie/boxes/line_processor.py:15-44."""
    lines = np.asarray(lines, np.int64).reshape(-1, 4)
    box = np.asarray(box, np.int64)
    if len(lines) == 0:
        return -1
    ok, iou = _viou(np.concatenate([box[None], lines]))
    ok, iou = ok[0, 1:], iou[0, 1:]
    idx = np.flatnonzero(ok)
    if len(idx) == 1:
        return int(idx[0]) + 1
    if len(idx) > 1:
        best, num = 0.0, -1
        for j in idx:
            if iou[j] > best:
                best, num = iou[j], int(j) + 1
        if num != -1:
            return num
    dy = np.abs((box[1] + box[3] // 2) - (lines[:, 1] + lines[:, 3]))
    best, num = 100000, -1
    for j, d in enumerate(dy):
        if d < best:
            best, num = d, j + 1
    return num


def lines_mask(xyxy: np.ndarray, height: int, width: int) -> np.ndarray:
    """The binary mask ``lines_from_bboxes`` labels (reference: marie/boxes/dit/ulim_dit_box_processor.py:218-257):
    reduced-height rectangles (both corners inclusive, clipped), then a horizontal closing of the rectangles with a
    (k, 1) box element anchored at k // 2; pixels outside the image never win the min/max (OpenCV's default morphology
    border).  Returns bool (True = line pixel)."""
    mask = np.zeros((height, width), bool)
    for x1, y1, x2, y2 in np.asarray(xyxy).astype(np.int32).tolist():
        q = (y2 - y1) // 8
        h = (y2 - y1) // 2 + q
        ya = y1 + h // 2 - q
        xa, xb, yb = max(x1, 0), min(x2, width - 1), min(ya + h, height - 1)
        ya = max(ya, 0)
        if xa <= xb and ya <= yb:
            mask[ya:yb + 1, xa:xb + 1] = True
    stride = width // min(160, width)
    k = stride if stride > 1 else width // 2
    a = k // 2
    # erode of the white background == grow black: black at x if any black in [x-a, x-a+k-1]
    pad = np.zeros((height, width + k - 1), bool)
    pad[:, a:a + width] = mask                                   # image x -> padded x + a; window -> [x, x + k - 1]
    c = np.concatenate([np.zeros((height, 1), np.int32), np.cumsum(pad, axis=1, dtype=np.int32)], axis=1)
    xs = np.arange(width)
    grown = (c[:, xs + k] - c[:, xs]) > 0
    # dilate of white == shrink black: black at x iff every in-image pixel of [x-a, x-a+k-1] is black
    cg = np.concatenate([np.zeros((height, 1), np.int32), np.cumsum(grown, axis=1, dtype=np.int32)], axis=1)
    wl = np.clip(xs - a, 0, width - 1)
    wr = np.clip(xs - a + k - 1, 0, width - 1)
    return (cg[:, wr + 1] - cg[:, wl]) == (wr - wl + 1)


def lines_from_bboxes(xyxy: np.ndarray, height: int, width: int) -> np.ndarray:
    """reference: marie/boxes/dit/ulim_dit_box_processor.py:201-288 — 4-connected components of ``lines_mask`` in raster
    order of their first pixel, stats boxes with h >= 2 and w >= 4, then ``line_merge``."""
    from scipy import ndimage

    mask = lines_mask(xyxy, height, width)
    lab, n = ndimage.label(mask, structure=[[0, 1, 0], [1, 1, 1], [0, 1, 0]])
    frags = []
    for sl in ndimage.find_objects(lab):
        y0, y1, x0, x1 = sl[0].start, sl[0].stop, sl[1].start, sl[1].stop
        if y1 - y0 < 2 or x1 - x0 < 4:
            continue
        frags.append([x0, y0, x1 - x0, y1 - y0])
    if not frags:
        return np.zeros((0, 4), np.int64)
    return line_merge(frags)
