"""ORACLE (test infrastructure, not product code) — CPU restatement of the page-size clamp the reference applies before OCR:
``ensure_max_page_size`` (marie/utils/image_utils.py:254-321) and the ``cv2.resize(..., interpolation=cv2.INTER_AREA)`` it
calls (:313-315); and of ``resize_image`` (marie/utils/resize_image.py:9-76) with its ``cv2.INTER_CUBIC`` shrink, which
frames small pages / region crops for the DiT detector (marie/boxes/dit/ulim_dit_box_processor.py:524-540).

Pinning: the SHAPE RULE is pinned by the reference's own tests (tests/imaging/test_image_resizing.py:7-34: three cases
agree with the reference's code; the fourth, ``test_max_page_001`` :37-44, expects (3200, 2600) for a 4171 x 2569 frame,
which the reference's code — and this restatement — turn into (3795, 2337): a stale test, kept out of the fixtures and
recorded in tests/test_oracle_ingest.py).  The RESAMPLING ARITHMETIC is PARITY UNPINNED: opencv-python 4.8.1.78 is a
third-party dependency, absent from /root/reference and not installed here, and no reference fixture holds resized
pixels.  It is restated from OpenCV's published area resampler (modules/imgproc/src/resize.cpp: ``computeResizeAreaTab``,
``resizeArea_``, ``resizeAreaFast_``): coverage tables in double, float32 accumulation along x then y in table order,
round-half-even, saturate; exactly integral scales sum blocks (2 x 2: ``(a + b + c + d + 2) >> 2``).
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this module.
"""
from __future__ import annotations

import math
from typing import List, Tuple

import numpy as np


def max_page_size(width: int, height: int, max_page_size: Tuple[int, int] = (2550, 3300), expand_ratio: float = 0.15):
    """image_utils.py:275-310 for one frame -> (changed, new_width, new_height)."""
    max_width_portrait, max_height_portrait = max_page_size
    if width > height:
        max_width, max_height = max_height_portrait, max_width_portrait
    else:
        max_width, max_height = max_width_portrait, max_height_portrait
    max_width = max_width + int(max_width * expand_ratio)
    max_height = max_height + int(max_height * expand_ratio)
    if not (width > max_width or height > max_height):
        return False, width, height
    aspect_ratio = width / height
    if width > height:
        new_width = min(width, max_width)
        new_height = int(new_width / aspect_ratio)
        if new_height > max_height:
            new_height = max_height
            new_width = int(new_height * aspect_ratio)
    else:
        new_height = min(height, max_height)
        new_width = int(new_height * aspect_ratio)
        if new_width > max_width:
            new_width = max_width
            new_height = int(new_width / aspect_ratio)
    return True, new_width, new_height


def _area_tab(ssize: int, dsize: int, scale: float):
    """computeResizeAreaTab: per destination index the list of (source index, float32 weight), in table order."""
    tab = []
    for d in range(dsize):
        f1 = d * scale
        f2 = f1 + scale
        cell = min(scale, ssize - f1)
        s1, s2 = math.ceil(f1), math.floor(f2)
        s2 = min(s2, ssize - 1)
        s1 = min(s1, s2)
        ent = []
        if s1 - f1 > 1e-3:
            ent.append((s1 - 1, np.float32((s1 - f1) / cell)))
        for s in range(s1, s2):
            ent.append((s, np.float32(1.0 / cell)))
        if f2 - s2 > 1e-3:
            ent.append((s2, np.float32(min(min(f2 - s2, 1.0), cell) / cell)))
        tab.append(ent)
    return tab


def _pack(tab):
    """ragged (index, weight) lists -> dense [n][depth] arrays + counts (vectorised accumulation in table order)."""
    depth = max(len(e) for e in tab)
    idx = np.zeros((len(tab), depth), np.int64)
    wgt = np.zeros((len(tab), depth), np.float32)
    cnt = np.array([len(e) for e in tab])
    for d, ent in enumerate(tab):
        for r, (s, w) in enumerate(ent):
            idx[d, r], wgt[d, r] = s, w
    return idx, wgt, cnt


def _sat_u8(v: np.ndarray) -> np.ndarray:
    return np.clip(np.rint(v), 0, 255).astype(np.uint8)          # np.rint: round half to even, as cvRound


def resize_area(img: np.ndarray, new_width: int, new_height: int) -> np.ndarray:
    """cv2.resize(img, (new_width, new_height), interpolation=cv2.INTER_AREA) for uint8 HxW / HxWx{1,3}, shrinking only."""
    src = img if img.ndim == 3 else img[:, :, None]
    sh, sw, cn = src.shape
    dh, dw = new_height, new_width
    assert src.dtype == np.uint8 and 0 < dh <= sh and 0 < dw <= sw
    scale_x, scale_y = 1.0 / (dw / sw), 1.0 / (dh / sh)
    ix, iy = int(round(scale_x)), int(round(scale_y))
    eps = np.finfo(np.float64).eps
    if abs(scale_x - ix) < eps and abs(scale_y - iy) < eps:          # resizeAreaFast_
        blocks = src[:dh * iy, :dw * ix].astype(np.int64).reshape(dh, iy, dw, ix, cn).sum(axis=(1, 3))
        if ix == 2 and iy == 2:
            out = ((blocks + 2) >> 2).astype(np.uint8)
        else:
            out = _sat_u8(blocks.astype(np.float32) * np.float32(np.float32(1.0) / np.float32(ix * iy)))
        return out if img.ndim == 3 else out[:, :, 0]
    xi, xw, xc = _pack(_area_tab(sw, dw, scale_x))
    ytab = _area_tab(sh, dh, scale_y)
    out = np.empty((dh, dw, cn), np.uint8)
    for dy, yent in enumerate(ytab):
        total = np.zeros((dw, cn), np.float32)
        for sy, beta in yent:
            row = src[sy].astype(np.float32)                         # (sw, cn)
            buf = np.zeros((dw, cn), np.float32)
            for r in range(xi.shape[1]):
                live = (xc > r)[:, None]
                term = row[xi[:, r]] * xw[:, r:r + 1]                # float32 product, rounded once
                buf = np.where(live, buf + term, buf)                # float32 sum, rounded once (no fused multiply-add)
            total = total + beta * buf
        out[dy] = _sat_u8(total)
    return out if img.ndim == 3 else out[:, :, 0]


def ensure_max_page_size(frames: List[np.ndarray], max_page_size_: Tuple[int, int] = (2550, 3300),
                         expand_ratio: float = 0.15):
    """image_utils.py:254-321 -> (changed, frames)."""
    out, changed = [], False
    for frame in frames:
        h, w = frame.shape[:2]
        ch, nw, nh = max_page_size(w, h, max_page_size_, expand_ratio)
        if ch:
            changed = True
            out.append(resize_area(frame, nw, nh))
        else:
            out.append(frame)
    return changed, out


# ---------------------------------------------------------------------------------------------- cv2.INTER_CUBIC (8-bit)
def _cubic_tab(dsize: int, scale: float):
    """per destination index: first tap's source index - 1 ... and four 11-bit fixed-point Keys (A = -0.75) weights, the
    way OpenCV's resize() fills xofs / ialpha: float32 polynomial, x 2048, round half to even."""
    d = np.arange(dsize, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    fl = np.floor(f)
    x = (f - fl).astype(np.float32)
    one, A = np.float32(1.0), np.float32(-0.75)
    x1 = x + one
    w0 = ((A * x1 - np.float32(5.0) * A) * x1 + np.float32(8.0) * A) * x1 - np.float32(4.0) * A
    w1 = ((A + np.float32(2.0)) * x - (A + np.float32(3.0))) * x * x + one
    xm = one - x
    w2 = ((A + np.float32(2.0)) * xm - (A + np.float32(3.0))) * xm * xm + one
    w3 = one - w0 - w1 - w2
    w = np.stack([w0, w1, w2, w3], axis=1).astype(np.float32)
    coef = np.clip(np.rint(w * np.float32(2048.0)), -32768, 32767).astype(np.int64)
    return fl.astype(np.int64), coef


def resize_cubic(img: np.ndarray, new_width: int, new_height: int) -> np.ndarray:
    """cv2.resize(img, (new_width, new_height), interpolation=cv2.INTER_CUBIC) for uint8 HxW / HxWx{1,3}: OpenCV's generic
    separable fixed-point path (integer row combination, (v + 2^21) >> 22, saturate).  PARITY UNPINNED (see the module
    header); x86 builds of OpenCV combine the rows of 8-pixel groups in float32 instead, which can differ by one level."""
    src = img if img.ndim == 3 else img[:, :, None]
    sh, sw, cn = src.shape
    dh, dw = int(new_height), int(new_width)
    sx, ca = _cubic_tab(dw, 1.0 / (dw / sw))
    sy, cb = _cubic_tab(dh, 1.0 / (dh / sh))
    s64 = src.astype(np.int64)
    hor = np.zeros((sh, dw, cn), np.int64)
    for j in range(4):
        xx = np.clip(sx - 1 + j, 0, sw - 1)
        hor += s64[:, xx, :] * ca[None, :, j, None]
    assert np.abs(hor).max() < 2 ** 31
    acc = np.zeros((dh, dw, cn), np.int64)
    for k in range(4):
        yy = np.clip(sy - 1 + k, 0, sh - 1)
        acc += hor[yy] * cb[:, k, None, None]
    out = np.clip((acc + (1 << 21)) >> 22, 0, 255).astype(np.uint8)
    return out if img.ndim == 3 else out[:, :, 0]


def resize_image(image: np.ndarray, desired_size, color=(255, 255, 255), keep_max_size: bool = False):
    """marie/utils/resize_image.py:9-76 -> (image, (x, y, w, h))."""
    if image.shape[0] == desired_size[0] and image.shape[1] == desired_size[1]:
        return image, (0, 0, image.shape[1], image.shape[0])
    size = image.shape[:2]

    def border(img, top, bottom, left, right):
        out = np.empty((img.shape[0] + top + bottom, img.shape[1] + left + right) + img.shape[2:], img.dtype)
        out[...] = np.asarray(color, img.dtype) if img.ndim == 3 else color[0]
        out[top:top + img.shape[0], left:left + img.shape[1]] = img
        return out

    if keep_max_size:
        h, w = size
        dh, dw = desired_size
        if w > dw and h < dh:
            delta_h = max(0, desired_size[0] - size[0])
            top, bottom = delta_h // 2, delta_h - (delta_h // 2)
            image = border(image, top, bottom, 40, 40)
            size = image.shape[:2]
            return image, (40, top, size[1], size[0])
    if size[0] > desired_size[0] or size[1] > desired_size[1]:
        ratio = min(float(desired_size[0]) / size[0], float(desired_size[1]) / size[1])
        new_size = tuple(int(x * ratio) for x in size)
        image = resize_cubic(image, new_size[1], new_size[0])
        size = image.shape
    delta_w = max(0, desired_size[1] - size[1])
    delta_h = max(0, desired_size[0] - size[0])
    top, bottom = delta_h // 2, delta_h - (delta_h // 2)
    left, right = delta_w // 2, delta_w - (delta_w // 2)
    image = border(image, top, bottom, left, right)
    return image, (left, top, size[1], size[0])
