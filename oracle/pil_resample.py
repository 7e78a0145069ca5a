"""ORACLE (test infrastructure, not product code) — numpy restatement of the recognizer's crop batcher:
BGR fragment -> PIL "L" -> ``Image.resize((w', 32), Image.BICUBIC)`` -> replicate-pad to imgW.

reference: MemoryDataset.__getitem__ (marie/models/icr/memory_dataset.py:40-55) and AlignCollate / NormalizePAD
(marie/models/icr/dataset.py:275-324).  The arithmetic restated here is Pillow's (libImaging/Convert.c ``rgb2l`` and
libImaging/Resample.c, 8 bits per channel): Pillow IS installed in the build container, so this restatement is PINNED
directly against ``PIL.Image`` in tests/test_oracle_resample.py (Pillow 12.2 here vs ~=9.5 in the reference's
requirements; the resampling code is unchanged between them).
"""
from __future__ import annotations

import math
from typing import List, Sequence

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def bgr_to_l(img_bgr: np.ndarray) -> np.ndarray:
    """cv2 BGR -> RGB -> ``Image.convert("L")``: L = (R*19595 + G*38470 + B*7471 + 0x8000) >> 16."""
    b = img_bgr[:, :, 0].astype(np.uint32)
    g = img_bgr[:, :, 1].astype(np.uint32)
    r = img_bgr[:, :, 2].astype(np.uint32)
    return ((r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16).astype(np.uint8)


def _bicubic(x: float) -> float:
    a = -0.5
    if x < 0.0:
        x = -x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def precompute_coeffs(in_size: int, out_size: int):
    """Resample.c ``precompute_coeffs`` + ``normalize_coeffs_8bpc`` for the full-image box, bicubic (support 2)."""
    scale = float(in_size) / out_size
    filterscale = max(scale, 1.0)
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int64)
    kk = np.zeros((out_size, ksize), np.int64)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = [_bicubic((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        for x in range(xmax):
            v = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _resample_axis(img: np.ndarray, out_size: int, axis: int) -> np.ndarray:
    src = np.moveaxis(img, axis, 1).astype(np.int64)         # (other, in)
    bounds, kk = precompute_coeffs(src.shape[1], out_size)
    out = np.empty((src.shape[0], out_size), np.uint8)
    for xx in range(out_size):
        xmin, xmax = bounds[xx]
        ss0 = (1 << (PRECISION_BITS - 1)) + (src[:, xmin:xmin + xmax] * kk[xx, :xmax][None, :]).sum(axis=1)
        out[:, xx] = np.clip(ss0 >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return np.moveaxis(out, 1, axis)


def resize_bicubic_l(img: np.ndarray, out_w: int, out_h: int) -> np.ndarray:
    """``Image.fromarray(img, "L").resize((out_w, out_h), Image.BICUBIC)``: horizontal pass, 8-bit rounding,
    vertical pass (ImagingResampleInner)."""
    if (img.shape[1], img.shape[0]) == (out_w, out_h):
        return img.copy()
    tmp = _resample_axis(img, out_w, 1) if img.shape[1] != out_w else img
    return _resample_axis(tmp, out_h, 0) if tmp.shape[0] != out_h else tmp


def resized_width(w: int, h: int, img_h: int, img_w: int) -> int:
    """AlignCollate's width rule (dataset.py:313-318)."""
    ratio = w / float(h)
    return img_w if math.ceil(img_h * ratio) > img_w else int(math.ceil(img_h * ratio))


def align_collate_u8(images: Sequence[np.ndarray], img_w: int, img_h: int = 32) -> np.ndarray:
    """(n, img_h, img_w) uint8: gray, aspect-preserving bicubic resize to height img_h, right edge replicated."""
    out = np.empty((len(images), img_h, img_w), np.uint8)
    for i, im in enumerate(images):
        a = np.asarray(im)
        g = bgr_to_l(a) if a.ndim == 3 else a
        rw = max(1, resized_width(g.shape[1], g.shape[0], img_h, img_w))
        r = resize_bicubic_l(g, rw, img_h)
        out[i, :, :rw] = r
        if rw < img_w:
            out[i, :, rw:] = r[:, rw - 1:rw]
    return out


def align_collate_pil(images: Sequence[np.ndarray], img_w: int, img_h: int = 32) -> np.ndarray:
    """The same through Pillow itself (what the reference executes) — used to pin the restatement."""
    from PIL import Image

    out = np.empty((len(images), img_h, img_w), np.uint8)
    for i, im in enumerate(images):
        a = np.asarray(im)
        pil = Image.fromarray(np.ascontiguousarray(a[:, :, ::-1])).convert("L") if a.ndim == 3 else Image.fromarray(a)
        w, h = pil.size
        rw = max(1, resized_width(w, h, img_h, img_w))
        g = np.asarray(pil.resize((rw, img_h), Image.BICUBIC), dtype=np.uint8)
        out[i, :, :rw] = g
        if rw < img_w:
            out[i, :, rw:] = g[:, rw - 1:rw]
    return out
