"""ORACLE (test infrastructure, not product code) — CPU restatement of the reference's CRAFT text
detector: pre-processing, network forward, score-map post-processing, box/crop extraction.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this.

Pinning status (also recorded in DESIGN.md):
  * network forward ``craft_forward`` — PINNED for the U-net/heads: ``oracle/gen_golden.py`` runs the
    reference's own unmodified ``CRAFT`` class (marie/models/craft/craft.py) on seeded weights.  Its VGG16-BN
    backbone comes from torchvision, which is NOT in this container and not vendored by the reference
    (pinned only as "torchvision"); the generator supplies the published torchvision
    ``vgg16_bn().features`` layer list (cfg "D" + BatchNorm, ``ReLU(inplace=True)``) so that the reference's
    slicing code (basenet/vgg16_bn.py:23-74) runs on it.
  * ``cv_resize_linear_u8``, ``connected_components``, ``min_area_rect``/``box_points`` restate OpenCV
    4.8 (opencv-python==4.8.1.78 in the reference's requirements; absent here): PARITY UNPINNED, anchored on
    the reference call sites cited at each function and cross-checked against independent implementations
    (scipy.ndimage) in tests/.
"""
from __future__ import annotations

import math
from typing import Dict, List, Tuple

import numpy as np

F32 = np.float32


# --------------------------------------------------------------------------- #
# pre-processing
# --------------------------------------------------------------------------- #
def _cv_round_short(v: np.ndarray) -> np.ndarray:
    """saturate_cast<short>(float): round half to even, clamp to int16."""
    return np.clip(np.rint(v), -32768, 32767).astype(np.int32)


def _linear_coeffs(src: int, dst: int):
    """OpenCV resize INTER_LINEAR coefficient tables for one axis (8-bit path, 11-bit fixed point)."""
    scale = float(src) / float(dst)
    d = np.arange(dst, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int32)
    f = (f - s.astype(np.float32)).astype(np.float32)
    lo = s < 0
    f[lo] = 0.0
    s[lo] = 0
    hi = s >= src - 1
    f[hi] = 0.0
    s[hi] = src - 1
    a0 = _cv_round_short((F32(1.0) - f) * F32(2048.0))
    a1 = _cv_round_short(f * F32(2048.0))
    return s, a0, a1


def cv_resize_linear_u8(img: np.ndarray, dst_w: int, dst_h: int) -> np.ndarray:
    """``cv2.resize(img_u8, (dst_w, dst_h), interpolation=cv2.INTER_LINEAR)``.

    reference call site: marie/models/craft/imgproc.py:58.  OpenCV's 8-bit bilinear is fixed point:
    horizontal pass in int with 2048-scaled weights, vertical pass
    ``(((b0*(S0>>4))>>16) + ((b1*(S1>>4))>>16) + 2) >> 2``.
    """
    h, w = img.shape[:2]
    if (dst_w, dst_h) == (w, h):
        return img.copy()
    sx, ax0, ax1 = _linear_coeffs(w, dst_w)
    sy, by0, by1 = _linear_coeffs(h, dst_h)
    src = img.astype(np.int32)
    sx1 = np.minimum(sx + 1, w - 1)
    rows = src[:, sx, :] * ax0[None, :, None] + src[:, sx1, :] * ax1[None, :, None]  # (h, dst_w, c) int32
    sy1 = np.minimum(sy + 1, h - 1)
    s0 = rows[sy] >> 4
    s1 = rows[sy1] >> 4
    out = (((by0[:, None, None] * s0) >> 16) + ((by1[:, None, None] * s1) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def craft_target_size(h: int, w: int, canvas_size: int, mag_ratio: float = 1.0):
    """reference: resize_aspect_ratio, marie/models/craft/imgproc.py:45-71."""
    target_size = mag_ratio * max(h, w)
    if target_size > canvas_size:
        target_size = canvas_size
    ratio = target_size / max(h, w)
    th, tw = int(h * ratio), int(w * ratio)
    th32 = th + (32 - th % 32) if th % 32 else th
    tw32 = tw + (32 - tw % 32) if tw % 32 else tw
    return ratio, th, tw, th32, tw32


def craft_preprocess(img_u8: np.ndarray, canvas_size: int, mag_ratio: float = 1.0):
    """uint8 HxWx3 -> normalised fp32 canvas (1,3,H32,W32), ratio, (th, tw).

    reference: resize_aspect_ratio + normalizeMeanVariance (imgproc.py:26-33,45-71): the zero canvas is
    filled BEFORE normalisation, so the padding becomes (0-127.5)/127.5 = -1."""
    h, w = img_u8.shape[:2]
    ratio, th, tw, th32, tw32 = craft_target_size(h, w, canvas_size, mag_ratio)
    proc = cv_resize_linear_u8(img_u8, tw, th)
    canvas = np.zeros((th32, tw32, 3), dtype=F32)
    canvas[:th, :tw] = proc
    canvas -= np.array([127.5, 127.5, 127.5], dtype=F32)
    canvas /= np.array([127.5, 127.5, 127.5], dtype=F32)
    return canvas.transpose(2, 0, 1)[None].copy(), ratio, (th, tw)


# --------------------------------------------------------------------------- #
# network
# --------------------------------------------------------------------------- #
def craft_forward(x: np.ndarray, st: Dict[str, np.ndarray]) -> Tuple[np.ndarray, np.ndarray]:
    """``CRAFT.forward`` (marie/models/craft/craft.py:59-81) on CPU fp32 with torch functional ops.

    x: (1,3,H,W) normalised.  Returns (y (1,H/2,W/2,2), feature (1,32,H/2,W/2)).
    Tap semantics: torchvision's ReLUs are in-place, so the slice outputs relu2_2 / relu3_2 / relu4_3 that the
    U-net reads are POST-ReLU (the next slice's first ReLU overwrites the saved tensor), while relu5_3 is the raw
    BatchNorm output because slice5 starts with a MaxPool (basenet/vgg16_bn.py:29-49,60-74).
    """
    import torch
    import torch.nn.functional as F

    t = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in st.items() if v.ndim > 0}

    def conv(h, name, pad=1, dil=1):
        return F.conv2d(h, t[name + ".weight"], t[name + ".bias"], padding=pad, dilation=dil)

    def bn(h, name):
        return F.batch_norm(h, t[name + ".running_mean"], t[name + ".running_var"], t[name + ".weight"],
                            t[name + ".bias"], False, 0.0, 1e-5)

    def cbr(h, c, b):
        return F.relu(bn(conv(h, c), b))

    with torch.no_grad():
        h = torch.from_numpy(x)
        p = "basenet."
        h = cbr(h, p + "slice1.0", p + "slice1.1")
        h = cbr(h, p + "slice1.3", p + "slice1.4")
        h = F.max_pool2d(h, 2, 2)
        h = cbr(h, p + "slice1.7", p + "slice1.8")
        relu2_2 = cbr(h, p + "slice1.10", p + "slice1.11")
        h = F.max_pool2d(relu2_2, 2, 2)
        h = cbr(h, p + "slice2.14", p + "slice2.15")
        relu3_2 = cbr(h, p + "slice2.17", p + "slice2.18")
        h = cbr(relu3_2, p + "slice3.20", p + "slice3.21")
        h = F.max_pool2d(h, 2, 2)
        h = cbr(h, p + "slice3.24", p + "slice3.25")
        relu4_3 = cbr(h, p + "slice3.27", p + "slice3.28")
        h = cbr(relu4_3, p + "slice4.30", p + "slice4.31")
        h = F.max_pool2d(h, 2, 2)
        h = cbr(h, p + "slice4.34", p + "slice4.35")
        relu5_3 = bn(conv(h, p + "slice4.37"), p + "slice4.38")          # no ReLU on this tap
        h = F.max_pool2d(relu5_3, 3, 1, 1)
        h = conv(h, p + "slice5.1", pad=6, dil=6)
        fc7 = conv(h, p + "slice5.2", pad=0)

        def up(y, c1, b1, c3, b3):
            y = F.relu(bn(conv(y, c1, pad=0), b1))
            return F.relu(bn(conv(y, c3), b3))

        y = up(torch.cat([fc7, relu5_3], 1), "upconv1.conv.0", "upconv1.conv.1", "upconv1.conv.3", "upconv1.conv.4")
        y = F.interpolate(y, size=relu4_3.shape[2:], mode="bilinear", align_corners=False)
        y = up(torch.cat([y, relu4_3], 1), "upconv2.conv.0", "upconv2.conv.1", "upconv2.conv.3", "upconv2.conv.4")
        y = F.interpolate(y, size=relu3_2.shape[2:], mode="bilinear", align_corners=False)
        y = up(torch.cat([y, relu3_2], 1), "upconv3.conv.0", "upconv3.conv.1", "upconv3.conv.3", "upconv3.conv.4")
        y = F.interpolate(y, size=relu2_2.shape[2:], mode="bilinear", align_corners=False)
        feature = up(torch.cat([y, relu2_2], 1), "upconv4.conv.0", "upconv4.conv.1", "upconv4.conv.3", "upconv4.conv.4")
        y = F.relu(conv(feature, "conv_cls.0"))
        y = F.relu(conv(y, "conv_cls.2"))
        y = F.relu(conv(y, "conv_cls.4"))
        y = F.relu(conv(y, "conv_cls.6", pad=0))
        y = conv(y, "conv_cls.8", pad=0)
        return y.permute(0, 2, 3, 1).contiguous().numpy(), feature.numpy()


# --------------------------------------------------------------------------- #
# post-processing
# --------------------------------------------------------------------------- #
def connected_components(mask: np.ndarray):
    """4-connected components with stats, labels numbered in raster order of each component's first pixel
    (what ``cv2.connectedComponentsWithStats(..., connectivity=4)`` yields; reference call site
    marie/models/craft/craft_utils.py:36-38).  Returns (n_labels incl. background, labels int32,
    stats[n,5] = left, top, width, height, area)."""
    h, w = mask.shape
    fg = mask != 0
    idx = np.arange(h * w, dtype=np.int64).reshape(h, w)
    # iterate min-propagation to a fixed point (vectorised label propagation; fine for test sizes)
    lab = np.where(fg, idx, np.int64(h * w))
    big = np.int64(h * w)
    while True:
        new = lab.copy()
        m = fg[:, 1:] & fg[:, :-1]
        new[:, 1:] = np.where(m, np.minimum(new[:, 1:], lab[:, :-1]), new[:, 1:])
        new[:, :-1] = np.where(m, np.minimum(new[:, :-1], lab[:, 1:]), new[:, :-1])
        m = fg[1:, :] & fg[:-1, :]
        new[1:, :] = np.where(m, np.minimum(new[1:, :], lab[:-1, :]), new[1:, :])
        new[:-1, :] = np.where(m, np.minimum(new[:-1, :], lab[1:, :]), new[:-1, :])
        # pointer jump: a pixel's label is itself a pixel index; adopt that pixel's label
        flat = new.reshape(-1)
        valid = flat < big
        jumped = flat.copy()
        jumped[valid] = flat[flat[valid]]
        new = jumped.reshape(h, w)
        if (new == lab).all():
            break
        lab = new
    roots = np.unique(lab[fg])                       # ascending = raster order of first pixel
    labels = np.zeros((h, w), dtype=np.int32)
    if roots.size:
        labels[fg] = (np.searchsorted(roots, lab[fg]) + 1).astype(np.int32)
    n = int(roots.size) + 1
    stats = np.zeros((n, 5), dtype=np.int64)
    ys, xs = np.nonzero(fg)
    ls = labels[ys, xs]
    for k in range(1, n):
        sel = ls == k
        x0, x1 = xs[sel].min(), xs[sel].max()
        y0, y1 = ys[sel].min(), ys[sel].max()
        stats[k] = (x0, y0, x1 - x0 + 1, y1 - y0 + 1, sel.sum())
    return n, labels, stats


def _convex_hull(pts: np.ndarray) -> np.ndarray:
    """Andrew monotone chain on integer points; returns hull vertices in counter-clockwise order
    (y down: visually clockwise), no collinear points."""
    p = np.unique(pts, axis=0)
    if len(p) <= 2:
        return p
    p = p[np.lexsort((p[:, 1], p[:, 0]))]

    def cross(o, a, b):
        return (a[0] - o[0]) * (b[1] - o[1]) - (a[1] - o[1]) * (b[0] - o[0])

    lower: List = []
    for q in p:
        while len(lower) >= 2 and cross(lower[-2], lower[-1], q) <= 0:
            lower.pop()
        lower.append(q)
    upper: List = []
    for q in p[::-1]:
        while len(upper) >= 2 and cross(upper[-2], upper[-1], q) <= 0:
            upper.pop()
        upper.append(q)
    return np.array(lower[:-1] + upper[:-1])


def min_area_rect_box(pts: np.ndarray) -> np.ndarray:
    """4 corners (float32, (4,2)) of the minimum-area enclosing rectangle of integer points — the composition
    ``cv2.boxPoints(cv2.minAreaRect(pts))`` used at marie/models/craft/craft_utils.py:79-80.

    Rotating calipers over the convex hull edges: for every hull edge direction the enclosing rectangle is
    computed in float32 and the smallest area wins (first minimum in hull order).  The rectangle itself is
    unique up to float rounding; the corner ORDER is normalised by the caller (clockwise from min x+y,
    craft_utils.py:91-94), so OpenCV's own corner order does not matter here.
    """
    hull = _convex_hull(np.asarray(pts, dtype=np.int64))
    n = len(hull)
    if n == 1:
        c = hull[0].astype(F32)
        return np.array([c, c, c, c], dtype=F32)
    if n == 2:
        a, b = hull[0].astype(F32), hull[1].astype(F32)
        return np.array([a, a, b, b], dtype=F32)
    hf = hull.astype(F32)
    best_area, best = None, None
    for i in range(n):
        e = hf[(i + 1) % n] - hf[i]
        ln = F32(math.sqrt(float(e[0]) * float(e[0]) + float(e[1]) * float(e[1])))
        ux, uy = F32(e[0] / ln), F32(e[1] / ln)          # edge direction
        vx, vy = F32(-uy), ux                             # normal
        pu = hf[:, 0] * ux + hf[:, 1] * uy
        pv = hf[:, 0] * vx + hf[:, 1] * vy
        u0, u1, v0, v1 = pu.min(), pu.max(), pv.min(), pv.max()
        area = F32((u1 - u0) * (v1 - v0))
        if best_area is None or area < best_area:
            best_area = area
            best = (ux, uy, vx, vy, u0, u1, v0, v1)
    ux, uy, vx, vy, u0, u1, v0, v1 = best
    corners = [(u0, v0), (u1, v0), (u1, v1), (u0, v1)]
    return np.array([[F32(a * ux + b * vx), F32(a * uy + b * vy)] for a, b in corners], dtype=F32)


def dilate_rect(seg: np.ndarray, k: int) -> np.ndarray:
    """``cv2.dilate(seg, getStructuringElement(MORPH_RECT, (k, k)))``: anchor at k//2, constant border that never
    wins the max (reference call site: craft_utils.py:73-74)."""
    if k <= 1:
        return seg.copy()
    h, w = seg.shape
    a = k // 2
    pad = np.zeros((h + k - 1, w + k - 1), dtype=seg.dtype)
    pad[a:a + h, a:a + w] = seg
    out = np.zeros_like(seg)
    for dy in range(k):
        for dx in range(k):
            # output (y,x) = max over src (y + dy - a, x + dx - a)
            out = np.maximum(out, pad[dy:dy + h, dx:dx + w])
    return out


def get_det_boxes(textmap: np.ndarray, linkmap: np.ndarray, text_threshold: float, link_threshold: float,
                  low_text: float):
    """``getDetBoxes_core`` — marie/models/craft/craft_utils.py:25-98 (minus its debug PNG writes)."""
    img_h, img_w = textmap.shape
    text_score = (textmap > F32(low_text)).astype(F32)            # cv2.threshold(..., 1, THRESH_BINARY)
    link_score = (linkmap > F32(link_threshold)).astype(F32)
    comb = np.clip(text_score + link_score, 0, 1).astype(np.uint8)
    n, labels, stats = connected_components(comb)
    det, mapper = [], []
    remove = np.logical_and(link_score == 1, text_score == 0)
    for k in range(1, n):
        x, y, w, h, size = (int(v) for v in stats[k])
        if size < 10:
            continue
        if textmap[labels == k].max() < text_threshold:
            continue
        segmap = np.zeros(textmap.shape, dtype=np.uint8)
        segmap[labels == k] = 255
        segmap[remove] = 0
        niter = int(math.sqrt(size * min(w, h) / (w * h)) * 2)
        sx, ex, sy, ey = x - niter, x + w + niter + 1, y - niter, y + h + niter + 1
        sx, sy = max(sx, 0), max(sy, 0)
        ex, ey = min(ex, img_w), min(ey, img_h)
        segmap[sy:ey, sx:ex] = dilate_rect(segmap[sy:ey, sx:ex], 1 + niter)
        ys, xs = np.nonzero(segmap)
        np_contours = np.stack([xs, ys], axis=1)
        if len(np_contours) == 0:
            # every pixel of the component was link-only; cv2.minAreaRect of an empty set is a zero box
            box = np.zeros((4, 2), dtype=F32)
        else:
            box = min_area_rect_box(np_contours)
            w_, h_ = np.linalg.norm(box[0] - box[1]), np.linalg.norm(box[1] - box[2])
            box_ratio = max(w_, h_) / (min(w_, h_) + 1e-5)
            if abs(1 - box_ratio) <= 0.1:
                l, r = xs.min(), xs.max()
                t, b = ys.min(), ys.max()
                box = np.array([[l, t], [r, t], [r, b], [l, b]], dtype=F32)
        startidx = box.sum(axis=1).argmin()
        box = np.roll(box, 4 - startidx, 0)
        det.append(np.array(box))
        mapper.append(k)
    return det, labels, mapper


def boxes_to_rects(boxes: List[np.ndarray], ratio: float, max_w: int, max_h: int) -> np.ndarray:
    """adjustResultCoordinates (craft_utils.py:268-274, ratio_net = 2) followed by the box processor's
    boundingRect + 2/4-pixel expansion (marie/boxes/craft_box_processor.py:499-520).  Returns (K,4) int32 xywh.
    ``cv2.boundingRect`` of int points: x = min, y = min, w = max-min+1, h = max-min+1."""
    out = []
    ratio_w = ratio_h = 1 / ratio
    for b in boxes:
        # in-place float32 *= (python floats): computed in float64, stored back as float32, then truncated
        poly = (np.array(b, dtype=F32) * np.array((ratio_w * 2, ratio_h * 2), dtype=np.float64)).astype(F32)
        region = poly.astype(np.int32).reshape(-1, 2)
        x0, y0 = region[:, 0].min(), region[:, 1].min()
        bw, bh = region[:, 0].max() - x0 + 1, region[:, 1].max() - y0 + 1
        out.append([max(0, x0 - 2), max(0, y0 - 2), min(max_w, bw + 4), min(max_h, bh + 4)])
    return np.array(out, dtype=np.int32).reshape(-1, 4)


def crop_fragments(img: np.ndarray, rects: np.ndarray) -> List[np.ndarray]:
    """``crop_poly_low`` on the expanded axis-aligned polygon (craft_box_processor.py:42-73,510-525): the filled
    mask covers the whole bounding rect of [x, x+w] x [y, y+h], i.e. the plain crop img[y:y+h+1, x:x+w+1]."""
    return [img[y:y + h + 1, x:x + w + 1].copy() for x, y, w, h in rects.tolist()]


def detect_page(img_u8: np.ndarray, st, text_threshold=0.7, link_threshold=0.45, low_text=0.3):
    """``BoxProcessorCraft.psm_sparse`` + box extraction (craft_box_processor.py:333-353,499-527) in one call."""
    h, w = img_u8.shape[:2]
    x, ratio, _ = craft_preprocess(img_u8, canvas_size=w, mag_ratio=1.0)
    y, _ = craft_forward(x, st)
    boxes, _, _ = get_det_boxes(y[0, :, :, 0], y[0, :, :, 1], text_threshold, link_threshold, low_text)
    rects = boxes_to_rects(boxes, ratio, w, h)
    # boxes lying entirely in the /32 canvas padding have no page pixels (the reference would raise inside cv2)
    rects = rects[(rects[:, 0] < w) & (rects[:, 1] < h)] if len(rects) else rects
    return rects, y
