"""Pin / cross-check the CRAFT CPU oracle (oracle/craft_ref.py)."""
import os

import numpy as np
import pytest

from marie_icr_amd.weights import make_craft_state, make_page_bgr, state_checksum
from oracle import craft_ref

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("tag", ["a", "b"])
def test_forward_matches_reference_craft_class(tag):
    g = np.load(os.path.join(GOLD, f"craft_net_{tag}.npz"))
    st = make_craft_state(int(g["weight_seed"]))
    assert state_checksum(st) == str(g["weight_sha256"])
    h, w = (int(v) for v in g["page_hw"])
    page = make_page_bgr(int(g["page_seed"]), h, w)
    x, ratio, _ = craft_ref.craft_preprocess(page, canvas_size=w)
    assert tuple(x.shape) == tuple(g["x_shape"])
    y, feat = craft_ref.craft_forward(x, st)
    np.testing.assert_allclose(y, g["y"], atol=2e-4, rtol=0)
    np.testing.assert_allclose(feat[:, ::4, ::3, ::3], g["feature_sub"], atol=2e-4, rtol=0)


def test_resize_identity_and_bounds():
    img = make_page_bgr(3, 60, 80)
    assert np.array_equal(craft_ref.cv_resize_linear_u8(img, 80, 60), img)
    out = craft_ref.cv_resize_linear_u8(img, 61, 47)
    assert out.shape == (47, 61, 3) and out.dtype == np.uint8
    # a constant image stays constant under fixed-point bilinear
    const = np.full((33, 41, 3), 137, np.uint8)
    assert (craft_ref.cv_resize_linear_u8(const, 29, 20) == 137).all()
    # against float bilinear (half-pixel centres): within 1 grey level
    yy = (np.arange(47) + 0.5) * 60 / 47 - 0.5
    xx = (np.arange(61) + 0.5) * 80 / 61 - 0.5
    y0 = np.clip(np.floor(yy).astype(int), 0, 59); y1 = np.clip(y0 + 1, 0, 59); fy = np.clip(yy - np.floor(yy), 0, 1)
    x0 = np.clip(np.floor(xx).astype(int), 0, 79); x1 = np.clip(x0 + 1, 0, 79); fx = np.clip(xx - np.floor(xx), 0, 1)
    fy[yy < 0] = 0; fx[xx < 0] = 0
    f = img.astype(np.float64)
    top = f[y0][:, x0] * (1 - fx)[None, :, None] + f[y0][:, x1] * fx[None, :, None]
    bot = f[y1][:, x0] * (1 - fx)[None, :, None] + f[y1][:, x1] * fx[None, :, None]
    ref = top * (1 - fy)[:, None, None] + bot * fy[:, None, None]
    assert np.abs(out.astype(np.float64) - ref).max() <= 1.01


def test_connected_components_vs_scipy():
    from scipy import ndimage

    rng = np.random.default_rng(0)
    mask = (rng.random((70, 93)) > 0.55).astype(np.uint8)
    n, labels, stats = craft_ref.connected_components(mask)
    ref, nref = ndimage.label(mask, structure=[[0, 1, 0], [1, 1, 1], [0, 1, 0]])
    assert n == nref + 1
    # same partition
    pairs = set(zip(labels[mask > 0].tolist(), ref[mask > 0].tolist()))
    assert len(pairs) == nref
    # labels are numbered in raster order of each component's first pixel
    first = [np.flatnonzero(labels.reshape(-1) == k)[0] for k in range(1, n)]
    assert first == sorted(first)
    for k in range(1, n):
        ys, xs = np.nonzero(labels == k)
        assert tuple(stats[k]) == (xs.min(), ys.min(), xs.max() - xs.min() + 1, ys.max() - ys.min() + 1, len(xs))


def test_min_area_rect_properties():
    rng = np.random.default_rng(1)
    # axis-aligned blob -> its bounding box
    ys, xs = np.mgrid[5:12, 20:51]
    pts = np.stack([xs.ravel(), ys.ravel()], 1)
    box = craft_ref.min_area_rect_box(pts)
    assert sorted(map(tuple, np.round(box).astype(int).tolist())) == [(20, 5), (20, 11), (50, 5), (50, 11)]
    # rotated cloud: every point inside, area <= axis-aligned bbox area and <= any sampled orientation
    ang = 0.5
    base = rng.integers(0, 40, size=(200, 2)) * np.array([1.0, 0.25])
    rot = np.array([[np.cos(ang), -np.sin(ang)], [np.sin(ang), np.cos(ang)]])
    pts = np.round(base @ rot.T).astype(np.int64) + 50
    box = craft_ref.min_area_rect_box(pts).astype(np.float64)
    e0, e1 = box[1] - box[0], box[3] - box[0]
    area = np.linalg.norm(e0) * np.linalg.norm(e1)
    bw = np.ptp(pts[:, 0]); bh = np.ptp(pts[:, 1])
    assert area <= bw * bh + 1e-3
    u0, u1 = e0 / np.linalg.norm(e0), e1 / np.linalg.norm(e1)
    rel = pts - box[0]
    assert (rel @ u0 >= -1e-3).all() and (rel @ u0 <= np.linalg.norm(e0) + 1e-3).all()
    assert (rel @ u1 >= -1e-3).all() and (rel @ u1 <= np.linalg.norm(e1) + 1e-3).all()
    for a in np.linspace(0, np.pi / 2, 181):
        d = np.array([np.cos(a), np.sin(a)]); nrm = np.array([-d[1], d[0]])
        assert area <= np.ptp(pts @ d) * np.ptp(pts @ nrm) + 1e-2


def test_get_det_boxes_on_synthetic_maps():
    text = np.full((60, 120), -1.0, np.float32)
    link = np.full((60, 120), -1.0, np.float32)
    text[10:20, 10:40] = 0.9           # word 1
    text[10:20, 46:80] = 0.8           # word 2, linked to word 1
    link[12:18, 38:48] = 0.9
    text[40:48, 10:30] = 0.5           # weak word: max < text_threshold -> dropped
    text[40:43, 60:62] = 0.9           # 6 px: area < 10 -> dropped
    boxes, labels, mapper = craft_ref.get_det_boxes(text, link, 0.7, 0.45, 0.3)
    assert len(boxes) == 1
    b = boxes[0]
    assert b.shape == (4, 2)
    # clockwise from the top-left corner
    assert b[0].sum() == b.sum(axis=1).min()
    l, t, r, bt = b[:, 0].min(), b[:, 1].min(), b[:, 0].max(), b[:, 1].max()
    assert l <= 10 and r >= 79 and t <= 10 and bt >= 19 and (r - l) < 90 and (bt - t) < 30
    rects = craft_ref.boxes_to_rects(boxes, 0.5, 1000, 1000)
    assert rects.shape == (1, 4) and (rects[:, 2:] > 0).all()


def test_dilate_rect_matches_bruteforce():
    rng = np.random.default_rng(2)
    seg = ((rng.random((17, 23)) > 0.85) * 255).astype(np.uint8)
    for k in (1, 2, 3, 4, 5):
        out = craft_ref.dilate_rect(seg, k)
        a = k // 2
        ref = np.zeros_like(seg)
        for y in range(17):
            for x in range(23):
                y0, y1 = max(0, y - a), min(17, y - a + k)
                x0, x1 = max(0, x - a), min(23, x - a + k)
                ref[y, x] = seg[y0:y1, x0:x1].max() if y1 > y0 and x1 > x0 else 0
        assert np.array_equal(out, ref), k
