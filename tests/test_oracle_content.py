"""oracle/content_ref.py: the OpenCV steps it restates, against independent formulations (OpenCV itself is not installed here and
the reference holds no fixture for crop_to_content / crop_to_content_box: parity unpinned, see the oracle's header)."""
import numpy as np
import pytest

from oracle import content_ref as R


def _page(seed, h=120, w=200, ink=((30, 50, 60, 140),)):
    """White paper with a little noise and blocks of dark 3-px strokes every 8 px (text-like)."""
    rng = np.random.default_rng(seed)
    img = np.full((h, w, 3), 255, np.uint8)
    img -= rng.integers(0, 6, img.shape, dtype=np.uint8)
    for (y0, y1, x0, x1) in ink:
        for x in range(x0, x1 - 2, 8):
            img[y0:y1, x:x + 3] = rng.integers(0, 60, (y1 - y0, 3, 1), dtype=np.uint8)
    return img


def test_gray_weights_sum_to_one_and_match_float():
    rng = np.random.default_rng(0)
    f = rng.integers(0, 256, (50, 60, 3), dtype=np.uint8)
    g = R.bgr2gray(f)
    ref = 0.114 * f[..., 0] + 0.587 * f[..., 1] + 0.299 * f[..., 2]
    assert np.abs(g.astype(np.float64) - ref).max() <= 0.51
    v = np.full((4, 4, 3), 201, np.uint8)
    assert (R.bgr2gray(v) == 201).all()                           # 1868 + 9617 + 4899 = 2**14


@pytest.mark.parametrize("shape", [(1, 1), (1, 7), (2, 2), (3, 9), (40, 33)])
def test_gaussian_against_float_convolution(shape):
    rng = np.random.default_rng(1)
    g = rng.integers(0, 256, shape, dtype=np.uint8)
    k = np.array([1, 4, 6, 4, 1], np.float64) / 16
    h, w = shape
    pad = np.pad(g.astype(np.float64), ((2, 2), (2, 2)), mode="reflect") if min(h, w) > 2 else None
    if pad is None:                                               # np.pad's reflect needs pad < size: index by hand
        ys = [R._reflect101(np.array([y + d for d in range(-2, 3)]), h) for y in range(h)]
        xs = [R._reflect101(np.array([x + d for d in range(-2, 3)]), w) for x in range(w)]
        ref = np.array([[sum(k[a] * k[b] * g[ys[y][a], xs[x][b]] for a in range(5) for b in range(5)) for x in range(w)] for y in range(h)])
    else:
        ref = sum(k[a] * k[b] * pad[a:a + h, b:b + w] for a in range(5) for b in range(5))
    out = R.gaussian5(g).astype(np.float64)
    assert np.abs(out - np.floor(ref + 0.5)).max() == 0           # exact weights: round half up of the exact value


def test_divide_rounding_and_zero_divisor():
    a = np.array([[0, 1, 10, 255, 128, 3]], np.uint8)
    b = np.array([[0, 0, 20, 255, 255, 2]], np.uint8)
    assert R.divide255(a, b).tolist() == [[0, 0, 128, 255, 128, 255]]       # 127.5 -> 128 (even), 382.5 saturates


def test_otsu_is_the_first_maximum_of_the_between_class_variance():
    rng = np.random.default_rng(2)
    for _ in range(5):
        img = np.concatenate([rng.normal(60, 12, 3000), rng.normal(190, 20, 5000)]).clip(0, 255).astype(np.uint8)
        t = R.otsu_threshold(img)
        hist = np.bincount(img, minlength=256).astype(np.float64) / img.size
        best, arg = -1.0, 0
        for i in range(256):
            q1 = hist[:i + 1].sum()
            q2 = 1 - q1
            if q1 < 1e-9 or q2 < 1e-9:
                continue
            m1 = (np.arange(i + 1) * hist[:i + 1]).sum() / q1
            m2 = (np.arange(i + 1, 256) * hist[i + 1:]).sum() / q2
            s = q1 * q2 * (m1 - m2) ** 2
            if s > best + 1e-12:
                best, arg = s, i
        assert abs(t - arg) <= 1                                   # the running-mean form and the direct form agree
    assert R.otsu_threshold(np.full((10, 10), 77, np.uint8)) == 0      # one level: no split


def test_close_removes_single_holes_and_keeps_blocks():
    t = np.full((9, 9), 255, np.uint8)
    t[4, 4] = 0                                                   # a 1-pixel hole is closed
    t[1:4, 5:7] = 0                                               # a 2 (wide) x 3 (tall) block survives ...
    out = R._morph(R._morph(t, True), False)
    assert out[4, 4] == 255
    # ... one column to the right: OpenCV's dilate and erode both read src(x + x' - anchor.x, y + y' - anchor.y) (no reflected
    # element for the dilation), so closing with an even-width element and anchor (1, 1) moves what survives by one pixel
    assert (out[1:4, 6:8] == 0).all() and (out[1:4, 5] == 255).all()
    assert (out == 0).sum() == 6


def test_crop_to_content_page_rules():
    img = _page(3)
    out = R.crop_to_content(img, content_aware=True)
    assert out.shape[0] == img.shape[0]                           # content-aware keeps the full height
    xmin, ymin, xmax, ymax, n = R.content_extent(img, True)
    assert n > 0 and 58 <= xmin <= 62 and 132 <= xmax <= 141
    assert out.shape[1] == min(img.shape[1], xmax - max(0, xmin - 16) + 16) + 1
    plain = R.crop_to_content(img, content_aware=False)
    x0, y0, x1, y1, _ = R.content_extent(img, False)
    assert plain.shape[:2] == (y1 - y0 + 1, x1 - x0 + 1)
    white = np.full((40, 50, 3), 255, np.uint8)
    assert R.crop_to_content(white) is white                      # nothing found: the frame itself


def test_crop_to_content_box_offsets():
    img = _page(4, h=60, w=160, ink=((20, 40, 50, 110),))
    off, crop = R.crop_to_content_box(img, content_aware=False)
    x0, y0, x1, y1, _ = R.content_extent(img, False)
    assert off == [x0, y0, img.shape[1] - (x1 - x0), img.shape[0] - (y1 - y0)]
    assert crop.shape[:2] == (y1 - y0 + 1, x1 - x0 + 1)
    off_a, _ = R.crop_to_content_box(img, content_aware=True)
    assert off_a[0] <= x0 + 2 and off_a[1] <= y0 + 2
    assert R.crop_to_content_box(np.full((8, 8, 3), 255, np.uint8))[0] == [0, 0, 0, 0]
