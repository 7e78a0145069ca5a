"""CPU restatement of the reference's content cropping — TEST INFRASTRUCTURE (only tests/ may import it).

Follows ``marie/utils/image_utils.py:190-252`` (``crop_to_content``, the ``crop_to_content=True`` kwarg of
``OcrEngine.extract``, ``marie/ocr/ocr_engine.py:169-176``) and ``marie/boxes/dit/ulim_dit_box_processor.py:291-352``
(``crop_to_content_box``, the ``bbox_optimization`` option of ``psm_sparse``, ``:608-626``).  Both run the same OpenCV chain:

    gray = cvtColor(BGR2GRAY); content-aware: blur = GaussianBlur(gray, (5, 5), 0); divide = cv2.divide(gray, blur, scale=255);
    Otsu threshold; morphologyEx(MORPH_CLOSE, 2 x 3 rectangle) — else: Otsu threshold of gray; then the extent of the zero pixels.

**Parity unpinned**: OpenCV (opencv-python 4.8.1.78 in the reference's requirements) is not installed here and the reference holds
no fixture for these functions, so the OpenCV steps are restated from its published algorithms:
  * BGR2GRAY 8-bit: (1868 B + 9617 G + 4899 R + 8192) >> 14;
  * GaussianBlur 5 x 5, sigma 0, 8-bit: the fixed kernel [1 4 6 4 1] / 16 in both directions, BORDER_REFLECT_101, fixed-point
    arithmetic that is exact for these weights, i.e. (sum of w_i w_j p + 128) >> 8;
  * cv2.divide 8-bit with a scale: saturate(round-half-even(a * scale / b)), 0 where b == 0 (float32 arithmetic);
  * Otsu: getThreshVal_Otsu_8u's double-precision scan (first maximum of the between-class variance, FLT_EPSILON guards);
    THRESH_BINARY: 255 where the pixel is > threshold;
  * MORPH_CLOSE = dilate then erode with the 2 (wide) x 3 (tall) rectangle, anchor (1, 1): columns x - 1 .. x, rows y - 1 .. y + 1,
    neighbours outside the image ignored.
"""
import numpy as np

FLT_EPSILON = 1.1920928955078125e-07


def bgr2gray(frame: np.ndarray) -> np.ndarray:
    f = frame.astype(np.int64)
    return ((f[..., 0] * 1868 + f[..., 1] * 9617 + f[..., 2] * 4899 + 8192) >> 14).astype(np.uint8)


def _reflect101(i: np.ndarray, n: int) -> np.ndarray:
    if n == 1:
        return np.zeros_like(i)
    p = 2 * (n - 1)
    i = np.mod(i, p)
    return np.where(i >= n, p - i, i)


def gaussian5(gray: np.ndarray) -> np.ndarray:
    h, w = gray.shape
    wts = np.array([1, 4, 6, 4, 1], np.int64)
    g = gray.astype(np.int64)
    xs = _reflect101(np.arange(w)[None, :] + np.arange(-2, 3)[:, None], w)        # [5][w]
    hp = sum(wts[k] * g[:, xs[k]] for k in range(5))                              # [h][w]
    ys = _reflect101(np.arange(h)[None, :] + np.arange(-2, 3)[:, None], h)
    vp = sum(wts[k] * hp[ys[k], :] for k in range(5))
    return ((vp + 128) >> 8).astype(np.uint8)


def divide255(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    af, bf = a.astype(np.float32), b.astype(np.float32)
    with np.errstate(divide="ignore", invalid="ignore"):
        q = (af * np.float32(255.0)) / bf
    q = np.where(b == 0, np.float32(0), q)
    return np.clip(np.rint(q), 0, 255).astype(np.uint8)


def otsu_threshold(img: np.ndarray) -> int:
    hist = np.bincount(img.reshape(-1), minlength=256).astype(np.float64)
    n = img.size
    scale = 1.0 / n
    mu = float((np.arange(256) * hist).sum()) * scale
    mu1 = q1 = 0.0
    max_sigma, max_val = 0.0, 0
    for i in range(256):
        p_i = hist[i] * scale
        mu1 *= q1
        q1 += p_i
        q2 = 1.0 - q1
        if min(q1, q2) < FLT_EPSILON or max(q1, q2) > 1.0 - FLT_EPSILON:
            continue
        mu1 = (mu1 + i * p_i) / q1
        mu2 = (mu - q1 * mu1) / q2
        sigma = q1 * q2 * (mu1 - mu2) * (mu1 - mu2)
        if sigma > max_sigma:
            max_sigma, max_val = sigma, i
    return int(max_val)


def _morph(img: np.ndarray, dilate: bool) -> np.ndarray:
    h, w = img.shape
    fill = 0 if dilate else 255
    pad = np.full((h + 2, w + 1), fill, np.uint8)
    pad[1:h + 1, 1:w + 1] = img
    views = [pad[1 + dy:1 + dy + h, 1 + dx:1 + dx + w] for dy in (-1, 0, 1) for dx in (-1, 0)]
    return np.maximum.reduce(views) if dilate else np.minimum.reduce(views)


def op_frame(frame: np.ndarray, content_aware: bool) -> np.ndarray:
    gray = bgr2gray(frame) if frame.ndim == 3 else frame
    if content_aware:
        d = divide255(gray, gaussian5(gray))
        t = np.where(d > otsu_threshold(d), 255, 0).astype(np.uint8)
        return _morph(_morph(t, True), False)
    return np.where(gray > otsu_threshold(gray), 255, 0).astype(np.uint8)


def content_extent(frame: np.ndarray, content_aware: bool):
    """(xmin, ymin, xmax, ymax, count) of the zero pixels of the processed frame; count 0 -> the rest is meaningless."""
    ys, xs = np.nonzero(op_frame(frame, content_aware) == 0)
    if len(xs) == 0:
        return 0, 0, 0, 0, 0
    return int(xs.min()), int(ys.min()), int(xs.max()), int(ys.max()), int(len(xs))


def crop_to_content(frame: np.ndarray, content_aware: bool = True) -> np.ndarray:
    """image_utils.py:190-252."""
    xmin, ymin, xmax, ymax, n = content_extent(frame, content_aware)
    if n == 0:
        return frame
    img_h, img_w = frame.shape[:2]
    if content_aware:
        x = max(0, xmin - 16)
        y, h = 0, img_h
        w = min(img_w, xmax - x + 16)
    else:
        x, y, h, w = xmin, ymin, ymax - ymin, xmax - xmin
    return frame[y:y + h + 1, x:x + w + 1].copy()


def crop_to_content_box(frame: np.ndarray, content_aware: bool = False):
    """ulim_dit_box_processor.py:291-352 -> (offset [left, top, img_w - w, img_h - h], cropped)."""
    xmin, ymin, xmax, ymax, n = content_extent(frame, content_aware)
    if n == 0:
        return [0, 0, 0, 0], frame
    img_h, img_w = frame.shape[:2]
    if content_aware:
        x = max(0, xmin - 1)
        y = max(0, ymin - 1)
        h = min(img_h, ymax - y + 1)
        w = min(img_w, xmax - x + 1)
    else:
        x, y, h, w = xmin, ymin, ymax - ymin, xmax - xmin
    return [x, y, img_w - w, img_h - h], frame[y:y + h + 1, x:x + w + 1].copy()
