"""Pin the crop-batcher restatement (oracle/pil_resample.py) against Pillow itself."""
import numpy as np
import pytest

from oracle import pil_resample as pr


def _rand_crop(rng, h, w):
    base = rng.integers(0, 256, size=(h // 3 + 1, w // 3 + 1, 3)).astype(np.float32)
    up = np.repeat(np.repeat(base, 3, 0), 3, 1)[:h, :w]
    return np.clip(up + rng.integers(-20, 21, size=(h, w, 3)), 0, 255).astype(np.uint8)


def test_gray_conversion_matches_pil():
    from PIL import Image

    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, size=(37, 53, 3)).astype(np.uint8)
    ref = np.asarray(Image.fromarray(np.ascontiguousarray(img[:, :, ::-1])).convert("L"))
    assert np.array_equal(pr.bgr_to_l(img), ref)


@pytest.mark.parametrize("h,w,ow,oh", [(40, 100, 80, 32), (20, 31, 50, 32), (64, 900, 256, 32), (32, 77, 77, 32),
                                       (33, 8, 8, 32), (5, 5, 32, 32), (200, 40, 7, 32), (32, 256, 256, 32)])
def test_resize_matches_pil(h, w, ow, oh):
    from PIL import Image

    rng = np.random.default_rng(h * 1000 + w)
    g = rng.integers(0, 256, size=(h, w)).astype(np.uint8)
    ref = np.asarray(Image.fromarray(g).resize((ow, oh), Image.BICUBIC))
    got = pr.resize_bicubic_l(g, ow, oh)
    assert np.array_equal(got, ref)


def test_align_collate_matches_pil():
    rng = np.random.default_rng(5)
    crops = [_rand_crop(rng, int(h), int(w)) for h, w in
             [(38, 120), (25, 60), (48, 700), (32, 256), (17, 9), (90, 30), (41, 333)]]
    for img_w in (256, 100):
        assert np.array_equal(pr.align_collate_u8(crops, img_w), pr.align_collate_pil(crops, img_w))
