"""Page ingest on the CPU: the oracle's size rule against the reference's own test cases, the library's host-side rule
against the oracle, properties of the restated INTER_AREA resampler, and the TIFF burst of ``load_image``."""
import numpy as np
import pytest

from marie_icr_amd import ingest
from oracle import ingest_ref

# tests/imaging/test_image_resizing.py:7-34 of the reference: (frame shape (h, w), expand_ratio or None = default, changed,
# expected shape).  The fourth case of that file (:37-44: 4171 x 2569 -> (3200, 2600), changed) contradicts the reference's
# own code (image_utils.py:275-310 yields (3795, 2337)) and is checked below as "stale".
REFERENCE_CASES = [
    ((3200, 2550), 0, False, (3200, 2550)),
    ((3200, 2600), 0, True, (3138, 2550)),
    ((3200, 2600), None, False, (3200, 2600)),
]


@pytest.mark.parametrize("shape,ratio,changed,expect", REFERENCE_CASES)
def test_size_rule_matches_reference_tests(shape, ratio, changed, expect):
    kw = {} if ratio is None else {"expand_ratio": ratio}
    ch, nw, nh = ingest_ref.max_page_size(shape[1], shape[0], **kw)
    assert ch is changed and (nh, nw) == expect
    ch2, nw2, nh2 = ingest.page_size_rule(shape[1], shape[0], **kw)
    assert ch2 is changed and (nh2, nw2) == expect


def test_stale_reference_case_follows_the_reference_code():
    # image_utils.py: max = (2550 + 382, 3300 + 495); 4171 > 3795 -> new_h 3795, new_w int(3795 * 2569 / 4171) = 2337
    assert ingest_ref.max_page_size(2569, 4171) == (True, 2337, 3795)
    assert ingest.page_size_rule(2569, 4171) == (True, 2337, 3795)


def test_library_rule_equals_oracle_on_random_sizes():
    rng = np.random.default_rng(7)
    for _ in range(3000):
        w, h = int(rng.integers(1, 9000)), int(rng.integers(1, 9000))
        mps = (int(rng.integers(100, 4000)), int(rng.integers(100, 4000)))
        ratio = float(rng.choice([0.0, 0.15, 0.05, 0.3]))
        assert ingest.page_size_rule(w, h, mps, ratio) == ingest_ref.max_page_size(w, h, mps, ratio), (w, h, mps, ratio)


def test_area_resampler_properties():
    rng = np.random.default_rng(1)
    const = np.full((37, 53, 3), 201, np.uint8)
    assert (ingest_ref.resize_area(const, 41, 30) == 201).all()              # weights of a cell sum to 1
    img = rng.integers(0, 256, (48, 60, 3), dtype=np.uint8)
    assert np.array_equal(ingest_ref.resize_area(img, 60, 48), img)           # scale 1: identity
    two = ingest_ref.resize_area(img, 30, 24)                                 # 2 x 2 blocks: (sum + 2) >> 2
    ref2 = (img.astype(np.int64).reshape(24, 2, 30, 2, 3).sum(axis=(1, 3)) + 2) >> 2
    assert np.array_equal(two, ref2.astype(np.uint8))
    three = ingest_ref.resize_area(img, 20, 16)                               # 3 x 3 blocks: mean, ties to even
    mean3 = img.astype(np.float64).reshape(16, 3, 20, 3, 3).sum(axis=(1, 3)) / 9.0
    assert np.abs(three.astype(np.float64) - mean3).max() <= 0.5 + 1e-4
    frac = ingest_ref.resize_area(img[:, :, 0], 47, 37)                       # fractional scale, gray
    assert frac.shape == (37, 47) and abs(float(frac.mean()) - float(img[:, :, 0].mean())) < 1.0
    # a smooth ramp stays within one grey level of the exact box integral
    ramp = np.clip(np.add.outer(np.arange(64) * 2.0, np.arange(80) * 1.5), 0, 255).astype(np.uint8)
    small = ingest_ref.resize_area(ramp, 50, 40).astype(np.float64)
    assert np.all(np.diff(small, axis=1) >= 0) and np.all(np.diff(small, axis=0) >= 0)


def test_ensure_max_page_size_keeps_small_frames(monkeypatch):
    frames = [np.zeros((3200, 2550), np.uint8), np.zeros((100, 80, 3), np.uint8)]
    changed, out = ingest.ensure_max_page_size(frames, expand_ratio=0)        # nothing to resize: no GPU touched
    assert changed is False and out[0] is frames[0] and out[1] is frames[1]
    ch, oracle_out = ingest_ref.ensure_max_page_size(frames, expand_ratio=0)
    assert ch is False and oracle_out[0] is frames[0]


def test_load_image_bursts_tiff(tmp_path):
    from PIL import Image

    rng = np.random.default_rng(3)
    pages = [rng.integers(0, 256, (40, 30), dtype=np.uint8), rng.integers(0, 256, (50, 20, 3), dtype=np.uint8),
             (rng.integers(0, 2, (25, 35)) * 255).astype(np.uint8)]
    ims = [Image.fromarray(pages[0]), Image.fromarray(pages[1]), Image.fromarray(pages[2]).convert("1")]
    path = tmp_path / "doc.tif"
    ims[0].save(path, save_all=True, append_images=ims[1:])
    loaded, frames = ingest.load_image(str(path))
    assert loaded and len(frames) == 3
    assert all(f.dtype == np.uint8 and f.ndim == 3 and f.shape[2] == 3 for f in frames)
    assert np.array_equal(frames[0][:, :, 1], pages[0]) and np.array_equal(frames[1], pages[1])
    assert np.array_equal(frames[2][:, :, 0], pages[2])
    png = tmp_path / "one.png"
    Image.fromarray(pages[1]).save(png)
    assert np.array_equal(ingest.frames_from_file(str(png))[0], pages[1])
    with pytest.raises(FileNotFoundError):
        ingest.frames_from_file(str(tmp_path / "missing.png"))
    assert ingest.load_image(None) == (False, None)
    with pytest.raises(NotImplementedError):
        ingest.load_image("x.pdf")


def test_cubic_resampler_properties():
    rng = np.random.default_rng(2)
    img = rng.integers(0, 256, (40, 56, 3), dtype=np.uint8)
    assert np.array_equal(ingest_ref.resize_cubic(img, 56, 40), img)                    # scale 1: taps (0, 2048, 0, 0)
    const = np.full((30, 44), 77, np.uint8)
    assert (ingest_ref.resize_cubic(const, 17, 11) == 77).all() and (ingest_ref.resize_cubic(const, 90, 61) == 77).all()
    _, coef = ingest_ref._cubic_tab(97, 1.0 / (97 / 131))
    assert np.all(np.abs(coef.sum(axis=1) - 2048) <= 2)                                 # weights sum to ~1 in 11-bit fixed point
    ramp = np.clip(np.add.outer(np.arange(60) * 3, np.arange(80) * 1), 0, 255).astype(np.uint8)
    small = ingest_ref.resize_cubic(ramp, 40, 30).astype(int)
    exact = np.add.outer(((np.arange(30) + 0.5) * 2 - 0.5) * 3, ((np.arange(40) + 0.5) * 2 - 0.5) * 1)
    inner = (slice(2, -2), slice(2, -2))
    assert np.abs(small[inner] - exact[inner]).max() <= 1.0                             # a cubic reproduces linear ramps


def test_resize_image_oracle_branches():
    img = np.random.default_rng(4).integers(0, 256, (100, 120, 3), dtype=np.uint8)
    framed, coord = ingest_ref.resize_image(img, (160, 160), keep_max_size=True)
    assert framed.shape == (160, 160, 3) and coord == (20, 30, 120, 100) and np.array_equal(framed[30:130, 20:140], img)
    same, c0 = ingest_ref.resize_image(img, (100, 120))
    assert same is img and c0 == (0, 0, 120, 100)
    tall = np.random.default_rng(5).integers(0, 256, (300, 100, 3), dtype=np.uint8)
    f3, c3 = ingest_ref.resize_image(tall, (160, 160), keep_max_size=True)
    assert f3.shape == (160, 160, 3) and c3 == (53, 0, 53, 160) and (f3[:, :53] == 255).all()
