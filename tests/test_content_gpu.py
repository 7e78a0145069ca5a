"""GPU: csrc/content_ops.hip (the chain behind crop_to_content / crop_to_content_box) bit-exact against oracle/content_ref.py,
and the two switches that use it: OcrEngine.extract(crop_to_content=True) and psm_sparse(bbox_optimization=True).
The oracle restates OpenCV's steps (not installed here, no reference fixture: parity unpinned — see its header)."""
import numpy as np
import pytest

from marie_icr_amd.weights import make_craft_state, make_crnn_state, make_page_bgr

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from marie_icr_amd._lib import Context

    c = Context(0)
    yield c
    c.close()


def _stroke_page(seed, h, w, blocks):
    rng = np.random.default_rng(seed)
    img = np.full((h, w, 3), 255, np.uint8)
    img -= rng.integers(0, 6, img.shape, dtype=np.uint8)
    for (y0, y1, x0, x1) in blocks:
        for x in range(x0, x1 - 2, 8):
            img[y0:y1, x:x + 3] = rng.integers(0, 60, (y1 - y0, 3, 1), dtype=np.uint8)
    return img


def _plain(x):
    if isinstance(x, dict):
        return {k: _plain(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_plain(v) for v in x]
    if isinstance(x, np.ndarray):
        return x.tolist()
    if isinstance(x, np.generic):
        return x.item()
    return x


def _extents(ctx, page, rects, aware):
    import torch

    from marie_icr_amd.content import content_extents

    dev = torch.from_numpy(np.ascontiguousarray(page)).cuda()
    return content_extents(ctx, dev.data_ptr(), page.shape[0], page.shape[1], rects, aware)


@pytest.mark.parametrize("aware", [True, False])
def test_extents_equal_the_oracle_on_many_rectangles(ctx, aware):
    from oracle import content_ref as R

    rng = np.random.default_rng(7)
    pages = [_stroke_page(1, 300, 420, [(40, 70, 50, 300), (120, 160, 200, 400), (220, 290, 10, 120)]),
             rng.integers(0, 256, (97, 131, 3), dtype=np.uint8),                                   # pure noise
             np.repeat(np.repeat(rng.integers(0, 256, (20, 30, 3), dtype=np.uint8), 5, 0), 5, 1)]   # blocky
    for page in pages:
        h, w = page.shape[:2]
        rects = [[0, 0, w, h], [0, 0, 1, 1], [w - 1, h - 1, 1, 1], [3, 5, 1, 7], [3, 5, 7, 1], [10, 10, 2, 2], [5, 5, 0, 9]]
        for _ in range(25):
            x, y = int(rng.integers(0, w - 1)), int(rng.integers(0, h - 1))
            rects.append([x, y, int(rng.integers(1, w - x + 1)), int(rng.integers(1, h - y + 1))])
        got = _extents(ctx, page, rects, aware)
        for r, g in zip(rects, got):
            x, y, rw, rh = r
            want = R.content_extent(page[y:y + rh, x:x + rw], aware) if rw and rh else (0, 0, 0, 0, 0)
            assert tuple(int(v) for v in g) == tuple(want), (r, aware)


def test_rejects_rectangles_outside_the_page(ctx):
    from marie_icr_amd._lib import MarieHipError

    page = np.full((20, 30, 3), 255, np.uint8)
    with pytest.raises(MarieHipError):
        _extents(ctx, page, [[25, 0, 10, 5]], True)


@pytest.mark.parametrize("aware", [True, False])
def test_crop_functions_follow_the_reference_rules(ctx, aware):
    from marie_icr_amd.content import crop_to_content, crop_to_content_box
    from oracle import content_ref as R

    page = _stroke_page(2, 400, 600, [(100, 140, 200, 450), (250, 300, 80, 260)])
    assert np.array_equal(crop_to_content(ctx, page, aware), R.crop_to_content(page, aware))
    snippet = page[90:150, 180:470]
    off, crop = crop_to_content_box(ctx, snippet, aware)
    roff, rcrop = R.crop_to_content_box(snippet, aware)
    assert off == roff and np.array_equal(crop, rcrop)
    white = np.full((50, 60, 3), 255, np.uint8)
    assert crop_to_content(ctx, white, aware) is white
    assert crop_to_content_box(ctx, white, aware)[0] == [0, 0, 0, 0]
    gray = R.bgr2gray(page)
    assert np.array_equal(crop_to_content(ctx, gray, aware), R.crop_to_content(gray, aware))     # a gray frame is taken as it is


def test_full_size_page_content_extent(ctx):
    from oracle import content_ref as R

    page = make_page_bgr(11, 3300, 2550)
    for aware in (True, False):
        got = _extents(ctx, page, [[0, 0, 2550, 3300]], aware)[0]
        assert tuple(int(v) for v in got) == tuple(R.content_extent(page, aware))


@pytest.mark.parametrize("aware", [True, False])
def test_optimize_boxes_is_the_reference_loop(ctx, aware):
    import torch

    from marie_icr_amd.content import optimize_boxes
    from oracle import content_ref as R

    page = _stroke_page(3, 300, 500, [(50, 80, 60, 260), (150, 200, 300, 480), (230, 260, 40, 120)])
    boxes = np.array([[40.7, 40.2, 280.9, 95.5], [290.1, 140.0, 499.6, 210.3], [20.0, 220.0, 140.0, 275.0], [400.0, 10.0, 460.0, 40.0],
                      [0.0, 0.0, 500.0, 300.0]], np.float32)
    dev = torch.from_numpy(page).cuda()
    got = optimize_boxes(ctx, dev.data_ptr(), 300, 500, list(boxes), aware)
    for box, g in zip(boxes, got):                                  # ulim_dit_box_processor.py:608-626
        b = np.array(box).astype(np.int32)
        x0, y0, x1, y1 = b
        snippet = page[y0:y0 + (y1 - y0), x0:x0 + (x1 - x0)]
        off, _ = R.crop_to_content_box(snippet, aware)
        want = [b[0] + off[0], b[1] + off[1], b[2] - (off[2] - off[0]), b[3] - (off[3] - off[1])]
        assert [int(v) for v in g] == [int(v) for v in want]


def test_engine_crop_to_content_switch(ctx):
    """extract(crop_to_content=True) = extract() of the cropped page on its 4-px white canvas (ocr_engine.py:169-184)."""
    from marie_icr_amd.box_processor import PSMode
    from marie_icr_amd.craft import BoxProcessorCraft
    from marie_icr_amd.crnn import CrnnOcrProcessor
    from marie_icr_amd.ocr_engine import CoordinateFormat, MarieHipOcrEngine
    from oracle import content_ref as R

    bp = BoxProcessorCraft(state=make_craft_state(5), precision="f32", ctx=ctx)
    rec = CrnnOcrProcessor(state=make_crnn_state(0), precision="f32", img_w=128, ctx=ctx)
    eng = MarieHipOcrEngine(box_processor=bp, default_ocr_processor=rec)
    page = np.full((300, 420, 3), 255, np.uint8)
    page[40:260, 120:330] = make_page_bgr(5, 220, 210)
    got = eng.extract([page], PSMode.SPARSE, CoordinateFormat.XYWH, crop_to_content=True)
    cropped = R.crop_to_content(page, True)
    assert cropped.shape[1] < page.shape[1]
    canvas = np.full((cropped.shape[0] + 8, cropped.shape[1] + 8, 3), 255, np.uint8)
    canvas[4:-4, 4:-4] = cropped
    want = eng.extract([canvas], PSMode.SPARSE, CoordinateFormat.XYWH)
    assert _plain(got) == _plain(want)
    assert got[0]["meta"]["imageSize"] == {"width": canvas.shape[1], "height": canvas.shape[0]}
