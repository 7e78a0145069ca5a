"""Page ingest on the GPU: the HIP INTER_AREA kernel against the oracle (bit-exact), ensure_max_page_size end to end, the
pinned-memory page feeder."""
import ctypes as C

import numpy as np
import pytest

from marie_icr_amd import ingest
from marie_icr_amd._lib import Context, MarieHipError, check
from oracle import ingest_ref

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("shape,new_wh", [
    ((97, 131, 3), (120, 90)),        # fractional scales, RGB
    ((97, 131), (64, 95)),            # gray, different scales per axis
    ((128, 96, 3), (48, 64)),         # 2 x 2 blocks
    ((90, 120), (40, 30)),            # 3 x 3 blocks
    ((60, 80, 3), (20, 30)),          # 4 x 2 blocks
    ((64, 64, 3), (64, 64)),          # identity
    ((33, 70, 3), (1, 1)),            # everything into one pixel
    ((500, 7, 3), (7, 123)),          # x kept, y shrunk
])
def test_resize_area_bit_exact(ctx, shape, new_wh):
    img = np.random.default_rng(sum(shape)).integers(0, 256, shape, dtype=np.uint8)
    got = ingest.resize_area(ctx, img, *new_wh)
    ref = ingest_ref.resize_area(img, *new_wh)
    assert got.shape == ref.shape
    assert np.array_equal(got, ref), int(np.abs(got.astype(int) - ref.astype(int)).max())


def test_full_page_clamp_bit_exact(ctx):
    page = np.random.default_rng(5).integers(0, 256, (3200, 2600), dtype=np.uint8)      # the reference test's frame
    changed, frames = ingest.ensure_max_page_size([page], expand_ratio=0, ctx=ctx)
    assert changed is True and frames[0].shape == (3138, 2550)                            # test_image_resizing.py:18-26
    ref_changed, ref = ingest_ref.ensure_max_page_size([page], expand_ratio=0)
    assert ref_changed and np.array_equal(frames[0], ref[0])
    changed, frames = ingest.ensure_max_page_size([page], ctx=ctx)                        # with the default expansion: kept
    assert changed is False and frames[0] is page
    rgb = np.random.default_rng(6).integers(0, 256, (5100, 6600, 3), dtype=np.uint8)[:1000]     # landscape strip
    ch, out = ingest.ensure_max_page_size([rgb, page], ctx=ctx)
    rch, rout = ingest_ref.ensure_max_page_size([rgb, page])
    assert ch is rch is True and out[1] is page and np.array_equal(out[0], rout[0])


def test_resize_area_rejects_enlarging(ctx):
    img = np.zeros((10, 10, 3), np.uint8)
    with pytest.raises(MarieHipError):
        ingest.resize_area(ctx, img, 12, 8)
    with pytest.raises(ValueError):
        ingest.resize_area(ctx, np.zeros((4, 4, 2), np.uint8), 2, 2)


def test_page_feeder_overlaps_and_preserves_bytes(ctx):
    import torch

    rng = np.random.default_rng(9)
    H, W = 320, 260
    batches = [[rng.integers(0, 256, (H, W, 3), dtype=np.uint8) for _ in range(n)] for n in (3, 1, 4, 2, 3)]
    stream = torch.cuda.Stream()
    ctx.set_stream(stream.cuda_stream)
    feeder = ingest.PageFeeder(batches, capacity_bytes=4 * H * W * 3, consumer_stream=stream)
    outs = []
    with torch.cuda.stream(stream):
        for bi, (ptr, shape) in enumerate(feeder):
            assert shape == (len(batches[bi]), H, W, 3)
            for i in range(shape[0]):       # consumer: shrink every page of the batch on the consumer stream, device to device
                dst = torch.empty((H // 2, W // 2, 3), dtype=torch.uint8, device="cuda")
                check(ctx.h, ctx.lib.mhip_resize_area_u8(ctx.h, C.c_void_p(ptr + i * H * W * 3), H, W, 3, W * 3,
                                                         C.c_void_p(dst.data_ptr()), H // 2, W // 2), "mhip_resize_area_u8")
                outs.append((bi, i, dst))
    stream.synchronize()
    ctx.set_stream(None)
    assert feeder.bytes_moved == sum(len(b) for b in batches) * H * W * 3
    for bi, i, dst in outs:
        assert np.array_equal(dst.cpu().numpy(), ingest_ref.resize_area(batches[bi][i], W // 2, H // 2)), (bi, i)


@pytest.mark.parametrize("shape,new_wh", [
    ((97, 131, 3), (120, 90)),        # mild shrink
    ((300, 100, 3), (53, 160)),       # resize_image's case: ratio 0.533
    ((64, 64), (23, 17)),             # gray, strong shrink (taps skip pixels, as in OpenCV)
    ((40, 50, 3), (125, 100)),        # enlarge 2.5 x: borders replicate
    ((31, 33, 3), (33, 31)),          # identity: weights (0, 2048, 0, 0)
    ((5, 3), (1, 1)),
])
def test_resize_cubic_bit_exact(ctx, shape, new_wh):
    img = np.random.default_rng(sum(shape) + 1).integers(0, 256, shape, dtype=np.uint8)
    src = np.ascontiguousarray(img)
    cn = 1 if src.ndim == 2 else 3
    out = np.empty((new_wh[1], new_wh[0]) + (() if src.ndim == 2 else (3,)), np.uint8)
    check(ctx.h, ctx.lib.mhip_resize_cubic_u8_host(ctx.h, src.ctypes.data_as(C.c_void_p), src.shape[0], src.shape[1], cn,
                                                   out.ctypes.data_as(C.c_void_p), out.shape[0], out.shape[1]),
          "mhip_resize_cubic_u8_host")
    ref = ingest_ref.resize_cubic(img, *new_wh)
    assert np.array_equal(out, ref), int(np.abs(out.astype(int) - ref.astype(int)).max())
    if shape[:2] == (new_wh[1], new_wh[0]):
        assert np.array_equal(out, img)
