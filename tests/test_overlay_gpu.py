"""GPU parity of the overlay cleaner through the C ABI: the generator against goldens written by the
reference's own LocalEnhancer (tests/golden/overlay_*.npz) and against the pinned oracle on a ragged page; the tensor -> image
conversion and blend_to_text bit-exact against the oracle's restatement (OpenCV pixel formulas: parity unpinned)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def ctx():
    from marie_icr_amd._lib import Context

    c = Context(0)
    yield c
    c.close()


def _golden(tag):
    from marie_icr_amd.weights import make_image_u8, make_overlay_state

    g = np.load(os.path.join(GOLD, f"overlay_{tag}.npz"))
    st = make_overlay_state(int(g["weight_seed"]), int(g["ngf"]))
    h, w = (int(v) for v in g["hw"])
    rgb = make_image_u8(int(g["image_seed"]), 1, h, w)[0]
    return st, int(g["ngf"]), np.ascontiguousarray(rgb[:, :, ::-1]), g["out"]        # the product takes BGR frames


def test_generator_fp32_matches_reference_goldens(ctx):
    from marie_icr_amd._lib import PREC_F32
    from marie_icr_amd.overlay import OverlayModel

    st, ngf, bgr, ref = _golden("ngf32")
    m = OverlayModel(ctx, st, ngf, PREC_F32)
    fake, raw = m.forward_host(bgr, want_raw=True)
    m.close()
    assert raw.shape == ref.shape
    err = float(np.abs(raw - ref).max())
    assert err <= 1e-3, err                                  # fp32 bar (north_star: logits within 1e-3)
    want = ((ref + 1) / 2.0 * 255.0).astype(np.uint8)        # tensor2im
    assert (np.abs(fake.astype(np.int32) - want.astype(np.int32)) <= 1).all() and (fake == want).mean() >= 0.98


def test_generator_f16_matches_reference_goldens(ctx):
    from marie_icr_amd._lib import PREC_F16, PREC_F32
    from marie_icr_amd.overlay import OverlayModel

    st, ngf, bgr, ref = _golden("ngf64")
    for prec, tol in ((PREC_F32, 1e-3), (PREC_F16, 0.06)):
        m = OverlayModel(ctx, st, ngf, prec)
        fake, raw = m.forward_host(bgr, want_raw=True)
        m.close()
        err = float(np.abs(raw - ref).max())
        assert err <= tol, (prec, err)
        assert np.abs(raw - ref).mean() <= tol / 8


def test_processor_ragged_page_vs_oracle(ctx):
    """A page whose sides are not multiples of 32 through the OverlayProcessor surface (white canvas, crop back) in fp32, against
    oracle/overlay_torch.segment_frame; the blend kernel bit-exact on the same generator image."""
    from marie_icr_amd.overlay import OverlayProcessor
    from marie_icr_amd.weights import make_image_u8, make_overlay_state
    from oracle.overlay_torch import TorchOverlayOracle, blend_to_text, preprocess, segment_frame

    st = make_overlay_state(2, 32)
    frame = make_image_u8(9, 1, 75, 100)[0]
    frame[20:30, 10:60] = (30, 30, 250)                       # a saturated patch: the HSV range test has something to cut
    p = OverlayProcessor("/tmp/overlay", state=st, ngf=32, precision="f32", ctx=ctx)
    src, mask, blended = p.segment_frame("doc", frame)
    o = TorchOverlayOracle(st)
    osrc, omask, oblended = segment_frame(o, frame)
    assert src is frame and mask.shape == omask.shape == frame.shape and blended.shape == frame.shape
    d = np.abs(mask.astype(np.int32) - omask.astype(np.int32))
    assert d.max() <= 1 and (d == 0).mean() >= 0.98           # the u8 truncation can flip where (t + 1) * 127.5 sits on an integer
    # blend: bit-exact given the SAME generator image (padded canvas)
    real = preprocess(frame)
    assert real.shape == (96, 128, 3)
    rng = np.random.default_rng(0)
    fake = rng.integers(0, 256, real.shape, dtype=np.uint8)
    fake[:8, :8] = (255, 20, 20)
    np.testing.assert_array_equal(p.blend_to_text(real, fake), blend_to_text(real, fake))
    # and end to end the blended image differs only where the mask did
    assert (np.abs(blended.astype(np.int32) - oblended.astype(np.int32)) > 1).mean() <= 0.02
    with pytest.raises(Exception, match="Sizes of input"):
        p.blend_to_text(real, fake[:-1])
