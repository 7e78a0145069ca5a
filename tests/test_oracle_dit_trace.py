"""CPU check of the decision-margin analysis (oracle/dit_trace.py) that tests/test_fullsize_gpu.py leans on: a second run of
the oracle whose continuous tensors are perturbed by a known eps must (a) be accepted — every KEPT box found at IoU >= 0.999,
nothing foreign — and (b) a perturbation far beyond the declared eps must be caught."""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def small():
    from marie_icr_amd.weights import make_dit_state, make_image_u8
    from oracle.dit_torch import TorchDitOracle

    st = make_dit_state(0)
    page = make_image_u8(11, 1, 330, 255)[0]
    o = TorchDitOracle(st, min_size=160, max_size=400)
    boxes, scores, stages = o.detect(page, want_stages=True)
    return o, page, boxes, scores, stages


def _second_run(o, stages, page_hw, noise, seed):
    """The oracle's stages 4-6 on RPN-head / box-head tensors perturbed by uniform noise of amplitude ``noise``."""
    from oracle import dit_torch as dt

    rng = np.random.default_rng(seed)
    nh, nw = stages["resized_hw"]
    sizes = [f.shape[:2] for f in stages["fpn"]]
    heads = [h + rng.uniform(-noise, noise, h.shape).astype(np.float32) for h in stages["rpn_heads"]]
    props, ps = dt.rpn_proposals(heads, sizes, (4, 8, 16, 32, 64), (nh, nw), dt.cell_anchors())
    nhwc = [np.ascontiguousarray(f) for f in stages["fpn"][:4]]
    head = o.box_head(dt.roi_align(nhwc, (1 / 4, 1 / 8, 1 / 16, 1 / 32), props))
    head = head + rng.uniform(-noise, noise, head.shape).astype(np.float32)
    boxes, scores = dt.fast_rcnn_inference(head, props, (nh, nw), page_hw)
    return props, boxes, sizes


@pytest.mark.parametrize("noise", [1e-5, 1e-4])
def test_perturbed_run_is_explained(small, noise):
    from oracle import dit_torch as dt
    from oracle import dit_trace as tr

    o, page, rboxes, rscores, stages = small
    nh, nw = stages["resized_hw"]
    props2, boxes2, sizes = _second_run(o, stages, page.shape[:2], noise, 1)
    es = noise
    eb = 64 * 6.5 ** 0.5 * 2 * noise * 3         # |d box| <= anchor extent * (|d delta| * e^delta ...) — generous bound
    cand_p = tr.rpn_intervals(stages["rpn_heads"], sizes, (4, 8, 16, 32, 64), (nh, nw), es, eb)
    chk_p = tr.check_against(cand_p, props2, coord_tol=eb)
    assert not chk_p["missing_kept"] and not chk_p["foreign"], chk_p
    nhwc = [np.ascontiguousarray(f) for f in stages["fpn"][:4]]
    head_all = o.box_head(dt.roi_align(nhwc, (1 / 4, 1 / 8, 1 / 16, 1 / 32), cand_p["boxes"]))
    # second-stage error: head noise + the effect of the proposal coordinates moving by eb
    ep, eb2 = 0.25 * 2 * noise + 0.02 * eb + 1e-6, 80 * noise + 2 * eb
    cands = tr.final_intervals(head_all, cand_p["boxes"], cand_p["state"], (nh, nw), page.shape[:2], ep, eb2)
    chk = tr.check_against(cands, boxes2, coord_tol=eb2 * page.shape[0] / nh)
    assert not chk["missing_kept"] and not chk["foreign"], chk
    assert chk["kept"] >= 0.5 * len(rboxes), (chk, len(rboxes))      # small eps: most decisions are provably stable


def test_zero_eps_reproduces_the_oracle(small):
    from oracle import dit_torch as dt
    from oracle import dit_trace as tr

    o, page, rboxes, rscores, stages = small
    nh, nw = stages["resized_hw"]
    sizes = [f.shape[:2] for f in stages["fpn"]]
    cand_p = tr.rpn_intervals(stages["rpn_heads"], sizes, (4, 8, 16, 32, 64), (nh, nw), 0.0, 0.0)
    assert (cand_p["state"] == tr.KEPT).sum() == len(stages["proposals"])
    chk = tr.check_against({k: v[cand_p["state"] == tr.KEPT] for k, v in cand_p.items()}, stages["proposals"], 0.999999)
    assert not chk["missing_kept"] and not chk["foreign"]


def test_large_perturbation_is_caught(small):
    """A run whose tensors are off by 100x the declared eps must not pass as 'near-ties'."""
    from oracle import dit_trace as tr

    o, page, rboxes, rscores, stages = small
    nh, nw = stages["resized_hw"]
    props2, boxes2, sizes = _second_run(o, stages, page.shape[:2], 0.3, 2)
    cand_p = tr.rpn_intervals(stages["rpn_heads"], sizes, (4, 8, 16, 32, 64), (nh, nw), 1e-4, 1e-3)
    chk_p = tr.check_against(cand_p, props2)
    assert chk_p["missing_kept"] or chk_p["foreign"]
