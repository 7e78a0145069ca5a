"""ORACLE-SIDE TEST INFRASTRUCTURE (not product code) — decision-margin analysis of the detector's discrete stages.

The DiT Mask R-CNN detector (oracle/dit_torch.py, restating detectron2's ``find_top_rpn_proposals`` and
``fast_rcnn_inference`` as marie/detectron/detector.py:83-147 runs them) interleaves continuous stages (backbone, FPN, RPN
head, ROIAlign, box head) with discrete ones (per-level top-k, NMS at 0.7, post-NMS top-1000, score threshold 0.05, NMS at
0.5).  A second implementation whose continuous tensors differ from the oracle's by a small eps can only produce a different
box SET where a discrete decision of the oracle sits within eps of its threshold.  This module makes that statement checkable:
it re-runs the oracle's discrete stages as an *interval* computation — every candidate ends up ``KEPT`` (kept under every
perturbation of the inputs up to eps), ``DROPPED`` (dropped under every such perturbation) or ``UNSTABLE`` (a score within
eps of the k-th score / the threshold, or an IoU whose [lo, hi] interval straddles the NMS threshold against a box that may
itself be kept).  tests/test_fullsize_gpu.py then requires: every KEPT detection has a partner at IoU >= 0.999 in the other
implementation's output, and every box of the other implementation is a KEPT or UNSTABLE candidate — i.e. every miss is a
proven near-tie.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this module.
"""
from __future__ import annotations

from typing import Dict, List, Sequence, Tuple

import numpy as np
import torch

from oracle import dit_torch as dt

KEPT, DROPPED, UNSTABLE = 1, 0, 2


def iou_bounds(box: np.ndarray, others: np.ndarray, eb: float) -> Tuple[np.ndarray, np.ndarray]:
    """[lo, hi] of IoU(box, others[j]) when every coordinate of both boxes may move by up to ``eb``."""
    b = box.astype(np.float64)
    o = others.astype(np.float64)
    iw = np.minimum(b[2], o[:, 2]) - np.maximum(b[0], o[:, 0])
    ih = np.minimum(b[3], o[:, 3]) - np.maximum(b[1], o[:, 1])
    iw_lo, iw_hi = np.clip(iw - 2 * eb, 0, None), np.clip(iw + 2 * eb, 0, None)
    ih_lo, ih_hi = np.clip(ih - 2 * eb, 0, None), np.clip(ih + 2 * eb, 0, None)
    inter_lo, inter_hi = iw_lo * ih_lo, iw_hi * ih_hi

    def area(x, d):
        return np.clip(x[..., 2] - x[..., 0] + d, 0, None) * np.clip(x[..., 3] - x[..., 1] + d, 0, None)

    a_lo, a_hi = area(b, -2 * eb), area(b, 2 * eb)
    o_lo, o_hi = area(o, -2 * eb), area(o, 2 * eb)
    un_lo = np.clip(a_lo + o_lo - inter_hi, 1e-12, None)
    un_hi = np.clip(a_hi + o_hi - inter_lo, 1e-12, None)
    return inter_lo / un_hi, np.minimum(inter_hi / un_lo, 1.0)


def interval_nms(boxes: np.ndarray, scores: np.ndarray, present: np.ndarray, thr: float, es: float, eb: float) -> np.ndarray:
    """Greedy NMS (suppress IoU > thr, descending score) under perturbations: scores +-es, coordinates +-eb.
    ``present[i]``: KEPT = surely a candidate, UNSTABLE = may or may not be one.  Returns KEPT / DROPPED / UNSTABLE per box."""
    n = len(boxes)
    order = np.argsort(-scores, kind="stable")
    b, s, p = boxes[order], scores[order], present[order]
    state = np.full(n, UNSTABLE, np.int8)
    for i in range(n):
        lo, hi = iou_bounds(b[i], b, eb)
        surely_before = s > s[i] + 2 * es
        maybe_before = s >= s[i] - 2 * es
        maybe_before[i] = False
        # processed entries (index < i) carry their final state; entries at or after i within 2*es are not decided yet and
        # count as "may be kept"
        st = state.copy()
        st[i:] = UNSTABLE
        if np.any(surely_before & (st == KEPT) & (lo > thr)):
            state[i] = DROPPED
        elif np.any(maybe_before & (st != DROPPED) & (hi > thr)):
            state[i] = UNSTABLE
        else:
            state[i] = KEPT if p[i] == KEPT else UNSTABLE
    out = np.empty(n, np.int8)
    out[order] = state
    return out


def rpn_intervals(heads: Sequence[np.ndarray], sizes_hw, strides, img_hw, es: float, eb: float, pre_topk=1000, post_topk=1000,
                  nms_thr=0.7) -> Dict[str, np.ndarray]:
    """Interval version of oracle.dit_torch.rpn_proposals.  Returns every candidate that may reach the proposal list with
    its state: ``boxes`` (n, 4), ``scores`` (n,), ``state`` (n,) in {KEPT, UNSTABLE} (DROPPED ones are removed)."""
    anchors = dt.grid_anchors(dt.cell_anchors(), sizes_hw, strides)
    all_b, all_s, all_st = [], [], []
    for hd, anc in zip(heads, anchors):
        hd = torch.from_numpy(np.ascontiguousarray(hd))
        logits = hd[:, :3].reshape(-1)
        deltas = hd[:, 3:15].reshape(-1, 4)
        k = min(len(logits), pre_topk)
        srt = torch.sort(logits, descending=True, stable=True)
        kth = float(srt.values[k - 1])
        nxt = float(srt.values[k]) if len(logits) > k else -np.inf        # best score left outside the top-k
        cand = torch.nonzero(logits >= kth - 2 * es).squeeze(1)
        lg = logits[cand].numpy()
        sure = lg > nxt + 2 * es                                          # cannot be displaced from the top-k
        boxes = dt.apply_deltas(deltas[cand], anc[cand], (1.0, 1.0, 1.0, 1.0))
        ok = (torch.isfinite(boxes).all(dim=1)).numpy() & np.isfinite(lg)
        boxes = dt.clip_boxes(boxes, img_hw[0], img_hw[1]).numpy()
        w, h = boxes[:, 2] - boxes[:, 0], boxes[:, 3] - boxes[:, 1]
        # an empty box of the oracle (clipped to zero height at the image border) may be a sliver in a run whose coordinates
        # differ by eb: it stays a candidate, as UNSTABLE
        ok &= (w > -2 * eb) & (h > -2 * eb) & ((w > 0) & (h > 0) | (eb > 0))
        sure &= (w > 2 * eb) & (h > 2 * eb)
        boxes, lg, sure = boxes[ok], lg[ok], sure[ok]
        present = np.where(sure, KEPT, UNSTABLE).astype(np.int8)
        st = interval_nms(boxes, lg, present, nms_thr, es, eb)
        keep = st != DROPPED
        all_b.append(boxes[keep]); all_s.append(lg[keep]); all_st.append(st[keep])
    boxes, scores, state = np.concatenate(all_b), np.concatenate(all_s), np.concatenate(all_st)
    # post-NMS top-k over all levels
    n_sure_above = np.array([np.sum((state == KEPT) & (scores > s + 2 * es)) for s in scores])
    n_maybe_above = np.array([np.sum(scores >= s - 2 * es) - 1 for s in scores])
    out_state = state.copy()
    out_state[(state == KEPT) & (n_maybe_above >= post_topk)] = UNSTABLE
    drop = n_sure_above >= post_topk
    return {"boxes": boxes[~drop], "scores": scores[~drop], "state": out_state[~drop]}


def final_intervals(head: np.ndarray, rois: np.ndarray, present: np.ndarray, img_hw, page_hw, ep: float, eb: float,
                    score_thr=0.05, nms_thr=0.5) -> Dict[str, np.ndarray]:
    """Interval version of oracle.dit_torch.fast_rcnn_inference.  ``head`` (n, 6) / ``rois`` (n, 4) of every proposal that may
    exist, ``present`` its state.  ``ep`` bounds the class-probability difference, ``eb`` the decoded-box coordinate
    difference (network-input pixels).  Returns page-coordinate ``boxes``, ``scores``, ``state`` of the candidates that are
    not surely dropped."""
    h = torch.from_numpy(np.ascontiguousarray(head)).float()
    probs = torch.softmax(h[:, :2], dim=-1)[:, 0].numpy()
    boxes = dt.apply_deltas(h[:, 2:6], torch.from_numpy(np.ascontiguousarray(rois)).float(), (10.0, 10.0, 5.0, 5.0))
    ok = torch.isfinite(boxes).all(dim=1).numpy() & np.isfinite(probs)
    boxes = dt.clip_boxes(boxes, img_hw[0], img_hw[1]).numpy()
    ok &= probs > score_thr - ep
    boxes, probs, present = boxes[ok], probs[ok], present[ok].copy()
    present[probs <= score_thr + ep] = UNSTABLE
    st = interval_nms(boxes, probs, present, nms_thr, ep, eb)
    sx, sy = np.float32(page_hw[1] / img_hw[1]), np.float32(page_hw[0] / img_hw[0])
    out = boxes * np.array([sx, sy, sx, sy], np.float32)
    out = dt.clip_boxes(torch.from_numpy(out), page_hw[0], page_hw[1]).numpy()
    w, hh = out[:, 2] - out[:, 0], out[:, 3] - out[:, 1]
    tiny = (w <= 2 * eb * sx) | (hh <= 2 * eb * sy)
    st[(st == KEPT) & tiny] = UNSTABLE
    keep = (st != DROPPED) & (w > 0) & (hh > 0)
    return {"boxes": out[keep], "scores": probs[keep], "state": st[keep]}


def pair_iou(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    if len(a) == 0 or len(b) == 0:
        return np.zeros((len(a), len(b)), np.float64)
    a, b = a.astype(np.float64), b.astype(np.float64)
    x1 = np.maximum(a[:, None, 0], b[None, :, 0]); y1 = np.maximum(a[:, None, 1], b[None, :, 1])
    x2 = np.minimum(a[:, None, 2], b[None, :, 2]); y2 = np.minimum(a[:, None, 3], b[None, :, 3])
    inter = np.clip(x2 - x1, 0, None) * np.clip(y2 - y1, 0, None)
    aa = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1]); ab = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    return inter / np.clip(aa[:, None] + ab[None, :] - inter, 1e-12, None)


def check_against(cands: Dict[str, np.ndarray], got_boxes: np.ndarray, iou_bar: float = 0.999, coord_tol: float = 0.0
                  ) -> Dict[str, object]:
    """Every KEPT candidate must have a partner in ``got_boxes`` and every box of ``got_boxes`` must be a KEPT or UNSTABLE
    candidate.  Partner = IoU >= iou_bar, or (``coord_tol`` > 0, for boxes a few pixels wide whose IoU moves by percents
    under a 0.01 px shift) every coordinate within ``coord_tol``.  Returns the counts and the offenders (empty lists = every
    miss is a near-tie)."""
    iou = pair_iou(cands["boxes"], got_boxes)
    if coord_tol > 0 and len(got_boxes) and len(cands["boxes"]):
        d = np.abs(cands["boxes"][:, None, :].astype(np.float64) - got_boxes[None, :, :].astype(np.float64)).max(axis=2)
        iou = np.where(d <= coord_tol, 1.0, iou)
    best_c = iou.max(axis=1) if len(got_boxes) else np.zeros(len(cands["boxes"]))
    best_g = iou.max(axis=0) if len(cands["boxes"]) else np.zeros(len(got_boxes))
    kept = cands["state"] == KEPT
    missing = np.nonzero(kept & (best_c < iou_bar))[0]
    foreign = np.nonzero(best_g < iou_bar)[0]
    # how far a box of the other run sits from the candidate it is matched with (the candidate nearest in coordinates among
    # those that pass the bar): px; relative to the candidate's extent along the axis of each coordinate; and the largest px
    # deviation among the pairs that are NOT within 0.2 % of the extent (those matched by the coordinate tolerance: small boxes)
    dev_px = dev_rel = dev_px_small = 0.0
    if len(got_boxes) and len(cands["boxes"]):
        cb, gb = cands["boxes"].astype(np.float64), got_boxes.astype(np.float64)
        d = np.abs(cb[:, None, :] - gb[None, :, :]).max(axis=2)
        d = np.where(iou >= iou_bar, d, np.inf)
        j = d.argmin(axis=0)
        dm = d[j, np.arange(len(gb))]
        okm = np.isfinite(dm)
        if okm.any():
            c, g = cb[j][okm], gb[okm]
            ext = np.stack([c[:, 2] - c[:, 0], c[:, 3] - c[:, 1]] * 2, axis=1)
            rel = (np.abs(c - g) / np.clip(ext, 1e-6, None)).max(axis=1)
            dev_px, dev_rel = float(dm[okm].max()), float(rel.max())
            dev_px_small = float(dm[okm][rel > 2e-3].max()) if (rel > 2e-3).any() else 0.0
    return {"kept": int(kept.sum()), "unstable": int((~kept).sum()), "got": int(len(got_boxes)),
            "kept_matched": int((kept & (best_c >= iou_bar)).sum()), "missing_kept": missing.tolist(),
            "foreign": foreign.tolist(),
            "unstable_present": int(((~kept) & (best_c >= iou_bar)).sum()),
            "max_coord_dev_px": dev_px, "max_coord_dev_over_extent": dev_rel, "max_coord_dev_px_of_pairs_beyond_0.2pct": dev_px_small}


def explain_end_to_end(oracle, stages: Dict[str, object], out: Dict[str, object], page_hw, margin: float = 1.5
                       ) -> Dict[str, object]:
    """The whole argument for one page.  ``stages`` = the oracle's (``TorchDitOracle.detect(want_stages=True)``), ``out`` = the
    other implementation's taps (``DitModel.debug_host``: rpn_heads, proposals, head, boxes, sizes).  Measures the error of
    the continuous tensors (x ``margin``), runs the interval analysis with it and checks both box lists against it.
    A match is IoU >= 0.999, or — for boxes a few pixels wide — every coordinate within the measured coordinate error."""
    nh, nw = stages["resized_hw"]
    sizes, strides = [tuple(s) for s in out["sizes"]], (4, 8, 16, 32, 64)
    e_rpn = max(float(np.abs(g - r).max()) for g, r in zip(out["rpn_heads"], stages["rpn_heads"]))
    es = margin * e_rpn
    eb = 0.0
    anchors = dt.grid_anchors(dt.cell_anchors(), sizes, strides)
    for g, r, anc in zip(out["rpn_heads"], stages["rpn_heads"], anchors):
        lg = torch.from_numpy(np.ascontiguousarray(r[:, :3])).reshape(-1)
        srt = torch.sort(lg, descending=True, stable=True)
        kth = float(srt.values[min(len(lg), 1000) - 1])
        # every anchor that may reach the top-k under the measured logit error (rpn_intervals' candidate set), at least 1200
        n_c = max(1200, int((lg >= kth - 2 * es).sum()))
        idx = srt.indices[:n_c]
        a = dt.apply_deltas(torch.from_numpy(np.ascontiguousarray(r[:, 3:15])).reshape(-1, 4)[idx], anc[idx], (1.0,) * 4)
        b = dt.apply_deltas(torch.from_numpy(np.ascontiguousarray(g[:, 3:15])).reshape(-1, 4)[idx], anc[idx], (1.0,) * 4)
        eb = max(eb, float((a - b).abs().max()))
    eb *= margin
    props = rpn_intervals(stages["rpn_heads"], sizes, strides, (nh, nw), es, eb)
    chk_p = check_against(props, out["proposals"], coord_tol=eb)
    if chk_p["foreign"]:      # diagnosis: how far the unexplained proposals are from the nearest candidate
        fb = np.asarray(out["proposals"])[chk_p["foreign"]].astype(np.float64)
        dd = np.abs(props["boxes"][None, :, :].astype(np.float64) - fb[:, None, :]).max(axis=2)
        chk_p["foreign_nearest_candidate_px"] = dd.min(axis=1).tolist()
        chk_p["foreign_boxes"] = fb.tolist()
    nhwc = [np.ascontiguousarray(f) for f in stages["fpn"][:4]]
    head_all = oracle.box_head(dt.roi_align(nhwc, (1 / 4, 1 / 8, 1 / 16, 1 / 32), props["boxes"]))
    dist = np.abs(props["boxes"][:, None, :] - out["proposals"][None, :, :]).max(axis=2)
    j, both = dist.argmin(axis=1), dist.min(axis=1) <= eb
    pr_o = torch.softmax(torch.from_numpy(head_all[both, :2]), -1)[:, 0].numpy()
    pr_g = torch.softmax(torch.from_numpy(np.ascontiguousarray(out["head"][j[both], :2])), -1)[:, 0].numpy()
    w = (10.0, 10.0, 5.0, 5.0)
    bx_o = dt.apply_deltas(torch.from_numpy(head_all[both, 2:6]), torch.from_numpy(props["boxes"][both]), w)
    bx_g = dt.apply_deltas(torch.from_numpy(np.ascontiguousarray(out["head"][j[both], 2:6])),
                           torch.from_numpy(np.ascontiguousarray(out["proposals"][j[both]])), w)
    ep = margin * float(np.abs(pr_o - pr_g).max()) if both.any() else 0.0
    eb2 = margin * float((bx_o - bx_g).abs().max()) if both.any() else 0.0
    cands = final_intervals(head_all, props["boxes"], props["state"], (nh, nw), page_hw, ep, eb2)
    scale = max(page_hw[0] / nh, page_hw[1] / nw)
    return {"eps_logit": es, "eps_rpn_box_px": eb, "eps_prob": ep, "eps_final_box_px": eb2, "eps_page_px": eb2 * scale,
            "proposals_on_both": int(both.sum()), "proposals_check": chk_p,
            "boxes_check": check_against(cands, out["boxes"], coord_tol=eb2 * scale),
            "boxes_check_iou_only": check_against(cands, out["boxes"])}
