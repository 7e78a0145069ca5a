"""ORACLE (test infrastructure, not product code) — CPU fp32 restatement of the DiT Mask R-CNN text detector as the
reference runs it (OptimizedDetectronPredictor.invoke_model, marie/detectron/detector.py:83-147; configs under
config/zoo/unilm/dit/text_detection/).

The backbone is ``oracle/vit_torch.py`` (PINNED to the reference's beit.py).  Everything after it is detectron2
(third-party; not vendored in /root/reference and not installed here; the reference's Dockerfiles install git HEAD,
"expected version 0.6"): FPN, RPN, ROIAlign, box head, FastRCNN inference, detector_postprocess are restated from
detectron2 v0.6's published algorithm — PARITY UNPINNED (no reference fixture or test covers them).
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this module.
"""
from __future__ import annotations

import math
from typing import Dict, List, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F
from PIL import Image

from oracle.vit_torch import TorchVitOracle

SCALE_CLAMP = math.log(1000.0 / 16)
VIT_PREFIX = "backbone.bottom_up.backbone."


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def resize_shortest_edge_shape(h: int, w: int, size: int = 800, max_size: int = 1333) -> Tuple[int, int]:
    """detectron2 ResizeShortestEdge.get_output_shape."""
    scale = size * 1.0 / min(h, w)
    if h < w:
        newh, neww = size, scale * w
    else:
        newh, neww = scale * h, size
    if max(newh, neww) > max_size:
        scale = max_size * 1.0 / max(newh, neww)
        newh, neww = newh * scale, neww * scale
    return int(newh + 0.5), int(neww + 0.5)


def cell_anchors(sizes=(4, 8, 16, 32, 64), ratios=(1.5, 3.5, 6.5)) -> List[torch.Tensor]:
    """DefaultAnchorGenerator.generate_cell_anchors, one size per level."""
    out = []
    for s in sizes:
        rows = []
        for r in ratios:
            w = math.sqrt(s * s / r)
            h = r * w
            rows.append([-w / 2.0, -h / 2.0, w / 2.0, h / 2.0])
        out.append(torch.tensor(rows, dtype=torch.float32))
    return out


def grid_anchors(cells, sizes_hw, strides) -> List[torch.Tensor]:
    out = []
    for cell, (h, w), s in zip(cells, sizes_hw, strides):
        sx = torch.arange(0, w * s, step=s, dtype=torch.float32)
        sy = torch.arange(0, h * s, step=s, dtype=torch.float32)
        yy, xx = torch.meshgrid(sy, sx, indexing="ij")
        shifts = torch.stack((xx.reshape(-1), yy.reshape(-1), xx.reshape(-1), yy.reshape(-1)), dim=1)
        out.append((shifts.view(-1, 1, 4) + cell.view(1, -1, 4)).reshape(-1, 4))
    return out


def apply_deltas(deltas: torch.Tensor, boxes: torch.Tensor, weights) -> torch.Tensor:
    """Box2BoxTransform.apply_deltas."""
    wx, wy, ww, wh = weights
    widths = boxes[:, 2] - boxes[:, 0]
    heights = boxes[:, 3] - boxes[:, 1]
    ctr_x = boxes[:, 0] + 0.5 * widths
    ctr_y = boxes[:, 1] + 0.5 * heights
    dx, dy = deltas[:, 0] / wx, deltas[:, 1] / wy
    dw = torch.clamp(deltas[:, 2] / ww, max=SCALE_CLAMP)
    dh = torch.clamp(deltas[:, 3] / wh, max=SCALE_CLAMP)
    pcx, pcy = dx * widths + ctr_x, dy * heights + ctr_y
    pw, ph = torch.exp(dw) * widths, torch.exp(dh) * heights
    return torch.stack((pcx - 0.5 * pw, pcy - 0.5 * ph, pcx + 0.5 * pw, pcy + 0.5 * ph), dim=1)


def clip_boxes(b: torch.Tensor, h: int, w: int) -> torch.Tensor:
    return torch.stack((b[:, 0].clamp(0, w), b[:, 1].clamp(0, h), b[:, 2].clamp(0, w), b[:, 3].clamp(0, h)), dim=1)


def nms(boxes: torch.Tensor, scores: torch.Tensor, thr: float) -> torch.Tensor:
    """torchvision.ops.nms (CPU kernel): stable descending score order, suppress IoU > thr."""
    order = torch.argsort(scores, descending=True, stable=True)
    b = boxes[order].numpy()
    n = len(b)
    areas = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    dead = np.zeros(n, bool)
    keep = []
    for i in range(n):
        if dead[i]:
            continue
        keep.append(i)
        xx1, yy1 = np.maximum(b[i, 0], b[i + 1:, 0]), np.maximum(b[i, 1], b[i + 1:, 1])
        xx2, yy2 = np.minimum(b[i, 2], b[i + 1:, 2]), np.minimum(b[i, 3], b[i + 1:, 3])
        inter = np.maximum(np.float32(0), xx2 - xx1) * np.maximum(np.float32(0), yy2 - yy1)
        with np.errstate(divide="ignore", invalid="ignore"):
            ovr = inter / (areas[i] + areas[i + 1:] - inter)
        dead[i + 1:] |= ovr > np.float32(thr)
    return order[torch.tensor(keep, dtype=torch.long)]


def rpn_proposals(heads: Sequence[np.ndarray], sizes_hw, strides, img_hw, cells, pre_topk=1000, post_topk=1000,
                  nms_thr=0.7):
    """find_top_rpn_proposals for one image.  heads[l]: (H*W, 15) fp32 = 3 objectness logits then 3 x 4 deltas per
    location (location-major, anchor-minor — the order detectron2's permute/flatten produces)."""
    anchors = grid_anchors(cells, sizes_hw, strides)
    boxes_l, scores_l, lvl_l = [], [], []
    for l, (hd, anc) in enumerate(zip(heads, anchors)):
        hd = _t(hd)
        logits = hd[:, :3].reshape(-1)
        deltas = hd[:, 3:15].reshape(-1, 4)
        k = min(len(logits), pre_topk)
        order = torch.argsort(logits, descending=True, stable=True)[:k]      # topk; equal scores: lower index first
        boxes_l.append(apply_deltas(deltas[order], anc[order], (1.0, 1.0, 1.0, 1.0)))
        scores_l.append(logits[order])
        lvl_l.append(torch.full((k,), l, dtype=torch.long))
    boxes, scores, lvl = torch.cat(boxes_l), torch.cat(scores_l), torch.cat(lvl_l)
    valid = torch.isfinite(boxes).all(dim=1) & torch.isfinite(scores)
    boxes, scores, lvl = boxes[valid], scores[valid], lvl[valid]
    boxes = clip_boxes(boxes, img_hw[0], img_hw[1])
    keep = ((boxes[:, 2] - boxes[:, 0]) > 0) & ((boxes[:, 3] - boxes[:, 1]) > 0)
    boxes, scores, lvl = boxes[keep], scores[keep], lvl[keep]
    # batched_nms, per level (torchvision's per-class loop), result ordered by score (ties: concatenation order)
    kept = torch.zeros(len(boxes), dtype=torch.bool)
    for l in range(len(heads)):
        idx = torch.nonzero(lvl == l).squeeze(1)
        if len(idx):
            kept[idx[nms(boxes[idx], scores[idx], nms_thr)]] = True
    idx = torch.nonzero(kept).squeeze(1)
    idx = idx[torch.argsort(scores[idx], descending=True, stable=True)][:post_topk]
    return boxes[idx].numpy(), scores[idx].numpy()


def _bilinear(feat: torch.Tensor, y: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    """feat (H, W, C); y, x flat sample coordinates -> (n, C).  detectron2 ROIAlign's bilinear_interpolate."""
    H, W, _ = feat.shape
    out_of_range = (y < -1.0) | (y > H) | (x < -1.0) | (x > W)
    y = y.clamp(min=0)
    x = x.clamp(min=0)
    yl = y.floor().long()
    xl = x.floor().long()
    top = yl >= H - 1
    yl = torch.where(top, torch.full_like(yl, H - 1), yl)
    yh = torch.where(top, yl, yl + 1)
    y = torch.where(top, yl.float(), y)
    right = xl >= W - 1
    xl = torch.where(right, torch.full_like(xl, W - 1), xl)
    xh = torch.where(right, xl, xl + 1)
    x = torch.where(right, xl.float(), x)
    ly, lx = y - yl.float(), x - xl.float()
    hy, hx = 1.0 - ly, 1.0 - lx
    v = (hy * hx)[:, None] * feat[yl, xl] + (hy * lx)[:, None] * feat[yl, xh] + (ly * hx)[:, None] * feat[yh, xl] + \
        (ly * lx)[:, None] * feat[yh, xh]
    v[out_of_range] = 0
    return v


def roi_align(feats: Sequence[np.ndarray], scales, rois: np.ndarray, out: int = 7) -> np.ndarray:
    """ROIPooler(output 7, ROIAlignV2 aligned, sampling_ratio 0) incl. assign_boxes_to_levels (canonical 224 @ level 4).
    feats[l]: (H, W, C) NHWC.  Returns (n, 49*C) with k = bin*C + c (the build's layout)."""
    rois_t = _t(rois).float()
    area = (rois_t[:, 2] - rois_t[:, 0]) * (rois_t[:, 3] - rois_t[:, 1])
    lvl = torch.floor(4 + torch.log2(torch.sqrt(area) / 224 + 1e-8)).clamp(2, 5).long() - 2
    C = feats[0].shape[2]
    res = np.zeros((len(rois), out * out * C), np.float32)
    for r in range(len(rois)):
        f = _t(feats[int(lvl[r])])
        s = np.float32(scales[int(lvl[r])])
        x0, y0, x1, y1 = [np.float32(v) * s - np.float32(0.5) for v in rois[r]]
        rw, rh = x1 - x0, y1 - y0
        bw, bh = rw / np.float32(out), rh / np.float32(out)
        gh, gw = int(math.ceil(rh / out)), int(math.ceil(rw / out))
        count = max(gh * gw, 1)
        ph = torch.arange(out, dtype=torch.float32)
        iy = torch.arange(gh, dtype=torch.float32)
        ix = torch.arange(gw, dtype=torch.float32)
        ys = float(y0) + ph[:, None] * float(bh) + (iy[None, :] + 0.5) * float(bh) / gh          # (7, gh)
        xs = float(x0) + ph[:, None] * float(bw) + (ix[None, :] + 0.5) * float(bw) / gw          # (7, gw)
        Y = ys[:, None, :, None].expand(out, out, gh, gw).reshape(-1)
        X = xs[None, :, None, :].expand(out, out, gh, gw).reshape(-1)
        v = _bilinear(f, Y, X).view(out * out, gh * gw, C).sum(dim=1) / count
        res[r] = v.reshape(-1).numpy()
    return res


def fast_rcnn_inference(head: np.ndarray, rois: np.ndarray, img_hw, page_hw, score_thr=0.05, nms_thr=0.5, max_det=2000):
    """FastRCNNOutputLayers.inference for one image + detector_postprocess.  head (n, 6): 2 class scores (text,
    background) then 4 box deltas."""
    h = _t(head).float()
    probs = F.softmax(h[:, :2], dim=-1)[:, 0]
    boxes = apply_deltas(h[:, 2:6], _t(rois).float(), (10.0, 10.0, 5.0, 5.0))
    valid = torch.isfinite(boxes).all(dim=1) & torch.isfinite(probs)
    boxes, probs = boxes[valid], probs[valid]
    boxes = clip_boxes(boxes, img_hw[0], img_hw[1])
    m = probs > score_thr
    boxes, probs = boxes[m], probs[m]
    keep = nms(boxes, probs, nms_thr)[:max_det]
    boxes, probs = boxes[keep], probs[keep]
    sx, sy = np.float32(page_hw[1] / img_hw[1]), np.float32(page_hw[0] / img_hw[0])
    boxes = boxes * torch.tensor([sx, sy, sx, sy])
    boxes = clip_boxes(boxes, page_hw[0], page_hw[1])
    ne = ((boxes[:, 2] - boxes[:, 0]) > 0) & ((boxes[:, 3] - boxes[:, 1]) > 0)
    return boxes[ne].numpy(), probs[ne].numpy()


class TorchDitOracle:
    def __init__(self, state: Dict[str, np.ndarray], heads: int = 12, taps=(3, 5, 7, 11), min_size=800, max_size=1333,
                 detections_per_image=2000):
        self.st = {k: _t(v) for k, v in state.items()}
        vit_state = {k[len(VIT_PREFIX):]: v for k, v in state.items() if k.startswith(VIT_PREFIX)}
        self.vit = TorchVitOracle(vit_state, heads, taps=taps)
        self.min_size, self.max_size, self.max_det = min_size, max_size, detections_per_image

    def preprocess(self, page_bgr: np.ndarray):
        h, w = page_bgr.shape[:2]
        nh, nw = resize_shortest_edge_shape(h, w, self.min_size, self.max_size)
        rgb = np.ascontiguousarray(page_bgr[:, :, ::-1])
        img = np.asarray(Image.fromarray(rgb).resize((nw, nh), Image.BILINEAR))
        H32, W32 = (nh + 31) // 32 * 32, (nw + 31) // 32 * 32
        return TorchVitOracle.preprocess(img[None], H32, W32, swap_rb=False), (nh, nw)

    @torch.no_grad()
    def fpn(self, x: torch.Tensor) -> List[torch.Tensor]:
        st = self.st
        _, c = self.vit.forward_features(x)
        outs = []
        prev = None
        for lvl in (5, 4, 3, 2):
            lat = F.conv2d(c[lvl - 2], st[f"backbone.fpn_lateral{lvl}.weight"], st[f"backbone.fpn_lateral{lvl}.bias"])
            prev = lat if prev is None else lat + F.interpolate(prev, scale_factor=2.0, mode="nearest")
            outs.insert(0, F.conv2d(prev, st[f"backbone.fpn_output{lvl}.weight"], st[f"backbone.fpn_output{lvl}.bias"],
                                    padding=1))
        outs.append(F.max_pool2d(outs[-1], kernel_size=1, stride=2, padding=0))
        return outs                                             # p2..p6, NCHW

    @torch.no_grad()
    def rpn_heads(self, feats) -> List[np.ndarray]:
        st, r = self.st, "proposal_generator.rpn_head."
        out = []
        for f in feats:
            t = F.relu(F.conv2d(f, st[r + "conv.weight"], st[r + "conv.bias"], padding=1))
            obj = F.conv2d(t, st[r + "objectness_logits.weight"], st[r + "objectness_logits.bias"])[0]
            dl = F.conv2d(t, st[r + "anchor_deltas.weight"], st[r + "anchor_deltas.bias"])[0]
            out.append(torch.cat((obj.permute(1, 2, 0).reshape(-1, 3), dl.permute(1, 2, 0).reshape(-1, 12)), dim=1).numpy())
        return out

    @torch.no_grad()
    def box_head(self, pooled: np.ndarray) -> np.ndarray:
        """pooled (n, 49*256) in the k = bin*C + c layout -> (n, 6)."""
        st = self.st
        n = len(pooled)
        x = _t(pooled).view(n, 49, 256).permute(0, 2, 1).reshape(n, -1)          # torch flatten order c*49 + bin
        x = F.relu(F.linear(x, st["roi_heads.box_head.fc1.weight"], st["roi_heads.box_head.fc1.bias"]))
        x = F.relu(F.linear(x, st["roi_heads.box_head.fc2.weight"], st["roi_heads.box_head.fc2.bias"]))
        cls = F.linear(x, st["roi_heads.box_predictor.cls_score.weight"], st["roi_heads.box_predictor.cls_score.bias"])
        box = F.linear(x, st["roi_heads.box_predictor.bbox_pred.weight"], st["roi_heads.box_predictor.bbox_pred.bias"])
        return torch.cat((cls, box), dim=1).numpy()

    def detect(self, page_bgr: np.ndarray, want_stages: bool = False):
        x, (nh, nw) = self.preprocess(page_bgr)
        feats = self.fpn(x)
        heads = self.rpn_heads(feats)
        sizes = [tuple(f.shape[2:]) for f in feats]
        props, pscores = rpn_proposals(heads, sizes, (4, 8, 16, 32, 64), (nh, nw), cell_anchors())
        nhwc = [f[0].permute(1, 2, 0).contiguous().numpy() for f in feats[:4]]
        pooled = roi_align(nhwc, (1 / 4, 1 / 8, 1 / 16, 1 / 32), props)
        head = self.box_head(pooled)
        boxes, scores = fast_rcnn_inference(head, props, (nh, nw), page_bgr.shape[:2], max_det=self.max_det)
        if want_stages:
            return boxes, scores, {"fpn": [f[0].permute(1, 2, 0).numpy() for f in feats], "rpn_heads": heads,
                                   "proposals": props, "proposal_scores": pscores, "pooled": pooled, "head": head,
                                   "resized_hw": (nh, nw)}
        return boxes, scores
