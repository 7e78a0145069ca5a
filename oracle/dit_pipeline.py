"""ORACLE (test infrastructure, not product code) — the reference's DiT box-processor control flow
(marie/boxes/dit/ulim_dit_box_processor.py:424-832) on the CPU, over ``oracle/dit_torch.py`` (detector) and
``oracle/geometry_ref.py`` (merge_boxes / lines_from_bboxes / find_line_number).  Pinning status: see those modules.
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this module.
"""
from __future__ import annotations

import numpy as np

from oracle import geometry_ref as gr
from oracle.dit_torch import TorchDitOracle


def blackout_bboxes(image: np.ndarray, boxes) -> np.ndarray:
    """reference: ulim_dit_box_processor.py:140-198 (cv2 BGR2GRAY restated: (B*1868 + G*9617 + R*4899 + 8192) >> 14)."""
    for box in boxes:
        x0, y0, x1, y1 = [int(v) for v in box]
        sn = image[y0:y1, x0:x1].astype(np.int64)
        if sn.size == 0:
            continue
        gray = (sn[..., 0] * 1868 + sn[..., 1] * 9617 + sn[..., 2] * 4899 + 8192) >> 14
        framed = (gray[0] == 0).all() and (gray[-1] == 0).all() and (gray[:, 0] == 0).all() and (gray[:, -1] == 0).all()
        if framed or (gray == 0).sum() / gray.size > 0.5:
            continue
        image[y0:y1, x0:x1, :] = 255
    return image


def _box_iou(a, b):
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    aa = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
    ab = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    lt = np.maximum(a[:, None, :2], b[None, :, :2])
    rb = np.minimum(a[:, None, 2:], b[None, :, 2:])
    wh = np.clip(rb - lt, 0, None)
    inter = wh[..., 0] * wh[..., 1]
    return inter / (aa[:, None] + ab[None, :] - inter)


class OracleDitBoxProcessor:
    def __init__(self, state, refinement=True, **detector_kwargs):
        self.det = TorchDitOracle(state, **detector_kwargs)
        self.refinement = refinement

    def step(self, image):
        boxes, scores = self.det.detect(image)
        if len(boxes) == 0:
            return [], [], []
        keep = (boxes[:, 2] - boxes[:, 0] > 2) & (boxes[:, 3] - boxes[:, 1] > 2)
        boxes = boxes[keep]
        if len(boxes) == 0:
            return [], [], []
        return gr.merge_boxes(boxes), np.zeros(len(scores), np.int64), scores

    def psm_sparse(self, image, bbox_refinement=None):
        """pages at least MIN_SIZE_TEST on both sides (no resize_image framing)."""
        refinement = self.refinement if bbox_refinement is None else bbox_refinement
        bboxes, classes, scores = [], [], []
        work = image
        for i in range(3 if refinement else 1):
            b_, c_, s_ = self.step(work)
            before = work.copy()
            work = blackout_bboxes(work, b_)
            if i == 0:
                bboxes.extend(b_); classes.extend(c_); scores.extend(s_)
                continue
            if np.array_equal(before, work) or len(b_) == 0:
                break
            tgt = np.unique(np.nonzero(_box_iou(np.asarray(bboxes), np.asarray(b_)) > 0.1)[1])
            b_, c_, s_ = np.delete(b_, tgt, axis=0), np.delete(c_, tgt, axis=0), np.delete(s_, tgt, axis=0)
            bboxes.extend(b_); classes.extend(c_); scores.extend(s_)
        sel = [(b, c, s) for b, c, s in zip(bboxes, classes, scores) if (b[3] - b[1]) / (b[2] - b[0]) < 2.5]
        if not sel:
            return [], [], [], []
        bb = np.array([t[0] for t in sel]); cc = np.array([t[1] for t in sel]); sc = np.array([t[2] for t in sel])
        bb = bb[np.lexsort((bb[:, 0], bb[:, 1]))]
        return bb, cc, sc, gr.lines_from_bboxes(bb, image.shape[0], image.shape[1])

    def extract_bounding_boxes(self, img, bbox_refinement=None):
        bboxes, classes, scores, lines = self.psm_sparse(img.copy(), bbox_refinement)
        if len(bboxes) == 0:
            return [], [], [], lines
        bi = bboxes.astype(np.int32)
        xywh = np.stack([bi[:, 0], bi[:, 1], bi[:, 2] - bi[:, 0], bi[:, 3] - bi[:, 1]], 1)
        numbers = [gr.find_line_number(lines, r) for r in xywh]
        ind = np.lexsort((bboxes[:, 0], np.asarray(numbers)))
        frags = [img[r[1]:r[1] + r[3], r[0]:r[0] + r[2]] for r in xywh]
        return xywh[ind], [frags[i] for i in ind], numbers, lines
