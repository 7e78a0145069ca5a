"""DiT Mask R-CNN text detector handle over ``mhip_dit`` (libmarie_hip.so) and its stage entry points.

reference: marie/detectron/detector.py:83-147 (OptimizedDetectronPredictor), marie/boxes/dit/ditod/backbone.py:131-153,
config/zoo/unilm/dit/text_detection/*.yaml.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional, Sequence

import numpy as np

from ._lib import PREC_F16, Context, DitConfig, check
from .vit import load_tensors

MAX_ROIS = 1000
ANCHOR_SIZES = (4.0, 8.0, 16.0, 32.0, 64.0)
ASPECT_RATIOS = (1.5, 3.5, 6.5)


def _vp(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else C.c_void_p(0)


def default_config(lib, model: str = "base") -> DitConfig:
    cfg = DitConfig()
    rc = lib.mhip_dit_default_config(0 if model == "base" else 1, C.byref(cfg))
    if rc:
        raise ValueError(f"mhip_dit_default_config({model}) -> {rc}")
    return cfg


class DitModel:
    def __init__(self, ctx: Context, state: Optional[Dict[str, np.ndarray]] = None, model: str = "base",
                 precision: int = PREC_F16, config: Optional[DitConfig] = None):
        self.ctx, self.lib, self.precision = ctx, ctx.lib, int(precision)
        self.cfg = config or default_config(ctx.lib, model)
        h = C.c_void_p()
        check(ctx.h, self.lib.mhip_dit_create(ctx.h, self.precision, C.byref(self.cfg), C.byref(h)), "mhip_dit_create")
        self.h = h
        ctx.adopt(self)
        if state is not None:
            load_tensors(ctx, self.lib.mhip_dit_set_tensor, self.h, state, "mhip_dit_set_tensor")
            check(ctx.h, self.lib.mhip_dit_finalize(self.h), "mhip_dit_finalize")

    def resized_shape(self, h: int, w: int):
        nh, nw, H32, W32 = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        self.lib.mhip_dit_resized_shape(C.byref(self.cfg), h, w, C.byref(nh), C.byref(nw), C.byref(H32), C.byref(W32))
        return nh.value, nw.value, H32.value, W32.value

    def arenas(self):
        out = []
        for which in (0, 1):
            p, n = C.c_void_p(), C.c_size_t()
            check(self.ctx.h, self.lib.mhip_dit_arena(self.h, which, C.byref(p), C.byref(n)), "mhip_dit_arena")
            out.append((p.value, n.value))
        return out

    def alloc_arena(self):
        check(self.ctx.h, self.lib.mhip_dit_alloc_arena(self.h), "mhip_dit_alloc_arena")

    def _unpack(self, boxes, scores, counts):
        return [(boxes[b, : counts[b]].copy(), scores[b, : counts[b]].copy()) for b in range(len(counts))]

    def detect_host(self, pages_bgr: np.ndarray):
        """pages (B, h, w, 3) uint8 BGR -> list of (boxes (n, 4) xyxy fp32 page coordinates, scores (n,))."""
        pages = np.ascontiguousarray(pages_bgr, np.uint8)
        if pages.ndim == 3:
            pages = pages[None]
        B, h, w, _ = pages.shape
        boxes = np.empty((B, MAX_ROIS, 4), np.float32)
        scores = np.empty((B, MAX_ROIS), np.float32)
        counts = np.zeros((B,), np.int32)
        check(self.ctx.h, self.lib.mhip_dit_detect_host(self.h, _vp(pages), B, h, w, _vp(boxes), _vp(scores), _vp(counts)),
              "mhip_dit_detect_host")
        return self._unpack(boxes, scores, counts)

    def detect_device(self, page_ptrs: Sequence[int], h: int, w: int):
        """device pages (pointers to u8 BGR [h][w][3]) of one size."""
        B = len(page_ptrs)
        ptrs = (C.c_void_p * B)(*page_ptrs)
        boxes = np.empty((B, MAX_ROIS, 4), np.float32)
        scores = np.empty((B, MAX_ROIS), np.float32)
        counts = np.zeros((B,), np.int32)
        check(self.ctx.h, self.lib.mhip_dit_detect(self.h, ptrs, B, h, w, _vp(boxes), _vp(scores), _vp(counts)),
              "mhip_dit_detect")
        return self._unpack(boxes, scores, counts)

    def debug_host(self, page_bgr: np.ndarray):
        """One page plus what the parity tests look at: FPN maps, the RPN head outputs and the box-head outputs (the inputs
        of the two discrete stages as this run computed them), proposals and detections."""
        page = np.ascontiguousarray(page_bgr, np.uint8)
        h, w, _ = page.shape
        nh, nw, H32, W32 = self.resized_shape(h, w)
        sizes = [(H32 >> (2 + l), W32 >> (2 + l)) for l in range(4)]
        sizes.append(((sizes[3][0] + 1) // 2, (sizes[3][1] + 1) // 2))
        fpn = [np.empty((s[0], s[1], 256), np.float32) for s in sizes]
        rpn = [np.empty((s[0] * s[1], 16), np.float32) for s in sizes]
        boxes, scores = np.empty((MAX_ROIS, 4), np.float32), np.empty((MAX_ROIS,), np.float32)
        pb, ps = np.empty((MAX_ROIS, 4), np.float32), np.empty((MAX_ROIS,), np.float32)
        head = np.empty((MAX_ROIS, 8), np.float32)
        n, pn = C.c_int(0), C.c_int(0)
        fp = (C.c_void_p * 5)(*[f.ctypes.data for f in fpn])
        rp = (C.c_void_p * 5)(*[r.ctypes.data for r in rpn])
        check(self.ctx.h, self.lib.mhip_dit_debug_taps_host(self.h, _vp(page), h, w, _vp(boxes), _vp(scores), C.byref(n), fp, rp,
                                                            _vp(pb), _vp(ps), C.byref(pn), _vp(head)),
              "mhip_dit_debug_taps_host")
        return {"boxes": boxes[: n.value], "scores": scores[: n.value], "fpn": fpn, "proposals": pb[: pn.value],
                "proposal_scores": ps[: pn.value], "resized_hw": (nh, nw), "sizes": sizes,
                "rpn_heads": [r[:, :15].copy() for r in rpn], "head": head[: pn.value, :6].copy()}

    def close(self):
        if getattr(self, "h", None) is not None and self.h.value:
            self.lib.mhip_dit_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---------------------------------------------------------------------------------------------------- stage entries
def rpn_proposals(ctx: Context, heads: Sequence[np.ndarray], sizes_hw, strides, img_hw, nms_thresh: float = 0.7):
    """heads[l] (H*W, 15) fp32 (3 logits + 3 x 4 deltas) for p2..p6 -> (boxes (n, 4), logits (n,)), score-ordered."""
    padded = []
    for hd, (h, w) in zip(heads, sizes_hw):
        p = np.zeros((h * w, 16), np.float32)
        p[:, :15] = hd
        padded.append(p)
    ptrs = (C.c_void_p * 5)(*[p.ctypes.data for p in padded])
    H = (C.c_int * 5)(*[s[0] for s in sizes_hw])
    W = (C.c_int * 5)(*[s[1] for s in sizes_hw])
    S = (C.c_int * 5)(*strides)
    sz = (C.c_float * 5)(*ANCHOR_SIZES)
    ar = (C.c_float * 3)(*ASPECT_RATIOS)
    boxes, scores = np.empty((MAX_ROIS, 4), np.float32), np.empty((MAX_ROIS,), np.float32)
    n = C.c_int(0)
    check(ctx.h, ctx.lib.mhip_rpn_proposals_host(ctx.h, ptrs, H, W, S, sz, ar, int(img_hw[0]), int(img_hw[1]), nms_thresh,
                                                 _vp(boxes), _vp(scores), C.byref(n)), "mhip_rpn_proposals_host")
    return boxes[: n.value], scores[: n.value]


def roi_align(ctx: Context, feats: Sequence[np.ndarray], rois: np.ndarray) -> np.ndarray:
    feats = [np.ascontiguousarray(f, np.float32) for f in feats]
    rois = np.ascontiguousarray(rois, np.float32)
    Cn = feats[0].shape[2]
    ptrs = (C.c_void_p * 4)(*[f.ctypes.data for f in feats])
    H = (C.c_int * 4)(*[f.shape[0] for f in feats])
    W = (C.c_int * 4)(*[f.shape[1] for f in feats])
    out = np.empty((len(rois), 49 * Cn), np.float32)
    check(ctx.h, ctx.lib.mhip_roi_align_host(ctx.h, ptrs, H, W, Cn, _vp(rois), len(rois), _vp(out)), "mhip_roi_align_host")
    return out


def det_final(ctx: Context, head6: np.ndarray, rois: np.ndarray, img_hw, page_hw, score_thresh=0.05, nms_thresh=0.5,
              max_det=2000):
    n = len(rois)
    head = np.zeros((n, 8), np.float32)
    head[:, :6] = head6
    rois = np.ascontiguousarray(rois, np.float32)
    boxes, scores = np.empty((MAX_ROIS, 4), np.float32), np.empty((MAX_ROIS,), np.float32)
    cnt = C.c_int(0)
    check(ctx.h, ctx.lib.mhip_det_final_host(ctx.h, _vp(head), _vp(rois), n, int(img_hw[0]), int(img_hw[1]), int(page_hw[0]),
                                             int(page_hw[1]), score_thresh, nms_thresh, max_det, _vp(boxes), _vp(scores),
                                             C.byref(cnt)), "mhip_det_final_host")
    return boxes[: cnt.value], scores[: cnt.value]


def pil_resize_rgb(ctx: Context, img: np.ndarray, out_hw, bicubic: bool = False) -> np.ndarray:
    img = np.ascontiguousarray(img, np.uint8)
    out = np.empty((int(out_hw[0]), int(out_hw[1]), 3), np.uint8)
    check(ctx.h, ctx.lib.mhip_pil_resize_rgb_host(ctx.h, _vp(img), img.shape[0], img.shape[1], _vp(out), out.shape[0],
                                                  out.shape[1], 3 if bicubic else 2), "mhip_pil_resize_rgb_host")
    return out
