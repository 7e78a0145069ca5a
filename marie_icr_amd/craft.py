"""CRAFT text detector on MI355X behind the reference's ``BoxProcessor`` surface.

Mirrors ``BoxProcessorCraft`` (reference: marie/boxes/craft_box_processor.py:244-562): same constructor arguments,
``psm_word/psm_sparse/psm_line/psm_raw_line/psm_multiline(image)`` returning ``(bboxes, polys, score_text, lines)`` and
``extract_bounding_boxes(_id, key, img, psm)`` returning ``(rect_from_poly, fragments, rect_line_numbers,
prediction_result, lines_bboxes)``.  Resize, normalisation, the whole network and the O(pixels) post-processing run in
libmarie_hip.so; this file does the per-box arithmetic the reference also does in numpy (coordinate scaling, bounding
rect, 2/4-pixel expansion, crop views).

Reference behaviours kept on purpose:
  * the RefineNet is never used: ``__load`` sets ``refine = None`` (craft_box_processor.py:288-290), so
    ``lines_bboxes`` is ``[]`` in every mode and every ``rect_line_number`` is -1;
  * the page is fed in the channel order it arrives in (BGR), although normalizeMeanVariance says RGB.
Reference behaviours dropped: the debug PNG/JPEG dumps under /tmp (craft_utils.py:40-43,
craft_box_processor.py:100,533-550) and the JET colour map of the returned heat map (raw fp32 maps are returned).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional

import numpy as np

from ._lib import PREC_F16, PREC_F32, Context, MarieHipError, check
from .box_processor import PSMode, find_line_number
from .weights import strip_module_prefix


class CraftModel:
    """Device-resident CRAFT weights + forward/detect.  Thin handle over ``mhip_craft``."""

    def __init__(self, ctx: Context, state: Optional[Dict[str, np.ndarray]], precision: int = PREC_F16,
                 max_boxes: int = 65536):
        self.ctx = ctx
        self.lib = ctx.lib
        self.precision = int(precision)
        self.max_boxes = int(max_boxes)
        h = C.c_void_p()
        check(ctx.h, self.lib.mhip_craft_create(ctx.h, self.precision, C.byref(h)), "mhip_craft_create")
        self.h = h
        ctx.adopt(self)
        self._boxes = np.empty((self.max_boxes, 4, 2), np.float32)
        if state is not None:
            self.load_state(state)

    def load_state(self, state: Dict[str, np.ndarray]):
        """reference: net.load_state_dict(copyStateDict(torch.load(...))), craft_box_processor.py:272-278."""
        for key, val in strip_module_prefix(state).items():
            arr = np.ascontiguousarray(np.asarray(val), dtype=np.float32)
            shape = (C.c_int64 * max(arr.ndim, 1))(*arr.shape)
            check(self.ctx.h,
                  self.lib.mhip_craft_set_tensor(self.h, key.encode(), arr.ctypes.data_as(C.c_void_p), shape, arr.ndim),
                  f"mhip_craft_set_tensor({key})")
        check(self.ctx.h, self.lib.mhip_craft_finalize(self.h), "mhip_craft_finalize")

    def alloc_arena(self):
        check(self.ctx.h, self.lib.mhip_craft_alloc_arena(self.h), "mhip_craft_alloc_arena")

    def arena(self):
        p = C.c_void_p()
        n = C.c_size_t()
        check(self.ctx.h, self.lib.mhip_craft_arena(self.h, C.byref(p), C.byref(n)), "mhip_craft_arena")
        return p.value, n.value

    def geometry(self, h: int, w: int, canvas_size: int, mag_ratio: float = 1.0):
        r = C.c_double()
        th, tw, H, W = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        rc = self.lib.mhip_craft_geometry(int(h), int(w), int(canvas_size), float(mag_ratio), C.byref(r), C.byref(th),
                                          C.byref(tw), C.byref(H), C.byref(W))
        if rc:
            raise MarieHipError(f"mhip_craft_geometry({h},{w},{canvas_size}) failed ({rc})")
        return {"ratio": r.value, "th": th.value, "tw": tw.value, "H": H.value, "W": W.value}

    def detect_host(self, page_u8: np.ndarray, text_threshold: float, link_threshold: float, low_text: float,
                    canvas_size: Optional[int] = None, mag_ratio: float = 1.0, want_scores: bool = True):
        """page_u8: (h, w, 3) uint8 host array.  Returns (boxes (K,4,2) fp32 in score-map coords, scores (H2,W2,2)
        or None, ratio)."""
        page = np.ascontiguousarray(page_u8, dtype=np.uint8)
        if page.ndim != 3 or page.shape[2] != 3:
            raise ValueError(f"page must be (h, w, 3) uint8, got {page.shape}")
        h, w = page.shape[:2]
        canvas = int(canvas_size if canvas_size is not None else w)
        g = self.geometry(h, w, canvas, mag_ratio)
        scores = np.empty((g["H"] // 2, g["W"] // 2, 2), np.float32) if want_scores else None
        n = C.c_int()
        ratio = C.c_double()
        check(self.ctx.h,
              self.lib.mhip_craft_detect_host(
                  self.h, page.ctypes.data_as(C.c_void_p), h, w, canvas, float(mag_ratio), float(text_threshold),
                  float(link_threshold), float(low_text), self._boxes.ctypes.data_as(C.c_void_p), self.max_boxes,
                  C.byref(n), scores.ctypes.data_as(C.c_void_p) if scores is not None else C.c_void_p(0),
                  C.byref(ratio)),
              "mhip_craft_detect_host")
        return self._boxes[:n.value].copy(), scores, ratio.value

    def detect_device(self, page_ptr: int, h: int, w: int, text_threshold: float, link_threshold: float,
                      low_text: float, canvas_size: Optional[int] = None, mag_ratio: float = 1.0):
        """Page already in HBM (uint8 [h][w][3] at ``page_ptr``).  Returns (boxes (K,4,2) fp32, ratio)."""
        canvas = int(canvas_size if canvas_size is not None else w)
        n = C.c_int()
        ratio = C.c_double()
        check(self.ctx.h,
              self.lib.mhip_craft_detect(self.h, C.c_void_p(page_ptr), int(h), int(w), canvas, float(mag_ratio),
                                         float(text_threshold), float(link_threshold), float(low_text),
                                         self._boxes.ctypes.data_as(C.c_void_p), self.max_boxes, C.byref(n),
                                         C.c_void_p(0), C.byref(ratio)),
              "mhip_craft_detect")
        return self._boxes[:n.value].copy(), ratio.value

    def kernel_flops(self, h: int, w: int, canvas_size: Optional[int] = None, mag_ratio: float = 1.0):
        canvas = int(canvas_size if canvas_size is not None else w)
        return {self.lib.mhip_kernel_name(k).decode():
                self.lib.mhip_craft_kernel_flops(self.h, k, int(h), int(w), canvas, float(mag_ratio))
                for k in range(self.lib.mhip_kernel_count())}

    def close(self):
        if getattr(self, "h", None) is not None and self.h.value:
            self.lib.mhip_craft_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def adjust_result_coordinates(polys: np.ndarray, ratio_w: float, ratio_h: float, ratio_net: int = 2) -> np.ndarray:
    """reference: adjustResultCoordinates, marie/models/craft/craft_utils.py:268-274 (in-place fp32 scaling)."""
    polys = np.array(polys, dtype=np.float32)
    if len(polys) > 0:
        for k in range(len(polys)):
            polys[k] *= (ratio_w * ratio_net, ratio_h * ratio_net)
    return polys


def rects_from_boxes(bboxes: np.ndarray, max_w: int, max_h: int) -> np.ndarray:
    """cv2.boundingRect of the int32 polygon + the reference's 2/4-pixel expansion, vectorised.
    reference: marie/boxes/craft_box_processor.py:499-520.  Returns (K,4) int32 x, y, w, h."""
    if len(bboxes) == 0:
        return np.zeros((0, 4), np.int32)
    region = np.asarray(bboxes).astype(np.int32).reshape(len(bboxes), -1, 2)
    x0, y0 = region[:, :, 0].min(axis=1), region[:, :, 1].min(axis=1)
    bw, bh = region[:, :, 0].max(axis=1) - x0 + 1, region[:, :, 1].max(axis=1) - y0 + 1
    out = np.stack([np.maximum(0, x0 - 2), np.maximum(0, y0 - 2), np.minimum(max_w, bw + 4),
                    np.minimum(max_h, bh + 4)], axis=1)
    return out.astype(np.int32)


class BoxProcessorCraft:
    """Drop-in for marie/boxes/craft_box_processor.py:244 (``BoxProcessor`` surface, marie/boxes/box_processor.py:179-256)."""

    # (text_threshold, link_threshold, low_text) per mode — craft_box_processor.py:313-434
    _THRESHOLDS = {"word": (0.6, 0.8, 0.3), "sparse": (0.7, 0.45, 0.3), "line": (0.4, 0.2, 0.3),
                   "raw_line": (0.4, 0.2, 0.5), "multiline": (0.6, 0.3, 0.3)}

    def __init__(self, work_dir: str = "/tmp/boxes", models_dir: Optional[str] = None, cuda: bool = True, *,
                 state: Optional[Dict[str, np.ndarray]] = None, precision: str = "f16", device_id: int = 0,
                 ctx: Optional[Context] = None):
        if not cuda:
            raise MarieHipError("BoxProcessorCraft here is the MI355X path; cuda=False has no implementation")
        self.work_dir = work_dir
        self.cuda = cuda
        if state is None:
            # craft_box_processor.py:248,263: models_dir defaults to <model zoo>/craft, the checkpoint is craft_mlt_25k.pth in
            # it; looked up before a device context exists, so that a missing checkpoint is the loader's error on any machine
            from .constants import __model_path__

            models_dir = os.path.join(__model_path__, "craft") if models_dir is None else models_dir
            path = os.path.join(models_dir, "craft_mlt_25k.pth")
            if not os.path.exists(path):
                raise FileNotFoundError(f"File not found : {path}")
        self.ctx = ctx or Context(device_id)
        if state is None:
            import torch

            sd = torch.load(path, map_location="cpu", weights_only=True)
            state = {k: v.numpy() for k, v in sd.items()}
        prec = {"f16": PREC_F16, "fp16": PREC_F16, "f32": PREC_F32, "fp32": PREC_F32}[precision]
        self.model = CraftModel(self.ctx, state, precision=prec)

    # -- page segmentation modes ------------------------------------------------------------------------------
    def _predict(self, image: np.ndarray, mode: str):
        """reference: get_prediction, craft_box_processor.py:76-247 with refine_net = line_refine_net = None."""
        tt, lt, low = self._THRESHOLDS[mode]
        w = image.shape[1]
        boxes, scores, ratio = self.model.detect_host(image, tt, lt, low, canvas_size=w, mag_ratio=1.0)
        ratio_w = ratio_h = 1 / ratio
        bboxes = adjust_result_coordinates(boxes, ratio_w, ratio_h)
        polys = [b for b in bboxes]          # poly=False: every poly is its box (craft_box_processor.py:133-135)
        score_text = np.hstack((scores[:, :, 0], scores[:, :, 1]))
        return bboxes, polys, score_text, []

    def psm_word(self, image):
        return self._predict(image, "word")

    def psm_sparse(self, image):
        return self._predict(image, "sparse")

    def psm_line(self, image):
        return self._predict(image, "line")

    def psm_raw_line(self, image):
        return self._predict(image, "raw_line")

    def psm_multiline(self, image):
        return self._predict(image, "multiline")

    def extract_bounding_boxes(self, _id, key, img, psm=PSMode.SPARSE):
        """reference: craft_box_processor.py:436-562."""
        if img is None:
            raise Exception("Input image can't be empty")
        image = np.asarray(img)
        lines_bboxes: List = []
        if psm == PSMode.SPARSE:
            bboxes, polys, score_text, lines_bboxes = self.psm_sparse(image)
        elif psm == PSMode.LINE:
            bboxes, polys, score_text, lines_bboxes = self.psm_line(image)
        elif psm == PSMode.MULTI_LINE:
            bboxes, polys, score_text, lines_bboxes = self.psm_multiline(image)
        elif psm == PSMode.RAW_LINE or psm == PSMode.WORD:
            h, w = image.shape[0], image.shape[1]
            return [[0, 0, w, h]], [image], [0], dict(), lines_bboxes
        else:
            raise Exception(f"PSM mode not supported : {psm}")

        prediction_result = {"bboxes": bboxes, "polys": polys, "heatmap": score_text}
        max_h, max_w = image.shape[0], image.shape[1]
        rects = rects_from_boxes(bboxes, max_w, max_h)
        # Boxes that fall entirely into the /32 canvas padding (right of / below the page) have no pixels to crop;
        # the reference would die on them inside cv2 (empty snippet, craft_box_processor.py:522-535) — drop them.
        keep = (rects[:, 0] < max_w) & (rects[:, 1] < max_h) if len(rects) else np.zeros((0,), bool)
        if len(rects) and not keep.all():
            rects = rects[keep]
            prediction_result = {"bboxes": bboxes[keep], "polys": [p for p, k in zip(polys, keep) if k],
                                 "heatmap": score_text}
        rect_from_poly, fragments, rect_line_numbers = [], [], []
        for x, y, w, h in rects.tolist():
            # crop_poly_low on the expanded axis-aligned polygon == the plain crop of its bounding rect
            # (craft_box_processor.py:42-73): rows y..y+h and columns x..x+w inclusive
            fragments.append(image[y:y + h + 1, x:x + w + 1].copy())
            rect_from_poly.append([x, y, w, h])
            rect_line_numbers.append(find_line_number(lines_bboxes, [x, y, w, h]))
        return rect_from_poly, fragments, rect_line_numbers, prediction_result, lines_bboxes
