"""``BoxProcessorUlimDit`` — the reference's DiT word-box processor surface over the MI355X detector.

Mirrors marie/boxes/dit/ulim_dit_box_processor.py:358-832: same constructor arguments, ``psm_sparse_step`` /
``psm_sparse`` (refinement passes with blackout, IoU > 0.1 de-duplication, the swapped-name aspect filter, lexsort,
``lines_from_bboxes``) and ``extract_bounding_boxes`` (snippets cut from the ORIGINAL image, ``find_line_number``,
(line, x) lexsort that leaves ``rect_line_numbers`` unpermuted — quirk Q1 in SURVEY.md section 8).  The detector, the
blackout and the box/line geometry run in libmarie_hip.so; this file is the control flow between them.

``bbox_optimization`` (``crop_to_content_box``, off by default in the reference) shrinks every box to the ink of its snippet
through csrc/content_ops.hip (marie_icr_amd/content.py).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Any, Dict, List, Optional, Tuple

import numpy as np

from ._lib import PREC_F16, PREC_F32, Context, MarieHipError, check
from .box_processor import PSMode
from .content import optimize_boxes
from .dit import DitModel
from .geometry import find_line_numbers, lines_from_bboxes, merge_boxes


def resize_image(image: np.ndarray, desired_size, color=(255, 255, 255), keep_max_size: bool = False,
                 ctx: Optional[Context] = None):
    """marie/utils/resize_image.py:9-76: frame ``image`` on a ``desired_size`` (height, width) canvas of ``color``; an image
    larger than the canvas on either side is first shrunk, aspect kept, with cv2.INTER_CUBIC — here the HIP kernel
    ``mhip_resize_cubic_u8`` (``ctx`` required for that branch only).  Returns ``(image, (x, y, w, h))``."""
    if image.shape[0] == desired_size[0] and image.shape[1] == desired_size[1]:
        return image, (0, 0, image.shape[1], image.shape[0])
    size = image.shape[:2]

    def border(img, top, bottom, left, right):
        out = np.empty((img.shape[0] + top + bottom, img.shape[1] + left + right, img.shape[2]), img.dtype)
        out[...] = np.asarray(color, img.dtype)
        out[top:top + img.shape[0], left:left + img.shape[1]] = img
        return out

    if keep_max_size:
        h, w = size
        dh, dw = desired_size
        if w > dw and h < dh:
            delta_h = max(0, desired_size[0] - size[0])
            top, bottom = delta_h // 2, delta_h - (delta_h // 2)
            image = border(image, top, bottom, 40, 40)
            size = image.shape[:2]
            return image, (40, top, size[1], size[0])
    if size[0] > desired_size[0] or size[1] > desired_size[1]:
        if ctx is None:
            raise MarieHipError("resize_image: the INTER_CUBIC shrink runs on the GPU and needs a Context")
        ratio = min(float(desired_size[0]) / size[0], float(desired_size[1]) / size[1])
        new_size = tuple(int(x * ratio) for x in size)
        if new_size[0] < 1 or new_size[1] < 1:
            raise ValueError(f"resize_image: {size} does not fit {tuple(desired_size)}")
        src = np.ascontiguousarray(image, np.uint8)
        out = np.empty((new_size[0], new_size[1], src.shape[2]), np.uint8)
        check(ctx.h, ctx.lib.mhip_resize_cubic_u8_host(ctx.h, src.ctypes.data_as(C.c_void_p), src.shape[0], src.shape[1],
                                                       src.shape[2], out.ctypes.data_as(C.c_void_p), out.shape[0],
                                                       out.shape[1]), "mhip_resize_cubic_u8_host")
        image = out
        size = image.shape
    delta_w = max(0, desired_size[1] - size[1])
    delta_h = max(0, desired_size[0] - size[0])
    top, bottom = delta_h // 2, delta_h - (delta_h // 2)
    left, right = delta_w // 2, delta_w - (delta_w // 2)
    image = border(image, top, bottom, left, right)
    return image, (left, top, size[1], size[0])


def box_iou(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """torchvision.ops.box_iou on fp32 xyxy boxes."""
    a = np.asarray(a, np.float32).reshape(-1, 4)
    b = np.asarray(b, np.float32).reshape(-1, 4)
    area_a = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
    area_b = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    lt = np.maximum(a[:, None, :2], b[None, :, :2])
    rb = np.minimum(a[:, None, 2:], b[None, :, 2:])
    wh = np.clip(rb - lt, 0, None)
    inter = wh[..., 0] * wh[..., 1]
    return inter / (area_a[:, None] + area_b[None, :] - inter)


class _Done:
    """A finished upload (the single-page path needs no helper thread)."""

    def __init__(self, value):
        self.value = value

    def result(self):
        return self.value


class BoxProcessorUlimDit:
    """Drop-in for marie/boxes/dit/ulim_dit_box_processor.py:358."""

    def __init__(self, work_dir: str = "/tmp/boxes", models_dir: Optional[str] = None, cuda: bool = False,
                 refinement: bool = True, *, state: Optional[Dict[str, np.ndarray]] = None, model: str = "large",
                 precision: str = "f16", device_id: int = 0, ctx: Optional[Context] = None, config=None, det_batch: int = 8,
                 dit_model: Optional[DitModel] = None):
        if not cuda:
            raise MarieHipError("BoxProcessorUlimDit here is the MI355X path; cuda=False has no implementation")
        self.work_dir = work_dir
        self.cuda = cuda
        self.refinement = refinement
        self.strict_box_segmentation = False
        if state is None and dit_model is None:
            # the reference's default: MODEL.WEIGHTS of mask_rcnn_dit_prod.yaml:12 under the model zoo
            # (ulim_dit_box_processor.py:384-417).  Looked up before a device context exists, so that a missing checkpoint is
            # the loader's error whatever the machine.
            from .constants import __model_path__

            models_dir = __model_path__ if models_dir is None else models_dir
            path = os.path.join(models_dir, "unilm/dit/text_detection/tuned-4000-LARGE-05302024/model_0147999.pth")
            if not os.path.exists(path):
                raise FileNotFoundError(f"File not found : {path}")
        self.ctx = ctx or (dit_model.ctx if dit_model is not None else Context(device_id))
        if dit_model is not None:            # an already-loaded detector (its weights stay where they are)
            self.model = dit_model
            self.det_batch = int(det_batch)
            self.min_size_test = [self.model.cfg.min_size_test, self.model.cfg.min_size_test]
            return
        if state is None:
            import torch

            sd = torch.load(path, map_location="cpu", weights_only=True)
            sd = sd.get("model", sd)
            state = {k: (v.numpy() if hasattr(v, "numpy") else np.asarray(v)) for k, v in sd.items()}
        prec = {"f16": PREC_F16, "fp16": PREC_F16, "f32": PREC_F32, "fp32": PREC_F32}[precision]
        self.model = DitModel(self.ctx, state, model=model, precision=prec, config=config)
        self.det_batch = int(det_batch)
        self.min_size_test = [self.model.cfg.min_size_test, self.model.cfg.min_size_test]

    # -- device page helpers -------------------------------------------------------------------------------------
    def _upload(self, image: np.ndarray):
        import torch

        return torch.from_numpy(np.ascontiguousarray(image)).cuda()

    def _upload_all(self, images):
        """Host pages -> HBM on a helper thread and a copy stream of its own, in the order given; returns one future per page whose
        result is the device page, already ordered behind the caller's current stream.  A 2550 x 3300 page is 25 MB of pageable
        memory (the reference's contract: numpy frames): copied up front, one after the other, the 64 pages of a call cost as much
        wall time as their detector forwards — now page k + 1 crosses PCIe while the detector works on page k."""
        import concurrent.futures
        import torch

        if len(images) <= 1:
            return [_Done(self._upload(im)) for im in images]
        copy_stream = torch.cuda.Stream()
        user_stream = torch.cuda.current_stream()
        pool = concurrent.futures.ThreadPoolExecutor(max_workers=1, thread_name_prefix="marie-h2d")

        def one(im):
            with torch.cuda.stream(copy_stream):
                t = torch.from_numpy(np.ascontiguousarray(im)).cuda()      # pageable source: blocks this thread only
                ev = torch.cuda.Event()
                ev.record(copy_stream)
            return t, ev

        class _Page:
            def __init__(self, fut):
                self.fut = fut

            def result(self):
                t, ev = self.fut.result()
                user_stream.wait_event(ev)
                t.record_stream(user_stream)
                return t

        futs = [_Page(pool.submit(one, im)) for im in images]
        pool.shutdown(wait=False)
        return futs

    def _blackout(self, page_dev, boxes_xyxy) -> bool:
        b = np.ascontiguousarray(np.asarray(boxes_xyxy, np.float32).reshape(-1, 4).astype(np.int32))   # int(x): truncation
        changed = C.c_int(0)
        h, w = page_dev.shape[:2]
        check(self.ctx.h, self.ctx.lib.mhip_blackout_bboxes(self.ctx.h, C.c_void_p(page_dev.data_ptr()), h, w,
                                                            b.ctypes.data_as(C.c_void_p), len(b), C.byref(changed)),
              "mhip_blackout_bboxes")
        return bool(changed.value)

    def _detect_batch(self, page_devs, shape):
        """One detector forward over device pages of one shape -> [(boxes xyxy fp32 page coordinates, scores)] per page.  The one
        call the parity tests swap (a CPU detector under this class's control flow, and the other way round)."""
        import torch

        self.ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        out = []
        for s0 in range(0, len(page_devs), self.det_batch):
            chunk = page_devs[s0:s0 + self.det_batch]
            out.extend(self.model.detect_device([d.data_ptr() for d in chunk], shape[0], shape[1]))
        return out

    # -- page segmentation ------------------------------------------------------------------------------------------
    @staticmethod
    def _post_step(boxes, scores, shape, adj_x: int, adj_y: int):
        """reference: ulim_dit_box_processor.py:441-497 (what psm_sparse_step does with the predictor's output)."""
        if len(boxes) == 0:
            return [], [], []
        bboxes = boxes
        if adj_x != 0 or adj_y != 0:
            bboxes[:, 0::2] -= np.float32(adj_x)
            bboxes[:, 1::2] -= np.float32(adj_y)
            bboxes[:, 0::2] = np.clip(bboxes[:, 0::2], 0, shape[1])
            bboxes[:, 1::2] = np.clip(bboxes[:, 1::2], 0, shape[0])
        keep = (bboxes[:, 2] - bboxes[:, 0] > 2) & (bboxes[:, 3] - bboxes[:, 1] > 2)
        bboxes = bboxes[keep]
        if len(bboxes) == 0:
            return [], [], []
        bboxes = np.array(merge_boxes(bboxes, 0.08))
        # as in the reference, classes / scores keep the detector's length and order (they are not merged)
        return bboxes, np.zeros((len(scores),), np.int64), scores

    def psm_sparse_step(self, page_dev, shape, adj_x: int, adj_y: int):
        """reference: ulim_dit_box_processor.py:424-497.  ``page_dev``: CUDA uint8 tensor (h, w, 3) BGR."""
        (boxes, scores), = self._detect_batch([page_dev], shape)
        return self._post_step(boxes, scores, shape, adj_x, adj_y)

    def psm_sparse(self, image: np.ndarray, bbox_optimization: Optional[bool] = False,
                   bbox_context_aware: Optional[bool] = True, bbox_refinement: Optional[bool] = None,
                   enable_visualization: Optional[bool] = False):
        """reference: ulim_dit_box_processor.py:499-658."""
        return self.psm_sparse_batch([image], bbox_optimization, bbox_context_aware, bbox_refinement)[0][:5]

    def psm_sparse_batch(self, images, bbox_optimization: Optional[bool] = False, bbox_context_aware: Optional[bool] = True,
                         bbox_refinement: Optional[bool] = None):
        """``psm_sparse`` (ulim_dit_box_processor.py:499-658) for several pages at once: every refinement pass runs the
        detector over all pages that are still active, batched by page shape (``det_batch`` pages per forward; batching does
        not change a page's boxes).  Per page the control flow is the reference's: pass 0 keeps everything; a later pass stops
        the page when the blackout changed nothing or nothing was found, else drops boxes overlapping (IoU > 0.1) earlier
        ones.  Returns per page ``(bboxes, classes, scores, lines, classes, device page or None)`` — the last item is the
        pristine page in HBM when the page was not framed (fragments can then be read where they are)."""
        refinement = self.refinement if bbox_refinement is None else bbox_refinement
        steps = 3 if refinement else 1
        pages = []
        for image in images:
            adj_x = adj_y = 0
            framed = image
            if image.shape[0] < self.min_size_test[0] or image.shape[1] < self.min_size_test[1]:
                framed, coord = resize_image(image, (self.min_size_test[0], self.min_size_test[1]), keep_max_size=True,
                                             ctx=self.ctx)
                adj_x, adj_y = coord[0], coord[1]
            pages.append({"image": framed, "adj": (adj_x, adj_y), "unframed": framed is image,
                          "bboxes": [], "classes": [], "scores": [], "active": True})
        # pages go to the device in the order the first pass takes them (grouped by shape), under the detector forwards
        first: Dict[Tuple[int, ...], list] = {}
        for pg in pages:
            first.setdefault(tuple(pg["image"].shape), []).append(pg)
        in_order = [pg for group in first.values() for pg in group]
        for pg, fut in zip(in_order, self._upload_all([pg["image"] for pg in in_order])):
            pg["upload"] = fut

        def on_device(pg):
            if "pristine" not in pg:
                dev = pg.pop("upload").result()
                pg["pristine"] = dev
                pg["dev"] = dev if pg["unframed"] else None
                # the refinement image: boxes found so far are painted white on it; the pristine copy stays for the fragments
                pg["work"] = dev.clone() if steps > 1 else dev
            return pg["work"]

        for i in range(steps):
            groups: Dict[Tuple[int, ...], list] = {}
            for pg in pages:
                if pg["active"]:
                    groups.setdefault(tuple(pg["image"].shape), []).append(pg)
            if not groups:
                break
            # det_batch pages per detector call: a call starts as soon as ITS pages have arrived (a page's boxes do not depend on
            # what shares its forward)
            nb = max(1, int(self.det_batch))
            for shape, group in [(sh, whole[s0:s0 + nb]) for sh, whole in groups.items() for s0 in range(0, len(whole), nb)]:
                dets = self._detect_batch([on_device(pg) for pg in group], shape)
                for pg, (boxes, scores) in zip(group, dets):
                    bboxes_, classes_, scores_ = self._post_step(boxes, scores, shape, *pg["adj"])
                    # a single pass never looks at the whitened page again: the blackout is skipped (output-invariant)
                    changed = self._blackout(pg["work"], bboxes_) if (len(bboxes_) and steps > 1) else False
                    if i == 0:
                        pg["bboxes"].extend(bboxes_)
                        pg["classes"].extend(classes_)
                        pg["scores"].extend(scores_)
                        continue
                    if not changed or len(bboxes_) == 0:
                        pg["active"] = False
                        continue
                    ious = box_iou(np.asarray(pg["bboxes"], np.float32), np.asarray(bboxes_, np.float32))
                    tgt = np.unique(np.nonzero(ious > 0.1)[1])
                    pg["bboxes"].extend(np.delete(bboxes_, tgt, axis=0))
                    pg["classes"].extend(np.delete(classes_, tgt, axis=0))
                    pg["scores"].extend(np.delete(scores_, tgt, axis=0))
        out = []
        for pg in pages:
            if bbox_optimization and len(pg["bboxes"]):
                # ulim_dit_box_processor.py:608-626: every box shrinks to the ink of its snippet (all boxes of the page in one call)
                ph, pw = pg["image"].shape[:2]
                pg["bboxes"] = optimize_boxes(self.ctx, pg["pristine"].data_ptr(), ph, pw, pg["bboxes"], bool(bbox_context_aware))
            bb, cc, sc = [], [], []
            for box, cls, score in zip(pg["bboxes"], pg["classes"], pg["scores"]):   # names swapped as in the reference (Q4)
                h = box[2] - box[0]
                w = box[3] - box[1]
                if w / h < 2.5:
                    bb.append(box)
                    cc.append(cls)
                    sc.append(score)
            bboxes, classes, scores = np.array(bb), np.array(cc), np.array(sc)
            if len(bboxes) == 0:
                out.append(([], [], [], [], [], pg["dev"]))
                continue
            ind = np.lexsort((bboxes[:, 0], bboxes[:, 1]))
            bboxes = bboxes[ind]
            lines = lines_from_bboxes(pg["image"], bboxes)
            out.append((bboxes, classes, scores, lines, classes, pg["dev"]))
        return out

    def psm_word(self, image):
        return self.psm_sparse(image)

    def psm_line(self, image):
        return self.psm_sparse(image)

    def psm_raw_line(self, image):
        return self.psm_sparse(image)

    def psm_multiline(self, image):
        return self.psm_sparse(image)

    def extract_bounding_boxes(self, _id, key, img, psm=PSMode.SPARSE, bbox_optimization: Optional[bool] = False,
                               bbox_context_aware: Optional[bool] = True, bbox_refinement: Optional[bool] = None
                               ) -> Tuple[Any, Any, Any, Any, Any]:
        """reference: ulim_dit_box_processor.py:676-832."""
        return self.extract_bounding_boxes_batch(_id, key, [img], psm, bbox_optimization, bbox_context_aware,
                                                 bbox_refinement)[0]

    def extract_bounding_boxes_batch(self, _id, key, imgs, psm=PSMode.SPARSE, bbox_optimization: Optional[bool] = False,
                                     bbox_context_aware: Optional[bool] = True, bbox_refinement: Optional[bool] = None,
                                     copy_fragments: bool = True):
        """``extract_bounding_boxes`` (ulim_dit_box_processor.py:676-832) for a list of pages with the detector batched over
        them; one 5-tuple per page, identical to what the per-page call returns.  ``fragments`` is a ``FragmentList``: numpy
        windows of the ORIGINAL image as in the reference, plus — when the page sits in HBM unframed — the device window each
        was cut from."""
        from .fragments import FragmentList

        checked = []
        for img in imgs:
            if img is None:
                raise Exception("Input image can't be empty")
            if not isinstance(img, np.ndarray):
                if hasattr(img, "convert"):      # PIL image -> BGR ndarray
                    img = np.array(img.convert("RGB"), dtype=np.uint8)[:, :, ::-1].copy()
                else:
                    raise ValueError("Expected image in numpy format")
            checked.append(img)
        if psm in (PSMode.RAW_LINE, PSMode.WORD):
            return [([[0, 0, im.shape[1], im.shape[0]]], [im.copy()], [0], dict(), []) for im in checked]
        if psm not in (PSMode.SPARSE, PSMode.LINE, PSMode.MULTI_LINE):
            raise Exception(f"PSM mode not supported : {psm}")
        if psm == PSMode.SPARSE:
            found = self.psm_sparse_batch(checked, bbox_optimization, bbox_context_aware, bbox_refinement)
        else:
            found = self.psm_sparse_batch(checked)
        results = []
        for img, (bboxes, polys, scores, lines_bboxes, classes, dev) in zip(checked, found):
            rect_from_poly, rect_line_numbers, fragments, windows = [], [], [], []
            if len(bboxes):
                bi = np.asarray(bboxes).astype(np.int32)
                xywh = np.stack([bi[:, 0], bi[:, 1], bi[:, 2] - bi[:, 0], bi[:, 3] - bi[:, 1]], axis=1)
                numbers = find_line_numbers(lines_bboxes, xywh)
                H, W = img.shape[:2]
                base = dev.data_ptr() if dev is not None else 0
                for i in range(len(bboxes)):
                    if classes[i] == 0:
                        x0, y0, w, h = (int(v) for v in xywh[i])
                        frag = img[y0:y0 + h, x0:x0 + w:]
                        fragments.append(frag)
                        windows.append((base + (y0 * W + x0) * 3, frag.shape[0], frag.shape[1], W * 3, 3))
                        rect_from_poly.append([x0, y0, w, h])
                        rect_line_numbers.append(numbers[i])
            if len(bboxes) > 0:
                # the reference indexes rect_line_numbers by detection index here, so an (impossible on this model) non-text
                # class would mis-align them; with one class the two index spaces coincide
                aug = np.array([[b[0], b[1], b[2], b[3], rect_line_numbers[i]] for i, b in enumerate(bboxes)])
                ind = np.lexsort((aug[:, 0], aug[:, 4]))
                bboxes = bboxes[ind]
                scores = scores[ind]
                rect_from_poly = np.array(rect_from_poly)[ind]
                fragments = [fragments[i] for i in ind]
                windows = [windows[i] for i in ind]
            usable = dev is not None and img.ndim == 3 and all(f.size > 0 for f in fragments)
            # the reference returns copies (ulim_dit_box_processor.py:805); a caller that reads the device windows and drops the
            # list before the frames change (the engine's batched path) asks for views: the 40 line copies of a page cost 1.2 ms
            if copy_fragments or not usable:
                fragments = [np.array(f, dtype=np.uint8) for f in fragments]
            fragments = FragmentList(fragments, windows if usable else None, [dev] if usable else ())
            prediction_result = {"bboxes": bboxes, "polys": bboxes, "scores": scores, "heatmap": None}
            results.append((rect_from_poly, fragments, rect_line_numbers, prediction_result, lines_bboxes))
        return results
