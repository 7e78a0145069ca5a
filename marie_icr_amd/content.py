"""Content cropping: the reference's ``crop_to_content`` (marie/utils/image_utils.py:190-252) and ``crop_to_content_box``
(marie/boxes/dit/ulim_dit_box_processor.py:291-352) over the HIP kernels of csrc/content_ops.hip.

The kernels return, per rectangle of a device page, the extent of the pixels the reference's OpenCV chain turns to 0; the two
callers' padding rules (16 px / full height for a page, 1 px for a detector box) are the integer logic below, as in the reference.
The reference writes debug PNGs to /tmp/fragments on every call (image_utils.py:203,224); this does not.
"""
import ctypes as C
from typing import List, Sequence, Tuple

import numpy as np

from ._lib import check


def content_extents(ctx, page_dev_ptr: int, h: int, w: int, rects_xywh, content_aware: bool) -> np.ndarray:
    """[n][5] int32 (xmin, ymin, xmax, ymax, count) of the zero pixels of each rectangle of a BGR device page."""
    rects = np.ascontiguousarray(np.asarray(rects_xywh, np.int32).reshape(-1, 4))
    out = np.zeros((len(rects), 5), np.int32)
    if len(rects):
        import torch

        # the launches go to the caller's current stream (the one the page was uploaded / written on), not to whatever stream the
        # context last held — that one may belong to a call that has ended
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        check(ctx.h, ctx.lib.mhip_content_extents(ctx.h, C.c_void_p(page_dev_ptr), h, w, rects.ctypes.data_as(C.c_void_p), len(rects),
                                                  1 if content_aware else 0, out.ctypes.data_as(C.c_void_p)), "mhip_content_extents")
    return out


def _upload(frame: np.ndarray):
    import torch

    if frame.ndim == 2:
        frame = np.repeat(frame[:, :, None], 3, axis=2)       # a gray frame: BGR2GRAY of (v, v, v) is v
    if frame.ndim != 3 or frame.shape[2] != 3 or frame.dtype != np.uint8:
        raise ValueError(f"expected an 8-bit BGR or gray frame, got {frame.dtype} {frame.shape}")
    return torch.from_numpy(np.ascontiguousarray(frame)).cuda()


def crop_rect_page(ext, img_w: int, img_h: int, content_aware: bool):
    """image_utils.py:225-247: (x, y, w, h) of ``frame[y : y + h + 1, x : x + w + 1]`` or None when nothing was found."""
    xmin, ymin, xmax, ymax, n = (int(v) for v in ext)
    if n == 0:
        return None
    if content_aware:
        x = max(0, xmin - 16)
        return x, 0, min(img_w, xmax - x + 16), img_h
    return xmin, ymin, xmax - xmin, ymax - ymin


def crop_to_content(ctx, frame: np.ndarray, content_aware: bool = True) -> np.ndarray:
    """reference: image_utils.py:190-252."""
    img_h, img_w = frame.shape[:2]
    dev = _upload(frame)
    ext = content_extents(ctx, dev.data_ptr(), img_h, img_w, [[0, 0, img_w, img_h]], content_aware)[0]
    rect = crop_rect_page(ext, img_w, img_h, content_aware)
    if rect is None:
        return frame
    x, y, w, h = rect
    return frame[y:y + h + 1, x:x + w + 1].copy()


def box_offset(ext, img_w: int, img_h: int, content_aware: bool) -> List[int]:
    """ulim_dit_box_processor.py:327-352: the offset [left, top, img_w - w, img_h - h] of one snippet."""
    xmin, ymin, xmax, ymax, n = (int(v) for v in ext)
    if n == 0:
        return [0, 0, 0, 0]
    if content_aware:
        x = max(0, xmin - 1)
        y = max(0, ymin - 1)
        h = min(img_h, ymax - y + 1)
        w = min(img_w, xmax - x + 1)
    else:
        x, y, h, w = xmin, ymin, ymax - ymin, xmax - xmin
    return [x, y, img_w - w, img_h - h]


def crop_to_content_box(ctx, frame: np.ndarray, content_aware: bool = False) -> Tuple[List[int], np.ndarray]:
    """reference: ulim_dit_box_processor.py:291-352 for one snippet."""
    if frame is None:
        raise Exception("Frame can't be empty")
    img_h, img_w = frame.shape[:2]
    if img_h == 0 or img_w == 0:
        return [0, 0, 0, 0], frame
    dev = _upload(frame)
    ext = content_extents(ctx, dev.data_ptr(), img_h, img_w, [[0, 0, img_w, img_h]], content_aware)[0]
    off = box_offset(ext, img_w, img_h, content_aware)
    if ext[4] == 0:
        return off, frame
    x, y, w, h = off[0], off[1], img_w - off[2], img_h - off[3]
    return off, frame[y:y + h + 1, x:x + w + 1].copy()


def optimize_boxes(ctx, page_dev_ptr: int, page_h: int, page_w: int, bboxes: Sequence, content_aware: bool) -> list:
    """The ``bbox_optimization`` loop of psm_sparse (ulim_dit_box_processor.py:608-626) for all boxes of a page in one call: every
    xyxy box is truncated to int32, its snippet ``image[y0 : y0 + h, x0 : x0 + w]`` (numpy slicing: clipped to the page) is measured
    and the box shrunk by the snippet's offset."""
    if len(bboxes) == 0:
        return list(bboxes)
    ib = np.asarray(bboxes, np.float32).reshape(-1, 4).astype(np.int32)
    x0, y0 = np.clip(ib[:, 0], 0, page_w), np.clip(ib[:, 1], 0, page_h)
    # numpy's image[y0 : y0 + h, x0 : x0 + w] with a non-negative start and a stop clipped to the page
    w = np.clip(np.minimum(ib[:, 2], page_w) - x0, 0, None)
    h = np.clip(np.minimum(ib[:, 3], page_h) - y0, 0, None)
    ext = content_extents(ctx, page_dev_ptr, page_h, page_w, np.stack([x0, y0, w, h], axis=1), content_aware)
    out = []
    for box, e, sw, sh in zip(ib, ext, w, h):
        off = box_offset(e, int(sw), int(sh), content_aware) if sw > 0 and sh > 0 else [0, 0, 0, 0]
        out.append([box[0] + off[0], box[1] + off[1], box[2] - (off[2] - off[0]), box[3] - (off[3] - off[1])])
    return out
