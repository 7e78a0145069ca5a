"""Page ingest — the step before the detector.

reference: ``ensure_max_page_size`` (marie/utils/image_utils.py:254-321, called by the extraction executor at
marie/executor/text/text_extraction_executor.py:139), ``load_image`` / ``frames_from_file`` / ``convert_frames``
(marie/utils/docs.py:184-256,372-379).

* ``ensure_max_page_size`` keeps the reference's signature and return value; an oversized frame is shrunk by the HIP
  INTER_AREA kernel (``mhip_resize_area_u8``), the size rule is the library's ``mhip_max_page_size``.
* ``load_image`` / ``frames_from_file`` burst TIFFs (all pages) and read single images with Pillow and return RGB frames,
  as the reference does after its ``convert_frames``.  PDFs raise: the reference extracts embedded images with PyPDF4,
  which is not part of this path.
* ``PageFeeder`` moves frames host -> HBM through two pinned staging buffers on a copy stream of its own, so the copy of
  batch i + 1 runs under the kernels of batch i (25 MB per 2550 x 3300 page).

There is no CPU fallback: without the library / a GPU every function here raises ``MarieHipError``.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Iterable, Iterator, List, Optional, Sequence, Tuple

import numpy as np

from ._lib import Context, MarieHipError, check, load


def _vp(a):
    return a.ctypes.data_as(C.c_void_p)


def page_size_rule(width: int, height: int, max_page_size: Tuple[int, int] = (2550, 3300), expand_ratio: float = 0.15):
    """The size rule for one frame -> (changed, new_width, new_height).  Host only."""
    nw, nh = C.c_int(), C.c_int()
    changed = load().mhip_max_page_size(int(width), int(height), int(max_page_size[0]), int(max_page_size[1]),
                                        float(expand_ratio), C.byref(nw), C.byref(nh))
    return bool(changed), nw.value, nh.value


def resize_area(ctx: Context, frame: np.ndarray, new_width: int, new_height: int) -> np.ndarray:
    """cv2.resize(frame, (new_width, new_height), interpolation=cv2.INTER_AREA) for uint8 HxW / HxWx3 frames (shrink)."""
    if frame.dtype != np.uint8 or frame.ndim not in (2, 3) or (frame.ndim == 3 and frame.shape[2] not in (1, 3)):
        raise ValueError(f"resize_area: uint8 HxW or HxWx3 frames only, got {frame.dtype} {frame.shape}")
    src = np.ascontiguousarray(frame)
    cn = 1 if src.ndim == 2 else src.shape[2]
    out = np.empty((int(new_height), int(new_width)) + (() if src.ndim == 2 else (cn,)), np.uint8)
    check(ctx.h, ctx.lib.mhip_resize_area_u8_host(ctx.h, _vp(src), src.shape[0], src.shape[1], cn, _vp(out), out.shape[0],
                                                  out.shape[1]), "mhip_resize_area_u8_host")
    return out


def ensure_max_page_size(frames: List[np.ndarray], max_page_size: Tuple[int, int] = (2550, 3300),
                         expand_ratio: float = 0.15, ctx: Optional[Context] = None) -> Tuple[bool, List[np.ndarray]]:
    """Ensure frames do not exceed the max page size (portrait ``(width, height)``, swapped for landscape frames, grown by
    ``expand_ratio``).  Returns ``(changed, frames)``; frames within the limit are returned as they came."""
    out: List[np.ndarray] = []
    changed = False
    for frame in frames:
        height, width = frame.shape[:2]
        ch, nw, nh = page_size_rule(width, height, max_page_size, expand_ratio)
        if ch:
            changed = True
            if ctx is None:
                ctx = _default_ctx()
            out.append(resize_area(ctx, frame, nw, nh))
        else:
            out.append(frame)
    return changed, out


_ctx: Optional[Context] = None


def _default_ctx() -> Context:
    global _ctx
    if _ctx is None:
        _ctx = Context(int(os.environ.get("LOCAL_RANK", "0")))
    return _ctx


# ---------------------------------------------------------------------------------------------- files -> frames
def get_document_type(file_path: str) -> str:
    """docs.py:28-52: by extension."""
    ext = os.path.splitext(str(file_path))[1].lower().lstrip(".")
    if ext == "pdf":
        return "pdf"
    if ext in ("tif", "tiff"):
        return "tiff"
    return "image"


def convert_frames(frames: Sequence[np.ndarray], img_format: str = "cv") -> List[np.ndarray]:
    """docs.py:184-199: every frame becomes HxWx3 (gray frames are replicated).  Frames are RGB on both sides here (the
    reference's TIFF reader hands BGR frames to this function and swaps them)."""
    out = []
    for f in frames:
        f = np.asarray(f)
        out.append(np.repeat(f[:, :, None], 3, axis=2) if f.ndim == 2 else f[:, :, :3].copy())
    if img_format == "pil":
        from PIL import Image
        return [Image.fromarray(f) for f in out]
    return out


def load_image(img_path, img_format: str = "cv"):
    """docs.py:202-256 -> ``(loaded, frames)``: every page of a TIFF, or the single image, as RGB uint8 HxWx3."""
    if img_path is None:
        return False, None
    kind = get_document_type(img_path)
    if kind == "pdf":
        raise NotImplementedError("PDF bursting (PyPDF4 image extraction, docs.py:122-181) is outside this path")
    from PIL import Image, ImageSequence
    try:
        with Image.open(img_path) as im:
            if kind == "tiff":
                frames = [np.array(page.convert("RGB"), dtype=np.uint8) for page in ImageSequence.Iterator(im)]
            else:
                frames = [np.array(im.convert("RGB"), dtype=np.uint8)]
    except (OSError, ValueError):
        return False, []
    if not frames:
        return False, []
    return True, convert_frames(frames, img_format)


def frames_from_file(img_path) -> List[np.ndarray]:
    """docs.py:372-379."""
    if not os.path.exists(img_path):
        raise FileNotFoundError(f"File not found : {img_path}")
    loaded, frames = load_image(img_path)
    if not loaded:
        raise Exception(f"Unable to load image : {img_path}")
    return frames


# ---------------------------------------------------------------------------------------------- host -> HBM feeder
class PageFeeder:
    """Double-buffered host -> device page feeder.

    ``for dev_ptr, shapes in PageFeeder(batches)`` yields, per batch of same-shaped uint8 frames, the device address of the
    packed batch ``[n][H][W][C]`` and its shape.  Two pinned host buffers and two device buffers alternate; the copy of the
    next batch is issued on the feeder's own stream before the current one is handed out, and the consumer's stream is
    made to wait on the copy's event, so no host synchronisation sits between copy and compute.  The device buffer of a
    batch is reused two batches later: the feeder records an event on the consumer stream when the next batch is
    requested and the copy stream waits on it.  ``consumer_stream`` may be a list of streams (detector and recognizer
    reading the same pages): each waits for the copy, and the buffer is recycled after all of them.
    """

    def __init__(self, batches: Iterable[Sequence[np.ndarray]], capacity_bytes: int, consumer_stream=None, device: int = 0):
        import torch

        if not torch.cuda.is_available():
            raise MarieHipError("PageFeeder needs a GPU")
        self.torch = torch
        self.dev = torch.device("cuda", device)
        self.batches = iter(batches)
        self.cap = int(capacity_bytes)
        self.copy_stream = torch.cuda.Stream(self.dev)
        cs = consumer_stream if consumer_stream is not None else torch.cuda.current_stream(self.dev)
        self.consumers = list(cs) if isinstance(cs, (list, tuple)) else [cs]
        self.host = [torch.empty(self.cap, dtype=torch.uint8).pin_memory() for _ in range(2)]
        self.devbuf = [torch.empty(self.cap, dtype=torch.uint8, device=self.dev) for _ in range(2)]
        self.copied = [torch.cuda.Event() for _ in range(2)]
        self.released = [None, None]
        self.slot = 0
        self.pending = None
        self.bytes_moved = 0

    def _issue(self):
        try:
            frames = next(self.batches)
        except StopIteration:
            return None
        t = self.torch
        s = self.slot
        first = np.asarray(frames[0])
        nbytes = first.nbytes * len(frames)
        if nbytes > self.cap:
            raise ValueError(f"PageFeeder: batch of {nbytes} bytes exceeds capacity {self.cap}")
        self.copied[s].synchronize()                      # the pinned buffer's previous copy has left the host
        hv = self.host[s].numpy()[:nbytes].reshape((len(frames),) + first.shape)
        for i, f in enumerate(frames):
            if f.shape != first.shape or f.dtype != np.uint8:
                raise ValueError("PageFeeder: frames of one batch must share shape and be uint8")
            hv[i] = f
        with t.cuda.stream(self.copy_stream):
            for ev in self.released[s] or ():
                self.copy_stream.wait_event(ev)                    # every consumer is done with this device buffer
            self.devbuf[s][:nbytes].copy_(self.host[s][:nbytes], non_blocking=True)
            self.copied[s].record(self.copy_stream)
        self.bytes_moved += nbytes
        self.slot ^= 1
        return s, (len(frames),) + first.shape

    def __iter__(self) -> Iterator[Tuple[int, Tuple[int, ...]]]:
        nxt = self._issue()
        prev_slot = None
        while nxt is not None:
            s, shape = nxt
            if prev_slot is not None:                     # everything queued on the consumer so far used prev_slot
                evs = []
                for c in self.consumers:
                    ev = self.torch.cuda.Event()
                    ev.record(c)
                    evs.append(ev)
                self.released[prev_slot] = evs
            nxt = self._issue()                           # batch i + 1 starts moving while batch i is consumed
            for c in self.consumers:
                c.wait_event(self.copied[s])
            yield self.devbuf[s].data_ptr(), shape
            prev_slot = s
