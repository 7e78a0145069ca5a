"""CRNN-family recognizer (None-VGG-BiLSTM-CTC) on MI355X behind the reference's
``OcrProcessor`` surface.

Mirrors ``CraftOcrProcessor`` (reference: marie/document/craft_ocr_processor.py:26-286):
same constructor arguments, ``is_available()``, and
``recognize_from_fragments(images) -> [{"confidence", "id": "img-<k>", "text"}]``
with upper-cased text in input order.  All arithmetic of the path — normalise,
conv stack, BiLSTM, prediction, greedy CTC decode, confidence — runs in
libmarie_hip.so; this file only packs inputs and turns token ids into strings.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional, Sequence

import numpy as np

from . import _lib
from ._lib import PREC_F16, PREC_F32, Context, MarieHipError, check
from .ocr_processor import OcrProcessor
from .weights import CRNN_CHARSET, strip_module_prefix

IMG_H = 32


class CrnnModel:
    """Device-resident recognizer weights + forward.  Thin handle over ``mhip_crnn``."""

    def __init__(self, ctx: Context, state: Optional[Dict[str, np.ndarray]], num_class: int,
                 precision: int = PREC_F16):
        self.ctx = ctx
        self.lib = ctx.lib
        self.num_class = int(num_class)
        self.precision = int(precision)
        h = C.c_void_p()
        check(ctx.h, self.lib.mhip_crnn_create(ctx.h, self.precision, self.num_class, C.byref(h)), "mhip_crnn_create")
        self.h = h
        ctx.adopt(self)
        if state is not None:
            self.load_state(state)

    # -- weights ------------------------------------------------------------------------
    def load_state(self, state: Dict[str, np.ndarray]):
        """reference: ``model.load_state_dict(torch.load(...))`` craft_ocr_processor.py:146."""
        for key, val in strip_module_prefix(state).items():
            arr = np.ascontiguousarray(np.asarray(val), dtype=np.float32)
            shape = (C.c_int64 * max(arr.ndim, 1))(*arr.shape)
            check(self.ctx.h,
                  self.lib.mhip_crnn_set_tensor(self.h, key.encode(), arr.ctypes.data_as(C.c_void_p), shape, arr.ndim),
                  f"mhip_crnn_set_tensor({key})")
        check(self.ctx.h, self.lib.mhip_crnn_finalize(self.h), "mhip_crnn_finalize")

    def alloc_arena(self):
        check(self.ctx.h, self.lib.mhip_crnn_alloc_arena(self.h), "mhip_crnn_alloc_arena")

    def arena(self):
        """(device pointer, bytes) of the single packed weight arena (RCCL broadcast unit)."""
        p = C.c_void_p()
        n = C.c_size_t()
        check(self.ctx.h, self.lib.mhip_crnn_arena(self.h, C.byref(p), C.byref(n)), "mhip_crnn_arena")
        return p.value, n.value

    # -- forward --------------------------------------------------------------------------
    def seq_len(self, w: int) -> int:
        return self.lib.mhip_crnn_seq_len(int(w))

    def forward_host(self, crops_u8: np.ndarray, want_logits: bool = False):
        """crops_u8: (n, 32, w) uint8 host array.  Returns dict of host numpy arrays."""
        crops = np.ascontiguousarray(crops_u8, dtype=np.uint8)
        if crops.ndim != 3 or crops.shape[1] != IMG_H:
            raise ValueError(f"crops must be (n, {IMG_H}, w) uint8, got {crops.shape}")
        n, _, w = crops.shape
        t = self.seq_len(w)
        if n == 0:
            z = np.zeros((0, max(t, 0)), np.int32)
            return {"logits": np.zeros((0, max(t, 0), self.num_class), np.float32) if want_logits else None,
                    "argmax": z, "tokens": z.copy(), "lengths": np.zeros((0,), np.int32),
                    "confidence": np.zeros((0,), np.float32)}
        logits = np.empty((n, t, self.num_class), np.float32) if want_logits else None
        argmax = np.empty((n, t), np.int32)
        tokens = np.empty((n, t), np.int32)
        lengths = np.empty((n,), np.int32)
        conf = np.empty((n,), np.float32)
        vp = lambda a: a.ctypes.data_as(C.c_void_p) if a is not None else C.c_void_p(0)  # noqa: E731
        check(self.ctx.h,
              self.lib.mhip_crnn_forward_host(self.h, vp(crops), n, w, vp(logits), vp(argmax), vp(tokens),
                                              vp(lengths), vp(conf)),
              "mhip_crnn_forward_host")
        return {"logits": logits, "argmax": argmax, "tokens": tokens, "lengths": lengths, "confidence": conf}

    def _outputs(self, n: int, w: int, want_logits: bool):
        t = self.seq_len(w)
        return (np.empty((n, t, self.num_class), np.float32) if want_logits else None, np.empty((n, t), np.int32),
                np.empty((n, t), np.int32), np.empty((n,), np.int32), np.empty((n,), np.float32))

    def forward_fragments_host(self, images: Sequence[np.ndarray], img_w: int, want_logits: bool = False):
        """Arbitrary-size BGR/gray fragments -> GPU crop batcher (Pillow-exact bicubic to 32 x img_w) -> forward."""
        n = len(images)
        logits, argmax, tokens, lengths, conf = self._outputs(n, img_w, want_logits)
        if n:
            packed, descs = pack_fragments(images)
            vp = lambda a: a.ctypes.data_as(C.c_void_p) if a is not None else C.c_void_p(0)  # noqa: E731
            check(self.ctx.h,
                  self.lib.mhip_crnn_forward_fragments_host(self.h, vp(packed), packed.size, descs, n, int(img_w),
                                                            vp(logits), vp(argmax), vp(tokens), vp(lengths), vp(conf)),
                  "mhip_crnn_forward_fragments_host")
        return {"logits": logits, "argmax": argmax, "tokens": tokens, "lengths": lengths, "confidence": conf}

    def forward_rects_device(self, page_ptr: int, page_h: int, page_w: int, rects_xywh: np.ndarray, img_w: int,
                             channels: int = 3, inclusive: bool = True, want_logits: bool = False):
        """Fragments given as boxes on a page that already lives in HBM (uint8 [page_h][page_w][channels]).
        ``inclusive`` reproduces the detector's crop rule rows y..y+h, cols x..x+w (craft_box_processor.py:42-73)."""
        from ._lib import CropDesc

        rects = np.asarray(rects_xywh, np.int64).reshape(-1, 4)
        n = len(rects)
        logits, argmax, tokens, lengths, conf = self._outputs(n, img_w, want_logits)
        if n:
            descs = (CropDesc * n)()
            ext = 1 if inclusive else 0
            for i, (x, y, w, h) in enumerate(rects.tolist()):
                x1, y1 = min(page_w, x + w + ext), min(page_h, y + h + ext)
                x, y = max(0, x), max(0, y)
                if x1 <= x or y1 <= y:
                    raise ValueError(f"rect {i} is empty on a {page_w}x{page_h} page")
                descs[i] = CropDesc((y * page_w + x) * channels, y1 - y, x1 - x, page_w * channels, channels)
            vp = lambda a: a.ctypes.data_as(C.c_void_p) if a is not None else C.c_void_p(0)  # noqa: E731
            check(self.ctx.h,
                  self.lib.mhip_crnn_forward_crops(self.h, C.c_void_p(page_ptr), descs, n, int(img_w), vp(logits),
                                                   vp(argmax), vp(tokens), vp(lengths), vp(conf)),
                  "mhip_crnn_forward_crops")
        return {"logits": logits, "argmax": argmax, "tokens": tokens, "lengths": lengths, "confidence": conf}

    def forward_device(self, crops_ptr: int, n: int, w: int, logits_ptr: int, argmax_ptr: int, tokens_ptr: int,
                       lengths_ptr: int, conf_ptr: int):
        """All-device entry (pointers are HBM addresses); enqueues on the ctx stream, no sync."""
        check(self.ctx.h,
              self.lib.mhip_crnn_forward(self.h, C.c_void_p(crops_ptr), int(n), int(w), C.c_void_p(logits_ptr or 0),
                                         C.c_void_p(argmax_ptr), C.c_void_p(tokens_ptr), C.c_void_p(lengths_ptr),
                                         C.c_void_p(conf_ptr)),
              "mhip_crnn_forward")

    def kernel_flops(self, n: int, w: int) -> Dict[str, float]:
        return {self.lib.mhip_kernel_name(k).decode(): self.lib.mhip_crnn_kernel_flops(self.h, k, int(n), int(w))
                for k in range(self.lib.mhip_kernel_count())}

    def workspace_bytes(self, n: int, w: int) -> int:
        return int(self.lib.mhip_crnn_workspace_bytes(self.h, int(n), int(w)))

    def close(self):
        if getattr(self, "h", None) is not None and self.h.value:
            self.lib.mhip_crnn_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def tokens_to_text(tokens: np.ndarray, lengths: np.ndarray, charset: str) -> List[str]:
    """Collapsed CTC token ids -> upper-cased strings.
    reference: CTCLabelConverter.character table (marie/models/icr/utils.py:19) and ``pred.upper()``
    (marie/document/craft_ocr_processor.py:272)."""
    table = [""] + list(charset)  # index 0 = CTC blank, never present in `tokens`
    return ["".join(table[t] for t in row[:ln]).upper() for row, ln in zip(tokens.tolist(), lengths.tolist())]


def tokens_to_text_fast(tokens: np.ndarray, lengths: np.ndarray, charset: str) -> List[str]:
    """Same as ``tokens_to_text`` for ASCII charsets, vectorised: one table lookup for the whole
    batch, then a bytes slice per line (serving loop: ~0.3 ms per 1024 lines)."""
    table = np.frombuffer((" " + charset.upper()).encode("ascii"), dtype=np.uint8)
    chars = table[tokens]                       # (n, T) uint8
    raw = chars.tobytes()
    t = tokens.shape[1]
    return [raw[i * t:i * t + int(ln)].decode("ascii") for i, ln in enumerate(lengths)]


def pack_fragments(images: Sequence[np.ndarray]):
    """Pack HxWx3 (BGR) / HxW (gray) uint8 fragments back to back for one H2D copy.
    Returns (packed uint8 buffer, ctypes array of CropDesc)."""
    from ._lib import CropDesc

    n = len(images)
    descs = (CropDesc * n)()
    arrs = []
    off = 0
    for i, im in enumerate(images):
        a = np.asarray(im)
        if a.dtype != np.uint8 or a.ndim not in (2, 3) or (a.ndim == 3 and a.shape[2] != 3) or a.size == 0:
            raise ValueError(f"fragment {i} must be a non-empty uint8 HxWx3 or HxW array, got {a.dtype} {a.shape}")
        ch = 3 if a.ndim == 3 else 1
        h, w = a.shape[:2]
        descs[i] = CropDesc(off, h, w, w * ch, ch)
        arrs.append(a)
        off += h * w * ch
    packed = np.empty(off, np.uint8)
    for d, a in zip(descs, arrs):
        packed[d.src_offset:d.src_offset + a.size] = a.reshape(-1)
    return packed, descs


class CrnnOcrProcessor(OcrProcessor):
    """Drop-in for the reference's CRNN-family ``OcrProcessor``
    (marie/document/craft_ocr_processor.py:26; abstract surface marie/document/ocr_processor.py:34-96).

    ``state`` (a reference-format state_dict of numpy arrays) may be passed directly; otherwise
    ``models_dir/<model_name>/best_accuracy.pth`` is read with ``torch.load(weights_only=True)``.
    """

    def __init__(self, work_dir: str = "/tmp/icr", models_dir: Optional[str] = None, cuda: bool = True,
                 *, state: Optional[Dict[str, np.ndarray]] = None, character: str = CRNN_CHARSET,
                 img_w: int = 256, precision: str = "f16", device_id: int = 0,
                 model_name: str = "None-VGG-BiLSTM-CTC", ctx: Optional[Context] = None, **kwargs) -> None:
        super().__init__(work_dir, cuda)
        if not cuda:
            raise MarieHipError("CrnnOcrProcessor is the MI355X path; cuda=False has no implementation here")
        if img_w % 4 or img_w < 8:
            raise ValueError("img_w must be a multiple of 4 and >= 8")
        self.work_dir = work_dir
        self.cuda = cuda
        self.character = character
        self.img_w = int(img_w)
        self.batch_size = int(kwargs.get("batch_size", 4096))
        self.ctx = ctx or Context(device_id)
        if state is None:
            if models_dir is None:
                raise ValueError("either `state` or `models_dir` is required")
            import torch

            path = os.path.join(models_dir, model_name, "best_accuracy.pth")
            sd = torch.load(path, map_location="cpu", weights_only=True)
            state = {k: v.numpy() for k, v in sd.items()}
        prec = {"f16": PREC_F16, "fp16": PREC_F16, "f32": PREC_F32, "fp32": PREC_F32}[precision]
        self.model = CrnnModel(self.ctx, state, num_class=len(character) + 1, precision=prec)

    def is_available(self) -> bool:
        return self.model is not None

    def recognize_from_fragments(self, images, **kwargs) -> List[Dict[str, object]]:
        """reference: marie/document/craft_ocr_processor.py:184-286."""
        results: List[Dict[str, object]] = []
        for start in range(0, len(images), self.batch_size):
            batch = images[start:start + self.batch_size]
            out = self.model.forward_fragments_host(batch, self.img_w)
            texts = tokens_to_text(out["tokens"], out["lengths"], self.character)
            for k, (text, conf) in enumerate(zip(texts, out["confidence"].tolist())):
                results.append({"confidence": conf, "text": text, "id": f"img-{start + k}"})
        return results
