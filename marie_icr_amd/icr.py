"""Production ICR recognizer (TPS-ResNet-BiLSTM-Attn) on MI355X behind the reference's ``CraftOcrProcessor`` surface.

Mirrors ``CraftOcrProcessor`` (reference: marie/document/craft_ocr_processor.py:26-286): same constructor arguments
(``work_dir, models_dir, cuda``), the checkpoint path ``<models_dir>/TPS-ResNet-BiLSTM-Attn-case-sensitive-ft/
best_accuracy.pth`` (:39-43), ``is_available()`` and ``recognize_from_fragments(images)``.  Crop batching
(Pillow-exact, img_w = 100), TPS rectification, ResNet-45, BiLSTM and the 49-step attention decoder run in
libmarie_hip.so; this file applies the reference's "[s]" cut / confidence rule to the per-step arg-max and softmax-max.

Deviation: a line whose decoded string starts with "[s]" yields ``{"text": "", "confidence": 0}`` for THAT line only; the
reference raises inside its batch loop there and drops the remaining lines of the batch (:253-285).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional, Sequence

import numpy as np

from ._lib import PREC_F16, PREC_F32, Context, MarieHipError, check
from .crnn import IMG_H, pack_fragments
from .ocr_processor import OcrProcessor
from .weights import CRNN_CHARSET, ICR_IMG_W, strip_module_prefix


class IcrModel:
    """Device-resident TPS-ResNet-BiLSTM-Attn weights + forward.  Thin handle over ``mhip_icr``."""

    def __init__(self, ctx: Context, state: Optional[Dict[str, np.ndarray]], num_class: int = 96,
                 precision: int = PREC_F16):
        self.ctx = ctx
        self.lib = ctx.lib
        self.num_class = int(num_class)
        self.precision = int(precision)
        self.steps = self.lib.mhip_icr_steps()
        h = C.c_void_p()
        check(ctx.h, self.lib.mhip_icr_create(ctx.h, self.precision, self.num_class, C.byref(h)), "mhip_icr_create")
        self.h = h
        ctx.adopt(self)
        if state is not None:
            self.load_state(state)

    def load_state(self, state: Dict[str, np.ndarray]):
        for key, val in strip_module_prefix(state).items():
            arr = np.ascontiguousarray(np.asarray(val), dtype=np.float32)
            shape = (C.c_int64 * max(arr.ndim, 1))(*arr.shape)
            check(self.ctx.h,
                  self.lib.mhip_icr_set_tensor(self.h, key.encode(), arr.ctypes.data_as(C.c_void_p), shape, arr.ndim),
                  f"mhip_icr_set_tensor({key})")
        check(self.ctx.h, self.lib.mhip_icr_finalize(self.h), "mhip_icr_finalize")

    def alloc_arena(self):
        check(self.ctx.h, self.lib.mhip_icr_alloc_arena(self.h), "mhip_icr_alloc_arena")

    def arena(self):
        p = C.c_void_p()
        n = C.c_size_t()
        check(self.ctx.h, self.lib.mhip_icr_arena(self.h, C.byref(p), C.byref(n)), "mhip_icr_arena")
        return p.value, n.value

    def forward_host(self, crops_u8: np.ndarray, want_logits: bool = False, want_rectified: bool = False):
        """crops_u8: (n, 32, 100) uint8.  Returns dict of host arrays: argmax (n,49), pmax (n,49) [, logits, rectified]."""
        crops = np.ascontiguousarray(crops_u8, dtype=np.uint8)
        if crops.ndim != 3 or crops.shape[1:] != (IMG_H, ICR_IMG_W):
            raise ValueError(f"crops must be (n, {IMG_H}, {ICR_IMG_W}) uint8, got {crops.shape}")
        n = crops.shape[0]
        s = self.steps
        logits = np.empty((n, s, self.num_class), np.float32) if want_logits else None
        rect = np.empty((n, IMG_H, ICR_IMG_W), np.float32) if want_rectified else None
        argmax = np.empty((n, s), np.int32)
        pmax = np.empty((n, s), np.float32)
        if n:
            vp = lambda a: a.ctypes.data_as(C.c_void_p) if a is not None else C.c_void_p(0)  # noqa: E731
            check(self.ctx.h, self.lib.mhip_icr_forward_host(self.h, vp(crops), n, vp(logits), vp(argmax), vp(pmax),
                                                             vp(rect)), "mhip_icr_forward_host")
        return {"logits": logits, "argmax": argmax, "pmax": pmax, "rectified": rect}

    def close(self):
        if getattr(self, "h", None) is not None and self.h.value:
            self.lib.mhip_icr_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def attn_texts(argmax: np.ndarray, pmax: np.ndarray, charset: str):
    """The reference's decode + "[s]" cut + confidence (marie/document/craft_ocr_processor.py:244-272,
    AttnLabelConverter.decode marie/models/icr/utils.py:142-148): the cut position is found in the joined STRING and the
    same number is used to slice the per-step probabilities."""
    character = ["[GO]", "[s]"] + list(charset)
    texts: List[str] = []
    confs: List[float] = []
    for row, pm in zip(argmax.tolist(), pmax):
        pred = "".join(character[i] for i in row)
        eos = pred.find("[s]")
        pred, pm = pred[:eos], pm[:eos]
        if pm.size == 0:
            texts.append("")
            confs.append(0.0)
        else:
            conf = np.float32(1.0)
            for v in pm:                       # cumprod in fp32, left to right
                conf = np.float32(conf * v)
            texts.append(pred.upper())
            confs.append(float(conf))
    return texts, confs


class CraftOcrProcessor(OcrProcessor):
    """Drop-in for marie/document/craft_ocr_processor.py:26."""

    def __init__(self, work_dir: str = "/tmp/icr", models_dir: Optional[str] = None, cuda: bool = True, *,
                 state: Optional[Dict[str, np.ndarray]] = None, character: str = CRNN_CHARSET, precision: str = "f16",
                 device_id: int = 0, ctx: Optional[Context] = None, **kwargs) -> None:
        super().__init__(work_dir, cuda)
        if not cuda:
            raise MarieHipError("CraftOcrProcessor here is the MI355X path; cuda=False has no implementation")
        self.character = character
        if state is None:
            # craft_ocr_processor.py:30,38-42: models_dir defaults to <model zoo>/icr; looked up before a device context exists
            from .constants import __model_path__

            models_dir = os.path.join(__model_path__, "icr") if models_dir is None else models_dir
            path = os.path.join(models_dir, "TPS-ResNet-BiLSTM-Attn-case-sensitive-ft", "best_accuracy.pth")
            if not os.path.exists(path):
                raise FileNotFoundError(f"File not found : {path}")
        self.ctx = ctx or Context(device_id)
        self.batch_size = int(kwargs.get("batch_size", 2048))
        if state is None:
            import torch

            sd = torch.load(path, map_location="cpu", weights_only=True)
            state = {k: v.numpy() for k, v in sd.items()}
        prec = {"f16": PREC_F16, "fp16": PREC_F16, "f32": PREC_F32, "fp32": PREC_F32}[precision]
        self.model = IcrModel(self.ctx, state, num_class=len(character) + 2, precision=prec)

    def is_available(self) -> bool:
        return self.model is not None

    def _crops(self, images: Sequence[np.ndarray]) -> np.ndarray:
        """GPU crop batcher (Pillow-exact bicubic to 32 x 100 + replicate pad) -> host uint8 (n, 32, 100)."""
        import torch

        packed, descs = pack_fragments(images)
        d_in = torch.from_numpy(packed).cuda()
        d_out = torch.empty((len(images), IMG_H, ICR_IMG_W), dtype=torch.uint8, device="cuda")
        self.ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        check(self.ctx.h, self.ctx.lib.mhip_crop_batch(self.ctx.h, C.c_void_p(d_in.data_ptr()), descs, len(images),
                                                       ICR_IMG_W, C.c_void_p(d_out.data_ptr())), "mhip_crop_batch")
        torch.cuda.synchronize()
        return d_out.cpu().numpy()

    def recognize_from_fragments(self, images, **kwargs) -> List[Dict[str, object]]:
        results: List[Dict[str, object]] = []
        for start in range(0, len(images), self.batch_size):
            batch = images[start:start + self.batch_size]
            out = self.model.forward_host(self._crops(batch))
            texts, confs = attn_texts(out["argmax"], out["pmax"], self.character)
            for k, (text, conf) in enumerate(zip(texts, confs)):
                results.append({"confidence": conf, "text": text, "id": f"img-{start + k}"})
        return results
