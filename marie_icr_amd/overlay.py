"""``OverlayProcessor`` — the reference's form-overlay cleaner surface over the MI355X generator.

Mirrors marie/overlay/overlay.py:29-372: ``segment(document_id, img_path, checksum)`` and ``segment_frame(document_id, frame)``
return ``(original image, generated mask (BGR), blended text-only image)``; ``preprocess`` (white canvas, both sides to the next
multiple of 32), ``postprocess`` and ``blend_to_text`` keep their names and results.  The reference writes the page to a PNG,
reads it back through a dataset, runs pix2pixHD's ``LocalEnhancer`` and converts tensors to images on the host; here the page
goes to HBM once and the generator, the tensor -> image conversion and the blend run in libmarie_hip.so (overlay_api.hip).

``state``: the generator's state_dict (``netG``: ``model.1.weight_orig`` ...) as numpy arrays; otherwise
``models_dir/overlay/claim_mask/latest_net_G.pth`` is read with ``torch.load(weights_only=True)``.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Optional, Tuple

import numpy as np

from ._lib import PREC_F16, PREC_F32, Context, MarieHipError, check
from .vit import load_tensors


def _vp(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else C.c_void_p(0)


class OverlayModel:
    """Thin handle over ``mhip_overlay``."""

    def __init__(self, ctx: Context, state: Dict[str, np.ndarray], ngf: int = 64, precision: int = PREC_F16):
        self.ctx, self.lib, self.ngf, self.precision = ctx, ctx.lib, int(ngf), int(precision)
        h = C.c_void_p()
        check(ctx.h, self.lib.mhip_overlay_create(ctx.h, self.precision, self.ngf, C.byref(h)), "mhip_overlay_create")
        self.h = h
        ctx.adopt(self)
        load_tensors(ctx, self.lib.mhip_overlay_set_tensor, self.h, state, "mhip_overlay_set_tensor")
        check(ctx.h, self.lib.mhip_overlay_finalize(self.h), "mhip_overlay_finalize")

    def padded_shape(self, h: int, w: int) -> Tuple[int, int]:
        H, W = C.c_int(), C.c_int()
        self.lib.mhip_overlay_padded_shape(h, w, C.byref(H), C.byref(W))
        return H.value, W.value

    def forward_host(self, page_bgr: np.ndarray, want_raw: bool = False):
        """page (h, w, 3) uint8 BGR -> the generator's image (H, W, 3) uint8 RGB on the padded canvas [, raw tanh fp32]."""
        page = np.ascontiguousarray(page_bgr, np.uint8)
        h, w = page.shape[:2]
        H, W = self.padded_shape(h, w)
        fake = np.empty((H, W, 3), np.uint8)
        raw = np.empty((H, W, 3), np.float32) if want_raw else None
        check(self.ctx.h, self.lib.mhip_overlay_forward_host(self.h, _vp(page), h, w, _vp(fake), _vp(raw)), "mhip_overlay_forward_host")
        return (fake, raw) if want_raw else fake

    def forward_device(self, page_ptr: int, h: int, w: int, fake_ptr: int):
        check(self.ctx.h, self.lib.mhip_overlay_forward(self.h, C.c_void_p(page_ptr), h, w, C.c_void_p(fake_ptr)), "mhip_overlay_forward")

    def close(self):
        if getattr(self, "h", None) is not None and self.h.value:
            self.lib.mhip_overlay_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class OverlayProcessor:
    """Drop-in for marie/overlay/overlay.py:29."""

    def __init__(self, work_dir: str, models_dir: Optional[str] = None, cuda: bool = True, *,
                 state: Optional[Dict[str, np.ndarray]] = None, ngf: int = 64, precision: str = "f16", device_id: int = 0,
                 ctx: Optional[Context] = None, **kwargs) -> None:
        if not cuda:
            raise MarieHipError("OverlayProcessor here is the MI355X path; cuda=False has no implementation")
        self.cuda, self.models_dir, self.work_dir = cuda, models_dir, work_dir
        self.ctx = ctx or Context(device_id)
        if state is None:
            if models_dir is None:
                raise ValueError("either `state` or `models_dir` is required")
            import torch

            path = os.path.join(models_dir, "overlay", "claim_mask", "latest_net_G.pth")      # overlay.py:38,59 + base_model.load_networks
            sd = torch.load(path, map_location="cpu", weights_only=True)
            state = {k: v.float().numpy() for k, v in sd.items() if hasattr(v, "numpy")}
        prec = {"f16": PREC_F16, "fp16": PREC_F16, "f32": PREC_F32, "fp32": PREC_F32}[precision]
        self.model = OverlayModel(self.ctx, state, ngf, prec)
        self.initialized = False

    # ---- reference surface -------------------------------------------------------------------------------------------
    def preprocess(self, img: np.ndarray) -> np.ndarray:
        """overlay.py:147-163."""
        if len(img.shape) != 3:
            raise Exception("Image must be 3 channel")
        oh, ow, channels = img.shape
        if ow % 32 != 0 or oh % 32 != 0:
            h, w = oh // 32 * 32 + 32, ow // 32 * 32 + 32
            overlay = np.ones((h, w, channels), dtype=np.uint8) * 255
            overlay[:oh, :ow, :] = img
            return overlay
        return img

    def _run(self, src_img: np.ndarray):
        """page -> (real on the padded canvas, generator image RGB, blended) — one upload, everything else on the device."""
        import torch

        self.ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        real = self.preprocess(src_img)
        H, W = real.shape[:2]
        d_real = torch.from_numpy(np.ascontiguousarray(real)).cuda()
        d_fake = torch.empty((H, W, 3), dtype=torch.uint8, device="cuda")
        d_out = torch.empty((H, W, 3), dtype=torch.uint8, device="cuda")
        self.model.forward_device(d_real.data_ptr(), H, W, d_fake.data_ptr())
        check(self.ctx.h, self.ctx.lib.mhip_overlay_blend(self.ctx.h, C.c_void_p(d_real.data_ptr()), C.c_void_p(d_fake.data_ptr()),
                                                         C.c_void_p(d_out.data_ptr()), H * W), "mhip_overlay_blend")
        torch.cuda.current_stream().synchronize()
        return real, d_fake.cpu().numpy(), d_out.cpu().numpy()

    def postprocess(self, src_img: np.ndarray, real_img: np.ndarray, fake_mask: np.ndarray):
        """overlay.py:191-245 on host arrays (``fake_mask``: the generator's image, RGB): (mask BGR, blended), cropped to src."""
        fake_bgr = np.ascontiguousarray(fake_mask[:, :, ::-1])
        if real_img.shape != fake_bgr.shape:
            h, w = min(real_img.shape[0], fake_bgr.shape[0]), min(real_img.shape[1], fake_bgr.shape[1])
            real_img, fake_bgr, fake_mask = real_img[:h, :w], fake_bgr[:h, :w], fake_mask[:h, :w]
        blended = self.blend_to_text(real_img, fake_mask)
        return fake_bgr[: src_img.shape[0], : src_img.shape[1]], blended[: src_img.shape[0], : src_img.shape[1]]

    def blend_to_text(self, real_img: np.ndarray, mask_img: np.ndarray) -> np.ndarray:
        """overlay.py:247-291 (device kernel): the generator's image is read in its own channel order, as the reference does."""
        import torch

        if real_img.shape != mask_img.shape:
            raise Exception(f"Sizes of input arguments do not match(real, fake) : {real_img.shape} != {mask_img.shape}")
        self.ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        d_real = torch.from_numpy(np.ascontiguousarray(real_img)).cuda()
        d_mask = torch.from_numpy(np.ascontiguousarray(mask_img)).cuda()
        d_out = torch.empty_like(d_real)
        check(self.ctx.h, self.ctx.lib.mhip_overlay_blend(self.ctx.h, C.c_void_p(d_real.data_ptr()), C.c_void_p(d_mask.data_ptr()),
                                                         C.c_void_p(d_out.data_ptr()), real_img.shape[0] * real_img.shape[1]),
              "mhip_overlay_blend")
        torch.cuda.current_stream().synchronize()
        return d_out.cpu().numpy()

    def segment_frame(self, document_id: str, frame: np.ndarray) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        """overlay.py:353-372 without the PNG round trip: (frame, generated mask BGR, blended), both cropped to the frame."""
        if len(frame.shape) != 3:
            raise Exception("Expected image shape is h,w,c")
        real, fake_rgb, blended = self._run(frame)
        h, w = frame.shape[:2]
        return frame, np.ascontiguousarray(fake_rgb[:h, :w, ::-1]), blended[:h, :w]

    def segment(self, document_id: str, img_path: str, checksum: str = None, raise_oom: bool = False):
        """overlay.py:291-352: the file is decoded (cv2.imread order: BGR) and handed to ``segment_frame``."""
        if not os.path.exists(img_path):
            raise Exception("File not found : {}".format(img_path))
        from PIL import Image

        with Image.open(img_path) as im:
            src = np.array(im.convert("RGB"), dtype=np.uint8)[:, :, ::-1].copy()
        return self.segment_frame(document_id, src)
