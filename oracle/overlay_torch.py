"""ORACLE (test infrastructure, not product code) — CPU fp32 restatement of the reference's overlay cleaner:
``OverlayProcessor`` (marie/overlay/overlay.py:29-372) over the pix2pixHD ``LocalEnhancer`` generator
(marie/models/pix2pix/models/networks_hd.py:24-106, GlobalGenerator :109-160, ResnetBlock :165-213; built by
networks.py:189-196 with ngf 64, instance norm).

PINNED: the generator forward is checked against goldens written by the reference's own ``LocalEnhancer`` class
(tests/golden/overlay_*.npz, oracle/gen_golden.py --overlay-only loads networks_hd.py by path).
PARITY UNPINNED: the OpenCV pixel operations of ``blend_to_text`` (8-bit BGR2HSV, inRange, BGR2GRAY) — opencv-python is not
installed and the reference holds no fixture; restated from OpenCV's published integer formulas (color_hsv.cpp RGB2HSV_b,
color_rgb RGB2Gray with 14-bit coefficients).
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this module.
"""
from __future__ import annotations

from typing import Dict, Tuple

import numpy as np
import torch
import torch.nn.functional as F


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def effective_weight(st: Dict[str, np.ndarray], name: str, transposed: bool = False) -> torch.Tensor:
    """torch.nn.utils.spectral_norm in eval mode: weight_orig / (u^T W_mat v); W_mat has the output dim first (dim 1 of a
    ConvTranspose2d weight)."""
    w = _t(st[name + ".weight_orig"]).float()
    mat = (w.permute(1, 0, 2, 3) if transposed else w).reshape(w.shape[1] if transposed else w.shape[0], -1)
    sigma = torch.dot(_t(st[name + ".weight_u"]).float(), mat @ _t(st[name + ".weight_v"]).float())
    return w / sigma


def swish(x):
    return x * torch.sigmoid(x)


class TorchOverlayOracle:
    def __init__(self, state: Dict[str, np.ndarray]):
        self.st = state
        self.w = {}
        for k in state:
            if k.endswith(".weight_orig"):
                n = k[: -len(".weight_orig")]
                self.w[n] = effective_weight(state, n, transposed=(n == "model1_2.3"))
        self.b = {k[: -len(".bias")]: _t(v).float() for k, v in state.items() if k.endswith(".bias")}

    def _conv7(self, x, n):
        return F.conv2d(F.pad(x, (3, 3, 3, 3), mode="reflect"), self.w[n], self.b[n])

    def _res(self, x, n):
        h = F.conv2d(F.pad(x, (1, 1, 1, 1), mode="reflect"), self.w[n + ".conv_block.1"], self.b[n + ".conv_block.1"])
        h = swish(F.instance_norm(h, eps=1e-5))
        h = F.conv2d(F.pad(h, (1, 1, 1, 1), mode="reflect"), self.w[n + ".conv_block.5"], self.b[n + ".conv_block.5"])
        return x + F.instance_norm(h, eps=1e-5)

    @torch.no_grad()
    def generator(self, x: torch.Tensor) -> torch.Tensor:
        """x (1, 3, H, W) fp32 in [-1, 1], H and W multiples of 32 -> (1, 3, H, W) tanh output."""
        IN = lambda t: F.instance_norm(t, eps=1e-5)
        half = F.conv2d(x, _t(self.st["downsample.weight"]).float(), _t(self.st["downsample.bias"]).float(), stride=2, padding=1)
        g = swish(IN(self._conv7(half, "model.1")))
        for n in ("model.4", "model.7", "model.10"):
            g = swish(IN(F.conv2d(g, self.w[n], self.b[n], stride=2, padding=1)))
        for b in range(9):
            g = self._res(g, f"model.{13 + b}")
        for n in ("model.23", "model.26", "model.29"):
            g = F.interpolate(g, scale_factor=2, mode="bilinear", align_corners=True)
            g = swish(F.conv2d(g, self.w[n], self.b[n], padding=1))
        l = swish(IN(self._conv7(x, "model1_1.1")))
        l = swish(IN(F.conv2d(l, self.w["model1_1.4"], self.b["model1_1.4"], stride=2, padding=1)))
        y = l + g
        for b in range(3):
            y = self._res(y, f"model1_2.{b}")
        y = F.conv_transpose2d(y, self.w["model1_2.3"], self.b["model1_2.3"], stride=2, padding=1, output_padding=1)
        y = swish(IN(y))
        return torch.tanh(self._conv7(y, "model1_2.7"))


def preprocess(img_bgr: np.ndarray) -> np.ndarray:
    """OverlayProcessor.preprocess (overlay.py:147-163): pad to the next multiple of 32 on BOTH axes when either is ragged,
    white canvas, image at the top-left."""
    oh, ow, ch = img_bgr.shape
    if ow % 32 != 0 or oh % 32 != 0:
        h, w = oh // 32 * 32 + 32, ow // 32 * 32 + 32
        out = np.full((h, w, ch), 255, np.uint8)
        out[:oh, :ow] = img_bgr
        return out
    return img_bgr


def to_tensor(real_bgr: np.ndarray) -> torch.Tensor:
    """imwrite -> single dataset (PIL RGB) -> ToTensor -> Normalize(0.5, 0.5) (data/single_dataset.py, base_dataset.py)."""
    rgb = np.ascontiguousarray(real_bgr[:, :, ::-1]).astype(np.float32) / np.float32(255.0)
    return ((_t(rgb) - 0.5) / 0.5).permute(2, 0, 1).unsqueeze(0).contiguous()


def tensor2im(y: torch.Tensor) -> np.ndarray:
    """util/util.py:9-30: (x + 1) / 2 * 255, truncated to uint8, HWC RGB."""
    a = y[0].float().numpy()
    return ((np.transpose(a, (1, 2, 0)) + 1) / 2.0 * 255.0).astype(np.uint8)


def bgr2gray(img: np.ndarray) -> np.ndarray:
    i = img.astype(np.int64)
    return ((i[..., 0] * 1868 + i[..., 1] * 9617 + i[..., 2] * 4899 + 8192) >> 14).astype(np.uint8)


def bgr2hsv_u8(img: np.ndarray) -> np.ndarray:
    """OpenCV 8-bit BGR2HSV (H in [0, 180)): color_hsv.cpp RGB2HSV_b."""
    b, g, r = (img[..., k].astype(np.int64) for k in range(3))
    v = np.maximum(np.maximum(b, g), r)
    vmin = np.minimum(np.minimum(b, g), r)
    diff = v - vmin
    sdiv = np.zeros(256, np.int64)
    hdiv = np.zeros(256, np.int64)
    for i in range(1, 256):
        sdiv[i] = int(round((255 << 12) / (1.0 * i)))
        hdiv[i] = int(round((180 << 12) / (6.0 * i)))
    s = (diff * sdiv[v] + (1 << 11)) >> 12
    vr, vg = v == r, v == g
    h = np.where(vr, g - b, np.where(vg, b - r + 2 * diff, r - g + 4 * diff))
    h = (h * hdiv[diff] + (1 << 11)) >> 12
    h = h + np.where(h < 0, 180, 0)
    return np.stack([h, s, v], axis=-1).astype(np.uint8)


def blend_to_text(real_bgr: np.ndarray, mask_img: np.ndarray) -> np.ndarray:
    """overlay.py:247-291.  ``mask_img`` is what tensor2im returned (RGB order) and is fed to BGR2HSV / BGR2GRAY as it is —
    the reference does the same."""
    hsv = bgr2hsv_u8(mask_img)
    inr = (hsv[..., 1] >= 137) & (hsv[..., 2] >= 216) & (hsv[..., 0] <= 179)
    red = np.where(inr, 0, 255).astype(np.uint8)                 # bitwise_not(inRange); GRAY2BGR -> BGR2GRAY is the identity
    blended = (bgr2gray(real_bgr) | bgr2gray(mask_img)) & red
    return np.repeat(blended[:, :, None], 3, axis=2)


def segment_frame(oracle: TorchOverlayOracle, src_bgr: np.ndarray) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """OverlayProcessor.segment (overlay.py:291-352) on an in-memory frame: (src, fake mask BGR, blended), cropped to src."""
    real = preprocess(src_bgr)
    fake = tensor2im(oracle.generator(to_tensor(real)))           # RGB
    fake_bgr = np.ascontiguousarray(fake[:, :, ::-1])
    blended = blend_to_text(real, fake)
    h, w = src_bgr.shape[:2]
    return src_bgr, fake_bgr[:h, :w], blended[:h, :w]
