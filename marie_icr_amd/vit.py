"""ViT encoder handle (DiT / BEiT detector backbone, TrOCR DeiT encoder) over ``mhip_vit`` in libmarie_hip.so.

reference: marie/boxes/dit/ditod/beit.py:564-748 (BEiT, dit_base_patch16 :787, dit_large_patch16 :803),
marie/models/unilm/trocr/deit.py:59-146.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional, Sequence

import numpy as np

from ._lib import PREC_F16, Context, VitConfig, check


def dit_config(name: str = "base") -> VitConfig:
    """``dit_base_patch16`` / ``dit_large_patch16`` with the taps of VIT_Backbone (backbone.py:37-40)."""
    cfg = VitConfig()
    if name == "base":
        cfg.dim, cfg.depth, cfg.heads, taps = 768, 12, 12, (3, 5, 7, 11)
    elif name == "large":
        cfg.dim, cfg.depth, cfg.heads, taps = 1024, 24, 16, (7, 11, 15, 23)
    else:
        raise ValueError(name)
    cfg.patch, cfg.pos_h, cfg.pos_w = 16, 14, 14
    cfg.layer_scale, cfg.qkv_bias, cfg.final_norm, cfg.fpn = 1, 1, 0, 1
    cfg.taps = (C.c_int * 4)(*taps)
    cfg.ln_eps = 1e-6
    return cfg


def make_config(dim, depth, heads, taps=(0, 0, 0, 0), pos_hw=(14, 14), layer_scale=1, qkv_bias=1, final_norm=0, fpn=1,
                ln_eps=1e-6) -> VitConfig:
    cfg = VitConfig()
    cfg.dim, cfg.depth, cfg.heads, cfg.patch = dim, depth, heads, 16
    cfg.pos_h, cfg.pos_w = pos_hw
    cfg.layer_scale, cfg.qkv_bias, cfg.final_norm, cfg.fpn = layer_scale, qkv_bias, final_norm, fpn
    cfg.taps = (C.c_int * 4)(*taps)
    cfg.ln_eps = ln_eps
    return cfg


def load_tensors(ctx: Context, setter, handle, state: Dict[str, np.ndarray], what: str):
    for key, val in state.items():
        arr = np.ascontiguousarray(np.asarray(val), dtype=np.float32)
        shape = (C.c_int64 * max(arr.ndim, 1))(*arr.shape)
        check(ctx.h, setter(handle, key.encode(), arr.ctypes.data_as(C.c_void_p), shape, arr.ndim), f"{what}({key})")


class VitModel:
    def __init__(self, ctx: Context, cfg: VitConfig, state: Optional[Dict[str, np.ndarray]] = None,
                 precision: int = PREC_F16):
        self.ctx, self.lib, self.cfg, self.precision = ctx, ctx.lib, cfg, int(precision)
        h = C.c_void_p()
        check(ctx.h, self.lib.mhip_vit_create(ctx.h, self.precision, C.byref(cfg), C.byref(h)), "mhip_vit_create")
        self.h = h
        ctx.adopt(self)
        if state is not None:
            load_tensors(ctx, self.lib.mhip_vit_set_tensor, self.h, state, "mhip_vit_set_tensor")
            check(ctx.h, self.lib.mhip_vit_finalize(self.h), "mhip_vit_finalize")

    def forward_host(self, imgs_u8: np.ndarray, canvas_hw: Sequence[int], swap_rb: bool = True, want_tokens=False,
                     want_fpn=True):
        """imgs (B, th, tw, 3) uint8 -> dict(tokens (B, 1+np, D) | None, fpn list of NHWC fp32 maps | None)."""
        imgs = np.ascontiguousarray(imgs_u8, np.uint8)
        B, th, tw, _ = imgs.shape
        H32, W32 = int(canvas_hw[0]), int(canvas_hw[1])
        hp, wp, D = H32 // 16, W32 // 16, self.cfg.dim
        tokens = np.empty((B, 1 + hp * wp, D), np.float32) if want_tokens else None
        fpn = None
        if want_fpn and self.cfg.fpn:
            fpn = [np.empty((B, 4 * hp, 4 * wp, D), np.float32), np.empty((B, 2 * hp, 2 * wp, D), np.float32),
                   np.empty((B, hp, wp, D), np.float32), np.empty((B, hp // 2, wp // 2, D), np.float32)]
        vp = lambda a: a.ctypes.data_as(C.c_void_p) if a is not None else C.c_void_p(0)  # noqa: E731
        f = fpn or [None] * 4
        check(self.ctx.h, self.lib.mhip_vit_forward_host(self.h, vp(imgs), B, th, tw, H32, W32, int(swap_rb), vp(tokens),
                                                         vp(f[0]), vp(f[1]), vp(f[2]), vp(f[3])), "mhip_vit_forward_host")
        return {"tokens": tokens, "fpn": fpn}

    def close(self):
        if getattr(self, "h", None) is not None and self.h.value:
            self.lib.mhip_vit_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
