"""ORACLE (test infrastructure, not product code) — CPU fp32 restatement of the ViT encoders on the hot path with torch
functional ops: the DiT / BEiT detector backbone with its fpn1..fpn4 heads (marie/boxes/dit/ditod/beit.py:706-748) and
the TrOCR DeiT encoder (marie/models/unilm/trocr/deit.py:105-146).

PINNED (BEiT/DiT): ``oracle/gen_golden.py --vit-only`` runs the reference's unmodified ``beit.py`` (``BEiT`` class, timm
import satisfied by three pass-through helpers as in SURVEY.md Appendix B) and ``tests/test_oracle_vit.py`` checks this
restatement against those vectors.  PARITY UNPINNED (DeiT variant: timm is not installed; same block code path with
``qkv_bias=2 / layer_scale=False / final_norm=True``).
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this module.
"""
from __future__ import annotations

from typing import Dict, Sequence

import numpy as np
import torch
import torch.nn.functional as F


def _t(a):
    return torch.from_numpy(np.array(a, copy=True, order="C"))


class TorchVitOracle:
    def __init__(self, state: Dict[str, np.ndarray], heads: int, pos_hw=(14, 14), taps: Sequence[int] = (3, 5, 7, 11),
                 ln_eps: float = 1e-6):
        self.st = {k: _t(v) for k, v in state.items()}
        self.heads = heads
        self.pos_hw = pos_hw
        self.taps = list(taps)
        self.eps = ln_eps
        self.depth = 1 + max(int(k.split(".")[1]) for k in state if k.startswith("blocks."))

    @staticmethod
    def preprocess(imgs_u8: np.ndarray, H32: int, W32: int, swap_rb: bool) -> torch.Tensor:
        """(pixel - 127.5) / 127.5 per channel, CHW, zero canvas (detectron2 ImageList padding; deit Normalize(0.5, 0.5))."""
        x = _t(imgs_u8).float()
        if swap_rb:
            x = x.flip(-1)
        x = ((x - 127.5) / 127.5).permute(0, 3, 1, 2)
        out = torch.zeros(x.shape[0], 3, H32, W32)
        out[:, :, : x.shape[2], : x.shape[3]] = x
        return out

    def _block(self, x, i):
        st, p = self.st, f"blocks.{i}."
        B, N, C = x.shape
        h = F.layer_norm(x, (C,), st[p + "norm1.weight"], st[p + "norm1.bias"], self.eps)
        bias = None
        if p + "attn.q_bias" in st:
            bias = torch.cat((st[p + "attn.q_bias"], torch.zeros(C), st[p + "attn.v_bias"]))
        elif p + "attn.qkv.bias" in st:
            bias = st[p + "attn.qkv.bias"]
        qkv = F.linear(h, st[p + "attn.qkv.weight"], bias).reshape(B, N, 3, self.heads, -1).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0] * (C // self.heads) ** -0.5, qkv[1], qkv[2]
        a = (q @ k.transpose(-2, -1)).softmax(dim=-1)
        o = F.linear((a @ v).transpose(1, 2).reshape(B, N, C), st[p + "attn.proj.weight"], st[p + "attn.proj.bias"])
        x = x + (st[p + "gamma_1"] * o if p + "gamma_1" in st else o)
        h = F.layer_norm(x, (C,), st[p + "norm2.weight"], st[p + "norm2.bias"], self.eps)
        h = F.linear(F.gelu(F.linear(h, st[p + "mlp.fc1.weight"], st[p + "mlp.fc1.bias"])), st[p + "mlp.fc2.weight"],
                     st[p + "mlp.fc2.bias"])
        return x + (st[p + "gamma_2"] * h if p + "gamma_2" in st else h)

    @torch.no_grad()
    def tokens(self, x: torch.Tensor):
        """x (B,3,H,W) normalised -> (tokens (B,1+np,D) after all blocks (and the final norm if present), tap maps)."""
        st = self.st
        y = F.conv2d(x, st["patch_embed.proj.weight"], st["patch_embed.proj.bias"], stride=16)
        B, D, hp, wp = y.shape
        pos = st["pos_embed"][:, 1:, :].view(1, self.pos_hw[0], self.pos_hw[1], D).permute(0, 3, 1, 2)
        y = y + F.interpolate(pos, size=(hp, wp), mode="bicubic")
        y = y.flatten(2).transpose(1, 2)
        cls = (st["cls_token"] + st["pos_embed"][:, :1, :]).expand(B, -1, -1)
        t = torch.cat((cls, y), dim=1)
        feats = []
        for i in range(self.depth):
            t = self._block(t, i)
            if i in self.taps:
                feats.append(t[:, 1:, :].permute(0, 2, 1).reshape(B, D, hp, wp).contiguous())
        if "norm.weight" in st:
            t = F.layer_norm(t, (D,), st["norm.weight"], st["norm.bias"], self.eps)
        return t, feats

    @torch.no_grad()
    def fpn(self, feats):
        st = self.st
        f1 = F.conv_transpose2d(feats[0], st["fpn1.0.weight"], st["fpn1.0.bias"], stride=2)
        f1 = F.batch_norm(f1, st["fpn1.1.running_mean"], st["fpn1.1.running_var"], st["fpn1.1.weight"],
                          st["fpn1.1.bias"], False, 0.0, 1e-5)
        f1 = F.conv_transpose2d(F.gelu(f1), st["fpn1.3.weight"], st["fpn1.3.bias"], stride=2)
        f2 = F.conv_transpose2d(feats[1], st["fpn2.0.weight"], st["fpn2.0.bias"], stride=2)
        return [f1, f2, feats[2], F.max_pool2d(feats[3], 2, 2)]

    def forward_features(self, x: torch.Tensor):
        t, feats = self.tokens(x)
        return t, self.fpn(feats)
