"""The ViT oracle (oracle/vit_torch.py) against vectors produced by the reference's own beit.py."""
import os

import numpy as np
import pytest

from marie_icr_amd.weights import make_image_u8, make_vit_state, state_checksum
from oracle.vit_torch import TorchVitOracle

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("tag", ["small", "base"])
def test_backbone_matches_reference(tag):
    g = np.load(os.path.join(GOLD, f"vit_{tag}.npz"))
    st = make_vit_state(int(g["weight_seed"]), int(g["dim"]), int(g["depth"]), int(g["heads"]))
    assert state_checksum(st) == str(g["weight_sha256"])
    th, tw = g["image_hw"]
    H32, W32 = g["canvas_hw"]
    imgs = make_image_u8(int(g["image_seed"]), int(g["batch"]), int(th), int(tw))
    o = TorchVitOracle(st, int(g["heads"]), taps=g["taps"].tolist())
    _, fpn = o.forward_features(o.preprocess(imgs, int(H32), int(W32), swap_rb=True))
    step = int(g["channel_step"])
    for j, f in enumerate(fpn):
        ref = g[f"fpn{j}"]
        assert f.shape[2:] == ref.shape[2:]
        assert np.abs(f.numpy()[:, ::step] - ref).max() <= 2e-4, (j, np.abs(f.numpy()[:, ::step] - ref).max())
