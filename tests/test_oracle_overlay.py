"""Pins oracle/overlay_torch.py to goldens written by the reference's own LocalEnhancer (networks_hd.py loaded by path,
oracle/gen_golden.py --overlay-only), and checks the restated pixel operations of blend_to_text on hand-computed values."""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("tag", ["ngf32", "ngf64"])
def test_generator_matches_reference_goldens(tag):
    import torch

    from marie_icr_amd.weights import make_image_u8, make_overlay_state, state_checksum
    from oracle.overlay_torch import TorchOverlayOracle

    g = np.load(os.path.join(GOLD, f"overlay_{tag}.npz"))
    st = make_overlay_state(int(g["weight_seed"]), int(g["ngf"]))
    assert state_checksum(st) == str(g["weight_sha256"])
    h, w = (int(v) for v in g["hw"])
    img = make_image_u8(int(g["image_seed"]), 1, h, w)[0]
    x = ((torch.from_numpy(img.astype(np.float32) / np.float32(255.0)) - 0.5) / 0.5).permute(2, 0, 1).unsqueeze(0)
    y = TorchOverlayOracle(st).generator(x)[0].permute(1, 2, 0).numpy()
    assert np.abs(y - g["out"]).max() <= 2e-5


def test_preprocess_pads_both_axes_white():
    from oracle.overlay_torch import preprocess

    img = np.zeros((40, 64, 3), np.uint8)
    out = preprocess(img)                                   # 40 is ragged -> BOTH axes grow to the next multiple (64 -> 96)
    assert out.shape == (64, 96, 3) and (out[40:] == 255).all() and (out[:40, 64:] == 255).all() and (out[:40, :64] == 0).all()
    same = np.zeros((64, 96, 3), np.uint8)
    assert preprocess(same) is same


def test_hsv_gray_and_blend_known_values():
    from oracle.overlay_torch import bgr2gray, bgr2hsv_u8, blend_to_text

    px = np.array([[[0, 0, 255], [255, 0, 0], [0, 255, 0], [255, 255, 255], [0, 0, 0], [40, 40, 250], [128, 64, 32]]], np.uint8)  # BGR
    hsv = bgr2hsv_u8(px)[0]
    assert hsv[0].tolist() == [0, 255, 255]                  # pure red
    assert hsv[1].tolist() == [120, 255, 255]                # pure blue
    assert hsv[2].tolist() == [60, 255, 255]                 # pure green
    assert hsv[3].tolist() == [0, 0, 255] and hsv[4].tolist() == [0, 0, 0]
    assert hsv[5].tolist() == [0, 214, 250]
    assert hsv[6].tolist() == [110, 191, 128]                # b=128 g=64 r=32: h = (4*96 + 32 - 64) * 30/96 = 110
    assert bgr2gray(px)[0].tolist() == [76, 29, 150, 255, 0, 103, 62]
    real = np.full((1, 3, 3), 200, np.uint8)
    mask = np.array([[[250, 30, 30], [10, 10, 10], [255, 255, 255]]], np.uint8)     # "red" in the channel order blend_to_text sees: S, V high
    out = blend_to_text(real, mask)
    assert out[0, 0].tolist() == [0, 0, 0]                   # in range -> masked out
    assert out[0, 1].tolist() == [202, 202, 202]             # 200 | 10
    assert out[0, 2].tolist() == [255, 255, 255]
