"""Deterministic CRNN weight sets and checkpoint-key handling.

The recognizer checkpoint format is the reference's ``state_dict`` of
``Model(opt)`` (reference: marie/models/icr/model.py:25-68) — the keys
``FeatureExtraction.ConvNet.<i>.*``, ``SequenceModeling.<j>.rnn.*``,
``SequenceModeling.<j>.linear.*`` and ``Prediction.*``; production checkpoints
are saved through ``torch.nn.DataParallel`` and therefore carry a ``module.``
prefix (reference: marie/document/craft_ocr_processor.py:142-146).

No trained weights ship with the reference, so parity and the
bench run on seeded synthetic weights.  ``make_crnn_state`` draws them with
numpy's PCG64 so the very same arrays can be rebuilt on the GPU box without
shipping a 34 MB file; ``state_checksum`` pins them.
"""
from __future__ import annotations

import hashlib
from typing import Dict

import numpy as np

# 94-character set of the production recognizer
# (reference: marie/document/craft_ocr_processor.py:59)
CRNN_CHARSET = (
    "0123456789abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ"
    "!\"#$%&'()*+,-./:;<=>?@[\\]^_`{|}~"
)

# (key prefix, Cout, Cin, kh, kw, has_bias) — reference:
# marie/models/icr/modules/feature_extraction.py:13-25
VGG_CONVS = (
    ("FeatureExtraction.ConvNet.0", 64, 1, 3, 3, True),
    ("FeatureExtraction.ConvNet.3", 128, 64, 3, 3, True),
    ("FeatureExtraction.ConvNet.6", 256, 128, 3, 3, True),
    ("FeatureExtraction.ConvNet.8", 256, 256, 3, 3, True),
    ("FeatureExtraction.ConvNet.11", 512, 256, 3, 3, False),
    ("FeatureExtraction.ConvNet.14", 512, 512, 3, 3, False),
    ("FeatureExtraction.ConvNet.18", 512, 512, 2, 2, True),
)
VGG_BNS = ("FeatureExtraction.ConvNet.12", "FeatureExtraction.ConvNet.15")
HIDDEN = 256


def strip_module_prefix(state: Dict[str, np.ndarray]) -> Dict[str, np.ndarray]:
    """Drop the DataParallel ``module.`` prefix if present."""
    return {(k[7:] if k.startswith("module.") else k): v for k, v in state.items()}


def make_crnn_state(seed: int = 0, num_class: int = 95, logit_gain: float = 24.0) -> Dict[str, np.ndarray]:
    """Seeded None-VGG-BiLSTM-CTC weights with O(1) activations.

    Conv weights use the He-uniform bound sqrt(6/fan_in) so the signal neither
    dies nor explodes through the ReLU stack; BatchNorm gets non-trivial
    gamma/beta/running stats so the BN fold is actually exercised; the
    prediction layer is scaled by ``logit_gain`` so that logits have trained-like
    magnitude (a few units) and the 1e-3 absolute logit tolerance is meaningful.
    """
    rng = np.random.Generator(np.random.PCG64(seed))
    st: Dict[str, np.ndarray] = {}

    def uni(shape, bound):
        return rng.uniform(-bound, bound, size=shape).astype(np.float32)

    for name, co, ci, kh, kw, has_bias in VGG_CONVS:
        fan_in = ci * kh * kw
        st[name + ".weight"] = uni((co, ci, kh, kw), np.sqrt(6.0 / fan_in))
        if has_bias:
            st[name + ".bias"] = uni((co,), 0.1)
    for name in VGG_BNS:
        st[name + ".weight"] = rng.uniform(0.6, 1.4, size=(512,)).astype(np.float32)
        st[name + ".bias"] = uni((512,), 0.2)
        st[name + ".running_mean"] = uni((512,), 0.3)
        st[name + ".running_var"] = rng.uniform(0.5, 1.5, size=(512,)).astype(np.float32)
        st[name + ".num_batches_tracked"] = np.asarray(0, dtype=np.int64)
    for j, in_size in ((0, 512), (1, 256)):
        p = f"SequenceModeling.{j}."
        for sfx in ("", "_reverse"):
            st[p + "rnn.weight_ih_l0" + sfx] = uni((4 * HIDDEN, in_size), np.sqrt(3.0 / in_size))
            st[p + "rnn.weight_hh_l0" + sfx] = uni((4 * HIDDEN, HIDDEN), np.sqrt(3.0 / HIDDEN))
            st[p + "rnn.bias_ih_l0" + sfx] = uni((4 * HIDDEN,), 0.1)
            st[p + "rnn.bias_hh_l0" + sfx] = uni((4 * HIDDEN,), 0.1)
        st[p + "linear.weight"] = uni((HIDDEN, 2 * HIDDEN), np.sqrt(6.0 / (2 * HIDDEN)))
        st[p + "linear.bias"] = uni((HIDDEN,), 0.1)
    st["Prediction.weight"] = uni((num_class, HIDDEN), logit_gain * np.sqrt(3.0 / HIDDEN))
    st["Prediction.bias"] = uni((num_class,), 0.1)
    return st


def state_checksum(state: Dict[str, np.ndarray]) -> str:
    """sha256 over (key, dtype, shape, bytes) of every entry in sorted-key order."""
    h = hashlib.sha256()
    for k in sorted(state):
        a = np.ascontiguousarray(state[k])
        h.update(k.encode())
        h.update(str(a.dtype).encode())
        h.update(str(a.shape).encode())
        h.update(a.tobytes())
    return h.hexdigest()


def make_crnn_input(seed: int, n: int, h: int = 32, w: int = 256) -> np.ndarray:
    """Seeded uint8 grayscale line crops ``(n, h, w)``: smooth strokes on a light
    background, so neighbouring pixels correlate as real scans do."""
    rng = np.random.Generator(np.random.PCG64(seed + 7919))
    base = rng.integers(0, 256, size=(n, h // 4 + 1, w // 4 + 1)).astype(np.float32)
    up = np.repeat(np.repeat(base, 4, axis=1), 4, axis=2)[:, :h, :w]
    noise = rng.integers(-24, 25, size=(n, h, w)).astype(np.float32)
    return np.clip(up + noise, 0, 255).astype(np.uint8)


# --------------------------------------------------------------------------- #
# CRAFT text detector
# --------------------------------------------------------------------------- #
# (state_dict prefix, Cout, Cin, k, BatchNorm prefix or None) in forward order.  Checkpoint keys are
# the reference's CRAFT.state_dict() (marie/models/craft/craft.py:31-58,
# marie/models/craft/basenet/vgg16_bn.py:23-49: slices index into torchvision's vgg16_bn().features).
CRAFT_CONVS = (
    ("basenet.slice1.0", 64, 3, 3, "basenet.slice1.1"),
    ("basenet.slice1.3", 64, 64, 3, "basenet.slice1.4"),
    ("basenet.slice1.7", 128, 64, 3, "basenet.slice1.8"),
    ("basenet.slice1.10", 128, 128, 3, "basenet.slice1.11"),
    ("basenet.slice2.14", 256, 128, 3, "basenet.slice2.15"),
    ("basenet.slice2.17", 256, 256, 3, "basenet.slice2.18"),
    ("basenet.slice3.20", 256, 256, 3, "basenet.slice3.21"),
    ("basenet.slice3.24", 512, 256, 3, "basenet.slice3.25"),
    ("basenet.slice3.27", 512, 512, 3, "basenet.slice3.28"),
    ("basenet.slice4.30", 512, 512, 3, "basenet.slice4.31"),
    ("basenet.slice4.34", 512, 512, 3, "basenet.slice4.35"),
    ("basenet.slice4.37", 512, 512, 3, "basenet.slice4.38"),
    ("basenet.slice5.1", 1024, 512, 3, None),     # dilation 6, padding 6
    ("basenet.slice5.2", 1024, 1024, 1, None),
    ("upconv1.conv.0", 512, 1536, 1, "upconv1.conv.1"),
    ("upconv1.conv.3", 256, 512, 3, "upconv1.conv.4"),
    ("upconv2.conv.0", 256, 768, 1, "upconv2.conv.1"),
    ("upconv2.conv.3", 128, 256, 3, "upconv2.conv.4"),
    ("upconv3.conv.0", 128, 384, 1, "upconv3.conv.1"),
    ("upconv3.conv.3", 64, 128, 3, "upconv3.conv.4"),
    ("upconv4.conv.0", 64, 192, 1, "upconv4.conv.1"),
    ("upconv4.conv.3", 32, 64, 3, "upconv4.conv.4"),
    ("conv_cls.0", 32, 32, 3, None),
    ("conv_cls.2", 32, 32, 3, None),
    ("conv_cls.4", 16, 32, 3, None),
    ("conv_cls.6", 16, 16, 1, None),
    ("conv_cls.8", 2, 16, 1, None),
)


def make_craft_state(seed: int = 0, score_gain: float = 4.0) -> Dict[str, np.ndarray]:
    """Seeded CRAFT weights with O(1) activations (He-uniform convs, non-trivial BatchNorm statistics).
    The last layer is scaled so the two score maps straddle the reference thresholds (0.3 / 0.45 / 0.7),
    which makes the post-processing do real work on random weights."""
    rng = np.random.Generator(np.random.PCG64(seed + 104729))
    st: Dict[str, np.ndarray] = {}

    def uni(shape, bound):
        return rng.uniform(-bound, bound, size=shape).astype(np.float32)

    for name, co, ci, k, bn in CRAFT_CONVS:
        fan_in = ci * k * k
        gain = score_gain if name == "conv_cls.8" else 1.0
        st[name + ".weight"] = uni((co, ci, k, k), gain * np.sqrt(6.0 / fan_in))
        st[name + ".bias"] = uni((co,), 0.1)
        if bn:
            st[bn + ".weight"] = rng.uniform(0.6, 1.4, size=(co,)).astype(np.float32)
            st[bn + ".bias"] = uni((co,), 0.2)
            st[bn + ".running_mean"] = uni((co,), 0.3)
            st[bn + ".running_var"] = rng.uniform(0.5, 1.5, size=(co,)).astype(np.float32)
            st[bn + ".num_batches_tracked"] = np.asarray(0, dtype=np.int64)
    return st


def page_line_boxes(h: int, w: int, n_lines: int = 0) -> np.ndarray:
    """Ground-truth text-line boxes (n_lines, 4) int32 x, y, w, h of ``make_page_bgr``'s layout."""
    if n_lines <= 0:
        n_lines = max(1, h // 80)
    pitch = h // (n_lines + 1)
    gh = max(6, int(pitch * 0.5))
    x0 = w // 16
    return np.array([[x0, pitch * (li + 1) - gh // 2, w - 2 * x0, gh] for li in range(n_lines)], np.int32)


def make_page_bgr(seed: int, h: int, w: int, n_lines: int = 0) -> np.ndarray:
    """Seeded synthetic page, uint8 HxWx3 (BGR like OpenCV): white paper, dark word-like blocks laid out
    on text lines, mild noise.  Deterministic across platforms (PCG64 + integer arithmetic only)."""
    rng = np.random.Generator(np.random.PCG64(seed + 15485863))
    img = np.full((h, w, 3), 255, np.uint8)
    if n_lines <= 0:
        n_lines = max(1, h // 80)
    pitch = h // (n_lines + 1)
    gh = max(6, int(pitch * 0.5))
    for li in range(n_lines):
        y0 = pitch * (li + 1) - gh // 2
        x = w // 16
        while x < w - w // 16:
            ww = int(rng.integers(gh, 4 * gh + 1))
            if x + ww >= w - w // 16:
                break
            shade = int(rng.integers(0, 90))
            # a "word": a block with a few vertical gaps so that it has stroke-like structure
            block = np.full((gh, ww, 3), shade, np.uint8)
            for gx in range(gh // 2, ww, max(3, gh // 2)):
                block[:, gx:gx + 1] = 255
            img[y0:y0 + gh, x:x + ww] = block
            x += ww + int(rng.integers(gh // 2, gh + 1))
    noise = rng.integers(-6, 7, size=img.shape).astype(np.int16)
    return np.clip(img.astype(np.int16) + noise, 0, 255).astype(np.uint8)


def make_craft_bench_state() -> Dict[str, np.ndarray]:
    """Weights for throughput runs: ``make_craft_state(0)`` with the link-score bias lowered so that, on
    ``make_page_bgr`` pages, a few percent of the score map exceeds the reference thresholds in many small islands
    — the regime a trained detector produces (hundreds of word-sized components per page) — instead of one
    page-sized blob.  Calibrated from the score-map percentiles of the seed-0 weights (text: 7 % above low_text
    0.3; link: median 1.6 before the shift)."""
    st = make_craft_state(0)
    b = st["conv_cls.8.bias"].copy()
    b[1] -= 4.2
    st["conv_cls.8.bias"] = b
    return st


# --------------------------------------------------------------------------- #
# production ICR recognizer: TPS-ResNet-BiLSTM-Attn (marie/document/craft_ocr_processor.py:49-70)
# --------------------------------------------------------------------------- #
ICR_IMG_H, ICR_IMG_W, ICR_FIDUCIAL, ICR_MAX_LEN = 32, 100, 20, 48


def icr_conv_table():
    """(conv key, BN key, Cout, Cin, k) of every conv+BN pair of TPS's localization net and of ResNet-45, in
    state_dict order (reference: marie/models/icr/modules/transformation.py:51-62,
    marie/models/icr/modules/feature_extraction.py:153-246)."""
    t = []
    loc = "Transformation.LocalizationNetwork.conv."
    for ci, bi, co, cin in ((0, 1, 64, 1), (4, 5, 128, 64), (8, 9, 256, 128), (12, 13, 512, 256)):
        t.append((f"{loc}{ci}", f"{loc}{bi}", co, cin, 3))
    r = "FeatureExtraction.ConvNet."
    t.append((r + "conv0_1", r + "bn0_1", 32, 1, 3))
    t.append((r + "conv0_2", r + "bn0_2", 64, 32, 3))
    inpl = 64
    for li, (planes, blocks) in enumerate(((128, 1), (256, 2), (512, 5), (512, 3)), start=1):
        for b in range(blocks):
            p = f"{r}layer{li}.{b}."
            t.append((p + "conv1", p + "bn1", planes, inpl if b == 0 else planes, 3))
            t.append((p + "conv2", p + "bn2", planes, planes, 3))
            if b == 0 and inpl != planes:
                t.append((p + "downsample.0", p + "downsample.1", planes, inpl, 1))
        inpl = planes
        if li < 4:
            t.append((f"{r}conv{li}", f"{r}bn{li}", planes, planes, 3))
    t.append((r + "conv4_1", r + "bn4_1", 512, 512, 2))
    t.append((r + "conv4_2", r + "bn4_2", 512, 512, 2))
    return t


def tps_buffers(F: int = ICR_FIDUCIAL, h: int = ICR_IMG_H, w: int = ICR_IMG_W):
    """``GridGenerator``'s constant buffers inv_delta_C (F+3, F+3) and P_hat (h*w, F+3) in fp32 —
    reference: marie/models/icr/modules/transformation.py:103-156 (they are part of the checkpoint)."""
    cx = np.linspace(-1.0, 1.0, int(F / 2))
    C = np.concatenate([np.stack([cx, -1 * np.ones(int(F / 2))], axis=1), np.stack([cx, np.ones(int(F / 2))], axis=1)], 0)
    hat_C = np.zeros((F, F), dtype=float)
    for i in range(F):
        for j in range(i, F):
            r = np.linalg.norm(C[i] - C[j])
            hat_C[i, j] = hat_C[j, i] = r
    np.fill_diagonal(hat_C, 1)
    hat_C = (hat_C ** 2) * np.log(hat_C)
    delta_C = np.concatenate([np.concatenate([np.ones((F, 1)), C, hat_C], axis=1),
                              np.concatenate([np.zeros((2, 3)), np.transpose(C)], axis=1),
                              np.concatenate([np.zeros((1, 3)), np.ones((1, F))], axis=1)], axis=0)
    inv_delta_C = np.linalg.inv(delta_C)
    gx = (np.arange(-w, w, 2) + 1.0) / w
    gy = (np.arange(-h, h, 2) + 1.0) / h
    P = np.stack(np.meshgrid(gx, gy), axis=2).reshape([-1, 2])
    n = P.shape[0]
    P_diff = np.tile(np.expand_dims(P, axis=1), (1, F, 1)) - np.expand_dims(C, axis=0)
    rbf_norm = np.linalg.norm(P_diff, ord=2, axis=2, keepdims=False)
    rbf = np.multiply(np.square(rbf_norm), np.log(rbf_norm + 1e-6))
    P_hat = np.concatenate([np.ones((n, 1)), P, rbf], axis=1)
    return inv_delta_C.astype(np.float32), P_hat.astype(np.float32)


def make_icr_state(seed: int = 0, num_class: int = 96, logit_gain: float = 16.0) -> Dict[str, np.ndarray]:
    """Seeded TPS-ResNet-BiLSTM-Attn weights with O(1) activations and non-trivial BatchNorm statistics.  The TPS
    output layer gets small random weights on top of the reference's identity-grid bias so the rectification is a
    mild, input-dependent warp (exercises grid_sample away from the identity)."""
    rng = np.random.Generator(np.random.PCG64(seed + 2750159))
    st: Dict[str, np.ndarray] = {}

    def uni(shape, bound):
        return rng.uniform(-bound, bound, size=shape).astype(np.float32)

    def bn(prefix, c, gamma=(0.6, 1.4)):
        st[prefix + ".weight"] = rng.uniform(*gamma, size=(c,)).astype(np.float32)
        st[prefix + ".bias"] = uni((c,), 0.2)
        st[prefix + ".running_mean"] = uni((c,), 0.3)
        st[prefix + ".running_var"] = rng.uniform(0.5, 1.5, size=(c,)).astype(np.float32)
        st[prefix + ".num_batches_tracked"] = np.asarray(0, dtype=np.int64)

    for ck, bk, co, ci, k in icr_conv_table():
        st[ck + ".weight"] = uni((co, ci, k, k), np.sqrt(6.0 / (ci * k * k)))
        # residual branches: damp the second conv of a block so the sum of branches stays O(1)
        bn(bk, co, gamma=(0.3, 0.7) if bk.endswith("bn2") else (0.6, 1.4))
    loc = "Transformation.LocalizationNetwork."
    st[loc + "localization_fc1.0.weight"] = uni((256, 512), np.sqrt(6.0 / 512))
    st[loc + "localization_fc1.0.bias"] = uni((256,), 0.1)
    st[loc + "localization_fc2.weight"] = uni((2 * ICR_FIDUCIAL, 256), 0.02)
    cx = np.linspace(-1.0, 1.0, ICR_FIDUCIAL // 2)
    top = np.stack([cx, np.linspace(0.0, -1.0, num=ICR_FIDUCIAL // 2)], axis=1)
    bot = np.stack([cx, np.linspace(1.0, 0.0, num=ICR_FIDUCIAL // 2)], axis=1)
    st[loc + "localization_fc2.bias"] = np.concatenate([top, bot], axis=0).astype(np.float32).reshape(-1)
    inv_delta_C, P_hat = tps_buffers()
    st["Transformation.GridGenerator.inv_delta_C"] = inv_delta_C
    st["Transformation.GridGenerator.P_hat"] = P_hat
    for j, in_size in ((0, 512), (1, 256)):
        p = f"SequenceModeling.{j}."
        for sfx in ("", "_reverse"):
            st[p + "rnn.weight_ih_l0" + sfx] = uni((4 * HIDDEN, in_size), np.sqrt(3.0 / in_size))
            st[p + "rnn.weight_hh_l0" + sfx] = uni((4 * HIDDEN, HIDDEN), np.sqrt(3.0 / HIDDEN))
            st[p + "rnn.bias_ih_l0" + sfx] = uni((4 * HIDDEN,), 0.1)
            st[p + "rnn.bias_hh_l0" + sfx] = uni((4 * HIDDEN,), 0.1)
        st[p + "linear.weight"] = uni((HIDDEN, 2 * HIDDEN), np.sqrt(6.0 / (2 * HIDDEN)))
        st[p + "linear.bias"] = uni((HIDDEN,), 0.1)
    a = "Prediction.attention_cell."
    st[a + "i2h.weight"] = uni((HIDDEN, HIDDEN), np.sqrt(3.0 / HIDDEN))
    st[a + "h2h.weight"] = uni((HIDDEN, HIDDEN), np.sqrt(3.0 / HIDDEN))
    st[a + "h2h.bias"] = uni((HIDDEN,), 0.1)
    st[a + "score.weight"] = uni((1, HIDDEN), 4.0 * np.sqrt(3.0 / HIDDEN))
    st[a + "rnn.weight_ih"] = uni((4 * HIDDEN, HIDDEN + num_class), np.sqrt(3.0 / HIDDEN))
    st[a + "rnn.weight_hh"] = uni((4 * HIDDEN, HIDDEN), np.sqrt(3.0 / HIDDEN))
    st[a + "rnn.bias_ih"] = uni((4 * HIDDEN,), 0.1)
    st[a + "rnn.bias_hh"] = uni((4 * HIDDEN,), 0.1)
    st["Prediction.generator.weight"] = uni((num_class, HIDDEN), logit_gain * np.sqrt(3.0 / HIDDEN))
    st["Prediction.generator.bias"] = uni((num_class,), 0.1)
    return st


# ------------------------------------------------------------------------------------------------ ViT / DiT backbone
def make_vit_state(seed: int = 0, dim: int = 768, depth: int = 12, heads: int = 12, pos_hw=(14, 14), layer_scale=True,
                   qkv_bias: int = 1, fpn: bool = True, final_norm: bool = False) -> Dict[str, np.ndarray]:
    """Seeded BEiT / DiT (or DeiT when ``qkv_bias != 1``) weights under the reference module's own key names
    (marie/boxes/dit/ditod/beit.py:564-640).  Gains are chosen so attention is peaked and every branch contributes at
    O(1) — a much harder numerical case than the 0.02-std initialisation of an untrained checkpoint."""
    rng = np.random.Generator(np.random.PCG64(seed + 4256233))
    st: Dict[str, np.ndarray] = {}

    def uni(shape, bound):
        return rng.uniform(-bound, bound, size=shape).astype(np.float32)

    D = dim
    st["cls_token"] = uni((1, 1, D), 0.5)
    st["pos_embed"] = uni((1, 1 + pos_hw[0] * pos_hw[1], D), 0.5)
    st["patch_embed.proj.weight"] = uni((D, 3, 16, 16), np.sqrt(3.0 / 768) * 2.0)
    st["patch_embed.proj.bias"] = uni((D,), 0.1)
    for i in range(depth):
        p = f"blocks.{i}."
        st[p + "norm1.weight"] = rng.uniform(0.7, 1.3, size=(D,)).astype(np.float32)
        st[p + "norm1.bias"] = uni((D,), 0.1)
        st[p + "attn.qkv.weight"] = uni((3 * D, D), 2.0 * np.sqrt(3.0 / D))
        if qkv_bias == 1:
            st[p + "attn.q_bias"] = uni((D,), 0.2)
            st[p + "attn.v_bias"] = uni((D,), 0.2)
        elif qkv_bias == 2:
            st[p + "attn.qkv.bias"] = uni((3 * D,), 0.2)
        st[p + "attn.proj.weight"] = uni((D, D), np.sqrt(3.0 / D))
        st[p + "attn.proj.bias"] = uni((D,), 0.1)
        st[p + "norm2.weight"] = rng.uniform(0.7, 1.3, size=(D,)).astype(np.float32)
        st[p + "norm2.bias"] = uni((D,), 0.1)
        st[p + "mlp.fc1.weight"] = uni((4 * D, D), np.sqrt(3.0 / D))
        st[p + "mlp.fc1.bias"] = uni((4 * D,), 0.1)
        st[p + "mlp.fc2.weight"] = uni((D, 4 * D), np.sqrt(3.0 / (4 * D)))
        st[p + "mlp.fc2.bias"] = uni((D,), 0.1)
        if layer_scale:
            st[p + "gamma_1"] = rng.uniform(0.2, 0.6, size=(D,)).astype(np.float32)
            st[p + "gamma_2"] = rng.uniform(0.2, 0.6, size=(D,)).astype(np.float32)
    if final_norm:
        st["norm.weight"] = rng.uniform(0.7, 1.3, size=(D,)).astype(np.float32)
        st["norm.bias"] = uni((D,), 0.1)
    if fpn:
        for name in ("fpn1.0", "fpn1.3", "fpn2.0"):
            st[name + ".weight"] = uni((D, D, 2, 2), np.sqrt(3.0 / D))
            st[name + ".bias"] = uni((D,), 0.1)
        st["fpn1.1.weight"] = rng.uniform(0.6, 1.4, size=(D,)).astype(np.float32)
        st["fpn1.1.bias"] = uni((D,), 0.2)
        st["fpn1.1.running_mean"] = uni((D,), 0.3)
        st["fpn1.1.running_var"] = rng.uniform(0.5, 1.5, size=(D,)).astype(np.float32)
    return st


def make_image_u8(seed: int, n: int, h: int, w: int) -> np.ndarray:
    """(n, h, w, 3) uint8: smooth blobs + noise, deterministic."""
    rng = np.random.Generator(np.random.PCG64(seed + 7919))
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    out = np.empty((n, h, w, 3), np.uint8)
    for i in range(n):
        img = rng.uniform(0, 255, size=(h, w, 3)).astype(np.float32) * 0.35
        for _ in range(6):
            cy, cx, s = rng.uniform(0, h), rng.uniform(0, w), rng.uniform(4, 0.3 * max(h, w))
            amp = rng.uniform(-160, 160, size=3).astype(np.float32)
            img += np.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / (2 * s * s))[..., None] * amp
        out[i] = np.clip(img + 80, 0, 255).astype(np.uint8)
    return out


DIT_VIT_PREFIX = "backbone.bottom_up.backbone."


def make_dit_state(seed: int = 0, model: str = "base") -> Dict[str, np.ndarray]:
    """Seeded weights for the whole DiT Mask R-CNN text detector under detectron2 checkpoint key names
    (backbone.bottom_up.backbone.* = the BEiT module, backbone.fpn_*, proposal_generator.rpn_head.*, roi_heads.*).
    Gains keep objectness / class scores spread out so top-k, NMS and the score threshold all cut somewhere non-trivial."""
    dim, depth, heads = (768, 12, 12) if model == "base" else (1024, 24, 16)
    st = {DIT_VIT_PREFIX + k: v for k, v in make_vit_state(seed, dim, depth, heads).items()}
    rng = np.random.Generator(np.random.PCG64(seed + 60013))

    def uni(shape, bound):
        return rng.uniform(-bound, bound, size=shape).astype(np.float32)

    for lvl in (2, 3, 4, 5):
        st[f"backbone.fpn_lateral{lvl}.weight"] = uni((256, dim, 1, 1), np.sqrt(3.0 / dim) * 0.5)
        st[f"backbone.fpn_lateral{lvl}.bias"] = uni((256,), 0.1)
        st[f"backbone.fpn_output{lvl}.weight"] = uni((256, 256, 3, 3), np.sqrt(3.0 / (256 * 9)))
        st[f"backbone.fpn_output{lvl}.bias"] = uni((256,), 0.1)
    r = "proposal_generator.rpn_head."
    st[r + "conv.weight"] = uni((256, 256, 3, 3), np.sqrt(6.0 / (256 * 9)))
    st[r + "conv.bias"] = uni((256,), 0.1)
    st[r + "objectness_logits.weight"] = uni((3, 256, 1, 1), 4.0 * np.sqrt(3.0 / 256))
    st[r + "objectness_logits.bias"] = uni((3,), 0.5)
    st[r + "anchor_deltas.weight"] = uni((12, 256, 1, 1), 0.6 * np.sqrt(3.0 / 256))
    st[r + "anchor_deltas.bias"] = uni((12,), 0.2)
    h = "roi_heads.box_head."
    st[h + "fc1.weight"] = uni((1024, 256 * 49), np.sqrt(6.0 / (256 * 49)))
    st[h + "fc1.bias"] = uni((1024,), 0.1)
    st[h + "fc2.weight"] = uni((1024, 1024), np.sqrt(6.0 / 1024))
    st[h + "fc2.bias"] = uni((1024,), 0.1)
    p = "roi_heads.box_predictor."
    st[p + "cls_score.weight"] = uni((2, 1024), 3.0 * np.sqrt(3.0 / 1024))
    st[p + "cls_score.bias"] = uni((2,), 0.2)
    st[p + "bbox_pred.weight"] = uni((4, 1024), 2.0 * np.sqrt(3.0 / 1024))
    st[p + "bbox_pred.bias"] = uni((4,), 0.2)
    return st


# ------------------------------------------------------------------------------------------------ TrOCR
def make_trocr_state(seed: int = 0, enc=(768, 12, 12), dec=(1024, 12, 16, 4096), vocab: int = 50265, max_positions: int = 512,
                     pad: int = 1, img: int = 384, logit_gain: float = 6.0, eos: int = 2, eos_gain: float = 18.0
                     ) -> Dict[str, np.ndarray]:
    """Seeded TrOCR weights under fairseq checkpoint key names: ``encoder.deit.*`` (timm VisionTransformer, no qkv bias,
    final norm) and ``decoder.*`` (TransformerDecoder with RoBERTa arguments: learned positions, layernorm_embedding,
    post-LN layers, output projection tied to embed_tokens).  ``logit_gain`` spreads the token distribution so beam
    search has clear winners; ``eos_gain`` adds a constant to the ``</s>`` logit (through the last LayerNorm's bias
    direction) so hypotheses end at different lengths and crops of one batch finish at different steps."""
    ed, edepth, eheads = enc
    D, L, H, F = dec
    g = img // 16
    st = {"encoder.deit." + k: v for k, v in
          make_vit_state(seed, ed, edepth, eheads, pos_hw=(g, g), layer_scale=False, qkv_bias=0, fpn=False, final_norm=True).items()}
    rng = np.random.Generator(np.random.PCG64(seed + 1299709))

    def uni(shape, bound):
        return rng.uniform(-bound, bound, size=shape).astype(np.float32)

    st["decoder.embed_tokens.weight"] = uni((vocab, D), logit_gain * np.sqrt(3.0 / D))
    st["decoder.embed_positions.weight"] = uni((max_positions + pad + 1, D), 0.5)
    st["decoder.layernorm_embedding.weight"] = rng.uniform(0.7, 1.3, size=(D,)).astype(np.float32)
    st["decoder.layernorm_embedding.bias"] = uni((D,), 0.1)
    for l in range(L):
        p = f"decoder.layers.{l}."
        for att, kd in (("self_attn", D), ("encoder_attn", ed)):
            st[p + att + ".q_proj.weight"] = uni((D, D), 2.0 * np.sqrt(3.0 / D))
            st[p + att + ".k_proj.weight"] = uni((D, kd), 2.0 * np.sqrt(3.0 / kd))
            st[p + att + ".v_proj.weight"] = uni((D, kd), np.sqrt(3.0 / kd))
            st[p + att + ".out_proj.weight"] = uni((D, D), np.sqrt(3.0 / D))
            for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
                st[p + att + f".{n}.bias"] = uni((D,), 0.1)
        st[p + "fc1.weight"] = uni((F, D), np.sqrt(3.0 / D))
        st[p + "fc1.bias"] = uni((F,), 0.1)
        st[p + "fc2.weight"] = uni((D, F), np.sqrt(3.0 / F))
        st[p + "fc2.bias"] = uni((D,), 0.1)
        for n in ("self_attn_layer_norm", "encoder_attn_layer_norm", "final_layer_norm"):
            st[p + n + ".weight"] = rng.uniform(0.7, 1.3, size=(D,)).astype(np.float32)
            st[p + n + ".bias"] = uni((D,), 0.1)
    if eos_gain:
        b = uni((D,), 0.6)
        st[f"decoder.layers.{L - 1}.final_layer_norm.bias"] = b
        st["decoder.embed_tokens.weight"][eos] += b * np.float32(eos_gain / float((b * b).sum()))
    return st


def make_trocr_sharp_state(seed: int = 0, enc=(768, 12, 12), dec=(1024, 12, 16, 4096), vocab: int = 50265,
                           max_positions: int = 512, pad: int = 1, img: int = 384, eos: int = 2, top_logit: float = 16.0,
                           tok_gain: float = 3.0, pos_gain: float = 1.5, img_gain: float = 0.6, sub_gain: float = 0.05,
                           end_fraction: float = 0.08, fam_gain: float = 0.5, end_gain: float = 2.0) -> Dict[str, np.ndarray]:
    """Seeded TrOCR weights whose beam search has MARGINS, for parity statements that need them (the reduced-precision run
    must return the fp32 run's tokens whenever the fp32 top-1 / top-2 gaps exceed the measured error many times over).
    Same architecture and key names as ``make_trocr_state``; what differs is structure in the decoder's two ends:

    * the model dimension is split into a token part (first 3/4) and a family part (last 1/4).  A token's embedding is a
      random direction in the token part plus a small random vector in the family part; positions live mostly in the family part;
    * every token has one successor among the even tokens (family A) and one among the odd tokens (family B).  The output
      projection (untied: ``decoder.output_projection.weight``) of token k holds, in the token part, the embedding directions of
      the tokens k succeeds (so the two successors of the previous token score ``top_logit`` and everything else a few units),
      and in the family part +u / -u by k's family: which successor wins is decided by u . x_family, a sum of a term of the
      previous token (``tok_gain``), of the position (``pos_gain``), of the image through the encoder-attention
      (``img_gain`` on the family rows of its output projection) and of the small contributions of every other sub-layer
      (``sub_gain`` on their output projections: they stay alive, the residual stream stays close to the embedding);
    * ``end_fraction`` of the tokens raise a flag dimension that the ``</s>`` row reads (``end_gain`` x ``top_logit`` for a
      flag that survived the LayerNorms undiluted): lines end at different lengths.

    Decisions are binary with a typical margin of several nats, depend on the image, the position and the history, and the
    runner-up stays in the beam — sequences differ from line to line and within a line."""
    ed, edepth, eheads = enc
    D, L, H, F = dec
    g = img // 16
    st = {"encoder.deit." + k: v for k, v in
          make_vit_state(seed, ed, edepth, eheads, pos_hw=(g, g), layer_scale=False, qkv_bias=0, fpn=False, final_norm=True).items()}
    rng = np.random.Generator(np.random.PCG64(seed + 7368787))
    DT = D * 3 // 4
    DF = D - DT - 1                    # family dimensions; the last dimension is the end flag
    FLAG = D - 1

    def uni(shape, bound):
        return rng.uniform(-bound, bound, size=shape).astype(np.float32)

    def gauss(shape, std):
        return (rng.standard_normal(size=shape) * std).astype(np.float32)

    emb = np.zeros((vocab, D), np.float32)
    emb[:, :DT] = gauss((vocab, DT), 1.0)
    emb[:, DT:DT + DF] = gauss((vocab, DF), 0.5)
    ending = rng.random(vocab) < end_fraction
    ending[:4] = False
    emb[ending, FLAG] = 4.0
    st["decoder.embed_tokens.weight"] = emb
    pos = np.zeros((max_positions + pad + 1, D), np.float32)
    pos[:, :DT] = gauss((len(pos), DT), 0.1)
    pos[:, DT:DT + DF] = gauss((len(pos), DF), 0.5)
    st["decoder.embed_positions.weight"] = pos
    st["decoder.layernorm_embedding.weight"] = np.ones((D,), np.float32)
    st["decoder.layernorm_embedding.bias"] = np.zeros((D,), np.float32)
    for l in range(L):
        p = f"decoder.layers.{l}."
        for att, kd in (("self_attn", D), ("encoder_attn", ed)):
            st[p + att + ".q_proj.weight"] = uni((D, D), 2.0 * np.sqrt(3.0 / D))
            st[p + att + ".k_proj.weight"] = uni((D, kd), 2.0 * np.sqrt(3.0 / kd))
            st[p + att + ".v_proj.weight"] = uni((D, kd), np.sqrt(3.0 / kd))
            w = uni((D, D), sub_gain * np.sqrt(3.0 / D))
            if att == "encoder_attn":
                w[DT:DT + DF] *= np.float32(img_gain / sub_gain)
            st[p + att + ".out_proj.weight"] = w
            for n in ("q_proj", "k_proj", "v_proj"):
                st[p + att + f".{n}.bias"] = uni((D,), 0.1)
            st[p + att + ".out_proj.bias"] = np.zeros((D,), np.float32)
        st[p + "fc1.weight"] = uni((F, D), np.sqrt(3.0 / D))
        st[p + "fc1.bias"] = uni((F,), 0.1)
        st[p + "fc2.weight"] = uni((D, F), sub_gain * np.sqrt(3.0 / F))
        st[p + "fc2.bias"] = np.zeros((D,), np.float32)
        for n in ("self_attn_layer_norm", "encoder_attn_layer_norm", "final_layer_norm"):
            st[p + n + ".weight"] = np.ones((D,), np.float32)
            st[p + n + ".bias"] = np.zeros((D,), np.float32)
    # output projection: successors in the token part, family sign in the family part, end flag for </s>
    ids = np.arange(vocab)
    n_even = (vocab - 4 + 1) // 2
    succ_a = 4 + 2 * rng.integers(0, n_even, size=vocab)                       # even tokens >= 4
    succ_b = 5 + 2 * rng.integers(0, (vocab - 5 + 1) // 2, size=vocab)         # odd tokens >= 5
    unit = emb[:, :DT] / np.linalg.norm(emb[:, :DT], axis=1, keepdims=True)
    out = np.zeros((vocab, D), np.float32)
    a = np.float32(top_logit / np.sqrt(DT))        # a token's own direction has length ~ sqrt(DT) after the last LayerNorm
    np.add.at(out[:, :DT], succ_a, a * unit)
    np.add.at(out[:, :DT], succ_b, a * unit)
    u = gauss((DF,), 1.0)
    u /= np.linalg.norm(u)
    fam = np.where(ids % 2 == 0, 1.0, -1.0).astype(np.float32)
    out[:, DT:DT + DF] = fam[:, None] * u[None, :] * np.float32(fam_gain)
    out[:4] = 0.0
    out[eos, FLAG] = np.float32(end_gain * top_logit / 4.0)
    st["decoder.output_projection.weight"] = out
    # the family coordinate the decision reads: project the three sources onto u with the requested gains
    emb[:, DT:DT + DF] += gauss((vocab, 1), tok_gain) * u[None, :]
    pos[:, DT:DT + DF] += gauss((len(pos), 1), pos_gain) * u[None, :]
    return st


def make_ocr_result(seed: int, width: int, height: int, n_lines: int = 12, page: int = 0) -> dict:
    """A seeded page result in the engine's output layout (``{"meta", "words", "lines"}``, boxes xywh) — synthetic input
    for the step after the path (renderers, ``get_words_and_boxes``): ragged lines, gaps, words of 1..12 characters."""
    rng = np.random.Generator(np.random.PCG64(seed + 7919))
    alphabet = "abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ0123456789-.,#$%/"
    words, lines = [], []
    wid = 0
    y = int(rng.integers(10, 60))
    for ln in range(1, n_lines + 1):
        lh = int(rng.integers(18, 40))
        if y + lh >= height - 4:
            break
        x = int(rng.integers(0, max(1, width // 3)))
        ids, x0 = [], None
        k = 0
        while True:
            n_ch = int(rng.integers(1, 13))
            ww = int(n_ch * rng.integers(7, 12))
            if x + ww >= width - 2 or k >= 14:
                break
            text = "".join(alphabet[int(i)] for i in rng.integers(0, len(alphabet), n_ch))
            words.append({"id": wid, "text": text, "confidence": round(float(rng.uniform(0.3, 1.0)), 4),
                          "box": [x, y + int(rng.integers(0, 4)), ww, lh - int(rng.integers(0, 4))], "line": ln,
                          "word_index": k})
            ids.append(wid)
            x0 = x if x0 is None else x0
            wid += 1
            k += 1
            x += ww + int(rng.integers(4, 90))
        if ids:
            last = words[ids[-1]]["box"]
            lines.append({"line": ln, "wordids": ids, "text": " ".join(words[i]["text"] for i in ids),
                          "bbox": [x0, y, last[0] + last[2] - x0, lh], "confidence": 1.0})
        y += lh + int(rng.integers(2, 70))
    order = rng.permutation(len(words))                      # the engine does not promise id order in "words"
    return {"meta": {"imageSize": {"width": int(width), "height": int(height)}, "page": int(page), "lang": "en",
                     "format": "xywh"},
            "words": [words[int(i)] for i in order], "lines": lines}


def make_voting_case(seed: int, regions: bool):
    """Seeded recognizer outputs in the engine's layout: 2-4 recognizers, identical word ids / boxes, texts from a 3-word
    alphabet (so majorities, equal-size groups and equal confidence sums all occur), confidences on a 0.05 grid."""
    rng = np.random.Generator(np.random.PCG64(7000 + seed))
    n_proc = int(rng.integers(2, 5))
    names = ["default", "craft", "levocr", "tesseract"][:n_proc]
    n_units = int(rng.integers(1, 4))
    texts = ["ACME", "ACNE", "AGME"]
    agg = {}
    for name in names:
        units = []
        for u in range(n_units):
            r2 = np.random.Generator(np.random.PCG64(9000 + seed * 31 + u))          # same word layout for every recognizer
            nw = int(r2.integers(0, 7))
            words = []
            for i in r2.permutation(nw).tolist() if regions else range(nw):
                words.append({"id": int(i), "text": texts[int(rng.integers(0, 3))],
                              "confidence": round(float(rng.integers(1, 20)) * 0.05, 4),
                              "box": [10 * int(i), 5 * u, 8, 4], "line": 1 + int(i) // 3, "word_index": int(i)})
            unit = {"meta": {"page": u}, "words": words, "lines": []}
            if regions:
                unit["id"] = str(100 + u)
            units.append(unit)
        if regions:
            agg[name] = {"regions": [{"id": 100 + u, "text": " ".join(w["text"] for w in un["words"]), "confidence": 0.5}
                                     for u, un in enumerate(units)], "extended": units}
        else:
            agg[name] = units
    reg = [{"id": 100 + u, "pageIndex": 0, "x": 0, "y": 0, "w": 50, "h": 20} for u in range(n_units)] if regions else None
    return names, agg, reg


# ------------------------------------------------------------------------------------------------ overlay (pix2pixHD LocalEnhancer)
def overlay_conv_table(ngf: int = 64):
    """(state_dict prefix, kind, out channels, in channels, kernel) of every convolution of the reference's ``LocalEnhancer``
    generator (marie/models/pix2pix/models/networks_hd.py:24-106 with n_downsample_global 3, n_blocks_global 9,
    n_local_enhancers 1, n_blocks_local 3 — networks.py:189-196), in forward order.  kind: "sn" = spectral-normed Conv2d,
    "snT" = spectral-normed ConvTranspose2d (weight laid out [in][out][k][k]), "plain" = Conv2d without spectral norm."""
    G = 2 * ngf
    t = [("downsample", "plain", 3, 3, 3), ("model.1", "sn", G, 3, 7), ("model.4", "sn", 2 * G, G, 3),
         ("model.7", "sn", 4 * G, 2 * G, 3), ("model.10", "sn", 8 * G, 4 * G, 3)]
    for b in range(9):
        t += [(f"model.{13 + b}.conv_block.1", "sn", 8 * G, 8 * G, 3), (f"model.{13 + b}.conv_block.5", "sn", 8 * G, 8 * G, 3)]
    t += [("model.23", "sn", 4 * G, 8 * G, 3), ("model.26", "sn", 2 * G, 4 * G, 3), ("model.29", "sn", G, 2 * G, 3),
          ("model1_1.1", "sn", ngf, 3, 7), ("model1_1.4", "sn", 2 * ngf, ngf, 3)]
    for b in range(3):
        t += [(f"model1_2.{b}.conv_block.1", "sn", 2 * ngf, 2 * ngf, 3), (f"model1_2.{b}.conv_block.5", "sn", 2 * ngf, 2 * ngf, 3)]
    t += [("model1_2.3", "snT", ngf, 2 * ngf, 3), ("model1_2.7", "sn", 3, ngf, 7)]
    return t


def make_overlay_state(seed: int = 0, ngf: int = 64) -> Dict[str, np.ndarray]:
    """Seeded weights for the overlay generator under the reference's state_dict keys: ``weight_orig`` / ``weight_u`` /
    ``weight_v`` / ``bias`` per spectral-normed layer.  ``u`` and ``v`` come from ten power iterations, as in a trained
    checkpoint (at inference the layer divides ``weight_orig`` by u^T W v, torch.nn.utils.spectral_norm in eval mode)."""
    rng = np.random.Generator(np.random.PCG64(seed + 7919))
    st = {}
    for name, kind, co, ci, k in overlay_conv_table(ngf):
        shape = (ci, co, k, k) if kind == "snT" else (co, ci, k, k)
        w = rng.uniform(-1.0, 1.0, size=shape).astype(np.float32) * np.float32(np.sqrt(3.0 / (ci * k * k)))
        b = rng.uniform(-0.2, 0.2, size=(co,)).astype(np.float32)
        if kind == "plain":
            st[name + ".weight"], st[name + ".bias"] = w, b
            continue
        mat = (w.transpose(1, 0, 2, 3) if kind == "snT" else w).reshape(co, -1).astype(np.float64)   # dim = 1 for ConvTranspose
        u = rng.normal(size=(co,))
        u /= np.linalg.norm(u)
        for _ in range(10):
            v = mat.T @ u
            v /= np.linalg.norm(v) + 1e-12
            u = mat @ v
            u /= np.linalg.norm(u) + 1e-12
        st[name + ".weight_orig"], st[name + ".bias"] = w, b
        st[name + ".weight_u"], st[name + ".weight_v"] = u.astype(np.float32), v.astype(np.float32)
    return st
