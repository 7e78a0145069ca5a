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
