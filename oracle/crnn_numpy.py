"""ORACLE (test infrastructure, not product code) — CPU restatement in numpy of
the reference's CRNN-family recognizer forward, None-VGG-BiLSTM-CTC.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module; the product path (``marie_icr_amd``) never does.

Parity is PINNED: ``oracle/gen_golden.py`` runs the reference's own unmodified
``Model(opt)`` (imported from /root/reference/marie/models/icr) on seeded
weights/inputs and commits its outputs under ``tests/golden/crnn_*.npz``;
``tests/test_oracle_crnn.py`` checks this restatement against those vectors.

Each function cites the reference lines it follows.  Everything is fp32 (the
reference CPU path never enables autocast — SURVEY.md §8 Q3).
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import numpy as np

F32 = np.float32


# --------------------------------------------------------------------------- #
# pre-processing
# --------------------------------------------------------------------------- #
def normalize_u8(crops_u8: np.ndarray) -> np.ndarray:
    """uint8 (N,H,W) -> fp32 (N,1,H,W) in [-1,1].

    reference: marie/models/icr/dataset.py:275-283 (``ToTensor`` = /255, then
    ``sub_(0.5).div_(0.5)``), same op order so the fp32 rounding matches.
    """
    x = crops_u8.astype(F32) / F32(255.0)
    x = (x - F32(0.5)) / F32(0.5)
    return x[:, None, :, :]


# --------------------------------------------------------------------------- #
# layers
# --------------------------------------------------------------------------- #
def conv2d(x: np.ndarray, w: np.ndarray, b, pad: int) -> np.ndarray:
    """NCHW stride-1 cross-correlation (``nn.Conv2d``), fp32, via im2col + GEMM.

    reference: marie/models/icr/modules/feature_extraction.py:13-25.
    """
    n, ci, h, wd = x.shape
    co, ci2, kh, kw = w.shape
    assert ci == ci2
    if pad:
        x = np.pad(x, ((0, 0), (0, 0), (pad, pad), (pad, pad)))
    ho = x.shape[2] - kh + 1
    wo = x.shape[3] - kw + 1
    cols = np.empty((n, ci, kh, kw, ho, wo), dtype=F32)
    for dy in range(kh):
        for dx in range(kw):
            cols[:, :, dy, dx] = x[:, :, dy:dy + ho, dx:dx + wo]
    cols = cols.reshape(n, ci * kh * kw, ho * wo)
    out = np.matmul(w.reshape(co, -1)[None], cols)  # (n, co, ho*wo)
    if b is not None:
        out = out + b[None, :, None]
    return out.reshape(n, co, ho, wo).astype(F32, copy=False)


def relu(x: np.ndarray) -> np.ndarray:
    return np.maximum(x, F32(0))


def maxpool(x: np.ndarray, kh: int, kw: int) -> np.ndarray:
    """``nn.MaxPool2d((kh,kw),(kh,kw))`` — floor mode, no padding."""
    n, c, h, w = x.shape
    ho, wo = h // kh, w // kw
    x = x[:, :, :ho * kh, :wo * kw].reshape(n, c, ho, kh, wo, kw)
    return x.max(axis=(3, 5))


def batchnorm_eval(x, gamma, beta, mean, var, eps=1e-5):
    """``nn.BatchNorm2d`` in eval mode (running statistics)."""
    inv = (gamma / np.sqrt(var + F32(eps))).astype(F32)
    return (x - mean[None, :, None, None]) * inv[None, :, None, None] + beta[None, :, None, None]


def vgg_features(x: np.ndarray, st: Dict[str, np.ndarray]) -> np.ndarray:
    """``VGG_FeatureExtractor.ConvNet`` —
    reference: marie/models/icr/modules/feature_extraction.py:12-28."""
    p = "FeatureExtraction.ConvNet."
    x = maxpool(relu(conv2d(x, st[p + "0.weight"], st[p + "0.bias"], 1)), 2, 2)
    x = maxpool(relu(conv2d(x, st[p + "3.weight"], st[p + "3.bias"], 1)), 2, 2)
    x = relu(conv2d(x, st[p + "6.weight"], st[p + "6.bias"], 1))
    x = maxpool(relu(conv2d(x, st[p + "8.weight"], st[p + "8.bias"], 1)), 2, 1)
    x = conv2d(x, st[p + "11.weight"], None, 1)
    x = relu(batchnorm_eval(x, st[p + "12.weight"], st[p + "12.bias"],
                            st[p + "12.running_mean"], st[p + "12.running_var"]))
    x = conv2d(x, st[p + "14.weight"], None, 1)
    x = relu(batchnorm_eval(x, st[p + "15.weight"], st[p + "15.bias"],
                            st[p + "15.running_mean"], st[p + "15.running_var"]))
    x = maxpool(x, 2, 1)
    x = relu(conv2d(x, st[p + "18.weight"], st[p + "18.bias"], 0))
    return x.astype(F32, copy=False)


def _sigmoid(x):
    return F32(1) / (F32(1) + np.exp(-x))


def lstm_direction(x, w_ih, w_hh, b_ih, b_hh, reverse: bool) -> np.ndarray:
    """One direction of ``nn.LSTM(batch_first=True)``; gate order i, f, g, o.

    reference call site: marie/models/icr/modules/sequence_modeling.py:8,17.
    """
    n, t, _ = x.shape
    hid = w_hh.shape[1]
    xp = np.matmul(x, w_ih.T) + (b_ih + b_hh)  # (n, t, 4h)
    h = np.zeros((n, hid), dtype=F32)
    c = np.zeros((n, hid), dtype=F32)
    out = np.empty((n, t, hid), dtype=F32)
    order = range(t - 1, -1, -1) if reverse else range(t)
    for k in order:
        g = xp[:, k] + np.matmul(h, w_hh.T)
        i = _sigmoid(g[:, :hid])
        f = _sigmoid(g[:, hid:2 * hid])
        gg = np.tanh(g[:, 2 * hid:3 * hid])
        o = _sigmoid(g[:, 3 * hid:])
        c = (f * c + i * gg).astype(F32)
        h = (o * np.tanh(c)).astype(F32)
        out[:, k] = h
    return out


def bidirectional_lstm(x, st, prefix: str) -> np.ndarray:
    """``BidirectionalLSTM.forward`` = BiLSTM -> Linear(2h -> h).

    reference: marie/models/icr/modules/sequence_modeling.py:11-19.
    """
    r = prefix + "rnn."
    fw = lstm_direction(x, st[r + "weight_ih_l0"], st[r + "weight_hh_l0"],
                        st[r + "bias_ih_l0"], st[r + "bias_hh_l0"], False)
    bw = lstm_direction(x, st[r + "weight_ih_l0_reverse"], st[r + "weight_hh_l0_reverse"],
                        st[r + "bias_ih_l0_reverse"], st[r + "bias_hh_l0_reverse"], True)
    rec = np.concatenate([fw, bw], axis=2)
    return (np.matmul(rec, st[prefix + "linear.weight"].T) + st[prefix + "linear.bias"]).astype(F32)


def crnn_logits(x: np.ndarray, st: Dict[str, np.ndarray]) -> np.ndarray:
    """``Model.forward`` for Trans=None, Feat=VGG, Seq=BiLSTM, Pred=CTC.

    x: fp32 (N,1,32,W) in [-1,1] -> logits (N, T, num_class), T = W/4 - 1.
    reference: marie/models/icr/model.py:70-92.
    """
    v = vgg_features(x, st)                      # (N, 512, 1, T)
    v = v.transpose(0, 3, 1, 2).mean(axis=3)     # permute(0,3,1,2) + AdaptiveAvgPool((None,1)) + squeeze
    c = bidirectional_lstm(v.astype(F32), st, "SequenceModeling.0.")
    c = bidirectional_lstm(c, st, "SequenceModeling.1.")
    return (np.matmul(c, st["Prediction.weight"].T) + st["Prediction.bias"]).astype(F32)


# --------------------------------------------------------------------------- #
# greedy CTC decode + confidence
# --------------------------------------------------------------------------- #
def ctc_greedy(logits: np.ndarray, charset: str) -> Tuple[np.ndarray, List[str], np.ndarray]:
    """argmax -> collapse -> confidence.

    reference: ``preds.max(2)`` marie/document/craft_ocr_processor.py:240;
    ``CTCLabelConverter.decode`` marie/models/icr/utils.py:41-54 (drop blank=0
    and repeats); confidence = ``softmax(dim=2).max(dim=2).cumprod(0)[-1]``
    marie/document/craft_ocr_processor.py:255-271; text is upper-cased (:272).

    Returns (argmax indices (N,T) int32, upper-cased strings, confidences (N,) fp32).
    """
    character = ["[CTCblank]"] + list(charset)
    idx = logits.argmax(axis=2).astype(np.int32)           # first max on ties, like torch CPU
    m = logits.max(axis=2, keepdims=True)
    e = np.exp((logits - m).astype(F32)).astype(F32)
    pmax = (F32(1) / e.sum(axis=2, dtype=F32)).astype(F32)  # softmax value at the argmax
    conf = np.ones((logits.shape[0],), dtype=F32)
    for t in range(logits.shape[1]):                        # cumprod in fp32, left to right
        conf = (conf * pmax[:, t]).astype(F32)
    texts = []
    for row in idx:
        chars = []
        for i, t in enumerate(row):
            if t != 0 and not (i > 0 and row[i - 1] == t):
                chars.append(character[t])
        texts.append("".join(chars).upper())
    return idx, texts, conf


def recognize_crops_u8(crops_u8: np.ndarray, st, charset: str):
    """uint8 pre-cropped lines (N,32,W) -> (logits, argmax, strings, confidences)."""
    logits = crnn_logits(normalize_u8(crops_u8), st)
    idx, texts, conf = ctc_greedy(logits, charset)
    return logits, idx, texts, conf
