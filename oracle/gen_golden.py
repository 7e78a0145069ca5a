"""Generate golden vectors by running the REFERENCE's own recognizer code.

Run only in the build container (``/root/reference`` does not travel to the GPU
box):  ``python oracle/gen_golden.py``  → writes ``tests/golden/crnn_*.npz``.

What is imported from the reference, unmodified, by path:
``marie/models/icr/model.py`` (``Model``) and ``marie/models/icr/utils.py``
(``CTCLabelConverter``).  The fixtures hold DATA only: input crops, logits,
argmax indices, decoded strings, confidences, and the sha256 of the weight set
(the weights themselves are rebuilt from the seed by ``marie_icr_amd.weights``
or ``oracle.crnn_torch.default_init_state``).
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True
REF_ICR = "/root/reference/marie/models/icr"

from marie_icr_amd.weights import CRNN_CHARSET, make_crnn_input, make_crnn_state, state_checksum  # noqa: E402
from oracle.crnn_torch import default_init_state  # noqa: E402


class _Opt:
    pass


def _ref_model(img_w: int, state=None, seed=None):
    sys.path.insert(0, REF_ICR)
    from model import Model  # reference: marie/models/icr/model.py:25
    from utils import CTCLabelConverter  # reference: marie/models/icr/utils.py:7

    opt = _Opt()
    opt.Transformation, opt.FeatureExtraction = "None", "VGG"
    opt.SequenceModeling, opt.Prediction = "BiLSTM", "CTC"
    opt.imgH, opt.imgW, opt.num_fiducial = 32, img_w, 20
    opt.input_channel, opt.output_channel, opt.hidden_size = 1, 512, 256
    opt.batch_max_length = 48
    conv = CTCLabelConverter(CRNN_CHARSET)
    opt.num_class = len(conv.character)
    if seed is not None:
        torch.manual_seed(seed)
    m = Model(opt).eval()
    if state is not None:
        m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in state.items()})
    return m, conv


def _run(m, conv, crops_u8):
    # pre-processing as the reference's NormalizePAD does for a full-width crop
    # (marie/models/icr/dataset.py:275-283): ToTensor (/255) then sub 0.5 div 0.5
    x = torch.from_numpy(crops_u8).float().div(255).sub_(0.5).div_(0.5).unsqueeze(1)
    with torch.no_grad():
        preds = m(x, None)
        # decode exactly as marie/document/craft_ocr_processor.py:236-272
        preds_size = torch.IntTensor([preds.size(1)] * preds.size(0))
        _, idx = preds.max(2)
        strs = conv.decode(idx, preds_size)
        prob = F.softmax(preds, dim=2)
        pmax, _ = prob.max(dim=2)
        conf = torch.stack([p.cumprod(dim=0)[-1] for p in pmax])
    return (preds.numpy(), idx.numpy().astype(np.int32),
            np.array([s.upper() for s in strs]), conf.numpy())


def main():
    out_dir = os.path.join(ROOT, "tests", "golden")
    os.makedirs(out_dir, exist_ok=True)

    # (1) seeded "scaled" weights, 8 crops 32x256 (BASELINE config-2 shape)
    # (2) same weights, ragged width 32x100 (the production imgW) — exercises odd W
    for tag, n, w, wseed, iseed in (("scaled_w256", 8, 256, 0, 0), ("scaled_w100", 5, 100, 1, 3)):
        st = make_crnn_state(wseed)
        m, conv = _ref_model(w, state=st)
        crops = make_crnn_input(iseed, n, 32, w)
        logits, idx, strs, conf = _run(m, conv, crops)
        np.savez_compressed(
            os.path.join(out_dir, f"crnn_{tag}.npz"),
            weight_seed=wseed, input_seed=iseed, weight_sha256=state_checksum(st),
            crops_u8=crops, logits=logits, argmax=idx, strings=strs, confidence=conf)
        print(tag, logits.shape, "max|logit|", np.abs(logits).max(), strs[:2], conf[:2])

    # (3) PyTorch default init under torch.manual_seed(0) — BASELINE.md config 2.
    m, conv = _ref_model(256, seed=0)
    ref_state = {k: v.numpy() for k, v in m.state_dict().items()}
    ours = default_init_state(0)
    for k, v in ref_state.items():
        assert np.array_equal(ours[k], v), f"default-init mirror differs at {k}"
    crops = make_crnn_input(11, 4, 32, 256)
    logits, idx, strs, conf = _run(m, conv, crops)
    np.savez_compressed(
        os.path.join(out_dir, "crnn_default_w256.npz"),
        weight_seed=0, input_seed=11, weight_sha256=state_checksum(ours),
        crops_u8=crops, logits=logits, argmax=idx, strings=strs, confidence=conf)
    print("default", logits.shape, "max|logit|", np.abs(logits).max(), strs[:2], conf[:2])


if __name__ == "__main__":
    main()
