"""ORACLE (test infrastructure, not product code) — the same CRNN restatement as
``oracle/crnn_numpy.py`` written with ``torch.nn.functional`` on CPU fp32.

It exists for two reasons: (1) it is what the reference itself executes on a
CPU host (``nn.Conv2d``/``nn.LSTM``/``nn.Linear`` → oneDNN/MKL), so it is the
honest ``cpu_baseline`` ("port") for ``bench.py``; (2) it is fast enough to act
as the checker at BASELINE config-2 size (1024 lines) in the GPU tests.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module.  Pinned against the reference-generated goldens by
``tests/test_oracle_crnn.py``.
"""
from __future__ import annotations

from typing import Dict

import numpy as np
import torch
import torch.nn.functional as F


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


class TorchCrnnOracle:
    """fp32 CPU forward of None-VGG-BiLSTM-CTC.

    reference: marie/models/icr/model.py:70-92,
    marie/models/icr/modules/feature_extraction.py:12-28,
    marie/models/icr/modules/sequence_modeling.py:11-19.
    """

    def __init__(self, state: Dict[str, np.ndarray], threads: int | None = None):
        if threads:
            torch.set_num_threads(int(threads))
        self.st = {k: _t(v) for k, v in state.items() if v.ndim > 0}
        self.lstm = []
        for j, insz in ((0, 512), (1, 256)):
            m = torch.nn.LSTM(insz, 256, bidirectional=True, batch_first=True)
            p = f"SequenceModeling.{j}.rnn."
            with torch.no_grad():
                for name, par in m.named_parameters():
                    par.copy_(self.st[p + name])
            m.eval()
            self.lstm.append(m)

    @torch.no_grad()
    def logits(self, x: np.ndarray) -> np.ndarray:
        """x: fp32 (N,1,32,W) -> (N,T,C)."""
        st = self.st
        p = "FeatureExtraction.ConvNet."
        x = _t(x)
        x = F.max_pool2d(F.relu(F.conv2d(x, st[p + "0.weight"], st[p + "0.bias"], padding=1)), 2, 2)
        x = F.max_pool2d(F.relu(F.conv2d(x, st[p + "3.weight"], st[p + "3.bias"], padding=1)), 2, 2)
        x = F.relu(F.conv2d(x, st[p + "6.weight"], st[p + "6.bias"], padding=1))
        x = F.max_pool2d(F.relu(F.conv2d(x, st[p + "8.weight"], st[p + "8.bias"], padding=1)), (2, 1), (2, 1))
        x = F.conv2d(x, st[p + "11.weight"], None, padding=1)
        x = F.relu(F.batch_norm(x, st[p + "12.running_mean"], st[p + "12.running_var"],
                                st[p + "12.weight"], st[p + "12.bias"], False, 0.0, 1e-5))
        x = F.conv2d(x, st[p + "14.weight"], None, padding=1)
        x = F.relu(F.batch_norm(x, st[p + "15.running_mean"], st[p + "15.running_var"],
                                st[p + "15.weight"], st[p + "15.bias"], False, 0.0, 1e-5))
        x = F.max_pool2d(x, (2, 1), (2, 1))
        x = F.relu(F.conv2d(x, st[p + "18.weight"], st[p + "18.bias"]))
        x = x.permute(0, 3, 1, 2).mean(dim=3)            # AdaptiveAvgPool2d((None,1)) + squeeze(3)
        for j in (0, 1):
            r, _ = self.lstm[j](x)
            x = F.linear(r, st[f"SequenceModeling.{j}.linear.weight"], st[f"SequenceModeling.{j}.linear.bias"])
        y = F.linear(x.contiguous(), st["Prediction.weight"], st["Prediction.bias"])
        return y.numpy()

    @torch.no_grad()
    def decode(self, logits: np.ndarray, charset: str):
        """greedy CTC + confidence, exactly the reference's torch calls —
        marie/document/craft_ocr_processor.py:240-272, marie/models/icr/utils.py:41-54."""
        preds = _t(logits)
        _, idx = preds.max(2)
        prob = F.softmax(preds, dim=2)
        pmax, _ = prob.max(dim=2)
        conf = pmax.cumprod(dim=1)[:, -1]
        character = ["[CTCblank]"] + list(charset)
        texts = []
        for row in idx.tolist():
            chars = [character[t] for i, t in enumerate(row) if t != 0 and not (i > 0 and row[i - 1] == t)]
            texts.append("".join(chars).upper())
        return idx.numpy().astype(np.int32), texts, conf.numpy()


def default_init_state(seed: int = 0, num_class: int = 95) -> Dict[str, np.ndarray]:
    """PyTorch default initialisation of None-VGG-BiLSTM-CTC under
    ``torch.manual_seed(seed)``, creating the parameters in the same order as the
    reference constructor (marie/models/icr/model.py:41-64) so that the RNG
    stream — and therefore every tensor — is identical to ``Model(opt)``'s.
    ``oracle/gen_golden.py`` asserts that identity against the real reference.
    """
    import torch.nn as nn

    torch.manual_seed(seed)
    oc = [64, 128, 256, 512]
    convnet = nn.Sequential(
        nn.Conv2d(1, oc[0], 3, 1, 1), nn.ReLU(True), nn.MaxPool2d(2, 2),
        nn.Conv2d(oc[0], oc[1], 3, 1, 1), nn.ReLU(True), nn.MaxPool2d(2, 2),
        nn.Conv2d(oc[1], oc[2], 3, 1, 1), nn.ReLU(True),
        nn.Conv2d(oc[2], oc[2], 3, 1, 1), nn.ReLU(True), nn.MaxPool2d((2, 1), (2, 1)),
        nn.Conv2d(oc[2], oc[3], 3, 1, 1, bias=False), nn.BatchNorm2d(oc[3]), nn.ReLU(True),
        nn.Conv2d(oc[3], oc[3], 3, 1, 1, bias=False), nn.BatchNorm2d(oc[3]), nn.ReLU(True),
        nn.MaxPool2d((2, 1), (2, 1)),
        nn.Conv2d(oc[3], oc[3], 2, 1, 0), nn.ReLU(True))
    seq = []
    for insz in (512, 256):
        rnn = nn.LSTM(insz, 256, bidirectional=True, batch_first=True)
        lin = nn.Linear(512, 256)
        seq.append((rnn, lin))
    pred = nn.Linear(256, num_class)
    st: Dict[str, np.ndarray] = {}
    for k, v in convnet.state_dict().items():
        st["FeatureExtraction.ConvNet." + k] = v.numpy().copy()
    for j, (rnn, lin) in enumerate(seq):
        for k, v in rnn.state_dict().items():
            st[f"SequenceModeling.{j}.rnn.{k}"] = v.numpy().copy()
        for k, v in lin.state_dict().items():
            st[f"SequenceModeling.{j}.linear.{k}"] = v.numpy().copy()
    for k, v in pred.state_dict().items():
        st["Prediction." + k] = v.numpy().copy()
    return st
