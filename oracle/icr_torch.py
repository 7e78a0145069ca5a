"""ORACLE (test infrastructure, not product code) — CPU fp32 restatement of the reference's production recognizer
TPS-ResNet-BiLSTM-Attn (``CraftOcrProcessor``'s model, marie/document/craft_ocr_processor.py:49-70) with torch
functional ops, and of its decode rule.

PINNED: ``oracle/gen_golden.py`` runs the reference's unmodified ``Model(opt)`` (marie/models/icr/model.py) on seeded
weights and ``tests/test_oracle_icr.py`` checks this restatement against those vectors.
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this module.
"""
from __future__ import annotations

from typing import Dict, List

import numpy as np
import torch
import torch.nn.functional as F


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


class TorchIcrOracle:
    def __init__(self, state: Dict[str, np.ndarray], batch_max_length: int = 48):
        self.st = {k: _t(v) for k, v in state.items() if np.ndim(v) > 0}
        self.max_len = batch_max_length
        self.lstm = []
        for j, insz in ((0, 512), (1, 256)):
            m = torch.nn.LSTM(insz, 256, bidirectional=True, batch_first=True)
            with torch.no_grad():
                for name, par in m.named_parameters():
                    par.copy_(self.st[f"SequenceModeling.{j}.rnn.{name}"])
            self.lstm.append(m.eval())

    # -- helpers ---------------------------------------------------------------------------------------------------
    def _cb(self, x, conv, bn, stride=1, padding=1, relu=True):
        st = self.st
        y = F.conv2d(x, st[conv + ".weight"], None, stride=stride, padding=padding)
        y = F.batch_norm(y, st[bn + ".running_mean"], st[bn + ".running_var"], st[bn + ".weight"], st[bn + ".bias"],
                         False, 0.0, 1e-5)
        return F.relu(y) if relu else y

    def _block(self, x, p, downsample):
        """BasicBlock — marie/models/icr/modules/feature_extraction.py:116-150."""
        out = self._cb(x, p + "conv1", p + "bn1")
        out = self._cb(out, p + "conv2", p + "bn2", relu=False)
        res = self._cb(x, p + "downsample.0", p + "downsample.1", padding=0, relu=False) if downsample else x
        return F.relu(out + res)

    # -- stages ------------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def tps(self, x: torch.Tensor) -> torch.Tensor:
        """TPS_SpatialTransformerNetwork.forward — marie/models/icr/modules/transformation.py:32-42,78-86,158-167."""
        st = self.st
        p = "Transformation.LocalizationNetwork."
        h = F.max_pool2d(self._cb(x, p + "conv.0", p + "conv.1"), 2, 2)
        h = F.max_pool2d(self._cb(h, p + "conv.4", p + "conv.5"), 2, 2)
        h = F.max_pool2d(self._cb(h, p + "conv.8", p + "conv.9"), 2, 2)
        h = self._cb(h, p + "conv.12", p + "conv.13")
        h = F.adaptive_avg_pool2d(h, 1).view(x.shape[0], -1)
        h = F.relu(F.linear(h, st[p + "localization_fc1.0.weight"], st[p + "localization_fc1.0.bias"]))
        c_prime = F.linear(h, st[p + "localization_fc2.weight"], st[p + "localization_fc2.bias"]).view(x.shape[0], -1, 2)
        g = "Transformation.GridGenerator."
        b = x.shape[0]
        cz = torch.cat((c_prime, torch.zeros(b, 3, 2)), dim=1)
        t = torch.bmm(st[g + "inv_delta_C"].repeat(b, 1, 1), cz)
        p_prime = torch.bmm(st[g + "P_hat"].repeat(b, 1, 1), t)
        grid = p_prime.reshape(b, x.shape[2], x.shape[3], 2)
        return F.grid_sample(x, grid, padding_mode="border", align_corners=True)

    @torch.no_grad()
    def resnet(self, x: torch.Tensor) -> torch.Tensor:
        """ResNet.forward — marie/models/icr/modules/feature_extraction.py:212-246."""
        r = "FeatureExtraction.ConvNet."
        x = self._cb(x, r + "conv0_1", r + "bn0_1")
        x = self._cb(x, r + "conv0_2", r + "bn0_2")
        x = F.max_pool2d(x, 2, 2)
        x = self._block(x, r + "layer1.0.", True)
        x = self._cb(x, r + "conv1", r + "bn1")
        x = F.max_pool2d(x, 2, 2)
        x = self._block(x, r + "layer2.0.", True)
        x = self._block(x, r + "layer2.1.", False)
        x = self._cb(x, r + "conv2", r + "bn2")
        x = F.max_pool2d(x, 2, (2, 1), (0, 1))
        x = self._block(x, r + "layer3.0.", True)
        for i in range(1, 5):
            x = self._block(x, r + f"layer3.{i}.", False)
        x = self._cb(x, r + "conv3", r + "bn3")
        for i in range(3):
            x = self._block(x, r + f"layer4.{i}.", False)
        x = self._cb(x, r + "conv4_1", r + "bn4_1", stride=(2, 1), padding=(0, 1))
        x = self._cb(x, r + "conv4_2", r + "bn4_2", stride=1, padding=0)
        return x

    @torch.no_grad()
    def attention(self, batch_h: torch.Tensor) -> torch.Tensor:
        """Attention.forward(is_train=False) + AttentionCell — marie/models/icr/modules/prediction.py:27-83."""
        st = self.st
        a = "Prediction.attention_cell."
        b = batch_h.shape[0]
        steps = self.max_len + 1
        nc = st["Prediction.generator.weight"].shape[0]
        h = torch.zeros(b, 256)
        c = torch.zeros(b, 256)
        targets = torch.zeros(b, dtype=torch.long)
        probs = torch.zeros(b, steps, nc)
        for i in range(steps):
            onehot = F.one_hot(targets, nc).float()
            hproj = F.linear(batch_h, st[a + "i2h.weight"])
            e = F.linear(torch.tanh(hproj + F.linear(h, st[a + "h2h.weight"], st[a + "h2h.bias"]).unsqueeze(1)),
                         st[a + "score.weight"])
            alpha = F.softmax(e, dim=1)
            context = torch.bmm(alpha.permute(0, 2, 1), batch_h).squeeze(1)
            gates = F.linear(torch.cat([context, onehot], 1), st[a + "rnn.weight_ih"], st[a + "rnn.bias_ih"]) + \
                F.linear(h, st[a + "rnn.weight_hh"], st[a + "rnn.bias_hh"])
            ig, fg, gg, og = gates.chunk(4, 1)
            c = torch.sigmoid(fg) * c + torch.sigmoid(ig) * torch.tanh(gg)
            h = torch.sigmoid(og) * torch.tanh(c)
            step = F.linear(h, st["Prediction.generator.weight"], st["Prediction.generator.bias"])
            probs[:, i, :] = step
            targets = step.max(1)[1]
        return probs

    @torch.no_grad()
    def logits(self, x: np.ndarray, want_stages: bool = False):
        """x fp32 (N,1,32,100) in [-1,1] -> (N, 49, 96).  reference: marie/models/icr/model.py:70-92."""
        xt = _t(x)
        rect = self.tps(xt)
        feat = self.resnet(rect)
        v = feat.permute(0, 3, 1, 2).mean(dim=3)                      # AdaptiveAvgPool2d((None, 1)) + squeeze
        hcur = v
        for j in (0, 1):
            r, _ = self.lstm[j](hcur)
            hcur = F.linear(r, self.st[f"SequenceModeling.{j}.linear.weight"], self.st[f"SequenceModeling.{j}.linear.bias"])
        out = self.attention(hcur.contiguous())
        if want_stages:
            return out.numpy(), {"rectified": rect.numpy(), "features": feat.numpy(), "contextual": hcur.numpy()}
        return out.numpy()


def attn_decode(logits: np.ndarray, charset: str):
    """Greedy decode of the attention head exactly as the reference does it
    (marie/document/craft_ocr_processor.py:244-272 with AttnLabelConverter.decode, marie/models/icr/utils.py:142-148):
    join the token STRINGS ('[GO]', '[s]', characters), cut at the first "[s]" found in that string, and take the
    product of the per-step max softmax probabilities over the same number of leading STEPS as the cut position in the
    string (a quirk when a '[GO]' token precedes the end: string position != step index).  Lines whose cut is empty get
    text "" and confidence 0 (the reference raises IndexError there and abandons the rest of its batch).
    Returns (argmax (N,S) int32, upper-cased strings, confidences fp32)."""
    character = ["[GO]", "[s]"] + list(charset)
    lt = _t(logits)
    idx = lt.max(2)[1]
    pmax = F.softmax(lt, dim=2).max(dim=2)[0]
    texts: List[str] = []
    confs = np.zeros((logits.shape[0],), np.float32)
    for n, row in enumerate(idx.tolist()):
        pred = "".join(character[i] for i in row)
        eos = pred.find("[s]")
        pred = pred[:eos]
        pm = pmax[n][:eos]
        if pm.numel() == 0:
            texts.append("")
            confs[n] = 0.0
        else:
            texts.append(pred.upper())
            confs[n] = float(pm.cumprod(dim=0)[-1])
    return idx.numpy().astype(np.int32), texts, confs
