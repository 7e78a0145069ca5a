"""ORACLE (test infrastructure, not product code) — CPU fp32 restatement of the TrOCR recognizer as the reference runs it
(marie/document/trocr_ocr_processor.py:116-180): PIL bicubic 384 x 384 -> DeiT encoder (oracle/vit_torch.py, DeiT flavour)
-> fairseq TransformerDecoder with RoBERTa arguments -> TextRecognitionGenerator._generate
(marie/models/unilm/trocr/generator.py:11-374, the reference's own file, read as text) with fairseq's BeamSearch.step and
finalize_hypos.

PARITY UNPINNED: fairseq and timm are third-party, absent from /root/reference and not installed here (unpinned git HEAD in
the reference's Dockerfiles; timm==0.6.12); no reference test or fixture holds outputs of this path.  The decoder layer,
incremental decoding, beam step and finalisation are restated from fairseq v0.12's published algorithm.
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this module.
"""
from __future__ import annotations

import math
from typing import Dict, List

import numpy as np
import torch
import torch.nn.functional as F
from PIL import Image

from oracle.vit_torch import TorchVitOracle

ENC_PREFIX = "encoder.deit."


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def preprocess_fragments(fragments_bgr, size: int = 384) -> np.ndarray:
    """MemoryDataset (BGR -> RGB PIL) + preprocess_image: convert("RGB").resize((384, 384), BICUBIC) -> uint8 RGB."""
    out = np.empty((len(fragments_bgr), size, size, 3), np.uint8)
    for i, f in enumerate(fragments_bgr):
        out[i] = np.asarray(Image.fromarray(np.ascontiguousarray(f[:, :, ::-1])).convert("RGB").resize((size, size), Image.BICUBIC))
    return out


class TorchTrocrOracle:
    def __init__(self, state: Dict[str, np.ndarray], enc_heads: int, dec_heads: int, beam: int = 3, max_len_b: int = 200,
                 min_len: int = 1, pad: int = 1, eos: int = 2, embed_scale: float = 1.0, img: int = 384):
        self.st = {k: _t(v) for k, v in state.items() if k.startswith("decoder.")}
        enc = {k[len(ENC_PREFIX):]: v for k, v in state.items() if k.startswith(ENC_PREFIX)}
        g = img // 16
        self.vit = TorchVitOracle(enc, enc_heads, pos_hw=(g, g), taps=())
        self.H = dec_heads
        self.beam, self.min_len, self.pad, self.eos, self.embed_scale, self.img = beam, min_len, pad, eos, embed_scale, img
        self.layers = 1 + max(int(k.split(".")[2]) for k in self.st if k.startswith("decoder.layers."))
        max_positions = self.st["decoder.embed_positions.weight"].shape[0] - pad - 1
        self.max_len = min(max_len_b, max_positions - 1)
        self.vocab = self.st["decoder.embed_tokens.weight"].shape[0]

    @torch.no_grad()
    def encode(self, crops_rgb_u8: np.ndarray) -> torch.Tensor:
        x = TorchVitOracle.preprocess(crops_rgb_u8, self.img, self.img, swap_rb=False)
        t, _ = self.vit.tokens(x)
        return t                                                # (B, 577, E)

    def _mha(self, p, q_in, k, v):
        """fairseq MultiheadAttention for one query position; k, v already projected: (B, S, D)."""
        st = self.st
        B, D = q_in.shape
        hd = D // self.H
        q = F.linear(q_in, st[p + "q_proj.weight"], st[p + "q_proj.bias"]) * hd ** -0.5
        q = q.view(B, self.H, 1, hd)
        kh = k.view(B, -1, self.H, hd).transpose(1, 2)
        vh = v.view(B, -1, self.H, hd).transpose(1, 2)
        a = torch.softmax(q @ kh.transpose(-2, -1), dim=-1)
        o = (a @ vh).transpose(1, 2).reshape(B, D)
        return F.linear(o, st[p + "out_proj.weight"], st[p + "out_proj.bias"])

    @torch.no_grad()
    def decoder_step(self, tok: torch.Tensor, step: int, hist, cross):
        """TransformerDecoder.forward in incremental mode: last token only; ``hist[l]`` = (K, V) of the previous steps."""
        st = self.st
        D = st["decoder.embed_tokens.weight"].shape[1]
        x = self.embed_scale * st["decoder.embed_tokens.weight"][tok] + st["decoder.embed_positions.weight"][self.pad + step + 1]
        x = F.layer_norm(x, (D,), st["decoder.layernorm_embedding.weight"], st["decoder.layernorm_embedding.bias"], 1e-5)
        for l in range(self.layers):
            p = f"decoder.layers.{l}."
            k_new = F.linear(x, st[p + "self_attn.k_proj.weight"], st[p + "self_attn.k_proj.bias"]).unsqueeze(1)
            v_new = F.linear(x, st[p + "self_attn.v_proj.weight"], st[p + "self_attn.v_proj.bias"]).unsqueeze(1)
            K = k_new if hist[l] is None else torch.cat((hist[l][0], k_new), dim=1)
            V = v_new if hist[l] is None else torch.cat((hist[l][1], v_new), dim=1)
            hist[l] = (K, V)
            x = F.layer_norm(x + self._mha(p + "self_attn.", x, K, V), (D,), st[p + "self_attn_layer_norm.weight"],
                             st[p + "self_attn_layer_norm.bias"], 1e-5)
            x = F.layer_norm(x + self._mha(p + "encoder_attn.", x, cross[l][0], cross[l][1]), (D,),
                             st[p + "encoder_attn_layer_norm.weight"], st[p + "encoder_attn_layer_norm.bias"], 1e-5)
            h = F.linear(F.gelu(F.linear(x, st[p + "fc1.weight"], st[p + "fc1.bias"])), st[p + "fc2.weight"], st[p + "fc2.bias"])
            x = F.layer_norm(x + h, (D,), st[p + "final_layer_norm.weight"], st[p + "final_layer_norm.bias"], 1e-5)
        w = st.get("decoder.output_projection.weight", st["decoder.embed_tokens.weight"])
        return F.linear(x, w)

    @torch.no_grad()
    def score_tokens(self, crops_rgb_u8: np.ndarray, token_lists) -> List[float]:
        """Teacher-forced score of a given hypothesis per crop (tokens incl. the final eos): sum of the step log-probabilities
        / length — the quantity ``generate`` ranks finished hypotheses by.  Used to show that a hypothesis chosen by a
        reduced-precision run is a near-tie of the oracle's best."""
        st = self.st
        enc = self.encode(crops_rgb_u8)
        out = []
        for i, toks in enumerate(token_lists):
            cross = []
            for l in range(self.layers):
                p = f"decoder.layers.{l}.encoder_attn."
                cross.append((F.linear(enc[i:i + 1], st[p + "k_proj.weight"], st[p + "k_proj.bias"]),
                              F.linear(enc[i:i + 1], st[p + "v_proj.weight"], st[p + "v_proj.bias"])))
            hist = [None] * self.layers
            prev, total = self.eos, 0.0
            for step, t in enumerate(toks):
                logits = self.decoder_step(torch.tensor([prev]), step, hist, cross)
                total += float(F.log_softmax(logits.float(), dim=-1)[0, int(t)])
                prev = int(t)
            out.append(total / max(len(toks), 1))
        return out

    @torch.no_grad()
    def generate(self, crops_rgb_u8: np.ndarray, want_step0: bool = False, want_trace: bool = False):
        """Returns per crop (tokens incl. eos, normalised score) of the best hypothesis.  ``want_trace``: also the candidate
        list of every step as BeamSearch.step produced it, one entry longer than the generator reads (2 * beam + 1: the first
        candidate that did not make the list bounds the last gap) — {"scores", "tokens", "beams"}: [steps][bsz][2 * beam + 1],
        "active": [steps][bsz] (False once a crop has finished), "prefixes": per step, per crop, the token prefix of every beam
        row BEFORE the step (so a candidate (beam, token) names a hypothesis independently of row order)."""
        st = self.st
        beam, K2, eos, pad, ML = self.beam, 2 * self.beam, self.eos, self.pad, self.max_len
        enc = self.encode(crops_rgb_u8)
        bsz = enc.shape[0]
        enc_b = enc.repeat_interleave(beam, dim=0)                  # reorder_encoder_out with new_order
        cross = []
        for l in range(self.layers):
            p = f"decoder.layers.{l}.encoder_attn."
            cross.append((F.linear(enc_b, st[p + "k_proj.weight"], st[p + "k_proj.bias"]),
                          F.linear(enc_b, st[p + "v_proj.weight"], st[p + "v_proj.bias"])))
        hist = [None] * self.layers
        M = bsz * beam
        tokens = torch.full((M, ML + 2), pad, dtype=torch.long)
        tokens[:, 0] = eos
        scores = torch.zeros(M, ML + 1)
        finalized: List[List[dict]] = [[] for _ in range(bsz)]
        finished = [False] * bsz
        ignore = torch.zeros(bsz, beam, dtype=torch.bool)
        remaining = bsz
        step0 = None
        trace = {"scores": [], "tokens": [], "beams": [], "active": [], "prefixes": []}
        for step in range(ML + 1):
            logits = self.decoder_step(tokens[:, step], step, hist, cross)
            if step == 0 and want_step0:
                step0 = logits.view(bsz, beam, -1)[:, 0].numpy().copy()
            lprobs = F.log_softmax(logits.float(), dim=-1)
            lprobs[lprobs != lprobs] = -math.inf
            lprobs[:, pad] = -math.inf
            if step >= ML:
                lprobs[:, :eos] = -math.inf
                lprobs[:, eos + 1:] = -math.inf
            elif step < self.min_len:
                lprobs[:, eos] = -math.inf
            lp = lprobs.view(bsz, beam, -1)
            if step == 0:
                lp = lp[:, ::beam, :].contiguous()
            else:
                lp = lp + scores.view(bsz, beam, -1)[:, :, step - 1].unsqueeze(-1)
            flat = lp.view(bsz, -1)
            order = torch.argsort(flat, dim=1, descending=True, stable=True)[:, :K2]      # topk; ties: lower flat index
            cand_scores = torch.gather(flat, 1, order)
            cand_beams = torch.div(order, self.vocab, rounding_mode="trunc")
            cand_tokens = order.fmod(self.vocab)
            if want_trace:
                o7 = torch.argsort(flat, dim=1, descending=True, stable=True)[:, :K2 + 1]
                trace["scores"].append(torch.gather(flat, 1, o7).numpy().copy())
                trace["beams"].append(torch.div(o7, self.vocab, rounding_mode="trunc").numpy().copy())
                trace["tokens"].append(o7.fmod(self.vocab).numpy().copy())
                trace["active"].append(np.array([not f for f in finished]))
                trace["prefixes"].append([[tuple(int(v) for v in tokens[s * beam + b, 1:step + 1]) for b in range(beam)]
                                          for s in range(bsz)])
            new_tokens, new_scores = tokens.clone(), scores.clone()
            parent = torch.arange(M)
            for s in range(bsz):
                if finished[s]:
                    continue
                eos_mask = (cand_tokens[s] == eos) & (cand_scores[s] != -math.inf)
                eos_mask[:beam][ignore[s]] = False
                for j in range(beam):
                    if eos_mask[j] and len(finalized[s]) < beam:
                        src = s * beam + int(cand_beams[s, j])
                        toks = tokens[src, 1:step + 2].clone()
                        toks[step] = eos
                        finalized[s].append({"tokens": toks, "score": float(cand_scores[s, j]) / (step + 1)})
                if eos_mask[:beam].any() and (len(finalized[s]) == beam or step == ML):
                    finished[s] = True
                    remaining -= 1
                    continue
                eos_mask[:beam] = eos_mask[:beam] | ignore[s]
                active_mask = eos_mask.long() * K2 + torch.arange(K2)
                vals, hyp = torch.topk(active_mask, k=beam, largest=False)
                ignore[s] = vals >= K2
                for b in range(beam):
                    j = int(hyp[b])
                    src, row = s * beam + int(cand_beams[s, j]), s * beam + b
                    new_tokens[row, :step + 1] = tokens[src, :step + 1]
                    new_tokens[row, step + 1] = cand_tokens[s, j]
                    if step > 0:
                        new_scores[row, :step] = scores[src, :step]
                    new_scores[row, step] = cand_scores[s, j]
                    parent[row] = src
            tokens, scores = new_tokens, new_scores
            hist = [(k.index_select(0, parent), v.index_select(0, parent)) for k, v in hist]      # reorder_incremental_state
            if remaining == 0:
                break
        out = []
        for s in range(bsz):
            sc = torch.tensor([h["score"] for h in finalized[s]])
            best = finalized[s][int(torch.sort(sc, descending=True, stable=True)[1][0])]
            out.append((best["tokens"].numpy(), best["score"]))
        if want_trace:
            tr = {k: (np.stack(v) if k not in ("prefixes",) else v) for k, v in trace.items()}
            tr["finalized"] = [[h["score"] for h in finalized[s]] for s in range(bsz)]
            return (out, step0, tr) if want_step0 else (out, tr)
        return (out, step0) if want_step0 else out
