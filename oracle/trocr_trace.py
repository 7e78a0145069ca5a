"""ORACLE-SIDE TEST INFRASTRUCTURE (not product code) — decision-margin analysis of the TrOCR beam search.

``TextRecognitionGenerator._generate`` (marie/models/unilm/trocr/generator.py:127-362, restated in oracle/trocr_torch.py) is a
chain of discrete decisions on continuous scores: at every step fairseq's ``BeamSearch.step`` sorts ``beam x vocabulary``
cumulative log-probabilities and keeps the best ``2 * beam``.  A second implementation whose scores differ from the oracle's by
a small eps can only take a different path where the oracle's own sorted list has two neighbours within 2 * eps of each other.
This module makes that statement checkable for a reduced-precision run that exposes its candidate lists
(``mhip_trocr_generate_trace_host``):

* ``walk`` goes through both searches step by step, per crop.  While the two candidate lists name the same (beam, token)
  sequence the searches are in the same state and the score difference of every candidate is MEASURED (eps of this crop so
  far).  At the first step where the lists differ, the difference must be explained: the oracle's list (one entry longer than
  the generator reads) is cut into groups of neighbours closer than ``2 * margin * eps``; the other list may permute candidates
  inside a group, nothing else.  A crop that never diverges must return the oracle's tokens.
* ``certificate`` is the statement for models with margins (north_star: "string-exact for the same decode rule"): when the
  oracle's best hypothesis is the chain of its top-1 candidates and every top-1 / top-2 gap along it, and its lead over the
  other finished hypotheses, exceed ``factor`` times the measured error, the reduced-precision run has to return the same
  tokens.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this module.
"""
from __future__ import annotations

from typing import Dict, List, Sequence

import numpy as np


def _groups(scores: np.ndarray, thr: float) -> np.ndarray:
    """group id per position of a descending list: neighbours closer than ``thr`` (or both -inf) share a group"""
    gid = np.zeros(len(scores), np.int64)
    for j in range(1, len(scores)):
        a, b = scores[j - 1], scores[j]
        same = (np.isneginf(a) and np.isneginf(b)) or (np.isfinite(a) and np.isfinite(b) and a - b <= thr)
        gid[j] = gid[j - 1] if same else gid[j - 1] + 1
    return gid


def walk(oracle_trace: Dict[str, np.ndarray], gpu_trace: Dict[str, np.ndarray], margin: float = 1.5) -> List[Dict[str, object]]:
    """Per crop: {"steps_equal", "diverged_at" (None = never), "explained" (the divergence is a near-tie under the error measured
    up to and including that step), "eps" (max |score difference| over the candidates compared), "eps_by_step"}."""
    so, to, bo, act = oracle_trace["scores"], oracle_trace["tokens"], oracle_trace["beams"], oracle_trace["active"]
    sg, tg, bg = gpu_trace["scores"], gpu_trace["tokens"], gpu_trace["beams"]
    n, k2 = so.shape[1], sg.shape[2]
    out = []
    for s in range(n):
        eps, eps_by_step, diverged, explained, steps_equal = 0.0, [], None, True, 0
        for t in range(so.shape[0]):
            if not act[t, s]:
                break
            if t >= sg.shape[0]:
                diverged, explained = t, False          # the other search stopped while the oracle's was still running
                break
            o_s, g_s = so[t, s], sg[t, s]
            o_key = [(int(bo[t, s, j]), int(to[t, s, j])) for j in range(k2 + 1)]
            g_key = [(int(bg[t, s, j]), int(tg[t, s, j])) for j in range(k2)]
            o_map = {k: float(o_s[j]) for j, k in enumerate(o_key) if np.isfinite(o_s[j])}
            e_t = max([abs(float(g_s[j]) - o_map[k]) for j, k in enumerate(g_key) if np.isfinite(g_s[j]) and k in o_map] or [0.0])
            eps = max(eps, e_t)
            eps_by_step.append(e_t)
            same = all(o_key[j] == g_key[j] for j in range(k2) if np.isfinite(o_s[j]) or np.isfinite(g_s[j]))
            if same:
                steps_equal += 1
                continue
            diverged = t
            gid = _groups(o_s, 2.0 * margin * eps)
            members: Dict[int, set] = {}
            for j, k in enumerate(o_key):
                members.setdefault(int(gid[j]), set()).add(k)
            open_group = int(gid[k2])                    # reaches the end of the oracle's list: may hold candidates it does not show
            for j in range(k2):
                if not (np.isfinite(o_s[j]) or np.isfinite(g_s[j])):
                    continue
                g = int(gid[j])
                if g_key[j] not in members[g] and g != open_group:
                    explained = False
            break
        out.append({"steps_equal": steps_equal, "diverged_at": diverged, "explained": explained, "eps": eps,
                    "eps_by_step": eps_by_step})
    return out


def certificate(oracle_trace: Dict[str, np.ndarray], hyps: Sequence, finalized_scores: Sequence[Sequence[float]], eos: int,
                eps: float, factor: float = 10.0) -> List[Dict[str, object]]:
    """Per crop: does the oracle's own run carry margins ``factor`` x ``eps`` everywhere it matters?  (``eps``: one bound, or one
    per crop — the score error ``walk`` measured on that crop's own candidates.)
    (1) its best hypothesis is the chain of top-1 candidates (each continuing the previous one: row 0), ending in </s>;
    (2) every top-1 / top-2 gap of cumulative score along the chain is at least ``factor * eps``;
    (3) its normalised score leads every other finished hypothesis of the oracle by at least ``factor * eps / length``.
    Returns {"holds", "min_gap", "final_lead", "chain"}."""
    so, to, bo, act = oracle_trace["scores"], oracle_trace["tokens"], oracle_trace["beams"], oracle_trace["active"]
    out = []
    for s, (tokens, score) in enumerate(hyps):
        tokens = [int(v) for v in tokens]
        chain_ok, min_gap = True, float("inf")
        L = len(tokens)
        for t in range(L):
            if t >= so.shape[0] or not act[t, s]:
                chain_ok = False
                break
            if int(to[t, s, 0]) != tokens[t] or (t > 0 and int(bo[t, s, 0]) != 0):
                chain_ok = False
                break
            min_gap = min(min_gap, float(so[t, s, 0] - so[t, s, 1]))
        chain_ok = chain_ok and L > 0 and tokens[-1] == eos
        others = sorted((float(v) for v in finalized_scores[s]), reverse=True)[1:]
        lead = float(score) - others[0] if others else float("inf")
        e_s = float(eps[s]) if hasattr(eps, "__len__") else float(eps)        # one bound for all crops, or the one measured per crop
        holds = bool(chain_ok and min_gap >= factor * e_s and lead >= factor * e_s / max(L, 1))
        out.append({"holds": holds, "min_gap": min_gap, "final_lead": lead, "chain": chain_ok})
    return out
