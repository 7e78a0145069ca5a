"""VotingOcrEngine's evaluator (marie/ocr/voting_ocr_engine.py:186-482) restated in marie_icr_amd.ocr_engine — host logic,
checked on hand-built recognizer outputs and against goldens written by the reference's own evaluator (bottom of this file)."""
from collections import OrderedDict
from copy import deepcopy

from marie_icr_amd.ocr_engine import vote_words, voting_evaluator


def _w(i, text, conf):
    return {"id": i, "text": text, "confidence": conf, "box": [0, 0, 1, 1], "line": 1, "word_index": i}


def _page(words):
    return {"meta": {"page": 0}, "words": words, "lines": []}


def test_majority_wins_and_records_votes():
    c = [_w(0, "ACME", 0.5), _w(0, "ACNE", 0.99), _w(0, "ACME", 0.4)]
    got = vote_words(deepcopy(c))
    assert got["text"] == "ACME" and got["confidence"] == 0.5
    assert got["strategy"]["type"] == "voting" and got["strategy"]["candidates"] == 2
    assert all("box" not in v for v in got["strategy"]["votes"])


def test_equal_group_sizes_use_confidence_sum_then_first():
    c = [_w(0, "A", 0.3), _w(0, "B", 0.4), _w(0, "A", 0.3), _w(0, "B", 0.5)]
    assert vote_words(deepcopy(c))["text"] == "B"
    c = [_w(0, "A", 0.4), _w(0, "B", 0.4), _w(0, "A", 0.4), _w(0, "B", 0.4)]
    assert vote_words(deepcopy(c))["text"] == "A"          # equal sums: the earlier group stays


def test_no_majority_falls_back_to_default_or_confidence():
    got = vote_words([_w(0, "X", 0.9), _w(0, "Y", 0.5)])
    assert got["text"] == "X" and got["strategy"] == {"type": "default"}
    got = vote_words([_w(0, "X", 0.2), _w(0, "Y", 0.5), _w(0, "Z", 0.7)])
    assert got["text"] == "Z" and got["strategy"] == {"type": "confidence", "confidence": 0.7}


def test_page_mode_evaluator():
    agg = OrderedDict()
    agg["default"] = [_page([_w(0, "INVOICE", 0.8), _w(1, "T0TAL", 0.6)])]
    agg["craft"] = [_page([_w(0, "INVOICE", 0.7), _w(1, "TOTAL", 0.9)])]
    out = voting_evaluator(agg, agg["default"])
    words = out[0]["words"]
    assert [w["text"] for w in words] == ["INVOICE", "TOTAL"]
    assert words[0]["strategy"]["type"] == "voting" and words[0]["processor"] == "default"
    assert words[1]["strategy"]["type"] == "confidence" and words[1]["processor"] == "craft"
    assert all(isinstance(w["id"], str) for w in words)


def test_region_mode_evaluator_and_empty_aggregate():
    regions = [{"id": 7, "pageIndex": 0, "x": 0, "y": 0, "w": 5, "h": 5}]
    ext_a = {"id": "7", "words": [_w(0, "12", 0.5), _w(1, "34", 0.9)]}
    ext_b = {"id": "7", "words": [_w(0, "12", 0.6), _w(1, "84", 0.3)]}
    agg = OrderedDict()
    agg["default"] = {"regions": [{"id": 7, "text": "12 34", "confidence": 0.7}], "extended": [ext_a]}
    agg["craft"] = {"regions": [{"id": 7, "text": "12 84", "confidence": 0.45}], "extended": [ext_b]}
    out = voting_evaluator(agg, agg["default"], regions)
    r = out["regions"][0]
    assert r["id"] == "7" and r["text"] == "12 34" and r["original_text"] == "12 34"
    assert r["confidence"] == round((0.5 + 0.9) / 2, 4)
    empty = voting_evaluator(OrderedDict(), None, deepcopy(regions))
    assert empty["regions"][0]["text"] == "" and empty["regions"][0]["confidence"] == 0


# ---- goldens written by the reference's own VotingOcrEngine.voting_evaluator (oracle/gen_golden.py --voting-only loads
# marie/ocr/voting_ocr_engine.py by path): 12 page-mode and 12 region-mode seeded cases with 2-4 recognizers, plus the
# "nothing to evaluate" branch
def _golden_cases():
    import json
    import os

    with open(os.path.join(os.path.dirname(__file__), "golden", "voting.json"), encoding="UTF-8") as f:
        return json.load(f)["cases"]


def test_evaluator_matches_reference_goldens():
    from marie_icr_amd.weights import make_voting_case

    n_conf = n_vote = n_default = 0
    for case in _golden_cases():
        names, agg, reg = make_voting_case(case["seed"], case["regions"])
        if case.get("empty"):
            got = voting_evaluator(OrderedDict(), None, deepcopy(reg))
        else:
            a = OrderedDict((n, deepcopy(agg[n])) for n in names)
            got = voting_evaluator(a, a[names[0]], deepcopy(reg))
        assert got == case["expected"], case["seed"]
        units = got if not case["regions"] else got.get("extended", [])
        for u in units:
            for w in u["words"]:
                t = w["strategy"]["type"]
                n_conf += t == "confidence"; n_vote += t == "voting"; n_default += t == "default"
    assert n_conf > 5 and n_vote > 20 and n_default > 5          # every rule of the evaluator is exercised by the cases
