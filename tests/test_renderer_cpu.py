"""The step after the path: TextRenderer / get_words_and_boxes against text written by the reference's own TextRenderer
(tests/golden/text_renderer.json, made by ``oracle/gen_golden.py --renderer-only`` from marie/renderer/text_renderer.py)."""
import copy
import io
import json
import os

import numpy as np
import pytest

from marie_icr_amd.renderer import TextRenderer, get_words_and_boxes
from marie_icr_amd.weights import make_ocr_result

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "text_renderer.json"), encoding="UTF-8"))


@pytest.mark.parametrize("case", GOLD["cases"], ids=lambda c: f"seed{c['seed']}")
def test_text_renderer_matches_reference_output(case, tmp_path):
    res = make_ocr_result(case["seed"], case["width"], case["height"], case["n_lines"])
    out = tmp_path / "page.txt"
    TextRenderer().render([np.zeros((case["height"], case["width"], 3), np.uint8)], [copy.deepcopy(res)], str(out))
    assert out.read_text(encoding="UTF-8") == case["text"]


def test_multi_page_document_and_streams(tmp_path):
    mp = GOLD["multi_page"]
    results = [make_ocr_result(s, mp["width"], mp["height"], mp["n_lines"], page=i) for i, s in enumerate(mp["seeds"])]
    frames = [np.zeros((mp["height"], mp["width"], 3), np.uint8)] * 3
    out = tmp_path / "doc.txt"
    r = TextRenderer(config={"preserve_interword_spaces": "true"})
    assert r.name == "TextRenderer" and r.preserve_interword_spaces is True
    r.render(frames, copy.deepcopy(results), str(out))
    text = out.read_text(encoding="UTF-8")
    assert text == mp["text"] and text.count("\f") == 2
    buf = io.BytesIO()
    TextRenderer().render(frames, copy.deepcopy(results), buf)
    assert buf.getvalue().decode("UTF-8") == mp["text"]


def test_failed_page_is_skipped_and_format_is_converted(tmp_path):
    good = make_ocr_result(3, 640, 480, 3)
    bad = make_ocr_result(3, 640, 480, 3)
    bad["words"][0]["box"][0] = 5000                                   # starts outside the page: that page raises
    out = tmp_path / "two.txt"
    TextRenderer().render([np.zeros((480, 640, 3), np.uint8)] * 2, [bad, copy.deepcopy(good)], str(out))
    assert out.read_text(encoding="UTF-8") == "\f" + GOLD["cases"][3]["text"]
    xyxy = copy.deepcopy(good)
    xyxy["meta"]["format"] = "xyxy"
    before = [list(w["box"]) for w in xyxy["words"]]
    TextRenderer().render([np.zeros((480, 640, 3), np.uint8)], [xyxy], str(out))
    assert [w["box"] for w in xyxy["words"]] == [[x, y, x + w, y + h] for x, y, w, h in before]   # renderer.py:47-63


def test_get_words_and_boxes():
    res = [make_ocr_result(0, 850, 1100, 14), make_ocr_result(1, 850, 1100, 5)]
    words, boxes = get_words_and_boxes(res, 1)
    assert words == [w["text"] for w in res[1]["words"]] and boxes == [w["box"] for w in res[1]["words"]]
    w3, b3, l3 = get_words_and_boxes(res, 0, include_lines=True)
    assert l3 == [w["line"] for w in res[0]["words"]] and len(w3) == len(b3) == len(l3)
    assert get_words_and_boxes([], 0) == ([], [])
    with pytest.raises(ValueError):
        get_words_and_boxes(res, 2)
