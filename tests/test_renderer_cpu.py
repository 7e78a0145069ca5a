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


# ---- the two XML on-disk formats: files byte-equal to what the reference's own classes write (oracle/gen_golden.py --xml-renderers-only)
XML_GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "xml_renderers.json"), encoding="UTF-8"))


def _mask_date(text):
    import re

    return re.sub(r'FIELD="CreationDate" VALUE="[^"]*"', 'FIELD="CreationDate" VALUE="*"', text)


@pytest.mark.parametrize("doc", range(len(XML_GOLD["documents"])))
def test_xml_renderers_write_the_reference_files(doc, tmp_path):
    import copy

    from marie_icr_amd.renderer import AdlibRenderer, BlobRenderer
    from oracle.gen_golden import xml_render_results

    results = xml_render_results()[doc]
    gold = XML_GOLD["documents"][doc]
    frames = [np.zeros((r["meta"]["imageSize"]["height"], r["meta"]["imageSize"]["width"], 3), np.uint8) for r in results]
    for key, cls in (("blob", BlobRenderer), ("adlib", AdlibRenderer)):
        out = tmp_path / key
        out.mkdir()
        cls(config={}).render(frames, copy.deepcopy(results), str(out))
        written = {name: (out / name).read_bytes().decode("UTF-8") for name in sorted(os.listdir(out))}
        assert sorted(written) == sorted(gold[key])
        for name in written:
            assert _mask_date(written[name]) == _mask_date(gold[key][name]), (key, name)


def test_xml_renderers_need_a_directory_and_skip_broken_pages(tmp_path):
    from marie_icr_amd.renderer import AdlibRenderer, BlobRenderer
    from marie_icr_amd.weights import make_ocr_result

    good = make_ocr_result(5, 600, 400, 2)
    broken = {"meta": good["meta"], "words": good["words"]}            # no "lines": the reference logs the KeyError and goes on
    frames = [np.zeros((400, 600, 3), np.uint8)] * 2
    for cls in (BlobRenderer, AdlibRenderer):
        with pytest.raises(ValueError):
            cls().render(frames, [good, good], str(tmp_path / "missing"))
    out = tmp_path / "o"
    out.mkdir()
    BlobRenderer().render(frames, [broken, good], str(out))
    assert sorted(os.listdir(out)) == ["2.BLOBS.XML"]
    AdlibRenderer(summary_filename="s.xml").render(frames, [broken, good], str(out), filename_generator=lambda n: f"p{n}.xml")
    assert sorted(os.listdir(out)) == ["2.BLOBS.XML", "p2.xml", "s.xml"]
    assert 'Filename="p1.xml"' in (out / "s.xml").read_text()          # the summary lists every frame, written or not
