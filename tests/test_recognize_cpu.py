"""OcrProcessor.recognize (marie/document/ocr_processor.py:87-267) on hand-built boxes / lines, including the reference's quirk Q1
(SURVEY.md section 8): BoxProcessorUlimDit returns boxes and fragments permuted by its final (line, x) lexsort but
``rect_line_numbers`` in detection order (ulim_dit_box_processor.py:799-823), so ``recognize`` pairs sorted box i with the line
number of detection i.  The expectation below is the reference's own procedure written out (per line id, a scan over all words)."""
import numpy as np
import pytest

from marie_icr_amd.ocr_processor import OcrProcessor


class _Canned(OcrProcessor):
    def __init__(self, texts):
        self.texts = texts

    def is_available(self):
        return True

    def recognize_from_fragments(self, images, **kwargs):
        return [{"confidence": 0.5 + 0.01 * k, "id": f"img-{k}", "text": self.texts[k]} for k in range(len(images))]


def _by_the_book(boxes, lines, results):
    """ocr_processor.py:160-253 as written there."""
    boxes, lines = np.array(boxes), np.array(lines)
    words = []
    for i, index in enumerate(np.argsort(boxes[:, 0])):
        words.append({"id": i, "text": results[index]["text"], "confidence": round(results[index]["confidence"], 3),
                      "box": boxes[index].tolist(), "line": int(lines[index])})
    out_lines, aligned, word_index = [], [], 0
    for i, line_numer in enumerate(sorted(np.unique(lines))):
        ids, picks, txt, conf = [], [], [], []
        for word in words:
            if line_numer == word["line"]:
                word["word_index"] = word_index
                ids.append(word["id"]); picks.append(word["box"]); txt.append(word["text"]); conf.append(word["confidence"])
                aligned.append(word)
                word_index += 1
        p = np.array(picks)
        x0, y0 = p[:, 0].min(), p[:, 1].min()
        x1, y1 = (p[:, 0] + p[:, 2]).max(), (p[:, 1] + p[:, 3]).max()
        out_lines.append({"line": i + 1, "wordids": ids, "text": " ".join(txt), "bbox": [int(x0), int(y0), int(x1 - x0), int(y1 - y0)],
                          "confidence": round(float(np.average(conf)), 4)})
    return aligned, out_lines


def _plain(v):
    if isinstance(v, dict):
        return {k: _plain(x) for k, x in v.items()}
    if isinstance(v, (list, tuple, np.ndarray)):
        return [_plain(x) for x in v]
    if isinstance(v, np.generic):
        return v.item()
    return v


CASES = {
    # detection order = (y, x) lexsort of a slightly skewed page: the second row's first word sits higher than the first row's last
    "q1_skew": dict(det_boxes=[[10, 10, 40, 12], [60, 12, 40, 12], [10, 20, 40, 12], [110, 15, 40, 12], [60, 31, 40, 12]],
                    det_lines=[1, 1, 2, 1, 2]),
    "in_order": dict(det_boxes=[[5, 5, 20, 10], [30, 5, 20, 10], [5, 25, 20, 10]], det_lines=[1, 1, 2]),
    "single": dict(det_boxes=[[3, 4, 9, 9]], det_lines=[1]),
    "gap_in_ids": dict(det_boxes=[[5, 5, 20, 10], [30, 45, 20, 10], [60, 5, 20, 10]], det_lines=[1, 4, 1]),
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_recognize_pairs_sorted_boxes_with_detection_order_lines(name):
    c = CASES[name]
    det_boxes, det_lines = np.array(c["det_boxes"]), list(c["det_lines"])
    ind = np.lexsort((det_boxes[:, 0], np.array(det_lines)))          # what the box processor does to boxes and fragments ...
    boxes = det_boxes[ind]
    fragments = [np.zeros((int(b[3]), int(b[2]), 3), np.uint8) for b in boxes]
    lines = det_lines                                                 # ... and does not do to the line numbers (Q1)
    texts = [f"W{k}" for k in range(len(boxes))]
    proc = _Canned(texts)
    result, overlay = proc.recognize("id", "key", np.zeros((100, 200, 3), np.uint8), boxes, fragments, lines)
    assert overlay is None
    words, out_lines = _by_the_book(boxes, lines, proc.recognize_from_fragments(fragments))
    assert _plain(result["words"]) == _plain(words)
    assert _plain(list(result["lines"])) == _plain(out_lines)
    assert result["meta"] == {"imageSize": {"width": 200, "height": 100}, "page": 0, "lang": "en"}
    if name == "q1_skew":
        assert not np.array_equal(ind, np.arange(len(ind)))           # the lexsort really permutes here
        mis = [w for w in result["words"] if det_lines[int(np.flatnonzero((det_boxes == w["box"]).all(1))[0])] != w["line"]]
        assert mis, "this case must contain a word whose line differs from its detection's line (the quirk)"


def test_blank_page_and_length_checks():
    proc = _Canned([])
    res, overlay = proc.recognize("id", "key", np.zeros((20, 30, 3), np.uint8), [], [], [])
    assert res == {"meta": {"imageSize": {"width": 30, "height": 20}, "page": 0, "lang": "en"}, "words": [], "lines": []}
    assert overlay.shape == (20, 30, 3) and (overlay == 255).all()
    with pytest.raises(AssertionError):
        proc.recognize("id", "key", np.zeros((20, 30, 3), np.uint8), [[0, 0, 1, 1]], [], [1])
    with pytest.raises(Exception):
        proc.recognize("id", "key", None, [], [], [])
