"""BASELINE configs[0] as a plumbing case, no GPU: 4 synthetic 2550 x 3300 pages through the ``OcrEngine`` surface
(marie/ocr/ocr_engine.py:93-221, examples/batch_document_ocr.py drives the same call remotely) with CPU processors — here the
oracles (oracle/dit_pipeline.py, oracle/trocr_torch.py) wrapped in the BoxProcessor / OcrProcessor interfaces.  What is checked
is the host flow the product shares with the GPU path: ``process_single`` -> ``extract_bounding_boxes`` -> ``recognize`` ->
coordinate conversion -> meta, on full-size pages.  (The product itself has no CPU mode by design: its processors raise without
a GPU; tests/test_abi_cpu.py.)  The detector runs at a reduced MIN_SIZE_TEST and the recognizer is the small seeded model so
that the case takes well under a minute."""
import numpy as np
import pytest

PAGE_H, PAGE_W = 3300, 2550


class _OracleBox:
    def __init__(self, state):
        from oracle.dit_pipeline import OracleDitBoxProcessor

        self.o = OracleDitBoxProcessor(state, refinement=False, min_size=160, max_size=400)
        self.calls = 0

    def extract_bounding_boxes(self, _id, key, img, psm=None, **kw):
        self.calls += 1
        rects, frags, numbers, lines = self.o.extract_bounding_boxes(img)
        return rects, frags, numbers, {"bboxes": rects}, lines


def _make_ocr(state, enc, dec):
    from marie_icr_amd.ocr_processor import OcrProcessor
    from oracle.trocr_torch import TorchTrocrOracle, preprocess_fragments

    class _OracleOcr(OcrProcessor):
        def __init__(self):
            super().__init__("/tmp/icr", False)
            self.o = TorchTrocrOracle(state, enc[2], dec[2], beam=3, max_len_b=6)
            self.batches = []

        def recognize_from_fragments(self, images, **kw):
            self.batches.append(len(images))
            out = []
            for s0 in range(0, len(images), 64):
                hyps = self.o.generate(preprocess_fragments(images[s0:s0 + 64]))
                for k, (toks, score) in enumerate(hyps):
                    out.append({"confidence": round(float(np.exp(score)), 4), "id": f"img-{s0 + k}",
                                "text": " ".join(str(int(t)) for t in toks if t not in (0, 2))})
            return out

    return _OracleOcr()


def test_four_full_size_pages_through_the_engine_surface_on_cpu():
    import torch

    from marie_icr_amd.box_processor import PSMode
    from marie_icr_amd.ocr_engine import CoordinateFormat, MarieHipOcrEngine
    from marie_icr_amd.weights import make_dit_state, make_page_bgr, make_trocr_state

    torch.set_num_threads(8)
    enc, dec = (256, 2, 4), (256, 2, 4, 512)
    box = _OracleBox(make_dit_state(0))
    ocr = _make_ocr(make_trocr_state(0, enc, dec, 97, 32), enc, dec)
    eng = MarieHipOcrEngine(box_processor=box, default_ocr_processor=ocr)
    pages = [make_page_bgr(2000 + i, PAGE_H, PAGE_W, n_lines=40) for i in range(4)]
    res = eng.extract(pages, PSMode.SPARSE, CoordinateFormat.XYXY)
    assert len(res) == 4 and box.calls == 4 and len(ocr.batches) == 4          # the reference's per-page loop
    for i, r in enumerate(res):
        assert r["meta"]["page"] == i and r["meta"]["format"] == "xyxy"
        assert r["meta"]["imageSize"] == {"width": PAGE_W, "height": PAGE_H}
        words = r["words"]
        assert len(words) == ocr.batches[i] > 0
        for w in words:
            x0, y0, x1, y1 = (int(v) for v in w["box"])
            assert 0 <= x0 < x1 <= PAGE_W and 0 <= y0 < y1 <= PAGE_H
            assert 0.0 <= w["confidence"] <= 1.0 and isinstance(w["text"], str)
        assert [w["word_index"] for w in words] == list(range(len(words)))
        assert sorted({int(w["line"]) for w in words}) == sorted(int(l) for l in np.unique(r["meta"]["lines"]))
        assert len(r["lines"]) == len({int(w["line"]) for w in words})
        for ln in r["lines"]:
            assert ln["text"] == " ".join(w["text"] for w in words if w["id"] in ln["wordids"])
    # XYWH of the same page: boxes convert back
    again = eng.extract(pages[:1], PSMode.SPARSE, CoordinateFormat.XYWH)[0]
    for a, b in zip(again["words"], res[0]["words"]):
        x, y, w, h = (int(v) for v in a["box"])
        assert [x, y, x + w, y + h] == [int(v) for v in b["box"]] and a["text"] == b["text"]
