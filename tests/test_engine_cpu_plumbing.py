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


# ---- the batched path's producer thread (detector of batch k + 1 under the recognizer of batch k) -----------------------------
class _Ctx:
    """stands in for a device context: the engine only compares the two processors' contexts by identity here"""

    def set_stream(self, _):
        pass


class _CountingBox:
    def __init__(self):
        self.ctx, self.batches = _Ctx(), []

    def extract_bounding_boxes_batch(self, _id, key, frames, psm=None):
        self.batches.append(len(frames))
        out = []
        for f in frames:
            boxes = [[1, 1, 5, 5], [8, 1, 5, 5]]
            out.append((boxes, [f[1:6, 1:6], f[1:6, 8:13]], [1, 1], {}, [[1, 1, 12, 5]]))
        return out


class _FailingOcr:
    def __init__(self, fail_on_call):
        self.ctx, self.calls, self.fail_on_call = _Ctx(), 0, fail_on_call

    def recognize_pages(self, _id, key, pages):
        self.calls += 1
        if self.calls == self.fail_on_call:
            raise RuntimeError("recognizer out of memory")
        return [({"meta": {}, "words": [{"box": [1, 1, 5, 5]}], "lines": []}, None) for _ in pages]


def test_recognizer_error_in_the_batched_path_stops_the_detector_thread():
    """ocr_engine.OcrEngine._fullpage_batched: when the consumer side raises, the producer must not stay blocked in q.put
    holding detected pages; the call raises the recognizer's error and leaves no thread behind."""
    import threading

    from marie_icr_amd.box_processor import PSMode
    from marie_icr_amd.ocr_engine import CoordinateFormat, MarieHipOcrEngine

    frames = [np.full((16, 16, 3), 255, np.uint8) for _ in range(10)]
    box, ocr = _CountingBox(), _FailingOcr(fail_on_call=2)
    eng = MarieHipOcrEngine(box_processor=box, default_ocr_processor=ocr)
    eng.page_batch, eng.first_batch = 2, 2                     # 5 chunks; the recognizer fails on the second
    before = threading.active_count()
    with pytest.raises(RuntimeError, match="out of memory"):
        eng.extract(frames, PSMode.SPARSE, CoordinateFormat.XYWH)
    assert threading.active_count() == before                 # the producer thread has exited
    assert len(box.batches) < 5                               # and did not run on to the end of the document
    # the same engine still works afterwards (nothing left locked or queued)
    ocr.fail_on_call = -1
    res = eng.extract(frames, PSMode.SPARSE, CoordinateFormat.XYWH)
    assert [r["meta"]["page"] for r in res] == list(range(10))


def test_batched_path_uses_a_short_first_batch_when_batches_overlap():
    from marie_icr_amd.box_processor import PSMode
    from marie_icr_amd.ocr_engine import CoordinateFormat, MarieHipOcrEngine

    frames = [np.full((16, 16, 3), 255, np.uint8) for _ in range(21)]
    box, ocr = _CountingBox(), _FailingOcr(fail_on_call=-1)
    eng = MarieHipOcrEngine(box_processor=box, default_ocr_processor=ocr)
    eng.page_batch, eng.first_batch = 8, 2
    res = eng.extract(frames, PSMode.SPARSE, CoordinateFormat.XYWH)
    assert box.batches == [2, 8, 8, 3] and len(res) == 21
    # one shared context: no overlap, plain batches
    ocr.ctx = box.ctx
    box.batches.clear()
    eng.extract(frames, PSMode.SPARSE, CoordinateFormat.XYWH)
    assert box.batches == [8, 8, 5]


# ---- constructor contract (ocr_engine.py:35-70, default_ocr_engine.py:32-58, voting_ocr_engine.py:34-64) --------------------------
def test_engine_constructors_build_the_reference_default_processors(monkeypatch, tmp_path):
    import marie_icr_amd.craft as craft
    import marie_icr_amd.dit_box_processor as ditbp
    import marie_icr_amd.icr as icr
    import marie_icr_amd.trocr as trocr
    from marie_icr_amd.ocr_engine import MarieHipOcrEngine, MarieHipVotingOcrEngine

    made = []

    def fake(name):
        class _P:
            def __init__(self, *a, **kw):
                made.append((name, a, kw))
        return _P

    monkeypatch.setattr(ditbp, "BoxProcessorUlimDit", fake("dit"))
    monkeypatch.setattr(craft, "BoxProcessorCraft", fake("craft"))
    monkeypatch.setattr(trocr, "TrOcrProcessor", fake("trocr"))
    monkeypatch.setattr(icr, "CraftOcrProcessor", fake("icr"))

    MarieHipOcrEngine()                                                    # box_segmentation_mode defaults to "1" = DiT
    assert [m[0] for m in made] == ["dit", "trocr"]
    assert made[0][2] == {"work_dir": "/tmp/boxes", "models_dir": None, "cuda": True}
    assert made[1][2] == {"work_dir": "/tmp/icr", "cuda": True, "model_name_or_path": None}
    made.clear()
    MarieHipOcrEngine(box_segmentation_mode="2", models_dir=str(tmp_path))
    assert [m[0] for m in made] == ["craft", "trocr"]
    assert made[0][2]["models_dir"] == str(tmp_path / "craft")
    assert made[1][2]["model_name_or_path"] == str(tmp_path / "trocr" / "trocr-large-printed.pt")
    made.clear()
    with pytest.raises(Exception, match="Unsupported box segmentation mode : 3"):
        MarieHipOcrEngine(box_segmentation_mode=3)
    made.clear()
    monkeypatch.setenv("MARIE_DISABLE_CUDA", "1")                          # ocr_engine.py:49-52
    MarieHipOcrEngine()
    assert made[0][2]["cuda"] is False and made[1][2]["cuda"] is False
    monkeypatch.delenv("MARIE_DISABLE_CUDA")
    made.clear()
    eng = MarieHipVotingOcrEngine()                                        # default = TrOCR, "craft" = the ICR recognizer
    assert [m[0] for m in made] == ["dit", "trocr", "icr"]
    assert list(eng.processors) == ["default", "craft"] and eng.processors["default"]["default"] is True
    # an explicit processor is used as given, nothing is built
    made.clear()
    box = object()
    e2 = MarieHipOcrEngine(box_processor=box, default_ocr_processor="ocr")
    assert made == [] and e2.box_processor is box and e2.ocr_processor == "ocr"


def test_default_processors_raise_the_loaders_error_for_a_missing_checkpoint(monkeypatch, tmp_path):
    """No GPU needed: the checkpoint is looked up before a device context is made, and the error is the loader's own
    (trocr_ocr_processor.py:216-217 ``FileNotFoundError(f"File not found : {model_path}")``)."""
    from marie_icr_amd import constants
    from marie_icr_amd.craft import BoxProcessorCraft
    from marie_icr_amd.dit_box_processor import BoxProcessorUlimDit
    from marie_icr_amd.icr import CraftOcrProcessor
    from marie_icr_amd.ocr_engine import MarieHipOcrEngine
    from marie_icr_amd.trocr import TrOcrProcessor

    monkeypatch.setattr(constants, "__model_path__", str(tmp_path))
    with pytest.raises(FileNotFoundError, match="model_0147999.pth"):
        BoxProcessorUlimDit(cuda=True)
    with pytest.raises(FileNotFoundError, match="craft_mlt_25k.pth"):
        BoxProcessorCraft(cuda=True)
    with pytest.raises(FileNotFoundError, match="trocr-large-printed.pt"):
        TrOcrProcessor(cuda=True)
    with pytest.raises(FileNotFoundError, match="best_accuracy.pth"):
        CraftOcrProcessor(cuda=True)
    with pytest.raises(FileNotFoundError, match="model_0147999.pth"):
        MarieHipOcrEngine()
