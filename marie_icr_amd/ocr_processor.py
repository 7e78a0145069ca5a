"""``OcrProcessor`` base: the word/line assembly around a recognizer.

Restates ``OcrProcessor.recognize`` (reference: marie/document/ocr_processor.py:87-267) and
``merge_bboxes_as_block`` (marie/utils/overlap.py:186-204).  This is per-word host logic in the reference too; what it
drops are the reference's side effects only: /tmp debug directories (:131-132) and the optional PIL overlay drawing
(``return_overlay=True`` returns a blank white overlay here).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import numpy as np


def merge_bboxes_as_block(bboxes) -> list:
    """reference: marie/utils/overlap.py:186-204."""
    bboxes = np.array(bboxes)
    min_x = bboxes[:, 0].min()
    min_y = bboxes[:, 1].min()
    max_h = (bboxes[:, 1] + bboxes[:, 3]).max() - min_y
    max_w = (bboxes[:, 0] + bboxes[:, 2]).max() - min_x
    return [round(k, 6) for k in [min_x, min_y, max_w, max_h]]


class OcrProcessor:
    """Base class of OCR processors (surface of marie/document/ocr_processor.py:34-96)."""

    def __init__(self, work_dir: str = "/tmp/icr", cuda: bool = True, **kwargs) -> None:
        self.cuda = cuda
        self.work_dir = work_dir

    def is_available(self) -> bool:
        raise NotImplementedError

    def recognize_from_fragments(self, image_fragments, **kwargs) -> List[Dict[str, object]]:
        raise Exception("Not Implemented")

    def recognize_from_boxes(self, image, boxes, **kwargs):
        raise Exception("Not yet implemented")

    def extract_text(self, _id, key, image):
        """reference: ocr_processor.py:49-66."""
        results = self.recognize_from_boxes([image], [0, 0, image.shape[1], image.shape[0]])
        if len(results) == 1:
            r = results[0]
            return r["text"], r["confidence"]
        return None, 0

    @staticmethod
    def _check_inputs(img, boxes, fragments, lines):
        if img is None:
            raise Exception("Input image can't be empty")
        if not isinstance(img, np.ndarray):
            try:
                from PIL import Image

                if isinstance(img, Image.Image):           # PIL RGB -> OpenCV BGR (:112-114)
                    img = np.asarray(img)[:, :, ::-1].copy()
            except ImportError:
                pass
        if not isinstance(img, np.ndarray):
            raise Exception("Expected image in numpy format but got {}".format(type(img)))
        assert len(boxes) == len(fragments), "You must provide the same number of box groups as images."
        assert len(boxes) == len(lines), "You must provide the same number of lines as boxes."
        return img

    def recognize(self, _id, key, img: np.ndarray, boxes, fragments, lines,
                  return_overlay: Optional[bool] = False) -> Tuple[Dict, Optional[np.ndarray]]:
        """reference: ocr_processor.py:87-267."""
        img = self._check_inputs(img, boxes, fragments, lines)
        if len(boxes) == 0:                                 # blank page (:147-154)
            return self._assemble(img.shape, boxes, lines, [], True)
        results = self.recognize_from_fragments(fragments)
        return self._assemble(img.shape, boxes, lines, results, return_overlay)

    def recognize_pages(self, _id, key, pages, return_overlay: Optional[bool] = False):
        """``recognize`` for several pages with ONE recognizer call: ``pages`` = [(img, boxes, fragments, lines), ...]; the
        fragments of all pages are pooled into a single ``recognize_from_fragments`` batch (the reference calls its
        recognizer page by page, ocr_engine.py:172-199; a page's result does not depend on its batch).  Returns the list of
        ``recognize`` results."""
        from .fragments import FragmentList

        checked = [(self._check_inputs(img, b, f, l), b, f, l) for img, b, f, l in pages]
        pooled = FragmentList.concat([f for _, b, f, _ in checked if len(b)])
        results = self.recognize_from_fragments(pooled) if len(pooled) else []
        assert len(results) == len(pooled), "You must provide the same number of results as fragments."
        out, k = [], 0
        for img, boxes, fragments, lines in checked:
            if len(boxes) == 0:
                out.append(self._assemble(img.shape, boxes, lines, [], True))
                continue
            out.append(self._assemble(img.shape, boxes, lines, results[k:k + len(fragments)], return_overlay))
            k += len(fragments)
        return out

    def _assemble(self, shape, boxes, lines, results, return_overlay):
        """Words in reading order, lines, confidences — reference: ocr_processor.py:140-267."""
        meta = {"imageSize": {"width": shape[1], "height": shape[0]}, "page": 0, "lang": "en"}
        if len(boxes) == 0:                                 # blank page (:147-154)
            overlay_image = np.ones((shape[0], shape[1], 3), dtype=np.uint8) * 255
            return {"meta": meta, "words": [], "lines": []}, overlay_image

        assert len(results) == len(boxes), "You must provide the same number of results as fragments."
        words = []
        boxes = np.array(boxes)
        lines = np.array(lines)
        indices = np.argsort(boxes[:, 0])                   # LTR reading order (:164)
        for i, index in enumerate(indices):
            extraction = results[index]
            words.append({"id": i, "text": extraction["text"], "confidence": round(extraction["confidence"], 3),
                          "box": boxes[index], "line": lines[index]})

        unique_line_ids = sorted(np.unique(lines))
        line_results = np.empty(len(unique_line_ids), dtype=object)
        aligned_words = []
        word_index = 0
        by_line: Dict[int, list] = {}                       # the reference scans all words per line id; same order, one pass
        for word in words:
            by_line.setdefault(int(word["line"]), []).append(word)
        # per line: the block around its boxes (merge_bboxes_as_block) and the mean confidence (np.average) — the same numpy
        # reductions as the reference's per-line calls, run once per page over segments instead of 8 small calls per line
        starts, order, confs = [], [], []
        for line_numer in unique_line_ids:
            group = by_line.get(int(line_numer), ())
            if len(group) == 0:
                raise Exception("Every word needs to be associated with a box")
            starts.append(len(order))
            for word in group:
                order.append(indices[word["id"]])
                confs.append(word["confidence"])
        bsel = boxes[np.asarray(order)]
        st = np.asarray(starts)
        min_x = np.minimum.reduceat(bsel[:, 0], st)
        min_y = np.minimum.reduceat(bsel[:, 1], st)
        max_h = np.maximum.reduceat(bsel[:, 1] + bsel[:, 3], st) - min_y
        max_w = np.maximum.reduceat(bsel[:, 0] + bsel[:, 2], st) - min_x
        counts = np.diff(np.append(st, len(order)))
        means = np.add.reduceat(np.asarray(confs, dtype=np.float64), st) / counts
        for i, line_numer in enumerate(unique_line_ids):
            word_ids, _w = [], []
            for word in by_line.get(int(line_numer), ()):
                word["word_index"] = word_index
                word_ids.append(word["id"])
                _w.append(word["text"])
                aligned_words.append(word)
                word_index += 1
            line_results[i] = {"line": i + 1, "wordids": word_ids, "text": " ".join(_w),
                               "bbox": [round(k, 6) for k in [min_x[i], min_y[i], max_w[i], max_h[i]]],
                               "confidence": round(means[i], 4)}
        result = {"meta": meta, "words": aligned_words, "lines": line_results}
        if len(words) != len(aligned_words):
            raise Exception(f"Aligned words should match original words got: {len(aligned_words)}, {len(words)}")
        overlay_image = np.ones((shape[0], shape[1], 3), dtype=np.uint8) * 255 if return_overlay else None
        return result, overlay_image
