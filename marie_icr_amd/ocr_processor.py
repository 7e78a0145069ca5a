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

    def recognize(self, _id, key, img: np.ndarray, boxes, fragments, lines,
                  return_overlay: Optional[bool] = False) -> Tuple[Dict, Optional[np.ndarray]]:
        """reference: ocr_processor.py:87-267."""
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

        shape = img.shape
        meta = {"imageSize": {"width": img.shape[1], "height": img.shape[0]}, "page": 0, "lang": "en"}
        if len(boxes) == 0:                                 # blank page (:147-154)
            overlay_image = np.ones((shape[0], shape[1], 3), dtype=np.uint8) * 255
            return {"meta": meta, "words": [], "lines": []}, overlay_image

        results = self.recognize_from_fragments(fragments)
        assert len(results) == len(fragments), "You must provide the same number of results as fragments."
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
        for i, line_numer in enumerate(unique_line_ids):
            word_ids, box_picks, _w, _conf = [], [], [], []
            for word in words:
                if line_numer == word["line"]:
                    word["word_index"] = word_index
                    word_ids.append(word["id"])
                    box_picks.append(word["box"])
                    _w.append(word["text"])
                    _conf.append(word["confidence"])
                    aligned_words.append(word)
                    word_index += 1
            if len(box_picks) == 0:
                raise Exception("Every word needs to be associated with a box")
            line_results[i] = {"line": i + 1, "wordids": word_ids, "text": " ".join(_w),
                               "bbox": merge_bboxes_as_block(np.array(box_picks)),
                               "confidence": round(np.average(_conf), 4)}
        result = {"meta": meta, "words": aligned_words, "lines": line_results}
        if len(words) != len(aligned_words):
            raise Exception(f"Aligned words should match original words got: {len(aligned_words)}, {len(words)}")
        overlay_image = np.ones((shape[0], shape[1], 3), dtype=np.uint8) * 255 if return_overlay else None
        return result, overlay_image
