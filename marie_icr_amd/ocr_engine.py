"""``OcrEngine`` surface over the MI355X box processor and recognizer.

Mirrors ``OcrEngine`` / ``DefaultOcrEngine`` (reference: marie/ocr/ocr_engine.py:28-433,
marie/ocr/default_ocr_engine.py:15-98) and ``CoordinateFormat`` (marie/ocr/coordinate_format.py:6-55): same
``extract(frames, pms_mode, coordinate_format, regions, queue_id, **kwargs)`` signature, same result dictionaries for
full-page and region extraction.

Output-invariant work of the reference that is NOT repeated per page (SURVEY.md §3.1 hot-loop notes): the deep copy of
every frame (ocr_engine.py:118), the md5 of all pixels used only as a debug-directory key (:119), and the white-canvas
copy when the padding is 0 (:181-184).  ``crop_to_content`` (an OpenCV blur/Otsu/morphology chain,
marie/utils/image_utils.py:190-252) is not provided on this path and raises if requested.
"""
from __future__ import annotations

import hashlib
from enum import Enum
from itertools import chain
from math import ceil
from typing import Any, Dict, List, Optional

import numpy as np

from .box_processor import PSMode


class CoordinateFormat(Enum):
    XYWH = "xywh"
    XYXY = "xyxy"

    @staticmethod
    def from_value(value):
        if value is None:
            return CoordinateFormat.XYWH
        for data in CoordinateFormat:
            if data.value == str(value).lower():
                return data
        return CoordinateFormat.XYWH

    @staticmethod
    def convert(box, from_mode: "CoordinateFormat", to_mode: "CoordinateFormat"):
        """reference: marie/ocr/coordinate_format.py:23-55."""
        arr = np.array(box)
        assert arr.shape == (4,), "CoordinateFormat.convert takes either a 4-tuple/list"
        if from_mode == to_mode:
            return box
        original_type = type(box)
        arr = arr.reshape(-1, 4)
        if to_mode == CoordinateFormat.XYXY and from_mode == CoordinateFormat.XYWH:
            arr[:, 2] += arr[:, 0]
            arr[:, 3] += arr[:, 1]
        elif from_mode == CoordinateFormat.XYXY and to_mode == CoordinateFormat.XYWH:
            arr[:, 2] -= arr[:, 0]
            arr[:, 3] -= arr[:, 1]
        else:
            raise RuntimeError("Cannot be here!")
        return original_type(arr.flatten())


def hash_frames_fast(frames, blocksize=2 ** 20) -> str:
    """md5 over the raw pixels in 1 MiB blocks — reference: marie/utils/image_utils.py:136-149."""
    md5 = hashlib.md5()
    if isinstance(frames, np.ndarray) and frames.ndim <= 3:
        frames = [frames] if frames.ndim == 3 else list(frames)
    for frame in frames:
        buf = np.ravel(frame)
        for k in range(ceil(len(buf) / blocksize)):
            md5.update(buf[k * blocksize:min((k + 1) * blocksize, len(buf))])
    return md5.hexdigest()


bbox_cache: Dict[str, Any] = {}   # module-global like the reference's (ocr_engine.py:20-25); not thread-safe


class OcrEngine:
    """reference: marie/ocr/ocr_engine.py:28-433."""

    def __init__(self, models_dir: Optional[str] = None, cuda: bool = True, *, box_processor=None, **kwargs) -> None:
        if box_processor is None:
            raise ValueError("pass the MI355X box processor explicitly (e.g. marie_icr_amd.craft.BoxProcessorCraft)")
        self.has_cuda = cuda
        self.box_processor = box_processor

    def extract(self, frames, pms_mode: PSMode = PSMode.SPARSE,
                coordinate_format: CoordinateFormat = CoordinateFormat.XYXY, regions=None, queue_id: str = None,
                **kwargs):
        raise NotImplementedError

    @staticmethod
    def as_frames(frames) -> List[np.ndarray]:
        """Frames as a list of HxWx3 uint8 BGR arrays.  PIL images are converted RGB -> BGR as
        ``OcrEngine.copy_frames`` does (ocr_engine.py:416-433); ndarrays are passed through without copying."""
        if isinstance(frames, np.ndarray) and frames.ndim == 3:
            frames = [frames]
        out = []
        for f in frames:
            if not isinstance(f, np.ndarray):
                f = np.asarray(f)[:, :, ::-1].copy()      # PIL RGB -> BGR
            out.append(f)
        return out

    def process_single(self, box_processor, icr_processor, frames, pms_mode: PSMode = PSMode.SPARSE,
                       coordinate_format: CoordinateFormat = CoordinateFormat.XYWH, regions=None,
                       queue_id: str = None, **kwargs: Any):
        """reference: ocr_engine.py:93-152."""
        queue_id = "0000-0000-0000-0000" if queue_id is None else queue_id
        regions = [] if regions is None else regions
        ro_frames = OcrEngine.as_frames(frames)
        checksum = "unhashed"       # only ever a debug-directory key in the reference
        if len(regions) == 0:
            return self._process_extract_fullpage(ro_frames, queue_id, checksum, pms_mode, coordinate_format,
                                                  box_processor, icr_processor, **kwargs)
        return self._process_extract_regions(ro_frames, queue_id, checksum, pms_mode, regions, box_processor,
                                             icr_processor, **kwargs)

    def _process_extract_fullpage(self, frames, queue_id, checksum, pms_mode, coordinate_format, box_processor,
                                  icr_processor, **kwargs):
        """reference: ocr_engine.py:154-221."""
        if kwargs.get("crop_to_content", False):
            raise NotImplementedError("crop_to_content is not available on the MI355X path")
        results = []
        for i, img in enumerate(frames):
            overlay = img                                   # padding == 0: the white canvas equals the page
            boxes, img_fragments, lines, _, line_bboxes = box_processor.extract_bounding_boxes(
                queue_id, checksum, overlay, pms_mode)
            result, _ = icr_processor.recognize(queue_id, checksum, overlay, boxes, img_fragments, lines)
            if CoordinateFormat.XYXY == coordinate_format:
                for word in result["words"]:
                    x, y, w, h = word["box"]
                    word["box"] = [x, y, x + w, y + h]
            result["meta"]["page"] = i
            result["meta"]["lines"] = lines
            result["meta"]["lines_bboxes"] = line_bboxes
            result["meta"]["format"] = coordinate_format.name.lower()
            results.append(result)
        return results

    def _process_extract_regions(self, frames, queue_id, checksum, pms_mode, regions, box_processor, icr_processor,
                                 **kwargs):
        """reference: ocr_engine.py:223-414."""
        output, extended = [], []
        for region in regions:
            if not all(key in region for key in ("id", "pageIndex", "x", "y", "w", "h")):
                raise Exception(f"Required key missing in region : {region}")
        pages: Dict[int, list] = {}
        for region in regions:
            pages.setdefault(region["pageIndex"], []).append(region)
        for page_index, page_regions in pages.items():
            img = frames[page_index]
            x_batch, y_batch, w_batch, h_batch = img.shape[1], img.shape[0], 0, 0
            region_ids = []
            bbox_results_batch = []
            for region in page_regions:
                rid = region["id"]
                region_ids.append(rid)
                x, y, w, h = region["x"], region["y"], region["w"], region["h"]
                if w == 0 or h == 0:
                    output.append({"id": rid, "text": "", "confidence": 0.0})
                    continue
                if y + h > img.shape[0] or x + w > img.shape[1]:
                    output.append({"id": rid, "text": "", "confidence": 0.0})
                    continue
                x_batch = min(x, x_batch)
                y_batch = min(y, y_batch)
                w_batch = max(x + w, x_batch + w_batch) - x_batch
                h_batch = max(y + h, h_batch + y_batch) - y_batch
                padding = 4
                region_overlay = np.ones((h + padding * 2, w + padding * 2, 3), dtype=np.uint8) * 255
                region_overlay[padding:h + padding, padding:w + padding] = img[y:y + h, x:x + w]
                mode = PSMode.from_value(region["mode"]) if "mode" in region else pms_mode
                cache_key = f"{id(region)}_{hash_frames_fast(region_overlay)}"
                bbox_results = bbox_cache.get(cache_key)
                if bbox_results is None:
                    bbox_results = box_processor.extract_bounding_boxes(queue_id, checksum, region_overlay, psm=mode)
                    bbox_cache[cache_key] = bbox_results
                bbox_results_batch.append(bbox_results)
            if not bbox_results_batch:
                # every region of this page was rejected above; the reference would fail on the empty zip(*[])
                extended.append({"meta": {}, "words": [], "lines": []})
                continue
            batch_crop = img[y_batch:y_batch + h_batch, x_batch:x_batch + w_batch]
            boxes, img_fragments, lines, _, lines_bboxes = (list(chain.from_iterable(x))
                                                            for x in zip(*bbox_results_batch))
            batch_result, _ = icr_processor.recognize(queue_id, checksum, batch_crop, boxes, img_fragments, lines)
            extended.append(batch_result)
            if "words" in batch_result and len(batch_result["words"]) == len(region_ids):
                for words, rid in zip(batch_result["words"], region_ids):
                    output.append({"id": rid, "text": words["text"], "confidence": words["confidence"]})
            else:
                for rid in region_ids:
                    output.append({"id": rid, "text": "", "confidence": 0.0})
        return {"regions": output, "extended": extended}


class MarieHipOcrEngine(OcrEngine):
    """Counterpart of ``DefaultOcrEngine`` (marie/ocr/default_ocr_engine.py:15-98) wired to the MI355X processors."""

    def __init__(self, models_dir: Optional[str] = None, cuda: bool = True, *, box_processor=None,
                 default_ocr_processor=None, **kwargs) -> None:
        super().__init__(models_dir=models_dir, cuda=cuda, box_processor=box_processor, **kwargs)
        if default_ocr_processor is None:
            raise ValueError("pass the MI355X recognizer explicitly (e.g. marie_icr_amd.crnn.CrnnOcrProcessor)")
        self.ocr_processor = default_ocr_processor

    def extract(self, frames, pms_mode: PSMode = PSMode.SPARSE,
                coordinate_format: CoordinateFormat = CoordinateFormat.XYWH, regions=None, queue_id: str = None,
                **kwargs: Any) -> List[Dict]:
        return self.process_single(self.box_processor, self.ocr_processor, frames, pms_mode, coordinate_format,
                                   regions, queue_id, **kwargs)
