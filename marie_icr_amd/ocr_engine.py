"""``OcrEngine`` surface over the MI355X box processor and recognizer.

Mirrors ``OcrEngine`` / ``DefaultOcrEngine`` (reference: marie/ocr/ocr_engine.py:28-433,
marie/ocr/default_ocr_engine.py:15-98) and ``CoordinateFormat`` (marie/ocr/coordinate_format.py:6-55): same
``extract(frames, pms_mode, coordinate_format, regions, queue_id, **kwargs)`` signature, same result dictionaries for
full-page and region extraction.

Output-invariant work of the reference that is NOT repeated per page (SURVEY.md §3.1 hot-loop notes): the deep copy of
every frame (ocr_engine.py:118), the md5 of all pixels used only as a debug-directory key (:119), and the white-canvas
copy when the padding is 0 (:181-184).  ``crop_to_content`` (an OpenCV blur/Otsu/morphology chain,
marie/utils/image_utils.py:190-252) runs in csrc/content_ops.hip (marie_icr_amd/content.py); the page then sits on a white canvas
with 4 px of padding as in :169-184.
"""
from __future__ import annotations

import contextlib
import hashlib
import os
from copy import deepcopy
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
        """reference: ocr_engine.py:35-70.  Without a box processor one is built from ``box_segmentation_mode`` (1 = DiT,
        2 = CRAFT) with the model zoo's default checkpoints; a missing checkpoint is the loader's FileNotFoundError.  The
        reference ignores ``models_dir`` here (its processors read ``__model_path__``); given explicitly it is handed on."""
        self.work_dir_icr = "/tmp/icr"
        has_cuda = cuda
        if os.environ.get("MARIE_DISABLE_CUDA"):
            has_cuda = False
        self.has_cuda = has_cuda
        if box_processor is not None:
            self.box_processor = box_processor
            return
        box_segmentation_mode = int(kwargs.get("box_segmentation_mode", "1"))
        if box_segmentation_mode == 1:
            from .dit_box_processor import BoxProcessorUlimDit

            self.box_processor = BoxProcessorUlimDit(work_dir="/tmp/boxes", models_dir=models_dir, cuda=has_cuda)
        elif box_segmentation_mode == 2:
            from .craft import BoxProcessorCraft

            self.box_processor = BoxProcessorCraft(
                work_dir="/tmp/boxes", models_dir=None if models_dir is None else os.path.join(models_dir, "craft"),
                cuda=has_cuda)
        else:
            raise Exception(f"Unsupported box segmentation mode : {box_segmentation_mode}")

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
            # ocr_engine.py:169-184: the page is cropped to its content and set on a white canvas with 4 px of padding
            from .content import crop_to_content

            ctx = getattr(box_processor, "ctx", None) or getattr(icr_processor, "ctx", None)
            if ctx is None:
                raise RuntimeError("crop_to_content needs a processor with a device context (ctx)")
            padded = []
            for img in frames:
                img = crop_to_content(ctx, img)
                h, w = img.shape[:2]
                canvas = np.full((h + 8, w + 8, 3), 255, np.uint8)
                canvas[4:h + 4, 4:w + 4] = img if img.ndim == 3 else img[:, :, None]
                padded.append(canvas)
            frames = padded
        if hasattr(box_processor, "extract_bounding_boxes_batch") and hasattr(icr_processor, "recognize_pages"):
            return self._fullpage_batched(frames, queue_id, checksum, pms_mode, coordinate_format, box_processor,
                                          icr_processor)
        results = []
        for i, img in enumerate(frames):
            overlay = img                                   # padding == 0: the white canvas equals the page
            boxes, img_fragments, lines, _, line_bboxes = box_processor.extract_bounding_boxes(
                queue_id, checksum, overlay, pms_mode)
            result, _ = icr_processor.recognize(queue_id, checksum, overlay, boxes, img_fragments, lines)
            results.append(self._finish_page(result, i, lines, line_bboxes, coordinate_format))
        return results

    @staticmethod
    def _finish_page(result, i, lines, line_bboxes, coordinate_format):
        """reference: ocr_engine.py:201-219."""
        if CoordinateFormat.XYXY == coordinate_format:
            for word in result["words"]:
                x, y, w, h = word["box"]
                word["box"] = [x, y, x + w, y + h]
        result["meta"]["page"] = i
        result["meta"]["lines"] = lines
        result["meta"]["lines_bboxes"] = line_bboxes
        result["meta"]["format"] = coordinate_format.name.lower()
        return result

    phase_align = False  # see _fullpage_batched
    page_batch = 32      # pages per detector / recognizer batch of the batched full-page path
    stream_recognizer = True
    stream_batch = 8     # pages per detector batch when the recognizer takes its pages in batches and finishes them together
    first_batch = 8      # pages of the first batch when batches overlap: nothing hides the first detector batch, so it is short

    def _fullpage_batched(self, frames, queue_id, checksum, pms_mode, coordinate_format, box_processor, icr_processor):
        """The per-page loop of ocr_engine.py:172-221 with both models batched: ``page_batch`` pages go through the detector
        together (same-size pages share a forward) and their fragments are pooled into one recognizer batch.  A page's result
        is what the per-page loop returns (tests/test_pipeline_gpu.py).  With more than one batch and the two processors on
        different contexts, the detector of batch k + 1 runs on its own stream and host thread under the recognizer of batch
        k — under its DECODE phase when the recognizer has one and says where it starts (``decode_gate``, a
        ``_lib.PhaseGate``): both models are MFMA-bound while the recognizer encodes, so racing them there only stretches
        both."""
        import queue
        import threading

        import torch

        # a recognizer that encodes page batches as they arrive and searches once over all of them (TrOcrProcessor): the detector
        # hands over small batches, nothing is recognized twice, and the beam search runs at the size of the whole call
        streaming = self.stream_recognizer and all(hasattr(icr_processor, a) for a in
                                                   ("recognize_pages_begin", "recognize_pages_add", "recognize_pages_finish"))
        B = max(1, int(self.stream_batch if streaming else self.page_batch))
        overlap = len(frames) > 1 and getattr(box_processor, "ctx", None) is not getattr(icr_processor, "ctx", None)
        head = min(B, max(1, int(self.first_batch))) if overlap and len(frames) > B // 2 else B
        starts = [0] + list(range(head, len(frames), B))
        chunks = [list(range(s, min(len(frames), e))) for s, e in zip(starts, starts[1:] + [len(frames)])]
        overlap = overlap and len(chunks) > 1

        import inspect

        # fragments as views of the frames (read on the device, dropped before this call returns) where the processor offers it
        views = ({"copy_fragments": False}
                 if "copy_fragments" in inspect.signature(box_processor.extract_bounding_boxes_batch).parameters else {})

        def detect(idx):
            return box_processor.extract_bounding_boxes_batch(queue_id, checksum, [frames[i] for i in idx], pms_mode, **views)

        handed = []          # streaming: (page index, detector output) of every page given to the recognizer so far

        def recognize(idx, found):
            pages = [(frames[i], f[0], f[1], f[2]) for i, f in zip(idx, found)]
            if streaming:
                icr_processor.recognize_pages_add(pages)
                handed.extend(zip(idx, found))
                return []
            recs = icr_processor.recognize_pages(queue_id, checksum, pages)
            return [self._finish_page(r, i, f[2], f[4], coordinate_format) for i, f, (r, _) in zip(idx, found, recs)]

        def finish():
            if not streaming:
                return []
            recs = icr_processor.recognize_pages_finish()
            return [self._finish_page(r, i, f[2], f[4], coordinate_format) for (i, f), (r, _) in zip(handed, recs)]

        results: List[Dict] = []
        if streaming:
            icr_processor.recognize_pages_begin(len(frames))
        if not overlap:
            for idx in chunks:
                results.extend(recognize(idx, detect(idx)))
            results.extend(finish())
            return results
        q: "queue.Queue" = queue.Queue(maxsize=2)
        stop = threading.Event()
        # the phase gate (detector forwards of batch k + 1 held back until the recognizer of batch k decodes) is off by default here:
        # measured on 64-page calls it costs 3 % (the detector starts later and still shares the GPU; profiles/r03), where the
        # steady-state loop of bench.py gains 0.6 % from it
        gate = getattr(icr_processor, "decode_gate", None) if getattr(self, "phase_align", False) else None
        targets: "queue.Queue" = queue.Queue()        # consumer -> producer: the gate signal that opens the next detector batch
        # processors without a device (the CPU stand-ins of the plumbing tests) overlap as plain host threads
        on_gpu = torch.cuda.is_available()
        det_stream = torch.cuda.Stream() if on_gpu else None
        det_ctx = getattr(box_processor, "ctx", None) if on_gpu else None

        def hand_over(item) -> bool:
            while not stop.is_set():
                try:
                    q.put(item, timeout=0.1)
                    return True
                except queue.Full:
                    pass
            return False

        def producer():
            try:
                with (torch.cuda.stream(det_stream) if on_gpu else contextlib.nullcontext()):
                    for k, idx in enumerate(chunks):
                        if gate is not None and k > 0:
                            seq = None
                            while seq is None and not stop.is_set():
                                try:
                                    seq = targets.get(timeout=0.1)
                                except queue.Empty:
                                    pass
                            if seq is not None and det_ctx is not None:
                                det_ctx.set_stream(det_stream.cuda_stream)
                                gate.wait(det_ctx, seq, timeout_ms=30000)
                        if stop.is_set() or not hand_over((idx, detect(idx))):
                            return
            except BaseException as e:              # surfaced on the consumer side
                hand_over(e)
            finally:
                # det_stream dies with this call: the detector's context must not keep its handle (a later call sets its own
                # stream again; a destroy that synchronised the dead handle would abort the process)
                if det_ctx is not None:
                    det_stream.synchronize()
                    det_ctx.set_stream(None)

        th = threading.Thread(target=producer, daemon=True)
        th.start()
        try:
            for _ in chunks:
                item = q.get()
                if isinstance(item, BaseException):
                    raise item
                if gate is not None:
                    targets.put(gate.count() + 1)       # the first decode phase of this batch's recognizer call
                results.extend(recognize(*item))
            results.extend(finish())
        finally:
            # whatever ended the loop (a recognizer error included): the producer must not stay blocked on the queue or the
            # gate holding device pages, and its stream must leave the detector's context
            stop.set()
            if gate is not None:
                gate.open(True)
            while th.is_alive():
                try:
                    q.get(timeout=0.05)
                except queue.Empty:
                    pass
            th.join()
            if gate is not None:
                gate.open(False)
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
        """reference: default_ocr_engine.py:32-58 — the recognizer defaults to ``TrOcrProcessor`` on the zoo's checkpoint."""
        super().__init__(models_dir=models_dir, cuda=cuda, box_processor=box_processor, **kwargs)
        if default_ocr_processor is None:
            from .trocr import TrOcrProcessor

            default_ocr_processor = TrOcrProcessor(
                work_dir=self.work_dir_icr, cuda=self.has_cuda,
                model_name_or_path=None if models_dir is None else os.path.join(models_dir, "trocr", "trocr-large-printed.pt"))
        self.ocr_processor = default_ocr_processor

    def extract(self, frames, pms_mode: PSMode = PSMode.SPARSE,
                coordinate_format: CoordinateFormat = CoordinateFormat.XYWH, regions=None, queue_id: str = None,
                **kwargs: Any) -> List[Dict]:
        return self.process_single(self.box_processor, self.ocr_processor, frames, pms_mode, coordinate_format,
                                   regions, queue_id, **kwargs)


class MockOcrEngine(OcrEngine):
    """reference: marie/ocr/mock_ocr_engine.py:17-53 — returns the results a previous run stored under
    ``/tmp/generators/<md5 of the frames>/results/results.json``; raises when there are none.  (The reference's constructor
    drops ``box_processor`` on its way to the base class and so builds a default detector it never uses; this one does not.)"""

    def __init__(self, models_dir: Optional[str] = None, cuda: bool = True, *, box_processor=None, **kwargs) -> None:
        self.has_cuda = cuda
        self.box_processor = box_processor

    def extract(self, frames, pms_mode: PSMode = PSMode.SPARSE,
                coordinate_format: CoordinateFormat = CoordinateFormat.XYWH, regions=None, queue_id: str = None, **kwargs):
        import json

        path = os.path.join("/tmp/generators", hash_frames_fast(frames), "results", "results.json")
        with open(path, "r", encoding="utf-8") as f:
            return json.load(f)


class _MemoBoxProcessor:
    """One detection per (frame, mode) inside a single ``extract`` call: every recognizer of the vote sees the same boxes
    and fragments (the reference re-runs its detector per recognizer on full pages and gets the same answer)."""

    def __init__(self, inner):
        self.inner = inner
        self._memo: Dict[Any, Any] = {}

    def extract_bounding_boxes(self, _id, key, img, psm=PSMode.SPARSE, **kwargs):
        k = (id(img), psm, tuple(sorted(kwargs.items())))
        if k not in self._memo:
            self._memo[k] = (img, self.inner.extract_bounding_boxes(_id, key, img, psm, **kwargs))   # keeps img alive: id stays unique
        return self._memo[k][1]


def vote_words(candidates: List[Dict], min_vote_count: int = 2, by_confidence: Optional[Dict] = None,
               first_greater: bool = False) -> Dict:
    """Pick one word among the recognizers' candidates for the same box (all carry the same ``id``).
    reference: VotingOcrEngine.get_words_by_vote_by_selector / voting_evaluator, marie/ocr/voting_ocr_engine.py:186-254,
    420-466: the largest group of identical texts wins if it has at least ``min_vote_count`` members (equal sizes: larger
    confidence sum; the first such group keeps ties); otherwise the default recognizer's word (the first candidate), unless
    a more confident candidate exists — page mode takes the MOST confident one (:458-466 scans the per-id maxima), region mode
    the FIRST candidate, in recognizer order, that beats the default (:239-247 scans the candidates and breaks)."""
    groups: Dict[str, List[Dict]] = {}
    for w in candidates:
        groups.setdefault(w["text"], []).append(w)
    best: List[Dict] = []
    for g in groups.values():
        if len(g) > len(best) or (len(g) == len(best) and sum(w["confidence"] for w in g) > sum(w["confidence"] for w in best)):
            best = g
    if len(best) >= min_vote_count:
        chosen = best[0]
        votes = deepcopy(best)
        for v in votes:
            v.pop("box", None)
            v.pop("strategy", None)
        chosen["strategy"] = {"type": "voting", "candidates": len(best), "votes": votes}
        return chosen
    chosen = candidates[0]
    chosen["strategy"] = {"type": "default"}
    if first_greater:
        top = next((w for w in candidates if w["confidence"] > chosen["confidence"]), None)
    else:
        top = by_confidence if by_confidence is not None else max(candidates, key=lambda w: w["confidence"])
    if top is not None and top["confidence"] > chosen["confidence"]:
        chosen = top
        chosen["strategy"] = {"type": "confidence", "confidence": top["confidence"]}
    return chosen


def voting_evaluator(aggregated_results: "OrderedDict[str, Any]", default_results, regions=None):
    """reference: VotingOcrEngine.voting_evaluator, marie/ocr/voting_ocr_engine.py:256-482.  Debug JSON dumps and prints
    are dropped.  Region mode: the reference looks results up by ``extended[i]["id"]``, a key its own engine never sets
    (ocr_engine.py:370); here an entry without ``id`` is addressed by its position."""
    has_regions = regions is not None and len(regions) > 0
    if has_regions and len(aggregated_results) == 0:
        out = {} if default_results is None else deepcopy(default_results)
        out["regions"] = []
        for region in regions:
            region.update({"confidence": 0, "text": "", "original_text": "", "words": []})
            out["regions"].append(region)
        return out
    by_unit: Dict[Any, Dict[str, List[Dict]]] = {}
    for name, res in aggregated_results.items():
        units = res["extended"] if has_regions else res
        for idx, unit in enumerate(units):
            uid = unit.get("id", idx) if has_regions else idx
            slot = by_unit.setdefault(uid, {})
            for word in unit.get("words", []):
                word["id"] = str(word["id"]) if not has_regions else word["id"]
                word["processor"] = name
                slot.setdefault(word["id"], []).append(word)
    voted: Dict[Any, List[Dict]] = {}
    for uid, words_by_id in by_unit.items():
        voted[uid] = []
        for wid, cands in words_by_id.items():
            top = cands[0]
            for c in cands:                     # first strictly-larger confidence wins, as in the reference's scan
                if c["confidence"] > top["confidence"]:
                    top = c
            voted[uid].append(vote_words(cands, 2, top, first_greater=has_regions))
    out = deepcopy(default_results)
    if not has_regions:
        for idx, page in enumerate(out):
            page["words"] = voted[idx]
        return out
    for idx, ext in enumerate(out["extended"]):
        uid = ext.get("id", idx)
        if uid in voted:
            ext["words"] = voted[uid]
    for region in out["regions"]:
        rid = region["id"] = str(region["id"])
        for idx, ext in enumerate(out["extended"]):
            if str(ext.get("id", "")) == rid:
                ext["words"].sort(key=lambda w: w["word_index"])
                conf = sum(w["confidence"] for w in ext["words"]) / len(ext["words"]) if ext["words"] else 0
                region["original_text"] = region["text"]
                region["text"] = " ".join(w["text"] for w in ext["words"])
                region["confidence"] = round(conf, 4)
    return out


class MarieHipVotingOcrEngine(OcrEngine):
    """Counterpart of ``VotingOcrEngine`` (marie/ocr/voting_ocr_engine.py:22-482): every enabled recognizer reads the same
    boxes, then the words are voted on.  ``processors``: ordered mapping name -> recognizer; the first one is the default
    (the reference wires TrOcrProcessor as "default" and CraftOcrProcessor as "craft")."""

    def __init__(self, models_dir: Optional[str] = None, cuda: bool = True, *, box_processor=None,
                 default_ocr_processor=None, processors=None, **kwargs) -> None:
        super().__init__(models_dir=models_dir, cuda=cuda, box_processor=box_processor, **kwargs)
        from collections import OrderedDict

        self.processors = OrderedDict()
        if default_ocr_processor is None:
            # voting_ocr_engine.py:49-64: "default" = TrOcrProcessor, "craft" = CraftOcrProcessor on the zoo's checkpoints
            from .trocr import TrOcrProcessor

            default_ocr_processor = TrOcrProcessor(
                work_dir=self.work_dir_icr, cuda=self.has_cuda,
                model_name_or_path=None if models_dir is None else os.path.join(models_dir, "trocr", "trocr-large-printed.pt"))
            if processors is None:
                from .icr import CraftOcrProcessor

                processors = {"craft": CraftOcrProcessor(
                    work_dir=self.work_dir_icr, cuda=self.has_cuda,
                    models_dir=None if models_dir is None else os.path.join(models_dir, "icr"))}
        self.processors["default"] = {"enabled": True, "default": True, "processor": default_ocr_processor}
        for name, proc in (processors or {}).items():
            self.processors[name] = proc if isinstance(proc, dict) else {"enabled": True, "processor": proc}
        first = next(iter(self.processors.values()))
        first.setdefault("default", True)

    def extract(self, frames, pms_mode: PSMode = PSMode.SPARSE,
                coordinate_format: CoordinateFormat = CoordinateFormat.XYXY, regions=None, queue_id: str = None,
                **kwargs: Any):
        from collections import OrderedDict

        global bbox_cache
        ro_frames = OcrEngine.as_frames(frames)
        memo = _MemoBoxProcessor(self.box_processor)
        aggregated, default_results, is_default = OrderedDict(), None, False
        for name, val in self.processors.items():
            if not val.get("enabled", True):
                continue
            is_default = is_default or bool(val.get("default"))
            try:
                results = self.process_single(memo, val["processor"], ro_frames, pms_mode, coordinate_format, regions,
                                              queue_id, **kwargs)
            except Exception:                    # a failing recognizer drops out of the vote, as in the reference
                import traceback

                traceback.print_exc()
                continue
            finally:
                bbox_cache = {}
            aggregated[name] = results
            if is_default:                      # the reference's flag stays set once the default recognizer was seen
                default_results = results
        return voting_evaluator(aggregated, default_results, regions)
