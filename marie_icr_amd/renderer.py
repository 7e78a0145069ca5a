"""The step after the path: what consumes the engine's page results.

reference: ``ResultRenderer`` (marie/renderer/renderer.py:11-63), ``TextRenderer`` (marie/renderer/text_renderer.py:12-175)
and ``get_words_and_boxes`` (marie/ocr/util.py:17-41).  Host-side text work on the result dictionaries the engine returns —
no GPU involved; same class and method names, arguments and output bytes as the reference (goldens written by the
reference's own ``TextRenderer``: tests/golden/text_renderer.json).  The reference's progress prints are not reproduced.

``BlobRenderer`` (marie/renderer/blob_renderer.py:13-90: one ``<n>.BLOBS.XML`` per page) and ``AdlibRenderer``
(marie/renderer/adlib_renderer.py:14-161: one ``<n>.tif.xml`` per page + ``summary.xml``) are the two XML on-disk formats; their
files are byte-equal to what the reference classes write (tests/golden/xml_renderers.json; the summary's ``CreationDate`` is the
time of the call in both).  ``PdfRenderer`` needs reportlab and PyPDF4, which are not installed here: absent.
"""
from __future__ import annotations

import io
import logging
from abc import ABC, abstractmethod
from math import ceil
from os import PathLike
from typing import Any, Callable, Dict, List, Optional, Union

import numpy as np


def get_words_and_boxes(ocr_results, page_index: int, include_lines: bool = False):
    """marie/ocr/util.py:17-41: ``(words, boxes)`` or ``(words, boxes, lines)`` of one page of results."""
    words, boxes, lines = [], [], []
    if not ocr_results:
        return words, boxes
    if page_index >= len(ocr_results):
        raise ValueError(f"Page index {page_index} is out of range.")
    for w in ocr_results[page_index]["words"]:
        boxes.append(w["box"])
        words.append(w["text"])
        lines.append(w["line"])
    if include_lines:
        return words, boxes, lines
    return words, boxes


def _strtobool(val) -> bool:
    if isinstance(val, bool):
        return val
    v = str(val).lower()
    if v in ("y", "yes", "t", "true", "on", "1"):
        return True
    if v in ("n", "no", "f", "false", "off", "0"):
        return False
    raise ValueError(f"invalid truth value {val!r}")


class ResultRenderer(ABC):
    """marie/renderer/renderer.py:11-63."""

    def __init__(self, config=None):
        self.config = {} if config is None else config
        self.logger = logging.getLogger(ResultRenderer.__name__)

    @property
    @abstractmethod
    def name(self) -> str:
        ...

    @abstractmethod
    def render(self, frames, results: List[Dict[str, Any]], output_file_or_dir: Union[str, PathLike, io.BytesIO],
               filename_generator: Optional[Callable[[int], str]] = None, **kwargs: Any) -> None:
        ...

    def check_format_xywh(self, result, convert=True):
        """renderer.py:47-63 — note what it does, not what it is called: when ``meta["format"]`` is set and is not
        ``"xywh"`` every word box ``[x, y, w, h]`` is rewritten to ``[x, y, x + w, y + h]``, in place."""
        meta = result["meta"]
        if convert and "format" in meta and meta["format"] != "xywh":
            for word in result["words"]:
                x, y, w, h = word["box"]
                word["box"] = [x, y, x + w, y + h]


class TextRenderer(ResultRenderer):
    """marie/renderer/text_renderer.py:12-175: words dropped onto a fixed-pitch character grid (8.44 x 16 px cells)."""

    def __init__(self, config=None):
        super().__init__(config)
        self.preserve_interword_spaces = False
        if "preserve_interword_spaces" in self.config:
            self.preserve_interword_spaces = _strtobool(self.config["preserve_interword_spaces"])

    @property
    def name(self):
        return "TextRenderer"

    def _render_page(self, image: np.ndarray, result: Dict[str, Any], page_index: int) -> str:
        if image is None:
            raise Exception("Image or list of images expected")
        self.check_format_xywh(result, True)
        h, w = image.shape[0], image.shape[1]
        char_width, char_height = 8.44, 16
        cols = ceil(w // char_width)
        x_space = np.arange(0, w, 1)
        bins = np.array(np.linspace(0, w, cols)).astype(np.int32)
        x_hist = np.digitize(x_space, bins, right=True)             # pixel column -> character column
        words, lines = result["words"], result["lines"]
        buffer = ""
        start_cell_y = 1
        max_characters_per_line = ceil(w // char_width)
        for i, line in enumerate(lines):
            wordids = line["wordids"]
            _, y, _, lh = line["bbox"]
            cell_y = (y + lh) // char_height                        # the baseline's character row
            delta_cell_y = cell_y - start_cell_y
            start_cell_y = cell_y
            for _ in range(1, delta_cell_y):
                buffer += "\n"
            aligned = [wd for wd in words if wd["id"] in wordids]
            order = np.argsort(np.array([wd["word_index"] for wd in aligned]), kind="stable") if aligned else []
            line_buffer = " " * max_characters_per_line
            for k in order:
                word = aligned[int(k)]
                grid_space = x_hist[word["box"][0]]                 # raises for a box starting outside the page, as the reference
                line_buffer = line_buffer[:grid_space] + word["text"] + line_buffer[grid_space:]
            buffer += line_buffer
            if i < len(lines) - 1:
                buffer += "\n"
        return buffer

    def render(self, frames, results: List[Dict[str, Any]], output_file_or_dir: Union[str, PathLike, io.BytesIO],
               filename_generator: Optional[Callable[[int], str]] = None, **kwargs: Any) -> None:
        """Pages separated by a form feed; a page whose rendering raises is logged and left out, as in the reference.
        ``results`` in a format other than xywh are converted in place (``check_format_xywh``)."""
        buffer = ""
        for page_index, (image, result) in enumerate(zip(frames, results)):
            try:
                buffer += self._render_page(image, result, page_index)
            except Exception as e:  # noqa: BLE001 - the reference swallows per-page failures
                self.logger.error(e, exc_info=True)
            if len(frames) > 1 and page_index < len(frames) - 1:
                buffer += "\f"
        if isinstance(output_file_or_dir, io.IOBase):
            data = buffer.encode("UTF-8") if isinstance(output_file_or_dir, (io.BytesIO, io.BufferedIOBase)) else buffer
            output_file_or_dir.write(data)
            return
        with open(output_file_or_dir, "w", encoding="UTF-8") as text_file:
            text_file.write(buffer)


class BlobRenderer(ResultRenderer):
    """reference: marie/renderer/blob_renderer.py:13-90."""

    def __init__(self, config=None):
        super().__init__(config)

    @property
    def name(self):
        return "BlobRenderer"

    def _render_page(self, image: np.ndarray, result: Dict[str, Any], page_index: int):
        """blob_renderer.py:22-58: a ``blobs`` root (300 dpi, angle 0) with one ``blob`` per word; the page is 1-based."""
        import xml.etree.ElementTree as gfg
        from xml.sax.saxutils import escape

        root = gfg.Element("blobs")
        root.set("angle", "0.0")
        root.set("yres", "300")
        root.set("xres", "300")
        root.set("page", str(page_index))
        result["meta"], result["lines"]                   # the reference reads both keys first (KeyError when one is missing)
        for word in result["words"]:
            x, y, w, h = word["box"]
            m1 = gfg.Element("blob")
            m1.set("x", str(x))
            m1.set("y", str(y))
            m1.set("w", str(w))
            m1.set("h", str(h))
            m1.set("text", escape(word["text"]))          # escaped here AND by ElementTree on output, as in the reference
            b1 = gfg.SubElement(m1, "page")
            b1.text = str(page_index + 1)
            root.append(m1)
        return gfg.ElementTree(root)

    def render(self, frames, results: List[Dict[str, Any]], output_path: Union[str, PathLike],
               filename_generator: Optional[Callable[[int], str]] = None) -> None:
        """blob_renderer.py:60-90: ``output_path`` is a directory; a page that fails is logged and skipped."""
        import os

        if not os.path.isdir(output_path):
            raise ValueError("output_path should be a directory")
        filename_generator = filename_generator or (lambda x: f"{x}.BLOBS.XML")
        for page_index, (image, result) in enumerate(zip(frames, results)):
            try:
                tree = self._render_page(image, result, page_index)
                with open(os.path.join(output_path, filename_generator(page_index + 1)), "wb") as fs:
                    tree.write(fs)
            except Exception as e:
                self.logger.error(e, stack_info=True, exc_info=True)


class AdlibRenderer(ResultRenderer):
    """reference: marie/renderer/adlib_renderer.py:14-161."""

    def __init__(self, summary_filename="summary.xml", config=None):
        super().__init__(config)
        self.summary_filename = summary_filename

    @property
    def name(self):
        return "AdlibRenderer"

    def write_adlib_summary_tree(self, frames, filename_generator: Callable[[int], str]):
        """adlib_renderer.py:30-63."""
        import xml.etree.ElementTree as gfg
        from datetime import datetime

        def _meta(field, val):
            meta = gfg.Element("METADATAELEMENT")
            meta.set("FIELD", str(field))
            meta.set("VALUE", str(val))
            return meta

        root = gfg.Element("OCR")
        metas = gfg.Element("METADATAELEMENTS")
        metas.append(_meta("OCR", "MARIE-AI"))
        metas.append(_meta("CreationDate", datetime.now().strftime("%Y-%m-%d %H:%M:%S")))
        root.append(metas)
        pages_node = gfg.Element("PAGES")
        for page_index, _path in enumerate(frames):
            node = gfg.Element("PAGE")
            node.set("Filename", filename_generator(page_index + 1))
            node.set("NUMBER", str(page_index + 1))
            pages_node.append(node)
        root.append(pages_node)
        return gfg.ElementTree(root)

    def _render_page(self, image: np.ndarray, result: Dict[str, Any], page_index: int):
        """adlib_renderer.py:65-126: inches at 300 dpi; TOP = y - h and BOTTOM = y + h as the reference computes them."""
        import xml.etree.ElementTree as gfg

        meta, words = result["meta"], result["words"]
        result["lines"]
        dpi_x = dpi_y = 300.0
        pagenumber = page_index + 1
        root = gfg.Element("PAGE")
        root.set("HEIGHT", str(meta["imageSize"]["height"] / dpi_y))
        root.set("WIDTH", str(meta["imageSize"]["width"] / dpi_x))
        root.set("ImageType", "Unknown")
        root.set("NUMBER", str(pagenumber))
        root.set("OCREndTime", "0")
        root.set("OCRStartTime", "0")
        root.set("Producer", "marie")
        root.set("XRESOLUTION", str(dpi_x))
        root.set("YRESOLUTION", str(dpi_y))
        root.append(gfg.Element("TEXT"))
        for word in words:
            x1, y1, w1, h1 = word["box"]
            x, y, w, h = x1 / dpi_x, y1 / dpi_y, w1 / dpi_x, h1 / dpi_y
            m1 = gfg.Element("TEXTSTRING")
            m1.set("CONSECUTIVE", "FALSE")
            m1.set("FONTNAME", "Courier")
            m1.set("FONTSIZE", "32")
            m1.set("NoLocation", "FALSE")
            m1.set("PageNumber", str(pagenumber))
            m1.set("LEFT", f"{x:.4f}")
            m1.set("RIGHT", f"{x + w:.4f}")
            m1.set("TOP", f"{y - h:.4f}")
            m1.set("BOTTOM", f"{y + h:.4f}")
            m1.set("WORD", str(word["text"]))
            root.append(m1)
        return gfg.ElementTree(root)

    def render(self, frames, results: List[Dict[str, Any]], output_file_or_dir: Union[str, PathLike],
               filename_generator: Optional[Callable[[int], str]] = None, **kwargs: Any) -> None:
        """adlib_renderer.py:128-161: page files, then the summary."""
        import os

        if not os.path.isdir(output_file_or_dir):
            raise ValueError("output_file_or_dir should be a directory")
        filename_generator = filename_generator or (lambda x: f"{x}.tif.xml")
        for page_index, (image, result) in enumerate(zip(frames, results)):
            try:
                tree = self._render_page(image, result, page_index)
                with open(os.path.join(output_file_or_dir, filename_generator(page_index + 1)), "wb") as fs:
                    tree.write(fs)
            except Exception as e:
                self.logger.error(e, stack_info=True, exc_info=True)
        tree = self.write_adlib_summary_tree(frames, filename_generator)
        with open(os.path.join(output_file_or_dir, self.summary_filename), "wb") as ws:
            tree.write(ws)
