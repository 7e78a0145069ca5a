"""Text fragments that remember where they live in HBM.

The reference hands fragments from the box processor to the recognizer as a list of numpy arrays cut from the page
(marie/boxes/dit/ulim_dit_box_processor.py:795-800, marie/document/ocr_processor.py:156).  Here the page is already on the
device when the detector has run, so the same list can carry, per fragment, the window of the device page it was cut from:
the recognizer then reads the pixels where they are (one ``mhip_crop_desc`` each) instead of packing the arrays and copying
them to the device again.  To every other consumer a ``FragmentList`` is a plain list of HxWx3 uint8 arrays.
"""
from __future__ import annotations

from typing import Iterable, List, Optional, Sequence, Tuple


class FragmentList(list):
    """``windows[i]`` = (device address of the first pixel, rows, columns, row stride in bytes, channels) of fragment i, or
    ``None`` for the whole list when any fragment has no device copy.  ``keepalive`` holds the device tensors."""

    def __init__(self, items: Iterable = (), windows: Optional[List[Tuple[int, int, int, int, int]]] = None, keepalive=()):
        super().__init__(items)
        self.windows = windows if windows is not None and len(windows) == len(self) else None
        self.keepalive = list(keepalive)

    @staticmethod
    def concat(lists: Sequence[Sequence]) -> "FragmentList":
        items, windows, keep, ok = [], [], [], True
        for l in lists:
            items.extend(l)
            w = getattr(l, "windows", None)
            if w is None and len(l):
                ok = False
            elif w is not None:
                windows.extend(w)
                keep.extend(getattr(l, "keepalive", ()))
        return FragmentList(items, windows if ok else None, keep)

    def device_descs(self, start: int = 0, stop: Optional[int] = None):
        """(base address, ctypes array of CropDesc relative to it) for fragments [start, stop); None without device windows."""
        from ._lib import CropDesc

        if self.windows is None:
            return None
        win = self.windows[start:stop]
        if not win:
            return None
        base = min(w[0] for w in win)
        descs = (CropDesc * len(win))()
        for i, (addr, h, w, stride, ch) in enumerate(win):
            descs[i] = CropDesc(addr - base, h, w, stride, ch)
        return base, descs
