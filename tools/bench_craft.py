#!/usr/bin/env python3
"""Time the CRAFT detector on full-size synthetic pages (per-kernel device time from HIP events + wall).
usage (GPU box): python tools/bench_craft.py [--pages 4] [--h 3300 --w 2550] [--precision f16]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from marie_icr_amd._lib import PREC_F16, PREC_F32, Context  # noqa: E402
from marie_icr_amd.craft import CraftModel  # noqa: E402
from marie_icr_amd.weights import make_craft_bench_state, make_page_bgr  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pages", type=int, default=4)
    ap.add_argument("--h", type=int, default=3300)
    ap.add_argument("--w", type=int, default=2550)
    ap.add_argument("--precision", default="f16")
    a = ap.parse_args()
    ctx = Context(0)
    m = CraftModel(ctx, make_craft_bench_state(), precision=PREC_F16 if a.precision == "f16" else PREC_F32)
    pages = [make_page_bgr(1000 + i, a.h, a.w, n_lines=40) for i in range(min(a.pages, 2))]
    g = m.geometry(a.h, a.w, a.w)
    print("geometry", g)
    boxes, scores, ratio = m.detect_host(pages[0], 0.7, 0.45, 0.3, want_scores=False)   # warm-up (allocations)
    print("warm-up boxes:", len(boxes))
    ctx.profile_reset()
    ctx.profile_enable(True)
    t0 = time.perf_counter()
    nb = 0
    for i in range(a.pages):
        boxes, _, _ = m.detect_host(pages[i % len(pages)], 0.7, 0.45, 0.3, want_scores=False)
        nb += len(boxes)
    dt = time.perf_counter() - t0
    prof = ctx.profile_read()
    ctx.profile_enable(False)
    print(f"{a.pages} pages in {dt * 1e3:.1f} ms -> {a.pages / dt:.2f} pages/s (host-inclusive, H2D of the page included); "
          f"boxes/page {nb / a.pages:.0f}")
    for k, v in prof.items():
        if v["launches"]:
            print(f"  {k:12s} {v['total_ms'] / a.pages:8.3f} ms/page  ({v['launches'] / a.pages:.0f} launches/page)")
    flops = 3.58e12 * (g["H"] * g["W"]) / (1984 * 2560)
    print(f"  conv_igemm algorithmic ~{flops / 1e12:.2f} TFLOP/page -> "
          f"{flops / (prof['conv_igemm']['total_ms'] / a.pages * 1e-3) / 1e12:.0f} TFLOP/s")


if __name__ == "__main__":
    main()
