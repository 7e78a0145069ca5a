#!/usr/bin/env python3
"""Time the DiT backbone (ViT encoder + fpn1..4) on page-sized inputs: per-kernel device time from HIP events.
usage (GPU box): python tools/bench_vit.py [--batch 4] [--iters 5] [--model base] [--h 1035 --w 800]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from marie_icr_amd._lib import PREC_F16, Context  # noqa: E402
from marie_icr_amd.vit import VitModel, dit_config  # noqa: E402
from marie_icr_amd.weights import make_vit_state  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--model", default="base")
    ap.add_argument("--h", type=int, default=1035)
    ap.add_argument("--w", type=int, default=800)
    a = ap.parse_args()
    ctx = Context(0)
    cfg = dit_config(a.model)
    m = VitModel(ctx, cfg, make_vit_state(0, cfg.dim, cfg.depth, cfg.heads), PREC_F16)
    H32, W32 = (a.h + 31) // 32 * 32, (a.w + 31) // 32 * 32
    imgs = np.random.default_rng(0).integers(0, 256, size=(a.batch, a.h, a.w, 3), dtype=np.uint8)
    m.forward_host(imgs, (H32, W32), want_fpn=False)
    ctx.profile_reset()
    ctx.profile_enable(True)
    t0 = time.perf_counter()
    for _ in range(a.iters):
        m.forward_host(imgs, (H32, W32), want_fpn=False)
    dt = time.perf_counter() - t0
    prof = ctx.profile_read()
    ctx.profile_enable(False)
    n = a.iters * a.batch
    print(f"{n} pages in {dt * 1e3:.1f} ms -> {n / dt:.1f} pages/s backbone only (canvas {H32}x{W32}, batch {a.batch})")
    for k, v in prof.items():
        if v["launches"]:
            tf = f"  {v['flops'] / (v['total_ms'] * 1e-3) / 1e12:7.1f} TFLOP/s" if v["flops"] else ""
            print(f"  {k:12s} {v['total_ms'] / n:8.3f} ms/page  ({v['launches'] / n:.1f} launches/page){tf}")


if __name__ == "__main__":
    main()
