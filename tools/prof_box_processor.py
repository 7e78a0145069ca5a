"""Where the wall time of BoxProcessorUlimDit.extract_bounding_boxes_batch goes on N full-size host pages (nothing else on the
GPU): upload, detector forwards, host post-processing.  python tools/prof_box_processor.py [pages]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch

    import marie_icr_amd.dit_box_processor as dbp
    from marie_icr_amd._lib import Context, PREC_F16
    from marie_icr_amd.box_processor import PSMode
    from marie_icr_amd.dit import DitModel
    from marie_icr_amd.weights import make_dit_state, make_page_bgr, page_line_boxes

    n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    ctx = Context(0)
    det = DitModel(ctx, make_dit_state(0, "base"), model="base", precision=PREC_F16)
    pool = [make_page_bgr(1000 + i, 3300, 2550, n_lines=40) for i in range(8)]
    frames = [pool[i % 8] for i in range(n)]
    gt = page_line_boxes(3300, 2550, 40)
    gt_xyxy = np.stack([gt[:, 0], gt[:, 1], gt[:, 0] + gt[:, 2] + 1, gt[:, 1] + gt[:, 3] + 1], 1).astype(np.float32)
    spent = {}

    def wrap(obj, name, key):
        fn = getattr(obj, name)

        def w(*a, **k):
            t = time.perf_counter()
            try:
                return fn(*a, **k)
            finally:
                spent[key] = spent.get(key, 0.0) + time.perf_counter() - t
        setattr(obj, name, w)

    for fixed in (True, False):
        bp = dbp.BoxProcessorUlimDit(cuda=True, refinement=False, dit_model=det, det_batch=8)
        real = bp._detect_batch
        if fixed:
            def det_fixed(page_devs, shape, real=real):
                real(page_devs, shape)
                return [(gt_xyxy.copy(), np.ones(len(gt_xyxy), np.float32)) for _ in page_devs]
            bp._detect_batch = det_fixed
        wrap(bp, "_detect_batch", "detector forwards (incl. waiting for the pages)")
        wrap(bp, "_post_step", "post_step (merge_boxes)")
        wrap(bp, "psm_sparse_batch", "psm_sparse_batch (all of the above + uploads + lines)")
        wrap(dbp, "lines_from_bboxes", "lines_from_bboxes")
        wrap(dbp, "find_line_numbers", "find_line_numbers")
        bp.extract_bounding_boxes_batch("q", "k", frames[:8], PSMode.SPARSE)      # warm-up
        for rep in range(2):
            spent.clear()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            res = bp.extract_bounding_boxes_batch("q", "k", frames, PSMode.SPARSE)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        words = sum(len(r[0]) for r in res) / len(res)
        print(f"{'fixed 40 lines' if fixed else 'detector boxes'}: {n} pages {dt * 1e3:.1f} ms = {dt / n * 1e3:.2f} ms/page, {words:.0f} boxes/page")
        for k, v in sorted(spent.items(), key=lambda x: -x[1]):
            print(f"    {v * 1e3:8.1f} ms  {k}")
        dbp.lines_from_bboxes = dbp.lines_from_bboxes.__closure__[0].cell_contents if False else dbp.lines_from_bboxes
    # the upload alone
    t0 = time.perf_counter()
    devs = [torch.from_numpy(f).cuda() for f in frames]
    torch.cuda.synchronize()
    print(f"upload alone, one after the other: {(time.perf_counter() - t0) * 1e3:.1f} ms for {n} pages")
    pin = [torch.from_numpy(f).pin_memory() for f in frames[:8]]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    devs = [p.cuda(non_blocking=True) for p in pin]
    torch.cuda.synchronize()
    print(f"8 pinned pages: {(time.perf_counter() - t0) * 1e3:.1f} ms")
    t0 = time.perf_counter()
    det.detect_device([d.data_ptr() for d in devs], 3300, 2550)
    torch.cuda.synchronize()
    print(f"one detector forward of 8 resident pages: {(time.perf_counter() - t0) * 1e3:.1f} ms")


if __name__ == "__main__":
    main()
