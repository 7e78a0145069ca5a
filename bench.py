#!/usr/bin/env python3
"""bench.py — throughput of the MI355X OCR hot path on synthetic data.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Default workload ``dit_trocr`` = BASELINE configs[2], the configuration the pages/sec metric is quoted on: DiT-base
Mask R-CNN detector + TrOCR-base recognizer on 2550x3300 pages resident in HBM.  One *step* = ``--pages`` (64) pages per GPU:
    Pillow-exact bilinear resize to 1035x800 -> DiT-base backbone + FPN + RPN + ROI heads + FastRCNN inference
    (one detector pass: bbox_refinement=False) -> boxes to the host;
    40 ground-truth line boxes per page (fixed recognizer work, SURVEY.md section 8d "headline") -> Pillow-exact bicubic to
    384x384 -> TrOCR-base encoder -> 12-layer decoder, beam 3, ``--decode-len`` + 1 steps (15 + EOS) -> token ids.
Seeded random weights; with random weights the detector's own boxes are noise, which is why the recognizer consumes the
generator's line boxes (the detector still runs in full and is timed).

Workload ``craft_crnn`` (BASELINE.json metric: pages/sec on 2550x3300 pages, ~40 lines/page): one *step* = one
pass of detect -> crop -> recognize over a batch of ``--pages`` synthetic 2550x3300x3 uint8 pages per GPU that are
already resident in HBM:
    CRAFT detector (cv2-exact resize to the 1970x2550 canvas, normalise, VGG16-BN + U-net, score maps) ->
    GPU threshold + connected components + statistics -> host box finalisation (minAreaRect etc.) ->
    crop batcher (Pillow-exact bicubic to 32 x 256) -> CRNN recognizer (None-VGG-BiLSTM-CTC) -> greedy CTC decode ->
    token ids / confidences to pinned host memory -> strings.
Fixed recognizer work: the detector runs and is timed in full, the recognizer consumes the page
generator's 40 ground-truth line boxes per page (``--crops detector`` feeds the detector's own boxes instead; with
random weights their number is arbitrary, so that mode is the secondary number).  Seeded random weights.

``--workload crnn`` = BASELINE configs[1]: the CRNN recognizer alone on 1024 pre-cropped 32x256 lines per GPU.

Multi-GPU: pages are independent, so each rank owns its own pages — weak scaling, no data-path
collective.  ``python bench.py --gpus N`` (N > 1) with no launcher around it starts the N ranks itself (a child
``torch.distributed.run`` before anything touches the GPU); under a launcher WORLD_SIZE must equal ``--gpus`` or the run
exits non-zero.  Rank 0 packs the weights and the packed arenas are broadcast once over RCCL/xGMI at start-up; the timed
region is bracketed by barrier + synchronize and the maximum over ranks is reported.

Secondary legs of the default workload, all in the same JSON line (never ``value``):
  ``stream``      BASELINE configs[3] — a fixed stream (2048 pages when N > 1) sharded round-robin over the ranks
                  (``marie_icr_amd.dist.shard_indices``), boxes-per-page and token ids gathered in page order
                  (``gather_in_order``); strong scaling; ``result_checksum`` is the same for every N.
  ``mixed_dpi``   BASELINE configs[4] — 150/200/300-DPI pages interleaved, chunks pulled from a host work queue, pages
                  bucketed by size, crops pooled.
  ``engine_api``  the same models behind ``MarieHipOcrEngine.extract`` (host frames in, dictionaries out).
  ``det_passes_3``, ``pcie_inclusive``, ``parity`` (GPU vs oracle on the cpu_baseline sample).

Rank 0 prints ONE JSON line.  ``roofline`` covers the dominant kernel (conv_igemm, MFMA-bound): algorithmic FLOPs of all
its launches in a step (2*MAC, real channel counts, accumulated by the launcher: mhip_profile_flops) divided by their
summed device time, measured live with HIP events on the launch streams in one more step run exactly like the timed ones
(``frac``: in situ, two streams) and once with the detector and the recognizer alone (``isolated``); ``by_tile`` splits
both by tile shape.  ``cpu_baseline`` times the CPU oracle (torch CPU ops — what the reference executes on a CPU host —
plus the reference's numpy post-processing restated) on rank 0's host cores on a bounded sample of the same workload.
"""
from __future__ import annotations

import argparse
import functools
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

LINES_PER_PAGE = 40          # BASELINE.json metric: "~40 lines/page"
PAGE_H, PAGE_W = 3300, 2550  # 300-dpi letter page
PEAK_MFMA_TFLOPS_F16 = 2500  # dense f16/bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_MFMA_TFLOPS_F32 = 157.3
THRESH = (0.7, 0.45, 0.3)    # psm_sparse: text_threshold, link_threshold, low_text


def host_cores():
    try:
        c = len(os.sched_getaffinity(0))
    except AttributeError:
        c = os.cpu_count() or 1
    return max(1, min(c, 64))


def cpu_baseline_crnn(state, charset, img_w, target_s=12.0):
    """Time the CPU oracle of the recognizer on a bounded sample (rank 0, N=1 only)."""
    from marie_icr_amd.weights import make_crnn_input
    from oracle import crnn_numpy
    from oracle.crnn_torch import TorchCrnnOracle

    cores = host_cores()
    o = TorchCrnnOracle(state, threads=cores)
    probe = make_crnn_input(1, 64, 32, img_w)
    x = crnn_numpy.normalize_u8(probe)
    o.decode(o.logits(x), charset)                       # warm-up
    t0 = time.perf_counter()
    o.decode(o.logits(x), charset)
    per_line = (time.perf_counter() - t0) / 64
    n = int(min(1024, max(64, (target_s / max(per_line, 1e-6)) // 64 * 64)))
    crops = make_crnn_input(2, n, 32, img_w)
    t0 = time.perf_counter()
    done = 0
    for s in range(0, n, 128):                           # the reference CPU path batches too
        xb = crnn_numpy.normalize_u8(crops[s:s + 128])
        o.decode(o.logits(xb), charset)
        done += xb.shape[0]
    dt = time.perf_counter() - t0
    return {"value": done / dt, "unit": "lines/s", "cores": cores, "kind": "port",
            "sample": f"{done} of the same seeded 32x{img_w} lines in batches of 128, fp32, "
                      f"oracle/crnn_torch.py (torch CPU ops), {dt:.1f} s"}


def cpu_baseline_pages(craft_state, crnn_state, charset, img_w, n_lines):
    """One full page through the CPU oracle pipeline (rank 0, N=1 only): torch-CPU CRAFT forward, the reference's
    numpy post-processing restated, Pillow crop batcher, torch-CPU CRNN.  ~15-30 s of CPU work."""
    import torch

    from marie_icr_amd.weights import make_page_bgr, page_line_boxes
    from oracle import craft_ref, crnn_numpy
    from oracle import pil_resample as pr
    from oracle.crnn_torch import TorchCrnnOracle

    cores = host_cores()
    torch.set_num_threads(cores)
    page = make_page_bgr(999, PAGE_H, PAGE_W, n_lines=n_lines)
    lines = page_line_boxes(PAGE_H, PAGE_W, n_lines)
    rec = TorchCrnnOracle(crnn_state, threads=cores)
    t0 = time.perf_counter()
    rects, _ = craft_ref.detect_page(page, craft_state, *THRESH)
    t1 = time.perf_counter()
    frags = [page[y:y + h + 1, x:x + w + 1] for x, y, w, h in lines.tolist()]
    crops = pr.align_collate_pil(frags, img_w)
    rec.decode(rec.logits(crnn_numpy.normalize_u8(crops)), charset)
    t2 = time.perf_counter()
    return {"value": 1.0 / (t2 - t0), "unit": "pages/s", "cores": cores, "kind": "port",
            "sample": f"1 of the same seeded {PAGE_W}x{PAGE_H} pages, fp32: detector {t1 - t0:.1f} s "
                      f"({len(rects)} boxes; torch CPU forward + the reference's per-component numpy loop), "
                      f"{n_lines} line crops + recognizer {t2 - t1:.2f} s"}

def host_cores_physical():
    """(physical cores the process may use, logical CPUs it may use): lscpu -p lists one row per logical CPU with its core
    and socket; unique (core, socket) pairs among the CPUs in our affinity mask are physical cores."""
    import subprocess

    try:
        aff = os.sched_getaffinity(0)
    except AttributeError:
        aff = set(range(os.cpu_count() or 1))
    try:
        rows = subprocess.run(["lscpu", "-p=cpu,core,socket"], capture_output=True, text=True, timeout=10).stdout.splitlines()
        cores = {tuple(r.split(",")[1:3]) for r in rows if r and not r.startswith("#") and int(r.split(",")[0]) in aff}
        phys = len(cores) or len(aff)
    except Exception:
        phys = len(aff)
    return max(1, phys), max(1, len(aff))


def _edit_rate(refs, hyps) -> float:
    """sum of Levenshtein distances / sum of reference lengths (the reference's char-error definition)."""
    dist = total = 0
    for r, h in zip(refs, hyps):
        prev = list(range(len(h) + 1))
        for i, a in enumerate(r, 1):
            cur = [i]
            for j, b in enumerate(h, 1):
                cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (a != b)))
            prev = cur
        dist += prev[-1]
        total += len(r)
    return dist / max(1, total)


def _iou_match(ref, got, bar):
    if len(ref) == 0 or len(got) == 0:
        return 0.0
    a, b = np.asarray(ref, np.float64), np.asarray(got, np.float64)
    x1 = np.maximum(a[:, None, 0], b[None, :, 0]); y1 = np.maximum(a[:, None, 1], b[None, :, 1])
    x2 = np.minimum(a[:, None, 2], b[None, :, 2]); y2 = np.minimum(a[:, None, 3], b[None, :, 3])
    inter = np.clip(x2 - x1, 0, None) * np.clip(y2 - y1, 0, None)
    aa = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1]); ab = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    iou = inter / (aa[:, None] + ab[None, :] - inter + 1e-12)
    return float((iou.max(axis=1) >= bar).mean())


def cpu_baseline_dit_trocr(dit_state, trocr_state, trocr_dims, decode_len, n_lines, det, rec, precision):
    """The CPU oracle pipeline on a bounded sample (rank 0, N=1 only): one full 2550x3300 page through the DiT-base
    detector (one pass) and its n_lines line crops through TrOCR-base, beam 3, in one batch.
    The same page and crops also go through the GPU models that were just timed, and the two outputs are compared
    (``parity``; the assertions live in tests/test_fullsize_gpu.py)."""
    import torch

    from marie_icr_amd.weights import make_page_bgr, page_line_boxes
    from oracle.dit_torch import TorchDitOracle
    from oracle.trocr_torch import TorchTrocrOracle, preprocess_fragments

    phys, logical = host_cores_physical()
    page = make_page_bgr(999, PAGE_H, PAGE_W, n_lines=n_lines)
    lines = page_line_boxes(PAGE_H, PAGE_W, n_lines)
    odet = TorchDitOracle(dit_state)
    # the CPU path gets the thread count it runs fastest with: all physical cores of a two-socket host are slower than 32
    # threads for these GEMM sizes, so a small window of the page is timed at a few counts first (also the warm-up)
    probe = page[:1320, :1020].copy()
    best = (None, 1e30)
    for nt in sorted({c for c in (16, 32, 64, phys) if c <= phys}):
        torch.set_num_threads(nt)
        odet.detect(probe)
        tp = time.perf_counter()
        odet.detect(probe)
        tp = time.perf_counter() - tp
        if tp < best[1]:
            best = (nt, tp)
    threads = best[0]
    torch.set_num_threads(threads)
    t0 = time.perf_counter()
    boxes, scores = odet.detect(page)
    t1 = time.perf_counter()
    enc, dec, vocab = trocr_dims
    orec = TorchTrocrOracle(trocr_state, enc[2], dec[2], beam=3, max_len_b=decode_len)
    k = n_lines                                                  # every line of the page: the parity sample below has N = 40
    frags = [page[y:y + h + 1, x:x + w + 1] for x, y, w, h in lines.tolist()]
    crops = preprocess_fragments(frags)
    t2 = time.perf_counter()
    ref, otr = orec.generate(crops, want_trace=True)
    t3 = time.perf_counter()
    per_page = (t1 - t0) + (t3 - t2)
    base = {"value": 1.0 / per_page, "unit": "pages/s", "cores": threads, "physical_cores": phys, "logical_cpus": logical,
            "kind": "port",
            "sample": f"1 of the same seeded {PAGE_W}x{PAGE_H} pages through the detector oracle ({t1 - t0:.1f} s, "
                      f"{len(boxes)} boxes, torch CPU fp32, {threads} threads — the fastest of 16/32/64/{phys} on a probe window; the host has "
                      f"{phys} physical cores in the affinity mask) + all {k} of "
                      f"its line crops through the TrOCR oracle in one batch ({(t3 - t2) / k:.2f} s/crop, beam 3, "
                      f"{decode_len}+1 steps)"}
    # ---- the GPU path on the same inputs ----
    from oracle import trocr_trace as tt

    (gb, gs), = det.detect_host(page[None])
    got, gtr = rec.generate_trace_host(crops)
    walk = tt.walk(otr, gtr)
    equal = [bool(len(g[0]) == len(r[0]) and np.array_equal(g[0], r[0])) for g, r in zip(got, ref)]
    own = orec.score_tokens(crops, [g[0] for g in got])
    gaps = [float(r[1] - s_) for r, s_ in zip(ref, own)]
    never = [w["diverged_at"] is None for w in walk]
    parity = {"dtype": precision, "page": "seed 999",
              "boxes": {"oracle": int(len(boxes)), "gpu": int(len(gb)),
                        "matched_iou_0.999": _iou_match(boxes, gb, 0.999), "matched_iou_0.99": _iou_match(boxes, gb, 0.99),
                        "matched_iou_0.9": _iou_match(boxes, gb, 0.9), "matched_iou_0.5": _iou_match(boxes, gb, 0.5)},
              "trocr": {"crops": k, "hypotheses_token_equal": int(sum(equal)),
                        # BASELINE.json's second metric ("char-error vs ref"): Levenshtein distance / reference length; the
                        # seeded models have no real vocabulary, so the symbols compared are the token ids
                        "symbol_error_rate": _edit_rate([list(map(int, r[0])) for r in ref], [list(map(int, g[0])) for g in got]),
                        "symbols": int(sum(len(r[0]) for r in ref)),
                        "max_abs_score_diff_where_equal": max([abs(g[1] - r[1]) for g, r, e in zip(got, ref, equal) if e] or [0.0]),
                        # the two beam searches walked side by side (oracle/trocr_trace.py): a line either takes the oracle's
                        # decisions at every step, or first differs where the oracle's own candidate list has a near-tie
                        # under the score error measured on that line
                        "beam_search_never_diverged": int(sum(never)),
                        "diverged_at_a_proven_near_tie": int(sum((not n) and w["explained"] for n, w in zip(never, walk))),
                        "diverged_unexplained": int(sum((not n) and (not w["explained"]) for n, w in zip(never, walk))),
                        "max_candidate_score_error": float(max(w["eps"] for w in walk)),
                        "max_oracle_best_minus_oracle_score_of_gpu_hypothesis": float(max(gaps)),
                        "min_oracle_best_minus_oracle_score_of_gpu_hypothesis": float(min(gaps))},
              "note": "N = 1 page (its boxes) and its 40 line crops.  f16 operands re-order near-tied discrete choices of a "
                      "random-weight model: every difference is shown to be a near-tie under the measured error by the interval "
                      "analysis (detector) and the candidate-list walk (recognizer) in tests/test_fullsize_gpu.py, which also "
                      "holds the fp32 bars (911 of 911 boxes matched: 900 at IoU >= 0.999, 11 boxes under 16 px by every "
                      "coordinate within 0.004 px; tokens exact on all 40 lines) and the string-exact bar on the decoder with margins"}
    return base, parity


def _pmc_traffic(config):
    """profiles/rNN/pmc_traffic.json (rocprofv3 --pmc passes folded by tools/pmc_traffic.py), newest round first, if it was taken
    on `config`."""
    for rnd in ("r03", "r02", "r01"):
        path = os.path.join(ROOT, "profiles", rnd, "p_pmc_traffic.json" if rnd == "r01" else "pmc_traffic.json")
        try:
            with open(path) as f:
                d = json.load(f)
        except (OSError, ValueError):
            continue
        if d.get("config") == config:
            d["source"] = os.path.relpath(path, ROOT)
            return d
    return None


MIXED_DPI_SIZES = ((1650, 1275), (2200, 1700), (3300, 2550))      # 150 / 200 / 300 DPI letter pages, (h, w)
IGEMM_VARIANTS = ("conv_igemm<64>", "conv_igemm<128>", "conv_igemm<256>", "conv_igemm<1128>", "conv3x3_patch")


class _WorkQueue:
    """Host work queue shared by the ranks of one node (BASELINE configs[4]): an atomic counter in torch.distributed's
    store — ``next()`` hands out chunk ids 0, 1, 2 ... to whichever rank asks first."""

    def __init__(self, dist, name):
        self.local = 0
        self.store = None
        self.key = name
        if dist is not None:
            from torch.distributed import distributed_c10d as c10d

            self.store = c10d._get_default_store()

    def next(self):
        if self.store is None:
            self.local += 1
            return self.local - 1
        return int(self.store.add(self.key, 1)) - 1


def _checksum(records):
    """Order-sensitive checksum of gathered results [(page index, n boxes, token lists)]."""
    acc = 0
    for i, (pi, nb, toks) in enumerate(records):
        t = sum((j + 1) * int(sum(int(v) for v in tk)) for j, tk in enumerate(toks))
        acc = (acc * 1000003 + (i + 1) * (pi + 1) * 7919 + nb * 104729 + t) % (1 << 61)
    return acc


def run_dit_trocr(args, torch, dist, rank, local_rank, world, prec):
    """BASELINE configs[2] (default), configs[3] (--stream-pages) and configs[4] (mixed-DPI leg): DiT-base detector +
    TrOCR-base recognizer, full pages, one GPU per rank."""
    import threading

    from marie_icr_amd._lib import Context, CropDesc
    from marie_icr_amd.dist import arenas_checksum, broadcast_arenas, gather_in_order, shard_indices
    from marie_icr_amd.dit import DitModel
    from marie_icr_amd.trocr import TrocrModel, default_config as trocr_config
    from marie_icr_amd.weights import make_dit_state, make_page_bgr, make_trocr_state, page_line_boxes

    P, DB = args.pages, max(1, min(args.det_batch, args.pages))
    streams = [torch.cuda.current_stream(), torch.cuda.Stream()]
    ctxs = [Context(local_rank), Context(local_rank)]
    for c, s_ in zip(ctxs, streams):
        c.set_stream(s_.cuda_stream)
    tcfg = trocr_config(ctxs[1].lib, "base")
    tcfg.max_len_b = args.decode_len
    dims = ((tcfg.enc_dim, tcfg.enc_depth, tcfg.enc_heads), (tcfg.dec_dim, tcfg.dec_layers, tcfg.dec_heads, tcfg.dec_ffn),
            tcfg.vocab)
    need_state = rank == 0
    pg_info = None
    dit_state = make_dit_state(0, args.model) if need_state else None
    trocr_state = make_trocr_state(0, dims[0], dims[1], dims[2], tcfg.max_positions) if need_state else None
    det = DitModel(ctxs[0], dit_state, model=args.model, precision=prec)
    rec = TrocrModel(ctxs[1], trocr_state, tcfg, prec)
    # the recognizer says where its decode phase starts in its stream; the detector's stream waits there (--no-phase-align:
    # the two streams race from the start of the step, the round-2 behaviour)
    gate = None if args.no_phase_align else rec.decode_gate()
    if dist is not None:   # rank 0 packed the weights; everyone else receives the packed arenas over RCCL / xGMI (a launcher
        # with one rank goes through the same collectives: the RCCL path is exercised on a one-GPU box too)
        for m, c, s_ in ((det, ctxs[0], streams[0]), (rec, ctxs[1], streams[1])):
            if rank != 0:
                m.alloc_arena()
            with torch.cuda.stream(s_):
                broadcast_arenas(m, c, dist, src=0)
                sums = [None] * world
                dist.all_gather_object(sums, arenas_checksum(m, c))
                if len(set(sums)) != 1:
                    raise SystemExit(f"rank {rank}: weight arenas differ across ranks after the broadcast: {sums}")
        pg_info = {"backend": dist.get_backend(), "world": world, "arena_checksums_equal": True}
    n_pool = min(P, 4)
    host_pages = np.stack([make_page_bgr(1000 + rank * 97 + i, PAGE_H, PAGE_W, n_lines=LINES_PER_PAGE) for i in range(n_pool)])
    pages = torch.from_numpy(host_pages[np.arange(P) % n_pool]).cuda()     # [P][H][W][3] in HBM
    page_bytes = PAGE_H * PAGE_W * 3
    gt = page_line_boxes(PAGE_H, PAGE_W, LINES_PER_PAGE)
    def make_descs(slots):
        """crop windows of the generator's line boxes for pages sitting in slots `slots` of a packed page buffer"""
        d = (CropDesc * (len(slots) * LINES_PER_PAGE))()
        i = 0
        for sl in slots:
            for x, y, w, h in gt.tolist():
                d[i] = CropDesc(sl * page_bytes + (y * PAGE_W + x) * 3, h + 1, w + 1, PAGE_W * 3, 3)
                i += 1
        return d

    stats = {"boxes": 0}
    last = [None]
    last_det = [None]
    det_passes = [args.det_passes]
    # what a step works on: device address of a packed page buffer, the slots of the step's pages in it, their crop windows
    cur = {"base": pages.data_ptr(), "slots": list(range(P)), "descs": make_descs(range(P))}

    gate_seq = [None]          # signal this step's detector waits for (None: not gated)

    def detect_all():
        nb = 0
        per_page = []
        slots = cur["slots"]
        if gate is not None and gate_seq[0] is not None:
            gate.wait(ctxs[0], gate_seq[0], timeout_ms=60000)
            gate_seq[0] += 1
        for pz in range(det_passes[0]):
            for s0 in range(0, len(slots), DB):
                ptrs = [cur["base"] + sl * page_bytes for sl in slots[s0:s0 + DB]]
                for boxes, _scores in det.detect_device(ptrs, PAGE_H, PAGE_W):
                    nb += len(boxes)
                    if pz == 0:
                        per_page.append(len(boxes))
        stats["boxes"] += nb // det_passes[0]
        last_det[0] = per_page

    def recognize_all():
        last[0] = rec.generate_fragments(cur["base"], cur["descs"], len(cur["slots"]) * LINES_PER_PAGE, swap_rb=True)

    def run(k, serial=False):
        # the detector and the recognizer of a step work on the same pages but do not depend on each other here (the
        # recognizer consumes ground-truth boxes), so they run as two host threads on two streams
        if serial:
            for _ in range(k):
                detect_all()
                torch.cuda.synchronize()
                recognize_all()
            return
        def loop(fn):
            for _ in range(k):
                fn()
        # step j's detector starts where step j's recognizer begins to decode: signal number (count now) + j + 1
        gate_seq[0] = gate.count() + 1 if gate is not None else None
        ths = [threading.Thread(target=loop, args=(f,)) for f in (detect_all, recognize_all)]
        for th in ths:
            th.start()
        try:
            for th in ths:
                th.join()
        finally:
            gate_seq[0] = None

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    run(max(1, args.warmup), args.serial)
    fence()
    stats["boxes"] = 0
    fence()
    t0 = time.perf_counter()
    run(args.steps, args.serial)
    fence()
    dt = time.perf_counter() - t0
    boxes_pp = stats["boxes"] / (args.steps * P)

    def profiled(serial):
        for c in ctxs:
            c.profile_reset()
            c.profile_enable(True)
        tA = time.perf_counter()
        if serial:
            detect_all()
            torch.cuda.synchronize()
            tB = time.perf_counter()
            recognize_all()
        else:
            run(1)
            tB = tA
        fence()
        tC = time.perf_counter()
        prof = {}
        for c in ctxs:
            for name, v in c.profile_read().items():
                acc = prof.setdefault(name, {"total_ms": 0.0, "launches": 0, "flops": 0.0})
                for kk in acc:
                    acc[kk] += v[kk]
            c.profile_enable(False)
        return prof, 1e3 * (tB - tA), 1e3 * (tC - tB)

    prof = prof_iso = None
    if not args.no_kernel_timing:      # two more steps with HIP events around every kernel: as timed (two streams), then serial
        prof, _, wall_ms = profiled(args.serial)
        prof_iso, det_ms, rec_ms = profiled(True)
        alone_ms = {"detector_ms_per_step_alone": det_ms, "recognizer_ms_per_step_alone": rec_ms}
    # ---- secondary figures (N = 1 only, after the timed region) ----------------------------------------------------------
    extra = {}
    if world == 1 and not args.no_secondary:
        det_passes[0] = 3                      # worst case of the reference's default refinement loop (3 detector forwards)
        run(1)
        fence()
        t1 = time.perf_counter()
        run(2)
        fence()
        extra["det_passes_3"] = {"value": 2 * P / (time.perf_counter() - t1), "unit": "pages/s",
                                 "what": "same step with three detector forwards per page (bbox_refinement=True worst case)"}
        det_passes[0] = args.det_passes
    pcie = None
    if world == 1 and args.host_steps > 0:
        # The same step with the pages arriving in (pinned) host memory: marie_icr_amd.ingest.PageFeeder copies step
        # i + 1's pages on its own stream while step i computes.  Reported beside `value`, never as `value`.
        from marie_icr_amd.ingest import PageFeeder

        host_batch = [host_pages[i % len(host_pages)] for i in range(P)]
        hs = args.host_steps
        feeder = PageFeeder((host_batch for _ in range(hs + 1)), capacity_bytes=P * page_bytes, consumer_stream=streams,
                            device=local_rank)
        t1 = None
        for i, (ptr, _shape) in enumerate(feeder):
            if i == 1:                 # step 0 warms the pinned buffers and the copy stream up
                torch.cuda.synchronize()
                t1 = time.perf_counter()
            cur["base"] = ptr
            run(1)
        torch.cuda.synchronize()
        dth = time.perf_counter() - t1
        cur["base"] = pages.data_ptr()
        pcie = {"value": P * hs / dth, "unit": "pages/s", "steps": hs, "ms_per_step": 1e3 * dth / hs,
                "h2d_gb_per_step": P * page_bytes / 1e9,
                "how": "pages in pinned host memory, double-buffered H2D on a copy stream under the previous step's kernels"}
        del feeder

    # ---- BASELINE configs[3]: a fixed stream of pages sharded over the ranks, results gathered in page order -------------
    stream = None
    total = args.stream_pages if args.stream_pages >= 0 else (2048 if world > 1 else 2 * P)
    if total > 0:
        mine = shard_indices(total, rank, world)          # rank r owns pages r, r + world, ...
        # page i of the stream = seeded page i % 4, the same on every rank (so the gathered result does not depend on N)
        spool = torch.from_numpy(np.stack([make_page_bgr(2000 + i, PAGE_H, PAGE_W, n_lines=LINES_PER_PAGE)
                                           for i in range(4)])).cuda()
        saved = dict(cur)
        desc_cache = {}
        fence()
        t1 = time.perf_counter()
        local = []
        for s0 in range(0, len(mine), P):
            ids = mine[s0:s0 + P]
            slots = [pi % 4 for pi in ids]
            key = tuple(slots)
            if key not in desc_cache:
                desc_cache[key] = make_descs(slots)
            cur.update({"base": spool.data_ptr(), "slots": slots, "descs": desc_cache[key]})
            run(1)
            hyps = last[0]
            for j, pi in enumerate(ids):
                toks = [hyps[j * LINES_PER_PAGE + c][0].tolist() for c in range(LINES_PER_PAGE)]
                local.append((pi, int(last_det[0][j]), toks))
        cur.update(saved)
        allrec = gather_in_order(local, total, dist)       # all_gather_object: variable-length records, page order
        fence()
        ds = time.perf_counter() - t1
        if dist is not None:
            tm = torch.tensor([ds], dtype=torch.float64, device="cuda")
            dist.all_reduce(tm, op=dist.ReduceOp.MAX)
            ds = float(tm.item())
        assert [r[0] for r in allrec] == list(range(total))
        stream = {"config": "BASELINE configs[3]: fixed stream, static round-robin shard (marie_icr_amd.dist.shard_indices), "
                            "boxes-per-page + token ids gathered in page order (gather_in_order)",
                  "pages": total, "seconds": ds, "value": total / ds, "unit": "pages/s", "scaling": "strong",
                  "pages_per_rank": len(mine), "steps_per_rank": (len(mine) + P - 1) // P,
                  "result_checksum": _checksum(allrec)}

    # ---- BASELINE configs[4]: mixed-DPI stream, host work queue, pages bucketed by size, crops pooled ---------------------
    mixed = None
    if not args.no_mixed_dpi:
        mixed = run_mixed_dpi(args, torch, dist, rank, world, det, rec, ctxs, streams, fence)

    if dist is not None:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    engine = overlay = None
    if world == 1 and not args.no_secondary:
        engine = run_engine_api(args, torch, det, rec, ctxs, streams, host_pages, gt)
        overlay = run_overlay_leg(args, torch, ctxs[0], pages, prec)
    if rank != 0:
        return
    rate = world * P * args.steps / dt
    nh, nw, H32, W32 = det.resized_shape(PAGE_H, PAGE_W)
    out = {
        "metric": "pages/sec (2550x3300, ~40 lines/page)", "value": rate, "unit": "pages/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
        "config": {
            "workload": f"BASELINE configs[2]: full detect->crop->recognize, {P} synthetic {PAGE_W}x{PAGE_H}x3 u8 pages per "
                        f"GPU per step resident in HBM; DiT-{args.model} Mask R-CNN detector (resize to {nh}x{nw}, "
                        f"{args.det_passes} pass, {args.det_batch} pages per forward) + TrOCR-base recognizer on the "
                        f"generator's {LINES_PER_PAGE} ground-truth line boxes per page (fixed work), beam 3, "
                        f"{args.decode_len}+1 decoder steps; seeded random weights; f16 operands with fp32 accumulation is the "
                        f"16-bit mode of this build (configs[2] says bf16: same width; the reference's GPU path is .half())"
                        + ("; detector and recognizer run one after the other (--serial)" if args.serial else ""),
            "pages_per_gpu_per_step": P, "crops_per_page": LINES_PER_PAGE, "detector_boxes_per_page": boxes_pp,
            "parallelism": f"dp{world} (independent pages); detector and recognizer on two streams per GPU, the detector "
                           + ("racing the recognizer" if args.no_phase_align else "started where the recognizer begins to decode"),
        },
    }
    if prof is not None:
        peak = PEAK_MFMA_TFLOPS_F16 if args.precision == "f16" else PEAK_MFMA_TFLOPS_F32

        def tf(k):
            return k["flops"] / (k["total_ms"] * 1e-3) / 1e12 if k["total_ms"] > 0 else 0.0

        k, ki = prof["conv_igemm"], prof_iso["conv_igemm"]
        ach = tf(k)
        out["roofline"] = {
            "bound": "mfma", "kernel": "conv_igemm", "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
            "traffic": None, "launches_per_step": k["launches"], "avg_launch_ms": k["total_ms"] / max(k["launches"], 1),
            "algorithmic_gflop_per_step": k["flops"] / 1e9, "algorithmic_gflop_per_launch": k["flops"] / 1e9 / max(k["launches"], 1),
            "measured": "HIP events on the launch streams around every conv_igemm launch of one more step run exactly as the "
                        "timed steps (detector and recognizer concurrently on two streams: in situ); FLOPs = 2*MAC of every "
                        "launch (mhip_profile_flops).  profiles/r03/c_*_kernel_stats.csv is rocprofv3 --kernel-trace --stats of "
                        "this command: sum its conv_igemm_kernel + conv3x3_patch_kernel rows",
            "isolated": {"achieved": tf(ki), "frac": tf(ki) / peak, "avg_launch_ms": ki["total_ms"] / max(ki["launches"], 1),
                         "measured": "same step, detector then recognizer alone (nothing else on the GPU); rocprofv3 of "
                                     "`bench.py --serial` reproduces it"},
            "by_tile": {n: {"launches": prof[n]["launches"], "gflop": prof[n]["flops"] / 1e9, "ms": prof[n]["total_ms"],
                            "tflops": tf(prof[n]), "tflops_isolated": tf(prof_iso[n])} for n in IGEMM_VARIANTS if prof[n]["launches"]},
        }
        # HBM-side bytes per launch: PMC counters cannot be read from inside this process, so the committed result of
        # the rocprofv3 --pmc passes over this same configuration is reported (null for any other configuration)
        pmc = _pmc_traffic({"workload": "dit_trocr", "pages": P, "det_batch": DB, "decode_len": args.decode_len,
                            "model": args.model, "det_passes": args.det_passes, "precision": args.precision})
        if pmc is not None and "conv_igemm" in pmc["kernels"]:
            out["roofline"]["traffic"] = pmc["kernels"]["conv_igemm"]["bytes_per_launch"]
            out["roofline"]["traffic_unit"] = f"bytes per launch, HBM side (FETCH_SIZE x 2 + WRITE_SIZE; {pmc['source']})"
        a, ai = prof["attn_flash"], prof_iso["attn_flash"]
        out["roofline_attention"] = {"kernel": "attn_flash", "bound": "mfma", "achieved": tf(a), "isolated": tf(ai),
                                     "peak": peak, "unit": "TFLOP/s", "algorithmic_gflop_per_step": a["flops"] / 1e9}
        agg = {n: v["total_ms"] for n, v in prof.items() if v["launches"] and n not in IGEMM_VARIANTS}
        out["kernels_ms_per_step"] = agg
        out["kernels_ms_per_step_isolated"] = {n: v["total_ms"] for n, v in prof_iso.items() if v["launches"] and n not in IGEMM_VARIANTS}
        out["kernel_ms_over_wall_ms"] = sum(agg.values()) / (1e3 * dt / args.steps)
        out.update(alone_ms)
    out.update(extra)
    if pg_info is not None:
        out["process_group"] = pg_info
    if pcie is not None:
        out["pcie_inclusive"] = pcie
    if stream is not None:
        out["stream"] = stream
    if mixed is not None:
        out["mixed_dpi"] = mixed
    if engine is not None:
        out["engine_api"] = engine
    if overlay is not None:
        out["overlay"] = overlay
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"], out["parity"] = cpu_baseline_dit_trocr(dit_state, trocr_state, dims, args.decode_len,
                                                                    LINES_PER_PAGE, det, rec, args.precision)
    out["sample_output"] = [[int(t) for t in last[0][0][0]], last[0][0][1]] if last[0] else None
    print(json.dumps(out), flush=True)


def run_mixed_dpi(args, torch, dist, rank, world, det, rec, ctxs, streams, fence):
    """BASELINE configs[4]: pages scanned at 150 / 200 / 300 DPI interleaved (page i has size i % 3).  Ranks pull chunks of
    24 consecutive pages from a host work queue (whoever is free takes the next chunk: big and small pages cost different
    time), bucket the chunk's pages by size for the detector (8 same-size pages per forward) and pool the line crops of the
    whole chunk into one recognizer batch (every crop is resized to 384 x 384 on the device whatever its source DPI)."""
    import threading

    from marie_icr_amd._lib import CropDesc
    from marie_icr_amd.dist import gather_in_order
    from marie_icr_amd.weights import make_page_bgr, page_line_boxes

    CH = 24
    total = max(CH, (args.mixed_pages // CH) * CH) * world
    n_chunks = total // CH
    pool, gts = [], []
    for k, (h, w) in enumerate(MIXED_DPI_SIZES):            # two distinct pages per size, resident in HBM
        hp = np.stack([make_page_bgr(3000 + 10 * k + j, h, w, n_lines=LINES_PER_PAGE) for j in range(2)])
        pool.append(torch.from_numpy(hp).cuda())
        gts.append(page_line_boxes(h, w, LINES_PER_PAGE))
    base = min(p.data_ptr() for p in pool)

    def page_ptr(i):
        k = i % 3
        return pool[k].data_ptr() + ((i // 3) % 2) * pool[k][0].numel(), k

    def chunk_inputs(c):
        ids = list(range(c * CH, (c + 1) * CH))
        buckets = {0: [], 1: [], 2: []}
        descs = (CropDesc * (CH * LINES_PER_PAGE))()
        n = 0
        for i in ids:
            ptr, k = page_ptr(i)
            buckets[k].append(ptr)
            h, w = MIXED_DPI_SIZES[k]
            for x, y, bw, bh in gts[k].tolist():
                descs[n] = CropDesc(ptr - base + (y * w + x) * 3, bh + 1, bw + 1, w * 3, 3)
                n += 1
        return ids, buckets, descs, n

    gate = None if args.no_phase_align else rec.decode_gate()

    def process(c):
        ids, buckets, descs, n = chunk_inputs(c)
        res = {}
        seq = gate.count() + 1 if gate is not None else None     # this chunk's recognizer call: where its decoding starts

        def detect():
            counts = {}
            if gate is not None:
                gate.wait(ctxs[0], seq, timeout_ms=60000)
            for k, ptrs in buckets.items():
                h, w = MIXED_DPI_SIZES[k]
                out = []
                for s0 in range(0, len(ptrs), args.det_batch):
                    out += [len(b) for b, _ in det.detect_device(ptrs[s0:s0 + args.det_batch], h, w)]
                counts[k] = out
            res["det"] = counts

        def recognize():
            res["hyp"] = rec.generate_fragments(base, descs, n, swap_rb=True)

        ths = [threading.Thread(target=f) for f in (detect, recognize)]
        for th in ths:
            th.start()
        for th in ths:
            th.join()
        seen = {0: 0, 1: 0, 2: 0}
        recs = []
        for j, i in enumerate(ids):
            k = i % 3
            nb = res["det"][k][seen[k]]
            seen[k] += 1
            recs.append((i, int(nb), [res["hyp"][j * LINES_PER_PAGE + c_][0].tolist() for c_ in range(LINES_PER_PAGE)]))
        return recs

    process(0)                                              # warm-up: three page geometries, one chunk-sized recognizer batch
    fence()
    q = _WorkQueue(dist, "mixed_dpi_next")
    t0 = time.perf_counter()
    local, mine = [], []
    while True:
        c = q.next()
        if c >= n_chunks:
            break
        mine.append(c)
        local.extend(process(c))
    if dist is not None:
        parts = [None] * world
        dist.all_gather_object(parts, local)
        allrec = sorted((r for p_ in parts for r in p_), key=lambda r: r[0])
    else:
        allrec = local
    fence()
    ds = time.perf_counter() - t0
    if dist is not None:
        tm = torch.tensor([ds], dtype=torch.float64, device="cuda")
        dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        ds = float(tm.item())
    assert [r[0] for r in allrec] == list(range(total)), "mixed-DPI gather lost or duplicated pages"
    if rank != 0:
        return None
    return {"config": "BASELINE configs[4]: 150/200/300-DPI pages interleaved (i % 3), chunks of 24 pages pulled from a host "
                      "work queue (torch.distributed store counter), pages bucketed by size for the detector, line crops of a "
                      "chunk pooled into one recognizer batch; f16", "pages": total, "seconds": ds, "value": total / ds,
            "unit": "pages/s", "scaling": "strong", "chunks_this_rank": len(mine), "sizes_hw": [list(s) for s in MIXED_DPI_SIZES],
            "result_checksum": _checksum(allrec)}


def run_overlay_leg(args, torch, ctx, pages, prec):
    """SURVEY.md 8(f) row 3, the step before the path: the overlay cleaner's generator (pix2pixHD LocalEnhancer, ngf 64, seeded
    weights) on the same resident 2550 x 3300 pages — page in HBM -> generator -> image in HBM."""
    import ctypes as C

    from marie_icr_amd.overlay import OverlayModel
    from marie_icr_amd.weights import make_overlay_state

    m = OverlayModel(ctx, make_overlay_state(0, 64), 64, prec)
    H, W = m.padded_shape(PAGE_H, PAGE_W)
    fake = torch.empty((H, W, 3), dtype=torch.uint8, device="cuda")
    page_bytes = PAGE_H * PAGE_W * 3
    n = 3
    m.forward_device(pages.data_ptr(), PAGE_H, PAGE_W, fake.data_ptr())          # warm-up (workspace growth)
    torch.cuda.synchronize()
    ctx.profile_reset()
    ctx.profile_enable(True)
    t0 = time.perf_counter()
    for i in range(n):
        m.forward_device(pages.data_ptr() + (i % len(pages)) * page_bytes, PAGE_H, PAGE_W, fake.data_ptr())
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    prof = ctx.profile_read()
    ctx.profile_enable(False)
    k = prof["conv_igemm"]
    m.close()
    peak = PEAK_MFMA_TFLOPS_F16 if args.precision == "f16" else PEAK_MFMA_TFLOPS_F32
    tf = k["flops"] / (k["total_ms"] * 1e-3) / 1e12 if k["total_ms"] > 0 else 0.0
    return {"value": n / dt, "unit": "pages/s", "ms_per_page": 1e3 * dt / n, "canvas_hw": [H, W],
            "conv_igemm": {"gflop_per_page": k["flops"] / 1e9 / n, "ms_per_page": k["total_ms"] / n, "tflops": tf, "frac": tf / peak},
            "kernels_ms_per_page": {name: v["total_ms"] / n for name, v in prof.items() if v["launches"] and name not in IGEMM_VARIANTS},
            "what": "OverlayProcessor's generator alone (netG local, ngf 64, instance norm), one page per forward as in the "
                    "reference (batch_size 1), page resident in HBM; FLOPs as launched (stride-2 convolutions at full width, the "
                    "transposed convolution over a zero-inserted image)"}


def run_engine_api(args, torch, det, rec, ctxs, streams, host_pages, gt):
    """The same models behind the drop-in surface: ``MarieHipOcrEngine.extract(frames)`` with host frames in, result
    dictionaries out — H2D copy, detector, host geometry (merge_boxes, aspect filter, lines_from_bboxes, find_line_numbers,
    sort), fragment windows, recognizer, ``OcrProcessor.recognize`` bookkeeping and token -> text all inside the timed call."""
    from marie_icr_amd.box_processor import PSMode
    from marie_icr_amd.dit_box_processor import BoxProcessorUlimDit
    from marie_icr_amd.ocr_engine import CoordinateFormat, MarieHipOcrEngine
    from marie_icr_amd.trocr import TrOcrProcessor

    P = args.pages
    frames = [host_pages[i % len(host_pages)] for i in range(P)]
    gt_xyxy = np.stack([gt[:, 0], gt[:, 1], gt[:, 0] + gt[:, 2] + 1, gt[:, 1] + gt[:, 3] + 1], 1).astype(np.float32)

    def make_box(fixed_lines, refinement):
        bp = BoxProcessorUlimDit(cuda=True, refinement=refinement, dit_model=det, det_batch=args.det_batch)
        if fixed_lines:
            real = bp._detect_batch

            def fixed(page_devs, shape):
                real(page_devs, shape)                            # the detector runs in full; its (random-weight) boxes are dropped
                return [(gt_xyxy.copy(), np.ones(len(gt_xyxy), np.float32)) for _ in page_devs]
            bp._detect_batch = fixed
        return bp

    tp = TrOcrProcessor(trocr_model=rec, batch_size=min(P, 32) * LINES_PER_PAGE)
    if args.no_phase_align:
        tp.decode_gate = None
    out = {}
    for name, fixed, refine, n_pages in (("fixed_lines", True, False, P), ("detector_driven", False, False, min(P, 8)),
                                         ("detector_driven_refinement", False, True, min(P, 8))):
        bp = make_box(fixed, refine)
        eng = MarieHipOcrEngine(box_processor=bp, default_ocr_processor=tp)
        eng.page_batch = min(P, args.engine_page_batch)    # several batches: the detector of batch k + 1 runs under the recognizer of batch k
        if args.engine_first_batch > 0:
            eng.first_batch = args.engine_first_batch
        if os.environ.get("MARIE_ENGINE_STREAM_BATCH"):            # tuning aids (tools/r03_engine_sweep2.sh)
            eng.stream_batch = int(os.environ["MARIE_ENGINE_STREAM_BATCH"])
        if os.environ.get("MARIE_ENGINE_NO_STREAM"):
            eng.stream_recognizer = False
        # wall time spent inside the two processors (they run on two host threads, so the two can add up to more than the call)
        spent = {"detect_s": 0.0, "recognize_s": 0.0}
        spans = []          # (what, pages, start, end) of every processor call: the timeline of the last extract()

        def timed(fn, key):
            @functools.wraps(fn)          # the engine looks at the signature (copy_fragments)
            def wrapper(*a, **k):
                t = time.perf_counter()
                try:
                    return fn(*a, **k)
                finally:
                    e = time.perf_counter()
                    spent[key] += e - t
                    spans.append((key[:-2], len(a[2]) if len(a) > 2 else (len(a[0]) if len(a) == 1 else 0), t, e))
            return wrapper
        bp.extract_bounding_boxes_batch = timed(bp.extract_bounding_boxes_batch, "detect_s")
        orig_rec = tp.recognize_pages
        tp.recognize_pages = timed(orig_rec, "recognize_s")
        orig_add, orig_fin = tp.recognize_pages_add, tp.recognize_pages_finish       # the engine's streaming path: encode per batch,
        tp.recognize_pages_add = timed(orig_add, "recognize_s")                      # one beam search at the end
        tp.recognize_pages_finish = timed(orig_fin, "recognize_s")
        fr = frames[:n_pages]
        reps = 2 if fixed else 1
        if fixed:
            eng.extract(fr, PSMode.SPARSE, CoordinateFormat.XYXY)     # warm-up
        torch.cuda.synchronize()
        spent.update(detect_s=0.0, recognize_s=0.0)
        t0 = time.perf_counter()
        for _ in range(reps):
            del spans[:]
            tl0 = time.perf_counter()
            res = eng.extract(fr, PSMode.SPARSE, CoordinateFormat.XYXY)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        tp.recognize_pages, tp.recognize_pages_add, tp.recognize_pages_finish = orig_rec, orig_add, orig_fin
        words = sum(len(r["words"]) for r in res) / len(res)
        out[name] = {"value": reps * n_pages / dt, "unit": "pages/s", "pages_per_call": n_pages, "words_per_page": words,
                     "lines_per_page": sum(len(r["lines"]) for r in res) / len(res), "page_batch": eng.page_batch,
                     "s_per_call": dt / reps, "s_in_box_processor": spent["detect_s"] / reps,
                     "s_in_ocr_processor": spent["recognize_s"] / reps,
                     "timeline_ms": [[w, n, round((a - tl0) * 1e3, 1), round((b - tl0) * 1e3, 1)] for w, n, a, b in sorted(spans, key=lambda x: x[2])]}
    out["what"] = ("MarieHipOcrEngine.extract(frames) end to end, host numpy frames in (H2D inside), result dictionaries out; "
                   "fixed_lines: the detector runs in full but the generator's 40 line boxes go on (the headline's fixed work, "
                   "bbox_refinement=False); detector_driven: whatever the random-weight detector emits becomes a crop "
                   "(SURVEY 8d B); _refinement: the reference's default 3-pass loop")
    return out


def launch_ranks(n: int) -> int:
    """``python bench.py --gpus N`` without a launcher: start N ranks (one per GPU) under torch.distributed.run and hand
    back its exit code.  Runs BEFORE this process has imported torch or touched the GPU — the ranks are fresh children."""
    import socket
    import subprocess

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--engine-page-batch", type=int, default=32, help="pages per detector / recognizer batch of the engine_api leg")
    ap.add_argument("--engine-first-batch", type=int, default=0, help="pages of the first batch of the engine_api leg (0 = the engine's default)")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", choices=["dit_trocr", "craft_crnn", "pages", "crnn"], default="dit_trocr")
    ap.add_argument("--pages", type=int, default=0, help="pages per GPU per step (default: 64 dit_trocr = BASELINE configs[2], 12 craft_crnn)")
    ap.add_argument("--det-batch", type=int, default=8, help="pages per detector forward (dit_trocr)")
    ap.add_argument("--decode-len", type=int, default=15, help="generated tokens before the forced EOS (dit_trocr)")
    ap.add_argument("--model", choices=["base", "large"], default="base", help="DiT detector size (dit_trocr)")
    ap.add_argument("--det-passes", type=int, default=1, choices=[1, 2, 3],
                    help="detector forwards per page (dit_trocr): 1 = bbox_refinement False (the headline); 3 = the worst "
                         "case of the reference's default refinement loop (psm_sparse re-runs the detector on the "
                         "blacked-out page up to 3 times)")
    ap.add_argument("--serial", action="store_true",
                    help="dit_trocr: detector and recognizer one after the other instead of on two streams (what "
                         "roofline.isolated measures; for rocprofv3 runs of the isolated kernels)")
    ap.add_argument("--no-phase-align", action="store_true",
                    help="dit_trocr: let the detector and the recognizer streams race from the start of a step instead of "
                         "starting the detector where the recognizer begins to decode")
    ap.add_argument("--stream-pages", type=int, default=-1,
                    help="dit_trocr, BASELINE configs[3]: also time a fixed stream of this many pages sharded over the ranks "
                         "with the results gathered in page order (default: 2048 when N > 1, two steps' worth at N = 1; 0 = off)")
    ap.add_argument("--mixed-pages", type=int, default=48,
                    help="dit_trocr, BASELINE configs[4]: pages per GPU of the mixed-DPI stream leg (multiple of 24)")
    ap.add_argument("--no-mixed-dpi", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip det_passes_3 and the engine_api legs (N = 1)")
    ap.add_argument("--crops", choices=["lines", "detector"], default="lines")
    ap.add_argument("--inflight", type=int, default=6,
                    help="page pipelines per GPU (one context + stream + host thread each): the host-side box "
                         "finalisation of one page overlaps the kernels of another")
    ap.add_argument("--lines", type=int, default=1024, help="lines per GPU per step (workload crnn)")
    ap.add_argument("--width", type=int, default=256)
    ap.add_argument("--precision", choices=["f16", "f32"], default="f16")
    ap.add_argument("--host-steps", type=int, default=2,
                    help="extra steps with the pages fed from pinned host memory (dit_trocr, N=1; reported as pcie_inclusive)")
    ap.add_argument("--no-kernel-timing", action="store_true", help="no per-kernel HIP events in the timed region")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()
    if args.workload == "pages":
        args.workload = "craft_crnn"
    if args.pages <= 0:
        args.pages = 64 if args.workload == "dit_trocr" else 12       # BASELINE configs[2]: 64 pages
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")

    # N > 1 without a launcher: become the launcher (nothing below this line has run, no GPU call has been made)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(launch_ranks(args.gpus))

    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: refusing to report a number for the wrong N")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    # Rehearsal of the N > 1 path on a ONE-GPU box (not for measurements): MARIE_BENCH_REHEARSE=1 puts every rank on
    # device 0 and uses gloo for the start-up broadcast / barriers (RCCL refuses two ranks on one device).
    rehearse = os.environ.get("MARIE_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    elif world > torch.cuda.device_count():
        raise SystemExit(f"bench.py: --gpus {world} but only {torch.cuda.device_count()} device(s) visible")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or "WORLD_SIZE" in os.environ:
        # under a launcher — `python -m torch.distributed.run --nproc-per-node=1 bench.py` included — the process group is
        # real: weights broadcast, checksums gathered, timings reduced and the work queue kept in the store, also with one rank
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group(backend="nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))

    from marie_icr_amd._lib import PREC_F16, PREC_F32, Context
    from marie_icr_amd.craft import CraftModel, adjust_result_coordinates, rects_from_boxes
    from marie_icr_amd.crnn import CrnnModel, tokens_to_text_fast
    from marie_icr_amd.dist import broadcast_arena
    from marie_icr_amd.weights import (CRNN_CHARSET, make_craft_bench_state, make_crnn_input, make_crnn_state,
                                       make_page_bgr, page_line_boxes)

    prec = PREC_F16 if args.precision == "f16" else PREC_F32
    if args.workload == "dit_trocr":
        run_dit_trocr(args, torch, dist, rank, local_rank, world, prec)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return
    n_pipe = max(1, args.inflight) if args.workload == "craft_crnn" else 1
    streams = [torch.cuda.current_stream()] + [torch.cuda.Stream() for _ in range(n_pipe - 1)]
    ctxs = [Context(local_rank) for _ in range(n_pipe)]
    for c, s_ in zip(ctxs, streams):
        c.set_stream(s_.cuda_stream)
    ctx, stream = ctxs[0], streams[0]

    # ---- weights: rank 0 packs, everyone else receives the packed arenas over RCCL ----------
    crnn_state = make_crnn_state(0)
    craft_state = make_craft_bench_state() if args.workload == "craft_crnn" else None

    def load(cls, c, state, **kw):
        if world > 1:
            m = cls(c, state if rank == 0 else None, precision=prec, **kw)
            if rank != 0:
                m.alloc_arena()
            with torch.cuda.stream(streams[ctxs.index(c)]):
                broadcast_arena(m, c, dist, src=0)
            return m
        return cls(c, state, precision=prec, **kw)

    recs = [load(CrnnModel, c, crnn_state, num_class=95) for c in ctxs]
    dets = [load(CraftModel, c, craft_state) for c in ctxs] if args.workload == "craft_crnn" else []
    rec = recs[0]
    det = dets[0] if dets else None

    w = args.width
    T = rec.seq_len(w)
    last_texts = [None]
    units_per_step = 0
    stats = {"boxes": 0, "crops": 0}

    if args.workload == "crnn":
        n = args.lines
        units_per_step = n
        crops = torch.from_numpy(make_crnn_input(1000 + rank, n, 32, w)).cuda()
        d_arg = torch.empty((n, T), dtype=torch.int32, device="cuda")
        d_tok = [torch.empty((n, T), dtype=torch.int32, device="cuda") for _ in range(2)]
        d_len = [torch.empty((n,), dtype=torch.int32, device="cuda") for _ in range(2)]
        d_cnf = [torch.empty((n,), dtype=torch.float32, device="cuda") for _ in range(2)]
        h_tok = [torch.empty((n, T), dtype=torch.int32).pin_memory() for _ in range(2)]
        h_len = [torch.empty((n,), dtype=torch.int32).pin_memory() for _ in range(2)]
        h_cnf = [torch.empty((n,), dtype=torch.float32).pin_memory() for _ in range(2)]
        ev = [torch.cuda.Event() for _ in range(2)]

        def step(i):
            b = i & 1
            rec.forward_device(crops.data_ptr(), n, w, 0, d_arg.data_ptr(), d_tok[b].data_ptr(), d_len[b].data_ptr(),
                               d_cnf[b].data_ptr())
            h_tok[b].copy_(d_tok[b], non_blocking=True)
            h_len[b].copy_(d_len[b], non_blocking=True)
            h_cnf[b].copy_(d_cnf[b], non_blocking=True)
            ev[b].record(stream)

        def collect(i):
            b = i & 1
            ev[b].synchronize()
            last_texts[0] = tokens_to_text_fast(h_tok[b].numpy(), h_len[b].numpy(), CRNN_CHARSET)

        def run(k):
            for i in range(k):
                step(i)
                if i > 0:
                    collect(i - 1)     # host string decode of step i-1 overlaps GPU step i
            collect(k - 1)
    else:
        P = args.pages
        units_per_step = P
        host_pages = np.stack([make_page_bgr(1000 + rank * 97 + i, PAGE_H, PAGE_W, n_lines=LINES_PER_PAGE)
                               for i in range(min(P, 4))])
        pages = torch.from_numpy(host_pages[np.arange(P) % len(host_pages)]).cuda()     # [P][H][W][3] in HBM
        page_bytes = PAGE_H * PAGE_W * 3
        gt = page_line_boxes(PAGE_H, PAGE_W, LINES_PER_PAGE)

        import threading

        lock = threading.Lock()

        def worker(t, k):
            """pipeline t owns pages t, t+n_pipe, ... of every step"""
            d_, r_ = dets[t], recs[t]
            mine = list(range(t, P, n_pipe))
            for _ in range(k):
                rect_list = []
                nb = 0
                for pi in mine:
                    boxes, ratio = d_.detect_device(pages.data_ptr() + pi * page_bytes, PAGE_H, PAGE_W, *THRESH)
                    bboxes = adjust_result_coordinates(boxes, 1 / ratio, 1 / ratio)
                    rects = rects_from_boxes(bboxes, PAGE_W, PAGE_H)
                    rects = rects[(rects[:, 0] < PAGE_W) & (rects[:, 1] < PAGE_H)] if len(rects) else rects
                    nb += len(rects)
                    use = gt if args.crops == "lines" else rects
                    r = use.astype(np.int64).copy()
                    r[:, 1] += pi * PAGE_H          # the P pages are one [P*H][W][3] image for the crop batcher
                    rect_list.append(r)
                allr = np.concatenate(rect_list) if rect_list else np.zeros((0, 4), np.int64)
                texts = []
                for s0 in range(0, len(allr), 4096):
                    out = r_.forward_rects_device(pages.data_ptr(), P * PAGE_H, PAGE_W, allr[s0:s0 + 4096], w)
                    texts += tokens_to_text_fast(out["tokens"], out["lengths"], CRNN_CHARSET)
                with lock:
                    stats["boxes"] += nb
                    stats["crops"] += len(allr)
                    last_texts[0] = texts

        def run(k):
            if n_pipe == 1:
                worker(0, k)
                return
            ths = [threading.Thread(target=worker, args=(t, k)) for t in range(n_pipe)]
            for th in ths:
                th.start()
            for th in ths:
                th.join()

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    run(max(1, args.warmup))
    fence()
    stats["boxes"] = stats["crops"] = 0
    fence()
    t0 = time.perf_counter()
    run(args.steps)
    fence()
    dt = time.perf_counter() - t0
    counted = dict(stats)

    # ---- per-kernel device time: HIP events on the launch stream, ONE pipeline at a time so that concurrent streams
    # do not stretch each other's kernels (a separate pass of the same step, right after the timed region) ----------
    prof = None
    prof_steps = 0
    if not args.no_kernel_timing:
        prof_steps = 1 if args.workload == "craft_crnn" else min(args.steps, 5)
        ctx.profile_reset()
        ctx.profile_enable(True)
        if args.workload == "craft_crnn":
            worker_all = n_pipe
            n_pipe_saved = n_pipe
            # pipeline 0 processes every page of the step alone
            single = list(range(P))
            d_, r_ = dets[0], recs[0]
            for _ in range(prof_steps):
                rl = []
                for pi in single:
                    boxes, ratio = d_.detect_device(pages.data_ptr() + pi * page_bytes, PAGE_H, PAGE_W, *THRESH)
                    if args.crops == "lines":
                        use = gt
                    else:
                        use = rects_from_boxes(adjust_result_coordinates(boxes, 1 / ratio, 1 / ratio), PAGE_W, PAGE_H)
                        use = use[(use[:, 0] < PAGE_W) & (use[:, 1] < PAGE_H)] if len(use) else use
                    r = use.astype(np.int64).copy()
                    r[:, 1] += pi * PAGE_H
                    rl.append(r)
                allr = np.concatenate(rl)
                for s0 in range(0, len(allr), 4096):
                    r_.forward_rects_device(pages.data_ptr(), P * PAGE_H, PAGE_W, allr[s0:s0 + 4096], w)
        else:
            run(prof_steps)
        fence()
        prof = ctx.profile_read()
        ctx.profile_enable(False)
    stats.update(counted)

    if dist is not None:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    if rank == 0:
        rate = world * units_per_step * args.steps / dt
        if args.workload == "craft_crnn":
            P = args.pages
            crops_pp = stats["crops"] / (args.steps * P)
            out = {
                "metric": "pages/sec (2550x3300, ~40 lines/page)",
                "value": rate, "unit": "pages/s",
                "config": {
                    "workload": f"detect->crop->recognize, {P} synthetic {PAGE_W}x{PAGE_H}x3 u8 pages per GPU per step "
                                f"resident in HBM: CRAFT detector (box_segmentation_mode 2) + CRNN recognizer "
                                f"(None-VGG-BiLSTM-CTC, 32x{w} crops), seeded random weights; recognizer input = "
                                + ("the generator's 40 ground-truth line boxes per page (fixed work)"
                                   if args.crops == "lines" else "the detector's own boxes"),
                    "pages_per_gpu_per_step": P, "crops_per_page": crops_pp,
                    "detector_boxes_per_page": stats["boxes"] / (args.steps * P),
                    "parallelism": f"dp{world} (independent pages), {n_pipe} page pipelines in flight per GPU",
                    "note": "secondary workload; the default (dit_trocr) is BASELINE configs[2]",
                },
            }
            flops_step = P * det.kernel_flops(PAGE_H, PAGE_W)["conv_igemm"] + \
                rec.kernel_flops(max(1, int(round(crops_pp * P))), w)["conv_igemm"]
        else:
            n = args.lines
            out = {
                "metric": "lines/sec (CRNN recognizer stage of the pages/sec path)",
                "value": rate, "unit": "lines/s",
                "config": {
                    "workload": f"BASELINE configs[1]: CRNN recognizer only (None-VGG-BiLSTM-CTC), {n} pre-cropped "
                                f"32x{w} u8 lines per GPU per step, seeded random weights, greedy CTC decode to strings",
                    "lines_per_gpu_per_step": n, "img_w": w, "parallelism": f"dp{world} (independent batches)",
                    "pages_per_sec_equiv": rate / LINES_PER_PAGE,
                    "pages_note": "recognizer stage only at 40 lines/page; the detector is not in this number",
                },
            }
            flops_step = rec.kernel_flops(n, w)["conv_igemm"]
        out.update({"n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
                    "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.precision,
                    "data": "synthetic"})
        if prof is not None:
            k = prof["conv_igemm"]
            per_step_ms = k["total_ms"] / prof_steps
            achieved = flops_step / (per_step_ms * 1e-3) / 1e12 if per_step_ms > 0 else 0.0
            peak = PEAK_MFMA_TFLOPS_F16 if args.precision == "f16" else PEAK_MFMA_TFLOPS_F32
            out["roofline"] = {
                "bound": "mfma", "kernel": "conv_igemm", "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                "frac": achieved / peak, "traffic": None,
                "launches_per_step": k["launches"] / prof_steps,
                "measured": "HIP events on the launch stream, one page pipeline alone, same step right after the "
                            "timed region" if args.workload == "craft_crnn" else "HIP events on the launch stream",
                "avg_launch_ms": k["total_ms"] / max(k["launches"], 1),
                "algorithmic_gflop_per_step": flops_step / 1e9,
            }
            out["kernels_ms_per_step"] = {name: v["total_ms"] / prof_steps for name, v in prof.items()
                                          if v["launches"] and name not in IGEMM_VARIANTS}
            out["kernel_ms_over_wall_ms"] = sum(out["kernels_ms_per_step"].values()) / (1e3 * dt / args.steps)
        if world == 1 and not args.no_cpu_baseline:
            if args.workload == "craft_crnn":
                out["cpu_baseline"] = cpu_baseline_pages(craft_state, crnn_state, CRNN_CHARSET, w, LINES_PER_PAGE)
            else:
                out["cpu_baseline"] = cpu_baseline_crnn(crnn_state, CRNN_CHARSET, w)
        out["sample_output"] = last_texts[0][:2] if last_texts[0] else None
        print(json.dumps(out), flush=True)

    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
