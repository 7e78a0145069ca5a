#!/usr/bin/env python3
"""bench.py — throughput of the MI355X OCR hot path on synthetic data.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): the CRNN recognizer (None-VGG-BiLSTM-CTC) on a batch of
1024 pre-cropped 32x256 grayscale text lines per GPU, seeded random weights.  One *step* = one
pass of the recognizer hot path over one such batch: uint8 crops resident in HBM -> normalise ->
conv stack -> BiLSTM x2 -> prediction -> greedy CTC decode + confidence -> token ids / lengths /
confidences copied to pinned host memory and turned into strings (one step behind the GPU).

Multi-GPU: lines/pages are independent, so each rank owns its own batch — weak
scaling, no data-path collective.  Rank 0 packs the weights and the packed arena is broadcast
once over RCCL/xGMI at start-up; the timed region is bracketed by barrier + synchronize and the
maximum over ranks is reported.

Rank 0 prints ONE JSON line.  ``roofline`` covers the dominant kernel (conv_igemm, MFMA-bound):
its algorithmic FLOPs (mhip_crnn_kernel_flops, = SURVEY.md §8d's per-line figure x lines) divided
by its device time, measured live with HIP events on the launch stream inside the timed steps.
``cpu_baseline`` times the CPU oracle (oracle/crnn_torch.py — the same torch CPU ops the reference
executes) on rank 0's host cores on a bounded sample of the same workload.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

LINES_PER_PAGE = 40          # BASELINE.json metric: "~40 lines/page"
PEAK_MFMA_TFLOPS_F16 = 2500  # dense f16/bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_MFMA_TFLOPS_F32 = 157.3


def cpu_baseline(state, charset, img_w, target_s=12.0):
    """Time the CPU oracle on a bounded sample (rank 0, N=1 only)."""
    from marie_icr_amd.weights import make_crnn_input
    from oracle import crnn_numpy
    from oracle.crnn_torch import TorchCrnnOracle

    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 64))
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--lines", type=int, default=1024, help="lines per GPU per step")
    ap.add_argument("--width", type=int, default=256)
    ap.add_argument("--precision", choices=["f16", "f32"], default="f16")
    ap.add_argument("--no-kernel-timing", action="store_true", help="no per-kernel HIP events in the timed region")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))

    from marie_icr_amd._lib import PREC_F16, PREC_F32, Context
    from marie_icr_amd.crnn import CrnnModel, tokens_to_text_fast
    from marie_icr_amd.weights import CRNN_CHARSET, make_crnn_input, make_crnn_state

    prec = PREC_F16 if args.precision == "f16" else PREC_F32
    ctx = Context(local_rank)
    stream = torch.cuda.current_stream()
    ctx.set_stream(stream.cuda_stream)

    # ---- weights: rank 0 packs, everyone else receives the packed arena over RCCL ----------
    state = make_crnn_state(0)
    if world > 1:
        model = CrnnModel(ctx, state if rank == 0 else None, num_class=95, precision=prec)
        if rank != 0:
            model.alloc_arena()
        from marie_icr_amd.dist import broadcast_arena

        broadcast_arena(model, ctx, dist, src=0)
    else:
        model = CrnnModel(ctx, state, num_class=95, precision=prec)

    n, w = args.lines, args.width
    T = model.seq_len(w)
    crops = torch.from_numpy(make_crnn_input(1000 + rank, n, 32, w)).cuda()
    d_arg = torch.empty((n, T), dtype=torch.int32, device="cuda")
    d_tok = [torch.empty((n, T), dtype=torch.int32, device="cuda") for _ in range(2)]
    d_len = [torch.empty((n,), dtype=torch.int32, device="cuda") for _ in range(2)]
    d_cnf = [torch.empty((n,), dtype=torch.float32, device="cuda") for _ in range(2)]
    h_tok = [torch.empty((n, T), dtype=torch.int32).pin_memory() for _ in range(2)]
    h_len = [torch.empty((n,), dtype=torch.int32).pin_memory() for _ in range(2)]
    h_cnf = [torch.empty((n,), dtype=torch.float32).pin_memory() for _ in range(2)]
    ev = [torch.cuda.Event() for _ in range(2)]
    last_texts = [None]

    def step(i):
        b = i & 1
        model.forward_device(crops.data_ptr(), n, w, 0, d_arg.data_ptr(), d_tok[b].data_ptr(),
                             d_len[b].data_ptr(), d_cnf[b].data_ptr())
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

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    run(max(1, args.warmup))
    ktime = not args.no_kernel_timing
    fence()
    if ktime:
        ctx.profile_reset()
        ctx.profile_enable(True)
    fence()
    t0 = time.perf_counter()
    run(args.steps)
    fence()
    dt = time.perf_counter() - t0
    prof = None
    if ktime:
        prof = ctx.profile_read()
        ctx.profile_enable(False)

    if dist is not None:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    if rank == 0:
        lines_per_s = world * n * args.steps / dt
        out = {
            "metric": "lines/sec (CRNN recognizer stage of the pages/sec path)",
            "value": lines_per_s,
            "unit": "lines/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.precision,
            "data": "synthetic",
            "config": {
                "workload": f"BASELINE configs[1]: CRNN recognizer only (None-VGG-BiLSTM-CTC), {n} pre-cropped "
                            f"32x{w} u8 lines per GPU per step, seeded random weights, greedy CTC decode to strings",
                "lines_per_gpu_per_step": n, "img_w": w, "parallelism": f"dp{world} (independent batches)",
                "pages_per_sec_equiv": lines_per_s / LINES_PER_PAGE,
                "pages_note": "recognizer stage only at 40 lines/page; the detector is not in this number",
            },
        }
        if prof is not None:
            flops = model.kernel_flops(n, w)
            k = prof["conv_igemm"]
            per_step_ms = k["total_ms"] / args.steps
            achieved = flops["conv_igemm"] / (per_step_ms * 1e-3) / 1e12 if per_step_ms > 0 else 0.0
            peak = PEAK_MFMA_TFLOPS_F16 if args.precision == "f16" else PEAK_MFMA_TFLOPS_F32
            out["roofline"] = {
                "bound": "mfma", "kernel": "conv_igemm", "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                "frac": achieved / peak, "traffic": None,
                "launches_per_step": k["launches"] / args.steps,
                "avg_launch_ms": k["total_ms"] / max(k["launches"], 1),
                "algorithmic_gflop_per_step": flops["conv_igemm"] / 1e9,
            }
            out["kernels_ms_per_step"] = {name: v["total_ms"] / args.steps for name, v in prof.items()}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(state, CRNN_CHARSET, w)
        out["sample_output"] = last_texts[0][:2] if last_texts[0] else None
        print(json.dumps(out), flush=True)

    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
