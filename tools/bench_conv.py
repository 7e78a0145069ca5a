#!/usr/bin/env python3
"""Microbenchmark of conv_igemm on the recognizer's (and CRAFT's) layer shapes: device time per launch from
HIP events on the launch stream (mhip_profile_*), interleaved rounds in ONE process, random operands.
usage (GPU box): python tools/bench_conv.py [--rounds 7] [--set crnn|craft]"""
import argparse
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from marie_icr_amd._lib import PREC_F16, PREC_F32, ConvDesc, Context  # noqa: E402

SETS = {
    "crnn": [  # 1024 lines of 32x256
        ("L1 64->128 p2x2", (1024, 16, 128, 64, 3, 1, 128, 1, 1, 0)),
        ("L2 128->256", (1024, 8, 64, 128, 3, 1, 256, 0, 1, 0)),
        ("L3 256->256 p2x1", (1024, 8, 64, 256, 3, 1, 256, 2, 1, 0)),
        ("L4 256->512", (1024, 4, 64, 256, 3, 1, 512, 0, 1, 0)),
        ("L5 512->512 p2x1", (1024, 4, 64, 512, 3, 1, 512, 2, 1, 0)),
        ("L6 2x2 512->512", (1024, 2, 64, 512, 2, 0, 512, 0, 1, 0)),
        ("xproj1 512->2048 f32", (64512, 1, 1, 512, 1, 0, 2048, 0, 0, 1)),
        ("lin 512->256", (64512, 1, 1, 512, 1, 0, 256, 0, 0, 0)),
    ],
    "vit": [  # ViT encoder GEMMs: 640 TrOCR crops x 640 rows, and 8 DiT pages x 3328 rows
        ("trocr qk 768->1536", (640, 1, 640, 768, 1, 0, 1536, 0, 0, 0)),
        ("trocr proj 768->768 f32", (640, 1, 640, 768, 1, 0, 768, 0, 0, 1)),
        ("trocr fc1 768->3072 gelu", (640, 1, 640, 768, 1, 0, 3072, 0, 2, 0)),
        ("trocr fc2 3072->768 f32", (640, 1, 640, 3072, 1, 0, 768, 0, 0, 1)),
        ("dit qk 768->1536", (8, 1, 3328, 768, 1, 0, 1536, 0, 0, 0)),
        ("dit fc1 768->3072 gelu", (8, 1, 3328, 768, 1, 0, 3072, 0, 2, 0)),
        ("dit fc2 3072->768 f32", (8, 1, 3328, 3072, 1, 0, 768, 0, 0, 1)),
        ("dec fc1 1024->4096 M=1920", (1, 1, 1920, 1024, 1, 0, 4096, 0, 2, 0)),
        ("dec out 1024->50265 M=1920 f32", (1, 1, 1920, 1024, 1, 0, 50265, 0, 0, 1)),
    ],
    "ksweep": [("K=%d N=1536" % k, (640, 1, 640, k, 1, 0, 1536, 0, 0, 0)) for k in (64, 128, 256, 512, 768, 1536, 3072)] +
              [("K=768 N=%d" % n, (640, 1, 640, 768, 1, 0, n, 0, 0, 0)) for n in (128, 256, 512, 3072)],
    "craft": [  # one 1984x2560 page through VGG16-BN (Cin >= 64 layers)
        ("c1_2 64->64 @1984x2560 p2x2", (1, 1984, 2560, 64, 3, 1, 64, 1, 1, 0)),
        ("c2_1 64->128 @992x1280", (1, 992, 1280, 64, 3, 1, 128, 0, 1, 0)),
        ("c3_1 128->256 @496x640", (1, 496, 640, 128, 3, 1, 256, 0, 1, 0)),
        ("c4_1 256->512 @248x320", (1, 248, 320, 256, 3, 1, 512, 0, 1, 0)),
        ("c5_1 512->512 @124x160", (1, 124, 160, 512, 3, 1, 512, 0, 1, 0)),
    ],
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--set", default="crnn")
    ap.add_argument("--precision", default="f16")
    a = ap.parse_args()
    ctx = Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    prec = PREC_F16 if a.precision == "f16" else PREC_F32
    tdt = torch.float16 if a.precision == "f16" else torch.float32
    bufs = []
    for name, (B, H, W, Cin, K, pad, N, pool, relu, of32) in SETS[a.set]:
        x = (torch.rand((B, H, W, Cin), device="cuda") * 2 - 1).to(tdt)
        w = ((torch.rand((N, K, K, Cin), device="cuda") * 2 - 1) * (3.0 / (K * K * Cin)) ** 0.5).to(tdt)
        bias = torch.rand((N,), device="cuda") - 0.5
        Ho, Wo = H + 2 * pad - K + 1, W + 2 * pad - K + 1
        Hp, Wp = (Ho // 2, Wo // 2) if pool == 1 else ((Ho // 2, Wo) if pool == 2 else (Ho, Wo))
        out = torch.empty((B, Hp, Wp, N), dtype=torch.float32 if of32 else tdt, device="cuda")
        d = ConvDesc(B, H, W, Cin, K, K, pad, N, pool, relu, of32, 1, 0)
        flops = 2.0 * B * Ho * Wo * N * K * K * Cin
        bufs.append((name, d, x, w, bias, out, flops))
    times = {b[0]: [] for b in bufs}
    for r in range(a.rounds + 1):
        for name, d, x, w, bias, out, flops in bufs:
            ctx.profile_reset()
            ctx.profile_enable(True)
            ctx.conv2d_nhwc(prec, d, x.data_ptr(), w.data_ptr(), 0, bias.data_ptr(), out.data_ptr())
            ms = ctx.profile_read()["conv_igemm"]["total_ms"]
            ctx.profile_enable(False)
            if r > 0:
                times[name].append(ms)
    tot = 0.0
    for name, d, x, w, bias, out, flops in bufs:
        med, mn = statistics.median(times[name]), min(times[name])
        tot += med
        print(f"{name:34s} median {med * 1e3:8.1f} us  min {mn * 1e3:8.1f} us  {flops / med / 1e9:7.0f} TFLOP/s (median)")
        if os.environ.get("IGEMM_DBG_CLOCK"):      # library built with EXTRA=-DIGEMM_DBG_CLOCK: main-loop cycles / 10 ns ticks of one tile
            torch.cuda.synchronize()
            raw = out.view(-1).view(torch.uint8)[:8 * 74].cpu().numpy().view("<u8")
            cyc, ticks = int(raw[0]), int(raw[1])
            st = raw[2:74].reshape(8, 9).astype("int64")
            t0 = int(st[:, 0].min())
            print("    middle K slice, cycles since the first wave's top: top | waited | barrier | staged | reads0 | mfma0 | reads1 | mfma1 | next top")
            for wv in range(8):
                print("      wave %d: " % wv + " ".join("%5d" % (int(v) - t0) for v in st[wv]))
            if ticks:
                print(f"    main loop of one tile: {cyc} cycles in {ticks * 10} ns -> {cyc / (ticks * 10.0):.2f} GHz, "
                      f"{cyc / max(1, d.Cin * d.KH * d.KW // (64 if a.precision == 'f16' else 32)):.0f} cycles per K slice")
    print(f"sum of medians {tot * 1e3:.1f} us")


if __name__ == "__main__":
    main()
