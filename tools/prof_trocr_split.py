"""Encoder-per-batch / decoder-once timing of the TrOCR recognizer (mhip_trocr_encode_fragments / mhip_trocr_decode) against
generate_fragments on the same crops: python tools/prof_trocr_split.py [pages]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch

    from marie_icr_amd._lib import Context, CropDesc, PREC_F16
    from marie_icr_amd.trocr import TrocrModel, default_config
    from marie_icr_amd.weights import make_page_bgr, make_trocr_state, page_line_boxes

    pages = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    ctx = Context(0)
    cfg = default_config(ctx.lib, "base")
    cfg.max_len_b = 15
    st = make_trocr_state(0, (cfg.enc_dim, cfg.enc_depth, cfg.enc_heads), (cfg.dec_dim, cfg.dec_layers, cfg.dec_heads, cfg.dec_ffn), cfg.vocab, cfg.max_positions)
    m = TrocrModel(ctx, st, cfg, PREC_F16)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    pool = np.stack([make_page_bgr(1000 + i, 3300, 2550, n_lines=40) for i in range(8)])
    dev = torch.from_numpy(pool[np.arange(pages) % 8]).cuda()
    gt = page_line_boxes(3300, 2550, 40)
    page_bytes = 3300 * 2550 * 3

    def descs(p0, p1):
        d = (CropDesc * ((p1 - p0) * len(gt)))()
        k = 0
        for p in range(p0, p1):
            for x, y, w, h in gt.tolist():
                d[k] = CropDesc(p * page_bytes + (y * 2550 + x) * 3, h + 1, w + 1, 2550 * 3, 3)
                k += 1
        return d, k

    def sync():
        torch.cuda.synchronize()
        return time.perf_counter()

    dall, nall = descs(0, pages)
    m.generate_fragments(dev.data_ptr(), dall, nall)
    for rep in range(2):
        t0 = sync()
        ref = m.generate_fragments(dev.data_ptr(), dall, nall)
        t1 = sync()
    print(f"generate_fragments, {pages} pages ({nall} crops) in one call: {(t1 - t0) * 1e3:.1f} ms")
    for chunk in (8, 16, 32, pages):
        for rep in range(2):
            t0 = sync()
            m.encode_begin(nall)
            te = []
            for p0 in range(0, pages, chunk):
                d, k = descs(p0, min(pages, p0 + chunk))
                a = time.perf_counter()
                m.encode_fragments(dev.data_ptr(), d, k)
                te.append(time.perf_counter() - a)
            t1 = sync()
            got = m.decode()
            t2 = sync()
        same = all(np.array_equal(a[0], b[0]) and a[1] == b[1] for a, b in zip(ref, got))
        print(f"encode in batches of {chunk:3d} pages: {(t1 - t0) * 1e3:7.1f} ms (host time inside the calls {sum(te) * 1e3:.1f} ms), decode {(t2 - t1) * 1e3:.1f} ms, "
              f"total {(t2 - t0) * 1e3:.1f} ms, hypotheses equal to the one-call run: {same}")


if __name__ == "__main__":
    main()
