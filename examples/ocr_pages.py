#!/usr/bin/env python3
"""In-process page OCR with the MI355X path behind the reference's engine surface — what the reference's executor does with a
document (marie/executor/text/text_extraction_executor.py -> marie/pipe/components.py:620-651 ``ocr_frames`` + the renderers), without
the service around it:

    frames  = load_image / ensure_max_page_size            (marie/utils/docs.py, image_utils.py)
    results = engine.extract(frames, PSMode.SPARSE, CoordinateFormat.XYWH)
    results.json, <name>.txt (TextRenderer), <n>.BLOBS.XML (BlobRenderer), <n>.tif.xml + summary.xml (AdlibRenderer)

    python examples/ocr_pages.py --out /tmp/out page1.png scan.tif            # images / multi-page TIFFs
    python examples/ocr_pages.py --out /tmp/out --synthetic 4                 # seeded 2550x3300 pages, random weights (plumbing)
    python examples/ocr_pages.py --out /tmp/out --models-dir /opt/model_zoo scan.tif   # the reference's checkpoints (DiT + TrOCR)

Without --models-dir the models carry seeded random weights (there are no checkpoints in this tree): boxes and texts are noise,
the plumbing, formats and timings are real.  ``--engine craft_crnn`` uses the small CRAFT + CRNN pair (fast), the default
``dit_trocr`` is BASELINE configs[2].
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def build_engine(args):
    from marie_icr_amd._lib import Context
    from marie_icr_amd.ocr_engine import MarieHipOcrEngine

    ctx = Context(args.device)
    if args.engine == "craft_crnn":
        from marie_icr_amd.craft import BoxProcessorCraft
        from marie_icr_amd.crnn import CrnnOcrProcessor
        from marie_icr_amd.weights import make_craft_state, make_crnn_state

        box = BoxProcessorCraft(state=make_craft_state(5), precision=args.precision, ctx=ctx)
        rec = CrnnOcrProcessor(state=make_crnn_state(0), precision=args.precision, img_w=256, ctx=ctx)
        return MarieHipOcrEngine(box_processor=box, default_ocr_processor=rec)
    from marie_icr_amd.dit_box_processor import BoxProcessorUlimDit
    from marie_icr_amd.trocr import TrOcrProcessor

    if args.models_dir:
        box = BoxProcessorUlimDit(models_dir=args.models_dir, cuda=True, refinement=not args.no_refinement, precision=args.precision, ctx=ctx)
        rec = TrOcrProcessor(model_name_or_path=os.path.join(args.models_dir, "trocr/trocr-large-printed.pt"), cuda=True, model="large",
                             dict_path=args.dict_path, encoder_json=args.encoder_json, precision=args.precision)
    else:
        from marie_icr_amd.dit import DitModel
        from marie_icr_amd._lib import PREC_F16, PREC_F32
        from marie_icr_amd.trocr import TrocrModel, default_config
        from marie_icr_amd.weights import make_dit_state, make_trocr_state

        prec = PREC_F16 if args.precision == "f16" else PREC_F32
        det = DitModel(ctx, make_dit_state(0), model="base", precision=prec)
        tcfg = default_config(ctx.lib, "base")
        tcfg.max_len_b = args.decode_len
        rctx = Context(args.device)
        trocr = TrocrModel(rctx, make_trocr_state(0, (tcfg.enc_dim, tcfg.enc_depth, tcfg.enc_heads),
                                                  (tcfg.dec_dim, tcfg.dec_layers, tcfg.dec_heads, tcfg.dec_ffn), tcfg.vocab, tcfg.max_positions),
                           tcfg, prec)
        box = BoxProcessorUlimDit(cuda=True, refinement=not args.no_refinement, dit_model=det)
        rec = TrOcrProcessor(trocr_model=trocr)
    return MarieHipOcrEngine(box_processor=box, default_ocr_processor=rec)


def load_frames(args):
    from marie_icr_amd.ingest import ensure_max_page_size, frames_from_file
    from marie_icr_amd.weights import make_page_bgr

    frames = []
    for path in args.files:
        frames.extend(f[:, :, ::-1].copy() for f in frames_from_file(path))          # RGB frames -> the engine's BGR
    for i in range(args.synthetic):
        h, w = (3300, 2550) if args.engine == "dit_trocr" else (600, 480)
        frames.append(make_page_bgr(1000 + i, h, w))
    if not frames:
        raise SystemExit("nothing to do: give image files or --synthetic N")
    changed, frames = ensure_max_page_size(frames)
    return frames


def _plain(v):
    if isinstance(v, dict):
        return {k: _plain(x) for k, x in v.items()}
    if isinstance(v, (list, tuple, np.ndarray)):
        return [_plain(x) for x in v]
    if isinstance(v, np.generic):
        return v.item()
    return v


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("files", nargs="*", help="images or multi-page TIFFs")
    ap.add_argument("--out", required=True, help="output directory")
    ap.add_argument("--synthetic", type=int, default=0, help="add N seeded synthetic pages")
    ap.add_argument("--engine", choices=("dit_trocr", "craft_crnn"), default="dit_trocr")
    ap.add_argument("--models-dir", default=None, help="the reference's model zoo root (DiT + TrOCR checkpoints)")
    ap.add_argument("--dict-path", default=None)
    ap.add_argument("--encoder-json", default=None)
    ap.add_argument("--precision", choices=("f16", "f32"), default="f16")
    ap.add_argument("--decode-len", type=int, default=15, help="max_len_b of the seeded TrOCR model")
    ap.add_argument("--no-refinement", action="store_true", help="one detector pass instead of the reference's three")
    ap.add_argument("--crop-to-content", action="store_true")
    ap.add_argument("--device", type=int, default=0)
    args = ap.parse_args(argv)

    from marie_icr_amd.box_processor import PSMode
    from marie_icr_amd.ocr_engine import CoordinateFormat
    from marie_icr_amd.renderer import AdlibRenderer, BlobRenderer, TextRenderer

    os.makedirs(args.out, exist_ok=True)
    frames = load_frames(args)
    engine = build_engine(args)
    t0 = time.perf_counter()
    results = engine.extract(frames, PSMode.SPARSE, CoordinateFormat.XYWH, crop_to_content=args.crop_to_content)
    dt = time.perf_counter() - t0
    with open(os.path.join(args.out, "results.json"), "w", encoding="utf-8") as f:
        json.dump(_plain(results), f)
    TextRenderer(config={"preserve_interword_spaces": True}).render(frames, results, os.path.join(args.out, "results.txt"))
    for sub, renderer in (("blobs", BlobRenderer()), ("adlib", AdlibRenderer())):
        os.makedirs(os.path.join(args.out, sub), exist_ok=True)
        renderer.render(frames, results, os.path.join(args.out, sub))
    words = sum(len(r["words"]) for r in results)
    print(json.dumps({"pages": len(frames), "words": words, "seconds": round(dt, 3), "pages_per_s": round(len(frames) / dt, 2),
                      "out": args.out}))
    return 0


if __name__ == "__main__":
    sys.exit(main())
