"""GPU: crop batcher bit-exact vs Pillow (through the oracle restatement pinned to it), and the whole
detect -> crop -> recognize path behind the OcrEngine surface vs the CPU oracle pipeline."""
import numpy as np
import pytest

from marie_icr_amd.weights import CRNN_CHARSET, make_craft_state, make_crnn_state, make_page_bgr

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from marie_icr_amd._lib import Context

    c = Context(0)
    yield c
    c.close()


def _rand_crop(rng, h, w, ch=3):
    base = rng.integers(0, 256, size=(h // 3 + 1, w // 3 + 1, ch)).astype(np.float32)
    up = np.repeat(np.repeat(base, 3, 0), 3, 1)[:h, :w]
    a = np.clip(up + rng.integers(-20, 21, size=(h, w, ch)), 0, 255).astype(np.uint8)
    return a if ch == 3 else a[:, :, 0]


@pytest.mark.parametrize("img_w", [256, 100])
def test_crop_batcher_is_pillow_exact(ctx, img_w):
    import torch

    from marie_icr_amd.crnn import pack_fragments
    from oracle import pil_resample as pr

    rng = np.random.default_rng(img_w)
    shapes = [(38, 120), (25, 60), (48, 700), (32, 256), (17, 9), (90, 30), (41, 333), (1, 50), (64, 1), (33, 2000)]
    crops = [_rand_crop(rng, h, w) for h, w in shapes] + [_rand_crop(rng, 29, 77, ch=1)]
    packed, descs = pack_fragments(crops)
    d_in = torch.from_numpy(packed).cuda()
    d_out = torch.empty((len(crops), 32, img_w), dtype=torch.uint8, device="cuda")
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    from marie_icr_amd._lib import check
    import ctypes as C

    check(ctx.h, ctx.lib.mhip_crop_batch(ctx.h, C.c_void_p(d_in.data_ptr()), descs, len(crops), img_w,
                                         C.c_void_p(d_out.data_ptr())), "mhip_crop_batch")
    torch.cuda.synchronize()
    ref = pr.align_collate_pil(crops, img_w)          # Pillow itself
    assert np.array_equal(d_out.cpu().numpy(), ref)
    ctx.set_stream(None)


def test_engine_full_page_vs_oracle(ctx):
    from marie_icr_amd.box_processor import PSMode
    from marie_icr_amd.craft import BoxProcessorCraft
    from marie_icr_amd.crnn import CrnnOcrProcessor
    from marie_icr_amd.ocr_engine import CoordinateFormat, MarieHipOcrEngine
    from oracle import craft_ref, crnn_numpy
    from oracle import pil_resample as pr

    dst, rst = make_craft_state(5), make_crnn_state(0)
    pages = [make_page_bgr(5, 300, 240), make_page_bgr(6, 260, 330)]
    bp = BoxProcessorCraft(state=dst, precision="f32", ctx=ctx)
    rec = CrnnOcrProcessor(state=rst, precision="f32", img_w=128, ctx=ctx)
    eng = MarieHipOcrEngine(box_processor=bp, default_ocr_processor=rec)
    results = eng.extract(pages, PSMode.SPARSE, CoordinateFormat.XYWH)
    assert len(results) == 2
    for pi, (page, res) in enumerate(zip(pages, results)):
        assert res["meta"]["page"] == pi and res["meta"]["format"] == "xywh"
        assert res["meta"]["imageSize"] == {"width": page.shape[1], "height": page.shape[0]}
        # ---- oracle pipeline on the same page
        rects, _ = craft_ref.detect_page(page, dst)
        frags = craft_ref.crop_fragments(page, rects)
        words = res["words"]
        assert len(words) == len(rects)
        if len(rects) == 0:
            continue
        crops = pr.align_collate_u8(frags, 128)
        logits, idx, texts, conf = crnn_numpy.recognize_crops_u8(crops, rst, CRNN_CHARSET)
        order = np.argsort(rects[:, 0])                  # OcrProcessor.recognize orders words by x
        s = np.sort(logits, axis=2)
        safe = ((s[:, :, -1] - s[:, :, -2]) > 2e-3).all(axis=1)
        for wi, (word, oi) in enumerate(zip(words, order)):
            assert word["id"] == wi and word["line"] == -1
            assert list(word["box"]) == rects[oi].tolist()
            if safe[oi]:
                assert word["text"] == texts[oi]
                assert abs(word["confidence"] - round(float(conf[oi]), 3)) <= 2e-3
        # one line group (every word has line -1), text = words joined in x order
        assert len(res["lines"]) == 1
        assert res["lines"][0]["wordids"] == list(range(len(words)))
        if safe.all():
            assert res["lines"][0]["text"] == " ".join(texts[oi] for oi in order)
    # XYXY conversion and the region path
    r2 = eng.extract(pages[:1], PSMode.SPARSE, CoordinateFormat.XYXY)
    for a, b in zip(results[0]["words"], r2[0]["words"]):
        x, y, w, h = a["box"]
        assert list(b["box"]) == [x, y, x + w, y + h]
    reg = eng.extract(pages[:1], PSMode.RAW_LINE, CoordinateFormat.XYWH,
                      regions=[{"id": 7, "pageIndex": 0, "x": 10, "y": 20, "w": 120, "h": 30},
                               {"id": 8, "pageIndex": 0, "x": 0, "y": 0, "w": 0, "h": 10}])
    # reference quirk kept: a rejected region is reported at once AND again when the word count no longer matches
    # the region count, which also blanks the surviving region (ocr_engine.py:276-283,375-389)
    ids = sorted(r["id"] for r in reg["regions"])
    assert ids == [7, 8, 8] and len(reg["extended"]) == 1
    ok = eng.extract(pages[:1], PSMode.RAW_LINE, CoordinateFormat.XYWH,
                     regions=[{"id": 7, "pageIndex": 0, "x": 10, "y": 20, "w": 120, "h": 30}])
    assert [r["id"] for r in ok["regions"]] == [7] and isinstance(ok["regions"][0]["text"], str)


def test_engine_with_dit_detector_and_trocr_recognizer(ctx):
    """The production wiring (BoxProcessorUlimDit + TrOcrProcessor) behind OcrEngine.extract: structure of the result,
    word/box/line bookkeeping, and text = what the recognizer returns for each fragment of the box processor."""
    from marie_icr_amd.box_processor import PSMode
    from marie_icr_amd.dit import default_config as dit_config
    from marie_icr_amd.dit_box_processor import BoxProcessorUlimDit
    from marie_icr_amd.ocr_engine import CoordinateFormat, MarieHipOcrEngine
    from marie_icr_amd.trocr import TrOcrProcessor, default_config as trocr_config
    from marie_icr_amd.weights import make_dit_state, make_image_u8, make_trocr_state

    dcfg = dit_config(ctx.lib, "base")
    dcfg.min_size_test, dcfg.max_size_test = 160, 400
    enc, dec = (256, 2, 4), (256, 2, 4, 512)
    tcfg = trocr_config(ctx.lib, "base")
    tcfg.enc_dim, tcfg.enc_depth, tcfg.enc_heads = enc
    tcfg.dec_dim, tcfg.dec_layers, tcfg.dec_heads, tcfg.dec_ffn = dec
    tcfg.vocab, tcfg.max_positions, tcfg.max_len_b = 97, 32, 8
    box = BoxProcessorUlimDit(cuda=True, state=make_dit_state(0), model="base", precision="f16", ctx=ctx, config=dcfg,
                              refinement=False)
    rec = TrOcrProcessor(state=make_trocr_state(0, enc, dec, 97, 32), config=tcfg, precision="f16", ctx=ctx)
    eng = MarieHipOcrEngine(box_processor=box, default_ocr_processor=rec)
    page = make_image_u8(11, 1, 330, 255)[0]
    res = eng.extract([page], PSMode.SPARSE, CoordinateFormat.XYWH)
    assert len(res) == 1
    words, lines = res[0]["words"], res[0]["lines"]
    rects, frags, numbers, _, line_boxes = box.extract_bounding_boxes("t", "k", page, PSMode.SPARSE)
    assert len(words) == len(rects) > 5
    assert sorted(tuple(int(v) for v in w["box"]) for w in words) == sorted(tuple(int(v) for v in r) for r in rects)
    texts = {r["text"] for r in rec.recognize_from_fragments(frags)}
    assert {w["text"] for w in words} <= texts
    assert all(0.0 <= w["confidence"] <= 1.0 for w in words)
    assert {w["line"] for w in words} == {ln["line"] for ln in lines}
    assert len(line_boxes) >= 1


def test_voting_engine_two_recognizers_one_detection(ctx):
    """MarieHipVotingOcrEngine: CRAFT boxes once, CRNN (CTC) and the production ICR (Attn) recognizers both read them,
    words are voted per box."""
    from marie_icr_amd.box_processor import PSMode
    from marie_icr_amd.craft import BoxProcessorCraft
    from marie_icr_amd.crnn import CrnnOcrProcessor
    from marie_icr_amd.icr import CraftOcrProcessor
    from marie_icr_amd.ocr_engine import CoordinateFormat, MarieHipVotingOcrEngine
    from marie_icr_amd.weights import make_craft_state, make_crnn_state, make_icr_state, make_page_bgr

    calls = []
    box = BoxProcessorCraft(state=make_craft_state(5), precision="f32", ctx=ctx)
    inner = box.extract_bounding_boxes
    box.extract_bounding_boxes = lambda *a, **k: (calls.append(1), inner(*a, **k))[1]
    a = CrnnOcrProcessor(state=make_crnn_state(0), precision="f32", img_w=128, ctx=ctx)
    b = CraftOcrProcessor(state=make_icr_state(0), precision="f32", ctx=ctx)
    eng = MarieHipVotingOcrEngine(box_processor=box, default_ocr_processor=a, processors={"craft": b})
    page = make_page_bgr(5, 300, 240)
    res = eng.extract([page], PSMode.SPARSE, CoordinateFormat.XYWH)
    assert len(calls) == 1, "the detector must run once for both recognizers"
    words = res[0]["words"]
    assert len(words) == 5
    ra = {str(w["id"]): w for w in MarieHipVotingOcrEngine(box_processor=box, default_ocr_processor=a).extract([page], PSMode.SPARSE, CoordinateFormat.XYWH)[0]["words"]}
    for w in words:
        assert w["strategy"]["type"] in ("voting", "default", "confidence")
        assert w["processor"] in ("default", "craft")
        if w["strategy"]["type"] == "default":
            assert w["text"] == ra[w["id"]]["text"]


def test_mixed_dpi_pages_interleaved_through_ingest_and_engine(ctx):
    """BASELINE configs[4] as a parity case: pages scanned at 150 / 200 / 300 DPI (sizes 1 : 4/3 : 2) interleaved in one
    call, one of them over the page-size limit.  ``ensure_max_page_size`` clamps only that one; the engine's result for
    every page equals the result of that page processed alone (no cross-page state in the detector batch, the crop
    batcher or the recognizer's decoding batch), and the recognizer sees crops of all three scales in one batch."""
    from marie_icr_amd import ingest
    from marie_icr_amd.box_processor import PSMode
    from marie_icr_amd.dit import default_config as dit_config
    from marie_icr_amd.dit_box_processor import BoxProcessorUlimDit
    from marie_icr_amd.ocr_engine import CoordinateFormat, MarieHipOcrEngine
    from marie_icr_amd.trocr import TrOcrProcessor, default_config as trocr_config
    from marie_icr_amd.weights import make_dit_state, make_image_u8, make_trocr_state
    from oracle import ingest_ref

    dcfg = dit_config(ctx.lib, "base")
    dcfg.min_size_test, dcfg.max_size_test = 160, 400
    enc, dec = (256, 2, 4), (256, 2, 4, 512)
    tcfg = trocr_config(ctx.lib, "base")
    tcfg.enc_dim, tcfg.enc_depth, tcfg.enc_heads = enc
    tcfg.dec_dim, tcfg.dec_layers, tcfg.dec_heads, tcfg.dec_ffn = dec
    tcfg.vocab, tcfg.max_positions, tcfg.max_len_b = 97, 32, 8
    box = BoxProcessorUlimDit(cuda=True, state=make_dit_state(0), model="base", precision="f16", ctx=ctx, config=dcfg,
                              refinement=False)
    rec = TrOcrProcessor(state=make_trocr_state(0, enc, dec, 97, 32), config=tcfg, precision="f16", ctx=ctx)
    eng = MarieHipOcrEngine(box_processor=box, default_ocr_processor=rec)
    sizes = [(200, 160), (266, 213), (400, 320), (266, 213), (200, 160), (520, 400)]       # the last one is oversized
    frames = [make_image_u8(40 + i, 1, h, w)[0] for i, (h, w) in enumerate(sizes)]
    limit = (320, 400)                                                                     # (width, height) portrait
    changed, clamped = ingest.ensure_max_page_size(frames, max_page_size=limit, expand_ratio=0.0, ctx=ctx)
    assert changed is True
    assert all(c is f for c, f in zip(clamped[:5], frames[:5]))
    ref_changed, ref = ingest_ref.ensure_max_page_size(frames, limit, 0.0)
    assert ref_changed and clamped[5].shape == ref[5].shape == (400, 307, 3) and np.array_equal(clamped[5], ref[5])
    together = eng.extract(clamped, PSMode.SPARSE, CoordinateFormat.XYWH)
    assert len(together) == len(clamped)
    for i, page in enumerate(clamped):
        alone = eng.extract([page], PSMode.SPARSE, CoordinateFormat.XYWH)[0]
        a = [(tuple(int(v) for v in w["box"]), w["text"], w["line"]) for w in alone["words"]]
        t = [(tuple(int(v) for v in w["box"]), w["text"], w["line"]) for w in together[i]["words"]]
        assert a == t and len(a) > 0, i
        assert np.allclose([w["confidence"] for w in alone["words"]], [w["confidence"] for w in together[i]["words"]], atol=2e-3)
    assert together[0]["meta"]["imageSize"]["width"] == 160 and together[2]["meta"]["imageSize"]["width"] == 320


def test_engine_batched_path_equals_per_page_loop(ctx):
    """MarieHipOcrEngine.extract batches pages through the detector and pools their fragments into one recognizer batch
    (two contexts -> detector of batch k+1 overlaps the recognizer of batch k).  Every page's result must be what the
    reference's per-page loop (ocr_engine.py:172-221) gives: refinement passes on, a blank page and three page sizes in one call; fragments travel as device windows in the batched path and as host arrays in the loop."""
    from marie_icr_amd._lib import Context
    from marie_icr_amd.box_processor import PSMode
    from marie_icr_amd.dit import default_config as dit_config
    from marie_icr_amd.dit_box_processor import BoxProcessorUlimDit
    from marie_icr_amd.fragments import FragmentList
    from marie_icr_amd.ocr_engine import CoordinateFormat, MarieHipOcrEngine
    from marie_icr_amd.trocr import TrOcrProcessor, default_config as trocr_config
    from marie_icr_amd.weights import make_dit_state, make_image_u8, make_trocr_state

    ctx2 = Context(0)
    dcfg = dit_config(ctx.lib, "base")
    dcfg.min_size_test, dcfg.max_size_test = 160, 400
    enc, dec = (256, 2, 4), (256, 2, 4, 512)
    tcfg = trocr_config(ctx.lib, "base")
    tcfg.enc_dim, tcfg.enc_depth, tcfg.enc_heads = enc
    tcfg.dec_dim, tcfg.dec_layers, tcfg.dec_heads, tcfg.dec_ffn = dec
    tcfg.vocab, tcfg.max_positions, tcfg.max_len_b = 97, 32, 8
    box = BoxProcessorUlimDit(cuda=True, state=make_dit_state(0), model="base", precision="f16", ctx=ctx, config=dcfg,
                              refinement=True, det_batch=2)
    rec = TrOcrProcessor(state=make_trocr_state(0, enc, dec, 97, 32), config=tcfg, precision="f16", ctx=ctx2, batch_size=64)
    eng = MarieHipOcrEngine(box_processor=box, default_ocr_processor=rec)
    eng.page_batch = 2
    sizes = [(330, 255), (248, 192), (330, 255), (412, 318), (330, 255)]
    frames = [make_image_u8(60 + i, 1, h, w)[0] for i, (h, w) in enumerate(sizes)]
    frames.append(np.full((330, 255, 3), 255, np.uint8))                       # a blank page (the random-weight detector still emits boxes)
    # a page smaller than MIN_SIZE_TEST is framed on a canvas: its fragments have no device window (and, as in the reference,
    # may be empty where a box lies in the frame — which is why it is not part of the recognizer comparison below)
    small = box.extract_bounding_boxes("q", "k", make_image_u8(63, 1, 100, 120)[0], PSMode.SPARSE)
    assert isinstance(small[1], FragmentList) and small[1].windows is None
    got = eng.extract(frames, PSMode.SPARSE, CoordinateFormat.XYXY)
    assert len(got) == len(frames)
    for i, page in enumerate(frames):
        rects, frags, numbers, _, line_boxes = box.extract_bounding_boxes("q", "k", page, PSMode.SPARSE)
        assert isinstance(frags, FragmentList) and frags.windows is not None
        ref, _ = rec.recognize("q", "k", page, rects, list(frags), numbers)                   # plain list: packed + uploaded
        ref = MarieHipOcrEngine._finish_page(ref, i, numbers, line_boxes, CoordinateFormat.XYXY)
        a = [(tuple(int(v) for v in w["box"]), w["text"], int(w["line"]), w["id"], w["word_index"]) for w in ref["words"]]
        t = [(tuple(int(v) for v in w["box"]), w["text"], int(w["line"]), w["id"], w["word_index"]) for w in got[i]["words"]]
        assert a == t, i
        assert np.allclose([w["confidence"] for w in ref["words"]], [w["confidence"] for w in got[i]["words"]], atol=2e-3)
        assert [(ln["line"], ln["text"], ln["wordids"]) for ln in ref["lines"]] == \
               [(ln["line"], ln["text"], ln["wordids"]) for ln in got[i]["lines"]]
        assert got[i]["meta"]["page"] == i and got[i]["meta"]["format"] == "xyxy"
    ctx2.close()


def test_example_script_writes_all_formats(tmp_path):
    """examples/ocr_pages.py end to end on two synthetic pages (the small CRAFT + CRNN pair): results.json, the text file, one BLOBS
    and one Adlib file per page + the summary, and a JSON status line."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = tmp_path / "out"
    r = subprocess.run([sys.executable, os.path.join(root, "examples", "ocr_pages.py"), "--out", str(out), "--synthetic", "2", "--engine",
                        "craft_crnn", "--precision", "f32"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    status = json.loads(r.stdout.strip().splitlines()[-1])
    assert status["pages"] == 2 and status["words"] > 0
    results = json.load(open(out / "results.json"))
    assert len(results) == 2 and all(set(p) >= {"meta", "words", "lines"} for p in results)
    assert sum(len(p["words"]) for p in results) == status["words"]
    assert os.path.getsize(out / "results.txt") > 0
    assert sorted(os.listdir(out / "blobs")) == ["1.BLOBS.XML", "2.BLOBS.XML"]
    assert sorted(os.listdir(out / "adlib")) == ["1.tif.xml", "2.tif.xml", "summary.xml"]
