"""Generate golden vectors by running the REFERENCE's own recognizer code.

Run only in the build container (``/root/reference`` does not travel to the GPU
box):  ``python oracle/gen_golden.py``  → writes ``tests/golden/crnn_*.npz``.

What is imported from the reference, unmodified, by path:
``marie/models/icr/model.py`` (``Model``) and ``marie/models/icr/utils.py``
(``CTCLabelConverter``).  The fixtures hold DATA only: input crops, logits,
argmax indices, decoded strings, confidences, and the sha256 of the weight set
(the weights themselves are rebuilt from the seed by ``marie_icr_amd.weights``
or ``oracle.crnn_torch.default_init_state``).
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True
REF_ICR = "/root/reference/marie/models/icr"

from marie_icr_amd.weights import CRNN_CHARSET, make_crnn_input, make_crnn_state, state_checksum  # noqa: E402
from oracle.crnn_torch import default_init_state  # noqa: E402


class _Opt:
    pass


def _ref_model(img_w: int, state=None, seed=None):
    sys.path.insert(0, REF_ICR)
    from model import Model  # reference: marie/models/icr/model.py:25
    from utils import CTCLabelConverter  # reference: marie/models/icr/utils.py:7

    opt = _Opt()
    opt.Transformation, opt.FeatureExtraction = "None", "VGG"
    opt.SequenceModeling, opt.Prediction = "BiLSTM", "CTC"
    opt.imgH, opt.imgW, opt.num_fiducial = 32, img_w, 20
    opt.input_channel, opt.output_channel, opt.hidden_size = 1, 512, 256
    opt.batch_max_length = 48
    conv = CTCLabelConverter(CRNN_CHARSET)
    opt.num_class = len(conv.character)
    if seed is not None:
        torch.manual_seed(seed)
    m = Model(opt).eval()
    if state is not None:
        m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in state.items()})
    return m, conv


def _run(m, conv, crops_u8):
    # pre-processing as the reference's NormalizePAD does for a full-width crop
    # (marie/models/icr/dataset.py:275-283): ToTensor (/255) then sub 0.5 div 0.5
    x = torch.from_numpy(crops_u8).float().div(255).sub_(0.5).div_(0.5).unsqueeze(1)
    with torch.no_grad():
        preds = m(x, None)
        # decode exactly as marie/document/craft_ocr_processor.py:236-272
        preds_size = torch.IntTensor([preds.size(1)] * preds.size(0))
        _, idx = preds.max(2)
        strs = conv.decode(idx, preds_size)
        prob = F.softmax(preds, dim=2)
        pmax, _ = prob.max(dim=2)
        conf = torch.stack([p.cumprod(dim=0)[-1] for p in pmax])
    return (preds.numpy(), idx.numpy().astype(np.int32),
            np.array([s.upper() for s in strs]), conf.numpy())


def _install_torchvision_standin():
    """torchvision is not installed here and not vendored by the reference.  The reference only needs
    ``torchvision.models.vgg16_bn(pretrained).features`` (marie/models/craft/basenet/vgg16_bn.py:27); this is
    that published layer list (VGG cfg "D" with BatchNorm: conv3x3 p1 / BatchNorm2d / ReLU(inplace=True), "M" =
    MaxPool2d(2, 2)) — a restatement of the absent third-party module, recorded as such in DESIGN.md."""
    import types

    import torch.nn as nn

    cfg_d = [64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512, "M"]

    class _VGG(nn.Module):
        def __init__(self):
            super().__init__()
            layers, cin = [], 3
            for v in cfg_d:
                if v == "M":
                    layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
                else:
                    layers += [nn.Conv2d(cin, v, kernel_size=3, padding=1), nn.BatchNorm2d(v), nn.ReLU(inplace=True)]
                    cin = v
            self.features = nn.Sequential(*layers)

    tv = types.ModuleType("torchvision")
    models = types.ModuleType("torchvision.models")
    models.vgg16_bn = lambda pretrained=False, **kw: _VGG()
    tv.models = models
    sys.modules["torchvision"] = tv
    sys.modules["torchvision.models"] = models


def gen_craft(out_dir):
    """CRAFT network goldens: the reference's unmodified CRAFT class on seeded weights."""
    from marie_icr_amd.weights import make_craft_state, make_page_bgr
    from oracle.craft_ref import craft_preprocess

    _install_torchvision_standin()
    sys.path.insert(0, "/root/reference/marie/models/craft")
    from craft import CRAFT  # reference: marie/models/craft/craft.py:31

    for tag, seed, (h, w) in (("a", 0, (130, 170)), ("b", 1, (210, 160))):
        st = make_craft_state(seed)
        net = CRAFT(pretrained=False).eval()
        missing = net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in st.items()}, strict=True)
        page = make_page_bgr(seed, h, w)
        x, ratio, (th, tw) = craft_preprocess(page, canvas_size=w, mag_ratio=1.0)
        with torch.no_grad():
            y, feature = net(torch.from_numpy(x))
        np.savez_compressed(
            os.path.join(out_dir, f"craft_net_{tag}.npz"),
            weight_seed=seed, weight_sha256=state_checksum(st), page_seed=seed, page_hw=np.array([h, w]),
            x_shape=np.array(x.shape), y=y.numpy(), feature_sub=feature.numpy()[:, ::4, ::3, ::3])
        print("craft", tag, tuple(x.shape), "->", tuple(y.shape), "text range", float(y[..., 0].min()),
              float(y[..., 0].max()), "link range", float(y[..., 1].min()), float(y[..., 1].max()), missing)


def gen_icr(out_dir):
    """Production recognizer TPS-ResNet-BiLSTM-Attn: the reference's unmodified Model(opt) + AttnLabelConverter and
    the decode/confidence lines of marie/document/craft_ocr_processor.py:244-272."""
    from marie_icr_amd.weights import make_icr_state

    sys.path.insert(0, REF_ICR)
    from model import Model
    from utils import AttnLabelConverter

    opt = _Opt()
    opt.Transformation, opt.FeatureExtraction, opt.SequenceModeling, opt.Prediction = "TPS", "ResNet", "BiLSTM", "Attn"
    opt.imgH, opt.imgW, opt.num_fiducial = 32, 100, 20
    opt.input_channel, opt.output_channel, opt.hidden_size, opt.batch_max_length = 1, 512, 256, 48
    conv = AttnLabelConverter(CRNN_CHARSET)
    opt.num_class = len(conv.character)
    for tag, wseed, iseed, n in (("a", 0, 0, 6), ("b", 1, 5, 3)):
        st = make_icr_state(wseed)
        m = Model(opt).eval()
        m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in st.items()})
        crops = make_crnn_input(iseed, n, 32, 100)
        x = torch.from_numpy(crops).float().div(255).sub_(0.5).div_(0.5).unsqueeze(1)
        with torch.no_grad():
            rect = m.Transformation(x)
            preds = m(x, torch.zeros(n, 49, dtype=torch.long), is_train=False)
            _, idx = preds.max(2)
            strs = conv.decode(idx, torch.IntTensor([48] * n))
            pmax, _ = F.softmax(preds, dim=2).max(dim=2)
        texts, confs = [], []
        for pred, pm in zip(strs, pmax):
            eos = pred.find("[s]")
            pred, pm = pred[:eos], pm[:eos]
            if pm.numel() == 0:            # the reference raises here; recorded as ("", 0)
                texts.append("")
                confs.append(0.0)
            else:
                texts.append(pred.upper())
                confs.append(float(pm.cumprod(dim=0)[-1]))
        np.savez_compressed(
            os.path.join(out_dir, f"icr_attn_{tag}.npz"),
            weight_seed=wseed, input_seed=iseed, weight_sha256=state_checksum(st), crops_u8=crops,
            rectified=rect.numpy(), logits=preds.numpy(), argmax=idx.numpy().astype(np.int32),
            strings=np.array(texts), confidence=np.array(confs, np.float32))
        print("icr", tag, tuple(preds.shape), "max|logit|", float(preds.abs().max()), texts[:3], confs[:3])


def _load_ref_geometry():
    """overlap.py (numpy only) and line_processor.py, unmodified, by path.  line_processor's
    cv2 / PIL / logger imports are only touched when ``enable_visualization`` is set; placeholders satisfy the import."""
    import importlib.util
    import logging
    import types

    def load(name, path):
        spec = importlib.util.spec_from_file_location(name, path)
        mod = importlib.util.module_from_spec(spec)
        sys.modules[name] = mod
        spec.loader.exec_module(mod)
        return mod

    for name in ("cv2", "marie", "marie.logging_core", "marie.logging_core.predefined", "marie.utils"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["marie.logging_core.predefined"].default_logger = logging.getLogger("ref")
    ov = load("marie.utils.overlap", "/root/reference/marie/utils/overlap.py")
    lp = load("ref_line_processor", "/root/reference/marie/boxes/line_processor.py")
    return ov, lp


def geometry_cases():
    """Seeded word-box layouts: text lines with jitter, split words, skew, two columns, overlaps, single boxes."""
    cases = []
    for seed in range(24):
        rng = np.random.default_rng(100 + seed)
        n_lines = int(rng.integers(1, 28))
        skew = float(rng.uniform(-0.01, 0.01)) if seed % 3 == 0 else 0.0
        two_col = seed % 4 == 1
        boxes = []
        y = float(rng.uniform(40, 200))
        for _ in range(n_lines):
            hgt = float(rng.uniform(24, 60))
            for x0, x_end in (((100.0, 1150.0), (1350.0, 2400.0)) if two_col else ((120.0, 2400.0),)):
                x = x0 + float(rng.uniform(0, 80))
                while x < x_end - 60:
                    w = float(rng.uniform(30, 320))
                    gap = float(rng.uniform(-6, 40))
                    yy = y + skew * x + float(rng.uniform(-0.25, 0.25)) * hgt
                    hh = hgt * float(rng.uniform(0.7, 1.2))
                    boxes.append([x, yy, min(x + w, x_end), yy + hh])
                    x += w + gap
            y += hgt * float(rng.uniform(1.1, 2.2))
        b = np.asarray(boxes, np.float32)
        rng.shuffle(b, axis=0)
        cases.append(b)
    return cases


def gen_geometry(out_dir):
    ov, lp = _load_ref_geometry()
    out = {}
    for k, b in enumerate(geometry_cases()):
        merged = np.asarray(ov.merge_boxes([row for row in b], 0.08), np.float32).reshape(-1, 4)
        # word boxes -> (x, y, w, h) int32 as the box processor hands them on (ulim_dit_box_processor.py:757-763)
        bi = b.astype(np.int32)
        xywh = np.stack([bi[:, 0], bi[:, 1], bi[:, 2] - bi[:, 0], bi[:, 3] - bi[:, 1]], 1)
        img = np.zeros((3300, 2550), np.uint8)
        lines_platform = np.asarray(lp.line_merge(img, [list(r) for r in xywh]), np.int64).reshape(-1, 4)
        # The reference sorts by y with numpy's default (unstable, CPU-dispatched) argsort, so boxes with equal y come
        # out in a platform-dependent order.  Second run of the same reference code with that one numpy call forced
        # to kind="stable": this is the tie rule our build fixes, and the vector the oracle is held to everywhere.
        plain = np.argsort
        np.argsort = lambda a, *args, **kw: plain(a, kind="stable")
        try:
            lines = np.asarray(lp.line_merge(img, [list(r) for r in xywh]), np.int64).reshape(-1, 4)
        finally:
            np.argsort = plain
        out[f"lines_platform_{k}"] = lines_platform.astype(np.int32)
        nums = np.asarray([lp.find_line_number(lines, list(r)) for r in xywh], np.int32)
        blocks = np.asarray([ov.merge_bboxes_as_block(xywh[: 1 + (j % len(xywh))]) for j in range(0, len(xywh), 7)], np.int64)
        out[f"boxes_{k}"] = b
        out[f"merged_{k}"] = merged
        out[f"lines_{k}"] = lines.astype(np.int32)
        out[f"linenum_{k}"] = nums
        out[f"blocks_{k}"] = blocks.astype(np.int32)
        print("geometry", k, len(b), "->", len(merged), "merged,", len(lines), "lines")
    out["n_cases"] = np.int32(len(geometry_cases()))
    np.savez_compressed(os.path.join(out_dir, "geometry.npz"), **out)


def _load_ref_beit():
    """beit.py, unmodified, by path; its ``from timm.models.layers import drop_path, to_2tuple, trunc_normal_`` is
    satisfied by three pass-through helpers — none of them takes part in an eval forward."""
    import importlib.util
    import types

    for name in ("timm", "timm.models", "timm.models.layers"):
        sys.modules.setdefault(name, types.ModuleType(name))
    lay = sys.modules["timm.models.layers"]
    lay.drop_path = lambda x, p=0.0, training=False: x
    lay.to_2tuple = lambda v: tuple(v) if isinstance(v, (tuple, list)) else (v, v)
    lay.trunc_normal_ = lambda t, std=1.0: torch.nn.init.trunc_normal_(t, std=std)
    spec = importlib.util.spec_from_file_location("ref_beit", "/root/reference/marie/boxes/dit/ditod/beit.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


VIT_GOLDEN_CASES = (
    # tag, dim, depth, heads, taps, (th, tw) image, (H32, W32) canvas, weight seed, image seed, batch
    ("small", 256, 4, 4, (0, 1, 2, 3), (90, 60), (96, 64), 0, 0, 2),
    ("base", 768, 12, 12, (3, 5, 7, 11), (120, 96), (128, 96), 1, 1, 1),
)


def gen_vit(out_dir):
    from functools import partial

    from marie_icr_amd.weights import make_image_u8, make_vit_state
    from oracle.vit_torch import TorchVitOracle

    beit = _load_ref_beit()
    for tag, dim, depth, heads, taps, (th, tw), (H32, W32), wseed, iseed, B in VIT_GOLDEN_CASES:
        st = make_vit_state(wseed, dim, depth, heads)
        m = beit.BEiT(img_size=[224, 224], patch_size=16, embed_dim=dim, depth=depth, num_heads=heads, mlp_ratio=4,
                      qkv_bias=True, norm_layer=partial(torch.nn.LayerNorm, eps=1e-6), init_values=0.1,
                      out_features=[f"layer{t}" for t in taps], drop_path_rate=0.1, use_abs_pos_emb=True,
                      use_checkpoint=False).eval()
        missing, unexpected = m.load_state_dict({k: torch.from_numpy(v) for k, v in st.items()}, strict=False)
        assert not unexpected and all("num_batches_tracked" in k for k in missing), (missing, unexpected)
        imgs = make_image_u8(iseed, B, th, tw)
        x = TorchVitOracle.preprocess(imgs, H32, W32, swap_rb=True)
        with torch.no_grad():
            feats = m.forward_features(x)
        outs = [feats[f"layer{t}"].numpy() for t in taps]
        step = 1 if tag == "small" else 4        # keep the base fixture small: every 4th channel of each tap
        np.savez_compressed(os.path.join(out_dir, f"vit_{tag}.npz"), weight_seed=wseed, image_seed=iseed,
                            weight_sha256=state_checksum(st), dim=dim, depth=depth, heads=heads, taps=np.asarray(taps),
                            image_hw=np.asarray([th, tw]), canvas_hw=np.asarray([H32, W32]), batch=B, channel_step=step,
                            **{f"fpn{j}": o[:, ::step].astype(np.float32) for j, o in enumerate(outs)})
        print("vit", tag, [o.shape for o in outs], [float(np.abs(o).max()) for o in outs])


RENDER_CASES = ((0, 850, 1100, 14), (1, 2550, 3300, 40), (2, 1700, 600, 9), (3, 640, 480, 3), (4, 1275, 1650, 0))


def _load_ref_text_renderer():
    """marie/renderer/renderer.py and text_renderer.py, unmodified, by path.  Their package imports are satisfied by
    placeholders: ``marie.logging_core.logger.MarieLogger`` (a silent logger), ``marie.renderer`` (exposing the
    ``ResultRenderer`` just loaded) and ``marie.utils.types.strtobool`` (not reached: no config key is set)."""
    import importlib.util
    import types

    class _Quiet:
        def __init__(self, *a, **k):
            pass

        def info(self, *a, **k):
            pass

        error = warning = debug = info

    for name in ("marie", "marie.logging_core", "marie.logging_core.logger", "marie.utils", "marie.utils.types", "marie.renderer"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["marie.logging_core.logger"].MarieLogger = _Quiet
    sys.modules["marie.utils.types"].strtobool = lambda v: v if isinstance(v, bool) else str(v).lower() in ("1", "true", "yes", "y", "on")
    base = "/root/reference/marie/renderer/"
    spec = importlib.util.spec_from_file_location("ref_renderer_base", base + "renderer.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    sys.modules["marie.renderer"].ResultRenderer = mod.ResultRenderer
    spec = importlib.util.spec_from_file_location("ref_text_renderer", base + "text_renderer.py")
    tr = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tr)
    return tr.TextRenderer


def gen_renderer(out_dir):
    """Text the reference's own TextRenderer writes for seeded page results (the step after the path)."""
    import contextlib
    import copy
    import io
    import json
    import tempfile

    from marie_icr_amd.weights import make_ocr_result

    TextRenderer = _load_ref_text_renderer()
    cases = []
    with tempfile.TemporaryDirectory() as td:
        def run(frames, results):
            path = os.path.join(td, "out.txt")
            with contextlib.redirect_stdout(io.StringIO()):          # the reference prints its grid while rendering
                TextRenderer(config={}).render(frames, copy.deepcopy(results), path)
            with open(path, encoding="UTF-8") as f:
                return f.read()

        for seed, w, h, n in RENDER_CASES:
            res = make_ocr_result(seed, w, h, n)
            cases.append({"seed": seed, "width": w, "height": h, "n_lines": n,
                          "text": run([np.zeros((h, w, 3), np.uint8)], [res])})
        multi = [make_ocr_result(10 + i, 900, 700, 6, page=i) for i in range(3)]
        doc = run([np.zeros((700, 900, 3), np.uint8)] * 3, multi)
    with open(os.path.join(out_dir, "text_renderer.json"), "w", encoding="UTF-8") as f:
        json.dump({"cases": cases, "multi_page": {"seeds": [10, 11, 12], "width": 900, "height": 700, "n_lines": 6, "text": doc}}, f)
    print("renderer", [len(c["text"]) for c in cases], len(doc))


def _load_ref_xml_renderers():
    """marie/renderer/blob_renderer.py and adlib_renderer.py, unmodified, by path (placeholders as for the text renderer, plus
    ``marie.logging_core.predefined.default_logger`` and ``marie.renderer.renderer``)."""
    import importlib.util
    import types

    _load_ref_text_renderer()                         # registers the placeholders and marie.renderer.ResultRenderer

    class _Quiet:
        def info(self, *a, **k):
            pass

        error = warning = debug = info

    for name in ("marie.logging_core.predefined", "marie.renderer.renderer"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["marie.logging_core.predefined"].default_logger = _Quiet()
    sys.modules["marie.renderer.renderer"].ResultRenderer = sys.modules["marie.renderer"].ResultRenderer
    out = []
    for fname, cls in (("blob_renderer.py", "BlobRenderer"), ("adlib_renderer.py", "AdlibRenderer")):
        spec = importlib.util.spec_from_file_location("ref_" + fname[:-3], "/root/reference/marie/renderer/" + fname)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        out.append(getattr(mod, cls))
    return out


XML_RENDER_CASES = [(21, 1200, 900, 5), (22, 2550, 3300, 12), (23, 640, 480, 1)]


def xml_render_results():
    """The page results the XML renderer goldens are made from (seeded pages + one with characters XML escapes)."""
    from marie_icr_amd.weights import make_ocr_result

    docs = [[make_ocr_result(seed, w, h, n)] for seed, w, h, n in XML_RENDER_CASES]
    docs.append([make_ocr_result(30 + i, 900, 700, 4, page=i) for i in range(3)])
    special = make_ocr_result(40, 800, 600, 2)
    for word, text in zip(special["words"], ["A&B", "<tag>", "\"quoted\"", "it's", "x>y&&y<z", "\u00e9t\u00e9"]):
        word["text"] = text
    docs.append([special])
    return docs


def gen_xml_renderers(out_dir):
    """Files the reference's own BlobRenderer / AdlibRenderer write (the on-disk formats after the path)."""
    import copy
    import json
    import tempfile

    BlobRenderer, AdlibRenderer = _load_ref_xml_renderers()
    out = []
    for results in xml_render_results():
        frames = [np.zeros((r["meta"]["imageSize"]["height"], r["meta"]["imageSize"]["width"], 3), np.uint8) for r in results]
        rec = {}
        for key, cls in (("blob", BlobRenderer), ("adlib", AdlibRenderer)):
            with tempfile.TemporaryDirectory() as td:
                cls(config={}).render(frames, copy.deepcopy(results), td)
                rec[key] = {name: open(os.path.join(td, name), "rb").read().decode("UTF-8") for name in sorted(os.listdir(td))}
        out.append(rec)
    with open(os.path.join(out_dir, "xml_renderers.json"), "w", encoding="UTF-8") as f:
        json.dump({"documents": out}, f)
    print("xml renderers", [sorted(d["blob"]) + sorted(d["adlib"]) for d in out])


def _load_ref_voting_engine():
    """marie/ocr/voting_ocr_engine.py, unmodified, by path.  Its package imports (recognizer classes, PSMode, the engine base
    class, a JSON dump helper) are satisfied by placeholders — none of them is reached by the vote rules: an instance is made
    without ``__init__`` and only ``voting_evaluator`` / ``get_words_by_vote_by_selector`` / ``group_candidates_by_selector``
    run.  ``store_json_object`` (debug dumps under /tmp/marie) is a no-op."""
    import importlib.util
    import types

    class _Quiet:
        def info(self, *a, **k):
            pass

        error = warning = debug = info

    names = {"marie": [], "marie.boxes": ["PSMode"], "marie.boxes.box_processor": ["BoxProcessor"],
             "marie.constants": [], "marie.document": ["TrOcrProcessor"],
             "marie.document.craft_ocr_processor": ["CraftOcrProcessor"],
             "marie.document.lev_ocr_processor": ["LevenshteinOcrProcessor"],
             "marie.document.ocr_processor": ["OcrProcessor"],
             "marie.document.tesseract_ocr_processor": ["TesseractOcrProcessor"],
             "marie.ocr": ["CoordinateFormat", "OcrEngine"], "marie.ocr.ocr_engine": [], "marie.utils": [],
             "marie.utils.json": []}
    saved = {n: sys.modules.get(n) for n in names}
    for n, classes in names.items():
        m = types.ModuleType(n)
        for c in classes:
            setattr(m, c, type(c, (), {}))
        sys.modules[n] = m
    sys.modules["marie.constants"].__model_path__ = "/nonexistent"
    sys.modules["marie.boxes"].PSMode.SPARSE = "sparse"                 # default-argument values of extract(), never used
    sys.modules["marie.ocr"].CoordinateFormat.XYXY = "xyxy"
    sys.modules["marie.ocr.ocr_engine"].reset_bbox_cache = lambda: None
    sys.modules["marie.utils.json"].store_json_object = lambda *a, **k: None
    try:
        spec = importlib.util.spec_from_file_location("ref_voting_ocr_engine", "/root/reference/marie/ocr/voting_ocr_engine.py")
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    finally:
        for n, m in saved.items():
            if m is None:
                sys.modules.pop(n, None)
            else:
                sys.modules[n] = m
    eng = object.__new__(mod.VotingOcrEngine)
    eng.logger = _Quiet()
    return eng


def gen_voting(out_dir):
    """What the reference's own VotingOcrEngine.voting_evaluator returns for seeded recognizer outputs, page mode and region
    mode (marie/ocr/voting_ocr_engine.py:186-482)."""
    import contextlib
    import copy
    import io
    import json
    from collections import OrderedDict

    from marie_icr_amd.weights import make_voting_case

    eng = _load_ref_voting_engine()
    cases = []
    for seed in range(24):
        regions = seed % 2 == 1
        names, agg, reg = make_voting_case(seed, regions)
        a = OrderedDict((n, copy.deepcopy(agg[n])) for n in names)
        with contextlib.redirect_stdout(io.StringIO()):              # the reference prints its candidate tables
            out = eng.voting_evaluator(a, a[names[0]], copy.deepcopy(reg))
        cases.append({"seed": seed, "regions": regions, "expected": out})
    # the reference's "nothing to evaluate" branch
    _, _, reg = make_voting_case(1, True)
    cases.append({"seed": 1, "regions": True, "empty": True, "expected": eng.voting_evaluator(OrderedDict(), None, copy.deepcopy(reg))})
    with open(os.path.join(out_dir, "voting.json"), "w", encoding="UTF-8") as f:
        json.dump({"cases": cases}, f)
    print("voting", len(cases), sum(len(json.dumps(c)) for c in cases))


def _load_ref_local_enhancer():
    """marie/models/pix2pix/models/networks_hd.py (and the three pure-torch modules it imports: gausian.py, swish.py,
    spectral_discriminator.py), unmodified, by path, as modules of a throw-away package."""
    import importlib.util
    import types

    base = "/root/reference/marie/models/pix2pix/models/"
    pkg = types.ModuleType("refpix")
    pkg.__path__ = []
    sys.modules["refpix"] = pkg
    for name in ("gausian", "swish", "spectral_discriminator", "networks_hd"):
        spec = importlib.util.spec_from_file_location("refpix." + name, base + name + ".py")
        m = importlib.util.module_from_spec(spec)
        sys.modules["refpix." + name] = m
        spec.loader.exec_module(m)
    return sys.modules["refpix.networks_hd"].LocalEnhancer


def gen_overlay(out_dir):
    """Outputs of the reference's own LocalEnhancer (netG 'local', instance norm: overlay.py:58-83, networks.py:189-196) for
    seeded weights and images."""
    import contextlib
    import functools
    import io

    import torch.nn as nn

    from marie_icr_amd.weights import make_image_u8, make_overlay_state

    LocalEnhancer = _load_ref_local_enhancer()
    norm = functools.partial(nn.InstanceNorm2d, affine=False, track_running_stats=False)
    for tag, ngf, (h, w), wseed, iseed in (("ngf32", 32, (64, 96), 0, 3), ("ngf64", 64, (64, 64), 1, 4)):
        with contextlib.redirect_stdout(io.StringIO()):              # the constructor prints a channel count
            net = LocalEnhancer(3, 3, ngf, 3, 9, 1, 3, norm)
        st = make_overlay_state(wseed, ngf)
        missing = net.load_state_dict({k: torch.from_numpy(v) for k, v in st.items()}, strict=True)
        net.eval()
        img = make_image_u8(iseed, 1, h, w)[0]                         # RGB order here; the product is fed the BGR-reversed frame
        x = ((torch.from_numpy(img.astype(np.float32) / np.float32(255.0)) - 0.5) / 0.5).permute(2, 0, 1).unsqueeze(0)
        with torch.no_grad():
            y = net(x)[0].permute(1, 2, 0).numpy()
        np.savez_compressed(os.path.join(out_dir, f"overlay_{tag}.npz"), weight_seed=wseed, image_seed=iseed, ngf=ngf,
                            hw=np.asarray([h, w]), weight_sha256=state_checksum(st), out=y.astype(np.float32))
        print("overlay", tag, y.shape, float(np.abs(y).max()), float(np.abs(y).mean()), missing)


def main():
    out_dir = os.path.join(ROOT, "tests", "golden")
    os.makedirs(out_dir, exist_ok=True)
    if "--overlay-only" in sys.argv:
        gen_overlay(out_dir)
        return
    if "--voting-only" in sys.argv:
        gen_voting(out_dir)
        return
    if "--renderer-only" in sys.argv:
        gen_renderer(out_dir)
        return
    if "--xml-renderers-only" in sys.argv:
        gen_xml_renderers(out_dir)
        return
    if "--vit-only" in sys.argv:
        gen_vit(out_dir)
        return
    if "--geometry-only" in sys.argv:
        gen_geometry(out_dir)
        return
    if "--craft-only" in sys.argv:
        gen_craft(out_dir)
        return
    if "--icr-only" in sys.argv:
        gen_icr(out_dir)
        return

    # (1) seeded "scaled" weights, 8 crops 32x256 (BASELINE config-2 shape)
    # (2) same weights, ragged width 32x100 (the production imgW) — exercises odd W
    for tag, n, w, wseed, iseed in (("scaled_w256", 8, 256, 0, 0), ("scaled_w100", 5, 100, 1, 3)):
        st = make_crnn_state(wseed)
        m, conv = _ref_model(w, state=st)
        crops = make_crnn_input(iseed, n, 32, w)
        logits, idx, strs, conf = _run(m, conv, crops)
        np.savez_compressed(
            os.path.join(out_dir, f"crnn_{tag}.npz"),
            weight_seed=wseed, input_seed=iseed, weight_sha256=state_checksum(st),
            crops_u8=crops, logits=logits, argmax=idx, strings=strs, confidence=conf)
        print(tag, logits.shape, "max|logit|", np.abs(logits).max(), strs[:2], conf[:2])

    # (3) PyTorch default init under torch.manual_seed(0) — BASELINE.md config 2.
    m, conv = _ref_model(256, seed=0)
    ref_state = {k: v.numpy() for k, v in m.state_dict().items()}
    ours = default_init_state(0)
    for k, v in ref_state.items():
        assert np.array_equal(ours[k], v), f"default-init mirror differs at {k}"
    crops = make_crnn_input(11, 4, 32, 256)
    logits, idx, strs, conf = _run(m, conv, crops)
    np.savez_compressed(
        os.path.join(out_dir, "crnn_default_w256.npz"),
        weight_seed=0, input_seed=11, weight_sha256=state_checksum(ours),
        crops_u8=crops, logits=logits, argmax=idx, strings=strs, confidence=conf)
    print("default", logits.shape, "max|logit|", np.abs(logits).max(), strs[:2], conf[:2])
    gen_craft(out_dir)
    gen_icr(out_dir)


if __name__ == "__main__":
    main()
