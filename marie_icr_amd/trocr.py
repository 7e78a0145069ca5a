"""TrOCR recognizer on MI355X behind the reference's ``TrOcrProcessor`` surface.

Mirrors ``TrOcrProcessor`` (reference: marie/document/trocr_ocr_processor.py:182-367): ``recognize_from_fragments(images)``
returns ``[{"confidence": round(exp(score), 4), "id": "img-<k>", "text": TEXT.upper()}]`` in input order.  Fragment
resize (Pillow bicubic to 384 x 384), the DeiT encoder, the fairseq-style decoder and the beam search run in
libmarie_hip.so; this file turns token ids into text: fairseq ``Dictionary`` symbols -> GPT-2 byte-level BPE decode
(marie/models/unilm/trocr/bpe.py:59-67; ``get_text`` :142-180).

The dictionary (``gpt2_with_mask.dict.txt``) and GPT-2's ``encoder.json`` are assets the reference downloads at run time;
they are not in its tree.  Pass their paths (``dict_path``, ``encoder_json``); without either the processor still runs and
returns the space-joined token ids as text (``encoder_json`` without ``dict_path`` is refused).
"""
from __future__ import annotations

import ctypes as C
import json
import math
import os
from typing import Dict, List, Optional, Sequence

import numpy as np

from ._lib import PREC_F16, PREC_F32, Context, MarieHipError, TrocrConfig, check
from .crnn import pack_fragments
from .ocr_processor import OcrProcessor
from .vit import load_tensors


def default_config(lib, model: str = "base") -> TrocrConfig:
    cfg = TrocrConfig()
    rc = lib.mhip_trocr_default_config(0 if model == "base" else 1, C.byref(cfg))
    if rc:
        raise ValueError(f"mhip_trocr_default_config({model}) -> {rc}")
    return cfg


def _vp(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else C.c_void_p(0)


class TrocrModel:
    def __init__(self, ctx: Context, state: Optional[Dict[str, np.ndarray]], config: TrocrConfig, precision: int = PREC_F16):
        self.ctx, self.lib, self.cfg, self.precision = ctx, ctx.lib, config, int(precision)
        h = C.c_void_p()
        check(ctx.h, self.lib.mhip_trocr_create(ctx.h, self.precision, C.byref(config), C.byref(h)), "mhip_trocr_create")
        self.h = h
        ctx.adopt(self)
        self.max_len = self.lib.mhip_trocr_max_len(C.byref(config))
        if state is not None:
            load_tensors(ctx, self.lib.mhip_trocr_set_tensor, self.h, state, "mhip_trocr_set_tensor")
            check(ctx.h, self.lib.mhip_trocr_finalize(self.h), "mhip_trocr_finalize")

    def arenas(self):
        out = []
        for which in (0, 1):
            p, n = C.c_void_p(), C.c_size_t()
            check(self.ctx.h, self.lib.mhip_trocr_arena(self.h, which, C.byref(p), C.byref(n)), "mhip_trocr_arena")
            out.append((p.value, n.value))
        return out

    def alloc_arena(self):
        check(self.ctx.h, self.lib.mhip_trocr_alloc_arena(self.h), "mhip_trocr_alloc_arena")

    def _outputs(self, n):
        return (np.empty((n, self.max_len + 1), np.int32), np.empty((n,), np.int32), np.empty((n,), np.float32))

    @staticmethod
    def _unpack(tokens, lengths, scores):
        return [(tokens[i, : lengths[i]].copy(), float(scores[i])) for i in range(len(lengths))]

    def generate_host(self, crops_u8: np.ndarray, swap_rb: bool = False, want_taps: bool = False):
        """crops (n, 384, 384, 3) uint8 -> [(token ids incl. eos, normalised log-prob)] (+ encoder tokens, step-0 logits)."""
        crops = np.ascontiguousarray(crops_u8, np.uint8)
        n = crops.shape[0]
        tokens, lengths, scores = self._outputs(n)
        enc = np.empty((n, 1 + (self.cfg.img_size // 16) ** 2, self.cfg.enc_dim), np.float32) if want_taps else None
        lg = np.empty((n, self.cfg.vocab), np.float32) if want_taps else None
        check(self.ctx.h, self.lib.mhip_trocr_generate_host(self.h, _vp(crops), n, int(swap_rb), _vp(tokens), _vp(lengths),
                                                            _vp(scores), _vp(enc), _vp(lg)), "mhip_trocr_generate_host")
        res = self._unpack(tokens, lengths, scores)
        return (res, enc, lg) if want_taps else res

    def generate_trace_host(self, crops_u8: np.ndarray, swap_rb: bool = False):
        """``generate_host`` + the beam search's candidate lists: (hypotheses, {"scores", "tokens", "beams"} each
        [steps][n][2 * beam]) — what the generator saw at every step (parity tests)."""
        crops = np.ascontiguousarray(crops_u8, np.uint8)
        n, k2 = crops.shape[0], 2 * self.cfg.beam
        tokens, lengths, scores = self._outputs(n)
        ts = np.zeros((self.max_len + 1, n, k2), np.float32)
        tt = np.zeros((self.max_len + 1, n, k2), np.int32)
        tb = np.zeros((self.max_len + 1, n, k2), np.int32)
        steps = C.c_int(0)
        check(self.ctx.h, self.lib.mhip_trocr_generate_trace_host(self.h, _vp(crops), n, int(swap_rb), _vp(tokens), _vp(lengths),
                                                                  _vp(scores), _vp(ts), _vp(tt), _vp(tb), C.byref(steps)),
              "mhip_trocr_generate_trace_host")
        k = steps.value
        return self._unpack(tokens, lengths, scores), {"scores": ts[:k], "tokens": tt[:k], "beams": tb[:k]}

    def generate_device(self, crops_ptr: int, n: int, swap_rb: bool = False):
        tokens, lengths, scores = self._outputs(n)
        check(self.ctx.h, self.lib.mhip_trocr_generate(self.h, C.c_void_p(crops_ptr), n, int(swap_rb), _vp(tokens), _vp(lengths),
                                                       _vp(scores)), "mhip_trocr_generate")
        return self._unpack(tokens, lengths, scores)

    def generate_fragments(self, base_ptr: int, descs, n: int, swap_rb: bool = True):
        tokens, lengths, scores = self._outputs(n)
        check(self.ctx.h, self.lib.mhip_trocr_generate_fragments(self.h, C.c_void_p(base_ptr), descs, n, int(swap_rb),
                                                                 _vp(tokens), _vp(lengths), _vp(scores)),
              "mhip_trocr_generate_fragments")
        return self._unpack(tokens, lengths, scores)

    def decode_gate(self):
        """The model's phase gate (created on first use): signalled in every generate call where decoding starts."""
        if getattr(self, "_gate", None) is None:
            self._gate = self.ctx.make_gate()
            check(self.ctx.h, self.lib.mhip_trocr_set_decode_gate(self.h, self._gate.h), "mhip_trocr_set_decode_gate")
        return self._gate

    # the recognizer in two halves (include/marie_hip.h): encoder per batch of fragments, decoder once over all of them
    def encode_begin(self, max_crops: int = 0):
        check(self.ctx.h, self.lib.mhip_trocr_encode_begin(self.h, int(max_crops)), "mhip_trocr_encode_begin")

    def encode_fragments(self, base_ptr: int, descs, n: int, swap_rb: bool = True):
        check(self.ctx.h, self.lib.mhip_trocr_encode_fragments(self.h, C.c_void_p(base_ptr), descs, n, int(swap_rb)),
              "mhip_trocr_encode_fragments")

    def encoded(self) -> int:
        return int(self.lib.mhip_trocr_encoded(self.h))

    def decode(self):
        n = self.encoded()
        tokens, lengths, scores = self._outputs(n)
        check(self.ctx.h, self.lib.mhip_trocr_decode(self.h, _vp(tokens), _vp(lengths), _vp(scores)), "mhip_trocr_decode")
        return self._unpack(tokens, lengths, scores)

    def close(self):
        if getattr(self, "h", None) is not None and self.h.value:
            self.lib.mhip_trocr_set_decode_gate(self.h, None)
            self.lib.mhip_trocr_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---------------------------------------------------------------------------------------------------- text side
def load_fairseq_dictionary(path: str, pad_to_multiple: int = 1) -> List[str]:
    """fairseq ``Dictionary.load`` (``add_from_file``) as the reference's task calls it
    (marie/models/unilm/trocr/task.py:86-101): ``<s>``, ``<pad>``, ``</s>``, ``<unk>``, then one symbol per
    ``<symbol> <count>[ #fairseq:overwrite]`` line.  A repeated symbol is an error unless its line carries the overwrite flag,
    in which case it gets a NEW index (the earlier one keeps its slot).  ``Dictionary.load`` does not pad: the 50 265 entries
    of ``gpt2_with_mask.dict.txt`` are 4 specials + 50 260 lines (its own ``madeupword0000..2`` among them) + ``<mask>``.
    ``pad_to_multiple`` > 1 appends fairseq's ``madeupwordNNNN`` fillers (``Dictionary.pad_to_multiple_``, used by
    ``finalize`` at preprocessing time, not by ``load``)."""
    symbols = ["<s>", "<pad>", "</s>", "<unk>"]
    seen = set(symbols)
    with open(path, "r", encoding="utf-8") as f:
        for raw in f:
            try:
                line, field = raw.rstrip().rsplit(" ", 1)
                overwrite = field == "#fairseq:overwrite"
                if overwrite:
                    line, field = line.rsplit(" ", 1)
                int(field)
            except ValueError:
                raise ValueError(f"Incorrect dictionary format, expected '<token> <cnt> [flags]': \"{raw}\"")
            if line in seen and not overwrite:
                raise RuntimeError(f"Duplicate word found when loading Dictionary: '{line}'")
            seen.add(line)
            symbols.append(line)
    i = 0
    while pad_to_multiple > 1 and len(symbols) % pad_to_multiple != 0:
        symbols.append(f"madeupword{i:04d}")
        i += 1
    return symbols


def _bytes_to_unicode() -> Dict[int, str]:
    """GPT-2's reversible byte <-> printable-unicode table."""
    bs = list(range(ord("!"), ord("~") + 1)) + list(range(ord("¡"), ord("¬") + 1)) + list(range(ord("®"), ord("ÿ") + 1))
    cs = bs[:]
    n = 0
    for b in range(256):
        if b not in bs:
            bs.append(b)
            cs.append(256 + n)
            n += 1
    return dict(zip(bs, [chr(c) for c in cs]))


class Gpt2Decoder:
    """``bpe.decode`` of GPT2BPEEnhancedSpace (INSERT_OR_REPLACE = 0): ids -> byte-level BPE strings -> utf-8, with the
    ``<s>`` space marker removed."""

    def __init__(self, encoder_json: str):
        with open(encoder_json, "r", encoding="utf-8") as f:
            enc = json.load(f)
        self.decoder = {v: k for k, v in enc.items()}
        self.byte_decoder = {v: k for k, v in _bytes_to_unicode().items()}

    def decode(self, hypo_str: str) -> str:
        parts = []
        for tok in hypo_str.split():
            parts.append(tok if tok in {"<unk>", "<mask>", "<s>"} else self.decoder[int(tok)])
        text = "".join(parts)
        return bytearray([self.byte_decoder[c] for c in text]).decode("utf-8", errors="replace").replace("<s>", "")


def hypo_string(tokens: Sequence[int], symbols: Optional[List[str]], eos: int = 2, bos: int = 0, unk: int = 3) -> str:
    """``Dictionary.string`` as ``utils.post_process_prediction`` calls it: symbols joined by spaces, eos / bos skipped."""
    out = []
    for t in tokens:
        t = int(t)
        if t == eos or t == bos:
            continue
        out.append("<unk>" if t == unk else (symbols[t] if symbols is not None else str(t)))
    return " ".join(out)


class TrOcrProcessor(OcrProcessor):
    """Drop-in for marie/document/trocr_ocr_processor.py:182."""

    def __init__(self, work_dir: str = "/tmp/icr", model_name_or_path: Optional[str] = None, cuda: bool = True, *,
                 state: Optional[Dict[str, np.ndarray]] = None, config: Optional[TrocrConfig] = None, model: str = "base",
                 precision: str = "f16", device_id: int = 0, ctx: Optional[Context] = None, dict_path: Optional[str] = None,
                 encoder_json: Optional[str] = None, batch_size: int = 256, trocr_model: Optional[TrocrModel] = None,
                 **kwargs) -> None:
        super().__init__(work_dir, cuda)
        if not cuda:
            raise MarieHipError("TrOcrProcessor here is the MI355X path; cuda=False has no implementation")
        if state is None and trocr_model is None:
            # trocr_ocr_processor.py:199-217: <model zoo>/trocr/trocr-large-printed.pt unless a path is given; a missing file
            # is FileNotFoundError.  Resolved (and read) before a device context exists.
            from .constants import __model_path__

            model_path = os.path.join(__model_path__, "trocr", "trocr-large-printed.pt")
            if model_name_or_path:
                model_path = model_name_or_path
            if not os.path.exists(model_path):
                raise FileNotFoundError(f"File not found : {model_path}")
            import torch

            ck = torch.load(model_path, map_location="cpu", weights_only=True)
            sd = ck.get("model", ck)
            state = {k: v.float().numpy() for k, v in sd.items() if hasattr(v, "numpy")}
            if config is None:
                # the checkpoint says which architecture it is (trocr_models.py:423-470: base = DeiT 768, large = 1024)
                pe = state.get("encoder.deit.pos_embed")
                if pe is not None:
                    model = "large" if pe.shape[-1] == 1024 else "base"
        self.ctx = ctx or (trocr_model.ctx if trocr_model is not None else Context(device_id))
        cfg = trocr_model.cfg if trocr_model is not None else (config or default_config(self.ctx.lib, model))
        if encoder_json and not dict_path:
            # the dictionary's symbols ARE the GPT-2 BPE ids; decoding dictionary indices as BPE ids would be silently wrong
            raise ValueError("encoder_json needs dict_path: GPT-2 BPE ids are the fairseq dictionary's symbols, not its indices")
        self.symbols = load_fairseq_dictionary(dict_path) if dict_path else None
        if self.symbols is not None and len(self.symbols) != cfg.vocab:
            raise ValueError(f"dictionary has {len(self.symbols)} symbols, the model expects {cfg.vocab}")
        self.bpe = Gpt2Decoder(encoder_json) if encoder_json else None
        prec = {"f16": PREC_F16, "fp16": PREC_F16, "f32": PREC_F32, "fp32": PREC_F32}[precision]
        self.model = trocr_model if trocr_model is not None else TrocrModel(self.ctx, state, cfg, prec)
        self.batch_size = int(batch_size)
        # where the decode phase of a generate call starts in this recognizer's stream (the engine runs the next page batch's
        # detector under it, ocr_engine.OcrEngine._fullpage_batched)
        self.decode_gate = self.model.decode_gate()

    def is_available(self) -> bool:
        return self.model is not None

    def _text(self, tokens) -> str:
        s = hypo_string(tokens, self.symbols, eos=self.model.cfg.eos)
        return self.bpe.decode(s) if self.bpe is not None else s

    def _results(self, hyps, first_id: int = 0) -> List[Dict[str, object]]:
        out = []
        for k, (tokens, score) in enumerate(hyps):
            conf = round(math.exp(score), 6)                 # get_text: round(exp(score), 6), then round(score, 4)
            text = self._text(tokens)
            out.append({"confidence": round(conf, 4), "id": f"img-{first_id + k}", "text": text.upper() if text is not None else ""})
        return out

    # ---- several page batches, one beam search (OcrEngine's batched path) ---------------------------------------------------
    decode_batch = 4096      # crops per beam search: more than this many pending crops are decoded before the next batch is added

    def recognize_pages_begin(self, expected_pages: int = 0):
        """Start a call whose pages arrive in batches (``recognize_pages_add``) and are finished together
        (``recognize_pages_finish``): the image encoder runs on every batch as it arrives, the beam search once over the crops of
        all batches — its 16 steps cost ~5 ms each however few crops there are, so one search over 2560 crops is ~80 ms cheaper
        than two over 1280.  A page's result equals what ``recognize`` returns for it (tests/test_pipeline_gpu.py)."""
        self._pending = []          # (img shape, boxes, lines, n fragments) in arrival order
        self._done = []             # results of the pages already decoded
        self._pending_results = []  # fragment results of the pages in _pending that have been decoded (always empty or complete)
        self.model.encode_begin(0)

    def recognize_pages_add(self, pages):
        import torch

        from .fragments import FragmentList

        self.ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        for img, boxes, fragments, lines in pages:
            img = self._check_inputs(img, boxes, fragments, lines)
            n = len(fragments) if len(boxes) else 0
            if n and self.model.encoded() + n > self.decode_batch and self.model.encoded():
                self._flush()
            if n:
                frl = fragments if isinstance(fragments, FragmentList) else FragmentList(list(fragments))
                dd = frl.device_descs(0, n)
                if dd is not None:
                    self.model.encode_fragments(dd[0], dd[1], n, swap_rb=True)                 # fragments are BGR
                else:
                    packed, descs = pack_fragments([f if np.ndim(f) == 3 else np.repeat(np.asarray(f)[:, :, None], 3, axis=2)
                                                    for f in fragments])
                    d_in = torch.from_numpy(packed).cuda()
                    self.model.encode_fragments(d_in.data_ptr(), descs, n, swap_rb=True)
                    torch.cuda.current_stream().synchronize()                                  # d_in dies here
            self._pending.append((img.shape, boxes, lines, n))

    def _flush(self):
        results = self._results(self.model.decode()) if self.model.encoded() else []
        k = 0
        for shape, boxes, lines, n in self._pending:
            self._done.append(self._assemble(shape, boxes, lines, results[k:k + n], n == 0))
            k += n
        assert k == len(results), "You must provide the same number of results as fragments."
        self._pending = []

    def recognize_pages_finish(self):
        self._flush()
        out, self._done = self._done, []
        return out

    def recognize_from_fragments(self, src_images, **kwargs) -> List[Dict[str, object]]:
        """reference: trocr_ocr_processor.py:241-367.  Fragments that carry the window of a device page they were cut from
        (``FragmentList`` from the MI355X box processor) are read where they are; anything else is packed and uploaded."""
        import torch

        results: List[Dict[str, object]] = []
        self.ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        on_device = getattr(src_images, "windows", None) is not None
        for start in range(0, len(src_images), self.batch_size):
            stop = min(len(src_images), start + self.batch_size)
            if on_device:
                base, descs = src_images.device_descs(start, stop)
                hyps = self.model.generate_fragments(base, descs, stop - start, swap_rb=True)              # fragments are BGR
            else:
                batch = src_images[start:stop]
                packed, descs = pack_fragments([f if np.ndim(f) == 3 else np.repeat(np.asarray(f)[:, :, None], 3, axis=2)
                                                for f in batch])
                d_in = torch.from_numpy(packed).cuda()
                hyps = self.model.generate_fragments(d_in.data_ptr(), descs, len(batch), swap_rb=True)
            results.extend(self._results(hyps, start))
        return results
