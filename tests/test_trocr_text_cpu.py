"""TrOCR text side (T5): fairseq dictionary -> hypothesis string -> GPT-2 byte-level BPE decode -> text.

Mirrors marie/document/trocr_ocr_processor.py:142-180 (``get_text``), marie/models/unilm/trocr/task.py:86-101
(``Dictionary.load``) and marie/models/unilm/trocr/bpe.py:59-67 (``GPT2BPEEnhancedSpace.decode``, INSERT_OR_REPLACE = 0).
fairseq is absent from the reference tree and from this image (parity unpinned): ``Dictionary.add_from_file`` /
``Dictionary.string`` are restated from fairseq v0.12; the byte-level decode is cross-checked against the installed
``tokenizers`` library's ByteLevel decoder, an independent implementation of GPT-2's byte table."""
import json

import numpy as np
import pytest

from marie_icr_amd.trocr import Gpt2Decoder, _bytes_to_unicode, hypo_string, load_fairseq_dictionary

PIECES = ["H", "ello", " wor", "ld", "é", " €", "!", " ", "日本", "\n", "\x00", "ÿ", " <"]


def _byte_level(s: str) -> str:
    b2u = _bytes_to_unicode()
    return "".join(b2u[b] for b in s.encode("utf-8"))


@pytest.fixture()
def assets(tmp_path):
    # encoder.json: byte-level piece -> GPT-2 BPE id (ids deliberately not 0..n-1 in dictionary order)
    bpe_ids = [15496, 11, 995, 0, 50256, 3, 7, 220, 42, 198, 188, 12520, 1279]
    enc = {_byte_level(p): i for p, i in zip(PIECES, bpe_ids)}
    ej = tmp_path / "encoder.json"
    ej.write_text(json.dumps(enc), encoding="utf-8")
    # dict.txt in the layout of gpt2_with_mask.dict.txt: "<bpe id> <count>", frequency order, its own madeupword rows, <mask>
    order = [3, 0, 7, 1, 2, 4, 5, 6, 8, 9, 10, 11, 12]
    lines = [f"{bpe_ids[k]} {1000 - 7 * n}" for n, k in enumerate(order)]
    lines += ["madeupword0000 0", "madeupword0001 0", "<mask> 0"]
    dp = tmp_path / "dict.txt"
    dp.write_text("\n".join(lines) + "\n", encoding="utf-8")
    index_of_piece = {PIECES[k]: 4 + n for n, k in enumerate(order)}
    return str(dp), str(ej), index_of_piece, bpe_ids


def test_dictionary_load_layout(assets):
    dp, ej, idx, bpe_ids = assets
    sym = load_fairseq_dictionary(dp)
    assert sym[:4] == ["<s>", "<pad>", "</s>", "<unk>"]                 # fairseq Dictionary specials, in this order
    assert len(sym) == 4 + 13 + 3                                       # Dictionary.load does NOT pad
    assert sym[4] == str(bpe_ids[3]) and sym[-1] == "<mask>" and sym[-3] == "madeupword0000"
    assert sym[idx["ello"]] == "11"
    # pad_to_multiple_ (finalize-time behaviour) on request only
    padded = load_fairseq_dictionary(dp, pad_to_multiple=8)
    assert len(padded) == 24 and padded[20:] == ["madeupword0000", "madeupword0001", "madeupword0002", "madeupword0003"]


def test_dictionary_overwrite_flag_and_errors(tmp_path):
    p = tmp_path / "d.txt"
    p.write_text("a 5\nb 4\na 3 #fairseq:overwrite\nwith space 2\n", encoding="utf-8")
    sym = load_fairseq_dictionary(str(p))
    assert sym[4:] == ["a", "b", "a", "with space"]                    # the overwrite row gets a new index; symbols may hold spaces
    p.write_text("a 5\na 3\n", encoding="utf-8")
    with pytest.raises(RuntimeError, match="Duplicate"):
        load_fairseq_dictionary(str(p))
    p.write_text("a five\n", encoding="utf-8")
    with pytest.raises(ValueError, match="Incorrect dictionary format"):
        load_fairseq_dictionary(str(p))
    p.write_text("lonely\n", encoding="utf-8")
    with pytest.raises(ValueError):
        load_fairseq_dictionary(str(p))


def test_hypo_string_skips_eos_and_bos_keeps_unk(assets):
    dp, ej, idx, _ = assets
    sym = load_fairseq_dictionary(dp)
    toks = [0, idx["H"], idx["ello"], 3, idx["!"], 2]                  # <s> H ello <unk> ! </s>
    assert hypo_string(toks, sym) == "15496 11 <unk> 7"
    assert hypo_string(toks, None) == f"{idx['H']} {idx['ello']} <unk> {idx['!']}"      # no dictionary: raw indices
    assert hypo_string([2], sym) == ""
    assert hypo_string([1, 2], sym) == "<pad>"                          # pad is not in extra_symbols_to_ignore


def test_gpt2_decode_round_trip_multibyte_and_space_marker(assets):
    from tokenizers import decoders

    dp, ej, idx, _ = assets
    sym = load_fairseq_dictionary(dp)
    bpe = Gpt2Decoder(ej)
    seq = ["H", "ello", " wor", "ld", "!", " €", "é", "日本", " ", "\n", "ÿ", " <"]
    toks = [idx[p] for p in seq] + [2]
    text = bpe.decode(hypo_string(toks, sym))
    assert text == "Hello world! €é日本 \nÿ <"
    # independent byte-level decoder on the same pieces
    assert decoders.ByteLevel().decode([_byte_level(p) for p in seq]) == text
    # the "<s>" marker the enhanced-space encoder inserts for a blank is dropped on decode (INSERT_OR_REPLACE = 0), <unk> stays
    hs = hypo_string([idx["H"], 0, idx["ello"], 3, 2], sym)
    assert hs == "15496 11 <unk>"
    assert bpe.decode("15496 <s> 11 <unk> <mask>") == "Hello<unk><mask>"
    # a split multi-byte character decodes with U+FFFD, like bytearray.decode(errors="replace") in GPT-2's decoder
    enc = json.load(open(ej, encoding="utf-8"))
    enc[_bytes_to_unicode()[0xE6]] = 77                                 # first byte of 日 alone
    json.dump(enc, open(ej, "w", encoding="utf-8"))
    assert Gpt2Decoder(ej).decode("77 7") == "�!"
    # dictionary rows that are not BPE ids (madeupword fillers) fail loudly, as int(tok) does in the reference
    with pytest.raises(ValueError):
        bpe.decode(hypo_string([len(sym) - 3, 2], sym))


def test_byte_table_is_a_bijection_over_256_bytes():
    t = _bytes_to_unicode()
    assert sorted(t) == list(range(256)) and len(set(t.values())) == 256
    assert t[ord("A")] == "A" and t[ord(" ")] == "Ġ" and t[0] == "Ā" and t[0xAD] == "Ń"


def test_processor_refuses_encoder_without_dictionary(assets):
    """decoding dictionary INDICES as BPE ids is silently wrong text — refuse the combination.  The check sits
    before any GPU work, so it is testable without a device."""
    from marie_icr_amd import trocr

    dp, ej, *_ = assets

    class _Ctx:                                    # stands in for Context: the constructor must raise before using it
        lib = None

    cfg = trocr.TrocrConfig()
    cfg.vocab = 20
    with pytest.raises(ValueError, match="needs dict_path"):
        trocr.TrOcrProcessor(state={}, config=cfg, ctx=_Ctx(), encoder_json=ej)
    with pytest.raises(ValueError, match="dictionary has 20 symbols, the model expects 21"):
        cfg.vocab = 21
        trocr.TrOcrProcessor(state={}, config=cfg, ctx=_Ctx(), dict_path=dp, encoder_json=ej)
