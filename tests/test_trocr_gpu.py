"""GPU parity of the TrOCR recognizer (DeiT encoder + fairseq-style decoder + beam search) through the C ABI against
oracle/trocr_torch.py.  The encoder stack is pinned through the BEiT goldens (tests/test_vit_gpu.py); the decoder and
the generator are restated from fairseq (third-party, absent here) — parity unpinned, see the oracle's header."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ENC, DEC, VOCAB, MAXPOS = (256, 2, 4), (256, 2, 4, 512), 97, 32


@pytest.fixture(scope="module")
def ctx():
    from marie_icr_amd._lib import Context

    c = Context(0)
    yield c
    c.close()


def _cfg(ctx, beam=3, max_len_b=12):
    from marie_icr_amd.trocr import default_config

    cfg = default_config(ctx.lib, "base")
    cfg.enc_dim, cfg.enc_depth, cfg.enc_heads = ENC
    cfg.dec_dim, cfg.dec_layers, cfg.dec_heads, cfg.dec_ffn = DEC
    cfg.vocab, cfg.max_positions, cfg.beam, cfg.max_len_b = VOCAB, MAXPOS, beam, max_len_b
    return cfg


@pytest.fixture(scope="module")
def case():
    from marie_icr_amd.weights import make_image_u8, make_trocr_state
    from oracle.trocr_torch import TorchTrocrOracle

    st = make_trocr_state(0, ENC, DEC, VOCAB, MAXPOS)
    crops = make_image_u8(21, 8, 384, 384)
    o = TorchTrocrOracle(st, ENC[2], DEC[2], beam=3, max_len_b=12)
    ref, step0 = o.generate(crops, want_step0=True)
    return st, crops, o, ref, step0


def test_fp32_tokens_and_scores(ctx, case):
    from marie_icr_amd._lib import PREC_F32
    from marie_icr_amd.trocr import TrocrModel

    st, crops, o, ref, step0 = case
    m = TrocrModel(ctx, st, _cfg(ctx), PREC_F32)
    got, enc, lg = m.generate_host(crops, want_taps=True)
    assert np.abs(enc - o.encode(crops).numpy()).max() <= 1e-3
    assert np.abs(lg - step0).max() <= 2e-3
    lens = set()
    for (gt, gs), (rt, rs) in zip(got, ref):
        np.testing.assert_array_equal(gt, rt)
        assert abs(gs - rs) <= 1e-3
        lens.add(len(rt))
    assert len(lens) >= 3, f"the seeded model should end hypotheses at different lengths: {lens}"
    m.close()


def test_beam1_and_forced_max_len(ctx, case):
    from marie_icr_amd._lib import PREC_F32
    from marie_icr_amd.trocr import TrocrModel
    from oracle.trocr_torch import TorchTrocrOracle

    st, crops, *_ = case
    o = TorchTrocrOracle(st, ENC[2], DEC[2], beam=1, max_len_b=3)
    ref = o.generate(crops[:3])
    m = TrocrModel(ctx, st, _cfg(ctx, beam=1, max_len_b=3), PREC_F32)
    got = m.generate_host(crops[:3])
    for (gt, gs), (rt, rs) in zip(got, ref):
        np.testing.assert_array_equal(gt, rt)
        assert len(gt) <= 4 and gt[-1] == 2
        assert abs(gs - rs) <= 1e-3
    m.close()


def test_f16_and_processor_surface(ctx, case):
    from marie_icr_amd.trocr import TrOcrProcessor
    from oracle.trocr_torch import preprocess_fragments

    st, crops, o, ref, _ = case
    rng = np.random.default_rng(5)
    frags = [rng.integers(0, 256, size=(int(h), int(w), 3)).astype(np.uint8) for h, w in ((40, 130), (25, 300), (60, 61))]
    frags.append(rng.integers(0, 256, size=(30, 90)).astype(np.uint8))          # a gray fragment
    p = TrOcrProcessor(state=st, config=_cfg(ctx), precision="f32", ctx=ctx)
    res = p.recognize_from_fragments(frags)
    assert [r["id"] for r in res] == ["img-0", "img-1", "img-2", "img-3"]
    f3 = [f if f.ndim == 3 else np.repeat(f[:, :, None], 3, axis=2) for f in frags]
    oref = o.generate(preprocess_fragments(f3))
    for r, (rt, rs) in zip(res, oref):
        assert r["text"] == " ".join(str(int(t)) for t in rt if t not in (0, 2)).upper()
        assert abs(r["confidence"] - round(round(float(np.exp(rs)), 6), 4)) <= 2e-4
    # production precision: same tokens on these clear-margin cases, scores within 3 %
    p16 = TrOcrProcessor(state=st, config=_cfg(ctx), precision="f16", ctx=ctx)
    res16 = p16.recognize_from_fragments(frags)
    same = sum(a["text"] == b["text"] for a, b in zip(res, res16))
    assert same >= 3


@pytest.mark.parametrize("precision", ["f32", "f16"])
def test_encoder_per_batch_decoder_once_equals_one_call(ctx, case, precision):
    """The recognizer in two halves (mhip_trocr_encode_begin / encode_fragments / decode, the engine's batched path): fragments
    encoded in ragged batches — host fragments and windows of a device page mixed, the token store growing past its reservation —
    and searched once give the hypotheses of ``recognize_from_fragments`` on the concatenation, bit for bit; pages without boxes
    pass through; a second call on the same processor starts from an empty store; ``decode_batch`` splits the search."""
    import torch

    from marie_icr_amd.fragments import FragmentList
    from marie_icr_amd.trocr import TrOcrProcessor

    st = case[0]
    rng = np.random.default_rng(17)
    page = rng.integers(0, 256, size=(300, 700, 3)).astype(np.uint8)
    dev = torch.from_numpy(page).cuda()
    boxes = [(10, 20, 300, 40), (5, 100, 650, 33), (400, 200, 120, 60), (0, 0, 700, 25), (350, 150, 200, 18)]

    def windows(sel):
        fr = [page[y:y + h, x:x + w] for x, y, w, h in sel]
        win = [(dev.data_ptr() + (y * 700 + x) * 3, h, w, 700 * 3, 3) for x, y, w, h in sel]
        return FragmentList([np.array(f) for f in fr], win, [dev])

    host = [rng.integers(0, 256, size=(int(h), int(w), 3)).astype(np.uint8) for h, w in ((40, 130), (25, 300), (60, 61), (33, 512))]
    p = TrOcrProcessor(state=st, config=_cfg(ctx), precision=precision, ctx=ctx)
    # six batches: more than the four buffers of the context's pinned staging ring (every batch's descriptor table goes through it
    # without draining the stream)
    batches = [windows(boxes[:2]), FragmentList(host[:3]), windows(boxes[2:3]), FragmentList(host[3:]), windows(boxes[3:4]),
               windows(boxes[4:])]
    flat = [f for b in batches for f in b]
    ref = p.recognize_from_fragments(FragmentList(flat))
    shape = page.shape
    for decode_batch in (4096, 4):
        p.decode_batch = decode_batch
        p.recognize_pages_begin(5)
        p.recognize_pages_add([(page, [[0, 0, 1, 1]] * len(batches[0]), batches[0], [1] * len(batches[0]))])
        p.recognize_pages_add([(page, [], [], []),                                              # a blank page in the middle
                               (page, [[0, 0, 1, 1]] * len(batches[1]), batches[1], [1] * len(batches[1]))])
        for b in batches[2:]:
            p.recognize_pages_add([(page, [[0, 0, 1, 1]] * len(b), b, list(range(1, len(b) + 1)))])
        out = p.recognize_pages_finish()
        assert len(out) == 7
        got = [(w["text"], w["confidence"]) for res, _ in out for w in sorted(res["words"], key=lambda w: w["id"])]
        assert out[1][0]["words"] == [] and out[1][0]["meta"]["imageSize"] == {"width": shape[1], "height": shape[0]}
        want = [(r["text"], round(r["confidence"], 3)) for r in ref]
        assert sorted(got) == sorted(want)
        # per page: the page's own fragments, in order
        k = 0
        for j, b in zip((0, 2, 3, 4, 5, 6), batches):
            words = sorted(out[j][0]["words"], key=lambda w: w["id"])
            assert [w["text"] for w in words] == [r["text"] for r in ref[k:k + len(b)]]
            k += len(b)


def test_single_crop_and_extreme_fragments(ctx, case):
    """n = 1; 1-pixel-high, 1-pixel-wide and very wide fragments (Pillow-exact resize to 384 x 384 before the encoder)."""
    from marie_icr_amd.trocr import TrOcrProcessor
    from oracle.trocr_torch import preprocess_fragments

    st, crops, o, ref, _ = case
    rng = np.random.default_rng(8)
    frags = [rng.integers(0, 256, size=s).astype(np.uint8) for s in ((1, 90, 3), (70, 1, 3), (12, 2300, 3))]
    p = TrOcrProcessor(state=st, config=_cfg(ctx), precision="f32", ctx=ctx, batch_size=2)      # 2 + 1: a batch of one
    res = p.recognize_from_fragments(frags)
    oref = o.generate(preprocess_fragments(frags))
    assert [r["id"] for r in res] == ["img-0", "img-1", "img-2"]
    for r, (rt, rs) in zip(res, oref):
        assert r["text"] == " ".join(str(int(t)) for t in rt if t not in (0, 2))
    assert p.recognize_from_fragments([]) == []


@pytest.mark.parametrize("enc,dec,vocab,maxpos,beam", [
    ((256, 2, 4), (512, 2, 8, 640), 203, 24, 3),      # encoder narrower than the decoder (as trocr-base: 768 vs 1024)
    ((512, 1, 8), (256, 3, 4, 384), 61, 20, 4),       # encoder wider than the decoder, beam 4, short dictionary
    ((768, 1, 12), (1024, 1, 16, 1024), 1031, 16, 2), # the released base widths (one layer each), prime vocabulary, beam 2
])
def test_fp32_parity_with_unequal_encoder_and_decoder_widths(ctx, enc, dec, vocab, maxpos, beam):
    """Cross-attention projects encoder tokens of one width into a decoder of another (kdim / vdim = encoder width, as in
    every released TrOCR model); head counts, FFN widths, vocabulary sizes and beam widths away from the defaults."""
    from marie_icr_amd._lib import PREC_F32
    from marie_icr_amd.trocr import TrocrModel, default_config
    from marie_icr_amd.weights import make_image_u8, make_trocr_state
    from oracle.trocr_torch import TorchTrocrOracle

    st = make_trocr_state(5, enc, dec, vocab, maxpos)
    crops = make_image_u8(31, 5, 384, 384)
    max_len_b = min(10, maxpos - 4)
    o = TorchTrocrOracle(st, enc[2], dec[2], beam=beam, max_len_b=max_len_b)
    ref, step0 = o.generate(crops, want_step0=True)
    cfg = default_config(ctx.lib, "base")
    cfg.enc_dim, cfg.enc_depth, cfg.enc_heads = enc
    cfg.dec_dim, cfg.dec_layers, cfg.dec_heads, cfg.dec_ffn = dec
    cfg.vocab, cfg.max_positions, cfg.beam, cfg.max_len_b = vocab, maxpos, beam, max_len_b
    m = TrocrModel(ctx, st, cfg, PREC_F32)
    got, encoded, lg = m.generate_host(crops, want_taps=True)
    assert np.abs(encoded - o.encode(crops).numpy()).max() <= 1e-3
    assert np.abs(lg - step0).max() <= 2e-3
    for (gt, gs), (rt, rs) in zip(got, ref):
        np.testing.assert_array_equal(gt, rt)
        assert abs(gs - rs) <= 1e-3
    m.close()


@pytest.mark.parametrize("crops,beam,heads,n_tok,enc_dim", [
    (5, 3, 16, 577, 768),      # trocr-base: 19 key tiles, the last one with 1 valid key
    (3, 3, 4, 577, 256),       # the small test model
    (2, 4, 8, 100, 512),       # beam 4, 8 heads, fewer keys than four tiles
    (1, 1, 16, 33, 1024),      # trocr-large width (2-slot ring), a single hypothesis, one key in the second tile
    (4, 2, 12, 64, 768),       # 12 heads: head slots 12..15 of a wave idle; exactly two full tiles
])
def test_encoder_attention_with_absorbed_projections(ctx, crops, beam, heads, n_tok, enc_dim):
    """The f16 decoder attends over the encoder tokens themselves (W_k folded into the queries, W_v applied to the context;
    cross_attn.hip).  Against fairseq's MultiheadAttention arithmetic in fp64 on the same f16-rounded operands: the absorbed
    form drops q . b_k (constant over the keys) and must give the same output."""
    import ctypes as C

    rng = np.random.default_rng(crops * 1000 + enc_dim)
    D, M = heads * 64, crops * beam
    f16 = lambda a: a.astype(np.float16).astype(np.float32)
    q = f16(rng.normal(0, 0.6, (M, D)))
    enc = f16(rng.normal(0, 1.0, (crops, n_tok, enc_dim)))
    wk = f16(rng.normal(0, 2.0 / np.sqrt(enc_dim), (D, enc_dim)))
    wv = f16(rng.normal(0, 1.0 / np.sqrt(enc_dim), (D, enc_dim)))
    bk = rng.normal(0, 0.5, (D,)).astype(np.float32)        # present in the reference computation only
    bv = rng.normal(0, 0.3, (D,)).astype(np.float32)
    enc[0, n_tok // 2] *= 6.0                                 # a dominant key well into the sequence: the stale reference must rescale
    out = np.empty((M, D), np.float32)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    rc = ctx.lib.mhip_cross_attention_host(ctx.h, vp(q), vp(np.ascontiguousarray(enc)), vp(wk), vp(wv), vp(bv), crops, beam, heads,
                                           n_tok, enc_dim, vp(out))
    assert rc == 0, ctx.last_error() if hasattr(ctx, "last_error") else rc
    K = enc.astype(np.float64) @ wk.T.astype(np.float64) + bk            # (crops, n_tok, D)
    V = enc.astype(np.float64) @ wv.T.astype(np.float64) + bv
    ref = np.empty((M, D))
    for r in range(M):
        c = r // beam
        for h in range(heads):
            sl = slice(h * 64, h * 64 + 64)
            s = K[c][:, sl] @ q[r, sl].astype(np.float64)
            p = np.exp(s - s.max())
            ref[r, sl] = (p / p.sum()) @ V[c][:, sl]
    err = np.abs(out - ref).max()
    assert err <= 2e-2 * max(1.0, np.abs(ref).max()), (err, np.abs(ref).max())
    assert np.abs(out - ref).mean() <= 2e-3 * max(1.0, np.abs(ref).max())
