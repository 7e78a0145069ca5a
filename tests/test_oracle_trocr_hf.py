"""Independent cross-check of the TrOCR oracle (CPU).

fairseq and timm are absent from the reference tree and from this image, so ``oracle/trocr_torch.py`` is "parity unpinned"
(DESIGN.md §5).  The ``transformers`` library that IS installed carries its own implementations of the same two networks —
``ViTModel`` (timm ``VisionTransformer`` layout) and ``TrOCRForCausalLM`` (a port of fairseq's ``TransformerDecoder`` with
the TrOCR arguments: learned positions offset by pad + 1, ``layernorm_embedding``, post-LN layers, cross-attention over a
narrower encoder).  Built from a config only (no download), loaded with the SAME seeded weights under their own key
names, they must reproduce the oracle's encoder tokens and its incremental decoder logits.  This is a second opinion on
the restatement, not the oracle of record.
"""
import numpy as np
import pytest
import torch

from marie_icr_amd.weights import make_image_u8, make_trocr_state
from oracle.trocr_torch import TorchTrocrOracle

transformers = pytest.importorskip("transformers")

ENC = (128, 2, 4)          # width, depth, heads
DEC = (192, 2, 4, 384)     # width, layers, heads, ffn
VOCAB, MAXPOS, IMG = 61, 24, 64


def _state():
    return make_trocr_state(3, ENC, DEC, VOCAB, MAXPOS, img=IMG)


def _hf_vit(state):
    from transformers import ViTConfig, ViTModel

    ed, depth, heads = ENC
    cfg = ViTConfig(hidden_size=ed, num_hidden_layers=depth, num_attention_heads=heads, intermediate_size=4 * ed,
                    hidden_act="gelu", hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, layer_norm_eps=1e-6,
                    image_size=IMG, patch_size=16, num_channels=3, qkv_bias=True)
    m = ViTModel(cfg, add_pooling_layer=False).eval()
    p = "encoder.deit."
    t = lambda k: torch.from_numpy(np.ascontiguousarray(state[p + k]))
    sd = {"embeddings.cls_token": t("cls_token"), "embeddings.position_embeddings": t("pos_embed"),
          "embeddings.patch_embeddings.projection.weight": t("patch_embed.proj.weight"),
          "embeddings.patch_embeddings.projection.bias": t("patch_embed.proj.bias"),
          "layernorm.weight": t("norm.weight"), "layernorm.bias": t("norm.bias")}
    for i in range(depth):
        b, h = f"blocks.{i}.", f"layers.{i}."
        qkv = t(b + "attn.qkv.weight")
        qkv_b = t(b + "attn.qkv.bias") if (p + b + "attn.qkv.bias") in state else torch.zeros(3 * ed)
        for j, n in enumerate(("q_proj", "k_proj", "v_proj")):
            sd[h + f"attention.{n}.weight"] = qkv[j * ed:(j + 1) * ed]
            sd[h + f"attention.{n}.bias"] = qkv_b[j * ed:(j + 1) * ed]
        sd[h + "attention.o_proj.weight"] = t(b + "attn.proj.weight")
        sd[h + "attention.o_proj.bias"] = t(b + "attn.proj.bias")
        sd[h + "layernorm_before.weight"], sd[h + "layernorm_before.bias"] = t(b + "norm1.weight"), t(b + "norm1.bias")
        sd[h + "layernorm_after.weight"], sd[h + "layernorm_after.bias"] = t(b + "norm2.weight"), t(b + "norm2.bias")
        sd[h + "mlp.fc1.weight"], sd[h + "mlp.fc1.bias"] = t(b + "mlp.fc1.weight"), t(b + "mlp.fc1.bias")
        sd[h + "mlp.fc2.weight"], sd[h + "mlp.fc2.bias"] = t(b + "mlp.fc2.weight"), t(b + "mlp.fc2.bias")
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not missing and not unexpected, (missing, unexpected)
    return m


def _hf_decoder(state):
    from transformers import TrOCRConfig, TrOCRForCausalLM

    D, L, H, F = DEC
    cfg = TrOCRConfig(vocab_size=VOCAB, d_model=D, decoder_layers=L, decoder_attention_heads=H, decoder_ffn_dim=F,
                      activation_function="gelu", max_position_embeddings=MAXPOS, dropout=0.0, attention_dropout=0.0,
                      activation_dropout=0.0, decoder_layerdrop=0.0, use_cache=False, scale_embedding=False,
                      use_learned_position_embeddings=True, layernorm_embedding=True, cross_attention_hidden_size=ENC[0],
                      pad_token_id=1, bos_token_id=0, eos_token_id=2, decoder_start_token_id=2, tie_word_embeddings=False)
    m = TrOCRForCausalLM(cfg).eval()
    sd = {"model." + k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in state.items() if k.startswith("decoder.")}
    out_w = sd.pop("model.decoder.output_projection.weight", sd["model.decoder.embed_tokens.weight"])
    sd["output_projection.weight"] = out_w
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not missing and not unexpected, (missing, unexpected)
    return m


def test_encoder_tokens_match_transformers_vit():
    st = _state()
    crops = make_image_u8(5, 2, IMG, IMG)
    orc = TorchTrocrOracle(st, ENC[2], DEC[2], img=IMG)
    ours = orc.encode(crops)                                                   # (2, 17, 128)
    x = (torch.from_numpy(crops).permute(0, 3, 1, 2).float() / 255.0 - 0.5) / 0.5     # the reference's Normalize(0.5, 0.5)
    with torch.no_grad():
        theirs = _hf_vit(st)(pixel_values=x).last_hidden_state
    assert ours.shape == theirs.shape
    assert float((ours - theirs).abs().max()) <= 2e-4 * max(1.0, float(theirs.abs().max()))


def test_incremental_decoder_logits_match_transformers_trocr():
    st = _state()
    crops = make_image_u8(6, 2, IMG, IMG)
    orc = TorchTrocrOracle(st, ENC[2], DEC[2], img=IMG)
    enc = orc.encode(crops)
    cross = []
    for l in range(DEC[1]):
        p = f"decoder.layers.{l}.encoder_attn."
        cross.append((torch.nn.functional.linear(enc, orc.st[p + "k_proj.weight"], orc.st[p + "k_proj.bias"]),
                      torch.nn.functional.linear(enc, orc.st[p + "v_proj.weight"], orc.st[p + "v_proj.bias"])))
    rng = np.random.default_rng(0)
    T = 9
    toks = torch.from_numpy(rng.integers(3, VOCAB, size=(2, T)))
    toks[:, 0] = 2                                                             # fairseq starts every hypothesis with </s>
    hist = [None] * DEC[1]
    ours = torch.stack([orc.decoder_step(toks[:, s], s, hist, cross) for s in range(T)], dim=1)     # (2, T, V)
    with torch.no_grad():
        theirs = _hf_decoder(st)(input_ids=toks, encoder_hidden_states=enc).logits
    assert ours.shape == theirs.shape
    err = float((ours - theirs).abs().max())
    assert err <= 2e-4 * max(1.0, float(theirs.abs().max())), err
    assert torch.equal(ours.argmax(-1), theirs.argmax(-1))
