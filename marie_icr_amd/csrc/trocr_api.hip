// trocr_api.hip — the TrOCR recognizer (image encoder + autoregressive text decoder + beam search) behind the C ABI.
//
// Host-side counterpart of TrOcrProcessor's model path (marie/document/trocr_ocr_processor.py:116-180,251-367):
//   preprocess_image: PIL bicubic resize of the fragment to 384 x 384, /255, (x - 0.5) / 0.5        (:116-125)
//   TrOCREncoder.forward -> AdaptedVisionTransformer.forward_features (trocr_models.py:508-524; deit.py:105-146)
//   TextRecognitionGenerator._generate (generator.py:11-374) over fairseq's TransformerDecoder
//   (trocr_models.py:137-147,180-186): beam 3 (trocr_ocr_processor.py:228), max_len = min(200, max_positions - 1),
//   cand_size = 2 * beam, length-normalised scores; the top hypothesis per fragment is returned.
// fairseq is third-party and absent from the reference tree: decoder layer, incremental state, BeamSearch.step and
// finalize_hypos follow fairseq's published v0.12 behaviour.
//
// MI355X shape of the problem: the encoder (577 tokens x 12 layers per fragment) is >= 85 % of the FLOPs and runs on
// the ViT MFMA path; a decoder step is GEMMs with M = fragments x beams rows plus two HBM-bound attentions.  The
// self-attention cache is never re-ordered: an ancestry table maps (hypothesis, past step) -> cache slot.
#include <math.h>

#include "vit_internal.h"

struct mhip_trocr {
  mhip_ctx* ctx = nullptr;
  int precision = MHIP_PREC_F16;
  mhip_trocr_config cfg{};
  mhip_vit* vit = nullptr;
  TensorStore store;
  Arena arena;
  bool ready = false;
  // grow-only staging of generate_fragments (resized crops + resize scratch): no hipMalloc in steady state
  uint8_t* frag_crops = nullptr;
  size_t frag_crops_bytes = 0;
  void* frag_scratch = nullptr;
  size_t frag_scratch_bytes = 0;
  bool absorb = false;       // encoder-attention with absorbed K / V projections (f16 mode)
  mhip_gate* decode_gate = nullptr;   // signalled where the decode phase of a generate call starts in the stream
  // "crops not finished yet" after every step, copied to pinned memory; the host reads it two steps late (never drains the stream)
  int* h_remaining = nullptr;
  hipEvent_t rem_ev[2] = {nullptr, nullptr};
  // encoder tokens of the crops encoded since mhip_trocr_encode_begin (the encoder runs per batch of fragments as the page
  // batches come out of the detector; the autoregressive decoder then runs once over all of them): [enc_cap * npad + 64][enc_dim] T
  char* enc_store = nullptr;
  int enc_cap = 0, enc_count = 0;
  size_t esz() const { return precision == MHIP_PREC_F16 ? 2 : 4; }
};

namespace {

const char* kEncPrefix = "encoder.deit.";
constexpr float DEC_LN_EPS = 1e-5f;

std::string lay(int l, const char* s) { return "L" + std::to_string(l) + "." + s; }

}  // namespace

extern "C" int mhip_trocr_default_config(int model, mhip_trocr_config* c) {
  if (!c || model < 0 || model > 1) return MHIP_EINVAL;
  // trocr_base (beit_base_patch16_384) / trocr_large (beit_large_patch16_384): trocr_models.py:423-447
  c->enc_dim = model ? 1024 : 768; c->enc_depth = model ? 24 : 12; c->enc_heads = model ? 16 : 12;
  c->dec_dim = 1024; c->dec_layers = 12; c->dec_heads = 16; c->dec_ffn = 4096;
  c->vocab = 50265;            // fairseq Dictionary over gpt2_with_mask.dict.txt (file not in the reference tree)
  c->max_positions = 512;
  c->beam = 3; c->max_len_b = 200; c->min_len = 1;
  c->pad = 1; c->eos = 2;
  c->embed_scale = 1.0f;       // RoBERTa arguments: no_scale_embedding
  c->img_size = 384;
  return MHIP_OK;
}

extern "C" int mhip_trocr_create(mhip_ctx* ctx, int precision, const mhip_trocr_config* cfg, mhip_trocr** out) {
  if (!ctx || !cfg || !out) return MHIP_EINVAL;
  *out = nullptr;
  if (precision != MHIP_PREC_F16 && precision != MHIP_PREC_F32) return mhip_fail(ctx, MHIP_EINVAL, "unknown precision %d", precision);
  const mhip_trocr_config& c = *cfg;
  if (c.dec_dim != c.dec_heads * 64 || c.dec_dim % 256 || c.dec_dim > 1024 || c.dec_ffn % 64 || c.dec_layers < 1 ||
      c.vocab < 8 || c.beam < 1 || c.beam > 4 || c.img_size % 16 || c.max_len_b < 1 || c.max_positions < 2 ||
      c.pad < 0 || c.eos < 0 || c.pad >= c.vocab || c.eos >= c.vocab)
    return mhip_fail(ctx, MHIP_EINVAL, "trocr: unsupported configuration");
  mhip_vit_config vc{};
  vc.dim = c.enc_dim; vc.depth = c.enc_depth; vc.heads = c.enc_heads; vc.patch = 16;
  vc.pos_h = vc.pos_w = c.img_size / 16;
  vc.layer_scale = 0; vc.qkv_bias = 0; vc.final_norm = 1; vc.fpn = 0; vc.ln_eps = 1e-6f;
  mhip_vit* vit = nullptr;
  int rc = mhip_vit_create(ctx, precision, &vc, &vit);
  if (rc) return rc;
  mhip_trocr* m = new mhip_trocr();
  m->ctx = ctx; m->precision = precision; m->cfg = c; m->vit = vit;
  // f16 mode: encoder-attention over the encoder tokens themselves (cross_attn.hip); MARIE_HIP_NO_ABSORB=1 keeps the projected
  // K / V path (A/B measurements)
  m->absorb = precision == MHIP_PREC_F16 && mhip_cross_absorb_supported(c.enc_dim, c.beam, c.dec_heads) && !getenv("MARIE_HIP_NO_ABSORB");
  const size_t es = m->esz(), D = c.dec_dim, E = c.enc_dim, F = c.dec_ffn;
  Arena& a = m->arena;
  a.take("emb", (size_t)c.vocab * D * es);
  a.take("out_w", (size_t)c.vocab * D * es);
  a.take("pos", (size_t)(c.max_positions + c.pad + 1) * D * 4);
  a.take("lne_g", D * 4); a.take("lne_b", D * 4);
  for (int l = 0; l < c.dec_layers; ++l) {
    // the three self-attention input projections sit back to back — one [3 D][D] weight and one [3 D] bias for a single GEMM per
    // layer and step (q | k | v of a step are one row of the history, below)
    for (const char* n : {"sa_q", "sa_k", "sa_v"}) a.take(lay(l, n) + "_w", D * D * es);
    for (const char* n : {"sa_q", "sa_k", "sa_v"}) a.take(lay(l, n) + "_b", D * 4);
    for (const char* n : {"sa_o", "ca_q", "ca_o"}) { a.take(lay(l, n) + "_w", D * D * es); a.take(lay(l, n) + "_b", D * 4); }
    // absorbed encoder-attention: W_k lives only as ca_kt and b_k drops out of the soft-max — no ca_k slots in the arena
    for (const char* n : {"ca_k", "ca_v"}) {
      if (m->absorb && !strcmp(n, "ca_k")) continue;
      a.take(lay(l, n) + "_w", D * E * es); a.take(lay(l, n) + "_b", D * 4);
    }
    if (m->absorb) a.take(lay(l, "ca_kt"), D * E * 2);     // W_k per head, transposed, x log2(e): [heads][E][64] f16
    a.take(lay(l, "fc1_w"), F * D * es); a.take(lay(l, "fc1_b"), F * 4);
    a.take(lay(l, "fc2_w"), D * F * es); a.take(lay(l, "fc2_b"), D * 4);
    for (const char* n : {"sa_ln", "ca_ln", "fin_ln"}) { a.take(lay(l, n) + "_g", D * 4); a.take(lay(l, n) + "_b", D * 4); }
  }
  hipError_t he = hipHostMalloc((void**)&m->h_remaining, (size_t)(c.max_positions + 2) * sizeof(int));
  for (int i = 0; i < 2 && he == hipSuccess; ++i) he = hipEventCreateWithFlags(&m->rem_ev[i], hipEventDisableTiming);
  if (he != hipSuccess) {
    mhip_trocr_destroy(m);
    return mhip_fail(ctx, MHIP_EHIP, "trocr: pinned step counter: %s", hipGetErrorString(he));
  }
  *out = m;
  return MHIP_OK;
}

extern "C" int mhip_trocr_destroy(mhip_trocr* m) {
  if (!m) return MHIP_OK;
  mhip_quiesce(m->ctx);
  if (m->h_remaining) (void)hipHostFree(m->h_remaining);
  for (auto e : m->rem_ev)
    if (e) (void)hipEventDestroy(e);
  mhip_vit_destroy(m->vit);
  m->arena.release();
  if (m->frag_crops) (void)hipFree(m->frag_crops);
  if (m->frag_scratch) (void)hipFree(m->frag_scratch);
  if (m->enc_store) (void)hipFree(m->enc_store);
  delete m;
  return MHIP_OK;
}

extern "C" int mhip_trocr_set_tensor(mhip_trocr* m, const char* key, const float* data, const int64_t* shape, int ndim) {
  if (!m || !key) return MHIP_EINVAL;
  std::string k(key);
  if (k.rfind(kEncPrefix, 0) == 0) {
    const std::string sub = k.substr(strlen(kEncPrefix));
    if (sub.rfind("head.", 0) == 0 || sub.rfind("head_dist.", 0) == 0 || sub == "dist_token") return MHIP_OK;   // classifier heads: unused
    return mhip_vit_set_tensor(m->vit, sub.c_str(), data, shape, ndim);
  }
  if (k == "decoder.version" || k == "encoder.version" || k.find("_float_tensor") != std::string::npos) return MHIP_OK;
  if (k.rfind("decoder.", 0) != 0) return mhip_fail(m->ctx, MHIP_EINVAL, "unknown state_dict key %s", key);
  m->ready = false;
  return m->store.set(m->ctx, k, data, shape, ndim);
}

extern "C" int mhip_trocr_alloc_arena(mhip_trocr* m) {
  if (!m) return MHIP_EINVAL;
  int rc = m->arena.alloc(m->ctx);
  if (rc) return rc;
  if ((rc = mhip_vit_alloc_arena(m->vit))) return rc;
  m->ready = true;
  return MHIP_OK;
}

extern "C" int mhip_trocr_arena(mhip_trocr* m, int which, void** dev, size_t* bytes) {
  if (!m || which < 0 || which > 1) return MHIP_EINVAL;
  if (which == 0) return mhip_vit_arena(m->vit, dev, bytes);
  if (dev) *dev = m->arena.dev;
  if (bytes) *bytes = m->arena.bytes;
  return MHIP_OK;
}

extern "C" int mhip_trocr_finalize(mhip_trocr* m) {
  if (!m) return MHIP_EINVAL;
  mhip_ctx* ctx = m->ctx;
  int rc = mhip_vit_finalize(m->vit);
  if (rc) return rc;
  const mhip_trocr_config& c = m->cfg;
  const int prec = m->precision, D = c.dec_dim, E = c.enc_dim, F = c.dec_ffn;
  Arena& a = m->arena;
  const TensorStore& st = m->store;
  a.begin_fill();
  const HostTensor* emb = st.find(ctx, "decoder.embed_tokens.weight", {c.vocab, D});
  const HostTensor* pos = st.find(ctx, "decoder.embed_positions.weight", {c.max_positions + c.pad + 1, D});
  const HostTensor* lg = st.find(ctx, "decoder.layernorm_embedding.weight", {D});
  const HostTensor* lb = st.find(ctx, "decoder.layernorm_embedding.bias", {D});
  if (!emb || !pos || !lg || !lb) return MHIP_ESTATE;
  Arena::put(prec, a.h("emb"), emb->data.data(), emb->numel());
  if (st.has("decoder.output_projection.weight")) {
    const HostTensor* ow = st.find(ctx, "decoder.output_projection.weight", {c.vocab, D});
    if (!ow) return MHIP_ESTATE;
    Arena::put(prec, a.h("out_w"), ow->data.data(), ow->numel());
  } else {
    Arena::put(prec, a.h("out_w"), emb->data.data(), emb->numel());   // share_decoder_input_output_embed
  }
  memcpy(a.h("pos"), pos->data.data(), pos->numel() * 4);
  memcpy(a.h("lne_g"), lg->data.data(), D * 4);
  memcpy(a.h("lne_b"), lb->data.data(), D * 4);
  const float qscale = 0.125f;   // head_dim 64: MultiheadAttention scales q (bias included) by head_dim^-0.5
  for (int l = 0; l < c.dec_layers; ++l) {
    const std::string p = "decoder.layers." + std::to_string(l) + ".";
    struct Lin { const char* name; std::string key; int out, in; float scale; };
    const Lin lins[] = {{"sa_q", p + "self_attn.q_proj", D, D, qscale}, {"sa_k", p + "self_attn.k_proj", D, D, 1.f},
                        {"sa_v", p + "self_attn.v_proj", D, D, 1.f},    {"sa_o", p + "self_attn.out_proj", D, D, 1.f},
                        {"ca_q", p + "encoder_attn.q_proj", D, D, qscale}, {"ca_k", p + "encoder_attn.k_proj", D, E, 1.f},
                        {"ca_v", p + "encoder_attn.v_proj", D, E, 1.f}, {"ca_o", p + "encoder_attn.out_proj", D, D, 1.f},
                        {"fc1", p + "fc1", F, D, 1.f}, {"fc2", p + "fc2", D, F, 1.f}};
    for (const Lin& L : lins) {
      if (m->absorb && !strcmp(L.name, "ca_k")) continue;
      const HostTensor* w = st.find(ctx, L.key + ".weight", {L.out, L.in});
      const HostTensor* b = st.find(ctx, L.key + ".bias", {L.out});
      if (!w || !b) return MHIP_ESTATE;
      if (L.scale != 1.f) {
        std::vector<float> t(w->data);
        for (float& v : t) v *= L.scale;
        Arena::put(prec, a.h(lay(l, L.name) + "_w"), t.data(), t.size());
        float* bb = (float*)a.h(lay(l, L.name) + "_b");
        for (int i = 0; i < L.out; ++i) bb[i] = b->data[i] * L.scale;
      } else {
        Arena::put(prec, a.h(lay(l, L.name) + "_w"), w->data.data(), w->numel());
        memcpy(a.h(lay(l, L.name) + "_b"), b->data.data(), (size_t)L.out * 4);
      }
    }
    if (m->absorb) {
      const HostTensor* wk = st.find(ctx, p + "encoder_attn.k_proj.weight", {D, E});
      if (!wk) return MHIP_ESTATE;
      _Float16* kt = (_Float16*)a.h(lay(l, "ca_kt"));
      const float log2e = 1.4426950408889634f;
      for (int h = 0; h < c.dec_heads; ++h)
        for (int d = 0; d < E; ++d)
          for (int j = 0; j < 64; ++j) kt[((size_t)h * E + d) * 64 + j] = (_Float16)(wk->data[(size_t)(h * 64 + j) * E + d] * log2e);
    }
    const std::pair<const char*, std::string> lns[] = {{"sa_ln", p + "self_attn_layer_norm"}, {"ca_ln", p + "encoder_attn_layer_norm"},
                                                       {"fin_ln", p + "final_layer_norm"}};
    for (const auto& ln : lns) {
      const HostTensor* g = st.find(ctx, ln.second + ".weight", {D});
      const HostTensor* b = st.find(ctx, ln.second + ".bias", {D});
      if (!g || !b) return MHIP_ESTATE;
      memcpy(a.h(lay(l, ln.first) + "_g"), g->data.data(), D * 4);
      memcpy(a.h(lay(l, ln.first) + "_b"), b->data.data(), D * 4);
    }
  }
  if ((rc = a.upload(ctx))) return rc;
  m->ready = true;
  m->store.t.clear();
  return MHIP_OK;
}

extern "C" int mhip_trocr_set_decode_gate(mhip_trocr* m, mhip_gate* gate) {
  if (!m) return MHIP_EINVAL;
  m->decode_gate = gate;
  return MHIP_OK;
}

static int trocr_max_len(const mhip_trocr_config& c) { return std::min(c.max_len_b, c.max_positions - 1); }

extern "C" int mhip_trocr_max_len(const mhip_trocr_config* c) { return c ? trocr_max_len(*c) : MHIP_EINVAL; }

static size_t trocr_ws_bytes(const mhip_trocr* m, int n) {
  const mhip_trocr_config& c = m->cfg;
  const size_t es = m->esz(), D = c.dec_dim, M = (size_t)n * c.beam, ML = trocr_max_len(c);
  VitGeom vg;
  vit_geometry(m->vit, c.img_size, c.img_size, &vg);
  const size_t ldv = (c.vocab + 7) / 8 * 8;
  size_t b = vit_workspace_bytes(m->vit, n, vg);
  if (m->absorb) b += 2 * M * 16 * (size_t)c.enc_dim * 2 + 512;      // absorbed queries / contexts
  else b += 2 * (size_t)c.dec_layers * n * vg.npad * D * es;          // cross K / V
  b += 3 * (size_t)c.dec_layers * (ML + 1) * M * D * es;              // self q | k | v history
  b += M * D * 4 + 3 * M * D * es + M * c.dec_ffn * es + M * ldv * 4; // x, xt, q, ao, hidden, logits
  b += 2 * M * (ML + 2) * 4 + (size_t)n * 2 * c.beam * 12 + (size_t)vg.n_tok * c.enc_dim * 4 + (size_t)c.vocab * 4 + 4096;
  b += mhip_beam_state_bytes(n, c.beam, (int)ML) + (size_t)n * (ML + 3) * 4 + 1024;   // generator state + device copies of the outputs
  return b + (1 << 16);
}

extern "C" size_t mhip_trocr_workspace_bytes(mhip_trocr* m, int n) { return (m && n > 0) ? trocr_ws_bytes(m, n) : 0; }

// crops_dev: n images u8 [img][img][3].  tokens_out [n][max_len + 1] (the hypothesis without the leading eos, eos included,
// padded with `pad`), lengths_out [n], scores_out [n] (length-normalised log-probability of the best hypothesis).
struct TrocrTrace {          // host arrays [max_len + 1][n][2 * beam], filled for the steps that ran
  float* scores;
  int32_t* tokens;
  int32_t* beams;
  int steps;
};

static int trocr_decode(mhip_trocr* m, Carver& ws, const char* enc_tokens, const VitGeom& vg, int n, int32_t* tokens_out,
                        int32_t* lengths_out, float* scores_out, float* step0_logits_host, TrocrTrace* trace);

static int trocr_generate(mhip_trocr* m, const uint8_t* crops_dev, int n, int swap_rb, int32_t* tokens_out,
                          int32_t* lengths_out, float* scores_out, float* enc_tokens_host, float* step0_logits_host,
                          TrocrTrace* trace = nullptr) {
  mhip_ctx* ctx = m->ctx;
  if (!m->ready) return mhip_fail(ctx, MHIP_ESTATE, "trocr: weights not finalized");
  if (n < 1) return mhip_fail(ctx, MHIP_EINVAL, "trocr: empty batch");
  MHIP_HIP(ctx, hipSetDevice(ctx->device));
  const mhip_trocr_config& c = m->cfg;
  const int prec = m->precision, E = c.enc_dim;
  const size_t es = m->esz();
  int rc = mhip_ensure_workspace(ctx, trocr_ws_bytes(m, n));
  if (rc) return rc;
  Carver ws(ctx->ws);
  // ---- encoder --------------------------------------------------------------------------------------------------
  VitRun run;
  if ((rc = vit_encode(m->vit, ws, crops_dev, n, c.img_size, c.img_size, c.img_size, c.img_size, swap_rb, &run))) return rc;
  const VitGeom& vg = run.g;
  if (enc_tokens_host) {
    float* stage = ws.take<float>((size_t)vg.n_tok * E * 4);
    for (int i = 0; i < n; ++i) {
      if ((rc = mhip_launch_convert_rows(ctx, prec, run.tokens + (size_t)i * vg.npad * E * es, stage, vg.n_tok, E))) return rc;
      MHIP_HIP(ctx, hipMemcpyAsync(enc_tokens_host + (size_t)i * vg.n_tok * E, stage, (size_t)vg.n_tok * E * 4, hipMemcpyDeviceToHost, ctx->stream));
      MHIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
  }
  return trocr_decode(m, ws, run.tokens, vg, n, tokens_out, lengths_out, scores_out, step0_logits_host, trace);
}

// The autoregressive part (TextRecognitionGenerator._generate, generator.py:127-362) over the encoder tokens of n crops:
// enc_tokens T [n * vg.npad + 64 slack rows][enc_dim].
static int trocr_decode(mhip_trocr* m, Carver& ws, const char* enc_tokens, const VitGeom& vg, int n, int32_t* tokens_out,
                        int32_t* lengths_out, float* scores_out, float* step0_logits_host, TrocrTrace* trace) {
  mhip_ctx* ctx = m->ctx;
  const mhip_trocr_config& c = m->cfg;
  const int prec = m->precision, D = c.dec_dim, E = c.enc_dim, F = c.dec_ffn, beam = c.beam, L = c.dec_layers;
  const int K2 = 2 * beam, ML = trocr_max_len(c), M = n * beam;
  const size_t es = m->esz();
  const int ldv = (c.vocab + 7) / 8 * 8;
  const Arena& a = m->arena;
  int rc;
  struct { const char* tokens; } run{enc_tokens};
  // ---- encoder keys / values of every decoder layer (static over the steps) -----------------------------------------
  const size_t cross_l = (size_t)n * vg.npad * D * es;
  char* ck = m->absorb ? nullptr : ws.take(L * cross_l);
  char* cv = m->absorb ? nullptr : ws.take(L * cross_l);
  char* qt = m->absorb ? ws.take((size_t)M * 16 * E * 2) : nullptr;
  char* ct = m->absorb ? ws.take((size_t)M * 16 * E * 2) : nullptr;
  for (int l = 0; l < L && !m->absorb; ++l) {
    if ((rc = mhip_gemm(ctx, prec, run.tokens, a.d(lay(l, "ca_k") + "_w"), (long long)n * vg.npad, D, E, nullptr, a.d<float>(lay(l, "ca_k") + "_b"), ck + l * cross_l, ACT_NONE, 0))) return rc;
    if ((rc = mhip_gemm(ctx, prec, run.tokens, a.d(lay(l, "ca_v") + "_w"), (long long)n * vg.npad, D, E, nullptr, a.d<float>(lay(l, "ca_v") + "_b"), cv + l * cross_l, ACT_NONE, 0))) return rc;
  }
  // the MFMA-bound part of the call is enqueued; what follows is HBM- and latency-bound: whoever waits on the gate (the
  // detector of the next page batch) runs underneath it
  if (m->decode_gate && (rc = mhip_gate_signal(m->decode_gate, ctx))) return rc;
  // ---- decoder state ------------------------------------------------------------------------------------------------
  const size_t hist_s = (size_t)M * 3 * D * es;              // one step of one layer: rows of q | k | v
  char* hqkv = ws.take((size_t)L * (ML + 1) * hist_s);
  float* x = ws.take<float>((size_t)M * D * 4);
  char* xt = ws.take((size_t)M * D * es);
  char* qb = ws.take((size_t)M * D * es);
  char* ao = ws.take((size_t)M * D * es);
  char* hid = ws.take((size_t)M * F * es);
  char* logits = ws.take((size_t)M * ldv * 4);      // element type T: f16 logits in the f16 mode, fp32 in the parity mode
  const int anc_ld = ML + 2;
  int* anc[2] = {ws.take<int>((size_t)M * anc_ld * 4), ws.take<int>((size_t)M * anc_ld * 4)};
  float* d_cs = ws.take<float>((size_t)n * K2 * 4);
  int* d_ct = ws.take<int>((size_t)n * K2 * 4);
  int* d_cb = ws.take<int>((size_t)n * K2 * 4);
  // generator state (TextRecognitionGenerator._generate without batch compaction: finished crops keep their rows), all in HBM
  BeamState bs;
  if ((rc = mhip_beam_state_carve(ws.take(mhip_beam_state_bytes(n, beam, ML)), n, beam, ML, c.pad, c.eos, &bs))) return rc;
  bs.cand_scores = d_cs; bs.cand_tokens = d_ct; bs.cand_beams = d_cb;
  int* d_out_tok = ws.take<int>((size_t)n * (ML + 1) * 4);
  int* d_out_len = ws.take<int>((size_t)n * 4);
  float* d_out_score = ws.take<float>((size_t)n * 4);
  if ((rc = mhip_launch_beam_init(ctx, bs, anc[0], anc_ld))) return rc;
  int cur = 0, tcur = 0;
  for (int step = 0; step <= ML; ++step) {
    if (step >= 2) {
      // the counter of two steps ago: the stream still holds step - 1 while this one is enqueued, so the device never waits for
      // the host; when every crop has finished, one step too many has been enqueued — finished crops ignore it
      MHIP_HIP(ctx, hipEventSynchronize(m->rem_ev[step & 1]));
      if (m->h_remaining[step - 2] == 0) break;
    }
    if (step > 0) {
      if ((rc = mhip_launch_ancestry(ctx, anc[cur], anc[cur ^ 1], bs.parent, M, anc_ld, step - 1))) return rc;
      cur ^= 1;
    }
    // LearnedPositionalEmbedding, incremental: position = padding_idx + (step + 1)
    const float* pos_row = a.d<float>("pos") + (size_t)(c.pad + step + 1) * D;
    if ((rc = mhip_launch_embed_step(ctx, prec, bs.last_tok, a.d("emb"), pos_row, c.embed_scale, a.d<float>("lne_g"), a.d<float>("lne_b"), x, xt, M, D, DEC_LN_EPS))) return rc;
    for (int l = 0; l < L; ++l) {
      char* hl = hqkv + ((size_t)l * (ML + 1)) * hist_s;
      // self-attention over the hypothesis' own history (post-LN residual block): q | k | v of this step in ONE GEMM (the three
      // weights are adjacent in the arena), written as one row of the history
      if ((rc = mhip_gemm(ctx, prec, xt, a.d(lay(l, "sa_q") + "_w"), M, 3 * D, D, nullptr, a.d<float>(lay(l, "sa_q") + "_b"), hl + (size_t)step * hist_s, ACT_NONE, 0))) return rc;
      DecAttnDesc sa;
      sa.q = hl + (size_t)step * hist_s; sa.k = hl + (size_t)D * es; sa.v = hl + (size_t)2 * D * es; sa.out = ao;
      sa.anc = anc[cur]; sa.anc_ld = anc_ld; sa.slots = M;
      sa.ldq = sa.ldk = 3 * D; sa.ldo = D; sa.heads = c.dec_heads; sa.groups = M; sa.nq = 1; sa.n_keys = step + 1;
      if ((rc = mhip_launch_decode_attention(ctx, prec, sa))) return rc;
      if ((rc = mhip_gemm(ctx, prec, ao, a.d(lay(l, "sa_o") + "_w"), M, D, D, nullptr, a.d<float>(lay(l, "sa_o") + "_b"), x, ACT_NONE, 1, x))) return rc;
      if ((rc = mhip_launch_layernorm2(ctx, prec, x, a.d<float>(lay(l, "sa_ln") + "_g"), a.d<float>(lay(l, "sa_ln") + "_b"), x, xt, M, D, DEC_LN_EPS))) return rc;
      // attention over the crop's encoder tokens (keys / values shared by its beams)
      if ((rc = mhip_gemm(ctx, prec, xt, a.d(lay(l, "ca_q") + "_w"), M, D, D, nullptr, a.d<float>(lay(l, "ca_q") + "_b"), qb, ACT_NONE, 0))) return rc;
      if (m->absorb) {
        CrossAbsorbDesc cd;
        cd.q = qb; cd.ldq = D; cd.E = run.tokens; cd.kv_rows = vg.npad; cd.n_keys = vg.n_tok; cd.enc_dim = E;
        cd.wkt = a.d(lay(l, "ca_kt")); cd.wv = a.d(lay(l, "ca_v") + "_w"); cd.bv = a.d<float>(lay(l, "ca_v") + "_b");
        cd.qt = qt; cd.ct = ct; cd.ao = ao; cd.ldo = D; cd.crops = n; cd.beam = beam; cd.heads = c.dec_heads;
        if ((rc = mhip_launch_cross_absorbed(ctx, cd))) return rc;
      } else {
        DecAttnDesc ca;
        ca.q = qb; ca.k = ck + l * cross_l; ca.v = cv + l * cross_l; ca.out = ao; ca.kv_rows = vg.npad;
        ca.ldq = ca.ldk = ca.ldo = D; ca.heads = c.dec_heads; ca.groups = n; ca.nq = beam; ca.n_keys = vg.n_tok;
        if ((rc = mhip_launch_decode_attention(ctx, prec, ca))) return rc;
      }
      if ((rc = mhip_gemm(ctx, prec, ao, a.d(lay(l, "ca_o") + "_w"), M, D, D, nullptr, a.d<float>(lay(l, "ca_o") + "_b"), x, ACT_NONE, 1, x))) return rc;
      if ((rc = mhip_launch_layernorm2(ctx, prec, x, a.d<float>(lay(l, "ca_ln") + "_g"), a.d<float>(lay(l, "ca_ln") + "_b"), x, xt, M, D, DEC_LN_EPS))) return rc;
      // feed-forward
      if ((rc = mhip_gemm(ctx, prec, xt, a.d(lay(l, "fc1_w")), M, F, D, nullptr, a.d<float>(lay(l, "fc1_b")), hid, ACT_GELU, 0))) return rc;
      if ((rc = mhip_gemm(ctx, prec, hid, a.d(lay(l, "fc2_w")), M, D, F, nullptr, a.d<float>(lay(l, "fc2_b")), x, ACT_NONE, 1, x))) return rc;
      if ((rc = mhip_launch_layernorm2(ctx, prec, x, a.d<float>(lay(l, "fin_ln") + "_g"), a.d<float>(lay(l, "fin_ln") + "_b"), x, xt, M, D, DEC_LN_EPS))) return rc;
    }
    if ((rc = mhip_gemm(ctx, prec, xt, a.d("out_w"), M, c.vocab, D, nullptr, nullptr, logits, ACT_NONE, 0, nullptr, ldv, 1))) return rc;
    if (step == 0 && step0_logits_host) {
      float* stage = ws.take<float>((size_t)c.vocab * 4);
      for (int i = 0; i < n; ++i) {
        if ((rc = mhip_launch_convert_rows(ctx, prec, logits + (size_t)i * beam * ldv * es, stage, 1, c.vocab))) return rc;
        MHIP_HIP(ctx, hipMemcpyAsync(step0_logits_host + (size_t)i * c.vocab, stage, (size_t)c.vocab * 4, hipMemcpyDeviceToHost, ctx->stream));
        MHIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
      }
    }
    BeamCandDesc bc;
    bc.logits = logits; bc.logits_f16 = prec == MHIP_PREC_F16; bc.ld = ldv; bc.vocab = c.vocab; bc.beam = beam; bc.bsz = n; bc.cum = bs.cum; bc.step = step;
    bc.max_len = ML; bc.min_len = c.min_len; bc.pad = c.pad; bc.eos = c.eos;
    bc.cand_scores = d_cs; bc.cand_tokens = d_ct; bc.cand_beams = d_cb;
    if ((rc = mhip_launch_beam_candidates(ctx, bc))) return rc;
    if (trace) {       // parity tests: the step's candidate list as the generator sees it (this path drains the stream)
      const size_t off = (size_t)step * n * K2;
      MHIP_HIP(ctx, hipMemcpyAsync(trace->scores + off, d_cs, (size_t)n * K2 * 4, hipMemcpyDeviceToHost, ctx->stream));
      MHIP_HIP(ctx, hipMemcpyAsync(trace->tokens + off, d_ct, (size_t)n * K2 * 4, hipMemcpyDeviceToHost, ctx->stream));
      MHIP_HIP(ctx, hipMemcpyAsync(trace->beams + off, d_cb, (size_t)n * K2 * 4, hipMemcpyDeviceToHost, ctx->stream));
      MHIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
      trace->steps = step + 1;
    }
    if ((rc = mhip_launch_beam_select(ctx, bs, tcur, step))) return rc;
    tcur ^= 1;
    MHIP_HIP(ctx, hipMemcpyAsync(&m->h_remaining[step], bs.remaining, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    MHIP_HIP(ctx, hipEventRecord(m->rem_ev[step & 1], ctx->stream));
  }
  // best hypothesis per crop: highest score, first finalized wins ties (torch.sort(descending) on the score list)
  if ((rc = mhip_launch_beam_best(ctx, bs, d_out_tok, d_out_len, d_out_score))) return rc;
  MHIP_HIP(ctx, hipMemcpyAsync(tokens_out, d_out_tok, (size_t)n * (ML + 1) * 4, hipMemcpyDeviceToHost, ctx->stream));
  MHIP_HIP(ctx, hipMemcpyAsync(lengths_out, d_out_len, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
  MHIP_HIP(ctx, hipMemcpyAsync(scores_out, d_out_score, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
  MHIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return MHIP_OK;
}

extern "C" int mhip_trocr_generate(mhip_trocr* m, const uint8_t* crops_dev, int n, int swap_rb, int32_t* tokens_out,
                                   int32_t* lengths_out, float* scores_out) {
  if (!m || !crops_dev || !tokens_out || !lengths_out || !scores_out) return MHIP_EINVAL;
  return trocr_generate(m, crops_dev, n, swap_rb, tokens_out, lengths_out, scores_out, nullptr, nullptr);
}

extern "C" int mhip_trocr_generate_host(mhip_trocr* m, const uint8_t* crops_host, int n, int swap_rb, int32_t* tokens_out,
                                        int32_t* lengths_out, float* scores_out, float* enc_tokens_out,
                                        float* step0_logits_out) {
  if (!m || !crops_host || !tokens_out || !lengths_out || !scores_out || n < 1) return MHIP_EINVAL;
  mhip_ctx* ctx = m->ctx;
  MHIP_HIP(ctx, hipSetDevice(ctx->device));
  const size_t nb = (size_t)n * m->cfg.img_size * m->cfg.img_size * 3;
  uint8_t* dev = nullptr;
  MHIP_HIP(ctx, hipMalloc((void**)&dev, nb));
  hipError_t e = hipMemcpy(dev, crops_host, nb, hipMemcpyHostToDevice);
  int rc = e == hipSuccess ? MHIP_OK : mhip_fail(ctx, MHIP_EHIP, "crop upload: %s", hipGetErrorString(e));
  if (!rc) rc = trocr_generate(m, dev, n, swap_rb, tokens_out, lengths_out, scores_out, enc_tokens_out, step0_logits_out);
  (void)hipStreamSynchronize(ctx->stream);
  (void)hipFree(dev);
  return rc;
}

// generate_host + the candidate list of every step as the generator saw it (parity tests: the CPU restatement walks the two
// searches side by side).  trace_* : host arrays [max_len + 1][n][2 * beam]; *steps_out = steps that ran.
extern "C" int mhip_trocr_generate_trace_host(mhip_trocr* m, const uint8_t* crops_host, int n, int swap_rb, int32_t* tokens_out,
                                              int32_t* lengths_out, float* scores_out, float* trace_scores,
                                              int32_t* trace_tokens, int32_t* trace_beams, int* steps_out) {
  if (!m || !crops_host || !tokens_out || !lengths_out || !scores_out || !trace_scores || !trace_tokens || !trace_beams ||
      !steps_out || n < 1)
    return MHIP_EINVAL;
  mhip_ctx* ctx = m->ctx;
  MHIP_HIP(ctx, hipSetDevice(ctx->device));
  const size_t nb = (size_t)n * m->cfg.img_size * m->cfg.img_size * 3;
  uint8_t* dev = nullptr;
  MHIP_HIP(ctx, hipMalloc((void**)&dev, nb));
  hipError_t e = hipMemcpy(dev, crops_host, nb, hipMemcpyHostToDevice);
  int rc = e == hipSuccess ? MHIP_OK : mhip_fail(ctx, MHIP_EHIP, "crop upload: %s", hipGetErrorString(e));
  TrocrTrace tr{trace_scores, trace_tokens, trace_beams, 0};
  if (!rc) rc = trocr_generate(m, dev, n, swap_rb, tokens_out, lengths_out, scores_out, nullptr, nullptr, &tr);
  (void)hipStreamSynchronize(ctx->stream);
  (void)hipFree(dev);
  *steps_out = tr.steps;
  return rc;
}

// Fragments of any size (u8, 3 channels, rows of row_stride bytes at base_dev + src_offset) -> Pillow bicubic resize to
// img x img (aspect ratio not preserved, as preprocess_image does) -> generate.
extern "C" int mhip_trocr_generate_fragments(mhip_trocr* m, const uint8_t* base_dev, const mhip_crop_desc* descs_host, int n,
                                             int swap_rb, int32_t* tokens_out, int32_t* lengths_out, float* scores_out) {
  if (!m || !base_dev || !descs_host || !tokens_out || !lengths_out || !scores_out || n < 1) return MHIP_EINVAL;
  mhip_ctx* ctx = m->ctx;
  MHIP_HIP(ctx, hipSetDevice(ctx->device));
  const int S = m->cfg.img_size;
  const size_t scratch = mhip_pil_resize_fragments_scratch(descs_host, n, S, S, MHIP_PIL_BICUBIC);
  const size_t cb = (size_t)n * S * S * 3;
  if (cb > m->frag_crops_bytes || scratch > m->frag_scratch_bytes) {
    MHIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (cb > m->frag_crops_bytes) {
      if (m->frag_crops) (void)hipFree(m->frag_crops);
      m->frag_crops = nullptr; m->frag_crops_bytes = 0;
      MHIP_HIP(ctx, hipMalloc((void**)&m->frag_crops, cb));
      m->frag_crops_bytes = cb;
    }
    if (scratch > m->frag_scratch_bytes) {
      if (m->frag_scratch) (void)hipFree(m->frag_scratch);
      m->frag_scratch = nullptr; m->frag_scratch_bytes = 0;
      MHIP_HIP(ctx, hipMalloc(&m->frag_scratch, scratch));
      m->frag_scratch_bytes = scratch;
    }
  }
  int rc = mhip_pil_resize_fragments(ctx, base_dev, descs_host, n, m->frag_crops, S, S, MHIP_PIL_BICUBIC, m->frag_scratch, m->frag_scratch_bytes);
  if (!rc) rc = trocr_generate(m, m->frag_crops, n, swap_rb, tokens_out, lengths_out, scores_out, nullptr, nullptr);
  return rc;
}

// ---- the same recognizer in two halves: the image encoder per batch of fragments, the decoder once over all of them ----------
// (OcrEngine's batched path: page batches leave the detector one after the other; their crops are encoded as they come — the
// encoder's GEMMs are as efficient on 300 crops as on 3000 — while the decoder, whose 16 steps are latency-bound on small
// batches (~5 ms per step whatever the batch), runs once.)
static size_t trocr_decode_ws_bytes(const mhip_trocr* m, int n) {
  VitGeom vg;
  vit_geometry(m->vit, m->cfg.img_size, m->cfg.img_size, &vg);
  return trocr_ws_bytes(m, n) - vit_workspace_bytes(m->vit, n, vg);
}

extern "C" int mhip_trocr_encode_begin(mhip_trocr* m, int max_crops) {
  if (!m || max_crops < 0) return MHIP_EINVAL;
  mhip_ctx* ctx = m->ctx;
  MHIP_HIP(ctx, hipSetDevice(ctx->device));
  m->enc_count = 0;
  if (max_crops > m->enc_cap) {
    VitGeom vg;
    vit_geometry(m->vit, m->cfg.img_size, m->cfg.img_size, &vg);
    MHIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (m->enc_store) (void)hipFree(m->enc_store);
    m->enc_store = nullptr; m->enc_cap = 0;
    MHIP_HIP(ctx, hipMalloc((void**)&m->enc_store, ((size_t)max_crops * vg.npad + 64) * m->cfg.enc_dim * m->esz()));
    m->enc_cap = max_crops;
  }
  return MHIP_OK;
}

extern "C" int mhip_trocr_encoded(mhip_trocr* m) { return m ? m->enc_count : MHIP_EINVAL; }

extern "C" int mhip_trocr_encode_fragments(mhip_trocr* m, const uint8_t* base_dev, const mhip_crop_desc* descs_host, int n,
                                           int swap_rb) {
  if (!m || !base_dev || !descs_host || n < 1) return MHIP_EINVAL;
  mhip_ctx* ctx = m->ctx;
  if (!m->ready) return mhip_fail(ctx, MHIP_ESTATE, "trocr: weights not finalized");
  MHIP_HIP(ctx, hipSetDevice(ctx->device));
  const mhip_trocr_config& c = m->cfg;
  const int S = c.img_size;
  const size_t es = m->esz();
  VitGeom vg;
  vit_geometry(m->vit, S, S, &vg);
  if (m->enc_count + n > m->enc_cap) {      // grow, keeping what has been encoded
    const int cap = std::max(m->enc_count + n, 2 * m->enc_cap);
    char* bigger = nullptr;
    MHIP_HIP(ctx, hipMalloc((void**)&bigger, ((size_t)cap * vg.npad + 64) * c.enc_dim * es));
    if (m->enc_count) MHIP_HIP(ctx, hipMemcpyAsync(bigger, m->enc_store, (size_t)m->enc_count * vg.npad * c.enc_dim * es, hipMemcpyDeviceToDevice, ctx->stream));
    MHIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (m->enc_store) (void)hipFree(m->enc_store);
    m->enc_store = bigger;
    m->enc_cap = cap;
  }
  const size_t scratch = mhip_pil_resize_fragments_scratch(descs_host, n, S, S, MHIP_PIL_BICUBIC);
  const size_t cb = (size_t)n * S * S * 3;
  if (cb > m->frag_crops_bytes || scratch > m->frag_scratch_bytes) {
    MHIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (cb > m->frag_crops_bytes) {
      if (m->frag_crops) (void)hipFree(m->frag_crops);
      m->frag_crops = nullptr; m->frag_crops_bytes = 0;
      MHIP_HIP(ctx, hipMalloc((void**)&m->frag_crops, cb));
      m->frag_crops_bytes = cb;
    }
    if (scratch > m->frag_scratch_bytes) {
      if (m->frag_scratch) (void)hipFree(m->frag_scratch);
      m->frag_scratch = nullptr; m->frag_scratch_bytes = 0;
      MHIP_HIP(ctx, hipMalloc(&m->frag_scratch, scratch));
      m->frag_scratch_bytes = scratch;
    }
  }
  int rc = mhip_pil_resize_fragments(ctx, base_dev, descs_host, n, m->frag_crops, S, S, MHIP_PIL_BICUBIC, m->frag_scratch, m->frag_scratch_bytes);
  if (rc) return rc;
  if ((rc = mhip_ensure_workspace(ctx, vit_workspace_bytes(m->vit, n, vg) + (1 << 16)))) return rc;
  Carver ws(ctx->ws);
  VitRun run;
  run.tokens_dst = m->enc_store + (size_t)m->enc_count * vg.npad * c.enc_dim * es;
  if ((rc = vit_encode(m->vit, ws, m->frag_crops, n, S, S, S, S, swap_rb, &run))) return rc;
  m->enc_count += n;
  return MHIP_OK;
}

extern "C" int mhip_trocr_decode(mhip_trocr* m, int32_t* tokens_out, int32_t* lengths_out, float* scores_out) {
  if (!m || !tokens_out || !lengths_out || !scores_out) return MHIP_EINVAL;
  mhip_ctx* ctx = m->ctx;
  if (m->enc_count < 1) return mhip_fail(ctx, MHIP_EINVAL, "trocr: nothing encoded since mhip_trocr_encode_begin");
  MHIP_HIP(ctx, hipSetDevice(ctx->device));
  const int n = m->enc_count;
  VitGeom vg;
  vit_geometry(m->vit, m->cfg.img_size, m->cfg.img_size, &vg);
  // the 64 rows behind the last crop are read (masked) by its final key tiles: keep them finite
  MHIP_HIP(ctx, hipMemsetAsync(m->enc_store + (size_t)n * vg.npad * m->cfg.enc_dim * m->esz(), 0, (size_t)64 * m->cfg.enc_dim * m->esz(), ctx->stream));
  int rc = mhip_ensure_workspace(ctx, trocr_decode_ws_bytes(m, n) + (1 << 16));
  if (rc) return rc;
  Carver ws(ctx->ws);
  rc = trocr_decode(m, ws, m->enc_store, vg, n, tokens_out, lengths_out, scores_out, nullptr, nullptr);
  m->enc_count = 0;
  return rc;
}


// The decoder's encoder-attention stage on caller-supplied host inputs, through the absorbed kernels (f16 operands): what the
// parity tests compare with fairseq's MultiheadAttention (q already projected and scaled).  q [crops*beam][heads*64],
// enc [crops][n_tok][enc_dim], wk / wv [heads*64][enc_dim], bv [heads*64] -> out [crops*beam][heads*64], all fp32.
extern "C" int mhip_cross_attention_host(mhip_ctx* ctx, const float* q, const float* enc, const float* wk, const float* wv,
                                         const float* bv, int crops, int beam, int heads, int n_tok, int enc_dim, float* out) {
  if (!ctx || !q || !enc || !wk || !wv || !bv || !out || crops < 1 || n_tok < 1) return MHIP_EINVAL;
  if (!mhip_cross_absorb_supported(enc_dim, beam, heads)) return mhip_fail(ctx, MHIP_EINVAL, "cross_attention: unsupported shape");
  MHIP_HIP(ctx, hipSetDevice(ctx->device));
  const int D = heads * 64, M = crops * beam, npad = (n_tok + 7) / 8 * 8;
  const size_t rowsE = (size_t)crops * npad + 64;
  std::vector<_Float16> hq((size_t)M * D), hE(rowsE * enc_dim, (_Float16)0.f), hkt((size_t)D * enc_dim), hwv((size_t)D * enc_dim);
  for (size_t i = 0; i < hq.size(); ++i) hq[i] = (_Float16)q[i];
  for (int c = 0; c < crops; ++c)
    for (int s = 0; s < n_tok; ++s)
      for (int d = 0; d < enc_dim; ++d) hE[((size_t)c * npad + s) * enc_dim + d] = (_Float16)enc[((size_t)c * n_tok + s) * enc_dim + d];
  const float log2e = 1.4426950408889634f;
  for (int h = 0; h < heads; ++h)
    for (int d = 0; d < enc_dim; ++d)
      for (int j = 0; j < 64; ++j) hkt[((size_t)h * enc_dim + d) * 64 + j] = (_Float16)(wk[(size_t)(h * 64 + j) * enc_dim + d] * log2e);
  for (size_t i = 0; i < hwv.size(); ++i) hwv[i] = (_Float16)wv[i];
  const size_t scr = (size_t)M * 16 * enc_dim * 2;
  size_t need = hq.size() * 2 + hE.size() * 2 + hkt.size() * 2 + hwv.size() * 2 + (size_t)D * 4 + 2 * scr + (size_t)M * D * 2 + 4096;
  int rc = mhip_ensure_workspace(ctx, need);
  if (rc) return rc;
  Carver ws(ctx->ws);
  char* dq = ws.take(hq.size() * 2); char* dE = ws.take(hE.size() * 2); char* dkt = ws.take(hkt.size() * 2);
  char* dwv = ws.take(hwv.size() * 2); float* dbv = ws.take<float>((size_t)D * 4);
  char* qt = ws.take(scr); char* ct = ws.take(scr); char* ao = ws.take((size_t)M * D * 2);
  MHIP_HIP(ctx, hipMemcpyAsync(dq, hq.data(), hq.size() * 2, hipMemcpyHostToDevice, ctx->stream));
  MHIP_HIP(ctx, hipMemcpyAsync(dE, hE.data(), hE.size() * 2, hipMemcpyHostToDevice, ctx->stream));
  MHIP_HIP(ctx, hipMemcpyAsync(dkt, hkt.data(), hkt.size() * 2, hipMemcpyHostToDevice, ctx->stream));
  MHIP_HIP(ctx, hipMemcpyAsync(dwv, hwv.data(), hwv.size() * 2, hipMemcpyHostToDevice, ctx->stream));
  MHIP_HIP(ctx, hipMemcpyAsync(dbv, bv, (size_t)D * 4, hipMemcpyHostToDevice, ctx->stream));
  CrossAbsorbDesc cd;
  cd.q = dq; cd.ldq = D; cd.E = dE; cd.kv_rows = npad; cd.n_keys = n_tok; cd.enc_dim = enc_dim;
  cd.wkt = dkt; cd.wv = dwv; cd.bv = dbv; cd.qt = qt; cd.ct = ct; cd.ao = ao; cd.ldo = D;
  cd.crops = crops; cd.beam = beam; cd.heads = heads;
  if ((rc = mhip_launch_cross_absorbed(ctx, cd))) return rc;
  std::vector<_Float16> ho((size_t)M * D);
  MHIP_HIP(ctx, hipMemcpyAsync(ho.data(), ao, ho.size() * 2, hipMemcpyDeviceToHost, ctx->stream));
  MHIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  for (size_t i = 0; i < ho.size(); ++i) out[i] = (float)ho[i];
  return MHIP_OK;
}
