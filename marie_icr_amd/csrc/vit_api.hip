// vit_api.hip — ViT encoder (BEiT / DiT backbone with its four FPN taps, and the TrOCR DeiT encoder) behind the C ABI.
//
// Host-side counterpart of BEiT.forward_features (marie/boxes/dit/ditod/beit.py:706-748: PatchEmbed + bicubic abs
// pos-emb :362-376, Block :315-341, Attention :175-260, fpn1..fpn4 :609-624) and of AdaptedVisionTransformer.
// forward_features (marie/models/unilm/trocr/deit.py:105-146).  One object = one weight arena + the launch sequence.
#include <math.h>

#include <stdlib.h>

#include "vit_internal.h"

namespace {

constexpr float LOG2E = 1.4426950408889634f;

int gemm(mhip_ctx* ctx, int prec, const void* in, const void* w, long long M, int N, int K, const float* scale,
         const float* bias, void* out, int act, int out_f32, const void* res = nullptr, int ldc = 0, int pad_ok = 0) {
  ConvDesc c;
  c.in = in; c.w = w; c.scale = scale; c.bias = bias; c.out = out;
  c.B = 1; c.H = 1; c.W = (int)M; c.Cin = K; c.N = N;
  c.relu = act; c.out_f32 = out_f32; c.res = res; c.ldc = ldc; c.pad_cols_writable = pad_ok;
  return mhip_launch_conv_igemm(ctx, prec, c);
}

std::string blk(int i, const char* s) { return "blocks." + std::to_string(i) + "." + s; }

}  // namespace

int mhip_gemm(mhip_ctx* ctx, int prec, const void* in, const void* w, long long M, int N, int K, const float* scale,
              const float* bias, void* out, int act, int out_f32, const void* res, int ldc, int pad_ok) {
  return gemm(ctx, prec, in, w, M, N, K, scale, bias, out, act, out_f32, res, ldc, pad_ok);
}

// ---------------------------------------------------------------------------------------------------- lifecycle
extern "C" int mhip_vit_create(mhip_ctx* ctx, int precision, const mhip_vit_config* cfg, mhip_vit** out) {
  if (!ctx || !out || !cfg) return MHIP_EINVAL;
  *out = nullptr;
  if (precision != MHIP_PREC_F16 && precision != MHIP_PREC_F32) return mhip_fail(ctx, MHIP_EINVAL, "unknown precision %d", precision);
  if (cfg->dim != cfg->heads * 64 || cfg->dim % 256 || cfg->dim > 1024 || cfg->depth < 1 || cfg->patch != 16 ||
      cfg->pos_h < 1 || cfg->pos_w < 1 || cfg->qkv_bias < 0 || cfg->qkv_bias > 2)
    return mhip_fail(ctx, MHIP_EINVAL, "vit: unsupported configuration (dim %d heads %d patch %d)", cfg->dim, cfg->heads, cfg->patch);
  if (cfg->fpn)
    for (int j = 0; j < 4; ++j)
      if (cfg->taps[j] < 0 || cfg->taps[j] >= cfg->depth) return mhip_fail(ctx, MHIP_EINVAL, "vit: tap %d out of range", cfg->taps[j]);
  mhip_vit* m = new mhip_vit();
  m->ctx = ctx;
  m->precision = precision;
  m->cfg = *cfg;
  m->x16 = precision == MHIP_PREC_F16 && getenv("MARIE_HIP_RESIDUAL_F16") != nullptr;
  m->fold = precision == MHIP_PREC_F16 && !m->x16 && getenv("MARIE_HIP_NO_LN_FOLD") == nullptr;
  const size_t es = m->esz(), D = cfg->dim, K0 = 3 * 16 * 16;
  Arena& a = m->arena;
  if (m->fold) {
    a.take("cls_stats", D / 64 * 2 * 4);
    for (int i = 0; i < cfg->depth; ++i) {
      a.take(blk(i, "qk_cs"), 2 * D * 4);
      a.take(blk(i, "v_cs"), D * 4); a.take(blk(i, "v_rb"), D * 4);
      a.take(blk(i, "fc1_cs"), 4 * D * 4);
    }
  }
  a.take("pe_w", D * K0 * es);
  a.take("pe_b", D * 4);
  a.take("pos", (size_t)cfg->pos_h * cfg->pos_w * D * 4);
  a.take("cls", D * 4);
  for (int i = 0; i < cfg->depth; ++i) {
    a.take(blk(i, "ln1_g"), D * 4); a.take(blk(i, "ln1_b"), D * 4);
    a.take(blk(i, "qk_w"), 2 * D * D * es); a.take(blk(i, "qk_b"), 2 * D * 4);
    a.take(blk(i, "v_w"), D * D * es);
    a.take(blk(i, "proj_w"), D * D * es); a.take(blk(i, "proj_s"), D * 4); a.take(blk(i, "proj_b"), D * 4);
    a.take(blk(i, "ln2_g"), D * 4); a.take(blk(i, "ln2_b"), D * 4);
    a.take(blk(i, "fc1_w"), 4 * D * D * es); a.take(blk(i, "fc1_b"), 4 * D * 4);
    a.take(blk(i, "fc2_w"), 4 * D * D * es); a.take(blk(i, "fc2_s"), D * 4); a.take(blk(i, "fc2_b"), D * 4);
  }
  if (cfg->final_norm) { a.take("norm_g", D * 4); a.take("norm_b", D * 4); }
  if (cfg->fpn)
    for (const char* n : {"f1a", "f1b", "f2"}) {
      a.take(std::string(n) + "_w", 4 * D * D * es);
      a.take(std::string(n) + "_s", 4 * D * 4);
      a.take(std::string(n) + "_b", 4 * D * 4);
    }
  *out = m;
  return MHIP_OK;
}

extern "C" int mhip_vit_destroy(mhip_vit* m) {
  if (!m) return MHIP_OK;
  mhip_quiesce(m->ctx);
  m->arena.release();
  for (auto& t : m->pos_tables) { (void)hipFree(t.dev); if (t.dev16) (void)hipFree(t.dev16); }
  delete m;
  return MHIP_OK;
}

extern "C" int mhip_vit_set_tensor(mhip_vit* m, const char* key, const float* data, const int64_t* shape, int ndim) {
  if (!m || !key) return MHIP_EINVAL;
  std::string k(key);
  if (k.size() > 19 && k.compare(k.size() - 19, 19, "num_batches_tracked") == 0) return MHIP_OK;
  if (k.find("relative_position") != std::string::npos)
    return mhip_fail(m->ctx, MHIP_EINVAL, "vit: relative position bias (%s) is not part of this build (abs pos-emb only)", key);
  m->ready = false;
  return m->store.set(m->ctx, k, data, shape, ndim);
}

extern "C" int mhip_vit_alloc_arena(mhip_vit* m) {
  if (!m) return MHIP_EINVAL;
  int rc = m->arena.alloc(m->ctx);
  if (rc) return rc;
  m->ready = true;
  return MHIP_OK;
}

extern "C" int mhip_vit_arena(mhip_vit* m, void** dev, size_t* bytes) {
  if (!m) return MHIP_EINVAL;
  if (dev) *dev = m->arena.dev;
  if (bytes) *bytes = m->arena.bytes;
  return MHIP_OK;
}

// ConvTranspose2d(k=2, s=2) weight [Cin][Cout][2][2] -> GEMM weight [n = (dy*2+dx)*Cout + co][ci]
static void pack_convT(const mhip_vit* m, Arena& a, const std::string& name, const HostTensor& w, const HostTensor& b,
                       const float* ch_scale, const float* ch_shift) {
  const int D = m->cfg.dim;
  std::vector<float> tmp((size_t)4 * D * D);
  for (int ci = 0; ci < D; ++ci)
    for (int co = 0; co < D; ++co)
      for (int q = 0; q < 4; ++q) tmp[((size_t)q * D + co) * D + ci] = w.data[((size_t)ci * D + co) * 4 + q];
  Arena::put(m->precision, a.h(name + "_w"), tmp.data(), tmp.size());
  float* s = (float*)a.h(name + "_s");
  float* sh = (float*)a.h(name + "_b");
  for (int q = 0; q < 4; ++q)
    for (int co = 0; co < D; ++co) {
      const float sc = ch_scale ? ch_scale[co] : 1.f;
      s[q * D + co] = sc;
      sh[q * D + co] = (ch_shift ? ch_shift[co] : 0.f) + b.data[co] * sc;
    }
}

extern "C" int mhip_vit_finalize(mhip_vit* m) {
  if (!m) return MHIP_EINVAL;
  mhip_ctx* ctx = m->ctx;
  const mhip_vit_config& c = m->cfg;
  const int D = c.dim, prec = m->precision;
  const size_t es = m->esz();
  Arena& a = m->arena;
  const TensorStore& st = m->store;
  a.begin_fill();
  const HostTensor* pw = st.find(ctx, "patch_embed.proj.weight", {D, 3, 16, 16});
  const HostTensor* pb = st.find(ctx, "patch_embed.proj.bias", {D});
  const HostTensor* pos = st.find(ctx, "pos_embed", {1, 1 + c.pos_h * c.pos_w, D});
  const HostTensor* cls = st.find(ctx, "cls_token", {1, 1, D});
  if (!pw || !pb || !pos || !cls) return MHIP_ESTATE;
  Arena::put(prec, a.h("pe_w"), pw->data.data(), pw->numel());
  memcpy(a.h("pe_b"), pb->data.data(), D * 4);
  memcpy(a.h("pos"), pos->data.data() + D, (size_t)c.pos_h * c.pos_w * D * 4);
  for (int d = 0; d < D; ++d) ((float*)a.h("cls"))[d] = cls->data[d] + pos->data[d];
  if (m->fold) {   // row statistics of the cls row per 64-column chunk, as conv_igemm's split epilogue produces them for the other rows
    const float* cr = (const float*)a.h("cls");
    float* cst = (float*)a.h("cls_stats");
    for (int c = 0; c < D / 64; ++c) {
      float sm = 0.f, m2 = 0.f;
      for (int d = 0; d < 64; ++d) sm += cr[c * 64 + d];
      for (int d = 0; d < 64; ++d) { const float t = cr[c * 64 + d] - sm / 64.f; m2 += t * t; }
      cst[c * 2] = sm; cst[c * 2 + 1] = m2;
    }
  }
  const float qs = 0.125f * LOG2E;   // head_dim^-0.5 (64 -> 1/8) and the exp -> exp2 change of base, folded into W_q
  for (int i = 0; i < c.depth; ++i) {
    const HostTensor* g1 = st.find(ctx, blk(i, "norm1.weight"), {D});
    const HostTensor* b1 = st.find(ctx, blk(i, "norm1.bias"), {D});
    const HostTensor* qkv = st.find(ctx, blk(i, "attn.qkv.weight"), {3 * D, D});
    const HostTensor* pjw = st.find(ctx, blk(i, "attn.proj.weight"), {D, D});
    const HostTensor* pjb = st.find(ctx, blk(i, "attn.proj.bias"), {D});
    const HostTensor* g2 = st.find(ctx, blk(i, "norm2.weight"), {D});
    const HostTensor* b2 = st.find(ctx, blk(i, "norm2.bias"), {D});
    const HostTensor* f1w = st.find(ctx, blk(i, "mlp.fc1.weight"), {4 * D, D});
    const HostTensor* f1b = st.find(ctx, blk(i, "mlp.fc1.bias"), {4 * D});
    const HostTensor* f2w = st.find(ctx, blk(i, "mlp.fc2.weight"), {D, 4 * D});
    const HostTensor* f2b = st.find(ctx, blk(i, "mlp.fc2.bias"), {D});
    if (!g1 || !b1 || !qkv || !pjw || !pjb || !g2 || !b2 || !f1w || !f1b || !f2w || !f2b) return MHIP_ESTATE;
    std::vector<float> qb(D, 0.f), kb(D, 0.f), vb(D, 0.f);
    if (c.qkv_bias == 1) {
      const HostTensor* q = st.find(ctx, blk(i, "attn.q_bias"), {D});
      const HostTensor* v = st.find(ctx, blk(i, "attn.v_bias"), {D});
      if (!q || !v) return MHIP_ESTATE;
      qb = q->data; vb = v->data;
    } else if (c.qkv_bias == 2) {
      const HostTensor* b = st.find(ctx, blk(i, "attn.qkv.bias"), {3 * D});
      if (!b) return MHIP_ESTATE;
      qb.assign(b->data.begin(), b->data.begin() + D);
      kb.assign(b->data.begin() + D, b->data.begin() + 2 * D);
      vb.assign(b->data.begin() + 2 * D, b->data.end());
    }
    std::vector<float> gam1(D, 1.f), gam2(D, 1.f);
    if (c.layer_scale) {
      const HostTensor* ga = st.find(ctx, blk(i, "gamma_1"), {D});
      const HostTensor* gb = st.find(ctx, blk(i, "gamma_2"), {D});
      if (!ga || !gb) return MHIP_ESTATE;
      gam1 = ga->data; gam2 = gb->data;
    }
    memcpy(a.h(blk(i, "ln1_g")), g1->data.data(), D * 4);
    memcpy(a.h(blk(i, "ln1_b")), b1->data.data(), D * 4);
    memcpy(a.h(blk(i, "ln2_g")), g2->data.data(), D * 4);
    memcpy(a.h(blk(i, "ln2_b")), b2->data.data(), D * 4);
    std::vector<float> wq((size_t)D * D);
    for (size_t e = 0; e < wq.size(); ++e) wq[e] = qkv->data[e] * qs;
    Arena::put(prec, a.h(blk(i, "qk_w")), wq.data(), wq.size());
    Arena::put(prec, a.h(blk(i, "qk_w")) + (size_t)D * D * es, qkv->data.data() + (size_t)D * D, (size_t)D * D);
    float* qkb = (float*)a.h(blk(i, "qk_b"));
    for (int d = 0; d < D; ++d) { qkb[d] = qb[d] * qs; qkb[D + d] = kb[d]; }
    Arena::put(prec, a.h(blk(i, "v_w")), qkv->data.data() + (size_t)2 * D * D, (size_t)D * D);
    Arena::put(prec, a.h(blk(i, "proj_w")), pjw->data.data(), pjw->numel());
    // softmax rows sum to one, so the value bias passes through attention unchanged:  proj(o + b_v) = proj(o) + W_p b_v
    float* ps = (float*)a.h(blk(i, "proj_s"));
    float* pbb = (float*)a.h(blk(i, "proj_b"));
    for (int o = 0; o < D; ++o) {
      double acc = pjb->data[o];
      for (int k = 0; k < D; ++k) acc += (double)pjw->data[(size_t)o * D + k] * vb[k];
      ps[o] = gam1[o];
      pbb[o] = (float)acc * gam1[o];
    }
    Arena::put(prec, a.h(blk(i, "fc1_w")), f1w->data.data(), f1w->numel());
    memcpy(a.h(blk(i, "fc1_b")), f1b->data.data(), (size_t)4 * D * 4);
    if (m->fold) {
      // LayerNorm folded around its consumer GEMMs:  LN(x) W^T + b = rstd (x (g*W)^T - mean colsum(g*W)) + (b + W beta).
      // The column sums are those of the ROUNDED folded weights — what the matrix cores multiply — so that the mean term cancels
      // exactly; the beta term keeps the unrounded weights (it is added in fp32).
      auto r16 = [](float v) { return (float)(_Float16)v; };
      auto fold_rows = [&](const float* W, int rows, const std::vector<float>& gam, const std::vector<float>& bet, float pre,
                           char* w_dst, float* cs, float* beta_dot) {
        std::vector<float> wf((size_t)rows * D);
        for (int n = 0; n < rows; ++n) {
          double c = 0.0, bd = 0.0;
          for (int k = 0; k < D; ++k) {
            const float w = W[(size_t)n * D + k] * pre * gam[k];
            wf[(size_t)n * D + k] = w;
            c += r16(w);
            bd += (double)W[(size_t)n * D + k] * pre * bet[k];
          }
          cs[n] = (float)c;
          beta_dot[n] = (float)bd;
        }
        Arena::put(prec, w_dst, wf.data(), wf.size());
      };
      std::vector<float> bd((size_t)4 * D);
      float* qkcs = (float*)a.h(blk(i, "qk_cs"));
      fold_rows(qkv->data.data(), D, g1->data, b1->data, qs, a.h(blk(i, "qk_w")), qkcs, bd.data());
      for (int d = 0; d < D; ++d) qkb[d] += bd[d];
      fold_rows(qkv->data.data() + (size_t)D * D, D, g1->data, b1->data, 1.f, a.h(blk(i, "qk_w")) + (size_t)D * D * es, qkcs + D, bd.data());
      for (int d = 0; d < D; ++d) qkb[D + d] += bd[d];
      fold_rows(qkv->data.data() + (size_t)2 * D * D, D, g1->data, b1->data, 1.f, a.h(blk(i, "v_w")), (float*)a.h(blk(i, "v_cs")),
                (float*)a.h(blk(i, "v_rb")));
      fold_rows(f1w->data.data(), 4 * D, g2->data, b2->data, 1.f, a.h(blk(i, "fc1_w")), (float*)a.h(blk(i, "fc1_cs")), bd.data());
      float* f1bb = (float*)a.h(blk(i, "fc1_b"));
      for (int d = 0; d < 4 * D; ++d) f1bb[d] += bd[d];
    }
    Arena::put(prec, a.h(blk(i, "fc2_w")), f2w->data.data(), f2w->numel());
    float* fs = (float*)a.h(blk(i, "fc2_s"));
    float* fb = (float*)a.h(blk(i, "fc2_b"));
    for (int o = 0; o < D; ++o) { fs[o] = gam2[o]; fb[o] = f2b->data[o] * gam2[o]; }
  }
  if (c.final_norm) {
    const HostTensor* g = st.find(ctx, "norm.weight", {D});
    const HostTensor* b = st.find(ctx, "norm.bias", {D});
    if (!g || !b) return MHIP_ESTATE;
    memcpy(a.h("norm_g"), g->data.data(), D * 4);
    memcpy(a.h("norm_b"), b->data.data(), D * 4);
  }
  if (c.fpn) {
    const HostTensor* w0 = st.find(ctx, "fpn1.0.weight", {D, D, 2, 2});
    const HostTensor* b0 = st.find(ctx, "fpn1.0.bias", {D});
    const HostTensor* bg = st.find(ctx, "fpn1.1.weight", {D});
    const HostTensor* bb = st.find(ctx, "fpn1.1.bias", {D});
    const HostTensor* bm = st.find(ctx, "fpn1.1.running_mean", {D});
    const HostTensor* bv = st.find(ctx, "fpn1.1.running_var", {D});
    const HostTensor* w3 = st.find(ctx, "fpn1.3.weight", {D, D, 2, 2});
    const HostTensor* b3 = st.find(ctx, "fpn1.3.bias", {D});
    const HostTensor* w2 = st.find(ctx, "fpn2.0.weight", {D, D, 2, 2});
    const HostTensor* b2 = st.find(ctx, "fpn2.0.bias", {D});
    if (!w0 || !b0 || !bg || !bb || !bm || !bv || !w3 || !b3 || !w2 || !b2) return MHIP_ESTATE;
    std::vector<float> sc(D), sh(D);
    for (int o = 0; o < D; ++o) {
      sc[o] = bg->data[o] / sqrtf(bv->data[o] + 1e-5f);
      sh[o] = bb->data[o] - bm->data[o] * sc[o];
    }
    pack_convT(m, a, "f1a", *w0, *b0, sc.data(), sh.data());
    pack_convT(m, a, "f1b", *w3, *b3, nullptr, nullptr);
    pack_convT(m, a, "f2", *w2, *b2, nullptr, nullptr);
  }
  int rc = a.upload(ctx);
  if (rc) return rc;
  m->ready = true;
  m->store.t.clear();
  for (auto& t : m->pos_tables) { (void)hipFree(t.dev); if (t.dev16) (void)hipFree(t.dev16); }
  m->pos_tables.clear();
  return MHIP_OK;
}

// ---------------------------------------------------------------------------------------------------- forward
void vit_geometry(const mhip_vit* m, int H32, int W32, VitGeom* g) {
  g->hp = H32 / m->cfg.patch;
  g->wp = W32 / m->cfg.patch;
  g->np = g->hp * g->wp;
  g->n_tok = g->np + 1;
  g->npad = (g->n_tok + 7) / 8 * 8;      // 16-byte aligned V^T columns per image; tiles may run into the next image's rows
}

size_t vit_workspace_bytes(const mhip_vit* m, int B, const VitGeom& g) {
  const size_t es = m->esz(), D = m->cfg.dim, R = (size_t)B * g.npad;
  size_t b = 0;
  auto add = [&](size_t n) { b += (n + 255) / 256 * 256; };
  add(R * D * 4);              // x
  add(R * D * es);             // ln / attention output
  add((R + 128) * 2 * D * es); // q | k (+ slack: the last image's final query block / key tile reads past its rows)
  add((D * R + 128) * es);     // v^T (+ slack)
  add(R * D * es);             // attention output
  add(R * 4 * D * es);         // mlp hidden; also the patch matrix
  if (m->cfg.fpn) add(4 * (size_t)B * g.np * D * es);
  else add((R + 64) * D * es); // final tokens (+ slack rows)
  if (m->fold) { add(D / 64 * R * 8); add(R * 4); add(R * 4); }   // row statistics per chunk, rstd, mean * rstd
  return b;
}

int vit_encode(mhip_vit* m, Carver& ws, const uint8_t* imgs, int B, int th, int tw, int H32, int W32, int swap_rb,
               VitRun* run) {
  mhip_ctx* ctx = m->ctx;
  if (!m->ready) return mhip_fail(ctx, MHIP_ESTATE, "vit: weights not finalized");
  const mhip_vit_config& c = m->cfg;
  const int D = c.dim, prec = m->precision, P = c.patch;
  const size_t es = m->esz();
  if (B < 1 || th < 1 || tw < 1 || H32 % P || W32 % P || th > H32 || tw > W32) return mhip_fail(ctx, MHIP_EINVAL, "vit: bad image geometry");
  VitGeom g;
  vit_geometry(m, H32, W32, &g);
  run->g = g;
  const Arena& a = m->arena;
  float* pos_dev = nullptr;
  const bool x16 = m->x16 && prec == MHIP_PREC_F16;
  void* pos16 = nullptr;
  for (const auto& t : m->pos_tables)
    if (t.hp == g.hp && t.wp == g.wp) { pos_dev = t.dev; pos16 = t.dev16; }
  if (!pos_dev) {   // first page of this geometry: resize the position table once and keep it
    if (m->pos_tables.size() >= 64) return mhip_fail(ctx, MHIP_ENOMEM, "vit: more than 64 distinct page geometries");
    MHIP_HIP(ctx, hipMalloc((void**)&pos_dev, (size_t)g.np * D * 4));
    int rc = mhip_launch_posemb_bicubic(ctx, a.d<float>("pos"), c.pos_h, c.pos_w, pos_dev, g.hp, g.wp, D);
    if (rc) return rc;
    if (x16) {
      MHIP_HIP(ctx, hipMalloc(&pos16, (size_t)g.np * D * 2));
      if ((rc = mhip_launch_narrow_f16(ctx, pos_dev, pos16, (long long)g.np * D))) return rc;
    } else if (m->fold) {
      MHIP_HIP(ctx, hipMalloc(&pos16, (size_t)g.np * D * 4));
      if ((rc = mhip_launch_split_f16(ctx, pos_dev, pos16, (char*)pos16 + (size_t)g.np * D * 2, (long long)g.np * D))) return rc;
    }
    m->pos_tables.push_back({g.hp, g.wp, pos_dev, pos16});
  }
  const size_t R = (size_t)B * g.npad;
  void* x = ws.take<float>(R * D * 4);      // fp32 stream, or f16 in the first half of it
  char* ln = ws.take(R * D * es);
  char* qk = ws.take((R + 128) * 2 * D * es);
  char* vt = ws.take((D * R + 128) * es);
  // rows / columns past the last image are read by its final tiles (and masked): keep them finite
  MHIP_HIP(ctx, hipMemsetAsync(qk + R * 2 * D * es, 0, (size_t)128 * 2 * D * es, ctx->stream));
  MHIP_HIP(ctx, hipMemsetAsync(vt + (size_t)D * R * es, 0, 128 * es, ctx->stream));
  char* ao = ws.take(R * D * es);
  char* hid = ws.take(R * 4 * D * es);
  run->x = x;
  const bool fold = m->fold && prec == MHIP_PREC_F16 && !x16;
  const int chunks = D / 64;
  char* xlo = fold ? (char*)x + R * D * 2 : nullptr;          // the split stream: two f16 planes in the bytes of the fp32 one
  run->x_lo = xlo;
  float* stats = nullptr; float* rstd = nullptr; float* mur = nullptr;
  if (fold) {
    stats = ws.take<float>((size_t)chunks * R * 8);
    rstd = ws.take<float>(R * 4);
    mur = ws.take<float>(R * 4);
  }
  int rc;
  // patches -> hid (as the [B*np][768] patch matrix) -> x rows 1.. with bias + resized position table
  const int K0 = 3 * P * P;
  {   // all images at once: row q of the patch matrix -> token row (q / np) * npad + 1 + q % np, + position row q % np
    if ((rc = mhip_launch_patchify(ctx, prec, imgs, B, th, tw, g.hp, g.wp, P, swap_rb, 127.5f, 127.5f, hid, K0))) return rc;
    ConvDesc cd;
    cd.in = hid; cd.w = a.d("pe_w"); cd.bias = a.d<float>("pe_b"); cd.out = x; cd.res = (x16 || fold) ? pos16 : (void*)pos_dev;
    cd.B = 1; cd.H = 1; cd.W = B * g.np; cd.Cin = K0; cd.N = D; cd.out_f32 = (x16 || fold) ? 0 : 1;
    cd.row_period = g.np; cd.row_stride = g.npad; cd.row_offset = 1;
    if (fold) { cd.epi = EPI_SPLIT; cd.out2 = xlo; cd.res2 = (char*)pos16 + (size_t)g.np * D * 2; cd.stats = stats; cd.stats_ld = (int)R; }
    if ((rc = mhip_launch_conv_igemm(ctx, prec, cd))) return rc;
  }
  if (fold) {
    if ((rc = mhip_launch_token_init_split(ctx, x, xlo, a.d<float>("cls"), a.d<float>("cls_stats"), stats, (int)R, B, g.npad, g.n_tok, D))) return rc;
  } else if ((rc = mhip_launch_token_init(ctx, x, a.d<float>("cls"), B, g.npad, g.n_tok, D, x16))) return rc;
  // split stream: the producers of x (GEMMs with the split epilogue: out = x in place, statistics of the new rows) and the
  // consumers of LN(x) (GEMMs over the high plane with the folded epilogues)
  auto produce = [&](const void* in, const void* w, int K, const float* sc, const float* bi) {
    ConvDesc c;
    c.in = in; c.w = w; c.scale = sc; c.bias = bi; c.out = x; c.out2 = xlo; c.res = x; c.res2 = xlo;
    c.B = 1; c.H = 1; c.W = (int)R; c.Cin = K; c.N = D;
    c.epi = EPI_SPLIT; c.stats = stats; c.stats_ld = (int)R;
    return mhip_launch_conv_igemm(ctx, prec, c);
  };
  auto consume_rows = [&](const void* w, int N, const float* cs, const float* bi, void* out, int act) {
    ConvDesc c;
    c.in = x; c.w = w; c.bias = bi; c.out = out;
    c.B = 1; c.H = 1; c.W = (int)R; c.Cin = D; c.N = N; c.relu = act;
    c.epi = EPI_LN_ROWS; c.ln_a = rstd; c.ln_b = mur; c.ln_cs = cs;
    return mhip_launch_conv_igemm(ctx, prec, c);
  };
  int tap_at = 0;
  if (c.fpn) for (int j = 0; j < 4; ++j) run->tap[j] = ws.take((size_t)B * g.np * D * es);
  for (int i = 0; i < c.depth; ++i) {
    if (fold) {
      if ((rc = mhip_launch_ln_finalize(ctx, stats, chunks, (int)R, rstd, mur, (int)R, D, c.ln_eps))) return rc;
      if ((rc = consume_rows(a.d(blk(i, "qk_w")), 2 * D, a.d<float>(blk(i, "qk_cs")), a.d<float>(blk(i, "qk_b")), qk, ACT_NONE))) return rc;
      ConvDesc cv;      // V^T = W_v LN(X)^T: the tokens are the GEMM's columns
      cv.in = a.d(blk(i, "v_w")); cv.w = x; cv.out = vt;
      cv.B = 1; cv.H = 1; cv.W = D; cv.Cin = D; cv.N = (int)R;
      cv.epi = EPI_LN_COLS; cv.ln_a = rstd; cv.ln_b = mur; cv.ln_cs = a.d<float>(blk(i, "v_cs")); cv.row_bias = a.d<float>(blk(i, "v_rb"));
      if ((rc = mhip_launch_conv_igemm(ctx, prec, cv))) return rc;
    } else {
    if ((rc = mhip_launch_layernorm(ctx, prec, x, a.d<float>(blk(i, "ln1_g")), a.d<float>(blk(i, "ln1_b")), ln, (int)R, D, c.ln_eps, x16))) return rc;
    if ((rc = gemm(ctx, prec, ln, a.d(blk(i, "qk_w")), (long long)R, 2 * D, D, nullptr, a.d<float>(blk(i, "qk_b")), qk, ACT_NONE, 0))) return rc;
    if ((rc = gemm(ctx, prec, a.d(blk(i, "v_w")), ln, D, (int)R, D, nullptr, nullptr, vt, ACT_NONE, 0))) return rc;   // V^T = W_v X^T
    }
    AttnDesc ad;
    ad.q = qk; ad.k = qk + (size_t)D * es; ad.vt = vt; ad.out = ao;
    ad.ldq = ad.ldk = 2 * D; ad.ldv = (int)R; ad.ldo = D;
    ad.images = B; ad.heads = c.heads; ad.npad_q = ad.npad_k = g.npad; ad.n_queries = ad.n_keys = g.n_tok;
    if ((rc = mhip_launch_attention(ctx, prec, ad))) return rc;
    if (fold) {
      if ((rc = produce(ao, a.d(blk(i, "proj_w")), D, a.d<float>(blk(i, "proj_s")), a.d<float>(blk(i, "proj_b"))))) return rc;
      if ((rc = mhip_launch_ln_finalize(ctx, stats, chunks, (int)R, rstd, mur, (int)R, D, c.ln_eps))) return rc;
      if ((rc = consume_rows(a.d(blk(i, "fc1_w")), 4 * D, a.d<float>(blk(i, "fc1_cs")), a.d<float>(blk(i, "fc1_b")), hid, ACT_GELU))) return rc;
      if ((rc = produce(hid, a.d(blk(i, "fc2_w")), 4 * D, a.d<float>(blk(i, "fc2_s")), a.d<float>(blk(i, "fc2_b"))))) return rc;
    } else {
    if ((rc = gemm(ctx, prec, ao, a.d(blk(i, "proj_w")), (long long)R, D, D, a.d<float>(blk(i, "proj_s")), a.d<float>(blk(i, "proj_b")), x, ACT_NONE, x16 ? 0 : 1, x))) return rc;
    if ((rc = mhip_launch_layernorm(ctx, prec, x, a.d<float>(blk(i, "ln2_g")), a.d<float>(blk(i, "ln2_b")), ln, (int)R, D, c.ln_eps, x16))) return rc;
    if ((rc = gemm(ctx, prec, ln, a.d(blk(i, "fc1_w")), (long long)R, 4 * D, D, nullptr, a.d<float>(blk(i, "fc1_b")), hid, ACT_GELU, 0))) return rc;
    if ((rc = gemm(ctx, prec, hid, a.d(blk(i, "fc2_w")), (long long)R, D, 4 * D, a.d<float>(blk(i, "fc2_s")), a.d<float>(blk(i, "fc2_b")), x, ACT_NONE, x16 ? 0 : 1, x))) return rc;
    }
    if (c.fpn)
      for (int j = 0; j < 4; ++j)
        if (c.taps[j] == i) {
          // a map in f16 of a split stream is its high plane (hi = f16(x))
          if ((rc = mhip_launch_tokens_to_map(ctx, prec, x, run->tap[j], B, g.npad, g.np, D, x16 || fold))) return rc;
          ++tap_at;
        }
  }
  run->tokens = nullptr;
  if (c.final_norm) {
    // + 64 finite rows: the decoder's encoder-attention walks every image's tokens in 32-row tiles and so reads (masked) rows
    // past the last image
    if (run->tokens_dst) {
      run->tokens = run->tokens_dst;
    } else {
      run->tokens = ws.take((R + 64) * D * es);
      MHIP_HIP(ctx, hipMemsetAsync(run->tokens + R * D * es, 0, (size_t)64 * D * es, ctx->stream));
    }
    if ((rc = mhip_launch_layernorm(ctx, prec, x, a.d<float>("norm_g"), a.d<float>("norm_b"), run->tokens, (int)R, D, c.ln_eps, x16 || fold, xlo))) return rc;
  }
  (void)tap_at;
  return MHIP_OK;
}

size_t vit_fpn_workspace_bytes(const mhip_vit* m, int B, const VitGeom& g) {
  const size_t es = m->esz(), D = m->cfg.dim, M = (size_t)B * g.np;
  return (4 * M * D * es + 256) + (16 * M * D * es + 256) + (4 * M * D * es + 256) + (M / 4 * D * es + 256) + 4096;
}

// fpn1..fpn4 on the four taps.  ConvTranspose2d(2, 2) is per-pixel: a GEMM to 4*D columns whose output, read as
// [4*rows][D], is the 2x-upsampled map in NESTED order (row = parent*4 + dy*2 + dx).  Consumers that are per-pixel
// themselves (the FPN lateral 1x1) take it as is; mhip_launch_unnest restores raster order.
int vit_fpn(mhip_vit* m, Carver& ws, int B, const VitRun& run, VitFpnOut* out) {
  mhip_ctx* ctx = m->ctx;
  const int D = m->cfg.dim, prec = m->precision;
  const size_t es = m->esz();
  const VitGeom& g = run.g;
  const long long M = (long long)B * g.np;
  const Arena& a = m->arena;
  char* t1 = ws.take(4 * M * D * es);
  char* o1 = ws.take(16 * M * D * es);
  char* o2 = ws.take(4 * M * D * es);
  int rc;
  if ((rc = gemm(ctx, prec, run.tap[0], a.d("f1a_w"), M, 4 * D, D, a.d<float>("f1a_s"), a.d<float>("f1a_b"), t1, ACT_GELU, 0))) return rc;
  if ((rc = gemm(ctx, prec, t1, a.d("f1b_w"), 4 * M, 4 * D, D, a.d<float>("f1b_s"), a.d<float>("f1b_b"), o1, ACT_NONE, 0))) return rc;
  if ((rc = gemm(ctx, prec, run.tap[1], a.d("f2_w"), M, 4 * D, D, a.d<float>("f2_s"), a.d<float>("f2_b"), o2, ACT_NONE, 0))) return rc;
  out->level[0] = o1; out->nest[0] = 2;
  out->level[1] = o2; out->nest[1] = 1;
  out->level[2] = run.tap[2]; out->nest[2] = 0;
  const int h4 = g.hp / 2, w4 = g.wp / 2;   // MaxPool2d(2, 2) floors odd sizes
  char* o4 = ws.take((size_t)B * h4 * w4 * D * es);
  if ((rc = mhip_launch_maxpool(ctx, prec, 2, run.tap[3], o4, B, g.hp, g.wp, D))) return rc;
  out->level[3] = o4; out->nest[3] = 0;
  out->h[0] = 4 * g.hp; out->w[0] = 4 * g.wp;
  out->h[1] = 2 * g.hp; out->w[1] = 2 * g.wp;
  out->h[2] = g.hp; out->w[2] = g.wp;
  out->h[3] = h4; out->w[3] = w4;
  return MHIP_OK;
}

// ---------------------------------------------------------------------------------------------------- test / host entry
extern "C" int mhip_vit_forward_host(mhip_vit* m, const uint8_t* imgs_host, int B, int th, int tw, int H32, int W32,
                                     int swap_rb, float* tokens_out, float* fpn0, float* fpn1, float* fpn2, float* fpn3) {
  if (!m || !imgs_host) return MHIP_EINVAL;
  mhip_ctx* ctx = m->ctx;
  MHIP_HIP(ctx, hipSetDevice(ctx->device));
  VitGeom g;
  vit_geometry(m, H32, W32, &g);
  const size_t D = m->cfg.dim;
  const size_t img_bytes = (size_t)B * th * tw * 3;
  const size_t out_f32 = m->cfg.fpn ? (size_t)B * 16 * g.np * D * 4 : (size_t)B * g.npad * D * 4;
  const size_t need = vit_workspace_bytes(m, B, g) + (m->cfg.fpn ? vit_fpn_workspace_bytes(m, B, g) : 0) + img_bytes + out_f32 + 4096;
  int rc = mhip_ensure_workspace(ctx, need);
  if (rc) return rc;
  Carver ws(ctx->ws);
  uint8_t* imgs = ws.take<uint8_t>(img_bytes);
  float* stage = ws.take<float>(out_f32);
  MHIP_HIP(ctx, hipMemcpyAsync(imgs, imgs_host, img_bytes, hipMemcpyHostToDevice, ctx->stream));
  VitRun run;
  if ((rc = vit_encode(m, ws, imgs, B, th, tw, H32, W32, swap_rb, &run))) return rc;
  if (tokens_out) {
    // final tokens (after the last norm if configured, else the residual stream), valid rows only
    for (int b = 0; b < B; ++b) {
      if (m->cfg.final_norm) {
        if ((rc = mhip_launch_convert_rows(ctx, m->precision, run.tokens + (size_t)b * g.npad * D * m->esz(), stage, g.n_tok, (int)D))) return rc;
        MHIP_HIP(ctx, hipMemcpyAsync(tokens_out + (size_t)b * g.n_tok * D, stage, (size_t)g.n_tok * D * 4, hipMemcpyDeviceToHost, ctx->stream));
        MHIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
      } else {
        if (run.x_lo) {      // split stream: hi + lo through the staging buffer
          const size_t o = (size_t)b * g.npad * D * 2;
          if ((rc = mhip_launch_join_f16(ctx, (const char*)run.x + o, (const char*)run.x_lo + o, stage, (long long)g.n_tok * D))) return rc;
          MHIP_HIP(ctx, hipMemcpyAsync(tokens_out + (size_t)b * g.n_tok * D, stage, (size_t)g.n_tok * D * 4, hipMemcpyDeviceToHost, ctx->stream));
          MHIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
        } else if (m->x16 && m->precision == MHIP_PREC_F16) {      // f16 stream: widen through the staging buffer
          if ((rc = mhip_launch_convert_rows(ctx, m->precision, (const char*)run.x + (size_t)b * g.npad * D * 2, stage, g.n_tok, (int)D))) return rc;
          MHIP_HIP(ctx, hipMemcpyAsync(tokens_out + (size_t)b * g.n_tok * D, stage, (size_t)g.n_tok * D * 4, hipMemcpyDeviceToHost, ctx->stream));
          MHIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
        } else {
          MHIP_HIP(ctx, hipMemcpyAsync(tokens_out + (size_t)b * g.n_tok * D, (const float*)run.x + (size_t)b * g.npad * D, (size_t)g.n_tok * D * 4, hipMemcpyDeviceToHost, ctx->stream));
        }
      }
    }
  }
  if (m->cfg.fpn) {
    VitFpnOut fo;
    if ((rc = vit_fpn(m, ws, B, run, &fo))) return rc;
    float* outs[4] = {fpn0, fpn1, fpn2, fpn3};
    for (int j = 0; j < 4; ++j) {
      if (!outs[j]) continue;
      if ((rc = mhip_launch_unnest(ctx, m->precision, fo.level[j], nullptr, stage, 1, B, fo.h[j], fo.w[j], (int)D, fo.nest[j]))) return rc;
      MHIP_HIP(ctx, hipMemcpyAsync(outs[j], stage, (size_t)B * fo.h[j] * fo.w[j] * D * 4, hipMemcpyDeviceToHost, ctx->stream));
      MHIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
  }
  MHIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return MHIP_OK;
}
