// dit_api.hip — the DiT Mask R-CNN text detector (BoxProcessorUlimDit's model) behind the C ABI.
//
// Host-side counterpart of OptimizedDetectronPredictor.invoke_model (marie/detectron/detector.py:83-147) running
// detectron2's GeneralizedRCNN.inference with build_vit_fpn_backbone (marie/boxes/dit/ditod/backbone.py:131-153) and the
// configuration of config/zoo/unilm/dit/text_detection/{Base-RCNN-FPN,mask_rcnn_dit_base,mask_rcnn_dit_prod}.yaml:
//   ResizeShortestEdge(800, max 1333 | 4000) with PIL bilinear -> (x-127.5)/127.5, RGB, zero-pad to /32
//   -> DiT backbone + fpn1..4 (vit_api.hip) -> FPN (lateral 1x1, top-down nearest x2 + sum, output 3x3, p6 = p5[::2, ::2])
//   -> RPN (3x3 conv + ReLU, objectness + deltas; anchors 4..64 x ratios 1.5/3.5/6.5) -> 1000 proposals
//   -> ROIAlign 7x7 -> fc1/fc2 -> class + box regression -> softmax, NMS 0.5, rescale to the page.
// The mask head is not evaluated: the reference discards its output (ulim_dit_box_processor.py:441-455).
// detectron2 is third-party and absent from the reference tree; the stages follow its v0.6 semantics (see det_ops.hip).
#include <math.h>

#include "vit_internal.h"

struct mhip_dit {
  mhip_ctx* ctx = nullptr;
  int precision = MHIP_PREC_F16;
  mhip_dit_config cfg{};
  mhip_vit* vit = nullptr;
  TensorStore store;
  Arena arena;
  bool ready = false;
  size_t esz() const { return precision == MHIP_PREC_F16 ? 2 : 4; }
};

namespace {

constexpr int FPN_C = 256, FC_DIM = 1024, POOL = 7, MAX_ROIS = 1000;
const char* kVitPrefix = "backbone.bottom_up.backbone.";

struct DitGeom {
  int nh, nw, H32, W32;
  int lh[5], lw[5];   // p2..p6
};

void dit_geometry(const mhip_dit_config& c, int h, int w, DitGeom* g) {
  // detectron2 ResizeShortestEdge.get_output_shape
  const double size = c.min_size_test;
  double scale = size / (double)std::min(h, w);
  double newh, neww;
  if (h < w) { newh = size; neww = scale * w; } else { newh = scale * h; neww = size; }
  if (std::max(newh, neww) > c.max_size_test) {
    scale = (double)c.max_size_test / std::max(newh, neww);
    newh *= scale;
    neww *= scale;
  }
  g->nw = (int)(neww + 0.5);
  g->nh = (int)(newh + 0.5);
  g->H32 = (g->nh + 31) / 32 * 32;
  g->W32 = (g->nw + 31) / 32 * 32;
  for (int l = 0; l < 4; ++l) { g->lh[l] = g->H32 >> (2 + l); g->lw[l] = g->W32 >> (2 + l); }
  g->lh[4] = (g->lh[3] + 1) / 2;
  g->lw[4] = (g->lw[3] + 1) / 2;
}

int conv(mhip_ctx* ctx, int prec, const void* in, const void* w, const float* bias, void* out, int B, int H, int W, int Cin,
         int N, int k, int relu, int out_f32 = 0, int ldc = 0) {
  ConvDesc c;
  c.in = in; c.w = w; c.bias = bias; c.out = out;
  c.B = B; c.H = H; c.W = W; c.Cin = Cin; c.KH = c.KW = k; c.pad = k / 2; c.N = N;
  c.relu = relu; c.out_f32 = out_f32; c.ldc = ldc; c.pad_cols_writable = ldc ? 1 : 0;     // pitched outputs here are private [rows][16] / [rows][8] buffers
  return mhip_launch_conv_igemm(ctx, prec, c);
}

}  // namespace

extern "C" int mhip_dit_default_config(int model, mhip_dit_config* c) {
  if (!c || (model != 0 && model != 1)) return MHIP_EINVAL;
  c->model = model;
  c->min_size_test = 800;
  c->max_size_test = model ? 4000 : 1333;            // mask_rcnn_dit_prod.yaml:36-38; detectron2 default
  c->detections_per_image = model ? 2500 : 2000;     // mask_rcnn_dit_prod.yaml:32; mask_rcnn_dit_base.yaml:19
  const float sizes[5] = {4, 8, 16, 32, 64}, ratios[3] = {1.5f, 3.5f, 6.5f};
  for (int i = 0; i < 5; ++i) c->anchor_sizes[i] = sizes[i];
  for (int i = 0; i < 3; ++i) c->aspect_ratios[i] = ratios[i];
  c->rpn_nms_thresh = 0.7f;
  c->score_thresh = 0.05f;
  c->nms_thresh = 0.5f;
  return MHIP_OK;
}

extern "C" int mhip_dit_resized_shape(const mhip_dit_config* c, int h, int w, int* nh, int* nw, int* H32, int* W32) {
  if (!c || h < 1 || w < 1) return MHIP_EINVAL;
  DitGeom g;
  dit_geometry(*c, h, w, &g);
  if (nh) *nh = g.nh;
  if (nw) *nw = g.nw;
  if (H32) *H32 = g.H32;
  if (W32) *W32 = g.W32;
  return MHIP_OK;
}

extern "C" int mhip_dit_create(mhip_ctx* ctx, int precision, const mhip_dit_config* cfg, mhip_dit** out) {
  if (!ctx || !cfg || !out) return MHIP_EINVAL;
  *out = nullptr;
  if (precision != MHIP_PREC_F16 && precision != MHIP_PREC_F32) return mhip_fail(ctx, MHIP_EINVAL, "unknown precision %d", precision);
  if (cfg->min_size_test < 32 || cfg->max_size_test < cfg->min_size_test) return mhip_fail(ctx, MHIP_EINVAL, "dit: bad test sizes");
  mhip_vit_config vc{};
  vc.dim = cfg->model ? 1024 : 768;
  vc.depth = cfg->model ? 24 : 12;
  vc.heads = cfg->model ? 16 : 12;
  vc.patch = 16; vc.pos_h = vc.pos_w = 14;
  vc.layer_scale = 1; vc.qkv_bias = 1; vc.final_norm = 0; vc.fpn = 1;
  const int taps_b[4] = {3, 5, 7, 11}, taps_l[4] = {7, 11, 15, 23};
  for (int j = 0; j < 4; ++j) vc.taps[j] = cfg->model ? taps_l[j] : taps_b[j];
  vc.ln_eps = 1e-6f;
  mhip_vit* vit = nullptr;
  int rc = mhip_vit_create(ctx, precision, &vc, &vit);
  if (rc) return rc;
  mhip_dit* m = new mhip_dit();
  m->ctx = ctx; m->precision = precision; m->cfg = *cfg; m->vit = vit;
  const size_t es = m->esz(), D = vc.dim;
  Arena& a = m->arena;
  for (int s = 2; s <= 5; ++s) {
    a.take("lat" + std::to_string(s) + "_w", FPN_C * D * es); a.take("lat" + std::to_string(s) + "_b", FPN_C * 4);
    a.take("out" + std::to_string(s) + "_w", (size_t)FPN_C * 9 * FPN_C * es); a.take("out" + std::to_string(s) + "_b", FPN_C * 4);
  }
  a.take("rpn_conv_w", (size_t)FPN_C * 9 * FPN_C * es); a.take("rpn_conv_b", FPN_C * 4);
  a.take("rpn_head_w", (size_t)16 * FPN_C * es); a.take("rpn_head_b", 64 * 4);
  a.take("fc1_w", (size_t)FC_DIM * POOL * POOL * FPN_C * es); a.take("fc1_b", FC_DIM * 4);
  a.take("fc2_w", (size_t)FC_DIM * FC_DIM * es); a.take("fc2_b", FC_DIM * 4);
  a.take("pred_w", (size_t)64 * FC_DIM * es); a.take("pred_b", 64 * 4);
  *out = m;
  return MHIP_OK;
}

extern "C" int mhip_dit_destroy(mhip_dit* m) {
  if (!m) return MHIP_OK;
  mhip_quiesce(m->ctx);
  mhip_vit_destroy(m->vit);
  m->arena.release();
  delete m;
  return MHIP_OK;
}

extern "C" int mhip_dit_set_tensor(mhip_dit* m, const char* key, const float* data, const int64_t* shape, int ndim) {
  if (!m || !key) return MHIP_EINVAL;
  std::string k(key);
  if (k.rfind("module.", 0) == 0) k = k.substr(7);
  if (k.rfind(kVitPrefix, 0) == 0) return mhip_vit_set_tensor(m->vit, k.c_str() + strlen(kVitPrefix), data, shape, ndim);
  // present in detectron2 checkpoints, not used on this path
  if (k.rfind("roi_heads.mask_head.", 0) == 0 || k == "pixel_mean" || k == "pixel_std" ||
      k.rfind("proposal_generator.anchor_generator.", 0) == 0)
    return MHIP_OK;
  const bool known = k.rfind("backbone.fpn_", 0) == 0 || k.rfind("proposal_generator.rpn_head.", 0) == 0 ||
                     k.rfind("roi_heads.box_head.", 0) == 0 || k.rfind("roi_heads.box_predictor.", 0) == 0;
  if (!known) return mhip_fail(m->ctx, MHIP_EINVAL, "unknown state_dict key %s", key);
  m->ready = false;
  return m->store.set(m->ctx, k, data, shape, ndim);
}

extern "C" int mhip_dit_alloc_arena(mhip_dit* m) {
  if (!m) return MHIP_EINVAL;
  int rc = m->arena.alloc(m->ctx);
  if (rc) return rc;
  if ((rc = mhip_vit_alloc_arena(m->vit))) return rc;
  m->ready = true;
  return MHIP_OK;
}

// two arenas (backbone, heads): index 0 / 1
extern "C" int mhip_dit_arena(mhip_dit* m, int which, void** dev, size_t* bytes) {
  if (!m || which < 0 || which > 1) return MHIP_EINVAL;
  if (which == 0) return mhip_vit_arena(m->vit, dev, bytes);
  if (dev) *dev = m->arena.dev;
  if (bytes) *bytes = m->arena.bytes;
  return MHIP_OK;
}

// [Co][Ci][k][k] -> [Co][(dy*k+dx)*Ci + ci]
static void pack_conv(int prec, char* dst, const HostTensor& w, int co, int ci, int k) {
  std::vector<float> tmp((size_t)co * k * k * ci);
  for (int o = 0; o < co; ++o)
    for (int c = 0; c < ci; ++c)
      for (int t = 0; t < k * k; ++t) tmp[((size_t)o * k * k + t) * ci + c] = w.data[((size_t)o * ci + c) * k * k + t];
  Arena::put(prec, dst, tmp.data(), tmp.size());
}

extern "C" int mhip_dit_finalize(mhip_dit* m) {
  if (!m) return MHIP_EINVAL;
  mhip_ctx* ctx = m->ctx;
  int rc = mhip_vit_finalize(m->vit);
  if (rc) return rc;
  const int prec = m->precision, D = m->vit->cfg.dim;
  const size_t es = m->esz();
  Arena& a = m->arena;
  const TensorStore& st = m->store;
  a.begin_fill();
  for (int s = 2; s <= 5; ++s) {
    const std::string n = std::to_string(s);
    const HostTensor* lw = st.find(ctx, "backbone.fpn_lateral" + n + ".weight", {FPN_C, D, 1, 1});
    const HostTensor* lb = st.find(ctx, "backbone.fpn_lateral" + n + ".bias", {FPN_C});
    const HostTensor* ow = st.find(ctx, "backbone.fpn_output" + n + ".weight", {FPN_C, FPN_C, 3, 3});
    const HostTensor* ob = st.find(ctx, "backbone.fpn_output" + n + ".bias", {FPN_C});
    if (!lw || !lb || !ow || !ob) return MHIP_ESTATE;
    Arena::put(prec, a.h("lat" + n + "_w"), lw->data.data(), lw->numel());
    memcpy(a.h("lat" + n + "_b"), lb->data.data(), FPN_C * 4);
    pack_conv(prec, a.h("out" + n + "_w"), *ow, FPN_C, FPN_C, 3);
    memcpy(a.h("out" + n + "_b"), ob->data.data(), FPN_C * 4);
  }
  {
    const std::string r = "proposal_generator.rpn_head.";
    const HostTensor* cw = st.find(ctx, r + "conv.weight", {FPN_C, FPN_C, 3, 3});
    const HostTensor* cb = st.find(ctx, r + "conv.bias", {FPN_C});
    const HostTensor* ow = st.find(ctx, r + "objectness_logits.weight", {3, FPN_C, 1, 1});
    const HostTensor* ob = st.find(ctx, r + "objectness_logits.bias", {3});
    const HostTensor* dw = st.find(ctx, r + "anchor_deltas.weight", {12, FPN_C, 1, 1});
    const HostTensor* db = st.find(ctx, r + "anchor_deltas.bias", {12});
    if (!cw || !cb || !ow || !ob || !dw || !db) return MHIP_ESTATE;
    pack_conv(prec, a.h("rpn_conv_w"), *cw, FPN_C, FPN_C, 3);
    memcpy(a.h("rpn_conv_b"), cb->data.data(), FPN_C * 4);
    Arena::put(prec, a.h("rpn_head_w"), ow->data.data(), (size_t)3 * FPN_C);
    Arena::put(prec, a.h("rpn_head_w") + (size_t)3 * FPN_C * es, dw->data.data(), (size_t)12 * FPN_C);
    float* hb = (float*)a.h("rpn_head_b");
    for (int i = 0; i < 3; ++i) hb[i] = ob->data[i];
    for (int i = 0; i < 12; ++i) hb[3 + i] = db->data[i];
  }
  {
    const HostTensor* w1 = st.find(ctx, "roi_heads.box_head.fc1.weight", {FC_DIM, FPN_C * POOL * POOL});
    const HostTensor* b1 = st.find(ctx, "roi_heads.box_head.fc1.bias", {FC_DIM});
    const HostTensor* w2 = st.find(ctx, "roi_heads.box_head.fc2.weight", {FC_DIM, FC_DIM});
    const HostTensor* b2 = st.find(ctx, "roi_heads.box_head.fc2.bias", {FC_DIM});
    const HostTensor* cw = st.find(ctx, "roi_heads.box_predictor.cls_score.weight", {2, FC_DIM});
    const HostTensor* cb = st.find(ctx, "roi_heads.box_predictor.cls_score.bias", {2});
    const HostTensor* bw = st.find(ctx, "roi_heads.box_predictor.bbox_pred.weight", {4, FC_DIM});
    const HostTensor* bb = st.find(ctx, "roi_heads.box_predictor.bbox_pred.bias", {4});
    if (!w1 || !b1 || !w2 || !b2 || !cw || !cb || !bw || !bb) return MHIP_ESTATE;
    // torch flattens the pooled (C, 7, 7) as k = c*49 + bin; ROIAlign here writes k' = bin*C + c
    std::vector<float> tmp(w1->numel());
    const int K = FPN_C * POOL * POOL;
    for (int o = 0; o < FC_DIM; ++o)
      for (int c = 0; c < FPN_C; ++c)
        for (int b = 0; b < POOL * POOL; ++b) tmp[(size_t)o * K + (size_t)b * FPN_C + c] = w1->data[(size_t)o * K + (size_t)c * POOL * POOL + b];
    Arena::put(prec, a.h("fc1_w"), tmp.data(), tmp.size());
    memcpy(a.h("fc1_b"), b1->data.data(), FC_DIM * 4);
    Arena::put(prec, a.h("fc2_w"), w2->data.data(), w2->numel());
    memcpy(a.h("fc2_b"), b2->data.data(), FC_DIM * 4);
    Arena::put(prec, a.h("pred_w"), cw->data.data(), (size_t)2 * FC_DIM);
    Arena::put(prec, a.h("pred_w") + (size_t)2 * FC_DIM * es, bw->data.data(), (size_t)4 * FC_DIM);
    float* pb = (float*)a.h("pred_b");
    pb[0] = cb->data[0]; pb[1] = cb->data[1];
    for (int i = 0; i < 4; ++i) pb[2 + i] = bb->data[i];
  }
  if ((rc = a.upload(ctx))) return rc;
  m->ready = true;
  m->store.t.clear();
  return MHIP_OK;
}

static size_t dit_ws_bytes(const mhip_dit* m, int B, int h, int w, const DitGeom& g) {
  const size_t es = m->esz();
  VitGeom vg;
  vit_geometry(m->vit, g.H32, g.W32, &vg);
  size_t b = (size_t)B * g.nh * g.nw * 3 + 256 + mhip_pil_resize_scratch_bytes(h, w, g.nh, g.nw, MHIP_PIL_BILINEAR) + 256;
  b += vit_workspace_bytes(m->vit, B, vg) + vit_fpn_workspace_bytes(m->vit, B, vg);
  size_t px = 0;
  for (int l = 0; l < 5; ++l) px += (size_t)g.lh[l] * g.lw[l];
  b += 4 * ((size_t)B * px * FPN_C * es + 5 * 256);            // lateral, merged, p-levels, rpn conv
  b += (size_t)B * px * 16 * 4 + 5 * 256;                      // rpn head
  b += (size_t)B * (5 * MAX_ROIS * 5 * 4 + MAX_ROIS * 5 * 4 * 2 + 64) + 4096;
  b += (size_t)B * MAX_ROIS * (POOL * POOL * FPN_C + 2 * FC_DIM) * es + (size_t)B * MAX_ROIS * 8 * 4 + 4096;
  return b + (1 << 16);
}

extern "C" size_t mhip_dit_workspace_bytes(mhip_dit* m, int B, int h, int w) {
  if (!m || B < 1 || h < 1 || w < 1) return 0;
  DitGeom g;
  dit_geometry(m->cfg, h, w, &g);
  return dit_ws_bytes(m, B, h, w, g);
}

struct DitDebug {
  float* fpn[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};   // host fp32 NHWC p2..p6 (B = 1)
  float* prop_boxes = nullptr;   // host [1000][4]
  float* prop_scores = nullptr;  // host [1000]
  int* prop_count = nullptr;
  float* rpn_head[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};   // host fp32 [H*W][16] per level (3 logits, 12 deltas, pad)
  float* box_head = nullptr;     // host fp32 [1000][8] (2 class scores, 4 deltas, 2 pad), rows past prop_count undefined
};

static int dit_run(mhip_dit* m, const uint8_t* const* pages_dev, int B, int h, int w, float* boxes_host, float* scores_host,
                   int* counts_host, const DitDebug* dbg) {
  mhip_ctx* ctx = m->ctx;
  if (!m->ready) return mhip_fail(ctx, MHIP_ESTATE, "dit: weights not finalized");
  if (B < 1 || h < 1 || w < 1) return mhip_fail(ctx, MHIP_EINVAL, "dit: bad page geometry");
  MHIP_HIP(ctx, hipSetDevice(ctx->device));
  const int prec = m->precision;
  const size_t es = m->esz();
  DitGeom g;
  dit_geometry(m->cfg, h, w, &g);
  int rc = mhip_ensure_workspace(ctx, dit_ws_bytes(m, B, h, w, g) + (dbg ? (size_t)g.lh[0] * g.lw[0] * FPN_C * 4 + 256 : 0));
  if (rc) return rc;
  Carver ws(ctx->ws);
  const Arena& a = m->arena;
  // 1. resize (PIL bilinear, channel order irrelevant) into B x [nh][nw][3]
  uint8_t* imgs = ws.take<uint8_t>((size_t)B * g.nh * g.nw * 3);
  void* rscratch = ws.take(mhip_pil_resize_scratch_bytes(h, w, g.nh, g.nw, MHIP_PIL_BILINEAR));
  for (int b = 0; b < B; ++b)
    if ((rc = mhip_launch_pil_resize_rgb(ctx, pages_dev[b], h, w, (size_t)w * 3, imgs + (size_t)b * g.nh * g.nw * 3, g.nh, g.nw, MHIP_PIL_BILINEAR, rscratch))) return rc;
  // 2. backbone (pages arrive BGR, INPUT.FORMAT is RGB)
  VitRun run;
  if ((rc = vit_encode(m->vit, ws, imgs, B, g.nh, g.nw, g.H32, g.W32, 1, &run))) return rc;
  VitFpnOut fo;
  if ((rc = vit_fpn(m->vit, ws, B, run, &fo))) return rc;
  const int D = m->vit->cfg.dim;
  // 3. FPN: laterals (per-pixel, any row order), top-down merge restores raster order, 3x3 output convs
  char* lat[4];
  char* merged[4];
  char* pl[5];
  for (int l = 0; l < 4; ++l) {
    const size_t n = (size_t)B * g.lh[l] * g.lw[l] * FPN_C * es;
    lat[l] = ws.take(n); merged[l] = ws.take(n); pl[l] = ws.take(n);
  }
  pl[4] = ws.take((size_t)B * g.lh[4] * g.lw[4] * FPN_C * es);
  for (int l = 3; l >= 0; --l) {
    const std::string s = std::to_string(l + 2);
    const long long rows = (long long)B * g.lh[l] * g.lw[l];
    if ((rc = mhip_gemm(ctx, prec, fo.level[l], a.d("lat" + s + "_w"), rows, FPN_C, D, nullptr, a.d<float>("lat" + s + "_b"), lat[l], ACT_NONE, 0))) return rc;
    const char* prev = lat[l];
    if (l < 3 || fo.nest[l]) {
      if ((rc = mhip_launch_unnest(ctx, prec, lat[l], l < 3 ? merged[l + 1] : nullptr, merged[l], 0, B, g.lh[l], g.lw[l], FPN_C, fo.nest[l]))) return rc;
      prev = merged[l];
    } else {
      merged[l] = lat[l];
    }
    if ((rc = conv(ctx, prec, prev, a.d("out" + s + "_w"), a.d<float>("out" + s + "_b"), pl[l], B, g.lh[l], g.lw[l], FPN_C, FPN_C, 3, ACT_NONE))) return rc;
  }
  if ((rc = mhip_launch_subsample2(ctx, prec, pl[3], pl[4], B, g.lh[3], g.lw[3], FPN_C))) return rc;
  // 4. RPN head on p2..p6
  RpnDesc rd;
  char* rt = ws.take((size_t)B * g.lh[0] * g.lw[0] * FPN_C * es);
  for (int l = 0; l < 5; ++l) {
    float* ho = ws.take<float>((size_t)B * g.lh[l] * g.lw[l] * 16 * 4);
    if ((rc = conv(ctx, prec, pl[l], a.d("rpn_conv_w"), a.d<float>("rpn_conv_b"), rt, B, g.lh[l], g.lw[l], FPN_C, FPN_C, 3, ACT_RELU))) return rc;
    if ((rc = conv(ctx, prec, rt, a.d("rpn_head_w"), a.d<float>("rpn_head_b"), ho, B, g.lh[l], g.lw[l], FPN_C, 15, 1, ACT_NONE, 1, 16))) return rc;
    rd.head[l] = ho; rd.H[l] = g.lh[l]; rd.W[l] = g.lw[l]; rd.stride[l] = 4 << l;
  }
  mhip_rpn_cell_anchors(m->cfg.anchor_sizes, m->cfg.aspect_ratios, rd.cell);
  rd.images = B; rd.img_h = g.nh; rd.img_w = g.nw; rd.nms_thr = m->cfg.rpn_nms_thresh; rd.post_topk = MAX_ROIS;
  rd.lvl_boxes = ws.take<float>((size_t)B * 5 * MAX_ROIS * 4 * 4);
  rd.lvl_scores = ws.take<float>((size_t)B * 5 * MAX_ROIS * 4);
  rd.lvl_counts = ws.take<int>((size_t)B * 5 * 4);
  rd.out_boxes = ws.take<float>((size_t)B * MAX_ROIS * 4 * 4);
  rd.out_scores = ws.take<float>((size_t)B * MAX_ROIS * 4);
  rd.out_counts = ws.take<int>((size_t)B * 4);
  if ((rc = mhip_launch_rpn_proposals(ctx, rd))) return rc;
  // 5. box head
  RoiDesc ro;
  for (int l = 0; l < 4; ++l) { ro.feat[l] = pl[l]; ro.H[l] = g.lh[l]; ro.W[l] = g.lw[l]; ro.scale[l] = 1.f / (float)(4 << l); }
  ro.rois = rd.out_boxes; ro.counts = rd.out_counts; ro.images = B; ro.max_rois = MAX_ROIS; ro.C = FPN_C;
  const int K1 = POOL * POOL * FPN_C;
  char* pooled = ws.take((size_t)B * MAX_ROIS * K1 * es);
  MHIP_HIP(ctx, hipMemsetAsync(pooled, 0, (size_t)B * MAX_ROIS * K1 * es, ctx->stream));   // rows past the proposal count
  ro.out = pooled;
  if ((rc = mhip_launch_roi_align(ctx, prec, ro))) return rc;
  char* f1 = ws.take((size_t)B * MAX_ROIS * FC_DIM * es);
  char* f2 = ws.take((size_t)B * MAX_ROIS * FC_DIM * es);
  float* hd = ws.take<float>((size_t)B * MAX_ROIS * 8 * 4);
  const long long R = (long long)B * MAX_ROIS;
  if ((rc = mhip_gemm(ctx, prec, pooled, a.d("fc1_w"), R, FC_DIM, K1, nullptr, a.d<float>("fc1_b"), f1, ACT_RELU, 0))) return rc;
  if ((rc = mhip_gemm(ctx, prec, f1, a.d("fc2_w"), R, FC_DIM, FC_DIM, nullptr, a.d<float>("fc2_b"), f2, ACT_RELU, 0))) return rc;
  if ((rc = mhip_gemm(ctx, prec, f2, a.d("pred_w"), R, 6, FC_DIM, nullptr, a.d<float>("pred_b"), hd, ACT_NONE, 1, nullptr, 8, 1))) return rc;
  DetFinalDesc fd;
  fd.head = hd; fd.rois = rd.out_boxes; fd.counts = rd.out_counts; fd.images = B; fd.max_rois = MAX_ROIS;
  fd.img_h = g.nh; fd.img_w = g.nw; fd.out_h = h; fd.out_w = w;
  fd.score_thr = m->cfg.score_thresh; fd.nms_thr = m->cfg.nms_thresh; fd.max_det = m->cfg.detections_per_image;
  fd.out_boxes = ws.take<float>((size_t)B * MAX_ROIS * 4 * 4);
  fd.out_scores = ws.take<float>((size_t)B * MAX_ROIS * 4);
  fd.out_count = ws.take<int>((size_t)B * 4);
  if ((rc = mhip_launch_det_final(ctx, fd))) return rc;
  MHIP_HIP(ctx, hipMemcpyAsync(counts_host, fd.out_count, (size_t)B * 4, hipMemcpyDeviceToHost, ctx->stream));
  MHIP_HIP(ctx, hipMemcpyAsync(boxes_host, fd.out_boxes, (size_t)B * MAX_ROIS * 16, hipMemcpyDeviceToHost, ctx->stream));
  if (scores_host) MHIP_HIP(ctx, hipMemcpyAsync(scores_host, fd.out_scores, (size_t)B * MAX_ROIS * 4, hipMemcpyDeviceToHost, ctx->stream));
  if (dbg) {
    float* stage = ws.take<float>((size_t)g.lh[0] * g.lw[0] * FPN_C * 4);
    for (int l = 0; l < 5; ++l)
      if (dbg->fpn[l]) {
        const int rows = g.lh[l] * g.lw[l];
        if ((rc = mhip_launch_convert_rows(ctx, prec, pl[l], stage, rows, FPN_C))) return rc;
        MHIP_HIP(ctx, hipMemcpyAsync(dbg->fpn[l], stage, (size_t)rows * FPN_C * 4, hipMemcpyDeviceToHost, ctx->stream));
        MHIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
      }
    for (int l = 0; l < 5; ++l)
      if (dbg->rpn_head[l])
        MHIP_HIP(ctx, hipMemcpyAsync(dbg->rpn_head[l], rd.head[l], (size_t)g.lh[l] * g.lw[l] * 64, hipMemcpyDeviceToHost, ctx->stream));
    if (dbg->box_head) MHIP_HIP(ctx, hipMemcpyAsync(dbg->box_head, hd, (size_t)MAX_ROIS * 32, hipMemcpyDeviceToHost, ctx->stream));
    if (dbg->prop_boxes) MHIP_HIP(ctx, hipMemcpyAsync(dbg->prop_boxes, rd.out_boxes, MAX_ROIS * 16, hipMemcpyDeviceToHost, ctx->stream));
    if (dbg->prop_scores) MHIP_HIP(ctx, hipMemcpyAsync(dbg->prop_scores, rd.out_scores, MAX_ROIS * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (dbg->prop_count) MHIP_HIP(ctx, hipMemcpyAsync(dbg->prop_count, rd.out_counts, 4, hipMemcpyDeviceToHost, ctx->stream));
  }
  MHIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return MHIP_OK;
}

extern "C" int mhip_dit_detect(mhip_dit* m, const uint8_t* const* pages_dev, int B, int h, int w, float* boxes_host,
                               float* scores_host, int* counts_host) {
  if (!m || !pages_dev || !boxes_host || !counts_host) return MHIP_EINVAL;
  return dit_run(m, pages_dev, B, h, w, boxes_host, scores_host, counts_host, nullptr);
}

extern "C" int mhip_dit_detect_host(mhip_dit* m, const uint8_t* pages_host, int B, int h, int w, float* boxes_host,
                                    float* scores_host, int* counts_host) {
  if (!m || !pages_host || !boxes_host || !counts_host) return MHIP_EINVAL;
  mhip_ctx* ctx = m->ctx;
  MHIP_HIP(ctx, hipSetDevice(ctx->device));
  uint8_t* dev = nullptr;
  const size_t pb = (size_t)h * w * 3;
  MHIP_HIP(ctx, hipMalloc((void**)&dev, pb * B));
  hipError_t e = hipMemcpyAsync(dev, pages_host, pb * B, hipMemcpyHostToDevice, ctx->stream);
  int rc = MHIP_OK;
  if (e != hipSuccess) rc = mhip_fail(ctx, MHIP_EHIP, "page upload: %s", hipGetErrorString(e));
  if (!rc) {
    std::vector<const uint8_t*> ptrs(B);
    for (int b = 0; b < B; ++b) ptrs[b] = dev + pb * b;
    rc = dit_run(m, ptrs.data(), B, h, w, boxes_host, scores_host, counts_host, nullptr);
  }
  (void)hipStreamSynchronize(ctx->stream);
  (void)hipFree(dev);
  return rc;
}

// one page, with the intermediate maps the parity tests look at
extern "C" int mhip_dit_debug_host(mhip_dit* m, const uint8_t* page_host, int h, int w, float* boxes_host, float* scores_host,
                                   int* count_host, float* p2, float* p3, float* p4, float* p5, float* p6,
                                   float* prop_boxes, float* prop_scores, int* prop_count) {
  if (!m || !page_host || !boxes_host || !count_host) return MHIP_EINVAL;
  mhip_ctx* ctx = m->ctx;
  MHIP_HIP(ctx, hipSetDevice(ctx->device));
  uint8_t* dev = nullptr;
  const size_t pb = (size_t)h * w * 3;
  MHIP_HIP(ctx, hipMalloc((void**)&dev, pb));
  hipError_t e = hipMemcpy(dev, page_host, pb, hipMemcpyHostToDevice);
  int rc = e == hipSuccess ? MHIP_OK : mhip_fail(ctx, MHIP_EHIP, "page upload: %s", hipGetErrorString(e));
  if (!rc) {
    DitDebug d;
    d.fpn[0] = p2; d.fpn[1] = p3; d.fpn[2] = p4; d.fpn[3] = p5; d.fpn[4] = p6;
    d.prop_boxes = prop_boxes; d.prop_scores = prop_scores; d.prop_count = prop_count;
    const uint8_t* ptr = dev;
    rc = dit_run(m, &ptr, 1, h, w, boxes_host, scores_host, count_host, &d);
  }
  (void)hipStreamSynchronize(ctx->stream);
  (void)hipFree(dev);
  return rc;
}

// same, plus the inputs of the two discrete stages (RPN selection, FastRCNN inference) as this run computed them: the
// parity tests replay the CPU restatement of the discrete stages on them
extern "C" int mhip_dit_debug_taps_host(mhip_dit* m, const uint8_t* page_host, int h, int w, float* boxes_host,
                                        float* scores_host, int* count_host, float* const* fpn5, float* const* rpn_head5,
                                        float* prop_boxes, float* prop_scores, int* prop_count, float* box_head) {
  if (!m || !page_host || !boxes_host || !count_host) return MHIP_EINVAL;
  mhip_ctx* ctx = m->ctx;
  MHIP_HIP(ctx, hipSetDevice(ctx->device));
  uint8_t* dev = nullptr;
  const size_t pb = (size_t)h * w * 3;
  MHIP_HIP(ctx, hipMalloc((void**)&dev, pb));
  hipError_t e = hipMemcpy(dev, page_host, pb, hipMemcpyHostToDevice);
  int rc = e == hipSuccess ? MHIP_OK : mhip_fail(ctx, MHIP_EHIP, "page upload: %s", hipGetErrorString(e));
  if (!rc) {
    DitDebug d;
    for (int l = 0; l < 5; ++l) {
      if (fpn5) d.fpn[l] = fpn5[l];
      if (rpn_head5) d.rpn_head[l] = rpn_head5[l];
    }
    d.prop_boxes = prop_boxes; d.prop_scores = prop_scores; d.prop_count = prop_count; d.box_head = box_head;
    const uint8_t* ptr = dev;
    rc = dit_run(m, &ptr, 1, h, w, boxes_host, scores_host, count_host, &d);
  }
  (void)hipStreamSynchronize(ctx->stream);
  (void)hipFree(dev);
  return rc;
}

// replaces: blackout_bboxes, marie/boxes/dit/ulim_dit_box_processor.py:161-198 — in place on a device page (BGR);
// boxes int32 xyxy on the host; *changed = 1 if any pixel changed (the reference's np.array_equal early stop).
extern "C" int mhip_blackout_bboxes(mhip_ctx* ctx, uint8_t* page_dev, int h, int w, const int32_t* boxes_xyxy_host, int n,
                                    int* changed) {
  if (!ctx || !page_dev || (n && !boxes_xyxy_host) || !changed) return MHIP_EINVAL;
  *changed = 0;
  if (n <= 0) return MHIP_OK;
  MHIP_HIP(ctx, hipSetDevice(ctx->device));
  int rc = mhip_ensure_workspace(ctx, (size_t)n * 16 + 256);
  if (rc) return rc;
  int* flag = (int*)ctx->ws;
  int* boxes = flag + 64;
  MHIP_HIP(ctx, hipMemsetAsync(flag, 0, 4, ctx->stream));
  MHIP_HIP(ctx, hipMemcpyAsync(boxes, boxes_xyxy_host, (size_t)n * 16, hipMemcpyHostToDevice, ctx->stream));
  if ((rc = mhip_launch_blackout(ctx, page_dev, h, w, boxes, n, flag))) return rc;
  MHIP_HIP(ctx, hipMemcpyAsync(changed, flag, 4, hipMemcpyDeviceToHost, ctx->stream));
  MHIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return MHIP_OK;
}

// ---------------------------------------------------------------------------------------------------- stage entries
// The detectron2 stages on caller-supplied inputs (host buffers): what the parity tests drive with identical inputs on
// both sides, and a standalone use of each stage.
extern "C" int mhip_rpn_proposals_host(mhip_ctx* ctx, const float* const* heads_host, const int* H, const int* W,
                                       const int* strides, const float* anchor_sizes, const float* aspect_ratios, int img_h,
                                       int img_w, float nms_thresh, float* boxes_out, float* scores_out, int* count_out) {
  if (!ctx || !heads_host || !H || !W || !strides || !anchor_sizes || !aspect_ratios || !boxes_out || !scores_out || !count_out)
    return MHIP_EINVAL;
  MHIP_HIP(ctx, hipSetDevice(ctx->device));
  size_t need = 1 << 16;
  for (int l = 0; l < 5; ++l) need += (size_t)H[l] * W[l] * 64 + 256;
  need += (size_t)5 * MAX_ROIS * 20 + MAX_ROIS * 20 + 4096;
  int rc = mhip_ensure_workspace(ctx, need);
  if (rc) return rc;
  Carver ws(ctx->ws);
  RpnDesc rd;
  for (int l = 0; l < 5; ++l) {
    float* d = ws.take<float>((size_t)H[l] * W[l] * 64);
    MHIP_HIP(ctx, hipMemcpyAsync(d, heads_host[l], (size_t)H[l] * W[l] * 64, hipMemcpyHostToDevice, ctx->stream));
    rd.head[l] = d; rd.H[l] = H[l]; rd.W[l] = W[l]; rd.stride[l] = strides[l];
  }
  mhip_rpn_cell_anchors(anchor_sizes, aspect_ratios, rd.cell);
  rd.images = 1; rd.img_h = img_h; rd.img_w = img_w; rd.nms_thr = nms_thresh; rd.post_topk = MAX_ROIS;
  rd.lvl_boxes = ws.take<float>((size_t)5 * MAX_ROIS * 16);
  rd.lvl_scores = ws.take<float>((size_t)5 * MAX_ROIS * 4);
  rd.lvl_counts = ws.take<int>(5 * 4);
  rd.out_boxes = ws.take<float>((size_t)MAX_ROIS * 16);
  rd.out_scores = ws.take<float>((size_t)MAX_ROIS * 4);
  rd.out_counts = ws.take<int>(4);
  if ((rc = mhip_launch_rpn_proposals(ctx, rd))) return rc;
  MHIP_HIP(ctx, hipMemcpyAsync(boxes_out, rd.out_boxes, MAX_ROIS * 16, hipMemcpyDeviceToHost, ctx->stream));
  MHIP_HIP(ctx, hipMemcpyAsync(scores_out, rd.out_scores, MAX_ROIS * 4, hipMemcpyDeviceToHost, ctx->stream));
  MHIP_HIP(ctx, hipMemcpyAsync(count_out, rd.out_counts, 4, hipMemcpyDeviceToHost, ctx->stream));
  MHIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return MHIP_OK;
}

// feats_host[l]: fp32 NHWC [H[l]][W[l]][C] at strides 4, 8, 16, 32; rois [n][4] -> pooled fp32 [n][49*C] (k = bin*C + c)
extern "C" int mhip_roi_align_host(mhip_ctx* ctx, const float* const* feats_host, const int* H, const int* W, int C,
                                   const float* rois_host, int n, float* pooled_out) {
  if (!ctx || !feats_host || !H || !W || !rois_host || !pooled_out || n < 0 || n > MAX_ROIS || C < 1) return MHIP_EINVAL;
  MHIP_HIP(ctx, hipSetDevice(ctx->device));
  size_t need = (size_t)n * 49 * C * 4 + MAX_ROIS * 16 + 8192;
  for (int l = 0; l < 4; ++l) need += (size_t)H[l] * W[l] * C * 4 + 256;
  int rc = mhip_ensure_workspace(ctx, need);
  if (rc) return rc;
  Carver ws(ctx->ws);
  RoiDesc ro;
  for (int l = 0; l < 4; ++l) {
    float* d = ws.take<float>((size_t)H[l] * W[l] * C * 4);
    MHIP_HIP(ctx, hipMemcpyAsync(d, feats_host[l], (size_t)H[l] * W[l] * C * 4, hipMemcpyHostToDevice, ctx->stream));
    ro.feat[l] = d; ro.H[l] = H[l]; ro.W[l] = W[l]; ro.scale[l] = 1.f / (float)(4 << l);
  }
  float* rois = ws.take<float>(MAX_ROIS * 16);
  int* cnt = ws.take<int>(4);
  float* out = ws.take<float>((size_t)std::max(n, 1) * 49 * C * 4);
  MHIP_HIP(ctx, hipMemcpyAsync(rois, rois_host, (size_t)n * 16, hipMemcpyHostToDevice, ctx->stream));
  MHIP_HIP(ctx, hipMemcpyAsync(cnt, &n, 4, hipMemcpyHostToDevice, ctx->stream));
  ro.rois = rois; ro.counts = cnt; ro.images = 1; ro.max_rois = std::max(n, 1); ro.C = C; ro.out = out;
  if ((rc = mhip_launch_roi_align(ctx, MHIP_PREC_F32, ro))) return rc;
  MHIP_HIP(ctx, hipMemcpyAsync(pooled_out, out, (size_t)n * 49 * C * 4, hipMemcpyDeviceToHost, ctx->stream));
  MHIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return MHIP_OK;
}

// head_host [n][8] (2 class scores, 4 deltas, 2 pad), rois_host [n][4] -> detections in page coordinates
extern "C" int mhip_det_final_host(mhip_ctx* ctx, const float* head_host, const float* rois_host, int n, int img_h, int img_w,
                                   int page_h, int page_w, float score_thresh, float nms_thresh, int max_det,
                                   float* boxes_out, float* scores_out, int* count_out) {
  if (!ctx || !head_host || !rois_host || !boxes_out || !scores_out || !count_out || n < 0 || n > MAX_ROIS) return MHIP_EINVAL;
  MHIP_HIP(ctx, hipSetDevice(ctx->device));
  int rc = mhip_ensure_workspace(ctx, (size_t)MAX_ROIS * 80 + 8192);
  if (rc) return rc;
  Carver ws(ctx->ws);
  float* head = ws.take<float>(MAX_ROIS * 32);
  float* rois = ws.take<float>(MAX_ROIS * 16);
  int* cnt = ws.take<int>(4);
  DetFinalDesc fd;
  fd.out_boxes = ws.take<float>(MAX_ROIS * 16);
  fd.out_scores = ws.take<float>(MAX_ROIS * 4);
  fd.out_count = ws.take<int>(4);
  MHIP_HIP(ctx, hipMemcpyAsync(head, head_host, (size_t)n * 32, hipMemcpyHostToDevice, ctx->stream));
  MHIP_HIP(ctx, hipMemcpyAsync(rois, rois_host, (size_t)n * 16, hipMemcpyHostToDevice, ctx->stream));
  MHIP_HIP(ctx, hipMemcpyAsync(cnt, &n, 4, hipMemcpyHostToDevice, ctx->stream));
  fd.head = head; fd.rois = rois; fd.counts = cnt; fd.images = 1; fd.max_rois = MAX_ROIS;
  fd.img_h = img_h; fd.img_w = img_w; fd.out_h = page_h; fd.out_w = page_w;
  fd.score_thr = score_thresh; fd.nms_thr = nms_thresh; fd.max_det = max_det;
  if ((rc = mhip_launch_det_final(ctx, fd))) return rc;
  MHIP_HIP(ctx, hipMemcpyAsync(boxes_out, fd.out_boxes, MAX_ROIS * 16, hipMemcpyDeviceToHost, ctx->stream));
  MHIP_HIP(ctx, hipMemcpyAsync(scores_out, fd.out_scores, MAX_ROIS * 4, hipMemcpyDeviceToHost, ctx->stream));
  MHIP_HIP(ctx, hipMemcpyAsync(count_out, fd.out_count, 4, hipMemcpyDeviceToHost, ctx->stream));
  MHIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return MHIP_OK;
}
