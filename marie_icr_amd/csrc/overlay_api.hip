// overlay_api.hip — the overlay cleaner's generator (pix2pixHD LocalEnhancer, netG "local") behind the C ABI.
//
// replaces: OverlayProcessor.__extract_segmentation_mask + model.test() (marie/overlay/overlay.py:165-189 ->
// marie/models/pix2pix/models/test_model.py:66-68 -> networks_hd.py LocalEnhancer.forward :94-106).  The reference writes the
// page to a PNG, reads it back through a dataset, runs the network and converts the tensor to an image; here the page stays in
// HBM: u8 page -> normalised NHWC tensor -> generator -> u8 image, one launch sequence.
//   input pyramid   downsample = Conv2d(3, 3, k 3, s 2, p 1)
//   global branch   (half resolution, G = 2 ngf) ReflPad3 + Conv7 + IN + Swish; 3 x [Conv3 s2 + IN + Swish]; 9 ResnetBlocks at 8 G;
//                   3 x [bilinear x2 (align_corners) + Conv3 + Swish]
//   local branch    ReflPad3 + Conv7 (3 -> ngf) + IN + Swish; Conv3 s2 + IN + Swish; + global; 3 ResnetBlocks at 2 ngf;
//                   ConvTranspose2d(k 3, s 2) + IN + Swish; ReflPad3 + Conv7 (ngf -> 3) + Tanh
// Every convolution with >= 32 input channels runs on conv_igemm (MFMA): reflection padding is an explicit padded copy, a
// stride-2 convolution is computed at full width with vertical stride 2 and the InstanceNorm pass reads its even columns, the
// transposed convolution is a 3x3 convolution (flipped weights) over a zero-inserted image.  Spectral normalisation
// (weight_orig / u^T W v, eval mode) is folded into the packed weights.  The two 7x7 stems over the 3-channel image are GEMMs
// over an explicit [pixels][192] patch matrix (147 taps, zero-filled to 192); the 3 -> 3 input down-sampler is a direct kernel.
#include <math.h>

#include <string>
#include <vector>

#include "weights_util.h"

int mhip_ov_preprocess(mhip_ctx* ctx, int prec, const uint8_t* page, int h, int w, void* x, int H, int W);
int mhip_ov_conv_c3(mhip_ctx* ctx, int prec, const void* in, const float* wt, const float* bias, void* out, int H, int W, int cout, int K,
                    int stride, int pad, int refl);
int mhip_ov_reflect_pad(mhip_ctx* ctx, int prec, const void* in, void* out, int H, int W, int C, int p);
int mhip_ov_im2col7(mhip_ctx* ctx, int prec, const void* in, void* out, int H, int W);
int mhip_ov_instance_norm(mhip_ctx* ctx, int prec, const void* x, int Ho, int Wfull, int cstep, int C, float eps, int swish,
                          const void* res, void* out, float* stats);
int mhip_ov_upsample2x(mhip_ctx* ctx, int prec, const void* in, void* out, int H, int W, int C, int swish_in);
int mhip_ov_zero_insert(mhip_ctx* ctx, int prec, const void* in, void* out, int H, int W, int C);
int mhip_ov_add_swish(mhip_ctx* ctx, int prec, const void* a, const void* b, void* out, size_t n);
int mhip_ov_final(mhip_ctx* ctx, int prec, const void* y, uint8_t* rgb, float* raw, size_t P);
int mhip_ov_blend(mhip_ctx* ctx, const uint8_t* real_bgr, const uint8_t* mask, uint8_t* out, size_t P);

struct mhip_overlay {
  mhip_ctx* ctx = nullptr;
  int precision = MHIP_PREC_F16;
  int ngf = 64;
  TensorStore store;
  Arena arena;
  bool ready = false;
  size_t esz() const { return precision == MHIP_PREC_F16 ? 2 : 4; }
};

namespace {

constexpr float IN_EPS = 1e-5f;

struct ConvSpec { std::string name; int kind; int co, ci, k; };   // kind 0 plain, 1 spectral-normed, 2 spectral-normed transposed

std::vector<ConvSpec> conv_table(int ngf) {
  const int G = 2 * ngf;
  std::vector<ConvSpec> t = {{"downsample", 0, 3, 3, 3}, {"model.1", 1, G, 3, 7}, {"model.4", 1, 2 * G, G, 3},
                             {"model.7", 1, 4 * G, 2 * G, 3}, {"model.10", 1, 8 * G, 4 * G, 3}};
  for (int b = 0; b < 9; ++b) {
    t.push_back({"model." + std::to_string(13 + b) + ".conv_block.1", 1, 8 * G, 8 * G, 3});
    t.push_back({"model." + std::to_string(13 + b) + ".conv_block.5", 1, 8 * G, 8 * G, 3});
  }
  t.push_back({"model.23", 1, 4 * G, 8 * G, 3});
  t.push_back({"model.26", 1, 2 * G, 4 * G, 3});
  t.push_back({"model.29", 1, G, 2 * G, 3});
  t.push_back({"model1_1.1", 1, ngf, 3, 7});
  t.push_back({"model1_1.4", 1, 2 * ngf, ngf, 3});
  for (int b = 0; b < 3; ++b) {
    t.push_back({"model1_2." + std::to_string(b) + ".conv_block.1", 1, 2 * ngf, 2 * ngf, 3});
    t.push_back({"model1_2." + std::to_string(b) + ".conv_block.5", 1, 2 * ngf, 2 * ngf, 3});
  }
  t.push_back({"model1_2.3", 2, ngf, 2 * ngf, 3});
  t.push_back({"model1_2.7", 1, 3, ngf, 7});
  return t;
}

int conv(mhip_overlay* m, const void* in, const std::string& name, void* out, int H, int W, int Cin, int N, int k, int pad, int sy,
         int ldc = 0) {
  ConvDesc c;
  c.in = in; c.w = m->arena.d(name + "_w"); c.bias = m->arena.d<float>(name + "_b"); c.out = out;
  c.B = 1; c.H = H; c.W = W; c.Cin = Cin; c.KH = c.KW = k; c.pad = pad; c.N = N; c.sy = sy; c.ldc = ldc; c.pad_cols_writable = ldc ? 1 : 0;
  return mhip_launch_conv_igemm(m->ctx, m->precision, c);
}

}  // namespace

extern "C" int mhip_overlay_create(mhip_ctx* ctx, int precision, int ngf, mhip_overlay** out) {
  if (!ctx || !out) return MHIP_EINVAL;
  *out = nullptr;
  if (precision != MHIP_PREC_F16 && precision != MHIP_PREC_F32) return mhip_fail(ctx, MHIP_EINVAL, "unknown precision %d", precision);
  const int need = precision == MHIP_PREC_F16 ? 64 : 32;        // conv_igemm walks K in 128-byte slices
  if (ngf < need || ngf % need || ngf > 128)
    return mhip_fail(ctx, MHIP_EINVAL, "overlay: ngf %d (needs a multiple of %d, at most 128)", ngf, need);
  mhip_overlay* m = new mhip_overlay();
  m->ctx = ctx; m->precision = precision; m->ngf = ngf;
  const size_t es = m->esz();
  for (const ConvSpec& s : conv_table(ngf)) {
    const bool direct = s.ci == 3 && s.k == 3, stem = s.ci == 3 && s.k == 7;       // stems: GEMM over a [pixels][192] patch matrix
    m->arena.take(s.name + "_w", direct ? (size_t)s.k * s.k * 3 * s.co * 4 : (stem ? (size_t)s.co * 192 * es : (size_t)s.co * s.k * s.k * s.ci * es));
    m->arena.take(s.name + "_b", (size_t)s.co * 4);
  }
  *out = m;
  return MHIP_OK;
}

extern "C" int mhip_overlay_destroy(mhip_overlay* m) {
  if (!m) return MHIP_OK;
  mhip_quiesce(m->ctx);
  m->arena.release();
  delete m;
  return MHIP_OK;
}

extern "C" int mhip_overlay_set_tensor(mhip_overlay* m, const char* key, const float* data, const int64_t* shape, int ndim) {
  if (!m || !key) return MHIP_EINVAL;
  m->ready = false;
  std::string k(key);
  if (k.rfind("netG.", 0) == 0) k = k.substr(5);
  if (k.rfind("module.", 0) == 0) k = k.substr(7);
  return m->store.set(m->ctx, k, data, shape, ndim);
}

extern "C" int mhip_overlay_finalize(mhip_overlay* m) {
  if (!m) return MHIP_EINVAL;
  mhip_ctx* ctx = m->ctx;
  const TensorStore& st = m->store;
  Arena& a = m->arena;
  a.begin_fill();
  for (const ConvSpec& s : conv_table(m->ngf)) {
    const int kk = s.k * s.k;
    const std::vector<int64_t> wshape = s.kind == 2 ? std::vector<int64_t>{s.ci, s.co, s.k, s.k} : std::vector<int64_t>{s.co, s.ci, s.k, s.k};
    const HostTensor* w = st.find(ctx, s.name + (s.kind ? ".weight_orig" : ".weight"), wshape);
    const HostTensor* b = st.find(ctx, s.name + ".bias", {s.co});
    if (!w || !b) return MHIP_ESTATE;
    double sigma = 1.0;
    if (s.kind) {     // torch.nn.utils.spectral_norm, eval: sigma = u^T W_mat v, W_mat = weight with the output dim first
      const int cols = s.ci * kk;
      const HostTensor* u = st.find(ctx, s.name + ".weight_u", {s.co});
      const HostTensor* v = st.find(ctx, s.name + ".weight_v", {cols});
      if (!u || !v) return MHIP_ESTATE;
      sigma = 0.0;
      for (int o = 0; o < s.co; ++o) {
        double row = 0.0;
        for (int c = 0; c < s.ci; ++c)
          for (int t = 0; t < kk; ++t) {
            const float wv = s.kind == 2 ? w->data[((size_t)c * s.co + o) * kk + t] : w->data[((size_t)o * s.ci + c) * kk + t];
            row += (double)wv * v->data[(size_t)c * kk + t];
          }
        sigma += row * u->data[o];
      }
      if (!(fabs(sigma) > 1e-20)) return mhip_fail(ctx, MHIP_EINVAL, "overlay: degenerate spectral norm of %s", s.name.c_str());
    }
    const float inv = (float)(1.0 / sigma);
    auto at = [&](int o, int c, int ky, int kx) -> float {   // the equivalent forward-convolution weight
      if (s.kind == 2) return w->data[((size_t)c * s.co + o) * kk + (s.k - 1 - ky) * s.k + (s.k - 1 - kx)] * inv;
      return w->data[((size_t)o * s.ci + c) * kk + ky * s.k + kx] * inv;
    };
    if (s.ci == 3 && s.k == 7) {   // stem: [co][192], k = tap * 3 + c, zero beyond 147
      std::vector<float> tmp((size_t)s.co * 192, 0.f);
      for (int o = 0; o < s.co; ++o)
        for (int t = 0; t < kk; ++t)
          for (int c = 0; c < 3; ++c) tmp[(size_t)o * 192 + t * 3 + c] = at(o, c, t / s.k, t % s.k);
      Arena::put(m->precision, a.h(s.name + "_w"), tmp.data(), tmp.size());
    } else if (s.ci == 3) {   // direct kernel: fp32 [tap][ci][co]
      float* d = (float*)a.h(s.name + "_w");
      for (int t = 0; t < kk; ++t)
        for (int c = 0; c < 3; ++c)
          for (int o = 0; o < s.co; ++o) d[((size_t)t * 3 + c) * s.co + o] = at(o, c, t / s.k, t % s.k);
    } else {           // conv_igemm: [co][tap][ci]
      std::vector<float> tmp((size_t)s.co * kk * s.ci);
      for (int o = 0; o < s.co; ++o)
        for (int t = 0; t < kk; ++t)
          for (int c = 0; c < s.ci; ++c) tmp[((size_t)o * kk + t) * s.ci + c] = at(o, c, t / s.k, t % s.k);
      Arena::put(m->precision, a.h(s.name + "_w"), tmp.data(), tmp.size());
    }
    memcpy(a.h(s.name + "_b"), b->data.data(), (size_t)s.co * 4);
  }
  int rc = a.upload(ctx);
  if (rc) return rc;
  m->ready = true;
  m->store.t.clear();
  return MHIP_OK;
}

extern "C" int mhip_overlay_padded_shape(int h, int w, int* H, int* W) {
  if (!H || !W || h < 1 || w < 1) return MHIP_EINVAL;
  // OverlayProcessor.preprocess (overlay.py:147-163): when either side is ragged BOTH grow to the next multiple of 32
  if (h % 32 || w % 32) { *H = h / 32 * 32 + 32; *W = w / 32 * 32 + 32; }
  else { *H = h; *W = w; }
  return MHIP_OK;
}

static size_t overlay_ws_bytes(const mhip_overlay* m, int H, int W) {
  // exactly what overlay_run carves, in its order (every take is rounded up to 256 bytes)
  const size_t es = m->esz(), P = (size_t)H * W, ngf = m->ngf, G = 2 * ngf, Hh = H / 2, Wh = W / 2;
  auto up = [](size_t v) { return (v + 255) / 256 * 256; };
  size_t b = up(3 * 2048 * 4);                              // stats
  b += up(P * 4 * es) + up(Hh * Wh * 4 * es);               // x4, half4
  b += 3 * up((Hh + 2) * (Wh + 2) * 2 * G * es);            // g, t1, t2
  b += up(P * 192 * es);                                    // patches of the 7x7 stems
  b += 2 * up((size_t)(H + 6) * (W + 6) * ngf * es);        // l1, l2
  b += up((size_t)(H + 2) * (W + 2) * G * es);              // zero-inserted canvas
  b += up(P * 8 * es);                                      // y8
  return b + 4096;
}

// page_dev u8 BGR [h][w][3] -> fake_rgb_dev u8 RGB [H][W][3] on the padded canvas (mhip_overlay_padded_shape); raw_dev (optional)
// fp32 [H][W][3] tanh outputs; real_bgr_dev (optional) the padded white-canvas copy of the page the blend needs.
static int overlay_run(mhip_overlay* m, const uint8_t* page_dev, int h, int w, uint8_t* fake_rgb_dev, float* raw_dev) {
  mhip_ctx* ctx = m->ctx;
  if (!m->ready) return mhip_fail(ctx, MHIP_ESTATE, "overlay: weights not finalized");
  MHIP_HIP(ctx, hipSetDevice(ctx->device));
  int H, W;
  mhip_overlay_padded_shape(h, w, &H, &W);
  const int prec = m->precision, ngf = m->ngf, G = 2 * ngf;
  const size_t es = m->esz();
  int rc = mhip_ensure_workspace(ctx, overlay_ws_bytes(m, H, W));
  if (rc) return rc;
  Carver ws(ctx->ws);
  const Arena& a = m->arena;
  float* stats = ws.take<float>(3 * 2048 * 4);
  const int Hh = H / 2, Wh = W / 2;
  // ---- input
  char* x4 = ws.take((size_t)H * W * 4 * es);
  if ((rc = mhip_ov_preprocess(ctx, prec, page_dev, h, w, x4, H, W))) return rc;
  char* half4 = ws.take((size_t)Hh * Wh * 4 * es);
  if ((rc = mhip_ov_conv_c3(ctx, prec, x4, a.d<float>("downsample_w"), a.d<float>("downsample_b"), half4, H, W, 3, 3, 2, 1, 0))) return rc;
  // scratch big enough for any map of the global branch incl. frames (the largest: 2 G channels at half resolution, the
  // up-sampled input of the last 3x3 convolution)
  const size_t gmax = (size_t)(Hh + 2) * (Wh + 2) * 2 * G * es;
  char* g = ws.take(gmax);
  char* t1 = ws.take(gmax);
  char* t2 = ws.take(gmax);
  // ---- global branch
  auto stem = [&](const void* img4, const char* name, void* out, int hs, int wsz, int cout, void* patches) -> int {
    int r = mhip_ov_im2col7(ctx, prec, img4, patches, hs, wsz);
    if (r) return r;
    ConvDesc c;
    c.in = patches; c.w = a.d(std::string(name) + "_w"); c.bias = a.d<float>(std::string(name) + "_b"); c.out = out;
    c.B = 1; c.H = hs; c.W = wsz; c.Cin = 192; c.KH = c.KW = 1; c.pad = 0; c.N = cout;
    return mhip_launch_conv_igemm(ctx, prec, c);
  };
  char* patches = ws.take((size_t)H * W * 192 * es);
  if ((rc = stem(half4, "model.1", t1, Hh, Wh, G, patches))) return rc;
  if ((rc = mhip_ov_instance_norm(ctx, prec, t1, Hh, Wh, 1, G, IN_EPS, 1, nullptr, g, stats))) return rc;
  int ch = G, hh = Hh, wh = Wh;
  for (const char* n : {"model.4", "model.7", "model.10"}) {
    // 3x3 stride 2: vertical stride in the convolution, horizontal stride = the even columns the norm pass reads
    if ((rc = conv(m, g, n, t1, hh, wh, ch, 2 * ch, 3, 1, 2))) return rc;
    hh /= 2;
    if ((rc = mhip_ov_instance_norm(ctx, prec, t1, hh, wh, 2, 2 * ch, IN_EPS, 1, nullptr, g, stats))) return rc;
    wh /= 2;
    ch *= 2;
  }
  for (int b = 0; b < 9; ++b) {
    const std::string n = "model." + std::to_string(13 + b) + ".conv_block.";
    if ((rc = mhip_ov_reflect_pad(ctx, prec, g, t1, hh, wh, ch, 1))) return rc;
    if ((rc = conv(m, t1, n + "1", t2, hh + 2, wh + 2, ch, ch, 3, 0, 1))) return rc;
    if ((rc = mhip_ov_instance_norm(ctx, prec, t2, hh, wh, 1, ch, IN_EPS, 1, nullptr, t2, stats))) return rc;
    if ((rc = mhip_ov_reflect_pad(ctx, prec, t2, t1, hh, wh, ch, 1))) return rc;
    if ((rc = conv(m, t1, n + "5", t2, hh + 2, wh + 2, ch, ch, 3, 0, 1))) return rc;
    if ((rc = mhip_ov_instance_norm(ctx, prec, t2, hh, wh, 1, ch, IN_EPS, 0, g, g, stats))) return rc;     // x + IN(conv)
  }
  int first = 1;
  for (const char* n : {"model.23", "model.26", "model.29"}) {
    if ((rc = mhip_ov_upsample2x(ctx, prec, g, t1, hh, wh, ch, first ? 0 : 1))) return rc;                // swish of the previous conv
    hh *= 2; wh *= 2;
    if ((rc = conv(m, t1, n, g, hh, wh, ch, ch / 2, 3, 1, 1))) return rc;
    ch /= 2;
    first = 0;
  }
  // g: [Hh][Wh][G] before its swish
  // ---- local branch
  const size_t lmax = (size_t)(H + 6) * (W + 6) * ngf * es;
  char* l1 = ws.take(lmax);
  char* l2 = ws.take(lmax);
  if ((rc = stem(x4, "model1_1.1", l1, H, W, ngf, patches))) return rc;
  if ((rc = mhip_ov_instance_norm(ctx, prec, l1, H, W, 1, ngf, IN_EPS, 1, nullptr, l1, stats))) return rc;
  if ((rc = conv(m, l1, "model1_1.4", l2, H, W, ngf, G, 3, 1, 2))) return rc;                                 // [Hh][W][G]
  if ((rc = mhip_ov_instance_norm(ctx, prec, l2, Hh, W, 2, G, IN_EPS, 1, nullptr, t1, stats))) return rc;     // [Hh][Wh][G]
  if ((rc = mhip_ov_add_swish(ctx, prec, t1, g, g, (size_t)Hh * Wh * G))) return rc;                          // model_downsample(x) + output_prev
  for (int b = 0; b < 3; ++b) {
    const std::string n = "model1_2." + std::to_string(b) + ".conv_block.";
    if ((rc = mhip_ov_reflect_pad(ctx, prec, g, t1, Hh, Wh, G, 1))) return rc;
    if ((rc = conv(m, t1, n + "1", t2, Hh + 2, Wh + 2, G, G, 3, 0, 1))) return rc;
    if ((rc = mhip_ov_instance_norm(ctx, prec, t2, Hh, Wh, 1, G, IN_EPS, 1, nullptr, t2, stats))) return rc;
    if ((rc = mhip_ov_reflect_pad(ctx, prec, t2, t1, Hh, Wh, G, 1))) return rc;
    if ((rc = conv(m, t1, n + "5", t2, Hh + 2, Wh + 2, G, G, 3, 0, 1))) return rc;
    if ((rc = mhip_ov_instance_norm(ctx, prec, t2, Hh, Wh, 1, G, IN_EPS, 0, g, g, stats))) return rc;
  }
  // ConvTranspose2d(G -> ngf, k 3, s 2, p 1, output_padding 1): zero-inserted canvas [H + 2][W + 2][G], 3x3 convolution, no padding
  char* zi = ws.take((size_t)(H + 2) * (W + 2) * G * es);
  if ((rc = mhip_ov_zero_insert(ctx, prec, g, zi, Hh, Wh, G))) return rc;
  if ((rc = conv(m, zi, "model1_2.3", l1, H + 2, W + 2, G, ngf, 3, 0, 1))) return rc;
  if ((rc = mhip_ov_instance_norm(ctx, prec, l1, H, W, 1, ngf, IN_EPS, 1, nullptr, l1, stats))) return rc;
  if ((rc = mhip_ov_reflect_pad(ctx, prec, l1, l2, H, W, ngf, 3))) return rc;
  char* y8 = ws.take((size_t)H * W * 8 * es);
  if ((rc = conv(m, l2, "model1_2.7", y8, H + 6, W + 6, ngf, 3, 7, 0, 1, 8))) return rc;
  return mhip_ov_final(ctx, prec, y8, fake_rgb_dev, raw_dev, (size_t)H * W);
}

extern "C" int mhip_overlay_forward(mhip_overlay* m, const uint8_t* page_dev, int h, int w, uint8_t* fake_rgb_dev) {
  if (!m || !page_dev || !fake_rgb_dev || h < 1 || w < 1) return MHIP_EINVAL;
  return overlay_run(m, page_dev, h, w, fake_rgb_dev, nullptr);
}

// host page in, host image out (+ the raw tanh outputs for the parity tests)
extern "C" int mhip_overlay_forward_host(mhip_overlay* m, const uint8_t* page_host, int h, int w, uint8_t* fake_rgb_host, float* raw_host) {
  if (!m || !page_host || !fake_rgb_host || h < 1 || w < 1) return MHIP_EINVAL;
  mhip_ctx* ctx = m->ctx;
  MHIP_HIP(ctx, hipSetDevice(ctx->device));
  int H, W;
  mhip_overlay_padded_shape(h, w, &H, &W);
  const size_t P = (size_t)H * W;
  uint8_t* dpage = nullptr;
  uint8_t* dfake = nullptr;
  float* draw = nullptr;
  MHIP_HIP(ctx, hipMalloc((void**)&dpage, (size_t)h * w * 3));
  hipError_t e = hipMalloc((void**)&dfake, P * 3);
  if (e == hipSuccess && raw_host) e = hipMalloc((void**)&draw, P * 3 * 4);
  int rc = e == hipSuccess ? MHIP_OK : mhip_fail(ctx, MHIP_ENOMEM, "overlay: %s", hipGetErrorString(e));
  if (!rc && hipMemcpy(dpage, page_host, (size_t)h * w * 3, hipMemcpyHostToDevice) != hipSuccess) rc = mhip_fail(ctx, MHIP_EHIP, "overlay: upload");
  if (!rc) rc = overlay_run(m, dpage, h, w, dfake, draw);
  if (!rc && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = mhip_fail(ctx, MHIP_EHIP, "overlay: kernel failure");
  if (!rc && hipMemcpy(fake_rgb_host, dfake, P * 3, hipMemcpyDeviceToHost) != hipSuccess) rc = mhip_fail(ctx, MHIP_EHIP, "overlay: download");
  if (!rc && raw_host && hipMemcpy(raw_host, draw, P * 3 * 4, hipMemcpyDeviceToHost) != hipSuccess) rc = mhip_fail(ctx, MHIP_EHIP, "overlay: download");
  (void)hipFree(dpage);
  (void)hipFree(dfake);
  if (draw) (void)hipFree(draw);
  return rc;
}

// blend_to_text on device images of P pixels: real BGR (the padded page), mask = the generator's image (RGB) -> BGR (gray x 3)
extern "C" int mhip_overlay_blend(mhip_ctx* ctx, const uint8_t* real_bgr_dev, const uint8_t* mask_dev, uint8_t* out_dev, size_t pixels) {
  if (!ctx || !real_bgr_dev || !mask_dev || !out_dev) return MHIP_EINVAL;
  MHIP_HIP(ctx, hipSetDevice(ctx->device));
  return mhip_ov_blend(ctx, real_bgr_dev, mask_dev, out_dev, pixels);
}
