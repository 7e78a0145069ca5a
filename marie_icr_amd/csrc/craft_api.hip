// craft_api.hip — the CRAFT text detector behind the C ABI: weight packing, network forward, and score-map
// post-processing down to word boxes.  Host-side counterpart of CRAFT (marie/models/craft/craft.py:31-81),
// vgg16_bn (marie/models/craft/basenet/vgg16_bn.py:23-74), get_prediction's resize/normalise
// (marie/boxes/craft_box_processor.py:94-110, marie/models/craft/imgproc.py:26-71) and getDetBoxes_core
// (marie/models/craft/craft_utils.py:25-98).
#include <math.h>

#include <algorithm>
#include <map>

#include "common.h"

namespace {

struct HostTensor {
  std::vector<int64_t> shape;
  std::vector<float> data;
};

// forward-order conv table; `cinp`/`coutp` are the channel counts the kernels see (thin layers are zero-padded
// to 64 so every layer is a multiple of the MFMA K slice and of the 16-byte lane vector).
struct CLayer {
  const char* key;
  int co, ci, k;
  const char* bn;
  int coutp, cinp;
};
const CLayer kL[27] = {
    {"basenet.slice1.0", 64, 3, 3, "basenet.slice1.1", 64, 3},
    {"basenet.slice1.3", 64, 64, 3, "basenet.slice1.4", 64, 64},
    {"basenet.slice1.7", 128, 64, 3, "basenet.slice1.8", 128, 64},
    {"basenet.slice1.10", 128, 128, 3, "basenet.slice1.11", 128, 128},
    {"basenet.slice2.14", 256, 128, 3, "basenet.slice2.15", 256, 128},
    {"basenet.slice2.17", 256, 256, 3, "basenet.slice2.18", 256, 256},
    {"basenet.slice3.20", 256, 256, 3, "basenet.slice3.21", 256, 256},
    {"basenet.slice3.24", 512, 256, 3, "basenet.slice3.25", 512, 256},
    {"basenet.slice3.27", 512, 512, 3, "basenet.slice3.28", 512, 512},
    {"basenet.slice4.30", 512, 512, 3, "basenet.slice4.31", 512, 512},
    {"basenet.slice4.34", 512, 512, 3, "basenet.slice4.35", 512, 512},
    {"basenet.slice4.37", 512, 512, 3, "basenet.slice4.38", 512, 512},
    {"basenet.slice5.1", 1024, 512, 3, nullptr, 1024, 512},
    {"basenet.slice5.2", 1024, 1024, 1, nullptr, 1024, 1024},
    {"upconv1.conv.0", 512, 1536, 1, "upconv1.conv.1", 512, 1536},
    {"upconv1.conv.3", 256, 512, 3, "upconv1.conv.4", 256, 512},
    {"upconv2.conv.0", 256, 768, 1, "upconv2.conv.1", 256, 768},
    {"upconv2.conv.3", 128, 256, 3, "upconv2.conv.4", 128, 256},
    {"upconv3.conv.0", 128, 384, 1, "upconv3.conv.1", 128, 384},
    {"upconv3.conv.3", 64, 128, 3, "upconv3.conv.4", 64, 128},
    {"upconv4.conv.0", 64, 192, 1, "upconv4.conv.1", 64, 192},
    {"upconv4.conv.3", 32, 64, 3, "upconv4.conv.4", 64, 64},
    {"conv_cls.0", 32, 32, 3, nullptr, 64, 64},
    {"conv_cls.2", 32, 32, 3, nullptr, 64, 64},
    {"conv_cls.4", 16, 32, 3, nullptr, 64, 64},
    {"conv_cls.6", 16, 16, 1, nullptr, 64, 64},
    {"conv_cls.8", 2, 16, 1, nullptr, 2, 64},
};
constexpr int NL = 27;

size_t al256(size_t v) { return (v + 255) / 256 * 256; }

}  // namespace

struct mhip_craft {
  mhip_ctx* ctx = nullptr;
  int precision = MHIP_PREC_F16;
  std::map<std::string, HostTensor> tensors;
  size_t w_off[NL] = {0}, s_off[NL] = {0}, b_off[NL] = {0};
  size_t head_off = 0;   // fp32 [w1 16x16][b1 16][w2 2x16][b2 2] of conv_cls.6 / conv_cls.8 for the fused score head
  size_t arena_bytes = 0;
  char* arena = nullptr;
  bool ready = false;
  // pinned host staging for the post-processing read-back
  void* hpin = nullptr;
  size_t hpin_bytes = 0;
  size_t esz() const { return precision == MHIP_PREC_F16 ? 2 : 4; }
};

namespace {

void craft_layout(mhip_craft* m) {
  size_t o = 0;
  for (int i = 0; i < NL; ++i) {
    const CLayer& L = kL[i];
    const size_t wbytes = (i == 0) ? (size_t)27 * 64 * 4 : (size_t)L.coutp * L.k * L.k * L.cinp * m->esz();
    m->w_off[i] = o;
    o = al256(o + wbytes);
    m->s_off[i] = o;
    o = al256(o + (size_t)std::max(L.coutp, 64) * 4);
    m->b_off[i] = o;
    o = al256(o + (size_t)std::max(L.coutp, 64) * 4);
  }
  m->head_off = o;
  o = al256(o + (256 + 16 + 32 + 2) * 4);
  m->arena_bytes = o;
}

const HostTensor* cfind(const mhip_craft* m, const std::string& k, std::initializer_list<int64_t> shape) {
  auto it = m->tensors.find(k);
  if (it == m->tensors.end()) {
    mhip_fail(m->ctx, MHIP_ESTATE, "missing tensor %s", k.c_str());
    return nullptr;
  }
  if (it->second.shape != std::vector<int64_t>(shape)) {
    mhip_fail(m->ctx, MHIP_EINVAL, "tensor %s has the wrong shape", k.c_str());
    return nullptr;
  }
  return &it->second;
}

}  // namespace

extern "C" int mhip_craft_create(mhip_ctx* ctx, int precision, mhip_craft** out) {
  if (!ctx || !out) return MHIP_EINVAL;
  *out = nullptr;
  if (precision != MHIP_PREC_F16 && precision != MHIP_PREC_F32)
    return mhip_fail(ctx, MHIP_EINVAL, "unknown precision %d", precision);
  mhip_craft* m = new mhip_craft();
  m->ctx = ctx;
  m->precision = precision;
  craft_layout(m);
  *out = m;
  return MHIP_OK;
}

extern "C" int mhip_craft_destroy(mhip_craft* m) {
  if (!m) return MHIP_OK;
  mhip_quiesce(m->ctx);
  if (m->arena) (void)hipFree(m->arena);
  if (m->hpin) (void)hipHostFree(m->hpin);
  delete m;
  return MHIP_OK;
}

extern "C" int mhip_craft_set_tensor(mhip_craft* m, const char* key, const float* data, const int64_t* shape,
                                     int ndim) {
  if (!m || !key) return MHIP_EINVAL;
  std::string k(key);
  if (k.rfind("module.", 0) == 0) k = k.substr(7);   // copyStateDict (marie/boxes/box_processor.py) strips it too
  if (k.size() > 19 && k.compare(k.size() - 19, 19, "num_batches_tracked") == 0) return MHIP_OK;
  bool known = k.rfind("basenet.", 0) == 0 || k.rfind("upconv", 0) == 0 || k.rfind("conv_cls.", 0) == 0;
  if (!known) return mhip_fail(m->ctx, MHIP_EINVAL, "unknown state_dict key %s", key);
  if (!data || ndim < 0 || ndim > 4 || (ndim > 0 && !shape)) return mhip_fail(m->ctx, MHIP_EINVAL, "bad tensor %s", key);
  HostTensor t;
  size_t n = 1;
  for (int i = 0; i < ndim; ++i) {
    if (shape[i] <= 0) return mhip_fail(m->ctx, MHIP_EINVAL, "bad shape for %s", key);
    t.shape.push_back(shape[i]);
    n *= (size_t)shape[i];
  }
  t.data.assign(data, data + n);
  m->tensors[k] = std::move(t);
  m->ready = false;
  return MHIP_OK;
}

extern "C" int mhip_craft_alloc_arena(mhip_craft* m) {
  if (!m) return MHIP_EINVAL;
  if (!m->arena && hipMalloc((void**)&m->arena, m->arena_bytes) != hipSuccess) {
    (void)hipGetLastError();
    return mhip_fail(m->ctx, MHIP_ENOMEM, "arena allocation of %zu bytes failed", m->arena_bytes);
  }
  m->ready = true;
  return MHIP_OK;
}

extern "C" int mhip_craft_arena(mhip_craft* m, void** dev, size_t* bytes) {
  if (!m) return MHIP_EINVAL;
  if (dev) *dev = m->arena;
  if (bytes) *bytes = m->arena_bytes;
  return MHIP_OK;
}

extern "C" int mhip_craft_finalize(mhip_craft* m) {
  if (!m) return MHIP_EINVAL;
  mhip_ctx* ctx = m->ctx;
  std::vector<char> host(m->arena_bytes, 0);
  char* h = host.data();
  for (int i = 0; i < NL; ++i) {
    const CLayer& L = kL[i];
    const HostTensor* w = cfind(m, std::string(L.key) + ".weight", {L.co, L.ci, L.k, L.k});
    const HostTensor* b = cfind(m, std::string(L.key) + ".bias", {L.co});
    if (!w || !b) return MHIP_ESTATE;
    const int taps = L.k * L.k;
    if (i == 0) {  // [64][3][3][3] -> [(tap*3 + c)][64] fp32 for the VALU first layer
      float* dw = (float*)(h + m->w_off[0]);
      for (int o = 0; o < 64; ++o)
        for (int c = 0; c < 3; ++c)
          for (int t = 0; t < 9; ++t) dw[(t * 3 + c) * 64 + o] = w->data[((size_t)o * 3 + c) * 9 + t];
    } else {       // [Co][Ci][k][k] -> [Cop][k][k][Cip], zero padded
      std::vector<float> tmp((size_t)L.coutp * taps * L.cinp, 0.f);
      for (int o = 0; o < L.co; ++o)
        for (int c = 0; c < L.ci; ++c)
          for (int t = 0; t < taps; ++t)
            tmp[((size_t)o * taps + t) * L.cinp + c] = w->data[((size_t)o * L.ci + c) * taps + t];
      if (m->precision == MHIP_PREC_F16) {
        _Float16* d = (_Float16*)(h + m->w_off[i]);
        for (size_t q = 0; q < tmp.size(); ++q) d[q] = (_Float16)tmp[q];
      } else {
        memcpy(h + m->w_off[i], tmp.data(), tmp.size() * 4);
      }
    }
    float* sc = (float*)(h + m->s_off[i]);
    float* bi = (float*)(h + m->b_off[i]);
    const int np = std::max(L.coutp, 64);
    for (int o = 0; o < np; ++o) {
      sc[o] = 1.f;
      bi[o] = 0.f;      // padded output channels: 0 * x + 0 -> ReLU -> 0
    }
    for (int o = 0; o < L.co; ++o) bi[o] = b->data[o];
    if (L.bn) {
      const HostTensor* g = cfind(m, std::string(L.bn) + ".weight", {L.co});
      const HostTensor* be = cfind(m, std::string(L.bn) + ".bias", {L.co});
      const HostTensor* mu = cfind(m, std::string(L.bn) + ".running_mean", {L.co});
      const HostTensor* va = cfind(m, std::string(L.bn) + ".running_var", {L.co});
      if (!g || !be || !mu || !va) return MHIP_ESTATE;
      for (int o = 0; o < L.co; ++o) {
        const float s = g->data[o] / sqrtf(va->data[o] + 1e-5f);
        sc[o] = s;
        bi[o] = be->data[o] + (bi[o] - mu->data[o]) * s;
      }
    }
  }
  {
    const HostTensor* w1 = cfind(m, "conv_cls.6.weight", {16, 16, 1, 1});
    const HostTensor* b1 = cfind(m, "conv_cls.6.bias", {16});
    const HostTensor* w2 = cfind(m, "conv_cls.8.weight", {2, 16, 1, 1});
    const HostTensor* b2 = cfind(m, "conv_cls.8.bias", {2});
    if (!w1 || !b1 || !w2 || !b2) return MHIP_ESTATE;
    float* hd = (float*)(h + m->head_off);
    for (int i = 0; i < 256; ++i) {
      float v = w1->data[i];
      hd[i] = m->precision == MHIP_PREC_F16 ? (float)(_Float16)v : v;   // same rounding as the MFMA path's weights
    }
    memcpy(hd + 256, b1->data.data(), 16 * 4);
    for (int i = 0; i < 32; ++i) {
      float v = w2->data[i];
      hd[272 + i] = m->precision == MHIP_PREC_F16 ? (float)(_Float16)v : v;
    }
    memcpy(hd + 304, b2->data.data(), 2 * 4);
  }
  int rc = mhip_craft_alloc_arena(m);
  if (rc) return rc;
  m->ready = false;
  MHIP_HIP(ctx, hipMemcpy(m->arena, h, m->arena_bytes, hipMemcpyHostToDevice));
  m->ready = true;
  m->tensors.clear();
  return MHIP_OK;
}

// resize_aspect_ratio geometry (marie/models/craft/imgproc.py:45-71), same double arithmetic as CPython
extern "C" int mhip_craft_geometry(int h, int w, int canvas_size, double mag_ratio, double* ratio, int* th, int* tw,
                                   int* H32, int* W32) {
  if (h < 1 || w < 1 || canvas_size < 1) return MHIP_EINVAL;
  const int mx = h > w ? h : w;
  double target = mag_ratio * (double)mx;
  if (target > (double)canvas_size) target = (double)canvas_size;
  const double r = target / (double)mx;
  const int a = (int)((double)h * r), b = (int)((double)w * r);
  if (a < 1 || b < 1) return MHIP_EINVAL;
  if (ratio) *ratio = r;
  if (th) *th = a;
  if (tw) *tw = b;
  if (H32) *H32 = a % 32 ? a + (32 - a % 32) : a;
  if (W32) *W32 = b % 32 ? b + (32 - b % 32) : b;
  return MHIP_OK;
}

namespace {

struct CPlan {
  int th, tw, H, W;   // resized image, /32 canvas
  double ratio;
  size_t resized, tabs, act[32], total;
  size_t ccl;
};

// activation slots (byte offsets); all NHWC with the padded channel counts
enum Slot { A1_1, P1, A2_1, A2_2, P2, A3_1, A3_2, A3_3P, A4_1, A4_2, A4_3P, A5_1, A5_2, P5, A6, FC7, U1A, U1B, UP1,
            U2A, U2B, UP2, U3A, U3B, UP3, U4A, U4B, C0, C1, C2, C3, NSLOT };

int make_cplan(const mhip_craft* m, int h, int w, int canvas, double mag, CPlan* p) {
  int rc = mhip_craft_geometry(h, w, canvas, mag, &p->ratio, &p->th, &p->tw, &p->H, &p->W);
  if (rc) return rc;
  const size_t es = m->esz();
  const size_t H = p->H, W = p->W;
  size_t o = 0;
  auto take = [&](size_t bytes) {
    size_t at = o;
    o = (o + bytes + 4095) / 4096 * 4096;
    return at;
  };
  p->resized = take((size_t)p->th * p->tw * 3);
  p->tabs = take(((size_t)p->th + p->tw) * 8 + 64);
  const size_t px1 = H * W, px2 = px1 / 4, px4 = px1 / 16, px8 = px1 / 64, px16 = px1 / 256;
  const size_t ch[NSLOT] = {64, 64, 128, 128, 128, 256, 256, 256, 512, 512, 512, 512, 512, 512, 1024, 1024, 512, 256,
                            256, 256, 128, 128, 128, 64, 64, 64, 64, 64, 64, 64, 64};
  const size_t px[NSLOT] = {px1, px2, px2, px2, px4, px4, px4, px8, px8, px8, px16, px16, px16, px16, px16, px16, px16,
                            px16, px8, px8, px8, px4, px4, px4, px2, px2, px2, px2, px2, px2, px2};
  for (int s = 0; s < NSLOT; ++s) p->act[s] = take(px[s] * ch[s] * es);
  p->ccl = take(mhip_ccl_workspace_bytes((int)(H / 2), (int)(W / 2)));
  p->total = o;
  return MHIP_OK;
}

}  // namespace

extern "C" size_t mhip_craft_workspace_bytes(mhip_craft* m, int h, int w, int canvas_size, double mag_ratio) {
  CPlan p;
  if (!m || make_cplan(m, h, w, canvas_size, mag_ratio, &p)) return 0;
  return p.total;
}

extern "C" double mhip_craft_kernel_flops(mhip_craft* m, int kid, int h, int w, int canvas_size, double mag_ratio) {
  CPlan p;
  if (!m || make_cplan(m, h, w, canvas_size, mag_ratio, &p)) return 0.0;
  const double px1 = (double)p.H * p.W;
  // resolution divisor (pixels = px1 / div) of every layer of kL, in table order
  static const int div[NL] = {1, 1, 4, 4, 16, 16, 16, 64, 64, 64, 256, 256, 256, 256, 256, 256, 64, 64, 16, 16, 4, 4,
                              4, 4, 4, 4, 4};
  double first = 0.0, rest = 0.0;
  for (int i = 0; i < NL; ++i) {
    const double f = 2.0 * (px1 / div[i]) * kL[i].co * kL[i].ci * kL[i].k * kL[i].k;
    if (i == 0) first = f;
    else if (i < 25) rest += f;   // conv_cls.6 / .8 run in the fused score-head kernel, not on conv_igemm
  }
  if (kid == MHIP_K_CONV_FIRST) return first;
  if (kid == MHIP_K_CONV_IGEMM) return rest;
  return 0.0;
}

// Network forward: page uint8 [h][w][3] (device) -> scores fp32 [H32/2][W32/2][2] (device).
extern "C" int mhip_craft_forward(mhip_craft* m, const uint8_t* page_dev, int h, int w, int canvas_size,
                                  double mag_ratio, float* scores_dev) {
  if (!m) return MHIP_EINVAL;
  mhip_ctx* ctx = m->ctx;
  if (!m->ready || !m->arena) return mhip_fail(ctx, MHIP_ESTATE, "craft: weights not finalized");
  if (!page_dev || !scores_dev) return mhip_fail(ctx, MHIP_EINVAL, "craft: null buffer");
  CPlan p;
  if (make_cplan(m, h, w, canvas_size, mag_ratio, &p)) return mhip_fail(ctx, MHIP_EINVAL, "craft: bad page %dx%d", h, w);
  if ((long long)p.H * p.W > 0x7fffff00LL / 4) return mhip_fail(ctx, MHIP_EINVAL, "craft: canvas too large");
  MHIP_HIP(ctx, hipSetDevice(ctx->device));
  int rc = mhip_ensure_workspace(ctx, p.total);
  if (rc) return rc;
  char* ws = (char*)ctx->ws;
  const char* A = m->arena;
  const int prec = m->precision;
  const int H = p.H, W = p.W;

  // ---- resize (cv2 INTER_LINEAR, 8-bit fixed point); identity when the size is unchanged ----
  const uint8_t* img = page_dev;
  if (p.th != h || p.tw != w) {
    std::vector<int> xo, yo;
    std::vector<short> xa, yb;
    mhip_resize_linear_tables(w, p.tw, xo, xa);
    mhip_resize_linear_tables(h, p.th, yo, yb);
    char* t = ws + p.tabs;
    int* d_xo = (int*)t;
    int* d_yo = d_xo + p.tw;
    short* d_xa = (short*)(d_yo + p.th);
    short* d_yb = d_xa + 2 * p.tw;
    MHIP_HIP(ctx, hipMemcpyAsync(d_xo, xo.data(), (size_t)p.tw * 4, hipMemcpyHostToDevice, ctx->stream));
    MHIP_HIP(ctx, hipMemcpyAsync(d_yo, yo.data(), (size_t)p.th * 4, hipMemcpyHostToDevice, ctx->stream));
    MHIP_HIP(ctx, hipMemcpyAsync(d_xa, xa.data(), (size_t)p.tw * 4, hipMemcpyHostToDevice, ctx->stream));
    MHIP_HIP(ctx, hipMemcpyAsync(d_yb, yb.data(), (size_t)p.th * 4, hipMemcpyHostToDevice, ctx->stream));
    MHIP_HIP(ctx, hipStreamSynchronize(ctx->stream));  // the host tables die with this scope
    rc = mhip_launch_resize_linear_u8(ctx, page_dev, h, w, (uint8_t*)(ws + p.resized), p.th, p.tw, d_xo, d_xa, d_yo,
                                      d_yb);
    if (rc) return rc;
    img = (const uint8_t*)(ws + p.resized);
  }

  auto W_ = [&](int i) { return (const void*)(A + m->w_off[i]); };
  auto S_ = [&](int i) { return kL[i].bn ? (const float*)(A + m->s_off[i]) : nullptr; };
  auto B_ = [&](int i) { return (const float*)(A + m->b_off[i]); };
  auto act = [&](int s) { return (void*)(ws + p.act[s]); };

  auto conv = [&](int li, const void* in, int hh, int ww, void* out, int pool, int relu, int dil = 1, int pad = -1,
                  const void* in2 = nullptr, int cin1 = 0, int out_f32 = 0) {
    const CLayer& L = kL[li];
    ConvDesc d;
    d.in = in; d.w = W_(li); d.scale = S_(li); d.bias = B_(li); d.out = out;
    d.B = 1; d.H = hh; d.W = ww; d.Cin = L.cinp;
    d.KH = L.k; d.KW = L.k; d.pad = pad >= 0 ? pad : (L.k == 3 ? 1 : 0);
    d.N = L.coutp; d.pool = pool; d.relu = relu; d.dil = dil;
    d.in2 = in2; d.Cin1 = cin1; d.out_f32 = out_f32;
    return mhip_launch_conv_igemm(ctx, prec, d);
  };
#define CK(x) do { rc = (x); if (rc) return rc; } while (0)

  const int H2 = H / 2, W2 = W / 2, H4 = H / 4, W4 = W / 4, H8 = H / 8, W8 = W / 8, H16 = H / 16, W16 = W / 16;
  // ---- VGG16-BN backbone (features[0..38]) -------------------------------------------------
  CK(mhip_launch_conv_rgb_first(ctx, prec, img, p.th, p.tw, H, W, (const float*)W_(0), (const float*)(A + m->s_off[0]),
                                B_(0), act(A1_1)));
  CK(conv(1, act(A1_1), H, W, act(P1), POOL_2x2, 1));                 // conv1_2 + ReLU + pool
  CK(conv(2, act(P1), H2, W2, act(A2_1), POOL_NONE, 1));             // conv2_1
  CK(conv(3, act(A2_1), H2, W2, act(A2_2), POOL_NONE, 1));           // conv2_2 -> tap relu2_2 (post-ReLU: in-place)
  CK(mhip_launch_maxpool(ctx, prec, 2, act(A2_2), act(P2), 1, H2, W2, 128));
  CK(conv(4, act(P2), H4, W4, act(A3_1), POOL_NONE, 1));             // conv3_1
  CK(conv(5, act(A3_1), H4, W4, act(A3_2), POOL_NONE, 1));           // conv3_2 -> tap relu3_2
  CK(conv(6, act(A3_2), H4, W4, act(A3_3P), POOL_2x2, 1));           // conv3_3 + pool
  CK(conv(7, act(A3_3P), H8, W8, act(A4_1), POOL_NONE, 1));          // conv4_1
  CK(conv(8, act(A4_1), H8, W8, act(A4_2), POOL_NONE, 1));           // conv4_2 -> tap relu4_3
  CK(conv(9, act(A4_2), H8, W8, act(A4_3P), POOL_2x2, 1));           // conv4_3 + pool
  CK(conv(10, act(A4_3P), H16, W16, act(A5_1), POOL_NONE, 1));       // conv5_1
  CK(conv(11, act(A5_1), H16, W16, act(A5_2), POOL_NONE, 0));        // conv5_2 + BN, NO ReLU -> tap relu5_3
  CK(mhip_launch_maxpool(ctx, prec, 3, act(A5_2), act(P5), 1, H16, W16, 512));
  CK(conv(12, act(P5), H16, W16, act(A6), POOL_NONE, 0, 6, 6));      // fc6: 3x3 dilation 6
  CK(conv(13, act(A6), H16, W16, act(FC7), POOL_NONE, 0));           // fc7: 1x1
  // ---- U-net ---------------------------------------------------------------------------------
  CK(conv(14, act(FC7), H16, W16, act(U1A), POOL_NONE, 1, 1, 0, act(A5_2), 1024));
  CK(conv(15, act(U1A), H16, W16, act(U1B), POOL_NONE, 1));
  CK(mhip_launch_upsample_bilinear(ctx, prec, act(U1B), act(UP1), 1, H16, W16, 256, H8, W8));
  CK(conv(16, act(UP1), H8, W8, act(U2A), POOL_NONE, 1, 1, 0, act(A4_2), 256));
  CK(conv(17, act(U2A), H8, W8, act(U2B), POOL_NONE, 1));
  CK(mhip_launch_upsample_bilinear(ctx, prec, act(U2B), act(UP2), 1, H8, W8, 128, H4, W4));
  CK(conv(18, act(UP2), H4, W4, act(U3A), POOL_NONE, 1, 1, 0, act(A3_2), 128));
  CK(conv(19, act(U3A), H4, W4, act(U3B), POOL_NONE, 1));
  CK(mhip_launch_upsample_bilinear(ctx, prec, act(U3B), act(UP3), 1, H4, W4, 64, H2, W2));
  CK(conv(20, act(UP3), H2, W2, act(U4A), POOL_NONE, 1, 1, 0, act(A2_2), 64));
  CK(conv(21, act(U4A), H2, W2, act(U4B), POOL_NONE, 1));            // feature (32 real + 32 zero channels)
  // ---- conv_cls ------------------------------------------------------------------------------
  CK(conv(22, act(U4B), H2, W2, act(C0), POOL_NONE, 1));
  CK(conv(23, act(C0), H2, W2, act(C1), POOL_NONE, 1));
  CK(conv(24, act(C1), H2, W2, act(C2), POOL_NONE, 1));
  {
    const float* hd = (const float*)(A + m->head_off);   // conv_cls.6 + ReLU + conv_cls.8 fused per pixel
    CK(mhip_launch_score_head(ctx, prec, act(C2), 64, hd, hd + 256, hd + 272, hd + 304, scores_dev,
                              (long long)H2 * W2));
  }
#undef CK
  return MHIP_OK;
}

// ---------------------------------------------------------------------------------------------------
// Host finalisation of getDetBoxes_core: per surviving component, dilate its mask, take the minimum-area
// rectangle of the dilated pixel set (convex hull + rotating calipers in fp32), diamond fix, clockwise order.
// O(sum of component windows); the O(page) work was done on the GPU.
// ---------------------------------------------------------------------------------------------------
namespace {

struct Pt { long long x, y; };

long long cross3(const Pt& o, const Pt& a, const Pt& b) { return (a.x - o.x) * (b.y - o.y) - (a.y - o.y) * (b.x - o.x); }

#pragma clang fp contract(off)
void min_area_rect_box(std::vector<Pt>& pts, float box[4][2]) {
  std::sort(pts.begin(), pts.end(), [](const Pt& a, const Pt& b) { return a.x < b.x || (a.x == b.x && a.y < b.y); });
  pts.erase(std::unique(pts.begin(), pts.end(), [](const Pt& a, const Pt& b) { return a.x == b.x && a.y == b.y; }),
            pts.end());
  std::vector<Pt> hull;
  if (pts.size() <= 2) {
    hull = pts;
  } else {
    std::vector<Pt> lower, upper;
    for (const Pt& q : pts) {
      while (lower.size() >= 2 && cross3(lower[lower.size() - 2], lower.back(), q) <= 0) lower.pop_back();
      lower.push_back(q);
    }
    for (size_t i = pts.size(); i-- > 0;) {
      const Pt& q = pts[i];
      while (upper.size() >= 2 && cross3(upper[upper.size() - 2], upper.back(), q) <= 0) upper.pop_back();
      upper.push_back(q);
    }
    lower.pop_back();
    upper.pop_back();
    hull = lower;
    hull.insert(hull.end(), upper.begin(), upper.end());
  }
  const size_t n = hull.size();
  if (n == 1) {
    for (int i = 0; i < 4; ++i) { box[i][0] = (float)hull[0].x; box[i][1] = (float)hull[0].y; }
    return;
  }
  if (n == 2) {
    box[0][0] = box[1][0] = (float)hull[0].x; box[0][1] = box[1][1] = (float)hull[0].y;
    box[2][0] = box[3][0] = (float)hull[1].x; box[2][1] = box[3][1] = (float)hull[1].y;
    return;
  }
  std::vector<float> hx(n), hy(n);
  for (size_t i = 0; i < n; ++i) { hx[i] = (float)hull[i].x; hy[i] = (float)hull[i].y; }
  bool have = false;
  float best_area = 0.f, bux = 0, buy = 0, bvx = 0, bvy = 0, bu0 = 0, bu1 = 0, bv0 = 0, bv1 = 0;
  for (size_t i = 0; i < n; ++i) {
    const float ex = hx[(i + 1) % n] - hx[i], ey = hy[(i + 1) % n] - hy[i];
    const float ln = (float)sqrt((double)ex * (double)ex + (double)ey * (double)ey);
    const float ux = ex / ln, uy = ey / ln;
    const float vx = -uy, vy = ux;
    float u0 = 0, u1 = 0, v0 = 0, v1 = 0;
    for (size_t j = 0; j < n; ++j) {
      const float a = hx[j] * ux, b = hy[j] * uy;
      const float pu = a + b;
      const float c = hx[j] * vx, d = hy[j] * vy;
      const float pv = c + d;
      if (j == 0) { u0 = u1 = pu; v0 = v1 = pv; }
      else { u0 = fminf(u0, pu); u1 = fmaxf(u1, pu); v0 = fminf(v0, pv); v1 = fmaxf(v1, pv); }
    }
    const float du = u1 - u0, dv = v1 - v0;
    const float area = du * dv;
    if (!have || area < best_area) {
      have = true; best_area = area;
      bux = ux; buy = uy; bvx = vx; bvy = vy; bu0 = u0; bu1 = u1; bv0 = v0; bv1 = v1;
    }
  }
  const float ca[4] = {bu0, bu1, bu1, bu0}, cb[4] = {bv0, bv0, bv1, bv1};
  for (int i = 0; i < 4; ++i) {
    const float a = ca[i] * bux, b = cb[i] * bvx;
    box[i][0] = a + b;
    const float c = ca[i] * buy, d = cb[i] * bvy;
    box[i][1] = c + d;
  }
}

float norm2(float ax, float ay) {
  const float a = ax * ax, b = ay * ay;
  return sqrtf(a + b);
}

}  // namespace

// Full detector: forward + post-processing.  boxes_host receives up to max_boxes x 8 floats (4 corners x,y in
// score-map coordinates, clockwise from the top-left-most corner) in OpenCV label order.
extern "C" int mhip_craft_detect(mhip_craft* m, const uint8_t* page_dev, int h, int w, int canvas_size,
                                 double mag_ratio, float text_threshold, float link_threshold, float low_text,
                                 float* boxes_host, int max_boxes, int* n_boxes, float* scores_host,
                                 double* ratio_out) {
  if (!m || !n_boxes || (max_boxes > 0 && !boxes_host)) return MHIP_EINVAL;
  mhip_ctx* ctx = m->ctx;
  CPlan p;
  if (make_cplan(m, h, w, canvas_size, mag_ratio, &p)) return mhip_fail(ctx, MHIP_EINVAL, "craft: bad page %dx%d", h, w);
  const int H2 = p.H / 2, W2 = p.W / 2, npx = H2 * W2;
  int rc = mhip_ensure_workspace(ctx, p.total + (size_t)npx * 8 + 4096);
  if (rc) return rc;
  float* scores_dev = (float*)((char*)ctx->ws + p.total);
  rc = mhip_craft_forward(m, page_dev, h, w, canvas_size, mag_ratio, scores_dev);
  if (rc) return rc;
  char* ws = (char*)ctx->ws;
  CclBuffers cb;
  mhip_ccl_carve(ws + p.ccl, H2, W2, &cb);
  rc = mhip_launch_ccl(ctx, scores_dev, H2, W2, low_text, link_threshold, cb);
  if (rc) return rc;

  // ---- read back: n_labels, then stats + labels + flags (+ scores on request) into pinned memory ----
  const size_t need = 64 + (size_t)npx * 5 + (size_t)(npx / 2 + 2) * 24 + (scores_host ? (size_t)npx * 8 : 0);
  if (need > m->hpin_bytes) {
    if (m->hpin) (void)hipHostFree(m->hpin);
    m->hpin = nullptr;
    m->hpin_bytes = 0;
    if (hipHostMalloc(&m->hpin, need + need / 4, hipHostMallocDefault) != hipSuccess) {
      (void)hipGetLastError();
      return mhip_fail(ctx, MHIP_ENOMEM, "pinned staging of %zu bytes failed", need);
    }
    m->hpin_bytes = need + need / 4;
  }
  char* hp = (char*)m->hpin;
  int* h_n = (int*)hp;
  int* h_labels = (int*)(hp + 64);
  uint8_t* h_flags = (uint8_t*)(hp + 64 + (size_t)npx * 4);
  int* h_stats = (int*)(hp + 64 + (((size_t)npx * 5 + 63) / 64) * 64);
  MHIP_HIP(ctx, hipMemcpyAsync(h_n, cb.n_labels, 4, hipMemcpyDeviceToHost, ctx->stream));
  MHIP_HIP(ctx, hipMemcpyAsync(h_labels, cb.labels, (size_t)npx * 4, hipMemcpyDeviceToHost, ctx->stream));
  MHIP_HIP(ctx, hipMemcpyAsync(h_flags, cb.flags, (size_t)npx, hipMemcpyDeviceToHost, ctx->stream));
  MHIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  const int nlab = *h_n;
  if (nlab > 1) {
    MHIP_HIP(ctx, hipMemcpyAsync(h_stats, cb.stats, (size_t)nlab * 24, hipMemcpyDeviceToHost, ctx->stream));
  }
  if (scores_host)
    MHIP_HIP(ctx, hipMemcpyAsync(scores_host, scores_dev, (size_t)npx * 8, hipMemcpyDeviceToHost, ctx->stream));
  MHIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (ratio_out) *ratio_out = p.ratio;

  int nb = 0;
  std::vector<uint8_t> seg, tmp;
  std::vector<Pt> pts;
  for (int k = 1; k < nlab; ++k) {
    const int* s = h_stats + (size_t)k * 6;
    const int x = s[0], y = s[1], bw = s[2] - s[0] + 1, bh = s[3] - s[1] + 1, size = s[4];
    if (size < 10) continue;                                              // craft_utils.py:49-51
    if (mhip_ordered_bits_to_float(s[5]) < text_threshold) continue;      // :54-55
    const int niter = (int)(sqrt((double)size * (double)std::min(bw, bh) / ((double)bw * (double)bh)) * 2.0);
    int sx = x - niter, ex = x + bw + niter + 1, sy = y - niter, ey = y + bh + niter + 1;   // :62-71
    if (sx < 0) sx = 0;
    if (sy < 0) sy = 0;
    if (ex >= W2) ex = W2;
    if (ey >= H2) ey = H2;
    const int ww = ex - sx, wh = ey - sy;
    seg.assign((size_t)ww * wh, 0);
    for (int yy = y; yy < y + bh; ++yy)
      for (int xx = x; xx < x + bw; ++xx) {
        const size_t i = (size_t)yy * W2 + xx;
        // segmap[labels == k] = 255; segmap[link_score == 1 & text_score == 0] = 0   (:58-60)
        if (h_labels[i] == k && h_flags[i] != 2) seg[(size_t)(yy - sy) * ww + (xx - sx)] = 255;
      }
    const int ks = 1 + niter, an = ks / 2;
    if (ks > 1) {   // cv2.dilate with a ks x ks rectangle, anchor ks/2; binary mask -> window sums, O(window)
      tmp.assign(seg.size(), 0);
      std::vector<int> pre((size_t)std::max(ww, wh) + 1);
      for (int yy = 0; yy < wh; ++yy) {
        pre[0] = 0;
        for (int xx = 0; xx < ww; ++xx) pre[xx + 1] = pre[xx] + (seg[(size_t)yy * ww + xx] ? 1 : 0);
        for (int xx = 0; xx < ww; ++xx) {
          const int lo = std::max(0, xx - an), hi = std::min(ww - 1, xx - an + ks - 1);
          tmp[(size_t)yy * ww + xx] = (hi >= lo && pre[hi + 1] - pre[lo] > 0) ? 255 : 0;
        }
      }
      for (int xx = 0; xx < ww; ++xx) {
        pre[0] = 0;
        for (int yy = 0; yy < wh; ++yy) pre[yy + 1] = pre[yy] + (tmp[(size_t)yy * ww + xx] ? 1 : 0);
        for (int yy = 0; yy < wh; ++yy) {
          const int lo = std::max(0, yy - an), hi = std::min(wh - 1, yy - an + ks - 1);
          seg[(size_t)yy * ww + xx] = (hi >= lo && pre[hi + 1] - pre[lo] > 0) ? 255 : 0;
        }
      }
    }
    pts.clear();
    long long l = 1LL << 40, r = -1, t = 1LL << 40, b = -1;
    for (int yy = 0; yy < wh; ++yy) {
      // only the extreme pixels of a row can be hull vertices; min/max bounds need them too
      int first = -1, last = -1;
      for (int xx = 0; xx < ww; ++xx)
        if (seg[(size_t)yy * ww + xx]) {
          if (first < 0) first = xx;
          last = xx;
        }
      if (first >= 0) {
        pts.push_back({sx + first, sy + yy});
        if (last != first) pts.push_back({sx + last, sy + yy});
        l = std::min<long long>(l, sx + first);
        r = std::max<long long>(r, sx + last);
        t = std::min<long long>(t, sy + yy);
        b = std::max<long long>(b, sy + yy);
      }
    }
    float box[4][2] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};
    if (!pts.empty()) {
      min_area_rect_box(pts, box);
      const float bw_ = norm2(box[0][0] - box[1][0], box[0][1] - box[1][1]);
      const float bh_ = norm2(box[1][0] - box[2][0], box[1][1] - box[2][1]);
      const float ratio = fmaxf(bw_, bh_) / (fminf(bw_, bh_) + 1e-5f);
      if (fabsf(1.f - ratio) <= 0.1f) {   // align diamond-shape (:82-88)
        box[0][0] = (float)l; box[0][1] = (float)t;
        box[1][0] = (float)r; box[1][1] = (float)t;
        box[2][0] = (float)r; box[2][1] = (float)b;
        box[3][0] = (float)l; box[3][1] = (float)b;
      }
    }
    int start = 0;
    float best = box[0][0] + box[0][1];
    for (int i = 1; i < 4; ++i) {
      const float sm = box[i][0] + box[i][1];
      if (sm < best) { best = sm; start = i; }
    }
    if (nb < max_boxes) {
      float* o = boxes_host + (size_t)nb * 8;
      for (int i = 0; i < 4; ++i) {   // np.roll(box, 4 - startidx, 0): out[i] = box[(i + start) % 4]
        o[2 * i] = box[(i + start) % 4][0];
        o[2 * i + 1] = box[(i + start) % 4][1];
      }
    }
    ++nb;
  }
  *n_boxes = nb;
  if (nb > max_boxes) return mhip_fail(ctx, MHIP_EINVAL, "craft: %d boxes exceed the caller's capacity %d", nb, max_boxes);
  return MHIP_OK;
}

extern "C" int mhip_craft_detect_host(mhip_craft* m, const uint8_t* page_host, int h, int w, int canvas_size,
                                      double mag_ratio, float text_threshold, float link_threshold, float low_text,
                                      float* boxes_host, int max_boxes, int* n_boxes, float* scores_host,
                                      double* ratio_out) {
  if (!m || !page_host) return MHIP_EINVAL;
  mhip_ctx* ctx = m->ctx;
  CPlan p;
  if (make_cplan(m, h, w, canvas_size, mag_ratio, &p)) return mhip_fail(ctx, MHIP_EINVAL, "craft: bad page %dx%d", h, w);
  const size_t npx = (size_t)(p.H / 2) * (p.W / 2);
  const size_t page_off = p.total + npx * 8 + 8192;
  int rc = mhip_ensure_workspace(ctx, page_off + (size_t)h * w * 3 + 4096);
  if (rc) return rc;
  uint8_t* page_dev = (uint8_t*)ctx->ws + page_off;
  MHIP_HIP(ctx, hipMemcpyAsync(page_dev, page_host, (size_t)h * w * 3, hipMemcpyHostToDevice, ctx->stream));
  return mhip_craft_detect(m, page_dev, h, w, canvas_size, mag_ratio, text_threshold, link_threshold, low_text,
                           boxes_host, max_boxes, n_boxes, scores_host, ratio_out);
}
