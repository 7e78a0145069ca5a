// icr_api.hip — the production recognizer TPS-ResNet-BiLSTM-Attn behind the C ABI: weight packing and forward.
// Host-side counterpart of Model(opt) with Transformation="TPS", FeatureExtraction="ResNet", SequenceModeling="BiLSTM",
// Prediction="Attn" (marie/models/icr/model.py:25-92; configured at marie/document/craft_ocr_processor.py:49-70:
// imgH 32, imgW 100, 20 fiducials, 1 input channel, 512 output channels, hidden 256, batch_max_length 48).
#include <math.h>

#include <map>

#include "common.h"

namespace {

struct HostTensor {
  std::vector<int64_t> shape;
  std::vector<float> data;
};

// conv + BatchNorm pairs in forward/state_dict order (marie_icr_amd/weights.py::icr_conv_table mirrors this)
struct CB {
  std::string conv, bn;
  int co, ci, k;
  int cop, cip;   // channel counts the kernels see (thin layers zero-padded to 64)
};

std::vector<CB> build_table() {
  std::vector<CB> t;
  const std::string loc = "Transformation.LocalizationNetwork.conv.";
  const int lc[4][4] = {{0, 1, 64, 1}, {4, 5, 128, 64}, {8, 9, 256, 128}, {12, 13, 512, 256}};
  for (auto& e : lc) t.push_back({loc + std::to_string(e[0]), loc + std::to_string(e[1]), e[2], e[3], 3, e[2], e[3]});
  const std::string r = "FeatureExtraction.ConvNet.";
  t.push_back({r + "conv0_1", r + "bn0_1", 32, 1, 3, 64, 1});
  t.push_back({r + "conv0_2", r + "bn0_2", 64, 32, 3, 64, 64});
  int inpl = 64;
  const int planes_[4] = {128, 256, 512, 512}, blocks_[4] = {1, 2, 5, 3};
  for (int li = 1; li <= 4; ++li) {
    const int planes = planes_[li - 1];
    for (int b = 0; b < blocks_[li - 1]; ++b) {
      const std::string p = r + "layer" + std::to_string(li) + "." + std::to_string(b) + ".";
      const int cin = b == 0 ? inpl : planes;
      t.push_back({p + "conv1", p + "bn1", planes, cin, 3, planes, cin});
      t.push_back({p + "conv2", p + "bn2", planes, planes, 3, planes, planes});
      if (b == 0 && inpl != planes) t.push_back({p + "downsample.0", p + "downsample.1", planes, inpl, 1, planes, inpl});
    }
    inpl = planes;
    if (li < 4) t.push_back({r + "conv" + std::to_string(li), r + "bn" + std::to_string(li), planes, planes, 3, planes, planes});
  }
  t.push_back({r + "conv4_1", r + "bn4_1", 512, 512, 2, 512, 512});
  t.push_back({r + "conv4_2", r + "bn4_2", 512, 512, 2, 512, 512});
  return t;
}

size_t al256(size_t v) { return (v + 255) / 256 * 256; }

}  // namespace

struct mhip_icr {
  mhip_ctx* ctx = nullptr;
  int precision = MHIP_PREC_F16;
  int num_class = 96;
  std::vector<CB> tab;
  std::map<std::string, int> tab_index;
  std::map<std::string, HostTensor> tensors;
  // arena offsets
  std::vector<size_t> w_off, s_off, b_off;
  size_t fc1_w = 0, fc1_b = 0, fc2_w = 0, fc2_b = 0, idc = 0, phat = 0;
  size_t ih_w[2] = {0}, ih_b[2] = {0}, hh_pack[2] = {0}, lin_w[2] = {0}, lin_b[2] = {0};
  size_t i2h_w = 0, hg_w = 0, hg_b = 0, score_w = 0, ihc_w = 0, onehot_w = 0, gen_w = 0, gen_b = 0;
  size_t arena_bytes = 0;
  char* arena = nullptr;
  bool ready = false;
  size_t esz() const { return precision == MHIP_PREC_F16 ? 2 : 4; }
};

namespace {

constexpr int IMG_H = 32, IMG_W = 100, NFID = 20, MAXLEN = 48, STEPS = MAXLEN + 1;

void icr_layout(mhip_icr* m) {
  size_t o = 0;
  const size_t es = m->esz();
  auto take = [&](size_t bytes) {
    size_t at = o;
    o = al256(o + bytes);
    return at;
  };
  const size_t n = m->tab.size();
  m->w_off.resize(n);
  m->s_off.resize(n);
  m->b_off.resize(n);
  for (size_t i = 0; i < n; ++i) {
    const CB& c = m->tab[i];
    const bool gray = c.ci == 1;
    m->w_off[i] = take(gray ? (size_t)9 * c.co * 4 : (size_t)c.cop * c.k * c.k * c.cip * es);
    m->s_off[i] = take((size_t)std::max(c.cop, 64) * 4);
    m->b_off[i] = take((size_t)std::max(c.cop, 64) * 4);
  }
  m->fc1_w = take((size_t)256 * 512 * es);
  m->fc1_b = take(256 * 4);
  m->fc2_w = take((size_t)2 * NFID * 256 * es);
  m->fc2_b = take(2 * NFID * 4);
  m->idc = take((size_t)(NFID + 3) * (NFID + 3) * 4);
  m->phat = take((size_t)IMG_H * IMG_W * (NFID + 3) * 4);
  for (int j = 0; j < 2; ++j) {
    const int in = j == 0 ? 512 : 256;
    m->ih_w[j] = take((size_t)2048 * in * es);
    m->ih_b[j] = take(2048 * 4);
    m->hh_pack[j] = take(mhip_lstm_wpack_bytes(m->precision));
    m->lin_w[j] = take((size_t)256 * 512 * es);
    m->lin_b[j] = take(256 * 4);
  }
  m->i2h_w = take((size_t)256 * 256 * es);
  m->hg_w = take((size_t)1280 * 256 * es);
  m->hg_b = take(1280 * 4);
  m->score_w = take(256 * 4);
  m->ihc_w = take((size_t)1024 * 256 * es);
  m->onehot_w = take((size_t)m->num_class * 1024 * 4);
  m->gen_w = take((size_t)m->num_class * 256 * es);
  m->gen_b = take((size_t)m->num_class * 4);
  m->arena_bytes = o;
}

const HostTensor* ifind(const mhip_icr* m, const std::string& k, std::vector<int64_t> shape) {
  auto it = m->tensors.find(k);
  if (it == m->tensors.end()) {
    mhip_fail(m->ctx, MHIP_ESTATE, "missing tensor %s", k.c_str());
    return nullptr;
  }
  if (it->second.shape != shape) {
    mhip_fail(m->ctx, MHIP_EINVAL, "tensor %s has the wrong shape", k.c_str());
    return nullptr;
  }
  return &it->second;
}

void putT(const mhip_icr* m, char* dst, const float* src, size_t n) {
  if (m->precision == MHIP_PREC_F16) {
    _Float16* d = (_Float16*)dst;
    for (size_t i = 0; i < n; ++i) d[i] = (_Float16)src[i];
  } else {
    memcpy(dst, src, n * 4);
  }
}

}  // namespace

extern "C" int mhip_icr_create(mhip_ctx* ctx, int precision, int num_class, mhip_icr** out) {
  if (!ctx || !out) return MHIP_EINVAL;
  *out = nullptr;
  if (precision != MHIP_PREC_F16 && precision != MHIP_PREC_F32)
    return mhip_fail(ctx, MHIP_EINVAL, "unknown precision %d", precision);
  if (num_class < 3 || num_class > 256 || (num_class * 4) % 16 != 0)
    return mhip_fail(ctx, MHIP_EINVAL, "num_class %d must be in [3,256] and a multiple of 4", num_class);
  mhip_icr* m = new mhip_icr();
  m->ctx = ctx;
  m->precision = precision;
  m->num_class = num_class;
  m->tab = build_table();
  for (size_t i = 0; i < m->tab.size(); ++i) m->tab_index[m->tab[i].conv] = (int)i;
  icr_layout(m);
  *out = m;
  return MHIP_OK;
}

extern "C" int mhip_icr_destroy(mhip_icr* m) {
  if (!m) return MHIP_OK;
  mhip_quiesce(m->ctx);
  if (m->arena) (void)hipFree(m->arena);
  delete m;
  return MHIP_OK;
}

extern "C" int mhip_icr_set_tensor(mhip_icr* m, const char* key, const float* data, const int64_t* shape, int ndim) {
  if (!m || !key) return MHIP_EINVAL;
  std::string k(key);
  if (k.rfind("module.", 0) == 0) k = k.substr(7);
  if (k.size() > 19 && k.compare(k.size() - 19, 19, "num_batches_tracked") == 0) return MHIP_OK;
  bool known = k.rfind("Transformation.", 0) == 0 || k.rfind("FeatureExtraction.ConvNet.", 0) == 0 ||
               k.rfind("SequenceModeling.", 0) == 0 || k.rfind("Prediction.", 0) == 0;
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

extern "C" int mhip_icr_alloc_arena(mhip_icr* m) {
  if (!m) return MHIP_EINVAL;
  if (!m->arena && hipMalloc((void**)&m->arena, m->arena_bytes) != hipSuccess) {
    (void)hipGetLastError();
    return mhip_fail(m->ctx, MHIP_ENOMEM, "arena allocation of %zu bytes failed", m->arena_bytes);
  }
  m->ready = true;
  return MHIP_OK;
}

extern "C" int mhip_icr_arena(mhip_icr* m, void** dev, size_t* bytes) {
  if (!m) return MHIP_EINVAL;
  if (dev) *dev = m->arena;
  if (bytes) *bytes = m->arena_bytes;
  return MHIP_OK;
}

extern "C" int mhip_icr_finalize(mhip_icr* m) {
  if (!m) return MHIP_EINVAL;
  mhip_ctx* ctx = m->ctx;
  const size_t es = m->esz();
  const int C = m->num_class;
  std::vector<char> host(m->arena_bytes, 0);
  char* h = host.data();
  for (size_t i = 0; i < m->tab.size(); ++i) {
    const CB& c = m->tab[i];
    const HostTensor* w = ifind(m, c.conv + ".weight", {c.co, c.ci, c.k, c.k});
    if (!w) return MHIP_ESTATE;
    const int taps = c.k * c.k;
    if (c.ci == 1) {   // [Co][1][3][3] -> [9][Co] fp32 for the VALU first-layer kernel
      float* dw = (float*)(h + m->w_off[i]);
      for (int o = 0; o < c.co; ++o)
        for (int t = 0; t < 9; ++t) dw[t * c.co + o] = w->data[(size_t)o * 9 + t];
    } else {
      std::vector<float> tmp((size_t)c.cop * taps * c.cip, 0.f);
      for (int o = 0; o < c.co; ++o)
        for (int ci = 0; ci < c.ci; ++ci)
          for (int t = 0; t < taps; ++t) tmp[((size_t)o * taps + t) * c.cip + ci] = w->data[((size_t)o * c.ci + ci) * taps + t];
      putT(m, h + m->w_off[i], tmp.data(), tmp.size());
    }
    const HostTensor* g = ifind(m, c.bn + ".weight", {c.co});
    const HostTensor* be = ifind(m, c.bn + ".bias", {c.co});
    const HostTensor* mu = ifind(m, c.bn + ".running_mean", {c.co});
    const HostTensor* va = ifind(m, c.bn + ".running_var", {c.co});
    if (!g || !be || !mu || !va) return MHIP_ESTATE;
    float* sc = (float*)(h + m->s_off[i]);
    float* bi = (float*)(h + m->b_off[i]);
    const int np = std::max(c.cop, 64);
    for (int o = 0; o < np; ++o) {
      sc[o] = 1.f;
      bi[o] = 0.f;
    }
    for (int o = 0; o < c.co; ++o) {
      const float s = g->data[o] / sqrtf(va->data[o] + 1e-5f);
      sc[o] = s;
      bi[o] = be->data[o] - mu->data[o] * s;
    }
  }
  const std::string loc = "Transformation.LocalizationNetwork.";
  {
    const HostTensor* w1 = ifind(m, loc + "localization_fc1.0.weight", {256, 512});
    const HostTensor* b1 = ifind(m, loc + "localization_fc1.0.bias", {256});
    const HostTensor* w2 = ifind(m, loc + "localization_fc2.weight", {2 * NFID, 256});
    const HostTensor* b2 = ifind(m, loc + "localization_fc2.bias", {2 * NFID});
    const HostTensor* idc = ifind(m, "Transformation.GridGenerator.inv_delta_C", {NFID + 3, NFID + 3});
    const HostTensor* ph = ifind(m, "Transformation.GridGenerator.P_hat", {IMG_H * IMG_W, NFID + 3});
    if (!w1 || !b1 || !w2 || !b2 || !idc || !ph) return MHIP_ESTATE;
    putT(m, h + m->fc1_w, w1->data.data(), w1->data.size());
    memcpy(h + m->fc1_b, b1->data.data(), 256 * 4);
    putT(m, h + m->fc2_w, w2->data.data(), w2->data.size());
    memcpy(h + m->fc2_b, b2->data.data(), 2 * NFID * 4);
    memcpy(h + m->idc, idc->data.data(), idc->data.size() * 4);
    memcpy(h + m->phat, ph->data.data(), ph->data.size() * 4);
  }
  for (int j = 0; j < 2; ++j) {
    const int in = j == 0 ? 512 : 256;
    const std::string p = "SequenceModeling." + std::to_string(j) + ".";
    const HostTensor* wih[2] = {ifind(m, p + "rnn.weight_ih_l0", {1024, in}), ifind(m, p + "rnn.weight_ih_l0_reverse", {1024, in})};
    const HostTensor* whh[2] = {ifind(m, p + "rnn.weight_hh_l0", {1024, 256}), ifind(m, p + "rnn.weight_hh_l0_reverse", {1024, 256})};
    const HostTensor* bih[2] = {ifind(m, p + "rnn.bias_ih_l0", {1024}), ifind(m, p + "rnn.bias_ih_l0_reverse", {1024})};
    const HostTensor* bhh[2] = {ifind(m, p + "rnn.bias_hh_l0", {1024}), ifind(m, p + "rnn.bias_hh_l0_reverse", {1024})};
    const HostTensor* lw = ifind(m, p + "linear.weight", {256, 512});
    const HostTensor* lb = ifind(m, p + "linear.bias", {256});
    for (int d = 0; d < 2; ++d)
      if (!wih[d] || !whh[d] || !bih[d] || !bhh[d]) return MHIP_ESTATE;
    if (!lw || !lb) return MHIP_ESTATE;
    for (int d = 0; d < 2; ++d) {
      float* bb = (float*)(h + m->ih_b[j]) + d * 1024;
      for (int col = 0; col < 1024; ++col) {
        const int n = mhip_lstm_xproj_row(col);
        putT(m, h + m->ih_w[j] + ((size_t)d * 1024 + col) * in * es, wih[d]->data.data() + (size_t)n * in, (size_t)in);
        bb[col] = bih[d]->data[n] + bhh[d]->data[n];
      }
    }
    mhip_lstm_pack_whh(m->precision, whh[0]->data.data(), whh[1]->data.data(), h + m->hh_pack[j]);
    putT(m, h + m->lin_w[j], lw->data.data(), (size_t)256 * 512);
    memcpy(h + m->lin_b[j], lb->data.data(), 256 * 4);
  }
  {
    const std::string a = "Prediction.attention_cell.";
    const HostTensor* i2h = ifind(m, a + "i2h.weight", {256, 256});
    const HostTensor* h2h = ifind(m, a + "h2h.weight", {256, 256});
    const HostTensor* h2hb = ifind(m, a + "h2h.bias", {256});
    const HostTensor* sw = ifind(m, a + "score.weight", {1, 256});
    const HostTensor* wih = ifind(m, a + "rnn.weight_ih", {1024, 256 + C});
    const HostTensor* whh = ifind(m, a + "rnn.weight_hh", {1024, 256});
    const HostTensor* bih = ifind(m, a + "rnn.bias_ih", {1024});
    const HostTensor* bhh = ifind(m, a + "rnn.bias_hh", {1024});
    const HostTensor* gw = ifind(m, "Prediction.generator.weight", {C, 256});
    const HostTensor* gb = ifind(m, "Prediction.generator.bias", {C});
    if (!i2h || !h2h || !h2hb || !sw || !wih || !whh || !bih || !bhh || !gw || !gb) return MHIP_ESTATE;
    putT(m, h + m->i2h_w, i2h->data.data(), (size_t)256 * 256);
    // one GEMM per step over h: rows 0..255 = h2h, rows 256..1279 = W_hh (gate order i,f,g,o kept)
    putT(m, h + m->hg_w, h2h->data.data(), (size_t)256 * 256);
    putT(m, h + m->hg_w + (size_t)256 * 256 * es, whh->data.data(), (size_t)1024 * 256);
    float* hb = (float*)(h + m->hg_b);
    for (int i = 0; i < 256; ++i) hb[i] = h2hb->data[i];
    for (int i = 0; i < 1024; ++i) hb[256 + i] = bih->data[i] + bhh->data[i];
    memcpy(h + m->score_w, sw->data.data(), 256 * 4);
    std::vector<float> ctxw((size_t)1024 * 256);
    float* oh = (float*)(h + m->onehot_w);
    for (int r = 0; r < 1024; ++r) {
      for (int k = 0; k < 256; ++k) ctxw[(size_t)r * 256 + k] = wih->data[(size_t)r * (256 + C) + k];
      for (int ch = 0; ch < C; ++ch) oh[(size_t)ch * 1024 + r] = wih->data[(size_t)r * (256 + C) + 256 + ch];
    }
    putT(m, h + m->ihc_w, ctxw.data(), ctxw.size());
    putT(m, h + m->gen_w, gw->data.data(), (size_t)C * 256);
    memcpy(h + m->gen_b, gb->data.data(), (size_t)C * 4);
  }
  int rc = mhip_icr_alloc_arena(m);
  if (rc) return rc;
  m->ready = false;
  MHIP_HIP(ctx, hipMemcpy(m->arena, h, m->arena_bytes, hipMemcpyHostToDevice));
  m->ready = true;
  m->tensors.clear();
  return MHIP_OK;
}

extern "C" int mhip_icr_steps(void) { return STEPS; }

// Forward of n crops (uint8 [n][32][100], device).  Outputs (device): logits fp32 [n][49][C]; argmax int32 [n][49];
// pmax fp32 [n][49] (softmax value at the arg-max); rectified fp32 [n][32][100] or NULL.
extern "C" int mhip_icr_forward(mhip_icr* m, const uint8_t* crops, int n, float* logits, int32_t* argmax,
                                float* pmax, float* rectified_out) {
  if (!m) return MHIP_EINVAL;
  mhip_ctx* ctx = m->ctx;
  if (!m->ready || !m->arena) return mhip_fail(ctx, MHIP_ESTATE, "icr: weights not finalized");
  if (!crops || !logits || !argmax || !pmax || n < 1 || n > 65535)
    return mhip_fail(ctx, MHIP_EINVAL, "icr: bad arguments (n=%d)", n);
  MHIP_HIP(ctx, hipSetDevice(ctx->device));
  const size_t es = m->esz();
  const int prec = m->precision, C = m->num_class;
  const int H = IMG_H, W = IMG_W;
  // ---- workspace: two ping-pong activation buffers + named scratch -------------------------------------
  size_t o = 0;
  auto take = [&](size_t bytes) {
    size_t at = o;
    o = (o + bytes + 4095) / 4096 * 4096;
    return at;
  };
  const size_t big = (size_t)n * H * W * 64 * es;          // largest activation ([n][32][100][64])
  const size_t bufA = take(big), bufB = take(big), bufC = take(big / 2), bufD = take(big / 2);
  const int T = 26;
  const size_t o_cprime = take((size_t)n * 2 * NFID * 4), o_rect = take((size_t)n * H * W * 4);
  const size_t o_xproj = take((size_t)n * T * 2048 * 4), o_hseq = take((size_t)n * T * 512 * es);
  const size_t o_lin0 = take((size_t)n * T * 256 * es), o_lin1 = take((size_t)n * T * 256 * es);
  const size_t o_hproj = take((size_t)n * T * 256 * 4), o_hg = take((size_t)n * 1280 * 4);
  const size_t o_ctx = take((size_t)n * 256 * es), o_gctx = take((size_t)n * 1024 * 4);
  const size_t o_h = take((size_t)n * 256 * es), o_c = take((size_t)n * 256 * 4), o_chars = take((size_t)n * 4);
  int rc = mhip_ensure_workspace(ctx, o);
  if (rc) return rc;
  char* ws = (char*)ctx->ws;
  const char* A = m->arena;
#define CK(x) do { rc = (x); if (rc) return rc; } while (0)
  auto conv = [&](const std::string& key, const void* in, int hh, int ww, void* out, int pool, int relu,
                  const void* res = nullptr, int sy = 1, int pad_y = -1, int pad_x = -1) {
    const int i = m->tab_index.at(key);
    const CB& c = m->tab[i];
    ConvDesc d;
    d.in = in; d.w = A + m->w_off[i]; d.scale = (const float*)(A + m->s_off[i]); d.bias = (const float*)(A + m->b_off[i]);
    d.out = out; d.B = n; d.H = hh; d.W = ww; d.Cin = c.cip; d.KH = c.k; d.KW = c.k;
    d.pad = pad_y >= 0 ? pad_y : (c.k == 3 ? 1 : 0);
    d.pad_x = pad_x >= 0 ? pad_x : d.pad;
    d.sy = sy; d.N = c.cop; d.pool = pool; d.relu = relu; d.res = res;
    return mhip_launch_conv_igemm(ctx, prec, d);
  };
  auto gemm = [&](const void* in, int rows, int K, size_t w_off, size_t b_off, int N, void* out, int relu, int out_f32,
                  int ldc = 0) {
    ConvDesc d;
    d.in = in; d.w = A + w_off; d.bias = b_off ? (const float*)(A + b_off) : nullptr; d.out = out;
    d.B = rows; d.H = 1; d.W = 1; d.Cin = K; d.N = N; d.relu = relu; d.out_f32 = out_f32; d.ldc = ldc;
    return mhip_launch_conv_igemm(ctx, prec, d);
  };
  auto gray = [&](const std::string& key, int in_is_u8, const void* img, void* out) {
    const int i = m->tab_index.at(key);
    const CB& c = m->tab[i];
    return mhip_launch_conv_gray_first(ctx, prec, in_is_u8, img, n, H, W, c.co, (const float*)(A + m->w_off[i]),
                                       (const float*)(A + m->s_off[i]), (const float*)(A + m->b_off[i]), out);
  };
  void *a = ws + bufA, *b = ws + bufB, *c2 = ws + bufC, *d2 = ws + bufD;

  // ---- TPS: localization network -> fiducials -> rectified crop ------------------------------------------
  const std::string loc = "Transformation.LocalizationNetwork.conv.";
  CK(gray(loc + "0", 1, crops, a));                                            // [n][32][100][64]
  CK(mhip_launch_maxpool(ctx, prec, 2, a, b, n, 32, 100, 64));                 // [n][16][50][64]
  CK(conv(loc + "4", b, 16, 50, a, POOL_2x2, 1));                              // [n][8][25][128]
  CK(conv(loc + "8", a, 8, 25, b, POOL_2x2, 1));                               // [n][4][12][256]
  CK(conv(loc + "12", b, 4, 12, a, POOL_NONE, 1));                             // [n][4][12][512]
  CK(mhip_launch_avgpool_hw(ctx, prec, a, b, n, 48, 512));                     // [n][512]
  CK(gemm(b, n, 512, m->fc1_w, m->fc1_b, 256, a, 1, 0));                       // fc1 + ReLU
  CK(gemm(a, n, 256, m->fc2_w, m->fc2_b, 2 * NFID, ws + o_cprime, 0, 1));      // C' fp32 [n][40]
  float* rect = rectified_out ? rectified_out : (float*)(ws + o_rect);
  CK(mhip_launch_tps_sample(ctx, crops, (const float*)(ws + o_cprime), (const float*)(A + m->idc),
                            (const float*)(A + m->phat), rect, n, H, W, NFID));

  // ---- ResNet-45 -------------------------------------------------------------------------------------------
  const std::string r = "FeatureExtraction.ConvNet.";
  CK(gray(r + "conv0_1", 0, rect, a));                                         // [n][32][100][32 -> 64 padded]
  CK(conv(r + "conv0_2", a, 32, 100, b, POOL_2x2, 1));                         // + maxpool1 -> [n][16][50][64]
  auto block = [&](const std::string& p, void* x, int hh, int ww, bool ds, void* t1, void* t2, void* y) -> int {
    int e = conv(p + "conv1", x, hh, ww, t1, POOL_NONE, 1);
    if (e) return e;
    const void* res = x;
    if (ds) {
      e = conv(p + "downsample.0", x, hh, ww, t2, POOL_NONE, 0);
      if (e) return e;
      res = t2;
    }
    return conv(p + "conv2", t1, hh, ww, y, POOL_NONE, 1, res);
  };
  CK(block(r + "layer1.0.", b, 16, 50, true, a, c2, d2));                      // -> d2 [n][16][50][128]
  CK(conv(r + "conv1", d2, 16, 50, a, POOL_2x2, 1));                           // + maxpool2 -> [n][8][25][128]
  CK(block(r + "layer2.0.", a, 8, 25, true, b, c2, d2));                       // -> d2 [n][8][25][256]
  CK(block(r + "layer2.1.", d2, 8, 25, false, b, c2, a));                      // -> a
  CK(conv(r + "conv2", a, 8, 25, b, POOL_NONE, 1));                            // [n][8][25][256]
  CK(mhip_launch_maxpool_s21_p01(ctx, prec, b, a, n, 8, 25, 256));             // maxpool3 -> [n][4][26][256]
  CK(block(r + "layer3.0.", a, 4, 26, true, b, c2, d2));                       // -> d2 [n][4][26][512]
  void* cur = d2;
  void* nxt = a;
  for (int i = 1; i < 5; ++i) {
    CK(block(r + "layer3." + std::to_string(i) + ".", cur, 4, 26, false, b, c2, nxt));
    std::swap(cur, nxt);
  }
  CK(conv(r + "conv3", cur, 4, 26, nxt, POOL_NONE, 1));
  std::swap(cur, nxt);
  for (int i = 0; i < 3; ++i) {
    CK(block(r + "layer4." + std::to_string(i) + ".", cur, 4, 26, false, b, c2, nxt));
    std::swap(cur, nxt);
  }
  CK(conv(r + "conv4_1", cur, 4, 26, nxt, POOL_NONE, 1, nullptr, 2, 0, 1));    // stride (2,1), pad (0,1) -> [n][2][27][512]
  CK(conv(r + "conv4_2", nxt, 2, 27, cur, POOL_NONE, 1, nullptr, 1, 0, 0));    // -> [n][1][26][512] = sequence

  // ---- BiLSTM x 2 (AdaptiveAvgPool over H is the identity: H = 1) ----------------------------------------------
  const void* seq_in = cur;
  int seq_ch = 512;
  const size_t lin_off[2] = {o_lin0, o_lin1};
  for (int j = 0; j < 2; ++j) {
    CK(gemm(seq_in, n * T, seq_ch, m->ih_w[j], m->ih_b[j], 2048, ws + o_xproj, 0, 1));
    CK(mhip_launch_lstm_rec(ctx, prec, (const float*)(ws + o_xproj), A + m->hh_pack[j], ws + o_hseq, n, T));
    CK(gemm(ws + o_hseq, n * T, 512, m->lin_w[j], m->lin_b[j], 256, ws + lin_off[j], 0, 0));
    seq_in = ws + lin_off[j];
    seq_ch = 256;
  }
  const void* batch_h = ws + o_lin1;

  // ---- attention decoder: 49 greedy steps ---------------------------------------------------------------------
  CK(gemm(batch_h, n * T, 256, m->i2h_w, 0, 256, ws + o_hproj, 0, 1));         // i2h(batch_H), loop-invariant
  MHIP_HIP(ctx, hipMemsetAsync(ws + o_h, 0, (size_t)n * 256 * es, ctx->stream));
  MHIP_HIP(ctx, hipMemsetAsync(ws + o_c, 0, (size_t)n * 256 * 4, ctx->stream));
  MHIP_HIP(ctx, hipMemsetAsync(ws + o_chars, 0, (size_t)n * 4, ctx->stream));  // [GO] = 0
  for (int s = 0; s < STEPS; ++s) {
    CK(gemm(ws + o_h, n, 256, m->hg_w, m->hg_b, 1280, ws + o_hg, 0, 1));       // [h2h(h)+b | W_hh h + b_ih + b_hh]
    CK(mhip_launch_attn_context(ctx, prec, (const float*)(ws + o_hproj), (const float*)(ws + o_hg), 1280,
                                (const float*)(A + m->score_w), batch_h, ws + o_ctx, n, T));
    CK(gemm(ws + o_ctx, n, 256, m->ihc_w, 0, 1024, ws + o_gctx, 0, 1));        // W_ih[:, :256] context
    CK(mhip_launch_attn_cell(ctx, prec, (const float*)(ws + o_gctx), (const float*)(ws + o_hg), 1280,
                             (const float*)(A + m->onehot_w), (const int*)(ws + o_chars), (float*)(ws + o_c),
                             ws + o_h, n));
    CK(gemm(ws + o_h, n, 256, m->gen_w, m->gen_b, C, logits + (size_t)s * C, 0, 1, STEPS * C));   // probs[:, s, :]
    CK(mhip_launch_argmax_rows(ctx, logits + (size_t)s * C, STEPS * C, C, (int*)(ws + o_chars), n));
  }
  CK(mhip_launch_rowmax_softmax(ctx, logits, n * STEPS, C, argmax, pmax));
#undef CK
  return MHIP_OK;
}

extern "C" int mhip_icr_forward_host(mhip_icr* m, const uint8_t* crops_h, int n, float* logits_h, int32_t* argmax_h,
                                     float* pmax_h, float* rectified_h) {
  if (!m || !crops_h || !argmax_h || !pmax_h || n < 1) return MHIP_EINVAL;
  mhip_ctx* ctx = m->ctx;
  MHIP_HIP(ctx, hipSetDevice(ctx->device));
  const int C = m->num_class;
  const size_t in_b = (size_t)n * IMG_H * IMG_W, lg_b = (size_t)n * STEPS * C * 4, it_b = (size_t)n * STEPS * 4,
               rc_b = (size_t)n * IMG_H * IMG_W * 4;
  // I/O staging in its own allocation so the forward's workspace planning stays independent
  char* io = nullptr;
  const size_t total = in_b + lg_b + 2 * it_b + rc_b + 4096 * 5;
  if (hipMalloc((void**)&io, total) != hipSuccess) {
    (void)hipGetLastError();
    return mhip_fail(ctx, MHIP_ENOMEM, "icr: I/O staging of %zu bytes failed", total);
  }
  auto al = [](size_t v) { return (v + 4095) / 4096 * 4096; };
  char* d_in = io;
  char* d_lg = d_in + al(in_b);
  char* d_am = d_lg + al(lg_b);
  char* d_pm = d_am + al(it_b);
  char* d_rc = d_pm + al(it_b);
  int rc = MHIP_OK;
  hipError_t e = hipMemcpyAsync(d_in, crops_h, in_b, hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) {
    rc = mhip_icr_forward(m, (const uint8_t*)d_in, n, (float*)d_lg, (int32_t*)d_am, (float*)d_pm,
                          rectified_h ? (float*)d_rc : nullptr);
    if (rc == MHIP_OK) {
      if (logits_h) e = hipMemcpyAsync(logits_h, d_lg, lg_b, hipMemcpyDeviceToHost, ctx->stream);
      if (e == hipSuccess) e = hipMemcpyAsync(argmax_h, d_am, it_b, hipMemcpyDeviceToHost, ctx->stream);
      if (e == hipSuccess) e = hipMemcpyAsync(pmax_h, d_pm, it_b, hipMemcpyDeviceToHost, ctx->stream);
      if (e == hipSuccess && rectified_h) e = hipMemcpyAsync(rectified_h, d_rc, rc_b, hipMemcpyDeviceToHost, ctx->stream);
    }
  }
  hipError_t e2 = hipStreamSynchronize(ctx->stream);
  (void)hipFree(io);
  if (rc) return rc;
  if (e != hipSuccess || e2 != hipSuccess)
    return mhip_fail(ctx, MHIP_EHIP, "icr forward_host: %s", hipGetErrorString(e != hipSuccess ? e : e2));
  return MHIP_OK;
}
