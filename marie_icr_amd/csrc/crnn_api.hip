// crnn_api.hip — context management and the CRNN-family recognizer (None-VGG-BiLSTM-CTC) behind
// the C ABI of include/marie_hip.h.  Host-side counterpart of Model(opt) in
// marie/models/icr/model.py:25-92 and of CraftOcrProcessor's forward/decode loop in
// marie/document/craft_ocr_processor.py:184-286.
#include <math.h>
#include <stdarg.h>

#include <map>
#include <memory>

#include "common.h"

// ======================================================================= context
int mhip_fail(mhip_ctx* ctx, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  if (ctx) ctx->err = buf;
  return code;
}

static const char* kKernelNames[MHIP_K_COUNT] = {"conv_first", "conv_igemm", "lstm_rec",   "ctc_decode",
                                                 "image_ops",  "ccl",        "crop_batch", "attn",
                                                 "attn_flash", "vit_ops",    "det_ops",    "dec_ops",
                                                 "conv_igemm<64>", "conv_igemm<128>", "conv_igemm<256>", "conv_igemm<1128>",
                                                 "conv3x3_patch", "cross_attn"};

extern "C" int mhip_kernel_count(void) { return MHIP_K_COUNT; }
extern "C" const char* mhip_kernel_name(int k) { return (k >= 0 && k < MHIP_K_COUNT) ? kKernelNames[k] : ""; }

extern "C" int mhip_init(int device_id, mhip_ctx** out) {
  if (!out) return MHIP_EINVAL;
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return MHIP_EHIP;
  if (device_id < 0 || device_id >= ndev) return MHIP_EINVAL;
  if (hipSetDevice(device_id) != hipSuccess) return MHIP_EHIP;
  mhip_ctx* ctx = new mhip_ctx();
  ctx->device = device_id;
  for (int k = MHIP_K_IGEMM_T64; k <= MHIP_K_IGEMM_PATCH; ++k) ctx->prof[k].parent = MHIP_K_CONV_IGEMM;
  if (hipMalloc(&ctx->zeros, MHIP_ZERO_BYTES) != hipSuccess || hipMemset(ctx->zeros, 0, MHIP_ZERO_BYTES) != hipSuccess) {
    delete ctx;
    return MHIP_ENOMEM;
  }
  *out = ctx;
  return MHIP_OK;
}

extern "C" int mhip_destroy(mhip_ctx* ctx) {
  if (!ctx) return MHIP_OK;
  (void)hipSetDevice(ctx->device);
  mhip_quiesce(ctx);
  for (auto& s : ctx->prof)
    for (auto& p : s.pending) {
      (void)hipEventDestroy(p.first);
      (void)hipEventDestroy(p.second);
    }
  for (auto e : ctx->event_pool) (void)hipEventDestroy(e);
  for (int i = 0; i < mhip_ctx::PinnedRing::N; ++i) {
    if (ctx->stage.ev[i]) (void)hipEventDestroy(ctx->stage.ev[i]);
    if (ctx->stage.buf[i]) (void)hipHostFree(ctx->stage.buf[i]);
  }
  if (ctx->ws) (void)hipFree(ctx->ws);
  if (ctx->zeros) (void)hipFree(ctx->zeros);
  delete ctx;
  return MHIP_OK;
}

int mhip_stage_h2d(mhip_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes) {
  if (!bytes) return MHIP_OK;
  mhip_ctx::PinnedRing& r = ctx->stage;
  const int i = r.next;
  r.next = (i + 1) % mhip_ctx::PinnedRing::N;
  if (r.ev[i]) MHIP_HIP(ctx, hipEventSynchronize(r.ev[i]));          // the copy that last read this buffer (four calls ago) is done
  else MHIP_HIP(ctx, hipEventCreateWithFlags(&r.ev[i], hipEventDisableTiming));
  if (bytes > r.cap[i]) {
    if (r.buf[i]) (void)hipHostFree(r.buf[i]);
    r.buf[i] = nullptr; r.cap[i] = 0;
    const size_t cap = std::max<size_t>(bytes, 64 * 1024);
    MHIP_HIP(ctx, hipHostMalloc(&r.buf[i], cap, hipHostMallocDefault));
    r.cap[i] = cap;
  }
  memcpy(r.buf[i], src_host, bytes);
  MHIP_HIP(ctx, hipMemcpyAsync(dst_dev, r.buf[i], bytes, hipMemcpyHostToDevice, ctx->stream));
  MHIP_HIP(ctx, hipEventRecord(r.ev[i], ctx->stream));
  return MHIP_OK;
}

extern "C" const char* mhip_last_error(mhip_ctx* ctx) { return ctx ? ctx->err.c_str() : "null ctx"; }

extern "C" int mhip_set_stream(mhip_ctx* ctx, void* s) {
  if (!ctx) return MHIP_EINVAL;
  ctx->stream = (hipStream_t)s;
  return MHIP_OK;
}

extern "C" int mhip_synchronize(mhip_ctx* ctx) {
  if (!ctx) return MHIP_EINVAL;
  MHIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return MHIP_OK;
}

extern "C" int mhip_device_info(mhip_ctx* ctx, char* arch, size_t arch_len, int* cu_count, size_t* hbm_bytes) {
  if (!ctx) return MHIP_EINVAL;
  hipDeviceProp_t p;
  MHIP_HIP(ctx, hipGetDeviceProperties(&p, ctx->device));
  if (arch && arch_len) {
    strncpy(arch, p.gcnArchName, arch_len - 1);
    arch[arch_len - 1] = 0;
  }
  if (cu_count) *cu_count = p.multiProcessorCount;
  if (hbm_bytes) *hbm_bytes = p.totalGlobalMem;
  return MHIP_OK;
}

extern "C" int mhip_memcpy_dev(mhip_ctx* ctx, void* dst, const void* src, size_t bytes) {
  if (!ctx || (bytes && (!dst || !src))) return MHIP_EINVAL;
  MHIP_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
  return MHIP_OK;
}

int mhip_ensure_workspace(mhip_ctx* ctx, size_t bytes) {
  if (bytes <= ctx->ws_bytes) return MHIP_OK;
  MHIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (ctx->ws) MHIP_HIP(ctx, hipFree(ctx->ws));
  ctx->ws = nullptr;
  ctx->ws_bytes = 0;
  size_t want = bytes + bytes / 8;
  if (hipMalloc(&ctx->ws, want) != hipSuccess) {
    (void)hipGetLastError();
    return mhip_fail(ctx, MHIP_ENOMEM, "workspace allocation of %zu bytes failed", want);
  }
  ctx->ws_bytes = want;
  return MHIP_OK;
}

// ------------------------------------------------------------------ profiling
static hipEvent_t get_event(mhip_ctx* ctx) {
  if (!ctx->event_pool.empty()) {
    hipEvent_t e = ctx->event_pool.back();
    ctx->event_pool.pop_back();
    return e;
  }
  hipEvent_t e = nullptr;
  (void)hipEventCreate(&e);
  return e;
}
void mhip_prof_begin(mhip_ctx* ctx, int kid, hipEvent_t* e0) {
  (void)kid;
  *e0 = get_event(ctx);
  (void)hipEventRecord(*e0, ctx->stream);
}
void mhip_prof_end(mhip_ctx* ctx, int kid, hipEvent_t e0) {
  hipEvent_t e1 = get_event(ctx);
  (void)hipEventRecord(e1, ctx->stream);
  ctx->prof[kid].pending.emplace_back(e0, e1);
}
static void prof_drain(mhip_ctx* ctx) {
  (void)hipStreamSynchronize(ctx->stream);
  for (auto& s : ctx->prof) {
    for (auto& p : s.pending) {
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, p.first, p.second) == hipSuccess) {
        s.total_ms += ms;
        s.launches += 1;
        if (s.parent >= 0) {
          ctx->prof[s.parent].total_ms += ms;
          ctx->prof[s.parent].launches += 1;
        }
      }
      ctx->event_pool.push_back(p.first);
      ctx->event_pool.push_back(p.second);
    }
    s.pending.clear();
  }
}
extern "C" int mhip_profile_enable(mhip_ctx* ctx, int enable) {
  if (!ctx) return MHIP_EINVAL;
  if (!enable) prof_drain(ctx);
  ctx->profiling = enable != 0;
  return MHIP_OK;
}
extern "C" int mhip_profile_reset(mhip_ctx* ctx) {
  if (!ctx) return MHIP_EINVAL;
  prof_drain(ctx);
  for (auto& s : ctx->prof) {
    s.total_ms = 0;
    s.launches = 0;
    s.flops = 0;
  }
  return MHIP_OK;
}
extern "C" int mhip_profile_flops(mhip_ctx* ctx, int kid, double* flops) {
  if (!ctx || kid < 0 || kid >= MHIP_K_COUNT || !flops) return MHIP_EINVAL;
  *flops = ctx->prof[kid].flops;
  return MHIP_OK;
}
extern "C" int mhip_profile_read(mhip_ctx* ctx, int kid, double* total_ms, int64_t* launches) {
  if (!ctx || kid < 0 || kid >= MHIP_K_COUNT) return MHIP_EINVAL;
  prof_drain(ctx);
  if (total_ms) *total_ms = ctx->prof[kid].total_ms;
  if (launches) *launches = ctx->prof[kid].launches;
  return MHIP_OK;
}

// ======================================================================= conv primitive
extern "C" int mhip_conv2d_nhwc(mhip_ctx* ctx, int precision, const mhip_conv_desc* d, const void* in,
                                const void* in2, const void* w, const float* scale, const float* bias, void* out) {
  if (!ctx || !d) return MHIP_EINVAL;
  if (precision != MHIP_PREC_F16 && precision != MHIP_PREC_F32)
    return mhip_fail(ctx, MHIP_EINVAL, "unknown precision %d", precision);
  if (d->B < 1 || d->H < 1 || d->W < 1 || d->N < 1 || d->KH < 1 || d->KW < 1 || d->pad < 0 || d->pool < 0 ||
      d->pool > 2)
    return mhip_fail(ctx, MHIP_EINVAL, "conv2d: bad descriptor");
  MHIP_HIP(ctx, hipSetDevice(ctx->device));
  ConvDesc c;
  c.in = in; c.w = w; c.scale = scale; c.bias = bias; c.out = out;
  c.B = d->B; c.H = d->H; c.W = d->W; c.Cin = d->Cin;
  c.KH = d->KH; c.KW = d->KW; c.pad = d->pad;
  c.N = d->N; c.pool = d->pool; c.relu = d->relu; c.out_f32 = d->out_f32;
  c.dil = d->dil > 0 ? d->dil : 1;
  c.in2 = in2; c.Cin1 = d->Cin1;
  c.ldc = d->ldc; c.pad_cols_writable = d->pad_cols_writable;
  return mhip_launch_conv_igemm(ctx, precision, c);
}

// ======================================================================= CRNN model
namespace {

struct HostTensor {
  std::vector<int64_t> shape;
  std::vector<float> data;
};

struct ConvSpec {
  const char* key;
  int co, ci, kh, kw;
  bool has_bias;
  const char* bn;  // BatchNorm key prefix or nullptr
};
// reference: marie/models/icr/modules/feature_extraction.py:13-25
const ConvSpec kConvs[7] = {
    {"FeatureExtraction.ConvNet.0", 64, 1, 3, 3, true, nullptr},
    {"FeatureExtraction.ConvNet.3", 128, 64, 3, 3, true, nullptr},
    {"FeatureExtraction.ConvNet.6", 256, 128, 3, 3, true, nullptr},
    {"FeatureExtraction.ConvNet.8", 256, 256, 3, 3, true, nullptr},
    {"FeatureExtraction.ConvNet.11", 512, 256, 3, 3, false, "FeatureExtraction.ConvNet.12"},
    {"FeatureExtraction.ConvNet.14", 512, 512, 3, 3, false, "FeatureExtraction.ConvNet.15"},
    {"FeatureExtraction.ConvNet.18", 512, 512, 2, 2, true, nullptr},
};

struct Arena {
  // byte offsets into the device arena
  size_t conv0_w = 0, conv0_b = 0;
  size_t conv_w[7] = {0}, conv_scale[7] = {0}, conv_bias[7] = {0};
  size_t ih_w[2] = {0}, ih_b[2] = {0}, hh_pack[2] = {0}, lin_w[2] = {0}, lin_b[2] = {0};
  size_t pred_w = 0, pred_b = 0;
  size_t bytes = 0;
};

size_t align_up(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }

}  // namespace

struct mhip_crnn {
  mhip_ctx* ctx = nullptr;
  int precision = MHIP_PREC_F16;
  int num_class = 0;
  std::map<std::string, HostTensor> tensors;
  Arena lay;
  char* arena = nullptr;
  bool ready = false;
  size_t esz() const { return precision == MHIP_PREC_F16 ? 2 : 4; }
};

namespace {

void build_layout(mhip_crnn* m) {
  Arena& L = m->lay;
  const size_t es = m->esz();
  size_t o = 0;
  auto take = [&](size_t bytes) {
    size_t at = o;
    o = align_up(o + bytes);
    return at;
  };
  L.conv0_w = take(9 * 64 * 4);
  L.conv0_b = take(64 * 4);
  for (int i = 1; i < 7; ++i) {
    const ConvSpec& c = kConvs[i];
    L.conv_w[i] = take((size_t)c.co * c.ci * c.kh * c.kw * es);
    L.conv_scale[i] = take((size_t)c.co * 4);
    L.conv_bias[i] = take((size_t)c.co * 4);
  }
  for (int j = 0; j < 2; ++j) {
    int in = j == 0 ? 512 : 256;
    L.ih_w[j] = take((size_t)2048 * in * es);
    L.ih_b[j] = take(2048 * 4);
    L.hh_pack[j] = take(mhip_lstm_wpack_bytes(m->precision));
    L.lin_w[j] = take((size_t)256 * 512 * es);
    L.lin_b[j] = take(256 * 4);
  }
  L.pred_w = take((size_t)m->num_class * 256 * es);
  L.pred_b = take((size_t)m->num_class * 4);
  L.bytes = o;
}

void put(const mhip_crnn* m, char* dst, const float* src, size_t n) {
  if (m->precision == MHIP_PREC_F16) {
    _Float16* d = (_Float16*)dst;
    for (size_t i = 0; i < n; ++i) d[i] = (_Float16)src[i];
  } else {
    memcpy(dst, src, n * 4);
  }
}

const HostTensor* find(const mhip_crnn* m, const std::string& k, std::initializer_list<int64_t> shape) {
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

extern "C" int mhip_crnn_create(mhip_ctx* ctx, int precision, int num_class, mhip_crnn** out) {
  if (!ctx || !out) return MHIP_EINVAL;
  *out = nullptr;
  if (precision != MHIP_PREC_F16 && precision != MHIP_PREC_F32)
    return mhip_fail(ctx, MHIP_EINVAL, "unknown precision %d", precision);
  if (num_class < 2 || num_class > 256) return mhip_fail(ctx, MHIP_EINVAL, "num_class %d not in [2,256]", num_class);
  mhip_crnn* m = new mhip_crnn();
  m->ctx = ctx;
  m->precision = precision;
  m->num_class = num_class;
  build_layout(m);
  *out = m;
  return MHIP_OK;
}

extern "C" int mhip_crnn_destroy(mhip_crnn* m) {
  if (!m) return MHIP_OK;
  if (m->arena) {
    mhip_quiesce(m->ctx);
    (void)hipFree(m->arena);
  }
  delete m;
  return MHIP_OK;
}

extern "C" int mhip_crnn_set_tensor(mhip_crnn* m, const char* key, const float* data, const int64_t* shape,
                                    int ndim) {
  if (!m || !key) return MHIP_EINVAL;
  std::string k(key);
  if (k.rfind("module.", 0) == 0) k = k.substr(7);
  if (k.size() > 19 && k.compare(k.size() - 19, 19, "num_batches_tracked") == 0) return MHIP_OK;
  bool known = k.rfind("FeatureExtraction.ConvNet.", 0) == 0 || k.rfind("SequenceModeling.", 0) == 0 ||
               k.rfind("Prediction.", 0) == 0;
  if (!known) return mhip_fail(m->ctx, MHIP_EINVAL, "unknown state_dict key %s", key);
  if (!data || ndim < 0 || ndim > 4 || (ndim > 0 && !shape))
    return mhip_fail(m->ctx, MHIP_EINVAL, "bad tensor %s", key);
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

extern "C" int mhip_crnn_alloc_arena(mhip_crnn* m) {
  if (!m) return MHIP_EINVAL;
  if (!m->arena) {
    if (hipMalloc((void**)&m->arena, m->lay.bytes) != hipSuccess) {
      (void)hipGetLastError();
      return mhip_fail(m->ctx, MHIP_ENOMEM, "arena allocation of %zu bytes failed", m->lay.bytes);
    }
  }
  m->ready = true;  // contents are the caller's responsibility (RCCL broadcast)
  return MHIP_OK;
}

extern "C" int mhip_crnn_arena(mhip_crnn* m, void** dev, size_t* bytes) {
  if (!m) return MHIP_EINVAL;
  if (dev) *dev = m->arena;
  if (bytes) *bytes = m->lay.bytes;
  return MHIP_OK;
}

extern "C" int mhip_crnn_finalize(mhip_crnn* m) {
  if (!m) return MHIP_EINVAL;
  mhip_ctx* ctx = m->ctx;
  const Arena& L = m->lay;
  const size_t es = m->esz();
  std::vector<char> host(L.bytes, 0);
  char* h = host.data();

  // conv0: [64][1][3][3] -> tap-major [9][64] fp32
  {
    const HostTensor* w = find(m, std::string(kConvs[0].key) + ".weight", {64, 1, 3, 3});
    const HostTensor* b = find(m, std::string(kConvs[0].key) + ".bias", {64});
    if (!w || !b) return MHIP_ESTATE;
    float* dw = (float*)(h + L.conv0_w);
    for (int c = 0; c < 64; ++c)
      for (int k = 0; k < 9; ++k) dw[k * 64 + c] = w->data[c * 9 + k];
    memcpy(h + L.conv0_b, b->data.data(), 64 * 4);
  }
  // conv1..6: [Co][Ci][kh][kw] -> [Co][kh][kw][Ci]; BN folded to per-channel scale/shift (fp32 epilogue)
  for (int i = 1; i < 7; ++i) {
    const ConvSpec& c = kConvs[i];
    const HostTensor* w = find(m, std::string(c.key) + ".weight", {c.co, c.ci, c.kh, c.kw});
    if (!w) return MHIP_ESTATE;
    std::vector<float> tmp((size_t)c.co * c.ci * c.kh * c.kw);
    const int taps = c.kh * c.kw;
    for (int o = 0; o < c.co; ++o)
      for (int ci = 0; ci < c.ci; ++ci)
        for (int t = 0; t < taps; ++t)
          tmp[((size_t)o * taps + t) * c.ci + ci] = w->data[((size_t)o * c.ci + ci) * taps + t];
    put(m, h + L.conv_w[i], tmp.data(), tmp.size());
    float* sc = (float*)(h + L.conv_scale[i]);
    float* bi = (float*)(h + L.conv_bias[i]);
    for (int o = 0; o < c.co; ++o) {
      sc[o] = 1.f;
      bi[o] = 0.f;
    }
    if (c.has_bias) {
      const HostTensor* b = find(m, std::string(c.key) + ".bias", {c.co});
      if (!b) return MHIP_ESTATE;
      memcpy(bi, b->data.data(), (size_t)c.co * 4);
    }
    if (c.bn) {
      const HostTensor* g = find(m, std::string(c.bn) + ".weight", {c.co});
      const HostTensor* be = find(m, std::string(c.bn) + ".bias", {c.co});
      const HostTensor* mu = find(m, std::string(c.bn) + ".running_mean", {c.co});
      const HostTensor* va = find(m, std::string(c.bn) + ".running_var", {c.co});
      if (!g || !be || !mu || !va) return MHIP_ESTATE;
      for (int o = 0; o < c.co; ++o) {
        float s = g->data[o] / sqrtf(va->data[o] + 1e-5f);  // nn.BatchNorm2d eps
        sc[o] = s;
        bi[o] = be->data[o] + (bi[o] - mu->data[o]) * s;
      }
    }
  }
  // BiLSTM layers
  for (int j = 0; j < 2; ++j) {
    const int in = j == 0 ? 512 : 256;
    const std::string p = "SequenceModeling." + std::to_string(j) + ".";
    const HostTensor* wih[2] = {find(m, p + "rnn.weight_ih_l0", {1024, in}),
                                find(m, p + "rnn.weight_ih_l0_reverse", {1024, in})};
    const HostTensor* whh[2] = {find(m, p + "rnn.weight_hh_l0", {1024, 256}),
                                find(m, p + "rnn.weight_hh_l0_reverse", {1024, 256})};
    const HostTensor* bih[2] = {find(m, p + "rnn.bias_ih_l0", {1024}), find(m, p + "rnn.bias_ih_l0_reverse", {1024})};
    const HostTensor* bhh[2] = {find(m, p + "rnn.bias_hh_l0", {1024}), find(m, p + "rnn.bias_hh_l0_reverse", {1024})};
    const HostTensor* lw = find(m, p + "linear.weight", {256, 512});
    const HostTensor* lb = find(m, p + "linear.bias", {256});
    for (int d = 0; d < 2; ++d)
      if (!wih[d] || !whh[d] || !bih[d] || !bhh[d]) return MHIP_ESTATE;
    if (!lw || !lb) return MHIP_ESTATE;
    // rows permuted so the GEMM writes xproj gate-interleaved, the order lstm.hip consumes
    for (int d = 0; d < 2; ++d) {
      float* bb = (float*)(h + L.ih_b[j]) + d * 1024;
      for (int col = 0; col < 1024; ++col) {
        const int n = mhip_lstm_xproj_row(col);
        put(m, h + L.ih_w[j] + ((size_t)d * 1024 + col) * in * es, wih[d]->data.data() + (size_t)n * in, (size_t)in);
        bb[col] = bih[d]->data[n] + bhh[d]->data[n];
      }
    }
    mhip_lstm_pack_whh(m->precision, whh[0]->data.data(), whh[1]->data.data(), h + L.hh_pack[j]);
    put(m, h + L.lin_w[j], lw->data.data(), (size_t)256 * 512);
    memcpy(h + L.lin_b[j], lb->data.data(), 256 * 4);
  }
  {
    const HostTensor* w = find(m, "Prediction.weight", {m->num_class, 256});
    const HostTensor* b = find(m, "Prediction.bias", {m->num_class});
    if (!w || !b) return MHIP_ESTATE;
    put(m, h + L.pred_w, w->data.data(), (size_t)m->num_class * 256);
    memcpy(h + L.pred_b, b->data.data(), (size_t)m->num_class * 4);
  }
  int rc = mhip_crnn_alloc_arena(m);
  if (rc) return rc;
  m->ready = false;
  MHIP_HIP(ctx, hipMemcpy(m->arena, h, L.bytes, hipMemcpyHostToDevice));
  m->ready = true;
  m->tensors.clear();  // host copies are no longer needed
  return MHIP_OK;
}

extern "C" int mhip_crnn_seq_len(int w) { return w / 4 - 1; }

namespace {

struct Plan {
  size_t act[7];   // outputs of conv layers 0..6
  size_t xproj, hseq, lin[2], logits, total;
  int T;
};

Plan make_plan(const mhip_crnn* m, int n, int w) {
  Plan p;
  const size_t es = m->esz();
  const int w2 = w / 2, w4 = w / 4, T = w4 - 1;
  size_t o = 0;
  auto take = [&](size_t bytes) {
    size_t at = o;
    o = align_up(o + bytes, 4096);
    return at;
  };
  p.T = T;
  p.act[0] = take((size_t)n * 16 * w2 * 64 * es);
  p.act[1] = take((size_t)n * 8 * w4 * 128 * es);
  p.act[2] = take((size_t)n * 8 * w4 * 256 * es);
  p.act[3] = take((size_t)n * 4 * w4 * 256 * es);
  p.act[4] = take((size_t)n * 4 * w4 * 512 * es);
  p.act[5] = take((size_t)n * 2 * w4 * 512 * es);
  p.act[6] = take((size_t)n * T * 512 * es);
  p.xproj = take((size_t)n * T * 2048 * 4);
  p.hseq = take((size_t)n * T * 512 * es);
  p.lin[0] = take((size_t)n * T * 256 * es);
  p.lin[1] = take((size_t)n * T * 256 * es);
  p.logits = take((size_t)n * T * m->num_class * 4);
  p.total = o;
  return p;
}

int check_shape(mhip_crnn* m, int n, int w) {
  if (!m) return MHIP_EINVAL;
  if (n < 1 || w < 8 || (w % 4) != 0)
    return mhip_fail(m->ctx, MHIP_EINVAL, "crnn: need n >= 1 and w >= 8 with w %% 4 == 0 (got n=%d w=%d)", n, w);
  if ((long long)n * 16 * (w / 2) * 4 > 0x7ffffff0LL)
    return mhip_fail(m->ctx, MHIP_EINVAL, "crnn: batch of %d x %d exceeds one launch; split it", n, w);
  return MHIP_OK;
}

}  // namespace

extern "C" size_t mhip_crnn_workspace_bytes(mhip_crnn* m, int n, int w) {
  if (!m || n < 1 || w < 8) return 0;
  return make_plan(m, n, w).total;
}

extern "C" double mhip_crnn_kernel_flops(mhip_crnn* m, int kid, int n, int w) {
  if (!m || n < 1 || w < 8) return 0.0;
  const double T = w / 4 - 1, w2 = w / 2, w4 = w / 4;
  switch (kid) {
    case MHIP_K_CONV_FIRST:
      return 2.0 * n * 32 * w * 64 * 9;
    case MHIP_K_CONV_IGEMM: {
      double f = 0;
      f += 2.0 * n * 16 * w2 * 128 * (9 * 64);
      f += 2.0 * n * 8 * w4 * 256 * (9 * 128);
      f += 2.0 * n * 8 * w4 * 256 * (9 * 256);
      f += 2.0 * n * 4 * w4 * 512 * (9 * 256);
      f += 2.0 * n * 4 * w4 * 512 * (9 * 512);
      f += 2.0 * n * T * 512 * (4 * 512);
      f += 2.0 * n * T * 2048 * 512 + 2.0 * n * T * 256 * 512;   // BiLSTM-0 input projection + linear
      f += 2.0 * n * T * 2048 * 256 + 2.0 * n * T * 256 * 512;   // BiLSTM-1
      f += 2.0 * n * T * m->num_class * 256;                     // prediction
      return f;
    }
    case MHIP_K_LSTM_REC:
      return 2.0 * (2.0 * n * T * 2 * 1024 * 256);
    default:
      return 0.0;
  }
}

extern "C" int mhip_crnn_forward(mhip_crnn* m, const uint8_t* crops, int n, int w, float* logits_out,
                                 int32_t* argmax, int32_t* tokens, int32_t* lengths, float* conf) {
  int rc = check_shape(m, n, w);
  if (rc) return rc;
  mhip_ctx* ctx = m->ctx;
  if (!m->ready || !m->arena) return mhip_fail(ctx, MHIP_ESTATE, "crnn: weights not finalized");
  if (!crops || !argmax || !tokens || !lengths || !conf) return mhip_fail(ctx, MHIP_EINVAL, "crnn: null buffer");
  MHIP_HIP(ctx, hipSetDevice(ctx->device));
  const Plan p = make_plan(m, n, w);
  rc = mhip_ensure_workspace(ctx, p.total);
  if (rc) return rc;
  char* ws = (char*)ctx->ws;
  const Arena& L = m->lay;
  const char* A = m->arena;
  const int prec = m->precision;
  const int w2 = w / 2, w4 = w / 4, T = p.T;

  rc = mhip_launch_conv_first(ctx, prec, crops, (const float*)(A + L.conv0_w), (const float*)(A + L.conv0_b),
                              ws + p.act[0], n, 32, w);
  if (rc) return rc;

  struct LayerShape { int H, W, pool; };
  const LayerShape shp[7] = {{0, 0, 0}, {16, w2, POOL_2x2}, {8, w4, POOL_NONE}, {8, w4, POOL_2x1},
                             {4, w4, POOL_NONE}, {4, w4, POOL_2x1}, {2, w4, POOL_NONE}};
  for (int i = 1; i < 7; ++i) {
    const ConvSpec& c = kConvs[i];
    ConvDesc d;
    d.in = ws + p.act[i - 1];
    d.w = A + L.conv_w[i];
    d.scale = c.bn ? (const float*)(A + L.conv_scale[i]) : nullptr;
    d.bias = (const float*)(A + L.conv_bias[i]);
    d.out = ws + p.act[i];
    d.B = n; d.H = shp[i].H; d.W = shp[i].W; d.Cin = c.ci;
    d.KH = c.kh; d.KW = c.kw; d.pad = (c.kh == 3) ? 1 : 0;
    d.N = c.co;
    d.pool = shp[i].pool;
    d.relu = 1;
    rc = mhip_launch_conv_igemm(ctx, prec, d);
    if (rc) return rc;
  }
  // AdaptiveAvgPool2d((None,1)) over H is the identity here: the VGG stack reduces imgH = 32 to H = 1
  // (marie/models/icr/model.py:77-78), so act[6] is already the [n*T][512] sequence.
  const void* seq_in = ws + p.act[6];
  int seq_ch = 512;
  for (int j = 0; j < 2; ++j) {
    ConvDesc g;
    g.in = seq_in; g.w = A + L.ih_w[j]; g.bias = (const float*)(A + L.ih_b[j]); g.out = ws + p.xproj;
    g.B = n * T; g.H = 1; g.W = 1; g.Cin = seq_ch; g.N = 2048; g.out_f32 = 1;
    rc = mhip_launch_conv_igemm(ctx, prec, g);
    if (rc) return rc;
    rc = mhip_launch_lstm_rec(ctx, prec, (const float*)(ws + p.xproj), A + L.hh_pack[j], ws + p.hseq, n, T);
    if (rc) return rc;
    ConvDesc l;
    l.in = ws + p.hseq; l.w = A + L.lin_w[j]; l.bias = (const float*)(A + L.lin_b[j]); l.out = ws + p.lin[j];
    l.B = n * T; l.H = 1; l.W = 1; l.Cin = 512; l.N = 256;
    rc = mhip_launch_conv_igemm(ctx, prec, l);
    if (rc) return rc;
    seq_in = ws + p.lin[j];
    seq_ch = 256;
  }
  float* logits = logits_out ? logits_out : (float*)(ws + p.logits);
  {
    ConvDesc g;
    g.in = seq_in; g.w = A + L.pred_w; g.bias = (const float*)(A + L.pred_b); g.out = logits;
    g.B = n * T; g.H = 1; g.W = 1; g.Cin = 256; g.N = m->num_class; g.out_f32 = 1;
    rc = mhip_launch_conv_igemm(ctx, prec, g);
    if (rc) return rc;
  }
  return mhip_launch_ctc_decode(ctx, logits, n, T, m->num_class, argmax, tokens, lengths, conf);
}

extern "C" int mhip_crnn_forward_host(mhip_crnn* m, const uint8_t* crops_h, int n, int w, float* logits_h,
                                      int32_t* argmax_h, int32_t* tokens_h, int32_t* lengths_h, float* conf_h) {
  int rc = check_shape(m, n, w);
  if (rc) return rc;
  mhip_ctx* ctx = m->ctx;
  if (!crops_h || !argmax_h || !tokens_h || !lengths_h || !conf_h)
    return mhip_fail(ctx, MHIP_EINVAL, "crnn: null host buffer");
  MHIP_HIP(ctx, hipSetDevice(ctx->device));
  const int T = w / 4 - 1, C = m->num_class;
  const size_t in_b = (size_t)n * 32 * w, lg_b = (size_t)n * T * C * 4, it_b = (size_t)n * T * 4;
  // I/O staging lives behind the forward workspace
  const Plan p = make_plan(m, n, w);
  size_t o = p.total;
  auto take = [&](size_t bytes) {
    size_t at = o;
    o = align_up(o + bytes, 4096);
    return at;
  };
  const size_t o_in = take(in_b), o_lg = take(lg_b), o_am = take(it_b), o_tk = take(it_b), o_ln = take((size_t)n * 4),
               o_cf = take((size_t)n * 4);
  rc = mhip_ensure_workspace(ctx, o);
  if (rc) return rc;
  char* ws = (char*)ctx->ws;
  MHIP_HIP(ctx, hipMemcpyAsync(ws + o_in, crops_h, in_b, hipMemcpyHostToDevice, ctx->stream));
  rc = mhip_crnn_forward(m, (const uint8_t*)(ws + o_in), n, w, (float*)(ws + o_lg), (int32_t*)(ws + o_am),
                         (int32_t*)(ws + o_tk), (int32_t*)(ws + o_ln), (float*)(ws + o_cf));
  if (rc) return rc;
  if (logits_h) MHIP_HIP(ctx, hipMemcpyAsync(logits_h, ws + o_lg, lg_b, hipMemcpyDeviceToHost, ctx->stream));
  MHIP_HIP(ctx, hipMemcpyAsync(argmax_h, ws + o_am, it_b, hipMemcpyDeviceToHost, ctx->stream));
  MHIP_HIP(ctx, hipMemcpyAsync(tokens_h, ws + o_tk, it_b, hipMemcpyDeviceToHost, ctx->stream));
  MHIP_HIP(ctx, hipMemcpyAsync(lengths_h, ws + o_ln, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
  MHIP_HIP(ctx, hipMemcpyAsync(conf_h, ws + o_cf, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
  MHIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return MHIP_OK;
}

// ======================================================================= crop batcher + composite entries
extern "C" int mhip_crop_batch(mhip_ctx* ctx, const uint8_t* base_dev, const mhip_crop_desc* descs, int n, int img_w,
                               uint8_t* out_dev) {
  if (!ctx || !descs || n < 1 || img_w < 1) return MHIP_EINVAL;
  MHIP_HIP(ctx, hipSetDevice(ctx->device));
  const size_t sb = mhip_crop_scratch_bytes(descs, n, 32, img_w);
  int rc = mhip_ensure_workspace(ctx, sb);
  if (rc) return rc;
  return mhip_launch_crop_batch(ctx, base_dev, descs, n, 32, img_w, ctx->ws, out_dev);
}

namespace {

int forward_crops_impl(mhip_crnn* m, const uint8_t* base_dev, const uint8_t* packed_host, size_t packed_bytes,
                       const mhip_crop_desc* descs, int n, int img_w, float* logits_h, int32_t* argmax_h,
                       int32_t* tokens_h, int32_t* lengths_h, float* conf_h) {
  int rc = check_shape(m, n, img_w);
  if (rc) return rc;
  mhip_ctx* ctx = m->ctx;
  if (!descs || !argmax_h || !tokens_h || !lengths_h || !conf_h) return mhip_fail(ctx, MHIP_EINVAL, "crnn: null buffer");
  MHIP_HIP(ctx, hipSetDevice(ctx->device));
  const int T = img_w / 4 - 1, C = m->num_class;
  const Plan p = make_plan(m, n, img_w);
  size_t o = p.total;
  auto take = [&](size_t bytes) {
    size_t at = o;
    o = align_up(o + bytes, 4096);
    return at;
  };
  const size_t it_b = (size_t)n * T * 4, lg_b = (size_t)n * T * C * 4;
  const size_t o_crops = take((size_t)n * 32 * img_w), o_scr = take(mhip_crop_scratch_bytes(descs, n, 32, img_w));
  const size_t o_lg = take(logits_h ? lg_b : 16), o_am = take(it_b), o_tk = take(it_b), o_ln = take((size_t)n * 4),
               o_cf = take((size_t)n * 4);
  const size_t o_pk = packed_host ? take(packed_bytes) : 0;
  rc = mhip_ensure_workspace(ctx, o);
  if (rc) return rc;
  char* ws = (char*)ctx->ws;
  if (packed_host) {
    MHIP_HIP(ctx, hipMemcpyAsync(ws + o_pk, packed_host, packed_bytes, hipMemcpyHostToDevice, ctx->stream));
    base_dev = (const uint8_t*)(ws + o_pk);
  }
  rc = mhip_launch_crop_batch(ctx, base_dev, descs, n, 32, img_w, ws + o_scr, (uint8_t*)(ws + o_crops));
  if (rc) return rc;
  rc = mhip_crnn_forward(m, (const uint8_t*)(ws + o_crops), n, img_w, logits_h ? (float*)(ws + o_lg) : nullptr,
                         (int32_t*)(ws + o_am), (int32_t*)(ws + o_tk), (int32_t*)(ws + o_ln), (float*)(ws + o_cf));
  if (rc) return rc;
  if (logits_h) MHIP_HIP(ctx, hipMemcpyAsync(logits_h, ws + o_lg, lg_b, hipMemcpyDeviceToHost, ctx->stream));
  MHIP_HIP(ctx, hipMemcpyAsync(argmax_h, ws + o_am, it_b, hipMemcpyDeviceToHost, ctx->stream));
  MHIP_HIP(ctx, hipMemcpyAsync(tokens_h, ws + o_tk, it_b, hipMemcpyDeviceToHost, ctx->stream));
  MHIP_HIP(ctx, hipMemcpyAsync(lengths_h, ws + o_ln, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
  MHIP_HIP(ctx, hipMemcpyAsync(conf_h, ws + o_cf, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
  MHIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return MHIP_OK;
}

}  // namespace

extern "C" int mhip_crnn_forward_crops(mhip_crnn* m, const uint8_t* base_dev, const mhip_crop_desc* descs, int n,
                                       int img_w, float* logits_h, int32_t* argmax_h, int32_t* tokens_h,
                                       int32_t* lengths_h, float* conf_h) {
  if (!m || !base_dev) return MHIP_EINVAL;
  return forward_crops_impl(m, base_dev, nullptr, 0, descs, n, img_w, logits_h, argmax_h, tokens_h, lengths_h, conf_h);
}

extern "C" int mhip_crnn_forward_fragments_host(mhip_crnn* m, const uint8_t* packed_host, size_t packed_bytes,
                                                const mhip_crop_desc* descs, int n, int img_w, float* logits_h,
                                                int32_t* argmax_h, int32_t* tokens_h, int32_t* lengths_h,
                                                float* conf_h) {
  if (!m || !packed_host || !packed_bytes) return MHIP_EINVAL;
  return forward_crops_impl(m, nullptr, packed_host, packed_bytes, descs, n, img_w, logits_h, argmax_h, tokens_h,
                            lengths_h, conf_h);
}
