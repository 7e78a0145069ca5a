// weights_util.h — host-side weight staging shared by the ViT / DiT / TrOCR model files: a state_dict-like tensor store,
// and an arena builder that lays named blocks out once (256-byte aligned) and fills them in the kernels' layouts.
#pragma once
#include <map>
#include <string>
#include <vector>

#include "common.h"

struct HostTensor {
  std::vector<int64_t> shape;
  std::vector<float> data;
  size_t numel() const { return data.size(); }
};

struct TensorStore {
  std::map<std::string, HostTensor> t;
  int set(mhip_ctx* ctx, const std::string& key, const float* data, const int64_t* shape, int ndim) {
    if (!data || ndim < 0 || ndim > 5 || (ndim > 0 && !shape)) return mhip_fail(ctx, MHIP_EINVAL, "bad tensor %s", key.c_str());
    HostTensor h;
    size_t n = 1;
    for (int i = 0; i < ndim; ++i) {
      if (shape[i] <= 0) return mhip_fail(ctx, MHIP_EINVAL, "bad shape for %s", key.c_str());
      h.shape.push_back(shape[i]);
      n *= (size_t)shape[i];
    }
    h.data.assign(data, data + n);
    t[key] = std::move(h);
    return MHIP_OK;
  }
  const HostTensor* find(mhip_ctx* ctx, const std::string& k, const std::vector<int64_t>& shape) const {
    auto it = t.find(k);
    if (it == t.end()) {
      mhip_fail(ctx, MHIP_ESTATE, "missing tensor %s", k.c_str());
      return nullptr;
    }
    if (it->second.shape != shape) {
      mhip_fail(ctx, MHIP_EINVAL, "tensor %s has the wrong shape", k.c_str());
      return nullptr;
    }
    return &it->second;
  }
  bool has(const std::string& k) const { return t.count(k) != 0; }
};

struct Arena {
  std::map<std::string, size_t> off;
  size_t bytes = 0;
  char* dev = nullptr;
  std::vector<char> host;
  size_t take(const std::string& name, size_t n) {
    size_t at = bytes;
    off[name] = at;
    bytes = (bytes + n + 255) / 256 * 256;
    return at;
  }
  bool known(const std::string& name) const { return off.count(name) != 0; }
  char* h(const std::string& name) { return host.data() + off.at(name); }
  template <typename P = char>
  P* d(const std::string& name) const { return (P*)(dev + off.at(name)); }
  void begin_fill() { host.assign(bytes, 0); }
  // fp32 source -> element type of `precision`
  static void put(int precision, char* dst, const float* src, size_t n) {
    if (precision == MHIP_PREC_F16) {
      _Float16* o = (_Float16*)dst;
      for (size_t i = 0; i < n; ++i) o[i] = (_Float16)src[i];
    } else {
      memcpy(dst, src, n * 4);
    }
  }
  int alloc(mhip_ctx* ctx) {
    if (!dev && hipMalloc((void**)&dev, bytes) != hipSuccess) {
      (void)hipGetLastError();
      return mhip_fail(ctx, MHIP_ENOMEM, "arena allocation of %zu bytes failed", bytes);
    }
    return MHIP_OK;
  }
  int upload(mhip_ctx* ctx) {
    int rc = alloc(ctx);
    if (rc) return rc;
    MHIP_HIP(ctx, hipMemcpy(dev, host.data(), bytes, hipMemcpyHostToDevice));
    host.clear();
    host.shrink_to_fit();
    return MHIP_OK;
  }
  void release() {
    if (dev) (void)hipFree(dev);
    dev = nullptr;
  }
};

// bump allocator over the context workspace for one forward
struct Carver {
  char* base;
  size_t off = 0;
  explicit Carver(void* b) : base((char*)b) {}
  template <typename P = char>
  P* take(size_t bytes) {
    P* p = (P*)(base + off);
    off = (off + bytes + 255) / 256 * 256;
    return p;
  }
};
