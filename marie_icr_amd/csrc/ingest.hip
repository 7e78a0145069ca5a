// ingest.hip — page ingest: the max-page-size clamp of marie/utils/image_utils.py:254-321 (ensure_max_page_size).
//
// The reference shrinks an oversized frame with cv2.resize(frame, (new_w, new_h), interpolation=cv2.INTER_AREA) on the
// host.  Here the frame is shrunk on the device, right where the detector and the crop batcher read it.  The arithmetic
// follows OpenCV's area resampler for 8-bit images as published (imgproc resize.cpp): source-cell coverage tables built
// in double precision, float32 accumulation along x and then along y in table order, round-half-even and saturate; an
// exactly integral scale takes the block-sum path (2x2: (a+b+c+d+2)>>2).  HBM-bound: every source byte is read about
// once (neighbouring destination pixels share at most one source column / row through L2), one destination byte written.
#include <math.h>

#include "common.h"

// The resampler's value is defined by separately rounded multiplies and adds (OpenCV's scalar loop is built without
// fused multiply-add).  hipcc contracts a * b + c by default and HIP's __fmul_rn / __fadd_rn are plain operators, so
// contraction is switched off for the whole file.
#pragma clang fp contract(off)

namespace {

struct AreaArgs {
  const uint8_t* src;
  uint8_t* dst;
  size_t src_pitch, dst_pitch;
  int sh, sw, dh, dw, cn;
  double scale_x, scale_y;
  int iscale_x, iscale_y;   // > 0: exactly integral scales (block-sum path)
};

// one axis' coverage of destination index d: up to `n` consecutive source indices starting at s0, weights w[]
// (first and last partial, the middle ones 1 / cellWidth)
struct Span {
  int s0, n;
  float first, mid, last;
  bool has_first, has_last;
};

__device__ __forceinline__ Span span_of(int d, double scale, int ssize) {
  Span sp;
  const double f1 = (double)d * scale;
  const double f2 = f1 + scale;
  const double cell = fmin(scale, (double)ssize - f1);
  int s1 = (int)ceil(f1), s2 = (int)floor(f2);
  s2 = min(s2, ssize - 1);
  s1 = min(s1, s2);
  sp.has_first = ((double)s1 - f1) > 1e-3;
  sp.first = (float)(((double)s1 - f1) / cell);
  sp.mid = (float)(1.0 / cell);
  sp.has_last = (f2 - (double)s2) > 1e-3;
  sp.last = (float)(fmin(fmin(f2 - (double)s2, 1.0), cell) / cell);
  sp.s0 = sp.has_first ? s1 - 1 : s1;
  sp.n = (s2 - s1) + (sp.has_first ? 1 : 0) + (sp.has_last ? 1 : 0);
  return sp;
}

__device__ __forceinline__ float span_weight(const Span& sp, int i) {
  if (i == 0 && sp.has_first) return sp.first;
  if (i == sp.n - 1 && sp.has_last) return sp.last;
  return sp.mid;
}

__device__ __forceinline__ uint8_t sat_u8(float v) {
  const float r = rintf(v);            // round half to even (cvRound)
  return (uint8_t)fminf(fmaxf(r, 0.f), 255.f);
}

template <int CN>
__global__ __launch_bounds__(256) void resize_area_kernel(AreaArgs p) {
  const int dx = blockIdx.x * 64 + (threadIdx.x & 63);
  const int dy = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (dx >= p.dw || dy >= p.dh) return;
  uint8_t* out = p.dst + (size_t)dy * p.dst_pitch + (size_t)dx * CN;
  if (p.iscale_x > 0) {                // exactly integral scales: block sums
    const int sx0 = dx * p.iscale_x, sy0 = dy * p.iscale_y;
    const int nx = min(p.iscale_x, p.sw - sx0), ny = min(p.iscale_y, p.sh - sy0);
    int sum[CN];
#pragma unroll
    for (int c = 0; c < CN; ++c) sum[c] = 0;
    for (int y = 0; y < ny; ++y) {
      const uint8_t* s = p.src + (size_t)(sy0 + y) * p.src_pitch + (size_t)sx0 * CN;
      for (int x = 0; x < nx; ++x)
#pragma unroll
        for (int c = 0; c < CN; ++c) sum[c] += s[x * CN + c];
    }
    const bool whole = nx == p.iscale_x && ny == p.iscale_y;
#pragma unroll
    for (int c = 0; c < CN; ++c) {
      if (whole && p.iscale_x == 2 && p.iscale_y == 2) out[c] = (uint8_t)((sum[c] + 2) >> 2);
      else if (whole) out[c] = sat_u8((float)sum[c] * (1.f / (float)(p.iscale_x * p.iscale_y)));
      else out[c] = sat_u8((float)sum[c] / (float)(nx * ny));
    }
    return;
  }
  const Span sx = span_of(dx, p.scale_x, p.sw), sy = span_of(dy, p.scale_y, p.sh);
  float sum[CN];
#pragma unroll
  for (int c = 0; c < CN; ++c) sum[c] = 0.f;
  for (int j = 0; j < sy.n; ++j) {
    const float beta = span_weight(sy, j);
    const uint8_t* s = p.src + (size_t)(sy.s0 + j) * p.src_pitch + (size_t)sx.s0 * CN;
    float buf[CN];
#pragma unroll
    for (int c = 0; c < CN; ++c) buf[c] = 0.f;
    for (int i = 0; i < sx.n; ++i) {
      const float alpha = span_weight(sx, i);
#pragma unroll
      for (int c = 0; c < CN; ++c) buf[c] = buf[c] + (float)s[i * CN + c] * alpha;   // two roundings (contract off)
    }
#pragma unroll
    for (int c = 0; c < CN; ++c) sum[c] = sum[c] + beta * buf[c];
  }
#pragma unroll
  for (int c = 0; c < CN; ++c) out[c] = sat_u8(sum[c]);
}

// ---- cv2.INTER_CUBIC for 8-bit images (resize_image's shrink branch, marie/utils/resize_image.py:53-61) --------------
// OpenCV's generic separable path: per axis  f = (float)((d + 0.5) * scale - 0.5), s = floor(f), t = f - s, four taps
// s-1 .. s+2 (replicated at the borders), Keys cubic with A = -0.75 evaluated in float32 and rounded to 11-bit fixed
// point (x 2048, half-even); rows are combined as 32-bit integers, the result is (v + 2^21) >> 22, saturated.
struct CubicArgs {
  const uint8_t* src;
  uint8_t* dst;
  size_t src_pitch, dst_pitch;
  int sh, sw, dh, dw;
  double scale_x, scale_y;
};

__device__ __forceinline__ void cubic_taps(int d, double scale, int& s, int c[4]) {
  float f = (float)(((double)d + 0.5) * scale - 0.5);
  const float fl = floorf(f);
  s = (int)fl;
  const float x = f - fl;
  const float A = -0.75f;
  float w[4];
  w[0] = ((A * (x + 1.f) - 5.f * A) * (x + 1.f) + 8.f * A) * (x + 1.f) - 4.f * A;
  w[1] = ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f;
  w[2] = ((A + 2.f) * (1.f - x) - (A + 3.f)) * (1.f - x) * (1.f - x) + 1.f;
  w[3] = 1.f - w[0] - w[1] - w[2];
#pragma unroll
  for (int k = 0; k < 4; ++k) c[k] = (int)fminf(fmaxf(rintf(w[k] * 2048.f), -32768.f), 32767.f);
}

template <int CN>
__global__ __launch_bounds__(256) void resize_cubic_kernel(CubicArgs p) {
  const int dx = blockIdx.x * 64 + (threadIdx.x & 63);
  const int dy = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (dx >= p.dw || dy >= p.dh) return;
  int sx, sy, ca[4], cb[4];
  cubic_taps(dx, p.scale_x, sx, ca);
  cubic_taps(dy, p.scale_y, sy, cb);
  long long acc[CN];
#pragma unroll
  for (int c = 0; c < CN; ++c) acc[c] = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int yy = min(max(sy - 1 + k, 0), p.sh - 1);
    const uint8_t* row = p.src + (size_t)yy * p.src_pitch;
    int h[CN];
#pragma unroll
    for (int c = 0; c < CN; ++c) h[c] = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int xx = min(max(sx - 1 + j, 0), p.sw - 1);
#pragma unroll
      for (int c = 0; c < CN; ++c) h[c] += (int)row[(size_t)xx * CN + c] * ca[j];
    }
#pragma unroll
    for (int c = 0; c < CN; ++c) acc[c] += (long long)h[c] * cb[k];
  }
  uint8_t* out = p.dst + (size_t)dy * p.dst_pitch + (size_t)dx * CN;
#pragma unroll
  for (int c = 0; c < CN; ++c) {
    const long long v = (acc[c] + (1ll << 21)) >> 22;
    out[c] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
  }
}

}  // namespace

// device entry: src u8 [sh][sw][cn] (row pitch in bytes) -> dst u8 [dh][dw][cn]; both axes shrink or keep (dh <= sh, dw <= sw)
extern "C" int mhip_resize_area_u8(mhip_ctx* ctx, const uint8_t* src_dev, int sh, int sw, int cn, size_t src_pitch,
                                   uint8_t* dst_dev, int dh, int dw) {
  if (!ctx) return MHIP_EINVAL;
  if (!src_dev || !dst_dev) return mhip_fail(ctx, MHIP_EINVAL, "resize_area: null buffer");
  if (cn != 1 && cn != 3) return mhip_fail(ctx, MHIP_EINVAL, "resize_area: %d channels (1 or 3)", cn);
  if (sh <= 0 || sw <= 0 || dh <= 0 || dw <= 0 || dh > sh || dw > sw)
    return mhip_fail(ctx, MHIP_EINVAL, "resize_area: %dx%d -> %dx%d is not a shrink", sw, sh, dw, dh);
  if (src_pitch < (size_t)sw * cn) return mhip_fail(ctx, MHIP_EINVAL, "resize_area: source pitch below the row size");
  AreaArgs a;
  a.src = src_dev; a.dst = dst_dev;
  a.src_pitch = src_pitch; a.dst_pitch = (size_t)dw * cn;
  a.sh = sh; a.sw = sw; a.dh = dh; a.dw = dw; a.cn = cn;
  const double inv_x = (double)dw / sw, inv_y = (double)dh / sh;      // cv::resize: scale = 1 / (dsize / ssize)
  a.scale_x = 1.0 / inv_x; a.scale_y = 1.0 / inv_y;
  const int ix = (int)lrint(a.scale_x), iy = (int)lrint(a.scale_y);   // saturate_cast<int>(double) rounds
  const bool fast = fabs(a.scale_x - ix) < 2.220446049250313e-16 && fabs(a.scale_y - iy) < 2.220446049250313e-16;
  a.iscale_x = fast ? ix : 0; a.iscale_y = fast ? iy : 0;
  MHIP_HIP(ctx, hipSetDevice(ctx->device));
  dim3 grid((unsigned)((dw + 63) / 64), (unsigned)((dh + 3) / 4)), block(256);
  if (cn == 1) PROF_LAUNCH(ctx, MHIP_K_IMAGE_OPS, hipLaunchKernelGGL(resize_area_kernel<1>, grid, block, 0, ctx->stream, a));
  else PROF_LAUNCH(ctx, MHIP_K_IMAGE_OPS, hipLaunchKernelGGL(resize_area_kernel<3>, grid, block, 0, ctx->stream, a));
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return mhip_fail(ctx, MHIP_EHIP, "resize_area launch: %s", hipGetErrorString(e));
  return MHIP_OK;
}

// host entry (tests / standalone): contiguous host buffers in and out
extern "C" int mhip_resize_area_u8_host(mhip_ctx* ctx, const uint8_t* src_host, int sh, int sw, int cn, uint8_t* dst_host,
                                        int dh, int dw) {
  if (!ctx || !src_host || !dst_host) return MHIP_EINVAL;
  if (sh <= 0 || sw <= 0 || dh <= 0 || dw <= 0 || (cn != 1 && cn != 3)) return mhip_fail(ctx, MHIP_EINVAL, "resize_area: bad shape");
  MHIP_HIP(ctx, hipSetDevice(ctx->device));
  const size_t sb = (size_t)sh * sw * cn, db = (size_t)dh * dw * cn;
  int rc = mhip_ensure_workspace(ctx, sb + db + 1024);
  if (rc) return rc;
  uint8_t* s = (uint8_t*)ctx->ws;
  uint8_t* d = s + (sb + 255) / 256 * 256;
  MHIP_HIP(ctx, hipMemcpyAsync(s, src_host, sb, hipMemcpyHostToDevice, ctx->stream));
  if ((rc = mhip_resize_area_u8(ctx, s, sh, sw, cn, (size_t)sw * cn, d, dh, dw))) return rc;
  MHIP_HIP(ctx, hipMemcpyAsync(dst_host, d, db, hipMemcpyDeviceToHost, ctx->stream));
  MHIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return MHIP_OK;
}

// device entry: cv2.resize(src, (dw, dh), interpolation=cv2.INTER_CUBIC), u8, 1 or 3 channels, any scale
extern "C" int mhip_resize_cubic_u8(mhip_ctx* ctx, const uint8_t* src_dev, int sh, int sw, int cn, size_t src_pitch,
                                    uint8_t* dst_dev, int dh, int dw) {
  if (!ctx) return MHIP_EINVAL;
  if (!src_dev || !dst_dev) return mhip_fail(ctx, MHIP_EINVAL, "resize_cubic: null buffer");
  if (cn != 1 && cn != 3) return mhip_fail(ctx, MHIP_EINVAL, "resize_cubic: %d channels (1 or 3)", cn);
  if (sh <= 0 || sw <= 0 || dh <= 0 || dw <= 0) return mhip_fail(ctx, MHIP_EINVAL, "resize_cubic: empty image");
  if (src_pitch < (size_t)sw * cn) return mhip_fail(ctx, MHIP_EINVAL, "resize_cubic: source pitch below the row size");
  CubicArgs a;
  a.src = src_dev; a.dst = dst_dev;
  a.src_pitch = src_pitch; a.dst_pitch = (size_t)dw * cn;
  a.sh = sh; a.sw = sw; a.dh = dh; a.dw = dw;
  a.scale_x = 1.0 / ((double)dw / sw); a.scale_y = 1.0 / ((double)dh / sh);
  MHIP_HIP(ctx, hipSetDevice(ctx->device));
  dim3 grid((unsigned)((dw + 63) / 64), (unsigned)((dh + 3) / 4)), block(256);
  if (cn == 1) PROF_LAUNCH(ctx, MHIP_K_IMAGE_OPS, hipLaunchKernelGGL(resize_cubic_kernel<1>, grid, block, 0, ctx->stream, a));
  else PROF_LAUNCH(ctx, MHIP_K_IMAGE_OPS, hipLaunchKernelGGL(resize_cubic_kernel<3>, grid, block, 0, ctx->stream, a));
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return mhip_fail(ctx, MHIP_EHIP, "resize_cubic launch: %s", hipGetErrorString(e));
  return MHIP_OK;
}

extern "C" int mhip_resize_cubic_u8_host(mhip_ctx* ctx, const uint8_t* src_host, int sh, int sw, int cn, uint8_t* dst_host,
                                         int dh, int dw) {
  if (!ctx || !src_host || !dst_host) return MHIP_EINVAL;
  if (sh <= 0 || sw <= 0 || dh <= 0 || dw <= 0 || (cn != 1 && cn != 3)) return mhip_fail(ctx, MHIP_EINVAL, "resize_cubic: bad shape");
  MHIP_HIP(ctx, hipSetDevice(ctx->device));
  const size_t sb = (size_t)sh * sw * cn, db = (size_t)dh * dw * cn;
  int rc = mhip_ensure_workspace(ctx, sb + db + 1024);
  if (rc) return rc;
  uint8_t* s = (uint8_t*)ctx->ws;
  uint8_t* d = s + (sb + 255) / 256 * 256;
  MHIP_HIP(ctx, hipMemcpyAsync(s, src_host, sb, hipMemcpyHostToDevice, ctx->stream));
  if ((rc = mhip_resize_cubic_u8(ctx, s, sh, sw, cn, (size_t)sw * cn, d, dh, dw))) return rc;
  MHIP_HIP(ctx, hipMemcpyAsync(dst_host, d, db, hipMemcpyDeviceToHost, ctx->stream));
  MHIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return MHIP_OK;
}

// The shape rule of ensure_max_page_size (image_utils.py:275-310) for one frame.  Returns 1 and the new size when the
// frame exceeds the (orientation-aware, expanded) maximum, 0 when it is kept as it is.  No GPU involved.
extern "C" int mhip_max_page_size(int width, int height, int max_w_portrait, int max_h_portrait, double expand_ratio,
                                  int* new_w, int* new_h) {
  int max_w = max_w_portrait, max_h = max_h_portrait;
  if (width > height) { max_w = max_h_portrait; max_h = max_w_portrait; }      // landscape: swap
  max_w = max_w + (int)((double)max_w * expand_ratio);                          // int() truncates
  max_h = max_h + (int)((double)max_h * expand_ratio);
  int nw = width, nh = height, changed = 0;
  if (width > max_w || height > max_h) {
    changed = 1;
    const double aspect = (double)width / (double)height;
    if (width > height) {
      nw = width < max_w ? width : max_w;
      nh = (int)((double)nw / aspect);
      if (nh > max_h) { nh = max_h; nw = (int)((double)nh * aspect); }
    } else {
      nh = height < max_h ? height : max_h;
      nw = (int)((double)nh * aspect);
      if (nw > max_w) { nw = max_w; nh = (int)((double)nw / aspect); }
    }
  }
  if (new_w) *new_w = nw;
  if (new_h) *new_h = nh;
  return changed;
}
