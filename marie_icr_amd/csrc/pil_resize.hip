// pil_resize.hip — Pillow's 8-bit Image.resize (BILINEAR / BICUBIC, antialiased) for whole pages and crops, bit-exact.
//
// Replaces: detectron2 ResizeShortestEdge -> ResizeTransform.apply_image's PIL bilinear resize reached from
// OptimizedDetectronPredictor.invoke_model (marie/detectron/detector.py:103-105), and TrOCR's
// ``im.convert("RGB").resize((384, 384), BICUBIC)`` (marie/document/trocr_ocr_processor.py:116-118).
//
// libImaging/Resample.c: per output coordinate a window [xmin, xmin+n) of weights filter((x - center + 0.5) * ss),
// normalised in double and rounded to 22-bit fixed point; horizontal pass rounded to uint8, then vertical pass.
// A tiny kernel builds the two coefficient tables on the device (IEEE double, contraction off — the same arithmetic as
// the C code); the passes are pure integer MACs, 3 interleaved channels per thread.
#include <math.h>

#include <algorithm>
#include <vector>

#include "common.h"

namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;

#pragma clang fp contract(off)
__device__ __forceinline__ double filt(int filter, double x) {
  if (x < 0.0) x = -x;
  if (filter == MHIP_PIL_BILINEAR) return x < 1.0 ? 1.0 - x : 0.0;
  const double a = -0.5;
  if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
  if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
  return 0.0;
}

// bounds[2*xx] = xmin, bounds[2*xx+1] = n ; kk[xx*ksize + x] = fixed-point weight
__global__ void pil_coeffs_kernel(int in_size, int out_size, int filter, int ksize, int* __restrict__ bounds,
                                  int* __restrict__ kk) {
  const int xx = blockIdx.x * blockDim.x + threadIdx.x;
  if (xx >= out_size) return;
  const double scale = (double)in_size / (double)out_size;
  double filterscale = scale;
  if (filterscale < 1.0) filterscale = 1.0;
  const double support = (filter == MHIP_PIL_BILINEAR ? 1.0 : 2.0) * filterscale;
  const double center = ((double)xx + 0.5) * scale;
  const double ss = 1.0 / filterscale;
  int xmin = (int)(center - support + 0.5);
  if (xmin < 0) xmin = 0;
  int xmax = (int)(center + support + 0.5);
  if (xmax > in_size) xmax = in_size;
  xmax -= xmin;
  double ww = 0.0;
  for (int x = 0; x < xmax; ++x) ww += filt(filter, ((double)(x + xmin) - center + 0.5) * ss);
  for (int x = 0; x < ksize; ++x) {
    int k = 0;
    if (x < xmax) {
      double w = filt(filter, ((double)(x + xmin) - center + 0.5) * ss);
      if (ww != 0.0) w /= ww;
      k = w < 0 ? (int)(-0.5 + w * (double)(1 << PRECISION_BITS)) : (int)(0.5 + w * (double)(1 << PRECISION_BITS));
    }
    kk[(size_t)xx * ksize + x] = k;
  }
  bounds[2 * xx] = xmin;
  bounds[2 * xx + 1] = xmax;
}

__device__ __forceinline__ uint8_t clip8(int v) {
  v >>= PRECISION_BITS;
  return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// horizontal: src [sh][src_stride bytes] RGB -> tmp [sh][dw][3]
__global__ __launch_bounds__(256) void pil_hpass_kernel(const uint8_t* __restrict__ src, int sh, size_t src_stride, int dw,
                                                        int ksize, const int* __restrict__ bounds,
                                                        const int* __restrict__ kk, uint8_t* __restrict__ tmp) {
  const long long total = (long long)sh * dw;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int yy = (int)(i / dw), xx = (int)(i - (long long)yy * dw);
    const int xmin = bounds[2 * xx], n = bounds[2 * xx + 1];
    const int* k = kk + (size_t)xx * ksize;
    const uint8_t* p = src + (size_t)yy * src_stride + (size_t)xmin * 3;
    int a0 = 1 << (PRECISION_BITS - 1), a1 = a0, a2 = a0;
    for (int x = 0; x < n; ++x) {
      const int w = k[x];
      a0 += (int)p[3 * x] * w; a1 += (int)p[3 * x + 1] * w; a2 += (int)p[3 * x + 2] * w;
    }
    uint8_t* o = tmp + (size_t)i * 3;
    o[0] = clip8(a0); o[1] = clip8(a1); o[2] = clip8(a2);
  }
}

// vertical: tmp [sh][dw][3] -> dst [dh][dw][3]
__global__ __launch_bounds__(256) void pil_vpass_kernel(const uint8_t* __restrict__ tmp, int dw, int dh, int ksize,
                                                        const int* __restrict__ bounds, const int* __restrict__ kk,
                                                        uint8_t* __restrict__ dst) {
  const long long total = (long long)dh * dw;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int yy = (int)(i / dw), xx = (int)(i - (long long)yy * dw);
    const int ymin = bounds[2 * yy], n = bounds[2 * yy + 1];
    const int* k = kk + (size_t)yy * ksize;
    const uint8_t* p = tmp + ((size_t)ymin * dw + xx) * 3;
    int a0 = 1 << (PRECISION_BITS - 1), a1 = a0, a2 = a0;
    for (int y = 0; y < n; ++y) {
      const int w = k[y];
      const uint8_t* q = p + (size_t)y * dw * 3;
      a0 += (int)q[0] * w; a1 += (int)q[1] * w; a2 += (int)q[2] * w;
    }
    uint8_t* o = dst + (size_t)i * 3;
    o[0] = clip8(a0); o[1] = clip8(a1); o[2] = clip8(a2);
  }
}

// ---- batched variant: n fragments of different sizes -> n images of one size, one launch per pass -------------------
struct FragDev {
  unsigned long long src_off;   // byte offset of the fragment's first pixel
  unsigned long long tmp_off;   // byte offset of its [h][dw][3] intermediate
  int h, w, row_stride;
};

// coefficient tables of all fragments: axis 0 = x (in_size = w), axis 1 = y (in_size = h); kmax taps per output
__global__ void pil_coeffs_batch_kernel(const FragDev* __restrict__ fr, int axis, int out_size, int filter, int kmax,
                                        int* __restrict__ bounds, int* __restrict__ kk) {
  const int f = blockIdx.y, xx = blockIdx.x * blockDim.x + threadIdx.x;
  if (xx >= out_size) return;
  const int in_size = axis ? fr[f].h : fr[f].w;
  const double scale = (double)in_size / (double)out_size;
  double filterscale = scale;
  if (filterscale < 1.0) filterscale = 1.0;
  const double support = (filter == MHIP_PIL_BILINEAR ? 1.0 : 2.0) * filterscale;
  const double center = ((double)xx + 0.5) * scale;
  const double ss = 1.0 / filterscale;
  int xmin = (int)(center - support + 0.5);
  if (xmin < 0) xmin = 0;
  int xmax = (int)(center + support + 0.5);
  if (xmax > in_size) xmax = in_size;
  xmax -= xmin;
  double ww = 0.0;
  for (int x = 0; x < xmax; ++x) ww += filt(filter, ((double)(x + xmin) - center + 0.5) * ss);
  int* k = kk + ((size_t)f * out_size + xx) * kmax;
  for (int x = 0; x < kmax; ++x) {
    int v = 0;
    if (x < xmax) {
      double w = filt(filter, ((double)(x + xmin) - center + 0.5) * ss);
      if (ww != 0.0) w /= ww;
      v = w < 0 ? (int)(-0.5 + w * (double)(1 << PRECISION_BITS)) : (int)(0.5 + w * (double)(1 << PRECISION_BITS));
    }
    k[x] = v;
  }
  bounds[((size_t)f * out_size + xx) * 2] = xmin;
  bounds[((size_t)f * out_size + xx) * 2 + 1] = xmax;
}

__global__ __launch_bounds__(256) void pil_hpass_batch_kernel(const uint8_t* __restrict__ base, const FragDev* __restrict__ fr,
                                                              int dw, int kmax, const int* __restrict__ bounds,
                                                              const int* __restrict__ kk, uint8_t* __restrict__ tmp) {
  const FragDev d = fr[blockIdx.y];
  const int total = d.h * dw;
  const int* bb = bounds + (size_t)blockIdx.y * dw * 2;
  const int* kf = kk + (size_t)blockIdx.y * dw * kmax;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const int yy = i / dw, xx = i - yy * dw;
    const int xmin = bb[2 * xx], n = bb[2 * xx + 1];
    const int* k = kf + (size_t)xx * kmax;
    const uint8_t* p = base + d.src_off + (size_t)yy * d.row_stride + (size_t)xmin * 3;
    int a0 = 1 << (PRECISION_BITS - 1), a1 = a0, a2 = a0;
    for (int x = 0; x < n; ++x) {
      const int w = k[x];
      a0 += (int)p[3 * x] * w; a1 += (int)p[3 * x + 1] * w; a2 += (int)p[3 * x + 2] * w;
    }
    uint8_t* o = tmp + d.tmp_off + (size_t)i * 3;
    o[0] = clip8(a0); o[1] = clip8(a1); o[2] = clip8(a2);
  }
}

__global__ __launch_bounds__(256) void pil_vpass_batch_kernel(const uint8_t* __restrict__ tmp, const FragDev* __restrict__ fr,
                                                              int dw, int dh, int kmax, const int* __restrict__ bounds,
                                                              const int* __restrict__ kk, uint8_t* __restrict__ dst) {
  const FragDev d = fr[blockIdx.y];
  const int total = dh * dw;
  const int* bb = bounds + (size_t)blockIdx.y * dh * 2;
  const int* kf = kk + (size_t)blockIdx.y * dh * kmax;
  uint8_t* out = dst + (size_t)blockIdx.y * total * 3;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const int yy = i / dw, xx = i - yy * dw;
    const int ymin = bb[2 * yy], n = bb[2 * yy + 1];
    const int* k = kf + (size_t)yy * kmax;
    const uint8_t* p = tmp + d.tmp_off + ((size_t)ymin * dw + xx) * 3;
    int a0 = 1 << (PRECISION_BITS - 1), a1 = a0, a2 = a0;
    for (int y = 0; y < n; ++y) {
      const int w = k[y];
      const uint8_t* q = p + (size_t)y * dw * 3;
      a0 += (int)q[0] * w; a1 += (int)q[1] * w; a2 += (int)q[2] * w;
    }
    uint8_t* o = out + (size_t)i * 3;
    o[0] = clip8(a0); o[1] = clip8(a1); o[2] = clip8(a2);
  }
}

int ksize_of(int in_size, int out_size, int filter) {
  double scale = (double)in_size / (double)out_size;
  if (scale < 1.0) scale = 1.0;
  const double support = (filter == MHIP_PIL_BILINEAR ? 1.0 : 2.0) * scale;
  return (int)ceil(support) * 2 + 1;
}

}  // namespace

size_t mhip_pil_resize_scratch_bytes(int sh, int sw, int dh, int dw, int filter) {
  const size_t kx = ksize_of(sw, dw, filter), ky = ksize_of(sh, dh, filter);
  auto al = [](size_t v) { return (v + 255) / 256 * 256; };
  return al((size_t)sh * dw * 3) + al((size_t)dw * 8) + al((size_t)dw * kx * 4) + al((size_t)dh * 8) + al((size_t)dh * ky * 4);
}

// src: u8 RGB rows of `src_stride` bytes; dst [dh][dw][3]; scratch from mhip_pil_resize_scratch_bytes
int mhip_launch_pil_resize_rgb(mhip_ctx* ctx, const uint8_t* src, int sh, int sw, size_t src_stride, uint8_t* dst, int dh,
                               int dw, int filter, void* scratch) {
  if (sh < 1 || sw < 1 || dh < 1 || dw < 1 || (filter != MHIP_PIL_BILINEAR && filter != MHIP_PIL_BICUBIC))
    return mhip_fail(ctx, MHIP_EINVAL, "pil_resize: bad arguments");
  const int kx = ksize_of(sw, dw, filter), ky = ksize_of(sh, dh, filter);
  auto al = [](size_t v) { return (v + 255) / 256 * 256; };
  char* p = (char*)scratch;
  uint8_t* tmp = (uint8_t*)p; p += al((size_t)sh * dw * 3);
  int* bx = (int*)p; p += al((size_t)dw * 8);
  int* kkx = (int*)p; p += al((size_t)dw * kx * 4);
  int* by = (int*)p; p += al((size_t)dh * 8);
  int* kky = (int*)p;
  PROF_LAUNCH(ctx, MHIP_K_IMAGE_OPS, {
    hipLaunchKernelGGL(pil_coeffs_kernel, dim3((dw + 255) / 256), dim3(256), 0, ctx->stream, sw, dw, filter, kx, bx, kkx);
    hipLaunchKernelGGL(pil_coeffs_kernel, dim3((dh + 255) / 256), dim3(256), 0, ctx->stream, sh, dh, filter, ky, by, kky);
    const long long t1 = (long long)sh * dw, t2 = (long long)dh * dw;
    hipLaunchKernelGGL(pil_hpass_kernel, dim3((unsigned)std::min<long long>((t1 + 255) / 256, 1 << 20)), dim3(256), 0, ctx->stream, src, sh, src_stride, dw, kx, bx, kkx, tmp);
    hipLaunchKernelGGL(pil_vpass_kernel, dim3((unsigned)std::min<long long>((t2 + 255) / 256, 1 << 20)), dim3(256), 0, ctx->stream, tmp, dw, dh, ky, by, kky, dst);
  });
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return mhip_fail(ctx, MHIP_EHIP, "pil_resize launch: %s", hipGetErrorString(e));
  return 0;
}

// n fragments (3-channel u8, descs on the host) -> dst [n][dh][dw][3].  Allocates its scratch from the context workspace
// TAIL (offset ws_off onwards), so callers that carve the head of the workspace are not disturbed.
int mhip_pil_resize_fragments(mhip_ctx* ctx, const uint8_t* base_dev, const mhip_crop_desc* descs, int n, uint8_t* dst, int dh,
                              int dw, int filter, void* scratch, size_t scratch_bytes) {
  if (n < 1) return 0;
  std::vector<FragDev> fr(n);
  size_t tmp_total = 0;
  int kx = 1, ky = 1, hmax = 1;
  for (int i = 0; i < n; ++i) {
    if (descs[i].channels != 3 || descs[i].h < 1 || descs[i].w < 1) return mhip_fail(ctx, MHIP_EINVAL, "pil_resize: fragment %d must be h x w x 3", i);
    fr[i].src_off = descs[i].src_offset; fr[i].tmp_off = tmp_total;
    fr[i].h = descs[i].h; fr[i].w = descs[i].w; fr[i].row_stride = descs[i].row_stride;
    tmp_total += ((size_t)descs[i].h * dw * 3 + 255) / 256 * 256;
    kx = std::max(kx, ksize_of(descs[i].w, dw, filter));
    ky = std::max(ky, ksize_of(descs[i].h, dh, filter));
    hmax = std::max(hmax, descs[i].h);
  }
  auto al = [](size_t v) { return (v + 255) / 256 * 256; };
  const size_t need = al(n * sizeof(FragDev)) + al(tmp_total) + al((size_t)n * dw * 8) + al((size_t)n * dw * kx * 4) +
                      al((size_t)n * dh * 8) + al((size_t)n * dh * ky * 4);
  if (need > scratch_bytes) return mhip_fail(ctx, MHIP_ENOMEM, "pil_resize: scratch %zu < %zu", scratch_bytes, need);
  char* p = (char*)scratch;
  FragDev* dfr = (FragDev*)p; p += al(n * sizeof(FragDev));
  uint8_t* tmp = (uint8_t*)p; p += al(tmp_total);
  int* bx = (int*)p; p += al((size_t)n * dw * 8);
  int* kkx = (int*)p; p += al((size_t)n * dw * kx * 4);
  int* by = (int*)p; p += al((size_t)n * dh * 8);
  int* kky = (int*)p;
  // fr is a host temporary: through pinned staging, without draining the stream (the engine encodes page batches back to back;
  // a drain here left the GPU idle while the host prepared the next batch: ~12 ms per 8 pages)
  {
    const int rc = mhip_stage_h2d(ctx, dfr, fr.data(), n * sizeof(FragDev));
    if (rc) return rc;
  }
  PROF_LAUNCH(ctx, MHIP_K_IMAGE_OPS, {
    hipLaunchKernelGGL(pil_coeffs_batch_kernel, dim3((dw + 255) / 256, n), dim3(256), 0, ctx->stream, dfr, 0, dw, filter, kx, bx, kkx);
    hipLaunchKernelGGL(pil_coeffs_batch_kernel, dim3((dh + 255) / 256, n), dim3(256), 0, ctx->stream, dfr, 1, dh, filter, ky, by, kky);
    hipLaunchKernelGGL(pil_hpass_batch_kernel, dim3((hmax * dw + 255) / 256, n), dim3(256), 0, ctx->stream, base_dev, dfr, dw, kx, bx, kkx, tmp);
    hipLaunchKernelGGL(pil_vpass_batch_kernel, dim3((dh * dw + 255) / 256, n), dim3(256), 0, ctx->stream, tmp, dfr, dw, dh, ky, by, kky, dst);
  });
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return mhip_fail(ctx, MHIP_EHIP, "pil_resize_fragments launch: %s", hipGetErrorString(e));
  return 0;
}

size_t mhip_pil_resize_fragments_scratch(const mhip_crop_desc* descs, int n, int dh, int dw, int filter) {
  auto al = [](size_t v) { return (v + 255) / 256 * 256; };
  size_t tmp_total = 0;
  int kx = 1, ky = 1;
  for (int i = 0; i < n; ++i) {
    tmp_total += ((size_t)std::max(descs[i].h, 1) * dw * 3 + 255) / 256 * 256;
    kx = std::max(kx, ksize_of(std::max(descs[i].w, 1), dw, filter));
    ky = std::max(ky, ksize_of(std::max(descs[i].h, 1), dh, filter));
  }
  return al(n * sizeof(FragDev)) + al(tmp_total) + al((size_t)n * dw * 8) + al((size_t)n * dw * kx * 4) + al((size_t)n * dh * 8) +
         al((size_t)n * dh * ky * 4) + 4096;
}

// replaces: Image.fromarray(rgb).resize((dw, dh), BILINEAR | BICUBIC) on host buffers (test / standalone entry)
extern "C" int mhip_pil_resize_rgb_host(mhip_ctx* ctx, const uint8_t* src_host, int sh, int sw, uint8_t* dst_host, int dh,
                                        int dw, int filter) {
  if (!ctx || !src_host || !dst_host) return MHIP_EINVAL;
  MHIP_HIP(ctx, hipSetDevice(ctx->device));
  const size_t sb = (size_t)sh * sw * 3, db = (size_t)dh * dw * 3;
  const size_t need = sb + db + mhip_pil_resize_scratch_bytes(sh, sw, dh, dw, filter) + 1024;
  int rc = mhip_ensure_workspace(ctx, need);
  if (rc) return rc;
  uint8_t* s = (uint8_t*)ctx->ws;
  uint8_t* d = s + (sb + 255) / 256 * 256;
  void* scratch = d + (db + 255) / 256 * 256;
  MHIP_HIP(ctx, hipMemcpyAsync(s, src_host, sb, hipMemcpyHostToDevice, ctx->stream));
  if ((rc = mhip_launch_pil_resize_rgb(ctx, s, sh, sw, (size_t)sw * 3, d, dh, dw, filter, scratch))) return rc;
  MHIP_HIP(ctx, hipMemcpyAsync(dst_host, d, db, hipMemcpyDeviceToHost, ctx->stream));
  MHIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return MHIP_OK;
}
