// content_ops.hip — where the ink is: the zero-pixel extent of the reference's content-cropping chain, for a batch of rectangles of
// one device page (HBM-bound byte kernels, one launch per stage whatever the number of rectangles).
//
// Replaces the OpenCV chain of crop_to_content (marie/utils/image_utils.py:190-252: the `crop_to_content` kwarg of
// OcrEngine.extract, marie/ocr/ocr_engine.py:169-176) and of crop_to_content_box (marie/boxes/dit/ulim_dit_box_processor.py:
// 291-352: the `bbox_optimization` option of psm_sparse, :608-626):
//     gray = BGR2GRAY;  content-aware: divide(gray, GaussianBlur(gray, 5x5, sigma 0), scale 255) -> Otsu -> close(2 x 3 rectangle)
//                       else:           Otsu(gray)
//     -> min / max x, y of the pixels that are 0.
// The callers turn the extent into their crop (the two functions pad it differently); that integer logic stays on the host
// (marie_icr_amd/content.py).  OpenCV's steps are restated from its published algorithms: BGR2GRAY (1868 B + 9617 G + 4899 R + 8192) >> 14;
// GaussianBlur 5x5 sigma 0 = [1 4 6 4 1] / 16 both ways, BORDER_REFLECT_101, exact fixed point; divide = saturate(rint(a * 255 / b)),
// 0 for b == 0; getThreshVal_Otsu_8u in double; erode / dilate read src(x + x' - anchor.x, y + y' - anchor.y), anchor (1, 1).
#include <limits.h>

#include "common.h"

namespace {

struct ContentRect {
  int x, y, w, h;
  unsigned long long off;     // this rectangle's first byte in the scratch planes
};

__device__ __forceinline__ int reflect101(int i, int n) {
  if (n == 1) return 0;
  const int p = 2 * (n - 1);
  i %= p;
  if (i < 0) i += p;
  return i >= n ? p - i : i;
}

// grid (chunks, rects): every workgroup walks its share of the rectangle's pixels
#define FOR_RECT_PIXELS(R, idx) \
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < (long long)(R).w * (R).h; idx += (long long)gridDim.x * blockDim.x)

__global__ __launch_bounds__(256) void content_gray_kernel(const uint8_t* __restrict__ page, int W, const ContentRect* __restrict__ rects,
                                                          uint8_t* __restrict__ G) {
  const ContentRect R = rects[blockIdx.y];
  FOR_RECT_PIXELS(R, idx) {
    const int py = (int)(idx / R.w), px = (int)(idx - (long long)py * R.w);
    const uint8_t* p = page + ((size_t)(R.y + py) * W + (R.x + px)) * 3;
    G[R.off + idx] = (uint8_t)((p[0] * 1868 + p[1] * 9617 + p[2] * 4899 + 8192) >> 14);
  }
}

// content-aware: D = saturate(rint(255 G / blur(G))) and its histogram; else the histogram of G
__global__ __launch_bounds__(256) void content_divide_kernel(const ContentRect* __restrict__ rects, const uint8_t* __restrict__ G,
                                                            uint8_t* __restrict__ D, unsigned* __restrict__ hist, int content_aware) {
  __shared__ unsigned lh[256];
  lh[threadIdx.x] = 0;
  __syncthreads();
  const ContentRect R = rects[blockIdx.y];
  const uint8_t* g = G + R.off;
  FOR_RECT_PIXELS(R, idx) {
    int v;
    if (content_aware) {
      const int py = (int)(idx / R.w), px = (int)(idx - (long long)py * R.w);
      const int wt[5] = {1, 4, 6, 4, 1};
      int xs[5];
#pragma unroll
      for (int k = 0; k < 5; ++k) xs[k] = reflect101(px + k - 2, R.w);
      int acc = 0;
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        const uint8_t* row = g + (size_t)reflect101(py + j - 2, R.h) * R.w;
        int hs = 0;
#pragma unroll
        for (int k = 0; k < 5; ++k) hs += wt[k] * row[xs[k]];
        acc += wt[j] * hs;
      }
      const int blur = (acc + 128) >> 8;
      const int a = g[idx];
      // cv2.divide, 8-bit with a scale: float32 arithmetic, round half to even, saturate; 0 where the divisor is 0
      v = blur == 0 ? 0 : min(255, (int)__builtin_rintf(((float)a * 255.0f) / (float)blur));
      D[R.off + idx] = (uint8_t)v;
    } else {
      v = g[idx];
    }
    atomicAdd(&lh[v], 1u);
  }
  __syncthreads();
  if (lh[threadIdx.x]) atomicAdd(&hist[(size_t)blockIdx.y * 256 + threadIdx.x], lh[threadIdx.x]);
}

// getThreshVal_Otsu_8u: one thread per rectangle, double precision, first maximum.  Also resets the rectangle's extent.
__global__ void content_otsu_kernel(const ContentRect* __restrict__ rects, const unsigned* __restrict__ hist, int* __restrict__ thr,
                                    int* __restrict__ ext, int n) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  const unsigned* h = hist + (size_t)r * 256;
  const double npix = (double)rects[r].w * (double)rects[r].h;
  int best = 0;
  if (npix > 0) {
    const double scale = 1.0 / npix;
    double mu = 0;
    for (int i = 0; i < 256; ++i) mu += i * (double)h[i];
    mu *= scale;
    double mu1 = 0, q1 = 0, max_sigma = 0;
    const double eps = 1.1920928955078125e-07;
    for (int i = 0; i < 256; ++i) {
      const double p_i = h[i] * scale;
      mu1 *= q1;
      q1 += p_i;
      const double q2 = 1.0 - q1;
      if (fmin(q1, q2) < eps || fmax(q1, q2) > 1.0 - eps) continue;
      mu1 = (mu1 + i * p_i) / q1;
      const double mu2 = (mu - q1 * mu1) / q2;
      const double sigma = q1 * q2 * (mu1 - mu2) * (mu1 - mu2);
      if (sigma > max_sigma) { max_sigma = sigma; best = i; }
    }
  }
  thr[r] = best;
  int* e = ext + (size_t)r * 5;
  e[0] = INT_MAX; e[1] = INT_MAX; e[2] = -1; e[3] = -1; e[4] = 0;
}

// dilate(D > thr) with the 2 x 3 rectangle, anchor (1, 1): columns x - 1 .. x, rows y - 1 .. y + 1 (outside: ignored) -> G
__global__ __launch_bounds__(256) void content_dilate_kernel(const ContentRect* __restrict__ rects, const uint8_t* __restrict__ D,
                                                            const int* __restrict__ thr, uint8_t* __restrict__ G) {
  const ContentRect R = rects[blockIdx.y];
  const int t = thr[blockIdx.y];
  const uint8_t* d = D + R.off;
  FOR_RECT_PIXELS(R, idx) {
    const int py = (int)(idx / R.w), px = (int)(idx - (long long)py * R.w);
    int any = 0;
    for (int dy = -1; dy <= 1; ++dy) {
      const int yy = py + dy;
      if (yy < 0 || yy >= R.h) continue;
      for (int dx = -1; dx <= 0; ++dx) {
        const int xx = px + dx;
        if (xx < 0) continue;
        any |= d[(size_t)yy * R.w + xx] > t;
      }
    }
    G[R.off + idx] = any ? 255 : 0;
  }
}

// the extent of the zero pixels of erode(G) (content-aware) or of G > thr (else)
__global__ __launch_bounds__(256) void content_extent_kernel(const ContentRect* __restrict__ rects, const uint8_t* __restrict__ G,
                                                            const int* __restrict__ thr, int* __restrict__ ext, int content_aware) {
  const ContentRect R = rects[blockIdx.y];
  const int t = thr[blockIdx.y];
  const uint8_t* g = G + R.off;
  int xmin = INT_MAX, ymin = INT_MAX, xmax = -1, ymax = -1, cnt = 0;
  FOR_RECT_PIXELS(R, idx) {
    const int py = (int)(idx / R.w), px = (int)(idx - (long long)py * R.w);
    bool zero;
    if (content_aware) {
      int all = 1;
      for (int dy = -1; dy <= 1; ++dy) {
        const int yy = py + dy;
        if (yy < 0 || yy >= R.h) continue;
        for (int dx = -1; dx <= 0; ++dx) {
          const int xx = px + dx;
          if (xx < 0) continue;
          all &= g[(size_t)yy * R.w + xx] != 0;
        }
      }
      zero = !all;
    } else {
      zero = !(g[idx] > t);
    }
    if (zero) {
      xmin = min(xmin, px); ymin = min(ymin, py); xmax = max(xmax, px); ymax = max(ymax, py);
      ++cnt;
    }
  }
  // wave reduction, then one set of atomics per wave
#pragma unroll
  for (int o = 32; o; o >>= 1) {
    xmin = min(xmin, __shfl_xor(xmin, o)); ymin = min(ymin, __shfl_xor(ymin, o));
    xmax = max(xmax, __shfl_xor(xmax, o)); ymax = max(ymax, __shfl_xor(ymax, o));
    cnt += __shfl_xor(cnt, o);
  }
  if ((threadIdx.x & 63) == 0 && cnt) {
    int* e = ext + (size_t)blockIdx.y * 5;
    atomicMin(&e[0], xmin); atomicMin(&e[1], ymin); atomicMax(&e[2], xmax); atomicMax(&e[3], ymax); atomicAdd(&e[4], cnt);
  }
}

}  // namespace

extern "C" int mhip_content_extents(mhip_ctx* ctx, const uint8_t* page_dev, int h, int w, const int32_t* rects_xywh_host, int n,
                                    int content_aware, int32_t* ext_host) {
  if (!ctx || !page_dev || !rects_xywh_host || !ext_host || h < 1 || w < 1 || n < 0) return MHIP_EINVAL;
  if (n == 0) return MHIP_OK;
  MHIP_HIP(ctx, hipSetDevice(ctx->device));
  std::vector<ContentRect> rects(n);
  unsigned long long total = 0;
  long long max_area = 0;
  for (int i = 0; i < n; ++i) {
    const int32_t* r = rects_xywh_host + 4 * i;
    if (r[0] < 0 || r[1] < 0 || r[2] < 0 || r[3] < 0 || (long long)r[0] + r[2] > w || (long long)r[1] + r[3] > h)
      return mhip_fail(ctx, MHIP_EINVAL, "content_extents: rectangle %d (%d, %d, %d, %d) leaves the %d x %d page", i, r[0], r[1], r[2], r[3], w, h);
    rects[i] = {r[0], r[1], r[2], r[3], total};
    const long long area = (long long)r[2] * r[3];
    total += (unsigned long long)((area + 15) / 16 * 16);
    max_area = area > max_area ? area : max_area;
  }
  if (max_area == 0) {
    for (int i = 0; i < n; ++i) { int32_t* e = ext_host + 5 * i; e[0] = e[1] = e[2] = e[3] = e[4] = 0; }
    return MHIP_OK;
  }
  auto up = [](size_t v) { return (v + 255) / 256 * 256; };
  const size_t o_rect = 0, o_hist = up(o_rect + (size_t)n * sizeof(ContentRect)), o_thr = up(o_hist + (size_t)n * 1024),
               o_ext = up(o_thr + (size_t)n * 4), o_g = up(o_ext + (size_t)n * 20), o_d = up(o_g + total), o_end = up(o_d + total);
  int rc = mhip_ensure_workspace(ctx, o_end);
  if (rc) return rc;
  char* ws = (char*)ctx->ws;
  ContentRect* d_rects = (ContentRect*)(ws + o_rect);
  unsigned* d_hist = (unsigned*)(ws + o_hist);
  int* d_thr = (int*)(ws + o_thr);
  int* d_ext = (int*)(ws + o_ext);
  uint8_t* G = (uint8_t*)(ws + o_g);
  uint8_t* D = (uint8_t*)(ws + o_d);
  MHIP_HIP(ctx, hipMemcpyAsync(d_rects, rects.data(), (size_t)n * sizeof(ContentRect), hipMemcpyHostToDevice, ctx->stream));
  MHIP_HIP(ctx, hipStreamSynchronize(ctx->stream));      // `rects` is pageable and dies with this call
  MHIP_HIP(ctx, hipMemsetAsync(d_hist, 0, (size_t)n * 1024, ctx->stream));
  const long long per_wg = 256 * 16;
  const unsigned chunks = (unsigned)std::min<long long>(2048, (max_area + per_wg - 1) / per_wg);
  const dim3 grid(chunks, (unsigned)n), block(256);
  PROF_LAUNCH(ctx, MHIP_K_IMAGE_OPS, hipLaunchKernelGGL(content_gray_kernel, grid, block, 0, ctx->stream, page_dev, w, d_rects, G));
  PROF_LAUNCH(ctx, MHIP_K_IMAGE_OPS,
              hipLaunchKernelGGL(content_divide_kernel, grid, block, 0, ctx->stream, d_rects, G, D, d_hist, content_aware));
  PROF_LAUNCH(ctx, MHIP_K_IMAGE_OPS,
              hipLaunchKernelGGL(content_otsu_kernel, dim3((n + 63) / 64), dim3(64), 0, ctx->stream, d_rects, d_hist, d_thr, d_ext, n));
  if (content_aware)
    PROF_LAUNCH(ctx, MHIP_K_IMAGE_OPS, hipLaunchKernelGGL(content_dilate_kernel, grid, block, 0, ctx->stream, d_rects, D, d_thr, G));
  PROF_LAUNCH(ctx, MHIP_K_IMAGE_OPS,
              hipLaunchKernelGGL(content_extent_kernel, grid, block, 0, ctx->stream, d_rects, G, d_thr, d_ext, content_aware));
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return mhip_fail(ctx, MHIP_EHIP, "content_extents launch: %s", hipGetErrorString(e));
  MHIP_HIP(ctx, hipMemcpyAsync(ext_host, d_ext, (size_t)n * 20, hipMemcpyDeviceToHost, ctx->stream));
  MHIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  for (int i = 0; i < n; ++i) {
    int32_t* x = ext_host + 5 * i;
    if (x[4] == 0) x[0] = x[1] = x[2] = x[3] = 0;
  }
  return MHIP_OK;
}
