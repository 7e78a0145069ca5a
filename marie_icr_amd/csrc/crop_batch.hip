// crop_batch.hip — the recognizer's line/word crop batcher on the GPU, bit-exact with Pillow.
//
// Replaces, per fragment: cv2 BGR->RGB + Image.fromarray(...).convert("L")
// (marie/models/icr/memory_dataset.py:40-55), the aspect-preserving
// image.resize((w', 32), Image.BICUBIC) and NormalizePAD's right-edge replication
// (marie/models/icr/dataset.py:275-324).  The /255, -0.5, /0.5 normalisation is fused into the recognizer's
// first conv kernel (conv_first.hip), so this stage writes uint8.
//
// Pillow's 8-bit resampling is integer work: per output pixel a short filter window whose weights are
// bicubic(a = -0.5) values normalised in double precision and rounded to 22-bit fixed point, a horizontal pass
// rounded to uint8, then a vertical pass (libImaging/Resample.c).  Each thread owns one output pixel and rebuilds
// its (<= 2*ceil(2*scale)+1) weights itself in IEEE double with FMA contraction off — identical to the C code
// Pillow runs — so no coefficient tables cross PCIe and the kernels are pure gather + integer MAC:
//   pass 1: BGR (or gray) fragment  -> tmp  [h][w']   (gray conversion fused: (R*19595+G*38470+B*7471+0x8000)>>16)
//   pass 2: tmp                     -> out  [32][imgW] with the last column replicated to imgW.
#include <math.h>

#include "common.h"

namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;

#pragma clang fp contract(off)
__device__ __forceinline__ double bicubic(double x) {
  const double a = -0.5;
  if (x < 0.0) x = -x;
  if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
  if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
  return 0.0;
}

// window [xmin, xmin+xmax) and normaliser ww of output sample xx (Resample.c precompute_coeffs)
__device__ __forceinline__ void window(int in_size, int out_size, int xx, int* xmin_o, int* xmax_o, double* center_o,
                                       double* ss_o, double* ww_o) {
  const double scale = (double)in_size / (double)out_size;
  double filterscale = scale;
  if (filterscale < 1.0) filterscale = 1.0;
  const double support = 2.0 * filterscale;
  const double center = ((double)xx + 0.5) * scale;
  const double ss = 1.0 / filterscale;
  int xmin = (int)(center - support + 0.5);
  if (xmin < 0) xmin = 0;
  int xmax = (int)(center + support + 0.5);
  if (xmax > in_size) xmax = in_size;
  xmax -= xmin;
  double ww = 0.0;
  for (int x = 0; x < xmax; ++x) ww += bicubic(((double)(x + xmin) - center + 0.5) * ss);
  *xmin_o = xmin;
  *xmax_o = xmax;
  *center_o = center;
  *ss_o = ss;
  *ww_o = ww;
}

__device__ __forceinline__ int fixed_weight(int x, int xmin, double center, double ss, double ww) {
  double w = bicubic(((double)(x + xmin) - center + 0.5) * ss);
  if (ww != 0.0) w /= ww;
  return w < 0 ? (int)(-0.5 + w * (double)(1 << PRECISION_BITS)) : (int)(0.5 + w * (double)(1 << PRECISION_BITS));
}

__device__ __forceinline__ uint8_t clip8(int v) {
  v >>= PRECISION_BITS;
  return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

struct CropDev {
  unsigned long long src_off;  // byte offset of the fragment's first pixel in `base`
  unsigned long long tmp_off;  // byte offset of its [h][rw] intermediate in `tmp`
  int h, w, row_stride, channels, rw;
};

// pass 1: horizontal resample (+ gray conversion).  grid = (ceil(maxrw*maxh / 256), n)
__global__ __launch_bounds__(256) void crop_hpass_kernel(const uint8_t* __restrict__ base,
                                                         const CropDev* __restrict__ descs,
                                                         uint8_t* __restrict__ tmp) {
  const CropDev d = descs[blockIdx.y];
  const int total = d.h * d.rw;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int yy = i / d.rw, xx = i - yy * d.rw;
    const uint8_t* row = base + d.src_off + (size_t)yy * d.row_stride;
    int xmin, xmax;
    double center, ss, ww;
    window(d.w, d.rw, xx, &xmin, &xmax, &center, &ss, &ww);
    int acc = 1 << (PRECISION_BITS - 1);
    for (int x = 0; x < xmax; ++x) {
      const int k = fixed_weight(x, xmin, center, ss, ww);
      int g;
      if (d.channels == 3) {
        const uint8_t* p = row + (size_t)(x + xmin) * 3;   // B, G, R
        g = (int)(((unsigned)p[2] * 19595u + (unsigned)p[1] * 38470u + (unsigned)p[0] * 7471u + 0x8000u) >> 16);
      } else {
        g = row[x + xmin];
      }
      acc += g * k;
    }
    tmp[d.tmp_off + (size_t)yy * d.rw + xx] = clip8(acc);
  }
}

// pass 2: vertical resample to out_h rows + replicate the last column up to img_w.  grid = (ceil(out_h*img_w/256), n)
__global__ __launch_bounds__(256) void crop_vpass_kernel(const uint8_t* __restrict__ tmp,
                                                         const CropDev* __restrict__ descs, uint8_t* __restrict__ out,
                                                         int out_h, int img_w) {
  const CropDev d = descs[blockIdx.y];
  const int total = out_h * img_w;
  uint8_t* o = out + (size_t)blockIdx.y * total;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int yy = i / img_w, xo = i - yy * img_w;
    const int xx = xo < d.rw ? xo : d.rw - 1;   // NormalizePAD: columns beyond the resized width repeat the last one
    int ymin, ymax;
    double center, ss, ww;
    window(d.h, out_h, yy, &ymin, &ymax, &center, &ss, &ww);
    int acc = 1 << (PRECISION_BITS - 1);
    const uint8_t* col = tmp + d.tmp_off + xx;
    for (int y = 0; y < ymax; ++y) acc += (int)col[(size_t)(y + ymin) * d.rw] * fixed_weight(y, ymin, center, ss, ww);
    o[i] = clip8(acc);
  }
}

}  // namespace

// Host: AlignCollate's width rule (marie/models/icr/dataset.py:313-318), same double arithmetic as CPython.
int mhip_crop_resized_width(int w, int h, int img_h, int img_w) {
  const double ratio = (double)w / (double)h;
  const double c = ceil((double)img_h * ratio);
  int rw = (c > (double)img_w) ? img_w : (int)c;
  return rw < 1 ? 1 : rw;
}

// descs_host: n x {src_off, h, w, row_stride, channels}.  scratch_dev must hold mhip_crop_scratch_bytes().
size_t mhip_crop_scratch_bytes(const mhip_crop_desc* descs, int n, int img_h, int img_w) {
  size_t tmp = 0;
  for (int i = 0; i < n; ++i) {
    const int rw = mhip_crop_resized_width(descs[i].w, descs[i].h, img_h, img_w);
    tmp += ((size_t)descs[i].h * rw + 15) / 16 * 16;
  }
  return ((size_t)n * sizeof(CropDev) + 255) / 256 * 256 + tmp + 256;
}

int mhip_launch_crop_batch(mhip_ctx* ctx, const uint8_t* base_dev, const mhip_crop_desc* descs, int n, int img_h,
                           int img_w, void* scratch_dev, uint8_t* out_dev) {
  if (n < 1 || img_h < 1 || img_w < 1 || !descs || !base_dev || !scratch_dev || !out_dev)
    return mhip_fail(ctx, MHIP_EINVAL, "crop_batch: bad arguments");
  std::vector<CropDev> dv((size_t)n);
  const size_t desc_bytes = ((size_t)n * sizeof(CropDev) + 255) / 256 * 256;
  size_t off = 0;
  int max_hrw = 1;
  for (int i = 0; i < n; ++i) {
    const mhip_crop_desc& s = descs[i];
    if (s.h < 1 || s.w < 1 || (s.channels != 1 && s.channels != 3) || s.row_stride < s.w * s.channels)
      return mhip_fail(ctx, MHIP_EINVAL, "crop_batch: bad fragment %d (%dx%d, %d ch, stride %d)", i, s.h, s.w,
                       s.channels, s.row_stride);
    CropDev& d = dv[i];
    d.src_off = s.src_offset;
    d.h = s.h; d.w = s.w; d.row_stride = s.row_stride; d.channels = s.channels;
    d.rw = mhip_crop_resized_width(s.w, s.h, img_h, img_w);
    d.tmp_off = off;
    off += ((size_t)d.h * d.rw + 15) / 16 * 16;
    max_hrw = std::max(max_hrw, d.h * d.rw);
  }
  CropDev* d_desc = (CropDev*)scratch_dev;
  uint8_t* d_tmp = (uint8_t*)scratch_dev + desc_bytes;
  MHIP_HIP(ctx, hipMemcpyAsync(d_desc, dv.data(), (size_t)n * sizeof(CropDev), hipMemcpyHostToDevice, ctx->stream));
  MHIP_HIP(ctx, hipStreamSynchronize(ctx->stream));   // dv dies with this scope
  hipEvent_t e0 = nullptr;
  if (ctx->profiling) mhip_prof_begin(ctx, MHIP_K_CROP_BATCH, &e0);
  const unsigned gx1 = (unsigned)std::min((max_hrw + 255) / 256, 64);
  hipLaunchKernelGGL(crop_hpass_kernel, dim3(gx1, n), dim3(256), 0, ctx->stream, base_dev, d_desc, d_tmp);
  const unsigned gx2 = (unsigned)std::min((img_h * img_w + 255) / 256, 64);
  hipLaunchKernelGGL(crop_vpass_kernel, dim3(gx2, n), dim3(256), 0, ctx->stream, d_tmp, d_desc, out_dev, img_h, img_w);
  if (ctx->profiling) mhip_prof_end(ctx, MHIP_K_CROP_BATCH, e0);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return mhip_fail(ctx, MHIP_EHIP, "crop_batch launch: %s", hipGetErrorString(e));
  return 0;
}
