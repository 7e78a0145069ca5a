// overlay_ops.hip — the pieces of the overlay cleaner (pix2pixHD LocalEnhancer generator + blend) that are not
// implicit-GEMM convolutions.
//
// replaces: marie/models/pix2pix/models/networks_hd.py:24-213 (ReflectionPad2d, InstanceNorm2d(affine=False), Swish,
// Upsample(bilinear, align_corners=True), ConvTranspose2d's zero insertion, Tanh), the dataset transform + tensor2im of
// marie/overlay/overlay.py:165-189 (ToTensor, Normalize(0.5), (x + 1) / 2 * 255 -> uint8) and blend_to_text :247-291.
// Activations are NHWC; every kernel here is HBM-bound element-wise / reduction work: 16-byte chunks per lane, one pass.
#include <math.h>

#include <algorithm>

#include "common.h"

namespace {

typedef float float4v __attribute__((ext_vector_type(4)));

template <typename T>
struct Chunk;          // 8 consecutive channels
template <>
struct Chunk<_Float16> {
  typedef _Float16 v8 __attribute__((ext_vector_type(8)));
  static __device__ __forceinline__ void load(const _Float16* p, float (&f)[8]) {
    const v8 v = *(const v8*)p;
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = (float)v[i];
  }
  static __device__ __forceinline__ void store(_Float16* p, const float (&f)[8]) {
    v8 v;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (_Float16)f[i];
    *(v8*)p = v;
  }
};
template <>
struct Chunk<float> {
  static __device__ __forceinline__ void load(const float* p, float (&f)[8]) {
    const float4v a = *(const float4v*)p, b = *(const float4v*)(p + 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) { f[i] = a[i]; f[4 + i] = b[i]; }
  }
  static __device__ __forceinline__ void store(float* p, const float (&f)[8]) {
    *(float4v*)p = (float4v){f[0], f[1], f[2], f[3]};
    *(float4v*)(p + 4) = (float4v){f[4], f[5], f[6], f[7]};
  }
};

__device__ __forceinline__ float swishf(float x) { return x / (1.f + __expf(-x)); }
__device__ __forceinline__ int reflect(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }

// page u8 BGR [h][w][3] on a white H x W canvas -> x [H][W][4] = ((rgb / 255) - 0.5) / 0.5, fourth channel 0
template <typename T>
__global__ void ov_pre_kernel(const uint8_t* __restrict__ page, int h, int w, T* __restrict__ x, int H, int W) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= H * W) return;
  const int y = idx / W, xx = idx - y * W;
  float v[3] = {255.f, 255.f, 255.f};
  if (y < h && xx < w) {
    const uint8_t* p = page + ((size_t)y * w + xx) * 3;
    v[0] = p[2]; v[1] = p[1]; v[2] = p[0];
  }
  T* o = x + (size_t)idx * 4;
#pragma unroll
  for (int c = 0; c < 3; ++c) o[c] = (T)((v[c] / 255.f - 0.5f) / 0.5f);
  o[3] = (T)0.f;
}

// Direct convolution of a 3-channel image (stored with 4 channels): the 7x7 stems (3 -> ngf / 2 ngf, reflection padding) and
// the 3x3 / stride-2 input down-sampler (3 -> 3, zero padding).  wt fp32 [K*K*3][COUT] staged in LDS; one thread = one output
// pixel x COUT/4 channels (COUT >= 16), or one pixel x all channels (COUT = 3, written as 4).
template <typename T, int COUT>
__global__ __launch_bounds__(256) void ov_conv_c3_kernel(const T* __restrict__ in, const float* __restrict__ wt,
                                                        const float* __restrict__ bias, T* __restrict__ out, int H, int W, int Ho,
                                                        int Wo, int K, int stride, int pad, int refl) {
  extern __shared__ __attribute__((aligned(16))) float lw[];
  constexpr int GROUPS = COUT >= 16 ? 4 : 1, CG = COUT >= 16 ? COUT / 4 : 4, CS = COUT >= 16 ? COUT : 4;   // CS: staged row width
  const int taps3 = K * K * 3;
  for (int i = threadIdx.x; i < taps3 * CS; i += 256) {
    const int r = i / CS, c = i - r * CS;
    lw[i] = c < COUT ? wt[(size_t)r * COUT + c] : 0.f;
  }
  __syncthreads();
  const int pix = blockIdx.x * (256 / GROUPS) + threadIdx.x / GROUPS, cg = threadIdx.x % GROUPS;
  if (pix >= Ho * Wo) return;
  const int oy = pix / Wo, ox = pix - oy * Wo;
  float acc[CG];
#pragma unroll
  for (int c = 0; c < CG; ++c) acc[c] = (cg * CG + c) < COUT ? bias[cg * CG + c] : 0.f;
  for (int ky = 0; ky < K; ++ky) {
    int iy = oy * stride - pad + ky;
    if (refl) iy = reflect(iy, H);
    if ((unsigned)iy >= (unsigned)H) continue;
    for (int kx = 0; kx < K; ++kx) {
      int ix = ox * stride - pad + kx;
      if (refl) ix = reflect(ix, W);
      if ((unsigned)ix >= (unsigned)W) continue;
      const T* p = in + ((size_t)iy * W + ix) * 4;
      const float v[3] = {(float)p[0], (float)p[1], (float)p[2]};
      const float* wr = lw + (size_t)((ky * K + kx) * 3) * CS + cg * CG;
#pragma unroll
      for (int ci = 0; ci < 3; ++ci)
#pragma unroll
        for (int c = 0; c < CG; ++c) acc[c] += v[ci] * wr[ci * CS + c];
    }
  }
  T* o = out + (size_t)pix * CS + cg * CG;
#pragma unroll
  for (int c = 0; c < CG; ++c) o[c] = (T)acc[c];
}

// 7x7 stems on the matrix cores: patch matrix [H*W][192] (k = (ky*7 + kx)*3 + c for the 147 taps, zeros up to 192) of the
// reflection-padded 3-channel image; the convolution is then a GEMM with K = 192.  One thread = 8 consecutive k of one pixel.
template <typename T>
__global__ void ov_im2col7_kernel(const T* __restrict__ in, T* __restrict__ out, int H, int W) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)H * W * 24) return;
  const int j = (int)(idx % 24);
  const size_t px = idx / 24;
  const int y = (int)(px / W), x = (int)(px - (size_t)y * W);
  float f[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int k = j * 8 + e;
    f[e] = 0.f;
    if (k < 147) {
      const int tap = k / 3, c = k - tap * 3, ky = tap / 7, kx = tap - ky * 7;
      f[e] = (float)in[((size_t)reflect(y + ky - 3, H) * W + reflect(x + kx - 3, W)) * 4 + c];
    }
  }
  Chunk<T>::store(out + idx * 8, f);
}

// out[(y, x)] = in[reflect(y - p), reflect(x - p)], 8 channels per thread
template <typename T>
__global__ void ov_reflect_pad_kernel(const T* __restrict__ in, T* __restrict__ out, int H, int W, int C8, int p) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int Hp = H + 2 * p, Wp = W + 2 * p;
  if (idx >= (size_t)Hp * Wp * C8) return;
  const int c = (int)(idx % C8);
  const size_t px = idx / C8;
  const int y = (int)(px / Wp), x = (int)(px - (size_t)y * Wp);
  const int sy = reflect(y - p, H), sx = reflect(x - p, W);
  float f[8];
  Chunk<T>::load(in + (((size_t)sy * W + sx) * C8 + c) * 8, f);
  Chunk<T>::store(out + idx * 8, f);
}

// Per-channel statistics over the pixels (y, xo * cstep), xo < Wo of x [Ho][Wfull][C] in ONE pass: with k[c] = the channel's value
// at pixel 0 (a shift that keeps the sums small next to the spread), stats[c] += sum (x - k), stats[C + c] += sum (x - k)^2,
// stats[2 C + c] = k.  A grid-strided walk of at most 2048 workgroups: partials reduced through LDS, one atomicAdd per channel
// and workgroup (one per 256 pixels was 33 k atomics per address at full resolution: 75 ms per page).
template <typename T>
__global__ __launch_bounds__(256) void ov_colsum_kernel(const T* __restrict__ x, int Wfull, int Wo, int cstep, int C, long long P,
                                                       float* __restrict__ stats) {
  __shared__ float red[256 * 16];
  const int C8 = C >> 3;
  const int lanes_p = 256 / C8 > 0 ? 256 / C8 : 1;        // C <= 2048
  const int cc = threadIdx.x % C8, lp = threadIdx.x / C8;
  float s1[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, s2[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, k[8];
  if (lp < lanes_p) {
    Chunk<T>::load(x + (size_t)cc * 8, k);
    if (blockIdx.x == 0 && lp == 0)
#pragma unroll
      for (int i = 0; i < 8; ++i) stats[2 * C + cc * 8 + i] = k[i];
    for (long long p = (long long)blockIdx.x * lanes_p + lp; p < P; p += (long long)gridDim.x * lanes_p) {
      const long long y = p / Wo, xo = p - y * Wo;
      float f[8];
      Chunk<T>::load(x + ((size_t)(y * Wfull + xo * cstep) * C8 + cc) * 8, f);
#pragma unroll
      for (int i = 0; i < 8; ++i) { const float d = f[i] - k[i]; s1[i] += d; s2[i] += d * d; }
    }
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) { red[threadIdx.x * 16 + i] = s1[i]; red[threadIdx.x * 16 + 8 + i] = s2[i]; }
  __syncthreads();
  if (lp == 0) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      float t = 0.f;
      for (int l = 0; l < lanes_p; ++l) t += red[(l * C8 + cc) * 16 + i];
      atomicAdd(stats + (i < 8 ? 0 : C) + cc * 8 + (i & 7), t);
    }
  }
}

// y = (x - mean) * rsqrt(var + eps) [swish] [+ residual]; reads (y, xo * cstep) of x [Ho][Wfull][C], writes compact [Ho][Wo][C]
template <typename T>
__global__ void ov_in_apply_kernel(const T* __restrict__ x, int Wfull, int Wo, int cstep, int C, long long P,
                                   const float* __restrict__ stats, float eps, int act, const T* __restrict__ res, T* __restrict__ out) {
  const int C8 = C >> 3;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)P * C8) return;
  const int cc = (int)(idx % C8);
  const long long p = (long long)(idx / C8), y = p / Wo, xo = p - y * Wo;
  float f[8], r[8];
  Chunk<T>::load(x + ((size_t)(y * Wfull + xo * cstep) * C8 + cc) * 8, f);
  if (res) Chunk<T>::load(res + idx * 8, r);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float m1 = stats[cc * 8 + i] / (float)P, var = fmaxf(stats[C + cc * 8 + i] / (float)P - m1 * m1, 0.f);
    float v = (f[i] - (stats[2 * C + cc * 8 + i] + m1)) * (1.f / sqrtf(var + eps));
    if (act) v = swishf(v);
    if (res) v += r[i];
    f[i] = v;
  }
  Chunk<T>::store(out + idx * 8, f);
}

// nn.Upsample(scale_factor=2, mode="bilinear", align_corners=True), optional swish on the inputs
template <typename T>
__global__ void ov_upsample2x_kernel(const T* __restrict__ in, T* __restrict__ out, int H, int W, int C8, int act_in) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int Ho = 2 * H, Wo = 2 * W;
  if (idx >= (size_t)Ho * Wo * C8) return;
  const int c = (int)(idx % C8);
  const size_t px = idx / C8;
  const int oy = (int)(px / Wo), ox = (int)(px - (size_t)oy * Wo);
  const float sy = Ho > 1 ? (float)(H - 1) / (float)(Ho - 1) : 0.f, sx = Wo > 1 ? (float)(W - 1) / (float)(Wo - 1) : 0.f;
  const float fy = sy * oy, fx = sx * ox;
  const int y0 = (int)fy, x0 = (int)fx, y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
  const float ly1 = fy - y0, ly0 = 1.f - ly1, lx1 = fx - x0, lx0 = 1.f - lx1;
  float a[8], b[8], cc[8], d[8], o[8];
  Chunk<T>::load(in + (((size_t)y0 * W + x0) * C8 + c) * 8, a);
  Chunk<T>::load(in + (((size_t)y0 * W + x1) * C8 + c) * 8, b);
  Chunk<T>::load(in + (((size_t)y1 * W + x0) * C8 + c) * 8, cc);
  Chunk<T>::load(in + (((size_t)y1 * W + x1) * C8 + c) * 8, d);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    if (act_in) { a[i] = swishf(a[i]); b[i] = swishf(b[i]); cc[i] = swishf(cc[i]); d[i] = swishf(d[i]); }
    o[i] = ly0 * (lx0 * a[i] + lx1 * b[i]) + ly1 * (lx0 * cc[i] + lx1 * d[i]);
  }
  Chunk<T>::store(out + idx * 8, o);
}

// ConvTranspose2d(k 3, stride 2, padding 1, output_padding 1) as a plain 3x3 convolution over this image: the input at
// (1 + 2 y, 1 + 2 x) of a zeroed [2H + 2][2W + 2] canvas
template <typename T>
__global__ void ov_zero_insert_kernel(const T* __restrict__ in, T* __restrict__ out, int H, int W, int C8) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)H * W * C8) return;
  const int c = (int)(idx % C8);
  const size_t px = idx / C8;
  const int y = (int)(px / W), x = (int)(px - (size_t)y * W);
  float f[8];
  Chunk<T>::load(in + idx * 8, f);
  Chunk<T>::store(out + (((size_t)(1 + 2 * y) * (2 * W + 2) + 1 + 2 * x) * C8 + c) * 8, f);
}

// out = a + swish(b)
template <typename T>
__global__ void ov_add_swish_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ out, size_t n8) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n8) return;
  float x[8], y[8];
  Chunk<T>::load(a + idx * 8, x);
  Chunk<T>::load(b + idx * 8, y);
#pragma unroll
  for (int i = 0; i < 8; ++i) x[i] += swishf(y[i]);
  Chunk<T>::store(out + idx * 8, x);
}

// y [P][8] (3 valid channels) -> tanh -> ((t + 1) / 2 * 255) truncated -> u8 RGB [P][3]; raw (optional) fp32 [P][3]
template <typename T>
__global__ void ov_final_kernel(const T* __restrict__ y, uint8_t* __restrict__ rgb, float* __restrict__ raw, size_t P) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= P) return;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float t = tanhf((float)y[idx * 8 + c]);
    if (raw) raw[idx * 3 + c] = t;
    rgb[idx * 3 + c] = (uint8_t)((t + 1.f) / 2.0f * 255.0f);
  }
}

__device__ __forceinline__ int gray14(int c0, int c1, int c2) { return (c0 * 1868 + c1 * 9617 + c2 * 4899 + 8192) >> 14; }

// blend_to_text (overlay.py:247-291): `mask` is the generator's image in ITS channel order (RGB) and is read as if it were BGR,
// exactly as the reference does.  OpenCV 8-bit BGR2HSV (H in [0, 180)), inRange([0,137,216],[179,255,255]) inverted, BGR2GRAY.
__global__ void ov_blend_kernel(const uint8_t* __restrict__ real_bgr, const uint8_t* __restrict__ mask, uint8_t* __restrict__ out,
                                size_t P) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= P) return;
  const int b = mask[idx * 3], g = mask[idx * 3 + 1], r = mask[idx * 3 + 2];
  const int v = max(max(b, g), r), vmin = min(min(b, g), r), diff = v - vmin;
  const int sdiv = v ? (int)rintf((float)(255 << 12) / (float)v) : 0;
  const int s = (diff * sdiv + (1 << 11)) >> 12;
  // hue is only needed for the range test h <= 179, which always holds (h in [0, 180))
  const bool inrange = s >= 137 && v >= 216;
  const int red = inrange ? 0 : 255;
  const int gr = gray14(real_bgr[idx * 3], real_bgr[idx * 3 + 1], real_bgr[idx * 3 + 2]), gm = gray14(b, g, r);
  const uint8_t o = (uint8_t)((gr | gm) & red);
  out[idx * 3] = out[idx * 3 + 1] = out[idx * 3 + 2] = o;
}

template <typename F>
int run(mhip_ctx* ctx, const char* what, F&& f) {
  PROF_LAUNCH(ctx, MHIP_K_IMAGE_OPS, f());
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return mhip_fail(ctx, MHIP_EHIP, "%s launch: %s", what, hipGetErrorString(e));
  return 0;
}
inline unsigned blocks(size_t n, int t = 256) { return (unsigned)((n + t - 1) / t); }

}  // namespace

#define OV_DISPATCH(prec, CALL16, CALL32) ((prec) == MHIP_PREC_F16 ? (CALL16) : (CALL32))

int mhip_ov_preprocess(mhip_ctx* ctx, int prec, const uint8_t* page, int h, int w, void* x, int H, int W) {
  return run(ctx, "ov_pre", [&] {
    if (prec == MHIP_PREC_F16) hipLaunchKernelGGL((ov_pre_kernel<_Float16>), dim3(blocks((size_t)H * W)), dim3(256), 0, ctx->stream, page, h, w, (_Float16*)x, H, W);
    else hipLaunchKernelGGL((ov_pre_kernel<float>), dim3(blocks((size_t)H * W)), dim3(256), 0, ctx->stream, page, h, w, (float*)x, H, W);
  });
}

template <typename T, int COUT>
static int conv_c3_t(mhip_ctx* ctx, const void* in, const float* wt, const float* bias, void* out, int H, int W, int Ho, int Wo, int K,
                     int stride, int pad, int refl) {
  constexpr int CS = COUT >= 16 ? COUT : 4, GROUPS = COUT >= 16 ? 4 : 1;
  const size_t lds = (size_t)K * K * 3 * CS * 4;
  static std::once_flag attr;
  std::call_once(attr, [&] {
    (void)hipFuncSetAttribute((const void*)ov_conv_c3_kernel<T, COUT>, hipFuncAttributeMaxDynamicSharedMemorySize, 147 * 256 * 4);
  });
  return run(ctx, "ov_conv_c3", [&] {
    hipLaunchKernelGGL((ov_conv_c3_kernel<T, COUT>), dim3(blocks((size_t)Ho * Wo, 256 / GROUPS)), dim3(256), lds, ctx->stream, (const T*)in,
                       wt, bias, (T*)out, H, W, Ho, Wo, K, stride, pad, refl);
  });
}

// in [H][W][4]; wt fp32 [K*K*3][cout] (tap-major, then input channel); out [Ho][Wo][cout] (cout = 3: [Ho][Wo][4])
int mhip_ov_conv_c3(mhip_ctx* ctx, int prec, const void* in, const float* wt, const float* bias, void* out, int H, int W, int cout, int K,
                    int stride, int pad, int refl) {
  const int Ho = (H + 2 * pad - K) / stride + 1, Wo = (W + 2 * pad - K) / stride + 1;
  if (K > 7 || (refl && (pad >= H || pad >= W))) return mhip_fail(ctx, MHIP_EINVAL, "ov_conv_c3: bad geometry");
#define C3(T)                                                                                                        \
  switch (cout) {                                                                                                    \
    case 3: return conv_c3_t<T, 3>(ctx, in, wt, bias, out, H, W, Ho, Wo, K, stride, pad, refl);                      \
    case 32: return conv_c3_t<T, 32>(ctx, in, wt, bias, out, H, W, Ho, Wo, K, stride, pad, refl);                    \
    case 64: return conv_c3_t<T, 64>(ctx, in, wt, bias, out, H, W, Ho, Wo, K, stride, pad, refl);                    \
    case 128: return conv_c3_t<T, 128>(ctx, in, wt, bias, out, H, W, Ho, Wo, K, stride, pad, refl);                  \
    case 256: return conv_c3_t<T, 256>(ctx, in, wt, bias, out, H, W, Ho, Wo, K, stride, pad, refl);                  \
    default: return mhip_fail(ctx, MHIP_EINVAL, "ov_conv_c3: %d output channels", cout);                             \
  }
  if (prec == MHIP_PREC_F16) { C3(_Float16) }
  C3(float)
#undef C3
}

int mhip_ov_im2col7(mhip_ctx* ctx, int prec, const void* in, void* out, int H, int W) {
  if (H < 4 || W < 4) return mhip_fail(ctx, MHIP_EINVAL, "ov_im2col7: image smaller than the reflection frame");
  const size_t n = (size_t)H * W * 24;
  return run(ctx, "ov_im2col7", [&] {
    if (prec == MHIP_PREC_F16) hipLaunchKernelGGL((ov_im2col7_kernel<_Float16>), dim3(blocks(n)), dim3(256), 0, ctx->stream, (const _Float16*)in, (_Float16*)out, H, W);
    else hipLaunchKernelGGL((ov_im2col7_kernel<float>), dim3(blocks(n)), dim3(256), 0, ctx->stream, (const float*)in, (float*)out, H, W);
  });
}

#define OV_T(prec, KERNEL, GRID, ...)                                                                                           \
  do {                                                                                                                          \
    if ((prec) == MHIP_PREC_F16) hipLaunchKernelGGL((KERNEL<_Float16>), dim3(GRID), dim3(256), 0, ctx->stream, __VA_ARGS__);   \
    else hipLaunchKernelGGL((KERNEL<float>), dim3(GRID), dim3(256), 0, ctx->stream, __VA_ARGS__);                              \
  } while (0)

int mhip_ov_reflect_pad(mhip_ctx* ctx, int prec, const void* in, void* out, int H, int W, int C, int p) {
  if (C % 8 || p >= H || p >= W) return mhip_fail(ctx, MHIP_EINVAL, "ov_reflect_pad: bad shape");
  const size_t n = (size_t)(H + 2 * p) * (W + 2 * p) * (C / 8);
  return run(ctx, "ov_reflect_pad", [&] {
    if (prec == MHIP_PREC_F16) hipLaunchKernelGGL((ov_reflect_pad_kernel<_Float16>), dim3(blocks(n)), dim3(256), 0, ctx->stream, (const _Float16*)in, (_Float16*)out, H, W, C / 8, p);
    else hipLaunchKernelGGL((ov_reflect_pad_kernel<float>), dim3(blocks(n)), dim3(256), 0, ctx->stream, (const float*)in, (float*)out, H, W, C / 8, p);
  });
}

// InstanceNorm2d(affine=False, eps) [+ swish] [+ residual] over the pixels (y, xo * cstep) of x [Ho][Wfull][C] -> out [Ho][Wfull/cstep][C].
// stats: 3 C floats of scratch.
int mhip_ov_instance_norm(mhip_ctx* ctx, int prec, const void* x, int Ho, int Wfull, int cstep, int C, float eps, int swish,
                          const void* res, void* out, float* stats) {
  if (C % 8 || C > 2048 || cstep < 1 || Wfull % cstep) return mhip_fail(ctx, MHIP_EINVAL, "ov_instance_norm: bad shape");
  const int Wo = Wfull / cstep;
  const long long P = (long long)Ho * Wo;
  MHIP_HIP(ctx, hipMemsetAsync(stats, 0, (size_t)2 * C * 4, ctx->stream));
  const unsigned sgrid = (unsigned)std::min<size_t>(blocks((size_t)P), 2048);
  int rc = run(ctx, "ov_colsum", [&] {
    if (prec == MHIP_PREC_F16) hipLaunchKernelGGL((ov_colsum_kernel<_Float16>), dim3(sgrid), dim3(256), 0, ctx->stream, (const _Float16*)x, Wfull, Wo, cstep, C, P, stats);
    else hipLaunchKernelGGL((ov_colsum_kernel<float>), dim3(sgrid), dim3(256), 0, ctx->stream, (const float*)x, Wfull, Wo, cstep, C, P, stats);
  });
  if (rc) return rc;
  const size_t n = (size_t)P * (C / 8);
  return run(ctx, "ov_in_apply", [&] {
    if (prec == MHIP_PREC_F16) hipLaunchKernelGGL((ov_in_apply_kernel<_Float16>), dim3(blocks(n)), dim3(256), 0, ctx->stream, (const _Float16*)x, Wfull, Wo, cstep, C, P, stats, eps, swish, (const _Float16*)res, (_Float16*)out);
    else hipLaunchKernelGGL((ov_in_apply_kernel<float>), dim3(blocks(n)), dim3(256), 0, ctx->stream, (const float*)x, Wfull, Wo, cstep, C, P, stats, eps, swish, (const float*)res, (float*)out);
  });
}

int mhip_ov_upsample2x(mhip_ctx* ctx, int prec, const void* in, void* out, int H, int W, int C, int swish_in) {
  if (C % 8) return mhip_fail(ctx, MHIP_EINVAL, "ov_upsample2x: bad shape");
  const size_t n = (size_t)4 * H * W * (C / 8);
  return run(ctx, "ov_upsample2x", [&] {
    if (prec == MHIP_PREC_F16) hipLaunchKernelGGL((ov_upsample2x_kernel<_Float16>), dim3(blocks(n)), dim3(256), 0, ctx->stream, (const _Float16*)in, (_Float16*)out, H, W, C / 8, swish_in);
    else hipLaunchKernelGGL((ov_upsample2x_kernel<float>), dim3(blocks(n)), dim3(256), 0, ctx->stream, (const float*)in, (float*)out, H, W, C / 8, swish_in);
  });
}

int mhip_ov_zero_insert(mhip_ctx* ctx, int prec, const void* in, void* out, int H, int W, int C) {
  if (C % 8) return mhip_fail(ctx, MHIP_EINVAL, "ov_zero_insert: bad shape");
  const size_t es = prec == MHIP_PREC_F16 ? 2 : 4, n = (size_t)H * W * (C / 8);
  MHIP_HIP(ctx, hipMemsetAsync(out, 0, (size_t)(2 * H + 2) * (2 * W + 2) * C * es, ctx->stream));
  return run(ctx, "ov_zero_insert", [&] {
    if (prec == MHIP_PREC_F16) hipLaunchKernelGGL((ov_zero_insert_kernel<_Float16>), dim3(blocks(n)), dim3(256), 0, ctx->stream, (const _Float16*)in, (_Float16*)out, H, W, C / 8);
    else hipLaunchKernelGGL((ov_zero_insert_kernel<float>), dim3(blocks(n)), dim3(256), 0, ctx->stream, (const float*)in, (float*)out, H, W, C / 8);
  });
}

int mhip_ov_add_swish(mhip_ctx* ctx, int prec, const void* a, const void* b, void* out, size_t n) {
  if (n % 8) return mhip_fail(ctx, MHIP_EINVAL, "ov_add_swish: bad size");
  return run(ctx, "ov_add_swish", [&] {
    if (prec == MHIP_PREC_F16) hipLaunchKernelGGL((ov_add_swish_kernel<_Float16>), dim3(blocks(n / 8)), dim3(256), 0, ctx->stream, (const _Float16*)a, (const _Float16*)b, (_Float16*)out, n / 8);
    else hipLaunchKernelGGL((ov_add_swish_kernel<float>), dim3(blocks(n / 8)), dim3(256), 0, ctx->stream, (const float*)a, (const float*)b, (float*)out, n / 8);
  });
}

int mhip_ov_final(mhip_ctx* ctx, int prec, const void* y, uint8_t* rgb, float* raw, size_t P) {
  return run(ctx, "ov_final", [&] {
    if (prec == MHIP_PREC_F16) hipLaunchKernelGGL((ov_final_kernel<_Float16>), dim3(blocks(P)), dim3(256), 0, ctx->stream, (const _Float16*)y, rgb, raw, P);
    else hipLaunchKernelGGL((ov_final_kernel<float>), dim3(blocks(P)), dim3(256), 0, ctx->stream, (const float*)y, rgb, raw, P);
  });
}

int mhip_ov_blend(mhip_ctx* ctx, const uint8_t* real_bgr, const uint8_t* mask, uint8_t* out, size_t P) {
  return run(ctx, "ov_blend", [&] { hipLaunchKernelGGL(ov_blend_kernel, dim3(blocks(P)), dim3(256), 0, ctx->stream, real_bgr, mask, out, P); });
}
