// image_ops.hip — the HBM-bound image / tensor kernels of the detector path (all NHWC, 16 B per lane).
//
//   resize_linear_u8      cv2.resize(u8, INTER_LINEAR)                 marie/models/craft/imgproc.py:58
//   conv_rgb_first        normalizeMeanVariance + canvas pad + conv1_1  imgproc.py:26-33,61-66; basenet/vgg16_bn.py:29 (features[0..2])
//   maxpool2x2 / 3x3s1    nn.MaxPool2d                                  basenet/vgg16_bn.py:27-43 (features "M", slice5[0])
//   upsample_bilinear     F.interpolate(mode='bilinear', align_corners=False)   marie/models/craft/craft.py:67-75
//
// None of these has reuse worth an LDS tile beyond a filter bank; they are written for coalescing:
// a lane owns 8 (f16) / 4 (f32) consecutive channels of one pixel = one 16-byte access.
#include <math.h>

#include "common.h"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float4v __attribute__((ext_vector_type(4)));

template <typename T>
struct Vec;
template <>
struct Vec<_Float16> {
  typedef half8 type;
  static constexpr int N = 8;
};
template <>
struct Vec<float> {
  typedef float4v type;
  static constexpr int N = 4;
};

// ------------------------------------------------------------------ cv2.resize INTER_LINEAR, 8-bit
// OpenCV's 8-bit bilinear is fixed point: 11-bit coefficients (built on the host with OpenCV's own float
// arithmetic), int horizontal pass, then (((b0*(S0>>4))>>16) + ((b1*(S1>>4))>>16) + 2) >> 2.
__global__ __launch_bounds__(256) void resize_linear_u8_kernel(const uint8_t* __restrict__ src, int sh, int sw,
                                                               uint8_t* __restrict__ dst, int dh, int dw,
                                                               const int* __restrict__ xofs,
                                                               const short* __restrict__ xa,
                                                               const int* __restrict__ yofs,
                                                               const short* __restrict__ yb) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y;
  if (x >= dw) return;
  const int x0 = xofs[x], x1 = min(x0 + 1, sw - 1);
  const int a0 = xa[2 * x], a1 = xa[2 * x + 1];
  const int y0 = yofs[y], y1 = min(y0 + 1, sh - 1);
  const int b0 = yb[2 * y], b1 = yb[2 * y + 1];
  const uint8_t* r0 = src + (size_t)y0 * sw * 3;
  const uint8_t* r1 = src + (size_t)y1 * sw * 3;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const int s0 = r0[x0 * 3 + c] * a0 + r0[x1 * 3 + c] * a1;
    const int s1 = r1[x0 * 3 + c] * a0 + r1[x1 * 3 + c] * a1;
    int v = (((b0 * (s0 >> 4)) >> 16) + ((b1 * (s1 >> 4)) >> 16) + 2) >> 2;
    dst[((size_t)y * dw + x) * 3 + c] = (uint8_t)min(max(v, 0), 255);
  }
}

// ------------------------------------------------------------------ first detector layer
// uint8 [th][tw][3] placed top-left on an [H][W] canvas -> (v-127.5)/127.5 (canvas padding is 0 BEFORE the
// normalisation, i.e. -1 after it) -> conv3x3 pad 1 (3 -> 64) -> scale/shift (folded BatchNorm) -> ReLU -> NHWC.
template <typename T>
__global__ __launch_bounds__(256) void conv_rgb_first_kernel(const uint8_t* __restrict__ img, int th, int tw,
                                                             int H, int W, const float* __restrict__ w27x64,
                                                             const float* __restrict__ scale,
                                                             const float* __restrict__ bias, T* __restrict__ out) {
  __shared__ float sw[27 * 64 + 128];
  for (int i = threadIdx.x; i < 27 * 64; i += 256) sw[i] = w27x64[i];
  if (threadIdx.x < 64) {
    sw[27 * 64 + threadIdx.x] = scale[threadIdx.x];
    sw[27 * 64 + 64 + threadIdx.x] = bias[threadIdx.x];
  }
  __syncthreads();
  const long long npix = (long long)H * W;
  const long long pp = (long long)blockIdx.x * 64 + (threadIdx.x >> 2);
  if (pp >= npix) return;
  const int cg = threadIdx.x & 3;
  const int x = (int)(pp % W), y = (int)(pp / W);
  float patch[27];
#pragma unroll
  for (int dy = 0; dy < 3; ++dy)
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      const int yy = y + dy - 1, xx = x + dx - 1;
      const bool in_canvas = (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
      const bool in_img = in_canvas && yy < th && xx < tw;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        float v = 0.f;                                       // conv zero padding (outside the canvas)
        if (in_canvas) {
          float u = in_img ? (float)img[((size_t)yy * tw + xx) * 3 + c] : 0.f;
          v = (u - 127.5f) / 127.5f;
        }
        patch[(dy * 3 + dx) * 3 + c] = v;
      }
    }
  float o[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) {
    const int ch = cg * 16 + c;
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 27; ++k) s = fmaf(patch[k], sw[k * 64 + ch], s);
    o[c] = fmaxf(s * sw[27 * 64 + ch] + sw[27 * 64 + 64 + ch], 0.f);
  }
  T* dst = out + (size_t)pp * 64 + cg * 16;
  if (sizeof(T) == 2) {
    half8 v0, v1;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      v0[c] = (_Float16)o[c];
      v1[c] = (_Float16)o[8 + c];
    }
    ((half8*)dst)[0] = v0;
    ((half8*)dst)[1] = v1;
  } else {
#pragma unroll
    for (int q = 0; q < 4; ++q) ((float4v*)dst)[q] = (float4v){o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]};
  }
}

// f16 mode: the same layer on the matrix cores.  K = 27 is padded to one 32-deep MFMA step
// (v_mfma_f32_16x16x32_f16).  The FILTERS are the A operand (rows = output channels, constant in registers for the
// whole kernel) and 16 PIXELS are the B operand, so in the C layout a lane ends up with 4 consecutive channels of one
// pixel: the 64-channel NHWC line of a pixel is written as 8-byte pieces that tile 2 KiB contiguously per wave.
// The im2col row of a pixel is 3 runs of 9 contiguous bytes (k = dy*9 + dx*3 + c), gathered per lane straight from
// the uint8 image; the kernel is bound by the 650 MB/page activation write, not by arithmetic.
typedef _Float16 half4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void conv_rgb_first_mfma_kernel(const uint8_t* __restrict__ img, int th, int tw,
                                                                  int H, int W, const float* __restrict__ w27x64,
                                                                  const float* __restrict__ scale,
                                                                  const float* __restrict__ bias,
                                                                  _Float16* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int frow = lane & 15, fg = lane >> 4;
  // A fragments: filter bank rows (channels) t*16 + frow, k = 8*fg + j  (zero for k >= 27)
  half8 wf[4];
  float sc[4][4], bi[4][4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = 8 * fg + j;
      wf[t][j] = (k < 27) ? (_Float16)w27x64[k * 64 + t * 16 + frow] : (_Float16)0.f;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {   // C rows of this lane: channels t*16 + fg*4 + r
      sc[t][r] = scale[t * 16 + fg * 4 + r];
      bi[t][r] = bias[t * 16 + fg * 4 + r];
    }
  }
  const long long npix = (long long)H * W;
  const long long ntiles = (npix + 15) / 16;
  const long long wave_id = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
  for (long long tile = wave_id; tile < ntiles; tile += nwaves) {
    const long long pp = tile * 16 + frow;        // this lane's pixel (B column)
    const int x = (int)(pp % W), y = (int)(pp / W);
    half8 bf;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = 8 * fg + j;                   // k = dy*9 + dx*3 + c
      const int dy = k / 9, rem = k - dy * 9, dx = rem / 3, c = rem - dx * 3;
      float v = 0.f;
      if (k < 27 && pp < npix) {
        const int yy = y + dy - 1, xx = x + dx - 1;
        if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) {
          const float u = (yy < th && xx < tw) ? (float)img[((size_t)yy * tw + xx) * 3 + c] : 0.f;
          v = (u - 127.5f) / 127.5f;
        }
      }
      bf[j] = (_Float16)v;
    }
    float4v acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[t], bf, (float4v){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
    if (pp < npix) {
      _Float16* dst = out + (size_t)pp * 64 + fg * 4;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        half4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (_Float16)fmaxf(acc[t][r] * sc[t][r] + bi[t][r], 0.f);
        *(half4*)(dst + t * 16) = o;
      }
    }
  }
}

// ------------------------------------------------------------------ detector score head
// conv_cls[6..8]: ReLU(conv1x1 16->16) then conv1x1 16->2 (marie/models/craft/craft.py:46-49), fused per pixel.
// 288 MACs per pixel are nothing; the layer is the read of 16 real channels (stored in a 64-channel padded NHWC
// line) and the write of two fp32 scores.  Keeping it off the 128-wide MFMA tile saves two launches that ran at
// < 2 % tile occupancy.  Weights: w1 [16][16], b1 [16], w2 [2][16], b2 [2] fp32 in LDS.
template <typename T>
__global__ __launch_bounds__(256) void score_head_kernel(const T* __restrict__ in, int in_stride,
                                                         const float* __restrict__ w1, const float* __restrict__ b1,
                                                         const float* __restrict__ w2, const float* __restrict__ b2,
                                                         float* __restrict__ scores, long long npix) {
  __shared__ float s1[16 * 16 + 16 + 2 * 16 + 2];
  for (int i = threadIdx.x; i < 256; i += 256) s1[i] = w1[i];
  if (threadIdx.x < 16) s1[256 + threadIdx.x] = b1[threadIdx.x];
  if (threadIdx.x < 32) s1[272 + threadIdx.x] = w2[threadIdx.x];
  if (threadIdx.x < 2) s1[304 + threadIdx.x] = b2[threadIdx.x];
  __syncthreads();
  typedef typename Vec<T>::type V;
  constexpr int N = Vec<T>::N;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < npix;
       i += (long long)gridDim.x * blockDim.x) {
    float x[16];
    const T* p = in + (size_t)i * in_stride;
#pragma unroll
    for (int q = 0; q < 16 / N; ++q) {
      const V v = *(const V*)(p + q * N);
#pragma unroll
      for (int k = 0; k < N; ++k) x[q * N + k] = (float)v[k];
    }
    float o0 = s1[304], o1 = s1[305];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      float h = s1[256 + j];
#pragma unroll
      for (int k = 0; k < 16; ++k) h = fmaf(x[k], s1[j * 16 + k], h);
      h = fmaxf(h, 0.f);
      if (sizeof(T) == 2) h = (float)(_Float16)h;   // the unfused f16 path rounds this activation to f16
      o0 = fmaf(h, s1[272 + j], o0);
      o1 = fmaf(h, s1[288 + j], o1);
    }
    ((float2*)scores)[i] = make_float2(o0, o1);
  }
}

// ------------------------------------------------------------------ max pooling
template <typename T>
__device__ __forceinline__ typename Vec<T>::type vmax(typename Vec<T>::type a, typename Vec<T>::type b) {
  typename Vec<T>::type r;
#pragma unroll
  for (int i = 0; i < Vec<T>::N; ++i) r[i] = a[i] > b[i] ? a[i] : b[i];
  return r;
}

// K = 2: 2x2 stride 2 (floor);  K = 3: 3x3 stride 1 pad 1 (padding never wins)
template <typename T, int K>
__global__ __launch_bounds__(256) void maxpool_kernel(const T* __restrict__ in, T* __restrict__ out, int B, int H,
                                                      int W, int C, int Ho, int Wo) {
  typedef typename Vec<T>::type V;
  constexpr int N = Vec<T>::N;
  const int cv = C / N;
  const long long total = (long long)B * Ho * Wo * cv;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % cv);
    long long r = i / cv;
    const int xo = (int)(r % Wo);
    r /= Wo;
    const int yo = (int)(r % Ho);
    const int b = (int)(r / Ho);
    V m;
    bool have = false;
#pragma unroll
    for (int dy = 0; dy < K; ++dy)
#pragma unroll
      for (int dx = 0; dx < K; ++dx) {
        const int y = (K == 2) ? 2 * yo + dy : yo + dy - 1;
        const int x = (K == 2) ? 2 * xo + dx : xo + dx - 1;
        if ((unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W) {
          V v = *(const V*)(in + (((size_t)b * H + y) * W + x) * C + (size_t)c * N);
          m = have ? vmax<T>(m, v) : v;
          have = true;
        }
      }
    *(V*)(out + (((size_t)b * Ho + yo) * Wo + xo) * C + (size_t)c * N) = m;
  }
}

// ------------------------------------------------------------------ bilinear up-sampling (align_corners = False)
// PyTorch's upsample_bilinear2d: src = scale*(dst+0.5)-0.5 clamped at 0 (fp32), i1 = min(i0+1, in-1),
// out = h0*(w0*v00 + w1*v01) + h1*(w0*v10 + w1*v11) in fp32.
template <typename T>
__global__ __launch_bounds__(256) void upsample_bilinear_kernel(const T* __restrict__ in, T* __restrict__ out, int B,
                                                                int Hi, int Wi, int C, int Ho, int Wo, float sh,
                                                                float sw) {
  typedef typename Vec<T>::type V;
  constexpr int N = Vec<T>::N;
  const int cv = C / N;
  const long long total = (long long)B * Ho * Wo * cv;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % cv);
    long long r = i / cv;
    const int xo = (int)(r % Wo);
    r /= Wo;
    const int yo = (int)(r % Ho);
    const int b = (int)(r / Ho);
    float fy = sh * ((float)yo + 0.5f) - 0.5f;
    fy = fy < 0.f ? 0.f : fy;
    float fx = sw * ((float)xo + 0.5f) - 0.5f;
    fx = fx < 0.f ? 0.f : fx;
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + ((y0 < Hi - 1) ? 1 : 0), x1 = x0 + ((x0 < Wi - 1) ? 1 : 0);
    const float h1 = fy - (float)y0, h0 = 1.f - h1;
    const float w1 = fx - (float)x0, w0 = 1.f - w1;
    const T* base = in + (size_t)b * Hi * Wi * C + (size_t)c * N;
    const V v00 = *(const V*)(base + ((size_t)y0 * Wi + x0) * C);
    const V v01 = *(const V*)(base + ((size_t)y0 * Wi + x1) * C);
    const V v10 = *(const V*)(base + ((size_t)y1 * Wi + x0) * C);
    const V v11 = *(const V*)(base + ((size_t)y1 * Wi + x1) * C);
    V o;
#pragma unroll
    for (int k = 0; k < N; ++k)
      o[k] = (T)(h0 * (w0 * (float)v00[k] + w1 * (float)v01[k]) + h1 * (w0 * (float)v10[k] + w1 * (float)v11[k]));
    *(V*)(out + (((size_t)b * Ho + yo) * Wo + xo) * C + (size_t)c * N) = o;
  }
}

unsigned grid_for(long long total) {
  long long g = (total + 255) / 256;
  return (unsigned)(g < 1 ? 1 : (g > 256 * 32 ? 256 * 32 : g));
}

}  // namespace

// Host side of cv2.resize's coefficient tables (same float arithmetic as OpenCV's resize.cpp).
void mhip_resize_linear_tables(int src, int dst, std::vector<int>& ofs, std::vector<short>& coef) {
  ofs.resize(dst);
  coef.resize(2 * (size_t)dst);
  const double scale = (double)src / (double)dst;
  for (int d = 0; d < dst; ++d) {
    float f = (float)((d + 0.5) * scale - 0.5);
    int s = (int)floorf(f);
    f -= (float)s;
    if (s < 0) {
      f = 0.f;
      s = 0;
    }
    if (s >= src - 1) {
      f = 0.f;
      s = src - 1;
    }
    ofs[d] = s;
    long a0 = lrintf((1.f - f) * 2048.f), a1 = lrintf(f * 2048.f);
    coef[2 * d] = (short)(a0 > 32767 ? 32767 : (a0 < -32768 ? -32768 : a0));
    coef[2 * d + 1] = (short)(a1 > 32767 ? 32767 : (a1 < -32768 ? -32768 : a1));
  }
}

int mhip_launch_resize_linear_u8(mhip_ctx* ctx, const uint8_t* src, int sh, int sw, uint8_t* dst, int dh, int dw,
                                 const int* xofs, const short* xa, const int* yofs, const short* yb) {
  if (sh < 1 || sw < 1 || dh < 1 || dw < 1) return mhip_fail(ctx, MHIP_EINVAL, "resize: bad shape");
  dim3 grid((dw + 255) / 256, dh), block(256);
  PROF_LAUNCH(ctx, MHIP_K_IMAGE_OPS,
              hipLaunchKernelGGL(resize_linear_u8_kernel, grid, block, 0, ctx->stream, src, sh, sw, dst, dh, dw, xofs,
                                 xa, yofs, yb));
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return mhip_fail(ctx, MHIP_EHIP, "resize launch: %s", hipGetErrorString(e));
  return 0;
}

int mhip_launch_conv_rgb_first(mhip_ctx* ctx, int precision, const uint8_t* img, int th, int tw, int H, int W,
                               const float* w27x64, const float* scale, const float* bias, void* out) {
  if (th < 1 || tw < 1 || H < th || W < tw) return mhip_fail(ctx, MHIP_EINVAL, "conv_rgb_first: bad shape");
  unsigned grid = (unsigned)(((long long)H * W + 63) / 64);
  if (precision == MHIP_PREC_F16) {
    const long long tiles = ((long long)H * W + 15) / 16;
    const unsigned g2 = (unsigned)std::min<long long>((tiles + 3) / 4, 256 * 16);
    PROF_LAUNCH(ctx, MHIP_K_CONV_FIRST,
                hipLaunchKernelGGL(conv_rgb_first_mfma_kernel, dim3(g2), dim3(256), 0, ctx->stream, img, th, tw, H, W,
                                   w27x64, scale, bias, (_Float16*)out));
  } else {
    PROF_LAUNCH(ctx, MHIP_K_CONV_FIRST,
                hipLaunchKernelGGL((conv_rgb_first_kernel<float>), dim3(grid), dim3(256), 0, ctx->stream, img, th, tw,
                                   H, W, w27x64, scale, bias, (float*)out));
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return mhip_fail(ctx, MHIP_EHIP, "conv_rgb_first launch: %s", hipGetErrorString(e));
  return 0;
}

int mhip_launch_maxpool(mhip_ctx* ctx, int precision, int k, const void* in, void* out, int B, int H, int W, int C) {
  const int vn = precision == MHIP_PREC_F16 ? 8 : 4;
  if ((k != 2 && k != 3) || B < 1 || H < 1 || W < 1 || C % vn) return mhip_fail(ctx, MHIP_EINVAL, "maxpool: bad args");
  const int Ho = k == 2 ? H / 2 : H, Wo = k == 2 ? W / 2 : W;
  if (Ho < 1 || Wo < 1) return mhip_fail(ctx, MHIP_EINVAL, "maxpool: empty output");
  const unsigned grid = grid_for((long long)B * Ho * Wo * (C / vn));
#define MP(T, K)                                                                                               \
  PROF_LAUNCH(ctx, MHIP_K_IMAGE_OPS,                                                                           \
              hipLaunchKernelGGL((maxpool_kernel<T, K>), dim3(grid), dim3(256), 0, ctx->stream, (const T*)in, \
                                 (T*)out, B, H, W, C, Ho, Wo))
  if (precision == MHIP_PREC_F16) {
    if (k == 2) MP(_Float16, 2); else MP(_Float16, 3);
  } else {
    if (k == 2) MP(float, 2); else MP(float, 3);
  }
#undef MP
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return mhip_fail(ctx, MHIP_EHIP, "maxpool launch: %s", hipGetErrorString(e));
  return 0;
}

int mhip_launch_upsample_bilinear(mhip_ctx* ctx, int precision, const void* in, void* out, int B, int Hi, int Wi,
                                  int C, int Ho, int Wo) {
  const int vn = precision == MHIP_PREC_F16 ? 8 : 4;
  if (B < 1 || Hi < 1 || Wi < 1 || Ho < 1 || Wo < 1 || C % vn)
    return mhip_fail(ctx, MHIP_EINVAL, "upsample: bad args");
  // area_pixel_compute_scale(align_corners = false, no explicit scale_factor): in / out, in fp32
  const float sh = (float)Hi / (float)Ho, sw = (float)Wi / (float)Wo;
  const unsigned grid = grid_for((long long)B * Ho * Wo * (C / vn));
  if (precision == MHIP_PREC_F16) {
    PROF_LAUNCH(ctx, MHIP_K_IMAGE_OPS,
                hipLaunchKernelGGL((upsample_bilinear_kernel<_Float16>), dim3(grid), dim3(256), 0, ctx->stream,
                                   (const _Float16*)in, (_Float16*)out, B, Hi, Wi, C, Ho, Wo, sh, sw));
  } else {
    PROF_LAUNCH(ctx, MHIP_K_IMAGE_OPS,
                hipLaunchKernelGGL((upsample_bilinear_kernel<float>), dim3(grid), dim3(256), 0, ctx->stream,
                                   (const float*)in, (float*)out, B, Hi, Wi, C, Ho, Wo, sh, sw));
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return mhip_fail(ctx, MHIP_EHIP, "upsample launch: %s", hipGetErrorString(e));
  return 0;
}

int mhip_launch_score_head(mhip_ctx* ctx, int precision, const void* in, int in_stride, const float* w1,
                           const float* b1, const float* w2, const float* b2, float* scores, long long npix) {
  if (npix < 1 || in_stride < 16) return mhip_fail(ctx, MHIP_EINVAL, "score_head: bad shape");
  const unsigned grid = grid_for(npix);
  if (precision == MHIP_PREC_F16) {
    PROF_LAUNCH(ctx, MHIP_K_IMAGE_OPS,
                hipLaunchKernelGGL((score_head_kernel<_Float16>), dim3(grid), dim3(256), 0, ctx->stream,
                                   (const _Float16*)in, in_stride, w1, b1, w2, b2, scores, npix));
  } else {
    PROF_LAUNCH(ctx, MHIP_K_IMAGE_OPS,
                hipLaunchKernelGGL((score_head_kernel<float>), dim3(grid), dim3(256), 0, ctx->stream, (const float*)in,
                                   in_stride, w1, b1, w2, b2, scores, npix));
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return mhip_fail(ctx, MHIP_EHIP, "score_head launch: %s", hipGetErrorString(e));
  return 0;
}
