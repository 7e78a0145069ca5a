// conv_first.hip — first recognizer layer, fused end to end:
//   uint8 crop -> /255 -> (x-0.5)/0.5 -> conv3x3 pad 1 (1 -> 64) + bias -> ReLU -> maxpool 2x2 -> NHWC
//
// Replaces: ToTensor + sub_(0.5).div_(0.5) (marie/models/icr/dataset.py:275-283) and
// ConvNet[0..2] (marie/models/icr/modules/feature_extraction.py:13-14).
//
// K = 9 is far too thin for the matrix cores, so this is a VALU kernel: a thread owns one pooled
// output pixel x 16 channels, keeps its 4x4 input patch in registers, reads the 9x64 filter bank
// from LDS (wave-broadcast) and writes 32 B (f16) / 64 B (f32) of contiguous NHWC output; the four
// threads of a pixel cover its full 64-channel line, so stores are fully coalesced.  The fp32
// normalised image never exists in HBM.
#include "common.h"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float4v __attribute__((ext_vector_type(4)));

template <typename T>
__global__ __launch_bounds__(256) void conv_first_kernel(const uint8_t* __restrict__ crops,
                                                         const float* __restrict__ w9x64,
                                                         const float* __restrict__ bias64, T* __restrict__ out,
                                                         int B, int H, int W) {
  __shared__ float sw[9 * 64 + 64];
  for (int i = threadIdx.x; i < 9 * 64; i += 256) sw[i] = w9x64[i];
  if (threadIdx.x < 64) sw[576 + threadIdx.x] = bias64[threadIdx.x];
  __syncthreads();

  const int Hp = H >> 1, Wp = W >> 1;
  const long long npix = (long long)B * Hp * Wp;
  const long long pp = (long long)blockIdx.x * 64 + (threadIdx.x >> 2);
  if (pp >= npix) return;
  const int cg = threadIdx.x & 3;
  const int xp = (int)(pp % Wp);
  const long long r = pp / Wp;
  const int yp = (int)(r % Hp);
  const int b = (int)(r / Hp);

  // 4x4 normalised input patch around the 2x2 pooling window (zero outside the image:
  // nn.Conv2d pads the *normalised* tensor with 0.0)
  float patch[4][4];
  const uint8_t* img = crops + (size_t)b * H * W;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int y = 2 * yp - 1 + i;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int x = 2 * xp - 1 + j;
      float v = 0.f;
      if ((unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W) {
        v = (float)img[(size_t)y * W + x] / 255.0f;
        v = (v - 0.5f) / 0.5f;
      }
      patch[i][j] = v;
    }
  }

  float o[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) {
    const int ch = cg * 16 + c;
    float wk[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) wk[k] = sw[k * 64 + ch];
    float best = -3.0e38f;
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) {
        float s = 0.f;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) s = fmaf(patch[dy + ky][dx + kx], wk[ky * 3 + kx], s);
        best = fmaxf(best, s);
      }
    o[c] = fmaxf(best + sw[576 + ch], 0.f);
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
    for (int q = 0; q < 4; ++q)
      ((float4v*)dst)[q] = (float4v){o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]};
  }
}

}  // namespace

int mhip_launch_conv_first(mhip_ctx* ctx, int precision, const uint8_t* crops, const float* w9x64,
                           const float* bias64, void* out, int B, int H, int W) {
  if (H < 2 || W < 2 || B < 1) return mhip_fail(ctx, MHIP_EINVAL, "conv_first: bad shape %dx%dx%d", B, H, W);
  long long npix = (long long)B * (H / 2) * (W / 2);
  unsigned grid = (unsigned)((npix + 63) / 64);
  if (precision == MHIP_PREC_F16) {
    PROF_LAUNCH(ctx, MHIP_K_CONV_FIRST,
                hipLaunchKernelGGL((conv_first_kernel<_Float16>), dim3(grid), dim3(256), 0, ctx->stream, crops,
                                   w9x64, bias64, (_Float16*)out, B, H, W));
  } else {
    PROF_LAUNCH(ctx, MHIP_K_CONV_FIRST,
                hipLaunchKernelGGL((conv_first_kernel<float>), dim3(grid), dim3(256), 0, ctx->stream, crops, w9x64,
                                   bias64, (float*)out, B, H, W));
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return mhip_fail(ctx, MHIP_EHIP, "conv_first launch: %s", hipGetErrorString(e));
  return 0;
}
