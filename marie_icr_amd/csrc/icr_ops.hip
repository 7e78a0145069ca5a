// icr_ops.hip — the non-GEMM kernels of the production recognizer TPS-ResNet-BiLSTM-Attn.
//
//   conv_gray_first    nn.Conv2d(1, C, 3, 1, 1, bias=False)+BatchNorm2d+ReLU on a 1-channel image
//                      (marie/models/icr/modules/transformation.py:52-53 C=64; feature_extraction.py:161-163 C=32)
//   maxpool_s21_p01    nn.MaxPool2d(kernel_size=2, stride=(2,1), padding=(0,1))   feature_extraction.py:180
//   avgpool_hw         nn.AdaptiveAvgPool2d(1)                                    transformation.py:61
//   tps_sample         GridGenerator.build_P_prime + F.grid_sample(border, align_corners=True)
//                      (transformation.py:32-42,158-167)
//   attn_context       AttentionCell: e = score(tanh(i2h(H) + h2h(h))), alpha = softmax_T(e), context = alpha^T H
//                      (marie/models/icr/modules/prediction.py:72-79)
//   attn_cell          nn.LSTMCell on [context, one_hot(char)] + state update (prediction.py:80-82)
//   argmax_rows        `_, next_input = probs_step.max(1)` (prediction.py:57-59)
// All are HBM / latency bound; the GEMM-shaped parts of the same modules run on conv_igemm.
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

__device__ __forceinline__ float norm_u8(uint8_t v) {
  float f = (float)v / 255.0f;     // ToTensor
  return (f - 0.5f) / 0.5f;        // sub_(0.5).div_(0.5)
}

// ------------------------------------------------------------------ 1 -> C 3x3 conv + scale/shift + ReLU
// IN = uint8_t: the crop, normalised on the fly;  IN = float: an fp32 image (the TPS-rectified crop).
// out: NHWC with 64 channels (channels >= C are written as zero so the next layer sees a full 128-byte K slice).
template <typename T, typename IN>
__global__ __launch_bounds__(256) void conv_gray_first_kernel(const IN* __restrict__ img, int B, int H, int W, int C,
                                                              const float* __restrict__ w9xC,
                                                              const float* __restrict__ scale,
                                                              const float* __restrict__ bias, T* __restrict__ out) {
  __shared__ float sw[9 * 64 + 128];
  for (int i = threadIdx.x; i < 9 * 64; i += 256) sw[i] = (i % 64 < C) ? w9xC[(i / 64) * C + (i % 64)] : 0.f;
  if (threadIdx.x < 64) {
    sw[576 + threadIdx.x] = threadIdx.x < C ? scale[threadIdx.x] : 0.f;
    sw[640 + threadIdx.x] = threadIdx.x < C ? bias[threadIdx.x] : 0.f;
  }
  __syncthreads();
  const long long npix = (long long)B * H * W;
  const long long pp = (long long)blockIdx.x * 64 + (threadIdx.x >> 2);
  if (pp >= npix) return;
  const int cg = threadIdx.x & 3;
  const int x = (int)(pp % W);
  const long long r = pp / W;
  const int y = (int)(r % H), b = (int)(r / H);
  const IN* im = img + (size_t)b * H * W;
  float patch[9];
#pragma unroll
  for (int dy = 0; dy < 3; ++dy)
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      const int yy = y + dy - 1, xx = x + dx - 1;
      float v = 0.f;
      if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) {
        if (sizeof(IN) == 1) v = norm_u8((uint8_t)im[(size_t)yy * W + xx]);
        else v = (float)im[(size_t)yy * W + xx];
      }
      patch[dy * 3 + dx] = v;
    }
  float o[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) {
    const int ch = cg * 16 + c;
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 9; ++k) s = fmaf(patch[k], sw[k * 64 + ch], s);
    o[c] = fmaxf(s * sw[576 + ch] + sw[640 + ch], 0.f);
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

// ------------------------------------------------------------------ MaxPool2d(2, stride (2,1), padding (0,1))
template <typename T>
__global__ __launch_bounds__(256) void maxpool_s21_p01_kernel(const T* __restrict__ in, T* __restrict__ out, int B,
                                                              int H, int W, int C) {
  typedef typename Vec<T>::type V;
  constexpr int N = Vec<T>::N;
  const int cv = C / N, Ho = H / 2, Wo = W + 1;
  const long long total = (long long)B * Ho * Wo * cv;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % cv);
    long long r = i / cv;
    const int xo = (int)(r % Wo);
    r /= Wo;
    const int yo = (int)(r % Ho), b = (int)(r / Ho);
    V m;
    bool have = false;
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) {
        const int y = 2 * yo + dy, x = xo + dx - 1;
        if (y < H && (unsigned)x < (unsigned)W) {
          V v = *(const V*)(in + (((size_t)b * H + y) * W + x) * C + (size_t)c * N);
          if (have) {
#pragma unroll
            for (int k = 0; k < N; ++k) m[k] = m[k] > v[k] ? m[k] : v[k];
          } else {
            m = v;
            have = true;
          }
        }
      }
    *(V*)(out + (((size_t)b * Ho + yo) * Wo + xo) * C + (size_t)c * N) = m;
  }
}

// ------------------------------------------------------------------ AdaptiveAvgPool2d(1): [B][HW][C] -> [B][C]
template <typename T>
__global__ __launch_bounds__(256) void avgpool_hw_kernel(const T* __restrict__ in, T* __restrict__ out, int B, int HW,
                                                         int C) {
  const long long total = (long long)B * C;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C), b = (int)(i / C);
    float s = 0.f;
    for (int p = 0; p < HW; ++p) s += (float)in[((size_t)b * HW + p) * C + c];
    out[i] = (T)(s / (float)HW);
  }
}

// ------------------------------------------------------------------ TPS grid + bilinear sampling
// One block per image.  T = inv_delta_C[:, :F] * C'  (the 3 appended rows of C' are zero), P' = P_hat * T,
// then grid_sample(padding_mode="border", align_corners=True) of the NORMALISED crop.
__global__ __launch_bounds__(256) void tps_sample_kernel(const uint8_t* __restrict__ crops,
                                                         const float* __restrict__ cprime,
                                                         const float* __restrict__ inv_delta_c,
                                                         const float* __restrict__ p_hat, float* __restrict__ out,
                                                         int H, int W, int F) {
  __shared__ float tmat[64 * 2];
  const int b = blockIdx.x, F3 = F + 3;
  const float* cp = cprime + (size_t)b * F * 2;
  for (int i = threadIdx.x; i < F3 * 2; i += blockDim.x) {
    const int r = i >> 1, c = i & 1;
    float s = 0.f;
    for (int k = 0; k < F; ++k) s += inv_delta_c[r * F3 + k] * cp[k * 2 + c];
    tmat[i] = s;
  }
  __syncthreads();
  const uint8_t* im = crops + (size_t)b * H * W;
  const int n = H * W;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    float gx = 0.f, gy = 0.f;
    const float* ph = p_hat + (size_t)i * F3;
    for (int k = 0; k < F3; ++k) {
      const float v = ph[k];
      gx += v * tmat[2 * k];
      gy += v * tmat[2 * k + 1];
    }
    float ix = ((gx + 1.f) / 2.f) * (float)(W - 1);
    float iy = ((gy + 1.f) / 2.f) * (float)(H - 1);
    ix = fminf(fmaxf(ix, 0.f), (float)(W - 1));
    iy = fminf(fmaxf(iy, 0.f), (float)(H - 1));
    const float fx0 = floorf(ix), fy0 = floorf(iy);
    const int x0 = (int)fx0, y0 = (int)fy0, x1 = x0 + 1, y1 = y0 + 1;
    const float nw = ((fx0 + 1.f) - ix) * ((fy0 + 1.f) - iy), ne = (ix - fx0) * ((fy0 + 1.f) - iy);
    const float sw = ((fx0 + 1.f) - ix) * (iy - fy0), se = (ix - fx0) * (iy - fy0);
    float v = 0.f;
    v += norm_u8(im[y0 * W + x0]) * nw;
    if (x1 < W) v += norm_u8(im[y0 * W + x1]) * ne;
    if (y1 < H) v += norm_u8(im[y1 * W + x0]) * sw;
    if (x1 < W && y1 < H) v += norm_u8(im[y1 * W + x1]) * se;
    out[(size_t)b * n + i] = v;
  }
}

// ------------------------------------------------------------------ attention decoder pieces
// One block (256 threads = hidden size) per batch row.
//   hproj  [B][T][256]  i2h(batch_H), fp32 (loop-invariant, computed once by a GEMM)
//   hp     [B][ld_hp]   h2h(h) + bias in columns 0..255 (the same GEMM row also holds W_hh*h + biases)
//   H      [B][T][256]  batch_H (activation type), context out [B][256] (activation type)
template <typename T>
__global__ __launch_bounds__(256) void attn_context_kernel(const float* __restrict__ hproj,
                                                           const float* __restrict__ hp, int ld_hp,
                                                           const float* __restrict__ score_w,
                                                           const T* __restrict__ H, T* __restrict__ ctx, int Tn) {
  __shared__ float e[64];
  __shared__ float part[4];
  const int b = blockIdx.x, j = threadIdx.x, lane = j & 63, wave = j >> 6;
  const float hj = hp[(size_t)b * ld_hp + j], sw = score_w[j];
  for (int t = 0; t < Tn; ++t) {
    float v = tanhf(hproj[((size_t)b * Tn + t) * 256 + j] + hj) * sw;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    if (lane == 0) part[wave] = v;
    __syncthreads();
    if (j == 0) e[t] = part[0] + part[1] + part[2] + part[3];
    __syncthreads();
  }
  // softmax over T (every thread redundantly: T <= 64)
  float mx = -INFINITY;
  for (int t = 0; t < Tn; ++t) mx = fmaxf(mx, e[t]);
  float den = 0.f;
  for (int t = 0; t < Tn; ++t) den += expf(e[t] - mx);
  float acc = 0.f;
  for (int t = 0; t < Tn; ++t) acc += (expf(e[t] - mx) / den) * (float)H[((size_t)b * Tn + t) * 256 + j];
  ctx[(size_t)b * 256 + j] = (T)acc;
}

// gates = gctx [B][1024] (W_ih[:, :256] * context) + ghid [B][ld_hp] columns 256.. (W_hh * h + b_ih + b_hh)
//         + w_onehot[char[b]][1024] (column char of W_ih[:, 256:], stored transposed); LSTMCell update in fp32.
template <typename T>
__global__ __launch_bounds__(256) void attn_cell_kernel(const float* __restrict__ gctx, const float* __restrict__ ghid,
                                                        int ld_hp, const float* __restrict__ w_onehot,
                                                        const int* __restrict__ chars, float* __restrict__ c,
                                                        T* __restrict__ h) {
  const int b = blockIdx.x, j = threadIdx.x;
  const float* gc = gctx + (size_t)b * 1024;
  const float* gh = ghid + (size_t)b * ld_hp + 256;
  const float* wo = w_onehot + (size_t)chars[b] * 1024;
  const float gi = gc[j] + gh[j] + wo[j];
  const float gf = gc[256 + j] + gh[256 + j] + wo[256 + j];
  const float gg = gc[512 + j] + gh[512 + j] + wo[512 + j];
  const float go = gc[768 + j] + gh[768 + j] + wo[768 + j];
  const float si = 1.f / (1.f + expf(-gi)), sf = 1.f / (1.f + expf(-gf)), so = 1.f / (1.f + expf(-go));
  const float cn = sf * c[(size_t)b * 256 + j] + si * tanhf(gg);
  c[(size_t)b * 256 + j] = cn;
  h[(size_t)b * 256 + j] = (T)(so * tanhf(cn));
}

// first maximum of every row of logits [B][ld] (first C columns) -> idx [B]
__global__ __launch_bounds__(64) void argmax_rows_kernel(const float* __restrict__ logits, int ld, int C,
                                                         int* __restrict__ idx) {
  const int b = blockIdx.x, lane = threadIdx.x;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int c = lane; c < C; c += 64) {
    const float v = logits[(size_t)b * ld + c];
    if (v > best) {
      best = v;
      bi = c;
    }
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const float ov = __shfl_xor(best, off);
    const int oi = __shfl_xor(bi, off);
    if (ov > best || (ov == best && oi < bi)) {
      best = ov;
      bi = oi;
    }
  }
  if (lane == 0) idx[b] = bi;
}

// per row: first arg-max and the softmax value at the arg-max (= 1 / sum exp(x - max)); one wave per row
__global__ __launch_bounds__(64) void rowmax_softmax_kernel(const float* __restrict__ logits, int rows, int C,
                                                            int* __restrict__ idx, float* __restrict__ pmax) {
  const int r = blockIdx.x, lane = threadIdx.x;
  const float* row = logits + (size_t)r * C;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int c = lane; c < C; c += 64) {
    const float v = row[c];
    if (v > best) {
      best = v;
      bi = c;
    }
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const float ov = __shfl_xor(best, off);
    const int oi = __shfl_xor(bi, off);
    if (ov > best || (ov == best && oi < bi)) {
      best = ov;
      bi = oi;
    }
  }
  float e = 0.f;
  for (int c = lane; c < C; c += 64) e += expf(row[c] - best);
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) e += __shfl_xor(e, off);
  if (lane == 0) {
    idx[r] = bi;
    pmax[r] = 1.0f / e;
  }
}

unsigned grid_for(long long total) {
  long long g = (total + 255) / 256;
  return (unsigned)(g < 1 ? 1 : (g > 256 * 32 ? 256 * 32 : g));
}

}  // namespace

#define CHECK_LAUNCH(what)                                                                          \
  do {                                                                                              \
    hipError_t e_ = hipGetLastError();                                                              \
    if (e_ != hipSuccess) return mhip_fail(ctx, MHIP_EHIP, what " launch: %s", hipGetErrorString(e_)); \
  } while (0)

int mhip_launch_conv_gray_first(mhip_ctx* ctx, int precision, int in_is_u8, const void* img, int B, int H, int W,
                                int C, const float* w9xC, const float* scale, const float* bias, void* out) {
  if (B < 1 || H < 1 || W < 1 || C < 1 || C > 64) return mhip_fail(ctx, MHIP_EINVAL, "conv_gray_first: bad shape");
  const unsigned grid = (unsigned)(((long long)B * H * W + 63) / 64);
#define L(TT, IN)                                                                                              \
  PROF_LAUNCH(ctx, MHIP_K_CONV_FIRST,                                                                          \
              hipLaunchKernelGGL((conv_gray_first_kernel<TT, IN>), dim3(grid), dim3(256), 0, ctx->stream,      \
                                 (const IN*)img, B, H, W, C, w9xC, scale, bias, (TT*)out))
  if (precision == MHIP_PREC_F16) {
    if (in_is_u8) L(_Float16, uint8_t); else L(_Float16, float);
  } else {
    if (in_is_u8) L(float, uint8_t); else L(float, float);
  }
#undef L
  CHECK_LAUNCH("conv_gray_first");
  return 0;
}

int mhip_launch_maxpool_s21_p01(mhip_ctx* ctx, int precision, const void* in, void* out, int B, int H, int W, int C) {
  const int vn = precision == MHIP_PREC_F16 ? 8 : 4;
  if (B < 1 || H < 2 || W < 1 || C % vn) return mhip_fail(ctx, MHIP_EINVAL, "maxpool_s21: bad shape");
  const unsigned grid = grid_for((long long)B * (H / 2) * (W + 1) * (C / vn));
  if (precision == MHIP_PREC_F16)
    PROF_LAUNCH(ctx, MHIP_K_IMAGE_OPS, hipLaunchKernelGGL((maxpool_s21_p01_kernel<_Float16>), dim3(grid), dim3(256), 0,
                                                          ctx->stream, (const _Float16*)in, (_Float16*)out, B, H, W, C));
  else
    PROF_LAUNCH(ctx, MHIP_K_IMAGE_OPS, hipLaunchKernelGGL((maxpool_s21_p01_kernel<float>), dim3(grid), dim3(256), 0,
                                                          ctx->stream, (const float*)in, (float*)out, B, H, W, C));
  CHECK_LAUNCH("maxpool_s21");
  return 0;
}

int mhip_launch_avgpool_hw(mhip_ctx* ctx, int precision, const void* in, void* out, int B, int HW, int C) {
  if (B < 1 || HW < 1 || C < 1) return mhip_fail(ctx, MHIP_EINVAL, "avgpool: bad shape");
  const unsigned grid = grid_for((long long)B * C);
  if (precision == MHIP_PREC_F16)
    PROF_LAUNCH(ctx, MHIP_K_IMAGE_OPS, hipLaunchKernelGGL((avgpool_hw_kernel<_Float16>), dim3(grid), dim3(256), 0,
                                                          ctx->stream, (const _Float16*)in, (_Float16*)out, B, HW, C));
  else
    PROF_LAUNCH(ctx, MHIP_K_IMAGE_OPS, hipLaunchKernelGGL((avgpool_hw_kernel<float>), dim3(grid), dim3(256), 0,
                                                          ctx->stream, (const float*)in, (float*)out, B, HW, C));
  CHECK_LAUNCH("avgpool");
  return 0;
}

int mhip_launch_tps_sample(mhip_ctx* ctx, const uint8_t* crops, const float* cprime, const float* inv_delta_c,
                           const float* p_hat, float* out, int B, int H, int W, int F) {
  if (B < 1 || H < 2 || W < 2 || F < 2 || F + 3 > 64) return mhip_fail(ctx, MHIP_EINVAL, "tps_sample: bad shape");
  PROF_LAUNCH(ctx, MHIP_K_IMAGE_OPS, hipLaunchKernelGGL(tps_sample_kernel, dim3(B), dim3(256), 0, ctx->stream, crops,
                                                        cprime, inv_delta_c, p_hat, out, H, W, F));
  CHECK_LAUNCH("tps_sample");
  return 0;
}

int mhip_launch_attn_context(mhip_ctx* ctx, int precision, const float* hproj, const float* hp, int ld_hp,
                             const float* score_w, const void* H, void* ctx_out, int B, int Tn) {
  if (B < 1 || Tn < 1 || Tn > 64) return mhip_fail(ctx, MHIP_EINVAL, "attn_context: bad shape");
  if (precision == MHIP_PREC_F16)
    PROF_LAUNCH(ctx, MHIP_K_ATTN, hipLaunchKernelGGL((attn_context_kernel<_Float16>), dim3(B), dim3(256), 0,
                                                     ctx->stream, hproj, hp, ld_hp, score_w, (const _Float16*)H,
                                                     (_Float16*)ctx_out, Tn));
  else
    PROF_LAUNCH(ctx, MHIP_K_ATTN, hipLaunchKernelGGL((attn_context_kernel<float>), dim3(B), dim3(256), 0, ctx->stream,
                                                     hproj, hp, ld_hp, score_w, (const float*)H, (float*)ctx_out, Tn));
  CHECK_LAUNCH("attn_context");
  return 0;
}

int mhip_launch_attn_cell(mhip_ctx* ctx, int precision, const float* gctx, const float* ghid, int ld_hp,
                          const float* w_onehot, const int* chars, float* c, void* h, int B) {
  if (B < 1) return mhip_fail(ctx, MHIP_EINVAL, "attn_cell: bad shape");
  if (precision == MHIP_PREC_F16)
    PROF_LAUNCH(ctx, MHIP_K_ATTN, hipLaunchKernelGGL((attn_cell_kernel<_Float16>), dim3(B), dim3(256), 0, ctx->stream,
                                                     gctx, ghid, ld_hp, w_onehot, chars, c, (_Float16*)h));
  else
    PROF_LAUNCH(ctx, MHIP_K_ATTN, hipLaunchKernelGGL((attn_cell_kernel<float>), dim3(B), dim3(256), 0, ctx->stream,
                                                     gctx, ghid, ld_hp, w_onehot, chars, c, (float*)h));
  CHECK_LAUNCH("attn_cell");
  return 0;
}

int mhip_launch_argmax_rows(mhip_ctx* ctx, const float* logits, int ld, int C, int* idx, int B) {
  if (B < 1 || C < 1 || ld < C) return mhip_fail(ctx, MHIP_EINVAL, "argmax_rows: bad shape");
  PROF_LAUNCH(ctx, MHIP_K_ATTN,
              hipLaunchKernelGGL(argmax_rows_kernel, dim3(B), dim3(64), 0, ctx->stream, logits, ld, C, idx));
  CHECK_LAUNCH("argmax_rows");
  return 0;
}

int mhip_launch_rowmax_softmax(mhip_ctx* ctx, const float* logits, int rows, int C, int* idx, float* pmax) {
  if (rows < 1 || C < 1) return mhip_fail(ctx, MHIP_EINVAL, "rowmax_softmax: bad shape");
  PROF_LAUNCH(ctx, MHIP_K_ATTN, hipLaunchKernelGGL(rowmax_softmax_kernel, dim3(rows), dim3(64), 0, ctx->stream, logits,
                                                   rows, C, idx, pmax));
  CHECK_LAUNCH("rowmax_softmax");
  return 0;
}
