// conv_igemm.hip — NHWC implicit-GEMM convolution / GEMM on the gfx950 matrix cores.
//
// Replaces the cuDNN/oneDNN convolutions and cuBLAS GEMMs the reference reaches through
// nn.Conv2d / nn.Linear in marie/models/icr/modules/feature_extraction.py:13-25,
// marie/models/icr/modules/sequence_modeling.py:9,18 and marie/models/icr/model.py:64.
//
//   out[m][n] = act( scale[n] * sum_k A[m][k] * W[n][k] + bias[n] ),   k = (dy*KW + dx)*Cin + c
//
// Design (MI355X-first, no im2col buffer ever exists in HBM):
//   * 128x128 output tile per 256-thread workgroup (4 waves, 64x64 each, 4x4 MFMA 16x16 tiles).
//   * K is walked in 128-byte slices (64 f16 / 32 f32 channels of one filter tap).  Both operand
//     slices go HBM -> LDS with global_load_lds_dwordx4 (LDS-DMA, no VGPR round trip): the A slice
//     is a *gather* — each lane's source address is its output pixel shifted by the tap, or a zero
//     page for padding — while the LDS image stays lane-linear.  Two LDS buffers, one barrier per
//     slice; the next slice's DMA is in flight while the current one feeds the MFMAs.
//   * LDS rows are 128 B; the 16-B slot index is XOR-swizzled with (row>>1)&7 on the SOURCE side
//     and on the ds_read_b128 side (cdna guide rule 21) so the 16 rows of a fragment read hit 16
//     distinct slots of the 256-B bank row.
//   * Rows (m) are enumerated so that a max-pool window is 4 (2x2) or 2 (2x1) consecutive rows.
//     In the 16x16 MFMA C layout a lane owns 4 consecutive rows of one column, so the pool is a
//     max over the lane's own accumulator registers — no extra pass, no extra HBM traffic.
//   * f16 operands use v_mfma_f32_16x16x32_f16; the exact-fp32 parity mode uses
//     v_mfma_f32_16x16x4_f32 on the same LDS image (k order permuted identically for A and W).
#include "common.h"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float4v __attribute__((ext_vector_type(4)));

namespace {

constexpr int BM = 128, BN = 128, ROWB = 128, NTHREADS = 256;
constexpr int TILE_BYTES = BM * ROWB;  // 16 KiB per operand slice

struct IgemmArgs {
  const char* in;
  const char* w;
  const float* scale;
  const float* bias;
  char* out;
  const char* zeros;
  int B, H, W, Cin;
  int KH, KW, pad;
  int Ho, Wo, Hp, Wp;
  int N, M;      // M = rows in pool-friendly order (pre-pool)
  int Ktot;      // KH*KW*Cin
  int nslices;   // Ktot / (ROWB/sizeof(T))
  int cpt;       // slices per tap
  int relu, out_f32;
};

template <typename T>
struct Tr;
template <>
struct Tr<_Float16> {
  static constexpr int E = 8;
  typedef half8 chunk_t;
  static __device__ __forceinline__ void mma(const chunk_t& a, const chunk_t& b, float4v& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  }
};
template <>
struct Tr<float> {
  static constexpr int E = 4;
  typedef float4v chunk_t;
  static __device__ __forceinline__ void mma(const chunk_t& a, const chunk_t& b, float4v& c) {
    // lane group g = lane>>4 holds k = 4g+j in element j; MFMA j contracts k = {j, 4+j, 8+j, 12+j}
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b[2], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b[3], c, 0, 0, 0);
  }
};

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

template <int POOL>
__device__ __forceinline__ void decode_row(const IgemmArgs& p, int m, int& b, int& y, int& x) {
  if (POOL == POOL_NONE) {
    x = m % p.Wo;
    int r = m / p.Wo;
    y = r % p.Ho;
    b = r / p.Ho;
  } else if (POOL == POOL_2x2) {
    int sub = m & 3, q = m >> 2;
    int xp = q % p.Wp;
    int r = q / p.Wp;
    int yp = r % p.Hp;
    b = r / p.Hp;
    y = 2 * yp + (sub >> 1);
    x = 2 * xp + (sub & 1);
  } else {
    int sub = m & 1, q = m >> 1;
    x = q % p.Wo;
    int r = q / p.Wo;
    int yp = r % p.Hp;
    b = r / p.Hp;
    y = 2 * yp + sub;
  }
}

template <typename T, int POOL>
__global__ __launch_bounds__(NTHREADS) void conv_igemm_kernel(IgemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef typename Tr<T>::chunk_t chunk_t;
  constexpr int E = Tr<T>::E;            // elements per 16-B chunk
  constexpr int BKE = ROWB / sizeof(T);  // elements per K slice

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;

  // ---- staging set-up: each thread moves 4 A chunks + 4 W chunks per slice -------------
  const int srow = wave * 8 + (lane >> 3);             // row within a 32-row group
  const int lchunk = (lane & 7) ^ ((srow >> 1) & 7);   // logical chunk this lane fetches
  const char* a_src[4];
  int a_y[4], a_x[4];
  const char* w_src[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    int m = m0 + q * 32 + srow;
    int b = 0, y = -100000, x = -100000;  // invalid rows fail every bounds test
    if (m < p.M) decode_row<POOL>(p, m, b, y, x);
    a_y[q] = y;
    a_x[q] = x;
    size_t pix = ((size_t)b * p.H + (y - p.pad)) * p.W + (x - p.pad);  // tap (0,0) position (may be OOB; guarded)
    a_src[q] = p.in + (pix * p.Cin + (size_t)lchunk * E) * sizeof(T);
    int n = n0 + q * 32 + srow;
    w_src[q] = (n < p.N) ? p.w + ((size_t)n * p.Ktot + (size_t)lchunk * E) * sizeof(T) : nullptr;
  }

  auto stage = [&](int it, int buf) {
    int tap = it / p.cpt, cc = it - tap * p.cpt;
    int dy = tap / p.KW, dx = tap - dy * p.KW;
    size_t a_off = (((size_t)dy * p.W + dx) * p.Cin + (size_t)cc * BKE) * sizeof(T);
    size_t w_off = (size_t)it * BKE * sizeof(T);
    char* la = smem + buf * (2 * TILE_BYTES) + wave * 8 * ROWB;
    char* lb = la + TILE_BYTES;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      int yy = a_y[q] + dy - p.pad, xx = a_x[q] + dx - p.pad;
      bool ok = (unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)p.W;
      const char* src = ok ? a_src[q] + a_off : p.zeros;
      glds16(src, la + q * 32 * ROWB);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const char* src = w_src[q] ? w_src[q] + w_off : p.zeros;
      glds16(src, lb + q * 32 * ROWB);
    }
  };

  // ---- accumulators ---------------------------------------------------------------------
  float4v acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (float4v){0.f, 0.f, 0.f, 0.f};

  const int wr = wave >> 1, wc = wave & 1;
  const int frow = lane & 15, fg = lane >> 4;

  stage(0, 0);
  for (int it = 0; it < p.nslices; ++it) {
    __syncthreads();  // drains this wave's LDS-DMA (vmcnt(0)) and orders everyone's
    if (it + 1 < p.nslices) stage(it + 1, (it + 1) & 1);
    const char* sa = smem + (it & 1) * (2 * TILE_BYTES);
    const char* sb = sa + TILE_BYTES;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      chunk_t a[4], b[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        int ra = wr * 64 + t * 16 + frow;
        a[t] = *(const chunk_t*)(sa + ra * ROWB + (((s * 4 + fg) ^ ((ra >> 1) & 7)) << 4));
        int rb = wc * 64 + t * 16 + frow;
        b[t] = *(const chunk_t*)(sb + rb * ROWB + (((s * 4 + fg) ^ ((rb >> 1) & 7)) << 4));
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) Tr<T>::mma(a[i], b[j], acc[i][j]);
    }
  }

  // ---- epilogue: scale/bias, ReLU, in-register max-pool, store NHWC ---------------------
  const int Mq = (POOL == POOL_2x2) ? (p.M >> 2) : (POOL == POOL_2x1) ? (p.M >> 1) : p.M;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    int n = n0 + wc * 64 + j * 16 + frow;
    if (n >= p.N) continue;
    float sc = p.scale ? p.scale[n] : 1.f;
    float bi = p.bias ? p.bias[n] : 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float t = acc[i][j][r] * sc + bi;
        v[r] = p.relu ? fmaxf(t, 0.f) : t;
      }
      int mrow = m0 + wr * 64 + i * 16 + fg * 4;  // first of this lane's 4 consecutive rows
      if (POOL == POOL_2x2) {
        int q = mrow >> 2;
        if (q < Mq) {
          float o = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
          size_t idx = (size_t)q * p.N + n;
          if (p.out_f32) ((float*)p.out)[idx] = o; else ((T*)p.out)[idx] = (T)o;
        }
      } else if (POOL == POOL_2x1) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          int q = (mrow >> 1) + h;
          if (q < Mq) {
            float o = fmaxf(v[2 * h], v[2 * h + 1]);
            size_t idx = (size_t)q * p.N + n;
            if (p.out_f32) ((float*)p.out)[idx] = o; else ((T*)p.out)[idx] = (T)o;
          }
        }
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          int q = mrow + r;
          if (q < Mq) {
            size_t idx = (size_t)q * p.N + n;
            if (p.out_f32) ((float*)p.out)[idx] = v[r]; else ((T*)p.out)[idx] = (T)v[r];
          }
        }
      }
    }
  }
}

template <typename T>
int launch_t(mhip_ctx* ctx, const IgemmArgs& a, int pool) {
  dim3 grid((a.M + BM - 1) / BM, (a.N + BN - 1) / BN), block(NTHREADS);
  size_t lds = 4 * TILE_BYTES;
  hipError_t e = hipSuccess;
  switch (pool) {
    case POOL_NONE:
      PROF_LAUNCH(ctx, MHIP_K_CONV_IGEMM,
                  hipLaunchKernelGGL((conv_igemm_kernel<T, POOL_NONE>), grid, block, lds, ctx->stream, a));
      break;
    case POOL_2x2:
      PROF_LAUNCH(ctx, MHIP_K_CONV_IGEMM,
                  hipLaunchKernelGGL((conv_igemm_kernel<T, POOL_2x2>), grid, block, lds, ctx->stream, a));
      break;
    default:
      PROF_LAUNCH(ctx, MHIP_K_CONV_IGEMM,
                  hipLaunchKernelGGL((conv_igemm_kernel<T, POOL_2x1>), grid, block, lds, ctx->stream, a));
      break;
  }
  e = hipGetLastError();
  if (e != hipSuccess) return mhip_fail(ctx, MHIP_EHIP, "conv_igemm launch: %s", hipGetErrorString(e));
  return 0;
}

}  // namespace

double mhip_conv_flops(const ConvDesc& d) {
  int Ho = d.H + 2 * d.pad - d.KH + 1, Wo = d.W + 2 * d.pad - d.KW + 1;
  return 2.0 * d.B * Ho * Wo * (double)d.N * d.KH * d.KW * d.Cin;
}

int mhip_launch_conv_igemm(mhip_ctx* ctx, int precision, const ConvDesc& d) {
  const int esz = precision == MHIP_PREC_F16 ? 2 : 4;
  const int bke = ROWB / esz;
  if (!d.in || !d.w || !d.out) return mhip_fail(ctx, MHIP_EINVAL, "conv_igemm: null operand");
  if (d.Cin % bke != 0)
    return mhip_fail(ctx, MHIP_EINVAL, "conv_igemm: Cin=%d must be a multiple of %d", d.Cin, bke);
  IgemmArgs a;
  a.in = (const char*)d.in;
  a.w = (const char*)d.w;
  a.scale = d.scale;
  a.bias = d.bias;
  a.out = (char*)d.out;
  a.zeros = (const char*)ctx->zeros;
  a.B = d.B; a.H = d.H; a.W = d.W; a.Cin = d.Cin;
  a.KH = d.KH; a.KW = d.KW; a.pad = d.pad;
  a.Ho = d.H + 2 * d.pad - d.KH + 1;
  a.Wo = d.W + 2 * d.pad - d.KW + 1;
  if (a.Ho <= 0 || a.Wo <= 0) return mhip_fail(ctx, MHIP_EINVAL, "conv_igemm: empty output %dx%d", a.Ho, a.Wo);
  a.Hp = a.Ho; a.Wp = a.Wo;
  long long M;
  if (d.pool == POOL_2x2) {
    a.Hp = a.Ho / 2; a.Wp = a.Wo / 2;
    M = (long long)d.B * a.Hp * a.Wp * 4;
  } else if (d.pool == POOL_2x1) {
    a.Hp = a.Ho / 2;
    M = (long long)d.B * a.Hp * a.Wo * 2;
  } else {
    M = (long long)d.B * a.Ho * a.Wo;
  }
  if (M <= 0 || M > 0x7fffffffLL - BM) return mhip_fail(ctx, MHIP_EINVAL, "conv_igemm: M=%lld out of range", M);
  a.M = (int)M;
  a.N = d.N;
  a.Ktot = d.KH * d.KW * d.Cin;
  a.cpt = d.Cin / bke;
  a.nslices = d.KH * d.KW * a.cpt;
  a.relu = d.relu;
  a.out_f32 = d.out_f32;
  static bool attr_set = false;
  if (!attr_set) {
    size_t lds = 4 * TILE_BYTES;
#define SETATTR(K) (void)hipFuncSetAttribute((const void*)K, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)
    SETATTR((conv_igemm_kernel<_Float16, POOL_NONE>));
    SETATTR((conv_igemm_kernel<_Float16, POOL_2x2>));
    SETATTR((conv_igemm_kernel<_Float16, POOL_2x1>));
    SETATTR((conv_igemm_kernel<float, POOL_NONE>));
    SETATTR((conv_igemm_kernel<float, POOL_2x2>));
    SETATTR((conv_igemm_kernel<float, POOL_2x1>));
#undef SETATTR
    attr_set = true;
  }
  if (precision == MHIP_PREC_F16) return launch_t<_Float16>(ctx, a, d.pool);
  return launch_t<float>(ctx, a, d.pool);
}
