// lstm.hip — bidirectional LSTM recurrence, persistent over the whole sequence.
//
// Replaces: nn.LSTM(input, 256, bidirectional=True, batch_first=True) as called at
// marie/models/icr/modules/sequence_modeling.py:8,17 (gate order i, f, g, o; h0 = c0 = 0).
// The input projection x_t W_ih^T + b_ih + b_hh of BOTH directions is one MFMA GEMM done
// beforehand by conv_igemm (N = 2048); this kernel does only the part that is sequential.
//
// MI355X design: the recurrence is independent per batch row, so a workgroup owns 16 batch rows
// of one direction and walks all T steps on its own — no grid-wide synchronisation, one launch per
// layer instead of T.  Per step it computes gates[16][1024] = xproj_t + h[16][256] * W_hh^T:
//   * h lives in LDS (double-buffered, XOR-swizzled so the 16-row fragment read is conflict-free);
//   * W_hh is pre-packed on the host in MFMA B-fragment order, so each wave streams its 128 KB
//     (f16) slice with fully coalesced 1-KiB global_load_dwordx4 from L2 (both directions' W_hh
//     = 1 MB stay L2-resident for the whole launch);
//   * the 4 waves split the 1024 gate columns so that a wave owns all four gates of the SAME 64
//     hidden units: the cell update (sigmoid/tanh, c, h) happens in the MFMA accumulator
//     registers of the lane that produced the gates, with c kept in registers across all T steps.
#include "common.h"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float4v __attribute__((ext_vector_type(4)));

template <typename T>
struct Lt;
template <>
struct Lt<_Float16> {
  static constexpr int E = 8;
  typedef half8 chunk_t;
  static __device__ __forceinline__ void mma(const chunk_t& a, const chunk_t& b, float4v& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  }
};
template <>
struct Lt<float> {
  static constexpr int E = 4;
  typedef float4v chunk_t;
  static __device__ __forceinline__ void mma(const chunk_t& a, const chunk_t& b, float4v& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b[2], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b[3], c, 0, 0, 0);
  }
};

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

constexpr int HID = 256, ROWS = 16;

template <typename T>
__global__ __launch_bounds__(256) void lstm_rec_kernel(const float* __restrict__ xproj,
                                                       const char* __restrict__ wpack, T* __restrict__ hseq,
                                                       int B, int Tn) {
  typedef typename Lt<T>::chunk_t chunk_t;
  constexpr int E = Lt<T>::E;
  constexpr int S = HID / (4 * E);           // k-groups per step (f16: 8, f32: 16)
  constexpr int ROWBYTES = HID * sizeof(T);  // 512 / 1024
  __shared__ __attribute__((aligned(16))) char hbuf[2][ROWS * ROWBYTES];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int frow = lane & 15, fg = lane >> 4;
  const int dir = blockIdx.y;
  const int row0 = blockIdx.x * ROWS;

  const chunk_t* wp = (const chunk_t*)wpack + (size_t)(dir * 4 + wave) * S * 16 * 64 + lane;

  for (int i = tid; i < ROWS * ROWBYTES / 4; i += 256) ((float*)hbuf[0])[i] = 0.f;
  float c[4][4];
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int r = 0; r < 4; ++r) c[u][r] = 0.f;
  __syncthreads();

  int cur = 0;
  for (int step = 0; step < Tn; ++step) {
    const int t = dir ? (Tn - 1 - step) : step;

    // input projection of this time step, in the accumulator layout (row = fg*4+r, col = unit)
    float4v acc[16];
#pragma unroll
    for (int nt = 0; nt < 16; ++nt) {
      const int col = dir * 1024 + (nt >> 2) * 256 + wave * 64 + (nt & 3) * 16 + frow;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int brow = row0 + fg * 4 + r;
        acc[nt][r] = (brow < B) ? xproj[((size_t)brow * Tn + t) * 2048 + col] : 0.f;
      }
    }

    const char* hb = hbuf[cur];
#pragma unroll 2
    for (int s = 0; s < S; ++s) {
      const chunk_t a = *(const chunk_t*)(hb + frow * ROWBYTES + (((s * 4 + fg) ^ frow) << 4));
#pragma unroll
      for (int nt = 0; nt < 16; ++nt) Lt<T>::mma(a, wp[(size_t)(s * 16 + nt) * 64], acc[nt]);
    }

    char* hn = hbuf[cur ^ 1];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int unit = wave * 64 + u * 16 + frow;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float ig = sigmoidf_(acc[u][r]);
        const float fgate = sigmoidf_(acc[4 + u][r]);
        const float gg = tanhf(acc[8 + u][r]);
        const float og = sigmoidf_(acc[12 + u][r]);
        const float cn = fgate * c[u][r] + ig * gg;
        c[u][r] = cn;
        const float h = og * tanhf(cn);
        const int row = fg * 4 + r;
        *(T*)(hn + row * ROWBYTES + ((((unit / E) ^ row)) << 4) + (unit % E) * sizeof(T)) = (T)h;
        const int brow = row0 + row;
        if (brow < B) hseq[((size_t)brow * Tn + t) * 512 + dir * 256 + unit] = (T)h;
      }
    }
    __syncthreads();
    cur ^= 1;
  }
}

}  // namespace

size_t mhip_lstm_wpack_bytes(int precision) {
  return (size_t)2 * 1024 * 256 * (precision == MHIP_PREC_F16 ? 2 : 4);
}

void mhip_lstm_pack_whh(int precision, const float* whh_fwd, const float* whh_bwd, void* dst) {
  const int E = precision == MHIP_PREC_F16 ? 8 : 4;
  const int S = HID / (4 * E);
  size_t o = 0;
  for (int dir = 0; dir < 2; ++dir) {
    const float* w = dir ? whh_bwd : whh_fwd;
    for (int wave = 0; wave < 4; ++wave)
      for (int s = 0; s < S; ++s)
        for (int nt = 0; nt < 16; ++nt)
          for (int lane = 0; lane < 64; ++lane) {
            const int n = (nt >> 2) * 256 + wave * 64 + (nt & 3) * 16 + (lane & 15);
            for (int j = 0; j < E; ++j, ++o) {
              const int k = s * 4 * E + E * (lane >> 4) + j;
              const float v = w[(size_t)n * HID + k];
              if (precision == MHIP_PREC_F16) ((_Float16*)dst)[o] = (_Float16)v;
              else ((float*)dst)[o] = v;
            }
          }
  }
}

int mhip_launch_lstm_rec(mhip_ctx* ctx, int precision, const float* xproj, const void* wpack, void* hseq, int B,
                         int T) {
  if (B < 1 || T < 1) return mhip_fail(ctx, MHIP_EINVAL, "lstm: bad shape B=%d T=%d", B, T);
  dim3 grid((B + ROWS - 1) / ROWS, 2), block(256);
  if (precision == MHIP_PREC_F16) {
    PROF_LAUNCH(ctx, MHIP_K_LSTM_REC,
                hipLaunchKernelGGL((lstm_rec_kernel<_Float16>), grid, block, 0, ctx->stream, xproj,
                                   (const char*)wpack, (_Float16*)hseq, B, T));
  } else {
    PROF_LAUNCH(ctx, MHIP_K_LSTM_REC,
                hipLaunchKernelGGL((lstm_rec_kernel<float>), grid, block, 0, ctx->stream, xproj,
                                   (const char*)wpack, (float*)hseq, B, T));
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return mhip_fail(ctx, MHIP_EHIP, "lstm launch: %s", hipGetErrorString(e));
  return 0;
}
