// lstm.hip — bidirectional LSTM recurrence, persistent over the whole sequence.
//
// Replaces: nn.LSTM(input, 256, bidirectional=True, batch_first=True) as called at
// marie/models/icr/modules/sequence_modeling.py:8,17 (gate order i, f, g, o; h0 = c0 = 0).
// The input projection x_t W_ih^T + b_ih + b_hh of BOTH directions is one MFMA GEMM done
// beforehand by conv_igemm (N = 2048); this file does only the part that is sequential.
//
// MI355X design: the recurrence is independent per batch row, so a workgroup owns 16 batch rows
// of one direction and walks all T steps on its own — no grid-wide synchronisation, one launch per
// layer instead of T.  Per step it computes gates[16][1024] = xproj_t + h[16][256] * W_hh^T:
//   * the 4 waves split the 1024 gate columns so that a wave owns all four gates of the SAME 64
//     hidden units: the cell update (sigmoid/tanh, c, h) happens in the MFMA accumulator
//     registers of the lane that produced the gates, with c kept in registers across all T steps;
//   * h lives in LDS (double-buffered, XOR-swizzled so the 16-row fragment read is conflict-free);
//     the new h is written only to LDS and streamed out to HBM one step later with 16-B stores;
//   * xproj columns are stored gate-interleaved ([dir][wave][u][unit16][gate], done by permuting
//     W_ih's rows at pack time) so a lane fetches its i,f,g,o pre-activations with one 16-B load.
//
// f16 mode (`lstm_rec_resident_f16`): W_hh of one direction is 512 KB in f16 — exactly what ONE
// CU can hold on chip: 4 waves x 64 lanes x 384 VGPRs (k-steps 0..5, MFMA B-fragment order) plus
// 128 KB of LDS (k-steps 6,7).  It is loaded once and the 63-step loop touches HBM/L2 only for
// xproj (64 B per lane per step) and the h write-out.  One wave per SIMD, 512-register budget.
//
// f32 parity mode (`lstm_rec_stream<float>`): W_hh is 1 MB per direction and cannot be resident;
// it is streamed from L2 in fragment order (1 KiB coalesced loads) every step.
#include "common.h"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float4v __attribute__((ext_vector_type(4)));
typedef unsigned int uint4v __attribute__((ext_vector_type(4)));

constexpr int HID = 256, ROWS = 16;

__device__ __forceinline__ float fast_sigmoid(float x) {
  // 1 / (1 + 2^(-x*log2 e)) on v_exp_f32 / v_rcp_f32 (1 ulp each)
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
__device__ __forceinline__ float fast_tanh(float x) {
  // tanh(x) = 1 - 2 / (1 + e^(2x)); saturates cleanly for |x| large (exp2 -> inf -> rcp -> 0)
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.8853900817779268f * x));
}

// xproj column of (dir, wave, u, frow, gate)
__device__ __forceinline__ int xcol(int dir, int wave, int u, int frow) {
  return dir * 1024 + wave * 256 + u * 64 + frow * 4;
}

// copy one step's h (16 rows x 256, swizzled in LDS) to hseq[b][t][dir*256 ..] with 16-B stores
template <typename T>
__device__ __forceinline__ void flush_h(const char* hb, T* hseq, int row0, int B, int Tn, int t, int dir, int tid) {
  constexpr int ROWBYTES = HID * sizeof(T);
  constexpr int CPR = ROWBYTES / 16;  // chunks per row: 32 (f16) / 64 (f32)
#pragma unroll
  for (int c = tid; c < ROWS * CPR; c += 256) {
    const int row = c / CPR, ch = c - row * CPR;
    const int brow = row0 + row;
    if (brow < B) {
      uint4v v = *(const uint4v*)(hb + row * ROWBYTES + ((ch ^ row) << 4));
      *(uint4v*)((char*)(hseq + ((size_t)brow * Tn + t) * 512 + dir * 256) + ch * 16) = v;
    }
  }
}

// ------------------------------------------------------------------ f16: W_hh resident on chip
// Register plan (one wave per SIMD, 512-entry unified file = 256 arch VGPRs + 256 AGPRs):
//   k-steps 0..3  -> 64 fragments in AGPRs (256 regs), read by the MFMA directly as its B operand;
//   k-steps 4,5   -> 32 fragments in VGPRs (128 regs);
//   k-steps 6,7   -> 32 fragments per wave in LDS (128 KiB per workgroup).
// hipcc cannot be talked into sourcing MFMA operands from AGPRs by itself (it allocates everything
// as VGPR, runs out at 256 and shuttles values through v_accvgpr_read), so the three MFMA forms are
// inline asm with explicit register classes.  Inline asm is invisible to the hazard recogniser: the
// only software-managed hazard here, MFMA result -> VALU read, is covered by mfma_fence() after each
// 32-MFMA chain; the chain itself rotates over 4 accumulators, so no MFMA reads a result younger
// than 3 instructions (hardware-interlocked SrcC forwarding).
constexpr int KS_AGPR = 4, KS_VGPR = 2, KS_LDS = 2;
constexpr int HBUF_BYTES = ROWS * HID * 2;          // 8 KiB
constexpr int WLDS_BYTES = 4 * KS_LDS * 16 * 1024;  // 128 KiB
constexpr int RES_LDS_BYTES = 2 * HBUF_BYTES + WLDS_BYTES;

__device__ __forceinline__ void mfma_init_a(float4v& acc, const half8& a, const half8& w) {
  asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=&v"(acc) : "v"(a), "a"(w));
}
__device__ __forceinline__ void mfma_acc_a(float4v& acc, const half8& a, const half8& w) {
  asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "a"(w));
}
__device__ __forceinline__ void mfma_acc_v(float4v& acc, const half8& a, const half8& w) {
  asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(w));
}
__device__ __forceinline__ void mfma_fence(float4v& a0, float4v& a1, float4v& a2, float4v& a3) {
  // >= 18 wait states between the last MFMA of a chain and the first VALU read of its result
  asm volatile("s_nop 15\n\ts_nop 7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
}

__global__ __launch_bounds__(256, 1) void lstm_rec_resident_f16(const float* __restrict__ xproj,
                                                                const half8* __restrict__ wpack,
                                                                _Float16* __restrict__ hseq, int B, int Tn) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* hbuf0 = smem;
  char* hbuf1 = smem + HBUF_BYTES;
  half8* wlds = (half8*)(smem + 2 * HBUF_BYTES);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int frow = lane & 15, fg = lane >> 4;
  const int dir = blockIdx.y;
  const int row0 = blockIdx.x * ROWS;

  // ---- one-time weight load: fragment (s, nt) of this wave, this lane's 16 bytes ----------
  const half8* wp = wpack + (size_t)(dir * 4 + wave) * 8 * 16 * 64 + lane;
  half8 wa[KS_AGPR][16], wv[KS_VGPR][16];
#pragma unroll
  for (int s = 0; s < KS_AGPR; ++s)
#pragma unroll
    for (int nt = 0; nt < 16; ++nt) wa[s][nt] = wp[(s * 16 + nt) * 64];
#pragma unroll
  for (int s = 0; s < KS_VGPR; ++s)
#pragma unroll
    for (int nt = 0; nt < 16; ++nt) wv[s][nt] = wp[((KS_AGPR + s) * 16 + nt) * 64];
  half8* wl = wlds + (size_t)wave * KS_LDS * 16 * 64 + lane;
#pragma unroll
  for (int i = 0; i < KS_LDS * 16; ++i) wl[i * 64] = wp[((KS_AGPR + KS_VGPR) * 16 + i) * 64];

  for (int i = tid; i < HBUF_BYTES / 4; i += 256) ((float*)hbuf0)[i] = 0.f;
  float c[4][4];
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int r = 0; r < 4; ++r) c[u][r] = 0.f;
  __syncthreads();

  int cur = 0;
  int t_prev = -1;
  for (int step = 0; step < Tn; ++step) {
    const int t = dir ? (Tn - 1 - step) : step;
    const char* hb = cur ? hbuf1 : hbuf0;
    char* hn = cur ? hbuf0 : hbuf1;
    if (t_prev >= 0) flush_h<_Float16>(hb, hseq, row0, B, Tn, t_prev, dir, tid);

    const char* arow = hb + frow * (HID * 2);  // A fragments are re-read per gate group: LDS is cheap, VGPRs are not
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      float4v xp[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int brow = row0 + fg * 4 + r;
        xp[r] = (brow < B) ? *(const float4v*)(xproj + ((size_t)brow * Tn + t) * 2048 + xcol(dir, wave, u, frow))
                           : (float4v){0.f, 0.f, 0.f, 0.f};
      }
      float4v acc[4];
      {
        const half8 a = *(const half8*)(arow + ((fg ^ frow) << 4));
#pragma unroll
        for (int q = 0; q < 4; ++q) mfma_init_a(acc[q], a, wa[0][q * 4 + u]);
      }
#pragma unroll
      for (int s = 1; s < KS_AGPR; ++s) {
        const half8 a = *(const half8*)(arow + (((s * 4 + fg) ^ frow) << 4));
#pragma unroll
        for (int q = 0; q < 4; ++q) mfma_acc_a(acc[q], a, wa[s][q * 4 + u]);
      }
#pragma unroll
      for (int s = 0; s < KS_VGPR; ++s) {
        const half8 a = *(const half8*)(arow + ((((KS_AGPR + s) * 4 + fg) ^ frow) << 4));
#pragma unroll
        for (int q = 0; q < 4; ++q) mfma_acc_v(acc[q], a, wv[s][q * 4 + u]);
      }
#pragma unroll
      for (int s = 0; s < KS_LDS; ++s) {
        const half8 a = *(const half8*)(arow + ((((KS_AGPR + KS_VGPR + s) * 4 + fg) ^ frow) << 4));
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const half8 w = wl[(s * 16 + q * 4 + u) * 64];
          mfma_acc_v(acc[q], a, w);
        }
      }
      mfma_fence(acc[0], acc[1], acc[2], acc[3]);

      const int unit = wave * 64 + u * 16 + frow;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float ig = fast_sigmoid(acc[0][r] + xp[r][0]);
        const float fgate = fast_sigmoid(acc[1][r] + xp[r][1]);
        const float gg = fast_tanh(acc[2][r] + xp[r][2]);
        const float og = fast_sigmoid(acc[3][r] + xp[r][3]);
        const float cn = fgate * c[u][r] + ig * gg;
        c[u][r] = cn;
        const float h = og * fast_tanh(cn);
        const int row = fg * 4 + r;
        *(_Float16*)(hn + row * (HID * 2) + (((unit >> 3) ^ row) << 4) + (unit & 7) * 2) = (_Float16)h;
      }
    }
    __syncthreads();
    cur ^= 1;
    t_prev = t;
  }
  flush_h<_Float16>(cur ? hbuf1 : hbuf0, hseq, row0, B, Tn, t_prev, dir, tid);
}

// ------------------------------------------------------------------ generic: W_hh streamed from L2
template <typename T>
struct Lt;
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

template <typename T>
__global__ __launch_bounds__(256) void lstm_rec_stream(const float* __restrict__ xproj,
                                                       const char* __restrict__ wpack, T* __restrict__ hseq,
                                                       int B, int Tn) {
  typedef typename Lt<T>::chunk_t chunk_t;
  [[maybe_unused]] constexpr int E = Lt<T>::E;
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

  int cur = 0, t_prev = -1;
  for (int step = 0; step < Tn; ++step) {
    const int t = dir ? (Tn - 1 - step) : step;
    const char* hb = hbuf[cur];
    char* hn = hbuf[cur ^ 1];
    if (t_prev >= 0) flush_h<T>(hb, hseq, row0, B, Tn, t_prev, dir, tid);

    float4v acc[16];  // index q*4+u
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int brow = row0 + fg * 4 + r;
        float4v x = (brow < B) ? *(const float4v*)(xproj + ((size_t)brow * Tn + t) * 2048 + xcol(dir, wave, u, frow))
                               : (float4v){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q * 4 + u][r] = x[q];
      }
#pragma unroll 2
    for (int s = 0; s < S; ++s) {
      const chunk_t a = *(const chunk_t*)(hb + frow * ROWBYTES + (((s * 4 + fg) ^ frow) << 4));
#pragma unroll
      for (int nt = 0; nt < 16; ++nt) Lt<T>::mma(a, wp[(size_t)(s * 16 + nt) * 64], acc[nt]);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int unit = wave * 64 + u * 16 + frow;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float ig = fast_sigmoid(acc[u][r]);
        const float fgate = fast_sigmoid(acc[4 + u][r]);
        const float gg = fast_tanh(acc[8 + u][r]);
        const float og = fast_sigmoid(acc[12 + u][r]);
        const float cn = fgate * c[u][r] + ig * gg;
        c[u][r] = cn;
        const float h = og * fast_tanh(cn);
        const int row = fg * 4 + r;
        *(T*)(hn + row * ROWBYTES + ((((unit / E) ^ row)) << 4) + (unit % E) * sizeof(T)) = (T)h;  // E: elems per chunk
      }
    }
    __syncthreads();
    cur ^= 1;
    t_prev = t;
  }
  flush_h<T>(hbuf[cur], hseq, row0, B, Tn, t_prev, dir, tid);
}

}  // namespace

size_t mhip_lstm_wpack_bytes(int precision) {
  return (size_t)2 * 1024 * 256 * (precision == MHIP_PREC_F16 ? 2 : 4);
}

// Row n of PyTorch's [4*256][in] gate matrices that lands in xproj column `col` (within one direction):
// col = wave*256 + u*64 + frow*4 + gate  <->  n = gate*256 + wave*64 + u*16 + frow.
int mhip_lstm_xproj_row(int col) {
  const int gate = col & 3, frow = (col >> 2) & 15, u = (col >> 6) & 3, wave = col >> 8;
  return gate * 256 + wave * 64 + u * 16 + frow;
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
    static std::once_flag attr_set;
    std::call_once(attr_set, [&] {
      (void)hipFuncSetAttribute((const void*)lstm_rec_resident_f16, hipFuncAttributeMaxDynamicSharedMemorySize,
                                RES_LDS_BYTES);
    });
    PROF_LAUNCH(ctx, MHIP_K_LSTM_REC,
                hipLaunchKernelGGL(lstm_rec_resident_f16, grid, block, RES_LDS_BYTES, ctx->stream, xproj,
                                   (const half8*)wpack, (_Float16*)hseq, B, T));
  } else {
    PROF_LAUNCH(ctx, MHIP_K_LSTM_REC,
                hipLaunchKernelGGL((lstm_rec_stream<float>), grid, block, 0, ctx->stream, xproj,
                                   (const char*)wpack, (float*)hseq, B, T));
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return mhip_fail(ctx, MHIP_EHIP, "lstm launch: %s", hipGetErrorString(e));
  return 0;
}
