// igemm_common.h — pieces shared by the implicit-GEMM convolution kernels (conv_igemm.hip, conv3x3_patch.hip).
#pragma once
#include "common.h"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float4v __attribute__((ext_vector_type(4)));
typedef unsigned int uint4v __attribute__((ext_vector_type(4)));

namespace igemm {

constexpr int ROWB = 128;       // bytes of one K slice row (64 f16 / 32 f32 channels)
constexpr int NTHREADS = 512;   // 8 waves

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
  int mtiles, ntiles;
  int dil;             // filter dilation (1 = dense)
  const char* in2;     // second input of a channel-concatenated 1x1 conv (DUAL kernels only)
  int Cin1;            // channels taken from `in`; the remaining Cin - Cin1 come from `in2`
  int sy;              // vertical stride (1 or 2); horizontal stride is always 1
  int pad_x;           // horizontal padding (vertical padding is `pad`)
  const char* res;     // optional residual tensor, same shape/type as the output, added before the ReLU
  int ldc;             // output row pitch in elements (>= N): lets a GEMM write into a slice of a wider tensor
  int pad_ok;          // the pad columns [N, roundup(N, 8)) of a pitched row may be overwritten (ConvDesc::pad_cols_writable)
  int row_period;      // > 0: GEMM row q lands in output row (q / period) * row_stride + row_offset + q % period and takes
  int row_stride;      //      residual row q % period — batched patch embedding: one GEMM over all images writes each
  int row_offset;      //      image's token block (after its cls row) and adds the shared position table
  int PH, PW;          // conv3x3_patch: output patch of one workgroup (PH*PW = 256 pixels)
  int tiles_x, tiles_y;
  // LayerNorm folded around the GEMM (ConvDesc::epi, f16 plain GEMMs only)
  int epi;             // EPI_NONE / EPI_LN_ROWS / EPI_LN_COLS / EPI_SPLIT
  const float* ln_a;   // EPI_LN_ROWS: rstd of GEMM row m;  EPI_LN_COLS: rstd of GEMM column n
  const float* ln_b;   // mean * rstd, indexed likewise
  const float* ln_cs;  // EPI_LN_ROWS: column sums of the folded weights [N];  EPI_LN_COLS: row sums [M]
  const float* row_bias;   // EPI_LN_COLS: bias per GEMM row m
  char* out2;          // EPI_SPLIT: the low plane of the output (out = high plane), same pitch
  const char* res2;    // EPI_SPLIT: the low plane of the residual
  int stagger;         // experiment (MARIE_HIP_STAGGER): s_sleep units the odd workgroups of the first round wait before starting
  float* stats;        // EPI_SPLIT: per 64-column chunk c and output row r: (sum, centred sum of squares) at stats[(c*stats_ld + r)*2]
  int stats_ld;
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


template <typename OT>
__device__ __forceinline__ void lds_put(char* base, int pitch, int row, int col, float v) {
  *(OT*)(base + row * pitch + col * (int)sizeof(OT)) = (OT)v;
}

}  // namespace igemm

// 3x3 / pad 1 / dense convolutions through the patch kernel; returns 1 if the shape is not eligible (caller falls
// back to the generic tap-gather kernel), 0 on launch, negative on error.
int mhip_try_launch_conv3x3_patch(mhip_ctx* ctx, int precision, const ConvDesc& d, igemm::IgemmArgs& a);
