// EXPERIMENT (round 2) — NOT part of libmarie_hip.so.  The plain f16 GEMM tile of conv_igemm.hip (256 x 256, 8 waves, 64 KiB K slices
// through a 2-slot LDS ring by LDS-DMA) with the two-group schedule of the CDNA guide's 8-phase template: waves 0-3 (rows 0-127) and
// waves 4-7 (rows 128-255) run one barrier apart, so on every SIMD one wave's fragment reads and DMA issue sit under the other wave's
// 32 MFMAs.  Per slice and wave:  R0 (reads of k-group 0) | M0 (32 MFMAs) | R1 | M1, four barriers; the second group starts with one
// extra barrier.  DMA of slice s+1: group 0 issues it in R0 / M0 of slice s and waits for it at the end of M1; group 1 issues
// half in M1 of slice s-1 and half in R0 of slice s and waits at the end of R1 — both waits precede the same workgroup barrier.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 gemm_twogroup_bench.hip -o ../gb_twogroup && ../gb_twogroup [rows] [iters]
// It times this kernel against the production one (interleaved launches, same process) and compares the outputs.
#include <stdarg.h>
#include <stdlib.h>

#include <vector>

#include "../../../marie_icr_amd/csrc/conv_igemm.hip"

int mhip_fail(mhip_ctx* ctx, int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vfprintf(stderr, fmt, ap);
  va_end(ap);
  fprintf(stderr, "\n");
  return code;
}
void mhip_prof_begin(mhip_ctx*, int, hipEvent_t*) {}
void mhip_prof_end(mhip_ctx*, int, hipEvent_t) {}
int mhip_try_launch_conv3x3_patch(mhip_ctx*, int, const ConvDesc&, igemm::IgemmArgs&) { return 1; }

namespace tg {

struct Args {
  const char* A;
  const char* W;
  const float* bias;
  char* out;
  int M, N, K, nslices, ntiles;
};

constexpr int A_BYTES = 256 * ROWB, STAGE = 512 * ROWB;

__global__ __launch_bounds__(512) void gemm_twogroup_kernel(Args p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, wc = wave & 3;
  int mt, nt;
  {
    const int nblk = gridDim.x, bid = blockIdx.x;
    const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7;
    const int L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    nt = L % p.ntiles;
    mt = L / p.ntiles;
  }
  const int srow = wave * 8 + (lane >> 3);
  const int lane_off = (((lane & 7) ^ ((srow >> 1) & 7)) << 4);
  const size_t rowb = (size_t)p.K * 2;
  const char* a_src[4];
  const char* w_src[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    a_src[q] = p.A + (size_t)min(mt * 256 + q * 64 + srow, p.M - 1) * rowb + lane_off;
    w_src[q] = p.W + (size_t)min(nt * 256 + q * 64 + srow, p.N - 1) * rowb + lane_off;
  }
  auto chunk = [&](int slot, int g) {
    char* la = smem + slot * STAGE + wave * (8 * ROWB);
    if (g < 4) { glds16(a_src[g], la + g * (64 * ROWB)); a_src[g] += ROWB; }
    else { glds16(w_src[g - 4], la + A_BYTES + (g - 4) * (64 * ROWB)); w_src[g - 4] += ROWB; }
  };
  const int frow = lane & 15, fg = lane >> 4;
  int a_off, b_off[2];
  {
    const int ra = grp * 128 + frow;
    a_off = ra * ROWB + ((fg ^ ((ra >> 1) & 7)) << 4);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int rb = wc * 64 + 8 * (frow >> 2) + 4 * h + (frow & 3);
      b_off[h] = A_BYTES + rb * ROWB + ((fg ^ ((rb >> 1) & 7)) << 4);
    }
  }
  float4v acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (float4v){0.f, 0.f, 0.f, 0.f};
  const int ns = p.nslices;

  // ---- prologue
#pragma unroll
  for (int g = 0; g < 8; ++g) chunk(0, g);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (grp == 1) {
    if (ns > 1) {
#pragma unroll
      for (int g = 0; g < 4; ++g) chunk(1, g);
    }
    __builtin_amdgcn_s_barrier();
  }

#pragma unroll 1
  for (int s = 0; s < ns; ++s) {
    const char* sb = smem + (s & 1) * STAGE;
    const int nslot = (s + 1) & 1;
    const bool next = s + 1 < ns, next2 = s + 2 < ns;
    half8 a[8], b[4];
    // ---- R0
#pragma unroll
    for (int j = 0; j < 4; ++j) b[j] = *(const half8*)(sb + b_off[j & 1] + (j >> 1) * (32 * ROWB));
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = *(const half8*)(sb + a_off + i * (16 * ROWB));
    if (next) {
      if (grp == 0) {
#pragma unroll
        for (int g = 0; g < 4; ++g) chunk(nslot, g);
      } else {
#pragma unroll
        for (int g = 4; g < 8; ++g) chunk(nslot, g);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // ---- M0
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if ((i & 1) == 0 && grp == 0 && next) chunk(nslot, 4 + (i >> 1));
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[j], a[i], acc[i][j], 0, 0, 0);
    }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_s_barrier();
    // ---- R1
#pragma unroll
    for (int j = 0; j < 4; ++j) b[j] = *(const half8*)(sb + (b_off[j & 1] ^ 64) + (j >> 1) * (32 * ROWB));
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = *(const half8*)(sb + (a_off ^ 64) + i * (16 * ROWB));
    if (grp == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // ---- M1
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if ((i & 1) == 0 && grp == 1 && next2) chunk(s & 1, i >> 1);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[j], a[i], acc[i][j], 0, 0, 0);
    }
    __builtin_amdgcn_s_setprio(0);
    if (grp == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  if (grp == 0) __builtin_amdgcn_s_barrier();

  // ---- epilogue: bias, f16, 16-byte stores straight from the (transposed) accumulators
  const int row0 = mt * 256 + grp * 128 + frow, col0 = nt * 256 + wc * 64 + 8 * fg;
#pragma unroll
  for (int jj = 0; jj < 2; ++jj) {
    const int col = col0 + 32 * jj;
    float bi[8];
    {
      const float4v b0 = *(const float4v*)(p.bias + min(col, p.N - 8)), b1 = *(const float4v*)(p.bias + min(col, p.N - 8) + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { bi[e] = b0[e]; bi[4 + e] = b1[e]; }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int row = row0 + 16 * i;
      half8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (_Float16)(acc[i][2 * jj + (e >> 2)][e & 3] + bi[e]);
      if (row < p.M && col < p.N) *(half8*)(p.out + ((size_t)row * p.N + col) * 2) = o;
    }
  }
}

}  // namespace tg


// ---- a third variant: FOUR waves (one per SIMD, up to 512 registers each), 128 x 128 per wave: 33 % fewer LDS fragment bytes per
// MFMA (16 ds_read_b128 per 64 MFMAs instead of 12 per 32), fragments double-buffered across k-groups so that every read is issued a
// whole phase before its MFMAs, one barrier per slice placed between the two k-groups (at it every wave holds the whole slice in
// registers: the slot is free, and the next slice — issued half a slice earlier — has landed).
namespace fw {

constexpr int A_BYTES = 256 * ROWB, STAGE = 512 * ROWB;

__global__ __launch_bounds__(256) void gemm_fourwave_kernel(tg::Args p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  int mt, nt;
  {
    const int nblk = gridDim.x, bid = blockIdx.x;
    const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7;
    const int L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    nt = L % p.ntiles;
    mt = L / p.ntiles;
  }
  // staging: instruction (4 q + wave) of an operand covers its rows 8 (4 q + wave) .. + 7; swizzle term (row >> 1) & 7 = 4 (wave & 1) + (r >> 1)
  const int r8 = lane >> 3;
  const int lane_off = (((lane & 7) ^ (4 * (wave & 1) + (r8 >> 1))) << 4);
  const size_t rowb = (size_t)p.K * 2;
  // uniform tile bases + 32-bit per-lane offsets (sixteen 64-bit pointers per lane spill: the fragments take 128 registers)
  const char* a_tile = p.A + (size_t)mt * 256 * rowb;
  const char* w_tile = p.W + (size_t)nt * 256 * rowb;
  unsigned a_vo[8], w_vo[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int row = (4 * q + wave) * 8 + r8;
    a_vo[q] = (unsigned)min(row, p.M - 1 - mt * 256) * (unsigned)rowb + lane_off;
    w_vo[q] = (unsigned)min(row, p.N - 1 - nt * 256) * (unsigned)rowb + lane_off;
  }
  unsigned kA = 0, kW = 0;                     // K byte offsets of the next A half / W half to issue (the halves of a slice go out at different times)
  auto chunk = [&](int slot, int g) {       // g < 8: A instruction g of this wave, else W instruction g - 8
    if (g < 8) glds16(a_tile + (a_vo[g] + kA), smem + slot * STAGE + (4 * g + wave) * 1024);
    else glds16(w_tile + (w_vo[g - 8] + kW), smem + slot * STAGE + A_BYTES + (4 * (g - 8) + wave) * 1024);
  };
  const int frow = lane & 15, fg = lane >> 4;
  int a_off, b_off[2];
  {
    const int ra = wr * 128 + frow;
    a_off = ra * ROWB + ((fg ^ ((ra >> 1) & 7)) << 4);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int rb = wc * 128 + 8 * (frow >> 2) + 4 * h + (frow & 3);
      b_off[h] = A_BYTES + rb * ROWB + ((fg ^ ((rb >> 1) & 7)) << 4);
    }
  }
  float4v acc[8][8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = (float4v){0.f, 0.f, 0.f, 0.f};
  const int ns = p.nslices;
  half8 a0[8], b0[8], a1[8], b1[8];
  auto load_frags = [&](half8 (&a)[8], half8 (&b)[8], const char* sb, int kg) {
#pragma unroll
    for (int j = 0; j < 8; ++j) b[j] = *(const half8*)(sb + (b_off[j & 1] ^ (kg << 6)) + (j >> 1) * (32 * ROWB));
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = *(const half8*)(sb + (a_off ^ (kg << 6)) + i * (16 * ROWB));
  };
  auto mfmas = [&](const half8 (&a)[8], const half8 (&b)[8]) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
#ifdef FW_ASM_MFMA
        // round 3: the accumulators pinned to AGPRs by the operand constraint — as the builtin compiles, hipcc moves them between the
        // register classes around every phase (~400 v_accvgpr_mov per iteration, section 9 of profiles/r02/f_gemm_tile_budget.txt)
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(b[j]), "v"(a[i]));
#else
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[j], a[i], acc[i][j], 0, 0, 0);
#endif
      }
    __builtin_amdgcn_s_setprio(0);
  };

  // ---- prologue: slice 0 landed, its first k-group in registers, the A half of slice 1 on its way
#pragma unroll
  for (int g = 0; g < 16; ++g) chunk(0, g);
  kA += ROWB; kW += ROWB;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  load_frags(a0, b0, smem, 0);
  if (ns > 1) {
#pragma unroll
    for (int g = 0; g < 8; ++g) chunk(1, g);
    kA += ROWB;
  }
#pragma unroll 1
  for (int t = 0; t < ns; ++t) {
    const char* sb = smem + (t & 1) * STAGE;
    const char* sn = smem + ((t + 1) & 1) * STAGE;
    // phase A: the W half of slice t + 1; fragments of (t, k-group 1); MFMAs of k-group 0
    if (t + 1 < ns) {
#pragma unroll
      for (int g = 8; g < 16; ++g) chunk((t + 1) & 1, g);
      kW += ROWB;
    }
    load_frags(a1, b1, sb, 1);
    mfmas(a0, b0);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");     // slice t + 1 landed; slice t is in registers
    __builtin_amdgcn_s_barrier();
    // phase B: the A half of slice t + 2 into the slot of slice t; fragments of (t + 1, k-group 0); MFMAs of k-group 1
    if (t + 2 < ns) {
#pragma unroll
      for (int g = 0; g < 8; ++g) chunk(t & 1, g);
      kA += ROWB;
    }
    if (t + 1 < ns) load_frags(a0, b0, sn, 0);
    mfmas(a1, b1);
  }

  // ---- epilogue: bias, f16, 16-byte stores straight from the (transposed) accumulators
#ifdef FW_ASM_MFMA
  asm volatile("s_nop 15\n s_nop 15" ::: "memory");      // the compiler does not know the asm above is an MFMA: no hazard slots of its own
#endif
  const int row0 = mt * 256 + wr * 128 + frow, col0 = nt * 256 + wc * 128 + 8 * fg;
#pragma unroll
  for (int jj = 0; jj < 4; ++jj) {
    const int col = col0 + 32 * jj;
    float bi[8];
    {
      const float4v q0 = *(const float4v*)(p.bias + min(col, p.N - 8)), q1 = *(const float4v*)(p.bias + min(col, p.N - 8) + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { bi[e] = q0[e]; bi[4 + e] = q1[e]; }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int row = row0 + 16 * i;
      half8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (_Float16)(acc[i][2 * jj + (e >> 2)][e & 3] + bi[e]);
      if (row < p.M && col < p.N) *(half8*)(p.out + ((size_t)row * p.N + col) * 2) = o;
    }
  }
}

}  // namespace fw

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void fill_kernel(_Float16* p, size_t n, unsigned seed, float amp) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned s = (unsigned)i * 2654435761u + seed;
    s ^= s >> 15; s *= 2246822519u; s ^= s >> 13;
    p[i] = (_Float16)(((int)(s >> 9) % 2001 - 1000) * 0.001f * amp);
  }
}
__global__ void diff_kernel(const _Float16* a, const _Float16* b, size_t n, unsigned* bad, float* maxd) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float d = fabsf((float)a[i] - (float)b[i]);
    if (d > 0.f) { atomicAdd(bad, 1u); atomicMax((unsigned*)maxd, __float_as_uint(d)); }
  }
}

int main(int argc, char** argv) {
  const int rows = argc > 1 ? atoi(argv[1]) : 640 * 577, iters = argc > 2 ? atoi(argv[2]) : 5;
  mhip_ctx ctx;
  CK(hipStreamCreate(&ctx.stream));
  CK(hipMalloc(&ctx.zeros, MHIP_ZERO_BYTES));
  CK(hipMemset(ctx.zeros, 0, MHIP_ZERO_BYTES));
  struct Shape { const char* name; int K, N; };
  const Shape shapes[] = {{"768 -> 2304", 768, 2304}, {"768 -> 768", 768, 768}, {"768 -> 3072", 768, 3072}, {"3072 -> 768", 3072, 768},
                          {"3072 -> 2304", 3072, 2304}, {"64 -> 512", 64, 512}, {"128 -> 1000", 128, 1000}};
  _Float16 *A, *W, *C0, *C1;
  float* bias;
  CK(hipMalloc((void**)&A, (size_t)rows * 3072 * 2)); CK(hipMalloc((void**)&C0, (size_t)rows * 3072 * 2)); CK(hipMalloc((void**)&C1, (size_t)rows * 3072 * 2));
  CK(hipMalloc((void**)&W, (size_t)3072 * 3072 * 2)); CK(hipMalloc((void**)&bias, 3072 * 4));
  fill_kernel<<<1024, 256, 0, ctx.stream>>>(A, (size_t)rows * 3072, 1u, 1.0f);
  fill_kernel<<<1024, 256, 0, ctx.stream>>>(W, (size_t)3072 * 3072, 2u, 0.05f);
  CK(hipMemset(bias, 0, 3072 * 4));
  unsigned* bad; float* maxd;
  CK(hipMalloc((void**)&bad, 4)); CK(hipMalloc((void**)&maxd, 4));
  (void)hipFuncSetAttribute((const void*)tg::gemm_twogroup_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * tg::STAGE);
  (void)hipFuncSetAttribute((const void*)fw::gemm_fourwave_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * fw::STAGE);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (const Shape& s : shapes) {
    ConvDesc d;
    d.in = A; d.w = W; d.bias = bias; d.out = C0; d.B = 1; d.H = 1; d.W = rows; d.Cin = s.K; d.N = s.N;
    tg::Args a;
    a.A = (const char*)A; a.W = (const char*)W; a.bias = bias; a.out = (char*)C1; a.M = rows; a.N = s.N; a.K = s.K; a.nslices = s.K / 64;
    a.ntiles = (s.N + 255) / 256;
    const int grid = ((rows + 255) / 256) * a.ntiles;
    float ms[3] = {0, 0, 0};
    const int nvar = getenv("GB_ONLY2") ? 2 : 3;
    for (int it = -1; it < iters; ++it)
      for (int which = 0; which < nvar; ++which) {
        CK(hipEventRecord(e0, ctx.stream));
        if (which == 0) { if (mhip_launch_conv_igemm(&ctx, MHIP_PREC_F16, d)) return 1; }
        else if (which == 1) hipLaunchKernelGGL(tg::gemm_twogroup_kernel, dim3(grid), dim3(512), 2 * tg::STAGE, ctx.stream, a);
        else hipLaunchKernelGGL(fw::gemm_fourwave_kernel, dim3(grid), dim3(256), 2 * fw::STAGE, ctx.stream, a);
        CK(hipEventRecord(e1, ctx.stream));
        CK(hipStreamSynchronize(ctx.stream));
        float t = 0;
        CK(hipEventElapsedTime(&t, e0, e1));
        if (it >= 0) ms[which] += t / iters;
      }
    CK(hipMemset(bad, 0, 4)); CK(hipMemset(maxd, 0, 4));
    diff_kernel<<<1024, 256, 0, ctx.stream>>>(C0, C1, (size_t)rows * s.N, bad, maxd);
    unsigned hb = 0; float hm = 0;
    CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&hm, maxd, 4, hipMemcpyDeviceToHost));
    const double fl = 2.0 * rows * (double)s.K * s.N;
    printf("%-14s production %8.3f ms %7.1f TFLOP/s | two-group %8.3f ms %7.1f | four-wave %8.3f ms %7.1f TFLOP/s | last variant: %u elements differ (max %.4g)\n",
           s.name, ms[0], fl / ms[0] * 1e-9, ms[1], fl / ms[1] * 1e-9, ms[2], nvar > 2 ? fl / ms[2] * 1e-9 : 0.0, hb, hm);
  }
  return 0;
}
