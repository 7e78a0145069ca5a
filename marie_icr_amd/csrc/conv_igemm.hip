// conv_igemm.hip — NHWC implicit-GEMM convolution / GEMM on the gfx950 matrix cores.
//
// Replaces the cuDNN/oneDNN convolutions and cuBLAS GEMMs the reference reaches through
// nn.Conv2d / nn.Linear in marie/models/icr/modules/feature_extraction.py:13-25,
// marie/models/icr/modules/sequence_modeling.py:9,18 and marie/models/icr/model.py:64.
//
//   out[m][n] = act( scale[n] * sum_k A[m][k] * W[n][k] + bias[n] ),   k = (dy*KW + dx)*Cin + c
//
// Design (MI355X-first, no im2col buffer ever exists in HBM):
//   * 256x128 output tile per 512-thread workgroup: 8 waves as 4(M) x 2(N), 64x64 per wave =
//     4x4 MFMA 16x16 tiles, two waves per SIMD so one wave's ds_reads hide under the other's MFMAs.
//   * K is walked in 128-byte slices (64 f16 / 32 f32 channels of one filter tap).  Both operand
//     slices go HBM/L2 -> LDS with global_load_lds_dwordx4 (LDS-DMA, no VGPR round trip): the A slice
//     is a *gather* - each lane's source address is its output pixel shifted by the tap, or a zero
//     page for padding - while the LDS image stays lane-linear.
//   * 3-slot LDS ring (3 x 48 KiB): the DMA of slices t+1 and t+2 stays in flight ACROSS the
//     barrier of slice t (raw s_barrier + counted s_waitcnt vmcnt(6), never vmcnt(0) in the loop),
//     one barrier per slice.
//   * LDS rows are 128 B; the 16-B slot index is XOR-swizzled with (row>>1)&7 on the SOURCE side
//     and on the ds_read_b128 side (cdna guide rule 21) so the 16 rows of a fragment read hit 16
//     distinct slots of the 256-B bank row.
//   * Rows (m) are enumerated so that a max-pool window is 4 (2x2) or 2 (2x1) consecutive rows.
//     In the 16x16 MFMA C layout a lane owns 4 consecutive rows of one column, so the pool is a
//     max over the lane's own accumulator registers - no extra pass, no extra HBM traffic.
//   * Epilogue: scale/bias/ReLU/pool in registers, tile transposed through the (now idle) LDS ring
//     and written with 16-B-per-lane stores, 256/512 contiguous bytes per output pixel.
//   * Workgroup ids are remapped so that the blocks resident on one XCD cover consecutive
//     (m-tile, all n-tiles): the n-tiles of an m-tile share their gathered A rows through that
//     XCD's L2, and vertically adjacent m-tiles share their halo rows.
//   * f16 operands use v_mfma_f32_16x16x32_f16; the exact-fp32 parity mode uses
//     v_mfma_f32_16x16x4_f32 on the same LDS image (k order permuted identically for A and W).
#include <stdlib.h>

#include <type_traits>

#include "igemm_common.h"

using namespace igemm;

namespace {

// Three tile shapes share one kernel body (8 waves each; every wave owns 64 output columns):
//   BN =  64, BM = 512: waves 8(M) x 1(N),  64x64 per wave, 2-slot ring of 72 KiB  (N <= 64: no half-empty N tile)
//   BN = 128, BM = 256: waves 4(M) x 2(N),  64x64 per wave, 3-slot ring of 48 KiB  (N <= 128)
//   BN = 256, BM = 256: waves 2(M) x 4(N), 128x64 per wave, 2-slot ring of 64 KiB  (N  > 128): 1.5x fewer L2->LDS
//             bytes and 25 % fewer LDS fragment reads per MFMA than the 128-wide tile.
//   code 1128: BN = 128, BM = 128: waves 4(M) x 2(N), 32x64 per wave, 4-slot ring of 32 KiB — for GEMMs with so few rows
//             (decoder steps: M = crops x beams) that the big tiles would leave most of the 256 CUs idle.
template <int BN_>
struct Cfg {
  static constexpr bool SMALL = BN_ / 1000 == 1;
  static constexpr int BN = BN_ % 1000;
  static constexpr int BM = SMALL ? 128 : ((BN == 64) ? 512 : 256);
  static constexpr int WN = BN / 64;             // waves along N
  static constexpr int WM = 8 / WN;              // waves along M
  static constexpr int MT = BM / WM / 16;        // 16-row MFMA tiles per wave along M (4 or 8)
  static constexpr bool KSPLIT = false;
  static constexpr int NSTAGE = (KSPLIT || SMALL) ? 4 : ((BN == 128) ? 3 : 2);
  static constexpr int HROWB = KSPLIT ? 64 : ROWB;        // bytes of one staged row
  static constexpr int A_BYTES = BM * HROWB;
  static constexpr int B_BYTES = BN * HROWB;
  static constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
  static constexpr int RPI = 64 * 16 / HROWB / 1;         // rows covered by one wave-instruction (8 or 16)
  static constexpr int ACHUNKS = BM * (HROWB / 16) / NTHREADS;   // A chunks staged per thread per (half-)slice
  static constexpr int WCHUNKS = BN * (HROWB / 16) / NTHREADS;   // W chunks
  static constexpr int GL = ACHUNKS + WCHUNKS;   // LDS-DMA instructions per wave per (half-)slice
  static constexpr int EPW = BN < 128 ? BN : 128;         // output columns per epilogue pass
  static constexpr int EPI_BYTES = BM * (EPW * 4 + 16);   // one epilogue pass, fp32 worst case
  static constexpr int RING_BYTES = NSTAGE * STAGE_BYTES;
  static constexpr int LDS_BYTES = RING_BYTES > EPI_BYTES ? RING_BYTES : EPI_BYTES;
};

// erf GELU (torch.nn.functional.gelu / nn.GELU()).  erf by Abramowitz & Stegun 7.1.26 — |error| <= 1.5e-7 absolute, i.e.
// about one fp32 ulp of (1 + erf) — branch-free: one v_rcp, one v_exp and five FMAs instead of ocml erff's two-branch
// polynomial (the fc1 GEMM applies this to 3072 columns of every token).
__device__ __forceinline__ float gelu_erf(float t) {
  // every multiply-add is spelled out: the epilogue exists in several copies (bounds-checked, straight-line) and a tile's values
  // must not depend on which copy wrote it — left to the compiler, the contraction of a * b + c may differ from copy to copy
  const float x = t * 0.70710678118654752f, ax = fabsf(x);
  const float u = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, ax, 1.f));
  float poly = __builtin_fmaf(1.061405429f, u, -1.453152027f);
  poly = __builtin_fmaf(poly, u, 1.421413741f);
  poly = __builtin_fmaf(poly, u, -0.284496736f);
  poly = __builtin_fmaf(poly, u, 0.254829592f);
  poly = poly * u;
  const float e = __builtin_fmaf(-poly, __builtin_amdgcn_exp2f((-1.4426950408889634f * ax) * ax), 1.f);
  const float h = 0.5f * t;
  return __builtin_fmaf(h, copysignf(e, x), h);
}

// sum over the 8 lanes of an aligned lane group without leaving the VALU (HIP's __shfl_xor is a ds_bpermute: an LDS-crossbar round
// trip per step): quad_perm [1,0,3,2], quad_perm [2,3,0,1], then row_half_mirror (lane i <-> 7 - i: the other quad's sum)
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float sum8(float v) {
  v += dpp_f<0xB1>(v);
  v += dpp_f<0x4E>(v);
  v += dpp_f<0x141>(v);
  return v;
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

template <typename T, int POOL, int BN_, bool DUAL, int EPI = EPI_NONE>
__global__ __launch_bounds__(NTHREADS) void conv_igemm_kernel(IgemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef Cfg<BN_> C;
  typedef typename Tr<T>::chunk_t chunk_t;
  constexpr int E = Tr<T>::E;            // elements per 16-B chunk
  constexpr int BKE = ROWB / sizeof(T);  // elements per K slice
  constexpr int MT = C::MT;
  constexpr int BM = C::BM;
  constexpr int A_BYTES = C::A_BYTES;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

  // ---- XCD-aware tile assignment (bijective for any grid size; speed only) ---------------
  int mt, nt;
  {
    const int nblk = gridDim.x, bid = blockIdx.x;
    const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7;
    const int L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    nt = L % p.ntiles;
    mt = L / p.ntiles;
  }
  const int m0 = mt * BM, n0 = nt * C::BN;
  if (p.stagger > 0 && blockIdx.x < 256 && (blockIdx.x & 8)) {       // experiment: half of the first round starts late
    for (int k = 0; k < p.stagger; ++k) __builtin_amdgcn_s_sleep(127);
  }

  // ---- staging set-up: each thread moves BM/64 A chunks + BN/64 W chunks per slice ---------
  // a wave-instruction covers RPI rows; a thread's q-th chunk sits RPI*8 rows further down
  constexpr int RPI = C::RPI, RSTEP = RPI * 8;
  const int srow = wave * RPI + (C::KSPLIT ? (lane >> 2) : (lane >> 3));   // row within a RSTEP-row group
  const int lchunk = C::KSPLIT ? ((lane & 3) ^ ((srow >> 1) & 3))          // logical chunk this lane fetches
                               : ((lane & 7) ^ ((srow >> 1) & 7));
  const char* a_src[C::ACHUNKS];
  int a_y[C::ACHUNKS], a_x[C::ACHUNKS];
  const char* w_src[C::WCHUNKS];
  // plain GEMM (see `stage_pure` below): pixel index = row index, no row decoding (three integer divisions per chunk)
  const bool pure = !DUAL && !C::KSPLIT && POOL == POOL_NONE && p.KH == 1 && p.KW == 1 && p.pad == 0 && p.pad_x == 0 &&
                    p.sy == 1 && (size_t)p.Ktot * sizeof(T) + ROWB <= (size_t)MHIP_ZERO_BYTES;
#pragma unroll
  for (int q = 0; q < C::ACHUNKS; ++q) {
    int m = m0 + q * RSTEP + srow;
    if (pure) {
      a_y[q] = a_x[q] = 0;
      a_src[q] = p.in + ((size_t)m * p.Cin + (size_t)lchunk * E) * sizeof(T);   // rows beyond M are redirected below
      continue;
    }
    int b = 0, y = -100000, x = -100000;  // invalid rows fail every bounds test
    if (m < p.M) decode_row<POOL>(p, m, b, y, x);
    a_y[q] = y;
    a_x[q] = x;
    if (m < p.M) {                       // input coordinates of tap (0,0)
      y = y * p.sy - p.pad;
      x = x - p.pad_x;
      a_y[q] = y;
      a_x[q] = x;
    }
    size_t pix = ((size_t)b * p.H + y) * p.W + x;  // may be out of bounds; every use is guarded
    if (DUAL) a_src[q] = (const char*)pix;  // 1x1 concat conv: keep the pixel index, pick the tensor per slice
    else a_src[q] = p.in + (pix * p.Cin + (size_t)lchunk * E) * sizeof(T);
  }
#pragma unroll
  for (int q = 0; q < C::WCHUNKS; ++q) {
    int n = n0 + q * RSTEP + srow;
    w_src[q] = (n < p.N) ? p.w + ((size_t)n * p.Ktot + (size_t)lchunk * E) * sizeof(T) : nullptr;
  }

  // Plain GEMMs (1x1, unpadded, unstrided, one input — every ViT / decoder product): slice `it` of chunk q is simply
  // a_src[q] + 128 it, so the pointers are advanced in place and a slice costs 2 VALU + 2 SALU per DMA instruction.  Rows
  // beyond M / N walk through the zero page instead (it holds a whole K row).  The general path below recomputes the tap
  // and the bounds of every chunk for every slice (~130 instructions per wave and slice, issued by both waves of a SIMD
  // right after the barrier, when nobody has MFMA work yet).
  const int wave_s = __builtin_amdgcn_readfirstlane(wave);
  if (pure) {
#pragma unroll
    for (int q = 0; q < C::ACHUNKS; ++q)
      if (m0 + q * RSTEP + srow >= p.M) a_src[q] = p.zeros + lchunk * 16;
#pragma unroll
    for (int q = 0; q < C::WCHUNKS; ++q)
      if (!w_src[q]) w_src[q] = p.zeros + lchunk * 16;
  }
  auto stage_pure = [&](int slot) {      // called once per (half-)slice, in K order
    char* la = smem + slot * C::STAGE_BYTES + wave_s * RPI * C::HROWB;
    char* lb = la + A_BYTES;
#pragma unroll
    for (int q = 0; q < C::ACHUNKS; ++q) {
      glds16(a_src[q], la + q * RSTEP * C::HROWB);
      a_src[q] += C::HROWB;
    }
#pragma unroll
    for (int q = 0; q < C::WCHUNKS; ++q) {
      glds16(w_src[q], lb + q * RSTEP * C::HROWB);
      w_src[q] += C::HROWB;
    }
  };
  // one DMA instruction of a (half-)slice: g < ACHUNKS -> A chunk g, else W chunk g - ACHUNKS.  The main loop issues them
  // BETWEEN groups of MFMAs: a wave that issues its 8 DMA instructions back to back sits in the vector-memory queue
  // behind the other waves' (64 KiB per slice drain at ~31 B/clk/CU: the second half of the waves measured ~2000 cycles
  // in `stage` before their first MFMA — profiles/r01/s_loop_phases.txt), and nothing overlaps that wait.
  auto stage_pure_chunk = [&](int slot, int g) {
    char* la = smem + slot * C::STAGE_BYTES + wave_s * RPI * C::HROWB;
    if (g < C::ACHUNKS) {
      glds16(a_src[g], la + g * RSTEP * C::HROWB);
      a_src[g] += C::HROWB;
    } else {
      const int q = g - C::ACHUNKS;
      glds16(w_src[q], la + A_BYTES + q * RSTEP * C::HROWB);
      w_src[q] += C::HROWB;
    }
  };
  auto stage_general = [&](int hs, int slot) {
    const int it = C::KSPLIT ? (hs >> 1) : hs;            // K slice
    const int hoff = C::KSPLIT ? (hs & 1) * 64 : 0;       // byte offset of the k-group half inside the slice
    int tap = it / p.cpt, cc = it - tap * p.cpt;
    int dy = (tap / p.KW) * p.dil, dx = (tap - (tap / p.KW) * p.KW) * p.dil;
    size_t a_off = (((size_t)dy * p.W + dx) * p.Cin + (size_t)cc * BKE) * sizeof(T) + hoff;
    size_t w_off = (size_t)it * BKE * sizeof(T) + hoff;
    char* la = smem + slot * C::STAGE_BYTES + wave_s * RPI * C::HROWB;
    char* lb = la + A_BYTES;
    // DUAL (KH = KW = 1, pad = 0): channel slice cc comes from `in` or from `in2`
    const int c0 = cc * BKE + lchunk * E + hoff / (int)sizeof(T);
    const bool first = c0 < p.Cin1;
    const char* dbase = first ? p.in : p.in2;
    const size_t dstride = (size_t)(first ? p.Cin1 : p.Cin - p.Cin1) * sizeof(T);
    const size_t dcol = (size_t)(first ? c0 : c0 - p.Cin1) * sizeof(T);
#pragma unroll
    for (int q = 0; q < C::ACHUNKS; ++q) {
      int yy = a_y[q] + dy, xx = a_x[q] + dx;
      bool ok = (unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)p.W;
      const char* src;
      if (DUAL) src = ok ? dbase + (size_t)a_src[q] * dstride + dcol : p.zeros;
      else src = ok ? a_src[q] + a_off : p.zeros;
      glds16(src, la + q * RSTEP * C::HROWB);
    }
#pragma unroll
    for (int q = 0; q < C::WCHUNKS; ++q) {
      const char* src = w_src[q] ? w_src[q] + w_off : p.zeros;
      glds16(src, lb + q * RSTEP * C::HROWB);
    }
  };
  auto stage = [&](int hs, int slot) {
    if (pure) stage_pure(slot);
    else stage_general(hs, slot);
  };

  // ---- accumulators ---------------------------------------------------------------------
  float4v acc[MT][4];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (float4v){0.f, 0.f, 0.f, 0.f};

  const int wr = wave / C::WN, wc = wave % C::WN;
  const int frow = lane & 15, fg = lane >> 4;
  // per-lane LDS byte offsets of k-group 0 (k-group 1 = offset ^ 64; KSPLIT: the other half-slice slot)
  int a_off0[MT], b_off0[4];
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    int ra = wr * (MT * 16) + t * 16 + frow;
    a_off0[t] = C::KSPLIT ? ra * 64 + ((fg ^ ((ra >> 1) & 3)) << 4) : ra * ROWB + ((fg ^ ((ra >> 1) & 7)) << 4);
  }
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    int rb = wc * 64 + t * 16 + frow;
    b_off0[t] = A_BYTES + (C::KSPLIT ? rb * 64 + ((fg ^ ((rb >> 1) & 3)) << 4) : rb * ROWB + ((fg ^ ((rb >> 1) & 7)) << 4));
  }

  // ---- main loop: NSTAGE-slot ring, GL LDS-DMA instructions per wave per (half-)slice -------
  constexpr int D = C::NSTAGE - 1;  // (half-)slices in flight ahead of the one being computed
  const int nsteps = C::KSPLIT ? 2 * p.nslices : p.nslices;
  if (nsteps > 0) stage(0, 0);
  if (D > 1 && nsteps > 1) stage(1, 1);
  if (D > 2 && nsteps > 2) stage(2, 2);
  int slot = 0, fill = D % C::NSTAGE;
  for (int it = 0; it < nsteps; ++it) {
    const int younger = min(D - 1, nsteps - 1 - it);   // (half-)slices issued after the one needed now
    if (C::GL == 6) {
      if (younger >= 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if (C::GL == 4 && D == 3) {
      if (younger >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else if (younger == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      static_assert(D == 1 || C::GL == 6 || (C::GL == 4 && D == 3), "counted vmcnt table");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    const bool feed = it + D < nsteps;
    const bool interleave = pure && feed;   // plain GEMMs: the next slice's DMA instructions go between this slice's MFMAs
    if (feed && !interleave) stage(it + D, fill);
    const char* sb = smem + slot * C::STAGE_BYTES;
#pragma unroll
    for (int s = 0; s < (C::KSPLIT ? 1 : 2); ++s) {
      chunk_t a[MT], b[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) b[t] = *(const chunk_t*)(sb + (b_off0[t] ^ (s << 6)));
#pragma unroll
      for (int t = 0; t < MT; ++t) a[t] = *(const chunk_t*)(sb + (a_off0[t] ^ (s << 6)));
      __builtin_amdgcn_s_setprio(1);  // keeps the MFMA cluster between the barriers (cdna guide T5)
      // plain GEMM: one DMA instruction of the next slice in front of every IL_EVERY-th MFMA of the slice (counted over
      // both k-groups), so that the waves' demand on the vector-memory queue stays near what it drains (~33 cycles per
      // 1 KiB instruction per CU) and no wave sits in the queue with its MFMAs behind it; the last one is issued early
      // enough to land under the rest of the slice.  One MFMA sequence for both cases — only the DMA instructions sit
      // behind a uniform branch.
      constexpr int NM = MT * 4, EVERY = C::KSPLIT ? (NM + C::GL - 1) / C::GL : ((MT == 8) ? 4 : (2 * NM + C::GL - 1) / C::GL / 2);
#pragma unroll
      for (int idx = 0; idx < NM; ++idx) {
        const int gidx = s * NM + idx;                     // position in the slice's MFMA sequence
        if (gidx % EVERY == 0 && gidx / EVERY < C::GL) {
          if (interleave) stage_pure_chunk(fill, gidx / EVERY);
        }
        Tr<T>::mma(a[idx / 4], b[idx % 4], acc[idx / 4][idx % 4]);
      }
      if (s == (C::KSPLIT ? 0 : 1) && interleave) {
#pragma unroll
        for (int g = ((C::KSPLIT ? 1 : 2) * NM + EVERY - 1) / EVERY; g < C::GL; ++g) stage_pure_chunk(fill, g);   // left-overs
      }
      __builtin_amdgcn_s_setprio(0);
    }
    slot = (slot == C::NSTAGE - 1) ? 0 : slot + 1;
    fill = (fill == C::NSTAGE - 1) ? 0 : fill + 1;
  }

  // ---- epilogue: scale/bias, activation, in-register max-pool -> wave-private LDS transpose -> coalesced NHWC stores ----
  // Each wave transposes its own 16-row x 64-column MFMA row-tile through a private 4 KiB fp32 staging area: 16
  // conflict-free ds_write_b32 (one per accumulator value) and then 16-byte reads of 4 consecutive columns of one row,
  // so a wave store covers 4 rows x 128 contiguous bytes (f16).  No workgroup barrier after the one that ends the main
  // loop, all 8 waves busy.  (History: staging the whole 256x128 tile with half the waves idle and two barriers per pass
  // cost 5 us per tile; before that an inlined erf-GELU made the epilogue instruction-fetch bound — profiles/r01/i_*.)
  constexpr int PF = (POOL == POOL_2x2) ? 4 : (POOL == POOL_2x1) ? 2 : 1;
  constexpr int RT = 16 / PF;          // staged rows per MFMA row-tile
  constexpr int SPW = 64 * 4 + 16;     // staged row pitch in bytes (272: rows 4 apart land 16 banks apart)
  const int Mq = p.M / PF;
  const int oe = p.out_f32 ? 4 : (int)sizeof(T);
  const size_t grow = (size_t)(p.ldc ? p.ldc : p.N) * oe;  // global bytes per output pixel
  const float lo = (p.relu == ACT_RELU && !p.res) ? 0.f : -INFINITY;   // ReLU without a residual: here, branch-free
  const float lo2 = (p.relu == ACT_RELU && p.res) ? 0.f : -INFINITY;   // with a residual: after the add
  const bool gelu = p.relu == ACT_GELU;
  // whole, 16-byte aligned column groups in every row — or a row pitch that leaves room for the last, partly filled group
  // that the caller has declared its own (pad_cols_writable — the logits GEMM: N = 50 265 columns in rows of 50 272; NOT a GEMM
  // that writes into a slice of a wider row, whose neighbours would be zeroed): the pad columns receive zeros (weight rows beyond N read the
  // zero page) instead of sending the whole GEMM down the element-wise path (measured 20 us of epilogue per tile there
  // against 4 us)
  const bool vec = (grow & 15) == 0 && ((p.N & 7) == 0 || (p.pad_ok && p.ldc >= ((p.N + 7) & ~7))) &&
                   (((unsigned long long)p.out | (unsigned long long)p.res) & 15) == 0;      // 16-byte stores / residual loads
  float sc4[4], bi4[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int n = n0 + wc * 64 + j * 16 + frow;
    sc4[j] = (p.scale && n < p.N) ? p.scale[n] : 1.f;
    bi4[j] = (p.bias && n < p.N) ? p.bias[n] : 0.f;
  }
  // LayerNorm folded around the GEMM (ConvDesc::epi): the vectors indexed by the GEMM column of this lane
  //   EPI_LN_ROWS: lnc_b = column sum of the folded weights;  EPI_LN_COLS: lnc_a = rstd, lnc_b = mean * rstd of token n
  float lnc_a[4] = {1.f, 1.f, 1.f, 1.f}, lnc_b[4] = {0.f, 0.f, 0.f, 0.f};
  if (EPI == EPI_LN_ROWS || EPI == EPI_LN_COLS) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wc * 64 + j * 16 + frow;
      if (n < p.N) {
        if (EPI == EPI_LN_ROWS) lnc_b[j] = p.ln_cs[n];
        else { lnc_a[j] = p.ln_a[n]; lnc_b[j] = p.ln_b[n]; }
      }
    }
  }
  // ... and the vectors indexed by the GEMM rows of this wave (MT * 16 of them): requested now, parked in a wave-private LDS
  // strip behind the staging areas once the ring is dead, read back 16 bytes at a time per row-tile (loading them where they are
  // used put a global-load latency in front of every row-tile)
  constexpr int WROWS = MT * 16, NRV = (WROWS + 63) / 64;
  float rv_a[NRV], rv_b[NRV];
  if (EPI == EPI_LN_ROWS || EPI == EPI_LN_COLS) {
#pragma unroll
    for (int k = 0; k < NRV; ++k) {
      const int m = m0 + wr * WROWS + k * 64 + lane;
      rv_a[k] = rv_b[k] = 0.f;
      if (k * 64 + lane < WROWS && m < p.M) {
        rv_a[k] = (EPI == EPI_LN_ROWS ? p.ln_a : p.ln_cs)[m];
        rv_b[k] = (EPI == EPI_LN_ROWS ? p.ln_b : p.row_bias)[m];
      }
    }
  }
  __syncthreads();                     // every wave is done reading the ring
  static_assert(EPI == EPI_NONE || 8 * (2 * 16 * SPW) + 8 * (2 * WROWS * 4) <= C::LDS_BYTES, "row-vector strip");
  float* rvs = (float*)(smem + 8 * (2 * 16 * SPW) + wave * (2 * WROWS * 4));
  if (EPI == EPI_LN_ROWS || EPI == EPI_LN_COLS) {
#pragma unroll
    for (int k = 0; k < NRV; ++k)
      if (k * 64 + lane < WROWS) { rvs[k * 64 + lane] = rv_a[k]; rvs[WROWS + k * 64 + lane] = rv_b[k]; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  char* wst0 = smem + wave * (2 * 16 * SPW);   // two private staging areas per wave: tile i+1 is written while tile i drains
  // One specialised copy of the row-tile loop per (output type, residual, GELU) combination, chosen once: the loop is
  // unrolled over the 8 row-tiles (accumulators are registers), so every uniform test left inside it is replicated and
  // the kernel's code outgrows the instruction cache two CUs share — which slows the neighbour's main loop as well.
  auto out_row = [&](int q) -> long long {
    return p.row_period ? (long long)(q / p.row_period) * p.row_stride + p.row_offset + q % p.row_period : (long long)q;
  };
  auto res_row = [&](int q) -> long long { return p.row_period ? (long long)(q % p.row_period) : (long long)q; };
  auto body = [&](auto OUT32_, auto RES_, auto GELU_, auto VEC_) {
    constexpr bool OUT32 = decltype(OUT32_)::value, RES = decltype(RES_)::value, GELU = decltype(GELU_)::value,
                   VEC = decltype(VEC_)::value;
    // EPI_SPLIT: row statistics of the passes this lane keeps (pass = 8 rows x this wave's 64 columns; lane j of an 8-lane
    // row group keeps pass j of every set of eight), stored once after the loop
    constexpr int NPASS = MT * ((RT * 8 + 63) / 64), NSET = (NPASS + 7) / 8;
    float keep_s[NSET], keep_q[NSET];
#pragma unroll
    for (int k = 0; k < NSET; ++k) keep_s[k] = keep_q[k] = 0.f;
    constexpr int NRES32 = RT * 16 / 64, NRES16 = (RT * 8 + 63) / 64;
    // row-tiles of residual requested ahead of the one being written.  Measured with 2 (tools/ubench/gemm_bench.hip, 369 280 rows):
    // no gain (proj 0.725 against 0.729 ms fp32, 0.748 against 0.720 ms split) at 26 more registers — these epilogues are bound by
    // the bytes of the stream (a proj GEMM moves 7.7 KB per row for 1.2 MFLOP: 3.9 TB/s on average, and every CU reaches its
    // epilogue at the same time), not by the latency of a request; 0 = requested at the top of their own row-tile
    constexpr int RES_AHEAD = 0;
    float4v rq32[RES && OUT32 && VEC ? MT : 1][RES && OUT32 && VEC ? NRES32 : 1];
    half8 rq16[RES && !OUT32 && VEC ? MT : 1][RES && !OUT32 && VEC ? NRES16 : 1];
    half8 rq16b[EPI == EPI_SPLIT ? MT : 1][EPI == EPI_SPLIT ? NRES16 : 1];
    auto load_res = [&](int i) {
      const int qb0 = (m0 + wr * (MT * 16) + i * 16) / PF;
      if (OUT32) {
#pragma unroll
        for (int t = 0; t < NRES32; ++t) {
          const int cidx = t * 64 + lane, row = cidx >> 4, ch = cidx & 15;
          const int q = qb0 + row, n = n0 + wc * 64 + ch * 4;
          rq32[RES && OUT32 && VEC ? i : 0][t] = (float4v){0.f, 0.f, 0.f, 0.f};
          if (q < Mq && n < p.N) rq32[RES && OUT32 && VEC ? i : 0][t] = *(const float4v*)(p.res + (size_t)res_row(q) * grow + (size_t)n * oe);
        }
      } else {
#pragma unroll
        for (int t = 0; t < NRES16; ++t) {
          const int cidx = t * 64 + lane, row = cidx >> 3, ch = cidx & 7;
          const int q = qb0 + row, n = n0 + wc * 64 + ch * 8;
          rq16[RES && !OUT32 && VEC ? i : 0][t] = (half8){0, 0, 0, 0, 0, 0, 0, 0};
          if (row < RT && q < Mq && n < p.N) rq16[RES && !OUT32 && VEC ? i : 0][t] = *(const half8*)(p.res + (size_t)res_row(q) * grow + (size_t)n * 2);
          if (EPI == EPI_SPLIT) {
            rq16b[EPI == EPI_SPLIT ? i : 0][t] = (half8){0, 0, 0, 0, 0, 0, 0, 0};
            if (row < RT && q < Mq && n < p.N) rq16b[EPI == EPI_SPLIT ? i : 0][t] = *(const half8*)(p.res2 + (size_t)res_row(q) * grow + (size_t)n * 2);
          }
        }
      }
    };
    if (RES && VEC) {
#pragma unroll
      for (int i = 0; i < RES_AHEAD && i < MT; ++i) load_res(i);
    }
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      char* wst = wst0 + (i & 1) * (16 * SPW);
      // The residual values of a row-tile are requested RES_AHEAD row-tiles before it is processed (load_res below): read next
      // to the store, every chunk of a wave paid a full load latency plus the drain of the previous store (vmcnt counts both), one
      // after the other; requested at the top of their own row-tile (rounds 1-2) the latency of one request per row-tile was
      // still in front of every row-tile — ~15 us of a 43 us proj tile (tools/ubench/gemm_bench.hip).  A chunk is read and
      // written by the same lane only, so in-place residuals (res == out) stay correct.
      if (RES && VEC && i + RES_AHEAD < MT) load_res(i + RES_AHEAD);
      float4v (&rres32)[RES && OUT32 && VEC ? NRES32 : 1] = rq32[RES && OUT32 && VEC ? i : 0];
      half8 (&rres16)[RES && !OUT32 && VEC ? NRES16 : 1] = rq16[RES && !OUT32 && VEC ? i : 0];
      half8 (&rres16b)[EPI == EPI_SPLIT ? NRES16 : 1] = rq16b[EPI == EPI_SPLIT ? i : 0];      // the low plane of a split residual
      // LayerNorm-folded consumers: the vectors indexed by the four GEMM rows this lane holds of the row-tile
      //   EPI_LN_ROWS: lr_a = rstd, lr_b = mean * rstd of token m;  EPI_LN_COLS: lr_a = row sum of the folded weights, lr_b = bias
      float4v lr_a = (float4v){0.f, 0.f, 0.f, 0.f}, lr_b = (float4v){0.f, 0.f, 0.f, 0.f};
      if (EPI == EPI_LN_ROWS || EPI == EPI_LN_COLS) {
        lr_a = *(const float4v*)(rvs + i * 16 + fg * 4);
        lr_b = *(const float4v*)(rvs + WROWS + i * 16 + fg * 4);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int lc = j * 16 + frow;
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (EPI == EPI_LN_ROWS) v[r] = __builtin_fmaf(acc[i][j][r], lr_a[r], __builtin_fmaf(-lr_b[r], lnc_b[j], bi4[j]));
          else if (EPI == EPI_LN_COLS) v[r] = __builtin_fmaf(acc[i][j][r], lnc_a[j], __builtin_fmaf(-lnc_b[j], lr_a[r], lr_b[r]));
          else v[r] = fmaxf(__builtin_fmaf(acc[i][j][r], sc4[j], bi4[j]), lo);
        }
        if (POOL == POOL_2x2) {
          lds_put<float>(wst, SPW, fg, lc, fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])));
        } else if (POOL == POOL_2x1) {
#pragma unroll
          for (int h = 0; h < 2; ++h) lds_put<float>(wst, SPW, fg * 2 + h, lc, fmaxf(v[2 * h], v[2 * h + 1]));
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) lds_put<float>(wst, SPW, fg * 4 + r, lc, v[r]);
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      const int qbase = (m0 + wr * (MT * 16) + i * 16) / PF;
      if (EPI == EPI_SPLIT) {
        // the residual stream as two f16 planes: x = acc * scale + bias + (res_hi + res_lo) in fp32, hi = f16(x), lo = f16(x - hi);
        // (sum, centred sum of squares) of the row over this wave's 64 columns ride along for the LayerNorm of the consumer
        half8 hv[NRES16], lv[NRES16];
        size_t ooff[NRES16];
        bool ook[NRES16];
#pragma unroll
        for (int t = 0; t < NRES16; ++t) {
          const int cidx = t * 64 + lane;
          const int row = cidx >> 3, ch = cidx & 7;
          const int q = qbase + row, n = n0 + wc * 64 + ch * 8;
          ook[t] = row < RT && q < Mq && n < p.N;
          ooff[t] = 0;
          float f[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
          if (ook[t]) {
            const float4v a0 = *(const float4v*)(wst + row * SPW + ch * 32), a1 = *(const float4v*)(wst + row * SPW + ch * 32 + 16);
            const half8 r8 = rres16[t], l8 = rres16b[t];
            f[0] = a0[0]; f[1] = a0[1]; f[2] = a0[2]; f[3] = a0[3]; f[4] = a1[0]; f[5] = a1[1]; f[6] = a1[2]; f[7] = a1[3];
#pragma unroll
            for (int k = 0; k < 8; ++k) f[k] += (float)r8[k] + (float)l8[k];
            ooff[t] = (size_t)out_row(q) * grow + (size_t)n * 2;
          }
          float sm = ((f[0] + f[1]) + (f[2] + f[3])) + ((f[4] + f[5]) + (f[6] + f[7]));
          sm = sum8(sm);
          const float mc = sm * (1.f / 64.f);
          float m2 = 0.f;
#pragma unroll
          for (int k = 0; k < 8; ++k) { const float d = f[k] - mc; m2 = __builtin_fmaf(d, d, m2); }
          m2 = sum8(m2);
          const int pi = i * NRES16 + t;
          if ((lane & 7) == (pi & 7)) { keep_s[pi >> 3] = sm; keep_q[pi >> 3] = m2; }
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            hv[t][k] = (_Float16)f[k];
            lv[t][k] = (_Float16)(f[k] - (float)hv[t][k]);
          }
        }
        unsigned long long dsth[NRES16], dstl[NRES16];
#pragma unroll
        for (int t = 0; t < NRES16; ++t) {
          dsth[t] = (unsigned long long)(p.out + ooff[t]);
          dstl[t] = (unsigned long long)(p.out2 + ooff[t]);
          asm volatile("" : "+v"(dsth[t]), "+v"(dstl[t]), "+v"(hv[t]), "+v"(lv[t]));
        }
#pragma unroll
        for (int t = 0; t < NRES16; ++t)
          if (ook[t]) {
            // as the other big tiles: non-temporal (at full batch the planes are gigabytes — they leave the caches before the next
            // GEMM reads them, and written through they leave the weights in the L2); the few-row tile's stay
            if (C::SMALL) {
              *(__attribute__((address_space(1))) half8*)dsth[t] = hv[t];
              *(__attribute__((address_space(1))) half8*)dstl[t] = lv[t];
            } else {
              __builtin_nontemporal_store(hv[t], (__attribute__((address_space(1))) half8*)dsth[t]);
              __builtin_nontemporal_store(lv[t], (__attribute__((address_space(1))) half8*)dstl[t]);
            }
          }
      } else if (VEC && !OUT32) {
        // f16 out: 8 columns per lane = one 16-byte store; 8 lanes cover a row's 64 columns (128 contiguous bytes).
        // All chunks of the row-tile are produced first and stored together: a store issued between two LDS reads that
        // reuse its data registers makes the compiler drain the store (vmcnt(0)) before every read.
        half8 ov[NRES16];
        size_t ooff[NRES16];
        bool ook[NRES16];
#pragma unroll
        for (int t = 0; t < NRES16; ++t) {
          const int cidx = t * 64 + lane;
          const int row = cidx >> 3, ch = cidx & 7;
          const int q = qbase + row, n = n0 + wc * 64 + ch * 8;
          ook[t] = row < RT && q < Mq && n < p.N;
          ooff[t] = 0;
          ov[t] = (half8){0, 0, 0, 0, 0, 0, 0, 0};
          if (ook[t]) {
            const float4v a0 = *(const float4v*)(wst + row * SPW + ch * 32), a1 = *(const float4v*)(wst + row * SPW + ch * 32 + 16);
            float f[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
            if (GELU)
#pragma unroll
              for (int k = 0; k < 8; ++k) f[k] = gelu_erf(f[k]);
            ooff[t] = (size_t)out_row(q) * grow + (size_t)n * 2;
            if (RES) {
              const half8 r8 = rres16[t];
#pragma unroll
              for (int k = 0; k < 8; ++k) f[k] = fmaxf(f[k] + (float)r8[k], lo2);
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) ov[t][k] = (_Float16)f[k];
          }
        }
        // every store of the group gets its own address and data registers, fixed BEFORE the first one is issued: with the
        // address formed next to each store the same register pair is recycled, and rewriting a register an in-flight
        // store still reads costs a full vmcnt(0) between consecutive stores
        unsigned long long dst16[NRES16];
#pragma unroll
        for (int t = 0; t < NRES16; ++t) {
          dst16[t] = (unsigned long long)(p.out + ooff[t]);
          asm volatile("" : "+v"(dst16[t]), "+v"(ov[t]));
        }
#pragma unroll
        for (int t = 0; t < NRES16; ++t)
          // global, not flat: the asm hides the provenance.  Non-temporal: the output is read next by another kernel, long after
          // it has left the L2, and kept out of it the weights stay (+3 % on the ViT GEMMs, tools/ubench/gemm_bench.hip)
          // The few-row tile's outputs (decoder steps) are small and read back at once: those stay in the L2.
          if (ook[t]) {
            if (C::SMALL) *(__attribute__((address_space(1))) half8*)dst16[t] = ov[t];
            else __builtin_nontemporal_store(ov[t], (__attribute__((address_space(1))) half8*)dst16[t]);
          }
      } else {
        float4v ov32[NRES32];
        size_t ooff32[NRES32];
        bool ook32[NRES32];
#pragma unroll
        for (int t = 0; t < NRES32; ++t) { ook32[t] = false; ooff32[t] = 0; ov32[t] = (float4v){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int t = 0; t < RT * 16 / 64; ++t) {
          const int cidx = t * 64 + lane;
          const int row = cidx >> 4, ch = cidx & 15;
          const int q = qbase + row, n = n0 + wc * 64 + ch * 4;
          float4v a = *(const float4v*)(wst + row * SPW + ch * 16);
          if (q < Mq && n < p.N) {
            if (GELU)
#pragma unroll
              for (int k = 0; k < 4; ++k) a[k] = gelu_erf(a[k]);
            const size_t off = (size_t)out_row(q) * grow + (size_t)n * oe;
            const size_t roff = (size_t)res_row(q) * grow + (size_t)n * oe;
            if (VEC) {     // fp32 out: kept in registers, stored after the loop (see the f16 branch)
              if (RES) {
                a += rres32[t];
#pragma unroll
                for (int k = 0; k < 4; ++k) a[k] = fmaxf(a[k], lo2);
              }
              ov32[t] = a;
              ooff32[t] = off;
              ook32[t] = true;
            } else {       // ragged N (e.g. the 95-class prediction layer): element-wise, everything decided at run time
#pragma unroll
              for (int k = 0; k < 4; ++k)
                if (n + k < p.N) {
                  float tv = a[k];
                  if (p.res) tv = fmaxf(tv + (oe == 4 ? *(const float*)(p.res + roff + k * 4) : (float)*(const T*)(p.res + roff + k * sizeof(T))), lo2);
                  if (oe == 4) *(float*)(p.out + off + k * 4) = tv;
                  else *(T*)(p.out + off + k * sizeof(T)) = (T)tv;
                }
            }
          }
        }
        if (VEC) {
          unsigned long long dst32[NRES32];
#pragma unroll
          for (int t = 0; t < NRES32; ++t) {
            dst32[t] = (unsigned long long)(p.out + (ook32[t] ? ooff32[t] : 0));
            asm volatile("" : "+v"(dst32[t]), "+v"(ov32[t]));
          }
#pragma unroll
          for (int t = 0; t < NRES32; ++t)
            if (ook32[t]) {
              if (C::SMALL) *(__attribute__((address_space(1))) float4v*)dst32[t] = ov32[t];
              else __builtin_nontemporal_store(ov32[t], (__attribute__((address_space(1))) float4v*)dst32[t]);
            }
        }
      }
    }
    if (EPI == EPI_SPLIT) {
      const int j = lane & 7, g = lane >> 3;
      if (n0 + wc * 64 < p.N) {
        const int c64 = (n0 + wc * 64) >> 6;
#pragma unroll
        for (int k = 0; k < NSET; ++k) {
          const int pass = k * 8 + j;
          if (pass < NPASS) {
            const int ii = pass / (NPASS / MT), tt = pass % (NPASS / MT);
            const int q = m0 + wr * (MT * 16) + ii * 16 + tt * 8 + g;
            if (q < Mq) {
              float* dst = p.stats + ((size_t)c64 * p.stats_ld + (size_t)out_row(q)) * 2;
              dst[0] = keep_s[k];
              dst[1] = keep_q[k];
            }
          }
        }
      }
    }
  };
  // EPI_SPLIT on a tile that lies entirely inside the output (all but the last row of tiles of a GEMM): the same arithmetic as
  // `body` as straight-line code — no bounds test, so no branch around any load or store, so the compiler counts its waits exactly
  // (with a branch per request it fell back to vmcnt(0) in front of the first use) — and the two planes of the residual are
  // requested SPLIT_AHEAD row-tiles before they are needed: the epilogue of a residual GEMM is bound by how many bytes a CU has in
  // flight (one row-tile per wave = 32 KiB per CU against ~2 us of loaded HBM latency: 15 us of a 43 us proj tile).
  auto split_fast = [&]() {
    constexpr int AH = 3;
    const int r8 = lane >> 3, c8 = lane & 7;
    const size_t lane_off = (size_t)(m0 + wr * (MT * 16) + r8) * grow + (size_t)(n0 + wc * 64 + c8 * 8) * 2;
    const char* rh = p.res + lane_off;
    const char* rl = p.res2 + lane_off;
    char* oh = p.out + lane_off;
    char* ol = p.out2 + lane_off;
    const size_t step8 = 8 * grow;
    half8 qh[MT][2], ql[MT][2];
    auto request = [&](int i) {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        qh[i][t] = *(const half8*)(rh + (size_t)(i * 2 + t) * step8);
        ql[i][t] = *(const half8*)(rl + (size_t)(i * 2 + t) * step8);
      }
    };
#pragma unroll
    for (int i = 0; i < AH && i < MT; ++i) request(i);
    constexpr int NPASS = MT * 2, NSET = (NPASS + 7) / 8;
    float keep_s[NSET], keep_q[NSET];
#pragma unroll
    for (int k = 0; k < NSET; ++k) keep_s[k] = keep_q[k] = 0.f;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      if (i + AH < MT) request(i + AH);
      char* wst = wst0 + (i & 1) * (16 * SPW);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) lds_put<float>(wst, SPW, fg * 4 + r, j * 16 + frow, fmaxf(__builtin_fmaf(acc[i][j][r], sc4[j], bi4[j]), lo));
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      half8 hv[2], lv[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int row = t * 8 + r8;
        const float4v a0 = *(const float4v*)(wst + row * SPW + c8 * 32), a1 = *(const float4v*)(wst + row * SPW + c8 * 32 + 16);
        float f[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
#pragma unroll
        for (int k = 0; k < 8; ++k) f[k] += (float)qh[i][t][k] + (float)ql[i][t][k];
        float sm = sum8(((f[0] + f[1]) + (f[2] + f[3])) + ((f[4] + f[5]) + (f[6] + f[7])));
        const float mc = sm * (1.f / 64.f);
        float m2 = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) { const float d = f[k] - mc; m2 = __builtin_fmaf(d, d, m2); }
        m2 = sum8(m2);
        const int pi = i * 2 + t;
        if ((lane & 7) == (pi & 7)) { keep_s[pi >> 3] = sm; keep_q[pi >> 3] = m2; }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          hv[t][k] = (_Float16)f[k];
          lv[t][k] = (_Float16)(f[k] - (float)hv[t][k]);
        }
      }
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        if (C::SMALL) {
          *(half8*)(oh + (size_t)(i * 2 + t) * step8) = hv[t];
          *(half8*)(ol + (size_t)(i * 2 + t) * step8) = lv[t];
        } else {
          __builtin_nontemporal_store(hv[t], (half8*)(oh + (size_t)(i * 2 + t) * step8));
          __builtin_nontemporal_store(lv[t], (half8*)(ol + (size_t)(i * 2 + t) * step8));
        }
      }
    }
    const int c64 = (n0 + wc * 64) >> 6;
#pragma unroll
    for (int k = 0; k < NSET; ++k) {
      const int pass = k * 8 + c8;
      if (pass < NPASS) {
        float* dst = p.stats + ((size_t)c64 * p.stats_ld + (size_t)(m0 + wr * (MT * 16) + pass * 8 + r8)) * 2;
        dst[0] = keep_s[k];
        dst[1] = keep_q[k];
      }
    }
  };
  // The same for the f16 outputs without a residual (q|k, V^T, fc1, every unpooled convolution into f16): `body` spends ~200
  // instructions per row-tile and wave on ~70 of arithmetic — bounds tests, a branch per chunk, 64-bit address arithmetic per
  // chunk, the row-period division — and both waves of a SIMD are in their epilogues together with the matrix cores idle.
  auto f16_fast = [&](auto GELU_) {
    constexpr bool GELU = decltype(GELU_)::value;
    const int r8 = lane >> 3, c8 = lane & 7;
    char* o = p.out + (size_t)(m0 + wr * (MT * 16) + r8) * grow + (size_t)(n0 + wc * 64 + c8 * 8) * 2;
    const size_t step8 = 8 * grow;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      char* wst = wst0 + (i & 1) * (16 * SPW);
      float4v lr_a = (float4v){0.f, 0.f, 0.f, 0.f}, lr_b = (float4v){0.f, 0.f, 0.f, 0.f};
      if (EPI == EPI_LN_ROWS || EPI == EPI_LN_COLS) {
        lr_a = *(const float4v*)(rvs + i * 16 + fg * 4);
        lr_b = *(const float4v*)(rvs + MT * 16 + i * 16 + fg * 4);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v;
          if (EPI == EPI_LN_ROWS) v = __builtin_fmaf(acc[i][j][r], lr_a[r], __builtin_fmaf(-lr_b[r], lnc_b[j], bi4[j]));
          else if (EPI == EPI_LN_COLS) v = __builtin_fmaf(acc[i][j][r], lnc_a[j], __builtin_fmaf(-lnc_b[j], lr_a[r], lr_b[r]));
          else v = fmaxf(__builtin_fmaf(acc[i][j][r], sc4[j], bi4[j]), lo);
          lds_put<float>(wst, SPW, fg * 4 + r, j * 16 + frow, v);
        }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      half8 ov[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int row = t * 8 + r8;
        const float4v a0 = *(const float4v*)(wst + row * SPW + c8 * 32), a1 = *(const float4v*)(wst + row * SPW + c8 * 32 + 16);
        float f[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
        if (GELU)
#pragma unroll
          for (int k = 0; k < 8; ++k) f[k] = gelu_erf(f[k]);
#pragma unroll
        for (int k = 0; k < 8; ++k) ov[t][k] = (_Float16)f[k];
      }
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        if (C::SMALL) *(half8*)(o + (size_t)(i * 2 + t) * step8) = ov[t];
        else __builtin_nontemporal_store(ov[t], (half8*)(o + (size_t)(i * 2 + t) * step8));
      }
    }
  };
  typedef std::true_type Y;
  typedef std::false_type N_;
  const bool has_res = p.res != nullptr;
  const bool full_tile = POOL == POOL_NONE && m0 + BM <= p.M && n0 + C::BN <= p.N && !p.row_period;
  if constexpr (EPI == EPI_SPLIT) {
    if (full_tile) split_fast();
    else body(N_{}, Y{}, N_{}, Y{});
  } else if constexpr (EPI == EPI_LN_ROWS) {
    if (full_tile) { if (gelu) f16_fast(Y{}); else f16_fast(N_{}); }
    else if (gelu) body(N_{}, N_{}, Y{}, Y{}); else body(N_{}, N_{}, N_{}, Y{});
  } else if constexpr (EPI == EPI_LN_COLS) {
    if (full_tile) f16_fast(N_{}); else body(N_{}, N_{}, N_{}, Y{});
  } else if (POOL == POOL_NONE && vec && oe == 2 && !has_res && full_tile) {
    if (gelu) f16_fast(Y{}); else f16_fast(N_{});
  } else if (!vec) {
    if (gelu) body(N_{}, N_{}, Y{}, N_{}); else body(N_{}, N_{}, N_{}, N_{});
  } else if (oe == 4) {
    if (gelu) body(Y{}, N_{}, Y{}, Y{}); else if (has_res) body(Y{}, Y{}, N_{}, Y{}); else body(Y{}, N_{}, N_{}, Y{});
  } else {
    if (gelu) body(N_{}, N_{}, Y{}, Y{}); else if (has_res) body(N_{}, Y{}, N_{}, Y{}); else body(N_{}, N_{}, N_{}, Y{});
  }
}

// the LayerNorm-folded epilogues (ConvDesc::epi): plain f16 GEMMs on the 256 x 256, 256 x 128 and 128 x 128 tiles
template <int BN_, int EPI>
int launch_epi(mhip_ctx* ctx, const IgemmArgs& a, int kid) {
  dim3 grid((unsigned)(a.mtiles * a.ntiles)), block(NTHREADS);
  const size_t lds = Cfg<BN_>::LDS_BYTES;
  static std::once_flag attr_set;
  std::call_once(attr_set, [&] {
    (void)hipFuncSetAttribute((const void*)conv_igemm_kernel<_Float16, POOL_NONE, BN_, false, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  });
  PROF_LAUNCH(ctx, kid, hipLaunchKernelGGL((conv_igemm_kernel<_Float16, POOL_NONE, BN_, false, EPI>), grid, block, lds, ctx->stream, a));
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return mhip_fail(ctx, MHIP_EHIP, "conv_igemm launch: %s", hipGetErrorString(e));
  return 0;
}

template <int BN_>
int launch_epi_any(mhip_ctx* ctx, const IgemmArgs& a, int kid) {
  switch (a.epi) {
    case EPI_LN_ROWS: return launch_epi<BN_, EPI_LN_ROWS>(ctx, a, kid);
    case EPI_LN_COLS: return launch_epi<BN_, EPI_LN_COLS>(ctx, a, kid);
    case EPI_SPLIT: return launch_epi<BN_, EPI_SPLIT>(ctx, a, kid);
    default: return mhip_fail(ctx, MHIP_EINVAL, "conv_igemm: unknown epilogue %d", a.epi);
  }
}

template <typename T, int BN_>
int launch_t(mhip_ctx* ctx, const IgemmArgs& a, int pool) {
  constexpr int KID = BN_ == 64 ? MHIP_K_IGEMM_T64 : (BN_ == 128 ? MHIP_K_IGEMM_T128 : MHIP_K_IGEMM_T256);
  if (a.epi != EPI_NONE) {
    if constexpr (std::is_same<T, _Float16>::value && BN_ != 64) return launch_epi_any<BN_>(ctx, a, KID);
    else return mhip_fail(ctx, MHIP_EINVAL, "conv_igemm: a LayerNorm-folded epilogue needs an f16 GEMM with N > 64");
  }
  dim3 grid((unsigned)(a.mtiles * a.ntiles)), block(NTHREADS);
  const size_t lds = Cfg<BN_>::LDS_BYTES;
  static std::once_flag attr_set;
  std::call_once(attr_set, [&] {
#define SETATTR(...) (void)hipFuncSetAttribute((const void*)__VA_ARGS__, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)
    SETATTR(conv_igemm_kernel<T, POOL_NONE, BN_, false>);
    SETATTR(conv_igemm_kernel<T, POOL_2x2, BN_, false>);
    SETATTR(conv_igemm_kernel<T, POOL_2x1, BN_, false>);
    SETATTR(conv_igemm_kernel<T, POOL_NONE, BN_, true>);
#undef SETATTR
  });
  if (a.in2) {
    PROF_LAUNCH(ctx, KID,
                hipLaunchKernelGGL((conv_igemm_kernel<T, POOL_NONE, BN_, true>), grid, block, lds, ctx->stream, a));
  } else {
    switch (pool) {
      case POOL_NONE:
        PROF_LAUNCH(ctx, KID,
                    hipLaunchKernelGGL((conv_igemm_kernel<T, POOL_NONE, BN_, false>), grid, block, lds, ctx->stream, a));
        break;
      case POOL_2x2:
        PROF_LAUNCH(ctx, KID,
                    hipLaunchKernelGGL((conv_igemm_kernel<T, POOL_2x2, BN_, false>), grid, block, lds, ctx->stream, a));
        break;
      default:
        PROF_LAUNCH(ctx, KID,
                    hipLaunchKernelGGL((conv_igemm_kernel<T, POOL_2x1, BN_, false>), grid, block, lds, ctx->stream, a));
        break;
    }
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return mhip_fail(ctx, MHIP_EHIP, "conv_igemm launch: %s", hipGetErrorString(e));
  return 0;
}

// the small 128x128 tile: plain (unpooled, single-input) convs / GEMMs only
template <typename T>
int launch_small(mhip_ctx* ctx, const IgemmArgs& a) {
  if (a.epi != EPI_NONE) {
    if constexpr (std::is_same<T, _Float16>::value) return launch_epi_any<1128>(ctx, a, MHIP_K_IGEMM_S128);
    else return mhip_fail(ctx, MHIP_EINVAL, "conv_igemm: a LayerNorm-folded epilogue needs an f16 GEMM");
  }
  dim3 grid((unsigned)(a.mtiles * a.ntiles)), block(NTHREADS);
  const size_t lds = Cfg<1128>::LDS_BYTES;
  static std::once_flag attr_set;
  std::call_once(attr_set, [&] {
    (void)hipFuncSetAttribute((const void*)conv_igemm_kernel<T, POOL_NONE, 1128, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  });
  PROF_LAUNCH(ctx, MHIP_K_IGEMM_S128,
              hipLaunchKernelGGL((conv_igemm_kernel<T, POOL_NONE, 1128, false>), grid, block, lds, ctx->stream, a));
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return mhip_fail(ctx, MHIP_EHIP, "conv_igemm launch: %s", hipGetErrorString(e));
  return 0;
}

}  // namespace

double mhip_conv_flops(const ConvDesc& d) {
  const int dil = d.dil > 0 ? d.dil : 1, sy = d.sy > 0 ? d.sy : 1, px = d.pad_x >= 0 ? d.pad_x : d.pad;
  int Ho = (d.H + 2 * d.pad - dil * (d.KH - 1) - 1) / sy + 1, Wo = d.W + 2 * px - dil * (d.KW - 1);
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
  a.dil = d.dil > 0 ? d.dil : 1;
  a.sy = d.sy > 0 ? d.sy : 1;
  a.pad_x = d.pad_x >= 0 ? d.pad_x : d.pad;
  a.res = (const char*)d.res;
  a.ldc = d.ldc;
  a.pad_ok = d.pad_cols_writable;
  a.row_period = d.row_period; a.row_stride = d.row_stride; a.row_offset = d.row_offset;
  if (d.row_period && (d.pool != POOL_NONE || d.row_period < 1 || d.row_stride < d.row_period))
    return mhip_fail(ctx, MHIP_EINVAL, "conv_igemm: periodic row mapping needs an unpooled GEMM and stride >= period");
  a.Ho = (d.H + 2 * d.pad - a.dil * (d.KH - 1) - 1) / a.sy + 1;
  a.Wo = d.W + 2 * a.pad_x - a.dil * (d.KW - 1);
  if (a.res && d.pool != POOL_NONE) return mhip_fail(ctx, MHIP_EINVAL, "conv_igemm: a residual needs an unpooled output");
  if (a.res && d.relu == ACT_GELU) return mhip_fail(ctx, MHIP_EINVAL, "conv_igemm: GELU is applied before any residual add");
  if (a.ldc && (a.ldc < d.N || d.pool != POOL_NONE))
    return mhip_fail(ctx, MHIP_EINVAL, "conv_igemm: bad output pitch %d", a.ldc);
  if ((a.res || a.ldc) && ((size_t)(a.ldc ? a.ldc : d.N) * (d.out_f32 ? 4 : esz)) % 16 != 0)
    return mhip_fail(ctx, MHIP_EINVAL, "conv_igemm: residual / pitched outputs need 16-byte aligned rows");
  static const int stagger_env = getenv("MARIE_HIP_STAGGER") ? atoi(getenv("MARIE_HIP_STAGGER")) : 0;
  a.stagger = stagger_env * (d.KH * d.KW * d.Cin / 64) / 16;      // in sixteenths of a slice count
  a.epi = d.epi;
  a.ln_a = d.ln_a; a.ln_b = d.ln_b; a.ln_cs = d.ln_cs; a.row_bias = d.row_bias;
  a.out2 = (char*)d.out2; a.res2 = (const char*)d.res2; a.stats = d.stats; a.stats_ld = d.stats_ld;
  if (d.epi != EPI_NONE) {
    const long long Mrows = (long long)d.B * d.H * d.W;
    if (precision != MHIP_PREC_F16 || d.KH != 1 || d.KW != 1 || d.pad != 0 || d.pool != POOL_NONE || d.in2 || d.out_f32 || d.ldc ||
        d.N % 8 != 0 || (d.epi != EPI_SPLIT && Mrows % 4 != 0) || (d.sy > 1))
      return mhip_fail(ctx, MHIP_EINVAL, "conv_igemm: a LayerNorm-folded epilogue needs a plain f16 GEMM (N %% 8 == 0, M %% 4 == 0)");
    if (d.epi == EPI_SPLIT && (!d.out2 || !d.res || !d.res2 || !d.stats || d.stats_ld <= 0 || d.N % 64 != 0 || d.relu != ACT_NONE))
      return mhip_fail(ctx, MHIP_EINVAL, "conv_igemm: the split epilogue needs both planes of output and residual, a statistics buffer and N %% 64 == 0");
    if (d.epi == EPI_LN_ROWS && (!d.ln_a || !d.ln_b || !d.ln_cs || d.res || d.row_period)) return mhip_fail(ctx, MHIP_EINVAL, "conv_igemm: EPI_LN_ROWS operands");
    if (d.epi == EPI_LN_COLS && (!d.ln_a || !d.ln_b || !d.ln_cs || !d.row_bias || d.res || d.relu != ACT_NONE || d.row_period))
      return mhip_fail(ctx, MHIP_EINVAL, "conv_igemm: EPI_LN_COLS operands");
    if (((unsigned long long)d.out | (unsigned long long)d.out2 | (unsigned long long)d.res | (unsigned long long)d.res2 |
         (unsigned long long)d.ln_a | (unsigned long long)d.ln_b | (unsigned long long)d.ln_cs | (unsigned long long)d.row_bias) & 15)
      return mhip_fail(ctx, MHIP_EINVAL, "conv_igemm: LayerNorm-folded epilogue operands must be 16-byte aligned");
  }
  a.in2 = (const char*)d.in2;
  a.Cin1 = d.in2 ? d.Cin1 : d.Cin;
  if (d.in2) {
    if (d.KH != 1 || d.KW != 1 || d.pad != 0 || d.pool != POOL_NONE)
      return mhip_fail(ctx, MHIP_EINVAL, "conv_igemm: a concatenated input needs a 1x1, unpadded, unpooled conv");
    if (d.Cin1 <= 0 || d.Cin1 >= d.Cin || d.Cin1 % bke != 0)
      return mhip_fail(ctx, MHIP_EINVAL, "conv_igemm: Cin1=%d must split Cin=%d on a multiple of %d", d.Cin1, d.Cin, bke);
  }
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
  if (M <= 0 || M > 0x7fffffffLL - 512) return mhip_fail(ctx, MHIP_EINVAL, "conv_igemm: M=%lld out of range", M);
  a.M = (int)M;
  a.N = d.N;
  a.Ktot = d.KH * d.KW * d.Cin;
  a.cpt = d.Cin / bke;
  a.nslices = d.KH * d.KW * a.cpt;
  a.relu = d.relu;
  a.out_f32 = d.out_f32;
  a.PH = a.PW = a.tiles_x = a.tiles_y = 0;
  const double fl = ctx->profiling ? mhip_conv_flops(d) : 0.0;
  if (ctx->profiling) ctx->prof[MHIP_K_CONV_IGEMM].flops += fl;
  if (d.epi == EPI_NONE) {
    const int r = mhip_try_launch_conv3x3_patch(ctx, precision, d, a);   // 3x3 / pad 1 / dense: patch kernel
    if (r == 0) ctx->prof[MHIP_K_IGEMM_PATCH].flops += fl;
    if (r <= 0) return r;
  }
  static const int force_bn = getenv("MARIE_HIP_FORCE_BN") ? atoi(getenv("MARIE_HIP_FORCE_BN")) : 0;   // tuning aid
  // (the LayerNorm-folded epilogues exist for the tiles of 128 columns and more)
  const int bn = force_bn ? force_bn : ((a.N > 128) ? 256 : ((a.N > 64 || d.epi != EPI_NONE) ? 128 : 64));
  const int bm = (bn == 64) ? 512 : 256;
  a.mtiles = (a.M + bm - 1) / bm;
  a.ntiles = (a.N + bn - 1) / bn;
  // too few 256 x 256 tiles to occupy the 256 CUs but enough 256 x 128 ones for one round (decoder-step GEMMs into 1024
  // columns: 7680 x 1024 = 120 / 240 tiles): the wider-than-tall tile stages 25 % fewer bytes per MFMA than 128 x 128
  static const bool mid_off = getenv("MARIE_HIP_NO_MID_TILE") != nullptr;      // A/B aid
  if (!mid_off && bn == 256 && a.mtiles * a.ntiles < 192 && a.mtiles * ((a.N + 127) / 128) >= 192 && !force_bn) {
    a.ntiles = (a.N + 127) / 128;
    ctx->prof[MHIP_K_IGEMM_T128].flops += fl;
    return precision == MHIP_PREC_F16 ? launch_t<_Float16, 128>(ctx, a, d.pool) : launch_t<float, 128>(ctx, a, d.pool);
  }
  // too few big tiles to occupy the 256 CUs (decoder-step GEMMs, heads on small maps): 128 x 128 tiles instead
  if (a.mtiles * a.ntiles < 192 && (a.N > 64 || d.epi != EPI_NONE) && d.pool == POOL_NONE && !d.in2) {
    a.mtiles = (a.M + 127) / 128;
    a.ntiles = (a.N + 127) / 128;
    ctx->prof[MHIP_K_IGEMM_S128].flops += fl;
    return precision == MHIP_PREC_F16 ? launch_small<_Float16>(ctx, a) : launch_small<float>(ctx, a);
  }
  ctx->prof[bn == 64 ? MHIP_K_IGEMM_T64 : (bn == 128 ? MHIP_K_IGEMM_T128 : MHIP_K_IGEMM_T256)].flops += fl;
  if (precision == MHIP_PREC_F16)
    return bn == 256 ? launch_t<_Float16, 256>(ctx, a, d.pool)
                     : (bn == 128 ? launch_t<_Float16, 128>(ctx, a, d.pool) : launch_t<_Float16, 64>(ctx, a, d.pool));
  return bn == 256 ? launch_t<float, 256>(ctx, a, d.pool)
                   : (bn == 128 ? launch_t<float, 128>(ctx, a, d.pool) : launch_t<float, 64>(ctx, a, d.pool));
}
