// EXPERIMENT (round 2) — NOT part of libmarie_hip.so.  A persistent, transposed-accumulator variant of the plain f16 GEMM that was
// built to remove the per-tile overhead of conv_igemm.hip; it computes the same results (checksums equal on the four ViT shapes)
// and is NOT faster: profiles/r02/f_gemm_tile_budget.txt has the measurements and what they say about where a tile's time goes
// (store instructions at ~75-100 cycles each per CU, both waves of a SIMD reading fragments at the same time, the clock the chip
// holds under load).  Kept as the record of that experiment and of its instrumentation (-DGB_STAMP: s_memtime stamps per slice).
// It was wired in through a hook in mhip_launch_conv_igemm (mhip_try_launch_gemm_f16, a kernel id of its own) that is not in
// the tree any more.
//
// gemm_f16.hip — persistent 256x256-tile f16 GEMM on the gfx950 matrix cores for the big plain products of the ViT encoders
// (qkv / proj / fc1 / fc2 of marie/boxes/dit/ditod/beit.py:89-341 and marie/models/unilm/trocr/deit.py:105-146, which the
// reference reaches through nn.Linear -> cuBLAS / oneDNN):
//
//   out[m][n] = act( scale[n] * sum_k A[m][k] W[n][k] + bias[n] ) (+ res[m][n]),   A [M][K], W [N][K], out / res [M][ldc], all f16
//
// Same operand staging as conv_igemm.hip's plain-GEMM path (128-byte K slices through a 2-slot LDS ring filled by LDS-DMA, XOR
// swizzle on the source side and on the fragment reads) — what differs is everything around the main loop, which for K = 768
// was 40 % of a tile's time there (profiles/r02/f_gemm_tile_budget.txt):
//   * one workgroup per CU walks a list of tiles.  The slices of all its tiles form ONE stream through the ring: the DMA of the
//     next tile's first slice is issued during this tile's last slice, the second one during the epilogue, so a tile starts
//     with its operands landed (no launch, no address set-up, no cold first slice per tile);
//   * the accumulators are held transposed (W rows are the MFMA's A operand): a lane owns 4 consecutive output COLUMNS of one
//     row per MFMA tile, and with the W rows of two MFMA tiles interleaved in groups of 4 it owns 8 consecutive columns —
//     one 16-byte store (and one 16-byte residual load) per (row tile, column pair) straight from registers.  No LDS
//     transpose, no barrier in the epilogue, the ring stays free for the next tile's operands;
//   * residual rows are requested right after the last MFMA and waited for together with the next tile's first slice, behind
//     the scale / bias pass;
//   * output stores are non-temporal (the output is read next by another kernel, long after it has left the L2; keeping it
//     from displacing the weights measured +3 %).
#include <stdlib.h>

#include "../../../marie_icr_amd/csrc/igemm_common.h"

using namespace igemm;

namespace {

typedef float float8v __attribute__((ext_vector_type(8)));

struct GemmArgs {
  const char* A;
  const char* W;
  const float* scale;
  const float* bias;
  char* out;
  const char* res;
  int M, N, K, ldc;
  int nslices, ntiles, total;
};

constexpr int BM = 256, BN = 256;
constexpr int A_BYTES = BM * ROWB, STAGE_BYTES = (BM + BN) * ROWB;   // 32 KiB + 32 KiB per slice
constexpr int SB_OFF = 2 * STAGE_BYTES;   // after the ring: [2 tiles][bias 256 floats | scale 256 floats]
constexpr int LDS_BYTES = SB_OFF + 2 * 2048;
constexpr int GL = 8;          // LDS-DMA instructions per wave and slice (4 A + 4 W)

__device__ __forceinline__ float gelu_erf(float t) {     // as conv_igemm.hip (Abramowitz & Stegun 7.1.26, |error| <= 1.5e-7)
  const float x = t * 0.70710678118654752f, ax = fabsf(x);
  const float u = __builtin_amdgcn_rcpf(1.f + 0.3275911f * ax);
  const float poly = ((((1.061405429f * u - 1.453152027f) * u + 1.421413741f) * u - 0.284496736f) * u + 0.254829592f) * u;
  const float e = 1.f - poly * __builtin_amdgcn_exp2f(-1.4426950408889634f * ax * ax);
  return 0.5f * t * (1.f + copysignf(e, x));
}

// A register load the compiler does not see (no entry in its s_waitcnt bookkeeping: beside LDS-DMA it would wait vmcnt(0) for an
// ordinary load, i.e. drain the ring).  The caller waits by hand and then passes the value through `landed` before any use.
__device__ __forceinline__ half8 gload16_hidden(const void* ptr) {
  half8 v;
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(ptr) : "memory");
  return v;
}
__device__ __forceinline__ void landed(half8& v) { asm volatile("" : "+v"(v)); }
// An LDS read the compiler does not see: a visible ds_read that follows LDS-DMA instructions gets an s_waitcnt vmcnt(0) in front
// (no alias information between the DMA's destination and the read), which here would wait for the residual rows as well.
__device__ __forceinline__ float4v lds_read16_hidden(const void* ptr) {
  float4v v;
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"((unsigned)(size_t)(const __attribute__((address_space(3))) void*)ptr) : "memory");
  return v;
}

#ifdef GB_TEMPORAL
#define GB_STORE(v, ptr) (*(ptr) = (v))
#else
#define GB_STORE(v, ptr) __builtin_nontemporal_store((v), (ptr))
#endif
#ifdef GB_STAMP
__device__ unsigned long long gb_stamps[256][16];
#define STAMP(x) unsigned long long x = __builtin_amdgcn_s_memtime()
#else
#define STAMP(x)
#endif

template <int ACT, bool RES>
__global__ __launch_bounds__(NTHREADS) void gemm_f16_kernel(GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef GB_STAMP
  unsigned long long st_loop = 0, st_pass = 0, st_wait = 0, st_store = 0, st_tiles = 0, st_drain = 0;
  unsigned long long sl0 = 0, sl1 = 0, sl2 = 0, sl3 = 0, sl4 = 0, sl5 = 0, sl6 = 0, sl7 = 0, slw = 0;
  STAMP(st_begin);
#endif
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // ---- this workgroup's tiles: the workgroups that share an XCD (blockIdx & 7) take a contiguous range of the tile list
  // (n-tile fastest) and walk it interleaved, so at any time they cover consecutive (m-tile, all n-tiles): one XCD's L2 serves
  // the A rows to all the n-tiles that need them.  Any grid size works (speed only).
  const int G = gridDim.x, xcd = blockIdx.x & 7;
  const int wx = (G - xcd + 7) >> 3;                       // workgroups on this XCD
  const int q8 = p.total >> 3, r8 = p.total & 7;
  const int Lend = xcd * q8 + min(xcd, r8) + q8 + (xcd < r8 ? 1 : 0);
  int L = xcd * q8 + min(xcd, r8) + (int)(blockIdx.x >> 3);
  if (L >= Lend) return;

  // ---- staging plan (as conv_igemm.hip): a wave-instruction moves 8 rows x 128 B; thread's chunk q sits 64 rows further down
  const int srow = wave * 8 + (lane >> 3);
  const int lane_off = (((lane & 7) ^ ((srow >> 1) & 7)) << 4);
  // source of chunk q = tile base (uniform) + K offset of the slice (uniform) + a per-lane 32-bit offset.  Rows beyond M / N
  // are clamped to the last one: what they produce is never stored, and every output row depends on its own A row only.
  unsigned a_voff[4], w_voff[4];
  const char* a_tile = nullptr;
  const char* w_tile = nullptr;
  const size_t rowb = (size_t)p.K * 2;
  int f_col = 0, f_par = 0;     // scale / bias of the cursor's tile: first column this lane copies, buffer parity
  auto set_feed = [&](int Lf) {
    const int nt = Lf % p.ntiles, mt = Lf / p.ntiles;
    f_col = min(nt * BN + 4 * lane, p.N - 4);
    a_tile = p.A + (size_t)mt * BM * rowb;
    w_tile = p.W + (size_t)nt * BN * rowb;
    const int mrem = p.M - mt * BM - 1, nrem = p.N - nt * BN - 1;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      a_voff[q] = (unsigned)min(q * 64 + srow, mrem) * (unsigned)rowb + lane_off;
      w_voff[q] = (unsigned)min(q * 64 + srow, nrem) * (unsigned)rowb + lane_off;
    }
  };
  // the feed cursor: next slice of the stream to issue
  int fL = L, fs = 0;
  bool feed_ok = true;
  set_feed(fL);
  auto advance = [&]() {
    if (++fs == p.nslices) {
      fs = 0;
      fL += wx;
      f_par ^= 1;
      feed_ok = fL < Lend;
      if (feed_ok) set_feed(fL);
    }
  };
  auto stage_chunk = [&](int slot, int g) {       // one of the GL instructions of the cursor's slice
    char* la = smem + slot * STAGE_BYTES + wave * (8 * ROWB);
    if (g == 0 && fs == 0) {
      // a tile's bias / scale columns travel with its first slice (one 1 KiB instruction each, waves 0 and 1): they are in
      // LDS a whole tile before the epilogue reads them, and no register load sits in the compiler's wait bookkeeping
      if (wave == 0 && p.bias) glds16(p.bias + f_col, smem + SB_OFF + f_par * 2048);
      if (wave == 1 && p.scale) glds16(p.scale + f_col, smem + SB_OFF + f_par * 2048 + 1024);
    }
    const unsigned koff = (unsigned)fs * ROWB;
    if (g < 4) glds16(a_tile + (a_voff[g] + koff), la + g * (64 * ROWB));
    else glds16(w_tile + (w_voff[g - 4] + koff), la + A_BYTES + (g - 4) * (64 * ROWB));
  };

  // ---- fragment read offsets.  wave = (wr, wc): rows wr*128.., columns wc*64..  A row tile i: rows 16 i + (lane & 15).
  // W tile j = 2 jj + h: MFMA row rho <-> column 32 jj + 8 (rho >> 2) + 4 h + (rho & 3), so that lane (n, g) ends up with
  // columns 32 jj + 8 g + [0, 8) of row n in acc[i][2 jj] (h = 0: first four) and acc[i][2 jj + 1] (last four).
  const int wr = wave >> 2, wc = wave & 3;
  const int frow = lane & 15, fg = lane >> 4;
  // (ra >> 1) & 7 does not depend on the row tile i (16 i >> 1 = 8 i), nor (rb >> 1) & 7 on jj: one register per operand and
  // k-group, the tiles are immediate offsets (2 KiB per A row tile, 4 KiB per W column pair)
  int a_off, b_off[2];
  {
    const int ra = wr * 128 + frow;
    a_off = ra * ROWB + ((fg ^ ((ra >> 1) & 7)) << 4);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int rb = wc * 64 + 8 * (frow >> 2) + 4 * h + (frow & 3);
      b_off[h] = A_BYTES + rb * ROWB + ((fg ^ ((rb >> 1) & 7)) << 4);
    }
  }

  // ---- prologue: first slice landed, second in flight
#pragma unroll
  for (int g = 0; g < GL; ++g) stage_chunk(0, g);
  advance();
  __builtin_amdgcn_s_waitcnt(0x0f70);      // vmcnt(0)
  __builtin_amdgcn_s_barrier();
  if (feed_ok) {
#pragma unroll
    for (int g = 0; g < GL; ++g) stage_chunk(1, g);
    advance();
  }

#ifdef GB_STAGGER
  {
    const unsigned long long t_begin = __builtin_amdgcn_s_memtime();
    const unsigned long long delay = (unsigned long long)((blockIdx.x >> 3) % GB_STAGGER) * (unsigned)(p.nslices * 3000 + 8000) / GB_STAGGER;
    while (__builtin_amdgcn_s_memtime() - t_begin < delay) __builtin_amdgcn_s_sleep(8);
  }
#endif
  int sigma = 0;        // stream index of the slice being computed: slot = sigma & 1
  int c_par = 0;        // scale / bias buffer of the tile being computed
  const size_t growb = (size_t)p.ldc * 2;

  // ---- the previous tile's output, rounded, waits here for its stores: a CU takes ~75 cycles per store instruction whatever
  // its width (128 of them per tile: ~10k cycles against the 25k of a 12-slice tile's MFMAs), and issued as one burst at
  // the tile's end they hold every wave in the vector-memory queue with no MFMA running.  Two per slice between the MFMAs of
  // the next tile cost nothing.  dout[2 i + jj] = columns 32 jj + 8 g + [0, 8) of row 16 i + (lane & 15).
  // Only the second column half waits (32 registers; both halves do not fit beside the accumulators): the first is stored at the
  // tile's end between the DMA instructions.
  half8 dout[8];
  char* dbase = nullptr;         // uniform: the wave's first row, second column half; dout[i] is 16 i rows further down
  int d_rows = 0;                // how many of the 8 row tiles are inside the matrix for this lane (0: nothing waits)
  const unsigned d_lane = (unsigned)(lane & 15) * (unsigned)growb + 16u * (lane >> 4);
  auto dstore = [&](int k, const half8& v) {
    if (k < d_rows) GB_STORE(v, (__attribute__((address_space(1))) half8*)(dbase + (size_t)k * 16 * growb + d_lane));
  };
  // static register indices: a run-time index would put dout in scratch
  auto dstore_one = [&](int kk) {
    if (kk == 0) dstore(0, dout[0]);
    else if (kk == 1) dstore(1, dout[1]);
    else if (kk == 2) dstore(2, dout[2]);
    else if (kk == 3) dstore(3, dout[3]);
    else if (kk == 4) dstore(4, dout[4]);
    else if (kk == 5) dstore(5, dout[5]);
    else if (kk == 6) dstore(6, dout[6]);
    else if (kk == 7) dstore(7, dout[7]);
  };

#pragma unroll 1
  for (; L < Lend; L += wx) {
    const int nt = L % p.ntiles, mt = L / p.ntiles;
    STAMP(t0);
    float4v acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = (float4v){0.f, 0.f, 0.f, 0.f};

#pragma unroll 1
    for (int t = 0; t < p.nslices; ++t, ++sigma) {
      // t == 0: the slice landed before the previous tile's epilogue (or in the prologue) and its successor is already on
      // its way.  t > 0: the slice issued one iteration ago has to land; the barrier also frees the slot before it
      const bool feed = feed_ok && t > 0;
      STAMP(ts_);
      if (t > 0) {
        __builtin_amdgcn_s_waitcnt(0x0f70);
        __builtin_amdgcn_s_barrier();
      }
#ifdef GB_STAMP
      { STAMP(tw_); slw += tw_ - ts_; }
#endif
      const char* sb = smem + (sigma & 1) * STAGE_BYTES;
      const int fill = (sigma + 1) & 1;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        // the A fragments are read two row tiles ahead of their MFMAs (12 registers instead of 32 for all eight: the
        // difference is part of what lets the previous tile's output wait in registers)
        half8 a[3], b[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = *(const half8*)(sb + (b_off[j & 1] ^ (s << 6)) + (j >> 1) * (32 * ROWB));
        a[0] = *(const half8*)(sb + (a_off ^ (s << 6)));
        a[1] = *(const half8*)(sb + (a_off ^ (s << 6)) + 16 * ROWB);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          if (i + 2 < 8) a[(i + 2) % 3] = *(const half8*)(sb + (a_off ^ (s << 6)) + (i + 2) * (16 * ROWB));
          // the next slice's DMA instructions go between the MFMAs of the first k-group, one per row tile (conv_igemm.hip:
          // issued back to back, a wave waits in the vector-memory queue with its MFMAs behind it); the previous tile's
          // stores between those of the second
          if (s == 0) {
            if (feed) stage_chunk(fill, i);
          } else if (i == 2) {
            if (!RES && dbase) dstore_one(t);          // 8 stores over slices 0..7; what a short K leaves is flushed at the tile's end
          }
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[j], a[i % 3], acc[i][j], 0, 0, 0);
        }
        __builtin_amdgcn_s_setprio(0);
      }
      if (feed) advance();
#ifdef GB_STAMP
      {
        STAMP(te);
        const unsigned long long d = te - ts_;
        if (t == 0) sl0 += d; else if (t == 1) sl1 += d; else if (t == 2) sl2 += d; else if (t == 3) sl3 += d;
        else if (t == 4) sl4 += d; else if (t == 5) sl5 += d; else if (t == 6) sl6 += d; else sl7 += d;
      }
#endif
    }
    if (!RES && dbase) {                   // K < 512: the rest of the previous tile's stores
#pragma unroll 1
      for (int kk = p.nslices; kk < 8; ++kk) dstore_one(kk);
    }

    // ---- tile end, in two column halves (jj: the lane's columns 32 jj + 8 g + [0, 8) of each row) so that the registers of
    // one half's residual rows fit beside the accumulators:
    //   residual rows of half 0 -> scale / bias / activation pass (the loads have that long to arrive) -> the wait that also
    //   covers the next tile's first slice -> the barrier that frees the last slice's slot -> half 0 rounded -> residual rows
    //   of half 1 -> the DMA of the stream's next slice -> half 1 rounded.  Nothing is stored here (see dout above).
    STAMP(t1);
    const int row0 = mt * BM + wr * 128 + frow;                      // + 16 i
    const int col0 = nt * BN + wc * 64 + 8 * fg;                     // + 32 jj
    half8 rres0[RES ? 8 : 1], rres1[RES ? 8 : 1];
    // addresses = a uniform base (the wave's first row and column) + a 32-bit per-lane offset: no 64-bit address registers per
    // row tile.  Residual rows: hidden loads, rows clamped to the last one (what a row beyond the edge reads is never stored)
    const size_t wave_off = (size_t)(mt * BM + wr * 128) * growb + (size_t)(nt * BN + wc * 64) * 2;
    const int mrem = p.M - 1 - (mt * BM + wr * 128);          // last valid row of the wave's 128 (may be negative or > 127)
    const bool cols_ok0 = col0 < p.N, cols_ok1 = col0 + 32 < p.N;
    auto load_res = [&](half8 (&r)[RES ? 8 : 1], int jj) {
      const char* rbase = p.res + wave_off;
      const unsigned coff = (jj ? (cols_ok1 ? 64u : 0u) : 0u) + 16u * fg;
#pragma unroll
      for (int i = 0; i < (RES ? 8 : 0); ++i)
        r[i] = gload16_hidden(rbase + ((unsigned)max(0, min(16 * i + frow, mrem)) * (unsigned)growb + coff));
    };
    if (RES) load_res(rres0, 0);
    const float* sbuf = (const float*)(smem + SB_OFF + c_par * 2048);
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const int lc = wc * 64 + 32 * jj + 8 * fg;       // tile-local column
      float4v sc[2] = {(float4v){1.f, 1.f, 1.f, 1.f}, (float4v){1.f, 1.f, 1.f, 1.f}};
      float4v bi[2] = {(float4v){0.f, 0.f, 0.f, 0.f}, (float4v){0.f, 0.f, 0.f, 0.f}};
      if (p.bias) { bi[0] = lds_read16_hidden(sbuf + lc); bi[1] = lds_read16_hidden(sbuf + lc + 4); }
      if (p.scale) { sc[0] = lds_read16_hidden(sbuf + 256 + lc); sc[1] = lds_read16_hidden(sbuf + 256 + lc + 4); }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      asm volatile("" : "+v"(sc[0]), "+v"(sc[1]), "+v"(bi[0]), "+v"(bi[1]));
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float v = acc[i][2 * jj + (e >> 2)][e & 3] * sc[e >> 2][e & 3] + bi[e >> 2][e & 3];
          if (ACT == ACT_GELU) v = gelu_erf(v);
          if (ACT == ACT_RELU && !RES) v = fmaxf(v, 0.f);
          acc[i][2 * jj + (e >> 2)][e & 3] = v;
        }
    }
    c_par ^= 1;
    STAMP(t2);
    __builtin_amdgcn_s_waitcnt(0x0f70);      // next tile's first slice (and the residual rows of half 0) have landed
    __builtin_amdgcn_s_barrier();            // every wave is done with the last slice: its slot takes the stream's next one
    STAMP(t3);
    const int fill = (sigma + 1) & 1;        // sigma already points at the next tile's first slice
    // one column half of a row tile: residual, rounding
    auto finish = [&](int i, int jj, const half8& r) {
      half8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float v = acc[i][2 * jj + (e >> 2)][e & 3];
        if (RES) {
          v += (float)r[e];
          if (ACT == ACT_RELU) v = fmaxf(v, 0.f);
        }
        o[e] = (_Float16)v;
      }
      return o;
    };
    if (RES) {
#pragma unroll
      for (int i = 0; i < 8; ++i) landed(rres0[i]);
    }
    half8 out0[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) out0[i] = finish(i, 0, rres0[RES ? i : 0]);
    if (RES) {
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(out0[i]));     // rounded before the next loads are issued
      load_res(rres1, 1);
    }
    char* obase = p.out + wave_off;
    const unsigned lane_out = (unsigned)frow * (unsigned)growb + 16u * fg;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (feed_ok) stage_chunk(fill, i);
      if (16 * i + frow <= mrem && cols_ok0)
        GB_STORE(out0[i], (__attribute__((address_space(1))) half8*)(obase + (size_t)i * 16 * growb + lane_out));
    }
    if (RES) {
      // With a residual the second half is stored here as well (its rows and the waiting output do not both fit in registers).
      // Younger than the residual rows of half 1 are the >= 8 DMA instructions just issued (when the stream goes on) and the
      // stores of half 0: "all but the 8 youngest" covers the residual rows
      if (feed_ok) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 0; i < 8; ++i) landed(rres1[i]);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const half8 o = finish(i, 1, rres1[i]);
        if (16 * i + frow <= mrem && cols_ok1)
          GB_STORE(o, (__attribute__((address_space(1))) half8*)(obase + (size_t)i * 16 * growb + (lane_out + 64)));
      }
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) dout[i] = finish(i, 1, rres1[0]);
      dbase = obase + 64;
      d_rows = cols_ok1 ? max(0, min(8, (mrem - frow + 16) >> 4)) : 0;
    }
    if (feed_ok) advance();
#ifdef GB_STAMP
    STAMP(t4);
    st_loop += t1 - t0; st_pass += t2 - t1; st_wait += t3 - t2; st_store += t4 - t3; st_tiles += 1;
#endif
  }
  // the last tile's output
  if (!RES) {
#pragma unroll 1
    for (int kk = 0; kk < 8; ++kk) dstore_one(kk);
  }
#ifdef GB_STAMP
  if (tid == 0 && blockIdx.x < 256) {
    STAMP(st_end);
    unsigned long long* o = gb_stamps[blockIdx.x];
    o[8] = sl0; o[9] = sl1; o[10] = sl2; o[11] = sl3; o[12] = sl4; o[13] = sl5; o[14] = sl6; o[15] = sl7; o[7] = slw;
    o[0] = st_end - st_begin; o[1] = st_loop; o[2] = st_pass; o[3] = st_wait; o[4] = st_store; o[5] = st_tiles; o[6] = st_drain;
  }
#endif
}

template <int ACT, bool RES>
void launch(mhip_ctx* ctx, const GemmArgs& a, int grid) {
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)gemm_f16_kernel<ACT, RES>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    attr = true;
  }
  PROF_LAUNCH(ctx, MHIP_K_IGEMM_T256,
              hipLaunchKernelGGL((gemm_f16_kernel<ACT, RES>), dim3(grid), dim3(NTHREADS), LDS_BYTES, ctx->stream, a));
}

}  // namespace

#ifdef GB_STAMP
void gb_print_stamps() {
  static unsigned long long h[256][16];
  (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(gb_stamps), sizeof(h));
  double s[16] = {0};
  for (int b = 0; b < 256; ++b) for (int k = 0; k < 16; ++k) s[k] += (double)h[b][k];
  printf("      slices t=0..6, rest (cycles per tile): %.0f %.0f %.0f %.0f %.0f %.0f %.0f | %.0f ; of which wait+barrier %.0f\n", s[8] / s[5], s[9] / s[5], s[10] / s[5],
         s[11] / s[5], s[12] / s[5], s[13] / s[5], s[14] / s[5], s[15] / s[5], s[7] / s[5]);
  if (s[6] > 0) printf("      drain after the stores: %.0f cycles per tile\n", s[6] / s[5]);
  printf("      stamps (cycles per tile, wave 0): total %.0f  loop %.0f  scale/act pass %.0f  wait+barrier %.0f  finish+stores %.0f   (%.1f tiles per WG)\n",
         s[0] / s[5], s[1] / s[5], s[2] / s[5], s[3] / s[5], s[4] / s[5], s[5] / 256);
}
#endif

// Plain f16 GEMMs with enough 256 x 256 tiles to occupy the chip; returns 1 if the call is not eligible (the caller goes on to
// the generic kernel), 0 on launch, negative on error.
int mhip_try_launch_gemm_f16(mhip_ctx* ctx, int precision, const ConvDesc& d, const igemm::IgemmArgs& ia) {
  static const bool off = getenv("MARIE_HIP_NO_GEMM_F16") != nullptr;     // A/B aid
  if (off || precision != MHIP_PREC_F16) return 1;
  if (d.KH != 1 || d.KW != 1 || d.pad != 0 || ia.pad_x != 0 || ia.sy != 1 || d.in2 || d.pool != POOL_NONE || d.out_f32 ||
      d.row_period)
    return 1;
  if (d.N <= 128 || (d.N & 7) || (d.Cin & 63) || d.Cin > (1 << 20)) return 1;
  if (d.relu != ACT_NONE && d.relu != ACT_RELU && d.relu != ACT_GELU) return 1;
  if (d.res && d.relu == ACT_GELU) return 1;
  const int ldc = d.ldc ? d.ldc : d.N;
  if ((ldc & 7) || (((unsigned long long)d.out | (unsigned long long)d.res | (unsigned long long)d.in | (unsigned long long)d.w) & 15))
    return 1;
  if (((unsigned long long)d.scale | (unsigned long long)d.bias) & 31) return 1;      // 32-byte scale / bias reads
  GemmArgs a;
  a.A = (const char*)d.in; a.W = (const char*)d.w; a.scale = d.scale; a.bias = d.bias; a.out = (char*)d.out;
  a.res = (const char*)d.res;
  a.M = ia.M; a.N = d.N; a.K = d.Cin; a.ldc = ldc;
  a.nslices = d.Cin / 64;
  a.ntiles = (d.N + BN - 1) / BN;
  const long long total = (long long)((ia.M + BM - 1) / BM) * a.ntiles;
  if (total < 192 || total > 0x7fffffff) return 1;
  a.total = (int)total;
  static int cus = 0;
  if (!cus) {
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device) != hipSuccess || cus < 8) cus = 256;
    cus &= ~7;
  }
  const int grid = a.total < cus ? a.total : cus;
  if (d.res) {
    if (d.relu == ACT_RELU) launch<ACT_RELU, true>(ctx, a, grid);
    else launch<ACT_NONE, true>(ctx, a, grid);
  } else if (d.relu == ACT_GELU) {
    launch<ACT_GELU, false>(ctx, a, grid);
  } else if (d.relu == ACT_RELU) {
    launch<ACT_RELU, false>(ctx, a, grid);
  } else {
    launch<ACT_NONE, false>(ctx, a, grid);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return mhip_fail(ctx, MHIP_EHIP, "gemm_f16 launch: %s", hipGetErrorString(e));
  return 0;
}
