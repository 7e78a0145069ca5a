// cross_attn.hip — the TrOCR decoder's encoder-attention with the key / value projections absorbed (f16 mode).
//
// What it replaces: fairseq MultiheadAttention(encoder_attn) of every decoder layer as TextRecognitionGenerator drives it
// (marie/models/unilm/trocr/generator.py:127-362 -> TransformerDecoderLayer.encoder_attn), i.e. per layer, crop and step
//     K = E W_k^T + b_k,  V = E W_v^T + b_v          (E: the crop's 577 encoder tokens of width ED, the SAME for all layers)
//     o_h = softmax_s(q_h . K_h[s]) V_h                (q pre-scaled by head_dim^-0.5, one query per beam and head)
// The straightforward kernel (trocr_ops.hip decode_attn) is HBM-bound on K and V: 2 x 577 x 1024 f16 = 2.36 MB per crop, layer
// and step, after two projection GEMMs per layer.  Algebraically
//     q_h . K_h[s]  =  (W_k,h^T q_h) . E[s] + q_h . b_k,h           the second term is constant over s: softmax drops it
//     o_h           =  W_v,h (sum_s p[s] E[s]) + b_v,h              (sum_s p[s] = 1)
// so attention can run over E itself with "absorbed" queries qt_h = W_k,h^T q_h (ED numbers per beam and head) and produce
// contexts ct_h = sum_s p[s] E[s] that W_v,h maps back: 0.89 MB per crop, layer and step instead of 2.36, no K / V tensors
// and no projection GEMMs.  Three kernels:
//   absorb_q    qt[r][h][:]  = W_k,h^T q[r][h*64 : h*64+64]                                 (grouped GEMM, K = 64)
//   cross_attn  one workgroup per crop, one wave per beam: S^T = E Qt^T and Ct^T = E^T P^T on the matrix cores
//   absorb_v    ao[r][h*64 : h*64+64] = W_v,h ct[r][h][:] + b_v,h                           (grouped GEMM, K = ED)
//
// cross_attn, MI355X shape: E tiles of 32 keys x ED (48 KiB at ED = 768) go HBM -> LDS by LDS-DMA into a 3-slot ring (two
// tiles in flight under the one being used); the tile is read twice from LDS — by rows (ds_read_b128: A operand of
// S^T = E Qt^T, contraction over ED) and by columns (ds_read_b64_tr_b16: A operand of Ct^T = E^T P^T, contraction over keys) —
// from ONE image: 16-byte chunks are XOR-swizzled inside every 256-byte group with x(row) = ((row & 7) << 1) | ((row >> 3) & 1),
// applied on the DMA source side, which keeps both kinds of read (nearly) conflict-free.  S^T is computed transposed so that a
// query is a lane column: the online soft-max is a per-lane loop plus two cross-lane max steps, keeps a stale reference (rescale
// only when a tile exceeds it by 2^8) and the probabilities a lane holds are, converted to f16, directly the B operand of the
// second product.  The wave's 16 queries are the 16 heads of its beam; Qt fragments (ED/32 x 4 VGPRs) live in registers, the
// ED x 16 fp32 context accumulators (ED/4 registers) in AGPRs.  Bound: HBM (0.89 MB per crop against 85 MFLOP).
#include <math.h>

#include "common.h"

namespace {

typedef _Float16 half8v __attribute__((ext_vector_type(8)));
typedef _Float16 half4v __attribute__((ext_vector_type(4)));
typedef __fp16 fp16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef float float4v __attribute__((ext_vector_type(4)));

#ifndef CA_E_AUX
#define CA_E_AUX 2      // cache policy of the encoder-token stream: non-temporal (read once per launch; -3 % on the kernel, tools/ubench/hbm_read.hip: 6.2 -> 7.0 TB/s on the bare stream)
#endif
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, CA_E_AUX);
}
__device__ __forceinline__ half4v lds_tr16(const char* p) {
  fp16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4*)p);
  return __builtin_bit_cast(half4v, v);
}
__device__ __forceinline__ int swz(int row) { return ((row & 7) << 1) | ((row >> 3) & 1); }

template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// wait until at most `tiles` x PER vector-memory operations are outstanding (tiles <= MAXT, wave-uniform)
template <int PER, int MAXT>
__device__ __forceinline__ void wait_vm_dyn(int tiles) {
  if constexpr (MAXT <= 0 || PER <= 0) {
    wait_vm<0>();
  } else {
    if (tiles >= MAXT) wait_vm<PER * MAXT>();
    else wait_vm_dyn<PER, MAXT - 1>(tiles);
  }
}

struct CrossArgs {
  const _Float16* E;     // [crops][kv_rows][ED] encoder tokens (rows >= n_keys of a crop are finite padding)
  const _Float16* qt;    // [crops*W][16][ED] absorbed queries (heads >= `heads` are not read)
  _Float16* ct;          // [crops*W][16][ED] contexts
  size_t crop_stride, tile_stride;   // bytes from crop to crop / from a crop's key tile to its next one
  int n_keys, heads;
};

#ifndef CA_TK
#define CA_TK 32
#endif
constexpr int TK = CA_TK;   // keys per tile: 32 (two 16-key row tiles, K = 32 in the second product) or 16 (one, K = 16)
static_assert(TK == 16 || TK == 32, "cross_attn key tile");
constexpr int RT = TK / 16;
// ring slots that fit beside the score-exchange buffer (2 W waves x 2 KiB) in 160 KiB of LDS
constexpr int cross_slots(int ed, int w) {
  const int n = (160 * 1024 - 4 * w * 1024) / (TK * ed * 2);
  return n > 6 ? 6 : n;
}

// One key tile of one wave.  `sb` (the tile being read), `feed_dst` (the ring slot the DMA of a later tile fills) and `xb` (score
// exchange) are distinct LDS regions; they are passed as __restrict__ pointers because that is what tells the compiler so: without
// it every ds_read_b64_tr_b16 that follows a global_load_lds gets an s_waitcnt vmcnt(0) in front (the transposed-read intrinsic
// carries no alias information), i.e. the whole ring drains before the second product — measured 306 us per launch against
// 244 us for the DMA stream alone.
// FEED (compile time: the tile loop is split into the tiles that refill the ring and the last DEPTH that do not): with the
// refill behind a run-time test every k-step was a basic block of its own — two LDS reads, a wait for exactly those two, two
// MFMAs — i.e. a full LDS latency in front of every MFMA pair, twelve times per product (round 3: ~6000 cycles per tile measured
// against ~2000 of stream at the rate a bare LDS-DMA ring reaches on this chip, tools/ubench/hbm_read.hip).  Straight-line, the
// fragment reads of a product run AHEAD k-steps in front of the MFMAs that consume them.
template <int ED, int W, int NPW, int NI, bool FEED>
__device__ __forceinline__ void cross_tile(const char* __restrict__ sb, char* __restrict__ feed_dst, float4v* __restrict__ xb,
                                           const char* __restrict__ feed_src, const int (&dma_off)[NPW],
                                           const int (&a_off)[4], const int (&t_off)[8], const half8v (&qf)[ED / 64],
                                           float4v (&acc)[ED / 32], float& mref, float& lsum, int t, int n_keys, int wave, int lane) {
  constexpr int NW = 2 * W, ROWB = ED * 2, KS = ED / 64, MT = ED / 32;
  const int g = lane >> 4;
    // ---- partial S^T[32 keys][16 heads] = E_tile[:, half] Qt[:, half]^T
#ifdef CA_DMA_ONLY      // measurement aid (tools/ubench): the stream alone, 0 = burst after the barrier, 1 = + the exchange barrier
    if (FEED) {
#pragma unroll
      for (int j = 0; j < NPW; ++j) {
        const int i = wave + j * NW;
        if (NI % NW == 0 || i < NI) glds16(feed_src + dma_off[j], feed_dst + i * 1024);
      }
    }
    if (CA_DMA_ONLY) __builtin_amdgcn_s_barrier();
    return;
#endif
    float4v s[2] = {(float4v){0.f, 0.f, 0.f, 0.f}, (float4v){0.f, 0.f, 0.f, 0.f}};
    constexpr int EVERY = (KS / NPW) > 0 ? (KS / NPW) : 1;
    constexpr int AHEAD = KS < 4 ? KS : 4;
    half8v a0[KS], a1[RT == 2 ? KS : 1];
#pragma unroll
    for (int ks = 0; ks < AHEAD; ++ks) {
      a0[ks] = *(const half8v*)(sb + a_off[ks & 3] + 256 * (ks >> 2));
      if (RT == 2) a1[ks] = *(const half8v*)(sb + a_off[ks & 3] + 256 * (ks >> 2) + 16 * ROWB);
    }
    __builtin_amdgcn_sched_barrier(0);      // the scheduler sinks such reads back to their use (fewer live registers) unless fenced
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      if (ks + AHEAD < KS) {
        a0[ks + AHEAD] = *(const half8v*)(sb + a_off[(ks + AHEAD) & 3] + 256 * ((ks + AHEAD) >> 2));
        if (RT == 2) a1[ks + AHEAD] = *(const half8v*)(sb + a_off[(ks + AHEAD) & 3] + 256 * ((ks + AHEAD) >> 2) + 16 * ROWB);
      }
      // the DMA instructions of the tile that refills the slot freed at this tile's barrier go out now, as early as they may,
      // one per k-step (not in one burst: the CU's vector-memory queue drains a 1 KiB instruction every ~33 cycles and a wave
      // that issues 8 back to back waits in it with its LDS reads and MFMAs behind it)
      if (FEED && ks % EVERY == 0 && ks / EVERY < NPW) {
        const int j = ks / EVERY, i = wave + j * NW;
        if (NI % NW == 0 || i < NI) glds16(feed_src + dma_off[j], feed_dst + i * 1024);
      }
      __builtin_amdgcn_sched_barrier(0);    // reads of k-step ks + AHEAD | MFMAs of k-step ks: the order stays
      s[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0[ks], qf[ks], s[0], 0, 0, 0);
      if (RT == 2) s[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1[RT == 2 ? ks : 0], qf[ks], s[1], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (FEED) {
#pragma unroll
      for (int j = (KS + EVERY - 1) / EVERY; j < NPW; ++j) {       // more DMA instructions than k-steps (narrow encoders)
        const int i = wave + j * NW;
        if (NI % NW == 0 || i < NI) glds16(feed_src + dma_off[j], feed_dst + i * 1024);
      }
    }
    // the other half's partial sums (same lane layout)
    xb[(wave * 2 + 0) * 64 + lane] = s[0];
    if (RT == 2) xb[(wave * 2 + 1) * 64 + lane] = s[1];
    __builtin_amdgcn_s_waitcnt(0xc07f);          // lgkmcnt(0): the LDS writes above
    __builtin_amdgcn_s_barrier();
    s[0] += xb[((wave ^ 1) * 2 + 0) * 64 + lane];
    if (RT == 2) s[1] += xb[((wave ^ 1) * 2 + 1) * 64 + lane];
    // lane (n, g) holds head n, keys t*32 + 16 mt + 4 g + e.  Keys past the end: -inf.
    const int key0 = t * TK;
    if (key0 + TK > n_keys) {
#pragma unroll
      for (int mt = 0; mt < RT; ++mt)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (key0 + 16 * mt + 4 * g + e >= n_keys) s[mt][e] = -INFINITY;
    }
    float mx = fmaxf(fmaxf(s[0][0], s[0][1]), fmaxf(s[0][2], s[0][3]));
    if (RT == 2) mx = fmaxf(mx, fmaxf(fmaxf(s[1][0], s[1][1]), fmaxf(s[1][2], s[1][3])));
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    if (t == 0) {
      mref = mx;                                    // tile 0 always holds valid keys
    } else if (__builtin_amdgcn_ballot_w64(mx > mref + 8.f) != 0) {   // stale reference: rescale only when a tile outgrows it
      const float nr = fmaxf(mref, mx), f = __builtin_amdgcn_exp2f(mref - nr);
      mref = nr;
      lsum *= f;
#pragma unroll
      for (int i = 0; i < MT; ++i) acc[i] *= f;
    }
    half8v pf = (half8v){0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int mt = 0; mt < RT; ++mt)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float pe = __builtin_amdgcn_exp2f(s[mt][e] - mref);
        lsum += pe;
        pf[mt * 4 + e] = (_Float16)pe;
      }
    // ---- Ct^T[half of ED][16 heads] += E_tile[:, half]^T P^T
    constexpr int TAHEAD = MT < 6 ? MT : 6;
    half4v tlo[MT], thi[RT == 2 ? MT : 1];
    const half4v pf4 = __builtin_shufflevector(pf, pf, 0, 1, 2, 3);
#pragma unroll
    for (int mt = 0; mt < TAHEAD; ++mt) {
      tlo[mt] = lds_tr16(sb + t_off[mt & 7] + 256 * (mt >> 3));
      if (RT == 2) thi[mt] = lds_tr16(sb + t_off[mt & 7] + 256 * (mt >> 3) + 16 * ROWB);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      if (mt + TAHEAD < MT) {
        tlo[mt + TAHEAD] = lds_tr16(sb + t_off[(mt + TAHEAD) & 7] + 256 * ((mt + TAHEAD) >> 3));
        if (RT == 2) thi[mt + TAHEAD] = lds_tr16(sb + t_off[(mt + TAHEAD) & 7] + 256 * ((mt + TAHEAD) >> 3) + 16 * ROWB);
      }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (RT == 2) {
        const half8v a = __builtin_shufflevector(tlo[mt], thi[mt], 0, 1, 2, 3, 4, 5, 6, 7);
        acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, pf, acc[mt], 0, 0, 0);
      } else {
        acc[mt] = __builtin_amdgcn_mfma_f32_16x16x16f16(tlo[mt], pf4, acc[mt], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
}

// 2 W waves: wave = (beam, half) — each wave owns one half of the ED dims for both products (so its Qt fragments and context
// accumulators are ED/64 + ED/8 registers and everything stays in architectural VGPRs: the file is compiled with
// -amdgpu-mfma-vgpr-form, a rescale of AGPR accumulators would cost two v_accvgpr moves per value).  The two partial score
// tiles of a beam are exchanged through 2 KiB of LDS per wave.
template <int ED, int W>
__global__ __launch_bounds__(128 * W) void cross_attn_kernel(CrossArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NW = 2 * W;
  constexpr int ROWB = ED * 2;                 // bytes of one key row
  constexpr int HB = ED;                       // bytes of a wave's half row
  constexpr int TILE_B = TK * ROWB;
  constexpr int NS = cross_slots(ED, W);       // ring slots
  constexpr int NI = TILE_B / 1024;            // LDS-DMA wave-instructions per tile
  constexpr int NPW = (NI + NW - 1) / NW;      // per wave (the last ones of a tile may be missing for some waves)
  constexpr int KS = ED / 64, MT = ED / 32;    // k-steps / 16-dim tiles of a half
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int beam = wave >> 1, hh = wave & 1;
  const int n = lane & 15, g = lane >> 4;
  const int crop = blockIdx.x;
  const char* Ec = (const char*)p.E + (size_t)crop * p.crop_stride;
  const int ntiles = (p.n_keys + TK - 1) / TK;
  float4v* xbuf = (float4v*)(smem + NS * TILE_B);          // [NW][2][64] partial score tiles

  // ---- DMA plan: instruction i of a tile fills LDS bytes [1024 i, 1024 i + 1024); lane l's 16 bytes are row r, swizzled chunk cp
  int dma_off[NPW];
#pragma unroll
  for (int j = 0; j < NPW; ++j) {
    const int i = wave + j * NW;
    const int byte = i * 1024 + lane * 16;
    const int r = byte / ROWB, cp = (byte - r * ROWB) >> 4;
    const int c = (cp & ~15) | ((cp & 15) ^ swz(r));
    dma_off[j] = r * ROWB + c * 16;
  }
  auto issue = [&](int t, int slot) {
    const char* src = Ec + (size_t)t * p.tile_stride;
    char* dst = smem + slot * TILE_B;
#pragma unroll
    for (int j = 0; j < NPW; ++j) {
      const int i = wave + j * NW;
      if (NI % NW == 0 || i < NI) glds16(src + dma_off[j], dst + i * 1024);
    }
  };
  // waves whose last instruction of a tile does not exist issue NPW - 1 per tile
  const bool short_wave = (NI % NW != 0) && (wave + (NPW - 1) * NW >= NI);

  // ---- absorbed queries of this beam, this half of the dims: B operand fragments, B[k = dim][n = head]
  half8v qf[KS];
  {
    const _Float16* qr = p.qt + ((size_t)(crop * W + beam) * 16 + n) * ED + hh * (ED / 2) + g * 8;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      if (n < p.heads) qf[ks] = *(const half8v*)(qr + ks * 32);
      else qf[ks] = (half8v){0, 0, 0, 0, 0, 0, 0, 0};
    }
  }
  // ---- LDS read offsets (bytes inside a tile)
  // row reads: key row 16 mt + n, chunk 4 (ks & 3) + g of 256-byte group ks >> 2 of the wave's half row.  swz(16 + n) = swz(n):
  // the second row tile is the first plus 16 rows (an immediate offset)
  int a_off[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) a_off[j] = n * ROWB + hh * HB + 16 * ((4 * j + g) ^ swz(n));
  // transposed reads: key rows 4 g + q (+ 16), chunk 2 (mt & 7) + (pp >> 1) of group mt >> 3, half pp & 1.  With x = swz(row):
  // 16 ((2 j + b) ^ x) = 32 (j ^ (x >> 1)) + 16 (b ^ (x & 1)), and x is the same for both row sets
  int t_off[8];
  {
    const int q = n >> 2, pp = n & 3, row = 4 * g + q, x = swz(row);
#pragma unroll
    for (int j = 0; j < 8; ++j) t_off[j] = row * ROWB + hh * HB + 32 * (j ^ (x >> 1)) + 16 * ((pp >> 1) ^ (x & 1)) + 8 * (pp & 1);
  }

  float4v acc[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) acc[i] = (float4v){0.f, 0.f, 0.f, 0.f};
  float mref = 0.f, lsum = 0.f;

  constexpr int DEPTH = NS - 1;                // tiles issued ahead of the one being consumed
  static_assert(NS >= 2 && DEPTH * NPW <= 60, "ring depth / vmcnt range");
#pragma unroll
  for (int i = 0; i < DEPTH; ++i)
    if (i < ntiles) issue(i, i);
  int slot = 0;
  // the tiles that refill the ring behind them: after each, DEPTH - 1 younger tiles are in flight
  const int nfeed = ntiles > DEPTH ? ntiles - DEPTH : 0;
  for (int t = 0; t < nfeed; ++t) {
    // tile t has landed once at most the DMA instructions of the tiles issued after it are outstanding
    if (short_wave) wait_vm<(NPW - 1) * (DEPTH - 1)>();
    else wait_vm<NPW * (DEPTH - 1)>();
    __builtin_amdgcn_s_barrier();
    // tile t + DEPTH goes into the slot tile t - 1 occupied (every wave is past its last read of it); cross_tile issues its DMA
    // instructions between the MFMAs of the first product
    const char* feed_src = Ec + (size_t)(t + DEPTH) * p.tile_stride;
    char* feed_dst = smem + (slot == 0 ? NS - 1 : slot - 1) * TILE_B;
    cross_tile<ED, W, NPW, NI, true>(smem + slot * TILE_B, feed_dst, xbuf, feed_src, dma_off, a_off, t_off, qf, acc, mref, lsum, t,
                                     p.n_keys, wave, lane);
    slot = (slot == NS - 1) ? 0 : slot + 1;
  }
  for (int t = nfeed; t < ntiles; ++t) {
    const int after = min(DEPTH - 1, ntiles - 1 - t);     // tiles issued after tile t
    if (short_wave) wait_vm_dyn<NPW - 1, DEPTH - 1>(after);
    else wait_vm_dyn<NPW, DEPTH - 1>(after);
    __builtin_amdgcn_s_barrier();
    cross_tile<ED, W, NPW, NI, false>(smem + slot * TILE_B, smem, xbuf, Ec, dma_off, a_off, t_off, qf, acc, mref, lsum, t, p.n_keys,
                                      wave, lane);
    slot = (slot == NS - 1) ? 0 : slot + 1;
  }
  lsum += __shfl_xor(lsum, 16);
  lsum += __shfl_xor(lsum, 32);
  const float inv = 1.f / lsum;
  if (n < p.heads) {
    _Float16* o = p.ct + ((size_t)(crop * W + beam) * 16 + n) * ED + hh * (ED / 2) + 4 * g;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      half4v h;
#pragma unroll
      for (int e = 0; e < 4; ++e) h[e] = (_Float16)(acc[mt][e] * inv);
      *(half4v*)(o + 16 * mt) = h;
    }
  }
}

// qt[r][h][d] = sum_j q[r][h*64 + j] wkt[h][d][j].  Computed transposed (rows of the MFMA tile = dims, columns = query rows) so
// that a lane ends up with 4 consecutive dims of one (row, head).  grid (rows / 64, heads), 4 waves; wave w = dims
// [w ED / 4, (w + 1) ED / 4) of all 64 rows: each W_k fragment it fetches (L2) serves 4 row tiles.  Results pass through a
// wave-private LDS tile so that the stores are 16 bytes per lane, 128 contiguous bytes per row (the output, 94 MB per launch at
// trocr-base sizes, is what bounds this kernel).  log2(e) (the soft-max runs on exp2) is folded into wkt.
template <int ED>
__global__ __launch_bounds__(256) void absorb_q_kernel(const _Float16* __restrict__ q, int ldq, const _Float16* __restrict__ wkt,
                                                      _Float16* __restrict__ qt, int rows) {
  __shared__ __attribute__((aligned(16))) _Float16 stage[4][64][72];      // [wave][row][64 dims + pad]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = lane & 15, g = lane >> 4;
  const int h = blockIdx.y, r0 = blockIdx.x * 64;
  half8v b[4][2];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int r = r0 + nt * 16 + n;
      if (r < rows) b[nt][ks] = *(const half8v*)(q + (size_t)r * ldq + h * 64 + ks * 32 + g * 8);
      else b[nt][ks] = (half8v){0, 0, 0, 0, 0, 0, 0, 0};
    }
  constexpr int MTW = ED / 64;      // 16-dim tiles per wave (a multiple of 4)
  const _Float16* wrow = wkt + ((size_t)h * ED + wave * (ED / 4) + n) * 64 + g * 8;
#pragma unroll 1
  for (int mg = 0; mg < MTW / 4; ++mg) {
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      const int mt = mg * 4 + mi;
      const half8v a0 = *(const half8v*)(wrow + (size_t)mt * 16 * 64);
      const half8v a1 = *(const half8v*)(wrow + (size_t)mt * 16 * 64 + 32);
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        float4v c = {0.f, 0.f, 0.f, 0.f};
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b[nt][0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b[nt][1], c, 0, 0, 0);
        half4v o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (_Float16)c[e];
        *(half4v*)&stage[wave][nt * 16 + n][mi * 16 + 4 * g] = o;       // row nt*16+n, dims mi*16 + 4g .. +3 of this group
      }
    }
    // 64 rows x 128 bytes -> 8 rows x 128 contiguous bytes per wave-instruction
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int row = i * 8 + (lane >> 3), ch = lane & 7, r = r0 + row;
      const uint4 v = *(const uint4*)&stage[wave][row][ch * 8];
      if (r < rows) *(uint4*)(qt + ((size_t)r * 16 + h) * ED + wave * (ED / 4) + mg * 64 + ch * 8) = v;
    }
  }
}

// ao[r][h*64 + j] = sum_d ct[r][h][d] wv[h*64 + j][d] + bv[h*64 + j].  Transposed like absorb_q: MFMA tile rows = j, columns =
// query rows.  grid (rows / 64, heads), 4 waves: all four work on the same 64 rows (4 column tiles x 4 row tiles of
// accumulators each) and split the contraction over ED four ways — four times the loads in flight of a one-wave-per-tile
// layout, which is what this kernel needs: it streams the contexts (94 MB per launch) once and is latency-bound otherwise.  The
// partial tiles are summed through LDS.
template <int ED>
__global__ __launch_bounds__(256) void absorb_v_kernel(const _Float16* __restrict__ ct, const _Float16* __restrict__ wv,
                                                      const float* __restrict__ bv, _Float16* __restrict__ ao, int ldo, int rows) {
  __shared__ __attribute__((aligned(16))) float4v part[4][16][64];        // [wave][mt * 4 + nt][lane]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = lane & 15, g = lane >> 4;
  const int h = blockIdx.y, r0 = blockIdx.x * 64;
  constexpr int KSW = ED / 32 / 4;                 // k-steps per wave
  const _Float16* brow[4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) brow[nt] = ct + ((size_t)min(r0 + nt * 16 + n, rows - 1) * 16 + h) * ED + wave * (ED / 4) + g * 8;
  const _Float16* arow = wv + ((size_t)h * 64 + n) * ED + wave * (ED / 4) + g * 8;
  float4v c[4][4];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) c[mt][nt] = (float4v){0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
  for (int ks = 0; ks < KSW; ++ks) {
    half8v a[4], b[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) b[nt] = *(const half8v*)(brow[nt] + ks * 32);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) a[mt] = *(const half8v*)(arow + (size_t)mt * 16 * ED + ks * 32);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) c[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[mt], b[nt], c[mt][nt], 0, 0, 0);
  }
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) part[wave][mt * 4 + nt][lane] = c[mt][nt];
  __syncthreads();
  // wave w finishes row tile nt = w: sum of the four partials, bias, 8-byte stores
  const int r = r0 + wave * 16 + n;
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    float4v v = part[0][mt * 4 + wave][lane];
#pragma unroll
    for (int w = 1; w < 4; ++w) v += part[w][mt * 4 + wave][lane];
    const float4v bb = *(const float4v*)(bv + h * 64 + mt * 16 + 4 * g);
    half4v o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = (_Float16)(v[e] + bb[e]);
    if (r < rows) *(half4v*)(ao + (size_t)r * ldo + h * 64 + mt * 16 + 4 * g) = o;
  }
}

template <int ED>
int launch_ed(mhip_ctx* ctx, const CrossAbsorbDesc& d) {
  const int rows = d.crops * d.beam;
  PROF_LAUNCH(ctx, MHIP_K_DEC_OPS, hipLaunchKernelGGL((absorb_q_kernel<ED>), dim3((rows + 63) / 64, d.heads), dim3(256), 0, ctx->stream,
                                                     (const _Float16*)d.q, d.ldq, (const _Float16*)d.wkt, (_Float16*)d.qt, rows));
  CrossArgs a;
  a.E = (const _Float16*)d.E; a.qt = (const _Float16*)d.qt; a.ct = (_Float16*)d.ct;
  constexpr int TILE_B = TK * ED * 2;
  a.n_keys = d.n_keys; a.heads = d.heads;
  a.crop_stride = (size_t)d.kv_rows * ED * 2; a.tile_stride = TILE_B;
  const size_t lds = (size_t)cross_slots(ED, d.beam) * TILE_B + (size_t)d.beam * 4096;
#define CROSS_LAUNCH(WV)                                                                                                       \
  do {                                                                                                                         \
    static std::once_flag attr;                                                                                                \
    std::call_once(attr, [&] {                                                                                                 \
      (void)hipFuncSetAttribute((const void*)cross_attn_kernel<ED, WV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    });                                                                                                                        \
    PROF_LAUNCH(ctx, MHIP_K_CROSS_ATTN,                                                                                        \
                hipLaunchKernelGGL((cross_attn_kernel<ED, WV>), dim3(d.crops), dim3(128 * WV), lds, ctx->stream, a));          \
  } while (0)
  switch (d.beam) {
    case 1: CROSS_LAUNCH(1); break;
    case 2: CROSS_LAUNCH(2); break;
    case 3: CROSS_LAUNCH(3); break;
    default: CROSS_LAUNCH(4); break;
  }
#undef CROSS_LAUNCH
  PROF_LAUNCH(ctx, MHIP_K_DEC_OPS, hipLaunchKernelGGL((absorb_v_kernel<ED>), dim3((rows + 63) / 64, d.heads), dim3(256), 0, ctx->stream,
                                                     (const _Float16*)d.ct, (const _Float16*)d.wv, d.bv, (_Float16*)d.ao, d.ldo, rows));
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return mhip_fail(ctx, MHIP_EHIP, "cross_attn launch: %s", hipGetErrorString(e));
  return 0;
}

}  // namespace

bool mhip_cross_absorb_supported(int enc_dim, int beam, int heads) {
  return (enc_dim == 256 || enc_dim == 512 || enc_dim == 768 || enc_dim == 1024) && beam >= 1 && beam <= 4 && heads >= 1 && heads <= 16;
}

int mhip_launch_cross_absorbed(mhip_ctx* ctx, const CrossAbsorbDesc& d) {
  if (!d.q || !d.E || !d.wkt || !d.wv || !d.bv || !d.qt || !d.ct || !d.ao || d.crops < 1)
    return mhip_fail(ctx, MHIP_EINVAL, "cross_attn: null operand");
  if (!mhip_cross_absorb_supported(d.enc_dim, d.beam, d.heads) || d.n_keys < 1 || d.kv_rows < d.n_keys || d.ldq % 8 || d.ldo % 4)
    return mhip_fail(ctx, MHIP_EINVAL, "cross_attn: unsupported shape (enc_dim %d, beam %d, heads %d)", d.enc_dim, d.beam, d.heads);
  switch (d.enc_dim) {
    case 256: return launch_ed<256>(ctx, d);
    case 512: return launch_ed<512>(ctx, d);
    case 768: return launch_ed<768>(ctx, d);
    default: return launch_ed<1024>(ctx, d);
  }
}
