// vit_ops.hip — the non-GEMM pieces of the ViT encoder stack shared by the DiT detector backbone and the TrOCR encoder:
// LayerNorm, fused softmax attention, patch extraction, token <-> feature-map moves, bicubic position-embedding resize.
//
// Replaces, in marie/boxes/dit/ditod/beit.py: Attention.forward's q@k^T -> softmax -> @v (:175-260), Block's
// nn.LayerNorm (:293,305), PatchEmbed's patch gather + bicubic pos-emb resize (:362-376), and the token -> NCHW tap
// reshape of forward_features (:730-732).  (GEMMs — qkv, proj, fc1, fc2, patch projection — are conv_igemm.hip.)
//
// Token layout: every image owns `npad` consecutive rows (npad % 8 == 0): row 0 = cls, rows 1..n_tok-1 = patches,
// the rest padding (finite values, masked as keys).  The residual stream is fp32; GEMM operands are T (f16 / fp32).
//
// attention (f16): one workgroup = 128 queries of one (image, head): 4 waves x 32 queries.  It is computed TRANSPOSED
// so that no operand ever needs a lane shuffle:  S^T = K Q^T  puts a query in a lane's column (C layout col = lane&15),
// so the online-softmax max/sum are per-lane loops plus two cross-lane steps, and the probabilities a lane holds after
// exp2 ARE the B operand of  O^T = V^T P^T  (k index = key) once the K rows of a tile are staged in the order
// key(kt, i) = 32(kt>>1) + 8(i>>2) + 4(kt&1) + (i&3).  V arrives pre-transposed ([d][token], written by its projection
// GEMM), so both K and V^T tiles go HBM -> LDS by LDS-DMA with 128-byte rows and the conflict-free XOR slot swizzle.
#include <algorithm>

#include "igemm_common.h"

using namespace igemm;

namespace {

constexpr int HD = 64;            // head dim of every model on this path (768/12, 1024/16)

// ------------------------------------------------------------------------------------------------------- LayerNorm
// one wave per row; D % 256 == 0, D <= 1024.  Two-pass (mean, then centred variance) in registers, fp32.  XT: element type of the
// residual stream that is normalised (fp32, or f16 when the model keeps an f16 stream as the reference's .half() path does).
// xlo: the low plane of a split stream (x = hi + lo, two f16 planes), or null.
template <typename T, typename XT>
__global__ __launch_bounds__(256) void layernorm_kernel(const XT* __restrict__ x, const float* __restrict__ g,
                                                        const float* __restrict__ b, T* __restrict__ out, int rows,
                                                        int D, float eps, const _Float16* __restrict__ xlo = nullptr) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const int nv = D >> 8;   // 4-element groups per lane
  float4v v[4];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
    if (i < nv) {
      if (sizeof(XT) == 4) {
        v[i] = *(const float4v*)((const float*)x + (size_t)row * D + (i * 64 + lane) * 4);
      } else {
        typedef _Float16 half4 __attribute__((ext_vector_type(4)));
        const half4 h = *(const half4*)((const _Float16*)x + (size_t)row * D + (i * 64 + lane) * 4);
        v[i] = (float4v){(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
        if (xlo) {
          const half4 l = *(const half4*)(xlo + (size_t)row * D + (i * 64 + lane) * 4);
          v[i] += (float4v){(float)l[0], (float)l[1], (float)l[2], (float)l[3]};
        }
      }
      s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    }
#pragma unroll
  for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o);
  const float mean = s / (float)D;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
    if (i < nv)
#pragma unroll
      for (int k = 0; k < 4; ++k) { float d = v[i][k] - mean; q += d * d; }
#pragma unroll
  for (int o = 32; o; o >>= 1) q += __shfl_xor(q, o);
  const float rstd = 1.f / sqrtf(q / (float)D + eps);
#pragma unroll
  for (int i = 0; i < 4; ++i)
    if (i < nv) {
      const int c = (i * 64 + lane) * 4;
      const float4v gg = *(const float4v*)(g + c), bb = *(const float4v*)(b + c);
      T o4[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) o4[k] = (T)((v[i][k] - mean) * rstd * gg[k] + bb[k]);
      if (sizeof(T) == 2) *(uint64_t*)(out + (size_t)row * D + c) = *(uint64_t*)o4;
      else *(float4v*)(out + (size_t)row * D + c) = *(float4v*)o4;
    }
}

// ------------------------------------------------------------------------------------------------------- attention
struct AttnArgs {
  const char* q;     // row pitch ldq elements; head h at column h*64 (pre-scaled by head_dim^-0.5 * log2 e)
  const char* k;     // row pitch ldk; head h at column h*64
  const char* vt;    // [heads*64][ldv]: V^T, column = image*npad_k + key
  char* out;         // [rows][ldo], head h at column h*64
  int ldq, ldk, ldv, ldo;
  int npad_q, npad_k;   // rows per image on the query / key side
  int n_keys;           // valid keys per image
  int heads, nqb;       // nqb = npad_q / 128
};

constexpr int AT_THREADS = 256, AT_QB = 128, AT_KT = 64;
constexpr int AT_TILE = AT_KT * 128;                 // bytes of a K tile (64 keys x 64 f16) == of a V^T tile (64 d x 64 keys)
constexpr int AT_STAGE = 2 * AT_TILE, AT_NSTAGE = 2;

__global__ __launch_bounds__(AT_THREADS) void attn_flash_f16_kernel(AttnArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, n = lane & 15;
  // consecutive logical ids on one XCD: the query blocks of an (image, head) share its K / V^T through that XCD's L2
  int qb, hb;
  {
    const int nblk = gridDim.x, bid = blockIdx.x;
    const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7;
    const int L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    qb = L % p.nqb;
    hb = L / p.nqb;
  }
  const int h = hb % p.heads, img = hb / p.heads;
  const size_t qrow0 = (size_t)img * p.npad_q + (size_t)qb * AT_QB + wave * 32;
  const size_t krow0 = (size_t)img * p.npad_k;

  // Q^T B-operands: lane (g, n) holds Q[query n of tile qt][d = 32 ks + 8 g + j]
  half8 qreg[2][2];
#pragma unroll
  for (int qt = 0; qt < 2; ++qt)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
      qreg[qt][ks] = *(const half8*)(p.q + ((qrow0 + qt * 16 + n) * p.ldq + h * HD + ks * 32 + g * 8) * 2);

  // staging: thread moves chunks (wave-instruction q covers 8 LDS rows): rows (q*4 + wave)*8 + (lane>>3), slot lane&7
  const char* ksrc[2];
  const char* vsrc[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int R = (q * 4 + wave) * 8 + (lane >> 3);
    const int lc = (lane & 7) ^ ((R >> 1) & 7);
    const int kt = R >> 4, i = R & 15;
    const int key = 32 * (kt >> 1) + 8 * (i >> 2) + 4 * (kt & 1) + (i & 3);
    ksrc[q] = p.k + ((krow0 + key) * p.ldk + h * HD + lc * 8) * 2;
    vsrc[q] = p.vt + (((size_t)h * HD + R) * p.ldv + krow0 + lc * 8) * 2;
  }
  // tiles are staged in order: the source pointers run along (a multiply-add per tile and pointer cost ~30 VALU instructions of
  // a loop that is VALU-bound)
  const size_t kstep = (size_t)AT_KT * p.ldk * 2;
  auto stage = [&](int slot) {
    char* la = smem + slot * AT_STAGE + wave * 1024;
#pragma unroll
    for (int q = 0; q < 2; ++q) { glds16(ksrc[q], la + q * 4096); ksrc[q] += kstep; }
#pragma unroll
    for (int q = 0; q < 2; ++q) { glds16(vsrc[q], la + AT_TILE + q * 4096); vsrc[q] += AT_KT * 2; }
  };

  float4v acc_o[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc_o[i][j] = (float4v){0.f, 0.f, 0.f, 0.f};
  // Online soft-max with a STALE reference: scores come out of the MFMA already minus the row's reference m (it is the
  // accumulator's initial value), and m only moves when a tile's maximum exceeds it by more than RESCALE_AT (2^8 in
  // probability) — then, and on the first tile, O^T and l are rescaled.  On every other tile the per-element subtraction,
  // the alpha exponentials and the 32 accumulator multiplies disappear from a loop whose VALU work (~800 cycles per tile
  // and wave) exceeds its MFMA work (512).  O / l at the end is independent of the reference.
  constexpr float RESCALE_AT = 8.f;
  float4v negm[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};   // -m of this lane's query, broadcast: the MFMA's C operand
  // The soft-max denominators come from the matrix cores as well: a 17th "value row" of ones, i.e. an A operand whose row 0 is
  // all ones (a constant fragment, no LDS), gives l[query] = sum of the SAME f16-rounded probabilities the numerator uses in
  // row 0 of acc_l (lanes g = 0, element 0).  That takes the 32 adds per tile (hipcc packs them into v_pk_add_f32, which
  // costs more than two plain adds beside MFMAs) out of the VALU stream for 4 more MFMAs per tile.
  const half8 ones = n == 0 ? (half8){1, 1, 1, 1, 1, 1, 1, 1} : (half8){0, 0, 0, 0, 0, 0, 0, 0};
  float4v acc_l[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};

  // fragment read offsets (row = n within a 16-row tile, logical slot = g (+4 for the second k step))
  int foff[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int R = t * 16 + n;
    foff[t] = R * 128 + ((g ^ ((R >> 1) & 7)) << 4);
  }

  const int ntiles = (p.n_keys + AT_KT - 1) / AT_KT;
  stage(0);
  int slot = 0, fill = 1;
  for (int t = 0; t < ntiles; ++t) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (t + 1 < ntiles) stage(fill);
    const char* sk = smem + slot * AT_STAGE;
    const char* sv = sk + AT_TILE;

    // ---- S^T = K Q^T ------------------------------------------------------------------------------------
    float4v s[4][2];
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      s[kt][0] = negm[0];
      s[kt][1] = negm[1];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const half8 a = *(const half8*)(sk + (foff[kt] ^ (ks << 6)));
        s[kt][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, qreg[0][ks], s[kt][0], 0, 0, 0);
        s[kt][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, qreg[1][ks], s[kt][1], 0, 0, 0);
      }
    }
    __builtin_amdgcn_s_setprio(0);
    if ((t + 1) * AT_KT > p.n_keys) {   // keys past the end of the image (padding rows)
      const int kbase = t * AT_KT;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = kbase + 32 * (kt >> 1) + 8 * g + 4 * (kt & 1) + r;
          if (key >= p.n_keys) { s[kt][0][r] = -INFINITY; s[kt][1][r] = -INFINITY; }
        }
    }
    // ---- online softmax (base 2; the log2 e factor lives in the query scale) -------------------------------
    half8 pb[2][2];
    float tmax[2];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      float mx = -INFINITY;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[kt][qt][r]);
      mx = fmaxf(mx, __shfl_xor(mx, 16));
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      tmax[qt] = mx;                       // the tile's maximum relative to the reference
    }
    const bool move = t == 0 || tmax[0] > RESCALE_AT || tmax[1] > RESCALE_AT;
    if (__builtin_amdgcn_ballot_w64(move) != 0) {      // wave-uniform: some query of this wave moves its reference
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) {
        // first tile: the reference becomes the tile maximum whatever it is; later: a query moves only on ITS OWN excess
        // (d = 0 leaves it bit-for-bit alone: alpha = 1, s - 0), so its arithmetic never depends on its wave's neighbours
        const float d = t == 0 ? tmax[qt] : (tmax[qt] > RESCALE_AT ? tmax[qt] : 0.f);
        const float alpha = t == 0 ? 0.f : __builtin_amdgcn_exp2f(-d);
        negm[qt] -= (float4v){d, d, d, d};
        acc_l[qt] *= alpha;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) acc_o[dt][qt] *= alpha;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) s[kt][qt] -= (float4v){d, d, d, d};
      }
    }
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) pb[qt][kt >> 1][(kt & 1) * 4 + r] = (_Float16)__builtin_amdgcn_exp2f(s[kt][qt][r]);
    // ---- O^T += V^T P^T ----------------------------------------------------------------------------------------
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const half8 a = *(const half8*)(sv + (foff[dt] ^ (c << 6)));
        acc_o[dt][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, pb[0][c], acc_o[dt][0], 0, 0, 0);
        acc_o[dt][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, pb[1][c], acc_o[dt][1], 0, 0, 0);
      }
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      acc_l[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ones, pb[0][c], acc_l[0], 0, 0, 0);
      acc_l[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ones, pb[1][c], acc_l[1], 0, 0, 0);
    }
    __builtin_amdgcn_s_setprio(0);
    slot = slot == AT_NSTAGE - 1 ? 0 : slot + 1;
    fill = fill == AT_NSTAGE - 1 ? 0 : fill + 1;
  }
  // ---- normalise and store: lane holds O^T[d = 16 dt + 4 g + r][query n] -> 4 consecutive d of one output row ------
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    const float inv = 1.f / __shfl(acc_l[qt][0], n);          // row 0 of the ones tile lives in lanes g = 0

    if (qb * AT_QB + wave * 32 + qt * 16 + n >= p.npad_q) continue;   // the last block may reach into the next image
    char* orow = p.out + ((qrow0 + qt * 16 + n) * p.ldo + h * HD) * 2;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      _Float16 o4[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) o4[r] = (_Float16)(acc_o[dt][qt][r] * inv);
      *(uint64_t*)(orow + (dt * 16 + g * 4) * 2) = *(uint64_t*)o4;
    }
  }
}

// fp32 parity mode: one wave per query, lane = key stripe; plain online softmax with fp32 FMAs (no matrix cores).
__global__ __launch_bounds__(256) void attn_simple_f32_kernel(AttnArgs p, int n_queries) {
  const int lane = threadIdx.x & 63;
  const int qi = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int h = blockIdx.y, img = blockIdx.z;
  if (qi >= n_queries) return;
  const float* Q = (const float*)p.q + ((size_t)img * p.npad_q + qi) * p.ldq + h * HD;
  const float* K = (const float*)p.k + (size_t)img * p.npad_k * p.ldk + h * HD;
  const float* VT = (const float*)p.vt + (size_t)h * HD * p.ldv + (size_t)img * p.npad_k;
  float q[HD], o[HD];
#pragma unroll
  for (int d = 0; d < HD; ++d) { q[d] = Q[d]; o[d] = 0.f; }
  float m = -INFINITY, l = 0.f;
  for (int key = lane; key < p.n_keys; key += 64) {
    const float* kr = K + (size_t)key * p.ldk;
    float s = 0.f;
#pragma unroll
    for (int d = 0; d < HD; ++d) s += q[d] * kr[d];
    const float mx = fmaxf(m, s);
    const float alpha = exp2f(m - mx), e = exp2f(s - mx);
    l = l * alpha + e;
#pragma unroll
    for (int d = 0; d < HD; ++d) o[d] = o[d] * alpha + e * VT[(size_t)d * p.ldv + key];
    m = mx;
  }
  float M = m;
#pragma unroll
  for (int off = 32; off; off >>= 1) M = fmaxf(M, __shfl_xor(M, off));
  const float f = (m == -INFINITY) ? 0.f : exp2f(m - M);
  l *= f;
#pragma unroll
  for (int off = 32; off; off >>= 1) l += __shfl_xor(l, off);
  float* out = (float*)p.out + ((size_t)img * p.npad_q + qi) * p.ldo + h * HD;
#pragma unroll
  for (int d = 0; d < HD; ++d) {
    float v = o[d] * f;
#pragma unroll
    for (int off = 32; off; off >>= 1) v += __shfl_xor(v, off);
    if (lane == (d & 63)) out[d] = v / l;
  }
}

// ------------------------------------------------------------------------------------------------------- patches
// resized page u8 [th][tw][3] (channel order as stored; `swap_rb` selects which stored channel is model channel 0)
// -> A[row0 + py*wp + px][k = (c*P + y)*P + x] = (pixel - mean) / std, or 0 outside (th, tw)   (zero canvas padding)
template <typename T>
__global__ __launch_bounds__(256) void patchify_kernel(const uint8_t* __restrict__ imgs, int th, int tw, int hp, int wp,
                                                        int P, int swap_rb, float mean, float stdv, T* __restrict__ outs,
                                                        int ld) {
  const uint8_t* img = imgs + (size_t)blockIdx.y * th * tw * 3;        // image blockIdx.y of the batch
  T* out = outs + (size_t)blockIdx.y * hp * wp * ld;
  const int kchunks = 3 * P * P / 8;                 // 8 consecutive x of one (c, y)
  const long long total = (long long)hp * wp * kchunks;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int kc = (int)(e % kchunks);
    const int patch = (int)(e / kchunks);
    const int k0 = kc * 8;
    const int c = k0 / (P * P), y = (k0 / P) % P, x0 = k0 % P;
    const int py = patch / wp, px = patch % wp;
    const int iy = py * P + y, sc = swap_rb ? 2 - c : c;
    T v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int ix = px * P + x0 + j;
      float f = 0.f;
      if (iy < th && ix < tw) f = ((float)img[((size_t)iy * tw + ix) * 3 + sc] - mean) / stdv;
      v[j] = (T)f;
    }
    T* dst = out + (size_t)patch * ld + k0;
    if (sizeof(T) == 2) *(uint4v*)dst = *(uint4v*)v;
    else { *(uint4v*)dst = *(uint4v*)v; *(uint4v*)(dst + 4) = *(uint4v*)(v + 4); }
  }
}

// x[img][0] = cls + pos[0]; rows n_tok.. npad-1 = 0   (fp32 residual stream)
template <typename XT>
__global__ void token_init_kernel(XT* x, const float* cls_row, int B, int npad, int n_tok, int D) {
  const int rows_per = 1 + (npad - n_tok);
  const long long total = (long long)B * rows_per * D;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const int d = (int)(e % D);
    const int r = (int)((e / D) % rows_per), b = (int)(e / ((long long)D * rows_per));
    const int row = r == 0 ? 0 : n_tok + r - 1;
    x[((size_t)b * npad + row) * D + d] = (XT)(r == 0 ? cls_row[d] : 0.f);
  }
}

// The split stream (x = hi + lo, two f16 planes; conv_igemm's EPI_SPLIT keeps it): cls row and pad rows of every image, and their
// row statistics per 64-column chunk — (sum, centred sum of squares), the cls row's precomputed on the host (cls_stats[chunk][2])
__global__ void token_init_split_kernel(_Float16* hi, _Float16* lo, const float* cls_row, const float* cls_stats, float* stats,
                                        int stats_ld, int B, int npad, int n_tok, int D) {
  const int rows_per = 1 + (npad - n_tok);
  const long long total = (long long)B * rows_per * D;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const int d = (int)(e % D);
    const int r = (int)((e / D) % rows_per), b = (int)(e / ((long long)D * rows_per));
    const int row = r == 0 ? 0 : n_tok + r - 1;
    const float v = r == 0 ? cls_row[d] : 0.f;
    const _Float16 h = (_Float16)v;
    const size_t o = ((size_t)b * npad + row) * D + d;
    hi[o] = h;
    lo[o] = (_Float16)(v - (float)h);
    if ((d & 63) == 0) {
      const int c = d >> 6;
      float* st = stats + ((size_t)c * stats_ld + (size_t)b * npad + row) * 2;
      st[0] = r == 0 ? cls_stats[c * 2] : 0.f;
      st[1] = r == 0 ? cls_stats[c * 2 + 1] : 0.f;
    }
  }
}

// row statistics of the split stream -> rstd and mean * rstd per row.  stats[(chunk * ld + row) * 2] = (sum, centred sum of squares)
// of the row over the 64 columns of the chunk; merged the pairwise way (Chan et al.): no cancellation of large means.
__global__ void ln_finalize_kernel(const float* __restrict__ stats, int chunks, int ld, float* __restrict__ rstd,
                                   float* __restrict__ mur, int rows, int D, float eps) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  float s[16], q[16], tot = 0.f;
#pragma unroll
  for (int c = 0; c < 16; ++c)
    if (c < chunks) {
      const float2 v = *(const float2*)(stats + ((size_t)c * ld + r) * 2);
      s[c] = v.x; q[c] = v.y; tot += v.x;
    }
  const float mean = tot / (float)D;
  float m2 = 0.f;
#pragma unroll
  for (int c = 0; c < 16; ++c)
    if (c < chunks) { const float d = s[c] * (1.f / 64.f) - mean; m2 += q[c] + 64.f * d * d; }
  const float rs = 1.f / sqrtf(m2 / (float)D + eps);
  rstd[r] = rs;
  mur[r] = mean * rs;
}

__global__ void split_f16_kernel(const float* __restrict__ in, _Float16* __restrict__ hi, _Float16* __restrict__ lo, long long n) {
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x) {
    const float v = in[e];
    const _Float16 h = (_Float16)v;
    hi[e] = h;
    lo[e] = (_Float16)(v - (float)h);
  }
}

__global__ void join_f16_kernel(const _Float16* __restrict__ hi, const _Float16* __restrict__ lo, float* __restrict__ out, long long n) {
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x) out[e] = (float)hi[e] + (float)lo[e];
}

// fp32 tokens (without cls) -> T feature map rows [B][n_tok-1][D]
template <typename T, typename XT>
__global__ void tokens_to_map_kernel(const XT* __restrict__ x, T* __restrict__ out, int B, int npad, int np, int D) {
  const long long total = (long long)B * np * (D / 4);
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const int c4 = (int)(e % (D / 4));
    const long long r = e / (D / 4);
    const int pidx = (int)(r % np), b = (int)(r / np);
    const XT* src = x + ((size_t)b * npad + 1 + pidx) * D + c4 * 4;
    T o4[4] = {(T)src[0], (T)src[1], (T)src[2], (T)src[3]};
    T* dst = out + ((size_t)b * np + pidx) * D + c4 * 4;
    if (sizeof(T) == 2) *(uint64_t*)dst = *(uint64_t*)o4;
    else *(float4v*)dst = *(float4v*)o4;
  }
}

// torch F.interpolate(mode="bicubic", align_corners=False) of a [gh][gw][D] table to [hp][wp][D]  (A = -0.75,
// source index scale*(dst+0.5)-0.5 unclamped, taps clamped to the border), fp32.
__device__ __forceinline__ void cubic_coeffs(float t, float c[4]) {
  const float A = -0.75f;
  float x = t + 1.f;
  c[0] = ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A;
  x = t;
  c[1] = ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f;
  x = 1.f - t;
  c[2] = ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f;
  x = 2.f - t;
  c[3] = ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A;
}

__global__ void posemb_bicubic_kernel(const float* __restrict__ tab, int gh, int gw, float* __restrict__ out, int hp,
                                      int wp, int D) {
  const long long total = (long long)hp * wp * D;
  const float sy = (float)gh / (float)hp, sx = (float)gw / (float)wp;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const int d = (int)(e % D);
    const int ox = (int)((e / D) % wp), oy = (int)(e / ((long long)D * wp));
    const float ry = sy * ((float)oy + 0.5f) - 0.5f, rx = sx * ((float)ox + 0.5f) - 0.5f;
    const int iy = (int)floorf(ry), ix = (int)floorf(rx);
    float cy[4], cx[4];
    cubic_coeffs(ry - (float)iy, cy);
    cubic_coeffs(rx - (float)ix, cx);
    float acc = 0.f;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int yy = min(max(iy - 1 + a, 0), gh - 1);
      float row = 0.f;
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int xx = min(max(ix - 1 + b, 0), gw - 1);
        row += tab[((size_t)yy * gw + xx) * D + d] * cx[b];
      }
      acc += row * cy[a];
    }
    out[e] = acc;
  }
}

// T rows -> fp32 rows (test / host read-back path)
template <typename T>
__global__ void convert_rows_kernel(const T* __restrict__ in, float* __restrict__ out, long long n) {
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x) out[e] = (float)in[e];
}

__global__ void narrow_f16_kernel(const float* __restrict__ in, _Float16* __restrict__ out, long long n) {
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x) out[e] = (_Float16)in[e];
}

// nested 2x2 row order (depth `nest`) -> raster NHWC, optionally adding the 2x nearest-upsampled coarser level
// (FPN top-down path: F.interpolate(scale_factor=2, mode="nearest") + lateral).  OT = T or float.
template <typename T, typename OT>
__global__ void unnest_kernel(const T* __restrict__ in, const T* __restrict__ coarse, OT* __restrict__ out, int B, int H,
                              int W, int C, int nest) {
  const long long total = (long long)B * H * W * (C / 4);
  const int h0 = H >> nest, w0 = W >> nest;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const int c4 = (int)(e % (C / 4));
    long long r = e / (C / 4);
    const int X = (int)(r % W);
    r /= W;
    const int Y = (int)(r % H), b = (int)(r / H);
    long long row = ((long long)b * h0 + (Y >> nest)) * w0 + (X >> nest);
    for (int l = nest - 1; l >= 0; --l) row = row * 4 + (((Y >> l) & 1) * 2 + ((X >> l) & 1));
    const T* src = in + row * C + c4 * 4;
    float v[4] = {(float)src[0], (float)src[1], (float)src[2], (float)src[3]};
    if (coarse) {
      const T* cs = coarse + (((long long)b * (H / 2) + (Y >> 1)) * (W / 2) + (X >> 1)) * C + c4 * 4;
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] += (float)cs[k];
    }
    OT* dst = out + (((long long)b * H + Y) * W + X) * C + c4 * 4;
#pragma unroll
    for (int k = 0; k < 4; ++k) dst[k] = (OT)v[k];
  }
}

inline int grid_for(long long total, int block) { return (int)std::min<long long>((total + block - 1) / block, 65535LL * 4); }

}  // namespace

#define CHECK_LAUNCH(ctx, what)                                                                              \
  do {                                                                                                       \
    hipError_t _e = hipGetLastError();                                                                       \
    if (_e != hipSuccess) return mhip_fail((ctx), MHIP_EHIP, what " launch: %s", hipGetErrorString(_e));    \
  } while (0)

int mhip_launch_layernorm(mhip_ctx* ctx, int precision, const void* x, const float* g, const float* b, void* out,
                          int rows, int D, float eps, int x_f16, const void* x_lo) {
  if (D % 256 != 0 || D > 1024 || rows <= 0) return mhip_fail(ctx, MHIP_EINVAL, "layernorm: D=%d rows=%d", D, rows);
  if (x_f16 && precision != MHIP_PREC_F16) return mhip_fail(ctx, MHIP_EINVAL, "layernorm: an f16 stream needs the f16 mode");
  if (x_lo && !x_f16) return mhip_fail(ctx, MHIP_EINVAL, "layernorm: a low plane belongs to an f16 high plane");
  dim3 grid((rows + 3) / 4), block(256);
  if (precision == MHIP_PREC_F16 && x_f16)
    PROF_LAUNCH(ctx, MHIP_K_VIT_OPS, hipLaunchKernelGGL((layernorm_kernel<_Float16, _Float16>), grid, block, 0, ctx->stream, (const _Float16*)x, g, b, (_Float16*)out, rows, D, eps, (const _Float16*)x_lo));
  else if (precision == MHIP_PREC_F16)
    PROF_LAUNCH(ctx, MHIP_K_VIT_OPS, hipLaunchKernelGGL((layernorm_kernel<_Float16, float>), grid, block, 0, ctx->stream, (const float*)x, g, b, (_Float16*)out, rows, D, eps));
  else
    PROF_LAUNCH(ctx, MHIP_K_VIT_OPS, hipLaunchKernelGGL((layernorm_kernel<float, float>), grid, block, 0, ctx->stream, (const float*)x, g, b, (float*)out, rows, D, eps));
  CHECK_LAUNCH(ctx, "layernorm");
  return 0;
}

int mhip_launch_attention(mhip_ctx* ctx, int precision, const AttnDesc& d) {
  if (d.images <= 0 || d.heads <= 0 || d.n_keys <= 0 || d.n_queries <= 0 || d.npad_k % 8 || d.n_queries > d.npad_q ||
      d.n_keys > d.npad_k)
    return mhip_fail(ctx, MHIP_EINVAL, "attention: bad shape (q %d/%d, k %d/%d)", d.n_queries, d.npad_q, d.n_keys, d.npad_k);
  const int esz = precision == MHIP_PREC_F16 ? 2 : 4;
  if ((d.ldq * esz) % 16 || (d.ldk * esz) % 16 || (d.ldv * esz) % 16 || (d.ldo * esz) % 8)
    return mhip_fail(ctx, MHIP_EINVAL, "attention: row pitches must keep 16-byte alignment");
  AttnArgs a;
  a.q = (const char*)d.q; a.k = (const char*)d.k; a.vt = (const char*)d.vt; a.out = (char*)d.out;
  a.ldq = d.ldq; a.ldk = d.ldk; a.ldv = d.ldv; a.ldo = d.ldo;
  a.npad_q = d.npad_q; a.npad_k = d.npad_k; a.n_keys = d.n_keys; a.heads = d.heads;
  a.nqb = (d.n_queries + AT_QB - 1) / AT_QB;
  if (ctx->profiling) ctx->prof[MHIP_K_ATTN_FLASH].flops += mhip_attention_flops(d);
  if (precision == MHIP_PREC_F16) {
    static std::once_flag attr;
    std::call_once(attr, [&] {
      (void)hipFuncSetAttribute((const void*)attn_flash_f16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, AT_NSTAGE * AT_STAGE);
    });
    dim3 grid((unsigned)(a.nqb * d.heads * d.images)), block(AT_THREADS);
    PROF_LAUNCH(ctx, MHIP_K_ATTN_FLASH,
                hipLaunchKernelGGL(attn_flash_f16_kernel, grid, block, AT_NSTAGE * AT_STAGE, ctx->stream, a));
  } else {
    dim3 grid((d.n_queries + 3) / 4, d.heads, d.images), block(256);
    PROF_LAUNCH(ctx, MHIP_K_ATTN_FLASH, hipLaunchKernelGGL(attn_simple_f32_kernel, grid, block, 0, ctx->stream, a, d.n_queries));
  }
  CHECK_LAUNCH(ctx, "attention");
  return 0;
}

double mhip_attention_flops(const AttnDesc& d) {
  return 4.0 * d.images * d.heads * (double)d.n_queries * d.n_keys * HD;
}

int mhip_launch_patchify(mhip_ctx* ctx, int precision, const uint8_t* img, int B, int th, int tw, int hp, int wp, int P,
                         int swap_rb, float mean, float stdv, void* out, int ld) {
  if (P != 16 && P != 8) return mhip_fail(ctx, MHIP_EINVAL, "patchify: patch size %d", P);
  const long long total = (long long)hp * wp * (3 * P * P / 8);
  dim3 grid((unsigned)std::min<long long>((total + 255) / 256, 4096), B), block(256);
  if (precision == MHIP_PREC_F16)
    PROF_LAUNCH(ctx, MHIP_K_VIT_OPS, hipLaunchKernelGGL(patchify_kernel<_Float16>, grid, block, 0, ctx->stream, img, th, tw, hp, wp, P, swap_rb, mean, stdv, (_Float16*)out, ld));
  else
    PROF_LAUNCH(ctx, MHIP_K_VIT_OPS, hipLaunchKernelGGL(patchify_kernel<float>, grid, block, 0, ctx->stream, img, th, tw, hp, wp, P, swap_rb, mean, stdv, (float*)out, ld));
  CHECK_LAUNCH(ctx, "patchify");
  return 0;
}

int mhip_launch_token_init(mhip_ctx* ctx, void* x, const float* cls_row, int B, int npad, int n_tok, int D, int x_f16) {
  const long long total = (long long)B * (1 + npad - n_tok) * D;
  if (x_f16)
    PROF_LAUNCH(ctx, MHIP_K_VIT_OPS, hipLaunchKernelGGL(token_init_kernel<_Float16>, dim3(grid_for(total, 256)), dim3(256), 0, ctx->stream, (_Float16*)x, cls_row, B, npad, n_tok, D));
  else
    PROF_LAUNCH(ctx, MHIP_K_VIT_OPS, hipLaunchKernelGGL(token_init_kernel<float>, dim3(grid_for(total, 256)), dim3(256), 0, ctx->stream, (float*)x, cls_row, B, npad, n_tok, D));
  CHECK_LAUNCH(ctx, "token_init");
  return 0;
}

int mhip_launch_token_init_split(mhip_ctx* ctx, void* hi, void* lo, const float* cls_row, const float* cls_stats, float* stats,
                                 int stats_ld, int B, int npad, int n_tok, int D) {
  if (D % 64 != 0 || D > 1024) return mhip_fail(ctx, MHIP_EINVAL, "token_init_split: D=%d", D);
  const long long total = (long long)B * (1 + npad - n_tok) * D;
  PROF_LAUNCH(ctx, MHIP_K_VIT_OPS, hipLaunchKernelGGL(token_init_split_kernel, dim3(grid_for(total, 256)), dim3(256), 0, ctx->stream, (_Float16*)hi, (_Float16*)lo, cls_row, cls_stats, stats, stats_ld, B, npad, n_tok, D));
  CHECK_LAUNCH(ctx, "token_init_split");
  return 0;
}

int mhip_launch_ln_finalize(mhip_ctx* ctx, const float* stats, int chunks, int ld, float* rstd, float* mur, int rows, int D, float eps) {
  if (chunks < 1 || chunks > 16 || chunks * 64 != D || rows <= 0) return mhip_fail(ctx, MHIP_EINVAL, "ln_finalize: chunks=%d D=%d rows=%d", chunks, D, rows);
  PROF_LAUNCH(ctx, MHIP_K_VIT_OPS, hipLaunchKernelGGL(ln_finalize_kernel, dim3((rows + 255) / 256), dim3(256), 0, ctx->stream, stats, chunks, ld, rstd, mur, rows, D, eps));
  CHECK_LAUNCH(ctx, "ln_finalize");
  return 0;
}

int mhip_launch_split_f16(mhip_ctx* ctx, const float* in, void* hi, void* lo, long long n) {
  PROF_LAUNCH(ctx, MHIP_K_VIT_OPS, hipLaunchKernelGGL(split_f16_kernel, dim3(grid_for(n, 256)), dim3(256), 0, ctx->stream, in, (_Float16*)hi, (_Float16*)lo, n));
  CHECK_LAUNCH(ctx, "split_f16");
  return 0;
}

int mhip_launch_join_f16(mhip_ctx* ctx, const void* hi, const void* lo, float* out, long long n) {
  PROF_LAUNCH(ctx, MHIP_K_VIT_OPS, hipLaunchKernelGGL(join_f16_kernel, dim3(grid_for(n, 256)), dim3(256), 0, ctx->stream, (const _Float16*)hi, (const _Float16*)lo, out, n));
  CHECK_LAUNCH(ctx, "join_f16");
  return 0;
}

int mhip_launch_tokens_to_map(mhip_ctx* ctx, int precision, const void* x, void* out, int B, int npad, int np, int D, int x_f16) {
  const long long total = (long long)B * np * (D / 4);
  dim3 grid(grid_for(total, 256)), block(256);
  if (precision == MHIP_PREC_F16 && x_f16)
    PROF_LAUNCH(ctx, MHIP_K_VIT_OPS, hipLaunchKernelGGL((tokens_to_map_kernel<_Float16, _Float16>), grid, block, 0, ctx->stream, (const _Float16*)x, (_Float16*)out, B, npad, np, D));
  else if (precision == MHIP_PREC_F16)
    PROF_LAUNCH(ctx, MHIP_K_VIT_OPS, hipLaunchKernelGGL((tokens_to_map_kernel<_Float16, float>), grid, block, 0, ctx->stream, (const float*)x, (_Float16*)out, B, npad, np, D));
  else
    PROF_LAUNCH(ctx, MHIP_K_VIT_OPS, hipLaunchKernelGGL((tokens_to_map_kernel<float, float>), grid, block, 0, ctx->stream, (const float*)x, (float*)out, B, npad, np, D));
  CHECK_LAUNCH(ctx, "tokens_to_map");
  return 0;
}

int mhip_launch_posemb_bicubic(mhip_ctx* ctx, const float* tab, int gh, int gw, float* out, int hp, int wp, int D) {
  const long long total = (long long)hp * wp * D;
  PROF_LAUNCH(ctx, MHIP_K_VIT_OPS, hipLaunchKernelGGL(posemb_bicubic_kernel, dim3(grid_for(total, 256)), dim3(256), 0, ctx->stream, tab, gh, gw, out, hp, wp, D));
  CHECK_LAUNCH(ctx, "posemb_bicubic");
  return 0;
}

int mhip_launch_convert_rows(mhip_ctx* ctx, int precision, const void* in, float* out, int rows, int D) {
  const long long n = (long long)rows * D;
  dim3 grid(grid_for(n, 256)), block(256);
  if (precision == MHIP_PREC_F16)
    PROF_LAUNCH(ctx, MHIP_K_VIT_OPS, hipLaunchKernelGGL(convert_rows_kernel<_Float16>, grid, block, 0, ctx->stream, (const _Float16*)in, out, n));
  else
    PROF_LAUNCH(ctx, MHIP_K_VIT_OPS, hipLaunchKernelGGL(convert_rows_kernel<float>, grid, block, 0, ctx->stream, (const float*)in, out, n));
  CHECK_LAUNCH(ctx, "convert_rows");
  return 0;
}

int mhip_launch_narrow_f16(mhip_ctx* ctx, const float* in, void* out, long long n) {
  PROF_LAUNCH(ctx, MHIP_K_VIT_OPS, hipLaunchKernelGGL(narrow_f16_kernel, dim3(grid_for(n, 256)), dim3(256), 0, ctx->stream, in, (_Float16*)out, n));
  CHECK_LAUNCH(ctx, "narrow_f16");
  return 0;
}

int mhip_launch_unnest(mhip_ctx* ctx, int precision, const void* in, const void* coarse, void* out, int out_f32, int B,
                       int H, int W, int C, int nest) {
  if (nest < 0 || nest > 2 || C % 4 || (H & ((1 << nest) - 1)) || (W & ((1 << nest) - 1)) || (coarse && ((H | W) & 1)))
    return mhip_fail(ctx, MHIP_EINVAL, "unnest: bad shape %dx%d nest %d", H, W, nest);
  const long long total = (long long)B * H * W * (C / 4);
  dim3 grid(grid_for(total, 256)), block(256);
#define UN(T, OT) PROF_LAUNCH(ctx, MHIP_K_VIT_OPS, hipLaunchKernelGGL((unnest_kernel<T, OT>), grid, block, 0, ctx->stream, (const T*)in, (const T*)coarse, (OT*)out, B, H, W, C, nest))
  if (precision == MHIP_PREC_F16) { if (out_f32) UN(_Float16, float); else UN(_Float16, _Float16); }
  else UN(float, float);
#undef UN
  CHECK_LAUNCH(ctx, "unnest");
  return 0;
}
