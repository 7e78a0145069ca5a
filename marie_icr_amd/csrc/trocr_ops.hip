// trocr_ops.hip — the per-step pieces of the TrOCR text decoder that are not GEMMs.
//
// The decoder is fairseq's TransformerDecoder (third-party; built at marie/models/unilm/trocr/trocr_models.py:137-147,
// 180-186 from RoBERTa arguments: post-LayerNorm layers, learned positions, layernorm_embedding, tied output projection)
// driven one token at a time by TextRecognitionGenerator._generate (marie/models/unilm/trocr/generator.py:11-374).
//   embed_step        embed_tokens[token] * embed_scale + embed_positions[pad + 1 + step] -> layernorm_embedding
//   layernorm2        post-LN after a residual GEMM: fp32 stream and the GEMM operand copy in one pass
//   decode_attention  one query per hypothesis against (a) its own key/value history, addressed through an ancestry
//                     table so that beam re-ordering never copies the cache, (b) the crop's 577 encoder keys/values,
//                     shared by the beams of the crop.  HBM-bound: every K/V element is read once per step.
//   beam_candidates   log_softmax over the vocabulary + cumulative score + the generator's pad / eos masks, then the
//                     2 x beam best (hypothesis, token) pairs per crop (BeamSearch.step)
#include <math.h>

#include <algorithm>

#include "common.h"

namespace {

typedef float float4v __attribute__((ext_vector_type(4)));
typedef unsigned long long u64;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// one wave per row: y = LN(x) (two-pass), written as fp32 (may alias x) and as T
template <typename T>
__device__ __forceinline__ void ln_row(const float* xr, const float* g, const float* b, float* yo, T* to, int D, float eps,
                                       int lane, float4v v[4]) {
  const int nv = D >> 8;
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
    if (i < nv) s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
  const float mean = wave_sum(s) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
    if (i < nv)
#pragma unroll
      for (int k = 0; k < 4; ++k) { const float d = v[i][k] - mean; q += d * d; }
  const float rstd = 1.f / sqrtf(wave_sum(q) / (float)D + eps);
#pragma unroll
  for (int i = 0; i < 4; ++i)
    if (i < nv) {
      const int c = (i * 64 + lane) * 4;
      const float4v gg = *(const float4v*)(g + c), bb = *(const float4v*)(b + c);
      float4v y;
      T o4[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) { y[k] = (v[i][k] - mean) * rstd * gg[k] + bb[k]; o4[k] = (T)y[k]; }
      if (yo) *(float4v*)(yo + c) = y;
      if (sizeof(T) == 2) *(uint64_t*)(to + c) = *(uint64_t*)o4;
      else *(float4v*)(to + c) = *(float4v*)o4;
    }
  (void)xr;
}

template <typename T>
__global__ __launch_bounds__(256) void layernorm2_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                         const float* __restrict__ b, float* yo, T* __restrict__ to,
                                                         int rows, int D, float eps) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  float4v v[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
    if (i < (D >> 8)) v[i] = *(const float4v*)(x + (size_t)row * D + (i * 64 + lane) * 4);
  ln_row<T>(nullptr, g, b, yo ? yo + (size_t)row * D : nullptr, to + (size_t)row * D, D, eps, lane, v);
}

template <typename T>
__global__ __launch_bounds__(256) void embed_step_kernel(const int* __restrict__ tokens, const T* __restrict__ emb,
                                                         const float* __restrict__ pos_row, float scale,
                                                         const float* __restrict__ g, const float* __restrict__ b, float* xo,
                                                         T* __restrict__ to, int rows, int D, float eps) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const T* e = emb + (size_t)tokens[row] * D;
  float4v v[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
    if (i < (D >> 8)) {
      const int c = (i * 64 + lane) * 4;
#pragma unroll
      for (int k = 0; k < 4; ++k) v[i][k] = (float)e[c + k] * scale + pos_row[c + k];
    }
  ln_row<T>(nullptr, g, b, xo + (size_t)row * D, to + (size_t)row * D, D, eps, lane, v);
}

// ---- decode attention ---------------------------------------------------------------------------------------------------
// grid (heads, groups); a group = NQ consecutive query rows that share one key/value set.
//   self-attention : NQ = 1, group = hypothesis row; key s lives at kbase + ((size_t)s * slots + anc[row*anc_ld + s]) * ld
//   cross-attention: NQ = beam, group = crop;        key s lives at kbase + ((size_t)group * kv_rows + s) * ld
struct DecAttnArgs {
  const void* q;     // [rows][ldq] T, pre-scaled by head_dim^-0.5
  const void* k;
  const void* v;
  void* out;         // [rows][ldo] T
  const int* anc;    // self: [rows][anc_ld] slot of each past step; nullptr for cross
  int anc_ld, slots;
  int kv_rows;       // cross: rows per group in k / v
  int ldq, ldk, ldo;
  int n_keys, nq;
};

constexpr int DA_WAVES = 4;   // (MAXQ, MAXK) instantiations: cross-attention (4, 640), self-attention (1, 256)

// One WAVE per (group, head): a 16-byte chunk of a key row per lane, so a wave-instruction reads whole 128-byte (f16) /
// 256-byte (fp32) head rows of 8 / 4 consecutive keys — every K and V byte is fetched once, fully coalesced.
// ANC (compile time): keys are addressed through the ancestry table (self-attention) or directly (cross-attention).  As a
// run-time test inside key_row() the table load sat behind a uniform branch, and the compiler then waits for ALL
// outstanding loads (vmcnt(0)) at every join — the four K / V loads of an iteration went out one at a time.
template <typename T, int DA_MAXQ, int DA_MAXK, bool ANC>
__global__ __launch_bounds__(64 * DA_WAVES) void decode_attn_kernel(DecAttnArgs p, int heads, int tasks) {
  constexpr int EPC = 16 / (int)sizeof(T);        // elements per chunk: 8 / 4
  constexpr int CPR = 64 / EPC;                   // chunks per head row: 8 / 16
  constexpr int KPI = 64 / CPR;                   // keys per wave-instruction: 8 / 4
  __shared__ float sp[DA_WAVES][DA_MAXQ][DA_MAXK];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int task = blockIdx.x * DA_WAVES + wave;
  const bool live = task < tasks;
  const int h = live ? task % heads : 0, grp = live ? task / heads : 0;
  const int row0 = grp * p.nq;
  const int c = lane % CPR, ks = lane / CPR;
  // self-attention: the hypothesis' ancestry row (slot of every past step) goes to LDS once, so that key addresses depend on
  // an LDS read instead of a global load in front of every K / V load
  __shared__ int sanc[ANC ? DA_WAVES : 1][ANC ? DA_MAXK : 1];
  if constexpr (ANC) {
    if (live)
      for (int s = lane; s < p.n_keys; s += 64) sanc[wave][s] = p.anc[(size_t)row0 * p.anc_ld + s];
    __syncthreads();
  }
  auto key_row = [&](int s) -> size_t {
    if constexpr (ANC) return (size_t)s * p.slots + sanc[wave][s];
    else return (size_t)grp * p.kv_rows + s;
  };
  float q[DA_MAXQ][EPC];
#pragma unroll
  for (int qi = 0; qi < DA_MAXQ; ++qi)
#pragma unroll
    for (int j = 0; j < EPC; ++j)
      q[qi][j] = (live && qi < p.nq) ? (float)((const T*)p.q)[(size_t)(row0 + qi) * p.ldq + h * 64 + c * EPC + j] : 0.f;
  // scores.  f16: Q K^T on the matrix core — the group's queries are rows 0..nq-1 of a 16-row A tile (other rows zero),
  // 16 keys are the columns of the B tile, so a wave-instruction streams 16 keys x 64 bytes and the scores of those keys
  // land in lanes 0..15 (accumulator rows 0..3): no cross-lane reduction, no FMAs.  fp32: lanes of one key hold partial
  // dot products (CPR lanes per key), reduced with shuffles.
  if (live && sizeof(T) == 2) {
    typedef _Float16 half8v __attribute__((ext_vector_type(8)));
    const int n = lane & 15, g = lane >> 4;
    half8v qa[2];
#pragma unroll
    for (int kstep = 0; kstep < 2; ++kstep)
#pragma unroll
      for (int j = 0; j < 8; ++j)
        qa[kstep][j] = n < p.nq ? ((const _Float16*)p.q)[(size_t)(row0 + n) * p.ldq + h * 64 + kstep * 32 + g * 8 + j] : (_Float16)0.f;
    for (int s0 = 0; s0 < p.n_keys; s0 += 32) {           // two 16-key tiles in flight
      half8v kb[2][2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int s = min(s0 + t * 16 + n, p.n_keys - 1);   // clamp: columns past the end are never stored
        const _Float16* kr = (const _Float16*)p.k + key_row(s) * p.ldk + h * 64 + g * 8;
        kb[t][0] = *(const half8v*)kr;
        kb[t][1] = *(const half8v*)(kr + 32);
      }
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        float4v c4 = {0.f, 0.f, 0.f, 0.f};
        c4 = __builtin_amdgcn_mfma_f32_16x16x32_f16(qa[0], kb[t][0], c4, 0, 0, 0);
        c4 = __builtin_amdgcn_mfma_f32_16x16x32_f16(qa[1], kb[t][1], c4, 0, 0, 0);
        const int s = s0 + t * 16 + n;
        if (g == 0 && s < p.n_keys)
#pragma unroll
          for (int qi = 0; qi < DA_MAXQ; ++qi)
            if (qi < p.nq) sp[wave][qi][s] = c4[qi];
      }
    }
  } else if (live)
    for (int s0 = ks; s0 < p.n_keys; s0 += 4 * KPI) {
      T kv[4][EPC];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int s = s0 + u * KPI;
        if (s < p.n_keys) *(uint4*)kv[u] = *(const uint4*)((const T*)p.k + key_row(s) * p.ldk + h * 64 + c * EPC);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int s = s0 + u * KPI;
        if (s >= p.n_keys) break;
        float part[DA_MAXQ];
#pragma unroll
        for (int qi = 0; qi < DA_MAXQ; ++qi) {
          part[qi] = 0.f;
#pragma unroll
          for (int j = 0; j < EPC; ++j) part[qi] += q[qi][j] * (float)kv[u][j];
#pragma unroll
          for (int o = 1; o < CPR; o <<= 1) part[qi] += __shfl_xor(part[qi], o);
        }
        if (c == 0)
          for (int qi = 0; qi < p.nq; ++qi) sp[wave][qi][s] = part[qi];
      }
    }
  __syncthreads();
  float inv[DA_MAXQ];
#pragma unroll
  for (int qi = 0; qi < DA_MAXQ; ++qi) inv[qi] = 0.f;
  if (live)
    for (int qi = 0; qi < p.nq; ++qi) {
      float m = -INFINITY;
      for (int s = lane; s < p.n_keys; s += 64) m = fmaxf(m, sp[wave][qi][s]);
#pragma unroll
      for (int o = 32; o; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
      float sum = 0.f;
      for (int s = lane; s < p.n_keys; s += 64) { const float e = expf(sp[wave][qi][s] - m); sp[wave][qi][s] = e; sum += e; }
      inv[qi] = 1.f / wave_sum(sum);
    }
  __syncthreads();
  if (!live) return;
  // weighted values: lane (c, ks) accumulates its chunk of d over keys ks, ks + KPI, ...
  float acc[DA_MAXQ][EPC];
#pragma unroll
  for (int qi = 0; qi < DA_MAXQ; ++qi)
#pragma unroll
    for (int j = 0; j < EPC; ++j) acc[qi][j] = 0.f;
  for (int s0 = ks; s0 < p.n_keys; s0 += 4 * KPI) {
    T vv[4][EPC];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int s = s0 + u * KPI;
      if (s < p.n_keys) *(uint4*)vv[u] = *(const uint4*)((const T*)p.v + key_row(s) * p.ldk + h * 64 + c * EPC);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int s = s0 + u * KPI;
      if (s >= p.n_keys) break;
#pragma unroll
      for (int qi = 0; qi < DA_MAXQ; ++qi) {
        const float w = qi < p.nq ? sp[wave][qi][s] : 0.f;
#pragma unroll
        for (int j = 0; j < EPC; ++j) acc[qi][j] += w * (float)vv[u][j];
      }
    }
  }
#pragma unroll
  for (int qi = 0; qi < DA_MAXQ; ++qi)
#pragma unroll
    for (int j = 0; j < EPC; ++j)
#pragma unroll
      for (int o = CPR; o < 64; o <<= 1) acc[qi][j] += __shfl_xor(acc[qi][j], o);
  if (ks == 0)
    for (int qi = 0; qi < p.nq; ++qi) {
      T o[EPC];
#pragma unroll
      for (int j = 0; j < EPC; ++j) o[j] = (T)(acc[qi][j] * inv[qi]);
      *(uint4*)((T*)p.out + (size_t)(row0 + qi) * p.ldo + h * 64 + c * EPC) = *(uint4*)o;
    }
}

// ---- self-attention of a decoder step, f16 (the history is short: <= max_len + 1 = 16 keys in the bench) ------------------------
// One wave per HYPOTHESIS, eight heads at a time: lane = (head of the group, 16-byte chunk of its 64 dims), so a wave-instruction
// reads 1 KiB of one key row (half of the 2 KiB a 1024-wide row holds), a score is a dot product over the 8 lanes of a head (three
// DPP adds), and the soft-max and the weighted values of a head never leave its 8 lanes: no LDS, no barrier, no cross-group
// reduction.  Keys go four at a time (online soft-max per chunk), their K and V rows requested together.  The generic kernel above
// spends a wave, an LDS score buffer, two barriers and an MFMA with one live row of sixteen on every (hypothesis, head): 105 us per
// launch of 7680 x 16 tasks against ~2 KB of history each.
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef _Float16 half8sa __attribute__((ext_vector_type(8)));
template <int CTRL>
__device__ __forceinline__ float sa_dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float sa_sum8(float v) {      // quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror
  v += sa_dpp<0xB1>(v);
  v += sa_dpp<0x4E>(v);
  v += sa_dpp<0x141>(v);
  return v;
}
__global__ __launch_bounds__(256) void decode_self_attn_f16_kernel(DecAttnArgs p, int heads, int rows) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + wave;
  if (r >= rows) return;
  const int hh = lane >> 3, c = lane & 7;
  const int my_slot = lane < p.n_keys ? p.anc[(size_t)r * p.anc_ld + lane] : 0;      // n_keys <= 64: one past step per lane
  const _Float16* K = (const _Float16*)p.k;
  const _Float16* V = (const _Float16*)p.v;
  for (int h0 = 0; h0 < heads; h0 += 8) {
    const int col = (h0 + hh) * 64 + c * 8;
    const half8sa q8 = *(const half8sa*)((const _Float16*)p.q + (size_t)r * p.ldq + col);
    half2v q2[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) q2[j] = (half2v){q8[2 * j], q8[2 * j + 1]};
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float m = -INFINITY, l = 0.f;
    for (int s0 = 0; s0 < p.n_keys; s0 += 4) {
      half8sa k8[4], v8[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int s = min(s0 + u, p.n_keys - 1);                                      // clamp: the extra keys of the last chunk get weight 0
        const size_t row = ((size_t)s * p.slots + __builtin_amdgcn_readlane(my_slot, s)) * p.ldk + col;
        k8[u] = *(const half8sa*)(K + row);
        v8[u] = *(const half8sa*)(V + row);
      }
      float sc[4], cm = m;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        float d = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) d = __builtin_amdgcn_fdot2(q2[j], (half2v){k8[u][2 * j], k8[u][2 * j + 1]}, d, false);
        sc[u] = s0 + u < p.n_keys ? sa_sum8(d) : -INFINITY;
        cm = fmaxf(cm, sc[u]);
      }
      const float alpha = __expf(m - cm);          // m = -inf on the first chunk: exp(-inf) = 0, acc and l are 0 anyway
      m = cm;
      l *= alpha;
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] *= alpha;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float pe = __expf(sc[u] - m);
        l += pe;
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = __builtin_fmaf(pe, (float)v8[u][j], acc[j]);
      }
    }
    const float inv = 1.f / l;
    half8sa o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (_Float16)(acc[j] * inv);
    *(half8sa*)((_Float16*)p.out + (size_t)r * p.ldo + col) = o;
  }
}

// ---- beam candidates ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned okey(float f) {
  unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float okey_inv(unsigned k) {
  return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

struct CandArgs {
  const void* logits;     // [bsz*beam][ld], fp32 or f16 (LT)
  int ld, vocab, beam;
  const float* cum;       // [bsz*beam] cumulative score of each hypothesis (scores[:, step-1]); unused at step 0
  int step, max_len, min_len;
  int pad, eos;
  float* cand_scores;     // [bsz][2*beam]
  int* cand_tokens;       // [bsz][2*beam]
  int* cand_beams;        // [bsz][2*beam]
};

constexpr int CAND_T = 1024, CAND_K = 8;   // per-thread shortlist length (>= 2 * beam)

// one workgroup per crop; LT = element type of the logits (the GEMM's output type)
template <typename LT>
__global__ __launch_bounds__(CAND_T) void beam_candidates_kernel(CandArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  u64* keys = (u64*)smem;                 // CAND_T * CAND_K
  __shared__ float rlse[8];
  const int sample = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int nb = p.step == 0 ? 1 : p.beam;          // step 0: all beams are identical, only the first one competes
  const int K = 2 * p.beam;
  // Two passes over the nb rows of the crop, 16 bytes per lane (the rows are 100 KB each at vocab 50 265: a 2-byte load per lane
  // and three passes made this kernel 2.2 % of the step at 2.1 TB/s).  Pass 1: log-sum-exp per row with a running maximum per
  // thread (one rescale per 16-byte chunk), combined across the workgroup.  Pass 2: scores and the per-thread shortlists.
  // (One pass — log-sum-exp and a per-row shortlist of logits together — measured 1.14 ms against 0.75 ms on the same box: a thread
  // sees only ~49 elements of a row, so a per-row shortlist admits a quarter of them; the kernel is bound by this bookkeeping, not by
  // the second read of rows that are still in the L2 / Infinity Cache.)
  constexpr int EPC = 16 / (int)sizeof(LT);
  constexpr float L2E = 1.4426950408889634f;
  const int nchunks = (p.vocab + EPC - 1) / EPC;
  __shared__ float wm[8][16], ws[8][16];
  for (int b = 0; b < nb; ++b) {
    const LT* row = (const LT*)p.logits + (size_t)(sample * p.beam + b) * p.ld;
    float m = -INFINITY, sum = 0.f;
    for (int c = tid; c < nchunks; c += CAND_T) {
      LT v[EPC];
      *(uint4*)v = *(const uint4*)(row + (size_t)c * EPC);
      float f[EPC], cm = -INFINITY;
#pragma unroll
      for (int e = 0; e < EPC; ++e) {
        f[e] = (c * EPC + e < p.vocab) ? (float)v[e] : -INFINITY;      // columns past the vocabulary are row padding
        cm = fmaxf(cm, f[e]);
      }
      if (cm == -INFINITY) continue;       // a chunk of -inf logits adds nothing; (-inf) - (-inf) below would be NaN
      if (cm > m) { sum *= __builtin_amdgcn_exp2f((m - cm) * L2E); m = cm; }     // m = -inf: exp2(-inf) = 0, sum stays 0
#pragma unroll
      for (int e = 0; e < EPC; ++e) sum += __builtin_amdgcn_exp2f((f[e] - m) * L2E);
    }
    // combine (m, sum) pairs: wave, then workgroup
#pragma unroll
    for (int o = 32; o; o >>= 1) {
      const float om = __shfl_xor(m, o), os = __shfl_xor(sum, o), nm = fmaxf(m, om);
      sum = (m == -INFINITY ? 0.f : sum * __builtin_amdgcn_exp2f((m - nm) * L2E)) + (om == -INFINITY ? 0.f : os * __builtin_amdgcn_exp2f((om - nm) * L2E));
      m = nm;
    }
    if (lane == 0) { wm[b][wv] = m; ws[b][wv] = sum; }
  }
  __syncthreads();
  if (tid < nb) {
    float m = -INFINITY;
    for (int w = 0; w < 16; ++w) m = fmaxf(m, wm[tid][w]);
    float ss = 0.f;
    for (int w = 0; w < 16; ++w) ss += wm[tid][w] == -INFINITY ? 0.f : ws[tid][w] * __builtin_amdgcn_exp2f((wm[tid][w] - m) * L2E);
    rlse[tid] = m + logf(ss);
  }
  __syncthreads();
  // per-thread shortlist of the K best (score, flat index) pairs; flat index = b * vocab + token
  u64 best[CAND_K];
#pragma unroll
  for (int j = 0; j < CAND_K; ++j) best[j] = 0;
  for (int b = 0; b < nb; ++b) {
    const LT* row = (const LT*)p.logits + (size_t)(sample * p.beam + b) * p.ld;
    const float base = (p.step == 0 ? 0.f : p.cum[sample * p.beam + b]), lse = rlse[b];
    for (int c = tid; c < nchunks; c += CAND_T) {
      LT v[EPC];
      *(uint4*)v = *(const uint4*)(row + (size_t)c * EPC);
#pragma unroll
      for (int e = 0; e < EPC; ++e) {
        const int i = c * EPC + e;
        if (i >= p.vocab) continue;
        float lp = (float)v[e] - lse;
        if (lp != lp) lp = -INFINITY;
        if (i == p.pad) lp = -INFINITY;
        if (p.step >= p.max_len && i != p.eos) lp = -INFINITY;
        if (p.step < p.min_len && i == p.eos) lp = -INFINITY;
        const float sc = lp + base;
        const u64 key = ((u64)okey(sc) << 32) | (u64)(0xffffffffu - (unsigned)(b * p.vocab + i));
        if (key > best[K - 1]) {
          best[K - 1] = key;
#pragma unroll
          for (int j = CAND_K - 1; j > 0; --j)
            if (j < K && best[j] > best[j - 1]) { const u64 t = best[j]; best[j] = best[j - 1]; best[j - 1] = t; }
        }
      }
    }
  }
  // K rounds of "largest key in the wave" (each lane's shortlist is sorted, so its head is its best remaining key),
  // then wave 0 repeats that over the 16 wave winners' lists — no sort of the 8192 shortlisted keys.
  {
    int head = 0;
    for (int r = 0; r < K; ++r) {
      const u64 mine = head < K ? best[head] : 0ull;
      u64 mx = mine;
#pragma unroll
      for (int o = 32; o; o >>= 1) { const u64 other = __shfl_xor(mx, o); mx = other > mx ? other : mx; }
      if (mine == mx && mx != 0ull) ++head;            // keys are unique (they embed the flat index)
      if (lane == 0) keys[wv * CAND_K + r] = mx;
    }
  }
  __syncthreads();
  if (wv == 0) {
    u64 mine[4];                                         // 16 waves x K <= 128 keys: lane holds up to 2 (K <= 8)
    const int total = 16 * K;
    int cnt = 0;
    for (int e = lane; e < total; e += 64) mine[cnt++] = keys[(e / K) * CAND_K + (e % K)];
    for (int r = 0; r < K; ++r) {
      u64 m = 0ull;
      for (int e = 0; e < cnt; ++e) m = mine[e] > m ? mine[e] : m;
      u64 mx = m;
#pragma unroll
      for (int o = 32; o; o >>= 1) { const u64 other = __shfl_xor(mx, o); mx = other > mx ? other : mx; }
      for (int e = 0; e < cnt; ++e)
        if (mine[e] == mx) mine[e] = 0ull;
      if (lane == 0) keys[1024 + r] = mx;
    }
  }
  __syncthreads();
  if (tid < K) {
    const u64 c = keys[1024 + tid];
    const unsigned flat = 0xffffffffu - (unsigned)(c & 0xffffffffu);
    p.cand_scores[sample * K + tid] = okey_inv((unsigned)(c >> 32));
    p.cand_tokens[sample * K + tid] = (int)(flat % (unsigned)p.vocab);
    p.cand_beams[sample * K + tid] = (int)(flat / (unsigned)p.vocab);
  }
}

// anc_new[r][0..step] = anc_old[parent[r]][0..step]; anc_new[r][step+1] = r
__global__ void ancestry_kernel(const int* __restrict__ anc_old, int* __restrict__ anc_new, const int* __restrict__ parent,
                                int rows, int ld, int step) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= rows * (step + 2)) return;
  const int r = e / (step + 2), s = e - r * (step + 2);
  anc_new[(size_t)r * ld + s] = s == step + 1 ? r : anc_old[(size_t)parent[r] * ld + s];
}

}  // namespace

#define CHECK_LAUNCH(ctx, what)                                                                              \
  do {                                                                                                       \
    hipError_t _e = hipGetLastError();                                                                       \
    if (_e != hipSuccess) return mhip_fail((ctx), MHIP_EHIP, what " launch: %s", hipGetErrorString(_e));    \
  } while (0)

int mhip_launch_layernorm2(mhip_ctx* ctx, int precision, const float* x, const float* g, const float* b, float* y_f32,
                           void* y_t, int rows, int D, float eps) {
  if (D % 256 != 0 || D > 1024 || rows <= 0) return mhip_fail(ctx, MHIP_EINVAL, "layernorm2: D=%d rows=%d", D, rows);
  dim3 grid((rows + 3) / 4), block(256);
  if (precision == MHIP_PREC_F16) PROF_LAUNCH(ctx, MHIP_K_VIT_OPS, hipLaunchKernelGGL(layernorm2_kernel<_Float16>, grid, block, 0, ctx->stream, x, g, b, y_f32, (_Float16*)y_t, rows, D, eps));
  else PROF_LAUNCH(ctx, MHIP_K_VIT_OPS, hipLaunchKernelGGL(layernorm2_kernel<float>, grid, block, 0, ctx->stream, x, g, b, y_f32, (float*)y_t, rows, D, eps));
  CHECK_LAUNCH(ctx, "layernorm2");
  return 0;
}

int mhip_launch_embed_step(mhip_ctx* ctx, int precision, const int* tokens, const void* emb, const float* pos_row, float scale,
                           const float* g, const float* b, float* x, void* xt, int rows, int D, float eps) {
  if (D % 256 != 0 || D > 1024 || rows <= 0) return mhip_fail(ctx, MHIP_EINVAL, "embed_step: D=%d rows=%d", D, rows);
  dim3 grid((rows + 3) / 4), block(256);
  if (precision == MHIP_PREC_F16) PROF_LAUNCH(ctx, MHIP_K_DEC_OPS, hipLaunchKernelGGL(embed_step_kernel<_Float16>, grid, block, 0, ctx->stream, tokens, (const _Float16*)emb, pos_row, scale, g, b, x, (_Float16*)xt, rows, D, eps));
  else PROF_LAUNCH(ctx, MHIP_K_DEC_OPS, hipLaunchKernelGGL(embed_step_kernel<float>, grid, block, 0, ctx->stream, tokens, (const float*)emb, pos_row, scale, g, b, x, (float*)xt, rows, D, eps));
  CHECK_LAUNCH(ctx, "embed_step");
  return 0;
}

int mhip_launch_decode_attention(mhip_ctx* ctx, int precision, const DecAttnDesc& d) {
  if (d.n_keys < 1 || d.n_keys > 640 || d.nq < 1 || d.nq > 4 || d.heads < 1 || d.groups < 1)
    return mhip_fail(ctx, MHIP_EINVAL, "decode_attention: n_keys %d nq %d", d.n_keys, d.nq);
  const bool small = d.nq == 1 && d.n_keys <= 256;   // self-attention over a short history: 4 KB of LDS, full occupancy
  DecAttnArgs a;
  a.q = d.q; a.k = d.k; a.v = d.v; a.out = d.out; a.anc = d.anc; a.anc_ld = d.anc_ld; a.slots = d.slots;
  a.kv_rows = d.kv_rows; a.ldq = d.ldq; a.ldk = d.ldk; a.ldo = d.ldo; a.n_keys = d.n_keys; a.nq = d.nq;
  static const bool generic_only = getenv("MARIE_HIP_GENERIC_SELF_ATTN") != nullptr;      // A/B aid
  if (precision == MHIP_PREC_F16 && d.anc && d.nq == 1 && d.n_keys <= 64 && d.heads % 8 == 0 && !generic_only &&
      d.ldq % 8 == 0 && d.ldk % 8 == 0 && d.ldo % 8 == 0) {
    PROF_LAUNCH(ctx, MHIP_K_DEC_OPS, hipLaunchKernelGGL(decode_self_attn_f16_kernel, dim3((d.groups + 3) / 4), dim3(256), 0, ctx->stream, a, d.heads, d.groups));
    CHECK_LAUNCH(ctx, "decode_self_attention");
    return 0;
  }
  const int tasks = d.heads * d.groups;
  dim3 grid((tasks + DA_WAVES - 1) / DA_WAVES), block(64 * DA_WAVES);
#define DA_LAUNCH1(T, Q, K, A) PROF_LAUNCH(ctx, MHIP_K_DEC_OPS, hipLaunchKernelGGL((decode_attn_kernel<T, Q, K, A>), grid, block, 0, ctx->stream, a, d.heads, tasks))
#define DA_LAUNCH(T, Q, K) do { if (d.anc) DA_LAUNCH1(T, Q, K, true); else DA_LAUNCH1(T, Q, K, false); } while (0)
  if (precision == MHIP_PREC_F16) { if (small) DA_LAUNCH(_Float16, 1, 256); else DA_LAUNCH(_Float16, 4, 640); }
  else { if (small) DA_LAUNCH(float, 1, 256); else DA_LAUNCH(float, 4, 640); }
#undef DA_LAUNCH
#undef DA_LAUNCH1
  CHECK_LAUNCH(ctx, "decode_attention");
  return 0;
}

int mhip_launch_beam_candidates(mhip_ctx* ctx, const BeamCandDesc& d) {
  if (d.beam < 1 || d.beam > 4 || d.vocab < 2 * d.beam) return mhip_fail(ctx, MHIP_EINVAL, "beam_candidates: beam %d", d.beam);
  if (d.ld % 8 || d.ld < (d.vocab + 7) / 8 * 8 || ((uintptr_t)d.logits & 15))
    return mhip_fail(ctx, MHIP_EINVAL, "beam_candidates: rows must be 16-byte aligned with the pitch rounded up to 8 columns");
  CandArgs a;
  a.logits = d.logits; a.ld = d.ld; a.vocab = d.vocab; a.beam = d.beam; a.cum = d.cum; a.step = d.step;
  a.max_len = d.max_len; a.min_len = d.min_len; a.pad = d.pad; a.eos = d.eos;
  a.cand_scores = d.cand_scores; a.cand_tokens = d.cand_tokens; a.cand_beams = d.cand_beams;
  const int lds = (1024 + 64) * 8;
  if (d.logits_f16) PROF_LAUNCH(ctx, MHIP_K_DEC_OPS, hipLaunchKernelGGL(beam_candidates_kernel<_Float16>, dim3(d.bsz), dim3(CAND_T), lds, ctx->stream, a));
  else PROF_LAUNCH(ctx, MHIP_K_DEC_OPS, hipLaunchKernelGGL(beam_candidates_kernel<float>, dim3(d.bsz), dim3(CAND_T), lds, ctx->stream, a));
  CHECK_LAUNCH(ctx, "beam_candidates");
  return 0;
}

int mhip_launch_ancestry(mhip_ctx* ctx, const int* anc_old, int* anc_new, const int* parent, int rows, int ld, int step) {
  const int total = rows * (step + 2);
  PROF_LAUNCH(ctx, MHIP_K_DEC_OPS, hipLaunchKernelGGL(ancestry_kernel, dim3((total + 255) / 256), dim3(256), 0, ctx->stream, anc_old, anc_new, parent, rows, ld, step));
  CHECK_LAUNCH(ctx, "ancestry");
  return 0;
}
