// det_ops.hip — the detector-head stages of the DiT Mask R-CNN text detector that are not convolutions / GEMMs.
//
// These are detectron2 (third-party, not vendored in the reference; "expected version 0.6", Dockerfiles/gpu.Dockerfile:91-95)
// stages the reference reaches through GeneralizedRCNN.inference from OptimizedDetectronPredictor.invoke_model
// (marie/detectron/detector.py:83-147), with the configuration of config/zoo/unilm/dit/text_detection/*.yaml:
//   rpn_level     RPN.predict_proposals for one FPN level: top-1000 objectness (PRE_NMS_TOPK_TEST), anchor decode
//                 (Box2BoxTransform weights 1,1,1,1, scale clamp log(1000/16)), clip, drop empty, NMS 0.7
//   rpn_merge     concatenate levels, order by score, keep POST_NMS_TOPK_TEST = 1000
//   roi_align     ROIPooler(7x7, scales 1/4..1/32, sampling_ratio 0, ROIAlignV2 = aligned) incl. level assignment
//   det_final     FastRCNNOutputLayers.inference: softmax, decode (weights 10,10,5,5), clip, score > 0.05, NMS 0.5,
//                 top-k, then detector_postprocess (rescale to the page, clip, drop empty)
//   blackout      blackout_bboxes of the refinement passes (marie/boxes/dit/ulim_dit_box_processor.py:161-198)
//
// All of it is small, branchy, latency-bound work (<= 160 k scores, <= 1000 boxes): one workgroup per (image, level)
// keeps every intermediate in LDS — radix-select of the k-th score, bitonic sort, bit-mask NMS — so a page costs a
// handful of launches and nothing round-trips through the host.
#include <math.h>

#include <algorithm>

#include "common.h"

#pragma clang fp contract(off)

namespace {

typedef unsigned long long u64;

__device__ __forceinline__ unsigned ordered_key(float f) {   // monotone float -> unsigned
  unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key_to_float(unsigned k) {
  unsigned u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
  return __uint_as_float(u);
}

// in-LDS bitonic sort, DESCENDING, n = power of two
__device__ void bitonic_desc(u64* a, int n) {
  for (int k = 2; k <= n; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      __syncthreads();
      for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int l = i ^ j;
        if (l > i) {
          const u64 x = a[i], y = a[l];
          const bool up = (i & k) == 0;          // this run is descending
          if (up ? (x < y) : (x > y)) { a[i] = y; a[l] = x; }
        }
      }
    }
  __syncthreads();
}

__device__ __forceinline__ void decode_box(const float a[4], const float d[4], float wx, float wy, float ww, float wh,
                                           float out[4]) {
  const float clampv = 4.135166556742356f;   // log(1000 / 16)
  const float widths = a[2] - a[0], heights = a[3] - a[1];
  const float ctr_x = a[0] + 0.5f * widths, ctr_y = a[1] + 0.5f * heights;
  const float dx = d[0] / wx, dy = d[1] / wy;
  const float dw = fminf(d[2] / ww, clampv), dh = fminf(d[3] / wh, clampv);
  const float pcx = dx * widths + ctr_x, pcy = dy * heights + ctr_y;
  const float pw = expf(dw) * widths, ph = expf(dh) * heights;
  out[0] = pcx - 0.5f * pw; out[1] = pcy - 0.5f * ph;
  out[2] = pcx + 0.5f * pw; out[3] = pcy + 0.5f * ph;
}

__device__ __forceinline__ bool iou_over(const float* a, const float* b, float thr) {
  const float areaa = (a[2] - a[0]) * (a[3] - a[1]), areab = (b[2] - b[0]) * (b[3] - b[1]);
  const float w = fmaxf(0.f, fminf(a[2], b[2]) - fmaxf(a[0], b[0]));
  const float h = fmaxf(0.f, fminf(a[3], b[3]) - fmaxf(a[1], b[1]));
  const float inter = w * h;
  return inter / (areaa + areab - inter) > thr;
}

// greedy NMS over n (<= 1024) score-sorted boxes in LDS: bit-mask build by all threads, sequential sweep by wave 0.
// mask: [n][16] u64 in LDS; keep_out[i] = 1 if kept.
__device__ void nms_sorted(const float* boxes, int n, float thr, u64* mask, unsigned char* keep_out) {
  const int words = (n + 63) >> 6;
  for (int e = threadIdx.x; e < n * words; e += blockDim.x) {
    const int i = e / words, wj = e - i * words;
    u64 bits = 0;
    const int j0 = wj * 64;
    for (int b = 0; b < 64; ++b) {
      const int j = j0 + b;
      if (j > i && j < n && iou_over(boxes + 4 * i, boxes + 4 * j, thr)) bits |= 1ull << b;
    }
    mask[i * 16 + wj] = bits;
  }
  __syncthreads();
  if (threadIdx.x < 64) {
    const int lane = threadIdx.x;
    u64 removed = 0;   // lane w owns word w of the suppressed set
    for (int i0 = 0; i0 < n; i0 += 8) {
      u64 rows[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) rows[t] = (lane < words && i0 + t < n) ? mask[(i0 + t) * 16 + lane] : 0ull;
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const int i = i0 + t;
        if (i < n) {
          const unsigned lo = __builtin_amdgcn_readlane((unsigned)removed, i >> 6);
          const unsigned hi = __builtin_amdgcn_readlane((unsigned)(removed >> 32), i >> 6);
          const u64 wsel = ((u64)hi << 32) | lo;
          const bool kept = !((wsel >> (i & 63)) & 1ull);
          removed |= kept ? rows[t] : 0ull;
          if (lane == 0) keep_out[i] = kept ? 1 : 0;
        }
      }
    }
  }
  __syncthreads();
}

constexpr int RPN_A = 3;            // anchors per location
constexpr int RPN_LD = 16;          // floats per pixel of the RPN head output: 3 logits, 12 deltas, 1 pad
constexpr int TOPK = 1000;
constexpr int CAP = 4096;           // candidates gathered for the exact top-k (all scores >= the k-th)
constexpr int RPN_THREADS = 1024;

struct RpnLevelArgs {
  const float* head[5];     // [B][Hl*Wl][16]
  int H[5], W[5], stride[5];
  float cell[5][RPN_A][4];  // cell anchors per level
  int img_h, img_w;         // resized image size (clip)
  float nms_thr;
  // outputs per (image, level): up to TOPK kept proposals, score-sorted
  float* boxes;             // [B][5][TOPK][4]
  float* scores;            // [B][5][TOPK]
  int* counts;              // [B][5]
};

// one workgroup per (level, image)
__global__ __launch_bounds__(RPN_THREADS) void rpn_level_kernel(RpnLevelArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lvl = blockIdx.x, img = blockIdx.y, tid = threadIdx.x;
  const int HW = p.H[lvl] * p.W[lvl], n = HW * RPN_A;
  const float* head = p.head[lvl] + (size_t)img * HW * RPN_LD;
  // LDS map (157 KB): [0, 16 K) hist[4096] u32, later scores[1024] f32 + keep[1024] u8
  //                    [16 K, 48 K) cand[CAP] u64, later boxes[1024][4] f32 in its lower half
  //                    [32 K, 157 K) NMS mask[1000][16] u64 (starts in cand's upper half, dead by then)
  unsigned* hist = (unsigned*)smem;
  u64* cand = (u64*)(smem + 16384);
  float* sbox = (float*)(smem + 16384);
  unsigned char* keep = (unsigned char*)(smem + 4096);
  u64* mask = (u64*)(smem + 32768);
  __shared__ unsigned s_prefix, s_above, s_ncand, chunk[64];

  const int k = n < TOPK ? n : TOPK;
  unsigned thresh = 0;   // k-th largest key
  if (n > CAP) {
    // radix select (12 + 12 + 8 bits) of the k-th largest ordered key
    unsigned prefix = 0, above = 0;
    const int shifts[3] = {20, 8, 0}, bits[3] = {12, 12, 8};
    for (int pass = 0; pass < 3; ++pass) {
      const int nb = 1 << bits[pass];
      for (int i = tid; i < 4096; i += RPN_THREADS) hist[i] = 0;
      __syncthreads();
      const int hs = shifts[pass] + bits[pass];   // bits above this digit
      for (int i = tid; i < n; i += RPN_THREADS) {
        const unsigned key = ordered_key(head[(size_t)(i / RPN_A) * RPN_LD + (i % RPN_A)]);
        if (pass == 0 || (key >> hs) == (prefix >> hs)) atomicAdd(&hist[(key >> shifts[pass]) & (nb - 1)], 1u);
      }
      __syncthreads();
      const int csize = nb >> 6;                     // 64 chunks of bins, summed in parallel, walked from the top
      if (tid < 64) {
        unsigned cs = 0;
        for (int b = 0; b < csize; ++b) cs += hist[tid * csize + b];
        chunk[tid] = cs;
      }
      __syncthreads();
      if (tid == 0) {
        unsigned acc = above;
        int c = 63;
        for (; c > 0; --c) {
          if (acc + chunk[c] >= (unsigned)k) break;
          acc += chunk[c];
        }
        int b = c * csize + csize - 1;
        for (; b > c * csize; --b) {
          if (acc + hist[b] >= (unsigned)k) break;
          acc += hist[b];
        }
        s_prefix = prefix | ((unsigned)b << shifts[pass]);
        s_above = acc;
      }
      __syncthreads();
      prefix = s_prefix;
      above = s_above;
      __syncthreads();
    }
    thresh = prefix;
  }
  // gather every score >= thresh as (key << 32 | ~index): descending sort = score desc, index asc
  if (tid == 0) s_ncand = 0;
  for (int i = tid; i < CAP; i += RPN_THREADS) cand[i] = 0;
  __syncthreads();
  for (int i = tid; i < n; i += RPN_THREADS) {
    const unsigned key = ordered_key(head[(size_t)(i / RPN_A) * RPN_LD + (i % RPN_A)]);
    if (key >= thresh) {
      const unsigned at = atomicAdd(&s_ncand, 1u);
      if (at < CAP) cand[at] = ((u64)key << 32) | (u64)(0xffffffffu - (unsigned)i);
    }
  }
  __syncthreads();
  const int ncand = s_ncand < (unsigned)CAP ? (int)s_ncand : CAP;
  int sort_n = 1024;
  while (sort_n < ncand) sort_n <<= 1;
  bitonic_desc(cand, sort_n);
  // decode the top k, clip, drop empty; compact in order (stable) into sbox/score registers
  // (cand and sbox alias: read the candidate into registers first, sync, then write boxes)
  float bx[4] = {0, 0, 0, 0}, sc = 0.f;
  bool valid = false;
  if (tid < k) {
    const u64 c = cand[tid];
    const unsigned idx = 0xffffffffu - (unsigned)(c & 0xffffffffu);
    sc = key_to_float((unsigned)(c >> 32));
    const int pix = idx / RPN_A, a = idx % RPN_A;
    const int y = pix / p.W[lvl], x = pix % p.W[lvl];
    const float sx = (float)(x * p.stride[lvl]), sy = (float)(y * p.stride[lvl]);
    const float anc[4] = {sx + p.cell[lvl][a][0], sy + p.cell[lvl][a][1], sx + p.cell[lvl][a][2], sy + p.cell[lvl][a][3]};
    const float* d = head + (size_t)pix * RPN_LD + RPN_A + a * 4;
    const float dl[4] = {d[0], d[1], d[2], d[3]};
    decode_box(anc, dl, 1.f, 1.f, 1.f, 1.f, bx);
    valid = isfinite(bx[0]) && isfinite(bx[1]) && isfinite(bx[2]) && isfinite(bx[3]) && isfinite(sc);
    bx[0] = fminf(fmaxf(bx[0], 0.f), (float)p.img_w); bx[2] = fminf(fmaxf(bx[2], 0.f), (float)p.img_w);
    bx[1] = fminf(fmaxf(bx[1], 0.f), (float)p.img_h); bx[3] = fminf(fmaxf(bx[3], 0.f), (float)p.img_h);
    valid = valid && (bx[2] - bx[0] > 0.f) && (bx[3] - bx[1] > 0.f);
  }
  __syncthreads();
  // order-preserving compaction of the valid ones (k <= 1024 = one element per thread): ballot + wave prefix
  __shared__ int wave_cnt[16], wave_off[16];
  const u64 bal = __ballot(valid);
  const int lane = tid & 63, wv = tid >> 6;
  if (lane == 0) wave_cnt[wv] = __popcll(bal);
  __syncthreads();
  if (tid == 0) {
    int acc = 0;
    for (int w = 0; w < 16; ++w) { wave_off[w] = acc; acc += wave_cnt[w]; }
    s_ncand = (unsigned)acc;
  }
  __syncthreads();
  const int m = (int)s_ncand;
  float* sscore = (float*)(smem);   // hist area is free now: [1024] scores
  if (valid) {
    const int at = wave_off[wv] + __popcll(bal & ((1ull << lane) - 1ull));
    sbox[4 * at] = bx[0]; sbox[4 * at + 1] = bx[1]; sbox[4 * at + 2] = bx[2]; sbox[4 * at + 3] = bx[3];
    sscore[at] = sc;
  }
  __syncthreads();
  nms_sorted(sbox, m, p.nms_thr, mask, keep);
  // write kept boxes, order preserved
  {
    const bool kv = tid < m && keep[tid];
    const u64 b2 = __ballot(kv);
    if (lane == 0) wave_cnt[wv] = __popcll(b2);
    __syncthreads();
    if (tid == 0) {
      int acc = 0;
      for (int w = 0; w < 16; ++w) { wave_off[w] = acc; acc += wave_cnt[w]; }
      p.counts[img * 5 + lvl] = acc;
    }
    __syncthreads();
    if (kv) {
      const int at = wave_off[wv] + __popcll(b2 & ((1ull << lane) - 1ull));
      float* ob = p.boxes + (((size_t)img * 5 + lvl) * TOPK + at) * 4;
      ob[0] = sbox[4 * tid]; ob[1] = sbox[4 * tid + 1]; ob[2] = sbox[4 * tid + 2]; ob[3] = sbox[4 * tid + 3];
      p.scores[((size_t)img * 5 + lvl) * TOPK + at] = sscore[tid];
    }
  }
}

// concatenate the levels of an image (p2..p6, each score-sorted), order by score (ties: concat order), keep post_topk.
__global__ __launch_bounds__(1024) void rpn_merge_kernel(const float* __restrict__ boxes, const float* __restrict__ scores,
                                                         const int* __restrict__ counts, int post_topk,
                                                         float* __restrict__ out_boxes, float* __restrict__ out_scores,
                                                         int* __restrict__ out_count) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  u64* keys = (u64*)smem;   // 8192
  const int img = blockIdx.x, tid = threadIdx.x;
  int off[6];
  off[0] = 0;
  for (int l = 0; l < 5; ++l) off[l + 1] = off[l] + counts[img * 5 + l];
  const int total = off[5];
  for (int i = tid; i < 8192; i += 1024) keys[i] = 0;
  __syncthreads();
  for (int l = 0; l < 5; ++l)
    for (int i = tid; i < off[l + 1] - off[l]; i += 1024) {
      const unsigned key = ordered_key(scores[((size_t)img * 5 + l) * TOPK + i]);
      // payload: concat position (for the tie rule) — level and rank are recovered from it
      keys[off[l] + i] = ((u64)key << 32) | (u64)(0xffffffffu - (unsigned)(off[l] + i));
    }
  __syncthreads();
  int sort_n = 1024;
  while (sort_n < total) sort_n <<= 1;
  bitonic_desc(keys, sort_n);
  const int m = total < post_topk ? total : post_topk;
  for (int i = tid; i < m; i += 1024) {
    const unsigned pos = 0xffffffffu - (unsigned)(keys[i] & 0xffffffffu);
    int l = 0;
    while (l < 4 && (int)pos >= off[l + 1]) ++l;
    const int r = (int)pos - off[l];
    const float* b = boxes + (((size_t)img * 5 + l) * TOPK + r) * 4;
    float* o = out_boxes + ((size_t)img * post_topk + i) * 4;
    o[0] = b[0]; o[1] = b[1]; o[2] = b[2]; o[3] = b[3];
    out_scores[(size_t)img * post_topk + i] = scores[((size_t)img * 5 + l) * TOPK + r];
  }
  if (tid == 0) out_count[img] = m;
}

// ROIAlignV2 (aligned = true, sampling_ratio = 0) over 4 NHWC feature levels; one block per (roi, image), thread = channel
struct RoiArgs {
  const void* feat[4];   // [B][Hl][Wl][C] T
  int H[4], W[4];
  float scale[4];        // 1/4 .. 1/32
  const float* rois;     // [B][max_rois][4]
  const int* counts;     // [B]
  int max_rois, C;
  void* out;             // [B][max_rois][49*C] T   (k = (ph*7 + pw)*C + c)
};

template <typename T>
__device__ __forceinline__ float bilinear_at(const T* f, int H, int W, int C, int c, float y, float x) {
  if (y < -1.0f || y > (float)H || x < -1.0f || x > (float)W) return 0.f;
  if (y <= 0.f) y = 0.f;
  if (x <= 0.f) x = 0.f;
  int yl = (int)y, xl = (int)x, yh, xh;
  if (yl >= H - 1) { yh = yl = H - 1; y = (float)yl; } else yh = yl + 1;
  if (xl >= W - 1) { xh = xl = W - 1; x = (float)xl; } else xh = xl + 1;
  const float ly = y - (float)yl, lx = x - (float)xl, hy = 1.f - ly, hx = 1.f - lx;
  const float v1 = (float)f[((size_t)yl * W + xl) * C + c], v2 = (float)f[((size_t)yl * W + xh) * C + c];
  const float v3 = (float)f[((size_t)yh * W + xl) * C + c], v4 = (float)f[((size_t)yh * W + xh) * C + c];
  return hy * hx * v1 + hy * lx * v2 + ly * hx * v3 + ly * lx * v4;
}

template <typename T>
__global__ __launch_bounds__(256) void roi_align_kernel(RoiArgs p) {
  const int r = blockIdx.x, img = blockIdx.y;
  if (r >= p.counts[img]) return;
  const float* b = p.rois + ((size_t)img * p.max_rois + r) * 4;
  // level: floor(4 + log2(sqrt(area) / 224 + 1e-8)) clamped to [2, 5]
  const float area = (b[2] - b[0]) * (b[3] - b[1]);
  int lvl = (int)floorf(4.f + log2f(sqrtf(area) / 224.f + 1e-8f));
  lvl = min(max(lvl, 2), 5) - 2;
  const int H = p.H[lvl], W = p.W[lvl], C = p.C;
  const T* f = (const T*)p.feat[lvl] + (size_t)img * H * W * C;
  const float s = p.scale[lvl];
  const float x0 = b[0] * s - 0.5f, y0 = b[1] * s - 0.5f, x1 = b[2] * s - 0.5f, y1 = b[3] * s - 0.5f;
  const float rw = x1 - x0, rh = y1 - y0;
  const float bw = rw / 7.f, bh = rh / 7.f;
  const int gh = (int)ceilf(rh / 7.f), gw = (int)ceilf(rw / 7.f);
  const float count = (float)max(gh * gw, 1);
  T* o = (T*)p.out + ((size_t)img * p.max_rois + r) * 49 * C;
  for (int c = threadIdx.x; c < C; c += 256)
    for (int ph = 0; ph < 7; ++ph)
      for (int pw = 0; pw < 7; ++pw) {
        float acc = 0.f;
        for (int iy = 0; iy < gh; ++iy) {
          const float y = y0 + ph * bh + ((float)iy + .5f) * bh / (float)gh;
          for (int ix = 0; ix < gw; ++ix) {
            const float x = x0 + pw * bw + ((float)ix + .5f) * bw / (float)gw;
            acc += bilinear_at<T>(f, H, W, C, c, y, x);
          }
        }
        o[(ph * 7 + pw) * C + c] = (T)(acc / count);
      }
}

struct FinalArgs {
  const float* head;      // [B][max_rois][8]: cols 0,1 = class scores (text, background), 2..5 = box deltas
  const float* rois;      // [B][max_rois][4]
  const int* counts;      // [B]
  int max_rois;
  int img_h, img_w;       // resized image size (clip before NMS)
  int out_h, out_w;       // page size (detector_postprocess)
  float score_thr, nms_thr;
  int max_det;
  float* out_boxes;       // [B][max_rois][4]
  float* out_scores;      // [B][max_rois]
  int* out_count;         // [B]
};

__global__ __launch_bounds__(1024) void det_final_kernel(FinalArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  u64* keys = (u64*)smem;                               // 1024 * 8
  float* sbox = (float*)(smem + 8192);                  // 1024 * 16
  float* sscore = (float*)(smem + 8192 + 16384);        // 1024 * 4
  unsigned char* keep = (unsigned char*)(smem + 8192 + 16384 + 4096);
  u64* mask = (u64*)(smem + 8192 + 16384 + 4096 + 1024);   // [1000][16]
  __shared__ int wave_cnt[16], wave_off[16];
  __shared__ int s_m;
  const int img = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int n = p.counts[img];
  float bx[4] = {0, 0, 0, 0}, prob = 0.f;
  bool ok = false;
  if (tid < n) {
    const float* h = p.head + ((size_t)img * p.max_rois + tid) * 8;
    const float* a = p.rois + ((size_t)img * p.max_rois + tid) * 4;
    const float mx = fmaxf(h[0], h[1]);
    const float e0 = expf(h[0] - mx), e1 = expf(h[1] - mx);
    prob = e0 / (e0 + e1);
    const float an[4] = {a[0], a[1], a[2], a[3]}, d[4] = {h[2], h[3], h[4], h[5]};
    decode_box(an, d, 10.f, 10.f, 5.f, 5.f, bx);
    ok = isfinite(bx[0]) && isfinite(bx[1]) && isfinite(bx[2]) && isfinite(bx[3]) && isfinite(prob);
    bx[0] = fminf(fmaxf(bx[0], 0.f), (float)p.img_w); bx[2] = fminf(fmaxf(bx[2], 0.f), (float)p.img_w);
    bx[1] = fminf(fmaxf(bx[1], 0.f), (float)p.img_h); bx[3] = fminf(fmaxf(bx[3], 0.f), (float)p.img_h);
    ok = ok && prob > p.score_thr;
  }
  keys[tid] = ok ? (((u64)ordered_key(prob) << 32) | (u64)(0xffffffffu - (unsigned)tid)) : 0ull;
  sbox[4 * tid] = bx[0]; sbox[4 * tid + 1] = bx[1]; sbox[4 * tid + 2] = bx[2]; sbox[4 * tid + 3] = bx[3];
  sscore[tid] = prob;
  const u64 bal = __ballot(ok);
  if (lane == 0) wave_cnt[wv] = __popcll(bal);
  __syncthreads();
  if (tid == 0) {
    int acc = 0;
    for (int w = 0; w < 16; ++w) acc += wave_cnt[w];
    s_m = acc;
  }
  bitonic_desc(keys, 1024);
  const int m = s_m;
  // gather into score order (registers, then back to LDS)
  float gb[4] = {0, 0, 0, 0}, gs = 0.f;
  if (tid < m) {
    const unsigned src = 0xffffffffu - (unsigned)(keys[tid] & 0xffffffffu);
    gb[0] = sbox[4 * src]; gb[1] = sbox[4 * src + 1]; gb[2] = sbox[4 * src + 2]; gb[3] = sbox[4 * src + 3];
    gs = sscore[src];
  }
  __syncthreads();
  sbox[4 * tid] = gb[0]; sbox[4 * tid + 1] = gb[1]; sbox[4 * tid + 2] = gb[2]; sbox[4 * tid + 3] = gb[3];
  sscore[tid] = gs;
  __syncthreads();
  nms_sorted(sbox, m, p.nms_thr, mask, keep);
  // top-k of the kept ones, then detector_postprocess: scale to the page, clip, drop empty — order preserved
  const float fx = (float)p.out_w / (float)p.img_w, fy = (float)p.out_h / (float)p.img_h;
  bool kv = tid < m && keep[tid];
  u64 b1 = __ballot(kv);
  if (lane == 0) wave_cnt[wv] = __popcll(b1);
  __syncthreads();
  if (tid == 0) { int acc = 0; for (int w = 0; w < 16; ++w) { wave_off[w] = acc; acc += wave_cnt[w]; } }
  __syncthreads();
  const int rank = wave_off[wv] + __popcll(b1 & ((1ull << lane) - 1ull));
  float ob[4] = {0, 0, 0, 0};
  if (kv) {
    ob[0] = fminf(fmaxf(sbox[4 * tid] * fx, 0.f), (float)p.out_w);
    ob[2] = fminf(fmaxf(sbox[4 * tid + 2] * fx, 0.f), (float)p.out_w);
    ob[1] = fminf(fmaxf(sbox[4 * tid + 1] * fy, 0.f), (float)p.out_h);
    ob[3] = fminf(fmaxf(sbox[4 * tid + 3] * fy, 0.f), (float)p.out_h);
    kv = rank < p.max_det && (ob[2] - ob[0] > 0.f) && (ob[3] - ob[1] > 0.f);
  }
  __syncthreads();
  const u64 b2 = __ballot(kv);
  if (lane == 0) wave_cnt[wv] = __popcll(b2);
  __syncthreads();
  if (tid == 0) {
    int acc = 0;
    for (int w = 0; w < 16; ++w) { wave_off[w] = acc; acc += wave_cnt[w]; }
    p.out_count[img] = acc;
  }
  __syncthreads();
  if (kv) {
    const int at = wave_off[wv] + __popcll(b2 & ((1ull << lane) - 1ull));
    float* o = p.out_boxes + ((size_t)img * p.max_rois + at) * 4;
    o[0] = ob[0]; o[1] = ob[1]; o[2] = ob[2]; o[3] = ob[3];
    p.out_scores[(size_t)img * p.max_rois + at] = sscore[tid];
  }
}

// ---- blackout_bboxes: white-fill each box unless its snippet is framed by black or mostly black ------------------------------
// gray = cv2.COLOR_BGR2GRAY: (B*1868 + G*9617 + R*4899 + 8192) >> 14.  One block per box; phase 1 decides, phase 2 fills.
__global__ __launch_bounds__(256) void blackout_kernel(uint8_t* __restrict__ page, int H, int W, const int* __restrict__ boxes,
                                                       int n, int* __restrict__ changed) {
  const int bi = blockIdx.x;
  if (bi >= n) return;
  // numpy slicing semantics of image[y0:y1, x0:x1]: negative starts would wrap; the caller clips boxes to the page first
  int x0 = boxes[4 * bi], y0 = boxes[4 * bi + 1], x1 = boxes[4 * bi + 2], y1 = boxes[4 * bi + 3];
  x0 = max(0, min(x0, W)); x1 = max(0, min(x1, W)); y0 = max(0, min(y0, H)); y1 = max(0, min(y1, H));
  const int w = x1 - x0, h = y1 - y0;
  if (w <= 0 || h <= 0) return;
  __shared__ int s_black, s_frame_bad;
  if (threadIdx.x == 0) { s_black = 0; s_frame_bad = 0; }
  __syncthreads();
  int black = 0, bad = 0;
  for (int e = threadIdx.x; e < w * h; e += 256) {
    const int yy = e / w, xx = e - yy * w;
    const uint8_t* px = page + ((size_t)(y0 + yy) * W + x0 + xx) * 3;
    const int g = (px[0] * 1868 + px[1] * 9617 + px[2] * 4899 + 8192) >> 14;
    if (g == 0) ++black;
    else if (yy == 0 || yy == h - 1 || xx == 0 || xx == w - 1) bad = 1;
  }
  atomicAdd(&s_black, black);
  if (bad) atomicOr(&s_frame_bad, 1);
  __syncthreads();
  const bool framed = s_frame_bad == 0;
  const bool mostly = (double)s_black / (double)(w * h) > 0.5;
  if (framed || mostly) return;
  int ch = 0;
  for (int e = threadIdx.x; e < w * h; e += 256) {
    const int yy = e / w, xx = e - yy * w;
    uint8_t* px = page + ((size_t)(y0 + yy) * W + x0 + xx) * 3;
    if (px[0] != 255 || px[1] != 255 || px[2] != 255) ch = 1;
    px[0] = 255; px[1] = 255; px[2] = 255;
  }
  if (ch) atomicOr(changed, 1);
}

// LastLevelMaxPool: F.max_pool2d(x, kernel_size=1, stride=2) == x[:, ::2, ::2]
template <typename T>
__global__ void subsample2_kernel(const T* __restrict__ in, T* __restrict__ out, int B, int H, int W, int C) {
  const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
  const long long total = (long long)B * Ho * Wo * C;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(e % C);
    long long r = e / C;
    const int x = (int)(r % Wo);
    r /= Wo;
    const int y = (int)(r % Ho), b = (int)(r / Ho);
    out[e] = in[(((size_t)b * H + 2 * y) * W + 2 * x) * C + c];
  }
}

}  // namespace

#define CHECK_LAUNCH(ctx, what)                                                                              \
  do {                                                                                                       \
    hipError_t _e = hipGetLastError();                                                                       \
    if (_e != hipSuccess) return mhip_fail((ctx), MHIP_EHIP, what " launch: %s", hipGetErrorString(_e));    \
  } while (0)

void mhip_rpn_cell_anchors(const float sizes[5], const float ratios[3], float out[5][3][4]) {
  for (int l = 0; l < 5; ++l)
    for (int a = 0; a < 3; ++a) {
      const double area = (double)sizes[l] * sizes[l];
      const double w = sqrt(area / (double)ratios[a]);
      const double h = (double)ratios[a] * w;
      out[l][a][0] = (float)(-w / 2.0); out[l][a][1] = (float)(-h / 2.0);
      out[l][a][2] = (float)(w / 2.0);  out[l][a][3] = (float)(h / 2.0);
    }
}

int mhip_launch_rpn_proposals(mhip_ctx* ctx, const RpnDesc& d) {
  RpnLevelArgs a;
  for (int l = 0; l < 5; ++l) {
    a.head[l] = d.head[l]; a.H[l] = d.H[l]; a.W[l] = d.W[l]; a.stride[l] = d.stride[l];
    if ((long long)d.H[l] * d.W[l] * RPN_A > (1 << 22)) return mhip_fail(ctx, MHIP_EINVAL, "rpn: level %d too large", l);
    for (int k = 0; k < 3; ++k)
      for (int c = 0; c < 4; ++c) a.cell[l][k][c] = d.cell[l][k][c];
  }
  a.img_h = d.img_h; a.img_w = d.img_w; a.nms_thr = d.nms_thr;
  a.boxes = d.lvl_boxes; a.scores = d.lvl_scores; a.counts = d.lvl_counts;
  const int lds1 = 32768 + 1000 * 16 * 8;
  const int lds2 = 8192 * 8;
  static std::once_flag attr;
  std::call_once(attr, [&] {
    (void)hipFuncSetAttribute((const void*)rpn_level_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds1);
    (void)hipFuncSetAttribute((const void*)rpn_merge_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds2);
  });
  PROF_LAUNCH(ctx, MHIP_K_DET_OPS, {
    hipLaunchKernelGGL(rpn_level_kernel, dim3(5, d.images), dim3(RPN_THREADS), lds1, ctx->stream, a);
    hipLaunchKernelGGL(rpn_merge_kernel, dim3(d.images), dim3(1024), lds2, ctx->stream, d.lvl_boxes, d.lvl_scores,
                       d.lvl_counts, d.post_topk, d.out_boxes, d.out_scores, d.out_counts);
  });
  CHECK_LAUNCH(ctx, "rpn_proposals");
  return 0;
}

int mhip_launch_roi_align(mhip_ctx* ctx, int precision, const RoiDesc& d) {
  RoiArgs a;
  for (int l = 0; l < 4; ++l) { a.feat[l] = d.feat[l]; a.H[l] = d.H[l]; a.W[l] = d.W[l]; a.scale[l] = d.scale[l]; }
  a.rois = d.rois; a.counts = d.counts; a.max_rois = d.max_rois; a.C = d.C; a.out = d.out;
  dim3 grid(d.max_rois, d.images), block(256);
  if (precision == MHIP_PREC_F16) PROF_LAUNCH(ctx, MHIP_K_DET_OPS, hipLaunchKernelGGL(roi_align_kernel<_Float16>, grid, block, 0, ctx->stream, a));
  else PROF_LAUNCH(ctx, MHIP_K_DET_OPS, hipLaunchKernelGGL(roi_align_kernel<float>, grid, block, 0, ctx->stream, a));
  CHECK_LAUNCH(ctx, "roi_align");
  return 0;
}

int mhip_launch_det_final(mhip_ctx* ctx, const DetFinalDesc& d) {
  if (d.max_rois > 1000) return mhip_fail(ctx, MHIP_EINVAL, "det_final: at most 1000 proposals per image");
  FinalArgs a;
  a.head = d.head; a.rois = d.rois; a.counts = d.counts; a.max_rois = d.max_rois;
  a.img_h = d.img_h; a.img_w = d.img_w; a.out_h = d.out_h; a.out_w = d.out_w;
  a.score_thr = d.score_thr; a.nms_thr = d.nms_thr; a.max_det = d.max_det;
  a.out_boxes = d.out_boxes; a.out_scores = d.out_scores; a.out_count = d.out_count;
  const int lds = 8192 + 16384 + 4096 + 1024 + 1000 * 16 * 8;
  static std::once_flag attr;
  std::call_once(attr, [&] {
    (void)hipFuncSetAttribute((const void*)det_final_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  });
  PROF_LAUNCH(ctx, MHIP_K_DET_OPS, hipLaunchKernelGGL(det_final_kernel, dim3(d.images), dim3(1024), lds, ctx->stream, a));
  CHECK_LAUNCH(ctx, "det_final");
  return 0;
}

int mhip_launch_blackout(mhip_ctx* ctx, uint8_t* page, int H, int W, const int* boxes_dev, int n, int* changed_dev) {
  if (n <= 0) return 0;
  PROF_LAUNCH(ctx, MHIP_K_DET_OPS, hipLaunchKernelGGL(blackout_kernel, dim3(n), dim3(256), 0, ctx->stream, page, H, W, boxes_dev, n, changed_dev));
  CHECK_LAUNCH(ctx, "blackout");
  return 0;
}

int mhip_launch_subsample2(mhip_ctx* ctx, int precision, const void* in, void* out, int B, int H, int W, int C) {
  const long long total = (long long)B * ((H + 1) / 2) * ((W + 1) / 2) * C;
  dim3 grid((unsigned)std::min<long long>((total + 255) / 256, 1 << 16)), block(256);
  if (precision == MHIP_PREC_F16) PROF_LAUNCH(ctx, MHIP_K_DET_OPS, hipLaunchKernelGGL(subsample2_kernel<_Float16>, grid, block, 0, ctx->stream, (const _Float16*)in, (_Float16*)out, B, H, W, C));
  else PROF_LAUNCH(ctx, MHIP_K_DET_OPS, hipLaunchKernelGGL(subsample2_kernel<float>, grid, block, 0, ctx->stream, (const float*)in, (float*)out, B, H, W, C));
  CHECK_LAUNCH(ctx, "subsample2");
  return 0;
}
