// ccl.hip — score-map post-processing on the GPU: binarise, 4-connected component labelling, statistics.
//
// Replaces the O(pixels) part of getDetBoxes_core (marie/models/craft/craft_utils.py:25-38,44-53):
//   cv2.threshold(textmap, low_text), cv2.threshold(linkmap, link_threshold), clip(sum),
//   cv2.connectedComponentsWithStats(connectivity=4) and the per-label area / bounding box / max(textmap).
//
// HBM-bound integer work: one pass to binarise (8 B in, 1 B out per pixel), a union-find labelling whose
// unions are atomicMin on a parent array (root = smallest pixel index of the component = its first pixel in
// raster order, which is exactly OpenCV's label order), a two-level prefix sum that turns roots into
// consecutive label numbers, and one pass of per-component atomics for the statistics.
#include "common.h"

namespace {

constexpr int SCAN_ITEMS = 2048;  // pixels per block in the root-ranking scan (256 threads x 8)

__device__ __forceinline__ int float_to_ordered(float f) {
  int b = __float_as_int(f);
  return b >= 0 ? b : b ^ 0x7fffffff;  // monotone: larger float -> larger int
}

// One block per score-map row: binarise, and point every foreground pixel at the first pixel of its horizontal
// run (prefix-max of "last background x" over the row).  Runs, not pixels, are then the union-find elements, so a
// page-sized blob costs one union per row instead of a million contended atomics.
__global__ __launch_bounds__(256) void ccl_rows_kernel(const float* __restrict__ scores, int H, int W,
                                                       float low_text, float link_thr, uint8_t* __restrict__ flags,
                                                       int* __restrict__ parent) {
  __shared__ int wmax[4];
  __shared__ int carry_s;
  const int y = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) carry_s = -1;
  __syncthreads();
  for (int x0 = 0; x0 < W; x0 += 256) {
    const int x = x0 + threadIdx.x;
    uint8_t f = 0;
    if (x < W) {
      const float2 s = ((const float2*)scores)[(size_t)y * W + x];
      f = (s.x > low_text ? 1 : 0) | (s.y > link_thr ? 2 : 0);   // cv2.threshold: strictly greater
      flags[(size_t)y * W + x] = f;
    }
    int v = (x < W && f == 0) ? x : -1;          // last background position so far
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {     // inclusive max-scan inside the wave
      int o = __shfl_up(v, off);
      if (lane >= off) v = max(v, o);
    }
    if (lane == 63) wmax[wave] = v;
    __syncthreads();
    int pre = carry_s;
    for (int q = 0; q < wave; ++q) pre = max(pre, wmax[q]);
    v = max(v, pre);
    if (x < W) parent[(size_t)y * W + x] = f ? y * W + (v + 1) : -1;
    __syncthreads();
    if (threadIdx.x == 255) carry_s = v;         // v of the last thread = max over the whole segment
    __syncthreads();
  }
}

__device__ __forceinline__ int find_root(volatile int* L, int i) {
  int p = L[i];
  while (p != i) {
    i = p;
    p = L[i];
  }
  return i;
}

__device__ __forceinline__ void unite(int* L, int a, int b) {
  while (true) {
    a = find_root(L, a);
    b = find_root(L, b);
    if (a == b) return;
    if (a > b) {
      int t = a;
      a = b;
      b = t;
    }
    const int old = atomicMin(&L[b], a);  // hang the larger root under the smaller one
    if (old == b) return;
    b = old;                              // someone re-parented b meanwhile: retry from there
  }
}

// one union per pair of vertically touching runs: at the first column of every contact segment
__global__ __launch_bounds__(256) void ccl_merge_kernel(int* __restrict__ parent, const uint8_t* __restrict__ flags,
                                                        int H, int W) {
  const int n = H * W;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x + W; i < n; i += gridDim.x * blockDim.x) {
    if (!flags[i] || !flags[i - W]) continue;
    const int x = i % W;
    if (x > 0 && flags[i - 1] && flags[i - W - 1]) continue;   // same contact segment as the pixel on the left
    unite(parent, parent[i], parent[i - W]);
  }
}

// compress the forest at the run heads (path to the root can be long for a tall blob) ...
__global__ __launch_bounds__(256) void ccl_compress_heads_kernel(int* __restrict__ parent,
                                                                 const uint8_t* __restrict__ flags, int H, int W) {
  const int n = H * W;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    if (!flags[i]) continue;
    const int x = i % W;
    if (x > 0 && flags[i - 1]) continue;            // not a run head
    const int r = find_root(parent, i);
    if (r != i) parent[i] = r;                      // roots never change in this kernel: race-free
  }
}

// ... then every other pixel takes its run head's (now final) root
__global__ __launch_bounds__(256) void ccl_flatten_kernel(int* __restrict__ parent, const uint8_t* __restrict__ flags,
                                                          int H, int W) {
  const int n = H * W;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    if (!flags[i]) continue;
    const int x = i % W;
    if (x > 0 && flags[i - 1]) parent[i] = parent[parent[i]];   // parent[i] is still the run head
  }
}

// level 1: roots per 2048-pixel block
__global__ __launch_bounds__(256) void ccl_count_roots_kernel(const int* __restrict__ parent, int n,
                                                              int* __restrict__ blocksum) {
  __shared__ int wsum[4];
  const int base = blockIdx.x * SCAN_ITEMS;
  int c = 0;
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS / 256; ++k) {
    const int i = base + k * 256 + threadIdx.x;
    c += (i < n && parent[i] == i) ? 1 : 0;
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) c += __shfl_xor(c, off);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) blocksum[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// level 2: exclusive scan of the block sums (one block), total -> n_labels (+1 for the background)
__global__ __launch_bounds__(1024) void ccl_scan_blocks_kernel(int* __restrict__ blocksum, int nblocks,
                                                               int* __restrict__ n_labels) {
  __shared__ int part[1024];
  const int per = (nblocks + 1023) / 1024;
  const int lo = threadIdx.x * per, hi = min(lo + per, nblocks);
  int s = 0;
  for (int i = lo; i < hi; ++i) s += blocksum[i];
  part[threadIdx.x] = s;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {   // Hillis-Steele inclusive scan
    int v = threadIdx.x >= off ? part[threadIdx.x - off] : 0;
    __syncthreads();
    part[threadIdx.x] += v;
    __syncthreads();
  }
  int run = threadIdx.x ? part[threadIdx.x - 1] : 0;
  for (int i = lo; i < hi; ++i) {
    const int v = blocksum[i];
    blocksum[i] = run;
    run += v;
  }
  if (threadIdx.x == 1023) *n_labels = part[1023] + 1;
}

// level 3: rank of every root = offset of its block + roots before it inside the block; written into labels[]
__global__ __launch_bounds__(256) void ccl_rank_roots_kernel(const int* __restrict__ parent, int n,
                                                             const int* __restrict__ blocksum,
                                                             int* __restrict__ labels) {
  __shared__ int wsum[4];
  const int base = blockIdx.x * SCAN_ITEMS;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int running = blocksum[blockIdx.x];
  for (int k = 0; k < SCAN_ITEMS / 256; ++k) {
    const int i = base + k * 256 + threadIdx.x;
    const bool root = i < n && parent[i] == i;
    const unsigned long long m = __ballot(root);
    const int before = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) wsum[wave] = __popcll(m);
    __syncthreads();
    int off = 0;
    for (int w = 0; w < wave; ++w) off += wsum[w];
    const int tot = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    if (root) labels[i] = running + off + before + 1;   // label numbers start at 1
    running += tot;
    __syncthreads();
  }
}

// labels for non-root foreground pixels + statistics.  A thread walks 8 consecutive pixels of a row and flushes
// its partial (area, x-range, max score) only when the label changes: 8x fewer atomics inside large components.
__global__ __launch_bounds__(256) void ccl_stats_kernel(const int* __restrict__ parent,
                                                        const float* __restrict__ scores, int H, int W,
                                                        int* __restrict__ labels, int* __restrict__ stats) {
  const int segs = (W + 7) / 8;
  const long long total = (long long)H * segs;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += (long long)gridDim.x * blockDim.x) {
    const int y = (int)(t / segs), xs = (int)(t % segs) * 8, xe = min(xs + 8, W);
    int cur = 0, area = 0, x0 = 0, x1 = 0, tmax = (int)0x80000000;
    for (int x = xs; x <= xe; ++x) {
      int k = 0;
      if (x < xe) {
        const size_t i = (size_t)y * W + x;
        const int r = parent[i];
        if (r >= 0) {
          k = labels[r];                 // roots were labelled by the previous kernel
          if ((size_t)r != i) labels[i] = k;
        } else {
          labels[i] = 0;
        }
      }
      if (k != cur) {
        if (cur) {
          int* s = stats + (size_t)cur * 6;
          atomicMin(&s[0], x0);
          atomicMin(&s[1], y);
          atomicMax(&s[2], x1);
          atomicMax(&s[3], y);
          atomicAdd(&s[4], area);
          atomicMax(&s[5], tmax);
        }
        cur = k;
        area = 0;
        x0 = x;
        tmax = (int)0x80000000;
      }
      if (k) {
        ++area;
        x1 = x;
        tmax = max(tmax, float_to_ordered(scores[2 * ((size_t)y * W + x)]));
      }
    }
  }
}

__global__ __launch_bounds__(256) void ccl_init_stats_kernel(int* __restrict__ stats, const int* __restrict__ n_labels,
                                                             int max_labels) {
  const int n = min(*n_labels, max_labels);
  for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
    int* s = stats + (size_t)k * 6;
    s[0] = 0x7fffffff;
    s[1] = 0x7fffffff;
    s[2] = -1;
    s[3] = -1;
    s[4] = 0;
    s[5] = (int)0x80000000;
  }
}

}  // namespace

float mhip_ordered_bits_to_float(int bits) {
  int b = bits >= 0 ? bits : bits ^ 0x7fffffff;
  float f;
  memcpy(&f, &b, 4);
  return f;
}

static size_t al(size_t v) { return (v + 255) / 256 * 256; }

size_t mhip_ccl_workspace_bytes(int H, int W) {
  const size_t n = (size_t)H * W;
  const size_t nb = (n + SCAN_ITEMS - 1) / SCAN_ITEMS + 1;
  return al(n) + al(n * 4) + al(n * 4) + al(nb * 4) + al((n / 2 + 2) * 6 * 4) + al(4);
}

void mhip_ccl_carve(char* base, int H, int W, CclBuffers* o) {
  const size_t n = (size_t)H * W;
  const size_t nb = (n + SCAN_ITEMS - 1) / SCAN_ITEMS + 1;
  o->flags = (uint8_t*)base;  base += al(n);
  o->parent = (int*)base;     base += al(n * 4);
  o->labels = (int*)base;     base += al(n * 4);
  o->blocksum = (int*)base;   base += al(nb * 4);
  o->stats = (int*)base;      base += al((n / 2 + 2) * 6 * 4);
  o->n_labels = (int*)base;
}

int mhip_launch_ccl(mhip_ctx* ctx, const float* scores, int H, int W, float low_text, float link_thr,
                    const CclBuffers& b) {
  if (H < 1 || W < 1 || (long long)H * W > 0x7ffffff0LL) return mhip_fail(ctx, MHIP_EINVAL, "ccl: bad shape");
  const int n = H * W;
  const int nblocks = (n + SCAN_ITEMS - 1) / SCAN_ITEMS;
  const unsigned g = (unsigned)std::min<long long>(((long long)n + 255) / 256, 256 * 16);
  hipEvent_t e0 = nullptr;
  if (ctx->profiling) mhip_prof_begin(ctx, MHIP_K_CCL, &e0);
  hipLaunchKernelGGL(ccl_rows_kernel, dim3(H), dim3(256), 0, ctx->stream, scores, H, W, low_text, link_thr, b.flags,
                     b.parent);
  hipLaunchKernelGGL(ccl_merge_kernel, dim3(g), dim3(256), 0, ctx->stream, b.parent, b.flags, H, W);
  hipLaunchKernelGGL(ccl_compress_heads_kernel, dim3(g), dim3(256), 0, ctx->stream, b.parent, b.flags, H, W);
  hipLaunchKernelGGL(ccl_flatten_kernel, dim3(g), dim3(256), 0, ctx->stream, b.parent, b.flags, H, W);
  hipLaunchKernelGGL(ccl_count_roots_kernel, dim3(nblocks), dim3(256), 0, ctx->stream, b.parent, n, b.blocksum);
  hipLaunchKernelGGL(ccl_scan_blocks_kernel, dim3(1), dim3(1024), 0, ctx->stream, b.blocksum, nblocks, b.n_labels);
  hipLaunchKernelGGL(ccl_rank_roots_kernel, dim3(nblocks), dim3(256), 0, ctx->stream, b.parent, n, b.blocksum,
                     b.labels);
  hipLaunchKernelGGL(ccl_init_stats_kernel, dim3(256), dim3(256), 0, ctx->stream, b.stats, b.n_labels, n / 2 + 2);
  hipLaunchKernelGGL(ccl_stats_kernel, dim3(g), dim3(256), 0, ctx->stream, b.parent, scores, H, W, b.labels, b.stats);
  if (ctx->profiling) mhip_prof_end(ctx, MHIP_K_CCL, e0);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return mhip_fail(ctx, MHIP_EHIP, "ccl launch: %s", hipGetErrorString(e));
  return 0;
}
