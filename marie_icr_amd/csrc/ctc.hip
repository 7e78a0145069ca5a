// ctc.hip — greedy CTC decode + confidence, one 64-lane wavefront per text line.
//
// Replaces: `_, preds_index = preds.max(2)` (marie/document/craft_ocr_processor.py:240),
// CTCLabelConverter.decode's collapse rule (marie/models/icr/utils.py:41-54: drop blank 0 and
// repeats) and the confidence `softmax(dim=2).max(dim=2).cumprod(0)[-1]`
// (marie/document/craft_ocr_processor.py:255-271).
//
// Each lane holds classes lane, lane+64, ...; arg-max and the softmax denominator are butterfly
// reductions over the wave with DPP/ds_swizzle shuffles (no LDS, no atomics).  Ties resolve to the
// LOWEST class index, which is what torch.max returns on CPU.  The maximum softmax probability is
// exp(0)/sum = 1/sum, so a single pass over the logits suffices.  Lane 0 keeps the running
// confidence product and the collapse state in scalar registers.
#include "common.h"

namespace {

constexpr int MAXC_PER_LANE = 4;  // up to 256 classes

__global__ __launch_bounds__(64) void ctc_decode_kernel(const float* __restrict__ logits, int n, int T, int C,
                                                        int32_t* __restrict__ argmax, int32_t* __restrict__ tokens,
                                                        int32_t* __restrict__ lengths, float* __restrict__ conf) {
  const int b = blockIdx.x;
  const int lane = threadIdx.x;
  const float* row = logits + (size_t)b * T * C;
  float cprod = 1.0f;
  int prev = -1, len = 0;
  for (int t = 0; t < T; ++t) {
    float v[MAXC_PER_LANE];
    float best = -INFINITY;
    int bidx = 0x7fffffff;
#pragma unroll
    for (int q = 0; q < MAXC_PER_LANE; ++q) {
      int cidx = lane + 64 * q;
      v[q] = (cidx < C) ? row[(size_t)t * C + cidx] : -INFINITY;
      if (cidx < C && v[q] > best) {  // strict > keeps the lowest index within the lane
        best = v[q];
        bidx = cidx;
      }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      float ov = __shfl_xor(best, off);
      int oi = __shfl_xor(bidx, off);
      if (ov > best || (ov == best && oi < bidx)) {
        best = ov;
        bidx = oi;
      }
    }
    float e = 0.f;
#pragma unroll
    for (int q = 0; q < MAXC_PER_LANE; ++q)
      if (lane + 64 * q < C) e += expf(v[q] - best);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) e += __shfl_xor(e, off);
    if (lane == 0) {
      cprod *= 1.0f / e;
      argmax[(size_t)b * T + t] = bidx;
      if (bidx != 0 && bidx != prev) tokens[(size_t)b * T + len++] = bidx;
      prev = bidx;
    }
  }
  if (lane == 0) {
    for (int i = len; i < T; ++i) tokens[(size_t)b * T + i] = 0;
    lengths[b] = len;
    conf[b] = cprod;
  }
}

}  // namespace

int mhip_launch_ctc_decode(mhip_ctx* ctx, const float* logits, int n, int T, int C, int32_t* argmax,
                           int32_t* tokens, int32_t* lengths, float* conf) {
  if (n < 1 || T < 1 || C < 1 || C > 64 * MAXC_PER_LANE)
    return mhip_fail(ctx, MHIP_EINVAL, "ctc_decode: bad shape n=%d T=%d C=%d", n, T, C);
  PROF_LAUNCH(ctx, MHIP_K_CTC_DECODE,
              hipLaunchKernelGGL(ctc_decode_kernel, dim3(n), dim3(64), 0, ctx->stream, logits, n, T, C, argmax,
                                 tokens, lengths, conf));
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return mhip_fail(ctx, MHIP_EHIP, "ctc_decode launch: %s", hipGetErrorString(e));
  return 0;
}
