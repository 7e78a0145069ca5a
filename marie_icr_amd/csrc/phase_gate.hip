// phase_gate.hip — ordering between two streams that are fed by two host threads.
//
// The engine runs the detector of page batch k + 1 under the recognizer of batch k (the reference's loop is strictly serial:
// marie/ocr/ocr_engine.py:172-221).  Both models are MFMA-bound while the recognizer encodes and the recognizer is HBM- and
// latency-bound while it decodes, so the detector belongs under the DECODE phase.  A gate is how the recognizer's thread says
// "my decode phase of call n starts here in my stream" and how the detector's thread makes its own stream wait for that point:
//   signal : hipEventRecord on the signaller's stream + sequence number n (host side, under a mutex)
//   wait   : the host blocks until signal n exists (bounded), then hipStreamWaitEvent on the waiter's stream
// An opened gate (mhip_gate_open) lets every present and future wait through at once — the error / shutdown path.
#include <chrono>
#include <condition_variable>

#include "common.h"

struct mhip_gate {
  static constexpr int RING = 16;
  int device = 0;
  std::mutex mu;
  std::condition_variable cv;
  long long count = 0;
  bool open = false;
  hipEvent_t ev[RING] = {};
};

extern "C" int mhip_gate_create(mhip_ctx* ctx, mhip_gate** out) {
  if (!ctx || !out) return MHIP_EINVAL;
  *out = nullptr;
  MHIP_HIP(ctx, hipSetDevice(ctx->device));
  mhip_gate* g = new mhip_gate();
  g->device = ctx->device;
  for (int i = 0; i < mhip_gate::RING; ++i) {
    hipError_t e = hipEventCreateWithFlags(&g->ev[i], hipEventDisableTiming);
    if (e != hipSuccess) {
      for (int j = 0; j < i; ++j) (void)hipEventDestroy(g->ev[j]);
      delete g;
      return mhip_fail(ctx, MHIP_EHIP, "gate: hipEventCreate: %s", hipGetErrorString(e));
    }
  }
  *out = g;
  return MHIP_OK;
}

extern "C" int mhip_gate_destroy(mhip_gate* g) {
  if (!g) return MHIP_OK;
  {
    std::lock_guard<std::mutex> lk(g->mu);
    g->open = true;
  }
  g->cv.notify_all();
  (void)hipSetDevice(g->device);
  for (auto e : g->ev) (void)hipEventDestroy(e);
  delete g;
  return MHIP_OK;
}

extern "C" int mhip_gate_signal(mhip_gate* g, mhip_ctx* ctx) {
  if (!g || !ctx) return MHIP_EINVAL;
  if (ctx->device != g->device) return mhip_fail(ctx, MHIP_EINVAL, "gate: context on device %d, gate on %d", ctx->device, g->device);
  {
    std::lock_guard<std::mutex> lk(g->mu);
    MHIP_HIP(ctx, hipEventRecord(g->ev[(g->count + 1) % mhip_gate::RING], ctx->stream));
    ++g->count;
  }
  g->cv.notify_all();
  return MHIP_OK;
}

extern "C" long long mhip_gate_count(mhip_gate* g) {
  if (!g) return MHIP_EINVAL;
  std::lock_guard<std::mutex> lk(g->mu);
  return g->count;
}

extern "C" int mhip_gate_open(mhip_gate* g, int open) {
  if (!g) return MHIP_EINVAL;
  {
    std::lock_guard<std::mutex> lk(g->mu);
    g->open = open != 0;
  }
  g->cv.notify_all();
  return MHIP_OK;
}

// 1 = the stream now waits for signal `seq` (or a later one), 0 = passed without a device-side wait (gate open, or the signal
// did not come within timeout_ms: the caller proceeds unordered — late is better than stuck), < 0 = error.
extern "C" int mhip_gate_wait(mhip_gate* g, mhip_ctx* ctx, long long seq, int timeout_ms) {
  if (!g || !ctx || seq < 1) return MHIP_EINVAL;
  if (ctx->device != g->device) return mhip_fail(ctx, MHIP_EINVAL, "gate: context on device %d, gate on %d", ctx->device, g->device);
  std::unique_lock<std::mutex> lk(g->mu);
  const bool ok = g->cv.wait_for(lk, std::chrono::milliseconds(timeout_ms < 0 ? 0 : timeout_ms), [&] { return g->open || g->count >= seq; });
  if (!ok || g->open) return 0;
  // the newest signal is at or after `seq` in the signaller's stream; its event cannot be re-recorded while the mutex is held
  MHIP_HIP(ctx, hipSetDevice(ctx->device));
  MHIP_HIP(ctx, hipStreamWaitEvent(ctx->stream, g->ev[g->count % mhip_gate::RING], 0));
  return 1;
}
