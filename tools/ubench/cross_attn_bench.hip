// Micro-benchmark of the absorbed encoder-attention kernels (marie_icr_amd/csrc/cross_attn.hip) on random device data:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form=1 [-DVARIANT...] cross_attn_bench.hip -o x
//   ./x [crops] [beam] [heads] [n_tok] [enc_dim] [iters]
// The production TU is included as is; only the two library hooks it needs are defined here.  Measurement aid, not shipped.
#include <stdarg.h>
#include <stdlib.h>

#include <vector>

#include "../../marie_icr_amd/csrc/cross_attn.hip"

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

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
  const int crops = argc > 1 ? atoi(argv[1]) : 1280, beam = argc > 2 ? atoi(argv[2]) : 3, heads = argc > 3 ? atoi(argv[3]) : 16;
  const int n_tok = argc > 4 ? atoi(argv[4]) : 577, ED = argc > 5 ? atoi(argv[5]) : 768, iters = argc > 6 ? atoi(argv[6]) : 20;
  const int D = heads * 64, M = crops * beam, npad = (n_tok + 7) / 8 * 8;
  mhip_ctx ctx;
  CK(hipStreamCreate(&ctx.stream));
  auto fill = [&](size_t n, float amp) {
    std::vector<_Float16> h(n);
    unsigned s = 12345u + (unsigned)n;
    for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = (_Float16)(((int)(s >> 9) % 2001 - 1000) * 0.001f * amp); }
    void* d = nullptr;
    if (hipMalloc(&d, n * 2) != hipSuccess) return (void*)nullptr;
    (void)hipMemcpy(d, h.data(), n * 2, hipMemcpyHostToDevice);
    return d;
  };
  CrossAbsorbDesc d;
  d.q = fill((size_t)M * D, 0.6f); d.ldq = D;
  d.E = fill(((size_t)crops * (npad + 64) + 64) * ED, 1.0f); d.kv_rows = npad; d.n_keys = n_tok; d.enc_dim = ED;
  d.wkt = fill((size_t)D * ED, 0.08f); d.wv = fill((size_t)D * ED, 0.04f);
  float* bv = nullptr;
  CK(hipMalloc((void**)&bv, D * 4));
  CK(hipMemset(bv, 0, D * 4));
  d.bv = bv;
  void *qt, *ct, *ao;
  CK(hipMalloc(&qt, (size_t)M * 16 * ED * 2)); CK(hipMalloc(&ct, (size_t)M * 16 * ED * 2)); CK(hipMalloc(&ao, (size_t)M * D * 2));
  d.qt = qt; d.ct = ct; d.ao = ao; d.ldo = D; d.crops = crops; d.beam = beam; d.heads = heads;
  if (!d.q || !d.E || !d.wkt || !d.wv) { fprintf(stderr, "alloc failed\n"); return 1; }
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int w = 0; w < 3; ++w) if (mhip_launch_cross_absorbed(&ctx, d)) return 1;
  CK(hipStreamSynchronize(ctx.stream));
  CK(hipEventRecord(e0, ctx.stream));
  for (int i = 0; i < iters; ++i) if (mhip_launch_cross_absorbed(&ctx, d)) return 1;
  CK(hipEventRecord(e1, ctx.stream));
  CK(hipStreamSynchronize(ctx.stream));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double gb = ((double)crops * n_tok * ED * 2 + 2.0 * M * 16 * ED * 2 * 2 + (double)M * D * 2 * 2) / 1e9;
  printf("crops %d beam %d heads %d n_tok %d ED %d: %.1f us per layer-step (3 kernels), %.2f TB/s over %.2f GB compulsory\n", crops, beam,
         heads, n_tok, ED, 1e3 * ms / iters, gb / (ms / iters * 1e-3) / 1e3, gb);
  // checksum so that variants can be compared for equal results
  std::vector<_Float16> ho((size_t)M * D);
  CK(hipMemcpy(ho.data(), ao, ho.size() * 2, hipMemcpyDeviceToHost));
  double cs = 0;
  for (size_t i = 0; i < ho.size(); ++i) cs += (double)ho[i] * ((i % 7) + 1);
  printf("checksum %.6f\n", cs);
  return 0;
}
