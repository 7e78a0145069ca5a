// Micro-benchmark of the plain-GEMM path of marie_icr_amd/csrc/conv_igemm.hip on the ViT encoder shapes:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DVARIANT...] gemm_bench.hip -o gb_x
//   ./gb_x [rows] [iters]
// The production TU is included as is; only the library hooks it needs are defined here.  Measurement aid, not shipped.
#include <stdarg.h>
#include <stdlib.h>

#include <vector>

#include "../../marie_icr_amd/csrc/conv_igemm.hip"

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
int mhip_try_launch_conv3x3_patch(mhip_ctx*, int, const ConvDesc&, igemm::IgemmArgs&) { return 1; }

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void fill_kernel(_Float16* p, size_t n, unsigned seed, float amp) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned s = (unsigned)i * 2654435761u + seed;
    s ^= s >> 15; s *= 2246822519u; s ^= s >> 13;
    p[i] = (_Float16)(((int)(s >> 9) % 2001 - 1000) * 0.001f * amp);
  }
}
__global__ void checksum_kernel(const _Float16* p, size_t n, size_t stride, double* out) {
  double s = 0;
  for (size_t i = threadIdx.x; i * stride < n; i += blockDim.x) s += (double)(float)p[i * stride] * (double)((i % 7) + 1);
  __shared__ double sh[256];
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o]; __syncthreads(); }
  if (threadIdx.x == 0) *out = sh[0];
}

int main(int argc, char** argv) {
  const int rows = argc > 1 ? atoi(argv[1]) : 640 * 577, iters = argc > 2 ? atoi(argv[2]) : 10;
  mhip_ctx ctx;
  CK(hipStreamCreate(&ctx.stream));
  CK(hipMalloc(&ctx.zeros, MHIP_ZERO_BYTES));
  CK(hipMemset(ctx.zeros, 0, MHIP_ZERO_BYTES));
  struct Shape { const char* name; int K, N, act, res; };
  const Shape shapes[] = {{"qkv  768->2304", 768, 2304, ACT_NONE, 0}, {"proj 768->768 +res", 768, 768, ACT_NONE, 1},
                          {"fc1  768->3072 gelu", 768, 3072, ACT_GELU, 0}, {"fc2  3072->768 +res", 3072, 768, ACT_NONE, 1},
                          {"sweep 1536->2304", 1536, 2304, ACT_NONE, 0}, {"sweep 3072->2304", 3072, 2304, ACT_NONE, 0},
                          {"sweep 768->768", 768, 768, ACT_NONE, 0}, {"sweep 3072->768", 3072, 768, ACT_NONE, 0}};
  _Float16 *A, *W, *C, *R;
  float* bias;
  const size_t amax = (size_t)rows * 3072, cmax = (size_t)rows * 3072;
  CK(hipMalloc((void**)&A, amax * 2 + 4096)); CK(hipMalloc((void**)&C, cmax * 2)); CK(hipMalloc((void**)&R, (size_t)rows * 768 * 2));
  CK(hipMalloc((void**)&W, (size_t)3072 * 3072 * 2)); CK(hipMalloc((void**)&bias, 3072 * 4));
  CK(hipMemset(bias, 0, 3072 * 4));
  const float amp = getenv("GB_ZERO") ? 0.f : 1.f;      // GB_ZERO=1: zero operands (what the clock does under load, not a result)
  fill_kernel<<<1024, 256, 0, ctx.stream>>>(A, amax, 1u, 1.0f * amp);
  fill_kernel<<<1024, 256, 0, ctx.stream>>>(W, (size_t)3072 * 3072, 2u, 0.05f * amp);
  fill_kernel<<<1024, 256, 0, ctx.stream>>>(R, (size_t)rows * 768, 3u, 1.0f);
  double* dsum;
  CK(hipMalloc((void**)&dsum, 8));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  double tot_fl = 0, tot_ms = 0;
  for (const Shape& s : shapes) {
    ConvDesc d;
    d.in = A; d.w = W; d.bias = bias; d.out = C; d.B = 1; d.H = 1; d.W = rows; d.Cin = s.K; d.N = s.N; d.relu = s.act;
    d.res = s.res ? R : nullptr;
    for (int w = 0; w < 2; ++w) if (mhip_launch_conv_igemm(&ctx, MHIP_PREC_F16, d)) return 1;
    CK(hipStreamSynchronize(ctx.stream));
    CK(hipEventRecord(e0, ctx.stream));
    for (int i = 0; i < iters; ++i) if (mhip_launch_conv_igemm(&ctx, MHIP_PREC_F16, d)) return 1;
    CK(hipEventRecord(e1, ctx.stream));
    CK(hipStreamSynchronize(ctx.stream));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= iters;
    checksum_kernel<<<1, 256, 0, ctx.stream>>>(C, (size_t)rows * s.N, 4099, dsum);
    double h = 0;
    CK(hipMemcpy(&h, dsum, 8, hipMemcpyDeviceToHost));
    const double fl = 2.0 * rows * (double)s.K * s.N;
    printf("%-22s %8.3f ms  %7.1f TFLOP/s  checksum %.6f\n", s.name, ms, fl / ms * 1e-9, h);
    if (s.name[0] != 's') { tot_fl += fl; tot_ms += ms; }
    const double tiles = (double)((rows + 255) / 256) * ((s.N + 255) / 256), rounds = tiles / 256.0;
    printf("      %.0f tiles, %.1f rounds of 256, %.2f us per tile (%d slices)\n", tiles, rounds, ms * 1e3 / rounds, s.K / 64);
  }
  printf("layer (4 GEMMs)        %8.3f ms  %7.1f TFLOP/s\n", tot_ms, tot_fl / tot_ms * 1e-9);
  return 0;
}
