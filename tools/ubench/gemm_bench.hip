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

  // ---- one encoder layer as vit_api.hip launches it, both ways of keeping the residual stream (D = 768) ----
  //   fp32 stream: q|k 768->1536, V^T (W_v as the row operand), proj -> fp32 + fp32 residual, fc1 + GELU, fc2 -> fp32 + fp32 residual
  //   split stream: the same five with the LayerNorm-folded epilogues (EPI_LN_ROWS / EPI_LN_COLS / EPI_SPLIT)
  {
    const int D = 768;
    float *X32, *vec;                 // fp32 stream; per-row / per-column vectors (rstd, mean * rstd, column sums, biases)
    _Float16 *Xhi, *Xlo, *VT;
    float* stats;
    CK(hipMalloc((void**)&X32, (size_t)rows * D * 4)); CK(hipMalloc((void**)&Xhi, (size_t)rows * D * 2)); CK(hipMalloc((void**)&Xlo, (size_t)rows * D * 2));
    CK(hipMalloc((void**)&VT, ((size_t)rows * D + 4096) * 2)); CK(hipMalloc((void**)&stats, (size_t)rows * (D / 64) * 8));
    const size_t nvec = (size_t)(rows > 3072 ? rows : 3072) + 64;
    CK(hipMalloc((void**)&vec, nvec * 4));
    CK(hipMemset(X32, 0, (size_t)rows * D * 4));
    std::vector<float> ones(nvec, 1.0f);
    CK(hipMemcpy(vec, ones.data(), nvec * 4, hipMemcpyHostToDevice));
    fill_kernel<<<1024, 256, 0, ctx.stream>>>(Xhi, (size_t)rows * D, 5u, 1.0f);
    fill_kernel<<<1024, 256, 0, ctx.stream>>>(Xlo, (size_t)rows * D, 6u, 0.0005f);
    struct G { const char* name; int split; ConvDesc d; double fl; };
    std::vector<G> gs;
    auto base = [&](const void* in, const void* w, long long M, int N, int K, void* out) {
      ConvDesc d; d.in = in; d.w = w; d.out = out; d.B = 1; d.H = 1; d.W = (int)M; d.Cin = K; d.N = N; d.bias = bias; return d;
    };
    for (int split = 0; split < 2; ++split) {
      ConvDesc qk = base(split ? (void*)Xhi : (void*)A, W, rows, 2 * D, D, C);
      ConvDesc vt = base(W, split ? (void*)Xhi : (void*)A, D, rows, D, VT); vt.bias = nullptr;
      ConvDesc pj = base(A, W, rows, D, D, split ? (void*)Xhi : (void*)X32);
      ConvDesc f1 = base(split ? (void*)Xhi : (void*)A, W, rows, 4 * D, D, C); f1.relu = ACT_GELU;
      ConvDesc f2 = base(A, W, rows, D, 4 * D, split ? (void*)Xhi : (void*)X32);
      if (split) {
        qk.epi = f1.epi = EPI_LN_ROWS; qk.ln_a = f1.ln_a = vec; qk.ln_b = f1.ln_b = vec; qk.ln_cs = f1.ln_cs = vec;
        vt.epi = EPI_LN_COLS; vt.ln_a = vec; vt.ln_b = vec; vt.ln_cs = vec; vt.row_bias = vec;
        for (ConvDesc* p : {&pj, &f2}) { p->epi = EPI_SPLIT; p->out2 = Xlo; p->res = Xhi; p->res2 = Xlo; p->stats = stats; p->stats_ld = rows; p->scale = vec; }
      } else {
        for (ConvDesc* p : {&pj, &f2}) { p->out_f32 = 1; p->res = X32; p->scale = vec; }
      }
      gs.push_back({split ? "split q|k" : "fp32  q|k", split, qk, 2.0 * rows * D * 2.0 * D});
      gs.push_back({split ? "split V^T" : "fp32  V^T", split, vt, 2.0 * rows * D * (double)D});
      gs.push_back({split ? "split proj" : "fp32  proj", split, pj, 2.0 * rows * D * (double)D});
      gs.push_back({split ? "split fc1" : "fp32  fc1", split, f1, 2.0 * rows * D * 4.0 * D});
      gs.push_back({split ? "split fc2" : "fp32  fc2", split, f2, 2.0 * rows * D * 4.0 * D});
    }
    std::vector<double> best(gs.size(), 1e30);
    for (int round = 0; round < 3; ++round)          // interleaved rounds in one process
      for (size_t k = 0; k < gs.size(); ++k) {
        for (int w = 0; w < 1; ++w) if (mhip_launch_conv_igemm(&ctx, MHIP_PREC_F16, gs[k].d)) return 1;
        CK(hipEventRecord(e0, ctx.stream));
        for (int i = 0; i < iters; ++i) if (mhip_launch_conv_igemm(&ctx, MHIP_PREC_F16, gs[k].d)) return 1;
        CK(hipEventRecord(e1, ctx.stream));
        CK(hipStreamSynchronize(ctx.stream));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms / iters < best[k]) best[k] = ms / iters;
      }
    double t[2] = {0, 0}, f[2] = {0, 0};
    for (size_t k = 0; k < gs.size(); ++k) {
      printf("%-12s %8.3f ms  %7.1f TFLOP/s\n", gs[k].name, best[k], gs[k].fl / best[k] * 1e-9);
      t[gs[k].split] += best[k]; f[gs[k].split] += gs[k].fl;
    }
    printf("encoder layer, fp32 stream  %8.3f ms  %7.1f TFLOP/s (+ two LayerNorm passes)\n", t[0], f[0] / t[0] * 1e-9);
    printf("encoder layer, split stream %8.3f ms  %7.1f TFLOP/s\n", t[1], f[1] / t[1] * 1e-9);
  }
  return 0;
}
