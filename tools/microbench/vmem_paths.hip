// Microbenchmark (tools/, not part of the library): how fast can one CU pull L2-resident data
//   (a) into LDS with global_load_lds_dwordx4 (LDS-DMA), and
//   (b) into VGPRs with global_load_dwordx4,
// 512 threads per workgroup, one workgroup per CU (128 KiB of LDS requested), every wave-instruction moves 1 KiB.
// Each workgroup walks a window of `window` bytes (shared by all workgroups: L2 / MALL resident after the first pass) in
// 64 KiB steps: 8 instructions per wave and step, `iters` steps, with a counted wait so that `depth` steps stay in flight.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float float4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void glds16(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

template <int MODE>   // 0 = LDS-DMA, 1 = VGPR loads, 2 = half / half, 3 = LDS-DMA with the GEMM's strided pattern
__global__ __launch_bounds__(512) void pull(const char* base, size_t window, int iters, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int wave_s = __builtin_amdgcn_readfirstlane(wave);
  float4v acc = {0.f, 0.f, 0.f, 0.f};
  float4v r[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) r[q] = (float4v){0.f, 0.f, 0.f, 0.f};
  size_t off = ((size_t)blockIdx.x * 65536) % window;
  for (int it = 0; it < iters; ++it) {
    const char* src = base + off + (size_t)wave * 8192 + (size_t)lane * 16;
    char* dst = smem + (it & 1) * 65536 + wave_s * 8192;
    if (MODE == 3) {
      // conv_igemm's staging of a K = 768 f16 GEMM: instruction q of a wave covers 8 rows x 128 B of rows 1536 B apart
      const size_t region = ((size_t)(it / 12) * 786432 + (size_t)blockIdx.x % 7 * 786432) % (window - 786432);
      const char* s3 = base + region + (size_t)(wave * 8 + (lane >> 3)) * 1536 + (size_t)(it % 12) * 128 + (size_t)(lane & 7) * 16;
#pragma unroll
      for (int q = 0; q < 8; ++q) glds16(s3 + (size_t)q * 64 * 1536, dst + q * 1024);
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      continue;
    }
    if (MODE != 0) {                    // consume the previous step's registers (this is where their loads are waited for)
#pragma unroll
      for (int q = (MODE == 2 ? 4 : 0); q < 8; ++q) acc += r[q];
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      if (MODE == 0 || (MODE == 2 && q < 4)) glds16(src + q * 1024, dst + q * 1024);
      else r[q] = *(const float4v*)(src + q * 1024);
    }
    if (MODE == 0) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // one step (8 instructions) stays in flight
    off += 65536 * 7;                                                  // stride through the window
    while (off >= window) off -= window;
  }
#pragma unroll
  for (int q = 0; q < 8; ++q) acc += r[q];
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) sink[tid] = acc[0];
  if (MODE != 1 && smem[tid] == 77 && iters < 0) sink[tid] = 1.f;
}

extern "C" int vmem_pull(int mode, const void* base, size_t window, int iters, int blocks, void* sink, float* ms_out) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const size_t lds = 131072;
  hipFuncSetAttribute((const void*)pull<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipFuncSetAttribute((const void*)pull<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipFuncSetAttribute((const void*)pull<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipFuncSetAttribute((const void*)pull<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0, 0);
    if (mode == 0) hipLaunchKernelGGL(pull<0>, dim3(blocks), dim3(512), lds, 0, (const char*)base, window, iters, (float*)sink);
    else if (mode == 1) hipLaunchKernelGGL(pull<1>, dim3(blocks), dim3(512), lds, 0, (const char*)base, window, iters, (float*)sink);
    else if (mode == 2) hipLaunchKernelGGL(pull<2>, dim3(blocks), dim3(512), lds, 0, (const char*)base, window, iters, (float*)sink);
    else hipLaunchKernelGGL(pull<3>, dim3(blocks), dim3(512), lds, 0, (const char*)base, window, iters, (float*)sink);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
  }
  hipEventElapsedTime(ms_out, e0, e1);
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  return (int)hipGetLastError();
}
