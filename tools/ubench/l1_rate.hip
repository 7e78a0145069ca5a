// How fast does one CU pull L2-resident data?  global_load_lds_dwordx4 (LDS-DMA) against global_load_dwordx4 (to VGPRs).
//   hipcc --offload-arch=gfx950 -O3 l1_rate.hip -o l1_rate && ./l1_rate
// Each workgroup (512 threads, one per CU: 100 KiB of LDS) walks its own window of `win` bytes `reps` times.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float float4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void glds16(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

// GEMM-like gather: an instruction covers 8 rows x 128 B at a row pitch of `pitch` bytes (lane -> row lane>>3, 16-byte chunk lane&7),
// a wave's 8 instructions of a step cover 64 rows, the 8 waves 512 rows x 128 B = 64 KiB; steps walk along the rows (K direction)
__global__ __launch_bounds__(512) void kg(const char* src, int pitch, int ksteps, int reps, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const char* base = src + (size_t)(wave * 64 + (lane >> 3)) * pitch + (lane & 7) * 16;
  for (int r = 0; r < reps; ++r) {
    for (int ks = 0; ks < ksteps; ++ks) {
#pragma unroll
      for (int j = 0; j < 8; ++j) glds16(base + (size_t)j * 8 * pitch + ks * 128, smem + wave * 8192 + j * 1024);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  if (*(float*)(smem + tid * 4) == 123.456f) sink[0] = 1.f;
}

template <int MODE>
__global__ __launch_bounds__(512) void k(const char* src, size_t win, int reps, float* sink, int shared_window) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const char* base = src + (shared_window ? 0 : (size_t)blockIdx.x * win);
  float4v acc = {0, 0, 0, 0};
  for (int r = 0; r < reps; ++r) {
    for (size_t off = 0; off < win; off += 8 * 8192) {     // 64 KiB per step: 8 instructions per wave
      const char* p = base + off + wave * 8192 + lane * 16;
      if (MODE == 0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) glds16(p + j * 1024, smem + wave * 8192 + j * 1024);
      } else {
        float4v v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = *(const float4v*)(p + j * 1024);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += v[j];
      }
    }
    if (MODE == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  if (MODE == 0) acc[0] = *(float*)(smem + tid * 4);
  if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) sink[0] = 1.f;
}

int main() {
  const int cus = 256;
  float* sink;
  char* src;
  hipMalloc((void**)&sink, 4);
  const size_t wins[] = {64 << 10, 256 << 10, 1 << 20};
  hipMalloc((void**)&src, (size_t)cus * (1 << 20));
  hipMemset(src, 1, (size_t)cus * (1 << 20));
  hipFuncSetAttribute((const void*)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 << 10);
  hipFuncSetAttribute((const void*)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 << 10);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int shared = 0; shared < 2; ++shared)
    for (size_t win : wins)
      for (int mode = 0; mode < 2; ++mode) {
        const int reps = (int)((size_t)(256 << 20) / win);
        for (int it = 0; it < 2; ++it) {
          hipEventRecord(e0);
          if (mode == 0) k<0><<<cus, 512, 100 << 10>>>(src, win, reps, sink, shared);
          else k<1><<<cus, 512, 100 << 10>>>(src, win, reps, sink, shared);
          hipEventRecord(e1);
          hipEventSynchronize(e1);
        }
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double bytes = (double)win * reps;     // per CU
        printf("%s window %5zu KiB/CU %s: %7.1f GB/s per CU (%.1f B/clk at 2.4 GHz), %6.2f TB/s chip\n", mode ? "load->VGPR" : "LDS-DMA   ",
               win >> 10, shared ? "(one window for all CUs)" : "(own window)", bytes / ms * 1e-6, bytes / ms * 1e-6 / 2.4, bytes * cus / ms * 1e-9);
      }
  hipFuncSetAttribute((const void*)kg, hipFuncAttributeMaxDynamicSharedMemorySize, 100 << 10);
  for (int pitch : {128, 1536, 1536 + 128, 6144, 6144 + 128, 2048, 4096}) {
    const int ksteps = pitch >= 1536 ? 12 : 1;                 // 512 rows x 1536 B = 768 KiB window, one for all CUs (L2-resident)
    const int reps = 4096 / ksteps;
    for (int it = 0; it < 2; ++it) {
      hipEventRecord(e0);
      kg<<<cus, 512, 100 << 10>>>(src, pitch, ksteps, reps, sink);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
    }
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double bytes = 65536.0 * ksteps * reps;
    printf("LDS-DMA gather 8 rows x 128 B per instruction, row pitch %5d B: %7.1f GB/s per CU (%.1f B/clk at 2.4 GHz)\n", pitch, bytes / ms * 1e-6,
           bytes / ms * 1e-6 / 2.4);
  }
  return 0;
}
