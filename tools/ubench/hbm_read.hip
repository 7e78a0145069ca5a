// What a read-only stream over a buffer far larger than the Infinity Cache reaches on this chip, by how it is issued:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 hbm_read.hip -o bin/hbm_read && ./bin/hbm_read [GB]
// (a) global_load_dwordx4 into registers, UNROLL loads in flight per lane, many workgroups per CU
// (b) LDS-DMA (global_load_lds_dwordx4) into a ring, one workgroup per CU, each workgroup streaming its own contiguous chunk —
//     the shape of cross_attn's encoder-token stream (marie_icr_amd/csrc/cross_attn.hip)
// Measurement aid, not shipped.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef float float4v __attribute__((ext_vector_type(4)));

template <int UNROLL>
__global__ __launch_bounds__(256) void read_regs(const float4v* __restrict__ p, size_t n16, float* out) {
  float4v acc = {0.f, 0.f, 0.f, 0.f};
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + (UNROLL - 1) * stride < n16; i += UNROLL * stride) {
    float4v v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) v[u] = __builtin_nontemporal_load(p + i + u * stride);
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) acc += v[u];
  }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[0] = acc[0];
}

// one workgroup = WAVES waves streams [chunk bytes] contiguous bytes through a ring of SLOTS x TILE bytes
template <int WAVES, int TILE, int SLOTS, int AUX>
__global__ __launch_bounds__(WAVES * 64) void read_dma(const char* __restrict__ p, size_t chunk, int chunks, float* out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int PER = TILE / 1024 / WAVES;      // DMA instructions per wave and tile
  for (int c = blockIdx.x; c < chunks; c += gridDim.x) {
    const char* src = p + (size_t)c * chunk;
    const int ntiles = (int)(chunk / TILE);
    auto issue = [&](int t, int slot) {
#pragma unroll
      for (int j = 0; j < PER; ++j) {
        const int i = wave + j * WAVES;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)t * TILE + i * 1024 + lane * 16),
                                         (__attribute__((address_space(3))) void*)(smem + slot * TILE + i * 1024), 16, 0, AUX);
      }
    };
    for (int i = 0; i < SLOTS - 1 && i < ntiles; ++i) issue(i, i);
    int slot = 0;
    for (int t = 0; t < ntiles; ++t) {
      // wait for tile t: at most (SLOTS - 2) tiles issued after it may still be outstanding
      if (t + SLOTS - 2 < ntiles) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((SLOTS - 2) * PER) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (t + SLOTS - 1 < ntiles) issue(t + SLOTS - 1, slot == 0 ? SLOTS - 1 : slot - 1);
      slot = slot == SLOTS - 1 ? 0 : slot + 1;
    }
    __builtin_amdgcn_s_barrier();
  }
  if (out && smem[threadIdx.x] == 123 && smem[4096 + threadIdx.x] == 77) out[1] = 1.f;
}

int main(int argc, char** argv) {
  const double gb = argc > 1 ? atof(argv[1]) : 2.3;
  const size_t chunk = 576 * 1536;                   // one crop's encoder tokens, rounded to whole 48 KiB tiles: 884 736 bytes
  const int chunks = (int)(gb * 1e9 / chunk);
  const size_t bytes = (size_t)chunks * chunk;
  char* buf;
  float* out;
  CK(hipMalloc((void**)&buf, bytes + (1 << 20)));
  CK(hipMalloc((void**)&out, 64));
  CK(hipMemset(buf, 1, bytes));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto time = [&](const char* name, auto launch) {
    for (int w = 0; w < 2; ++w) launch();
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    const int it = 10;
    for (int i = 0; i < it; ++i) launch();
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-64s %8.3f ms  %6.2f TB/s\n", name, ms / it, bytes / (ms / it) * 1e-9);
  };
  printf("%d chunks of %zu bytes = %.2f GB\n", chunks, chunk, bytes * 1e-9);
  time("registers, 4 loads in flight per lane, 2048 workgroups", [&] { hipLaunchKernelGGL(read_regs<4>, dim3(2048), dim3(256), 0, 0, (const float4v*)buf, bytes / 16, out); });
  time("registers, 8 loads in flight per lane, 2048 workgroups", [&] { hipLaunchKernelGGL(read_regs<8>, dim3(2048), dim3(256), 0, 0, (const float4v*)buf, bytes / 16, out); });
  time("registers, 8 loads in flight per lane, 4096 workgroups", [&] { hipLaunchKernelGGL(read_regs<8>, dim3(4096), dim3(256), 0, 0, (const float4v*)buf, bytes / 16, out); });
#define DMA(W, T, S, A, G, name)                                                                                          \
  do {                                                                                                                      \
    hipFuncSetAttribute((const void*)read_dma<W, T, S, A>, hipFuncAttributeMaxDynamicSharedMemorySize, T * S);             \
    time(name, [&] { hipLaunchKernelGGL((read_dma<W, T, S, A>), dim3(G), dim3(W * 64), T * S, 0, buf, chunk, chunks, out); }); \
  } while (0)
  DMA(6, 49152, 3, 0, chunks, "LDS-DMA 6 waves, 3 x 48 KiB ring, one workgroup per chunk");
  DMA(6, 49152, 3, 2, chunks, "LDS-DMA 6 waves, 3 x 48 KiB ring, one workgroup per chunk, nt");
  DMA(6, 49152, 3, 0, 256, "LDS-DMA 6 waves, 3 x 48 KiB ring, 256 persistent workgroups");
  DMA(4, 16384, 8, 0, chunks, "LDS-DMA 4 waves, 8 x 16 KiB ring, one workgroup per chunk");
  DMA(4, 16384, 8, 2, chunks, "LDS-DMA 4 waves, 8 x 16 KiB ring, one workgroup per chunk, nt");
  DMA(4, 16384, 4, 0, chunks, "LDS-DMA 4 waves, 4 x 16 KiB ring (2 workgroups per CU)");
  DMA(4, 16384, 2, 0, chunks, "LDS-DMA 4 waves, 2 x 16 KiB ring (4 workgroups per CU)");
  DMA(8, 16384, 8, 0, chunks, "LDS-DMA 8 waves, 8 x 16 KiB ring, one workgroup per chunk");
  DMA(2, 16384, 8, 0, chunks, "LDS-DMA 2 waves, 8 x 16 KiB ring, one workgroup per chunk");
  return 0;
}
