// Micro-benchmark (dev tool): write bandwidth of the stash store pattern of k_chain_bf16<bwd> in isolation, with and
// without the weight LDS-DMA stream beside it, on all or a quarter of the CUs.
// hipcc --offload-arch=gfx950 -O3 -o /tmp/store_bench tools/micro/store_bench.hip && /tmp/store_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
#define GPTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define LPTR(p) ((__attribute__((address_space(3))) void*)(p))

// per (tile, array): 16 stores of 1 KiB per wave (2 per 32-row output tile) and, with DMA, 2 KiB of LDS-DMA per wave and
// output tile from a 2 MiB L2-resident weight image (16 KiB per workgroup and output tile, as the chain kernel).
template <int STORES, int DMA>
__global__ void __launch_bounds__(512) k_mix(char* base, const char* wts, int64_t rows, int tiles, int arrays) {
  extern __shared__ char lds[];
  const int tid = threadIdx.x, lane = tid & 63, col = lane & 31, hh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  u32x4 v = {(unsigned)tid, 1u, 2u, 3u};
  uint32_t wpos = 0;
  for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const uint32_t m = (uint32_t)tile * 256 + wave * 32 + col;
    for (int l = 0; l < arrays; ++l) {
      char* lb = base + (size_t)l * rows * 512;
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        if (DMA) {
          asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "n"(STORES ? 2 : 0) : "memory");
          const char* src = wts + wpos;
          wpos = (wpos + 16384u) & 0x1fffffu;
          char* dst = lds + (t & 1) * 16384;
#pragma unroll
          for (int p = 0; p < 2; ++p)
            __builtin_amdgcn_global_load_lds(GPTR(src + p * 8192 + wave * 1024 + lane * 16), LPTR(dst + p * 8192 + wave * 1024), 16, 0, 0);
        }
        if (STORES) {
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            const size_t off = ((((size_t)(m >> 5) * 32 + (4 * t + 2 * s + hh)) << 5) + (m & 31)) << 4;
            __builtin_nontemporal_store(v, (u32x4*)(lb + off));
            v[1] += 1;
          }
        }
      }
    }
  }
}

int main() {
  const int tiles = 10112, arrays = 18;
  const int64_t rows = (int64_t)tiles * 256;
  const size_t bytes = (size_t)arrays * rows * 512;
  char *buf, *wts;
  CK(hipMalloc(&buf, bytes));
  CK(hipMalloc(&wts, 4 << 20));
  CK(hipMemset(wts, 1, 4 << 20));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto run = [&](const char* name, auto kern, int grid, int ntiles) -> int {
    for (int it = 0; it < 3; ++it) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(kern, dim3(grid), dim3(512), 32768, 0, buf, wts, rows, ntiles, arrays);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      const double sb = (double)arrays * ntiles * 256 * 512;
      if (it == 2) printf("%-22s grid %3d tiles %5d: %7.2f ms  stores %6.2f TB/s (%5.1f GB/s/CU)  dma %6.2f TB/s\n", name, grid, ntiles, ms,
                          sb / ms * 1e-9, sb / ms * 1e-6 / grid, (double)arrays * ntiles * 8 * 16384 / ms * 1e-9);
    }
    return 0;
  };
  for (int grid : {256, 64, 16}) {
    const int nt = tiles * grid / 256;
    if (run("stores only", k_mix<1, 0>, grid, nt)) return 1;
    if (run("dma only", k_mix<0, 1>, grid, nt)) return 1;
    if (run("stores + dma", k_mix<1, 1>, grid, nt)) return 1;
  }
  return 0;
}
