// fp6_probe2.hip - second probe for the 6-bit H stash:
//   D. ds_read_b96_tr_b6 on rows left in LDS by a 12-byte-per-lane LDS-DMA (global_load_lds_dwordx3): where does the DMA place lane l's
//      12 bytes (base + 12 l or base + 16 l), and which row addresses does the transposed read want (-DSTRIDE=12 / 16)?
//   E. issue cost of v_cvt_scalef32_pk32_bf6_f16 (32 values) against the 16 v_cvt_scalef32_pk_bf8_f16 it would replace (s_memtime around
//      256 back-to-back conversions of independent registers, one wave).
// hipcc --offload-arch=gfx950 -O3 -o tools/micro/fp6_probe2 tools/micro/fp6_probe2.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define GPTR(p) ((const __attribute__((address_space(1))) void*)(p))
#ifndef STRIDE
#define STRIDE 16
#endif
#define LPTR(p) ((__attribute__((address_space(3))) void*)(p))
typedef _Float16 f16x32 __attribute__((ext_vector_type(32)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x6 __attribute__((ext_vector_type(6)));
typedef int i32x3 __attribute__((ext_vector_type(3)));

__global__ void k(const unsigned char* rows, unsigned* outD, unsigned long long* outE) {
  __shared__ __attribute__((aligned(16))) unsigned char img[64 * 16];
  const int lane = threadIdx.x;
  // ---- D: 64 rows of 12 bytes, row r field c = (5 r + c) & 63
  for (int i = 0; i < 4; ++i) ((unsigned*)img)[lane * 4 + i] = 0xdeadbeefu;
  __syncthreads();
  __builtin_amdgcn_global_load_lds(GPTR(rows + lane * 12), LPTR(img), 12, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  // raw image, to see where the DMA put the bytes
  for (int i = 0; i < 4; ++i) outD[lane * 4 + i] = ((unsigned*)img)[lane * 4 + i];
  // 16-lane group g reads rows 16 g .. 16 g + 15 transposed
  i32x3 t;
  const uint32_t addr = (uint32_t)(uintptr_t)LPTR(img) + STRIDE * lane;
  asm volatile("ds_read_b96_tr_b6 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(t) : "v"(addr) : "memory");
  for (int i = 0; i < 3; ++i) outD[256 + lane * 3 + i] = (unsigned)t[i];
  // ---- E
  f16x32 v[4];
  for (int j = 0; j < 4; ++j)
    for (int i = 0; i < 32; ++i) v[j][i] = (_Float16)(0.25f * ((lane + i + j) & 15));
  u32x6 acc = {0, 0, 0, 0, 0, 0};
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  for (int it = 0; it < 64; ++it)
    for (int j = 0; j < 4; ++j) {
      u32x6 r = __builtin_amdgcn_cvt_scalef32_pk32_bf6_f16(v[j], 1.0f);
      asm volatile("" : "+v"(r));
      acc ^= r;
    }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  unsigned acc8 = 0;
  for (int it = 0; it < 64; ++it)
    for (int j = 0; j < 4; ++j)
      for (int q = 0; q < 8; ++q) {
        s16x2 w = {0, 0};
        const f16x2_t p0 = {v[j][4 * q], v[j][4 * q + 1]}, p1 = {v[j][4 * q + 2], v[j][4 * q + 3]};
        w = __builtin_amdgcn_cvt_scalef32_pk_bf8_f16(w, p0, 1.0f, false);
        w = __builtin_amdgcn_cvt_scalef32_pk_bf8_f16(w, p1, 1.0f, true);
        unsigned u = __builtin_bit_cast(unsigned, w);
        asm volatile("" : "+v"(u));
        acc8 ^= u;
      }
  unsigned long long t2 = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if (lane == 0) { outE[0] = t1 - t0; outE[1] = t2 - t1; outE[2] = acc[0] ^ acc[5] ^ acc8; }
}

static int field(const unsigned* w, int f) {
  const int bit = 6 * f, d = bit >> 5, s = bit & 31;
  unsigned long long x = w[d];
  if (s + 6 > 32) x |= (unsigned long long)w[d + 1] << 32;
  return (int)((x >> s) & 0x3f);
}

int main() {
  unsigned char h[64 * 12] = {0};
  for (int r = 0; r < 64; ++r)
    for (int c = 0; c < 16; ++c) {
      const unsigned val = (5 * r + c) & 63;
      const int bit = 96 * r + 6 * c;
      for (int b = 0; b < 6; ++b)
        if (val >> b & 1) h[(bit + b) >> 3] |= 1u << ((bit + b) & 7);
    }
  unsigned char* d; unsigned* dD; unsigned long long* dE;
  (void)hipMalloc(&d, sizeof h); (void)hipMalloc(&dD, 448 * 4); (void)hipMalloc(&dE, 24);
  (void)hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, dD, dE);
  unsigned hD[448]; unsigned long long hE[3];
  (void)hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost); (void)hipMemcpy(hE, dE, sizeof hE, hipMemcpyDeviceToHost);
  int bad_dma = 0, bad_tr = 0;
  int at16 = 0, pad_kept = 0;      // lane l's 12 bytes at base + 16 l, the 4 bytes behind them untouched?
  for (int l = 0; l < 64; ++l) {
    for (int i = 0; i < 3; ++i) at16 += hD[4 * l + i] == ((unsigned*)h)[3 * l + i];
    pad_kept += hD[4 * l + 3] == 0xdeadbeefu;
  }
  for (int i = 0; i < 192; ++i) bad_dma += hD[i] != ((unsigned*)h)[i];
  printf("D. global_load_lds_dwordx3: %d of 192 dwords found at base + 16 lane + 4 i; pad dword untouched in %d of 64 lanes\n", at16, pad_kept);
  for (int l = 0; l < 64; ++l)
    for (int f = 0; f < 16; ++f) {
      const int want = (5 * (16 * (l >> 4) + f) + (l & 15)) & 63;      // lane l = column l & 15 of rows 16 (l >> 4) + f
      bad_tr += field(hD + 256 + 3 * l, f) != want;
    }
  printf("   image == source bytes at a 12-byte stride: %s;  ds_read_b96_tr_b6 with row addresses base + %d lane: %s (%d wrong fields)\n",
         bad_dma ? "no" : "yes", STRIDE, bad_tr ? "WRONG" : "transposes correctly", bad_tr);
  if (bad_tr) {
    for (int l = 0; l < 64; l += 5) { printf("   lane %2d:", l); for (int f = 0; f < 16; ++f) printf(" %2d", field(hD + 256 + 3 * l, f)); printf("\n"); }
  }
  printf("E. 256 x cvt_scalef32_pk32_bf6_f16: %llu clocks (%.1f per 32 values);  256 x 16 cvt_scalef32_pk_bf8_f16: %llu clocks (%.1f per 32 values)  [s_memtime, 100 MHz]\n",
         hE[0], hE[0] / 256.0, hE[1], hE[1] / 256.0);
  return 0;
}
