// fp6_probe.hip - dev probe for a 6-bit (bf6 = e3m2) H-stash contracted on v_mfma_scale_f32_32x32x64_f8f6f4 against the bf8 dZ' stash
// (the "stash diet" the round-2 review asked for).  Three questions, answered by the hardware:
//   A. how v_cvt_scalef32_pk32_bf6_f16 packs its 32 results into 6 dwords (field order) - and that the matrix instruction reads the same packing;
//   B. with A = bf8 (cbsz 1) and B = bf6 (blgp 3): which B field (lane half, field index) meets which A element (lane half, register, byte),
//      and which fields an E8M0 scale supplied by the lanes of half 0 / half 1 applies to (the K-blocks);
//   C. what ds_read_b96_tr_b6 moves where: 16 lanes x 16 six-bit elements.
// hipcc --offload-arch=gfx950 -O3 -o /tmp/fp6_probe tools/micro/fp6_probe.hip && /tmp/fp6_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef _Float16 f16x32 __attribute__((ext_vector_type(32)));
typedef unsigned int u32x6 __attribute__((ext_vector_type(6)));
typedef int i32x3 __attribute__((ext_vector_type(3)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define ONE8 0x3C   // bf8 (e5m2) 1.0

// value of the non-negative e3m2 code c (0..31): sign 0, exponent c >> 2 (bias 3), mantissa c & 3
__host__ __device__ inline float e3m2(int c) {
  const int e = c >> 2, m = c & 3;
  return e == 0 ? m * 0.0625f : (1.f + 0.25f * m) * (float)(1 << e) / 8.f;
}

__global__ void k(unsigned* outA, float* outB, float* outS, unsigned* outC, float* outR) {
  __shared__ __attribute__((aligned(16))) unsigned char img[16 * 16];      // part C
  const int lane = threadIdx.x, hh = lane >> 5;
  // ---- A: element i = the value whose code is i -> dump the 6 dwords (lane 0)
  {
    f16x32 v;
    for (int i = 0; i < 32; ++i) v[i] = (_Float16)e3m2(i);
    const u32x6 r = __builtin_amdgcn_cvt_scalef32_pk32_bf6_f16(v, 1.0f);
    const f16x32 b = __builtin_amdgcn_cvt_scalef32_pk32_f16_bf6(r, 1.0f);
    if (lane == 0) {
      for (int i = 0; i < 6; ++i) outA[i] = r[i];
      for (int i = 0; i < 32; ++i) outR[i] = (float)b[i];
    }
  }
  // ---- B: A = bf8 1.0 in ONE element (half ha, k-position ka = 4 reg + byte), B = bf6 1.0 in ONE converter element (half hb, element fb):
  //         outB[(32 ha + ka) * 64 + 32 hb + fb] = D[0][0]
  for (int ea = 0; ea < 64; ++ea)
    for (int eb = 0; eb < 64; ++eb) {
      i32x8 a = {0, 0, 0, 0, 0, 0, 0, 0};
      if (hh == (ea >> 5)) a[(ea & 31) >> 2] = ONE8 << (8 * (ea & 3));
      f16x32 v;
      for (int i = 0; i < 32; ++i) v[i] = (_Float16)((hh == (eb >> 5) && i == (eb & 31)) ? 1.f : 0.f);
      const u32x6 r = __builtin_amdgcn_cvt_scalef32_pk32_bf6_f16(v, 1.0f);
      const i32x8 b = {(int)r[0], (int)r[1], (int)r[2], (int)r[3], (int)r[4], (int)r[5], 0, 0};
      f32x16 c = {0};
      c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 1, 3, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
      asm volatile("" :: "v"(a), "v"(b));      // keep the operands alive: hipcc otherwise lets the 16-register result overlap the A operand
      if (lane == 0) outB[ea * 64 + eb] = c[0];
    }
  // ---- B scales: everything 1.0; B's scale = 2 (E8M0 128) supplied by the lanes of half sb only; then A element (ha, ka) alone picks out which
  //      k-positions got doubled: outS[sb * 64 + 32 ha + ka] = D[0][0] with B all ones
  for (int sb = 0; sb < 2; ++sb)
    for (int ea = 0; ea < 64; ++ea) {
      i32x8 a = {0, 0, 0, 0, 0, 0, 0, 0};
      if (hh == (ea >> 5)) a[(ea & 31) >> 2] = ONE8 << (8 * (ea & 3));
      f16x32 v;
      for (int i = 0; i < 32; ++i) v[i] = (_Float16)1.f;
      const u32x6 r = __builtin_amdgcn_cvt_scalef32_pk32_bf6_f16(v, 1.0f);
      const i32x8 b = {(int)r[0], (int)r[1], (int)r[2], (int)r[3], (int)r[4], (int)r[5], 0, 0};
      const int scb = hh == sb ? 0x7f7f7f80 : 0x7f7f7f7f;
      f32x16 c = {0};
      c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 1, 3, 0, 0x7f7f7f7f, 0, scb);
      asm volatile("" :: "v"(a), "v"(b));
      if (lane == 0) outS[sb * 64 + ea] = c[0];
    }
  // ---- C: a 16 x 16 matrix of 6-bit elements in LDS, row r = 12 bytes at img + 16 r (element c of row r = code (r << 2 | (c & 3)) ^ ((c >> 2) << 4)...)
  //         simpler: two passes, element = r (4 bits) then element = c (4 bits); lane l of the first 16-lane group supplies the address of row l
  for (int pass = 0; pass < 2; ++pass) {
    __syncthreads();
    if (lane < 16) {
      unsigned long long lo = 0, hi = 0;      // 96 bits: 16 fields of 6 bits, field c at bit 6 c
      for (int c = 0; c < 16; ++c) {
        const unsigned long long val = (unsigned long long)(pass == 0 ? lane : c) & 0x3f;
        const int bit = 6 * c;
        if (bit < 64) { lo |= val << bit; if (bit + 6 > 64) hi |= val >> (64 - bit); }
        else hi |= val << (bit - 64);
      }
      *(unsigned long long*)(img + 16 * lane) = lo;
      *(unsigned*)(img + 16 * lane + 8) = (unsigned)hi;
    }
    __syncthreads();
    const i32x3 t = __builtin_amdgcn_ds_read_tr6_b96_v3i32((__attribute__((address_space(3))) i32x3*)(img + 16 * (lane & 15)));
    if (lane < 16)
      for (int i = 0; i < 3; ++i) outC[(pass * 16 + lane) * 3 + i] = (unsigned)t[i];
  }
}

static int field(const unsigned* w, int f) {      // 6-bit field f of a little-endian bit string
  const int bit = 6 * f, d = bit >> 5, s = bit & 31;
  unsigned long long x = w[d];
  if (s + 6 > 32) x |= (unsigned long long)w[d + 1] << 32;
  return (int)((x >> s) & 0x3f);
}

int main() {
  unsigned *dA, *dC; float *dB, *dS, *dR;
  (void)hipMalloc(&dA, 6 * 4); (void)hipMalloc(&dB, 64 * 64 * 4); (void)hipMalloc(&dS, 128 * 4); (void)hipMalloc(&dC, 2 * 16 * 3 * 4); (void)hipMalloc(&dR, 32 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dS, dC, dR);
  unsigned hA[6], hC[96]; static float hB[4096]; float hS[128], hR[32];
  (void)hipMemcpy(hA, dA, sizeof hA, hipMemcpyDeviceToHost); (void)hipMemcpy(hB, dB, sizeof hB, hipMemcpyDeviceToHost);
  (void)hipMemcpy(hS, dS, sizeof hS, hipMemcpyDeviceToHost); (void)hipMemcpy(hC, dC, sizeof hC, hipMemcpyDeviceToHost);
  (void)hipMemcpy(hR, dR, sizeof hR, hipMemcpyDeviceToHost);
  printf("A. cvt_scalef32_pk32_bf6_f16 of elements with codes 0..31: dwords %08x %08x %08x %08x %08x %08x\n   fields (little-endian, 6 bits each):", hA[0], hA[1], hA[2], hA[3], hA[4], hA[5]);
  for (int f = 0; f < 32; ++f) printf(" %d", field(hA, f));
  printf("\n   round trip through pk32_f16_bf6:");
  for (int i = 0; i < 32; ++i) printf(" %g", hR[i]);
  printf("\nB. A element (half, k) x B converter element (half, e) -> D[0][0]; listing the non-zero pairs:\n");
  for (int ea = 0; ea < 64; ++ea) {
    printf("   A(h%d,k%2d):", ea >> 5, ea & 31);
    for (int eb = 0; eb < 64; ++eb)
      if (hB[ea * 64 + eb] != 0.f) printf(" B(h%d,e%2d)=%g", eb >> 5, eb & 31, hB[ea * 64 + eb]);
    printf("\n");
  }
  for (int sb = 0; sb < 2; ++sb) {
    printf("   B scale x2 supplied by lane half %d: D[0][0] per A element (h0 k0..31, h1 k0..31):", sb);
    for (int ea = 0; ea < 64; ++ea) printf(" %g", hS[sb * 64 + ea]);
    printf("\n");
  }
  for (int pass = 0; pass < 2; ++pass) {
    printf("C. ds_read_b96_tr_b6, LDS element = %s: lane l receives fields", pass == 0 ? "its row" : "its column");
    for (int l = 0; l < 16; ++l) {
      printf("\n   lane %2d:", l);
      for (int f = 0; f < 16; ++f) printf(" %2d", field(hC + (pass * 16 + l) * 3, f));
    }
    printf("\n");
  }
  return 0;
}
