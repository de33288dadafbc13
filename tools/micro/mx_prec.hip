// mx_prec.hip — dev probe: accuracy of bf8 x bf8 dot products on v_mfma_scale_f32_32x32x64_f8f6f4 and on the non-scaled
// v_mfma_f32_32x32x16_bf8_bf8 against an exact (double) evaluation of the same bf8 operands, for operands with a wide
// dynamic range inside one K block.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <math.h>
#include <stdlib.h>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(const unsigned char* A, const unsigned char* B, float* Dmx, float* D16) {
  // A[32 rows][64 k], B[64 k][32 cols] bytes (bf8).  lane (r = lane&31, h = lane>>5): element (h, reg, byte) <-> k = 32h + 4reg + byte
  const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
  i32x8 a, b;
  for (int q = 0; q < 8; ++q) {
    unsigned wa = 0, wb = 0;
    for (int by = 0; by < 4; ++by) { const int kk = 32 * h + 4 * q + by; wa |= (unsigned)A[r * 64 + kk] << (8 * by); wb |= (unsigned)B[kk * 32 + r] << (8 * by); }
    a[q] = (int)wa; b[q] = (int)wb;
  }
  f32x16 c = {0};
  c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 1, 1, 0, 127, 0, 127);
  for (int i = 0; i < 16; ++i) Dmx[(8 * (i >> 2) + (i & 3) + 4 * h) * 32 + r] = c[i];
  // non-scaled K=16 form: 4 MFMAs, lane holds k = 16 step + 8h + j
  f32x16 d = {0};
  for (int st = 0; st < 4; ++st) {
    unsigned long long wa = 0, wb = 0;
    for (int j = 0; j < 8; ++j) { const int kk = 16 * st + 8 * h + j; wa |= (unsigned long long)A[r * 64 + kk] << (8 * j); wb |= (unsigned long long)B[kk * 32 + r] << (8 * j); }
    d = __builtin_amdgcn_mfma_f32_32x32x16_bf8_bf8((long)wa, (long)wb, d, 0, 0, 0);
  }
  for (int i = 0; i < 16; ++i) D16[(8 * (i >> 2) + (i & 3) + 4 * h) * 32 + r] = d[i];
}
static double bf8(unsigned char v) {
  const int s = v >> 7, e = (v >> 2) & 31, m = v & 3;
  double x = e == 0 ? ldexp(m / 4.0, -14) : ldexp(1.0 + m / 4.0, e - 15);
  return s ? -x : x;
}
int main() {
  unsigned char hA[32 * 64], hB[64 * 32];
  srand(3);
  for (int spread = 0; spread <= 24; spread += 8) {
    for (int i = 0; i < 32 * 64; ++i) { int e = 15 - rand() % (spread + 1); hA[i] = (unsigned char)(((rand() & 1) << 7) | (e << 2) | (rand() & 3)); }
    for (int i = 0; i < 64 * 32; ++i) { int e = 15 - rand() % 4; hB[i] = (unsigned char)((e << 2) | (rand() & 3)); }
    unsigned char *dA, *dB; float *d1, *d2;
    (void)hipMalloc(&dA, sizeof hA); (void)hipMalloc(&dB, sizeof hB); (void)hipMalloc(&d1, 4096); (void)hipMalloc(&d2, 4096);
    (void)hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); (void)hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, d1, d2);
    float h1[1024], h2[1024];
    (void)hipMemcpy(h1, d1, 4096, hipMemcpyDeviceToHost); (void)hipMemcpy(h2, d2, 4096, hipMemcpyDeviceToHost);
    double e1 = 0, e2 = 0, nrm = 0;
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
      double ref = 0;
      for (int kk = 0; kk < 64; ++kk) ref += bf8(hA[i * 64 + kk]) * bf8(hB[kk * 32 + j]);
      e1 += (h1[i * 32 + j] - ref) * (h1[i * 32 + j] - ref); e2 += (h2[i * 32 + j] - ref) * (h2[i * 32 + j] - ref); nrm += ref * ref;
    }
    printf("A exponent spread %2d binades: rel L2 error  MX 32x32x64 %.3e   32x32x16_bf8_bf8 %.3e\n", spread, sqrt(e1 / nrm), sqrt(e2 / nrm));
  }
  return 0;
}
