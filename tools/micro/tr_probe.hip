// tr_probe.hip — dev probe: (1) which stage row each element of a ds_read_b64_tr_b16 fragment comes from, with the
// address pattern of k_wgrad_bf16; (2) whether v_mfma_f32_32x32x16_f16 keeps f16 subnormal operands.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(unsigned short* out, float* mf) {
  __shared__ __attribute__((aligned(16))) unsigned short img[64 * 32];      // [row][col] 64 rows x 32 cols, value = row*256 + col
  const int lane = threadIdx.x;
  for (int i = lane; i < 64 * 32; i += 64) img[i] = (unsigned short)(((i / 32) << 8) | (i % 32));
  __syncthreads();
  const int g4 = lane >> 4, li = lane & 15, tq = li >> 2, tp = li & 3;
  for (int rd = 0; rd < 2; ++rd) {
    const int row = 8 * (g4 >> 1) + 4 * rd + tq;
    const int c = 16 * (g4 & 1) + 4 * tp;
    const s16x4 x = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(img + row * 32 + c));
    for (int e = 0; e < 4; ++e) out[lane * 8 + rd * 4 + e] = (unsigned short)x[e];
  }
  // subnormal test: A[i][k] = 2^-20 (f16 subnormal) for k = 0, B[k][j] = 1024: D = 2^-10 if subnormals are kept
  h8 a = {0, 0, 0, 0, 0, 0, 0, 0}, b = {0, 0, 0, 0, 0, 0, 0, 0};
  if (lane < 32) { a[0] = (_Float16)9.5367431640625e-07f; b[0] = (_Float16)1024.f; }
  f32x16 d = {0};
  d = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, d, 0, 0, 0);
  if (lane == 0) { mf[0] = d[0]; mf[1] = (float)a[0]; }
}
int main() {
  unsigned short* o; float* mf;
  hipMalloc(&o, 64 * 8 * 2); hipMalloc(&mf, 8);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o, mf);
  unsigned short h[512]; float hm[2];
  hipMemcpy(h, o, sizeof h, hipMemcpyDeviceToHost); hipMemcpy(hm, mf, sizeof hm, hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; l += 1) {
    printf("lane %2d:", l);
    for (int j = 0; j < 8; ++j) printf(" (r%2d,c%2d)", h[l * 8 + j] >> 8, h[l * 8 + j] & 255);
    printf("\n");
  }
  printf("mfma f16 subnormal: a=%g  d=%g (kept if 0.000976562)\n", hm[1], hm[0]);
  return 0;
}
