// tr8_probe.hip — dev probe: element order of ds_read_b64_tr_b8 and the scale semantics / overflow behaviour of
// v_cvt_scalef32_pk_bf8_f16 and v_cvt_scalef32_pk_f16_bf8 (the 8-bit stash of the f16 training mode).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef int i32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef short s2 __attribute__((ext_vector_type(2)));
__global__ void k(unsigned char* out, float* cv) {
  __shared__ __attribute__((aligned(16))) unsigned char img[64 * 16 * 2];      // 2 chunk columns x [64 rows][16 B]; value = (row << 4 | byte) (+128 for chunk 1)
  const int lane = threadIdx.x;
  for (int i = lane; i < 2 * 64 * 16; i += 64) {
    const int c = i / 1024, r = (i % 1024) / 16, b = i % 16;
    img[i] = (unsigned char)(((r & 7) << 4) | b) ^ (c ? 0x80 : 0);
  }
  __syncthreads();
  const int g4 = lane >> 4, li = lane & 15;
  // hypothesis: lane 2q+p of a 16-lane group supplies row q (0..7), bytes 8p..8p+7; lane i receives byte-column i of the 8 rows
  const int q = li >> 1, p = li & 1;
  const int row = 8 * (g4 >> 1) + q;
  const unsigned char* a = img + (g4 & 1) * 1024 + row * 16 + 8 * p;
  const i32x2 x = __builtin_amdgcn_ds_read_tr8_b64_v2i32((__attribute__((address_space(3))) i32x2*)a);
  for (int e = 0; e < 8; ++e) out[lane * 8 + e] = (unsigned char)((e < 4 ? (unsigned)x[0] >> (8 * e) : (unsigned)x[1] >> (8 * (e - 4))) & 0xff);
  if (lane == 0) {
    const float vals[6] = {1.0f, 0.001f, 3.0e-5f, 60000.f, 65504.f, 1.1f};
    for (int i = 0; i < 6; ++i) {
      h2 v = {(_Float16)vals[i], (_Float16)-vals[i]};
      s2 r = {0, 0};
      r = __builtin_amdgcn_cvt_scalef32_pk_bf8_f16(r, v, 1.0f, false);
      h2 b = __builtin_amdgcn_cvt_scalef32_pk_f16_bf8(__builtin_bit_cast(unsigned, r), 1.0f, false);
      cv[i * 4 + 0] = (float)b[0]; cv[i * 4 + 1] = (float)b[1];
      s2 r4 = {0, 0};
      r4 = __builtin_amdgcn_cvt_scalef32_pk_bf8_f16(r4, v, 4.0f, false);
      h2 b4 = __builtin_amdgcn_cvt_scalef32_pk_f16_bf8(__builtin_bit_cast(unsigned, r4), 1.0f, false);
      cv[i * 4 + 2] = (float)b4[0];
      h2 b5 = __builtin_amdgcn_cvt_scalef32_pk_f16_bf8(__builtin_bit_cast(unsigned, r), 4.0f, false);
      cv[i * 4 + 3] = (float)b5[0];
    }
  }
}
int main() {
  unsigned char* o; float* cv;
  (void)hipMalloc(&o, 512); (void)hipMalloc(&cv, 24 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o, cv);
  unsigned char h[512]; float hc[24];
  (void)hipMemcpy(h, o, sizeof h, hipMemcpyDeviceToHost); (void)hipMemcpy(hc, cv, sizeof hc, hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) {
    printf("lane %2d:", l);
    for (int j = 0; j < 8; ++j) printf(" (c%d r%d b%2d)", h[l * 8 + j] >> 7, (h[l * 8 + j] >> 4) & 7, h[l * 8 + j] & 15);
    printf("\n");
  }
  const float vals[6] = {1.0f, 0.001f, 3.0e-5f, 60000.f, 65504.f, 1.1f};
  for (int i = 0; i < 6; ++i)
    printf("v=%g: bf8 round trip (scale 1) = %g / %g ; to-bf8 scale 4 then back scale 1 = %g ; to-bf8 scale 1 then back scale 4 = %g\n",
           vals[i], hc[i * 4], hc[i * 4 + 1], hc[i * 4 + 2], hc[i * 4 + 3]);
  return 0;
}
