// mx_probe.hip — dev probe: k-structure and block-scale semantics of v_mfma_scale_f32_32x32x64_f8f6f4 with bf8 operands.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define ONE 0x3C   // bf8 (e5m2) 1.0
// out[0..255]: probe 1 (half_a, reg_a) x (half_b, reg_b) -> D[0][0]
// out[256..271]: probe 2 byte_a x byte_b within (half 0, reg 0)
// out[272..]: probe 3 scales
__global__ void k(float* out) {
  const int lane = threadIdx.x, hh = lane >> 5;
  for (int pa = 0; pa < 16; ++pa)
    for (int pb = 0; pb < 16; ++pb) {
      i32x8 a = {0, 0, 0, 0, 0, 0, 0, 0}, b = {0, 0, 0, 0, 0, 0, 0, 0};
      if (hh == (pa >> 3)) a[pa & 7] = ONE * 0x01010101;
      if (hh == (pb >> 3)) b[pb & 7] = ONE * 0x01010101;
      f32x16 c = {0};
      c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 1, 1, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
      if (lane == 0) out[pa * 16 + pb] = c[0];
    }
  for (int ba = 0; ba < 4; ++ba)
    for (int bb = 0; bb < 4; ++bb) {
      i32x8 a = {0, 0, 0, 0, 0, 0, 0, 0}, b = {0, 0, 0, 0, 0, 0, 0, 0};
      if (hh == 0) { a[0] = ONE << (8 * ba); b[0] = ONE << (8 * bb); }
      f32x16 c = {0};
      c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 1, 1, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
      if (lane == 0) out[256 + ba * 4 + bb] = c[0];
    }
  // probe 3: all ones; scale of A = 2.0 (E8M0 128) in lanes of half sa, byte position sel; D[0][0] and D[row 5][col 7]
  for (int t = 0; t < 8; ++t) {
    i32x8 a, b;
    for (int r = 0; r < 8; ++r) { a[r] = ONE * 0x01010101; b[r] = ONE * 0x01010101; }
    int sa = 0x7f7f7f7f, sb = 0x7f7f7f7f;
    if (t == 1 && hh == 0) sa = 0x7f7f7f80;           // byte 0, half 0 lanes
    if (t == 2 && hh == 1) sa = 0x7f7f7f80;           // byte 0, half 1 lanes
    if (t == 3 && lane == 0) sa = 0x7f7f7f80;         // only lane 0 (row 0, half 0)
    if (t == 4 && hh == 0) sb = 0x7f7f7f80;           // B scale, half 0
    if (t == 5 && lane == 7) sb = 0x7f7f7f80;         // B: only lane 7 (col 7, half 0)
    if (t == 6 && hh == 0) sa = 0x7f7f807f;           // byte 1 with opsel 0: ignored?
    f32x16 c = {0};
    if (t == 7) { if (hh == 0) sa = 0x7f7f807f; c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 1, 1, 1, sa, 0, sb); }
    else c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 1, 1, 0, sa, 0, sb);
    if (lane == 0) { out[272 + t * 4 + 0] = c[0]; out[272 + t * 4 + 1] = c[1]; }
    if (lane == 7) { out[272 + t * 4 + 2] = c[0]; }
    if (lane == 32 + 7) { out[272 + t * 4 + 3] = c[0]; }
  }
}
int main() {
  float* o;
  (void)hipMalloc(&o, 512 * 4);
  (void)hipMemset(o, 0, 512 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o);
  float h[512];
  (void)hipMemcpy(h, o, sizeof h, hipMemcpyDeviceToHost);
  printf("probe 1: rows (half_a,reg_a), cols (half_b,reg_b): D[0][0]\n");
  for (int i = 0; i < 16; ++i) { for (int j = 0; j < 16; ++j) printf("%2.0f ", h[i * 16 + j]); printf("\n"); }
  printf("probe 2: byte_a x byte_b (half 0, reg 0)\n");
  for (int i = 0; i < 4; ++i) { for (int j = 0; j < 4; ++j) printf("%2.0f ", h[256 + i * 4 + j]); printf("\n"); }
  const char* nm[8] = {"no scale", "A scale x2 half0 lanes", "A scale x2 half1 lanes", "A scale x2 lane 0 only", "B scale x2 half0 lanes", "B scale x2 lane 7 only", "A byte1=x2 opsel0", "A byte1=x2 opsel1"};
  for (int t = 0; t < 8; ++t) printf("probe 3 %-26s: D[0][0]=%g D[1?][0]=%g  lane7 c0 (D[0][7])=%g  lane39 c0 (D[4][7])=%g\n", nm[t], h[272 + t * 4], h[272 + t * 4 + 1], h[272 + t * 4 + 2], h[272 + t * 4 + 3]);
  return 0;
}
