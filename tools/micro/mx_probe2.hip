// mx_probe2.hip — which lanes' E8M0 scale applies to which of an operand's k-values (v_mfma_scale_f32_32x32x64_f8f6f4, bf8).
// One launch per case; A = ones in the selected (lane-half mask, register mask) elements, B = all ones (or the reverse).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(int on_b, int half_mask, int reg_mask, int scale_half_mask, float* out) {
  const int lane = threadIdx.x, hh = lane >> 5;
  i32x8 x, ones;
  for (int r = 0; r < 8; ++r) { ones[r] = 0x3c3c3c3c; x[r] = (((half_mask >> hh) & 1) && ((reg_mask >> r) & 1)) ? 0x3c3c3c3c : 0; }
  const int s = ((scale_half_mask >> hh) & 1) ? 128 : 127;
  f32x16 c;
  for (int i = 0; i < 16; ++i) c[i] = 0.f;
  if (on_b) c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(ones, x, c, 1, 1, 0, 127, 0, s);
  else c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(x, ones, c, 1, 1, 0, s, 0, 127);
  if (lane == 0) out[0] = c[0];
  if (lane == 37) out[1] = c[5];
}
int main() {
  float* o; (void)hipMalloc(&o, 64);
  for (int on_b = 0; on_b < 2; ++on_b)
    for (int hm = 1; hm <= 3; ++hm)
      for (int rm : {0xff, 0x0f, 0xf0, 0x33})
        for (int sm = 0; sm <= 3; ++sm) {
          hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, on_b, hm, rm, sm, o);
          float h[2]; (void)hipMemcpy(h, o, sizeof h, hipMemcpyDeviceToHost);
          printf("%s data: lane halves %d%d regs %02x | scale x2 in lane halves %d%d -> D[0][0] = %g, D[elsewhere] = %g\n", on_b ? "B" : "A", hm & 1, hm >> 1, rm, sm & 1, sm >> 1, h[0], h[1]);
        }
  return 0;
}
