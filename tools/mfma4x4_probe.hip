// Operand / result layout of v_mfma_f32_4x4x1_16b_f32 (16 blocks of a 4 x 4 outer product, one k step) on gfx950:
// prints which (block, i, j) each lane's result registers hold, from inputs that encode their own index.
//   hipcc -O3 --offload-arch=gfx950 -o tools/bin/mfma4x4_probe tools/mfma4x4_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(float *out) {
  const int l = threadIdx.x;
  const float a = __builtin_ldexpf(1.0f, l);      // A value of lane l: 2^l
  const float b = 1.0f + l / 64.0f;               // B value of lane l: 1 + l / 64  (product = 2^la * (1 + lb / 64))
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
  for (int i = 0; i < 4; ++i) out[l * 4 + i] = c[i];
}
int main() {
  float *d, h[256];
  hipMalloc(&d, sizeof(h));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; l += 1) {
    if (l >= 8 && l < 60) continue;
    printf("lane %2d:", l);
    for (int i = 0; i < 4; ++i) {
      const double v = h[l * 4 + i];
      int e = 0;
      const double m = frexp(v, &e);        // v = m * 2^e, m in [0.5, 1)
      const int la = e - 1, lbb = (int)((m * 2.0 - 1.0) * 64.0 + 0.5);
      printf("  reg %d = A(lane %2d) * B(lane %2d)", i, la, lbb);
    }
    printf("\n");
  }
  return 0;
}
