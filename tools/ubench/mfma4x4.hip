// Which lane / register holds D_b[i][j] of v_mfma_f32_4x4x1_16b_f32 (16 blocks of 4x4 += 4x1 * 1x4)?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(float* out) {
  const int l = threadIdx.x;
  f32x4 acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_f32_4x4x1f32(1.0f + l, 100.0f + l, acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[l * 4 + r] = acc[r];
}
int main() {
  float* d; hipMalloc(&d, 256 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  float h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int ok_a = 1, ok_b = 1;
  for (int l = 0; l < 64; ++l)
    for (int r = 0; r < 4; ++r) {
      const int b = l / 4;
      // hypothesis A: lane (b, j = l % 4), reg r = row i      hypothesis B: lane (b, i = l % 4), reg r = column j
      const float va = (1.0f + 4 * b + r) * (100.0f + 4 * b + l % 4), vb = (1.0f + 4 * b + l % 4) * (100.0f + 4 * b + r);
      if (h[l * 4 + r] != va) ok_a = 0;
      if (h[l * 4 + r] != vb) ok_b = 0;
    }
  printf("D layout: lane=(block, column j), reg=row i: %d ; lane=(block,row i), reg=column j: %d\n", ok_a, ok_b);
  printf("lane 5: %g %g %g %g\n", h[20], h[21], h[22], h[23]);
  return 0;
}
