// Does v_mfma_f32_32x32x16_f16 keep binary16 subnormal inputs?  (k_nearest_mfma's error bound wants to know.)
// hipcc --offload-arch=gfx950 -O2 -o mfma_f16_denorm tools/micro/mfma_f16_denorm.hip && ./mfma_f16_denorm
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
__global__ void k(float *out, float a_val, float b_val) {
  const int l = threadIdx.x;
  h8 a, b;
  for (int j = 0; j < 8; j++) { a[j] = (_Float16)0.0f; b[j] = (_Float16)0.0f; }
  if (l < 32) { a[0] = (_Float16)a_val; b[0] = (_Float16)b_val; }  // k = 0 of every row / column
  f16v z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  const f16v d = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, z, 0, 0, 0);
  if (l == 0) out[0] = d[0];
}
int main() {
  float *d;
  hipMalloc(&d, 4);
  const float cases[][2] = {{0x1p-20f, 1.0f}, {1.0f, 0x1p-24f}, {0x1p-15f, 0x1p-15f}, {0x1p-14f, 1.0f}, {0x1p-20f, 0x1p10f}};
  for (auto &c : cases) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, c[0], c[1]);
    float h = -1;
    hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
    printf("a = %a  b = %a  ->  d = %a  (exact product %a)\n", c[0], c[1], h, c[0] * c[1]);
  }
  return 0;
}
