// valu_rate.hip -- what one MI355X SIMD issues per cycle for the instruction kinds the filter kernels are
// made of: independent chains of v_fma_f32, v_pk_fma_f32, v_fma_f64, v_mul_f64 + v_add_f64, with 1..8 waves
// per SIMD.  Prints wave-instructions per second per SIMD and the implied cycles per wave64 instruction at
// the measured clock (s_memtime ticks / wall time).  Build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float v2f __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ void __launch_bounds__(256) k_rate(float *out, int iters, float seed) {
  // 16 independent accumulators per lane: no dependent-issue stalls
  float a[16];
  double d[16];
  v2f p[16];
#pragma unroll
  for (int k = 0; k < 16; k++) { a[k] = seed + k; d[k] = seed + k; p[k] = (v2f){seed + k, seed - k}; }
  const float m = 1.0000001f;
  const double md = 1.0000000001;
  const v2f mp = (v2f){m, m};
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int k = 0; k < 16; k++) {
      if (KIND == 0) a[k] = __builtin_fmaf(a[k], m, 0.5f);
      if (KIND == 1) p[k] = __builtin_elementwise_fma(p[k], mp, mp);
      if (KIND == 2) d[k] = __builtin_fma(d[k], md, 0.5);
      if (KIND == 3) { d[k] = d[k] * md; d[k] = d[k] + 0.5; }
    }
  }
  float s = 0;
#pragma unroll
  for (int k = 0; k < 16; k++) s += a[k] + (float)d[k] + p[k].x + p[k].y;
  if (s == 12345.678f) out[0] = s;
}

int main() {
  float *out;
  hipMalloc(&out, 4);
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  const char *names[4] = {"v_fma_f32", "v_pk_fma_f32", "v_fma_f64", "v_mul_f64+v_add_f64"};
  const int per_iter[4] = {16, 16, 16, 32};
  printf("{\"compute_units\": %d, \"clock_MHz\": %d, \"rows\": [\n", cus, prop.clockRate / 1000);
  bool first = true;
  for (int kind = 0; kind < 4; kind++) {
    for (int waves_per_simd : {1, 2, 4, 8}) {
      const int iters = 20000;
      const int blocks = cus * waves_per_simd;  // 256 threads = 4 waves = one per SIMD of a CU
      hipEvent_t e0, e1;
      hipEventCreate(&e0); hipEventCreate(&e1);
      auto launch = [&]() {
        if (kind == 0) hipLaunchKernelGGL(k_rate<0>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f);
        if (kind == 1) hipLaunchKernelGGL(k_rate<1>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f);
        if (kind == 2) hipLaunchKernelGGL(k_rate<2>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f);
        if (kind == 3) hipLaunchKernelGGL(k_rate<3>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f);
      };
      launch();
      hipDeviceSynchronize();
      hipEventRecord(e0);
      launch();
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms = 0;
      hipEventElapsedTime(&ms, e0, e1);
      const double insts_per_simd = (double)iters * per_iter[kind] * waves_per_simd;  // wave-instructions on one SIMD
      const double per_s = insts_per_simd / (ms * 1e-3);
      printf("%s {\"inst\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.4f, \"wave_insts_per_s_per_simd\": %.4g, "
             "\"cycles_per_inst_at_2.4GHz\": %.3f}",
             first ? "" : ",\n", names[kind], waves_per_simd, ms, per_s, 2.4e9 / per_s);
      first = false;
    }
  }
  printf("\n]}\n");
  return 0;
}
