// f64_latency.hip -- what one wave alone on a SIMD pays per float64 vector instruction (MI355X): a DEPENDENT chain of
// v_add_f64 / v_mul_f64 / v_fma_f64 against K independent chains interleaved, measured with s_memtime around
// 4096 instructions.  The projection's Newton step is one wave per SIMD running such chains.
// build: hipcc --offload-arch=gfx950 -O2 -o f64_latency f64_latency.hip ; run: ./f64_latency
#include <hip/hip_runtime.h>
#include <cstdio>

template <int OP, int K>
__global__ void chain(double *out, long long *cycles, double a, double b) {
  double x[K];
  for (int k = 0; k < K; k++) x[k] = a + k;
  const long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int it = 0; it < 4096 / K / 8; it++) {
#pragma unroll
    for (int u = 0; u < 8; u++) {
#pragma unroll
      for (int k = 0; k < K; k++) {
        if (OP == 0) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x[k]) : "v"(b));
        if (OP == 1) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x[k]) : "v"(b));
        if (OP == 2) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(x[k]) : "v"(b));
        if (OP == 3) asm volatile("v_add_f32 %0, %0, %1" : "+v"(((float *)&x[k])[0]) : "v"((float)b));
      }
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  double s = 0;
  for (int k = 0; k < K; k++) s += x[k];
  out[threadIdx.x] = s;
  if (threadIdx.x == 0) *cycles = t1 - t0;
}

template <int OP, int K>
void run(const char *name) {
  double *out;
  long long *cyc, h = 0;
  hipMalloc(&out, 64 * sizeof(double));
  hipMalloc(&cyc, sizeof(long long));
  for (int rep = 0; rep < 3; rep++) hipLaunchKernelGGL((chain<OP, K>), dim3(1), dim3(64), 0, 0, out, cyc, 1.0, 1.0000001);
  hipMemcpy(&h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  const int n = (4096 / K / 8) * 8 * K;
  printf("%-10s %d independent chain(s): %6.2f counter ticks per instruction (%d instructions, %lld ticks)\n", name, K, (double)h / n, n, h);
  hipFree(out); hipFree(cyc);
}

int main() {
  int clk = 0;
  hipDeviceGetAttribute(&clk, hipDeviceAttributeWallClockRate, 0);
  printf("wall clock rate (kHz) %d -- s_memtime / readcyclecounter ticks at this rate; shader clock 2.4 GHz\n", clk);
  run<0, 1>("v_add_f64"); run<0, 2>("v_add_f64"); run<0, 4>("v_add_f64"); run<0, 8>("v_add_f64");
  run<1, 1>("v_mul_f64"); run<1, 4>("v_mul_f64");
  run<2, 1>("v_fma_f64"); run<2, 2>("v_fma_f64"); run<2, 4>("v_fma_f64"); run<2, 8>("v_fma_f64");
  run<3, 1>("v_add_f32"); run<3, 4>("v_add_f32");
  return 0;
}
