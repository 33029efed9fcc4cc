// nn_loop.hip -- what the nearest-neighbour screen's loop can reach on one SIMD (MI355X): per tile of 32 nodes a wave
// issues eight v_mfma_f32_32x32x16_f16 (four sets of 32 queries x two binary16 terms) and folds the sixteen results per
// lane and set with eight v_min3_f32.  Variants: matrix instructions alone; folds alone; both, the folds of the tile
// before between the matrix instructions of this one (mjpl_nearest.h); with one and two waves per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 -o nn_loop nn_loop.hip ; run: ./nn_loop
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int MODE>  // 0: matrix only, 1: folds only, 2: interleaved, 3: matrix then folds (round 4's order)
__global__ void __launch_bounds__(256, 2) loop(const uint4 *__restrict__ tiles, int ntile, float *out, float thr) {
  const int l = threadIdx.x & 63;
  h8 bh[4], bl[4];
  for (int s = 0; s < 4; s++)
    for (int k = 0; k < 8; k++) { bh[s][k] = (_Float16)(0.01f * (l + s + k)); bl[s][k] = (_Float16)(0.001f * (l - s + k)); }
  const f16v zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  f16v ta[4], tb[4];
  for (int s = 0; s < 4; s++) { ta[s] = zero; tb[s] = zero; for (int i = 0; i < 16; i++) { ta[s][i] = 1.0f + i + l; tb[s][i] = 2.0f + i; } }
  float acc = 1e30f;
  int hits = 0;
  auto step = [&](f16v (&cur)[4], const f16v (&prev)[4], const h8 &av) {
    float mins[4];
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const int s = j & 3;
      if (MODE != 1) {
        cur[s] = j < 4 ? __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bh[s], zero, 0, 0, 0) : __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bl[s], cur[s], 0, 0, 0);
      }
      if (MODE == 2 || MODE == 1) {
        __builtin_amdgcn_sched_barrier(0);
        const f16v &t = prev[j >> 1];
        float m;
        if ((j & 1) == 0) {
          m = __builtin_fminf(__builtin_fminf(t[0], t[1]), t[2]);
          m = __builtin_fminf(__builtin_fminf(m, t[3]), t[4]);
          m = __builtin_fminf(__builtin_fminf(m, t[5]), t[6]);
          m = __builtin_fminf(__builtin_fminf(m, t[7]), t[8]);
        } else {
          m = mins[j >> 1];
          m = __builtin_fminf(__builtin_fminf(m, t[9]), t[10]);
          m = __builtin_fminf(__builtin_fminf(m, t[11]), t[12]);
          m = __builtin_fminf(__builtin_fminf(m, t[13]), t[14]);
          m = __builtin_fminf(m, t[15]);
        }
        mins[j >> 1] = m;
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (MODE == 3) {
#pragma unroll
      for (int s = 0; s < 4; s++) {
        const f16v &t = cur[s];
        float m = __builtin_fminf(__builtin_fminf(t[0], t[1]), t[2]);
#pragma unroll
        for (int i = 3; i < 15; i += 2) m = __builtin_fminf(__builtin_fminf(m, t[i]), t[i + 1]);
        mins[s] = __builtin_fminf(m, t[15]);
      }
    }
    if (MODE == 0) {
#pragma unroll
      for (int s = 0; s < 4; s++) mins[s] = cur[s][0];
    }
    bool any = false;
#pragma unroll
    for (int s = 0; s < 4; s++) any = any || (mins[s] <= thr);
    if (__ballot(any) != 0ull) { hits++; acc = fminf(acc, mins[0] + mins[1] + mins[2] + mins[3]); }
  };
  uint4 a0 = tiles[l], a1 = tiles[64 + l], a2 = tiles[128 + l], a3 = tiles[192 + l];
  for (int k = 0; k < ntile; k += 2) {
    h8 av;
    __builtin_memcpy(&av, &a0, 16);
    a0 = a1; a1 = a2; a2 = a3; a3 = tiles[(size_t)((k + 4) & 1023) * 64 + l];
    step(tb, ta, av);
    __builtin_memcpy(&av, &a0, 16);
    a0 = a1; a1 = a2; a2 = a3; a3 = tiles[(size_t)((k + 5) & 1023) * 64 + l];
    step(ta, tb, av);
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc + hits + ta[0][3] + tb[1][2];
}

template <int MODE>
void run(const char *name, int wgs_per_cu, const uint4 *tiles, float *out) {
  const int ntile = 8192;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = 256 * wgs_per_cu;  // four waves per workgroup: wgs_per_cu waves per SIMD
  hipLaunchKernelGGL(loop<MODE>, dim3(grid), dim3(256), 0, 0, tiles, ntile, out, -1e30f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(loop<MODE>, dim3(grid), dim3(256), 0, 0, tiles, ntile, out, -1e30f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double ns_per_tile = ms * 1e6 / ntile / wgs_per_cu;  // per tile of one wave, as a SIMD sees them one after the other
  printf("%-28s %d wave(s)/SIMD: %7.3f ms, %6.1f ns per wave-tile on a SIMD = %5.0f cycles at 2.4 GHz (matrix instructions alone: 256)\n", name,
         wgs_per_cu, ms, ns_per_tile, ns_per_tile * 2.4);
}

int main() {
  uint4 *tiles; float *out;
  hipMalloc(&tiles, 1024 * 64 * 16 + 4096);
  hipMemset(tiles, 0x11, 1024 * 64 * 16 + 4096);
  hipMalloc(&out, 512 * 256 * 4);
  for (int w = 1; w <= 2; w++) {
    run<0>("matrix only", w, tiles, out);
    run<1>("folds only", w, tiles, out);
    run<2>("interleaved (round 5)", w, tiles, out);
    run<3>("matrix, then folds (round 4)", w, tiles, out);
  }
  return 0;
}
