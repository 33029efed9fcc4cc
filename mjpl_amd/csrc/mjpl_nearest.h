// mjpl_nearest.h -- device code of the batched nearest neighbour:
// Tree.nearest_neighbor (reference src/mjpl/planning/tree.py:57-66) for many queries at once -- the plain float64
// scan, and for large trees the same scan behind a reduced-precision screen (binary32 on the vector units,
// two-term binary16 on the matrix cores).  Every variant returns the float64 scan's winner and distance.
// Included by mjpl_hip.hip (same translation unit: kBlock and the launch helpers are its).
#pragma once

#include <limits>

#include "mjpl_device.h"
#include "mjpl_filter.h"

namespace {

using namespace mjpl;

// Tree.nearest_neighbor (planning/tree.py:57-66) for a batch of queries: squared Euclidean
// distance in float64, node tiles staged through LDS, ties to the lowest node index.
__global__ void __launch_bounds__(kBlock)
k_nearest(const double *__restrict__ nodes, int64_t n, int64_t cap, const double *__restrict__ queries,
          int64_t M, int nplan, int32_t *__restrict__ out_idx, double *__restrict__ out_d2) {
  extern __shared__ double smem[];
  const int B = blockDim.x;
  double *qs = smem;                       // [nplan][B] this block's queries
  double *tile = smem + (size_t)nplan * B; // [nplan][B] node tile
  const int64_t j = (int64_t)blockIdx.x * B + threadIdx.x;
  for (int c = 0; c < nplan; c++) qs[c * B + threadIdx.x] = (j < M) ? queries[(int64_t)c * M + j] : 0.0;
  double best = std::numeric_limits<double>::infinity();
  int32_t besti = -1;
  for (int64_t base = 0; base < n; base += B) {
    __syncthreads();
    const int64_t src = base + threadIdx.x;
    for (int c = 0; c < nplan; c++) tile[c * B + threadIdx.x] = (src < n) ? nodes[(int64_t)c * cap + src] : 0.0;
    __syncthreads();
    const int lim = (int)((n - base) < B ? (n - base) : B);
    for (int t = 0; t < lim; t++) {
      double s = 0;
      for (int c = 0; c < nplan; c++) {
        double d = tile[c * B + t] - qs[c * B + threadIdx.x];
        s = s + d * d;
      }
      if (s < best) { best = s; besti = (int32_t)(base + t); }
    }
  }
  if (j < M) {
    out_idx[j] = besti;
    if (out_d2) out_d2[j] = best;
  }
}

// Partial nearest neighbour: block (x, y) scans node chunk y for 512 queries (four per thread, held
// in registers; node tiles are staged through LDS and read as broadcasts: 84 float64 operations
// per 7 LDS reads for a 7-joint arm).  Distances are the same sequential-sum squared norms as on
// the host (s = s + d * d, the product rounded on its own: numpy's norm of a row); ties go to the
// lowest node index (strict `<`, chunks reduced in order).  NP: the number of planning columns
// when it is one of the instantiated ones (registers and loops sized for it), 0 = any up to 16.
constexpr int kNNThreads = 128, kNNMaxPlan = 16, kNNQueries = 4;

template <int NP>
__global__ void __launch_bounds__(kNNThreads)
k_nearest_part(const double *__restrict__ nodes, int64_t n, int64_t cap, const double *__restrict__ queries,
               int64_t M, int nplan_rt, int64_t chunk, int32_t *__restrict__ pidx, double *__restrict__ pd2,
               int64_t stride = 1,  // (stride > 1: every stride-th node only -- a sample; indices are sample indices)
               const unsigned *__restrict__ only_if_wild = nullptr) {
  if (only_if_wild && only_if_wild[1] == 0u) return;  // (the matrix cores found the sample's bounds: k_nearest_mfma<NP, true>)
  constexpr int W = NP ? NP : kNNMaxPlan;
  const int nplan = NP ? NP : nplan_rt;
  __shared__ double tile[W * kNNThreads];
  const int t = threadIdx.x;
  const int64_t j0 = (int64_t)blockIdx.x * (kNNQueries * kNNThreads) + t;
  double q[kNNQueries][W];
#pragma unroll
  for (int a = 0; a < kNNQueries; a++)
#pragma unroll
    for (int c = 0; c < W; c++) {
      const int64_t j = j0 + (int64_t)a * kNNThreads;
      q[a][c] = (c < nplan && j < M) ? queries[(int64_t)c * M + j] : 0.0;
    }
  const int64_t lo = (int64_t)blockIdx.y * chunk, hi = (lo + chunk < n) ? lo + chunk : n;
  double best[kNNQueries];
  int32_t bi[kNNQueries];
#pragma unroll
  for (int a = 0; a < kNNQueries; a++) { best[a] = std::numeric_limits<double>::infinity(); bi[a] = -1; }
  for (int64_t base = lo; base < hi; base += kNNThreads) {
    __syncthreads();
    const int64_t src = base + t;
    for (int c = 0; c < nplan; c++) tile[c * kNNThreads + t] = (src < hi) ? nodes[(int64_t)c * cap + src * stride] : 0.0;
    __syncthreads();
    const int lim = (int)((hi - base) < kNNThreads ? (hi - base) : kNNThreads);
    for (int k = 0; k < lim; k++) {
      double s[kNNQueries];
#pragma unroll
      for (int a = 0; a < kNNQueries; a++) s[a] = 0;
#pragma unroll
      for (int c = 0; c < W; c++) {
        if (NP || c < nplan) {
          const double v = tile[c * kNNThreads + k];
#pragma unroll
          for (int a = 0; a < kNNQueries; a++) {
            const double d = v - q[a][c];
            s[a] = s[a] + d * d;
          }
        }
      }
#pragma unroll
      for (int a = 0; a < kNNQueries; a++)
        if (s[a] < best[a]) { best[a] = s[a]; bi[a] = (int32_t)(base + k); }
    }
  }
#pragma unroll
  for (int a = 0; a < kNNQueries; a++) {
    const int64_t j = j0 + (int64_t)a * kNNThreads;
    if (j < M) { pidx[(int64_t)blockIdx.y * M + j] = bi[a]; pd2[(int64_t)blockIdx.y * M + j] = best[a]; }
  }
}

// The same partial scan with a binary32 screen in front of the float64 arithmetic (large trees and
// query sets): eight queries per lane held as packed float pairs, the node tile as floats; a node's
// exact float64 squared distance -- the statements of k_nearest_part, so the same value and the same
// winner -- is evaluated only when its binary32 estimate does not rule it out:
//   s32 <= thr,   thr >= (R2 + 2 (2 sqrt(NP) e r + NP e^2)) (1 + 1e-6),   e = 2^-22 X,
// with R2 = min(best exact squared distance so far in this chunk, bound2), r = sqrt(R2), bound2 = the
// exact squared distance from the query to SOME node of the tree (found beforehand over a strided
// sample: the answer is never farther), and X the largest coordinate magnitude among this tile's
// nodes and the query.  (Rounding node and query to binary32 and their difference: |error| <= e per
// column; over the NP columns that moves the squared distance of a node no farther than r by at most
// 2 sqrt(NP) e r + NP e^2; NP fused multiply-adds lose at most NP 2^-24 of it.  The factor two and
// the 1e-6 are margin.)  A node farther than r cannot be the answer; one at most that far always
// passes the screen and is then compared exactly, in scan order with a strict <, so ties still go to
// the lowest index.  The sample keeps the screen tight from the first node on: without it a tree
// whose later nodes lie closer to the query (chains growing towards it) improves the running best
// at nearly every step, and every improvement is an exact evaluation.
constexpr int kNN32Queries = 8;

template <int NP>
__global__ void __launch_bounds__(kNNThreads, 4)  // (left alone the compiler prefetches tiles into 256 registers)
k_nearest_part32(const double *__restrict__ nodes, int64_t n, int64_t cap, const double *__restrict__ queries,
                 int64_t M, int64_t chunk, const double *__restrict__ bound2, int32_t *__restrict__ pidx,
                 double *__restrict__ pd2, const unsigned *__restrict__ only_if_wild = nullptr) {
  typedef float v2f __attribute__((ext_vector_type(2)));
  if (only_if_wild && only_if_wild[1] == 0u) return;  // (the matrix-core screen below serves this call)
  __shared__ float tile[NP * kNNThreads];
  __shared__ float wmax[kNNThreads / 64];
  const int t = threadIdx.x;
  const int64_t j0 = (int64_t)blockIdx.x * (kNN32Queries * kNNThreads) + t;
  auto qindex = [&](int a) -> int64_t { return j0 + (int64_t)a * kNNThreads; };
  v2f q[kNN32Queries / 2][NP];
  float qmax[kNN32Queries];
#pragma unroll
  for (int a = 0; a < kNN32Queries; a++) {
    float m = 0;
#pragma unroll
    for (int c = 0; c < NP; c++) {
      const double v = qindex(a) < M ? queries[(int64_t)c * M + qindex(a)] : 0.0;
      const float f = (float)v;
      if (a & 1) q[a / 2][c].y = f; else q[a / 2][c].x = f;
      m = fmaxf(m, fminf(fabsf(f), 1e30f));
    }
    qmax[a] = m;
  }
  const int64_t lo = (int64_t)blockIdx.y * chunk, hi = (lo + chunk < n) ? lo + chunk : n;
  const double kInf = std::numeric_limits<double>::infinity();
  double best[kNN32Queries], ref2[kNN32Queries], ref[kNN32Queries];  // ref2 = min(best, bound2), ref = sqrt(ref2)
  int32_t bi[kNN32Queries];
  float thr[kNN32Queries];
#pragma unroll
  for (int a = 0; a < kNN32Queries; a++) {
    best[a] = kInf;
    ref2[a] = (bound2 && qindex(a) < M) ? bound2[qindex(a)] : kInf;
    ref[a] = ref2[a] < kInf ? sqrt(ref2[a]) : kInf;
    bi[a] = -1;
    thr[a] = std::numeric_limits<float>::infinity();
  }
  for (int64_t base = lo; base < hi; base += kNNThreads) {
    __syncthreads();
    const int64_t src = base + t;
    float mine = 0;
#pragma unroll
    for (int c = 0; c < NP; c++) {
      const double dv = (src < hi) ? nodes[(int64_t)c * cap + src] : kInf;
      const float f = (float)dv;
      tile[c * kNNThreads + t] = f;
      // (an infinite node never wins: no part in X; a finite one beyond binary32's range makes the tile "wild")
      if (fabs(dv) < kInf) mine = fmaxf(mine, fminf(fabsf(f), 1e30f));
    }
    for (int o = 32; o > 0; o >>= 1) mine = fmaxf(mine, __shfl_xor(mine, o));
    if ((t & 63) == 0) wmax[t >> 6] = mine;
    __syncthreads();
    float tmax = wmax[0];
#pragma unroll
    for (int w = 1; w < kNNThreads / 64; w++) tmax = fmaxf(tmax, wmax[w]);
    // this tile's thresholds
#pragma unroll
    for (int a = 0; a < kNN32Queries; a++) {
      if (!(fmaxf(tmax, qmax[a]) < 1e18f)) {
        thr[a] = std::numeric_limits<float>::infinity();  // squares would overflow binary32: no screen here
      } else if (ref2[a] < kInf) {
        const double X = (double)fmaxf(tmax, qmax[a]) * (1.0 + 1e-6);
        const double e = X * 0x1p-22;
        const double slack = 2.0 * (2.0 * sqrt((double)NP) * e * ref[a] + NP * e * e);
        const double th = (ref2[a] + slack) * (1.0 + 1e-6);
        float f = (float)th;
        if ((double)f < th) f = __int_as_float(__float_as_int(f) + 1);  // (th >= 0: the next float up)
        thr[a] = f;
      }
    }
    const int lim = (int)((hi - base) < kNNThreads ? (hi - base) : kNNThreads);
    for (int k = 0; k < lim; k++) {
      v2f s[kNN32Queries / 2];
#pragma unroll
      for (int p2 = 0; p2 < kNN32Queries / 2; p2++) s[p2] = (v2f){0.0f, 0.0f};
#pragma unroll
      for (int c = 0; c < NP; c++) {
        const float v = tile[c * kNNThreads + k];
        const v2f vv = (v2f){v, v};
#pragma unroll
        for (int p2 = 0; p2 < kNN32Queries / 2; p2++) {
          const v2f d = vv - q[p2][c];
          s[p2] = __builtin_elementwise_fma(d, d, s[p2]);
        }
      }
      bool hit = false;
#pragma unroll
      for (int p2 = 0; p2 < kNN32Queries / 2; p2++) hit = hit || (s[p2].x <= thr[2 * p2]) || (s[p2].y <= thr[2 * p2 + 1]);
      if (__ballot(hit) != 0ull) {
        if (hit) {
          const int64_t node = base + k;
#pragma unroll
          for (int a = 0; a < kNN32Queries; a++) {
            const float sa = (a & 1) ? s[a / 2].y : s[a / 2].x;
            if (sa <= thr[a] && qindex(a) < M) {
              double ex = 0;
#pragma unroll
              for (int c = 0; c < NP; c++) {
                const double d = nodes[(int64_t)c * cap + node] - queries[(int64_t)c * M + qindex(a)];
                ex = ex + d * d;
              }
              if (ex < best[a]) {
                best[a] = ex;
                bi[a] = (int32_t)node;
                if (ex < ref2[a]) { ref2[a] = ex; ref[a] = sqrt(ex); }
                const double X = (double)fmaxf(tmax, qmax[a]) * (1.0 + 1e-6);
                const double e = X * 0x1p-22;
                const double th = (ref2[a] + 2.0 * (2.0 * sqrt((double)NP) * e * ref[a] + NP * e * e)) * (1.0 + 1e-6);
                float f = (float)th;
                if ((double)f < th) f = __int_as_float(__float_as_int(f) + 1);  // (th >= 0: the next float up)
                thr[a] = (fmaxf(tmax, qmax[a]) < 1e18f) ? f : std::numeric_limits<float>::infinity();
              }
            }
          }
        }
      }
    }
  }
#pragma unroll
  for (int a = 0; a < kNN32Queries; a++)
    if (qindex(a) < M) {
      pidx[(int64_t)blockIdx.y * M + qindex(a)] = bi[a];
      pd2[(int64_t)blockIdx.y * M + qindex(a)] = best[a];  // (+inf, -1: nothing within the bound in this chunk)
    }
}

// ---- the same screen on the matrix cores (large trees and query sets, coordinates of ordinary size) --------------
// Squared distances expand to |x|^2 - 2 q . x + |q|^2, and the cross terms of 32 nodes x 32 queries are two chained
// v_mfma_f32_32x32x16_f16.  Once per call (k_nn_pack) every coordinate v of a node or query is written as
// v_h + v_l, two binary16 numbers (v_h = v rounded, v_l = the rest rounded: 22 bits of v).  A node's row of the A
// operand is [x_h (7), n_1 | x_l (7), n_2] with n = |x~|^2, x~ = x_h + x_l, summed in binary32 and split in two
// binary16 parts; a query has two columns of B: [-2 q_h (7), 1 | -2 q_h (7), 1] and [-2 q_l (7), 0 | -2 q_l (7), 0].
// The first instruction returns n - 2 q_h . x~, the second adds -2 q_l . x~: t = |x~|^2 - 2 q~ . x~ for 1024 pairs,
// node on the register index, query on the lane, and a pair passes when t <= T_q = thr - |q~|^2.  Sixteen results
// per lane are folded with eight v_min3 and compared once; only a wave that sees a pass looks at the sixteen one
// by one and evaluates the exact float64 distance of those -- k_nearest_part's statements, in scan order with a
// strict <.
//   thr >= (R2 + 2 (4 sqrt(NP) e r + 4 NP e^2 + a)) (1 + 1e-6),  e = 1.5 2^-22 X + 2^-24,  a = 2^-18 (max |x~| + |q~|)^2,
// R2 = min(best exact squared distance so far of this lane, bound2), r = sqrt(R2), X = the largest finite coordinate
// magnitude among all nodes and queries (k_nn_pack).  (v -> binary32 loses 2^-24 |v|; v_h is within 2^-11 |v| of
// that, v_l within 2^-11 of the rest or 2^-25 where it is subnormal -- the matrix cores keep binary16 subnormals,
// tools/micro/mfma_f16_denorm.hip --: e per column, on either side; so the represented pair's distance differs from
// the true one by at most 2 sqrt(NP) e, and a node no farther than r has a represented squared distance of at most
// R2 + 4 sqrt(NP) e r + 4 NP e^2.  Arithmetic: the products are exact in binary32; the seven-term sums of |x~|^2
// and |q~|^2, the split of n (2^-22 n), the 32-term accumulation of the two instructions and the subtraction of
// |q~|^2 lose at most 50 2^-24 (|x~| + |q~|)^2 together: a takes 64 2^-24 of that with the query's own |q~| and the largest
// |x~| among the nodes (round 4: 2^-16 NP X^2 for all pairs, four to seven times as much).  The factor two and the 1e-6 are margin.)
// With single binary16 coordinates -- one instruction, e = 2^-11 X -- the screen let through every node within
// 0.02 rad of the bound: a quarter of a tree whose chains all start at one root.  A node at +inf gives +inf or NaN,
// which v_min3 and the ordered compare ignore: never nearest, as in the other kernels.  Two lanes share a query
// (rows 4h .. 4h+3 of every eight: h = lane / 32), each keeps its own best; the reduction breaks ties by index.
// Coordinates of 256 or more (or NaN) anywhere, or a finite node whose squared norm does not fit binary16
// (>= 65504): k_nn_pack raises a flag, this kernel returns at once and the binary32 screen above does the work.
// nplan <= 7.
typedef _Float16 nn_h8 __attribute__((ext_vector_type(8)));
typedef float nn_f16 __attribute__((ext_vector_type(16)));
constexpr int kNNMSets = 4, kNNMWaves = 4;              // 32-query sets per wave, waves per workgroup
constexpr int kNNMQueries = kNNMSets * kNNMWaves * 32;  // 512 queries per workgroup
constexpr int kNNMMaxPlan = 7;
constexpr float kNNMWild = 256.0f;

// nodes / queries -> the operand rows, |x~|^2, and X.  Nodes: 32 bytes (k = 0..7 | 8..15); queries: 64 (two columns).
constexpr int kNNPackRows = 8;  // rows per thread: a launch has ceil(padded / (256 kNNPackRows)) workgroups
__global__ void __launch_bounds__(256)
k_nn_pack(const double *__restrict__ src, int64_t count, int64_t col_stride, int64_t padded, int nplan, int is_query,
          uint4 *__restrict__ out16, float *__restrict__ nrm, unsigned *__restrict__ xbits, int64_t row_stride = 1,
          const float *__restrict__ centre = nullptr) {
  // (centre, the cell-ordered scan: every coordinate relative to the middle of its column's range -- nodes and queries alike,
  //  so the distances are the same and the magnitudes, and with them the screen's arithmetic allowance, smaller)
  // (coordinate c of row i: src[c * col_stride + i * row_stride] -- columns [nplan][count] by default, rows of eight with (1, 8))
  float mx = 0, n2max = 0;
  bool wild = false;
  // (kNNPackRows rows per thread: the three atomics at the end are per wave, and with one row per thread a tree of two
  //  million nodes sent sixty thousand of them to one cache line -- 0.6 ms of a 0.66 ms kernel)
  for (int rep = 0; rep < kNNPackRows; rep++) {
  const int64_t i = ((int64_t)blockIdx.x * kNNPackRows + rep) * blockDim.x + threadIdx.x;
  if (i < padded) {
    union { _Float16 h[16]; uint4 u[2]; } hi, lo;
    for (int k = 0; k < 16; k++) { hi.h[k] = (_Float16)0.0f; lo.h[k] = (_Float16)0.0f; }
    float n2 = 0;
    if (i < count) {
      for (int c = 0; c < nplan; c++) {
        const double v = src[(int64_t)c * col_stride + i * row_stride] - (centre ? (double)centre[c] : 0.0);
        const float f = (float)v;
        if (v != v) wild = true;
        if (fabs(v) < std::numeric_limits<double>::infinity()) {
          if (!(fabs(v) < (double)kNNMWild)) wild = true;
          mx = fmaxf(mx, fminf(fabsf(f), 1e30f));
        }
        const _Float16 vh = (_Float16)f;
        const float rest = f - (float)vh;  // exact (or inf - inf: a node that never passes)
        const _Float16 vl = (_Float16)rest;
        const float rep = (float)vh + (float)vl;
        n2 = n2 + rep * rep;
        if (is_query) {
          hi.h[c] = hi.h[8 + c] = (_Float16)(-2.0f * (float)vh);
          lo.h[c] = lo.h[8 + c] = (_Float16)(-2.0f * (float)vl);
        } else {
          hi.h[c] = vh;
          hi.h[8 + c] = vl;
        }
      }
    } else if (!is_query) {
      hi.h[0] = (_Float16)std::numeric_limits<float>::infinity();  // padding: a node that never passes
      n2 = std::numeric_limits<float>::infinity();
    }
    if (is_query) {
      hi.h[7] = hi.h[15] = (_Float16)1.0f;
      out16[4 * i] = hi.u[0];
      out16[4 * i + 1] = hi.u[1];
      out16[4 * i + 2] = lo.u[0];
      out16[4 * i + 3] = lo.u[1];
    } else {
      // |x~|^2 travels as two binary16 numbers: beyond binary16's largest finite value (65504: one coordinate of
      // 256, or seven of 97) n1 would be +inf, the rest -inf and the instruction's sum NaN -- a node that can
      // never pass the screen although it may be the nearest.  Such a call belongs to the binary32 screen.
      if (n2 < std::numeric_limits<float>::infinity() && !(n2 < 65504.0f)) wild = true;
      const _Float16 n1 = (_Float16)n2;
      const float rest = n2 - (float)n1;  // exact; |rest| <= 2^-11 n2
      hi.h[7] = n1;
      hi.h[15] = (n2 < std::numeric_limits<float>::infinity()) ? (_Float16)rest : (_Float16)0.0f;
      out16[2 * i] = hi.u[0];
      out16[2 * i + 1] = hi.u[1];
    }
    if (nrm) nrm[i] = n2;
    // (the largest finite |x~|^2 among the nodes: the screen's arithmetic allowance is taken per query against it, below)
    if (!is_query && i < count && n2 < std::numeric_limits<float>::infinity()) n2max = fmaxf(n2max, n2);
  }
  }
  for (int o = 32; o > 0; o >>= 1) n2max = fmaxf(n2max, __shfl_xor(n2max, o));
  if ((threadIdx.x & 63) == 0 && n2max > 0) atomicMax(&xbits[2], __float_as_uint(n2max));
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  if ((threadIdx.x & 63) == 0 && mx > 0) atomicMax(&xbits[0], __float_as_uint(mx));
  if (__ballot(wild) != 0ull && (threadIdx.x & 63) == 0) atomicOr(&xbits[1], 1u);
}

__global__ void __launch_bounds__(256)
k_nn_fill_inf(double *__restrict__ x, int64_t M, const unsigned *__restrict__ xbits) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j < M && xbits[1] == 0u) x[j] = std::numeric_limits<double>::infinity();
}

template <int NP>
__device__ __forceinline__ float nn_mfma_threshold(double ref2, double ref, double e, double a, float nq) {
  const double th = (ref2 + 2.0 * (4.0 * sqrt((double)NP) * e * ref + 4.0 * NP * e * e + a)) * (1.0 + 1e-6) - (double)nq;
  float f = (float)th;
  if ((double)f < th) f = nextafterf(f, std::numeric_limits<float>::infinity());
  return f;
}

// The scan's loop, round 5 (profiles/README.md): (1) a wave keeps kNNMAhead node tiles in flight -- with one, every
// iteration waited for a trip to L2 that the 0.1 us of arithmetic of a tile does not cover; (2) a pair that passes
// the screen is PARKED -- (node, set) in the lane's own list in LDS, kNNMPark entries -- and the lists are worked off
// together when one fills up and at the end of the chunk: the exact distance wants fourteen loads from HBM, and
// taken on the spot a wave waited ~1.5 us for them at one or two live lanes, a hundred and fifty times per chunk.
// A lane works its list off in the order it was filled, which is scan order per query, with a strict <: the same
// winner.  Between two passes over the lists the thresholds stand still (they only ever fall), so a few more pairs
// are parked than the immediate evaluation would have looked at; every one of them is compared exactly.
// SAMPLE: the pass in front of the scan that gives every query its bound (bound2: the exact squared distance to SOME
// node).  Round 4 found it with a float64 scan of a strided sample of 16 384 nodes (1.7 ms at planner size, 40 % of
// the scan behind it); here the same instructions pick the sample node with the smallest screened value -- no
// threshold, no exact arithmetic in the loop -- and ONE exact distance per lane at the end is the bound: any node's
// distance is a valid one, and the screen's argmin is within its error of the sample's best.
constexpr int kNNMAhead = 4, kNNMPark = 32;

// CELLS (round 6; mjpl_nearest_cells.h): the nodes arrive sorted along a space-filling curve of their
// cells, so that the kNNCellSub nodes of a "sub-chunk" lie together in a small box; the queries arrive sorted by where on
// that curve the sample pass found their bound, so that the 128 queries of a wave look at the same part of the tree.  A
// pass in front of the scan (k_nn_candidates) marks, per wave, the sub-chunks whose box comes within SOME query's bound of
// that query; the wave scans those and nothing else.  What is skipped holds no node within the bound of any of the wave's
// queries, so the winners are those of the full scan; equal distances go to the lower node id through the permutation.
constexpr int kNNCellSub = 256;    // nodes per sub-chunk (a multiple of 32)
struct NnCells {
  const int32_t *list;             // [waves of 128 sorted queries][list_pitch]: the wave's candidate sub-chunks, ascending
  const int32_t *count;            // [waves]: how many
  const int32_t *perm_n;           // sorted position -> node id
  int64_t list_pitch;
  int second;                      // the parked pairs' second screen is on (option "nn_second_screen")
  int node_rows, query_rows;       // the float64 nodes / queries are rows of eight ([count][8]) instead of columns
  const float *nodes32, *queries32; // the same rows in binary32, centred ([count][8]), for the parked pairs' second screen
  int probe;                       // (option "nn_probe" = 2: the pairs that reach the exact evaluation are counted in counter[0]; answers unchanged)
  unsigned *counter;
  int pack_idx;                    // (SAMPLE) the low 16 bits of a bound carry the sample position it was found at (scaled by idx_shift)
  int idx_shift;
};

// (NW: waves per workgroup.  The full scan's four waves walk the same chunk and end together; a cell-ordered scan's waves
//  have lists of their own, and a workgroup held its slot until the slowest of four was through -- half the chip's wave
//  slots stood empty on average (SQ_WAVE_CYCLES / SQ_BUSY_CYCLES 7.4 against the full scan's 14.2): ONE wave per workgroup there.)
template <int NP, bool SAMPLE, bool CELLS = false, int NW = CELLS ? 1 : kNNMWaves>
__global__ void __launch_bounds__(NW * 64, 2)
k_nearest_mfma(const double *__restrict__ nodes, int64_t n, int64_t cap, const double *__restrict__ queries, int64_t M,
               const uint4 *__restrict__ nodes16, const uint4 *__restrict__ queries16, const float *__restrict__ qnorm,
               const unsigned *__restrict__ xbits, int64_t chunk, int64_t stride, double *__restrict__ bound2,
               int32_t *__restrict__ pidx, double *__restrict__ pd2, NnCells cells = NnCells{}) {
  static_assert(!(SAMPLE && CELLS), "the sample pass is a plain scan");
  static_assert(NP <= kNNMMaxPlan, "seven coordinate slots per half of the operand");
  if (xbits[1] != 0u) return;  // wild coordinates: the binary32 screen (and the float64 sample scan) serve this call
  // a lane's state that the loop itself does not touch lives in LDS (the two accumulator sets, the queries' operands and
  // the tiles in flight leave ~50 registers): the parked pairs, and per set the best exact distance, its node, the bound
  constexpr int kThreads = NW * 64;
  __shared__ int32_t park[SAMPLE ? 1 : kNNMPark * kThreads];
  __shared__ double best_l[SAMPLE ? 1 : kNNMSets * kThreads], ref2_l[SAMPLE ? 1 : kNNMSets * kThreads];
  __shared__ int32_t bi_l[kNNMSets * kThreads];
  const int l = threadIdx.x & 63, r = l & 31, h = l >> 5, w = threadIdx.x >> 6;
  const int64_t q0 = ((int64_t)blockIdx.x * NW + w) * (kNNMSets * 32) + r;
  const double kInf = std::numeric_limits<double>::infinity();
  const double X = (double)__uint_as_float(xbits[0]);
  // The arithmetic allowance `a` of nn_mfma_threshold, per query: the sums lose at most 50 2^-24 (|x~| + |q~|)^2 (comment above);
  // round 4 took the bound for ALL pairs, 2^-16 NP X^2.  With a query's own |q~| and the largest |x~| among the nodes
  // (k_nn_pack) it is several times smaller -- and in a dense tree every node within sqrt(2 a) of a query's best passes the
  // screen: at 2 10^6 nodes on a five-dimensional manifold a look-up of configurations ON the tree took 160 ms with
  // sqrt(2 a) = 0.045 (profiles/README.md, round 5).
  const double N2 = (double)__uint_as_float(xbits[2]) * (1.0 + 1e-5);
  const double e = 1.5 * 0x1p-22 * X + 0x1p-24;
  auto allowance = [&](float nqv) {
    const double sxy = sqrt(N2) + sqrt((double)nqv * (1.0 + 1e-5));
    return 0x1p-18 * sxy * sxy;
  };
  nn_h8 bh[kNNMSets], bl[kNNMSets];
  float T[kNNMSets];
  double *best = best_l + (SAMPLE ? 0 : threadIdx.x), *ref2 = ref2_l + (SAMPLE ? 0 : threadIdx.x);  // [set * kThreads]
  int32_t *bi = bi_l + threadIdx.x;
#pragma unroll
  for (int s = 0; s < kNNMSets; s++) {
    const int64_t q = q0 + 32 * s;  // (the packed queries are padded to whole workgroups)
    const uint4 u = queries16[4 * q + h], v = queries16[4 * q + 2 + h];
    __builtin_memcpy(&bh[s], &u, 16);
    __builtin_memcpy(&bl[s], &v, 16);
    bi[s * kThreads] = -1;
    if constexpr (SAMPLE) {
      T[s] = std::numeric_limits<float>::infinity();  // (the smallest screened value so far: a strict < below)
    } else {
      const double b2 = (q < M) ? bound2[q] : 0.0;
      best[s * kThreads] = kInf;
      ref2[s * kThreads] = b2;
      T[s] = b2 < kInf ? nn_mfma_threshold<NP>(b2, sqrt(b2), e, allowance(qnorm[q]), qnorm[q]) : std::numeric_limits<float>::infinity();
      if (q >= M) T[s] = -std::numeric_limits<float>::infinity();
    }
  }
  // the node range under the loop: a chunk of the launch -- or (CELLS) one run of sub-chunks after another; parked pairs
  // count their nodes from glo, the start of everything this workgroup may scan
  int64_t lo = (int64_t)blockIdx.y * chunk, hi = (lo + chunk < n) ? lo + chunk : n;
  const int64_t glo = CELLS ? 0 : lo;  // (CELLS: a row of the grid takes sub-chunks from all over the tree; n < 2^26)
  const nn_f16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  int32_t *mypark = park + (SAMPLE ? 0 : threadIdx.x);
  int parked = 0;
  // (CELLS) the second screen's thresholds: a node within sqrt(b2) of its query in float64 has a direct binary32 squared
  // distance of at most (sqrt(b2) + d)^2 (1 + 2^-18), d = sqrt(NP) 2^-22 X: both rows were rounded to binary32 (2^-24 |x| each)
  // and so was their difference -- 2^-22 X per column at most, X the largest magnitude among the centred coordinates (these
  // rows are centred like the screen's operands); the NP fused multiply-adds lose less than 2^-20 of the sum.
  float thr32[kNNMSets];
  auto second_screen = [&](double b2) -> float {
    if (!(b2 < kInf)) return std::numeric_limits<float>::infinity();
    const double d = sqrt((double)NP) * 0x1p-22 * X * (1.0 + 1e-6);
    const double rr = sqrt(b2) + d;
    const double th = rr * rr * (1.0 + 0x1p-18);
    float f = (float)th;
    if ((double)f < th) f = nextafterf(f, std::numeric_limits<float>::infinity());
    return f;
  };
  if constexpr (CELLS) {
#pragma unroll
    for (int s = 0; s < kNNMSets; s++) thr32[s] = second_screen(ref2[s * kThreads]);
  }

  // exact distance of node `node` from query q: k_nearest_part's statements
  auto exact = [&](int64_t node, int64_t q) {
    double ex = 0;
    // (the cell-ordered scan hands over rows of eight doubles -- a node is ONE cache line, not seven: in a dense tree a
    //  look-up evaluates hundreds of millions of these)
    if constexpr (CELLS) {  // (both in rows: four 16-byte loads each)
      double nv[8], qv[8];
      const double2 *np2 = reinterpret_cast<const double2 *>(nodes + 8 * node), *qp2 = reinterpret_cast<const double2 *>(queries + 8 * q);
#pragma unroll
      for (int k = 0; k < (NP + 1) / 2; k++) {
        const double2 a = np2[k], b = qp2[k];
        nv[2 * k] = a.x; nv[2 * k + 1] = a.y; qv[2 * k] = b.x; qv[2 * k + 1] = b.y;
      }
#pragma unroll
      for (int c = 0; c < NP; c++) {
        const double d = nv[c] - qv[c];
        ex = ex + d * d;
      }
      return ex;
    }
    const double *nrow = cells.node_rows ? nodes + 8 * node : nodes + node;
    const double *qrow = cells.query_rows ? queries + 8 * q : queries + q;
    const int64_t ncs = cells.node_rows ? 1 : cap, qcs = cells.query_rows ? 1 : M;
#pragma unroll
    for (int c = 0; c < NP; c++) {
      const double d = nrow[(int64_t)c * ncs] - qrow[(int64_t)c * qcs];
      ex = ex + d * d;
    }
    return ex;
  };
  // the lanes' lists, entry j of every lane at a time (the loads of a pass travel together)
  auto work_off = [&]() {
    int most = parked;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const int other = __shfl_xor(most, o);
      most = other > most ? other : most;
    }
    unsigned touched = 0;
    if constexpr (CELLS) {
      if (cells.probe == 2) {
        int tot = parked;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) tot += __shfl_xor(tot, o);
        if (l == 0 && tot > 0) atomicAdd(cells.counter, (unsigned)tot);
      }
    }
    for (int j = 0; j < most; j++) {
      if (j < parked) {
        const int32_t ent = mypark[j * kThreads];
        const int s = ent & 3;
        const int64_t node = glo + (int64_t)(ent >> 2);
        if (CELLS && cells.second) {
          // The second screen (round 6).  In a dense tree thousands of nodes lie within the matrix-core screen's arithmetic
          // allowance of a query's best (0.02 rad where a step is 0.05; 7 10^8 parked pairs per look-up once two trees
          // mirror each other): the direct binary32 distance, ONE 32-byte row per node, tells all but the near-ties apart.
          // A pair within ref2 of its query has a binary32 distance of at most thr32 (below); the others cannot win.
          const float4 *nr = reinterpret_cast<const float4 *>(cells.nodes32 + 8 * node);
          const float4 *qr = reinterpret_cast<const float4 *>(cells.queries32 + 8 * (q0 + 32 * s));
          const float4 n0 = nr[0], n1 = nr[1], c0 = qr[0], c1 = qr[1];
          float d32 = 0, df;
          df = n0.x - c0.x; d32 = __builtin_fmaf(df, df, d32);
          if (NP > 1) { df = n0.y - c0.y; d32 = __builtin_fmaf(df, df, d32); }
          if (NP > 2) { df = n0.z - c0.z; d32 = __builtin_fmaf(df, df, d32); }
          if (NP > 3) { df = n0.w - c0.w; d32 = __builtin_fmaf(df, df, d32); }
          if (NP > 4) { df = n1.x - c1.x; d32 = __builtin_fmaf(df, df, d32); }
          if (NP > 5) { df = n1.y - c1.y; d32 = __builtin_fmaf(df, df, d32); }
          if (NP > 6) { df = n1.z - c1.z; d32 = __builtin_fmaf(df, df, d32); }
          if (!(d32 <= thr32[s])) continue;  // (NaN -- an infinite node -- never passes either)
        }
        const double ex = exact(node, q0 + 32 * s);
        bool better = ex < best[s * kThreads];
        if constexpr (CELLS)  // (sorted rows are not in id order: an equal distance goes to the lower node id)
          if (ex == best[s * kThreads] && bi[s * kThreads] >= 0) better = cells.perm_n[node] < cells.perm_n[bi[s * kThreads]];
        if (better) {
          best[s * kThreads] = ex;
          bi[s * kThreads] = (int32_t)node;
          if (ex < ref2[s * kThreads]) {
            ref2[s * kThreads] = ex;
            touched |= 1u << s;
            if constexpr (CELLS) thr32[s] = second_screen(ex);
          }
        }
      }
    }
#pragma unroll
    for (int k = 0; k < kNNMSets; k++)
      if ((touched >> k) & 1u) {  // (a set's lower bound: its threshold falls; q < M, or nothing of it was ever parked)
        const double b2 = ref2[k * kThreads];
        T[k] = nn_mfma_threshold<NP>(b2, sqrt(b2), e, allowance(qnorm[q0 + 32 * k]), qnorm[q0 + 32 * k]);
      }
    parked = 0;
  };

  // One step: the eight instructions of tile `base` into `cur`, and -- issued between them, so that the vector pipe
  // works while the matrix pipe does (a wave issues in order: with the folds BEHIND its own eight instructions it
  // waited through them, then left the matrix pipe idle through its folds, and two waves per SIMD only partly fill
  // each other's gaps: 900 cycles per pair of tiles where the instructions alone are 512) -- the folds of the tile
  // before it, whose accumulators `prev` are complete.  Then that tile's passes, if any.
  auto step = [&](nn_f16 (&cur)[kNNMSets], const nn_f16 (&prev)[kNNMSets], const nn_h8 &av, int64_t pbase, bool have_prev) {
    static_assert(kNNMSets == 4, "eight matrix instructions, half a set's fold behind each");
    float mins[kNNMSets];
#pragma unroll
    for (int j = 0; j < 2 * kNNMSets; j++) {
      const int s = j & 3;
      cur[s] = j < kNNMSets ? __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bh[s], zero, 0, 0, 0)
                            : __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bl[s], cur[s], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);  // (the order written here IS the schedule: nothing moves across)
      const nn_f16 &t = prev[j >> 1];
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
    bool hit[kNNMSets];
    bool any = false;
#pragma unroll
    for (int s = 0; s < kNNMSets; s++) {
      hit[s] = have_prev && (SAMPLE ? mins[s] < T[s] : mins[s] <= T[s]);
      any = any || hit[s];
    }
    if (__ballot(any) == 0ull) return;
    if constexpr (SAMPLE) {
#pragma unroll
      for (int s = 0; s < kNNMSets; s++) {
        if (!hit[s]) continue;
        const nn_f16 &t = prev[s];
        int at = 15;
#pragma unroll
        for (int i = 14; i >= 0; i--) at = t[i] == mins[s] ? i : at;
        T[s] = mins[s];
        bi[s * kThreads] = (int32_t)(pbase + (at & 3) + 8 * (at >> 2) + 4 * h);
      }
    } else {
#pragma unroll
      for (int s = 0; s < kNNMSets; s++) {
        if (__ballot(hit[s]) == 0ull) continue;
        if (__ballot(parked + 16 > kNNMPark) != 0ull) {  // (room for a set's sixteen on every lane, or the lists are worked off first)
          work_off();
          if (__ballot(mins[s] <= T[s]) == 0ull) continue;
        }
        const nn_f16 &t = prev[s];
#pragma unroll
        for (int i = 0; i < 16; i++) {
          const int64_t node = pbase + (i & 3) + 8 * (i >> 2) + 4 * h;
          if (t[i] <= T[s] && node < hi) {
            mypark[parked * (NW * 64)] = (int32_t)(((node - glo) << 2) | s);
            parked++;
          }
        }
      }
    }
  };
  static_assert(kNNMAhead == 4, "four named registers hold the tiles in flight (an indexed array would live in scratch)");
  nn_f16 ta[kNNMSets], tb[kNNMSets];
#pragma unroll
  for (int s = 0; s < kNNMSets; s++) ta[s] = tb[s] = zero;
  typedef unsigned nn_u4 __attribute__((ext_vector_type(4)));
  if constexpr (!CELLS) {
  // ---- the screened scan of nodes [lo, hi)
  const int ntile = __builtin_amdgcn_readfirstlane((int)((hi - lo + 31) >> 5));  // (wave-uniform, and told so: a scalar register)
  // tile k of this lane through a buffer descriptor: the chunk's first row in scalar registers, the lane's 32-bit byte
  // offset in ONE vector register, the tile's offset in a scalar one (a 64-bit address per lane was two registers more
  // than there are, and its reload from scratch -- a vector-memory operation like the tiles -- waited for every tile in flight)
  const int64_t rstride = SAMPLE ? stride : 1;  // rows of the packed nodes between two rows of a tile
  const __amdgpu_buffer_rsrc_t tiles =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4 *>(nodes16 + 2 * lo * rstride), 0, 0x7fffffff, 0x00020000);
  const unsigned loff = (unsigned)((2 * (int64_t)r * rstride + h) * 16);
  const unsigned tbytes = (unsigned)(64 * rstride * 16);  // (a chunk's tiles span less than 2^31 bytes: mjpl_nearest_dev sizes them so)
  auto fetch = [&](int k) -> uint4 {
    if (k >= ntile) return uint4{0, 0, 0, 0};
    const nn_u4 v = __builtin_amdgcn_raw_buffer_load_b128(tiles, loff, (unsigned)k * tbytes, 0);
    return uint4{v[0], v[1], v[2], v[3]};
  };
  uint4 a0 = fetch(0), a1 = fetch(1), a2 = fetch(2), a3 = fetch(3);
  // step k: the instructions of tile k (of a tile of zeros behind the last one) and the folds of tile k - 1
  auto turn = [&](nn_f16 (&cur)[kNNMSets], const nn_f16 (&prev)[kNNMSets], int k) {
    nn_h8 av;
    __builtin_memcpy(&av, &a0, 16);
    a0 = a1; a1 = a2; a2 = a3;
    a3 = fetch(k + kNNMAhead);
    step(cur, prev, av, lo + 32 * (int64_t)(k - 1), k > 0);
  };
  for (int k = 0; k <= ntile; k += 2) {  // (<=: one step more, which folds the last tile)
    turn(tb, ta, k);
    if (k + 1 <= ntile) turn(ta, tb, k + 1);
  }
  } else {
    // ---- the screened scan of this wave's candidate sub-chunks (k_nn_candidates / k_nn_compact: their numbers, ascending,
    // cells.count of them): row y of the grid's rows takes the y-th slice of the list, and walks its tiles -- kNNCellSub / 32
    // per sub-chunk -- as ONE stream through the same pipeline; the list comes through the scalar cache, an entry ahead.
    // (Range by range -- every run of consecutive sub-chunks its own prologue and drain -- the loop spent two thirds of
    //  its time filling and emptying the pipeline: 3.6 ms where the stream takes 1.3, profiles/README.md round 6.)
    constexpr int kTiles = kNNCellSub / 32;
    static_assert((kTiles & (kTiles - 1)) == 0, "a power of two");
    const int wq = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * NW + w));
    const int32_t *list = cells.list + (int64_t)wq * cells.list_pitch;
    const int cnt = __builtin_amdgcn_readfirstlane(cells.count[wq]);
    const int e0 = __builtin_amdgcn_readfirstlane((int)((int64_t)cnt * blockIdx.y / gridDim.y));
    const int e1 = __builtin_amdgcn_readfirstlane((int)((int64_t)cnt * (blockIdx.y + 1) / gridDim.y));
    const int ntile = (e1 - e0) * kTiles;
    hi = n;
    const __amdgpu_buffer_rsrc_t tiles = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4 *>(nodes16), 0, 0x7fffffff, 0x00020000);
    const unsigned loff = (unsigned)((2 * r + h) * 16);
    // two cursors over the list: the tile to fetch next, the tile whose results are folded next
    int fk = 0, f_sc = e0 < e1 ? list[e0] : 0, f_nx = e0 + 1 < e1 ? list[e0 + 1] : 0;
    auto fetch = [&]() -> uint4 {
      if (fk >= ntile) return uint4{0, 0, 0, 0};
      const unsigned soff = ((unsigned)f_sc * (unsigned)kNNCellSub + (unsigned)(fk & (kTiles - 1)) * 32u) * 32u;  // (n < 2^25: below 2^30 bytes)
      const nn_u4 v = __builtin_amdgcn_raw_buffer_load_b128(tiles, loff, soff, 0);
      fk++;
      if ((fk & (kTiles - 1)) == 0) {
        f_sc = f_nx;
        const int at = e0 + (fk / kTiles) + 1;
        f_nx = at < e1 ? list[at] : 0;
      }
      return uint4{v[0], v[1], v[2], v[3]};
    };
    uint4 a0 = fetch(), a1 = fetch(), a2 = fetch(), a3 = fetch();
    int pk = 0, p_sc = f_sc, p_nx = 0;  // (set below, at the first fold)
    p_sc = e0 < e1 ? list[e0] : 0;
    p_nx = e0 + 1 < e1 ? list[e0 + 1] : 0;
    auto turn = [&](nn_f16 (&cur)[kNNMSets], const nn_f16 (&prev)[kNNMSets], int k) {
      nn_h8 av;
      __builtin_memcpy(&av, &a0, 16);
      a0 = a1; a1 = a2; a2 = a3;
      a3 = fetch();
      int64_t pbase = 0;
      if (k > 0) {  // tile k - 1 = tile pk of the stream
        pbase = (int64_t)p_sc * kNNCellSub + (int64_t)(pk & (kTiles - 1)) * 32;
        pk++;
        if ((pk & (kTiles - 1)) == 0) {
          p_sc = p_nx;
          const int at = e0 + (pk / kTiles) + 1;
          p_nx = at < e1 ? list[at] : 0;
        }
      }
      step(cur, prev, av, pbase, k > 0);
    };
    for (int k = 0; k <= ntile; k += 2) {
      turn(tb, ta, k);
      if (k + 1 <= ntile) turn(ta, tb, k + 1);
    }
  }
  if constexpr (SAMPLE) {
    // one exact distance per lane and set: the bound of its query (the better of the two lanes that share it)
#pragma unroll
    for (int s = 0; s < kNNMSets; s++) {
      const int64_t q = q0 + 32 * s;
      double ex = kInf;
      const int32_t liked = bi[s * kThreads];
      if (q < M && liked >= 0) ex = exact((int64_t)liked * stride, q);
      const double own = ex;
      const double other = __shfl_xor(ex, 32);
      ex = other < ex ? other : ex;
      // (the sample may be cut into chunks, blockIdx.y: the smallest of their bounds -- non-negative doubles order like their bits;
      //  k_nn_fill_inf set every bound to +inf before the launch)
      unsigned long long word = (unsigned long long)__double_as_longlong(ex);
      if (cells.pack_idx) {
        // (the scan sorts its queries by where along the curve their bound was found: the bound, rounded up to a multiple of
        //  2^16 ulps -- still the distance to no node nearer than it says -- carries the sample position in its low bits)
        const int32_t mine = liked >= 0 ? liked : 0;
        const int32_t theirs = __shfl_xor(mine, 32);
        const int32_t at = other < own ? theirs : mine;
        if (ex < kInf) word = ((word + 0xffffull) & ~0xffffull) | (unsigned long long)((unsigned)(at >> cells.idx_shift) & 0xffffu);
      }
      if (q < M && h == 0) atomicMin(reinterpret_cast<unsigned long long *>(bound2 + q), word);
    }
  } else {
    work_off();
#pragma unroll
    for (int s = 0; s < kNNMSets; s++) {
      const int64_t q = q0 + 32 * s;
      if (q < M) {
        const int64_t at = ((int64_t)blockIdx.y * 2 + h) * M + q;
        int32_t won = bi[s * kThreads];
        if constexpr (CELLS) won = won >= 0 ? cells.perm_n[won] : won;
        pidx[at] = won;
        pd2[at] = best[s * kThreads];  // (+inf, -1: nothing within the bound among this lane's rows)
      }
    }
  }
}

// (partial results that are not in index order -- the two lanes of a query above: equal distances go to the lower index)
__global__ void __launch_bounds__(kBlock)
k_nearest_reduce_ties(const int32_t *__restrict__ pidx, const double *__restrict__ pd2, int64_t M, int nparts,
                      int32_t *__restrict__ out_idx, double *__restrict__ out_d2, const unsigned *__restrict__ xbits,
                      const int32_t *__restrict__ wild_pidx, const double *__restrict__ wild_pd2, int wild_nparts) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= M) return;
  if (xbits[1] != 0u) { pidx = wild_pidx; pd2 = wild_pd2; nparts = wild_nparts; }  // (the binary32 screen did the work)
  double best = std::numeric_limits<double>::infinity();
  int32_t bi = -1;
  for (int y = 0; y < nparts; y++) {
    const double d = pd2[(int64_t)y * M + j];
    const int32_t k = pidx[(int64_t)y * M + j];
    if (k >= 0 && (d < best || (d == best && (bi < 0 || k < bi)))) { best = d; bi = k; }
  }
  out_idx[j] = bi;
  if (out_d2) out_d2[j] = best;
}

// A look-up over the node range [index0, n) only, behind an answer for the nodes below it (the planner looks a round's
// targets up in the nodes a tree had a round ago while that round's tail runs, and in the nodes the round added afterwards):
// k_nearest_bound_min lowers every query's bound (from the range's own sample) to the distance already found;
// k_nearest_rebase turns the range's indices into node ids and keeps the earlier answer unless the range holds a strictly nearer node (lower indices win ties).
__global__ void __launch_bounds__(kBlock)
k_nearest_bound_min(double *__restrict__ bound2, const double *__restrict__ outer_d2, int64_t M) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j < M && outer_d2[j] < bound2[j]) bound2[j] = outer_d2[j];
}
__global__ void __launch_bounds__(kBlock)
k_nearest_rebase(int64_t M, int64_t index0, int have_range, int32_t *__restrict__ idx, double *__restrict__ d2, double *__restrict__ d2_out,
                 const int32_t *__restrict__ outer_idx, const double *__restrict__ outer_d2) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= M) return;
  int32_t k = have_range ? idx[j] : -1;
  double d = have_range ? d2[j] : std::numeric_limits<double>::infinity();
  if (k >= 0) k += (int32_t)index0;
  if (outer_idx && outer_idx[j] >= 0 && !(k >= 0 && d < outer_d2[j])) { k = outer_idx[j]; d = outer_d2[j]; }
  idx[j] = k;
  if (d2_out) d2_out[j] = d;
}

__global__ void __launch_bounds__(kBlock)
k_nearest_reduce(const int32_t *__restrict__ pidx, const double *__restrict__ pd2, int64_t M, int nchunks,
                 int32_t *__restrict__ out_idx, double *__restrict__ out_d2,
                 const int32_t *__restrict__ seed_idx = nullptr, const double *__restrict__ seed_d2 = nullptr,
                 const unsigned *__restrict__ only_if_wild = nullptr) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= M) return;
  if (only_if_wild && only_if_wild[1] == 0u) return;
  // (seed: the result over the nodes below the chunks' range -- lower indices, so it goes first)
  double best = seed_d2 ? seed_d2[j] : std::numeric_limits<double>::infinity();
  int32_t bi = seed_idx ? seed_idx[j] : -1;
  for (int y = 0; y < nchunks; y++) {
    const double d = pd2[(int64_t)y * M + j];
    if (d < best) { best = d; bi = pidx[(int64_t)y * M + j]; }
  }
  out_idx[j] = bi;
  if (out_d2) out_d2[j] = best;
}

}  // namespace
