// mjpl_nearest_cells.h -- the cell-ordered nearest-neighbour scan (round 6): what turns
// Tree.nearest_neighbor for a batch (reference src/mjpl/planning/tree.py:57-66) from "every query against every node"
// into "every query against the part of the tree it can have its answer in", with the answers of the full scan.
//
//   1. the nodes are sorted along a space-filling curve: a 32-bit key interleaves the bits of a node's cell coordinates,
//      every bit halving the column whose cells are widest at that point (k_nnc_plan), so cells are near-cubes;
//   2. every kNNCellSub consecutive sorted nodes -- a sub-chunk -- get their bounding box (binary32, rounded outwards);
//   3. the sample pass (k_nearest_mfma<NP, true>, over a strided sample of the SORTED nodes) gives every query its bound
//      -- the exact distance to some node -- and the sample position it was found at, which is a place on the curve;
//   4. the queries are sorted by that position: the 128 queries of a scan wave look at the same part of the tree;
//   5. k_nn_candidates marks, per wave, the sub-chunks whose box comes within SOME query's bound of that query
//      (point-to-box distance, per query -- a box over the wave's queries would be as wide as the joint ranges);
//   6. k_nearest_mfma<NP, false, true> scans the marked sub-chunks with the matrix-core screen in front of the exact
//      float64 distances, as the full scan does, and the reduction scatters the answers back to the callers' order.
// A sub-chunk left out holds no node within the bound of any of the wave's queries, and a query's answer lies within
// its bound: the winners (and, through the permutation, the lowest node id among equal distances) are the full scan's.
// Measured on the planner's trees (tools/nn_prune_study.py, profiles/README.md round 6): a wave scans 3 - 10 % of a tree.
#pragma once

#include <rocprim/device/device_radix_sort.hpp>

#include "mjpl_nearest.h"

namespace {

using namespace mjpl;

constexpr int kNNCKeyBits = 32;

// floats ordered like unsigned integers (for atomicMin / atomicMax over finite values of either sign)
__device__ __forceinline__ unsigned nnc_ord(float f) {
  const unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float nnc_unord(unsigned u) {
  return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}
__device__ __forceinline__ float nnc_down(double v) {  // the nearest binary32 not above v
  float f = (float)v;
  if ((double)f > v) f = nextafterf(f, -std::numeric_limits<float>::infinity());
  return f;
}
__device__ __forceinline__ float nnc_up(double v) {
  float f = (float)v;
  if ((double)f < v) f = nextafterf(f, std::numeric_limits<float>::infinity());
  return f;
}

// mm[c] = ord(min), mm[8 + c] = ord(max) of column c over the finite coordinates (initialised to ~0 / 0)
constexpr int kNNCMinmaxRows = 32;  // rows per thread
__global__ void __launch_bounds__(256)
k_nnc_minmax(const double *__restrict__ src, int64_t count, int64_t col_stride, int nplan, unsigned *__restrict__ mm) {
  __shared__ float part[2][4];
  const int64_t i0 = (int64_t)blockIdx.x * blockDim.x * kNNCMinmaxRows + threadIdx.x;
  for (int c = 0; c < nplan; c++) {
    float lo = std::numeric_limits<float>::infinity(), hi = -std::numeric_limits<float>::infinity();
    for (int k = 0; k < kNNCMinmaxRows; k++) {
      const int64_t i = i0 + (int64_t)k * blockDim.x;
      if (i < count) {
        const double v = src[(int64_t)c * col_stride + i];
        if (fabs(v) < 1e30) { lo = fminf(lo, nnc_down(v)); hi = fmaxf(hi, nnc_up(v)); }
      }
    }
    for (int o = 32; o > 0; o >>= 1) { lo = fminf(lo, __shfl_xor(lo, o)); hi = fmaxf(hi, __shfl_xor(hi, o)); }
    // (one pair of atomics per workgroup and column: per wave they were the kernel's whole time)
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { part[0][threadIdx.x >> 6] = lo; part[1][threadIdx.x >> 6] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
      lo = fminf(fminf(part[0][0], part[0][1]), fminf(part[0][2], part[0][3]));
      hi = fmaxf(fmaxf(part[1][0], part[1][1]), fmaxf(part[1][2], part[1][3]));
      if (lo <= hi) { atomicMin(&mm[c], nnc_ord(lo)); atomicMax(&mm[8 + c], nnc_ord(hi)); }
    }
  }
}

// the key's plan: plan[b] = the column bit b (most significant first) halves -- always the column whose cells are widest;
// rng[c] = column minimum, rng[8 + c] = 1 / width (0: a column without extent), nb[c] = bits of column c
struct NncPlan {
  float lo[8], inv[8];
  int nb[8];
  int col[kNNCKeyBits];
  float ctr[8];  // the middle of every column's range: the screen's operands are packed relative to it (smaller magnitudes, a smaller allowance)
};
__global__ void k_nnc_plan(const unsigned *__restrict__ mm, int nplan, NncPlan *__restrict__ out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  float w[8];
  for (int c = 0; c < 8; c++) {
    out->nb[c] = 0;
    out->lo[c] = 0;
    out->inv[c] = 0;
    out->ctr[c] = 0;
    w[c] = -1.0f;
    if (c < nplan && mm[c] != 0xffffffffu) {
      const float lo = nnc_unord(mm[c]), hi = nnc_unord(mm[8 + c]);
      out->lo[c] = lo;
      out->ctr[c] = 0.5f * lo + 0.5f * hi;
      w[c] = hi - lo;
      out->inv[c] = w[c] > 0 ? 1.0f / w[c] : 0.0f;
    }
  }
  for (int b = 0; b < kNNCKeyBits; b++) {
    int best = 0;
    for (int c = 1; c < 8; c++)
      if (w[c] > w[best]) best = c;
    out->col[b] = best;
    out->nb[best]++;
    w[best] *= 0.5f;
  }
}

__global__ void __launch_bounds__(256)
k_nnc_keys(const double *__restrict__ src, int64_t count, int64_t col_stride, int nplan, const NncPlan *__restrict__ plan,
           unsigned *__restrict__ keys, int32_t *__restrict__ iota) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  unsigned u[8], left[8];
  for (int c = 0; c < 8; c++) {
    u[c] = 0;
    left[c] = (unsigned)plan->nb[c];
    if (c < nplan && left[c] > 0) {
      const double v = src[(int64_t)c * col_stride + i];
      const float cells = (float)(1u << left[c]);
      float x = ((float)v - plan->lo[c]) * plan->inv[c] * cells;  // (NaN / inf: the comparisons below send them to an end)
      x = x > 0.0f ? x : 0.0f;
      x = x < cells - 1.0f ? x : cells - 1.0f;
      u[c] = (unsigned)x;
    }
  }
  unsigned key = 0;
  for (int b = 0; b < kNNCKeyBits; b++) {
    const int c = plan->col[b];
    left[c]--;
    key = (key << 1) | ((u[c] >> left[c]) & 1u);
  }
  keys[i] = key;
  iota[i] = (int32_t)i;
}

// rows in sorted order, eight doubles each -- one cache line per node for the exact distances: dst[j][c] = src[c][perm[j]]
// (rows beyond `count` up to `padded`: +inf -- a node that never wins; slot 7: 0)
__global__ void __launch_bounds__(256)
k_nnc_gather(const double *__restrict__ src, int64_t col_stride, const int32_t *__restrict__ perm, int64_t count, int64_t padded,
             int nplan, double *__restrict__ dst, float *__restrict__ dst32, const NncPlan *__restrict__ plan) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= padded) return;
  const int64_t i = j < count ? perm[j] : -1;
  double row[8];
  for (int c = 0; c < 8; c++)
    row[c] = c < nplan ? (i >= 0 ? src[(int64_t)c * col_stride + i] : std::numeric_limits<double>::infinity()) : 0.0;
  double2 *out = reinterpret_cast<double2 *>(dst + 8 * j);
  for (int k = 0; k < 4; k++) out[k] = double2{row[2 * k], row[2 * k + 1]};
  // ... and the same row in binary32 (32 bytes), relative to the columns' middles like the screen's operands: the parked
  // pairs' second screen (k_nearest_mfma: work_off) reads these
  float r32[8];
  for (int c = 0; c < 8; c++) r32[c] = c < nplan ? (float)(row[c] - (double)plan->ctr[c]) : 0.0f;
  float4 *o32 = reinterpret_cast<float4 *>(dst32 + 8 * j);
  o32[0] = float4{r32[0], r32[1], r32[2], r32[3]};
  o32[1] = float4{r32[4], r32[5], r32[6], r32[7]};
}

// the box of every sub-chunk of the sorted nodes: nbox[c][s] = min, nbox[8 + c][s] = max (columns beyond nplan: -inf / +inf)
__global__ void __launch_bounds__(256)
k_nnc_boxes(const double *__restrict__ nodes_s, int64_t n, int nplan, int nsub, int nsubp, float *__restrict__ nbox) {
  const int s = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (s >= nsub) return;
  for (int c = 0; c < 8; c++) {
    float lo = std::numeric_limits<float>::infinity(), hi = -std::numeric_limits<float>::infinity();
    if (c < nplan) {
      for (int k = lane; k < kNNCellSub; k += 64) {
        const int64_t i = (int64_t)s * kNNCellSub + k;
        if (i < n) {
          const double v = nodes_s[8 * i + c];  // (rows of eight)
          // (a NaN never wins and makes the call "wild" anyway; an infinite coordinate opens the box to that side)
          if (v == v) { lo = fminf(lo, nnc_down(v)); hi = fmaxf(hi, nnc_up(v)); }
        }
      }
      for (int o = 32; o > 0; o >>= 1) { lo = fminf(lo, __shfl_xor(lo, o)); hi = fmaxf(hi, __shfl_xor(hi, o)); }
    } else {
      lo = -std::numeric_limits<float>::infinity();
      hi = std::numeric_limits<float>::infinity();
    }
    if (lane == 0) { nbox[(int64_t)c * nsubp + s] = lo; nbox[(int64_t)(8 + c) * nsubp + s] = hi; }
  }
}

// a bound from an earlier answer (a ranged look-up) under the sample's: the smaller of the two, the sample position kept
__global__ void __launch_bounds__(kBlock)
k_nnc_bound_min(double *__restrict__ bound2, const double *__restrict__ outer_d2, int64_t M, const unsigned *__restrict__ xbits) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= M) return;
  if (xbits[1] != 0u) {  // (wild coordinates: the float64 sample scan wrote plain bounds)
    if (outer_d2[j] < bound2[j]) bound2[j] = outer_d2[j];
    return;
  }
  if (outer_d2[j] < bound2[j]) {
    const unsigned long long old = (unsigned long long)__double_as_longlong(bound2[j]);
    unsigned long long w = (unsigned long long)__double_as_longlong(outer_d2[j]);
    w = ((w + 0xffffull) & ~0xffffull) | (old & 0xffffull);
    bound2[j] = __longlong_as_double((long long)w);
  }
}

__global__ void __launch_bounds__(kBlock)
k_nnc_query_keys(const double *__restrict__ bound2, int64_t M, unsigned *__restrict__ keys, int32_t *__restrict__ iota) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= M) return;
  keys[j] = (unsigned)((unsigned long long)__double_as_longlong(bound2[j]) & 0xffffull);
  iota[j] = (int32_t)j;
}

// the queries in scan order: float64 columns (the exact distances), their bounds, and a binary32 row per query for the
// candidate pass: [x (7, columns beyond nplan: 0), the bound rounded up with the pass's own allowance]
__global__ void __launch_bounds__(kBlock)
k_nnc_gather_queries(const double *__restrict__ src, const double *__restrict__ bound2, const int32_t *__restrict__ perm, int64_t M,
                     int64_t Mpad, int nplan, double *__restrict__ dst, double *__restrict__ bound2_s, float *__restrict__ qf,
                     float *__restrict__ q32c, const NncPlan *__restrict__ plan) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= Mpad) return;
  float row[8] = {0, 0, 0, 0, 0, 0, 0, -1.0f};  // (a padding row: no sub-chunk is a candidate for it)
  float rowc[8] = {0, 0, 0, 0, 0, 0, 0, 0};     // the row relative to the columns' middles (the second screen's)
  if (j < M) {
    const int64_t i = perm[j];
    float amax = 0;
    for (int c = 0; c < 8; c++) {
      const double v = c < nplan ? src[(int64_t)c * M + i] : 0.0;
      dst[8 * j + c] = v;  // (rows of eight, as the nodes)
      if (c < nplan) {
        row[c] = (float)v;
        rowc[c] = (float)(v - (double)plan->ctr[c]);
        amax = fmaxf(amax, fabsf(row[c]));
      }
    }
    const double b2 = bound2[i];
    bound2_s[j] = b2;
    // The pass computes the squared distance from the binary32 query to a box rounded outwards: the query moves by
    // at most 2^-24 |x| per column, so the distance by at most e = sqrt(nplan) 2^-24 max|x|, and the binary32 sum of
    // squares loses at most 2^-20 of itself: a box is a candidate when that sum <= (sqrt(b2) + e)^2 (1 + 2^-18).
    if (b2 < std::numeric_limits<double>::infinity()) {
      const double e = sqrt((double)nplan) * 0x1p-24 * (double)amax;
      const double r = sqrt(b2) + e;
      row[7] = nnc_up(r * r * (1.0 + 0x1p-18));
    } else {
      row[7] = std::numeric_limits<float>::infinity();
    }
  }
  for (int k = 0; k < 8; k++) { qf[8 * j + k] = row[k]; q32c[8 * j + k] = rowc[k]; }
}

// Every query's bound, tightened where its answer is likely to be: the exact float64 distances to the nodes of its HOME
// sub-chunk -- the one its sample node lies in (the bound's low 16 bits say where on the curve) -- and the two beside it.
// In a dense tree the strided sample's nearest node is several nodes' spacings away while thousands lie nearer: with the
// sample's bound alone a connect-phase look-up parked 7 10^8 pairs (every row of the scan's grid works its way down from
// the same loose bound on its own); the home nodes bring the bound to the answer's neighbourhood before anything is
// scanned, and the candidate lists shrink with it.  One lane per sorted query -- neighbours on the curve, so a wave's
// lanes read the same rows -- and the bound of the candidate pass (qf row, slot 7) is made here from the result.
__global__ void __launch_bounds__(256)
k_nnc_home(const double *__restrict__ nodes_s, int64_t n, const double *__restrict__ queries_s, int64_t M, int nplan,
           int64_t mstride, int idx_shift, int nsub, double *__restrict__ bound2_s, float *__restrict__ qf,
           const unsigned *__restrict__ xbits) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= M || xbits[1] != 0u) return;
  double q[8];
  const double2 *qp = reinterpret_cast<const double2 *>(queries_s + 8 * j);
  for (int k = 0; k < 4; k++) { const double2 v = qp[k]; q[2 * k] = v.x; q[2 * k + 1] = v.y; }
  const double b0 = bound2_s[j];
  double best = b0;
  if (b0 < std::numeric_limits<double>::infinity()) {
    const unsigned long long word = (unsigned long long)__double_as_longlong(b0);
    const int64_t pos = (int64_t)((word & 0xffffull) << idx_shift) * mstride;
    int sc = (int)(pos / kNNCellSub);
    sc = sc < nsub ? sc : nsub - 1;
    const int64_t lo = (int64_t)(sc > 0 ? sc - 1 : 0) * kNNCellSub;
    int64_t hi = (int64_t)(sc + 2) * kNNCellSub;
    hi = hi < n ? hi : n;
    for (int64_t i = lo; i < hi; i++) {
      const double2 *np2 = reinterpret_cast<const double2 *>(nodes_s + 8 * i);
      double x[8];
      for (int k = 0; k < 4; k++) { const double2 v = np2[k]; x[2 * k] = v.x; x[2 * k + 1] = v.y; }
      double ex = 0;
      for (int c = 0; c < 7; c++) {  // (the statements of the exact distance; slots beyond nplan hold zeros on both sides)
        const double d = x[c] - q[c];
        ex = ex + d * d;
      }
      best = ex < best ? ex : best;
    }
    bound2_s[j] = best;
  }
  // the candidate pass's bound (see k_nnc_gather_queries for the allowance)
  float amax = 0;
  for (int c = 0; c < nplan; c++) amax = fmaxf(amax, fabsf((float)q[c]));
  float r2 = std::numeric_limits<float>::infinity();
  if (best < std::numeric_limits<double>::infinity()) {
    const double e = sqrt((double)nplan) * 0x1p-24 * (double)amax;
    const double r = sqrt(best) + e;
    r2 = nnc_up(r * r * (1.0 + 0x1p-18));
  }
  qf[8 * j + 7] = r2;
}

// The candidate pass.  grid (waves of 128 sorted queries, ceil(nwords / 4)), 256 threads: wave w of the workgroup tests the
// 64 sub-chunks of mask word 4 blockIdx.y + w -- one per lane, its box in registers -- against the 128 queries one after
// the other (their rows come through the scalar cache: the address is the wave's), and writes the word.
__global__ void __launch_bounds__(256)
k_nn_candidates(const float *__restrict__ qf, int64_t M, const float *__restrict__ nbox, int nsub, int nsubp, int nwords,
                unsigned long long *__restrict__ masks, const unsigned *__restrict__ xbits) {
  if (xbits[1] != 0u) return;  // (wild coordinates: the binary32 screen serves the call)
  const int word = __builtin_amdgcn_readfirstlane((int)(blockIdx.y * 4 + (threadIdx.x >> 6)));
  if (word >= nwords) return;
  const int lane = threadIdx.x & 63;
  const int sc = word * 64 + lane;
  float nlo[7], nhi[7];
#pragma unroll
  for (int c = 0; c < 7; c++) {
    nlo[c] = sc < nsub ? nbox[(int64_t)c * nsubp + sc] : std::numeric_limits<float>::infinity();
    nhi[c] = sc < nsub ? nbox[(int64_t)(8 + c) * nsubp + sc] : std::numeric_limits<float>::infinity();
  }
  const int64_t q0 = (int64_t)blockIdx.x * 128;
  bool any = false;
  for (int k = 0; k < 128; k++) {
    const float *row = qf + 8 * (q0 + k);  // (padded to whole workgroups of the scan: always there)
    float lb2 = 0;
#pragma unroll
    for (int c = 0; c < 7; c++) {
      const float x = row[c];
      const float gap = fmaxf(fmaxf(nlo[c] - x, x - nhi[c]), 0.0f);
      lb2 = __builtin_fmaf(gap, gap, lb2);
    }
    any = any || lb2 <= row[7];
  }
  const unsigned long long m = __ballot(any && sc < nsub);
  if (lane == 0) masks[(int64_t)blockIdx.x * nwords + word] = m;
}

// the candidate bits of every wave as a list of sub-chunk numbers, ascending (what the scan walks): one wave per row of masks
__global__ void __launch_bounds__(256)
k_nn_compact(const unsigned long long *__restrict__ masks, int nwords, int nwaves, int32_t *__restrict__ list, int64_t list_pitch,
             int32_t *__restrict__ count, const unsigned *__restrict__ xbits) {
  if (xbits[1] != 0u) return;
  const int wq = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
  if (wq >= nwaves) return;
  const int lane = threadIdx.x & 63;
  int32_t *mine = list + (int64_t)wq * list_pitch;
  int base = 0;
  for (int word = 0; word < nwords; word++) {
    const unsigned long long m = masks[(int64_t)wq * nwords + word];
    if ((m >> lane) & 1ull) mine[base + __popcll(m & ((1ull << lane) - 1ull))] = word * 64 + lane;
    base += __popcll(m);
  }
  if (lane == 0) count[wq] = base;
}

// the partial results of the scan (sorted query order, node ids already the callers') -> the callers' query order
__global__ void __launch_bounds__(kBlock)
k_nnc_reduce(const int32_t *__restrict__ pidx, const double *__restrict__ pd2, int64_t M, int nparts, const int32_t *__restrict__ perm_q,
             int32_t *__restrict__ out_idx, double *__restrict__ out_d2, const unsigned *__restrict__ xbits,
             const int32_t *__restrict__ wild_pidx, const double *__restrict__ wild_pd2, int wild_nparts) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= M) return;
  int64_t to = perm_q[j];
  if (xbits[1] != 0u) { pidx = wild_pidx; pd2 = wild_pd2; nparts = wild_nparts; to = j; }  // (the binary32 screen did the work, in the callers' order)
  double best = std::numeric_limits<double>::infinity();
  int32_t bi = -1;
  for (int y = 0; y < nparts; y++) {
    const double d = pd2[(int64_t)y * M + j];
    const int32_t k = pidx[(int64_t)y * M + j];
    if (k >= 0 && (d < best || (d == best && (bi < 0 || k < bi)))) { best = d; bi = k; }
  }
  out_idx[to] = bi;
  if (out_d2) out_d2[to] = best;
}

}  // namespace
