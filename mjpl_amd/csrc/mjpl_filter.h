// mjpl_filter.h -- the float32 filter kernels of the edge / configuration pipeline and what they
// share (LDS carve, per-lane configuration check, waypoint expansion), as templates on a `Spec`
// policy: void = the interpreter of mjpl_device.h walks the compiled model program; a generated
// struct (tools: mjpl_amd/specialise.py) supplies the same per-configuration check as straight-line
// code for ONE model -- the same kernels are then instantiated in that model's own library
// (libmjpl_spec_<hash>.so) and launched by the engine instead of the generic ones.
#pragma once

#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "../../include/mjpl_hip.h"
#include "mjpl_device.h"

namespace mjpl {

constexpr int kBlock = 256;           // threads per workgroup: 4 wavefronts
constexpr int kFilterBlock = 256;     // queued filter kernels: four wavefronts share one LDS table copy
constexpr int kStatusNonFinite = 1;
// version of the contract between libmjpl_hip.so and a per-model specialised library
// (kernel signatures of this header + table layouts of mjpl_device.h)
#define MJPL_SPEC_ABI 14
// scene-generic specialised libraries: cull rows per moving geom, moving geoms at most,
// floats of the scene header in front of the rows
// Table: [header | per moving geom: kSceneRows cull rows of 4 floats, then kSceneRows descriptor words | one
// spare chunk of four rows (the code fetches two rows ahead)].  Rows 0 and 1 of a geom are its plane partners.
constexpr int kSceneRows = 32, kScenePlaneRows = 2, kSceneMaxStages = 24, kSceneHeader = 32;
constexpr int kSceneStageFloats = kSceneRows * 4 + kSceneRows;
inline size_t scene_floats(int nstage) { return (size_t)kSceneHeader + (size_t)nstage * kSceneStageFloats + 16; }
// digest of the headers both sides are built from (mjpl_amd/build.py: src_stamp); 0 = built by hand
#ifndef MJPL_SRC_STAMP
#define MJPL_SRC_STAMP 0ull
#endif
#ifndef MJPL_MBOX_WAVES
#define MJPL_MBOX_WAVES 1  // the 24-slot moving-box build: ~360 VGPRs; bound to two waves per SIMD it spills 200 dwords and is 4x slower
#endif
// A model's own straight-line code is asked to fit three waves per SIMD (168 VGPRs): its register
// pressure is a few registers above that without the bound, and the third wave is worth more
#ifndef MJPL_SPEC_WAVES
#define MJPL_SPEC_WAVES 3
#endif
template <class Spec, int MAXS = 0> constexpr int kMinWaves = std::is_void<Spec>::value ? (MAXS == 24 ? MJPL_MBOX_WAVES : 1) : MJPL_SPEC_WAVES;

// LDS carve shared by all kernels: [tables (A/B build only) | float64 columns | pose saves].
// With the default build the tables stay in global memory (scalar loads).
template <class T>
struct Carve {
  double *col0, *col1;
  T *save;
  char *qmem;  // per-wave candidate queues (queued kernels), after the pose saves
  T *ltab;     // queued kernels: the workgroup's LDS copy of the constant table (drain gathers)
  IP ip;
  typename Real<T>::Tab tp;
};

// queued narrowphase: the float32 filter, except in the one general build for models that keep
// more than kQueuedMaxSlots geoms in the slot file (immediate interpreter)
constexpr int kQueuedMaxSlots = 24;
template <class T, int MAXS>
constexpr bool kQueued = !Real<T>::exact && MAXS <= kQueuedMaxSlots;

template <class T, bool MBOX>
__device__ __forceinline__ WaveQueue<T, MBOX> wave_queue(char *qmem) {
  char *base = qmem + (size_t)(threadIdx.x >> 6) * WaveQueue<T, MBOX>::bytes();
  WaveQueue<T, MBOX> wq;
  wq.carve(base);
  return wq;
}

template <class T, int MAXS, bool WBOX, bool MBOX, class Spec = void, class QT = double>
__device__ __forceinline__ int check_one(const Carve<T> &c, const QT *q, int B, bool active, T tol,
                                         int64_t row, const UndecidedConfigs &uc = UndecidedConfigs{},
                                         int idx = 0, const int *item_edge = nullptr,
                                         const int *item_idx = nullptr, const EdgeSource &src = EdgeSource{}) {
  const int qstride = B;  // q[k * qstride]: an LDS column slice, like the pose saves
#ifdef MJPL_X_NOCHECK  // timing-only build: what the item kernel costs around the check
  if (item_idx) return V_NONE;
#endif
  PatchSink ps;
  ps.uc = uc;
  // where the owner's configuration is found when a pair is handed to the exact re-check: the
  // float64 LDS columns -- or, with `src` (float32 columns hold a rounded copy), the caller's
  // rows / the recurrence from them
  ps.qcol = c.col0 + (threadIdx.x & ~63);
  ps.B = B;
  ps.L = 1;
  ps.nplan = c.ip[H_NPLAN];
  ps.idx = idx;
  ps.item_edge = item_edge;
  ps.item_idx = item_idx;
  ps.src = src;
  ps.perm = c.ip + c.ip[H_OFF_PERM];
  if constexpr (kQueued<T, MAXS>) {
    WaveQueue<T, MBOX> wq = wave_queue<T, MBOX>(c.qmem);
    if constexpr (!std::is_void<Spec>::value)  // this model's own straight-line check (same contract)
      return Spec::run(c.tp, c.ltab, q, qstride, c.save + threadIdx.x, B, active, tol, wq, (int)row, ps);
    else
      return run_config_queued<T, MAXS, WBOX, MBOX>(c.ip, c.tp, c.ltab, q, qstride, c.save + threadIdx.x, B, active,
                                                    tol, wq, (int)row, ps);
  } else {
    FkOut none = {};
    if constexpr (Real<T>::exact) ps.uc.count = nullptr;  // (the exact path decides everything itself)
    return run_config<T, MAXS, false, WBOX, MBOX>(c.ip, c.tp, q, qstride, c.save + threadIdx.x, B, active, tol,
                                                  none, row, ps);
  }
}

template <class T>
__device__ __forceinline__ Carve<T> carve_lds(double *smem, const int *__restrict__ gip, int nip,
                                              const T *__restrict__ gtp, int ntp, int nplan, int ncolsets,
                                              int B, size_t colscalar = sizeof(double),
                                              size_t qbytes = WaveQueue<T>::bytes()) {
  Carve<T> c;
#if MJPL_TABLES_LDS
  // A/B build: [control words | constants | columns | saves | queues | ...], tables first so that
  // their place does not depend on what the kernel carves behind the saves
  int *il = reinterpret_cast<int *>(smem);
  T *tl = reinterpret_cast<T *>(smem + (nip * sizeof(int) + 7) / 8);
  const int tdoubles = (int)((nip * sizeof(int) + 7) / 8 + (ntp * sizeof(T) + 7) / 8);
  c.col0 = smem + tdoubles;
#else
  c.col0 = smem;
#endif
  c.col1 = c.col0 + (size_t)nplan * B;
  // (float32 filter kernels keep binary32 columns: a third workgroup fits a CU's LDS with them)
  const size_t colbytes = ((size_t)ncolsets * nplan * B * colscalar + 7) & ~(size_t)7;
  c.save = reinterpret_cast<T *>(reinterpret_cast<char *>(c.col0) + colbytes);
  const int nsave = gip[H_NSAVE];
  // queue memory starts 8-byte aligned after the saves
  c.qmem = reinterpret_cast<char *>(c.col0) +
           ((colbytes + (size_t)nsave * 7 * B * sizeof(T) + 7) & ~(size_t)7);
  // queued kernels: constant-table copy behind the queues (staged below, before the barrier)
  c.ltab = reinterpret_cast<T *>(c.qmem + (size_t)(B / 64) * qbytes);  // (qbytes: one wave's queues)
  if (!Real<T>::exact)
    for (int k = threadIdx.x; k < ntp; k += blockDim.x) c.ltab[k] = gtp[k];
#if MJPL_TABLES_LDS
  for (int k = threadIdx.x; k < ntp; k += blockDim.x) tl[k] = gtp[k];
  for (int k = threadIdx.x; k < nip; k += blockDim.x) il[k] = gip[k];
  c.ip = il;
  c.tp = tl;
#else
  c.ip = (IP)gip;
  c.tp = (typename Real<T>::Tab)gtp;
#endif
  return c;
}

// Number of planning columns when the kernel is built for one model (0: read from the program).
// With it the per-column loops below unroll and their loads go out together.
template <class Spec> struct StaticNplan { static constexpr int value = Spec::kNplan; };
template <> struct StaticNplan<void> { static constexpr int value = 0; };

// f(k, Q[row i][k]) for all planning columns, eight independent loads at a time (a plain loop
// over a run-time nplan waits out one memory round trip per column)
template <class F>
__device__ __forceinline__ void for_row(const double *__restrict__ Q, int64_t N, int64_t i, int nplan,
                                        int layout, bool on, F &&f) {
  for (int k0 = 0; k0 < nplan; k0 += 8) {
    double v[8];
#pragma unroll
    for (int j = 0; j < 8; j++)
      v[j] = (on && k0 + j < nplan) ? ((layout == MJPL_SOA) ? Q[(int64_t)(k0 + j) * N + i] : Q[i * nplan + k0 + j]) : 0.0;
#pragma unroll
    for (int j = 0; j < 8; j++)
      if (k0 + j < nplan) f(k0 + j, v[j]);
  }
}
// the same over two arrays of one shape: f(k, A[i][k], B[i][k])
template <class F>
__device__ __forceinline__ void for_rows(const double *__restrict__ A, const double *__restrict__ Bq, int64_t N,
                                         int64_t i, int nplan, int layout, bool on, F &&f) {
  for (int k0 = 0; k0 < nplan; k0 += 8) {
    double a[8], b[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const bool in = on && k0 + j < nplan;
      const int64_t at = (layout == MJPL_SOA) ? (int64_t)(k0 + j) * N + i : i * nplan + k0 + j;
      a[j] = in ? A[at] : 0.0;
      b[j] = in ? Bq[at] : 0.0;
    }
#pragma unroll
    for (int j = 0; j < 8; j++)
      if (k0 + j < nplan) f(k0 + j, a[j], b[j]);
  }
}

// planning columns of configuration i -> this lane's LDS column slice
template <class QT>
__device__ __forceinline__ void load_columns(QT *col, int B, const double *__restrict__ Q,
                                             int64_t N, int64_t i, int nplan, int layout, bool active) {
  for_row(Q, N, i, nplan, layout, active, [&](int c, double v) { col[c * B] = (QT)v; });
}


// ---- lane-per-waypoint interior pass --------------------------------------------------------
// The walking kernel gives every surviving edge one lane for all of its waypoints: at config-3
// size that is 2 300 waves for 1 024 SIMDs, each alive for the whole kernel.  Here the waypoints
// themselves become the work items: the endpoint kernel walks the reference's recurrence for
// every edge whose endpoint passed (float64, the statements of edge_body) and writes the interior
// waypoints into a dense item buffer; k_filter_items checks one waypoint per lane.  Consecutive
// items are consecutive waypoints of one edge, so the lanes of a wave see similar poses and pass
// the same bounding culls.  Edges with many waypoints stay with the walking kernel: `llist`.
struct ItemBuffers {
  int *edge, *idx;  // [cap] which edge, which check index (1..K)
  // The item space is split into `regions` equal parts of `regcap` slots (a multiple of the block
  // size), each with its own fill counter count[r * kCounterStride]: a workgroup reserves in region
  // blockIdx % regions.  One shared counter would take a returning atomic from every wave of the
  // endpoint kernel on one address -- at 4 096 waves that serialisation was a fifth of the kernel.
  int *count;       // items reserved, per region (may exceed regcap: the surplus went to llist)
  int cap;          // regions * regcap
  int regions, regcap;
  int *scount;      // survivors per region (statistic: edges reaching the interior pass)
  int *llist, *lcount;  // edges left to the walking kernel
  int kmax;             // edges with more interior waypoints than this stay with the walking kernel
  double *tstep;        // [E] step / |QB - QA| of the edges that have items
  double *ckpt;         // [cap][nplan] exact waypoints of the items with idx % kCkptEvery == 0 (launches
                        // with kmax >= kCkptEvery; else null): where exact_waypoint starts from
  int *claim;           // [E] per-edge claim word (see k_filter_items)
  int gen;              // this launch's generation: claim[edge] == gen <=> the edge is in ulist already
};
constexpr int kCounterStride = 32;    // ints between device counters: one 128-byte line each
constexpr int kItemRegions = 32;
// device counters of one launch: five scalars (edge-level undecided list, undecided pairs, edges whose
// endpoint passed, unused, edges left to the walking kernel), the per-region item fills, the per-region
// survivor counts.  The engine keeps two such sets and alternates: the first kernel of a launch clears
// the set the NEXT launch will use (nobody touches it meanwhile), which saves a fill kernel per launch.
constexpr int kNumCounters = 5 + 4 * kItemRegions + 2;
constexpr int kCtrTailDone = 5 + 4 * kItemRegions;  // k_tail: walking workgroups that are through
constexpr int kCtrCertified = 5 + 4 * kItemRegions + 1;  // fused kernel: surviving edges its certificate spared the waypoint checks
// tile queues of the persistent kernels (k_filter_endpoints_pw / k_filter_items_pw), one per region:
// the next tile of that region to hand out
constexpr int kCtrEndpointTiles = 5 + 2 * kItemRegions;
constexpr int kCtrItemTiles = 5 + 3 * kItemRegions;
__device__ __forceinline__ void zero_counters(int *next) {
  if (next && blockIdx.x == 0)
    for (int k = threadIdx.x; k < kNumCounters * kCounterStride; k += blockDim.x) next[k] = 0;
}
constexpr int kExpandMinWaypoints = 24;  // kmax is at least this; small batches get more (item space / E)

// One step of the recurrence: _step(w, QB, step)  (planning/utils.py:182-185; the statements of
// edge_body) on this lane's LDS rows qw (waypoint, stride ws) and qe (QB, stride B); returns true
// when the walk has arrived at QB.  `degenerate`: the squared distance under- or overflowed.
template <class Perm>
__device__ __forceinline__ bool walk_step(Perm perm, int nplan, const double *qe, int B, double *qw, int ws,
                                          double step, bool &degenerate) {
  double s = 0;
  for (int k = 0; k < nplan; k++) {
    const int col = perm[k];
    double d = qe[col * B] - qw[col * ws];
    s = s + d * d;
  }
  const double mag = sqrt(s);
  degenerate = degenerate || !(mag > 0.0) || !(mag <= 1.79769313486231570815e+308);
  const double sm = step < mag ? step : mag;
  // The nplan quotients d / mag share their divisor.  The compiler expands each division into
  // div_scale x2, rcp, two Newton steps on the reciprocal, a product, a remainder, div_fmas,
  // div_fixup; with every operand well inside the normal range (no scaling, no special case)
  // that is  r = refined rcp(mag);  q0 = d r;  q = fma(fma(-mag, q0, d), r, q0)  -- the same
  // operations on the same values, with r computed once.  Anything else divides as usual.
  bool plain = !(mag >= 0x1p-400 && mag <= 0x1p400);
  for (int k = 0; k < nplan; k++) {
    const double ad = fabs(qe[k * B] - qw[k * ws]);
    plain = plain || !(ad == 0.0 || ad >= 0x1p-400);
  }
#ifdef MJPL_X_PLAINDIV  // timing-only build: every quotient by the compiler's division
  plain = true;
#endif
  bool eq = true;
  if (__ballot(plain) != 0ull) {  // (wave-uniform: a per-lane select would compute both)
    for (int k = 0; k < nplan; k++) {
      const double ek = qe[k * B];
      const double d = ek - qw[k * ws];
      const double nw = qw[k * ws] + (d / mag) * sm;
      qw[k * ws] = nw;
      eq = eq && (nw == ek);
    }
  } else {
    double r = __builtin_amdgcn_rcp(mag);
    r = fma(r, fma(-mag, r, 1.0), r);
    r = fma(r, fma(-mag, r, 1.0), r);
    for (int k = 0; k < nplan; k++) {
      const double ek = qe[k * B];
      const double d = ek - qw[k * ws];
      const double q0 = d * r;
      double quo = fma(fma(-mag, q0, d), r, q0);
      quo = (d == 0.0) ? d : quo;  // (+-0 / mag keeps its sign)
      const double nw = qw[k * ws] + quo * sm;
      qw[k * ws] = nw;
      eq = eq && (nw == ek);
    }
  }
  return eq;
}

// How many interior waypoints does edge i have?  Walks the recurrence (`todo` lanes; qe: the
// lane's QB row in LDS, qw: scratch for the walking waypoint).  Returns the count, or -1 for an
// edge that stays with the walking kernel (longer than kmax waypoints, or degenerate: that kernel
// reports it).  Leaves step / |QB - QA| in ib.tstep[i] for k_filter_items.
// The items carry (edge, index) only.  The reference's waypoint idx is idx steps of length
// `step` from QA towards QB, each with a freshly computed direction: it lies within a few
// idx * 2^-53 |q| of QA + idx * (step / |QB - QA|) * (QB - QA), which is what k_filter_items tests
// (its float32 bounds absorb 1e-5; the exact re-check rebuilds the waypoint by the recurrence).
// So the walk is needed for the COUNT only -- where `step` divides the edge length the count
// hangs on the last bit of the running distance -- and no waypoint is stored.
__device__ __forceinline__ int count_waypoints_walk(const int *__restrict__ gip, double step, bool todo, bool at_end,
                                                    const double *qe, int B, double *qw, int ws, int kmax, int nplan,
                                                    double &tstep);
__device__ __forceinline__ int count_waypoints_ts(const int *__restrict__ gip, const double *__restrict__ QA,
                                                  int64_t E, int64_t i, double step, int layout, bool todo,
                                                  const double *qe, int B, double *qw, int ws,
                                                  int kmax, int nplan, double &tstep) {
  bool at_end = true;
  for_row(QA, E, i, nplan, layout, todo, [&](int k, double a) {
    qw[k * ws] = a;
    at_end = at_end && (a == qe[k * B]);
  });
  return count_waypoints_walk(gip, step, todo, at_end, qe, B, qw, ws, kmax, nplan, tstep);
}
// ... the walk itself: qw holds the lane's QA row on entry (at_end: QA == QB)
__device__ __forceinline__ int count_waypoints_walk(const int *__restrict__ gip, double step, bool todo, bool at_end,
                                                    const double *qe, int B, double *qw, int ws, int kmax, int nplan,
                                                    double &tstep) {
  const int *perm = gip + gip[H_OFF_PERM];
  double s0 = 0;
  for (int k = 0; k < nplan; k++) {
    const int col = perm[k];
    const double d = qe[col * B] - qw[col * ws];
    s0 = s0 + d * d;
  }
  tstep = 0.0;
  if (!todo || at_end) return 0;
  const bool long_edge = !(sqrt(s0) <= step * (double)(kmax - 2));
  int K = 0;
  bool degenerate = false;
  bool walking = !long_edge;
  while (__ballot(walking) != 0ull) {
    if (walking) {
      if (walk_step(perm, nplan, qe, B, qw, ws, step, degenerate)) walking = false;
      else if (++K > kmax) walking = false;
#ifdef MJPL_X_SHORTWALK  // timing-only build: one step of the walk
      walking = false;
#endif
      if (degenerate) { walking = false; K = kmax + 1; }
    }
  }
  if (long_edge || K > kmax) return -1;  // (the estimate can be off by a step)
  if (K > 0) tstep = step / sqrt(s0);
  return K;
}
__device__ __forceinline__ int count_waypoints(const int *__restrict__ gip, const double *__restrict__ QA,
                                               int64_t E, int64_t i, double step, int layout, bool todo,
                                               const double *qe, int B, double *qw, int ws,
                                               const ItemBuffers &ib, int nplan) {
  double ts;
  const int K = count_waypoints_ts(gip, QA, E, i, step, layout, todo, qe, B, qw, ws, ib.kmax, nplan, ts);
  if (K > 0) ib.tstep[i] = ts;
  return K;
}

// Turn the counts of a wave's surviving edges into items: one reservation per wave, lane l owns
// slots [base + sum_{l' < l} K_l', +K_l).  With checkpoints (ib.ckpt: few-edge launches with long
// items, path shortcutting) long edges walk once more through qe / qw and leave every
// kCkptEvery-th waypoint behind, so that rebuilding one never takes more than that many steps.
__device__ __forceinline__ void emit_items(const int *__restrict__ gip, const double *__restrict__ QA, int64_t E,
                                           int64_t i, double step, int layout, bool survive, int K,
                                           const double *qe, int B, double *qw, int ws, const ItemBuffers &ib,
                                           int nplan, int region_of_wave = -1) {
  const int lane = threadIdx.x & 63;
  bool done = !survive;
  if (!done && K < 0) {  // the walking kernel takes the edge
    ib.llist[atomicAdd(ib.lcount, 1)] = (int)i;
    done = true;
  }
  if (done) K = 0;
  int incl = K;
  for (int off = 1; off < 64; off <<= 1) {
    const int up = __shfl_up(incl, off);
    if (lane >= off) incl += up;
  }
  const int total = __shfl(incl, 63);
  if (total == 0) return;
  int base = 0;
  const int region = region_of_wave >= 0 ? region_of_wave : (int)(blockIdx.x % (unsigned)ib.regions);
  if (lane == 0) base = atomicAdd(ib.count + region * kCounterStride, total);
  base = __builtin_amdgcn_readfirstlane(base);
  const int rel = base + incl - K, first = region * ib.regcap + rel;
  if (!done && rel + K > ib.regcap) {  // out of item space: the walking kernel takes the edge
    ib.llist[atomicAdd(ib.lcount, 1)] = (int)i;
    for (int r = rel > 0 ? rel : 0; r < ib.regcap; r++) ib.edge[region * ib.regcap + r] = -1;  // reserved but void
    done = true;
  }
#ifdef MJPL_X_NOSTORE  // timing-only build: items reserved, not written
  if (K < 1000) return;
#endif
  if (!done)
    for (int idx = 1; idx <= K; idx++) {
      ib.edge[first + idx - 1] = (int)i;
      ib.idx[first + idx - 1] = idx;
    }
  if (ib.ckpt) {
    const bool longe = !done && K >= kCkptEvery;
    if (__ballot(longe) != 0ull) {
      const int *perm = gip + gip[H_OFF_PERM];
      for_row(QA, E, i, nplan, layout, longe, [&](int k, double a) { qw[k * ws] = a; });
      bool degenerate = false;
      for (int idx = 1; __ballot(longe && idx <= K) != 0ull; idx++) {
        if (longe && idx <= K) {
          walk_step(perm, nplan, qe, B, qw, ws, step, degenerate);
          if (idx % kCkptEvery == 0)
            for (int k = 0; k < nplan; k++) ib.ckpt[(size_t)(first + idx - 1) * nplan + k] = qw[k * ws];
        }
      }
    }
  }
}

// Endpoint pass of the two-pass edge filter: check 0 (the endpoint QB, utils.py:144) for every
// edge with full lanes; edges whose endpoint is free (or undecided) are appended to `slist` for
// the interior pass, so that pass runs only on edges that still need it -- in an RRT batch a
// large share of the candidate edges ends inside an obstacle, and in the one-pass kernel their
// lanes idle through every later waypoint of the wave.
template <class Spec, int MAXS, bool WBOX, bool MBOX>
__global__ void __launch_bounds__(kBlock, (kMinWaves<Spec, MAXS>))
k_filter_endpoints(const int *__restrict__ gip, int nip, const float *__restrict__ gfp, int nfp,
                   const double *__restrict__ QA, const double *__restrict__ QB, int64_t E, int layout,
                   float tol, uint8_t *__restrict__ valid, int32_t *__restrict__ first_bad,
                   int *__restrict__ status, int *__restrict__ ulist, int *__restrict__ ucount,
                   UndecidedConfigs uc, int *__restrict__ slist, int *__restrict__ scount, ItemBuffers ib,
                   double step, int *__restrict__ zero_next) {
  extern __shared__ double smem[];
  const int B = blockDim.x;
  zero_counters(zero_next);
  const int nplan = StaticNplan<Spec>::value ? StaticNplan<Spec>::value : gip[H_NPLAN];
  // (immediate interpreter + item expansion: a second column set holds the walking waypoint)
  // queued interpreter: binary32 columns for the check (what goes to the exact re-check is read
  // from QB again), and the walk below reuses the workgroup's columns, saves and queues
  constexpr bool kQ = kQueued<float, MAXS>;
  typedef typename std::conditional<kQ, float, double>::type QT;
  Carve<float> c = carve_lds<float>(smem, gip, nip, gfp, nfp, nplan, (!kQ && ib.count) ? 2 : 1, B, sizeof(QT),
                                    WaveQueue<float, MBOX>::bytes());
  const int64_t i = (int64_t)blockIdx.x * B + threadIdx.x;
  const bool active = i < E;
  bool finite = true;
  for_rows(QA, QB, E, i, nplan, layout, active, [&](int, double a, double b) {
    finite = finite && (fabs(a) <= 1.79769313486231570815e+308) && (fabs(b) <= 1.79769313486231570815e+308);
  });
  // Lane-per-waypoint interior pass, queued interpreter: the waypoint COUNT of every edge is
  // taken first (float64 rows QB / walking waypoint laid over the columns, saves and queues the
  // check is about to use).  Counting after the check instead would spare the edges whose
  // endpoint is in contact, but a wave walks as long as its longest edge either way, and it
  // would have to wait for the slowest check of the workgroup before it may reuse the memory;
  // here the waves arrive at the barrier together.  (With checkpoints -- few long edges -- the
  // count follows the check: the survivors walk twice.)
  const size_t idle = (size_t)(reinterpret_cast<char *>(c.ltab) - reinterpret_cast<char *>(c.col0));
  const bool fits = kQ && (size_t)2 * nplan * B * sizeof(double) <= idle;
#ifdef MJPL_X_LATECOUNT  // timing-only build: count after the check, as with checkpoints
  const bool early = false;
#else
  const bool early = kQ && ib.count && fits && !ib.ckpt;
#endif
  int K = 0;
  if (early) {
    double *qe = c.col0 + threadIdx.x;
    load_columns(qe, B, QB, E, i, nplan, layout, active);
    K = count_waypoints(gip, QA, E, i, step, layout, active && finite, qe, B, qe + (size_t)nplan * B, B, ib, nplan);
    __syncthreads();
  }
  QT *qw = reinterpret_cast<QT *>(c.col0) + threadIdx.x;
  // (a NaN / inf row never reaches the check, not even on a lane that only keeps company: zeros in its place --
  // a NaN passes the negated compare of the plane culls, and the lane would queue candidates)
  load_columns(qw, B, QB, E, i, nplan, layout, active && finite);
  __syncthreads();
  const bool run = active && finite;
  const int code = check_one<float, MAXS, WBOX, MBOX, Spec>(c, qw, B, run, tol, i, uc, 0, nullptr, nullptr,
                                                            kQ ? EdgeSource{QB, QB, E, layout, 0.0, nullptr} : EdgeSource{});
  bool survive = run && code != V_CONTACT;
  if (active) {
    if (!finite) {
      valid[i] = 0;
      if (first_bad) first_bad[i] = -2;
      atomicOr(status, kStatusNonFinite);
    } else if (code == V_CONTACT) {
      valid[i] = 0;
      if (first_bad) first_bad[i] = 0;
    } else {
      bool whole_edge = false;
      if (code == V_UNSURE) {  // the endpoint itself goes to the exact configuration kernel
        const int j = kQueued<float, MAXS> ? uc.cap : atomicAdd(uc.count, 1);  // (queued: the whole edge)
        if (j < uc.cap) {
          for (int k = 0; k < nplan; k++) uc.q[(size_t)j * nplan + k] = (double)qw[k * B];  // (float64 columns here)
          uc.edge[j] = (int)i;
          uc.idx[j] = 0;
          uc.ga[j] = uc.gb[j] = -1;
        } else {
          whole_edge = true;
        }
      }
      if (whole_edge) {
        ulist[atomicAdd(ucount, 1)] = (int)i;  // the exact edge kernel writes valid / first_bad
        survive = false;
      } else {
        valid[i] = 1;  // so far; the interior pass and the patch pass may clear it
        if (first_bad) first_bad[i] = -1;
      }
    }
  }
#ifdef MJPL_X_NOEXPAND  // timing-only build: the endpoint kernel without the waypoint count / the items
  return;
#endif
  const unsigned long long m = __ballot(survive);
  if (ib.count) {
    // lane-per-waypoint interior pass: emit this edge's interior waypoints as work items
    if constexpr (kQ) {
      if (early) {
        if (m == 0ull) return;
        emit_items(gip, QA, E, i, step, layout, survive, K, nullptr, B, nullptr, B, ib, nplan);
        if ((threadIdx.x & 63) == 0)
          atomicAdd(ib.scount + (blockIdx.x % (unsigned)ib.regions) * kCounterStride, (int)__builtin_popcountll(m));
        return;
      }
      // count after the check: the two float64 rows per lane take over the workgroup's columns,
      // saves and queues once every wave is through with its check
      __syncthreads();
      if (m == 0ull) return;
      if (fits) {
        double *qe = c.col0 + threadIdx.x;
        load_columns(qe, B, QB, E, i, nplan, layout, active);
        K = count_waypoints(gip, QA, E, i, step, layout, survive, qe, B, qe + (size_t)nplan * B, B, ib, nplan);
        emit_items(gip, QA, E, i, step, layout, survive, K, qe, B, qe + (size_t)nplan * B, B, ib, nplan);
        if ((threadIdx.x & 63) == 0)
          atomicAdd(ib.scount + (blockIdx.x % (unsigned)ib.regions) * kCounterStride, (int)__builtin_popcountll(m));
        return;
      }
    } else {
      if (m == 0ull) return;
      K = count_waypoints(gip, QA, E, i, step, layout, survive, qw, B, c.col1 + threadIdx.x, B, ib, nplan);
      emit_items(gip, QA, E, i, step, layout, survive, K, qw, B, c.col1 + threadIdx.x, B, ib, nplan);
      if ((threadIdx.x & 63) == 0) atomicAdd(scount, (int)__builtin_popcountll(m));
      return;
    }
    if (survive) ib.llist[atomicAdd(ib.lcount, 1)] = (int)i;  // too many columns: walking kernel
    return;
  }
  if (m == 0ull) return;
  const int lane = threadIdx.x & 63;
  int base = 0;
  if (lane == 0) base = atomicAdd(scount, (int)__builtin_popcountll(m));
  base = __builtin_amdgcn_readfirstlane(base);
  if (survive)
    slist[base + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u))] = (int)i;
}

template <class Spec, int MAXS, bool WBOX, bool MBOX>
__global__ void __launch_bounds__(kBlock, (kMinWaves<Spec, MAXS>))
k_filter_configs(const int *__restrict__ gip, int nip, const float *__restrict__ gfp, int nfp,
                 const double *__restrict__ Q, int64_t N, int layout, float tol,
                 uint8_t *__restrict__ valid, int *__restrict__ ulist, int *__restrict__ ucount,
                 UndecidedConfigs uc, int *__restrict__ zero_next) {
  extern __shared__ double smem[];
  const int B = blockDim.x;
  zero_counters(zero_next);
  const int nplan = StaticNplan<Spec>::value ? StaticNplan<Spec>::value : gip[H_NPLAN];
  constexpr bool kQ = kQueued<float, MAXS>;
  typedef typename std::conditional<kQ, float, double>::type QT;  // (see k_filter_endpoints)
  Carve<float> c = carve_lds<float>(smem, gip, nip, gfp, nfp, nplan, 1, B, sizeof(QT), WaveQueue<float, MBOX>::bytes());
  const int64_t i = (int64_t)blockIdx.x * B + threadIdx.x;
  const bool active = i < N;
  QT *qc = reinterpret_cast<QT *>(c.col0) + threadIdx.x;
  load_columns(qc, B, Q, N, i, nplan, layout, active);
  __syncthreads();
  // queued interpreter: undecided pairs go to k_patch_pairs through `uc` (which clears valid[i]
  // on a contact); V_UNSURE comes back only for what could not be handed over
  const int code = check_one<float, MAXS, WBOX, MBOX, Spec>(c, qc, B, active, tol, i, uc, 0, nullptr, nullptr,
                                                            kQ ? EdgeSource{Q, Q, N, layout, 0.0, nullptr} : EdgeSource{});
  if (active) {
    if (code == V_UNSURE) ulist[atomicAdd(ucount, 1)] = (int)i;
    else valid[i] = (code == V_CONTACT) ? 0 : 1;
  }
}


#ifdef MJPL_X_ITEMS_4WAVES
#define MJPL_ITEMS_BOUNDS __launch_bounds__(kBlock, 4)
#else
#define MJPL_ITEMS_BOUNDS __launch_bounds__(kBlock, (kMinWaves<Spec, MAXS>))
#endif
template <class Spec, int MAXS, bool WBOX, bool MBOX>
__global__ void MJPL_ITEMS_BOUNDS
k_filter_items(const int *__restrict__ gip, int nip, const float *__restrict__ gfp, int nfp, ItemBuffers ib,
               EdgeSource src, float tol, uint8_t *__restrict__ valid, int32_t *__restrict__ first_bad,
               int *__restrict__ ulist, int *__restrict__ ucount, UndecidedConfigs uc) {
  extern __shared__ double smem[];
  const int B = blockDim.x;
  // this block's region of the item space (round robin: the blocks with work come first in the
  // grid, the surplus ones after them), and how far that region is filled
  const int region = (int)(blockIdx.x % (unsigned)ib.regions);
  const int64_t start = (int64_t)region * ib.regcap + (int64_t)(blockIdx.x / (unsigned)ib.regions) * B;
  const int fill = ib.count[region * kCounterStride];
  const int64_t n = (int64_t)region * ib.regcap + (fill < ib.regcap ? fill : ib.regcap);
  if (start >= n) return;
  const int nplan = StaticNplan<Spec>::value ? StaticNplan<Spec>::value : gip[H_NPLAN];
  Carve<float> c = carve_lds<float>(smem, gip, nip, gfp, nfp, nplan, 1, B, sizeof(float), WaveQueue<float, MBOX>::bytes());
  const int64_t it = start + threadIdx.x;
  const int64_t itc = it < (int64_t)ib.cap ? it : 0;
  const int ed = it < n ? ib.edge[itc] : -1;
  const bool active = ed >= 0;  // (a void slot: reserved by an edge that did not fit)
  // the waypoint in closed form (see count_waypoints): QA + min(idx * step / |QB - QA|, 1) (QB - QA),
  // rounded to binary32 as the check would round it anyway
  float *qw = reinterpret_cast<float *>(c.col0) + threadIdx.x;
  {
    const int64_t e0 = active ? ed : 0;
    double t = active ? (double)ib.idx[itc] * ib.tstep[e0] : 0.0;
    t = t < 1.0 ? t : 1.0;
    for_rows(src.QA, src.QB, src.E, e0, nplan, src.layout, true, [&](int k, double a, double b) {
      qw[k * B] = active ? (float)fma(t, b - a, a) : 0.0f;
    });
  }
  __syncthreads();
  const int code = check_one<float, MAXS, WBOX, MBOX, Spec>(c, qw, B, active, tol, it, uc, 0, ib.edge,
                                                            ib.idx, src);
  if (active && code != V_NONE) {
    if (code == V_CONTACT) {
      valid[ed] = 0;
      if (first_bad) atomicMin(reinterpret_cast<unsigned *>(first_bad) + ed, (unsigned)ib.idx[it]);
    } else {
      bool handed = false;
      if constexpr (!kQueued<float, MAXS>) {
        // immediate interpreter: the whole configuration goes to the exact configuration kernel
        if (uc.count) {
          const int j = atomicAdd(uc.count, 1);
          if (j < uc.cap) {
            exact_waypoint(src, gip + gip[H_OFF_PERM], nplan, ed, ib.idx[it], uc.q + (size_t)j * nplan, it);
            uc.edge[j] = ed;
            uc.idx[j] = ib.idx[it];
            uc.ga[j] = uc.gb[j] = -1;
            handed = true;
          }
        }
      }
      // could not hand the undecided item over: the exact edge kernel redoes the whole edge.  An
      // edge has up to K items here, but `ulist` holds E entries and the re-run has one lane per
      // entry: the first item to claim the edge (generation-stamped word, never cleared between
      // launches) lists it, the others find it listed.
      if (!handed && atomicExch(&ib.claim[ed], ib.gen) != ib.gen) ulist[atomicAdd(ucount, 1)] = ed;
    }
  }
}

// ---- the walking kernel's body (one lane per edge): float32 checks for the filter, float64 for the verdicts
// Work assignment shared by the exact kernels: lane j of (virtual) block vb takes item j, or -- when
// re-running the filter's uncertain items -- item ulist[j] for j < *ucount.
__device__ __forceinline__ bool pick_item(int64_t n, const int *__restrict__ ulist,
                                          const int *__restrict__ ucount, int64_t *item, int64_t vb) {
  const int64_t j = vb * blockDim.x + threadIdx.x;
  if (ulist) {
    const bool a = j < (int64_t)*ucount;
    *item = a ? (int64_t)ulist[j] : 0;
    return a;
  }
  *item = j;
  return j < n;
}
__device__ __forceinline__ bool pick_item(int64_t n, const int *__restrict__ ulist,
                                          const int *__restrict__ ucount, int64_t *item) {
  return pick_item(n, ulist, ucount, item, (int64_t)blockIdx.x);
}

constexpr int kMaxWaypoints = 1 << 20;  // per-edge guard; the reference would spin forever

// One lane per edge.  Check 0 is the endpoint QB; checks 1..K are the interior waypoints of
// _valid_collision_interval(QA, QB, step) generated on chip by the reference's own recurrence
//   w <- w + ((QB - w)/||QB - w||) * min(step, ||QB - w||)      (planning/utils.py:182-185)
// until w == QB (np.array_equal, :211).  ||.|| is the sequential-sum 2-norm over qpos
// addresses in ascending order (see DESIGN.md "waypoint semantics").  The recurrence always
// runs in float64, also in the filter kernel (FILTER = true), whose per-configuration checks
// run in float32 and which hands every edge it cannot decide within `tol` to the exact kernel
// through `ulist` / `ucount`.
template <class T, int MAXS, bool WBOX, bool MBOX>
__device__ __forceinline__ void edge_body(const int *__restrict__ gip, int nip, const T *__restrict__ gtp,
                                          int ntp, const double *__restrict__ QA,
                                          const double *__restrict__ QB, int64_t E, double step, int layout,
                                          int flags, T tol, uint8_t *__restrict__ valid,
                                          int32_t *__restrict__ first_bad, int *__restrict__ status,
                                          int *__restrict__ ulist, int *__restrict__ ucount,
                                          const int *__restrict__ rlist, const int *__restrict__ rcount,
                                          UndecidedConfigs uc, int vblock0 = -1, int nvblocks = 0) {
  extern __shared__ double smem[];
  const int B = blockDim.x;
  // (virtual) blocks vblock0, vblock0 + nvblocks, ... of the work: a kernel of its own passes nothing and
  // every workgroup takes the block of its own number, once; a role inside k_tail strides over the list
  if (vblock0 < 0) { vblock0 = (int)blockIdx.x; nvblocks = (int)gridDim.x; }
  const int64_t total = rlist ? (int64_t)*rcount : E;
  if ((int64_t)vblock0 * B >= total) return;
  const int nplan = gip[H_NPLAN];
  Carve<T> c = carve_lds<T>(smem, gip, nip, gtp, ntp, nplan, 1, B, sizeof(double), WaveQueue<T, MBOX>::bytes());
  for (int64_t vb = vblock0; vb * B < total; vb += nvblocks) {
  if (vb != vblock0) __syncthreads();  // the columns and queues change hands
  int64_t i;
  const bool active = pick_item(E, rlist, rcount, &i, vb);
  // The walking waypoint lives in LDS (starts at QB for check 0, then QA + steps); the edge end
  // is re-read from global memory when needed (L2-resident, coalesced in the SoA layout):
  // halving the per-lane LDS footprint buys an extra wave per SIMD.
  double *qw = c.col0 + threadIdx.x;
  auto end_col = [&](int k) -> double {
    return active ? ((layout == MJPL_SOA) ? QB[(int64_t)k * E + i] : QB[i * nplan + k]) : 0.0;
  };
  auto start_col = [&](int k) -> double {
    return active ? ((layout == MJPL_SOA) ? QA[(int64_t)k * E + i] : QA[i * nplan + k]) : 0.0;
  };
  load_columns(qw, B, QB, E, i, nplan, layout, active);
  __syncthreads();
  IP perm = c.ip + c.ip[H_OFF_PERM];
  // the walking list of a launch that made the endpoints items (mjpl_fused.h, small batches): an edge whose endpoint
  // was found in contact has its verdict already -- the interior walk must not write over it
  const bool verdict_stands = active && rlist && (flags & MJPL_EDGE_INTERIOR_ONLY) && valid[i] == 0;

  bool finite = true, at_end = true;
  for (int k = 0; k < nplan; k++) {
    double a = start_col(k), b = qw[k * B];
    finite = finite && (fabs(a) <= 1.79769313486231570815e+308) && (fabs(b) <= 1.79769313486231570815e+308);
    at_end = at_end && (a == b);
  }
  bool done = !active || verdict_stands;
  bool ok = true, unsure = false;
  int fb = -1;
  if (active && finite) {
    // an edge that needs more than kMaxWaypoints steps would only be found out after walking
    // them all: say so at once (same verdict: reported like a non-finite edge)
    double s0 = 0;
    for (int k = 0; k < nplan; k++) {
      const double d = qw[k * B] - start_col(k);
      s0 = s0 + d * d;
    }
    if (!(sqrt(s0) <= step * (kMaxWaypoints + 2.0))) finite = false;
  }
  if (active && !finite && !verdict_stands) {
    done = true; ok = false; fb = -2;
    atomicOr(status, kStatusNonFinite);
  }
  if (active && !finite)  // (the lane keeps company through the checks: zeros, never a NaN / inf row -- see k_filter_endpoints)
    for (int k = 0; k < nplan; k++) qw[k * B] = 0.0;

  // Iteration 0 checks the endpoint (apply_constraints validates q before the interval,
  // utils.py:144); every later iteration advances the waypoint and checks it.  One call site
  // of run_config keeps a single copy of the interpreter in the instruction stream.
  int idx = 0;
  bool stepped = false;
  bool first = (flags & MJPL_EDGE_INTERIOR_ONLY) == 0;
  if (!first && !done && at_end) done = true;
  while (__ballot(!done) != 0ull) {
    if (!first && !done) {
      if (idx == 0 && !stepped) {  // leaving check 0: the walk starts at QA
        for (int k = 0; k < nplan; k++) qw[k * B] = start_col(k);
        stepped = true;
      }
      // _step(w, QB, step)
      double s = 0;
      for (int k = 0; k < nplan; k++) {
        const int col = perm[k];
        double d = end_col(col) - qw[col * B];
        s = s + d * d;
      }
      const double mag = sqrt(s);
      const double sm = step < mag ? step : mag;
      bool eq = true;
      for (int k = 0; k < nplan; k++) {
        const double ek = end_col(k);
        double d = ek - qw[k * B];
        double nw = qw[k * B] + (d / mag) * sm;
        qw[k * B] = nw;
        eq = eq && (nw == ek);
      }
      if (!(mag > 0.0) || !(mag <= 1.79769313486231570815e+308)) {
        // the squared distance under- or overflowed: the reference's recurrence yields NaN from
        // here on and never ends -- reported like a non-finite edge
        done = true; ok = false; fb = -2;
        atomicOr(status, kStatusNonFinite);
      } else if (eq) {
        done = true;  // reached QB: that element is dropped by waypoints[1:-1]
      } else {
        idx++;
        if (idx > kMaxWaypoints) {
          done = true; ok = false; fb = -2;
          atomicOr(status, kStatusNonFinite);
        }
      }
    }
    // idx is the same for every lane that is still walking (they all started together)
    const unsigned long long walking = __ballot(!done);
    const int widx = __builtin_amdgcn_readfirstlane(__shfl(idx, walking ? __ffsll((long long)walking) - 1 : 0));
    const int code = check_one<T, MAXS, WBOX, MBOX>(c, qw, B, !done, tol, i, uc, widx);
    if (!done && code == V_CONTACT) { done = true; ok = false; fb = idx; }
    if (!done && code == V_UNSURE) {
      // the filter cannot decide this configuration: hand IT (not the whole edge) to the exact
      // configuration kernel and walk on as if it were valid; that kernel lowers first_bad and
      // clears valid if it finds a contact.  Only if the hand-off buffer is full does the whole
      // edge go to the exact edge kernel.  (The queued interpreter hands over single pairs by
      // itself; when IT reports a configuration undecidable, the whole edge goes.)
      const int j = kQueued<T, MAXS> ? uc.cap : atomicAdd(uc.count, 1);
      if (j < uc.cap) {
        for (int k = 0; k < nplan; k++) uc.q[(size_t)j * nplan + k] = qw[k * B];
        uc.edge[j] = (int)i;
        uc.idx[j] = idx;
        uc.ga[j] = uc.gb[j] = -1;  // the whole configuration (immediate interpreter)
      } else {
        done = true; unsure = true;
      }
    }
    if (first && !done && at_end) done = true;  // waypoints == [start]: nothing interior
    first = false;
  }
  if (active && !verdict_stands) {
    if (unsure) {
      ulist[atomicAdd(ucount, 1)] = (int)i;  // the exact kernel writes valid / first_bad
    } else {
      valid[i] = ok ? 1 : 0;
      if (first_bad) first_bad[i] = fb;
    }
  }
  }
}

// ---- persistent kernels: one wavefront per tile of 64, tiles handed out by a device counter -----------
// k_filter_endpoints / k_filter_items give every workgroup 256 consecutive units and die with it: a
// workgroup's four waves hold their SIMD slots and its 50 KiB of LDS until the slowest of them is through
// (a wave whose lanes drain more candidates takes longer), the last third of a round of workgroups runs
// on a third of the wave slots, and the item kernel is launched for the whole item SPACE because the host
// does not know the count.  Measured: 2.0 of 3 wave slots occupied on average (SQ_WAVE_CYCLES).  Here the
// grid is what the chip holds at once (three workgroups per CU) for the whole kernel; every WAVE takes
// the next tile of 64 units from a counter until there is none, with its own private slice of the
// workgroup's LDS -- [columns | pose saves | candidate queues], contiguous per wave, so that no barrier
// is needed after the one that publishes the workgroup's copy of the constant table.
template <class T, bool MBOX>
struct WaveLds {
  float *col;   // [nplan][64] binary32 planning columns of this wave's lanes
  T *save;      // [nsave * 7][64]
  char *qmem;   // this wave's candidate queues
  T *ltab;      // the workgroup's copy of the constant table
  char *base;   // start of the wave's slice (the endpoint kernel lays its float64 walking rows over it)
  size_t bytes; // of the slice
};
__host__ __device__ constexpr size_t wave_slice_bytes(int nplan, int nsave, size_t scalar, size_t qbytes) {
  return (((size_t)nplan * 64 * sizeof(float) + 7) & ~(size_t)7) + (((size_t)nsave * 7 * 64 * scalar + 7) & ~(size_t)7) + ((qbytes + 7) & ~(size_t)7);
}
template <class T, bool MBOX>
__device__ __forceinline__ WaveLds<T, MBOX> carve_wave(double *smem, const T *__restrict__ gtp, int ntp, int nplan, int nsave) {
  WaveLds<T, MBOX> w;
  const size_t qbytes = WaveQueue<T, MBOX>::bytes();
  w.bytes = wave_slice_bytes(nplan, nsave, sizeof(T), qbytes);
  const int nwaves = blockDim.x >> 6;
  char *b = reinterpret_cast<char *>(smem) + (size_t)(threadIdx.x >> 6) * w.bytes;
  w.base = b;
  w.col = reinterpret_cast<float *>(b);
  w.save = reinterpret_cast<T *>(b + (((size_t)nplan * 64 * sizeof(float) + 7) & ~(size_t)7));
  w.qmem = reinterpret_cast<char *>(w.save) + (((size_t)nsave * 7 * 64 * sizeof(T) + 7) & ~(size_t)7);
  w.ltab = reinterpret_cast<T *>(reinterpret_cast<char *>(smem) + (size_t)nwaves * w.bytes);
  for (int k = threadIdx.x; k < ntp; k += blockDim.x) w.ltab[k] = gtp[k];
  return w;
}

// the per-configuration check on a wave's private slice (queued interpreter or the model's own code)
template <int MAXS, bool WBOX, bool MBOX, class Spec>
__device__ __forceinline__ int check_wave(const int *__restrict__ gip, const float *__restrict__ gfp, const WaveLds<float, MBOX> &w,
                                          bool active, float tol, int64_t row, const UndecidedConfigs &uc,
                                          const int *item_edge, const int *item_idx, const EdgeSource &src,
                                          const _Float16 *adq = nullptr) {
  const int lane = threadIdx.x & 63;
  PatchSink ps;
  ps.uc = uc;
  ps.adq = adq;  // (the edge certificate of the fused kernel's endpoint tiles; null: an ordinary check)
  ps.qcol = nullptr;  // (src is always set here: what goes to the exact re-check is read / rebuilt from the caller's rows)
  ps.B = 64;
  ps.L = 1;
  ps.nplan = gip[H_NPLAN];
  ps.idx = 0;
  ps.item_edge = item_edge;
  ps.item_idx = item_idx;
  ps.src = src;
  ps.perm = (IP)gip + gip[H_OFF_PERM];
  WaveQueue<float, MBOX> wq;
  wq.carve(w.qmem);
  if constexpr (!std::is_void<Spec>::value)
    return Spec::run((FP)gfp, w.ltab, w.col + lane, 64, w.save + lane, 64, active, tol, wq, (int)row, ps);
  else
    return run_config_queued<float, MAXS, WBOX, MBOX>((IP)gip, (FP)gfp, w.ltab, w.col + lane, 64, w.save + lane, 64, active, tol, wq,
                                                      (int)row, ps);
}

// Tile queues.  One counter for the whole grid does not work: returning atomics on ONE address are served
// at about 18 ns apiece on this chip, whoever asks -- 10 354 item tiles + one failed attempt per wave made
// the item kernel 0.185 ms long (0.128 without the counter), the endpoint kernel 0.126 (0.096).  So there
// is a queue per region (up to 32 addresses, a 128-byte line each): a wave starts at the queue of its own
// number in the grid and moves on to the next queue when one is empty; after a full turn without a tile
// it is done.  Within a queue the tiles go out in order.
struct TileQueues {
  int *ctr;         // [nq] counters, kCounterStride ints apart
  int nq, q;        // queues; the one this wave is on
  int rank, peers;  // this wave's number among the `peers` waves whose home is queue q
  int nwaves;
  int round = 0, home_rounds = -1;  // static rounds taken / to take on the home queue (-1: not known yet)
  bool outstanding = false;         // an ask() has gone out whose answer take() has not looked at
  bool dynamic;                     // false: the remainder is dealt out statically, too (no counter at all)
  __device__ __forceinline__ TileQueues(int *c, int n, bool dyn) : ctr(c), nq(n), dynamic(dyn) {
    const int w = (int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
    nwaves = (int)(gridDim.x * (blockDim.x >> 6));
    q = w % n;
    rank = w / n;
    peers = (nwaves - q + n - 1) / n;
  }
  // Tiles of a queue come in two parts.  The first peers * floor(ntiles / peers) of them are dealt out
  // statically -- the r-th round of the home wave with rank u is tile u + r * peers: no memory operation,
  // and every home wave of the queue gets the same number.  The remainder (less than one round) goes
  // through the queue's counter, and so does whatever other queues still hold when this one is empty:
  // that is what evens out the end of the kernel.  (A returning atomic per tile, even spread over 32
  // addresses, costs more than it balances -- item kernel 0.153 ms against 0.130 with static tiles
  // only; asked for one tile ahead and only for the remainder it does not show.)
  __device__ __forceinline__ int static_part(int nt, int qq) const {
    const int p = (nwaves - qq + nq - 1) / nq;
    return p > 0 ? (nt / p) * p : 0;
  }
  __device__ __forceinline__ int ask() const {
    int j = 0;
    if ((threadIdx.x & 63) == 0) j = atomicAdd(ctr + q * kCounterStride, 1);
    return j;
  }
  // Called before a tile's arithmetic: if the NEXT take() will be answered by a counter, ask it now, so
  // that the answer's round trip is over when it is needed.  -> the value to hand to take().
  __device__ __forceinline__ int ask_ahead(int asked) {
    if (dynamic && home_rounds >= 0 && round >= home_rounds && !outstanding) {
      outstanding = true;
      return ask();
    }
    return asked;
  }
  // ntiles(q): how many tiles queue q holds.  -> queue of the next tile (index in *index), or -1: nothing
  // is left.  When a counter's answer says its queue is empty the wave looks at ALL counters at once (one
  // load, a lane per queue; a counter only grows, so a queue seen empty is empty), moves to the nearest
  // queue that still has tiles and asks there; a wave that sees none is done.
  template <class F>
  __device__ __forceinline__ int take(int asked, F &&ntiles, int *index) {
    if (home_rounds < 0) home_rounds = peers > 0 ? ntiles(q) / peers : 0;
    if (round < home_rounds) {
      *index = rank + round * peers;
      round++;
      return q;
    }
    if (!dynamic) {  // the remainder: one more tile for the lowest ranks, and that is all
      const int j = rank + round * peers;
      if (round > home_rounds || j >= ntiles(q)) return -1;
      round++;
      *index = j;
      return q;
    }
    if (!outstanding) asked = ask();
    outstanding = false;
    int j = __builtin_amdgcn_readfirstlane(asked);
    for (;;) {
      const int nt = ntiles(q), base = static_part(nt, q);
      if (base + j < nt) {
        *index = base + j;
        return q;
      }
      const int lane = threadIdx.x & 63;
      int left = 0;
      if (lane < nq) {
        const int ntl = ntiles(lane);
        left = ntl - static_part(ntl, lane) - __hip_atomic_load(ctr + lane * kCounterStride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      const unsigned long long m = __ballot(left > 0);
      if (m == 0ull) return -1;
      // the first queue with tiles at or after q + 1, cyclically (nq <= 64)
      const int s = q + 1 == nq ? 0 : q + 1;
      const unsigned long long hi = m >> s;
      q = hi ? s + (int)__builtin_ctzll(hi) : (int)__builtin_ctzll(m);
      j = __builtin_amdgcn_readfirstlane(ask());
    }
  }
};
// a wave's LDS slice changes hands between differently typed views (float64 walking rows, binary32
// columns, queue records) without a workgroup barrier: keep the compiler and the LDS queue in order
__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

template <class Spec, int MAXS, bool WBOX, bool MBOX>
__global__ void __launch_bounds__(kBlock, (kMinWaves<Spec, MAXS>))
k_filter_items_pw(const int *__restrict__ gip, int nip, const float *__restrict__ gfp, int nfp, ItemBuffers ib,
                  EdgeSource src, float tol, uint8_t *__restrict__ valid, int32_t *__restrict__ first_bad,
                  int *__restrict__ ulist, int *__restrict__ ucount, UndecidedConfigs uc, int *__restrict__ tiles) {
  static_assert(kQueued<float, MAXS>, "persistent kernels serve the queued interpreter");
  extern __shared__ double smem[];
  const int lane = threadIdx.x & 63;
  // The grid is sized for the item SPACE (the host does not know the count).  A launch with few items -- the
  // late chunks of a planner's extension: a handful of lanes -- needs few workgroups: when no queue holds a
  // round of tiles for all its home waves, every tile goes out through the counters, any wave can take any
  // of them, and workgroups beyond the tile count leave before they stage anything (a launch of 768
  // workgroups for three tiles took 60 us, most of it table copies nobody used).
  {
    int nt = 0, stat = 0;
    if (lane < ib.regions) {
      const int f = ib.count[lane * kCounterStride];
      nt = ((f < ib.regcap ? f : ib.regcap) + 63) >> 6;
      const int nwaves = (int)(gridDim.x * (blockDim.x >> 6));
      stat = nt >= (nwaves - lane + ib.regions - 1) / ib.regions ? 1 : 0;
    }
    for (int o = 32; o > 0; o >>= 1) { nt += __shfl_xor(nt, o); stat |= __shfl_xor(stat, o); }
    if (!stat && (int)(blockIdx.x * (blockDim.x >> 6)) >= nt) return;
  }
  const int nplan = StaticNplan<Spec>::value ? StaticNplan<Spec>::value : gip[H_NPLAN];
  WaveLds<float, MBOX> w = carve_wave<float, MBOX>(smem, gfp, nfp, nplan, gip[H_NSAVE]);
  __syncthreads();  // the table copy; from here on every wave is on its own
  // a tile = 64 consecutive items of one region; the regions are the queues.  Tiles cost what their
  // candidates cost (a wave that drains more takes longer): the remainder is dealt out dynamically
  TileQueues tq(tiles, ib.regions, true);
  auto ntiles = [&](int r) {
    const int f = ib.count[r * kCounterStride];
    return ((f < ib.regcap ? f : ib.regcap) + 63) >> 6;
  };
  int asked = 0;
  for (;;) {
    int j = 0;
    const int region = tq.take(asked, ntiles, &j);
    if (region < 0) break;
    const int fill = ib.count[region * kCounterStride];
    const int64_t n = (int64_t)region * ib.regcap + (fill < ib.regcap ? fill : ib.regcap);
    const int64_t it = (int64_t)region * ib.regcap + (int64_t)j * 64 + lane;
    const int64_t itc = it < (int64_t)ib.cap ? it : 0;
    const int ed = it < n ? ib.edge[itc] : -1;
    const bool active = ed >= 0;  // (a void slot: reserved by an edge that did not fit)
    float *qw = w.col + lane;
    {
      const int64_t e0 = active ? ed : 0;
      double tt = active ? (double)ib.idx[itc] * ib.tstep[e0] : 0.0;
      tt = tt < 1.0 ? tt : 1.0;
      for_rows(src.QA, src.QB, src.E, e0, nplan, src.layout, true, [&](int k, double a, double b) {
        qw[k * 64] = active ? (float)fma(tt, b - a, a) : 0.0f;
      });
    }
    wave_lds_fence();
    asked = tq.ask_ahead(asked);  // (the next tile, if a counter hands it out: answered while this one is checked)
    const int code = check_wave<MAXS, WBOX, MBOX, Spec>(gip, gfp, w, active, tol, it, uc, ib.edge, ib.idx, src);
    if (active && code != V_NONE) {
      if (code == V_CONTACT) {
        valid[ed] = 0;
        if (first_bad) atomicMin(reinterpret_cast<unsigned *>(first_bad) + ed, (unsigned)ib.idx[it]);
      } else if (atomicExch(&ib.claim[ed], ib.gen) != ib.gen) {
        ulist[atomicAdd(ucount, 1)] = ed;  // (see k_filter_items: the exact edge kernel redoes the whole edge, once)
      }
    }
    wave_lds_fence();
  }
}

template <class Spec, int MAXS, bool WBOX, bool MBOX>
__global__ void __launch_bounds__(kBlock, (kMinWaves<Spec, MAXS>))
k_filter_endpoints_pw(const int *__restrict__ gip, int nip, const float *__restrict__ gfp, int nfp,
                      const double *__restrict__ QA, const double *__restrict__ QB, int64_t E, int layout,
                      float tol, uint8_t *__restrict__ valid, int32_t *__restrict__ first_bad,
                      int *__restrict__ status, int *__restrict__ ulist, int *__restrict__ ucount,
                      UndecidedConfigs uc, ItemBuffers ib, double step, int *__restrict__ zero_next, int *__restrict__ tiles) {
  static_assert(kQueued<float, MAXS>, "persistent kernels serve the queued interpreter");
  extern __shared__ double smem[];
  zero_counters(zero_next);
  const int nplan = StaticNplan<Spec>::value ? StaticNplan<Spec>::value : gip[H_NPLAN];
  WaveLds<float, MBOX> w = carve_wave<float, MBOX>(smem, gfp, nfp, nplan, gip[H_NSAVE]);
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int ntile = (int)((E + 63) >> 6);
  // the two float64 rows per lane of the waypoint count lie over the wave's own slice
  const bool fits = (size_t)2 * nplan * 64 * sizeof(double) <= w.bytes;
  double *qe = reinterpret_cast<double *>(w.base) + lane;
  double *qx = qe + (size_t)nplan * 64;
  // tile t = edges [64 t, 64 t + 64); queue (= item region) t % regions holds the tiles t = j * regions + q
  // (endpoint tiles cost much the same: all static -- measured 0.096 ms against 0.109 with a dynamic remainder)
  TileQueues tq(tiles, ib.regions, false);
  auto ntiles = [&](int q) { return (ntile - q + ib.regions - 1) / ib.regions; };
  int asked = 0;
  for (;;) {
    int j = 0;
    const int region = tq.take(asked, ntiles, &j);
    if (region < 0) break;
    const int64_t i = ((int64_t)j * ib.regions + region) * 64 + lane;
    const bool active = i < E;
    bool finite = true;
    for_rows(QA, QB, E, i, nplan, layout, active, [&](int, double a, double b) {
      finite = finite && (fabs(a) <= 1.79769313486231570815e+308) && (fabs(b) <= 1.79769313486231570815e+308);
    });
    // the waypoint COUNT first (see k_filter_endpoints); with checkpoints the count follows the check
    const bool early = fits && !ib.ckpt;
    int K = 0;
    if (early) {
      load_columns(qe, 64, QB, E, i, nplan, layout, active);
      K = count_waypoints(gip, QA, E, i, step, layout, active && finite, qe, 64, qx, 64, ib, nplan);
      wave_lds_fence();
    }
    float *qw = w.col + lane;
    load_columns(qw, 64, QB, E, i, nplan, layout, active && finite);  // (see k_filter_endpoints)
    wave_lds_fence();
    const bool run = active && finite;
    asked = tq.ask_ahead(asked);  // (the next tile, if a counter hands it out: answered while this one is checked)
    const int code = check_wave<MAXS, WBOX, MBOX, Spec>(gip, gfp, w, run, tol, i, uc, nullptr, nullptr,
                                                        EdgeSource{QB, QB, E, layout, 0.0, nullptr});
    bool survive = run && code != V_CONTACT;
    if (active) {
      if (!finite) {
        valid[i] = 0;
        if (first_bad) first_bad[i] = -2;
        atomicOr(status, kStatusNonFinite);
      } else if (code == V_CONTACT) {
        valid[i] = 0;
        if (first_bad) first_bad[i] = 0;
      } else if (code == V_UNSURE) {  // (queued interpreter: the whole edge goes to the exact edge kernel)
        ulist[atomicAdd(ucount, 1)] = (int)i;
        survive = false;
      } else {
        valid[i] = 1;  // so far; the interior pass and the patch pass may clear it
        if (first_bad) first_bad[i] = -1;
      }
    }
    const unsigned long long m = __ballot(survive);
    wave_lds_fence();
    if (m == 0ull) continue;
    if (!fits) {
      if (survive) ib.llist[atomicAdd(ib.lcount, 1)] = (int)i;  // too many columns: walking kernel
      continue;
    }
    if (!early) {
      load_columns(qe, 64, QB, E, i, nplan, layout, active);
      K = count_waypoints(gip, QA, E, i, step, layout, survive, qe, 64, qx, 64, ib, nplan);
    }
    emit_items(gip, QA, E, i, step, layout, survive, K, early ? nullptr : qe, 64, early ? nullptr : qx, 64, ib, nplan, region);
    if (lane == 0) atomicAdd(ib.scount + region * kCounterStride, (int)__builtin_popcountll(m));
    wave_lds_fence();
  }
}

// dynamic LDS of a persistent kernel's workgroup: the waves' slices, then the table copy
inline size_t persistent_lds_bytes(int nplan, int nsave, size_t ntab, bool mbox, int block = kBlock) {
  const size_t q = mbox ? WaveQueue<float, true>::bytes() : WaveQueue<float, false>::bytes();
  return (size_t)(block / 64) * wave_slice_bytes(nplan, nsave, sizeof(float), q) + ((ntab * sizeof(float) + 7) & ~(size_t)7);
}
// grid of a persistent kernel: as many workgroups as the device holds at once (asked of the runtime once
// per kernel and LDS size), never more than there are tiles for (four per workgroup)
template <class K>
inline unsigned persistent_grid(K kernel, size_t lds, long long ntile) {
  static thread_local const void *last_k = nullptr;
  static thread_local size_t last_lds = 0;
  static thread_local int last_dev = -1, resident = 0;
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (last_k != reinterpret_cast<const void *>(kernel) || last_lds != lds || last_dev != dev) {
    int per_cu = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void *>(kernel), kBlock, lds) != hipSuccess || per_cu < 1) per_cu = 1;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
    if (const char *f = getenv("MJPL_PW_BLOCKS_PER_CU")) {  // (A/B measurements)
      fprintf(stderr, "persistent_grid: runtime says %d workgroups per CU at %zu B of LDS, %d CUs\n", per_cu, lds, cus);
      if (atoi(f) > 0) per_cu = atoi(f);
    }
    resident = per_cu * cus;
    last_k = reinterpret_cast<const void *>(kernel); last_lds = lds; last_dev = dev;
  }
  const long long want = (ntile + kBlock / 64 - 1) / (kBlock / 64);
  return (unsigned)(want < 1 ? 1 : (want < resident ? want : resident));
}

// ---- exact re-check of single geom pairs the filter could not decide -------------------------
// Item u = (configuration uc.q[u], moving geom uc.ga[u], partner geom uc.gb[u]).  One lane per
// item: float64 FK of the moving bodies (same statements as run_config), capturing the world
// pose of the one or two geoms it needs, then the bounding cull and the narrowphase routine of
// that pair exactly as the full exact kernel would run them.  A contact clears valid[edge] and
// lowers first_bad[edge] (unsigned min; -1 = valid so far).  Latency of a wave is one FK instead
// of FK + all culls + every narrowphase call some lane of the wave needs.
struct GeomTable {           // per model geom, float64 [GT_LEN]
  const double *t;
  int moving_base;           // id of the first moving geom (a scene-generic ExactSpec numbers geoms from it)
};
enum : int { GTB_TYPE = 0, GTB_SIZE = 1, GTB_RBOUND = 4, GTB_MARGIN = 5, GTB_STATIC = 6, GTB_XPOS = 7,
             GTB_XMAT = 10, GTB_LEN = 19 };

// ESpec: void = the float64 FK below walks the compiled program; a generated struct
// (mjpl_amd/specialise.py: generate_exact) supplies the SAME statements with the model's constants as
// literals -- a wave of this kernel runs alone on its SIMD, and ~700 scalar table loads per wave,
// each waited for, were most of its 30 us.
template <class ESpec>
__device__ __forceinline__ void patch_pairs_body(const int *__restrict__ gip, int nip, const double *__restrict__ gdp, int ndp,
                                                 GeomTable gt, UndecidedConfigs uc, uint8_t *__restrict__ valid,
                                                 int32_t *__restrict__ first_bad, int vblock0, int nvblocks) {
  extern __shared__ double smem[];
  const int B = blockDim.x;
  const int64_t n = *uc.count < uc.cap ? *uc.count : uc.cap;
  if ((int64_t)vblock0 * B >= n) return;
  const int nplan = gip[H_NPLAN];
  Carve<double> c = carve_lds<double>(smem, gip, nip, gdp, ndp, nplan, 1, B);
  // the number of items is read on the device and may exceed the grid: blocks stride over it
  for (int64_t base = (int64_t)vblock0 * B; base < n; base += (int64_t)nvblocks * B) {
  const int64_t u = base + threadIdx.x;
  bool active = u < n;
  const int ga = active ? uc.ga[u] : -1;
  const int gb = active ? uc.gb[u] : -1;
  active = active && ga >= 0;
  double *q = c.col0 + threadIdx.x;
  load_columns(q, B, uc.q, n, u < n ? u : 0, nplan, MJPL_AOS, active);
  if (__ballot(active) == 0ull) continue;

  typedef GeomT<double> Geom;
  IP ip = c.ip;
  DP tp = c.tp;
  double *save = c.save + threadIdx.x;
  const int sstride = B;
  Geom A, Bg;
#pragma unroll
  for (int k = 0; k < 3; k++) A.pos[k] = Bg.pos[k] = 0;
#pragma unroll
  for (int k = 0; k < 9; k++) A.m[k] = Bg.m[k] = 0;
  double p[3] = {0, 0, 0}, qt[4] = {1, 0, 0, 0}, R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  if constexpr (!std::is_void<ESpec>::value) {
    ESpec::fk_pair(q, B, save, sstride, active, ga - ESpec::kRelative * gt.moving_base, gb - ESpec::kRelative * gt.moving_base, A, Bg);
  } else {
  const int nbodyops = uni(ip[H_NBODYOPS]);
  int pc = uni(ip[H_OFF_BODYOPS]);
  for (int b = 0; b < nbodyops; b++) {
    const int parent = uni(ip[pc + B_PARENT]);
    DP bd = tp + uni(ip[pc + B_DOFF]);
    const int njnt = uni(ip[pc + B_NJNT]);
    const int save_slot = uni(ip[pc + B_SAVE]);
    const int ngeom = uni(ip[pc + B_NGEOM]);
    pc += B_SIZE;
    double pp[3], pq[4], pR[9];
    if (parent == PARENT_CUR) {
#pragma unroll
      for (int k = 0; k < 3; k++) pp[k] = p[k];
#pragma unroll
      for (int k = 0; k < 4; k++) pq[k] = qt[k];
#pragma unroll
      for (int k = 0; k < 9; k++) pR[k] = R[k];
    } else if (parent == PARENT_STATIC) {
#pragma unroll
      for (int k = 0; k < 3; k++) pp[k] = bd[7 + k];
#pragma unroll
      for (int k = 0; k < 4; k++) pq[k] = bd[10 + k];
#pragma unroll
      for (int k = 0; k < 9; k++) pR[k] = bd[14 + k];
    } else {
      const double *sv = save + (size_t)(parent - 1) * 7 * sstride;
#pragma unroll
      for (int k = 0; k < 3; k++) pp[k] = sv[k * sstride];
#pragma unroll
      for (int k = 0; k < 4; k++) pq[k] = sv[(3 + k) * sstride];
      quat2mat(pR, pq);
    }
    double np[3], nq[4];
    {
      double bpos[3] = {bd[0], bd[1], bd[2]};
      double bquat[4] = {bd[3], bd[4], bd[5], bd[6]};
      mul_mat_vec3(np, pR, bpos);
      np[0] += pp[0]; np[1] += pp[1]; np[2] += pp[2];
      mul_quat(nq, pq, bquat);
    }
    for (int j = 0; j < njnt; j++) {
      const int jtype = uni(ip[pc + J_TYPE]);
      const int qsrc = uni(ip[pc + J_QSRC]);
      const int jflags = uni(ip[pc + J_FLAGS]);
      DP jd = tp + uni(ip[pc + J_DOFF]);
      pc += J_SIZE;
      const double qv = (qsrc >= 0) ? q[qsrc * B] : jd[7];
      const double dq = qv - jd[6];
      double jaxis[3] = {jd[0], jd[1], jd[2]};
      double jpos[3] = {jd[3], jd[4], jd[5]};
      if (jtype == JT_SLIDE) {
        double xaxis[3];
        rot_vec_quat(xaxis, jaxis, nq);
        np[0] += xaxis[0] * dq; np[1] += xaxis[1] * dq; np[2] += xaxis[2] * dq;
      } else {
        double xanchor[3] = {np[0], np[1], np[2]};
        if (jflags & JF_POS_NONZERO) {
          rot_vec_quat(xanchor, jpos, nq);
          xanchor[0] += np[0]; xanchor[1] += np[1]; xanchor[2] += np[2];
        }
        double sn, cs;
        sincos_half(dq * 0.5, &sn, &cs);
        double qloc[4] = {cs, jaxis[0] * sn, jaxis[1] * sn, jaxis[2] * sn};
        mul_quat(nq, nq, qloc);
        if (jflags & JF_POS_NONZERO) {
          double vec[3];
          rot_vec_quat(vec, jpos, nq);
          np[0] = xanchor[0] - vec[0]; np[1] = xanchor[1] - vec[1]; np[2] = xanchor[2] - vec[2];
        }
      }
    }
    normalize4(nq);
#pragma unroll
    for (int k = 0; k < 3; k++) p[k] = np[k];
#pragma unroll
    for (int k = 0; k < 4; k++) qt[k] = nq[k];
    quat2mat(R, qt);
    if (save_slot >= 0) {
      double *sv = save + (size_t)save_slot * 7 * sstride;
#pragma unroll
      for (int k = 0; k < 3; k++) sv[k * sstride] = p[k];
#pragma unroll
      for (int k = 0; k < 4; k++) sv[(3 + k) * sstride] = qt[k];
    }
    for (int g = 0; g < ngeom; g++) {
      const int gflags = uni(ip[pc + G_FLAGS]);
      DP gd = tp + uni(ip[pc + G_DOFF]);
      const int geom_id = uni(ip[pc + G_GEOMID]);
      pc += G_SIZE + MAX_SLOTS;
      if (__ballot(active && (geom_id == ga || geom_id == gb)) == 0ull) continue;
      Geom cur;
      if (gflags & GF_SAMEPOS) {
        cur.pos[0] = p[0]; cur.pos[1] = p[1]; cur.pos[2] = p[2];
      } else {
        double lpos[3] = {gd[0], gd[1], gd[2]};
        mul_mat_vec3(cur.pos, R, lpos);
        cur.pos[0] += p[0]; cur.pos[1] += p[1]; cur.pos[2] += p[2];
      }
      if (gflags & GF_SAMEROT) {
#pragma unroll
        for (int k = 0; k < 9; k++) cur.m[k] = R[k];
      } else {
        double lq[4] = {gd[3], gd[4], gd[5], gd[6]}, gq[4];
        mul_quat(gq, qt, lq);
        quat2mat(cur.m, gq);
      }
      const bool isa = geom_id == ga, isb = geom_id == gb;
#pragma unroll
      for (int k = 0; k < 3; k++) { A.pos[k] = isa ? cur.pos[k] : A.pos[k]; Bg.pos[k] = isb ? cur.pos[k] : Bg.pos[k]; }
#pragma unroll
      for (int k = 0; k < 9; k++) { A.m[k] = isa ? cur.m[k] : A.m[k]; Bg.m[k] = isb ? cur.m[k] : Bg.m[k]; }
    }
  }
  }
  if (!active) continue;
  const double *ta = gt.t + (size_t)ga * GTB_LEN, *tb = gt.t + (size_t)gb * GTB_LEN;
  if (tb[GTB_STATIC] != 0.0) {
#pragma unroll
    for (int k = 0; k < 3; k++) Bg.pos[k] = tb[GTB_XPOS + k];
#pragma unroll
    for (int k = 0; k < 9; k++) Bg.m[k] = tb[GTB_XMAT + k];
  }
  const int tya = (int)ta[GTB_TYPE], tyb = (int)tb[GTB_TYPE];
  const double sa[3] = {ta[GTB_SIZE], ta[GTB_SIZE + 1], ta[GTB_SIZE + 2]};
  const bool bplane = tyb == GT_PLANE;  // the interpreter passes no size for a plane
  const double sb[3] = {bplane ? 0.0 : tb[GTB_SIZE], bplane ? 0.0 : tb[GTB_SIZE + 1], bplane ? 0.0 : tb[GTB_SIZE + 2]};
  // pair margin and bounding cull in mj_collision's (g1 < g2) order, as compile_program folds them
  const int g1 = ga < gb ? ga : gb, g2 = ga < gb ? gb : ga;
  const double *t1 = gt.t + (size_t)g1 * GTB_LEN, *t2 = gt.t + (size_t)g2 * GTB_LEN;
  const double margin = fmax(t1[GTB_MARGIN], t2[GTB_MARGIN]);
  const double r1 = t1[GTB_RBOUND], r2 = t2[GTB_RBOUND];
  bool pass = true;
  if (tyb == GT_PLANE) {
    if (ta[GTB_RBOUND] > 0) {
      const double n[3] = {Bg.m[2], Bg.m[5], Bg.m[8]};
      const double dif[3] = {A.pos[0] - Bg.pos[0], A.pos[1] - Bg.pos[1], A.pos[2] - Bg.pos[2]};
      pass = !(dot3(dif, n) > margin + ta[GTB_RBOUND]);
    }
  } else if (r1 > 0 && r2 > 0) {
    const double bsum = r1 + r2 + margin;
    const double dx = A.pos[0] - Bg.pos[0], dy = A.pos[1] - Bg.pos[1], dz = A.pos[2] - Bg.pos[2];
    pass = !(dx * dx + dy * dy + dz * dz > bsum * bsum);
  }
  if (!pass) continue;
  // the partner is the first geom of the pair if its type is smaller, geom id breaking ties
  const bool pfirst = (tyb < tya) || (tyb == tya && gb < ga);
  const int code = pair_contact<double, true, true>(tya, A, sa, tyb, Bg, sb, pfirst, margin, 0.0);
  if (code == V_CONTACT) {
    const int ed = uc.edge[u];
    valid[ed] = 0;
    if (first_bad) atomicMin(reinterpret_cast<unsigned *>(first_bad) + ed, (unsigned)uc.idx[u]);
  }
  }
}

template <class ESpec>
__global__ void __launch_bounds__(kBlock)
k_patch_pairs(const int *__restrict__ gip, int nip, const double *__restrict__ gdp, int ndp, GeomTable gt,
              UndecidedConfigs uc, uint8_t *__restrict__ valid, int32_t *__restrict__ first_bad) {
  patch_pairs_body<ESpec>(gip, nip, gdp, ndp, gt, uc, valid, first_bad, (int)blockIdx.x, (int)gridDim.x);
}

// ---- one kernel for everything behind the item pass ---------------------------------------------------
// What is left when the endpoint and item kernels are through: (W) the edges that did not become items --
// longer than the item space allows per edge, or more than a region held -- for the walking body, (P) the
// geom pairs the filter could not decide, for the float64 pair re-check, (X) the edges that have to be
// redone in float64 as a whole.  In an RRT batch W and X are empty and P is ~30 waves: three launches cost
// three kernel boundaries (6.5 + 21 + 6.5 us of a 0.25 ms step) for one wave's worth of work.  Here they are
// roles of ONE grid: workgroups [0, nw) walk, the next np re-check pairs, the last nx redo edges, each
// role striding over its list.  W may add to the lists of P and X, so those wait for it: every W workgroup
// counts itself out (release), P and X spin on that count (a handful of microseconds when W has nothing to
// do).  W comes first in the grid, so it is resident before anything waits for it; a wait that outlasts
// every plausible W pass sets a status bit instead of hanging.
constexpr int kStatusTailTimeout = 2;
struct TailArgs {
  const int *ip; int nip;
  const float *fp; int nfp;
  const double *dp; int ndp;
  GeomTable gt;
  UndecidedConfigs uc;
  const double *QA, *QB;
  long long E;
  double step;
  int layout, flags;
  float tol;
  uint8_t *valid;
  int32_t *first_bad;
  int *status, *ulist, *ucount;
  const int *llist, *lcount;
  int *done;       // W workgroups that are through (cleared with the launch's counters)
  int nw, np, nx;  // workgroups per role
};

template <class ESpec, int MAXS_F, int MAXS_D, bool WBOX, bool MBOX>
__global__ void __launch_bounds__(kBlock)
k_tail(TailArgs a) {
  const int b = (int)blockIdx.x;
  if (b < a.nw) {
    // (interior waypoints only: check 0 was the endpoint kernel's)
    edge_body<float, MAXS_F, WBOX, MBOX>(a.ip, a.nip, a.fp, a.nfp, a.QA, a.QB, a.E, a.step, a.layout,
                                         a.flags | MJPL_EDGE_INTERIOR_ONLY, a.tol, a.valid, a.first_bad, a.status, a.ulist,
                                         a.ucount, a.llist, a.lcount, a.uc, b, a.nw);
    if (*a.lcount > 0) {
      __syncthreads();
      if (threadIdx.x == 0) {
        __threadfence();  // what this workgroup added to the lists, before its count
        __hip_atomic_fetch_add(a.done, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    return;
  }
  // (the walking list was closed by the endpoint kernel: when it is empty -- the usual case -- the walking
  // workgroups add nothing to anybody's list and nobody waits for them)
  // Thread 0 waits and decides for the workgroup; the decision travels through one LDS word, so every thread
  // takes the same side of the branch below (the bodies behind it hold __syncthreads: threads that each read the
  // status word themselves could see another workgroup's time-out between two of their loads and part ways).
  __shared__ int give_up;
  if (threadIdx.x == 0) {
    int gave = 0;
    if (*a.lcount > 0) {
      int spins = 0;
      while (__hip_atomic_load(a.done, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < a.nw) {
        __builtin_amdgcn_s_sleep(8);
        if (++spins > (1 << 24)) {  // (seconds: no W pass takes that long -- report, do not hang)
          atomicOr(a.status, kStatusTailTimeout);
          break;
        }
      }
      __threadfence();
      gave = (__hip_atomic_load(a.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & kStatusTailTimeout) ? 1 : 0;
    }
    give_up = gave;
  }
  __syncthreads();
  // (a wait that gave up: the lists may still be growing under a walking workgroup, so nothing read from them
  // now can be trusted -- leave the batch alone; the status bit is sticky and every synchronising entry point,
  // mjpl_take_status and the planner's rounds report it as MJPL_E_HIP)
  if (give_up) return;
  if (b < a.nw + a.np)
    patch_pairs_body<ESpec>(a.ip, a.nip, a.dp, a.ndp, a.gt, a.uc, a.valid, a.first_bad, b - a.nw, a.np);
  else
    edge_body<double, MAXS_D, WBOX, MBOX>(a.ip, a.nip, a.dp, a.ndp, a.QA, a.QB, a.E, a.step, a.layout, a.flags, 0.0, a.valid,
                                          a.first_bad, a.status, nullptr, nullptr, a.ulist, a.ucount, UndecidedConfigs{}, b - a.nw - a.np,
                                          a.nx);
}


}  // namespace mjpl

#include "mjpl_fused.h"
