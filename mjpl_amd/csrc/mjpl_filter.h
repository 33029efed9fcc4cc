// mjpl_filter.h -- the float32 filter kernels of the edge / configuration pipeline and what they
// share (LDS carve, per-lane configuration check, waypoint expansion), as templates on a `Spec`
// policy: void = the interpreter of mjpl_device.h walks the compiled model program; a generated
// struct (tools: mjpl_amd/specialise.py) supplies the same per-configuration check as straight-line
// code for ONE model -- the same kernels are then instantiated in that model's own library
// (libmjpl_spec_<hash>.so) and launched by the engine instead of the generic ones.
#pragma once

#include <type_traits>

#include "../../include/mjpl_hip.h"
#include "mjpl_device.h"

namespace mjpl {

constexpr int kBlock = 256;           // threads per workgroup: 4 wavefronts
constexpr int kFilterBlock = 256;     // queued filter kernels: four wavefronts share one LDS table copy
constexpr int kStatusNonFinite = 1;
// version of the contract between libmjpl_hip.so and a per-model specialised library
// (kernel signatures of this header + table layouts of mjpl_device.h)
#define MJPL_SPEC_ABI 1
// A model's own straight-line code is asked to fit three waves per SIMD (168 VGPRs): its register
// pressure is a few registers above that without the bound, and the third wave is worth more
template <class Spec> constexpr int kMinWaves = std::is_void<Spec>::value ? 1 : 3;

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

// queued narrowphase: the float32 filter of models without moving boxes
template <class T, bool MBOX>
constexpr bool kQueued = !Real<T>::exact && !MBOX;

template <class T>
__device__ __forceinline__ WaveQueue<T> wave_queue(char *qmem) {
  char *base = qmem + (size_t)(threadIdx.x >> 6) * WaveQueue<T>::bytes();
  WaveQueue<T> wq;
  wq.carve(base);
  return wq;
}

template <class T, int MAXS, bool WBOX, bool MBOX, class Spec = void>
__device__ __forceinline__ int check_one(const Carve<T> &c, const double *q, int B, bool active, T tol,
                                         int64_t row, const UndecidedConfigs &uc = UndecidedConfigs{},
                                         int idx = 0, const int *item_edge = nullptr,
                                         const int *item_idx = nullptr, const double *sink_q = nullptr,
                                         int sink_stride = 0, int qstride = 0) {
  if (qstride == 0) qstride = B;  // q[k * qstride]; the LDS pose saves always use stride B
  if constexpr (kQueued<T, MBOX>) {
    WaveQueue<T> wq = wave_queue<T>(c.qmem);
    PatchSink ps;
    ps.uc = uc;
    // where a drain lane finds the owner's configuration: the LDS columns, or (lane-per-item
    // kernel reading its configurations straight from the item buffer) global memory
    ps.qcol = sink_q ? sink_q : c.col0 + (threadIdx.x & ~63);
    ps.B = sink_q ? 1 : B;
    ps.L = sink_q ? sink_stride : 1;
    ps.nplan = c.ip[H_NPLAN];
    ps.idx = idx;
    ps.item_edge = item_edge;
    ps.item_idx = item_idx;
    if constexpr (!std::is_void<Spec>::value)  // this model's own straight-line check (same contract)
      return Spec::run(c.tp, c.ltab, q, qstride, c.save + threadIdx.x, B, active, tol, wq, (int)row, ps);
    else
      return run_config_queued<T, MAXS, WBOX>(c.ip, c.tp, c.ltab, q, qstride, c.save + threadIdx.x, B, active, tol,
                                              wq, (int)row, ps);
  } else {
    FkOut none = {};
    return run_config<T, MAXS, false, WBOX, MBOX>(c.ip, c.tp, q, qstride, c.save + threadIdx.x, B, active, tol,
                                                  none, row);
  }
}

template <class T>
__device__ __forceinline__ Carve<T> carve_lds(double *smem, const int *__restrict__ gip, int nip,
                                              const T *__restrict__ gtp, int ntp, int nplan, int ncolsets,
                                              int B) {
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
  c.save = reinterpret_cast<T *>(c.col0 + (size_t)ncolsets * nplan * B);
  const int nsave = gip[H_NSAVE];
  // queue memory starts 8-byte aligned after the saves
  c.qmem = reinterpret_cast<char *>(c.col0) +
           (((size_t)ncolsets * nplan * B * sizeof(double) + (size_t)nsave * 7 * B * sizeof(T) + 7) & ~(size_t)7);
  // queued kernels: constant-table copy behind the queues (staged below, before the barrier)
  c.ltab = reinterpret_cast<T *>(c.qmem + (size_t)(B / 64) * WaveQueue<T>::bytes());
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

// planning columns of configuration i -> this lane's LDS column slice
__device__ __forceinline__ void load_columns(double *col, int B, const double *__restrict__ Q,
                                             int64_t N, int64_t i, int nplan, int layout, bool active) {
  for (int c = 0; c < nplan; c++) {
    double v = 0.0;
    if (active) v = (layout == MJPL_SOA) ? Q[(int64_t)c * N + i] : Q[i * nplan + c];
    col[c * B] = v;
  }
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
  double *w;        // [cap][nplan] waypoints, one row per item (a lane's items are adjacent rows,
                    // so the lines it writes during its walk fill up in cache)
  int *edge, *idx;  // [cap] which edge, which check index (1..K)
  int *count;       // items written
  int cap;
  int *llist, *lcount;  // edges left to the walking kernel
  int kmax;             // edges with more interior waypoints than this stay with the walking kernel
  int *claim;           // [E] per-edge claim word (see k_filter_items)
  int gen;              // this launch's generation: claim[edge] == gen <=> the edge is in ulist already
};
constexpr int kExpandMinWaypoints = 24;  // kmax is at least this; small batches get more (item space / E)

// qe: this lane's edge end QB (LDS, stride B); qw: scratch for the walking waypoint (LDS, stride
// ws).  `todo` lanes own an edge i whose interior waypoints are wanted.
__device__ __forceinline__ void expand_edge(const int *__restrict__ gip, const double *__restrict__ QA,
                                            int64_t E, int64_t i, double step, int layout, bool todo,
                                            const double *qe, int B, double *qw, int ws,
                                            const ItemBuffers &ib) {
  const int nplan = gip[H_NPLAN];
  const int *perm = gip + gip[H_OFF_PERM];
  const int lane = threadIdx.x & 63;
  auto start_col = [&](int k) -> double {
    return todo ? ((layout == MJPL_SOA) ? QA[(int64_t)k * E + i] : QA[i * nplan + k]) : 0.0;
  };
  bool at_end = true;
  for (int k = 0; k < nplan; k++) {
    const double a = start_col(k);
    qw[k * ws] = a;
    at_end = at_end && (a == qe[k * B]);
  }
  double s0 = 0;
  for (int k = 0; k < nplan; k++) {
    const int col = perm[k];
    const double d = qe[col * B] - qw[col * ws];
    s0 = s0 + d * d;
  }
  bool done = !todo || at_end;
  if (!done && !(sqrt(s0) <= step * (double)(ib.kmax - 2))) {  // long edge
    ib.llist[atomicAdd(ib.lcount, 1)] = (int)i;
    done = true;
  }
  // one step of the recurrence: _step(w, QB, step)  (planning/utils.py:182-185; the statements of
  // edge_body); returns true when the walk has arrived at QB
  bool degenerate = false;  // the squared distance under- or overflowed (see edge_body)
  auto advance = [&]() -> bool {
    double s = 0;
    for (int k = 0; k < nplan; k++) {
      const int col = perm[k];
      double d = qe[col * B] - qw[col * ws];
      s = s + d * d;
    }
    const double mag = sqrt(s);
    degenerate = degenerate || !(mag > 0.0) || !(mag <= 1.79769313486231570815e+308);
    const double sm = step < mag ? step : mag;
    bool eq = true;
    for (int k = 0; k < nplan; k++) {
      const double ek = qe[k * B];
      double d = ek - qw[k * ws];
      double nw = qw[k * ws] + (d / mag) * sm;
      qw[k * ws] = nw;
      eq = eq && (nw == ek);
    }
    return eq;
  };
  // first walk: how many interior waypoints does this edge have?
  int K = 0;
  {
    bool walking = !done;
    while (__ballot(walking) != 0ull) {
      if (walking) {
        if (advance()) walking = false;
        else if (++K > ib.kmax) walking = false;
        if (degenerate) { walking = false; K = ib.kmax + 1; }  // the walking kernel reports it
      }
    }
  }
  if (!done && K > ib.kmax) {  // the estimate was off: the walking kernel takes the edge
    ib.llist[atomicAdd(ib.lcount, 1)] = (int)i;
    done = true;
  }
  if (done) K = 0;
  // one reservation per wave: lane l owns slots [base + sum_{l' < l} K_l', +K_l)
  int incl = K;
  for (int off = 1; off < 64; off <<= 1) {
    const int up = __shfl_up(incl, off);
    if (lane >= off) incl += up;
  }
  const int total = __shfl(incl, 63);
  if (total == 0) return;
  int base = 0;
  if (lane == 0) base = atomicAdd(ib.count, total);
  base = __builtin_amdgcn_readfirstlane(base);
  const int first = base + incl - K;
  if (!done && first + K > ib.cap) {  // out of item space: the walking kernel takes the edge
    ib.llist[atomicAdd(ib.lcount, 1)] = (int)i;
    for (int slot = first; slot < ib.cap; slot++) ib.edge[slot] = -1;  // reserved but void
    done = true;
  }
  // second walk: write the waypoints
  for (int k = 0; k < nplan; k++) qw[k * ws] = start_col(k);
  for (int idx = 1; __ballot(!done && idx <= K) != 0ull; idx++) {
    if (!done && idx <= K) {
      advance();
      const int slot = first + idx - 1;
      for (int k = 0; k < nplan; k++) ib.w[(size_t)slot * nplan + k] = qw[k * ws];
      ib.edge[slot] = (int)i;
      ib.idx[slot] = idx;
    }
  }
}

// Endpoint pass of the two-pass edge filter: check 0 (the endpoint QB, utils.py:144) for every
// edge with full lanes; edges whose endpoint is free (or undecided) are appended to `slist` for
// the interior pass, so that pass runs only on edges that still need it -- in an RRT batch a
// large share of the candidate edges ends inside an obstacle, and in the one-pass kernel their
// lanes idle through every later waypoint of the wave.
template <class Spec, int MAXS, bool WBOX, bool MBOX>
__global__ void __launch_bounds__(kBlock, kMinWaves<Spec>)
k_filter_endpoints(const int *__restrict__ gip, int nip, const float *__restrict__ gfp, int nfp,
                   const double *__restrict__ QA, const double *__restrict__ QB, int64_t E, int layout,
                   float tol, uint8_t *__restrict__ valid, int32_t *__restrict__ first_bad,
                   int *__restrict__ status, int *__restrict__ ulist, int *__restrict__ ucount,
                   UndecidedConfigs uc, int *__restrict__ slist, int *__restrict__ scount, ItemBuffers ib,
                   double step) {
  extern __shared__ double smem[];
  const int B = blockDim.x;
  const int nplan = gip[H_NPLAN];
  // (immediate interpreter + item expansion: a second column set holds the walking waypoint)
  Carve<float> c = carve_lds<float>(smem, gip, nip, gfp, nfp, nplan, (!kQueued<float, MBOX> && ib.count) ? 2 : 1, B);
  const int64_t i = (int64_t)blockIdx.x * B + threadIdx.x;
  const bool active = i < E;
  double *qw = c.col0 + threadIdx.x;
  load_columns(qw, B, QB, E, i, nplan, layout, active);
  __syncthreads();
  bool finite = true;
  for (int k = 0; k < nplan; k++) {
    const double a = active ? ((layout == MJPL_SOA) ? QA[(int64_t)k * E + i] : QA[i * nplan + k]) : 0.0;
    const double b = qw[k * B];
    finite = finite && (fabs(a) <= 1.79769313486231570815e+308) && (fabs(b) <= 1.79769313486231570815e+308);
  }
  const bool run = active && finite;
  const int code = check_one<float, MAXS, WBOX, MBOX, Spec>(c, qw, B, run, tol, i, uc, 0);
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
        const int j = kQueued<float, MBOX> ? uc.cap : atomicAdd(uc.count, 1);  // (queued: the whole edge)
        if (j < uc.cap) {
          for (int k = 0; k < nplan; k++) uc.q[(size_t)j * nplan + k] = qw[k * B];
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
  const unsigned long long m = __ballot(survive);
  if (m == 0ull) return;
  if (ib.count) {
    // lane-per-waypoint interior pass: emit this edge's interior waypoints as work items.  The
    // walking waypoint lives in this wave's (now idle) candidate-queue memory.
    if constexpr (kQueued<float, MBOX>) {
      if ((size_t)nplan * 64 * sizeof(double) <= WaveQueue<float>::bytes()) {
        double *scratch = reinterpret_cast<double *>(c.qmem + (size_t)(threadIdx.x >> 6) * WaveQueue<float>::bytes()) +
                          (threadIdx.x & 63);
        expand_edge(gip, QA, E, i, step, layout, survive, qw, B, scratch, 64, ib);
        if ((threadIdx.x & 63) == 0) atomicAdd(scount, (int)__builtin_popcountll(m));
        return;
      }
    } else {
      expand_edge(gip, QA, E, i, step, layout, survive, qw, B, c.col1 + threadIdx.x, B, ib);
      if ((threadIdx.x & 63) == 0) atomicAdd(scount, (int)__builtin_popcountll(m));
      return;
    }
    if (survive) ib.llist[atomicAdd(ib.lcount, 1)] = (int)i;  // too many columns: walking kernel
    return;
  }
  const int lane = threadIdx.x & 63;
  int base = 0;
  if (lane == 0) base = atomicAdd(scount, (int)__builtin_popcountll(m));
  base = __builtin_amdgcn_readfirstlane(base);
  if (survive)
    slist[base + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u))] = (int)i;
}

template <class Spec, int MAXS, bool WBOX, bool MBOX>
__global__ void __launch_bounds__(kBlock, kMinWaves<Spec>)
k_filter_configs(const int *__restrict__ gip, int nip, const float *__restrict__ gfp, int nfp,
                 const double *__restrict__ Q, int64_t N, int layout, float tol,
                 uint8_t *__restrict__ valid, int *__restrict__ ulist, int *__restrict__ ucount,
                 UndecidedConfigs uc) {
  extern __shared__ double smem[];
  const int B = blockDim.x;
  const int nplan = gip[H_NPLAN];
  Carve<float> c = carve_lds<float>(smem, gip, nip, gfp, nfp, nplan, 1, B);
  const int64_t i = (int64_t)blockIdx.x * B + threadIdx.x;
  const bool active = i < N;
  load_columns(c.col0 + threadIdx.x, B, Q, N, i, nplan, layout, active);
  __syncthreads();
  // queued interpreter: undecided pairs go to k_patch_pairs through `uc` (which clears valid[i]
  // on a contact); V_UNSURE comes back only for what could not be handed over
  const int code = check_one<float, MAXS, WBOX, MBOX, Spec>(c, c.col0 + threadIdx.x, B, active, tol, i, uc, 0);
  if (active) {
    if (code == V_UNSURE) ulist[atomicAdd(ucount, 1)] = (int)i;
    else valid[i] = (code == V_CONTACT) ? 0 : 1;
  }
}


#ifdef MJPL_X_ITEMS_4WAVES
#define MJPL_ITEMS_BOUNDS __launch_bounds__(kBlock, 4)
#else
#define MJPL_ITEMS_BOUNDS __launch_bounds__(kBlock, kMinWaves<Spec>)
#endif
template <class Spec, int MAXS, bool WBOX, bool MBOX>
__global__ void MJPL_ITEMS_BOUNDS
k_filter_items(const int *__restrict__ gip, int nip, const float *__restrict__ gfp, int nfp, ItemBuffers ib,
               float tol, uint8_t *__restrict__ valid, int32_t *__restrict__ first_bad,
               int *__restrict__ ulist, int *__restrict__ ucount, UndecidedConfigs uc) {
  extern __shared__ double smem[];
  const int B = blockDim.x;
  const int64_t n = *ib.count < ib.cap ? *ib.count : ib.cap;
  if ((int64_t)blockIdx.x * B >= n) return;
  const int nplan = gip[H_NPLAN];
  // the configuration is read straight from the item buffer (coalesced, once per joint): no LDS
  // columns in this kernel
  Carve<float> c = carve_lds<float>(smem, gip, nip, gfp, nfp, nplan, 0, B);
  const int64_t it = (int64_t)blockIdx.x * B + threadIdx.x;
  const bool active = it < n && ib.edge[it] >= 0;  // (a void slot: reserved by an edge that did not fit)
  __syncthreads();
  const int64_t itc = it < (int64_t)ib.cap ? it : 0;
  const int code = check_one<float, MAXS, WBOX, MBOX, Spec>(c, ib.w + itc * nplan, B, active, tol, it, uc, 0, ib.edge,
                                                      ib.idx, ib.w + (itc - (threadIdx.x & 63)) * nplan, nplan, 1);
  if (active && code != V_NONE) {
    const int ed = ib.edge[it];
    if (code == V_CONTACT) {
      valid[ed] = 0;
      if (first_bad) atomicMin(reinterpret_cast<unsigned *>(first_bad) + ed, (unsigned)ib.idx[it]);
    } else {
      bool handed = false;
      if constexpr (!kQueued<float, MBOX>) {
        // immediate interpreter: the whole configuration goes to the exact configuration kernel
        if (uc.count) {
          const int j = atomicAdd(uc.count, 1);
          if (j < uc.cap) {
            for (int k = 0; k < nplan; k++) uc.q[(size_t)j * nplan + k] = ib.w[itc * nplan + k];
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

}  // namespace mjpl
