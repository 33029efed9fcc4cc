// mjpl_hip.hip -- libmjpl_hip.so: gfx950 kernels + the C ABI of include/mjpl_hip.h.
//
// Replaces, for whole batches, the per-configuration body of the reference's
// CollisionConstraint.valid_config (src/mjpl/constraint/collision_constraint.py:26-30) and the
// edge discretisation of _valid_collision_interval (src/mjpl/planning/utils.py:188-216).
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared (see __graft_entry__.py).
// There is deliberately no CPU fallback in this file.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <dlfcn.h>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <map>
#include <memory>
#include <mutex>
#include <set>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

#include "../../include/mjpl_hip.h"
#include "mjpl_device.h"
#include "mjpl_filter.h"
#include "mjpl_pose.h"
#include "mjpl_project.h"
#include "mjpl_nearest.h"
#include "mjpl_nearest_cells.h"

namespace {

using namespace mjpl;

constexpr int kCtr = kCounterStride;  // ints between device counters: one 128-byte line each
constexpr int kNumCtr = kNumCounters;
constexpr size_t kFusedDbgWaves = 256 * 4 * 3 * 2;  // (wave slots of the chip, with room)

thread_local std::string g_err;

int fail(int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

#define HIP_TRY(expr)                                                                   \
  do {                                                                                  \
    hipError_t _e = (expr);                                                             \
    if (_e != hipSuccess)                                                               \
      return fail(MJPL_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),    \
                  __FILE__, __LINE__);                                                  \
  } while (0)

// ------------------------------------------------------------------------------- kernels

// ---- exact path (float64): final verdicts --------------------------------------------------

template <int MAXS, bool WBOX, bool MBOX>
__global__ void __launch_bounds__(kBlock)
k_check_configs(const int *__restrict__ gip, int nip, const double *__restrict__ gdp, int ndp,
                const double *__restrict__ Q, int64_t N, int layout, uint8_t *__restrict__ valid,
                unsigned long long *__restrict__ bits, const int *__restrict__ ulist,
                const int *__restrict__ ucount, UndecidedConfigs uc, int32_t *__restrict__ first_bad) {
  extern __shared__ double smem[];
  const int B = blockDim.x;
  if (ulist && (int64_t)blockIdx.x * B >= (int64_t)*ucount) return;  // nothing left to re-run
  const int nplan = gip[H_NPLAN];
  Carve<double> c = carve_lds<double>(smem, gip, nip, gdp, ndp, nplan, 1, B);
  FkOut none = {};
  if (uc.count) {
    // patch mode: the rows of uc.q are undecided waypoints of edges; their number is read on the
    // device and may exceed what the grid covers in one pass: blocks stride over it
    const int64_t n = *uc.count < uc.cap ? *uc.count : uc.cap;
    for (int64_t base = (int64_t)blockIdx.x * B; base < n; base += (int64_t)gridDim.x * B) {
      const int64_t i = base + threadIdx.x;
      const bool active = i < n && uc.ga[i] < 0;  // pair-level items belong to k_patch_pairs
      load_columns(c.col0 + threadIdx.x, B, Q, n, i < n ? i : 0, nplan, layout, active);
      const bool hit = run_config<double, MAXS, false, WBOX, MBOX>(c.ip, c.tp, c.col0 + threadIdx.x, B,
                                                                   c.save + threadIdx.x, B, active, 0.0, none,
                                                                   i) == V_CONTACT;
      if (active && hit) {
        const int ed = uc.edge[i];
        valid[ed] = 0;
        // first_bad holds -1 (= UINT_MAX) for "valid so far": an unsigned min keeps the lowest index
        if (first_bad) atomicMin(reinterpret_cast<unsigned *>(first_bad) + ed, (unsigned)uc.idx[i]);
      }
    }
    return;
  }
  int64_t i;
  const bool active = pick_item(N, ulist, ucount, &i);
  load_columns(c.col0 + threadIdx.x, B, Q, N, i, nplan, layout, active);
  __syncthreads();
  const bool hit = run_config<double, MAXS, false, WBOX, MBOX>(c.ip, c.tp, c.col0 + threadIdx.x, B,
                                                               c.save + threadIdx.x, B, active, 0.0, none,
                                                               i) == V_CONTACT;
  if (valid && active) valid[i] = hit ? 0 : 1;
  if (bits) {
    unsigned long long m = __ballot(active && !hit);
    if ((threadIdx.x & 63) == 0 && active) bits[i >> 6] = m;
  }
}

__global__ void __launch_bounds__(kBlock)
k_fk(const int *__restrict__ gip, int nip, const double *__restrict__ gdp, int ndp,
     const double *__restrict__ Q, int64_t N, int layout, FkOut out) {
  extern __shared__ double smem[];
  const int B = blockDim.x;
  const int nplan = gip[H_NPLAN];
  Carve<double> c = carve_lds<double>(smem, gip, nip, gdp, ndp, nplan, 1, B);
  const int64_t i = (int64_t)blockIdx.x * B + threadIdx.x;
  const bool active = i < N;
  load_columns(c.col0 + threadIdx.x, B, Q, N, i, nplan, layout, active);
  __syncthreads();
  run_config<double, 1, true, true, true>(c.ip, c.tp, c.col0 + threadIdx.x, B, c.save + threadIdx.x, B,
                                          active, 0.0, out, i);
}

template <int MAXS, bool WBOX, bool MBOX>
__global__ void __launch_bounds__(kBlock)
k_check_edges(const int *__restrict__ gip, int nip, const double *__restrict__ gdp, int ndp,
              const double *__restrict__ QA, const double *__restrict__ QB, int64_t E, double step,
              int layout, int flags, uint8_t *__restrict__ valid, int32_t *__restrict__ first_bad,
              int *__restrict__ status, const int *__restrict__ rlist, const int *__restrict__ rcount) {
  UndecidedConfigs none = {};
  edge_body<double, MAXS, WBOX, MBOX>(gip, nip, gdp, ndp, QA, QB, E, step, layout, flags, 0.0, valid,
                                      first_bad, status, nullptr, nullptr, rlist, rcount, none);
}

// ---- filter path (float32): decides what it can, lists the rest ----------------------------

template <int MAXS, bool WBOX, bool MBOX>
__global__ void __launch_bounds__(kBlock)
k_filter_edges(const int *__restrict__ gip, int nip, const float *__restrict__ gfp, int nfp,
               const double *__restrict__ QA, const double *__restrict__ QB, int64_t E, double step,
               int layout, int flags, float tol, uint8_t *__restrict__ valid,
               int32_t *__restrict__ first_bad, int *__restrict__ status, int *__restrict__ ulist,
               int *__restrict__ ucount, UndecidedConfigs uc, const int *__restrict__ rlist,
               const int *__restrict__ rcount, int *__restrict__ zero_next) {
  zero_counters(zero_next);  // (only when this is the first kernel of the launch: one-pass mode)
  edge_body<float, MAXS, WBOX, MBOX>(gip, nip, gfp, nfp, QA, QB, E, step, layout, flags, tol, valid,
                                     first_bad, status, ulist, ucount, rlist, rcount, uc);
}

// (pose_project_lane, k_pose_apply, k_rrt_gen_project: mjpl_project.h -- per-model libraries instantiate them, too)

__global__ void __launch_bounds__(kPoseBlock)
k_pose_valid(const int *__restrict__ pi, const double *__restrict__ pd, const double *__restrict__ Q,
             int64_t N, uint8_t *__restrict__ valid, double *__restrict__ xpos, double *__restrict__ xmat) {
  extern __shared__ double smem[];
  constexpr int B = kPoseBlock;
  const int lane = threadIdx.x;
  const int nq = pi[PH_NQ];
  const double *tail = pd + pi[PH_OFF_TAIL];
  const double *jrange = pd + pi[PH_OFF_JRANGE];
  double *qw = smem + lane;
  const int64_t i = (int64_t)blockIdx.x * B + lane;
  const bool active = i < N;
  bool inside = true;
  for (int k = 0; k < nq; k++) {
    const double v = active ? Q[i * nq + k] : 0.0;
    qw[k * B] = v;
    inside = inside && (v >= jrange[2 * k] && v <= jrange[2 * k + 1]);
  }
  PoseChainOut o;
  pose_chain(pi, pd, qw, B, nullptr, 0, o);
  double dx[6], qs[4];
  pose_displacement(tail, o, dx, qs);
  if (active) {
    if (valid) valid[i] = (inside && norm6(dx) <= tail[PT_TOL]) ? 1 : 0;
    if (xpos) for (int k = 0; k < 3; k++) xpos[i * 3 + k] = o.site_xpos[k];
    if (xmat) for (int k = 0; k < 9; k++) xmat[i * 9 + k] = o.site_xmat[k];
  }
}


// (k_ik_solve: mjpl_project.h -- interpreting, or around a model's generated chain)

// ------------------------------------------------------------------------------- host model

struct HostModel {
  int nq = 0, njnt = 0, nbody = 0, ngeom = 0;
  std::vector<int> body_parentid, body_weldid, body_jntadr, body_jntnum;
  std::vector<double> body_pos, body_quat;
  std::vector<int> jnt_type, jnt_qposadr;
  std::vector<double> jnt_axis, jnt_pos, qpos0;
  std::vector<int> geom_type, geom_bodyid, geom_contype, geom_conaffinity;
  std::vector<double> geom_size, geom_pos, geom_quat, geom_rbound, geom_margin;
};

template <class T>
std::vector<T> copy_n(const T *p, size_t n) {
  return std::vector<T>(p, p + n);
}

// entry points of a per-model specialised library (see load_spec)
struct SpecLib {
  typedef int (*ConfigsFn)(hipStream_t, unsigned, unsigned, size_t, const int *, int, const float *, int, const double *,
                           int64_t, int, float, uint8_t *, int *, int *, UndecidedConfigs, int *);
  typedef int (*EndpointsFn)(hipStream_t, unsigned, unsigned, size_t, const int *, int, const float *, int, const double *,
                             const double *, int64_t, int, float, uint8_t *, int32_t *, int *, int *, int *, UndecidedConfigs,
                             int *, int *, ItemBuffers, double, int *);
  typedef int (*PatchFn)(hipStream_t, unsigned, unsigned, size_t, const int *, int, const double *, int, GeomTable,
                         UndecidedConfigs, uint8_t *, int32_t *);
  typedef int (*ItemsFn)(hipStream_t, unsigned, unsigned, size_t, const int *, int, const float *, int, ItemBuffers, EdgeSource, float,
                         uint8_t *, int32_t *, int *, int *, UndecidedConfigs);
  typedef int (*EndpointsPwFn)(hipStream_t, size_t, const int *, int, const float *, int, const double *, const double *, int64_t, int,
                               float, uint8_t *, int32_t *, int *, int *, int *, UndecidedConfigs, ItemBuffers, double, int *, int *);
  typedef int (*ItemsPwFn)(hipStream_t, size_t, const int *, int, const float *, int, ItemBuffers, EdgeSource, float, uint8_t *,
                           int32_t *, int *, int *, UndecidedConfigs, int *);
  typedef int (*TailFn)(hipStream_t, size_t, TailArgs);
  typedef int (*FusedFn)(hipStream_t, int, size_t, FusedArgs);  // (waves per workgroup, LDS bytes, arguments)
  FusedFn fused = nullptr;
  FusedFn fused_f64 = nullptr;  // the float64 pool kernel around the model's generated check (mjpl_spec_launch_fused_f64), if it has one
  int fused_waves = kFusedWaves;  // what its fused kernel was built for (mjpl_spec_fused_waves)
  bool fused_cert = false;        // ... and whether with the edge certificate (mjpl_spec_fused_cert): LDS rows of |QB - QA|
  // generated PoseConstraint projections, one per site body of the model (mjpl_project.h): chain hashes and launchers
  // (row kernels of mjpl_rows.h: projection index, lanes per row, stream, waves, ...; 0 launched, -1 refused, -2 HIP error)
  typedef int (*PoseApplyFn)(int, int, hipStream_t, unsigned, const int *, const double *, const double *, const double *, int64_t, int64_t,
                             double *, uint8_t *, int32_t *, PosePhase);
  typedef int (*GenProjectFn)(int, int, hipStream_t, unsigned, int, int, int, double, int, const int *, const double *, const int *,
                              const double *, const uint8_t *, const double *, const double *, const double *, RrtLanes, RrtCand, int *);
  typedef int (*IkSolveFn)(int, int, hipStream_t, unsigned, const int *, const double *, const double *, int64_t, int64_t, double *, uint8_t *,
                           int32_t *, double *, int, unsigned long long);
  int pose_count = 0;
  unsigned long long (*pose_hash)(int) = nullptr;
  PoseApplyFn pose_apply = nullptr;
  GenProjectFn gen_project = nullptr;
  IkSolveFn ik_solve = nullptr;
  EndpointsPwFn endpoints_pw = nullptr;
  ItemsPwFn items_pw = nullptr;
  TailFn tail = nullptr;
  void *lib = nullptr;
  int generic_rows = 0, generic_stages = 0;  // scene-generic libraries: rows per moving geom, moving geoms
  ConfigsFn configs = nullptr;
  EndpointsFn endpoints = nullptr;
  ItemsFn items = nullptr;
  PatchFn patch = nullptr;
};

}  // namespace

struct mjpl_engine {
  int device = 0;
  hipStream_t stream = nullptr;
  hipDeviceProp_t prop{};
  HostModel m;
  std::set<std::pair<int, int>> allowed;  // sorted body-id pairs (collision_constraint.py:60-64)
  std::vector<int> qidx;
  std::vector<double> qbase;
  // compiled program
  std::vector<int> ip;
  std::vector<double> dp;
  std::vector<float> fp;  // the filter's float32 image of dp (cull bounds widened by filter_tol)
  int *d_ip = nullptr;
  double *d_dp = nullptr;
  float *d_fp = nullptr;
  int *d_status = nullptr;
  // float32 filter + exact re-run of what it cannot decide
  bool filter = true;
  float filter_tol = 1e-4f;       // tolerance band in force (>= filter_tol_req)
  float filter_tol_req = 1e-4f;   // what the caller (or the default) asked for
  bool filter_tol_user = false;   // asked for through mjpl_set_filter / MJPL_FILTER_TOL
  bool filter_usable = true;      // false: this model's binary32 error floor is too high, exact path only
  double ferr_a = 0, ferr_b = 0;  // |pose error| <= ferr_a + ferr_b * max coordinate
  double fmax_coord = 0;
  int npoisoned = 0;              // static geoms too large / far for binary32: their pairs are always undecided
  uint64_t program_hash = 0;      // FNV-1a of the compiled tables (ip, fp, dp), the kernel variant and the header digest
  const SpecLib *spec = nullptr;  // this model's own filter kernels, if a library for program_hash was found
  // ... or, failing that, a scene-generic library of the ROBOT (robot_hash: moving bodies, their geoms and
  // self pairs, planning set, tolerance -- nothing of the static geoms): its generated code takes every
  // static partner's cull row from a table in front of the float32 tables (scene_floats of them), so the
  // obstacles may change without a compiler
  bool spec_generic = false;
  uint64_t robot_hash = 0;
  int moving_base = 0;           // model id of the first moving geom
  std::vector<float> scene;      // [scene header | per moving geom: kSceneRows rows of 8] (compile_program)
  float *d_fp_base = nullptr;    // allocation behind d_fp (= d_fp_base + scene.size())
  bool spec_off = false;          // mjpl_set_spec(e, 0): run the interpreting kernels whatever libraries exist
  bool spec_generic_only = false; // mjpl_set_spec(e, 2): pass over the program's own library, take the robot's scene-generic one
  int *d_ulist = nullptr;   // items (configurations / whole edges) the filter left undecided
  int *d_ucount_base = nullptr;  // both counter sets; d_ucount = the one the last launch used
  bool counters_stale = false;
  int *d_ucount = nullptr;  // [0] how many of those, [1] undecided waypoints of edges, [2] edges in d_slist
  int *d_slist = nullptr;   // two-pass edge filter: edges whose endpoint passed
  size_t slist_cap = 0;
  bool two_pass = true;
  size_t ulist_cap = 0;
  // timing runs (mjpl_time_edges_stages_dev): marks[k] is recorded after stage k - 1 of the launch
  // (marks[0] at its start); nullptr in ordinary launches
  hipEvent_t *marks = nullptr;
  // option "kernel_timer" (bench.py: every line's roofline from THIS run), a bit mask: 1 = the dominant kernel of the configuration /
  // projection / IK / planner-extension launches is bracketed with events, 2 = the nearest-neighbour scans; "kernel_timer_ms" /
  // "kernel_timer2_ms" read the sums since the option was set, "kernel_timer[2]_launches" how many
  int kt_mode = 0, kt_used[2] = {0, 0};  // (kt_mode: bit 0 = class 1, bit 1 = class 2)
  std::vector<hipEvent_t> kt_ev[2];
  double *d_ucq = nullptr;  // undecided waypoints: rows of nplan float64
  int *d_ucedge = nullptr, *d_ucidx = nullptr, *d_ucga = nullptr, *d_ucgb = nullptr;
  double *d_geomtab = nullptr;  // GTB_LEN doubles per model geom (k_patch_pairs)
  // lane-per-waypoint interior pass: (edge, idx) items, long-edge list
  double *d_tstep = nullptr, *d_itemck = nullptr;  // (checkpoint rows: allocated by the first launch with long items)
  size_t itemck_cap = 0;
  int *d_itemedge = nullptr, *d_itemidx = nullptr, *d_llist = nullptr, *d_icount = nullptr;
  int *d_eclaim = nullptr;  // [llist_cap] per-edge claim word: the launch generation that listed the edge in d_ulist
  int claim_gen = 0;
  size_t item_cap = 0, llist_cap = 0;
  bool expand = true;
  // endpoint and item kernels as persistent grids of waves with tile queues (queued interpreter only).
  // -1: where it pays -- a model's own specialised kernels (0.257 -> 0.245 ms per step); the interpreting
  // kernels, whose waves differ more in what a tile costs them, are faster as ordinary grids (0.397 vs
  // 0.459 ms).  MJPL_PERSIST=0 / 1 forces either (tests run both ways).
  int persist = -1;
  bool fused_tail = true;  // MJPL_TAIL: walking kernel, pair re-check and exact edge kernel as roles of one launch (k_tail)
  // ONE filter kernel per edge launch (k_edges_fused, mjpl_fused.h): endpoint and waypoint tiles served by the same
  // resident workgroups from a work pool in LDS.  MJPL_FUSED=0 restores the two persistent kernels;
  // MJPL_FUSED_POLICY (bit 0: item tiles first), MJPL_FUSED_KMAX: A/B measurements
  bool fused = true;
  int fused_policy = 0, fused_kmax = 4096, fused_pool_cap = 0;  // (MJPL_FUSED_POOL: at most that many ring slots)
  bool f64_spec = true;  // the float64 pool kernel around the library's generated check, when it has one (MJPL_F64_SPEC=0: interpreting)
  const SpecLib *spec_cert = nullptr;  // the model's library built WITH the edge certificate (spec/cert/), if there is one ...
  int64_t fused_cert_min_edges = (int64_t)1 << 20;  // ... takes the fused launches of at least this many edges (option "fused_cert_min_edges"; 0: never)
  int fused_cert = 1;  // the fused kernel's edge certificate (mjpl_fused.h; MJPL_FUSED_CERT=0: every surviving edge's waypoints are checked)
  bool fused_mbox = false;
  bool f64_queued = true;  // MJPL_F64_QUEUED: the float64 pool kernel checks through the candidate queues (A/B switch)
  int fused_single_max = 32768;  // MJPL_FUSED_SINGLE: batches up to this many edges check every configuration of an edge in one round (measured: 0.068 vs 0.087 ms at 1 024 edges, 0.090 vs 0.102 at 32 768, 0.116 vs 0.106 at 65 536)
  bool fused_skip_once = false;  // mjpl_check_edges: this launch holds a few long edges -> the kernels with checkpoints
  const char *fused_dbg_path = nullptr;  // MJPL_FUSED_DEBUG=<file> (with a -DMJPL_FUSED_DEBUG build of the kernels)
  unsigned long long *d_fused_dbg = nullptr;
  size_t item_cap_limit = (size_t)1 << 26;  // MJPL_ITEM_CAP: edges beyond it take the walking kernel
  void *d_nn = nullptr;         // nearest neighbour: per-chunk partial results
  size_t nn_bytes = 0;
  void *d_nn16 = nullptr;       // ... the matrix-core screen's operands (binary16 rows) and partial results
  size_t nn16_bytes = 0;
  int nn_mfma = 1;              // MJPL_NN_MFMA=0: binary32 screen only
  void *d_nn_tmp = nullptr; size_t nn_tmp_bytes = 0;  // distances of a ranged look-up whose caller wants none
  int64_t nn_reserve_nodes = 0; // packed-node rows the screened look-up's scratch is sized for at least (mjpl_rrt_create: the planner's capacity, 2^23 at most)
  int64_t nn_sample = 65536;    // MJPL_NN_SAMPLE: nodes of the strided sample the matrix cores take every query's bound from
  // switches that used to be read from the environment wherever they were used (round 6: options, include/mjpl_hip.h)
  size_t zero_copy_bytes = (size_t)256 << 10;  // "zero_copy_bytes"
  int rows_g = 0;               // "rows_g": lanes per row of the projection / IK row kernels forced (0: by the batch size)
  int pose_spec = 1;            // "pose_spec": 0 = projections and IK seeds on the interpreting kernels
  int pose_phase_steps = 0;     // "pose_phase_steps"
  int rrt_trace = 0;            // "rrt_trace": 1 = a planner round's phases on stderr, 2 = every extension chunk (synchronises)
  // ... and the planner's (mjpl_rrt_create copies them into the handle it makes)
  // (rrt_early_nn: the look-ups on a second stream beside the first extension's last chunks -- round 5's answer to tails of a
  //  thousand steps.  With the chains capped the last chunks are a millisecond and the look-ups two or three: rounds of
  //  13.0 - 13.5 ms with it at any threshold, 13.1 without (profiles/r06_rrt_early_ab.txt, one box): OFF by default from round 6
  //  on -- one stream, one scratch arena --, still an option and still tested.)
  int rrt_exact_counts = 1, rrt_early_nn = 0, rrt_early_lanes = 4096, rrt_early_next = 1, rrt_proj_steps = 1024, rrt_proj_g = 0,
      rrt_ahead = 1, rrt_ahead_lanes = 1024, rrt_proj_waves = 1024;
  int64_t rrt_early_min_nodes = 65536, rrt_proj_slots = 1 << 20;
  int nn_cells = 1;             // the cell-ordered scan for big trees (mjpl_nearest_cells.h); option "nn_cells"
  int64_t nn_cells_min = 131072; // ... from this many nodes on; option "nn_cells_min_nodes"
  int64_t nn_cells_sample = 32768; // ... with bounds from a strided sample of this many nodes ("nn_cells_sample"; measured on the planner's trees,
                                 //     ms per look-up targets / connect phase: 8 192: 4.79 / 2.83, 16 384: 4.55 / 2.61, 32 768: 4.64 / 2.68, 65 536: 5.02 / 2.83)
  // options, both OFF: every query's bound tightened on its home sub-chunks before the candidate pass; a binary32 second screen
  // of the parked pairs.  Measured on one box, 14 planner rounds (profiles/README.md round 6): the second screen 21.2 ms per
  // round against 19.5 without (it evaluates every parked pair twice where the first evaluation decides most), the home pass
  // 19.6 (what it saves the scan it costs itself).  Kept for trees and batches of other shapes.
  int nn_home = 0, nn_second_screen = 0;
  int nn_last_cells = 0;        // the last look-up took it
  int nn_probe = 0;             // option "nn_probe" (2: count the exact evaluations)
  const int32_t *nn_last_count = nullptr; int nn_last_waves = 0, nn_last_nsub = 0;  // the last cell-ordered scan's candidate counts
  int nn_last = 0;              // what the last mjpl_nearest_dev launched: 0 float64 scan, 1 binary32 screen, 2 matrix-core screen (or 1: see its flag)
  size_t uc_cap = 0, uc_cap_limit = 0;
  int nslots = 0, nsave = 0, maxs = 4;
  bool wbox = false, mbox = false;
  // the immediate (non-queued) interpreter serves models with moving boxes and models that keep
  // more than 16 geoms in the slot file (one general <32, true, true> build)
  // Which builds a model runs.  Exact kernels: <4|8|16, wbox, false>, or the general <32, true, true>
  // for moving boxes / more than 16 stored geoms.  Filter kernels: the same small builds, then
  // <24, true, true> -- still the queued interpreter, moving boxes through its box queue -- and
  // only beyond 24 stored geoms the immediate interpreter <32, true, true>.
  bool exact_general() const { return mbox || maxs > 16; }
  bool force_immediate = false;  // MJPL_FORCE_IMMEDIATE (create-time, tests)
  bool immediate() const { return nslots > kQueuedMaxSlots || (force_immediate && exact_general()); }
  bool filter_mbox() const { return exact_general() && !immediate(); }
  int npairs = 0, npairs_world = 0, nmoving = 0, nstatic = 0;
  // static poses for FK output
  std::vector<double> st_xpos, st_xquat, st_gxpos, st_gxmat;
  std::vector<char> body_static, geom_static;
  // RCCL communicator of the frontier planner's exchange (mjpl_comm_init); none = a world of one
  void *comm = nullptr;
  int comm_rank = 0, comm_world = 1;
  // grow-only staging buffers for the host-pointer entry points
  void *stage[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};  // ([6]: the parked-row list of a two-launch projection)
  size_t stage_bytes[7] = {0, 0, 0, 0, 0, 0, 0};
  // pinned, grow-only host block of the fused small-batch path of the host-pointer entry points
  // (one H2D and one D2H per call instead of five copies from / to pageable memory)
  void *h_pin = nullptr;
  size_t h_pin_bytes = 0;
};

namespace {

void load_spec(mjpl_engine *e, bool generic_ok, int nstage);

int stage_reserve(mjpl_engine *e, int k, size_t bytes) {
  if (bytes <= e->stage_bytes[k]) return MJPL_OK;
  if (e->stage[k]) HIP_TRY(hipFree(e->stage[k]));
  e->stage[k] = nullptr;
  e->stage_bytes[k] = 0;
  size_t want = std::max<size_t>(bytes, 1 << 16);
  HIP_TRY(hipMalloc(&e->stage[k], want));
  e->stage_bytes[k] = want;
  return MJPL_OK;
}

// ---- per-model specialised filter kernels (mjpl_amd/specialise.py builds them) ----
// A library libmjpl_spec_<hash>.so holds the three float32 filter kernels of mjpl_filter.h instantiated
// with generated straight-line code for ONE compiled program.  It is looked up by the program's hash
// in $MJPL_SPEC_DIR (default: the directory `spec` next to this library); MJPL_SPEC=0 disables it.
// The cache belongs to the process, engines to their threads ("one engine per thread", include/mjpl_hip.h):
// every look-up and insertion happens under spec_mutex().  Entries are never erased and std::map never moves
// its nodes, so the pointer an engine keeps stays valid without the lock.
std::map<uint64_t, SpecLib> &spec_cache() {
  static std::map<uint64_t, SpecLib> c;
  return c;
}
std::mutex &spec_mutex() {
  static std::mutex m;
  return m;
}

// mjpl_set_spec_dir: per-model libraries are looked for there instead of beside this library (process-wide)
std::string &spec_dir_store() {
  static std::string d;
  return d;
}
std::string spec_dir_override() {
  std::lock_guard<std::mutex> lock(spec_mutex());
  return spec_dir_store();
}

// dlopen <dir>/<prefix><hash>.so and check that it was built for this hash from these headers
const SpecLib *find_spec(uint64_t hash, bool generic, const char *subdir = nullptr) {
  std::string dir_override = spec_dir_override();
  std::lock_guard<std::mutex> lock(spec_mutex());
  auto &cache = spec_cache();
  uint64_t key = hash ^ (generic ? 0x9e3779b97f4a7c15ull : 0ull);
  if (subdir) {  // (a build of another kind beside the default ones: spec/cert/ holds the edge-certificate builds)
    if (dir_override.empty()) {
      Dl_info info0;
      if (dladdr((const void *)&mjpl_version, &info0) && info0.dli_fname) {
        dir_override = info0.dli_fname;
        const size_t slash0 = dir_override.rfind('/');
        dir_override = (slash0 == std::string::npos ? std::string(".") : dir_override.substr(0, slash0)) + "/spec";
      }
    }
    dir_override += std::string("/") + subdir;
  }
  const char *env_dir = dir_override.empty() ? nullptr : dir_override.c_str();
  if (env_dir)  // (libraries of another directory are other libraries: e.g. the certificate builds under spec/cert)
    for (const char *c = env_dir; *c; c++) key = (key ^ (uint64_t)(unsigned char)*c) * 0x100000001b3ull;
  auto it = cache.find(key);
  if (it != cache.end()) return it->second.lib ? &it->second : nullptr;
  SpecLib sl;
  std::string dir;
  if (env_dir) {
    dir = env_dir;
  } else {
    Dl_info info;
    if (dladdr((const void *)&mjpl_version, &info) && info.dli_fname) {
      dir = info.dli_fname;
      const size_t slash = dir.rfind('/');
      dir = (slash == std::string::npos ? std::string(".") : dir.substr(0, slash)) + "/spec";
    }
  }
  char name[64];
  snprintf(name, sizeof(name), "/libmjpl_spec%s_%016llx.so", generic ? "g" : "", (unsigned long long)hash);
  void *lib = dir.empty() ? nullptr : dlopen((dir + name).c_str(), RTLD_NOW | RTLD_LOCAL);
  if (lib) {
    auto abi = (int (*)())dlsym(lib, "mjpl_spec_abi");
    auto hashf = (unsigned long long (*)())dlsym(lib, "mjpl_spec_hash");
    auto stamp = (unsigned long long (*)())dlsym(lib, "mjpl_spec_src_stamp");
    auto gen = (int (*)())dlsym(lib, "mjpl_spec_generic");  // 0, or rows per moving geom << 8 | moving geoms
    sl.configs = (SpecLib::ConfigsFn)dlsym(lib, "mjpl_spec_launch_configs");
    sl.endpoints = (SpecLib::EndpointsFn)dlsym(lib, "mjpl_spec_launch_endpoints");
    sl.items = (SpecLib::ItemsFn)dlsym(lib, "mjpl_spec_launch_items");
    sl.patch = (SpecLib::PatchFn)dlsym(lib, "mjpl_spec_launch_patch");
    sl.endpoints_pw = (SpecLib::EndpointsPwFn)dlsym(lib, "mjpl_spec_launch_endpoints_pw");
    sl.items_pw = (SpecLib::ItemsPwFn)dlsym(lib, "mjpl_spec_launch_items_pw");
    sl.tail = (SpecLib::TailFn)dlsym(lib, "mjpl_spec_launch_tail");
    sl.fused = (SpecLib::FusedFn)dlsym(lib, "mjpl_spec_launch_fused");
    sl.fused_f64 = (SpecLib::FusedFn)dlsym(lib, "mjpl_spec_launch_fused_f64");
    if (auto fw = (int (*)())dlsym(lib, "mjpl_spec_fused_waves")) sl.fused_waves = fw();
    if (auto fc = (int (*)())dlsym(lib, "mjpl_spec_fused_cert")) sl.fused_cert = fc() != 0;
    sl.pose_hash = (unsigned long long (*)(int))dlsym(lib, "mjpl_spec_pose_hash");
    sl.pose_apply = (SpecLib::PoseApplyFn)dlsym(lib, "mjpl_spec_launch_pose_apply");
    sl.gen_project = (SpecLib::GenProjectFn)dlsym(lib, "mjpl_spec_launch_gen_project");
    sl.ik_solve = (SpecLib::IkSolveFn)dlsym(lib, "mjpl_spec_launch_ik_solve");
    if (auto pc = (int (*)())dlsym(lib, "mjpl_spec_pose_count"))
      sl.pose_count = (sl.pose_hash && sl.pose_apply && sl.gen_project && sl.ik_solve) ? pc() : 0;
    // (the stamp: both libraries built from the same mjpl_filter.h / mjpl_device.h / mjpl_trig.h -- the
    // structs that cross this boundary by value and the table layouts live there)
    const int g = gen ? gen() : 0;
    if (abi && hashf && stamp && abi() == MJPL_SPEC_ABI && stamp() == (unsigned long long)MJPL_SRC_STAMP && hashf() == hash &&
        (g != 0) == generic && sl.configs && sl.endpoints && sl.items && sl.patch && sl.endpoints_pw && sl.items_pw && sl.tail && sl.fused) {
      sl.lib = lib;
      sl.generic_rows = g >> 8;
      sl.generic_stages = g & 255;
    } else {
      dlclose(lib);
    }
  }
  auto ins = cache.emplace(key, sl).first;
  return ins->second.lib ? &ins->second : nullptr;
}

// `generic_ok`: the program satisfies what a scene-generic library assumes (compile_program)
void load_spec(mjpl_engine *e, bool generic_ok = false, int nstage = 0) {
  e->spec = nullptr;
  e->spec_generic = false;
  if (e->spec_off || e->immediate() || (e->exact_general() && !e->filter_mbox()) || !e->filter_usable) return;  // (the generator covers the queued builds)
  e->spec_cert = nullptr;
  if (!e->spec_generic_only) e->spec = find_spec(e->program_hash, false);
  // (round 6: the certificate pays from about a million edges per launch on -- +7 ... 11 % -- and costs 4 % at 262 144,
  //  profiles/README.md round 5: both builds are loaded, launch_edges picks by the batch's size)
  if (e->spec && !e->spec->fused_cert) {
    const SpecLib *c = find_spec(e->program_hash, false, "cert");
    if (c && c->fused_cert && c->fused_waves == e->spec->fused_waves) e->spec_cert = c;
  }
  if (e->spec || !generic_ok) return;
  const SpecLib *g = find_spec(e->robot_hash, true);
  if (g && g->generic_rows == kSceneRows && g->generic_stages == nstage) {
    e->spec = g;
    e->spec_generic = true;
  }
}

int pin_reserve(mjpl_engine *e, size_t bytes) {
  if (bytes <= e->h_pin_bytes) return MJPL_OK;
  if (e->h_pin) HIP_TRY(hipHostFree(e->h_pin));
  e->h_pin = nullptr;
  e->h_pin_bytes = 0;
  const size_t want = std::max<size_t>(bytes, 1 << 16);
  HIP_TRY(hipHostMalloc(&e->h_pin, want));
  e->h_pin_bytes = want;
  return MJPL_OK;
}
constexpr size_t kFusedHostBytes = (size_t)256 << 10;  // batches up to this size take the fused path
// ... and up to this size the kernels read the pinned block and write into it themselves, over the bus: no copy
// operation on the stream at all (a scalar valid_config pays launches and one synchronisation; MJPL_ZERO_COPY_BYTES)
size_t zero_copy_bytes(const mjpl_engine *e) { return e->zero_copy_bytes; }  // (option "zero_copy_bytes")

// mj_collision pair filters [MJ-recalled: engine_collision_driver.c filterBitmask /
// filterBodyPair] + the a6 ruleset folded in.  returns true if the pair is tested.
bool pair_enabled(const mjpl_engine *e, int g1, int g2) {
  const HostModel &m = e->m;
  const int ct1 = m.geom_contype[g1], ca1 = m.geom_conaffinity[g1];
  const int ct2 = m.geom_contype[g2], ca2 = m.geom_conaffinity[g2];
  if (!(ct1 & ca2) && !(ct2 & ca1)) return false;
  const int b1 = m.geom_bodyid[g1], b2 = m.geom_bodyid[g2];
  const int w1 = m.body_weldid[b1], w2 = m.body_weldid[b2];
  if (w1 == w2) return false;
  const int wp1 = m.body_weldid[m.body_parentid[w1]];
  const int wp2 = m.body_weldid[m.body_parentid[w2]];
  if (w1 != 0 && w2 != 0 && (w1 == wp2 || w2 == wp1)) return false;
  const int t1 = m.geom_type[g1], t2 = m.geom_type[g2];
  if (t1 == GT_PLANE && t2 == GT_PLANE) return false;  // no collision function
  // CollisionRuleset: a contact between an allowed body pair never invalidates
  if (e->allowed.count({std::min(b1, b2), std::max(b1, b2)})) return false;
  return true;
}

bool type_supported(int t) { return t == GT_PLANE || t == GT_SPHERE || t == GT_CAPSULE || t == GT_BOX; }

// Compile the model into the ip/dp program the kernels interpret.
int compile_program(mjpl_engine *e) {
  const HostModel &m = e->m;
  const int nb = m.nbody, ng = m.ngeom;

  for (int j = 0; j < m.njnt; j++)
    if (m.jnt_type[j] != JT_SLIDE && m.jnt_type[j] != JT_HINGE)
      return fail(MJPL_E_JOINT, "joint %d has type %d; only slide(2)/hinge(3) are supported", j,
                  m.jnt_type[j]);

  // ---- static (world-welded) bodies: poses folded here with the kernels' own arithmetic
  e->body_static.assign(nb, 0);
  e->st_xpos.assign(3 * nb, 0.0);
  e->st_xquat.assign(4 * nb, 0.0);
  std::vector<double> st_xmat(9 * nb, 0.0);
  e->st_xquat[0] = 1.0;
  st_xmat[0] = st_xmat[4] = st_xmat[8] = 1.0;
  e->body_static[0] = 1;
  for (int b = 1; b < nb; b++) {
    if (m.body_weldid[b] != 0) continue;
    if (m.body_jntnum[b] != 0) return fail(MJPL_E_ARG, "body %d is welded to the world but has joints", b);
    const int p = m.body_parentid[b];
    if (!e->body_static[p]) return fail(MJPL_E_ARG, "body %d: weld id 0 below a moving parent", b);
    e->body_static[b] = 1;
    double np[3], nq[4];
    mul_mat_vec3(np, &st_xmat[9 * p], &m.body_pos[3 * b]);
    for (int k = 0; k < 3; k++) np[k] += e->st_xpos[3 * p + k];
    mul_quat(nq, &e->st_xquat[4 * p], &m.body_quat[4 * b]);
    normalize4(nq);
    for (int k = 0; k < 3; k++) e->st_xpos[3 * b + k] = np[k];
    for (int k = 0; k < 4; k++) e->st_xquat[4 * b + k] = nq[k];
    quat2mat(&st_xmat[9 * b], nq);
  }

  // ---- geoms
  e->geom_static.assign(ng, 0);
  e->st_gxpos.assign(3 * ng, 0.0);
  e->st_gxmat.assign(9 * ng, 0.0);
  std::vector<int> world_row(ng, -1);
  std::vector<double> wcull_tab, wnarrow_tab;
  std::vector<int> winfo;
  std::vector<std::pair<size_t, int>> info_at;  // dp index -> int stored there (first 4 bytes)
  std::vector<size_t> sq_bound_at, plane_bound_at;  // dp indices of cull bounds
  std::vector<int> poison_rows;                     // world rows whose binary32 narrowphase data is NaN
  e->nstatic = e->nmoving = 0;
  for (int g = 0; g < ng; g++) {
    const int b = m.geom_bodyid[g];
    if (!type_supported(m.geom_type[g])) {
      // a geom that can never collide is harmless; otherwise refuse
      bool used = false;
      for (int h = 0; h < ng && !used; h++)
        if (h != g) used = pair_enabled(e, std::min(g, h), std::max(g, h));
      if (used) return fail(MJPL_E_PAIRTYPE, "geom %d has unsupported type %d", g, m.geom_type[g]);
    }
    if (!e->body_static[b]) { e->nmoving++; continue; }
    e->geom_static[g] = 1;
    e->nstatic++;
    double gp[3], gq[4];
    mul_mat_vec3(gp, &st_xmat[9 * b], &m.geom_pos[3 * g]);
    for (int k = 0; k < 3; k++) e->st_gxpos[3 * g + k] = gp[k] + e->st_xpos[3 * b + k];
    mul_quat(gq, &e->st_xquat[4 * b], &m.geom_quat[4 * g]);
    quat2mat(&e->st_gxmat[9 * g], gq);
    world_row[g] = (int)winfo.size();
    {
      const double *gmx = &e->st_gxmat[9 * g];
      double rc[WC_LEN] = {0}, rn[WN_LEN] = {0};
      for (int k = 0; k < 3; k++) {
        rc[WC_POS + k] = e->st_gxpos[3 * g + k];
        rn[WN_XAXIS + k] = gmx[3 * k + 0];
        rn[WN_YAXIS + k] = gmx[3 * k + 1];
        rn[WN_ZAXIS + k] = gmx[3 * k + 2];
        rn[WN_SIZE + k] = m.geom_size[3 * g + k];
      }
      const int32_t info[2] = {m.geom_type[g] | (g << 8), 0};
      memcpy(&rc[WC_INFO], info, sizeof(double));
      const int w = (int)winfo.size();
      winfo.push_back(info[0]);
      // four rows side by side per chunk (wc_at); one spare chunk: the kernels may prefetch ahead
      wcull_tab.resize((size_t)((w >> 2) + 2) * 16, 0.0);
      for (int f = 0; f < WC_LEN; f++) wcull_tab[wc_at(w, f)] = rc[f];
      wnarrow_tab.insert(wnarrow_tab.end(), rn, rn + WN_LEN);
    }
  }
  const int nworld = (int)winfo.size();
  const int nwpad = (nworld + 3) / 4 * 4;
  wcull_tab.resize((size_t)(nwpad + 4) * WC_LEN, 0.0);
  if (nworld > 64) return fail(MJPL_E_CAPACITY, "%d static geoms; this build enables at most 64 per moving geom", nworld);
  if (ng >= (1 << 23)) return fail(MJPL_E_CAPACITY, "too many geoms");

  // ---- moving bodies in id order (parents precede children)
  std::vector<int> order;
  for (int b = 1; b < nb; b++)
    if (!e->body_static[b]) order.push_back(b);
  std::vector<int> save_slot(nb, -1);
  int nsave = 0;
  for (size_t k = 0; k < order.size(); k++) {
    const int p = m.body_parentid[order[k]];
    if (e->body_static[p]) continue;
    if (k > 0 && order[k - 1] == p) continue;
    if (save_slot[p] < 0) save_slot[p] = nsave++;
  }

  // moving geoms in processing order, their partners, and register-slot allocation
  std::vector<int> mgeoms;
  for (int b : order)
    for (int g = 0; g < ng; g++)
      if (m.geom_bodyid[g] == b) mgeoms.push_back(g);
  const int nm = (int)mgeoms.size();
  std::vector<std::vector<int>> stored_partners(nm), world_partners(nm);
  std::vector<int> last_user(nm, -1);
  e->npairs = e->npairs_world = 0;
  e->wbox = e->mbox = false;
  auto note_pair = [&](int ga, int gb) {  // ga is the moving geom being placed
    if (m.geom_type[ga] == GT_BOX) e->mbox = true;
    if (m.geom_type[gb] == GT_BOX) (e->geom_static[gb] ? e->wbox : e->mbox) = true;
  };
  for (int k = 0; k < nm; k++) {
    const int g = mgeoms[k];
    for (int s = 0; s < ng; s++)
      if (e->geom_static[s] && pair_enabled(e, std::min(g, s), std::max(g, s))) {
        world_partners[k].push_back(s);
        note_pair(g, s);
        e->npairs++; e->npairs_world++;
      }
    for (int k2 = 0; k2 < k; k2++) {
      const int h = mgeoms[k2];
      if (!pair_enabled(e, std::min(g, h), std::max(g, h))) continue;
      stored_partners[k].push_back(k2);
      note_pair(g, h);
      last_user[k2] = k;
      e->npairs++;
    }
    if (m.geom_type[g] == GT_PLANE) return fail(MJPL_E_PAIRTYPE, "plane geom %d on a moving body", g);
  }
  // register slots: one per kept sphere/capsule, two per kept box; a slot is reusable once the
  // last geom that needs its occupant has been processed (a geom is stored after its own tests)
  std::vector<int> slot_of(nm, -1);
  {
    std::vector<int> free_at;  // slot -> index of the last geom that reads it
    auto take = [&](int k) {
      for (size_t t = 0; t < free_at.size(); t++)
        if (free_at[t] <= k) { free_at[t] = last_user[k]; return (int)t; }
      free_at.push_back(last_user[k]);
      return (int)free_at.size() - 1;
    };
    for (int k = 0; k < nm; k++) {
      if (last_user[k] < 0) continue;
      const int s1 = take(k);
      const int s2 = (m.geom_type[mgeoms[k]] == GT_BOX) ? take(k) : (int)SLOT_NONE;
      slot_of[k] = s1 | (s2 << 6);
    }
    e->nslots = (int)free_at.size();
  }
  if (e->nslots > MAX_SLOTS)
    return fail(MJPL_E_CAPACITY, "%d moving geoms must be held at once; this build has %d register slots",
                e->nslots, (int)MAX_SLOTS);
  e->maxs = e->nslots <= 4 ? 4 : (e->nslots <= 8 ? 8 : (e->nslots <= 16 ? 16 : 32));  // vector widths with indirect addressing
  e->nsave = nsave;

  // ---- emit
  std::vector<int> &ip = e->ip;
  std::vector<double> &dp = e->dp;
  ip.assign(H_SIZE, 0);
  dp.clear();
  const int nplan = (int)e->qidx.size();
  std::vector<int> col_of(m.nq, -1);
  for (int c = 0; c < nplan; c++) col_of[e->qidx[c]] = c;

  ip[H_NPLAN] = nplan;
  ip[H_NSAVE] = nsave;
  ip[H_NSLOTS] = e->nslots;
  ip[H_NBODYOPS] = (int)order.size();
  ip[H_OFF_WCULL] = 0;
  ip[H_NWORLD] = nworld;
  ip[H_NWPAD] = nwpad;
  dp = wcull_tab;
  for (int w = 0; w < nworld; w++) info_at.push_back({(size_t)wc_at(w, WC_INFO), winfo[w]});
  ip[H_OFF_WNARROW] = (int)dp.size();
  dp.insert(dp.end(), wnarrow_tab.begin(), wnarrow_tab.end());

  // column permutation: ascending qpos address (the order np.linalg.norm sums the full vector)
  ip[H_OFF_PERM] = (int)ip.size();
  {
    std::vector<int> perm(nplan);
    for (int c = 0; c < nplan; c++) perm[c] = c;
    std::sort(perm.begin(), perm.end(), [&](int a, int b) { return e->qidx[a] < e->qidx[b]; });
    for (int c : perm) ip.push_back(c);
  }

  ip[H_OFF_BODYOPS] = (int)ip.size();
  int gk = 0;  // index into mgeoms
  for (size_t k = 0; k < order.size(); k++) {
    const int b = order[k], p = m.body_parentid[b];
    int parent_src;
    if (e->body_static[p]) parent_src = PARENT_STATIC;
    else if (k > 0 && order[k - 1] == p) parent_src = PARENT_CUR;
    else parent_src = save_slot[p] + 1;
    const size_t base = ip.size();
    ip.resize(base + B_SIZE);
    ip[base + B_PARENT] = parent_src;
    ip[base + B_DOFF] = (int)dp.size();
    ip[base + B_BODYID] = b;
    ip[base + B_NJNT] = m.body_jntnum[b];
    ip[base + B_SAVE] = save_slot[b];
    for (int k3 = 0; k3 < 3; k3++) dp.push_back(m.body_pos[3 * b + k3]);
    for (int k4 = 0; k4 < 4; k4++) dp.push_back(m.body_quat[4 * b + k4]);
    if (parent_src == PARENT_STATIC) {
      for (int k3 = 0; k3 < 3; k3++) dp.push_back(e->st_xpos[3 * p + k3]);
      for (int k4 = 0; k4 < 4; k4++) dp.push_back(e->st_xquat[4 * p + k4]);
      for (int k9 = 0; k9 < 9; k9++) dp.push_back(st_xmat[9 * p + k9]);
    }
    for (int j = 0; j < m.body_jntnum[b]; j++) {
      const int jid = m.body_jntadr[b] + j;
      const int qadr = m.jnt_qposadr[jid];
      const double *jp = &m.jnt_pos[3 * jid];
      ip.push_back(m.jnt_type[jid]);
      ip.push_back(col_of[qadr]);
      ip.push_back((jp[0] != 0 || jp[1] != 0 || jp[2] != 0) ? JF_POS_NONZERO : 0);
      ip.push_back((int)dp.size());
      for (int k3 = 0; k3 < 3; k3++) dp.push_back(m.jnt_axis[3 * jid + k3]);
      for (int k3 = 0; k3 < 3; k3++) dp.push_back(jp[k3]);
      dp.push_back(m.qpos0[qadr]);
      dp.push_back(e->qbase[qadr]);
    }
    int ngeom_here = 0;
    for (; gk < nm && m.geom_bodyid[mgeoms[gk]] == b; gk++, ngeom_here++) {
      const int g = mgeoms[gk];
      const double *gp = &m.geom_pos[3 * g], *gq = &m.geom_quat[4 * g];
      int flags = 0;
      if (gp[0] == 0 && gp[1] == 0 && gp[2] == 0) flags |= GF_SAMEPOS;
      if (gq[0] == 1 && gq[1] == 0 && gq[2] == 0 && gq[3] == 0) flags |= GF_SAMEROT;
      unsigned long long wmask = 0, pmask = 0;
      for (int sgeom : world_partners[gk])
        (m.geom_type[sgeom] == GT_PLANE ? pmask : wmask) |= 1ull << world_row[sgeom];
      unsigned smask = 0;
      for (int k2 : stored_partners[gk]) smask |= 1u << (slot_of[k2] & 63);
      ip.push_back(m.geom_type[g]);
      ip.push_back(flags);
      ip.push_back((int)dp.size());
      ip.push_back(slot_of[gk]);
      ip.push_back(g);
      ip.push_back((int)smask);
      ip.push_back((int)(uint32_t)(wmask & 0xffffffffull));
      ip.push_back((int)(uint32_t)(wmask >> 32));
      ip.push_back((int)(uint32_t)(pmask & 0xffffffffull));
      ip.push_back((int)(uint32_t)(pmask >> 32));
      const size_t swords_at = ip.size();
      ip.insert(ip.end(), MAX_SLOTS, 0);
      for (int k3 = 0; k3 < 3; k3++) dp.push_back(gp[k3]);
      for (int k4 = 0; k4 < 4; k4++) dp.push_back(gq[k4]);
      for (int k3 = 0; k3 < 3; k3++) dp.push_back(m.geom_size[3 * g + k3]);
      dp.push_back((double)g);  // GD_GEOMID
      dp.push_back(0.0);

      // cull bound and margin of the pair (g, h) in mj_collision's (g1 < g2) order
      auto pair_bound = [&](int h, double *bound, double *margin) {
        const int g1 = std::min(g, h), g2 = std::max(g, h);
        *margin = std::fmax(m.geom_margin[g1], m.geom_margin[g2]);
        const double r1 = m.geom_rbound[g1], r2 = m.geom_rbound[g2];
        *bound = std::numeric_limits<double>::infinity();
        if (r1 > 0 && r2 > 0) {
          const double bsum = r1 + r2 + *margin;
          *bound = bsum * bsum;
        } else if (m.geom_type[h] == GT_PLANE && m.geom_rbound[g] > 0) {
          *bound = *margin + m.geom_rbound[g];
        }
      };
      const double inf = std::numeric_limits<double>::infinity();
      {
        // rows that are no partner of this geom: -inf, so the queued culls need no enable mask
        std::vector<double> wb(nwpad, -inf), wm(nwpad, 0.0);
        for (int sgeom : world_partners[gk]) {
          pair_bound(sgeom, &wb[world_row[sgeom]], &wm[world_row[sgeom]]);
          (m.geom_type[sgeom] == GT_PLANE ? plane_bound_at : sq_bound_at).push_back(dp.size() + world_row[sgeom]);
        }
        dp.insert(dp.end(), wb.begin(), wb.end());
        dp.insert(dp.end(), wm.begin(), wm.end());
      }
      {
        std::vector<double> sb(MAX_SLOTS, inf), sm(MAX_SLOTS, 0.0), ss(3 * MAX_SLOTS, 0.0), sg(MAX_SLOTS, -1.0);
        for (int k2 : stored_partners[gk]) {
          const int h = mgeoms[k2];
          const int s1 = slot_of[k2] & 63, s2 = (slot_of[k2] >> 6) & 63;
          const int g1 = std::min(g, h), g2 = std::max(g, h);
          const int first = (m.geom_type[g1] > m.geom_type[g2]) ? g2 : g1;
          ip[swords_at + s1] = s2 | (m.geom_type[h] << 12) | (first == h ? P_FIRST : 0);
          pair_bound(h, &sb[s1], &sm[s1]);
          sq_bound_at.push_back(dp.size() + s1);
          for (int k3 = 0; k3 < 3; k3++) ss[3 * s1 + k3] = m.geom_size[3 * h + k3];
          sg[s1] = (double)h;
        }
        dp.insert(dp.end(), sb.begin(), sb.end());
        dp.insert(dp.end(), sm.begin(), sm.end());
        dp.insert(dp.end(), ss.begin(), ss.end());
        dp.insert(dp.end(), sg.begin(), sg.end());  // GS_GEOMID
      }
    }
    ip[base + B_NGEOM] = ngeom_here;
  }

  // the kernels prefetch one entry past the one they test: keep that read inside the tables
  ip.insert(ip.end(), 32, 0);
  dp.insert(dp.end(), 24, 0.0);

  // ---- binary32 error bound of the filter.  eps = 2^-24.  For every
  // moving body b, by induction along the chain (every operation of run_config_queued counted with
  // its worst-case rounding; fused multiply-adds only lower these):
  //   rot(b) <= rot(parent) + (20 + 42 * hinges(b)) eps          orientation error, radians
  //   pos(b) <= posA(b) + posB(b) * C                            position error, metres, where C
  //             bounds every moving coordinate magnitude (enforced per lane: FC_MAXCOORD), and
  //   posA(b) = posA(parent) + L_b (rot(parent) + 8 eps) + 3 eps L_b + sum_hinges 2 |jnt_pos| (rot(b) + 8 eps)
  //   posB(b) = posB(parent) + sqrt(3) eps (1 + slides(b) + 2 offcentre_hinges(b))
  // A geom adds |lpos| (rot + 8 eps) + extent (rot + 24 eps) + 2 eps |size| and sqrt(3) eps C; a static
  // geom is off by the rounding of its constants; evaluating a narrowphase formula on binary32
  // poses adds 16 eps (pair scale) + 4 eps C.  Signed distances are 1-Lipschitz in every point of
  // either geom, so |distance32 - distance64| <= E = A + B C over all enabled pairs.
  {
    const double eps = std::ldexp(1.0, -24);
    const double r3 = std::sqrt(3.0);
    auto norm3 = [](const double *v) { return std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); };
    std::vector<double> rot(nb, 0.0), posA(nb, 0.0), posB(nb, 0.0);
    for (int b = 0; b < nb; b++)
      if (e->body_static[b]) { rot[b] = 2 * eps; posA[b] = r3 * eps * norm3(&e->st_xpos[3 * b]); }
    for (int b : order) {
      const int p = m.body_parentid[b];
      int hinges = 0, slides = 0, off = 0;
      double jp = 0;
      for (int j = 0; j < m.body_jntnum[b]; j++) {
        const int jid = m.body_jntadr[b] + j;
        if (m.jnt_type[jid] == JT_HINGE) {
          hinges++;
          const double l = norm3(&m.jnt_pos[3 * jid]);
          if (l > 0) { off++; jp += l; }
        } else {
          slides++;
        }
      }
      const double L = norm3(&m.body_pos[3 * b]);
      rot[b] = rot[p] + (20.0 + 42.0 * hinges) * eps;
      posA[b] = posA[p] + L * (rot[p] + 8 * eps) + 3 * eps * L + 2 * jp * (rot[b] + 8 * eps);
      posB[b] = posB[p] + r3 * eps * (1 + slides + 2 * off);
    }
    auto extent = [&](int g) {  // farthest point of the geom from its frame origin along rotating directions
      const double *sz = &m.geom_size[3 * g];
      if (m.geom_type[g] == GT_CAPSULE) return sz[1];
      if (m.geom_type[g] == GT_BOX) return norm3(sz);
      return 0.0;
    };
    std::vector<double> gA(ng, 0.0), gB(ng, 0.0);
    e->npoisoned = 0;
    std::vector<char> poisoned(ng, 0);
    const double tolh_req = 0.5 * e->filter_tol_req;
    for (int g = 0; g < ng; g++) {
      const int b = m.geom_bodyid[g];
      const double lp = norm3(&m.geom_pos[3 * g]), sz = norm3(&m.geom_size[3 * g]);
      if (e->geom_static[g]) {
        gA[g] = r3 * eps * (norm3(&e->st_gxpos[3 * g]) + extent(g)) + 2 * eps * sz;
        // a static geom whose own constants do not fit binary32 within an eighth of the band: its
        // narrowphase rows are NaN in the float tables, so every pair that passes its (widened)
        // cull comes out undecided and is settled by the float64 pair kernel
        if (gA[g] > 0.25 * tolh_req) { poisoned[g] = 1; e->npoisoned++; }
      } else {
        gA[g] = posA[b] + lp * (rot[b] + 8 * eps) + extent(g) * (rot[b] + 24 * eps) + 2 * eps * sz;
        gB[g] = posB[b] + r3 * eps;
      }
    }
    double A = 0, B = 0;
    for (int k = 0; k < nm; k++) {
      const int g = mgeoms[k];
      auto pair = [&](int h) {
        if (poisoned[h]) return;
        const double scale = m.geom_rbound[g] + m.geom_rbound[h] + std::fmax(m.geom_margin[g], m.geom_margin[h]) +
                             (m.geom_type[h] == GT_PLANE ? norm3(&e->st_gxpos[3 * h]) : 0.0);
        A = std::fmax(A, gA[g] + gA[h] + 16 * eps * scale);
        B = std::fmax(B, gB[g] + gB[h] + 4 * eps);
      };
      for (int sgeom : world_partners[k]) pair(sgeom);
      for (int k2 : stored_partners[k]) pair(mgeoms[k2]);
    }
    e->ferr_a = A;
    e->ferr_b = B;
    // half the band is the error budget: E(C) = A + B C <= tol / 2.  A default tolerance grows with
    // the model's floor; one the caller asked for is kept, and if the floor does not fit under it
    // the filter steps aside for this model (exact path only).
    double tol = e->filter_tol_req;
    e->filter_usable = true;
    if (A > 0.4 * tol) {
      if (e->filter_tol_user) e->filter_usable = false;
      else tol = A / 0.4;
      if (!(tol < 1e-2)) e->filter_usable = false;  // a band of centimetres decides nothing useful
    }
    e->filter_tol = (float)tol;
    // (a candidate record of the filter's queues carries its geom's table offset in 16 bits -- the other half of the word
    //  is the candidate's certificate margin, mjpl_device.h: queue_drain --: a table of 64 K entries or more, far beyond
    //  any model the slot file holds, takes the exact path)
    if (dp.size() + 64 >= 65536) e->filter_usable = false;
    double maxc = (B > 0) ? (0.5 * tol - A) / B : 1e6;
    maxc = std::fmin(std::fmax(maxc, 0.0), 1e6);
    e->fmax_coord = e->filter_usable ? maxc : 0.0;
    ip[H_OFF_FCONST] = (int)dp.size();
    double fc[FC_SIZE] = {0};
    fc[FC_MAXCOORD] = e->fmax_coord;
    fc[FC_MAXANGLE] = kFilterMaxAngle;
    dp.insert(dp.end(), fc, fc + FC_SIZE);
    // NaN rows: written into the float image below
    for (int g = 0; g < ng; g++)
      if (poisoned[g]) poison_rows.push_back(world_row[g]);
  }

  // ---- the filter's float32 image: same offsets; cull bounds widened by the tolerance so that
  // a pair culled in float32 is certainly culled (or contact-free) in float64
  std::vector<float> &fp = e->fp;
  fp.resize(dp.size());
  for (size_t k = 0; k < dp.size(); k++) fp[k] = (float)dp[k];
  // (a poisoned static geom's own rounding may exceed the band: its bounds are widened by that, too)
  const double tol = e->filter_tol;
  double poison_slack = 0;
  for (int w : poison_rows) {
    const double *wt = &dp[(size_t)ip[H_OFF_WCULL]];
    const double rc[3] = {wt[wc_at(w, 0)], wt[wc_at(w, 1)], wt[wc_at(w, 2)]};
    const double *rn = &dp[(size_t)ip[H_OFF_WNARROW] + (size_t)w * WN_LEN];
    const double mag = std::fabs(rc[0]) + std::fabs(rc[1]) + std::fabs(rc[2]) + std::fabs(rn[WN_SIZE]) +
                       std::fabs(rn[WN_SIZE + 1]) + std::fabs(rn[WN_SIZE + 2]);
    poison_slack = std::fmax(poison_slack, 4 * std::ldexp(1.0, -24) * mag);
  }
  for (size_t k : sq_bound_at)
    if (std::isfinite(dp[k])) {
      const double r = std::sqrt(dp[k]) + tol + poison_slack;
      fp[k] = (float)(r * r * (1.0 + 1e-6));
    }
  for (size_t k : plane_bound_at)
    if (std::isfinite(dp[k])) fp[k] = (float)(dp[k] + tol + poison_slack + 1e-6 * std::fabs(dp[k]));
  for (auto &kv : info_at) memcpy(&fp[kv.first], &kv.second, sizeof(float));
  // the whole narrowphase row (axes and sizes; the position belongs to the cull table): every
  // routine then computes NaN and classifies the pair as undecided
  for (int w : poison_rows)
    for (int k = 0; k < WN_LEN; k++) fp[(size_t)ip[H_OFF_WNARROW] + (size_t)w * WN_LEN + k] = std::numeric_limits<float>::quiet_NaN();

  // ---- identity of the compiled program: what a per-model specialised library is keyed by
  {
    uint64_t h = 0xcbf29ce484222325ull;
    auto mix = [&](const void *ptr, size_t n) {
      const unsigned char *b = (const unsigned char *)ptr;
      for (size_t k = 0; k < n; k++) { h ^= b[k]; h *= 0x100000001b3ull; }
    };
    mix(ip.data(), ip.size() * sizeof(int));
    mix(fp.data(), fp.size() * sizeof(float));
    // the float64 table as well: the generated exact pair re-check (ExactSpec::fk_pair) carries ITS values
    // as literals, and two programs may share a binary32 image while their float64 constants differ
    mix(dp.data(), dp.size() * sizeof(double));
    const int shape[4] = {e->maxs, e->wbox ? 1 : 0, e->mbox ? 1 : 0, MJPL_SPEC_ABI};
    mix(shape, sizeof(shape));
    const unsigned long long stamp = MJPL_SRC_STAMP;
    mix(&stamp, sizeof(stamp));
    e->program_hash = h;
  }
  // ---- identity of the ROBOT alone, and the cull table a scene-generic library reads.
  // Everything the generated code of such a library carries as literals goes into robot_hash: the moving
  // bodies with their constants, joints (planning column or constant), geoms, register slots and self
  // pairs with their bounds, tolerance, kernel shape.  Nothing of the static geoms: those reach the code
  // through `scene` -- a header, then for every moving geom kSceneRows cull rows [a0 a1 a2 thr] followed by
  // kSceneRows descriptor words: a plane partner (rows 0, 1) passes its cull when a . c <= thr (a =
  // normal), any other (rows 2 ..) when |c|^2 + a . c <= thr (a = -2 X: the expanded form, threshold
  // raised by the form's rounding bound for a centre within the geom's reach); the descriptor says what a
  // candidate of that pair is queued as.  Rows that are no pair of the geom never pass (thr = -inf).
  const int nstage = nm;
  e->moving_base = nm > 0 ? mgeoms[0] : 0;
  int nplanes = 0;
  for (int w = 0; w < nworld; w++) nplanes += (winfo[w] & 255) == GT_PLANE ? 1 : 0;
  // (a robot with moving boxes, or 17 .. 24 stored geoms: the 24-slot queued build has generated code, too)
  bool generic_ok = !e->immediate() && e->filter_usable && nplanes <= kScenePlaneRows && nworld - nplanes <= kSceneRows - kScenePlaneRows &&
                    nstage <= kSceneMaxStages && nstage > 0;
  for (int k = 1; k < nm && generic_ok; k++) generic_ok = mgeoms[k] == mgeoms[0] + k;  // (the pair re-check counts geoms from the first moving one)
  {
    uint64_t h = 0xcbf29ce484222325ull;
    auto mix = [&](const void *ptr, size_t n) {
      const unsigned char *b = (const unsigned char *)ptr;
      for (size_t k = 0; k < n; k++) { h ^= b[k]; h *= 0x100000001b3ull; }
    };
    auto mixi = [&](int v) { mix(&v, sizeof(v)); };
    auto mixd = [&](const double *v, int n) { mix(v, sizeof(double) * (size_t)n); };
    mixi(nplan); mixi(e->maxs); mixi(e->nslots); mixi(nsave); mixi(MJPL_SPEC_ABI); mixi(kSceneRows);
    const unsigned long long stamp = MJPL_SRC_STAMP;
    mix(&stamp, sizeof(stamp));
    const float tolf = e->filter_tol;
    mix(&tolf, sizeof(tolf));
    int gk2 = 0;
    for (size_t k = 0; k < order.size(); k++) {
      const int b = order[k], p = m.body_parentid[b];
      const int psrc = e->body_static[p] ? (int)PARENT_STATIC : ((k > 0 && order[k - 1] == p) ? (int)PARENT_CUR : save_slot[p] + 1);
      mixi(psrc);
      mixd(&m.body_pos[3 * b], 3); mixd(&m.body_quat[4 * b], 4);
      if (psrc == PARENT_STATIC) { mixd(&e->st_xpos[3 * p], 3); mixd(&e->st_xquat[4 * p], 4); mixd(&st_xmat[9 * p], 9); }
      mixi(m.body_jntnum[b]); mixi(save_slot[b]);
      for (int j = 0; j < m.body_jntnum[b]; j++) {
        const int jid = m.body_jntadr[b] + j, qadr = m.jnt_qposadr[jid];
        mixi(m.jnt_type[jid]); mixi(col_of[qadr]);
        mixd(&m.jnt_axis[3 * jid], 3); mixd(&m.jnt_pos[3 * jid], 3); mixd(&m.qpos0[qadr], 1);
        const double qc = col_of[qadr] < 0 ? e->qbase[qadr] : 0.0;
        mixd(&qc, 1);
      }
      for (; gk2 < nm && m.geom_bodyid[mgeoms[gk2]] == b; gk2++) {
        const int g = mgeoms[gk2];
        mixi(m.geom_type[g]); mixi(slot_of[gk2]);
        mixd(&m.geom_pos[3 * g], 3); mixd(&m.geom_quat[4 * g], 4); mixd(&m.geom_size[3 * g], 3);
        mixd(&m.geom_rbound[g], 1); mixd(&m.geom_margin[g], 1);
        for (int k2 : stored_partners[gk2]) { mixi(k2); mixi(slot_of[k2]); }
        mixi(-1);
      }
      mixi(-2);
    }
    e->robot_hash = h;
  }
  e->scene.clear();
  if (generic_ok) {
    const size_t S = scene_floats(nstage);
    e->scene.assign(S, 0.0f);
    auto seti = [&](size_t at, int v) { memcpy(&e->scene[at], &v, sizeof(float)); };
    const double u24 = std::ldexp(1.0, -24);
    auto round_up = [](double v) {
      float f = (float)v;
      if ((double)f < v) f = std::nextafterf(f, std::numeric_limits<float>::infinity());
      return f;
    };
    // how far from the origin a moving geom's centre can be while its lane is alive (specialise.py: the same)
    const double box_reach = std::sqrt(3.0) * (double)(float)e->fmax_coord;
    std::vector<double> breach(nb, 0.0);
    auto n3 = [](const double *v) { return std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); };
    for (int b : order) {
      const int p = m.body_parentid[b];
      double r = (e->body_static[p] ? n3(&e->st_xpos[3 * p]) : breach[p]) + n3(&m.body_pos[3 * b]);
      for (int j = 0; j < m.body_jntnum[b]; j++) {
        const int jid = m.body_jntadr[b] + j;
        if (m.jnt_type[jid] == JT_SLIDE) r = std::numeric_limits<double>::infinity();
        else r += 2.0 * n3(&m.jnt_pos[3 * jid]);
      }
      breach[b] = std::fmin(r, box_reach);
    }
    // [0] planes (0 .. 2: the last rows), [1] first pair of rows in use, [2] nwpad, [3] offset of the narrowphase table
    // The rows of a geom are filled from the END: the planes last, the bounded geoms below them; the code is one
    // straight line over all kSceneRows rows, entered at the first pair of rows that holds anything ([1]).
    const int nbounded = nworld - nplanes;
    const int first_row = std::min(kSceneRows - 2, (kSceneRows - nplanes - nbounded) & ~1);
    seti(0, nplanes); seti(1, first_row);
    seti(2, nwpad); seti(3, ip[H_OFF_WNARROW]);
    e->scene[4] = (float)e->fmax_coord;
    e->scene[5] = kFilterMaxAngle;
    int gk3 = 0, pc = ip[H_OFF_BODYOPS];
    for (size_t k = 0; k < order.size(); k++) {
      const int b = order[k];
      const int njnt = ip[pc + B_NJNT], ngeom_here = ip[pc + B_NGEOM];
      pc += B_SIZE + njnt * J_SIZE;
      for (int gi = 0; gi < ngeom_here; gi++, gk3++) {
        const int g = mgeoms[gk3], gtype = m.geom_type[g], gdoff = ip[pc + G_DOFF];
        pc += G_SIZE + MAX_SLOTS;
        seti(8 + gk3, gdoff);
        const double reach = breach[b] + n3(&m.geom_pos[3 * g]);
        std::set<int> partners(world_partners[gk3].begin(), world_partners[gk3].end());
        float *rows = &e->scene[(size_t)kSceneHeader + (size_t)gk3 * kSceneStageFloats];
        float *descs = rows + (size_t)kSceneRows * 4;
        for (int r0 = 0; r0 < kSceneRows; r0++) rows[(size_t)r0 * 4 + 3] = -std::numeric_limits<float>::infinity();
        for (int pass = 0; pass < 2; pass++) {  // the planes in the last rows, the others below them
          int r = pass == 0 ? kSceneRows - nplanes : kSceneRows - nplanes - nbounded;
          for (int sgeom = 0; sgeom < ng; sgeom++) {
            if (!e->geom_static[sgeom] || world_row[sgeom] < 0) continue;
            const int w = world_row[sgeom], ptype = winfo[w] & 255;
            if ((ptype == GT_PLANE) != (pass == 0)) continue;
            float *row = rows + (size_t)r * 4;
            float *dword = descs + r;
            r++;
            if (!partners.count(sgeom)) continue;  // (not a pair of this geom: a row that never passes)
            const double X[3] = {(double)fp[wc_at(w, 0)], (double)fp[wc_at(w, 1)], (double)fp[wc_at(w, 2)]};
            const double bound = (double)fp[(size_t)gdoff + GD_WBOUND + w];
            int desc;
            if (ptype == GT_PLANE) {
              const float *rw = &fp[(size_t)ip[H_OFF_WNARROW] + (size_t)w * WN_LEN];
              const double n[3] = {(double)rw[WN_ZAXIS], (double)rw[WN_ZAXIS + 1], (double)rw[WN_ZAXIS + 2]};
              const double off = n[0] * X[0] + n[1] * X[1] + n[2] * X[2];
              for (int c = 0; c < 3; c++) row[c] = (float)n[c];
              // three fused multiply-adds on a centre within `reach`: each rounds by at most u (reach + |n . p0| + |bound|)
              row[3] = std::isfinite(reach) ? round_up(bound + off + 4.0 * u24 * (1.01 * reach + std::fabs(off) + std::fabs(bound)))
                                            : std::numeric_limits<float>::infinity();
              // (bit 15: the candidate goes to the box queue -- a static box, or ANY partner of a moving box, whose
              //  records carry whole frames)
              desc = EK_PLANE | (w << 2) | (GT_PLANE << 10) | (1 << 14) | ((gtype == GT_BOX && e->filter_mbox() ? 1 : 0) << 15);
            } else {
              const double nx = std::sqrt(X[0] * X[0] + X[1] * X[1] + X[2] * X[2]);
              for (int c = 0; c < 3; c++) row[c] = (float)(-2.0 * X[c]);
              // (the allowance of specialise.py: expanded_threshold)
              const double allow = 8.0 * u24 * (1.01 * reach + nx) * (1.01 * reach + nx);
              row[3] = std::isfinite(reach) ? round_up(bound - (X[0] * X[0] + X[1] * X[1] + X[2] * X[2]) + allow)
                                            : std::numeric_limits<float>::infinity();
              if (!std::isfinite(bound)) row[3] = (float)bound;
              const int pgid = winfo[w] >> 8;
              const int pfirst = (ptype < gtype || (ptype == gtype && pgid < g)) ? 1 : 0;
              desc = EK_STATIC | (w << 2) | (ptype << 10) | (pfirst << 14) |
                     (((ptype == GT_BOX || (gtype == GT_BOX && e->filter_mbox())) ? 1 : 0) << 15);
            }
            memcpy(dword, &desc, sizeof(float));
          }
        }
      }
    }
  }
  if (e->device < 0) return MJPL_OK;  // mjpl_program_dump: host tables only
  load_spec(e, generic_ok, nstage);
  if (!e->spec_generic) e->scene.clear();

  // ---- upload
  if (e->d_ip) (void)hipFree(e->d_ip);
  if (e->d_dp) (void)hipFree(e->d_dp);
  if (e->d_fp_base) (void)hipFree(e->d_fp_base);
  e->d_ip = nullptr;
  e->d_dp = nullptr;
  e->d_fp = e->d_fp_base = nullptr;
  HIP_TRY(hipMalloc(&e->d_ip, ip.size() * sizeof(int)));
  HIP_TRY(hipMalloc(&e->d_dp, dp.size() * sizeof(double)));
  HIP_TRY(hipMalloc(&e->d_fp_base, (e->scene.size() + fp.size()) * sizeof(float)));
  e->d_fp = e->d_fp_base + e->scene.size();  // (the kernels' table pointer: a scene-generic library reads backwards from it)
  HIP_TRY(hipMemcpy(e->d_ip, ip.data(), ip.size() * sizeof(int), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(e->d_dp, dp.data(), dp.size() * sizeof(double), hipMemcpyHostToDevice));
  if (!e->scene.empty()) HIP_TRY(hipMemcpy(e->d_fp_base, e->scene.data(), e->scene.size() * sizeof(float), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(e->d_fp, fp.data(), fp.size() * sizeof(float), hipMemcpyHostToDevice));
  return MJPL_OK;
}

size_t lds_bytes(const mjpl_engine *e, int ncolsets, size_t scalar = sizeof(double), int block = kBlock,
                 bool queued = false, size_t colscalar = sizeof(double)) {
  const size_t nplan = e->qidx.size();
  size_t bytes = (((size_t)ncolsets * nplan * block * colscalar + 7) & ~(size_t)7) + (size_t)e->nsave * 7 * block * scalar;
  bytes = (bytes + 7) & ~(size_t)7;
  if (queued)
    bytes += (size_t)(block / 64) * (e->filter_mbox() ? WaveQueue<float, true>::bytes() : WaveQueue<float, false>::bytes()) +
             ((e->fp.size() * sizeof(float) + 7) & ~(size_t)7);
#if MJPL_TABLES_LDS
  bytes += ((e->dp.size() * scalar + 7) / 8) * 8 + ((e->ip.size() * sizeof(int) + 7) / 8) * 8;
#endif
  return bytes ? bytes : 8;
}

int uc_reserve(mjpl_engine *e, int64_t n) {
  // MJPL_UC_CAP (create-time, tests): a small hand-over buffer, to exercise its overflow paths
  const size_t want = e->uc_cap_limit ? e->uc_cap_limit : (size_t)std::max<int64_t>(4096, n);
  if (want <= e->uc_cap) return MJPL_OK;
  if (e->d_ucq) HIP_TRY(hipFree(e->d_ucq));
  if (e->d_ucedge) HIP_TRY(hipFree(e->d_ucedge));
  if (e->d_ucidx) HIP_TRY(hipFree(e->d_ucidx));
  if (e->d_ucga) HIP_TRY(hipFree(e->d_ucga));
  if (e->d_ucgb) HIP_TRY(hipFree(e->d_ucgb));
  e->d_ucq = nullptr; e->d_ucedge = e->d_ucidx = e->d_ucga = e->d_ucgb = nullptr; e->uc_cap = 0;
  HIP_TRY(hipMalloc(&e->d_ucq, want * std::max<size_t>(1, e->m.nq) * sizeof(double)));
  HIP_TRY(hipMalloc(&e->d_ucedge, want * sizeof(int)));
  HIP_TRY(hipMalloc(&e->d_ucidx, want * sizeof(int)));
  HIP_TRY(hipMalloc(&e->d_ucga, want * sizeof(int)));
  HIP_TRY(hipMalloc(&e->d_ucgb, want * sizeof(int)));
  if (!e->d_geomtab) {
    const HostModel &m = e->m;
    std::vector<double> t((size_t)m.ngeom * GTB_LEN, 0.0);
    for (int g = 0; g < m.ngeom; g++) {
      double *r = &t[(size_t)g * GTB_LEN];
      r[GTB_TYPE] = m.geom_type[g];
      for (int k = 0; k < 3; k++) r[GTB_SIZE + k] = m.geom_size[3 * g + k];
      r[GTB_RBOUND] = m.geom_rbound[g];
      r[GTB_MARGIN] = m.geom_margin[g];
      r[GTB_STATIC] = e->geom_static[g] ? 1.0 : 0.0;
      for (int k = 0; k < 3; k++) r[GTB_XPOS + k] = e->st_gxpos[3 * g + k];
      for (int k = 0; k < 9; k++) r[GTB_XMAT + k] = e->st_gxmat[9 * g + k];
    }
    HIP_TRY(hipMalloc(&e->d_geomtab, t.size() * sizeof(double)));
    HIP_TRY(hipMemcpy(e->d_geomtab, t.data(), t.size() * sizeof(double), hipMemcpyHostToDevice));
  }
  e->uc_cap = want;
  return MJPL_OK;
}

int ulist_reserve(mjpl_engine *e, int64_t n) {
  if (!e->d_ucount) {
    // five device counters, each on its own 128-byte line (they are hammered by wave-level atomics
    // of the same kernel: sharing a line costs ~7 % of a step), cleared by one memset per launch:
    //   [0] edge-level undecided list, [kCtr] undecided pairs, [2 kCtr] edges whose endpoint passed,
    //   [3 kCtr] (unused), [4 kCtr] edges left to the walking kernel,
    //   [5 kCtr ..] waypoint items per region, then edges whose endpoint passed per region
    // two sets, used alternately (next_counters): the first kernel of a launch clears the other one
    HIP_TRY(hipMalloc(&e->d_ucount_base, 2 * kNumCtr * kCtr * sizeof(int)));
    HIP_TRY(hipMemset(e->d_ucount_base, 0, 2 * kNumCtr * kCtr * sizeof(int)));
    e->d_ucount = e->d_ucount_base;
    e->d_icount = e->d_ucount + 3 * kCtr;
  }
  if ((size_t)n > e->ulist_cap) {
    if (e->d_ulist) HIP_TRY(hipFree(e->d_ulist));
    e->d_ulist = nullptr;
    e->ulist_cap = 0;
    HIP_TRY(hipMalloc(&e->d_ulist, (size_t)n * sizeof(int)));
    e->ulist_cap = (size_t)n;
  }
  return MJPL_OK;
}

// Raise a kernel's dynamic-LDS limit when a launch needs more than was granted before (per
// device: the attribute belongs to the loaded code object).  Cached: the driver call costs a
// few microseconds, and a step is several launches.
template <class K>
int allow_lds(K kernel, size_t bytes) {
  if (bytes > 160 * 1024) return fail(MJPL_E_CAPACITY, "model needs %zu B of LDS per workgroup (> 160 KiB)", bytes);
  static thread_local std::map<std::pair<const void *, int>, size_t> granted;
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  size_t &have = granted[{reinterpret_cast<const void *>(kernel), dev}];
  if (bytes <= have) return MJPL_OK;
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  have = bytes;
  return MJPL_OK;
}

// Can this engine's edge launches run the fused filter kernel (mjpl_fused.h), and how: waves per workgroup
// (twelve = one workgroup per CU at three waves per SIMD; four for the one-wave-per-SIMD build of models with
// moving boxes), entries of a workgroup's pool (what the CU's LDS leaves, at least 64 per wave: one endpoint tile
// each) and the dynamic LDS of a workgroup.
bool fused_plan(const mjpl_engine *e, int *nwaves, size_t *lds, int *ring = nullptr, const SpecLib *lib = nullptr) {
  const SpecLib *spec = lib ? lib : e->spec;  // (lib: the certificate build instead of the engine's default library)
  if (!e->fused || !e->filter || !e->filter_usable || !e->two_pass || !e->expand || e->immediate()) return false;
  const int nplan = (int)e->qidx.size();
  const bool mbox = e->filter_mbox();
  // (models with moving boxes run one wave per SIMD: four waves per workgroup, and measured 5 % SLOWER fused than as the
  // two kernels -- 1.63 vs 1.55 ms on Franka-P with the ten pad boxes; MJPL_FUSED_MBOX=1 runs them fused all the same)
  if (mbox && !e->fused_mbox && !spec) return false;
  if (!fused_fits(nplan, e->nsave, mbox)) return false;
  const int nw = spec ? spec->fused_waves : (mbox ? 4 : kFusedWaves);
  const size_t budget = (size_t)160 * 1024;
  const bool cert = spec && spec->fused_cert;
  const size_t base = fused_lds_bytes(nw, nplan, e->nsave, e->fp.size(), mbox, 0, cert);
  if (base + (size_t)(64 * nw + 64) * kFusedEntryBytes > budget) return false;
  int r = (int)std::min<size_t>(kFusedMaxPool, (budget - base) / kFusedEntryBytes / 64 * 64);
  if (e->fused_pool_cap > 0) r = std::max(64 * nw + 64, std::min(r, e->fused_pool_cap / 64 * 64));  // (tests: a ring that wraps)
  if (nwaves) *nwaves = nw;
  if (lds) *lds = fused_lds_bytes(nw, nplan, e->nsave, e->fp.size(), mbox, r, cert);
  if (ring) *ring = r;
  return true;
}

// ... and with the filter off: the float64 checks through the same pool (k_edges_fused_f64)
bool fused_f64_ok(const mjpl_engine *e) {
  return e->fused && e->two_pass && e->expand &&
         fused_f64_lds_bytes(kFusedF64Waves, (int)e->qidx.size(), e->nsave, 64 * kFusedF64Waves + 64) <= (size_t)160 * 1024;
}

// A filter launch takes the cleared counter set and returns the other one, which its first kernel is
// to clear for the launch after it.
int *next_counters(mjpl_engine *e) {
  if (e->counters_stale) {  // a launch failed between taking its set and its first kernel
    (void)hipMemsetAsync(e->d_ucount_base, 0, 2 * (size_t)kNumCtr * kCtr * sizeof(int), e->stream);
    e->counters_stale = false;
  }
  int *other = e->d_ucount;
  e->d_ucount = (e->d_ucount == e->d_ucount_base) ? e->d_ucount_base + (size_t)kNumCtr * kCtr : e->d_ucount_base;
  e->d_icount = e->d_ucount + 3 * kCtr;
  return other;
}

// pick the <MAXS, WBOX, MBOX> instantiation of the EXACT kernels for this model
template <class F>
int dispatch_variant(const mjpl_engine *e, F &&f) {
  if (e->exact_general()) return f(std::integral_constant<int, 32>{}, std::true_type{}, std::true_type{});
  auto with_box = [&](auto S) -> int {
    if (e->wbox) return f(S, std::true_type{}, std::false_type{});
    return f(S, std::false_type{}, std::false_type{});
  };
  switch (e->maxs) {
    case 4: return with_box(std::integral_constant<int, 4>{});
    case 8: return with_box(std::integral_constant<int, 8>{});
    case 16: return with_box(std::integral_constant<int, 16>{});
    default: return f(std::integral_constant<int, 32>{}, std::true_type{}, std::true_type{});
  }
}

// ... and of the FILTER kernels
template <class F>
int dispatch_filter(const mjpl_engine *e, F &&f) {
  if (e->immediate()) return f(std::integral_constant<int, 32>{}, std::true_type{}, std::true_type{});
  if (e->filter_mbox()) return f(std::integral_constant<int, kQueuedMaxSlots>{}, std::true_type{}, std::true_type{});
  return dispatch_variant(e, f);
}

// option "kernel_timer": an event on either side of the launches of class `mode`
void kt_mark(mjpl_engine *e, int mode, hipStream_t st) {
  if (!(e->kt_mode & mode)) return;
  const int c = mode == 2 ? 1 : 0;
  if (e->kt_used[c] + 1 > (int)e->kt_ev[c].size()) {
    if (e->kt_ev[c].size() >= 4096) return;  // (a run of more launches than that: the later ones go untimed)
    hipEvent_t ev = nullptr;
    if (hipEventCreate(&ev) != hipSuccess) return;
    e->kt_ev[c].push_back(ev);
  }
  (void)hipEventRecord(e->kt_ev[c][(size_t)e->kt_used[c]++], st);
}

struct KtScope {  // ... around everything a scope enqueues on the engine's stream
  mjpl_engine *e; int mode;
  KtScope(mjpl_engine *e_, int mode_) : e(e_), mode(mode_) { kt_mark(e, mode, e->stream); }
  ~KtScope() { kt_mark(e, mode, e->stream); }
};

// exact re-check of the undecided pairs: the model's own straight-line float64 FK if it has a library
int launch_patch(mjpl_engine *e, unsigned pgrid, size_t ldsc, const UndecidedConfigs &uc, uint8_t *dvalid, int32_t *dfb) {
  GeomTable gt = {e->d_geomtab, e->moving_base};
  if (e->spec)
    return e->spec->patch(e->stream, pgrid, (unsigned)kBlock, ldsc, e->d_ip, (int)e->ip.size(), e->d_dp, (int)e->dp.size(), gt,
                          uc, dvalid, dfb) == 0 ? MJPL_OK : fail(MJPL_E_HIP, "specialised pair kernel failed to launch");
  int rc = allow_lds(k_patch_pairs<void>, ldsc);
  if (rc != MJPL_OK) return rc;
  hipLaunchKernelGGL(k_patch_pairs<void>, dim3(pgrid), dim3(kBlock), ldsc, e->stream, e->d_ip, (int)e->ip.size(), e->d_dp,
                     (int)e->dp.size(), gt, uc, dvalid, dfb);
  return MJPL_OK;
}

struct CounterGuard {  // marks the counter sets for a full clear unless the launch got through
  mjpl_engine *e;
  bool armed;
  ~CounterGuard() { if (armed) e->counters_stale = true; }
};

int launch_configs(mjpl_engine *e, const double *dQ, int64_t N, int layout, uint8_t *dvalid,
                   unsigned long long *dbits) {
  if (N == 0) return MJPL_OK;
  const unsigned grid = (unsigned)((N + kBlock - 1) / kBlock);
  const bool filter = e->filter && e->filter_usable && dvalid && !dbits && N < (int64_t)1 << 29;
  if (filter) {
    int rc = ulist_reserve(e, N);
    if (rc != MJPL_OK) return rc;
    int *zero_next = next_counters(e);
    CounterGuard guard{e, true};
    UndecidedConfigs uc = {};
    if (rc == MJPL_OK) {  // (both interpreters hand single undecided pairs over)
      rc = uc_reserve(e, N);
      uc.q = e->d_ucq; uc.edge = e->d_ucedge; uc.idx = e->d_ucidx;
      uc.ga = e->d_ucga; uc.gb = e->d_ucgb;
      uc.count = e->d_ucount + kCtr;
      uc.cap = (int)std::min<size_t>(e->uc_cap, (size_t)1 << 30);
    }
    if (rc != MJPL_OK) return rc;
    const int fblock = e->immediate() ? kBlock : kFilterBlock;
    const unsigned fgrid = (unsigned)((N + fblock - 1) / fblock);
    // (queued interpreter: binary32 columns)
    const size_t ldsf = lds_bytes(e, 1, sizeof(float), fblock, !e->immediate(), e->immediate() ? sizeof(double) : sizeof(float));
    kt_mark(e, 1, e->stream);
    if (e->spec)
      rc = e->spec->configs(e->stream, fgrid, (unsigned)fblock, ldsf, e->d_ip, (int)e->ip.size(), e->d_fp, (int)e->fp.size(), dQ, N,
                            layout, e->filter_tol, dvalid, e->d_ulist, e->d_ucount, uc, zero_next) == 0 ? MJPL_OK
           : fail(MJPL_E_HIP, "specialised configuration kernel failed to launch");
    else rc = dispatch_filter(e, [&](auto S, auto W, auto M) -> int {
      auto kern = k_filter_configs<void, decltype(S)::value, decltype(W)::value, decltype(M)::value>;
      int r = allow_lds(kern, ldsf);
      if (r != MJPL_OK) return r;
      hipLaunchKernelGGL(kern, dim3(fgrid), dim3(fblock), ldsf, e->stream, e->d_ip, (int)e->ip.size(),
                         e->d_fp, (int)e->fp.size(), dQ, N, layout, e->filter_tol, dvalid, e->d_ulist,
                         e->d_ucount, uc, zero_next);
      return MJPL_OK;
    });
    kt_mark(e, 1, e->stream);
    if (rc != MJPL_OK) return rc;
    guard.armed = false;
    if (uc.count) {  // exact re-check of the undecided pairs; rows here are planning columns, AoS
      const size_t ldsc = lds_bytes(e, 1);
      const unsigned pgrid = (unsigned)std::min<size_t>((uc.cap + kBlock - 1) / kBlock, 1024);
      rc = launch_patch(e, pgrid, ldsc, uc, dvalid, nullptr);
      if (rc != MJPL_OK) return rc;
    }
  }
  const size_t lds = lds_bytes(e, 1);
  int rc = dispatch_variant(e, [&](auto S, auto W, auto M) -> int {
    auto kern = k_check_configs<decltype(S)::value, decltype(W)::value, decltype(M)::value>;
    int r = allow_lds(kern, lds);
    if (r != MJPL_OK) return r;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kBlock), lds, e->stream, e->d_ip, (int)e->ip.size(),
                       e->d_dp, (int)e->dp.size(), dQ, N, layout, dvalid, dbits,
                       filter ? e->d_ulist : nullptr, filter ? e->d_ucount : nullptr, UndecidedConfigs{},
                       (int32_t *)nullptr);
    return MJPL_OK;
  });
  if (rc != MJPL_OK) return rc;
  HIP_TRY(hipGetLastError());
  return MJPL_OK;
}

// stages of one edge launch, in stream order (include/mjpl_hip.h: MJPL_STAGE_*)
#define MJPL_MARK(k)                                                          \
  do {                                                                        \
    if (e->marks) HIP_TRY(hipEventRecord(e->marks[k], e->stream));            \
  } while (0)

int launch_edges(mjpl_engine *e, const double *dQA, const double *dQB, int64_t E, double step,
                 int layout, int flags, uint8_t *dvalid, int32_t *dfb) {
  if (E == 0) return MJPL_OK;
  const unsigned grid = (unsigned)((E + kBlock - 1) / kBlock);
  const bool filter = e->filter && e->filter_usable && E < (int64_t)1 << 29;  // item ids travel in 29 bits of the per-lane flag words
  UndecidedConfigs uc = {};
  if (filter) {
    int rc = ulist_reserve(e, E);
    if (rc == MJPL_OK) rc = uc_reserve(e, E);
    if (rc != MJPL_OK) return rc;
    int *zero_next = next_counters(e);  // (this launch's own set was cleared by the launch before it)
    CounterGuard guard{e, true};
    uc.q = e->d_ucq; uc.edge = e->d_ucedge; uc.idx = e->d_ucidx;
    uc.ga = e->d_ucga; uc.gb = e->d_ucgb;
    uc.count = e->d_ucount + kCtr;
    uc.cap = (int)std::min<size_t>(e->uc_cap, (size_t)1 << 30);
    MJPL_MARK(0);
    const int fblock = e->immediate() ? kBlock : kFilterBlock;
    const unsigned fgrid = (unsigned)((E + fblock - 1) / fblock);
    const size_t ldsf = lds_bytes(e, 1, sizeof(float), fblock, !e->immediate());  // walking kernel: float64 columns
    // endpoint and item kernels of the queued interpreter keep binary32 columns
    const size_t ldsq = lds_bytes(e, 1, sizeof(float), fblock, !e->immediate(), e->immediate() ? sizeof(double) : sizeof(float));
    // the endpoint kernel of the immediate interpreter walks the waypoint recurrence in a second
    // column set (the queued one reuses the workgroup's columns, saves and queues)
    const size_t ldse = e->immediate() ? lds_bytes(e, 2, sizeof(float), fblock, false) : ldsq;
    // two passes unless only the interior was asked for: endpoints of all edges, then the interior
    // waypoints of the edges whose endpoint passed
    const bool two_pass = e->two_pass && !(flags & MJPL_EDGE_INTERIOR_ONLY);
    // interior waypoints: one lane per waypoint for ordinary edges (the endpoint kernel emits the
    // waypoints as items), the walking kernel for long edges and for models with moving boxes
    const bool expand = two_pass && e->expand;
    const int *rlist = nullptr, *rcount = nullptr;  // work list of the walking kernel
    ItemBuffers ib = {};
    int fwaves = 0, fring = 0;
    size_t flds = 0;
    // (a big batch goes to the model's certificate build, if the engine found one: load_spec)
    const SpecLib *flib = (e->spec_cert && e->fused_cert && e->fused_cert_min_edges > 0 && E >= e->fused_cert_min_edges &&
                           E > (int64_t)e->fused_single_max) ? e->spec_cert : e->spec;
    const bool fused = expand && !e->fused_skip_once && fused_plan(e, &fwaves, &flds, &fring, flib);
    e->fused_skip_once = false;  // (set by the host-pointer entry point for the launch that follows it)
    if (expand) {
      // per-edge scratch: walking list, step fractions, claim words of the undecided-edge list
      if ((size_t)E > e->llist_cap) {
        for (void *ptr : {(void *)e->d_tstep, (void *)e->d_llist, (void *)e->d_eclaim})
          if (ptr) HIP_TRY(hipFree(ptr));
        e->d_tstep = nullptr;
        e->d_llist = e->d_eclaim = nullptr;
        e->llist_cap = 0;
        HIP_TRY(hipMalloc(&e->d_llist, (size_t)E * sizeof(int)));
        HIP_TRY(hipMalloc(&e->d_tstep, (size_t)E * sizeof(double)));
        HIP_TRY(hipMalloc(&e->d_eclaim, (size_t)E * sizeof(int)));
        HIP_TRY(hipMemsetAsync(e->d_eclaim, 0, (size_t)E * sizeof(int), e->stream));
        e->claim_gen = 0;
        e->llist_cap = (size_t)E;
      }
      if (++e->claim_gen == std::numeric_limits<int>::max()) {  // generations never repeat between clears
        HIP_TRY(hipMemsetAsync(e->d_eclaim, 0, e->llist_cap * sizeof(int), e->stream));
        e->claim_gen = 1;
      }
    }
    if (expand && !fused) {
      // room for 8 waypoints per edge on average, and never less than a quarter of a million items:
      // a handful of long edges (path shortcutting) is best served one waypoint per lane, too
      size_t want = std::min<size_t>(std::max<size_t>((size_t)E * 8, (size_t)1 << 18) + 4096, e->item_cap_limit);
      want = (want + fblock - 1) / fblock * fblock;  // whole blocks (the item space is split into regions of them)
      if (want > e->item_cap) {
        for (void *ptr : {(void *)e->d_itemedge, (void *)e->d_itemidx})
          if (ptr) HIP_TRY(hipFree(ptr));
        e->d_itemedge = e->d_itemidx = nullptr;
        e->item_cap = 0;
        HIP_TRY(hipMalloc(&e->d_itemedge, want * sizeof(int)));
        HIP_TRY(hipMalloc(&e->d_itemidx, want * sizeof(int)));
        e->item_cap = want;
      }
      const int kmax = (int)std::min<size_t>(std::max<size_t>(e->item_cap / (size_t)E, kExpandMinWaypoints), 1 << 16);
      double *ckpt = nullptr;
      if (kmax >= kCkptEvery) {
        // rows are sized by nq, the upper bound of nplan (mjpl_set_planning may widen the planning set)
        if (e->itemck_cap < e->item_cap) {
          if (e->d_itemck) HIP_TRY(hipFree(e->d_itemck));
          e->d_itemck = nullptr; e->itemck_cap = 0;
          HIP_TRY(hipMalloc(&e->d_itemck, e->item_cap * std::max<size_t>(1, e->m.nq) * sizeof(double)));
          e->itemck_cap = e->item_cap;
        }
        ckpt = e->d_itemck;
      }
      // regions: enough workgroups behind each counter to fill it evenly, few enough waves to not queue up
      const int regions = (int)std::max<size_t>(1, std::min<size_t>(std::min<size_t>(kItemRegions, fgrid / 8), e->item_cap / fblock));
      const int regcap = (int)(e->item_cap / regions / fblock * fblock);
      ib = ItemBuffers{e->d_itemedge, e->d_itemidx, e->d_ucount + 5 * kCtr, regions * regcap, regions, regcap,
                       e->d_ucount + (5 + kItemRegions) * kCtr, e->d_llist,
                       e->d_icount + kCtr, kmax, e->d_tstep, ckpt, e->d_eclaim, e->claim_gen};
    }
    if (fused) {
      // ---- ONE filter kernel for the launch (mjpl_fused.h): endpoint tiles and waypoint tiles from one work pool
      FusedArgs fa = {};
      fa.ip = e->d_ip; fa.nip = (int)e->ip.size();
      fa.fp = e->d_fp; fa.nfp = (int)e->fp.size();
      fa.QA = dQA; fa.QB = dQB; fa.E = (long long)E; fa.layout = layout;
      fa.tol = e->filter_tol; fa.step = step;
      fa.valid = dvalid; fa.first_bad = dfb;
      fa.status = e->d_status; fa.ulist = e->d_ulist; fa.ucount = e->d_ucount;
      fa.uc = uc;
      fa.llist = e->d_llist; fa.lcount = e->d_icount + kCtr;
      fa.claim = e->d_eclaim; fa.gen = e->claim_gen;
      fa.tstep = e->d_tstep;
      fa.item_count = e->d_ucount + 5 * kCtr;
      fa.surv_count = e->d_ucount + (5 + kItemRegions) * kCtr;
      fa.cert_count = e->d_ucount + (size_t)kCtrCertified * kCtr;
      fa.cert = e->fused_cert;
      fa.zero_next = zero_next;
      fa.kmax = e->fused_kmax; fa.pool = fring; fa.policy = e->fused_policy;
      // a batch that leaves most of the chip idle: one round of checks instead of two (the endpoint as an item)
      fa.single = (E <= (int64_t)e->fused_single_max) ? 1 : 0;
      if (e->fused_dbg_path) {  // diagnostic runs: per-wave counters of the most recent launch, written out at mjpl_destroy
        if (!e->d_fused_dbg) {
          HIP_TRY(hipMalloc(&e->d_fused_dbg, kFusedDbgWaves * 8 * sizeof(unsigned long long)));
        }
        HIP_TRY(hipMemsetAsync(e->d_fused_dbg, 0, kFusedDbgWaves * 8 * sizeof(unsigned long long), e->stream));
        fa.dbg = e->d_fused_dbg;
      }
      if (flib) {
        rc = flib->fused(e->stream, fwaves, flds, fa) == 0 ? MJPL_OK : fail(MJPL_E_HIP, "specialised fused kernel failed to launch");
      } else {
        rc = dispatch_filter(e, [&](auto S, auto W, auto M) -> int {
          if constexpr (decltype(S)::value <= kQueuedMaxSlots) {
            constexpr int NW = decltype(M)::value ? 4 : kFusedWaves;
            auto kern = k_edges_fused<void, decltype(S)::value, decltype(W)::value, decltype(M)::value, NW>;
            if (fwaves != NW) return fail(MJPL_E_ARG, "fused kernel: %d waves per workgroup asked for, this build has %d", fwaves, NW);
            int r = allow_lds(kern, flds);
            if (r != MJPL_OK) return r;
            if (fused_launch(kern, NW, flds, fa, e->stream) != hipSuccess) return fail(MJPL_E_HIP, "the fused kernel failed to launch");
            return MJPL_OK;
          } else {
            return fail(MJPL_E_ARG, "the fused kernel serves the queued interpreter");
          }
        });
      }
      if (rc != MJPL_OK) return rc;
      guard.armed = false;
      MJPL_MARK(1);
      rlist = e->d_llist;  // what is left for the walking role
      rcount = e->d_icount + kCtr;
    }
    if (two_pass && !fused) {
      if ((size_t)E > e->slist_cap) {
        if (e->d_slist) HIP_TRY(hipFree(e->d_slist));
        e->d_slist = nullptr; e->slist_cap = 0;
        HIP_TRY(hipMalloc(&e->d_slist, (size_t)E * sizeof(int)));
        e->slist_cap = (size_t)E;
      }
      rlist = e->d_slist;
      rcount = e->d_ucount + 2 * kCtr;
      // persistent grids (one wave per tile of 64, tiles from a device counter): the default for the
      // queued interpreter when the interior waypoints become items
      const bool pw = (e->persist < 0 ? e->spec != nullptr : e->persist != 0) && expand && !e->immediate();
      const size_t ldsp = persistent_lds_bytes((int)e->qidx.size(), e->nsave, e->fp.size(), e->filter_mbox());
      int *etiles = e->d_ucount + kCtrEndpointTiles * kCtr;
      if (pw && e->spec)
        rc = e->spec->endpoints_pw(e->stream, ldsp, e->d_ip, (int)e->ip.size(), e->d_fp, (int)e->fp.size(), dQA, dQB, E, layout,
                                   e->filter_tol, dvalid, dfb, e->d_status, e->d_ulist, e->d_ucount, uc, ib, step, zero_next,
                                   etiles) == 0 ? MJPL_OK : fail(MJPL_E_HIP, "specialised endpoint kernel failed to launch");
      else if (pw)
        rc = dispatch_filter(e, [&](auto S, auto W, auto M) -> int {
          if constexpr (decltype(S)::value <= kQueuedMaxSlots) {
            auto kern = k_filter_endpoints_pw<void, decltype(S)::value, decltype(W)::value, decltype(M)::value>;
            int r = allow_lds(kern, ldsp);
            if (r != MJPL_OK) return r;
            hipLaunchKernelGGL(kern, dim3(persistent_grid(kern, ldsp, (E + 63) / 64)), dim3(kBlock), ldsp, e->stream, e->d_ip,
                               (int)e->ip.size(), e->d_fp, (int)e->fp.size(), dQA, dQB, E, layout, e->filter_tol, dvalid, dfb,
                               e->d_status, e->d_ulist, e->d_ucount, uc, ib, step, zero_next, etiles);
            return MJPL_OK;
          } else {
            return fail(MJPL_E_ARG, "persistent kernels serve the queued interpreter");
          }
        });
      else if (e->spec)
        rc = e->spec->endpoints(e->stream, fgrid, (unsigned)fblock, ldse, e->d_ip, (int)e->ip.size(), e->d_fp, (int)e->fp.size(), dQA,
                                dQB, E, layout, e->filter_tol, dvalid, dfb, e->d_status, e->d_ulist, e->d_ucount, uc, e->d_slist,
                                e->d_ucount + 2 * kCtr, ib, step, zero_next) == 0 ? MJPL_OK
             : fail(MJPL_E_HIP, "specialised endpoint kernel failed to launch");
      else rc = dispatch_filter(e, [&](auto S, auto W, auto M) -> int {
        auto kern = k_filter_endpoints<void, decltype(S)::value, decltype(W)::value, decltype(M)::value>;
        int r = allow_lds(kern, ldse);
        if (r != MJPL_OK) return r;
        hipLaunchKernelGGL(kern, dim3(fgrid), dim3(fblock), ldse, e->stream, e->d_ip, (int)e->ip.size(),
                           e->d_fp, (int)e->fp.size(), dQA, dQB, E, layout, e->filter_tol, dvalid, dfb,
                           e->d_status, e->d_ulist, e->d_ucount, uc, e->d_slist, e->d_ucount + 2 * kCtr, ib, step,
                           zero_next);
        return MJPL_OK;
      });
      if (rc != MJPL_OK) return rc;
      guard.armed = false;
    }
    if (!fused) MJPL_MARK(1);  // after k_filter_endpoints (nothing ran yet in a one-pass launch)
    if (expand && !fused) {
      const unsigned igrid = (unsigned)(ib.cap / fblock);  // (regions of whole blocks)
      const EdgeSource src = {dQA, dQB, (long long)E, layout, step, ib.ckpt};
      const bool pw = (e->persist < 0 ? e->spec != nullptr : e->persist != 0) && !e->immediate();
      const size_t ldsp = persistent_lds_bytes((int)e->qidx.size(), e->nsave, e->fp.size(), e->filter_mbox());
      int *itiles = e->d_ucount + kCtrItemTiles * kCtr;
      if (pw && e->spec)
        rc = e->spec->items_pw(e->stream, ldsp, e->d_ip, (int)e->ip.size(), e->d_fp, (int)e->fp.size(), ib, src, e->filter_tol, dvalid,
                               dfb, e->d_ulist, e->d_ucount, uc, itiles) == 0 ? MJPL_OK
             : fail(MJPL_E_HIP, "specialised item kernel failed to launch");
      else if (pw)
        rc = dispatch_filter(e, [&](auto S, auto W, auto M) -> int {
          if constexpr (decltype(S)::value <= kQueuedMaxSlots) {
            auto kern = k_filter_items_pw<void, decltype(S)::value, decltype(W)::value, decltype(M)::value>;
            int r = allow_lds(kern, ldsp);
            if (r != MJPL_OK) return r;
            hipLaunchKernelGGL(kern, dim3(persistent_grid(kern, ldsp, (long long)ib.cap / 64)), dim3(kBlock), ldsp, e->stream, e->d_ip,
                               (int)e->ip.size(), e->d_fp, (int)e->fp.size(), ib, src, e->filter_tol, dvalid, dfb, e->d_ulist,
                               e->d_ucount, uc, itiles);
            return MJPL_OK;
          } else {
            return fail(MJPL_E_ARG, "persistent kernels serve the queued interpreter");
          }
        });
      else if (e->spec)
        rc = e->spec->items(e->stream, igrid, (unsigned)fblock, ldsq, e->d_ip, (int)e->ip.size(), e->d_fp, (int)e->fp.size(), ib,
                            src, e->filter_tol, dvalid, dfb, e->d_ulist, e->d_ucount, uc) == 0 ? MJPL_OK
             : fail(MJPL_E_HIP, "specialised item kernel failed to launch");
      else rc = dispatch_filter(e, [&](auto S, auto W, auto M) -> int {
        auto kern = k_filter_items<void, decltype(S)::value, decltype(W)::value, decltype(M)::value>;
        int r = allow_lds(kern, ldsq);
        if (r != MJPL_OK) return r;
        hipLaunchKernelGGL(kern, dim3(igrid), dim3(fblock), ldsq, e->stream, e->d_ip, (int)e->ip.size(),
                           e->d_fp, (int)e->fp.size(), ib, src, e->filter_tol, dvalid, dfb, e->d_ulist, e->d_ucount,
                           uc);
        return MJPL_OK;
      });
      if (rc != MJPL_OK) return rc;
      rlist = e->d_llist;  // what is left for the walking kernel
      rcount = e->d_icount + kCtr;
    }
    MJPL_MARK(2);  // after k_filter_items
    if (expand && e->fused_tail && !e->immediate()) {
      // everything behind the item pass in one launch: walking role over the long-edge list, pair
      // re-check, exact edge role over the undecided-edge list (k_tail)
      MJPL_MARK(3);
      const size_t ldst = std::max(ldsf, lds_bytes(e, 1));
      TailArgs ta = {};
      ta.ip = e->d_ip; ta.nip = (int)e->ip.size();
      ta.fp = e->d_fp; ta.nfp = (int)e->fp.size();
      ta.dp = e->d_dp; ta.ndp = (int)e->dp.size();
      ta.gt = GeomTable{e->d_geomtab, e->moving_base};
      ta.uc = uc;
      ta.QA = dQA; ta.QB = dQB; ta.E = (long long)E; ta.step = step; ta.layout = layout; ta.flags = flags;
      ta.tol = e->filter_tol;
      ta.valid = dvalid; ta.first_bad = dfb;
      ta.status = e->d_status; ta.ulist = e->d_ulist; ta.ucount = e->d_ucount;
      ta.llist = e->d_llist; ta.lcount = e->d_icount + kCtr;
      ta.done = e->d_ucount + kCtrTailDone * kCtr;
      const int eblocks = (int)std::min<int64_t>((E + kBlock - 1) / kBlock, 64);
      ta.nw = eblocks;
      ta.np = (int)std::min<size_t>((uc.cap + kBlock - 1) / kBlock, 64);
      ta.nx = eblocks;
      if (e->spec) {
        rc = e->spec->tail(e->stream, ldst, ta) == 0 ? MJPL_OK : fail(MJPL_E_HIP, "specialised tail kernel failed to launch");
      } else {
        auto go = [&](auto SF, auto SD, auto W, auto M) -> int {
          auto kern = k_tail<void, decltype(SF)::value, decltype(SD)::value, decltype(W)::value, decltype(M)::value>;
          int r = allow_lds(kern, ldst);
          if (r != MJPL_OK) return r;
          hipLaunchKernelGGL(kern, dim3((unsigned)(ta.nw + ta.np + ta.nx)), dim3(kBlock), ldst, e->stream, ta);
          return MJPL_OK;
        };
        if (e->exact_general())  // (moving boxes: 24-slot queued filter build, general exact build)
          rc = go(std::integral_constant<int, kQueuedMaxSlots>{}, std::integral_constant<int, 32>{}, std::true_type{}, std::true_type{});
        else
          rc = dispatch_variant(e, [&](auto S, auto W, auto M) -> int { return go(S, S, W, M); });
      }
      if (rc != MJPL_OK) return rc;
      guard.armed = false;
      MJPL_MARK(4);
      MJPL_MARK(5);
      HIP_TRY(hipGetLastError());
      return MJPL_OK;
    }
    rc = dispatch_filter(e, [&](auto S, auto W, auto M) -> int {
      auto kern = k_filter_edges<decltype(S)::value, decltype(W)::value, decltype(M)::value>;
      int r = allow_lds(kern, ldsf);
      if (r != MJPL_OK) return r;
      hipLaunchKernelGGL(kern, dim3(fgrid), dim3(fblock), ldsf, e->stream, e->d_ip, (int)e->ip.size(),
                         e->d_fp, (int)e->fp.size(), dQA, dQB, E, step, layout,
                         two_pass ? (flags | MJPL_EDGE_INTERIOR_ONLY) : flags, e->filter_tol, dvalid,
                         dfb, e->d_status, e->d_ulist, e->d_ucount, uc, rlist, rcount,
                         two_pass ? (int *)nullptr : zero_next);
      return MJPL_OK;
    });
    if (rc != MJPL_OK) return rc;
    guard.armed = false;
    MJPL_MARK(3);  // after k_filter_edges (the walking kernel)
    // undecided waypoints: exact configuration kernel in patch mode (grid sized for a generous
    // share of the batch; surplus blocks return at once)
    const size_t ldsc = lds_bytes(e, 1);
    const unsigned pgrid = (unsigned)std::min<size_t>((uc.cap + kBlock - 1) / kBlock, 1024);
    if (e->immediate()) {  // immediate filter: also whole configurations (entries with ga < 0)
      rc = dispatch_variant(e, [&](auto S, auto W, auto M) -> int {
        auto kern = k_check_configs<decltype(S)::value, decltype(W)::value, decltype(M)::value>;
        int r = allow_lds(kern, ldsc);
        if (r != MJPL_OK) return r;
        hipLaunchKernelGGL(kern, dim3(pgrid), dim3(kBlock), ldsc, e->stream, e->d_ip, (int)e->ip.size(),
                           e->d_dp, (int)e->dp.size(), (const double *)uc.q, (int64_t)0, (int)MJPL_AOS, dvalid,
                           (unsigned long long *)nullptr, (const int *)nullptr, (const int *)nullptr, uc, dfb);
        return MJPL_OK;
      });
      if (rc != MJPL_OK) return rc;
    }
    {  // single geom pairs (entries with ga >= 0)
      rc = launch_patch(e, pgrid, ldsc, uc, dvalid, dfb);
      if (rc != MJPL_OK) return rc;
    }
    MJPL_MARK(4);  // after the exact re-check of undecided pairs / configurations
  } else if (fused_f64_ok(e) && !(flags & MJPL_EDGE_INTERIOR_ONLY) && E < (int64_t)1 << 30) {
    // ---- filter off, the float64 checks through the pool (mjpl_fused.h: k_edges_fused_f64): endpoints, then the interior
    // waypoints of the survivors as items; edges of more than kFusedF64Kmax waypoints are left to k_check_edges below
    int rc = ulist_reserve(e, E);
    if (rc != MJPL_OK) return rc;
    if ((size_t)E > e->llist_cap) {
      for (void *ptr : {(void *)e->d_tstep, (void *)e->d_llist, (void *)e->d_eclaim})
        if (ptr) HIP_TRY(hipFree(ptr));
      e->d_tstep = nullptr;
      e->d_llist = e->d_eclaim = nullptr;
      e->llist_cap = 0;
      HIP_TRY(hipMalloc(&e->d_llist, (size_t)E * sizeof(int)));
      HIP_TRY(hipMalloc(&e->d_tstep, (size_t)E * sizeof(double)));
      HIP_TRY(hipMalloc(&e->d_eclaim, (size_t)E * sizeof(int)));
      HIP_TRY(hipMemsetAsync(e->d_eclaim, 0, (size_t)E * sizeof(int), e->stream));
      e->claim_gen = 0;
      e->llist_cap = (size_t)E;
    }
    int *zero_next = next_counters(e);
    CounterGuard guard{e, true};
    const int nplan = (int)e->qidx.size();
    // the check through the candidate queues (narrowphase with full lanes) where the model has the queued build's
    // shape and eight waves' queues fit beside the rows and a pool; else the immediate interpreter (MJPL_F64_QUEUED=0)
    const bool queued = e->f64_queued && !e->exact_general() &&
                        fused_f64_lds_bytes(kFusedF64Waves, nplan, e->nsave, 64 * kFusedF64Waves + 64, true) <= (size_t)160 * 1024;
    const size_t base = fused_f64_lds_bytes(kFusedF64Waves, nplan, e->nsave, 0, queued);
    int pool = (int)std::min<size_t>(kFusedMaxPool, ((size_t)160 * 1024 - base) / kFusedEntryBytes / 64 * 64);
    if (e->fused_pool_cap > 0) pool = std::max(64 * kFusedF64Waves + 64, std::min(pool, e->fused_pool_cap / 64 * 64));
    const size_t flds = fused_f64_lds_bytes(kFusedF64Waves, nplan, e->nsave, pool, queued);
    FusedArgs fa = {};
    fa.ip = e->d_ip; fa.nip = (int)e->ip.size();
    fa.dp = e->d_dp; fa.ndp = (int)e->dp.size();
    fa.QA = dQA; fa.QB = dQB; fa.E = (long long)E; fa.layout = layout;
    fa.step = step;
    fa.valid = dvalid; fa.first_bad = dfb;
    fa.status = e->d_status;
    fa.llist = e->d_llist; fa.lcount = e->d_icount + kCtr;
    fa.item_count = e->d_ucount + 5 * kCtr;
    fa.surv_count = e->d_ucount + (5 + kItemRegions) * kCtr;
    fa.cert_count = e->d_ucount + (size_t)kCtrCertified * kCtr;
    fa.cert = 0;
    fa.zero_next = zero_next;
    fa.kmax = kFusedF64Kmax; fa.pool = pool; fa.policy = e->fused_policy;
    MJPL_MARK(0);
    // a model's library may carry the check as straight-line code (ExactFull; MJPL_F64_SPEC=0: the interpreting kernel)
    bool generated = false;
    if (queued && e->spec && e->spec->fused_f64 && e->f64_spec && e->spec->generic_rows == 0) {
      const int r2 = e->spec->fused_f64(e->stream, kFusedF64Waves, flds, fa);
      if (r2 == 0) generated = true;
      else if (r2 != -1) return fail(MJPL_E_HIP, "the generated float64 pool kernel failed to launch");
    }
    rc = generated ? MJPL_OK : dispatch_variant(e, [&](auto S, auto W, auto M) -> int {
      auto go = [&](auto kern) -> int {
        int r = allow_lds(kern, flds);
        if (r != MJPL_OK) return r;
        if (fused_launch(kern, kFusedF64Waves, flds, fa, e->stream) != hipSuccess) return fail(MJPL_E_HIP, "the float64 pool kernel failed to launch");
        return MJPL_OK;
      };
      if constexpr (!decltype(M)::value && decltype(S)::value <= 16) {
        if (queued) return go(k_edges_fused_f64<decltype(S)::value, decltype(W)::value, false, kFusedF64Waves, true>);
      }
      return go(k_edges_fused_f64<decltype(S)::value, decltype(W)::value, decltype(M)::value, kFusedF64Waves, false>);
    });
    if (rc != MJPL_OK) return rc;
    guard.armed = false;
    for (int k = 1; k <= 4; k++) MJPL_MARK(k);
    // the walking list (long edges): k_check_edges, interior waypoints only (the endpoint's verdict stands)
    const size_t ldsw = lds_bytes(e, 1);
    rc = dispatch_variant(e, [&](auto S, auto W, auto M) -> int {
      auto kern = k_check_edges<decltype(S)::value, decltype(W)::value, decltype(M)::value>;
      int r = allow_lds(kern, ldsw);
      if (r != MJPL_OK) return r;
      hipLaunchKernelGGL(kern, dim3(grid), dim3(kBlock), ldsw, e->stream, e->d_ip, (int)e->ip.size(), e->d_dp, (int)e->dp.size(),
                         dQA, dQB, E, step, layout, flags | MJPL_EDGE_INTERIOR_ONLY, dvalid, dfb, e->d_status,
                         (const int *)e->d_llist, (const int *)(e->d_icount + kCtr));
      return MJPL_OK;
    });
    if (rc != MJPL_OK) return rc;
    MJPL_MARK(5);
    HIP_TRY(hipGetLastError());
    return MJPL_OK;
  } else {
    for (int k = 0; k <= 4; k++) MJPL_MARK(k);  // filter off: the exact kernel is the whole launch
  }
  const size_t lds = lds_bytes(e, 1);
  int rc = dispatch_variant(e, [&](auto S, auto W, auto M) -> int {
    auto kern = k_check_edges<decltype(S)::value, decltype(W)::value, decltype(M)::value>;
    int r = allow_lds(kern, lds);
    if (r != MJPL_OK) return r;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kBlock), lds, e->stream, e->d_ip, (int)e->ip.size(),
                       e->d_dp, (int)e->dp.size(), dQA, dQB, E, step, layout, flags, dvalid, dfb, e->d_status,
                       filter ? (const int *)e->d_ulist : nullptr, filter ? (const int *)e->d_ucount : nullptr);
    return MJPL_OK;
  });
  if (rc != MJPL_OK) return rc;
  MJPL_MARK(5);  // after k_check_edges
  HIP_TRY(hipGetLastError());
  return MJPL_OK;
}

int check_common(const mjpl_engine *e, const void *a, int64_t n, int layout) {
  if (!e) return fail(MJPL_E_ARG, "engine is NULL");
  if (n < 0) return fail(MJPL_E_ARG, "negative batch size");
  if (n > 0 && !a) return fail(MJPL_E_ARG, "NULL batch pointer");
  if (layout != MJPL_SOA && layout != MJPL_AOS) return fail(MJPL_E_ARG, "layout must be MJPL_SOA or MJPL_AOS");
  return MJPL_OK;
}

}  // namespace

// ------------------------------------------------------------------------------- C ABI

extern "C" {

const char *mjpl_last_error(void) { return g_err.c_str(); }

#ifdef MJPL_STAMPS
// diagnostic builds only: read and clear the phase counters of the queued interpreter
int mjpl_debug_stamps(unsigned long long *out) {
  unsigned long long z[16] = {0};
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(mjpl::g_stamps), sizeof(z)) != hipSuccess) return -1;
  if (hipMemcpyToSymbol(HIP_SYMBOL(mjpl::g_stamps), z, sizeof(z)) != hipSuccess) return -1;
  return 0;
}
#endif
const char *mjpl_version(void) { return "mjpl_hip 0.1 (gfx950)"; }

int mjpl_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

namespace {
// copy the model tables, the allowed body pairs and the environment overrides into a fresh engine
int engine_from_desc(mjpl_engine *e, const mjpl_model_desc *d, const int32_t *allowed_bodies, int32_t nallowed) {
  HostModel &m = e->m;
  m.nq = d->nq; m.njnt = d->njnt; m.nbody = d->nbody; m.ngeom = d->ngeom;
  m.body_parentid = copy_n(d->body_parentid, m.nbody);
  m.body_weldid = copy_n(d->body_weldid, m.nbody);
  m.body_jntadr = copy_n(d->body_jntadr, m.nbody);
  m.body_jntnum = copy_n(d->body_jntnum, m.nbody);
  m.body_pos = copy_n(d->body_pos, 3 * (size_t)m.nbody);
  m.body_quat = copy_n(d->body_quat, 4 * (size_t)m.nbody);
  m.jnt_type = copy_n(d->jnt_type, m.njnt);
  m.jnt_qposadr = copy_n(d->jnt_qposadr, m.njnt);
  m.jnt_axis = copy_n(d->jnt_axis, 3 * (size_t)m.njnt);
  m.jnt_pos = copy_n(d->jnt_pos, 3 * (size_t)m.njnt);
  m.qpos0 = copy_n(d->qpos0, m.nq);
  m.geom_type = copy_n(d->geom_type, m.ngeom);
  m.geom_bodyid = copy_n(d->geom_bodyid, m.ngeom);
  m.geom_contype = copy_n(d->geom_contype, m.ngeom);
  m.geom_conaffinity = copy_n(d->geom_conaffinity, m.ngeom);
  m.geom_size = copy_n(d->geom_size, 3 * (size_t)m.ngeom);
  m.geom_pos = copy_n(d->geom_pos, 3 * (size_t)m.ngeom);
  m.geom_quat = copy_n(d->geom_quat, 4 * (size_t)m.ngeom);
  m.geom_rbound = copy_n(d->geom_rbound, m.ngeom);
  m.geom_margin = copy_n(d->geom_margin, m.ngeom);
  for (int b = 0; b < m.nbody; b++)
    if (m.body_parentid[b] < 0 || m.body_parentid[b] > b || m.body_weldid[b] < 0 || m.body_weldid[b] > b)
      return (fail(MJPL_E_ARG, "body %d: parent/weld ids must precede the body", b));
  for (int g = 0; g < m.ngeom; g++)
    if (m.geom_bodyid[g] < 0 || m.geom_bodyid[g] >= m.nbody)
      return (fail(MJPL_E_ARG, "geom %d: body id out of range", g));
  for (int a = 0; a < nallowed; a++) {
    int b1 = allowed_bodies[2 * a], b2 = allowed_bodies[2 * a + 1];
    if (b1 < 0 || b2 < 0 || b1 >= m.nbody || b2 >= m.nbody)
      return (fail(MJPL_E_ARG, "allowed body pair %d out of range", a));
    e->allowed.insert({std::min(b1, b2), std::max(b1, b2)});
  }
  e->qidx.resize(m.nq);
  for (int k = 0; k < m.nq; k++) e->qidx[k] = k;
  e->qbase = m.qpos0;
  // (no switch is read from the environment here: mjpl_set_option; MJPL_DEBUG=1 lets the variables of old through, below)
  return MJPL_OK;
}
}  // namespace

void apply_debug_environment(mjpl_engine *e);

int mjpl_create(const mjpl_model_desc *d, const int32_t *allowed_bodies, int32_t nallowed,
                int32_t device, mjpl_engine **out) {
  if (!d || !out) return fail(MJPL_E_ARG, "mjpl_create: NULL argument");
  *out = nullptr;
  if (d->nq < 0 || d->njnt < 0 || d->nbody < 1 || d->ngeom < 0 || nallowed < 0)
    return fail(MJPL_E_ARG, "mjpl_create: negative size");
  if (d->nq != d->njnt) return fail(MJPL_E_JOINT, "nq != njnt: only 1-DoF joints are supported");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(MJPL_E_NODEVICE, "no HIP device available (this library has no CPU fallback)");
  if (device < 0 || device >= ndev) return fail(MJPL_E_NODEVICE, "device %d out of range [0,%d)", device, ndev);

  mjpl_engine *e = new mjpl_engine();
  e->device = device;
  auto bail = [&](int rc) { mjpl_destroy(e); return rc; };
  if (hipSetDevice(device) != hipSuccess) return bail(fail(MJPL_E_HIP, "hipSetDevice(%d) failed", device));
  if (hipGetDeviceProperties(&e->prop, device) != hipSuccess) return bail(fail(MJPL_E_HIP, "hipGetDeviceProperties failed"));
  if (hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking) != hipSuccess) return bail(fail(MJPL_E_HIP, "hipStreamCreate failed"));
  if (hipMalloc(&e->d_status, sizeof(int)) != hipSuccess) return bail(fail(MJPL_E_HIP, "hipMalloc failed"));
  (void)hipMemset(e->d_status, 0, sizeof(int));

  int rc = engine_from_desc(e, d, allowed_bodies, nallowed);
  if (rc != MJPL_OK) return bail(rc);
  apply_debug_environment(e);
  rc = compile_program(e);
  if (rc != MJPL_OK) return bail(rc);
  *out = e;
  return MJPL_OK;
}

void mjpl_destroy(mjpl_engine *e) {
  if (e && e->d_fused_dbg && e->fused_dbg_path) {
    (void)hipSetDevice(e->device);
    (void)hipStreamSynchronize(e->stream);
    std::vector<unsigned long long> h(kFusedDbgWaves * 8);
    if (hipMemcpy(h.data(), e->d_fused_dbg, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost) == hipSuccess)
      if (FILE *f = fopen(e->fused_dbg_path, "wb")) {
        fwrite(h.data(), sizeof(unsigned long long), h.size(), f);
        fclose(f);
      }
    (void)hipFree(e->d_fused_dbg);
  }
  if (!e) return;
  (void)hipSetDevice(e->device);
  if (e->stream) (void)hipStreamSynchronize(e->stream);
  if (e->comm) (void)mjpl_comm_destroy(e);
  for (int k = 0; k < 7; k++)
    if (e->stage[k]) (void)hipFree(e->stage[k]);
  if (e->h_pin) (void)hipHostFree(e->h_pin);
  if (e->d_ip) (void)hipFree(e->d_ip);
  if (e->d_dp) (void)hipFree(e->d_dp);
  if (e->d_fp_base) (void)hipFree(e->d_fp_base);
  if (e->d_ulist) (void)hipFree(e->d_ulist);
  if (e->d_ucount_base) (void)hipFree(e->d_ucount_base);
  if (e->d_slist) (void)hipFree(e->d_slist);
  if (e->d_ucga) (void)hipFree(e->d_ucga);
  if (e->d_ucgb) (void)hipFree(e->d_ucgb);
  if (e->d_geomtab) (void)hipFree(e->d_geomtab);
  if (e->d_nn) (void)hipFree(e->d_nn);
  if (e->d_nn16) (void)hipFree(e->d_nn16);
  if (e->d_nn_tmp) (void)hipFree(e->d_nn_tmp);
  if (e->d_tstep) (void)hipFree(e->d_tstep);
  if (e->d_itemck) (void)hipFree(e->d_itemck);
  if (e->d_itemedge) (void)hipFree(e->d_itemedge);
  if (e->d_itemidx) (void)hipFree(e->d_itemidx);
  if (e->d_llist) (void)hipFree(e->d_llist);
  if (e->d_eclaim) (void)hipFree(e->d_eclaim);
  if (e->d_ucq) (void)hipFree(e->d_ucq);
  if (e->d_ucedge) (void)hipFree(e->d_ucedge);
  if (e->d_ucidx) (void)hipFree(e->d_ucidx);
  if (e->d_status) (void)hipFree(e->d_status);
  if (e->stream) (void)hipStreamDestroy(e->stream);
  delete e;
}

int mjpl_program_dump(const mjpl_model_desc *d, const int32_t *allowed_bodies, int32_t nallowed, const int32_t *qidx,
                      int32_t nplan, const double *qpos_base, double filter_tol, int32_t *ip, int32_t *nip, float *fp,
                      double *dp, int32_t *ntab, mjpl_program_info *info) {
  if (!d || !nip || !ntab || !info) return fail(MJPL_E_ARG, "mjpl_program_dump: NULL argument");
  if (d->nq < 0 || d->njnt < 0 || d->nbody < 1 || d->ngeom < 0 || nallowed < 0) return fail(MJPL_E_ARG, "negative size");
  if (d->nq != d->njnt) return fail(MJPL_E_JOINT, "nq != njnt: only 1-DoF joints are supported");
  std::unique_ptr<mjpl_engine> e(new mjpl_engine());
  e->device = -1;  // host tables only: nothing is allocated or uploaded
  int rc = engine_from_desc(e.get(), d, allowed_bodies, nallowed);
  if (rc != MJPL_OK) return rc;
  if (qidx) {
    e->qidx.assign(qidx, qidx + nplan);
    for (int c : e->qidx)
      if (c < 0 || c >= d->nq) return fail(MJPL_E_ARG, "planning index %d out of range", c);
  }
  if (qpos_base) e->qbase.assign(qpos_base, qpos_base + d->nq);
  if (filter_tol > 0.0) { e->filter_tol_req = (float)filter_tol; e->filter_tol_user = true; }
  if ((rc = compile_program(e.get())) != MJPL_OK) return rc;
  memset(info, 0, sizeof(*info));
  info->hash = e->program_hash;
  // (maxs: the slot-file width of the FILTER kernels a library is built around -- 24 for the queued build of models with
  // moving boxes; immediate: only the immediate interpreter serves this program, nothing to specialise)
  info->maxs = e->filter_mbox() ? kQueuedMaxSlots : e->maxs; info->wbox = e->wbox; info->mbox = e->mbox || e->filter_mbox();
  info->immediate = e->immediate() || (e->exact_general() && !e->filter_mbox());
  info->filter_usable = e->filter_usable; info->filter_tol = e->filter_tol; info->nslots = e->nslots; info->nsave = e->nsave;
  info->spec_abi = MJPL_SPEC_ABI;
  info->robot_hash = e->robot_hash;
  info->scene_rows = kSceneRows;
  info->scene_ok = e->scene.empty() ? 0 : 1;
  const int32_t want_ip = (int32_t)e->ip.size(), want_tab = (int32_t)e->fp.size();
  const bool fits = ip && fp && *nip >= want_ip && *ntab >= want_tab;
  *nip = want_ip;
  *ntab = want_tab;
  if (!ip && !fp && !dp) return MJPL_OK;  // size query
  if (!fits) return fail(MJPL_E_CAPACITY, "mjpl_program_dump: buffers too small (%d control words, %d constants)", want_ip, want_tab);
  memcpy(ip, e->ip.data(), e->ip.size() * sizeof(int));
  memcpy(fp, e->fp.data(), e->fp.size() * sizeof(float));
  if (dp) memcpy(dp, e->dp.data(), e->dp.size() * sizeof(double));
  return MJPL_OK;
}

int mjpl_spec_probe(uint64_t hash, int32_t generic) {
  return find_spec(hash, generic != 0) ? 1 : 0;
}

int mjpl_spec_loaded(const mjpl_engine *e) { return (e && e->spec) ? (e->spec_generic ? 2 : 1) : 0; }

int mjpl_set_spec(mjpl_engine *e, int32_t enable) {
  if (!e) return fail(MJPL_E_ARG, "mjpl_set_spec: NULL engine");
  HIP_TRY(hipSetDevice(e->device));
  HIP_TRY(hipStreamSynchronize(e->stream));
  e->spec_off = enable == 0;
  e->spec_generic_only = enable == 2;
  return compile_program(e);  // (a scene-generic library wants its cull table in front of the float32 tables)
}

int mjpl_set_planning(mjpl_engine *e, const int32_t *qidx, int32_t nplan, const double *qpos_base) {
  if (!e || nplan < 0 || (nplan > 0 && !qidx)) return fail(MJPL_E_ARG, "mjpl_set_planning: bad argument");
  std::vector<int> q(qidx, qidx + nplan);
  std::vector<char> seen(e->m.nq, 0);
  for (int c : q) {
    if (c < 0 || c >= e->m.nq || seen[c]) return fail(MJPL_E_ARG, "planning index %d invalid or repeated", c);
    seen[c] = 1;
  }
  HIP_TRY(hipSetDevice(e->device));
  HIP_TRY(hipStreamSynchronize(e->stream));
  e->qidx = q;
  if (qpos_base) e->qbase.assign(qpos_base, qpos_base + e->m.nq);
  return compile_program(e);
}

int mjpl_set_filter(mjpl_engine *e, int32_t enable, double tol) {
  if (!e) return fail(MJPL_E_ARG, "mjpl_set_filter: NULL engine");
  if (enable && !(tol > 0.0 && tol < 1.0)) return fail(MJPL_E_ARG, "filter tolerance must be in (0, 1) metres");
  HIP_TRY(hipSetDevice(e->device));
  HIP_TRY(hipStreamSynchronize(e->stream));
  e->filter = enable != 0;
  if (enable && (!e->filter_tol_user || (float)tol != e->filter_tol_req)) {
    e->filter_tol_req = (float)tol;
    e->filter_tol_user = true;
    return compile_program(e);
  }
  return MJPL_OK;
}

int64_t mjpl_filter_last_undecided(mjpl_engine *e) {
  if (!e || !e->filter || !e->filter_usable || !e->d_ucount) return 0;
  int n[5 * kCtr];
  if (hipSetDevice(e->device) != hipSuccess) return -1;
  if (hipStreamSynchronize(e->stream) != hipSuccess) return -1;
  if (hipMemcpy(n, e->d_ucount, sizeof(n), hipMemcpyDeviceToHost) != hipSuccess) return -1;
  return (int64_t)n[0] + n[kCtr];
}

// Diagnostic: the (edge, check index, geom a, geom b) records the last filter launch handed to the exact pair
// kernel -- which pairs the float32 filter could not decide.  Returns how many there were (the arrays receive up to
// `cap` of them), or -1.
int64_t mjpl_filter_undecided_pairs(mjpl_engine *e, int32_t *edge, int32_t *idx, int32_t *ga, int32_t *gb, int64_t cap) {
  if (!e || !e->filter || !e->filter_usable || !e->d_ucount || !e->d_ucedge) return 0;
  if (hipSetDevice(e->device) != hipSuccess) return -1;
  if (hipStreamSynchronize(e->stream) != hipSuccess) return -1;
  int n = 0;
  if (hipMemcpy(&n, e->d_ucount + kCtr, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return -1;
  const size_t m = (size_t)std::max<int64_t>(0, std::min<int64_t>(std::min<int64_t>(n, cap), (int64_t)e->uc_cap));
  int32_t *dst[4] = {edge, idx, ga, gb};
  const int *src[4] = {e->d_ucedge, e->d_ucidx, e->d_ucga, e->d_ucgb};
  for (int k = 0; k < 4; k++)
    if (dst[k] && m && hipMemcpy(dst[k], src[k], m * sizeof(int32_t), hipMemcpyDeviceToHost) != hipSuccess) return -1;
  return n;
}

namespace {
// the launch's counters on the host (synchronises the stream)
bool read_counters(mjpl_engine *e, std::vector<int> &n) {
  n.assign((size_t)kNumCtr * kCtr, 0);
  if (hipSetDevice(e->device) != hipSuccess) return false;
  if (hipStreamSynchronize(e->stream) != hipSuccess) return false;
  return hipMemcpy(n.data(), e->d_ucount, n.size() * sizeof(int), hipMemcpyDeviceToHost) == hipSuccess;
}
}  // namespace

int64_t mjpl_filter_last_interior_edges(mjpl_engine *e) {
  if (!e || !e->filter || !e->d_ucount) return -1;
  std::vector<int> n;
  if (!read_counters(e, n)) return -1;
  int64_t total = n[2 * kCtr];  // (one counter without the lane-per-waypoint pass, one per region with it)
  for (int r = 0; r < kItemRegions; r++) total += n[(size_t)(5 + kItemRegions + r) * kCtr];
  return total;
}

int64_t mjpl_filter_last_items(mjpl_engine *e) {
  if (!e || !e->filter || !e->d_ucount) return -1;
  std::vector<int> n;
  if (!read_counters(e, n)) return -1;
  int64_t total = 0;  // (reserved slots, as before: a region's surplus went to the walking kernel)
  for (int r = 0; r < kItemRegions; r++) total += n[(size_t)(5 + r) * kCtr];
  return total;
}

int64_t mjpl_filter_last_certified(mjpl_engine *e) {
  if (!e || !e->filter || !e->d_ucount) return -1;
  std::vector<int> n;
  if (!read_counters(e, n)) return -1;
  return n[(size_t)kCtrCertified * kCtr];
}

int mjpl_get_info(const mjpl_engine *e, mjpl_info *out) {
  if (!e || !out) return fail(MJPL_E_ARG, "mjpl_get_info: NULL argument");
  memset(out, 0, sizeof(*out));
  out->device = e->device;
  out->nplan = (int)e->qidx.size();
  out->nmoving_geoms = e->nmoving;
  out->nstatic_geoms = e->nstatic;
  out->npairs = e->npairs;
  out->npairs_world = e->npairs_world;
  out->nslots = e->nslots;
  out->nsaves = e->nsave;
  out->lds_bytes_configs = (int)lds_bytes(e, 1);
  out->lds_bytes_edges = (int)lds_bytes(e, 1);
  out->filter_enabled = (e->filter && e->filter_usable) ? 1 : 0;
  out->filter_tol = e->filter_tol;
  out->filter_max_coord = (float)e->fmax_coord;
  out->filter_err_a = (float)e->ferr_a;
  out->filter_err_b = (float)e->ferr_b;
  out->filter_poisoned_geoms = e->npoisoned;
  out->filter_interpreter = e->immediate() ? 2 : (e->filter_mbox() ? 1 : 0);
  out->filter_block_threads = e->immediate() ? kBlock : kFilterBlock;
  out->lds_bytes_filter = (int)lds_bytes(e, 1, sizeof(float), out->filter_block_threads, !e->immediate(),
                                         e->immediate() ? sizeof(double) : sizeof(float));
  out->persistent_kernels = ((e->persist < 0 ? e->spec != nullptr : e->persist != 0) && e->two_pass && e->expand && !e->immediate()) ? 1 : 0;
  out->fused_tail = (e->fused_tail && e->two_pass && e->expand && !e->immediate()) ? 1 : 0;
  if (e->filter && e->filter_usable) {
    int nw = 0;
    out->fused_edges = fused_plan(e, &nw, nullptr) ? 1 : 0;
    out->fused_waves = out->fused_edges ? nw : 0;
  } else {  // (filter off: the float64 checks through the pool)
    out->fused_edges = fused_f64_ok(e) ? 1 : 0;
    out->fused_waves = out->fused_edges ? kFusedF64Waves : 0;
  }
  out->block_threads = kBlock;
  out->compute_units = e->prop.multiProcessorCount;
  strncpy(out->arch, e->prop.gcnArchName, sizeof(out->arch) - 1);
  return MJPL_OK;
}

// ---- device-resident

int mjpl_check_configs_dev(mjpl_engine *e, const double *dQ, int64_t N, int32_t layout, uint8_t *dvalid) {
  int rc = check_common(e, dQ, N, layout);
  if (rc != MJPL_OK) return rc;
  if (N > 0 && !dvalid) return fail(MJPL_E_ARG, "NULL output pointer");
  HIP_TRY(hipSetDevice(e->device));
  return launch_configs(e, dQ, N, layout, dvalid, nullptr);
}

int mjpl_check_configs_bits_dev(mjpl_engine *e, const double *dQ, int64_t N, int32_t layout, uint64_t *dbits) {
  int rc = check_common(e, dQ, N, layout);
  if (rc != MJPL_OK) return rc;
  if (N > 0 && !dbits) return fail(MJPL_E_ARG, "NULL output pointer");
  HIP_TRY(hipSetDevice(e->device));
  return launch_configs(e, dQ, N, layout, nullptr, reinterpret_cast<unsigned long long *>(dbits));
}

int mjpl_check_edges_dev(mjpl_engine *e, const double *dQA, const double *dQB, int64_t E, double step_dist,
                         int32_t layout, int32_t flags, uint8_t *dvalid, int32_t *dfirst_bad) {
  int rc = check_common(e, dQA, E, layout);
  if (rc != MJPL_OK) return rc;
  if (E > 0 && (!dQB || !dvalid)) return fail(MJPL_E_ARG, "NULL pointer");
  if (!(step_dist > 0.0)) return fail(MJPL_E_ARG, "`step_dist` must be > 0");
  HIP_TRY(hipSetDevice(e->device));
  return launch_edges(e, dQA, dQB, E, step_dist, layout, flags, dvalid, dfirst_bad);
}

int mjpl_take_status(mjpl_engine *e, int32_t *status) {
  if (!e || !status) return fail(MJPL_E_ARG, "mjpl_take_status: NULL argument");
  HIP_TRY(hipSetDevice(e->device));
  int s = 0;
  HIP_TRY(hipMemcpyAsync(&s, e->d_status, sizeof(int), hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipMemsetAsync(e->d_status, 0, sizeof(int), e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  if (s & (kStatusTailTimeout | kStatusFusedTimeout)) return fail(MJPL_E_HIP, "a kernel of the edge pipeline gave up waiting (status %d: 2 = the tail kernel for its walking workgroups, 4 = a wave of the fused kernel for its work pool)", s);
  *status = (s & kStatusNonFinite) ? MJPL_E_NONFINITE : MJPL_OK;
  return MJPL_OK;
}

// ---- options through the ABI (include/mjpl_hip.h).  One table: name -> how to read it, how to set it.
namespace {
double kt_sum(mjpl_engine *e, int c) {
  if (hipDeviceSynchronize() != hipSuccess) return -1.0;
  double sum = 0;
  for (int k = 0; k + 1 < e->kt_used[c]; k += 2) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, e->kt_ev[c][(size_t)k], e->kt_ev[c][(size_t)k + 1]) != hipSuccess) return -1.0;
    sum += ms;
  }
  return sum;
}
struct EngineOption {
  const char *name;
  double (*get)(mjpl_engine *);
  bool (*set)(mjpl_engine *, double);  // false: value out of range
};
#define MJPL_OPT_BOOL(NAME, FIELD) \
  {NAME, [](mjpl_engine *e) { return (double)(e->FIELD ? 1 : 0); }, [](mjpl_engine *e, double v) { e->FIELD = v != 0.0; return true; }}
#define MJPL_OPT_INT(NAME, FIELD, LO, HI)                                   \
  {NAME, [](mjpl_engine *e) { return (double)e->FIELD; }, [](mjpl_engine *e, double v) { \
     if (!(v >= (double)(LO) && v <= (double)(HI))) return false;            \
     e->FIELD = (decltype(e->FIELD))v;                                       \
     return true;                                                            \
   }}
const EngineOption kEngineOptions[] = {
    // ---- what shapes the compiled model or the launch pipeline (take effect at the next mjpl_set_planning / mjpl_set_spec /
    //      mjpl_set_filter, which compile again; the launch-time ones at the next launch)
    MJPL_OPT_BOOL("filter", filter),
    MJPL_OPT_BOOL("two_pass", two_pass),
    MJPL_OPT_BOOL("force_immediate", force_immediate),
    MJPL_OPT_BOOL("expand", expand),
    MJPL_OPT_BOOL("persist", persist),
    MJPL_OPT_BOOL("tail", fused_tail),
    MJPL_OPT_BOOL("fused", fused),
    MJPL_OPT_INT("fused_policy", fused_policy, -1000000, 1000000),
    MJPL_OPT_BOOL("fused_cert", fused_cert),
    MJPL_OPT_INT("fused_cert_min_edges", fused_cert_min_edges, 0, 1ll << 40),
    {"spec_cert_loaded", [](mjpl_engine *e) { return (double)(e->spec_cert ? 1 : 0); }, nullptr},
    MJPL_OPT_BOOL("f64_spec", f64_spec),
    MJPL_OPT_INT("fused_pool", fused_pool_cap, 0, 1 << 30),
    MJPL_OPT_BOOL("fused_mbox", fused_mbox),
    MJPL_OPT_BOOL("f64_queued", f64_queued),
    MJPL_OPT_INT("fused_single", fused_single_max, 0, 1 << 30),
    MJPL_OPT_INT("fused_kmax", fused_kmax, 2, 1 << 16),
    MJPL_OPT_INT("item_cap", item_cap_limit, 64, 1ll << 40),
    MJPL_OPT_INT("uc_cap", uc_cap_limit, 1, 1ll << 40),
    {"filter_tol", [](mjpl_engine *e) { return (double)e->filter_tol_req; },
     [](mjpl_engine *e, double v) { if (!(v > 0.0 && v < 1.0)) return false; e->filter_tol_req = (float)v; e->filter_tol_user = true; return true; }},
    MJPL_OPT_INT("zero_copy_bytes", zero_copy_bytes, 0, 1ll << 40),
    // ---- projection / IK rows
    MJPL_OPT_INT("rows_g", rows_g, 0, 64),
    MJPL_OPT_BOOL("pose_spec", pose_spec),
    MJPL_OPT_INT("pose_phase_steps", pose_phase_steps, 0, 1 << 20),
    // ---- the planner (read by mjpl_rrt_create)
    MJPL_OPT_INT("rrt_trace", rrt_trace, 0, 2),
    MJPL_OPT_BOOL("rrt_exact_counts", rrt_exact_counts),
    MJPL_OPT_BOOL("rrt_early_nn", rrt_early_nn),
    MJPL_OPT_INT("rrt_early_lanes", rrt_early_lanes, 1, 1 << 30),
    MJPL_OPT_INT("rrt_early_min_nodes", rrt_early_min_nodes, 1, 1ll << 40),
    MJPL_OPT_BOOL("rrt_early_next", rrt_early_next),
    MJPL_OPT_INT("rrt_proj_steps", rrt_proj_steps, 1, 1 << 30),
    MJPL_OPT_INT("rrt_proj_slots", rrt_proj_slots, 1, 1ll << 40),
    {"rrt_proj_g", [](mjpl_engine *e) { return (double)e->rrt_proj_g; },
     [](mjpl_engine *e, double v) { const int g = (int)v; e->rrt_proj_g = (g == 1 || g == 4 || g == 8 || g == 16 || g == 64) ? g : 0; return true; }},
    MJPL_OPT_BOOL("rrt_ahead", rrt_ahead),
    MJPL_OPT_INT("rrt_ahead_lanes", rrt_ahead_lanes, 1, 1 << 30),
    MJPL_OPT_INT("rrt_proj_waves", rrt_proj_waves, 1, 1 << 20),
    // ---- nearest neighbour
    MJPL_OPT_BOOL("nn_cells", nn_cells),
    MJPL_OPT_BOOL("nn_home", nn_home),
    MJPL_OPT_BOOL("nn_second_screen", nn_second_screen),
    MJPL_OPT_BOOL("nn_mfma", nn_mfma),
    {"nn_cells_min_nodes", [](mjpl_engine *e) { return (double)e->nn_cells_min; },
     [](mjpl_engine *e, double v) { if (!(v >= 0 && v < 9e15)) return false; e->nn_cells_min = (int64_t)v; return true; }},
    {"nn_cells_sample", [](mjpl_engine *e) { return (double)e->nn_cells_sample; },
     [](mjpl_engine *e, double v) { if (!(v >= 32 && v < 9e15)) return false; e->nn_cells_sample = std::max<int64_t>(1024, (int64_t)v) / 32 * 32; return true; }},
    {"nn_sample", [](mjpl_engine *e) { return (double)e->nn_sample; },
     [](mjpl_engine *e, double v) { if (!(v >= 32 && v < 9e15)) return false; e->nn_sample = std::max<int64_t>(1024, (int64_t)v) / 32 * 32; return true; }},
    {"nn_last_cells", [](mjpl_engine *e) { return (double)e->nn_last_cells; }, nullptr},  // (read-only: the last look-up took the cell-ordered scan)
    {"kernel_timer", [](mjpl_engine *e) { return (double)e->kt_mode; },
     [](mjpl_engine *e, double v) {
       if (!(v >= 0 && v <= 3)) return false;
       e->kt_mode = (int)v;
       e->kt_used[0] = e->kt_used[1] = 0;
       for (int c = 0; c < 2; c++)  // (the events of a run's first launches are made here, not inside what is being timed)
         while (((int)v >> c & 1) && e->kt_ev[c].size() < 1024) {
           hipEvent_t ev = nullptr;
           if (hipEventCreate(&ev) != hipSuccess) break;
           e->kt_ev[c].push_back(ev);
         }
       return true;
     }},
    {"kernel_timer_launches", [](mjpl_engine *e) { return (double)(e->kt_used[0] / 2); }, nullptr},
    {"kernel_timer2_launches", [](mjpl_engine *e) { return (double)(e->kt_used[1] / 2); }, nullptr},
    // (read-only; synchronise the device: the launches may be on the planner's second stream)
    {"kernel_timer_ms", [](mjpl_engine *e) { return kt_sum(e, 0); }, nullptr},
    {"kernel_timer2_ms", [](mjpl_engine *e) { return kt_sum(e, 1); }, nullptr},
    // (2: the cell-ordered scan counts the pairs that reach their exact evaluation -- "nn_last_exact_pairs"; answers unchanged.
    //  A switch that dropped them unevaluated, for timing, gave wrong answers by design and is not in the product: see round 5.)
    {"nn_probe", [](mjpl_engine *e) { return (double)e->nn_probe; },
     [](mjpl_engine *e, double v) { if (v != 0.0 && v != 2.0) return false; e->nn_probe = (int)v; return true; }},
    {"nn_last_exact_pairs",  // (read-only; synchronises; counted only with "nn_probe" = 2)
     [](mjpl_engine *e) {
       unsigned x[4] = {0, 0, 0, 0};
       if (!e->d_nn16 || hipStreamSynchronize(e->stream) != hipSuccess || hipMemcpy(x, e->d_nn16, sizeof(x), hipMemcpyDeviceToHost) != hipSuccess)
         return -1.0;
       return (double)x[3];
     },
     nullptr},
    // (read-only; synchronises: the share of the tree's sub-chunks a wave of the last cell-ordered scan had on its list, averaged)
    {"nn_last_candidate_fraction",
     [](mjpl_engine *e) {
       if (!e->nn_last_cells || !e->nn_last_count || e->nn_last_waves < 1 || e->nn_last_nsub < 1) return -1.0;
       std::vector<int32_t> c((size_t)e->nn_last_waves);
       if (hipStreamSynchronize(e->stream) != hipSuccess ||
           hipMemcpy(c.data(), e->nn_last_count, c.size() * sizeof(int32_t), hipMemcpyDeviceToHost) != hipSuccess)
         return -1.0;
       double sum = 0;
       for (int32_t v : c) sum += v;
       return sum / ((double)c.size() * e->nn_last_nsub);
     },
     nullptr},
};
#undef MJPL_OPT_INT
#undef MJPL_OPT_BOOL
}  // namespace

int mjpl_set_option(mjpl_engine *e, const char *name, double value) {
  if (!e || !name) return fail(MJPL_E_ARG, "mjpl_set_option: NULL argument");
  for (const EngineOption &o : kEngineOptions)
    if (strcmp(o.name, name) == 0) {
      if (!o.set) return fail(MJPL_E_ARG, "mjpl_set_option: \"%s\" is read-only", name);
      if (!(value == value) || !o.set(e, value)) return fail(MJPL_E_ARG, "mjpl_set_option: value %g out of range for \"%s\"", value, name);
      return MJPL_OK;
    }
  return fail(MJPL_E_ARG, "mjpl_set_option: unknown option \"%s\"", name);
}

// MJPL_DEBUG=1 (and only then): every writable option NAME is also taken from the variable MJPL_<NAME> when an engine is
// made -- for shell tools and A/B scripts (tools/README.md).  The library reads no other switch from the environment.
void apply_debug_environment(mjpl_engine *e) {
  const char *dbg = getenv("MJPL_DEBUG");
  if (!dbg || atoi(dbg) == 0) return;
  for (const EngineOption &o : kEngineOptions) {
    if (!o.set) continue;
    std::string var = "MJPL_";
    for (const char *c = o.name; *c; c++) var += (char)toupper((unsigned char)*c);
    if (const char *v = getenv(var.c_str())) (void)o.set(e, atof(v));
  }
  e->fused_dbg_path = getenv("MJPL_FUSED_DEBUG");  // (a file the fused kernel's per-wave accounting is written to: tools/fused_debug.py)
}

int32_t mjpl_option_count(void) { return (int32_t)(sizeof(kEngineOptions) / sizeof(kEngineOptions[0])); }

const char *mjpl_option_name(int32_t index, int32_t *writable) {
  if (index < 0 || index >= mjpl_option_count()) return nullptr;
  if (writable) *writable = kEngineOptions[index].set ? 1 : 0;
  return kEngineOptions[index].name;
}

int mjpl_set_spec_dir(const char *dir) {
  std::lock_guard<std::mutex> lock(spec_mutex());
  spec_dir_store() = dir ? dir : "";
  return MJPL_OK;
}

int mjpl_get_option(mjpl_engine *e, const char *name, double *value) {
  if (!e || !name || !value) return fail(MJPL_E_ARG, "mjpl_get_option: NULL argument");
  for (const EngineOption &o : kEngineOptions)
    if (strcmp(o.name, name) == 0) { *value = o.get(e); return MJPL_OK; }
  return fail(MJPL_E_ARG, "mjpl_get_option: unknown option \"%s\"", name);
}

int32_t mjpl_nearest_last_screen(mjpl_engine *e) {
  if (!e) return -1;
  if (e->nn_last != 2) return e->nn_last;
  unsigned x[2] = {0, 0};
  if (hipSetDevice(e->device) != hipSuccess || hipStreamSynchronize(e->stream) != hipSuccess ||
      hipMemcpy(x, e->d_nn16, sizeof(x), hipMemcpyDeviceToHost) != hipSuccess)
    return -1;
  return x[1] ? 1 : 2;  // (coordinates too large for binary16: the binary32 screen ran)
}

}  // extern "C"

// the look-up over nodes [0, n) of the slab at `dnodes` (column stride `cap`); outer_d2 (or NULL): a distance every query
// already has an answer at -- nothing farther is of interest
static int nearest_core(mjpl_engine *e, const double *dnodes, int64_t n, int64_t cap, const double *dqueries, int64_t M,
                        int32_t *dout_idx, double *dout_dist2, const double *outer_d2);

// mjpl_nearest_dev over the node range [n0, n) behind an answer (outer_idx, outer_d2: NULL = none) for the nodes below n0
int nearest_range(mjpl_engine *e, const double *dnodes, int64_t n0, int64_t n, int64_t cap, const double *dqueries, int64_t M,
                  int32_t *dout_idx, double *dout_dist2, const int32_t *outer_idx, const double *outer_d2) {
  if (n0 == 0 && !outer_idx) return nearest_core(e, dnodes, n, cap, dqueries, M, dout_idx, dout_dist2, nullptr);
  if (n0 < 0 || n0 > n || (outer_idx && !outer_d2)) return fail(MJPL_E_ARG, "nearest_range: bad range");
  HIP_TRY(hipSetDevice(e->device));
  double *d2 = dout_dist2;
  if (!d2) {
    if ((size_t)M * sizeof(double) > e->nn_tmp_bytes) {
      if (e->d_nn_tmp) HIP_TRY(hipFree(e->d_nn_tmp));
      e->d_nn_tmp = nullptr; e->nn_tmp_bytes = 0;
      HIP_TRY(hipMalloc(&e->d_nn_tmp, (size_t)M * sizeof(double)));
      e->nn_tmp_bytes = (size_t)M * sizeof(double);
    }
    d2 = (double *)e->d_nn_tmp;
  }
  if (n > n0) {
    const int rc = nearest_core(e, dnodes + n0, n - n0, cap, dqueries, M, dout_idx, d2, outer_idx ? outer_d2 : nullptr);
    if (rc != MJPL_OK) return rc;
  }
  hipLaunchKernelGGL(k_nearest_rebase, dim3((unsigned)((M + kBlock - 1) / kBlock)), dim3(kBlock), 0, e->stream, M, n0, n > n0 ? 1 : 0, dout_idx,
                     d2, dout_dist2, outer_idx, outer_d2);
  HIP_TRY(hipGetLastError());
  return MJPL_OK;
}

extern "C" {

int mjpl_nearest_dev(mjpl_engine *e, const double *dnodes, int64_t n, int64_t cap, const double *dqueries,
                     int64_t M, int32_t *dout_idx, double *dout_dist2) {
  if (!e || n < 0 || M < 0 || cap < n) return fail(MJPL_E_ARG, "mjpl_nearest_dev: bad sizes");
  if (M == 0) return MJPL_OK;
  if (!dnodes || !dqueries || !dout_idx) return fail(MJPL_E_ARG, "mjpl_nearest_dev: NULL pointer");
  return nearest_core(e, dnodes, n, cap, dqueries, M, dout_idx, dout_dist2, nullptr);
}

int mjpl_nearest_range_dev(mjpl_engine *e, const double *dnodes, int64_t n0, int64_t n, int64_t cap, const double *dqueries,
                           int64_t M, int32_t *dout_idx, double *dout_dist2, const int32_t *dprev_idx, const double *dprev_dist2) {
  if (!e || n < 0 || n0 < 0 || n0 > n || M < 0 || cap < n) return fail(MJPL_E_ARG, "mjpl_nearest_range_dev: bad sizes");
  if (M == 0) return MJPL_OK;
  if (!dnodes || !dqueries || !dout_idx || ((dprev_idx == nullptr) != (dprev_dist2 == nullptr)))
    return fail(MJPL_E_ARG, "mjpl_nearest_range_dev: NULL pointer");
  return nearest_range(e, dnodes, n0, n, cap, dqueries, M, dout_idx, dout_dist2, dprev_idx, dprev_dist2);
}

}  // extern "C"

static int nearest_core(mjpl_engine *e, const double *dnodes, int64_t n, int64_t cap, const double *dqueries, int64_t M,
                        int32_t *dout_idx, double *dout_dist2, const double *outer_d2) {
  HIP_TRY(hipSetDevice(e->device));
  const int nplan = (int)e->qidx.size();
  if (nplan <= kNNMaxPlan && n > 0) {
    // 2-D decomposition: 512 queries per block x node chunks, then a reduction over the chunks;
    // chunks sized for about eight two-wave blocks per SIMD quartet (the scan is one dependent
    // chain per query: waves, not instructions, hide its latencies)
    const int qblock = kNNQueries * kNNThreads;
    const int64_t qtiles = (M + qblock - 1) / qblock;
    int64_t nchunks = std::max<int64_t>(1, (8192 + qtiles - 1) / qtiles);
    nchunks = std::min<int64_t>(nchunks, (n + kNNThreads - 1) / kNNThreads);
    nchunks = std::max<int64_t>(nchunks, 1);
    int64_t chunk = (n + nchunks - 1) / nchunks;
    chunk = (chunk + kNNThreads - 1) / kNNThreads * kNNThreads;
    nchunks = (n + chunk - 1) / chunk;
    // partial results: room for twice the chunks of the float64 scan (the screened scan below
    // cuts the nodes up to twice as fine), and one seed row
    const size_t maxchunks = 2 * (size_t)nchunks + 1;
    const size_t need = (maxchunks + 1) * (size_t)M * (sizeof(double) + sizeof(int32_t));
    if (need > e->nn_bytes) {
      if (e->d_nn) HIP_TRY(hipFree(e->d_nn));
      e->d_nn = nullptr; e->nn_bytes = 0;
      HIP_TRY(hipMalloc(&e->d_nn, need));
      e->nn_bytes = need;
    }
    double *pd2 = (double *)e->d_nn;
    double *seed_d2 = pd2 + maxchunks * (size_t)M;
    int32_t *pidx = (int32_t *)(seed_d2 + (size_t)M);
    int32_t *seed_idx = pidx + maxchunks * (size_t)M;
    // the float64 scan of nodes [0, nn) in chunks of `ch`
    auto scan64 = [&](int64_t nn, int64_t ch, int64_t nch, int64_t stride = 1) {
      const dim3 grid((unsigned)qtiles, (unsigned)nch);
#define MJPL_NN_CASE(NPV)                                                                                    \
      case NPV:                                                                                              \
        hipLaunchKernelGGL(k_nearest_part<NPV>, grid, dim3(kNNThreads), 0, e->stream, dnodes, nn, cap, dqueries, \
                           M, nplan, ch, pidx, pd2, stride);                                                 \
        break;
      switch (nplan) {
        MJPL_NN_CASE(2) MJPL_NN_CASE(3) MJPL_NN_CASE(4) MJPL_NN_CASE(5) MJPL_NN_CASE(6) MJPL_NN_CASE(7)
        MJPL_NN_CASE(8) MJPL_NN_CASE(9)
        default:
          hipLaunchKernelGGL(k_nearest_part<0>, grid, dim3(kNNThreads), 0, e->stream, dnodes, nn, cap, dqueries, M,
                             nplan, ch, pidx, pd2, stride);
      }
#undef MJPL_NN_CASE
    };
    const unsigned rgridM = (unsigned)((M + kBlock - 1) / kBlock);
    constexpr int64_t kSampleNodes = 16384;
    // (4 096 queries: the planner's look-up of its tail lanes.  A ranged look-up brings a bound for every query with it -- the
    //  distance already found below the range: screened from 8 192 nodes on.  It still takes the sample of its range: the
    //  nodes a round adds reach into places the tree was far from, and a query there would otherwise park every one of them)
    const bool have_bound = outer_d2 != nullptr;
    if (M >= 4096 && (n >= 16 * kSampleNodes || (have_bound && n >= 8192)) && nplan >= 2 && nplan <= 9) {
      // Large trees and query sets.  First a strided sample of the nodes, exactly: every query gets a
      // bound close to its answer.  Then the binary32-screened scan of all nodes, eight queries per
      // lane, which evaluates exactly only what lies within that bound.
      const int64_t stride = std::max<int64_t>(1, n / kSampleNodes);
      // (chunks of the float64 sample scan: whole multiples of the scan's tile, and never more of them than partial rows --
      //  16 384 / nc0 rounded DOWN gave one chunk too many for some node counts, whose row was the seed row: ADVICE r05)
      const int64_t nc0 = std::min<int64_t>(maxchunks, kSampleNodes / kNNThreads);
      const int64_t ch0 = ((kSampleNodes + nc0 - 1) / nc0 + kNNThreads - 1) / kNNThreads * kNNThreads;
      // the screened scan on the matrix cores (nplan <= 7), unless some coordinate is too large for binary16 -- which the
      // device finds out while it packs the operands: then those kernels return at once and the float64 sample scan and
      // the binary32 screen run (which in turn return at once when the matrix cores serve the call)
      // (the matrix-core kernel reads its node tiles through a buffer descriptor with 32-bit offsets: 2^26 nodes at most)
      const bool mfma = e->nn_mfma && nplan <= kNNMMaxPlan && n < ((int64_t)1 << 26);
      e->nn_last = mfma ? 2 : 1;
      unsigned *xbits = nullptr;
      int32_t *mp_idx = nullptr;
      double *mp_d2 = nullptr;
      int mparts = 0;
      const int64_t nsamp64 = std::min<int64_t>(kSampleNodes, n / kNNThreads * kNNThreads);  // (a short range: all of it)
      e->nn_last_cells = 0;
      if (mfma && e->nn_cells && n >= e->nn_cells_min && n >= 64 * kNNCellSub && n < ((int64_t)1 << 25)) {
        // ---- the cell-ordered scan (mjpl_nearest_cells.h): nodes sorted along a space-filling curve, queries by where on it
        // their bound was found, a wave scans the sub-chunks some query of it can have its answer in
        e->nn_last_cells = 1;
        const int64_t npad = (n + kNNCellSub - 1) / kNNCellSub * kNNCellSub, Mpad = (M + kNNMQueries - 1) / kNNMQueries * kNNMQueries;
        const int64_t qblocks = Mpad / kNNMQueries;
        const int nsub = (int)(npad / kNNCellSub), nwords = (nsub + 63) / 64, nsubp = nwords * 64;
        // (sub-chunks per workgroup row: whole mask words, ~2 048 workgroups in all, no more rows than words)
        const int nsplit = (int)std::max<int64_t>(1, std::min<int64_t>((4096 + qblocks - 1) / qblocks, nwords));
        mparts = 2 * nsplit;
        // the arena: what grows with the tree last
        auto al = [](size_t b) { return (b + 255) / 256 * 256; };
        size_t sort_tmp_n = 0, sort_tmp_q = 0;
        const int64_t nres = std::max<int64_t>(npad, std::min<int64_t>((cap + kNNCellSub - 1) / kNNCellSub * kNNCellSub,
                                                                       std::max<int64_t>(4 * npad, e->nn_reserve_nodes)));
        (void)rocprim::radix_sort_pairs(nullptr, sort_tmp_n, (const unsigned *)nullptr, (unsigned *)nullptr, (const int32_t *)nullptr,
                                        (int32_t *)nullptr, (size_t)nres, 0, kNNCKeyBits, e->stream);
        (void)rocprim::radix_sort_pairs(nullptr, sort_tmp_q, (const unsigned *)nullptr, (unsigned *)nullptr, (const int32_t *)nullptr,
                                        (int32_t *)nullptr, (size_t)M, 0, 16, e->stream);
        const size_t b_q16 = al((size_t)Mpad * 64), b_qn = al((size_t)Mpad * 4), b_qs = al((size_t)8 * M * 8), b_b2 = al((size_t)M * 8);
        const size_t b_qf = al((size_t)Mpad * 32), b_k = al((size_t)M * 4), b_pd = al((size_t)mparts * M * 8), b_pi = al((size_t)mparts * M * 4);
        const size_t b_masks = al((size_t)(Mpad / 128) * nwords * 8), b_tmpq = al(sort_tmp_q);
        const size_t b_cnt = al((size_t)(Mpad / 128) * 4);
        const size_t fixed = 1024 + 2 * b_q16 + 2 * b_qn + b_qs + b_b2 + 2 * b_qf + 4 * b_k + b_pd + b_pi + b_tmpq + b_cnt;
        auto grows = [&](int64_t np) {  // masks, lists, boxes, keys + permutation (in / out), sort space, sorted rows, packed rows
          const size_t words = (size_t)((np / kNNCellSub + 63) / 64);
          return al((size_t)(Mpad / 128) * words * 8) + al((size_t)(Mpad / 128) * words * 64 * 4) + al(16 * words * 64 * 4) +
                 4 * al((size_t)np * 4) + al(sort_tmp_n) + al((size_t)8 * np * 8) + 2 * al((size_t)np * 32);
        };
        if (fixed + grows(npad) > e->nn16_bytes) {
          const size_t room = fixed + grows(nres);
          if (e->d_nn16) HIP_TRY(hipFree(e->d_nn16));
          e->d_nn16 = nullptr; e->nn16_bytes = 0;
          HIP_TRY(hipMalloc(&e->d_nn16, room));
          e->nn16_bytes = room;
        }
        char *at = (char *)e->d_nn16;
        auto take = [&](size_t b) { char *p = at; at += b; return p; };
        xbits = (unsigned *)take(256);
        unsigned *mm = (unsigned *)take(256);
        NncPlan *plan = (NncPlan *)take(512);
        uint4 *q16u = (uint4 *)take(b_q16), *q16 = (uint4 *)take(b_q16);
        float *qnu = (float *)take(b_qn), *qn = (float *)take(b_qn);
        double *queries_s = (double *)take(b_qs), *bound2_s = (double *)take(b_b2);
        float *qf = (float *)take(b_qf), *q32c = (float *)take(b_qf);
        unsigned *kq_in = (unsigned *)take(b_k), *kq_out = (unsigned *)take(b_k);
        int32_t *pq_in = (int32_t *)take(b_k), *perm_q = (int32_t *)take(b_k);
        mp_d2 = (double *)take(b_pd);
        mp_idx = (int32_t *)take(b_pi);
        void *tmpq = take(b_tmpq);
        int32_t *ccount = (int32_t *)take(b_cnt);
        unsigned long long *masks = (unsigned long long *)take(b_masks);
        int32_t *clist = (int32_t *)take(al((size_t)(Mpad / 128) * nsubp * 4));
        float *nbox = (float *)take(al((size_t)16 * nsubp * 4));
        unsigned *kn_in = (unsigned *)take(al((size_t)npad * 4)), *kn_out = (unsigned *)take(al((size_t)npad * 4));
        int32_t *pn_in = (int32_t *)take(al((size_t)npad * 4)), *perm_n = (int32_t *)take(al((size_t)npad * 4));
        void *tmpn = take(al(sort_tmp_n));
        double *nodes_s = (double *)take(al((size_t)8 * npad * 8));
        uint4 *nodes16 = (uint4 *)take(al((size_t)npad * 32));
        float *nodes32 = (float *)take(al((size_t)npad * 32));
        // 1. the nodes along the curve
        HIP_TRY(hipMemsetAsync(xbits, 0, 16, e->stream));  // ([3]: option "nn_probe" = 2 counts the exact evaluations here)
        HIP_TRY(hipMemsetAsync(mm, 0xff, 32, e->stream));
        HIP_TRY(hipMemsetAsync(mm + 8, 0, 32, e->stream));
        hipLaunchKernelGGL(k_nnc_minmax, dim3((unsigned)((n + 256 * kNNCMinmaxRows - 1) / (256 * kNNCMinmaxRows))), dim3(256), 0, e->stream, dnodes, n, cap, nplan, mm);
        hipLaunchKernelGGL(k_nnc_plan, dim3(1), dim3(64), 0, e->stream, (const unsigned *)mm, nplan, plan);
        hipLaunchKernelGGL(k_nnc_keys, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, e->stream, dnodes, n, cap, nplan, (const NncPlan *)plan,
                           kn_in, pn_in);
        size_t tn = sort_tmp_n;
        if (rocprim::radix_sort_pairs(tmpn, tn, (const unsigned *)kn_in, kn_out, (const int32_t *)pn_in, perm_n, (size_t)n, 0, kNNCKeyBits,
                                      e->stream) != hipSuccess)
          return fail(MJPL_E_HIP, "nearest: sorting the nodes failed");
        hipLaunchKernelGGL(k_nnc_gather, dim3((unsigned)((npad + 255) / 256)), dim3(256), 0, e->stream, dnodes, cap, (const int32_t *)perm_n, n,
                           npad, nplan, nodes_s, nodes32, (const NncPlan *)plan);
        hipLaunchKernelGGL(k_nnc_boxes, dim3((unsigned)((nsub + 3) / 4)), dim3(256), 0, e->stream, (const double *)nodes_s, n, nplan, nsub,
                           nsubp, nbox);
        hipLaunchKernelGGL(k_nn_pack, dim3((unsigned)((npad + 256 * kNNPackRows - 1) / (256 * kNNPackRows))), dim3(256), 0, e->stream, (const double *)nodes_s, n, (int64_t)1, npad, nplan,
                           0, nodes16, (float *)nullptr, xbits, (int64_t)8, (const float *)plan->ctr);
        hipLaunchKernelGGL(k_nn_pack, dim3((unsigned)((Mpad + 256 * kNNPackRows - 1) / (256 * kNNPackRows))), dim3(256), 0, e->stream, dqueries, M, M, Mpad, nplan, 1, q16u, qnu,
                           xbits, (int64_t)1, (const float *)plan->ctr);
        // 2. every query's bound, and where on the curve it was found (wild coordinates: the float64 scan of a sample of the
        //    callers' rows, as in the plain path; either pair of kernels leaves at once when the other serves the call)
        const int64_t msample = std::max<int64_t>(32, std::min<int64_t>(e->nn_cells_sample, n / 16 / 32 * 32));
        const int64_t mstride = std::max<int64_t>(1, n / msample);
        const int64_t ssplit = std::max<int64_t>(1, std::min<int64_t>((1024 + qblocks - 1) / qblocks, msample / 256));
        const int64_t schunk = ((msample + ssplit - 1) / ssplit + 31) / 32 * 32;
        const dim3 gs((unsigned)qblocks, (unsigned)((msample + schunk - 1) / schunk));
        NnCells sc{};
        sc.pack_idx = 1;
        sc.node_rows = 1;  // (the sorted nodes: rows of eight; the queries still the callers' columns)
        for (sc.idx_shift = 0; (msample >> sc.idx_shift) > 65536; sc.idx_shift++) {}
        hipLaunchKernelGGL(k_nn_fill_inf, dim3(rgridM), dim3(kBlock), 0, e->stream, seed_d2, M, (const unsigned *)xbits);
#define MJPL_NNM_CASE(NPV)                                                                                                           \
        case NPV:                                                                                                                    \
          hipLaunchKernelGGL((k_nearest_mfma<NPV, true>), gs, dim3(kNNMWaves * 64), 0, e->stream, (const double *)nodes_s, msample, npad,  \
                             dqueries, M, (const uint4 *)nodes16, (const uint4 *)q16u, (const float *)qnu, (const unsigned *)xbits,  \
                             schunk, mstride, seed_d2, (int32_t *)nullptr, (double *)nullptr, sc);                                   \
          break;
        switch (nplan) {
          MJPL_NNM_CASE(2) MJPL_NNM_CASE(3) MJPL_NNM_CASE(4) MJPL_NNM_CASE(5) MJPL_NNM_CASE(6) MJPL_NNM_CASE(7)
        }
#undef MJPL_NNM_CASE
        {
          const dim3 grid((unsigned)qtiles, (unsigned)((nsamp64 + ch0 - 1) / ch0));
#define MJPL_NN_CASE(NPV)                                                                                    \
          case NPV:                                                                                          \
            hipLaunchKernelGGL(k_nearest_part<NPV>, grid, dim3(kNNThreads), 0, e->stream, dnodes, nsamp64, cap, dqueries, \
                               M, nplan, ch0, pidx, pd2, stride, (const unsigned *)xbits);                   \
            break;
          switch (nplan) {
            MJPL_NN_CASE(2) MJPL_NN_CASE(3) MJPL_NN_CASE(4) MJPL_NN_CASE(5) MJPL_NN_CASE(6) MJPL_NN_CASE(7)
          }
#undef MJPL_NN_CASE
        }
        hipLaunchKernelGGL(k_nearest_reduce, dim3(rgridM), dim3(kBlock), 0, e->stream, pidx, pd2, M, (int)((nsamp64 + ch0 - 1) / ch0), seed_idx,
                           seed_d2, (const int32_t *)nullptr, (const double *)nullptr, (const unsigned *)xbits);
        if (have_bound) hipLaunchKernelGGL(k_nnc_bound_min, dim3(rgridM), dim3(kBlock), 0, e->stream, seed_d2, outer_d2, M, (const unsigned *)xbits);
        const unsigned rgridM_home = (unsigned)((M + 255) / 256);
        // 3. the queries in scan order
        hipLaunchKernelGGL(k_nnc_query_keys, dim3(rgridM), dim3(kBlock), 0, e->stream, (const double *)seed_d2, M, kq_in, pq_in);
        size_t tq = sort_tmp_q;
        if (rocprim::radix_sort_pairs(tmpq, tq, (const unsigned *)kq_in, kq_out, (const int32_t *)pq_in, perm_q, (size_t)M, 0, 16, e->stream) !=
            hipSuccess)
          return fail(MJPL_E_HIP, "nearest: sorting the queries failed");
        hipLaunchKernelGGL(k_nnc_gather_queries, dim3((unsigned)((Mpad + kBlock - 1) / kBlock)), dim3(kBlock), 0, e->stream, dqueries,
                           (const double *)seed_d2, (const int32_t *)perm_q, M, Mpad, nplan, queries_s, bound2_s, qf, q32c, (const NncPlan *)plan);
        hipLaunchKernelGGL(k_nn_pack, dim3((unsigned)((Mpad + 256 * kNNPackRows - 1) / (256 * kNNPackRows))), dim3(256), 0, e->stream, (const double *)queries_s, M, (int64_t)1, Mpad, nplan,
                           1, q16, qn, xbits, (int64_t)8, (const float *)plan->ctr);
        if (e->nn_home)
          hipLaunchKernelGGL(k_nnc_home, dim3(rgridM_home), dim3(256), 0, e->stream, (const double *)nodes_s, n, (const double *)queries_s, M, nplan,
                           mstride, sc.idx_shift, nsub, bound2_s, qf, (const unsigned *)xbits);
        // 4. candidates, scan
        hipLaunchKernelGGL(k_nn_candidates, dim3((unsigned)(Mpad / 128), (unsigned)((nwords + 3) / 4)), dim3(256), 0, e->stream, (const float *)qf,
                           M, (const float *)nbox, nsub, nsubp, nwords, masks, (const unsigned *)xbits);
        hipLaunchKernelGGL(k_nn_compact, dim3((unsigned)((Mpad / 128 + 3) / 4)), dim3(256), 0, e->stream, (const unsigned long long *)masks, nwords,
                           (int)(Mpad / 128), clist, (int64_t)nsubp, ccount, (const unsigned *)xbits);
        NnCells cs{};
        cs.list = clist; cs.count = ccount; cs.perm_n = perm_n; cs.list_pitch = nsubp; cs.probe = e->nn_probe; cs.counter = xbits + 3; cs.node_rows = 1; cs.query_rows = 1; cs.nodes32 = nodes32; cs.queries32 = q32c; cs.second = e->nn_second_screen;
        e->nn_last_count = ccount; e->nn_last_waves = (int)((M + 127) / 128); e->nn_last_nsub = nsub;
        const dim3 gm((unsigned)(Mpad / 128), (unsigned)nsplit);  // (one wave of 128 queries per workgroup)
        kt_mark(e, 2, e->stream);
#define MJPL_NNM_CASE(NPV)                                                                                                           \
        case NPV:                                                                                                                    \
          hipLaunchKernelGGL((k_nearest_mfma<NPV, false, true>), gm, dim3(64), 0, e->stream, (const double *)nodes_s, n, npad, \
                             (const double *)queries_s, M, (const uint4 *)nodes16, (const uint4 *)q16, (const float *)qn,            \
                             (const unsigned *)xbits, (int64_t)0, (int64_t)1, bound2_s, mp_idx, mp_d2, cs);                          \
          break;
        switch (nplan) {
          MJPL_NNM_CASE(2) MJPL_NNM_CASE(3) MJPL_NNM_CASE(4) MJPL_NNM_CASE(5) MJPL_NNM_CASE(6) MJPL_NNM_CASE(7)
        }
#undef MJPL_NNM_CASE
        kt_mark(e, 2, e->stream);
        // 5. wild coordinates: the binary32 screen over the callers' rows; then the answers in the callers' order
        const int qb32c = kNN32Queries * kNNThreads;
        const int64_t qt32c = (M + qb32c - 1) / qb32c;
        int64_t nc32c = std::max<int64_t>(1, (8192 + qt32c - 1) / qt32c);
        nc32c = std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(nc32c, (int64_t)maxchunks), n / (8 * kNNThreads)));
        int64_t ch32c = (n + nc32c - 1) / nc32c;
        ch32c = (ch32c + kNNThreads - 1) / kNNThreads * kNNThreads;
        nc32c = (n + ch32c - 1) / ch32c;
        const dim3 g32c((unsigned)qt32c, (unsigned)nc32c);
#define MJPL_NN32_CASE(NPV)                                                                                   \
        case NPV:                                                                                             \
          hipLaunchKernelGGL(k_nearest_part32<NPV>, g32c, dim3(kNNThreads), 0, e->stream, dnodes, n, cap, dqueries, \
                             M, ch32c, (const double *)seed_d2, pidx, pd2, (const unsigned *)xbits);          \
          break;
        switch (nplan) {
          MJPL_NN32_CASE(2) MJPL_NN32_CASE(3) MJPL_NN32_CASE(4) MJPL_NN32_CASE(5) MJPL_NN32_CASE(6) MJPL_NN32_CASE(7)
        }
#undef MJPL_NN32_CASE
        hipLaunchKernelGGL(k_nnc_reduce, dim3(rgridM), dim3(kBlock), 0, e->stream, (const int32_t *)mp_idx, (const double *)mp_d2, M, mparts,
                           (const int32_t *)perm_q, dout_idx, dout_dist2, (const unsigned *)xbits, (const int32_t *)pidx, (const double *)pd2,
                           (int)nc32c);
        HIP_TRY(hipGetLastError());
        return MJPL_OK;
      }
      if (!mfma) {
        scan64(nsamp64, ch0, (nsamp64 + ch0 - 1) / ch0, stride);
        hipLaunchKernelGGL(k_nearest_reduce, dim3(rgridM), dim3(kBlock), 0, e->stream, pidx, pd2, M, (int)((nsamp64 + ch0 - 1) / ch0), seed_idx,
                           seed_d2, (const int32_t *)nullptr, (const double *)nullptr, (const unsigned *)nullptr);
        if (have_bound) hipLaunchKernelGGL(k_nearest_bound_min, dim3(rgridM), dim3(kBlock), 0, e->stream, seed_d2, outer_d2, M);
      } else {
        const int64_t npad = (n + 31) / 32 * 32, Mpad = (M + kNNMQueries - 1) / kNNMQueries * kNNMQueries;
        const int64_t qblocks = Mpad / kNNMQueries;
        int64_t nsplit = std::max<int64_t>(1, std::min<int64_t>((2048 + qblocks - 1) / qblocks, n / 4096));
        int64_t chm = ((n + nsplit - 1) / nsplit + 31) / 32 * 32;
        nsplit = (n + chm - 1) / chm;
        mparts = (int)(2 * nsplit);
        const size_t b_nodes = (size_t)npad * 32, b_q = (size_t)Mpad * 64, b_qn = (size_t)Mpad * sizeof(float);
        const size_t b_pd = (size_t)mparts * (size_t)M * sizeof(double), b_pi = (size_t)mparts * (size_t)M * sizeof(int32_t);
        const size_t need16 = 256 + b_nodes + b_q + b_qn + b_pd + b_pi + 32;
        if (need16 > e->nn16_bytes) {
          // (trees grow call by call, and a reallocation stalls the stream: room for four times the nodes, within the slab)
          // (a planner says how far its trees may grow: nn_reserve_nodes -- no reallocation in the middle of a search)
          const size_t room = need16 - b_nodes + (size_t)std::min<int64_t>((cap + 31) / 32 * 32, std::max<int64_t>(4 * npad, e->nn_reserve_nodes)) * 32;
          if (e->d_nn16) HIP_TRY(hipFree(e->d_nn16));
          e->d_nn16 = nullptr; e->nn16_bytes = 0;
          HIP_TRY(hipMalloc(&e->d_nn16, room));
          e->nn16_bytes = room;
        }
        char *at = (char *)e->d_nn16;  // (the node rows last: they are what grows)
        xbits = (unsigned *)at; at += 256;
        uint4 *q16 = (uint4 *)at; at += b_q;
        mp_d2 = (double *)at; at += b_pd;
        float *qn = (float *)at; at += b_qn;
        mp_idx = (int32_t *)at; at += (b_pi + 31) / 32 * 32;
        uint4 *nodes16 = (uint4 *)at;
        HIP_TRY(hipMemsetAsync(xbits, 0, 12, e->stream));  // [0] largest coordinate, [1] wild, [2] largest node norm
        hipLaunchKernelGGL(k_nn_pack, dim3((unsigned)((npad + 256 * kNNPackRows - 1) / (256 * kNNPackRows))), dim3(256), 0, e->stream, dnodes, n, cap, npad, nplan, 0,
                           nodes16, (float *)nullptr, xbits);
        hipLaunchKernelGGL(k_nn_pack, dim3((unsigned)((Mpad + 256 * kNNPackRows - 1) / (256 * kNNPackRows))), dim3(256), 0, e->stream, dqueries, M, M, Mpad, nplan, 1, q16,
                           qn, xbits);
        // every query's bound: the sample node the screen likes best, its distance exactly (wild coordinates: the
        // float64 scan of the sample as before; either pair of kernels leaves at once when the other serves the call)
        const int64_t msample = std::max<int64_t>(32, std::min<int64_t>(e->nn_sample, n / 16 / 32 * 32));
        const int64_t mstride = std::max<int64_t>(1, n / msample);
        // (the sample in as many chunks as give the chip ~1 024 workgroups: a look-up of few queries -- the planner's tail lanes --
        //  would otherwise scan it with eight)
        const int64_t ssplit = std::max<int64_t>(1, std::min<int64_t>((1024 + qblocks - 1) / qblocks, msample / 256));
        const int64_t schunk = ((msample + ssplit - 1) / ssplit + 31) / 32 * 32;
        const dim3 gm((unsigned)qblocks, (unsigned)nsplit), gs((unsigned)qblocks, (unsigned)((msample + schunk - 1) / schunk));
        hipLaunchKernelGGL(k_nn_fill_inf, dim3(rgridM), dim3(kBlock), 0, e->stream, seed_d2, M, (const unsigned *)xbits);
#define MJPL_NNM_CASE(NPV)                                                                                                       \
        case NPV:                                                                                                                \
          hipLaunchKernelGGL((k_nearest_mfma<NPV, true>), gs, dim3(kNNMWaves * 64), 0, e->stream, dnodes, msample, cap, dqueries, M, \
                             (const uint4 *)nodes16, (const uint4 *)q16, (const float *)qn, (const unsigned *)xbits, schunk,     \
                             mstride, seed_d2, (int32_t *)nullptr, (double *)nullptr);                                           \
          break;
        switch (nplan) {
          MJPL_NNM_CASE(2) MJPL_NNM_CASE(3) MJPL_NNM_CASE(4) MJPL_NNM_CASE(5) MJPL_NNM_CASE(6) MJPL_NNM_CASE(7)
        }
#undef MJPL_NNM_CASE
        {
          const dim3 grid((unsigned)qtiles, (unsigned)((nsamp64 + ch0 - 1) / ch0));
#define MJPL_NN_CASE(NPV)                                                                                    \
          case NPV:                                                                                          \
            hipLaunchKernelGGL(k_nearest_part<NPV>, grid, dim3(kNNThreads), 0, e->stream, dnodes, nsamp64, cap, dqueries, \
                               M, nplan, ch0, pidx, pd2, stride, (const unsigned *)xbits);                   \
            break;
          switch (nplan) {
            MJPL_NN_CASE(2) MJPL_NN_CASE(3) MJPL_NN_CASE(4) MJPL_NN_CASE(5) MJPL_NN_CASE(6) MJPL_NN_CASE(7)
          }
#undef MJPL_NN_CASE
        }
        hipLaunchKernelGGL(k_nearest_reduce, dim3(rgridM), dim3(kBlock), 0, e->stream, pidx, pd2, M, (int)((nsamp64 + ch0 - 1) / ch0), seed_idx,
                           seed_d2, (const int32_t *)nullptr, (const double *)nullptr, (const unsigned *)xbits);
        if (have_bound) hipLaunchKernelGGL(k_nearest_bound_min, dim3(rgridM), dim3(kBlock), 0, e->stream, seed_d2, outer_d2, M);
#define MJPL_NNM_CASE(NPV)                                                                                                \
        case NPV:                                                                                                         \
          hipLaunchKernelGGL((k_nearest_mfma<NPV, false>), gm, dim3(kNNMWaves * 64), 0, e->stream, dnodes, n, cap, dqueries, M,    \
                             (const uint4 *)nodes16, (const uint4 *)q16, (const float *)qn, (const unsigned *)xbits, chm, \
                             (int64_t)1, seed_d2, mp_idx, mp_d2);                                                         \
          break;
        switch (nplan) {
          MJPL_NNM_CASE(2) MJPL_NNM_CASE(3) MJPL_NNM_CASE(4) MJPL_NNM_CASE(5) MJPL_NNM_CASE(6) MJPL_NNM_CASE(7)
        }
#undef MJPL_NNM_CASE
      }
      const int qb32 = kNN32Queries * kNNThreads;
      const int64_t qt32 = (M + qb32 - 1) / qb32;
      int64_t nc32 = std::max<int64_t>(1, (8192 + qt32 - 1) / qt32);
      nc32 = std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(nc32, (int64_t)maxchunks), n / (8 * kNNThreads)));
      int64_t ch32 = (n + nc32 - 1) / nc32;
      ch32 = (ch32 + kNNThreads - 1) / kNNThreads * kNNThreads;
      nc32 = (n + ch32 - 1) / ch32;
      const dim3 g32((unsigned)qt32, (unsigned)nc32);
#define MJPL_NN32_CASE(NPV)                                                                                   \
      case NPV:                                                                                               \
        hipLaunchKernelGGL(k_nearest_part32<NPV>, g32, dim3(kNNThreads), 0, e->stream, dnodes, n, cap, dqueries, \
                           M, ch32, (const double *)seed_d2, pidx, pd2, (const unsigned *)xbits);             \
        break;
      switch (nplan) {
        MJPL_NN32_CASE(2) MJPL_NN32_CASE(3) MJPL_NN32_CASE(4) MJPL_NN32_CASE(5) MJPL_NN32_CASE(6) MJPL_NN32_CASE(7)
        MJPL_NN32_CASE(8) MJPL_NN32_CASE(9)
      }
#undef MJPL_NN32_CASE
      if (mfma)
        hipLaunchKernelGGL(k_nearest_reduce_ties, dim3(rgridM), dim3(kBlock), 0, e->stream, (const int32_t *)mp_idx, (const double *)mp_d2, M,
                           mparts, dout_idx, dout_dist2, (const unsigned *)xbits, (const int32_t *)pidx, (const double *)pd2, (int)nc32);
      else
        hipLaunchKernelGGL(k_nearest_reduce, dim3(rgridM), dim3(kBlock), 0, e->stream, pidx, pd2, M, (int)nc32, dout_idx,
                           dout_dist2, (const int32_t *)nullptr, (const double *)nullptr);
      HIP_TRY(hipGetLastError());
      return MJPL_OK;
    }
    e->nn_last = 0;
    scan64(n, chunk, nchunks);
    hipLaunchKernelGGL(k_nearest_reduce, dim3((unsigned)((M + kBlock - 1) / kBlock)), dim3(kBlock), 0, e->stream,
                       pidx, pd2, M, (int)nchunks, dout_idx, dout_dist2);
    HIP_TRY(hipGetLastError());
    return MJPL_OK;
  }
  const size_t lds = 2 * (size_t)nplan * kBlock * sizeof(double);
  int rc = allow_lds(k_nearest, lds);
  if (rc != MJPL_OK) return rc;
  const unsigned grid = (unsigned)((M + kBlock - 1) / kBlock);
  hipLaunchKernelGGL(k_nearest, dim3(grid), dim3(kBlock), lds, e->stream, dnodes, n, cap, dqueries, M, nplan,
                     dout_idx, dout_dist2);
  HIP_TRY(hipGetLastError());
  return MJPL_OK;
}

extern "C" {

// ---- host-buffer

int mjpl_check_configs(mjpl_engine *e, const double *Q, int64_t N, int32_t layout, uint8_t *valid) {
  int rc = check_common(e, Q, N, layout);
  if (rc != MJPL_OK) return rc;
  if (N == 0) return MJPL_OK;
  if (!valid) return fail(MJPL_E_ARG, "NULL output pointer");
  HIP_TRY(hipSetDevice(e->device));
  const size_t qb = (size_t)N * e->qidx.size() * sizeof(double);
  if ((rc = stage_reserve(e, 0, qb)) != MJPL_OK) return rc;
  if ((rc = stage_reserve(e, 2, (size_t)N)) != MJPL_OK) return rc;
  if (qb + (size_t)N <= zero_copy_bytes(e)) {
    if ((rc = pin_reserve(e, qb + (size_t)N)) != MJPL_OK) return rc;
    char *pin = (char *)e->h_pin, *dpin = nullptr;
    HIP_TRY(hipHostGetDevicePointer((void **)&dpin, pin, 0));
    memcpy(pin, Q, qb);
    if ((rc = launch_configs(e, (const double *)dpin, N, layout, (uint8_t *)(dpin + qb), nullptr)) != MJPL_OK) return rc;
    HIP_TRY(hipStreamSynchronize(e->stream));
    memcpy(valid, pin + qb, (size_t)N);
    return MJPL_OK;
  }
  if (qb + (size_t)N <= kFusedHostBytes) {
    // planner-sized batch: through the pinned block, one copy each way
    if ((rc = pin_reserve(e, qb + (size_t)N)) != MJPL_OK) return rc;
    char *pin = (char *)e->h_pin;
    memcpy(pin, Q, qb);
    HIP_TRY(hipMemcpyAsync(e->stage[0], pin, qb, hipMemcpyHostToDevice, e->stream));
    if ((rc = launch_configs(e, (const double *)e->stage[0], N, layout, (uint8_t *)e->stage[2], nullptr)) != MJPL_OK) return rc;
    HIP_TRY(hipMemcpyAsync(pin + qb, e->stage[2], (size_t)N, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    memcpy(valid, pin + qb, (size_t)N);
    return MJPL_OK;
  }
  HIP_TRY(hipMemcpyAsync(e->stage[0], Q, qb, hipMemcpyHostToDevice, e->stream));
  if ((rc = launch_configs(e, (const double *)e->stage[0], N, layout, (uint8_t *)e->stage[2], nullptr)) != MJPL_OK) return rc;
  HIP_TRY(hipMemcpyAsync(valid, e->stage[2], (size_t)N, hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  return MJPL_OK;
}

int mjpl_check_edges(mjpl_engine *e, const double *QA, const double *QB, int64_t E, double step_dist,
                     int32_t layout, int32_t flags, uint8_t *valid, int32_t *first_bad) {
  int rc = check_common(e, QA, E, layout);
  if (rc != MJPL_OK) return rc;
  if (!(step_dist > 0.0)) return fail(MJPL_E_ARG, "`step_dist` must be > 0");
  if (E == 0) return MJPL_OK;
  if (!QB || !valid) return fail(MJPL_E_ARG, "NULL pointer");
  HIP_TRY(hipSetDevice(e->device));
  const size_t qb = (size_t)E * e->qidx.size() * sizeof(double);
  // A handful of LONG edges (path shortcutting, smooth_path: planning/utils.py:9-87) is better served by the two
  // persistent kernels, which keep every 32nd exact waypoint as a checkpoint: the fused kernel rebuilds an undecided
  // waypoint by the recurrence from the start of its edge (measured: 64 edges of 2 400 waypoints 3.9 vs 7.6 ms).  The
  // rows are in host memory here, so a small batch can simply be looked at.
  e->fused_skip_once = false;
  if (E <= 4096) {
    const int64_t np = (int64_t)e->qidx.size();
    double longest2 = 0;
    for (int64_t i = 0; i < E; i++) {
      double s = 0;
      for (int64_t k = 0; k < np; k++) {
        const double d = layout == MJPL_SOA ? QB[k * E + i] - QA[k * E + i] : QB[i * np + k] - QA[i * np + k];
        s += d * d;
      }
      if (s > longest2) longest2 = s;  // (a NaN never compares greater: such an edge is reported by the kernels)
    }
    e->fused_skip_once = std::sqrt(longest2) > 64.0 * step_dist;
  }
  if (2 * qb + 5 * (size_t)E + 16 <= zero_copy_bytes(e)) {
    const size_t vb = ((size_t)E + 7) & ~(size_t)7, fbb = (size_t)E * sizeof(int32_t);
    if ((rc = pin_reserve(e, 2 * qb + vb + fbb + 8)) != MJPL_OK) return rc;
    char *pin = (char *)e->h_pin, *dpin = nullptr;
    HIP_TRY(hipHostGetDevicePointer((void **)&dpin, pin, 0));
    memcpy(pin, QA, qb);
    memcpy(pin + qb, QB, qb);
    HIP_TRY(hipMemsetAsync(e->d_status, 0, sizeof(int), e->stream));
    char *dout = dpin + 2 * qb, *hout = pin + 2 * qb;
    if ((rc = launch_edges(e, (const double *)dpin, (const double *)(dpin + qb), E, step_dist, layout, flags, (uint8_t *)dout,
                           first_bad ? (int32_t *)(dout + vb) : nullptr)) != MJPL_OK)
      return rc;
    HIP_TRY(hipMemcpyAsync(hout + vb + fbb, e->d_status, sizeof(int), hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    memcpy(valid, hout, (size_t)E);
    if (first_bad) memcpy(first_bad, hout + vb, fbb);
    int status = 0;
    memcpy(&status, hout + vb + fbb, sizeof(int));
    if (status & (kStatusTailTimeout | kStatusFusedTimeout)) return fail(MJPL_E_HIP, "a kernel of the edge pipeline gave up waiting (status %d: 2 = the tail kernel for its walking workgroups, 4 = a wave of the fused kernel for its work pool)", status);
    if (status & kStatusNonFinite)
      return fail(MJPL_E_NONFINITE, "an edge holds NaN/inf or needs more than %d waypoints", kMaxWaypoints);
    return MJPL_OK;
  }
  if (2 * qb + 5 * (size_t)E + 16 <= kFusedHostBytes) {
    // planner-sized batch: QA | QB go up in one copy from the pinned block, valid | first_bad |
    // status come back in one
    const size_t vb = ((size_t)E + 7) & ~(size_t)7, fbb = (size_t)E * sizeof(int32_t);
    if ((rc = stage_reserve(e, 0, 2 * qb)) != MJPL_OK) return rc;
    if ((rc = stage_reserve(e, 2, vb + fbb + 8)) != MJPL_OK) return rc;
    if ((rc = pin_reserve(e, 2 * qb + vb + fbb + 8)) != MJPL_OK) return rc;
    char *pin = (char *)e->h_pin;
    memcpy(pin, QA, qb);
    memcpy(pin + qb, QB, qb);
    char *dq = (char *)e->stage[0], *dout = (char *)e->stage[2];
    HIP_TRY(hipMemsetAsync(e->d_status, 0, sizeof(int), e->stream));
    HIP_TRY(hipMemcpyAsync(dq, pin, 2 * qb, hipMemcpyHostToDevice, e->stream));
    if ((rc = launch_edges(e, (const double *)dq, (const double *)(dq + qb), E, step_dist, layout, flags, (uint8_t *)dout,
                           first_bad ? (int32_t *)(dout + vb) : nullptr)) != MJPL_OK)
      return rc;
    HIP_TRY(hipMemcpyAsync(dout + vb + fbb, e->d_status, sizeof(int), hipMemcpyDeviceToDevice, e->stream));
    char *hout = pin + 2 * qb;
    HIP_TRY(hipMemcpyAsync(hout, dout, vb + fbb + 8, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    memcpy(valid, hout, (size_t)E);
    if (first_bad) memcpy(first_bad, hout + vb, fbb);
    int status = 0;
    memcpy(&status, hout + vb + fbb, sizeof(int));
    if (status & (kStatusTailTimeout | kStatusFusedTimeout)) return fail(MJPL_E_HIP, "a kernel of the edge pipeline gave up waiting (status %d: 2 = the tail kernel for its walking workgroups, 4 = a wave of the fused kernel for its work pool)", status);
    if (status & kStatusNonFinite)
      return fail(MJPL_E_NONFINITE, "an edge holds NaN/inf or needs more than %d waypoints", kMaxWaypoints);
    return MJPL_OK;
  }
  if ((rc = stage_reserve(e, 0, qb)) != MJPL_OK) return rc;
  if ((rc = stage_reserve(e, 1, qb)) != MJPL_OK) return rc;
  if ((rc = stage_reserve(e, 2, (size_t)E)) != MJPL_OK) return rc;
  if (first_bad && (rc = stage_reserve(e, 3, (size_t)E * sizeof(int32_t))) != MJPL_OK) return rc;
  HIP_TRY(hipMemsetAsync(e->d_status, 0, sizeof(int), e->stream));
  HIP_TRY(hipMemcpyAsync(e->stage[0], QA, qb, hipMemcpyHostToDevice, e->stream));
  HIP_TRY(hipMemcpyAsync(e->stage[1], QB, qb, hipMemcpyHostToDevice, e->stream));
  if ((rc = launch_edges(e, (const double *)e->stage[0], (const double *)e->stage[1], E, step_dist, layout, flags,
                         (uint8_t *)e->stage[2], first_bad ? (int32_t *)e->stage[3] : nullptr)) != MJPL_OK)
    return rc;
  int status = 0;
  HIP_TRY(hipMemcpyAsync(valid, e->stage[2], (size_t)E, hipMemcpyDeviceToHost, e->stream));
  if (first_bad) HIP_TRY(hipMemcpyAsync(first_bad, e->stage[3], (size_t)E * sizeof(int32_t), hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipMemcpyAsync(&status, e->d_status, sizeof(int), hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  if (status & (kStatusTailTimeout | kStatusFusedTimeout)) return fail(MJPL_E_HIP, "a kernel of the edge pipeline gave up waiting (status %d: 2 = the tail kernel for its walking workgroups, 4 = a wave of the fused kernel for its work pool)", status);
  if (status & kStatusNonFinite)
    return fail(MJPL_E_NONFINITE, "an edge holds NaN/inf or needs more than %d waypoints", kMaxWaypoints);
  return MJPL_OK;
}

int mjpl_fk(mjpl_engine *e, const double *Q, int64_t N, int32_t layout, double *xpos, double *xquat,
            double *geom_xpos, double *geom_xmat) {
  int rc = check_common(e, Q, N, layout);
  if (rc != MJPL_OK) return rc;
  if (N == 0) return MJPL_OK;
  HIP_TRY(hipSetDevice(e->device));
  const HostModel &m = e->m;
  const size_t qb = (size_t)N * e->qidx.size() * sizeof(double);
  const size_t sz[4] = {(size_t)N * m.nbody * 3 * 8, (size_t)N * m.nbody * 4 * 8, (size_t)N * m.ngeom * 3 * 8,
                        (size_t)N * m.ngeom * 9 * 8};
  double *host[4] = {xpos, xquat, geom_xpos, geom_xmat};
  if ((rc = stage_reserve(e, 0, qb)) != MJPL_OK) return rc;
  for (int k = 0; k < 4; k++)
    if (host[k] && (rc = stage_reserve(e, 2 + k, sz[k])) != MJPL_OK) return rc;
  HIP_TRY(hipMemcpyAsync(e->stage[0], Q, qb, hipMemcpyHostToDevice, e->stream));
  FkOut out;
  out.xpos = xpos ? (double *)e->stage[2] : nullptr;
  out.xquat = xquat ? (double *)e->stage[3] : nullptr;
  out.geom_xpos = geom_xpos ? (double *)e->stage[4] : nullptr;
  out.geom_xmat = geom_xmat ? (double *)e->stage[5] : nullptr;
  out.nbody = m.nbody;
  out.ngeom = m.ngeom;
  const size_t lds = lds_bytes(e, 1);
  const unsigned grid = (unsigned)((N + kBlock - 1) / kBlock);
  rc = allow_lds(k_fk, lds);
  if (rc == MJPL_OK)
    hipLaunchKernelGGL(k_fk, dim3(grid), dim3(kBlock), lds, e->stream, e->d_ip, (int)e->ip.size(), e->d_dp,
                       (int)e->dp.size(), (const double *)e->stage[0], N, layout, out);
  if (rc != MJPL_OK) return rc;
  HIP_TRY(hipGetLastError());
  for (int k = 0; k < 4; k++)
    if (host[k]) HIP_TRY(hipMemcpyAsync(host[k], e->stage[2 + k], sz[k], hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  // bodies/geoms welded to the world do not depend on qpos: poses folded at create
  for (int64_t i = 0; i < N; i++) {
    for (int b = 0; b < m.nbody; b++) {
      if (!e->body_static[b]) continue;
      if (xpos) memcpy(xpos + (i * m.nbody + b) * 3, &e->st_xpos[3 * b], 3 * sizeof(double));
      if (xquat) memcpy(xquat + (i * m.nbody + b) * 4, &e->st_xquat[4 * b], 4 * sizeof(double));
    }
    for (int g = 0; g < m.ngeom; g++) {
      if (!e->geom_static[g]) continue;
      if (geom_xpos) memcpy(geom_xpos + (i * m.ngeom + g) * 3, &e->st_gxpos[3 * g], 3 * sizeof(double));
      if (geom_xmat) memcpy(geom_xmat + (i * m.ngeom + g) * 9, &e->st_gxmat[9 * g], 9 * sizeof(double));
    }
  }
  return MJPL_OK;
}

// ---- memory / stream helpers

int mjpl_dev_alloc(mjpl_engine *e, size_t bytes, void **out) {
  if (!e || !out) return fail(MJPL_E_ARG, "mjpl_dev_alloc: NULL argument");
  HIP_TRY(hipSetDevice(e->device));
  HIP_TRY(hipMalloc(out, bytes ? bytes : 1));
  return MJPL_OK;
}

int mjpl_dev_free(mjpl_engine *e, void *p) {
  if (!e) return fail(MJPL_E_ARG, "mjpl_dev_free: NULL engine");
  HIP_TRY(hipSetDevice(e->device));
  if (p) HIP_TRY(hipFree(p));
  return MJPL_OK;
}

int mjpl_h2d(mjpl_engine *e, void *dst, const void *src, size_t bytes) {
  if (!e || (bytes && (!dst || !src))) return fail(MJPL_E_ARG, "mjpl_h2d: NULL argument");
  HIP_TRY(hipSetDevice(e->device));
  HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, e->stream));
  return MJPL_OK;
}

int mjpl_d2h(mjpl_engine *e, void *dst, const void *src, size_t bytes) {
  if (!e || (bytes && (!dst || !src))) return fail(MJPL_E_ARG, "mjpl_d2h: NULL argument");
  HIP_TRY(hipSetDevice(e->device));
  HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, e->stream));
  return MJPL_OK;
}

int mjpl_sync(mjpl_engine *e) {
  if (!e) return fail(MJPL_E_ARG, "mjpl_sync: NULL engine");
  HIP_TRY(hipSetDevice(e->device));
  HIP_TRY(hipStreamSynchronize(e->stream));
  return MJPL_OK;
}

void *mjpl_stream(mjpl_engine *e) { return e ? (void *)e->stream : nullptr; }

// ---- measurement: per-launch HIP-event timing on the engine's own stream

int mjpl_time_edges_stages_dev(mjpl_engine *e, const double *dQA, const double *dQB, int64_t E, double step_dist,
                               int32_t layout, uint8_t *dvalid, int32_t iters, int32_t sample_every,
                               float *ms_mean, float *stage_ms, int32_t *nsamples) {
  if (!e || iters < 0 || (iters > 0 && !ms_mean)) return fail(MJPL_E_ARG, "mjpl_time_edges_stages_dev: bad argument");
  HIP_TRY(hipSetDevice(e->device));
  if (sample_every < 1) sample_every = 1;
  // Two events around the whole run (the mean step) and, on every `sample_every`-th step, one
  // event after each stage of the launch: bracketing every kernel of every step would cost a few
  // percent of the throughput the run is there to measure.
  // (sample_every > iters: NO launch is bracketed -- a bracketed launch costs ~56 us more than a plain one, 1.4 % of a
  //  20-step region: bench.py takes the per-kernel durations of such a region from launches outside it)
  const bool none = sample_every > iters;
  const int nsamp = (iters > 0 && !none) ? (iters + sample_every - 1) / sample_every : 0;
  constexpr int NM = MJPL_NSTAGES + 1;
  std::vector<hipEvent_t> ev((size_t)nsamp * NM + 2);
  for (auto &x : ev) HIP_TRY(hipEventCreate(&x));
  hipEvent_t region_start = ev[(size_t)nsamp * NM], region_end = ev[(size_t)nsamp * NM + 1];
  int rc = MJPL_OK;
  HIP_TRY(hipEventRecord(region_start, e->stream));
  for (int k = 0; k < iters && rc == MJPL_OK; k++) {
    e->marks = (!none && k % sample_every == 0) ? &ev[(size_t)(k / sample_every) * NM] : nullptr;
    rc = mjpl_check_edges_dev(e, dQA, dQB, E, step_dist, layout, 0, dvalid, nullptr);
    e->marks = nullptr;
  }
  HIP_TRY(hipEventRecord(region_end, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  if (rc == MJPL_OK && iters > 0) {
    float total = 0;
    HIP_TRY(hipEventElapsedTime(&total, region_start, region_end));
    *ms_mean = total / (float)iters;
    if (stage_ms && nsamp > 0) {
      for (int st = 0; st < MJPL_NSTAGES; st++) {
        double acc = 0;
        for (int sidx = 0; sidx < nsamp; sidx++) {
          float t = 0;
          HIP_TRY(hipEventElapsedTime(&t, ev[(size_t)sidx * NM + st], ev[(size_t)sidx * NM + st + 1]));
          acc += t;
        }
        stage_ms[st] = (float)(acc / nsamp);
      }
    }
    if (nsamples) *nsamples = nsamp;
  }
  for (auto &x : ev) (void)hipEventDestroy(x);
  return rc;
}

int mjpl_time_edges_dev(mjpl_engine *e, const double *dQA, const double *dQB, int64_t E, double step_dist,
                        int32_t layout, uint8_t *dvalid, int32_t iters, float *ms, float *ms_first) {
  if (!e || iters < 0 || (iters > 0 && !ms)) return fail(MJPL_E_ARG, "mjpl_time_edges_dev: bad argument");
  float mean = 0, stage[MJPL_NSTAGES] = {0};
  const int rc = mjpl_time_edges_stages_dev(e, dQA, dQB, E, step_dist, layout, dvalid, iters, 1, &mean,
                                            ms_first ? stage : nullptr, nullptr);
  if (rc != MJPL_OK) return rc;
  const int main_stage = (e->filter && e->filter_usable) ? MJPL_STAGE_ITEMS : MJPL_STAGE_EXACT;
  for (int k = 0; k < iters; k++) {
    ms[k] = mean;
    if (ms_first) ms_first[k] = stage[main_stage];
  }
  return MJPL_OK;
}

int mjpl_time_configs_dev(mjpl_engine *e, const double *dQ, int64_t N, int32_t layout, uint8_t *dvalid,
                          int32_t iters, float *ms) {
  if (!e || iters < 0 || (iters > 0 && !ms)) return fail(MJPL_E_ARG, "mjpl_time_configs_dev: bad argument");
  HIP_TRY(hipSetDevice(e->device));
  std::vector<hipEvent_t> ev(2 * (size_t)iters);
  for (auto &x : ev) HIP_TRY(hipEventCreate(&x));
  int rc = MJPL_OK;
  for (int k = 0; k < iters && rc == MJPL_OK; k++) {
    HIP_TRY(hipEventRecord(ev[2 * k], e->stream));
    rc = mjpl_check_configs_dev(e, dQ, N, layout, dvalid);
    HIP_TRY(hipEventRecord(ev[2 * k + 1], e->stream));
  }
  HIP_TRY(hipStreamSynchronize(e->stream));
  for (int k = 0; k < iters && rc == MJPL_OK; k++) HIP_TRY(hipEventElapsedTime(&ms[k], ev[2 * k], ev[2 * k + 1]));
  for (auto &x : ev) (void)hipEventDestroy(x);
  return rc;
}

// ---- row f1: PoseConstraint ---------------------------------------------------------------

struct mjpl_pose {
  mjpl_engine *e = nullptr;
  std::vector<int> pi;
  std::vector<double> pd;
  int *d_pi = nullptr;
  double *d_pd = nullptr;
  int nq = 0, nj = 0;
  uint64_t chain_hash = 0;  // of the chain program (bodies, joints, their constants): names a generated projection
  bool spec_off = false;    // MJPL_POSE_SPEC=0 at creation: the interpreting kernels whatever the engine has loaded
};

namespace {
size_t pose_lds(const mjpl_pose *p) { return (size_t)kPoseBlock * sizeof(double) * ((size_t)p->nq + 6 * (size_t)p->nj); }

int pose_check(const mjpl_pose *p, const void *a, int64_t n) {
  if (!p) return fail(MJPL_E_ARG, "pose handle is NULL");
  if (n < 0) return fail(MJPL_E_ARG, "negative batch size");
  if (n > 0 && !a) return fail(MJPL_E_ARG, "NULL batch pointer");
  return MJPL_OK;
}
}  // namespace

namespace {
// chain program shared by the pose and IK handles: per body {njnt}, per joint {type, qadr, jid}
int build_chain(const HostModel &m, int site_body, std::vector<int> &pi, std::vector<double> &pd, int *nj) {
  std::vector<int> chain;
  for (int b = site_body; b > 0; b = m.body_parentid[b]) chain.push_back(b);
  std::reverse(chain.begin(), chain.end());
  pi.assign(PH_SIZE, 0);
  *nj = 0;
  for (int b : chain) {
    pi.push_back(m.body_jntnum[b]);
    for (int k = 0; k < 3; k++) pd.push_back(m.body_pos[3 * b + k]);
    for (int k = 0; k < 4; k++) pd.push_back(m.body_quat[4 * b + k]);
    for (int j = 0; j < m.body_jntnum[b]; j++) {
      const int jid = m.body_jntadr[b] + j;
      if (m.jnt_type[jid] != JT_SLIDE && m.jnt_type[jid] != JT_HINGE)
        return fail(MJPL_E_JOINT, "joint %d: only slide and hinge joints are supported", jid);
      pi.push_back(m.jnt_type[jid]);
      pi.push_back(m.jnt_qposadr[jid]);
      pi.push_back(jid);
      for (int k = 0; k < 3; k++) pd.push_back(m.jnt_axis[3 * jid + k]);
      for (int k = 0; k < 3; k++) pd.push_back(m.jnt_pos[3 * jid + k]);
      pd.push_back(m.qpos0[m.jnt_qposadr[jid]]);
      (*nj)++;
    }
  }
  pi[PH_NBODY] = (int)chain.size();
  pi[PH_NJOINT] = *nj;
  pi[PH_NQ] = m.nq;
  return MJPL_OK;
}
}  // namespace

namespace {
// what a generated projection carries as literals: the chain program without its run-time tail (site offset,
// constraint, tolerances, iteration bound), the library ABI and the digest of the shared headers
uint64_t chain_hash_of(const std::vector<int> &pi, const std::vector<double> &pd, size_t chain_doubles) {
  uint64_t h = 0xcbf29ce484222325ull;
  auto mix = [&](const void *ptr, size_t n) {
    const unsigned char *b = (const unsigned char *)ptr;
    for (size_t k = 0; k < n; k++) { h ^= b[k]; h *= 0x100000001b3ull; }
  };
  const int head[3] = {pi[PH_NBODY], pi[PH_NJOINT], pi[PH_NQ]};
  mix(head, sizeof(head));
  mix(pi.data() + PH_SIZE, (pi.size() - PH_SIZE) * sizeof(int));
  mix(pd.data(), chain_doubles * sizeof(double));
  const int abi = MJPL_SPEC_ABI;
  mix(&abi, sizeof(abi));
  const unsigned long long stamp = MJPL_SRC_STAMP;
  mix(&stamp, sizeof(stamp));
  return h;
}

// How a batch of N rows (projections, IK seeds, active planner lanes) goes to the row kernels of mjpl_rows.h.  The
// kernels hold one wave per SIMD (their registers), 1 024 waves on the chip.  While ALL rows can be resident at once
// a row gets as many lanes as that allows -- eight up to 8 192 rows, four up to 16 384 (the lanes split a third of a
// Newton step's instructions) -- and every row starts with the launch: such batches are bound by their slowest row,
// which must not wait for a free lane first.  Beyond that one lane per row, and a wave owns ceil(N / 1 024) >= 64 rows:
// it refills as rows end.  MJPL_ROWS_G: force the lanes per row (A/B timing).
struct RowsShape { int G; unsigned grid; int64_t per; };
RowsShape rows_shape(int64_t N, int forced = 0) {  // (forced: option "rows_g" -- tests compare the three kernels on one batch)
  const int64_t max_waves = 1024;  // one per SIMD: what the row kernels' registers allow (mjpl_rows.h)
  RowsShape r;
  r.G = N <= 8 * max_waves ? 8 : (N <= 16 * max_waves ? 4 : 1);
  if (forced == 1 || forced == 4 || forced == 8) r.G = forced;
  const int64_t rows_per_wave = 64 / r.G;
  const int64_t waves = std::min<int64_t>(max_waves, (N + rows_per_wave - 1) / rows_per_wave);
  r.per = (N + waves - 1) / waves;
  r.grid = (unsigned)((N + r.per - 1) / r.per);
  return r;
}

// the generated projection of this handle's chain in the library its engine has loaded NOW (set_planning may have
// changed it since the handle was made), or -1
int pose_spec_index(const mjpl_pose *p) {
  const SpecLib *sl = p->e->spec;
  if (!sl || p->spec_off) return -1;
  for (int k = 0; k < sl->pose_count; k++)
    if (sl->pose_hash(k) == p->chain_hash) return k;
  return -1;
}
}  // namespace

// Host only, no device needed: the chain program of (model, site body) -- what mjpl_pose_create compiles and
// mjpl_amd/specialise.py turns into straight-line code -- and its hash.  Call with pi = pd = NULL for the sizes.
int mjpl_pose_chain_dump(const mjpl_model_desc *d, int32_t site_body, int32_t *pi, int32_t *npi, double *pd, int32_t *npd,
                         uint64_t *hash) {
  if (!d || !npi || !npd || !hash) return fail(MJPL_E_ARG, "mjpl_pose_chain_dump: NULL argument");
  if (d->nq != d->njnt) return fail(MJPL_E_JOINT, "nq != njnt: only 1-DoF joints are supported");
  std::unique_ptr<mjpl_engine> e(new mjpl_engine());
  e->device = -1;
  int rc = engine_from_desc(e.get(), d, nullptr, 0);
  if (rc != MJPL_OK) return rc;
  if (site_body < 0 || site_body >= e->m.nbody) return fail(MJPL_E_ARG, "site body %d out of range", site_body);
  std::vector<int> vi;
  std::vector<double> vd;
  int nj = 0;
  if ((rc = build_chain(e->m, site_body, vi, vd, &nj)) != MJPL_OK) return rc;
  *hash = chain_hash_of(vi, vd, vd.size());
  const bool fits = pi && pd && *npi >= (int32_t)vi.size() && *npd >= (int32_t)vd.size();
  *npi = (int32_t)vi.size();
  *npd = (int32_t)vd.size();
  if (fits) {
    memcpy(pi, vi.data(), vi.size() * sizeof(int));
    memcpy(pd, vd.data(), vd.size() * sizeof(double));
  }
  return MJPL_OK;
}

// 0: the interpreting kernels serve this handle; 1: a generated projection of the engine's library does
int mjpl_pose_spec_loaded(mjpl_pose *p) { return (p && pose_spec_index(p) >= 0) ? 1 : 0; }

int mjpl_pose_create(mjpl_engine *e, const mjpl_pose_desc *d, mjpl_pose **out) {
  if (!e || !d || !out) return fail(MJPL_E_ARG, "mjpl_pose_create: NULL argument");
  const HostModel &m = e->m;
  if (d->site_body < 0 || d->site_body >= m.nbody) return fail(MJPL_E_ARG, "site body %d out of range", d->site_body);
  if (!d->jnt_range) return fail(MJPL_E_ARG, "jnt_range is NULL");
  if (d->tolerance < 0.0) return fail(MJPL_E_ARG, "`tolerance` must be >= 0.");
  if (!(d->q_step > 0.0)) return fail(MJPL_E_ARG, "`q_step` must be > 0.");
  if (m.nq != m.njnt) return fail(MJPL_E_JOINT, "pose projection needs 1-DoF joints only (nq %d != njnt %d)", m.nq, m.njnt);
  auto p = std::make_unique<mjpl_pose>();
  p->e = e;
  p->nq = m.nq;
  int rc = build_chain(m, d->site_body, p->pi, p->pd, &p->nj);
  if (rc != MJPL_OK) return rc;
  p->chain_hash = chain_hash_of(p->pi, p->pd, p->pd.size());
  p->spec_off = e->pose_spec == 0;  // (option "pose_spec")
  p->pi[PH_MAXIT] = d->max_iters > 0 ? d->max_iters : 1000;
  p->pi[PH_OFF_TAIL] = (int)p->pd.size();
  p->pd.resize(p->pd.size() + PT_SIZE);
  double *t = p->pd.data() + p->pi[PH_OFF_TAIL];
  for (int k = 0; k < 3; k++) { t[PT_SITE_POS + k] = d->site_pos[k]; t[PT_C_POS + k] = d->c_pos[k]; }
  for (int k = 0; k < 4; k++) { t[PT_SITE_QUAT + k] = d->site_quat[k]; t[PT_C_QUAT + k] = d->c_quat[k]; }
  for (int k = 0; k < 6; k++) { t[PT_LO + k] = d->lo[k]; t[PT_HI + k] = d->hi[k]; }
  t[PT_TOL] = d->tolerance;
  t[PT_QSTEP] = d->q_step;
  p->pi[PH_OFF_JRANGE] = (int)p->pd.size();
  for (int k = 0; k < 2 * m.njnt; k++) p->pd.push_back(d->jnt_range[k]);
  if (pose_lds(p.get()) > 64 * 1024)
    return fail(MJPL_E_CAPACITY, "pose projection: %d qpos + %d chain joints exceed the LDS budget", p->nq, p->nj);
  HIP_TRY(hipSetDevice(e->device));
  HIP_TRY(hipMalloc(&p->d_pi, p->pi.size() * sizeof(int)));
  HIP_TRY(hipMalloc(&p->d_pd, p->pd.size() * sizeof(double)));
  HIP_TRY(hipMemcpy(p->d_pi, p->pi.data(), p->pi.size() * sizeof(int), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(p->d_pd, p->pd.data(), p->pd.size() * sizeof(double), hipMemcpyHostToDevice));
  *out = p.release();
  return MJPL_OK;
}

void mjpl_pose_destroy(mjpl_pose *p) {
  if (!p) return;
  (void)hipSetDevice(p->e->device);
  if (p->d_pi) (void)hipFree(p->d_pi);
  if (p->d_pd) (void)hipFree(p->d_pd);
  delete p;
}

int mjpl_pose_set_q_step(mjpl_pose *p, double q_step) {
  if (!p) return fail(MJPL_E_ARG, "pose handle is NULL");
  if (!(q_step > 0.0)) return fail(MJPL_E_ARG, "`q_step` must be > 0.");
  const size_t at = (size_t)p->pi[PH_OFF_TAIL] + PT_QSTEP;
  p->pd[at] = q_step;
  HIP_TRY(hipSetDevice(p->e->device));
  HIP_TRY(hipStreamSynchronize(p->e->stream));
  HIP_TRY(hipMemcpy(p->d_pd + at, &p->pd[at], sizeof(double), hipMemcpyHostToDevice));
  return MJPL_OK;
}

int mjpl_pose_apply_dev(mjpl_pose *p, const double *dQold, const double *dQ, int64_t N, double *dQout,
                        uint8_t *dok, int32_t *diters) {
  int rc = pose_check(p, dQ, N);
  if (rc != MJPL_OK) return rc;
  if (N == 0) return MJPL_OK;
  if (!dQold || !dQout || !dok) return fail(MJPL_E_ARG, "NULL pointer");
  HIP_TRY(hipSetDevice(p->e->device));
  KtScope kt_scope(p->e, 1);  // (option "kernel_timer")
  const unsigned grid = (unsigned)((N + kPoseBlock - 1) / kPoseBlock);
  const int k = pose_spec_index(p);
  if (k >= 0) {
    mjpl_engine *e = p->e;
    const RowsShape rs = rows_shape(N, e->rows_g);
    // MJPL_POSE_PHASE_STEPS=k (an experiment, off by default): a batch too large for several lanes per row in two
    // launches (mjpl_rows.h: PosePhase) -- every row's first k Newton steps with one lane per row, then the rows that
    // want more, a list the first launch packs, with eight.  Same bytes; measured SLOWER at 131 072 rows (0.198 ms in
    // one launch; k = 2 / 3 / 4 / 6: 0.352 / 0.283 / 0.247 / 0.206 -- profiles/r05h_pose_phase.json): eight lanes per row
    // cost eight times the lanes for 1.6 times the speed, and 40 % of the rows want more than three steps.
    const int phase_steps = e->pose_phase_steps;  // (option "pose_phase_steps")
    if (rs.G == 1 && phase_steps > 0) {
      const size_t need = ((size_t)N * 2 + 16) * sizeof(int32_t);
      if ((rc = stage_reserve(e, 6, need)) != MJPL_OK) return rc;
      int32_t *list = (int32_t *)e->stage[6], *itst = list + N;
      int *count = (int *)(itst + N);
      HIP_TRY(hipMemsetAsync(count, 0, sizeof(int), e->stream));
      const PosePhase first = {phase_steps, nullptr, nullptr, list, count, itst};
      int rc2 = e->spec->pose_apply(k, 1, e->stream, rs.grid, p->d_pi, p->d_pd, dQold, dQ, N, rs.per, dQout, dok, diters, first);
      if (rc2 == 0) {
        const PosePhase second = {0, list, count, nullptr, nullptr, itst};
        rc2 = e->spec->pose_apply(k, 8, e->stream, 1024u, p->d_pi, p->d_pd, dQold, dQ, N, 0, dQout, dok, diters, second);
        if (rc2 == 0) return MJPL_OK;
        return fail(MJPL_E_HIP, "generated projection kernel (second launch) failed to launch");
      }
      if (rc2 != -1) return fail(MJPL_E_HIP, "generated projection kernel failed to launch");
    } else {
      const PosePhase whole = {0, nullptr, nullptr, nullptr, nullptr, nullptr};
      const int rc2 = e->spec->pose_apply(k, rs.G, e->stream, rs.grid, p->d_pi, p->d_pd, dQold, dQ, N, rs.per, dQout, dok, diters, whole);
      if (rc2 == 0) return MJPL_OK;
      if (rc2 != -1) return fail(MJPL_E_HIP, "generated projection kernel failed to launch");
    }
  }
  hipLaunchKernelGGL(k_pose_apply, dim3(grid), dim3(kPoseBlock), pose_lds(p), p->e->stream, p->d_pi, p->d_pd,
                     dQold, dQ, N, dQout, dok, diters);
  HIP_TRY(hipGetLastError());
  return MJPL_OK;
}

int mjpl_pose_valid_dev(mjpl_pose *p, const double *dQ, int64_t N, uint8_t *dvalid, double *dxpos, double *dxmat) {
  int rc = pose_check(p, dQ, N);
  if (rc != MJPL_OK) return rc;
  if (N == 0) return MJPL_OK;
  HIP_TRY(hipSetDevice(p->e->device));
  const unsigned grid = (unsigned)((N + kPoseBlock - 1) / kPoseBlock);
  hipLaunchKernelGGL(k_pose_valid, dim3(grid), dim3(kPoseBlock), (size_t)kPoseBlock * sizeof(double) * p->nq,
                     p->e->stream, p->d_pi, p->d_pd, dQ, N, dvalid, dxpos, dxmat);
  HIP_TRY(hipGetLastError());
  return MJPL_OK;
}

int mjpl_pose_apply(mjpl_pose *p, const double *Q_old, const double *Q, int64_t N, double *Q_out,
                    uint8_t *ok, int32_t *iters) {
  int rc = pose_check(p, Q, N);
  if (rc != MJPL_OK) return rc;
  if (N == 0) return MJPL_OK;
  if (!Q_old || !Q_out || !ok) return fail(MJPL_E_ARG, "NULL pointer");
  mjpl_engine *e = p->e;
  HIP_TRY(hipSetDevice(e->device));
  const size_t qb = (size_t)N * p->nq * sizeof(double);
  if ((rc = stage_reserve(e, 0, qb)) != MJPL_OK) return rc;
  if ((rc = stage_reserve(e, 1, qb)) != MJPL_OK) return rc;
  if ((rc = stage_reserve(e, 4, qb)) != MJPL_OK) return rc;
  if ((rc = stage_reserve(e, 2, (size_t)N)) != MJPL_OK) return rc;
  if ((rc = stage_reserve(e, 3, (size_t)N * sizeof(int32_t))) != MJPL_OK) return rc;
  HIP_TRY(hipMemcpyAsync(e->stage[0], Q_old, qb, hipMemcpyHostToDevice, e->stream));
  HIP_TRY(hipMemcpyAsync(e->stage[1], Q, qb, hipMemcpyHostToDevice, e->stream));
  if ((rc = mjpl_pose_apply_dev(p, (const double *)e->stage[0], (const double *)e->stage[1], N,
                                (double *)e->stage[4], (uint8_t *)e->stage[2], (int32_t *)e->stage[3])) != MJPL_OK)
    return rc;
  HIP_TRY(hipMemcpyAsync(Q_out, e->stage[4], qb, hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipMemcpyAsync(ok, e->stage[2], (size_t)N, hipMemcpyDeviceToHost, e->stream));
  if (iters) HIP_TRY(hipMemcpyAsync(iters, e->stage[3], (size_t)N * sizeof(int32_t), hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  return MJPL_OK;
}

int mjpl_pose_valid(mjpl_pose *p, const double *Q, int64_t N, uint8_t *valid, double *xpos, double *xmat) {
  int rc = pose_check(p, Q, N);
  if (rc != MJPL_OK) return rc;
  if (N == 0) return MJPL_OK;
  mjpl_engine *e = p->e;
  HIP_TRY(hipSetDevice(e->device));
  const size_t qb = (size_t)N * p->nq * sizeof(double);
  if ((rc = stage_reserve(e, 0, qb)) != MJPL_OK) return rc;
  if ((rc = stage_reserve(e, 2, (size_t)N)) != MJPL_OK) return rc;
  if ((rc = stage_reserve(e, 4, (size_t)N * 3 * sizeof(double))) != MJPL_OK) return rc;
  if ((rc = stage_reserve(e, 5, (size_t)N * 9 * sizeof(double))) != MJPL_OK) return rc;
  HIP_TRY(hipMemcpyAsync(e->stage[0], Q, qb, hipMemcpyHostToDevice, e->stream));
  if ((rc = mjpl_pose_valid_dev(p, (const double *)e->stage[0], N, (uint8_t *)e->stage[2], (double *)e->stage[4],
                                (double *)e->stage[5])) != MJPL_OK)
    return rc;
  if (valid) HIP_TRY(hipMemcpyAsync(valid, e->stage[2], (size_t)N, hipMemcpyDeviceToHost, e->stream));
  if (xpos) HIP_TRY(hipMemcpyAsync(xpos, e->stage[4], (size_t)N * 3 * sizeof(double), hipMemcpyDeviceToHost, e->stream));
  if (xmat) HIP_TRY(hipMemcpyAsync(xmat, e->stage[5], (size_t)N * 9 * sizeof(double), hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  return MJPL_OK;
}

// ---- row f3: IK seeds ------------------------------------------------------------------------


int mjpl_ik_solve_dev(mjpl_engine *e, const mjpl_ik_desc *d, const double *dQ, int64_t N, double *dQout,
                      uint8_t *dok, int32_t *diters, double *derr) {
  if (!e || !d) return fail(MJPL_E_ARG, "mjpl_ik_solve: NULL argument");
  if (N < 0) return fail(MJPL_E_ARG, "negative batch size");
  if (N == 0) return MJPL_OK;
  if (!dQ || !dQout || !dok) return fail(MJPL_E_ARG, "NULL pointer");
  const HostModel &m = e->m;
  if (d->site_body < 0 || d->site_body >= m.nbody) return fail(MJPL_E_ARG, "site body %d out of range", d->site_body);
  if (!d->jnt_range || !d->movable) return fail(MJPL_E_ARG, "jnt_range / movable is NULL");
  if (d->iterations < 1) return fail(MJPL_E_ARG, "`iterations` must be > 0.");
  if (m.nq != m.njnt) return fail(MJPL_E_JOINT, "IK needs 1-DoF joints only (nq %d != njnt %d)", m.nq, m.njnt);
  std::vector<int> pi;
  std::vector<double> pd;
  int nj = 0;
  int rc = build_chain(m, d->site_body, pi, pd, &nj);
  if (rc != MJPL_OK) return rc;
  // the chain as straight-line code, if the engine's library has it (same results; MJPL_POSE_SPEC=0: interpreted)
  int spec_k = -1;
  {
    if (e->spec && e->spec->pose_count > 0 && e->pose_spec != 0) {
      const uint64_t h = chain_hash_of(pi, pd, pd.size());
      for (int k = 0; k < e->spec->pose_count && spec_k < 0; k++)
        if (e->spec->pose_hash(k) == h) spec_k = k;
    }
  }
  pi[PH_MAXIT] = d->iterations;
  pi[PH_OFF_TAIL] = (int)pd.size();
  pd.resize(pd.size() + IT_SIZE);
  double *t = pd.data() + pi[PH_OFF_TAIL];
  for (int k = 0; k < 3; k++) { t[IT_SITE_POS + k] = d->site_pos[k]; t[IT_TGT_POS + k] = d->target_pos[k]; }
  for (int k = 0; k < 4; k++) { t[IT_SITE_QUAT + k] = d->site_quat[k]; t[IT_TGT_QUAT + k] = d->target_quat[k]; }
  t[IT_POS_TOL] = d->pos_tolerance;
  t[IT_ORI_TOL] = d->ori_tolerance;
  t[IT_DAMP] = d->damping > 0 ? d->damping : 1e-6;
  t[IT_LM] = d->lm_damping >= 0 ? d->lm_damping : 0.1;
  t[IT_MAX_STEP] = d->max_step > 0 ? d->max_step : 0.2;
  pi[PH_OFF_JRANGE] = (int)pd.size();
  for (int k = 0; k < 2 * m.njnt; k++) pd.push_back(d->jnt_range[k]);
  for (int k = 0; k < m.njnt; k++) pd.push_back(d->movable[k] ? 1.0 : 0.0);
  const size_t lds = (size_t)kPoseBlock * sizeof(double) * ((size_t)m.nq + 7 * (size_t)nj);  // + the step row
  if (lds > 64 * 1024) return fail(MJPL_E_CAPACITY, "IK: %d qpos + %d chain joints exceed the LDS budget", m.nq, nj);
  HIP_TRY(hipSetDevice(e->device));
  KtScope kt_scope(e, 1);  // (option "kernel_timer")
  // the program is tiny and changes with every target: staged through the engine's scratch
  const size_t ib = pi.size() * sizeof(int), db = pd.size() * sizeof(double);
  if ((rc = stage_reserve(e, 5, ((ib + 7) & ~(size_t)7) + db)) != MJPL_OK) return rc;
  int *d_pi = (int *)e->stage[5];
  double *d_pd = (double *)((char *)e->stage[5] + ((ib + 7) & ~(size_t)7));
  HIP_TRY(hipStreamSynchronize(e->stream));  // pageable host vectors go out of scope on return
  HIP_TRY(hipMemcpy(d_pi, pi.data(), ib, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(d_pd, pd.data(), db, hipMemcpyHostToDevice));
  const unsigned grid = (unsigned)((N + kPoseBlock - 1) / kPoseBlock);
  if (spec_k >= 0) {
    const RowsShape rs = rows_shape(N, e->rows_g);
    const int rc2 = e->spec->ik_solve(spec_k, rs.G, e->stream, rs.grid, d_pi, d_pd, dQ, N, rs.per, dQout, dok, diters, derr,
                                      d->restarts > 0 ? d->restarts : 0, (unsigned long long)d->restart_seed);
    if (rc2 == 0) return MJPL_OK;
    if (rc2 != -1) return fail(MJPL_E_HIP, "generated IK kernel failed to launch");
  }
  hipLaunchKernelGGL(k_ik_solve, dim3(grid), dim3(kPoseBlock), lds, e->stream, d_pi, d_pd, dQ, N, dQout, dok,
                     diters, derr, d->restarts > 0 ? d->restarts : 0, (uint64_t)d->restart_seed);
  HIP_TRY(hipGetLastError());
  return MJPL_OK;
}

int mjpl_ik_solve(mjpl_engine *e, const mjpl_ik_desc *d, const double *Q, int64_t N, double *Q_out,
                  uint8_t *ok, int32_t *iters, double *err) {
  if (!e || !d) return fail(MJPL_E_ARG, "mjpl_ik_solve: NULL argument");
  if (N < 0) return fail(MJPL_E_ARG, "negative batch size");
  if (N == 0) return MJPL_OK;
  if (!Q || !Q_out || !ok) return fail(MJPL_E_ARG, "NULL pointer");
  HIP_TRY(hipSetDevice(e->device));
  const size_t qb = (size_t)N * e->m.nq * sizeof(double);
  int rc;
  if ((rc = stage_reserve(e, 0, qb)) != MJPL_OK) return rc;
  if ((rc = stage_reserve(e, 1, qb)) != MJPL_OK) return rc;
  if ((rc = stage_reserve(e, 2, (size_t)N)) != MJPL_OK) return rc;
  if ((rc = stage_reserve(e, 3, (size_t)N * sizeof(int32_t))) != MJPL_OK) return rc;
  if ((rc = stage_reserve(e, 4, (size_t)N * 2 * sizeof(double))) != MJPL_OK) return rc;
  HIP_TRY(hipMemcpyAsync(e->stage[0], Q, qb, hipMemcpyHostToDevice, e->stream));
  if ((rc = mjpl_ik_solve_dev(e, d, (const double *)e->stage[0], N, (double *)e->stage[1], (uint8_t *)e->stage[2],
                              (int32_t *)e->stage[3], (double *)e->stage[4])) != MJPL_OK)
    return rc;
  HIP_TRY(hipMemcpyAsync(Q_out, e->stage[1], qb, hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipMemcpyAsync(ok, e->stage[2], (size_t)N, hipMemcpyDeviceToHost, e->stream));
  if (iters) HIP_TRY(hipMemcpyAsync(iters, e->stage[3], (size_t)N * sizeof(int32_t), hipMemcpyDeviceToHost, e->stream));
  if (err) HIP_TRY(hipMemcpyAsync(err, e->stage[4], (size_t)N * 2 * sizeof(double), hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  return MJPL_OK;
}

}  // extern "C"

#include "mjpl_rrt.h"
