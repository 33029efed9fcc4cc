// mjpl_rrt.h -- device-resident frontier bi-RRT and the RCCL exchange (SURVEY.md section 8e,
// BASELINE configs[3]).  Included at the end of mjpl_hip.hip: same translation unit, so it uses the
// engine's launch helpers directly.
//
// What it replaces, for L lanes at a time: RRT.plan_to_configs' sample / extend / connect loop
// (reference src/mjpl/planning/rrt.py:190-235) with _constrained_extend's per-step accept / stop
// rules (src/mjpl/planning/utils.py:139-164).  Both trees live in HBM as SoA slabs [nplan][cap]
// (the layout mjpl_nearest_dev reads); a round never moves tree data through the host: the host
// reads back one 64-byte counter block per extension chunk and per exchange, nothing else.
//
// The algorithm (one round r = 1, 2, ... on rank k of W, L lanes per rank) is stated in DESIGN.md
// section 7 and mirrored in NumPy by mjpl_amd/planning/parallel_rrt.py (the CPU / gloo flavour and
// the reference the GPU tests compare whole trees against):
//   1. targets: counter-based splitmix64 draws keyed by (seed, rank, round); goal bias; of the
//      biased lanes that share a target only the lowest takes part;
//   2. nearest node of the growing tree per lane (snapshot at round start);
//   3. extend: repeated _step(cur, target, eps) [-> PoseConstraint projection] -> joint limits ->
//      moved >= 1e-8 -> not farther from the target -> collision edge (endpoint + interval);
//      the accepted prefix of every lane becomes new nodes, ordered (lane, level);
//   4. the other tree extends towards what each lane reached; equal ends = a connection;
//   5. exchange: all ranks' new nodes are all-gathered (RCCL on the engine's stream) and appended
//      in rank order, so every rank holds bit-identical trees and node ids.
#pragma once

#include <dlfcn.h>

#include <chrono>
#include <mutex>

namespace {

using namespace mjpl;

// (the counter-based generator -- sm64 / rrt_key / rrt_u01 -- lives in mjpl_device.h)

struct RrtAcc {                // nodes accepted during the current extension, any order
  double *Q;                   // [cap][nplan]
  int32_t *lane, *level;
  int cap;
};

// Lanes whose chain was capped (mjpl_rrt_desc.max_steps_per_round), one record set per tree: the next time the tree grows
// such a lane goes on from the node it had reached towards the same target instead of drawing a new one (DESIGN.md
// section 7; mjpl_amd/planning/parallel_rrt.py: Carry).  flag == nullptr: no cap, nothing is ever carried.
struct RrtCarry {
  uint8_t *flag;               // [L] the lane's chain of this tree goes on
  double *T;                   // [nplan][L] its target
  int32_t *goal;               // [L] its goal pick (-1: an ordinary sample)
  int32_t *node;               // [L] the node it had reached: pending index -1 - k until the round's exchange, a node id after
};

// ----------------------------------------------------------------------------- kernels
constexpr int kRingStride = 32;  // ints per slot of the pinned counter ring

__global__ void __launch_bounds__(256)
k_rrt_sample(int L, int nplan, uint64_t key, double pgoal, int grow, int ngoal, const double *__restrict__ lo,
             const double *__restrict__ hi, const double *__restrict__ qinit, const double *__restrict__ goalQ,
             int64_t goalcap, RrtLanes ln, int32_t *__restrict__ first, RrtCarry cy) {
  const int l = blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= L) return;
  if (cy.flag && cy.flag[l]) {  // a carried lane: the target and the goal pick of the round it was capped in
    for (int c = 0; c < nplan; c++) ln.T[(int64_t)c * L + l] = cy.T[(int64_t)c * L + l];
    const int g = cy.goal[l];
    if (g >= 0) atomicMin(&first[g], l);  // (among the lanes of one goal it competes like any other)
    ln.goal[l] = g;
    return;
  }
  const uint64_t base = (uint64_t)l * (uint64_t)(nplan + 2);
  const bool biased = rrt_u01(key, base + nplan) <= pgoal;  // rng.random() <= p  (rrt.py:197)
  int g = -1;
  if (biased) {
    if (grow == 0) {  // the start tree grows: a random goal (rrt.py:201-203)
      g = (int)(rrt_u01(key, base + nplan + 1) * (double)ngoal);
      g = g < ngoal ? g : ngoal - 1;
      for (int c = 0; c < nplan; c++) ln.T[(int64_t)c * L + l] = goalQ[(int64_t)c * goalcap + g];
    } else {          // the goal tree grows: q_init (rrt.py:198-199)
      g = 0;
      for (int c = 0; c < nplan; c++) ln.T[(int64_t)c * L + l] = qinit[c];
    }
    atomicMin(&first[g], l);
  } else {
    for (int c = 0; c < nplan; c++) {
      const double u = rrt_u01(key, base + c);
      ln.T[(int64_t)c * L + l] = lo[c] + u * (hi[c] - lo[c]);
    }
  }
  ln.goal[l] = g;
}

// extension start: C = node nearest to the target, lane on unless it duplicates a lower biased lane
__global__ void __launch_bounds__(256)
k_rrt_begin(int L, int nplan, const double *__restrict__ Q, int64_t cap, RrtLanes ln, const int32_t *__restrict__ first,
            const double *__restrict__ Tgt, int second, int *__restrict__ ctr, RrtCarry cy, const int32_t *__restrict__ cf,
            uint8_t *__restrict__ coff) {
  const int l = blockIdx.x * blockDim.x + threadIdx.x;
  if (l == 0) { ctr[RC_EDGES] = 0; ctr[RC_ACTIVE] = 0; ctr[RC_ACC] = 0; }  // for the extension's first chunk
  if (l >= L) return;
  bool on;
  if (!second) {
    const int g = ln.goal[l];
    on = g < 0 || first[g] == l;
    ln.on[l] = on ? 1 : 0;
  } else {
    on = ln.on[l] != 0 && !(cy.flag && cy.flag[l]);  // (a lane carried by the first extension sits the connect phase out)
    // ... and of the lanes whose first extension added nothing and that stand on the same node of the tree -- their connect
    // phases would be the same chain towards the same configuration, node for node -- only the lowest takes part (k_rrt_conn_claim)
    const int ra = ln.refA[l];
    const bool dup = on && ra >= 0 && cf[ra] != l;
    coff[l] = dup ? 1 : 0;
    on = on && !dup;
  }
  int nn = ln.near[l];
  if (!second && cy.flag && cy.flag[l]) {  // a carried chain goes on from where it stopped (a lane that lost its goal to a lower one: dropped)
    if (on) { nn = cy.node[l]; ln.near[l] = nn; }
    cy.flag[l] = 0;  // (consumed: k_rrt_accept raises it again if the chain is capped again)
  }
  bool same = true;
  for (int c = 0; c < nplan; c++) {
    const double v = Q[(int64_t)c * cap + nn];
    ln.C[(int64_t)c * L + l] = v;
    same = same && (v == Tgt[(int64_t)c * L + l]);
  }
  ln.act[l] = (on && !same) ? 1 : 0;
  ln.cnt[l] = 0;
}

// a projecting extension works from packed lists of its active lanes (mjpl_project.h: RC_LISTN): list 0 for chunk 0
__global__ void __launch_bounds__(256)
k_rrt_list_begin(int L, RrtLanes ln, int *__restrict__ ctr) {
  const int l = blockIdx.x * blockDim.x + threadIdx.x;
  const bool act = l < L && ln.act[l] != 0;
  const unsigned long long m = __ballot(act);
  if (m == 0ull) return;
  const int lane = (int)(threadIdx.x & 63);
  int base = 0;
  if (lane == (int)__builtin_ctzll(m)) base = atomicAdd(&ctr[RC_LISTN], __popcll(m));
  base = __shfl(base, (int)__builtin_ctzll(m));
  if (act) ln.list[0][base + __popcll(m & ((1ull << lane) - 1ull))] = l;
}

// One chunk of an extension without a projecting constraint: every active lane walks up to S steps
// from C towards its target and emits the candidate edges (w -> q) it would like validated.  The
// candidates do not depend on the verdicts, so S may exceed 1; the rule checks (joint limits, moved,
// not farther) are made here.  (k_rrt_gen_project below: the same with a PoseConstraint.)
template <int NP>
__global__ void __launch_bounds__(256)
k_rrt_gen(int L, int nplan_arg, int S_arg, int max_steps, double eps, const double *__restrict__ lo,
          const double *__restrict__ hi, const double *__restrict__ Tgt, RrtLanes ln, RrtCand cd, int *__restrict__ ctr) {
  // (NP: the number of planning joints as a constant of the instantiation -- the rows below are registers then; with a
  //  run-time count they live in scratch, and a chunk of 512 lanes took 125 us instead of 50)
  const int nplan = NP > 0 ? NP : nplan_arg;
  const int l = blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63;
  const bool act = l < L && ln.act[l] != 0;
  // (the cap: no more steps than the lane may still add nodes this round -- at least one: a lane at its cap has been stopped)
  const int S = (act && max_steps > 0 && max_steps - ln.cnt[l] < S_arg) ? max_steps - ln.cnt[l] : S_arg;
  double T[kRrtMaxPlan], w[kRrtMaxPlan], q[kRrtMaxPlan], d[kRrtMaxPlan];
  if (act)
    for (int c = 0; c < nplan; c++) { T[c] = Tgt[(int64_t)c * L + l]; w[c] = ln.C[(int64_t)c * L + l]; }
  // one step of the walk; returns 0 = rule failure (nothing emitted), 1 = emit and continue,
  // 2 = emit, then the lane has arrived
  auto step = [&]() -> int {
    for (int c = 0; c < nplan; c++) d[c] = T[c] - w[c];
    const double mag = seqnorm(d, nplan);
    const double sm = eps < mag ? eps : mag;
    bool reach = true;
    for (int c = 0; c < nplan; c++) {
      q[c] = w[c] + (d[c] / mag) * sm;
      reach = reach && (q[c] == T[c]);
    }
    reach = reach || (mag <= eps);
    if (reach)
      for (int c = 0; c < nplan; c++) q[c] = T[c];  // a step of at most eps lands on the target
    bool ok = true;
    for (int c = 0; c < nplan; c++) ok = ok && (q[c] >= lo[c] && q[c] <= hi[c]);
    for (int c = 0; c < nplan; c++) d[c] = q[c] - w[c];
    ok = ok && !(seqnorm(d, nplan) < 1e-8);
    for (int c = 0; c < nplan; c++) d[c] = T[c] - q[c];
    ok = ok && !(seqnorm(d, nplan) > mag);
    return ok ? (reach ? 2 : 1) : 0;
  };
  // pass 1: count
  int count = 0, end = 0;
  if (act) {
    for (int s = 0; s < S; s++) {
      const int r = step();
      if (r == 0) { end = 1; break; }
      count++;
      for (int c = 0; c < nplan; c++) w[c] = q[c];
      if (r == 2) { end = 1; break; }
    }
  }
  // one reservation per wave
  int incl = count;
  for (int o = 1; o < 64; o <<= 1) {
    const int up = __shfl_up(incl, o);
    if (lane >= o) incl += up;
  }
  const int total = __shfl(incl, 63);
  int base = 0;
  if (lane == 0 && total > 0) base = atomicAdd(&ctr[RC_EDGES], total);
  base = __builtin_amdgcn_readfirstlane(base);
  if (!act) return;
  int first = base + incl - count;
  if (first + count > cd.cap) {  // out of candidate space: the lane waits for the next chunk
    atomicOr(&ctr[RC_OVERFLOW], 1);
    // the slots it reserved below the end of the buffer are validated with everything else: give
    // them a harmless zero-length edge at the lane's current configuration
    for (int slot = first; slot < cd.cap; slot++) {
      for (int c = 0; c < nplan; c++) {
        const double v = ln.C[(int64_t)c * L + l];
        cd.A[(int64_t)slot * nplan + c] = v;
        cd.B[(int64_t)slot * nplan + c] = v;
      }
      cd.lane[slot] = l; cd.level[slot] = 0; cd.rule[slot] = 0; cd.reach[slot] = 0;
    }
    count = 0; end = 0; first = 0;
  }
  ln.gfirst[l] = first;
  ln.gcount[l] = count;
  ln.gend[l] = (uint8_t)end;
  if (count == 0) {
    if (end) ln.act[l] = 0;  // stopped by a rule before emitting anything
    return;
  }
  atomicAdd(&ctr[RC_ACTIVE], 1);
  // pass 2: write
  for (int c = 0; c < nplan; c++) w[c] = ln.C[(int64_t)c * L + l];
  const int lvl0 = ln.cnt[l];
  for (int s = 0; s < count; s++) {
    const int r = step();
    const int slot = first + s;
    for (int c = 0; c < nplan; c++) {
      cd.A[(int64_t)slot * nplan + c] = w[c];
      cd.B[(int64_t)slot * nplan + c] = q[c];
    }
    cd.lane[slot] = l;
    cd.level[slot] = lvl0 + s;
    cd.rule[slot] = 1;
    cd.reach[slot] = (r == 2) ? 1 : 0;
    for (int c = 0; c < nplan; c++) w[c] = q[c];
  }
}

// (k_rrt_gen_project: mjpl_project.h)

// accept the leading valid candidates of every lane
__global__ void __launch_bounds__(256)
k_rrt_accept(int L, int nplan, RrtLanes ln, RrtCand cd, RrtAcc acc, int *__restrict__ ctr, int *__restrict__ host_slot, int seq,
             int next_list, int max_steps, int second, const double *__restrict__ Tgt, RrtCarry cy) {
  const int l = blockIdx.x * blockDim.x + threadIdx.x;
  // The chunk's counts are closed (the generating kernel is through): leave them in the pinned block the host
  // will look at two chunks from now (RC_SIZE ints and a sequence word written last) ...
  if (host_slot && l < RC_SIZE) {
    __hip_atomic_store(host_slot + l, ctr[l], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __threadfence_system();
    if (l == 0) __hip_atomic_store(host_slot + 16, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  // ... and clear them for the next chunk instead of two fills
  if (l == 0) { ctr[RC_EDGES] = 0; ctr[RC_ACTIVE] = 0; }
  // (every lane of the wave stays to the end: the accepted rows are copied by the whole wave)
  const bool live = l < L && ln.act[l] != 0;
  const int n = live ? ln.gcount[l] : 0;  // (0: waiting for candidate space, or not in this extension)
  const int f = n > 0 ? ln.gfirst[l] : 0;
  // the leading candidates that are valid and passed the rules -- their flags fetched sixteen at a time: a chunk may
  // hold a thousand steps of a lane
  int a = 0;
  bool arrived = false;
  for (bool more = true; more && a < n;) {
    const int m = n - a < 16 ? n - a : 16;
    uint8_t fv[16], fr[16], fc[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
      const int at = f + a + (k < m ? k : 0);
      fv[k] = cd.valid[at]; fr[k] = cd.rule[at]; fc[k] = cd.reach[at];
    }
    int k = 0;
#pragma unroll
    for (int j = 0; j < 16; j++) {
      const bool good = j < m && fv[j] && fr[j];
      if (more && good) { arrived = fc[j] != 0; k++; }
      else more = more && good;  // (the first that is not: everything behind it is ignored)
    }
    a += k;
    more = more && k == m;
  }
  int at = 0, lvl0 = 0;
  bool keep = a > 0;
  if (keep) {
    at = atomicAdd(&ctr[RC_ACC], a);
    if (at + a > acc.cap) {
      atomicOr(&ctr[RC_OVERFLOW], 2);
      ln.act[l] = 0;
      keep = false;
    } else {
      lvl0 = ln.cnt[l];
    }
  }
  // The accepted rows of a lane are contiguous on both sides (slots f .. f + a of the candidates, at .. at + a of the
  // accepted nodes): the wave copies them lane by lane, 64 elements at a time -- one lane copying a thousand rows of its
  // own, element after element, was a millisecond.
  const int wl = (int)(threadIdx.x & 63);
  for (unsigned long long todo = __ballot(keep); todo; todo &= todo - 1) {
    const int j = (int)__builtin_ctzll(todo);
    const int fj = __shfl(f, j), atj = __shfl(at, j), aj = __shfl(a, j), lj = __shfl(l, j), vj = __shfl(lvl0, j);
    const long long src = (long long)fj * nplan, dst = (long long)atj * nplan;
    for (int k = wl; k < aj * nplan; k += 64) acc.Q[dst + k] = cd.B[src + k];
    for (int k = wl; k < aj; k += 64) { acc.lane[atj + k] = lj; acc.level[atj + k] = vj + k; }
  }
  bool still = live && n == 0;  // (waiting for candidate space: still in the extension)
  if (live && n > 0) {
    if (keep) {
      for (int c = 0; c < nplan; c++) ln.C[(int64_t)c * L + l] = cd.B[(int64_t)(f + a - 1) * nplan + c];
      ln.cnt[l] = lvl0 + a;
    }
    if (!(a > 0 && !keep)) {  // (else: out of room for accepted nodes, the lane was stopped above)
      if (a < n || ln.gend[l] || arrived) ln.act[l] = 0;
      else still = true;
    }
    // the cap: a lane that has added max_steps nodes and is still under way stops for this round.  In the first extension
    // it is carried -- it sits the connect phase out, and goes on the next time its tree grows -- in the second it just stops
    if (still && max_steps > 0 && lvl0 + a >= max_steps) {
      still = false;
      ln.act[l] = 0;
      if (!second) {
        cy.flag[l] = 1;
        cy.goal[l] = ln.goal[l];
        for (int c = 0; c < nplan; c++) cy.T[(int64_t)c * L + l] = Tgt[(int64_t)c * L + l];
      }
    }
  }
  // projecting extensions: the lanes still extending, packed, are the next chunk's rows (next_list: 0 / 1; -1: none kept)
  if (next_list >= 0) {
    const unsigned long long m = __ballot(still);
    if (m != 0ull) {
      int base = 0;
      if (wl == (int)__builtin_ctzll(m)) base = atomicAdd(&ctr[RC_LISTN + next_list], __popcll(m));
      base = __shfl(base, (int)__builtin_ctzll(m));
      if (still) ln.list[next_list][base + __popcll(m & ((1ull << wl) - 1ull))] = l;
    }
  }
}

// exclusive scan of cnt[L] (deterministic node order: lanes ascending) in three small launches: sums of blocks of
// 1024 lanes, a scan of those sums by one workgroup, the scan inside every block on top of its offset.  (One
// workgroup looping over all of L took 0.22 ms for 131 072 lanes, four times a round.)
constexpr int kScanBlock = 1024;

__device__ __forceinline__ int block_scan_1024(int v, int *part, int *block_total) {  // inclusive; all 1024 threads call
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  int incl = v;
  for (int o = 1; o < 64; o <<= 1) {
    const int up = __shfl_up(incl, o);
    if (lane >= o) incl += up;
  }
  if (lane == 63) part[w] = incl;
  __syncthreads();
  if (w == 0) {
    int x = lane < kScanBlock / 64 ? part[lane] : 0, xi = x;
    for (int o = 1; o < 16; o <<= 1) {
      const int up = __shfl_up(xi, o);
      if (lane >= o) xi += up;
    }
    if (lane < kScanBlock / 64) part[lane] = xi - x;  // exclusive offsets of the waves
    if (lane == kScanBlock / 64 - 1) *block_total = xi;
  }
  __syncthreads();
  return incl + part[w];
}

__global__ void __launch_bounds__(kScanBlock)
k_rrt_scan_sums(int L, const int32_t *__restrict__ cnt, int32_t *__restrict__ sums) {
  __shared__ int part[kScanBlock / 64];
  __shared__ int total;
  const int i = blockIdx.x * kScanBlock + threadIdx.x;
  (void)block_scan_1024(i < L ? cnt[i] : 0, part, &total);
  if (threadIdx.x == 0) sums[blockIdx.x] = total;
}

__global__ void __launch_bounds__(kScanBlock)
k_rrt_scan_blocks(int nblocks, int32_t *__restrict__ sums, int *__restrict__ total) {  // nblocks <= 1024 per pass of the loop
  __shared__ int part[kScanBlock / 64];
  __shared__ int tot;
  int carry = 0;
  for (int base = 0; base < nblocks; base += kScanBlock) {
    const int i = base + threadIdx.x;
    const int v = i < nblocks ? sums[i] : 0;
    const int incl = block_scan_1024(v, part, &tot);
    if (i < nblocks) sums[i] = carry + incl - v;
    carry += tot;
    __syncthreads();
  }
  if (threadIdx.x == 0) *total = carry;
}

__global__ void __launch_bounds__(kScanBlock)
k_rrt_scan_apply(int L, const int32_t *__restrict__ cnt, const int32_t *__restrict__ sums, int32_t *__restrict__ off) {
  __shared__ int part[kScanBlock / 64];
  __shared__ int total;
  const int i = blockIdx.x * kScanBlock + threadIdx.x;
  const int v = i < L ? cnt[i] : 0;
  const int incl = block_scan_1024(v, part, &total);
  if (i < L) off[i] = sums[blockIdx.x] + incl - v;
}

// place the accepted nodes in (lane, level) order into this rank's pending slab of the tree
__global__ void __launch_bounds__(256)
k_rrt_place(int nacc, int nplan, RrtAcc acc, const int32_t *__restrict__ off, const int32_t *__restrict__ near,
            double *__restrict__ pendQ, int32_t *__restrict__ pendpar) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nacc) return;
  const int l = acc.lane[i], lv = acc.level[i];
  const int pos = off[l] + lv;
  for (int c = 0; c < nplan; c++) pendQ[(int64_t)pos * nplan + c] = acc.Q[(int64_t)i * nplan + c];
  pendpar[pos] = lv == 0 ? near[l] : -pos;  // >= 0: node id in the tree; < 0: -1 - (pending index pos - 1)
}

__global__ void __launch_bounds__(256)
k_rrt_finish(int L, int nplan, RrtLanes ln, int32_t *__restrict__ ref, double *__restrict__ reached) {
  const int l = blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= L) return;
  const int n = ln.cnt[l];
  ref[l] = n > 0 ? -1 - (ln.off[l] + n - 1) : ln.near[l];
  if (reached)
    for (int c = 0; c < nplan; c++) reached[(int64_t)c * L + l] = ln.C[(int64_t)c * L + l];
}

// the lanes capped in this round's first extension: their last node (a pending index of this rank's block) as a node id
__global__ void __launch_bounds__(256)
k_rrt_carry_node(int L, RrtCarry cy, const int32_t *__restrict__ ref, int base) {
  const int l = blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= L || !cy.flag[l]) return;
  cy.node[l] = base + (-1 - ref[l]);  // (a capped lane added max_steps >= 1 nodes: its reference is a pending one)
}

__global__ void k_rrt_first_init(int32_t *__restrict__ first, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) first[i] = 0x7fffffff;
}

// the early look-up of the connect phase's nearest nodes (mjpl_rrt: early_nn): what every lane has reached so far, and
// whether that is final -- a lane with act == 0 is through, nothing writes its C again in this extension
__global__ void __launch_bounds__(256)
k_rrt_early(int L, int nplan, RrtLanes ln, uint8_t *__restrict__ early, double *__restrict__ reached) {
  const int l = blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= L) return;
  early[l] = ln.act[l] == 0 ? 1 : 0;  // (on the extension's own stream, between two chunks: nothing else writes act or C meanwhile)
  for (int c = 0; c < nplan; c++) reached[(int64_t)c * L + l] = ln.C[(int64_t)c * L + l];
}

// ... the lanes that were still under way then, packed ([0] of ctr2: how many), with their final targets as queries;
// the rows behind them repeat the first (the look-up is sized for `room` queries whatever the count)
__global__ void __launch_bounds__(256)
k_rrt_late_list(int L, const uint8_t *__restrict__ early, int32_t *__restrict__ pos, int *__restrict__ count) {
  const int l = blockIdx.x * blockDim.x + threadIdx.x;
  const bool late = l < L && early[l] == 0;
  const unsigned long long m = __ballot(late);
  if (m == 0ull) return;
  const int lane = (int)(threadIdx.x & 63);
  int base = 0;
  if (lane == (int)__builtin_ctzll(m)) base = atomicAdd(count, __popcll(m));
  base = __shfl(base, (int)__builtin_ctzll(m));
  if (late) pos[l] = base + __popcll(m & ((1ull << lane) - 1ull));
}
__global__ void __launch_bounds__(256)
k_rrt_late_fill(int L, int nplan, int room, const double *__restrict__ Tgt, double *__restrict__ q) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= room) return;
  for (int c = 0; c < nplan; c++) q[(int64_t)c * room + i] = Tgt[(int64_t)c * L];  // (lane 0's target: a query like any other)
}
__global__ void __launch_bounds__(256)
k_rrt_late_queries(int L, int nplan, int room, const uint8_t *__restrict__ early, const int32_t *__restrict__ pos,
                   const double *__restrict__ Tgt, double *__restrict__ q, int *__restrict__ ctr) {
  const int l = blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= L || early[l]) return;
  const int at = pos[l];
  if (at >= room) { atomicOr(&ctr[RC_OVERFLOW], 4); return; }  // (cannot happen: lanes only ever leave an extension)
  for (int c = 0; c < nplan; c++) q[(int64_t)c * room + at] = Tgt[(int64_t)c * L + l];
}
__global__ void __launch_bounds__(256)
k_rrt_near_merge(int L, int room, const uint8_t *__restrict__ early, const int32_t *__restrict__ pos, const int32_t *__restrict__ near_e,
                 const int32_t *__restrict__ near_late, int32_t *__restrict__ near) {
  const int l = blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= L) return;
  near[l] = early[l] ? near_e[l] : near_late[pos[l] < room ? pos[l] : 0];
}

// between the two extensions: the lowest lane among those that stand on the same old node of the tree that grew
__global__ void __launch_bounds__(256)
k_rrt_conn_claim(int L, RrtLanes ln, RrtCarry cy, int32_t *__restrict__ cf) {
  const int l = blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= L || !ln.on[l] || (cy.flag && cy.flag[l])) return;
  const int ra = ln.refA[l];
  if (ra >= 0) atomicMin(&cf[ra], l);
}

__global__ void __launch_bounds__(256)
k_rrt_connect(int L, int nplan, RrtLanes ln, int *__restrict__ ctr, RrtCarry cy, int32_t *__restrict__ cf, const uint8_t *__restrict__ coff) {
  const int l = blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= L || !ln.on[l] || (cy.flag && cy.flag[l])) return;  // (carried: no connect phase this round)
  {
    const int ra = ln.refA[l];
    if (ra >= 0) cf[ra] = 0x7f7f7f7f;  // (the claims of this round: cleared by the lanes that made them)
  }
  if (coff[l]) return;  // (another lane ran this lane's connect phase)
  bool eq = true;
  for (int c = 0; c < nplan; c++) eq = eq && (ln.RA[(int64_t)c * L + l] == ln.C[(int64_t)c * L + l]);
  if (eq) atomicMin(&ctr[RC_CONN], l);
}

// this rank's exchange header: [new nodes of the tree that grew, of the other tree, connecting lane
// (INT_MAX: none), its reference in the start tree, in the goal tree, 0, 0, 0]
__global__ void k_rrt_header(RrtLanes ln, const int *__restrict__ ctr, int grow, int stop, int *__restrict__ head) {
  const int l = ctr[RC_CONN];
  head[0] = ctr[RC_NEWA];
  head[1] = ctr[RC_NEWB];
  head[2] = l;
  head[3] = head[4] = 0;
  if (l != 0x7fffffff) {  // [3]: the lane's node in the start tree, [4]: in the goal tree
    head[3] = grow == 0 ? ln.refA[l] : ln.refB[l];
    head[4] = grow == 0 ? ln.refB[l] : ln.refA[l];
  }
  head[5] = stop;  // this rank asks for the search to end (time limit): honoured by all ranks together
  head[6] = head[7] = 0;
}

__global__ void k_rrt_round_init(int *__restrict__ ctr, int32_t *__restrict__ first, int nfirst) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < RC_SIZE) ctr[i] = i == RC_CONN ? 0x7fffffff : 0;
  if (i < nfirst) first[i] = 0x7fffffff;
}

// append one rank's slab to a tree: rows [cnt][nplan] AoS -> SoA [nplan][cap] at base
__global__ void __launch_bounds__(256)
k_rrt_merge(int cnt, int nplan, const double *__restrict__ rows, const int32_t *__restrict__ par, double *__restrict__ Q,
            int32_t *__restrict__ parent, int64_t cap, int base) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= cnt) return;
  for (int c = 0; c < nplan; c++) Q[(int64_t)c * cap + base + i] = rows[(int64_t)i * nplan + c];
  const int p = par[i];
  parent[base + i] = p >= 0 ? p : base + (-1 - p);
}

// root-ward walk from `node`, rows written leaf first
__global__ void k_rrt_walk(const double *__restrict__ Q, const int32_t *__restrict__ parent, int64_t cap, int nplan, int node,
                           int maxlen, double *__restrict__ out, int *__restrict__ len) {
  int n = 0;
  while (node >= 0 && n < maxlen) {
    for (int c = 0; c < nplan; c++) out[(int64_t)n * nplan + c] = Q[(int64_t)c * cap + node];
    node = parent[node];
    n++;
  }
  *len = node >= 0 ? -1 : n;
}

// ----------------------------------------------------------------------------- RCCL, loaded on demand
struct Id128 { char b[128]; };  // ncclUniqueId
struct Rccl {
  void *lib = nullptr;
  int (*GetUniqueId)(void *) = nullptr;
  int (*CommInitRank)(void **, int, Id128, int) = nullptr;  // the id travels by value
  int (*AllGather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
  int (*CommDestroy)(void *) = nullptr;
  const char *(*GetErrorString)(int) = nullptr;
};

// (every engine of the process shares the one table: std::call_once, so that two threads creating their
// first communicators together load the library once and both see the filled table)
Rccl *rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    for (const char *name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
      r.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (r.lib) break;
    }
    if (r.lib) {
      r.GetUniqueId = (int (*)(void *))dlsym(r.lib, "ncclGetUniqueId");
      r.CommInitRank = (int (*)(void **, int, Id128, int))dlsym(r.lib, "ncclCommInitRank");
      r.AllGather = (int (*)(const void *, void *, size_t, int, void *, hipStream_t))dlsym(r.lib, "ncclAllGather");
      r.CommDestroy = (int (*)(void *))dlsym(r.lib, "ncclCommDestroy");
      r.GetErrorString = (const char *(*)(int))dlsym(r.lib, "ncclGetErrorString");
      if (!r.GetUniqueId || !r.CommInitRank || !r.AllGather || !r.CommDestroy) { dlclose(r.lib); r.lib = nullptr; }
    }
  });
  return r.lib ? &r : nullptr;
}

constexpr int kNcclInt8 = 0;  // ncclInt8 / ncclChar

}  // namespace

struct mjpl_rrt {
  mjpl_engine *e = nullptr;
  mjpl_pose *pose = nullptr;
  int nplan = 0, nq = 0, L = 0;
  int64_t cap = 0;
  double eps = 0.05, istep = 0, pgoal = 0.05;
  uint64_t seed = 0;
  int ngoal = 0, round = 0;
  int32_t *d_cf = nullptr;     // [cap] per node of the tree that grew: the lowest lane standing on it after the first extension (0x7f7f7f7f: none)
  uint8_t *d_coff = nullptr;   // [L] the lane's connect phase is another lane's
  int max_steps = 0;           // most nodes a lane adds per extension (0: no cap); the lanes capped in a tree's first extension are carried
  RrtCarry carry[2] = {{nullptr, nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr, nullptr}};  // per tree
  int n[2] = {0, 0};
  double *d_Q[2] = {nullptr, nullptr};
  int32_t *d_parent[2] = {nullptr, nullptr};
  double *d_lo = nullptr, *d_hi = nullptr, *d_qinit = nullptr, *d_qbase = nullptr;
  int *d_qidx = nullptr;
  uint8_t *d_isplan = nullptr;
  RrtLanes ln{};
  RrtCand cd{};
  RrtAcc acc{};
  int32_t *d_first = nullptr;
  int first_cap = 1;
  int32_t *d_scan = nullptr;   // block sums of the lane scan
  double *d_pendQ[2] = {nullptr, nullptr};
  int32_t *d_pendpar[2] = {nullptr, nullptr};
  int pendcap = 0;
  // projecting extensions: most steps a lane takes per chunk, and the candidate slots a chunk is sized for
  // (S = min(proj_steps_max, proj_slots / active lanes); MJPL_RRT_PROJ_STEPS / MJPL_RRT_PROJ_SLOTS)
  // (round 4, with the chain as generated code a step is cheap enough for the chunk's fixed cost to show: 64 / 65 536 ->
  //  65.2 ms per 131 072-lane round, 256 / 262 144 -> 62.6, 1 024 / 1 048 576 -> 59.8; the slots are bounded by the
  //  candidate buffer, 4 L, all the same)
  int proj_steps_max = 1024;
  int64_t proj_slots = 1 << 20;
  // ... and the shape of the generating launch (mjpl_rows.h; mjpl_hip.hip: rows_shape): lanes per row, waves per launch
  // (one per SIMD: the kernels' registers) -- MJPL_RRT_PROJ_G / MJPL_RRT_PROJ_WAVES for A/B timing
  int proj_g = 0;            // 0: lanes per row by the number of active lanes (rows_shape); 1 / 4 / 8 / 16 forced
  // rows of sixteen lanes that run a step ahead (mjpl_rows.h: k_rrt_gen_project_ahead), when no more lanes than this extend:
  // one row per wave (measured: with four rows per wave, 1 536 lanes, 14.0 ms against 13.5 with eight lanes per row -- the
  // statements of running ahead are paid by every row of the wave; one row per wave, 268 lanes x 1 024 steps: 20.5 ms
  // against 23.9).  MJPL_RRT_AHEAD=0: never; MJPL_RRT_AHEAD_LANES
  int ahead = 1, ahead_lanes = 1024;
  int proj_waves_max = 1024;
  // The connect phase's nearest neighbours, early (round 5).  A lane's round does not depend on any other lane's (both trees
  // are the round's snapshot), and the tail of the first extension is a handful of lanes taking a thousand sequential steps
  // on an otherwise idle chip: once no more than `early_lanes` lanes are still extending, the lanes that are through get
  // their nearest node of the OTHER tree on a second stream (what they reached is final: ln.C of a lane with act == 0);
  // the second extension then looks up only the lanes that were still under way, and waits for the early answers.  The
  // same queries against the same nodes: the same nodes.  MJPL_RRT_EARLY_NN=0 turns it off; MJPL_RRT_EARLY_LANES /
  // MJPL_RRT_EARLY_MIN_NODES move the thresholds (tests set them low).
  int exact_counts = 1;  // a projecting extension reads every chunk's lane count before it sizes the chunk (MJPL_RRT_EXACT_COUNTS=0: two chunks late, from the pinned ring)
  int early_nn = 1, early_lanes = 4096, early_next = 1;  // (early_next: the next round's look-up as well, MJPL_RRT_EARLY_NEXT=0: not)
  int64_t early_min_nodes = 65536;
  bool early_on = false;      // this round's first extension started the early look-up
  hipStream_t side = nullptr;
  hipEvent_t ev_tail = nullptr, ev_near = nullptr;
  double *d_RAe = nullptr;    // [nplan][L] what the lanes had reached when the early look-up started
  int32_t *d_near_e = nullptr, *d_late_pos = nullptr, *d_late_near = nullptr;
  uint8_t *d_early = nullptr; // the lane's answer is the early one
  double *d_late_q = nullptr; // [nplan][early_lanes] the other lanes' queries, packed (padded with the first of them)
  // ... and, behind that look-up on the same stream, the NEXT round's targets (counter-based draws: they depend on the seed,
  // the rank and the round number only) against the nodes its growing tree holds NOW -- the tree this round's connect phase
  // extends; what that adds is appended at the round's end, behind them, and is all the next round still has to scan
  // (nearest_range: lower indices win ties, as in one scan).  pre_round: the round the draws are for (0: none).
  int pre_round = 0;
  int pre_n0 = 0;             // nodes of that tree covered by the early answers
  double *d_Tn = nullptr;     // [nplan][L] the next round's targets
  int32_t *d_goal_n = nullptr, *d_first_n = nullptr, *d_pre_idx = nullptr;
  double *d_pre_d2 = nullptr;
  hipEvent_t ev_pre = nullptr;
  // the look-ups on the second stream use their own scratch (the engine's may be in use by a look-up on the first)
  struct NnScratch { void *nn = nullptr; size_t nn_bytes = 0; void *nn16 = nullptr; size_t nn16_bytes = 0; void *tmp = nullptr; size_t tmp_bytes = 0; } side_nn;
  int *d_ctr = nullptr, *h_ctr = nullptr;
  // projecting extensions read their chunk counters two chunks late (rrt_extend): a ring of pinned copies
  int *h_ring = nullptr;      // 4 slots of kRingStride ints: RC_SIZE counters, then the sequence word
  int ring_seq0 = 1;          // sequence number of chunk 0 of the extension under way (never repeats within the ring)
  // exchange
  int *d_heads = nullptr, *h_heads = nullptr;
  char *d_gather = nullptr;
  size_t gather_bytes = 0;
  double *d_path = nullptr;
  std::vector<void *> owned;
  // result of the last successful round
  int conn_start = -1, conn_goal = -1;
  // rank identity when the caller moves the slabs itself (mjpl_rrt_set_world); else the engine's communicator
  int rank = 0, world = 1;
  bool world_set = false;
  bool in_round = false;     // between round_begin and round_finish
  int *h_myhead = nullptr;   // pinned: this rank's header (error relay / round_begin's copy)
  std::string local_err;     // what went wrong on THIS rank in the round in flight
};

namespace {

template <class T>
int rrt_alloc(mjpl_rrt *r, T **p, size_t count) {
  void *v = nullptr;
  HIP_TRY(hipMalloc(&v, std::max<size_t>(count, 1) * sizeof(T)));
  r->owned.push_back(v);
  *p = (T *)v;
  return MJPL_OK;
}

inline unsigned rgrid(int64_t n) { return (unsigned)((n + 255) / 256); }

// MJPL_RRT_TRACE=1: where a round's wall time goes (stderr; synchronises the stream at every mark)
struct RrtTrace {
  bool on, chunks;  // option "rrt_trace": 1 = the phases, 2 = a line per extension chunk as well (the stream is synchronised at every mark)
  explicit RrtTrace(const mjpl_engine *e) : on(e->rrt_trace >= 1), chunks(e->rrt_trace >= 2) {}
  void chunk(hipStream_t st, int index, int active, int S, int G, unsigned grid) {
    if (!chunks) return;
    (void)hipStreamSynchronize(st);
    const auto t1 = std::chrono::steady_clock::now();
    fprintf(stderr, "[rrt]   chunk %3d  active <= %7d  S %5d  G %d  waves %5u  %9.3f ms\n", index, active, S, G, grid,
            std::chrono::duration<double, std::milli>(t1 - t0).count());
    t0 = std::chrono::steady_clock::now();
  }
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  // (after a look-up: what the cell-ordered scan had to do -- options "nn_last_candidate_fraction", "nn_last_exact_pairs")
  void nn(mjpl_engine *e) {
    if (!on) return;
    double frac = -1, pairs = -1;
    (void)mjpl_get_option(e, "nn_last_candidate_fraction", &frac);
    (void)mjpl_get_option(e, "nn_last_exact_pairs", &pairs);
    fprintf(stderr, "[rrt]   (look-up: %.4f of the tree's sub-chunks on a wave's list, %.0f exact distances)\n", frac, pairs);
  }
  void mark(hipStream_t st, const char *what, long n = -1) {
    if (!on) return;
    (void)hipStreamSynchronize(st);
    const auto t1 = std::chrono::steady_clock::now();
    fprintf(stderr, "[rrt] %-28s %9.3f ms", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
    if (n >= 0) fprintf(stderr, "  (%ld)", n);
    fprintf(stderr, "\n");
    t0 = std::chrono::steady_clock::now();
  }
};

int rrt_read_ctr(mjpl_rrt *r) {
  HIP_TRY(hipMemcpyAsync(r->h_ctr, r->d_ctr, RC_SIZE * sizeof(int), hipMemcpyDeviceToHost, r->e->stream));
  HIP_TRY(hipStreamSynchronize(r->e->stream));
  return MJPL_OK;
}

int rrt_rank(const mjpl_rrt *r);

// a look-up enqueued on the planner's second stream, with its scratch.  (The engine's stream and scratch pointers are swapped
// for the duration of the CALL -- host code, one thread per engine by the ABI's contract, nothing is enqueued in between --
// not for the duration of the kernels: those run on `side` with the side scratch while the main stream goes on with its own.
// The side scratch is allocated by its first look-up of a search, sized like the engine's: nn_reserve_nodes.)
int rrt_side_nearest(mjpl_rrt *r, const double *nodes, int64_t n, const double *queries, int64_t M, int32_t *idx, double *d2) {
  mjpl_engine *e = r->e;
  auto swap_scratch = [&]() {
    std::swap(e->d_nn, r->side_nn.nn); std::swap(e->nn_bytes, r->side_nn.nn_bytes);
    std::swap(e->d_nn16, r->side_nn.nn16); std::swap(e->nn16_bytes, r->side_nn.nn16_bytes);
    std::swap(e->d_nn_tmp, r->side_nn.tmp); std::swap(e->nn_tmp_bytes, r->side_nn.tmp_bytes);
  };
  hipStream_t keep = e->stream;
  const int last = e->nn_last;
  swap_scratch();
  e->stream = r->side;  // (the look-up's launches go where the engine's stream points)
  const int rc = mjpl_nearest_dev(e, nodes, n, r->cap, queries, M, idx, d2);
  e->stream = keep;
  swap_scratch();
  e->nn_last = last;    // (mjpl_nearest_last_screen speaks of the engine's own scratch)
  return rc;
}

// one extension of tree `t` towards the targets Tgt ([nplan][L]); `second`: the connect phase
int rrt_extend(mjpl_rrt *r, int t, const double *Tgt, int second, int *nnew) {
  mjpl_engine *e = r->e;
  hipStream_t st = e->stream;
  const int L = r->L, nplan = r->nplan;
  RrtTrace tr(e);
  tr.mark(st, "(before the extension)");
  int rc = MJPL_OK;
  if (second && r->early_on) {
    // the lanes that were through when the first extension's tail began have their answers (or will have: ev_near);
    // the others -- early_lanes at most -- are looked up now
    r->early_on = false;
    const int room = r->early_lanes;
    HIP_TRY(hipMemsetAsync(r->d_late_near, 0, sizeof(int), st));  // ([0] doubles as the list's counter until the look-up writes it)
    hipLaunchKernelGGL(k_rrt_late_list, dim3(rgrid(L)), dim3(256), 0, st, L, r->d_early, r->d_late_pos, (int *)r->d_late_near);
    hipLaunchKernelGGL(k_rrt_late_fill, dim3(rgrid(room)), dim3(256), 0, st, L, nplan, room, Tgt, r->d_late_q);
    hipLaunchKernelGGL(k_rrt_late_queries, dim3(rgrid(L)), dim3(256), 0, st, L, nplan, room, r->d_early, r->d_late_pos, Tgt, r->d_late_q,
                       r->d_ctr);
    HIP_TRY(hipStreamWaitEvent(st, r->ev_near, 0));  // (d_near_e: the early answers, written on the second stream)
    rc = mjpl_nearest_dev(e, r->d_Q[t], r->n[t], r->cap, r->d_late_q, room, r->d_late_near, nullptr);
    if (rc != MJPL_OK) return rc;
    hipLaunchKernelGGL(k_rrt_near_merge, dim3(rgrid(L)), dim3(256), 0, st, L, room, r->d_early, r->d_late_pos, r->d_near_e, r->d_late_near,
                       r->ln.near);
    tr.mark(st, "nearest neighbour (the lanes of the tail; the others were looked up early)", r->n[t]);
    tr.nn(e);
  } else if (!second && r->pre_round == r->round && r->pre_n0 <= r->n[t]) {
    // this round's targets were looked up in the tree's first pre_n0 nodes while the round before ran its tail
    r->pre_round = 0;
    HIP_TRY(hipStreamWaitEvent(st, r->ev_pre, 0));
    rc = nearest_range(e, r->d_Q[t], r->pre_n0, r->n[t], r->cap, Tgt, L, r->ln.near, nullptr, r->d_pre_idx, r->d_pre_d2);
    if (rc != MJPL_OK) return rc;
    tr.mark(st, "nearest neighbour (the nodes of the last round; the others were scanned early)", r->n[t] - r->pre_n0);
  } else {
    rc = mjpl_nearest_dev(e, r->d_Q[t], r->n[t], r->cap, Tgt, L, r->ln.near, nullptr);
    if (rc != MJPL_OK) return rc;
    tr.mark(st, "nearest neighbour", r->n[t]);
    tr.nn(e);
  }
  const bool projecting = r->pose != nullptr;
  if (projecting) HIP_TRY(hipMemsetAsync(r->d_ctr + RC_LISTN, 0, 2 * sizeof(int), st));  // (both lists of the extension: empty)
  hipLaunchKernelGGL(k_rrt_begin, dim3(rgrid(L)), dim3(256), 0, st, L, nplan, r->d_Q[t], r->cap, r->ln, r->d_first, Tgt,
                     second, r->d_ctr, r->carry[second ? 1 - t : t],  // (the records of the tree that grows this round)
                     (const int32_t *)r->d_cf, r->d_coff);
  if (projecting) hipLaunchKernelGGL(k_rrt_list_begin, dim3(rgrid(L)), dim3(256), 0, st, L, r->ln, r->d_ctr);
  // (without a projection: four steps per lane in the first chunk of a big batch, doubling; a small batch -- a planner of
  //  a few hundred lanes -- starts with as many as 2^16 candidate slots allow, up to 64: its chunks cost their launches'
  //  latency whatever they hold, and an extension of six chunks becomes one of two)
  int S = projecting ? 1 : (int)std::max<int64_t>(4, std::min<int64_t>(std::min<int64_t>(64, ((int64_t)1 << 16) / std::max(1, L)), (int64_t)r->cd.cap / std::max(1, L)));
  // With a projecting constraint an extension is hundreds of chunks, the late ones with a handful of lanes:
  // waiting for every chunk's counters before sizing its launches would leave the GPU idle while the host
  // enqueues the next kernels.  Lanes only ever leave an extension, and every active lane owns S slots of the
  // chunk's candidates, so the active count of two chunks ago bounds this chunk's: chunk i is sized by it -- the
  // surplus rows are earlier chunks' candidates, validated again and read by nobody -- and the loop ends two
  // chunks after the last lane has.  The counters come from a pinned ring k_rrt_accept writes (slot chunk % 4):
  // no copy, no event, the host spins on a sequence word.
  if (projecting && r->cd.cap < L) return fail(MJPL_E_CAPACITY, "rrt: candidate buffer too small for one step per lane");
  int chunks_done = 0;
  // Sequence numbers of the pinned ring never repeat -- on ANY way out of this function: an extension that
  // fails half-way has k_rrt_accept launches in flight that still write their slots, and the next extension
  // on this planner must neither meet their numbers again nor find a stale slot that already carries the
  // number it waits for.  A failing exit therefore also waits for the stream before the numbers move on.
  struct RingGuard {
    mjpl_rrt *r; const int *chunks; hipStream_t st; bool ok;
    ~RingGuard() {
      if (!ok) (void)hipStreamSynchronize(st);
      r->ring_seq0 += *chunks + 8;
    }
  } ring_guard{r, &chunks_done, st, false};
  // the look-ups that run beside the tail of the first extension (mjpl_rrt: early_nn), started once, when no more than
  // early_lanes lanes are still extending: everything enqueued so far (the acceptance of the chunk before last) decides
  // which lanes are through
  auto start_tail_lookups = [&](int active) -> int {
    if (second || !r->early_nn || r->early_on || active > r->early_lanes || r->n[1 - t] < r->early_min_nodes) return MJPL_OK;
    // (the snapshot of who is through, and of what they reached, is taken on the MAIN stream, between two chunks: on the
    //  second stream it would read act / C while a later chunk's acceptance kernel writes them)
    hipLaunchKernelGGL(k_rrt_early, dim3(rgrid(L)), dim3(256), 0, st, L, nplan, r->ln, r->d_early, r->d_RAe);
    HIP_TRY(hipEventRecord(r->ev_tail, st));
    HIP_TRY(hipStreamWaitEvent(r->side, r->ev_tail, 0));
    int nrc = rrt_side_nearest(r, r->d_Q[1 - t], r->n[1 - t], r->d_RAe, L, r->d_near_e, nullptr);
    if (nrc != MJPL_OK) return nrc;
    HIP_TRY(hipEventRecord(r->ev_near, r->side));
    if (r->early_next) {
      // the next round: its growing tree is 1 - t, its draws are keyed by the round number
      RrtLanes lnn = r->ln;
      lnn.T = r->d_Tn; lnn.goal = r->d_goal_n;
      const int nf = std::max(r->ngoal, 1);
      hipLaunchKernelGGL(k_rrt_first_init, dim3(rgrid(nf)), dim3(256), 0, r->side, r->d_first_n, nf);
      hipLaunchKernelGGL(k_rrt_sample, dim3(rgrid(L)), dim3(256), 0, r->side, L, nplan,
                         rrt_key(r->seed, (uint64_t)rrt_rank(r), (uint64_t)(r->round + 1)), r->pgoal, 1 - t, r->ngoal, r->d_lo, r->d_hi,
                         r->d_qinit, r->d_Q[1], r->cap, lnn, r->d_first_n, r->carry[1 - t]);
      nrc = rrt_side_nearest(r, r->d_Q[1 - t], r->n[1 - t], r->d_Tn, L, r->d_pre_idx, r->d_pre_d2);
      if (nrc != MJPL_OK) return nrc;
      HIP_TRY(hipEventRecord(r->ev_pre, r->side));
      r->pre_round = r->round + 1;
      r->pre_n0 = r->n[1 - t];
    }
    r->early_on = true;
    return MJPL_OK;
  };
  int active_bound = L;  // (projecting) no more lanes than this are active in the chunk about to be launched
  int steps_done = 0;    // (projecting) steps per lane of the chunks launched so far
  int trace_G = 0;
  unsigned trace_grid = 0;
  for (int chunk = 0;; chunk++) {
    chunks_done = chunk + 1;
    if (projecting) {
      if (r->exact_counts) {
        // Round 5: the chunk's list has been closed by the acceptance kernel before it (chunk 0: by k_rrt_list_begin); its
        // length IS the number of active lanes.  Reading it costs a trip to the host with the chip idle (~30 us) and buys
        // the right number of steps for the chunk: sized by the count of two chunks ago (below) an extension's first three
        // chunks all took four steps and it needed eight to ten chunks where five or six do (profiles/README.md).
        if ((rc = rrt_read_ctr(r)) != MJPL_OK) return rc;
        if (r->h_ctr[RC_OVERFLOW] & 2) return fail(MJPL_E_CAPACITY, "rrt: more new nodes in one extension than the pending slab holds");
        if (r->h_ctr[RC_OVERFLOW] & 1) return fail(MJPL_E_CAPACITY, "rrt: candidate buffer overrun in a projecting extension");
        if (r->h_ctr[RC_LISTN + (chunk & 1)] == 0) break;
        active_bound = r->h_ctr[RC_LISTN + (chunk & 1)];
        if (chunk >= 1 && (rc = start_tail_lookups(active_bound)) != MJPL_OK) return rc;
      } else if (chunk >= 2) {
        const int look = (chunk - 2) % 4;
        volatile int *slot = r->h_ring + look * kRingStride;
        const int want = r->ring_seq0 + chunk - 2;
        for (long spins = 0; __atomic_load_n(slot + 16, __ATOMIC_ACQUIRE) != want; spins++) {
          if (spins > 2000000000L) return fail(MJPL_E_HIP, "rrt: the counters of an extension chunk never arrived");
          __builtin_ia32_pause();
        }
        for (int k = 0; k < RC_SIZE; k++) r->h_ctr[k] = slot[k];
        if (r->h_ctr[RC_OVERFLOW] & 2) return fail(MJPL_E_CAPACITY, "rrt: more new nodes in one extension than the pending slab holds");
        if (r->h_ctr[RC_OVERFLOW] & 1) return fail(MJPL_E_CAPACITY, "rrt: candidate buffer overrun in a projecting extension");
        if (r->h_ctr[RC_ACTIVE] == 0) break;
        active_bound = r->h_ctr[RC_ACTIVE];
        if ((rc = start_tail_lookups(active_bound)) != MJPL_OK) return rc;
      }
      // Steps per chunk: one while the chunk's kernels are busy with the lanes there are; more when they are
      // not -- a chunk then costs the latency of its launches whatever it holds, and S steps share it.  The
      // speculation costs slots (S per lane) and nothing else: a lane that stops early stops generating.
      S = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(r->proj_steps_max, r->proj_slots / std::max(1, active_bound)),
                                                      (int64_t)r->cd.cap / std::max(1, active_bound)));
      // the cap: a lane still extending has added every step of every chunk so far (any other outcome ended it)
      if (r->max_steps > 0) S = std::min(S, std::max(1, r->max_steps - steps_done));
      {
        // The chunk's rows are the packed list chunk & 1 (at most active_bound entries).  A model's library walks them
        // with waves that refill (mjpl_rows.h): one lane per row while there are many, four or eight lanes per row once
        // all of them fit the chip at once (rows_shape) -- the tail of an extension, which is most of its time.
        const int par = chunk & 1;
        const size_t plds = pose_lds(r->pose) + (size_t)kPoseBlock * sizeof(double) * (size_t)r->nq;
        const int pk = pose_spec_index(r->pose);  // (the chain as straight-line code, if the engine's library has it)
        int launched = -1;
        KtScope kt_scope(e, 1);  // (option "kernel_timer": the generating kernel of every chunk)
        trace_G = 1; trace_grid = (unsigned)((active_bound + kPoseBlock - 1) / kPoseBlock);
        if (pk >= 0) {
          const RowsShape rs = r->proj_g > 0 ? RowsShape{r->proj_g, 0u, 0} : rows_shape(active_bound, e->rows_g);
          // (few rows: sixteen lanes each, the next step's first Newton pass beside this step's closing evaluation -- mjpl_rows.h,
          //  k_rrt_gen_project_ahead; a library without that kernel refuses and the chunk falls back below)
          const int G = (r->proj_g == 0 && r->ahead && active_bound <= r->ahead_lanes) ? (active_bound <= r->proj_waves_max ? 64 : 16) : rs.G;
          const int64_t rows_per_wave = G == 64 ? 1 : 64 / G;  // (64: sixteen lanes per row all the same, ONE row per wave)
          const unsigned rgridw = (unsigned)std::max<int64_t>(1, std::min<int64_t>(r->proj_waves_max, (active_bound + rows_per_wave - 1) / rows_per_wave));
          // (a library refuses -- nothing launched, -1 -- when it was built for another number of planning joints)
          trace_G = G; trace_grid = rgridw;
          launched = e->spec->gen_project(pk, G, st, rgridw, L, nplan, S, r->eps, par, r->pose->d_pi, r->pose->d_pd, r->d_qidx, r->d_qbase,
                                          r->d_isplan, r->d_lo, r->d_hi, Tgt, r->ln, r->cd, r->d_ctr);
          if (launched == -1 && (G == 16 || G == 64) && r->proj_g == 0) {  // (no such kernel in this library: eight lanes per row)
            const int64_t rpw = 64 / rs.G;
            const unsigned gw = (unsigned)std::max<int64_t>(1, std::min<int64_t>(r->proj_waves_max, (active_bound + rpw - 1) / rpw));
            trace_G = rs.G; trace_grid = gw;
            launched = e->spec->gen_project(pk, rs.G, st, gw, L, nplan, S, r->eps, par, r->pose->d_pi, r->pose->d_pd, r->d_qidx, r->d_qbase,
                                            r->d_isplan, r->d_lo, r->d_hi, Tgt, r->ln, r->cd, r->d_ctr);
          }
          if (launched != 0 && launched != -1) return fail(MJPL_E_HIP, "rrt: the generated extension kernel failed to launch");
        }
        if (launched != 0) {
          const unsigned pgrid = (unsigned)((active_bound + kPoseBlock - 1) / kPoseBlock);
          hipLaunchKernelGGL(k_rrt_gen_project, dim3(std::max(1u, pgrid)), dim3(kPoseBlock), plds, st, L, nplan, S, r->eps, par, r->pose->d_pi,
                             r->pose->d_pd, r->d_qidx, r->d_qbase, r->d_isplan, r->d_lo, r->d_hi, Tgt, r->ln, r->cd, r->d_ctr);
        }
      }
      if (!r->exact_counts && chunk < 2) {  // (the first two chunks: their own counts)
        if ((rc = rrt_read_ctr(r)) != MJPL_OK) return rc;
        if (r->h_ctr[RC_OVERFLOW] & 3) return fail(MJPL_E_CAPACITY, "rrt: buffer overrun in a projecting extension");
        if (r->h_ctr[RC_ACTIVE] == 0) break;
        active_bound = r->h_ctr[RC_ACTIVE];
      }
    } else {
      {
        auto gen = k_rrt_gen<0>;
        switch (nplan) {  // (the usual planning sets: their rows in registers)
#define MJPL_GEN_CASE(n) case n: gen = k_rrt_gen<n>; break;
          MJPL_GEN_CASE(1) MJPL_GEN_CASE(2) MJPL_GEN_CASE(3) MJPL_GEN_CASE(4) MJPL_GEN_CASE(5) MJPL_GEN_CASE(6) MJPL_GEN_CASE(7)
          MJPL_GEN_CASE(8) MJPL_GEN_CASE(9) MJPL_GEN_CASE(10) MJPL_GEN_CASE(12)
#undef MJPL_GEN_CASE
          default: break;
        }
        hipLaunchKernelGGL(gen, dim3(rgrid(L)), dim3(256), 0, st, L, nplan, S, r->max_steps, r->eps, r->d_lo, r->d_hi, Tgt, r->ln, r->cd, r->d_ctr);
      }
      if ((rc = rrt_read_ctr(r)) != MJPL_OK) return rc;
      if (r->h_ctr[RC_OVERFLOW] & 2) return fail(MJPL_E_CAPACITY, "rrt: more new nodes in one extension than the pending slab holds");
      if (r->h_ctr[RC_ACTIVE] == 0) {
        if (r->h_ctr[RC_OVERFLOW] & 1) {  // every waiting lane was refused: S is too large for the space left
          HIP_TRY(hipMemsetAsync(r->d_ctr + RC_OVERFLOW, 0, sizeof(int), st));
          HIP_TRY(hipMemsetAsync(r->d_ctr + RC_EDGES, 0, sizeof(int), st));  // (k_rrt_accept clears it otherwise)
          if (S == 1) return fail(MJPL_E_CAPACITY, "rrt: candidate buffer too small for one step per lane");
          S = 1;
          continue;
        }
        break;
      }
      if (r->h_ctr[RC_OVERFLOW] & 1) HIP_TRY(hipMemsetAsync(r->d_ctr + RC_OVERFLOW, 0, sizeof(int), st));
    }
    // (lanes that were refused candidate space have reserved slots all the same: RC_EDGES may exceed
    // the buffer, and only RC_ACTIVE says whether any lane emitted)
    const int E = projecting ? (int)std::min<int64_t>((int64_t)S * active_bound, r->cd.cap) : std::min(r->h_ctr[RC_EDGES], r->cd.cap);
    if (r->istep > 0)
      rc = launch_edges(e, r->cd.A, r->cd.B, E, r->istep, MJPL_AOS, 0, r->cd.valid, nullptr);
    else
      rc = launch_configs(e, r->cd.B, E, MJPL_AOS, r->cd.valid, nullptr);
    if (rc != MJPL_OK) return rc;
    hipLaunchKernelGGL(k_rrt_accept, dim3(rgrid(L)), dim3(256), 0, st, L, nplan, r->ln, r->cd, r->acc, r->d_ctr,
                       projecting ? r->h_ring + (chunk % 4) * kRingStride : (int *)nullptr, r->ring_seq0 + chunk,
                       projecting ? ((chunk + 1) & 1) : -1, r->max_steps, second, Tgt, r->carry[t]);
    steps_done += S;
    if (projecting) tr.chunk(st, chunk, active_bound, S, trace_G, trace_grid);
    // chunk sizes double: a chain of n steps costs O(log n) chunks and at most 2x its own checks
    if (!projecting && S < 64) {
      const int64_t room = (int64_t)r->cd.cap / std::max(1, r->h_ctr[RC_ACTIVE]);
      S = (int)std::max<int64_t>(1, std::min<int64_t>(2 * S, room));
    }
  }
  tr.mark(st, "extension chunks", chunks_done);
  // node order of the extension: lanes ascending, levels ascending within a lane
  {
    const int nsb = (L + kScanBlock - 1) / kScanBlock;
    hipLaunchKernelGGL(k_rrt_scan_sums, dim3(nsb), dim3(kScanBlock), 0, st, L, r->ln.cnt, r->d_scan);
    hipLaunchKernelGGL(k_rrt_scan_blocks, dim3(1), dim3(kScanBlock), 0, st, nsb, r->d_scan, r->d_ctr + (t == (r->round - 1) % 2 ? RC_NEWA : RC_NEWB));
    hipLaunchKernelGGL(k_rrt_scan_apply, dim3(nsb), dim3(kScanBlock), 0, st, L, r->ln.cnt, r->d_scan, r->ln.off);
  }
  if ((rc = rrt_read_ctr(r)) != MJPL_OK) return rc;
  if (r->h_ctr[RC_OVERFLOW] & 2) return fail(MJPL_E_CAPACITY, "rrt: more new nodes in one extension than the pending slab holds");
  {  // the validation launches of this extension never synchronised: a tail kernel that gave up waiting says so here
    int vstatus = 0;
    HIP_TRY(hipMemcpyAsync(&vstatus, e->d_status, sizeof(int), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (vstatus & (kStatusTailTimeout | kStatusFusedTimeout)) return fail(MJPL_E_HIP, "rrt: a kernel of a validation launch gave up waiting (status %d)", vstatus);
  }
  const int nacc = r->h_ctr[RC_ACC];
  *nnew = nacc;
  if (nacc > r->pendcap) return fail(MJPL_E_CAPACITY, "rrt: %d new nodes in one extension, pending slab holds %d", nacc, r->pendcap);
  if (nacc > 0)
    hipLaunchKernelGGL(k_rrt_place, dim3(rgrid(nacc)), dim3(256), 0, st, nacc, nplan, r->acc, r->ln.off, r->ln.near, r->d_pendQ[t],
                       r->d_pendpar[t]);
  hipLaunchKernelGGL(k_rrt_finish, dim3(rgrid(L)), dim3(256), 0, st, L, nplan, r->ln, second ? r->ln.refB : r->ln.refA,
                     second ? (double *)nullptr : r->ln.RA);
  HIP_TRY(hipGetLastError());
  tr.mark(st, "scan, place, finish", nacc);
  ring_guard.ok = true;
  return MJPL_OK;
}

}  // namespace

extern "C" {

int mjpl_comm_unique_id(void *id128) {
  if (!id128) return fail(MJPL_E_ARG, "mjpl_comm_unique_id: NULL");
  Rccl *nc = rccl();
  if (!nc) return fail(MJPL_E_HIP, "librccl.so could not be loaded");
  const int rc = nc->GetUniqueId(id128);
  if (rc != 0) return fail(MJPL_E_HIP, "ncclGetUniqueId failed: %s", nc->GetErrorString ? nc->GetErrorString(rc) : "?");
  return MJPL_OK;
}

int mjpl_comm_init(mjpl_engine *e, const void *id128, int32_t rank, int32_t world) {
  if (!e || !id128 || world < 1 || rank < 0 || rank >= world) return fail(MJPL_E_ARG, "mjpl_comm_init: bad argument");
  if (e->comm) return fail(MJPL_E_ARG, "mjpl_comm_init: the engine already has a communicator");
  Rccl *nc = rccl();
  if (!nc) return fail(MJPL_E_HIP, "librccl.so could not be loaded");
  HIP_TRY(hipSetDevice(e->device));
  Id128 id;
  memcpy(id.b, id128, sizeof(id.b));
  void *comm = nullptr;
  const int rc = nc->CommInitRank(&comm, world, id, rank);
  if (rc != 0) return fail(MJPL_E_HIP, "ncclCommInitRank failed: %s", nc->GetErrorString ? nc->GetErrorString(rc) : "?");
  e->comm = comm;
  e->comm_rank = rank;
  e->comm_world = world;
  return MJPL_OK;
}

int mjpl_comm_destroy(mjpl_engine *e) {
  if (!e) return fail(MJPL_E_ARG, "mjpl_comm_destroy: NULL engine");
  if (e->comm) {
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipStreamSynchronize(e->stream));
    Rccl *nc = rccl();
    if (nc) nc->CommDestroy(e->comm);
    e->comm = nullptr;
    e->comm_rank = 0;
    e->comm_world = 1;
  }
  return MJPL_OK;
}

int mjpl_allgather_dev(mjpl_engine *e, const void *dsend, void *drecv, size_t bytes_per_rank) {
  if (!e || (bytes_per_rank && (!dsend || !drecv))) return fail(MJPL_E_ARG, "mjpl_allgather_dev: bad argument");
  HIP_TRY(hipSetDevice(e->device));
  if (!e->comm) {  // no communicator: a world of one
    if (dsend != drecv && bytes_per_rank)
      HIP_TRY(hipMemcpyAsync(drecv, dsend, bytes_per_rank, hipMemcpyDeviceToDevice, e->stream));
    return MJPL_OK;
  }
  Rccl *nc = rccl();
  const int rc = nc->AllGather(dsend, drecv, bytes_per_rank, kNcclInt8, e->comm, e->stream);
  if (rc != 0) return fail(MJPL_E_HIP, "ncclAllGather failed: %s", nc->GetErrorString ? nc->GetErrorString(rc) : "?");
  return MJPL_OK;
}

void mjpl_rrt_destroy(mjpl_rrt *r) {
  if (!r) return;
  (void)hipSetDevice(r->e->device);
  (void)hipStreamSynchronize(r->e->stream);
  if (r->side) { (void)hipStreamSynchronize(r->side); (void)hipStreamDestroy(r->side); }
  if (r->ev_tail) (void)hipEventDestroy(r->ev_tail);
  if (r->ev_near) (void)hipEventDestroy(r->ev_near);
  if (r->ev_pre) (void)hipEventDestroy(r->ev_pre);
  if (r->side_nn.nn) (void)hipFree(r->side_nn.nn);
  if (r->side_nn.nn16) (void)hipFree(r->side_nn.nn16);
  if (r->side_nn.tmp) (void)hipFree(r->side_nn.tmp);
  for (void *p : r->owned) (void)hipFree(p);
  if (r->h_ctr) (void)hipHostFree(r->h_ctr);
  if (r->h_ring) (void)hipHostFree(r->h_ring);
  if (r->h_heads) (void)hipHostFree(r->h_heads);
  if (r->h_myhead) (void)hipHostFree(r->h_myhead);
  if (r->d_gather) (void)hipFree(r->d_gather);
  delete r;
}

int mjpl_rrt_create(mjpl_engine *e, const mjpl_rrt_desc *d, mjpl_rrt **out) {
  if (!e || !d || !out) return fail(MJPL_E_ARG, "mjpl_rrt_create: NULL argument");
  *out = nullptr;
  const int nplan = (int)e->qidx.size();
  if (nplan < 1 || nplan > kRrtMaxPlan) return fail(MJPL_E_CAPACITY, "rrt: %d planning columns (1..%d supported)", nplan, kRrtMaxPlan);
  if (d->lanes < 1 || d->capacity < 2 || !d->lo || !d->hi) return fail(MJPL_E_ARG, "mjpl_rrt_create: bad sizes");
  if (!(d->epsilon > 0.0)) return fail(MJPL_E_ARG, "`epsilon` must be > 0.0");
  if (!(d->goal_bias >= 0.0 && d->goal_bias <= 1.0)) return fail(MJPL_E_ARG, "`goal_biasing_probability` must be within [0.0, 1.0].");
  if (d->pose && d->pose->e != e) return fail(MJPL_E_ARG, "the pose handle belongs to another engine");
  if (d->pose && pose_lds(d->pose) + (size_t)kPoseBlock * sizeof(double) * (size_t)e->m.nq > 64 * 1024)
    return fail(MJPL_E_CAPACITY, "rrt: %d qpos + %d chain joints exceed the LDS budget of the projecting extension", e->m.nq, d->pose->nj);
  HIP_TRY(hipSetDevice(e->device));
  std::unique_ptr<mjpl_rrt, void (*)(mjpl_rrt *)> r(new mjpl_rrt(), mjpl_rrt_destroy);
  r->e = e;
  r->pose = d->pose;
  r->nplan = nplan;
  r->nq = e->m.nq;
  r->L = d->lanes;
  r->cap = d->capacity;
  e->nn_reserve_nodes = std::max<int64_t>(e->nn_reserve_nodes, std::min<int64_t>(d->capacity, (int64_t)1 << 23));
  r->eps = d->epsilon;
  r->istep = d->interval_step;
  r->pgoal = d->goal_bias;
  r->seed = d->seed;
  const int L = r->L;
  int rc = MJPL_OK;
#define RA(ptr, count) if ((rc = rrt_alloc(r.get(), &(ptr), (size_t)(count))) != MJPL_OK) return rc
  for (int t = 0; t < 2; t++) { RA(r->d_Q[t], (size_t)nplan * r->cap); RA(r->d_parent[t], r->cap); }
  RA(r->d_lo, nplan); RA(r->d_hi, nplan); RA(r->d_qinit, nplan); RA(r->d_qbase, r->nq); RA(r->d_qidx, nplan); RA(r->d_isplan, r->nq);
  RA(r->ln.T, (size_t)nplan * L); RA(r->ln.C, (size_t)nplan * L); RA(r->ln.RA, (size_t)nplan * L);
  RA(r->ln.near, L); RA(r->ln.refA, L); RA(r->ln.refB, L); RA(r->ln.on, L); RA(r->ln.act, L); RA(r->ln.cnt, L); RA(r->ln.off, L);
  RA(r->ln.gfirst, L); RA(r->ln.gcount, L); RA(r->ln.gend, L); RA(r->ln.goal, L);
  RA(r->ln.list[0], L); RA(r->ln.list[1], L);
  r->cd.cap = (int)std::min<int64_t>(std::max<int64_t>(4 * (int64_t)L, 1 << 16), (int64_t)1 << 27);
  RA(r->cd.A, (size_t)r->cd.cap * nplan); RA(r->cd.B, (size_t)r->cd.cap * nplan); RA(r->cd.lane, r->cd.cap); RA(r->cd.level, r->cd.cap);
  RA(r->cd.valid, r->cd.cap); RA(r->cd.rule, r->cd.cap); RA(r->cd.reach, r->cd.cap);
  r->pendcap = (int)std::min<int64_t>(r->cap, d->max_new_per_round > 0 ? d->max_new_per_round : std::max<int64_t>(128 * (int64_t)L, 1 << 16));
  r->acc.cap = r->pendcap;
  RA(r->acc.Q, (size_t)r->acc.cap * nplan); RA(r->acc.lane, r->acc.cap); RA(r->acc.level, r->acc.cap);
  for (int t = 0; t < 2; t++) { RA(r->d_pendQ[t], (size_t)r->pendcap * nplan); RA(r->d_pendpar[t], r->pendcap); }
  RA(r->d_ctr, RC_SIZE); RA(r->d_heads, 8 * 1024); RA(r->d_path, (size_t)nplan * 65536);
  RA(r->d_first, 1);  // re-allocated by reset for the number of goals
  RA(r->d_scan, (size_t)(L + kScanBlock - 1) / kScanBlock);
  RA(r->d_cf, r->cap); RA(r->d_coff, L);
  HIP_TRY(hipMemset(r->d_cf, 0x7f, (size_t)r->cap * sizeof(int32_t)));
  HIP_TRY(hipMemset(r->d_coff, 0, (size_t)L));
  if (d->max_steps_per_round < 0) return fail(MJPL_E_ARG, "`max_steps_per_round` must be >= 0 (0: no cap)");
  r->max_steps = d->max_steps_per_round;
  if (r->max_steps > 0)
    for (int t = 0; t < 2; t++) {
      RA(r->carry[t].flag, L); RA(r->carry[t].T, (size_t)nplan * L); RA(r->carry[t].goal, L); RA(r->carry[t].node, L);
    }
  // (the planner's switches: options of the engine -- "rrt_exact_counts", "rrt_early_nn", ... -- as they stand now)
  r->exact_counts = e->rrt_exact_counts; r->early_nn = e->rrt_early_nn; r->early_lanes = e->rrt_early_lanes;
  r->early_min_nodes = e->rrt_early_min_nodes; r->early_next = e->rrt_early_next;
  if (!d->pose) r->early_nn = 0;  // (extensions without a projecting constraint have no such tail)
  if (r->early_nn) {
    RA(r->d_RAe, (size_t)nplan * L); RA(r->d_near_e, L); RA(r->d_late_pos, L); RA(r->d_early, L);
    RA(r->d_late_near, std::max(r->early_lanes, 1)); RA(r->d_late_q, (size_t)nplan * r->early_lanes);
    HIP_TRY(hipStreamCreateWithFlags(&r->side, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&r->ev_tail, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&r->ev_near, hipEventDisableTiming));
    if (r->early_next) {
      RA(r->d_Tn, (size_t)nplan * L); RA(r->d_goal_n, L); RA(r->d_first_n, 1); RA(r->d_pre_idx, L); RA(r->d_pre_d2, L);
      HIP_TRY(hipEventCreateWithFlags(&r->ev_pre, hipEventDisableTiming));
    }
  }
#undef RA
  HIP_TRY(hipHostMalloc((void **)&r->h_ctr, RC_SIZE * sizeof(int)));
  HIP_TRY(hipHostMalloc((void **)&r->h_ring, 4 * kRingStride * sizeof(int)));
  memset(r->h_ring, 0, 4 * kRingStride * sizeof(int));
  r->proj_steps_max = e->rrt_proj_steps; r->proj_slots = e->rrt_proj_slots; r->proj_g = e->rrt_proj_g;
  r->ahead = e->rrt_ahead; r->ahead_lanes = e->rrt_ahead_lanes; r->proj_waves_max = e->rrt_proj_waves;
  HIP_TRY(hipHostMalloc((void **)&r->h_heads, 8 * 1024 * sizeof(int)));
  HIP_TRY(hipHostMalloc((void **)&r->h_myhead, 8 * sizeof(int)));
  std::vector<uint8_t> isplan(r->nq, 0);
  for (int c : e->qidx) isplan[c] = 1;
  HIP_TRY(hipMemcpy(r->d_lo, d->lo, nplan * sizeof(double), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(r->d_hi, d->hi, nplan * sizeof(double), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(r->d_qbase, e->qbase.data(), r->nq * sizeof(double), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(r->d_qidx, e->qidx.data(), nplan * sizeof(int), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(r->d_isplan, isplan.data(), r->nq, hipMemcpyHostToDevice));
  *out = r.release();
  return MJPL_OK;
}

int mjpl_rrt_reset(mjpl_rrt *r, const double *q_init, const double *q_goals, int32_t ngoal, uint64_t seed) {
  if (!r || !q_init || !q_goals || ngoal < 1) return fail(MJPL_E_ARG, "mjpl_rrt_reset: bad argument");
  if (ngoal >= r->cap) return fail(MJPL_E_CAPACITY, "rrt: more goals than node capacity");
  mjpl_engine *e = r->e;
  HIP_TRY(hipSetDevice(e->device));
  HIP_TRY(hipStreamSynchronize(e->stream));
  const int nplan = r->nplan;
  // start tree: node 0 = q_init.  goal tree: nodes 0..ngoal-1 = the goals (their common parent, the
  // reference's sink node at +inf, is implicit: parent -1; rrt.py:179-188)
  std::vector<double> col0((size_t)nplan), colg((size_t)nplan * ngoal);
  for (int c = 0; c < nplan; c++) {
    HIP_TRY(hipMemcpy(r->d_Q[0] + (int64_t)c * r->cap, q_init + c, sizeof(double), hipMemcpyHostToDevice));
    for (int g = 0; g < ngoal; g++) colg[(size_t)c * ngoal + g] = q_goals[(size_t)g * nplan + c];
    HIP_TRY(hipMemcpy(r->d_Q[1] + (int64_t)c * r->cap, &colg[(size_t)c * ngoal], ngoal * sizeof(double), hipMemcpyHostToDevice));
  }
  std::vector<int32_t> minus(ngoal, -1);
  HIP_TRY(hipMemcpy(r->d_parent[0], minus.data(), sizeof(int32_t), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(r->d_parent[1], minus.data(), ngoal * sizeof(int32_t), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(r->d_qinit, q_init, nplan * sizeof(double), hipMemcpyHostToDevice));
  if (r->side) HIP_TRY(hipStreamSynchronize(r->side));
  for (int t = 0; t < 2; t++)
    if (r->carry[t].flag) HIP_TRY(hipMemset(r->carry[t].flag, 0, (size_t)r->L));
  HIP_TRY(hipMemset(r->d_cf, 0x7f, (size_t)r->cap * sizeof(int32_t)));  // (a round that failed half-way leaves its claims)
  r->pre_round = 0;
  r->early_on = false;
  if (ngoal > r->first_cap) {
    int rc = rrt_alloc(r, &r->d_first, (size_t)ngoal);
    if (rc != MJPL_OK) return rc;
    if (r->early_next && r->early_nn && (rc = rrt_alloc(r, &r->d_first_n, (size_t)ngoal)) != MJPL_OK) return rc;
    r->first_cap = ngoal;
  }
  r->n[0] = 1;
  r->n[1] = ngoal;
  r->ngoal = ngoal;
  r->round = 0;
  r->seed = seed;
  r->conn_start = r->conn_goal = -1;
  r->in_round = false;
  return MJPL_OK;
}

namespace {

int rrt_rank(const mjpl_rrt *r) { return r->world_set ? r->rank : (r->e->comm ? r->e->comm_rank : 0); }
int rrt_world(const mjpl_rrt *r) { return r->world_set ? r->world : (r->e->comm ? r->e->comm_world : 1); }

// First half of a round, asynchronous: sample, extend the growing tree, extend the other towards what was
// reached, find this rank's connection; leaves the rank's header in d_heads[8 * rank ..] and its new
// nodes in the pending slabs.  A failure that only THIS rank sees (a full pending slab, a HIP error) must
// not keep it out of the exchange the other ranks are about to enter: it travels in the header
// (head[6]) and every rank returns it from the second half.
int rrt_begin(mjpl_rrt *r, int32_t request_stop) {
  if (r->n[0] < 1 || r->n[1] < 1) return fail(MJPL_E_ARG, "mjpl_rrt_round: call mjpl_rrt_reset first");
  if (r->in_round) return fail(MJPL_E_ARG, "mjpl_rrt_round_begin: the previous round has not been finished");
  mjpl_engine *e = r->e;
  HIP_TRY(hipSetDevice(e->device));
  hipStream_t st = e->stream;
  const int L = r->L, nplan = r->nplan;
  const int world = rrt_world(r), rank = rrt_rank(r);
  if (world > 1024) return fail(MJPL_E_CAPACITY, "rrt: world size %d not supported", world);  // (the same on every rank)
  r->round++;
  r->early_on = false;
  const int grow = (r->round - 1) % 2, other = 1 - grow;  // tree swap every round (rrt.py:234-235)
  if (r->pre_round == r->round) {
    // the round before drew this round's targets (and looked them up in the nodes there were): its buffers become the round's
    HIP_TRY(hipStreamWaitEvent(st, r->ev_pre, 0));
    std::swap(r->ln.T, r->d_Tn);
    std::swap(r->ln.goal, r->d_goal_n);
    std::swap(r->d_first, r->d_first_n);
    hipLaunchKernelGGL(k_rrt_round_init, dim3(rgrid((int)RC_SIZE)), dim3(256), 0, st, r->d_ctr, r->d_first, 0);
  } else {
    r->pre_round = 0;
    const int nf = std::max(r->ngoal, 1);
    hipLaunchKernelGGL(k_rrt_round_init, dim3(rgrid(std::max(nf, (int)RC_SIZE))), dim3(256), 0, st, r->d_ctr, r->d_first, nf);
    hipLaunchKernelGGL(k_rrt_sample, dim3(rgrid(L)), dim3(256), 0, st, L, nplan, rrt_key(r->seed, (uint64_t)rank, (uint64_t)r->round), r->pgoal,
                       grow, r->ngoal, r->d_lo, r->d_hi, r->d_qinit, r->d_Q[1], r->cap, r->ln, r->d_first, r->carry[grow]);
  }
  int newA = 0, newB = 0;
  int rc = rrt_extend(r, grow, r->ln.T, 0, &newA);
  if (rc == MJPL_OK) hipLaunchKernelGGL(k_rrt_conn_claim, dim3(rgrid(L)), dim3(256), 0, st, L, r->ln, r->carry[grow], r->d_cf);
  if (rc == MJPL_OK) rc = rrt_extend(r, other, r->ln.RA, 1, &newB);
  if (rc == MJPL_OK) {
    hipLaunchKernelGGL(k_rrt_connect, dim3(rgrid(L)), dim3(256), 0, st, L, nplan, r->ln, r->d_ctr, r->carry[grow], r->d_cf, (const uint8_t *)r->d_coff);
    hipLaunchKernelGGL(k_rrt_header, dim3(1), dim3(1), 0, st, r->ln, r->d_ctr, grow, request_stop ? 1 : 0, r->d_heads + 8 * rank);
    if (hipGetLastError() != hipSuccess) rc = fail(MJPL_E_HIP, "rrt: a kernel of the round failed to launch");
  }
  r->local_err.clear();
  if (rc != MJPL_OK) {
    r->pre_round = 0;
    r->local_err = g_err;
    const int h[8] = {0, 0, 0x7fffffff, 0, 0, request_stop ? 1 : 0, rc, 0};
    memcpy(r->h_myhead, h, sizeof(h));
    if (hipMemcpyAsync(r->d_heads + 8 * rank, r->h_myhead, sizeof(h), hipMemcpyHostToDevice, st) != hipSuccess)
      return rc;  // the device itself is gone: nothing can be relayed
  }
  r->in_round = true;
  return MJPL_OK;
}

// Second half: `heads` = the world's headers in rank order (host); allQ[p] / allP[p] = device buffers with
// every rank's slab of pass p (0: the tree that grew, 1: the other) at a stride of stride[p] rows.
int rrt_finish(mjpl_rrt *r, const int *heads, const char *const allQ[2], const char *const allP[2], const int64_t stride[2],
               mjpl_rrt_round_info *info) {
  mjpl_engine *e = r->e;
  hipStream_t st = e->stream;
  const int nplan = r->nplan;
  const int world = rrt_world(r), rank = rrt_rank(r);
  const int grow = (r->round - 1) % 2, other = 1 - grow;
  r->in_round = false;
  memset(info, 0, sizeof(*info));
  info->round = r->round;
  for (int k = 0; k < world; k++)
    if (heads[8 * k + 6] != 0)
      return fail(heads[8 * k + 6], "rrt: rank %d failed in round %d: %s", k, r->round,
                  k == rank ? r->local_err.c_str() : "(its own mjpl_last_error has the reason)");
  int64_t totA = 0, totB = 0;
  for (int k = 0; k < world; k++) {
    if (heads[8 * k + 5]) info->stop_requested = 1;
    if (heads[8 * k] < 0 || heads[8 * k + 1] < 0 || heads[8 * k] > stride[0] || heads[8 * k + 1] > stride[1])
      return fail(MJPL_E_ARG, "rrt: rank %d's header counts (%d, %d) do not fit the gathered slabs", k, heads[8 * k], heads[8 * k + 1]);
    totA += heads[8 * k];
    totB += heads[8 * k + 1];
  }
  if (r->n[grow] + totA > r->cap || r->n[other] + totB > r->cap) return fail(MJPL_E_CAPACITY, "rrt: node capacity %lld exhausted", (long long)r->cap);
  int win_rank = -1;
  for (int k = 0; k < world && win_rank < 0; k++)
    if (heads[8 * k + 2] != 0x7fffffff) win_rank = k;
  int baseA_of_winner = 0, baseB_of_winner = 0;
  for (int pass = 0; pass < 2; pass++) {  // pass 0: the tree that grew; 1: the other
    const int t = pass == 0 ? grow : other;
    int base = r->n[t];
    const size_t rowb = (size_t)stride[pass] * nplan * sizeof(double), parb = (size_t)stride[pass] * sizeof(int32_t);
    for (int k = 0; k < world; k++) {
      const int cnt = heads[8 * k + pass];
      if (k == win_rank) (pass == 0 ? baseA_of_winner : baseB_of_winner) = base;
      if (k == rank && pass == 0 && r->max_steps > 0)
        hipLaunchKernelGGL(k_rrt_carry_node, dim3(rgrid(r->L)), dim3(256), 0, st, r->L, r->carry[grow], r->ln.refA, base);
      if (cnt > 0) {
        if (!allQ[pass] || !allP[pass]) return fail(MJPL_E_ARG, "rrt: no gathered slab for pass %d", pass);
        hipLaunchKernelGGL(k_rrt_merge, dim3(rgrid(cnt)), dim3(256), 0, st, cnt, nplan, (const double *)(allQ[pass] + (size_t)k * rowb),
                           (const int32_t *)(allP[pass] + (size_t)k * parb), r->d_Q[t], r->d_parent[t], r->cap, base);
      }
      base += cnt;
    }
    r->n[t] = base;
  }
  HIP_TRY(hipGetLastError());
  info->new_nodes[grow] = (int32_t)totA;
  info->new_nodes[other] = (int32_t)totB;
  info->nodes[0] = r->n[0];
  info->nodes[1] = r->n[1];
  info->connected = win_rank >= 0 ? 1 : 0;
  if (win_rank >= 0) {
    // header refs: [3] in the start tree, [4] in the goal tree; pending indices are relative to
    // the winner's block of that tree (pass 0 = the tree that grew)
    const int ra = heads[8 * win_rank + 3], rb = heads[8 * win_rank + 4];
    const int base_start = grow == 0 ? baseA_of_winner : baseB_of_winner;
    const int base_goal = grow == 0 ? baseB_of_winner : baseA_of_winner;
    r->conn_start = ra >= 0 ? ra : base_start + (-1 - ra);
    r->conn_goal = rb >= 0 ? rb : base_goal + (-1 - rb);
    info->conn_start = r->conn_start;
    info->conn_goal = r->conn_goal;
    info->conn_rank = win_rank;
  }
  return MJPL_OK;
}

}  // namespace

int mjpl_rrt_set_world(mjpl_rrt *r, int32_t rank, int32_t world) {
  if (!r || world < 1 || world > 1024 || rank < 0 || rank >= world) return fail(MJPL_E_ARG, "mjpl_rrt_set_world: bad rank / world");
  if (r->in_round) return fail(MJPL_E_ARG, "mjpl_rrt_set_world: a round is in flight");
  r->rank = rank;
  r->world = world;
  r->world_set = true;
  r->pre_round = 0;  // (draws made ahead for the next round were keyed by the rank there was)
  return MJPL_OK;
}

int mjpl_rrt_round_begin(mjpl_rrt *r, int32_t request_stop, int32_t *head) {
  if (!r || !head) return fail(MJPL_E_ARG, "mjpl_rrt_round_begin: NULL argument");
  int rc = rrt_begin(r, request_stop);
  if (rc != MJPL_OK) return rc;
  hipStream_t st = r->e->stream;
  HIP_TRY(hipMemcpyAsync(r->h_myhead, r->d_heads + 8 * rrt_rank(r), 8 * sizeof(int), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  memcpy(head, r->h_myhead, 8 * sizeof(int));
  return MJPL_OK;
}

int mjpl_rrt_round_slabs(mjpl_rrt *r, int32_t pass, const void **drows, const void **dparents) {
  if (!r || pass < 0 || pass > 1 || !drows || !dparents) return fail(MJPL_E_ARG, "mjpl_rrt_round_slabs: bad argument");
  if (!r->in_round) return fail(MJPL_E_ARG, "mjpl_rrt_round_slabs: no round in flight");
  const int grow = (r->round - 1) % 2;
  const int t = pass == 0 ? grow : 1 - grow;
  *drows = r->d_pendQ[t];
  *dparents = r->d_pendpar[t];
  return MJPL_OK;
}

int mjpl_rrt_round_finish(mjpl_rrt *r, const int32_t *heads, const void *const *drows_all, const void *const *dparents_all,
                          const int32_t *stride_rows, mjpl_rrt_round_info *info) {
  if (!r || !heads || !drows_all || !dparents_all || !stride_rows || !info) return fail(MJPL_E_ARG, "mjpl_rrt_round_finish: NULL argument");
  if (!r->in_round) return fail(MJPL_E_ARG, "mjpl_rrt_round_finish: no round in flight");
  HIP_TRY(hipSetDevice(r->e->device));
  const char *q[2] = {(const char *)drows_all[0], (const char *)drows_all[1]};
  const char *p[2] = {(const char *)dparents_all[0], (const char *)dparents_all[1]};
  const int64_t stride[2] = {stride_rows[0], stride_rows[1]};
  return rrt_finish(r, heads, q, p, stride, info);
}

int mjpl_rrt_round(mjpl_rrt *r, int32_t request_stop, mjpl_rrt_round_info *info) {
  if (!r || !info) return fail(MJPL_E_ARG, "mjpl_rrt_round: NULL argument");
  mjpl_engine *e = r->e;
  const int world = rrt_world(r), rank = rrt_rank(r);
  if (world > 1 && !e->comm)
    return fail(MJPL_E_ARG, "mjpl_rrt_round: rank %d of %d without a communicator (mjpl_comm_init), use round_begin / round_finish", rank, world);
  hipStream_t st = e->stream;
  RrtTrace tr(e);
  int rc = rrt_begin(r, request_stop);
  if (rc != MJPL_OK) return rc;
  tr.mark(st, "round_begin");
  const int nplan = r->nplan;
  const int grow = (r->round - 1) % 2, other = 1 - grow;
  // ---- exchange: headers of all ranks, then the new-node slabs padded to the round's largest count
  auto bail = [&](int code) { r->in_round = false; return code; };
  rc = mjpl_allgather_dev(e, r->d_heads + 8 * rank, r->d_heads, 8 * sizeof(int));
  if (rc != MJPL_OK) return bail(rc);
  if (hipMemcpyAsync(r->h_heads, r->d_heads, (size_t)world * 8 * sizeof(int), hipMemcpyDeviceToHost, st) != hipSuccess ||
      hipStreamSynchronize(st) != hipSuccess)
    return bail(fail(MJPL_E_HIP, "rrt: reading the exchange headers failed"));
  int mx[2] = {0, 0};
  bool failed = false;
  for (int k = 0; k < world; k++) {
    failed = failed || r->h_heads[8 * k + 6] != 0;
    mx[0] = std::max(mx[0], r->h_heads[8 * k]);
    mx[1] = std::max(mx[1], r->h_heads[8 * k + 1]);
  }
  const char *allQ[2] = {nullptr, nullptr}, *allP[2] = {nullptr, nullptr};
  const int64_t stride[2] = {mx[0], mx[1]};
  if (!failed) {  // (every rank sees the same headers: all gather, or none does)
    const size_t rowb[2] = {(size_t)mx[0] * nplan * sizeof(double), (size_t)mx[1] * nplan * sizeof(double)};
    const size_t parb[2] = {(size_t)mx[0] * sizeof(int32_t), (size_t)mx[1] * sizeof(int32_t)};
    if (e->comm) {
      const size_t need = (size_t)world * (rowb[0] + parb[0] + rowb[1] + parb[1]);
      if (need > r->gather_bytes) {
        if (r->d_gather && hipFree(r->d_gather) != hipSuccess) return bail(fail(MJPL_E_HIP, "hipFree failed"));
        r->d_gather = nullptr; r->gather_bytes = 0;
        if (hipMalloc((void **)&r->d_gather, need) != hipSuccess) return bail(fail(MJPL_E_HIP, "rrt: no memory for the gathered slabs"));
        r->gather_bytes = need;
      }
      char *at = r->d_gather;
      for (int pass = 0; pass < 2; pass++) {
        const int t = pass == 0 ? grow : other;
        if (mx[pass] == 0) continue;
        char *gQ = at, *gP = at + (size_t)world * rowb[pass];
        at = gP + (size_t)world * parb[pass];
        if ((rc = mjpl_allgather_dev(e, r->d_pendQ[t], gQ, rowb[pass])) != MJPL_OK) return bail(rc);
        if ((rc = mjpl_allgather_dev(e, r->d_pendpar[t], gP, parb[pass])) != MJPL_OK) return bail(rc);
        allQ[pass] = gQ; allP[pass] = gP;
      }
    } else {  // a world of one: the pending slabs are the gathered ones
      for (int pass = 0; pass < 2; pass++) {
        const int t = pass == 0 ? grow : other;
        allQ[pass] = (const char *)r->d_pendQ[t];
        allP[pass] = (const char *)r->d_pendpar[t];
      }
    }
  }
  tr.mark(st, "exchange");
  rc = rrt_finish(r, r->h_heads, allQ, allP, stride, info);
  tr.mark(st, "round_finish");
  return rc;
}

int mjpl_rrt_path(mjpl_rrt *r, double *path, int32_t maxlen, int32_t *len) {
  if (!r || !path || !len || maxlen < 2) return fail(MJPL_E_ARG, "mjpl_rrt_path: bad argument");
  if (r->conn_start < 0) return fail(MJPL_E_ARG, "mjpl_rrt_path: no connection has been found");
  mjpl_engine *e = r->e;
  HIP_TRY(hipSetDevice(e->device));
  const int nplan = r->nplan, room = 65536;
  std::vector<double> a, b;
  for (int t = 0; t < 2; t++) {
    // the junction configuration is in both trees: the goal-tree half starts at its parent
    hipLaunchKernelGGL(k_rrt_walk, dim3(1), dim3(1), 0, e->stream, r->d_Q[t], r->d_parent[t], r->cap, nplan,
                       t == 0 ? r->conn_start : r->conn_goal, room, r->d_path, r->d_ctr);
    int n = 0;
    HIP_TRY(hipMemcpyAsync(&n, r->d_ctr, sizeof(int), hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    if (n < 0) return fail(MJPL_E_CAPACITY, "rrt: path longer than %d nodes", room);
    std::vector<double> &v = t == 0 ? a : b;
    v.resize((size_t)n * nplan);
    HIP_TRY(hipMemcpy(v.data(), r->d_path, v.size() * sizeof(double), hipMemcpyDeviceToHost));
  }
  const int na = (int)(a.size() / nplan), nb = (int)(b.size() / nplan);
  const int total = na + nb - 1;
  *len = total;
  if (total > maxlen) return fail(MJPL_E_CAPACITY, "mjpl_rrt_path: the path has %d waypoints, the buffer holds %d", total, maxlen);
  for (int i = 0; i < na; i++) memcpy(path + (size_t)i * nplan, &a[(size_t)(na - 1 - i) * nplan], nplan * sizeof(double));
  for (int i = 1; i < nb; i++) memcpy(path + (size_t)(na + i - 1) * nplan, &b[(size_t)i * nplan], nplan * sizeof(double));
  return MJPL_OK;
}

int mjpl_rrt_get_tree(mjpl_rrt *r, int32_t tree, double *Q, int32_t *parent, int64_t maxn, int64_t *n) {
  if (!r || tree < 0 || tree > 1 || !n) return fail(MJPL_E_ARG, "mjpl_rrt_get_tree: bad argument");
  mjpl_engine *e = r->e;
  HIP_TRY(hipSetDevice(e->device));
  HIP_TRY(hipStreamSynchronize(e->stream));
  *n = r->n[tree];
  if (!Q && !parent) return MJPL_OK;
  if (maxn < r->n[tree]) return fail(MJPL_E_CAPACITY, "mjpl_rrt_get_tree: %d nodes, the buffers hold %lld", r->n[tree], (long long)maxn);
  const int cnt = r->n[tree];
  if (Q) {  // rows [n][nplan]
    std::vector<double> col(cnt);
    for (int c = 0; c < r->nplan; c++) {
      HIP_TRY(hipMemcpy(col.data(), r->d_Q[tree] + (int64_t)c * r->cap, (size_t)cnt * sizeof(double), hipMemcpyDeviceToHost));
      for (int i = 0; i < cnt; i++) Q[(size_t)i * r->nplan + c] = col[i];
    }
  }
  if (parent) HIP_TRY(hipMemcpy(parent, r->d_parent[tree], (size_t)cnt * sizeof(int32_t), hipMemcpyDeviceToHost));
  return MJPL_OK;
}

// the lanes of the most recent round (tests): targets [L][nplan], on flags
int mjpl_rrt_get_lanes(mjpl_rrt *r, double *targets, uint8_t *on) {
  if (!r) return fail(MJPL_E_ARG, "mjpl_rrt_get_lanes: NULL");
  mjpl_engine *e = r->e;
  HIP_TRY(hipSetDevice(e->device));
  HIP_TRY(hipStreamSynchronize(e->stream));
  if (targets) {
    std::vector<double> col(r->L);
    for (int c = 0; c < r->nplan; c++) {
      HIP_TRY(hipMemcpy(col.data(), r->ln.T + (int64_t)c * r->L, (size_t)r->L * sizeof(double), hipMemcpyDeviceToHost));
      for (int i = 0; i < r->L; i++) targets[(size_t)i * r->nplan + c] = col[i];
    }
  }
  if (on) HIP_TRY(hipMemcpy(on, r->ln.on, (size_t)r->L, hipMemcpyDeviceToHost));
  return MJPL_OK;
}

}  // extern "C"
