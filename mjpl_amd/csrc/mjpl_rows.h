// mjpl_rows.h -- the Newton-chain kernels around a GENERATED chain (mjpl_amd/specialise.py: generate_pose): the
// PoseConstraint projection (pose_constraint.py:78-147), the planner's projecting extension chunk
// (planning/utils.py:139-164) and the batched IK seeds (inverse_kinematics/mink_ik_solver.py:91-115), with the waves
// kept full.  Included at the end of mjpl_project.h; per-model libraries instantiate these around their PoseSpec<k>.
//
// What round 4's kernels did (one row per lane, grid = rows / 64, a wave alive until its slowest row is through) left
// 7 .. 21 of 64 lanes live per vector instruction (profiles/r04h_pmc_*).  Two changes, results unchanged bit for bit:
//
// * ROWS, NOT LANES, ARE THE UNIT, and a wave refills.  Every wave owns a contiguous range of the launch's rows (of the
//   packed active-lane list, for the planner) and works on 64 / G of them at a time; a row that ends -- converged,
//   rejected, arrived -- hands its lanes to the next row of the range at the top of the iteration loop.  All rows of a
//   wave are always at the same point of the SAME loop body (one Newton iteration), so there is no divergence to pay
//   for beyond the refill itself; a wave retires when its range is empty.
// * A ROW MAY HAVE G = 4 OR 8 LANES (launches with few rows: the tail of an extension, an IK batch, a scalar call), which
//   split what splits: the seven hinges' half-angle sines and cosines one joint per lane, the inverse trigonometric
//   functions of the displacement and of the Jacobian's E_rpy one per lane, the site's sines / cosines one per lane --
//   about a third of a Newton step's vector instructions.  Every value is computed by the statements, in the order,
//   of the G = 1 code (and of the interpreting kernel): only WHICH lane executes them changes; the lanes of a row
//   exchange them through 26 doubles of LDS.  The chain itself, the 6x6 and the rules run on all lanes of the row
//   (the same values eight times: no exchange would be cheaper than the arithmetic).
//
// Registers: the generated chain with the joints' axes / anchors / columns, the 6x6 and a row's state wants ~470 registers
// (256 VGPRs + AGPRs as the compiler's spill space, no scratch traffic): ONE wave per SIMD, as round 4's generated kernels
// already were; held to 256 the kernels spill 200 .. 600 registers.  So the launches are sized for 1 024 resident waves
// (mjpl_hip.hip: rows_shape; mjpl_rrt.h: rrt_extend) and every wave owns as many rows as that leaves it.
#pragma once

namespace mjpl {

// Where planning column k sits in qpos: read from the planner's table (PlanQRuntime), or -- a model's library knows its
// program's planning joints -- a constant of the generated source (struct SpecPlanQ, mjpl_amd/specialise.py): the loops that
// scatter a lane's planning coordinates into a configuration and gather them back are unrolled over constants then, and the
// NP x NQ compare-and-select chains they were (~350 vector instructions per step of an extension) fold away.
struct PlanQRuntime {
  static __device__ __forceinline__ int at(int k, const int *__restrict__ qidx) { return qidx[k]; }
};

constexpr int kXchStride = 26;  // doubles per row of a wave's exchange area (rows land on disjoint LDS banks: 52 words apart)

__device__ __forceinline__ void row_fence() {  // LDS written by some lanes of the wave, read by others
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

template <int G>
struct RowId {
  static_assert(G == 1 || G == 4 || G == 8, "a row is one lane, four or eight");
  static constexpr int kRows = 64 / G;
  static constexpr unsigned long long kLeaders = G == 1 ? ~0ull : (G == 4 ? 0x1111111111111111ull : 0x0101010101010101ull);
  int lane, row, g;
  __device__ __forceinline__ RowId() : lane((int)(threadIdx.x & 63)), row(lane / G), g(lane % G) {}
  // the rows for which p holds (p: the same on all lanes of a row), one bit per row at its first lane
  __device__ __forceinline__ unsigned long long rows(bool p) const { return __ballot(p) & kLeaders; }
  // how many rows of `m` lie below this one
  __device__ __forceinline__ int rank(unsigned long long m) const { return __popcll(m & ((1ull << (row * G)) - 1ull)); }
};

// sines and cosines of the chain's half angles: one joint after the other (G == 1), or joint k on lane k % G of the row
template <class PS, int G>
__device__ __forceinline__ void hinge_sincos(const double (&qv)[PS::kNQ], double (&sn)[PS::kNJ > 0 ? PS::kNJ : 1],
                                             double (&cs)[PS::kNJ > 0 ? PS::kNJ : 1], double *x, int g) {
  constexpr int NJ = PS::kNJ;
  if constexpr (G == 1) {
#pragma unroll
    for (int k = 0; k < NJ; k++) {
      sn[k] = 0.0; cs[k] = 1.0;
      if (PS::jtype(k) == JT_HINGE) sincos_pi2(PS::half_angle(k, qv), &sn[k], &cs[k]);
    }
  } else {
    static_assert(2 * NJ <= kXchStride, "exchange area too small for this chain");
#pragma unroll
    for (int k0 = 0; k0 < NJ; k0 += G) {
      const int k = k0 + g;
      double s, c;
      sincos_pi2(PS::half_angle(k, qv), &s, &c);  // (k past the chain: half_angle gives 0)
      if (k < NJ) { x[2 * k] = s; x[2 * k + 1] = c; }
    }
    row_fence();
#pragma unroll
    for (int k = 0; k < NJ; k++) {
      sn[k] = x[2 * k]; cs[k] = x[2 * k + 1];
      if (PS::jtype(k) != JT_HINGE) { sn[k] = 0.0; cs[k] = 1.0; }
    }
    row_fence();
  }
}

// ---- row f1: the projection ---------------------------------------------------------------------------------------
template <class PS, int G>
struct PoseRows {
  static constexpr int NQ = PS::kNQ, NJ = PS::kNJ, NJX = NJ > 0 ? NJ : 1;
  struct Consts {
    const double *tail, *jrange;
    double tol, far_at;
    int maxit;
    __device__ __forceinline__ Consts(const int *__restrict__ pi, const double *__restrict__ pd)
        : tail(pd + pi[PH_OFF_TAIL]), jrange(pd + pi[PH_OFF_JRANGE]), tol(tail[PT_TOL]), far_at(2 * tail[PT_QSTEP]), maxit(pi[PH_MAXIT]) {}
  };
  struct Row {
    double qv[NQ], qold[NQ];
    int it;
  };

  // One pass of the loop of PoseConstraint.apply (pose_constraint.py:78-91) for the row: -1 = go on; 1 = within
  // tolerance, 0 = left the joint limits or went farther than 2 q_step from q_old, 2 = iteration bound.  Statement for
  // statement pose_project_lane (mjpl_project.h); r.it counts the Newton steps taken.
  static __device__ __forceinline__ int iterate(const Consts &c, Row &r, double *x, int g) {
    const double *tail = c.tail;
    double sn[NJX], cs[NJX], jx[NJX][6];
    hinge_sincos<PS, G>(r.qv, sn, cs, x, g);
    PoseChainOut o;
    PS::chain(r.qv, sn, cs, jx, o, tail);
    // pose_displacement (mjpl_pose.h), the inverse trigonometric functions of quat2rpy taken out of it
    const double cq[4] = {tail[PT_C_QUAT], tail[PT_C_QUAT + 1], tail[PT_C_QUAT + 2], tail[PT_C_QUAT + 3]};
    double qs[4], qc[4], t[3], d[6], dx[6];
    mat2quat(qs, o.site_xmat);
    mul_quat(qc, cq, qs);
    so3_apply(t, cq, o.site_xpos);
    d[0] = t[0] + tail[PT_C_POS]; d[1] = t[1] + tail[PT_C_POS + 1]; d[2] = t[2] + tail[PT_C_POS + 2];
    // quat2rpy(d + 3, qc); of quat2rpy(., qs) the Jacobian's E_rpy needs pitch and yaw
    double pitch_s, yaw_s;
    if constexpr (G == 1) {
      quat2rpy(d + 3, qc);
    } else {
      const double ay[3] = {2 * (qc[0] * qc[1] + qc[2] * qc[3]), 2 * (qc[0] * qc[3] + qc[1] * qc[2]), 2 * (qs[0] * qs[3] + qs[1] * qs[2])};
      const double ax[3] = {1 - 2 * (qc[1] * qc[1] + qc[2] * qc[2]), 1 - 2 * (qc[2] * qc[2] + qc[3] * qc[3]), 1 - 2 * (qs[2] * qs[2] + qs[3] * qs[3])};
      const double as[2] = {2 * (qc[0] * qc[2] - qc[3] * qc[1]), 2 * (qs[0] * qs[2] - qs[3] * qs[1])};
      // five values, value t on lane t % G in round t / G: even t an atan2, odd t an asin (every lane runs both routines
      // in a round -- one wave --, so a round costs one of each whatever G)
#pragma unroll
      for (int t0 = 0; t0 < 5; t0 += G) {
        const int t = t0 + g;
        const double y_ = t == 0 ? ay[0] : (t == 2 ? ay[1] : ay[2]), x_ = t == 0 ? ax[0] : (t == 2 ? ax[1] : ax[2]);
        const double s_ = t == 1 ? as[0] : as[1];
        const double ra = atan2(y_, x_), rs = asin(s_);
        if (t < 5) x[t] = (t & 1) ? rs : ra;
      }
      row_fence();
      d[3] = x[0]; d[4] = x[1]; d[5] = x[2];
      pitch_s = x[3]; yaw_s = x[4];
      row_fence();
    }
#pragma unroll
    for (int k = 0; k < 6; k++) {
      double v = 0;
      if (d[k] > tail[PT_HI + k]) v = d[k] - tail[PT_HI + k];
      if (d[k] < tail[PT_LO + k]) v = d[k] - tail[PT_LO + k];
      dx[k] = v;
    }
    if (norm6(dx) <= c.tol) return 1;
    if (r.it >= c.maxit) return 2;
    // _get_jacobian: E_rpy(world rpy of the site) @ [jacp; jacr]
    double c_p, c_y, s_p, s_y;
    if constexpr (G == 1) {
      double rpy[3];
      quat2rpy(rpy, qs);
      c_p = cos(rpy[1]); c_y = cos(rpy[2]); s_p = sin(rpy[1]); s_y = sin(rpy[2]);
    } else {
      const double ang = g == 0 ? pitch_s : yaw_s;  // lane 0: the pitch, lane 1: the yaw
      const double ca = cos(ang), sa = sin(ang);
      if (g < 2) { x[2 * g] = ca; x[2 * g + 1] = sa; }
      row_fence();
      c_p = x[0]; s_p = x[1]; c_y = x[2]; s_y = x[3];
      row_fence();
    }
    const double e33 = c_y / c_p, e34 = s_y / c_p, e43 = -s_y, e44 = c_p;
    const double e53 = c_y * (s_p / c_p), e54 = s_y * (s_p / c_p);
    double A[6][6];
#pragma unroll
    for (int i = 0; i < 6; i++)
#pragma unroll
      for (int j = 0; j < 6; j++) A[i][j] = 0;
#pragma unroll
    for (int jk = 0; jk < NJ; jk++) {
      const double ax[3] = {jx[jk][0], jx[jk][1], jx[jk][2]};
      double col[6];
      if (PS::jtype(jk) == JT_HINGE) {
        const double rr[3] = {o.site_xpos[0] - jx[jk][3], o.site_xpos[1] - jx[jk][4], o.site_xpos[2] - jx[jk][5]};
        col[0] = ax[1] * rr[2] - ax[2] * rr[1];
        col[1] = ax[2] * rr[0] - ax[0] * rr[2];
        col[2] = ax[0] * rr[1] - ax[1] * rr[0];
        col[3] = e33 * ax[0] + e34 * ax[1];
        col[4] = e43 * ax[0] + e44 * ax[1];
        col[5] = e53 * ax[0] + e54 * ax[1] + ax[2];
      } else {
        col[0] = ax[0]; col[1] = ax[1]; col[2] = ax[2];
        col[3] = e33 * 0.0 + e34 * 0.0;
        col[4] = e43 * 0.0 + e44 * 0.0;
        col[5] = e53 * 0.0 + e54 * 0.0 + 0.0;
      }
#pragma unroll
      for (int i = 0; i < 6; i++) jx[jk][i] = col[i];
#pragma unroll
      for (int i = 0; i < 6; i++)
#pragma unroll
        for (int j = 0; j < 6; j++) A[i][j] = A[i][j] + col[i] * col[j];
    }
    double y[6];
    const bool fast = spd6_solve_certified(A, dx, y, kPoseMaxCond);
    if (__ballot(!fast) != 0ull) {
      if (!fast) {
        double Ae[6][6];
#pragma unroll
        for (int i = 0; i < 6; i++)
#pragma unroll
          for (int j = 0; j < 6; j++) Ae[i][j] = A[i][j];
        pinv_sym6_apply(Ae, dx, y);
      }
    }
#pragma unroll
    for (int jk = 0; jk < NJ; jk++) {
      double acc = 0;
#pragma unroll
      for (int i = 0; i < 6; i++) acc = acc + jx[jk][i] * y[i];
      r.qv[PS::qadr(jk)] -= acc;
    }
    bool viol = false;
    double s = 0;
#pragma unroll
    for (int k = 0; k < NQ; k++) {
      const double v = r.qv[k];
      viol = viol || !(v >= c.jrange[2 * k] && v <= c.jrange[2 * k + 1]);
      const double dd = v - r.qold[k];
      s = s + dd * dd;
    }
    r.it++;
    if (viol || sqrt(s) > c.far_at) return 0;
    return -1;
  }
};

// PoseConstraint.apply for N rows: wave w owns rows [w * per, (w + 1) * per) and refills (header comment).
// Launch: ceil(N / per) workgroups of one wave; per >= 64 / G.
// Two launches for a large batch (mjpl_hip.hip: mjpl_pose_apply_dev): the rows' iteration counts differ -- 2.9 Newton
// steps on average, twenty for a few -- and a wave whose range is used up idles most of its lanes through its slowest
// rows' tail.  So the first launch (one lane per row) lets a row take `phase_steps` Newton steps at most; a row that
// wants more parks its working configuration in Qout and its step count in `it_store` and enters `out_list`; the second
// launch (eight lanes per row: few rows, latency matters) takes its rows from that list (`in_list`, `in_count` on the
// device: per = ceil(count / waves)) and finishes them.  The same iteration on the same values: the same results.
struct PosePhase {
  int phase_steps;           // > 0: a row is parked after that many Newton steps; 0: rows run to their end
  const int32_t *in_list;    // rows of this launch (null: 0 .. N - 1, read from Q)
  const int *in_count;
  int32_t *out_list;         // parked rows
  int *out_count;
  int32_t *it_store;         // [N] Newton steps taken by a parked row
};

template <class PS, int G>
__global__ void __launch_bounds__(kPoseBlock)
k_pose_apply_rows(const int *__restrict__ pi, const double *__restrict__ pd, const double *__restrict__ Qold,
                  const double *__restrict__ Q, int64_t N, int64_t per, double *__restrict__ Qout,
                  uint8_t *__restrict__ ok, int32_t *__restrict__ iters, PosePhase ph) {
  typedef PoseRows<PS, G> P;
  constexpr int NQ = PS::kNQ;
  __shared__ double xch[G == 1 ? 1 : RowId<G>::kRows * kXchStride];
  const RowId<G> id;
  double *x = xch + (G == 1 ? 0 : id.row * kXchStride);
  const typename P::Consts c(pi, pd);
  const bool resumed = ph.in_list != nullptr;
  if (resumed) {  // (the list's length is known on the device only)
    N = *ph.in_count;
    per = (N + (int64_t)gridDim.x - 1) / (int64_t)gridDim.x;
  }
  int64_t next = (int64_t)blockIdx.x * per;
  const int64_t end = next + per < N ? next + per : N;
  typename P::Row r;
  bool busy = false;
  int64_t i = 0;
  for (;;) {
    const unsigned long long free_rows = id.rows(!busy);
    if (free_rows != 0ull && next < end) {
      if (!busy) {
        const int64_t at = next + id.rank(free_rows);
        if (at < end) {
          i = resumed ? (int64_t)ph.in_list[at] : at;
          const double *src = resumed ? Qout : Q;
#pragma unroll
          for (int k = 0; k < NQ; k++) { r.qv[k] = src[i * NQ + k]; r.qold[k] = Qold[i * NQ + k]; }
          r.it = resumed ? ph.it_store[i] : 0;
          busy = true;
        }
      }
      next += __popcll(free_rows);
    }
    if (__ballot(busy) == 0ull) break;
    if (busy) {
      const int res = P::iterate(c, r, x, id.g);
      if (res >= 0) {
        if (id.g == 0) {
#pragma unroll
          for (int k = 0; k < NQ; k++) Qout[i * NQ + k] = r.qv[k];
          ok[i] = res == 1 ? 1 : 0;
          if (iters) iters[i] = res == 2 ? -r.it : r.it;
        }
        busy = false;
      }
    }
    // rows that have had their share of this launch: parked for the next one
    const bool park = busy && ph.phase_steps > 0 && r.it >= ph.phase_steps;
    const unsigned long long pm = id.rows(park);
    if (pm != 0ull) {
      int base = 0;
      if (id.lane == (int)__builtin_ctzll(pm)) base = atomicAdd(ph.out_count, __popcll(pm));
      base = __shfl(base, (int)__builtin_ctzll(pm));
      if (park && id.g == 0) {
#pragma unroll
        for (int k = 0; k < NQ; k++) Qout[i * NQ + k] = r.qv[k];
        ph.it_store[i] = r.it;
        ph.out_list[base + id.rank(pm)] = (int32_t)i;
      }
      if (park) busy = false;
    }
  }
}

// ---- row e: one chunk of an extension under a projecting constraint ---------------------------------------------------
// k_rrt_gen_project (mjpl_project.h) with the rows of the packed active-lane list: wave w owns entries
// [w * per, (w + 1) * per) of list `par` (per = ceil(entries / waves)), reserves S candidate slots for each of them at
// once, and walks them 64 / G at a time -- a lane's chain of up to S steps is one row; a step is _step, the Newton
// iterations of its projection, the rules of _constrained_extend (planning/utils.py:139-164); a row whose chain ends
// (rule failure, arrival, S steps) hands its lanes to the next entry.  What the chunk writes (candidates per lane,
// gfirst / gcount / gend) is what k_rrt_gen_project<void> writes: slot NUMBERS differ (nothing reads them across lanes).
template <class PS, int NP, int G, class PQ = PlanQRuntime>
__global__ void __launch_bounds__(kPoseBlock)
k_rrt_gen_project_rows(int L, int S, double eps, int par, const int *__restrict__ pi, const double *__restrict__ pd,
                       const int *__restrict__ qidx, const double *__restrict__ qbase, const uint8_t *__restrict__ isplan,
                       const double *__restrict__ lo, const double *__restrict__ hi, const double *__restrict__ Tgt, RrtLanes ln,
                       RrtCand cd, int *__restrict__ ctr) {
  static_assert(NP > 0, "the rows kernel keeps a lane's target and current rows in registers");
  typedef PoseRows<PS, G> P;
  constexpr int NQ = PS::kNQ;
  __shared__ double xch[G == 1 ? 1 : RowId<G>::kRows * kXchStride];
  const RowId<G> id;
  double *x = xch + (G == 1 ? 0 : id.row * kXchStride);
  const typename P::Consts c(pi, pd);
  const int n = ctr[RC_LISTN + par];
  const int32_t *__restrict__ list = ln.list[par];
  if (blockIdx.x == 0 && id.lane == 0) ctr[RC_LISTN + (par ^ 1)] = 0;  // (the acceptance kernel of this chunk fills it)
  const int per = (n + (int)gridDim.x - 1) / (int)gridDim.x;
  const int begin = (int)blockIdx.x * per;
  const int end = begin + per < n ? begin + per : n;
  if (begin >= end) return;
  int base = 0;
  if (id.lane == 0) {
    base = atomicAdd(&ctr[RC_EDGES], (end - begin) * S);
    atomicAdd(&ctr[RC_ACTIVE], end - begin);
  }
  base = __builtin_amdgcn_readfirstlane(base);
  int next = begin;
  typename P::Row r;
  double T[NP], w[NP], q[NP], d[NP];
  bool busy = false;
  int l = 0, first = 0, s = 0, count = 0, lvl0 = 0;

  // _step(w, T, eps) (planning/utils.py:167-186) into the row's working configuration
  auto begin_step = [&]() {
#pragma unroll
    for (int k = 0; k < NP; k++) d[k] = T[k] - w[k];
    const double mag = seqnorm(d, NP);
    const double sm = eps < mag ? eps : mag;
    bool reach = true;
#pragma unroll
    for (int k = 0; k < NP; k++) {
      q[k] = w[k] + (d[k] / mag) * sm;
      reach = reach && (q[k] == T[k]);
    }
    reach = reach || (mag <= eps);
    if (reach) {
#pragma unroll
      for (int k = 0; k < NP; k++) q[k] = T[k];  // a step of at most eps lands on the target
    }
#pragma unroll
    for (int k = 0; k < NQ; k++) { r.qold[k] = qbase[k]; r.qv[k] = qbase[k]; }
#pragma unroll
    for (int k = 0; k < NP; k++) {
      const int at = PQ::at(k, qidx);
#pragma unroll
      for (int j = 0; j < NQ; j++)
        if (j == at) { r.qold[j] = w[k]; r.qv[j] = q[k]; }
    }
    r.it = 0;
  };

  for (;;) {
    const unsigned long long free_rows = id.rows(!busy);
    if (free_rows != 0ull && next < end) {
      if (!busy) {
        const int at = next + id.rank(free_rows);
        if (at < end) {
          l = list[at];
          first = base + (at - begin) * S;
          s = 0; count = 0;
          lvl0 = ln.cnt[l];
#pragma unroll
          for (int k = 0; k < NP; k++) { T[k] = Tgt[(int64_t)k * L + l]; w[k] = ln.C[(int64_t)k * L + l]; }
          if (first + S > cd.cap) {  // (the host sizes S for the space there is: a lane refused here waits for the next chunk)
            if (id.g == 0) {
              atomicOr(&ctr[RC_OVERFLOW], 1);
              for (int slot = first; slot < cd.cap; slot++) {
                for (int k = 0; k < NP; k++) { cd.A[(int64_t)slot * NP + k] = w[k]; cd.B[(int64_t)slot * NP + k] = w[k]; }
                cd.lane[slot] = l; cd.level[slot] = 0; cd.rule[slot] = 0; cd.reach[slot] = 0;
              }
              ln.gfirst[l] = 0; ln.gcount[l] = 0; ln.gend[l] = 0;
            }
          } else {
            busy = true;
            begin_step();
          }
        }
      }
      next += __popcll(free_rows);
    }
    if (__ballot(busy) == 0ull) {
      if (next < end) continue;  // (every row of this pass was refused candidate space: on to the next entries)
      break;
    }
    if (busy) {
      const int result = P::iterate(c, r, x, id.g);
      if (result >= 0) {
        // the rules of _constrained_extend on what came back
        bool good = result == 1;
#pragma unroll
        for (int k = 0; k < NQ; k++)  // a projection that moves a joint outside the planning set is rejected
          if (!isplan[k]) good = good && (r.qv[k] == qbase[k]);
        bool reach = true;
#pragma unroll
        for (int k = 0; k < NP; k++) {
          const int at = PQ::at(k, qidx);
          double v = 0;
#pragma unroll
          for (int j = 0; j < NQ; j++)
            if (j == at) v = r.qv[j];
          q[k] = v;
          reach = reach && (q[k] == T[k]);
          good = good && (q[k] >= lo[k] && q[k] <= hi[k]);
        }
#pragma unroll
        for (int k = 0; k < NP; k++) d[k] = q[k] - w[k];
        good = good && !(seqnorm(d, NP) < 1e-8);
#pragma unroll
        for (int k = 0; k < NP; k++) d[k] = T[k] - q[k];
        const double after = seqnorm(d, NP);
#pragma unroll
        for (int k = 0; k < NP; k++) d[k] = T[k] - w[k];
        good = good && !(after > seqnorm(d, NP));
        const int slot = first + s;
        if (id.g == 0) {
#pragma unroll
          for (int k = 0; k < NP; k++) {
            cd.A[(int64_t)slot * NP + k] = w[k];
            cd.B[(int64_t)slot * NP + k] = good ? q[k] : w[k];  // (refused: a harmless edge for the validation launch)
          }
          cd.lane[slot] = l;
          cd.level[slot] = lvl0 + s;
          cd.rule[slot] = good ? 1 : 0;
          cd.reach[slot] = (good && reach) ? 1 : 0;
        }
        count++;
        s++;
        bool over = !good || reach;  // the lane ends after this chunk's candidates whatever their verdicts
        if (good) {
#pragma unroll
          for (int k = 0; k < NP; k++) w[k] = q[k];
        }
        if (!over && s < S) {
          begin_step();
        } else {
          if (id.g == 0) {
            for (int sl = first + count; sl < first + S; sl++) {  // slots behind the last candidate: a zero-length edge nobody reads
#pragma unroll
              for (int k = 0; k < NP; k++) { cd.A[(int64_t)sl * NP + k] = w[k]; cd.B[(int64_t)sl * NP + k] = w[k]; }
              cd.lane[sl] = l; cd.level[sl] = 0; cd.rule[sl] = 0; cd.reach[sl] = 0;
            }
            ln.gfirst[l] = first;
            ln.gcount[l] = count;
            ln.gend[l] = (uint8_t)(over ? 1 : 0);
          }
          busy = false;
        }
      }
    }
  }
}

// ---- row e, a step ahead ----------------------------------------------------------------------------------------------
// The tail of an extension is a few hundred lanes taking a thousand steps each, and a step is two passes of the loop
// above: the Newton update of the stepped configuration, then the evaluation that finds the result within tolerance.
// That second pass decides nothing the next step's first pass needs -- IF it comes back within tolerance, the candidate
// is the configuration it was given.  So a row takes SIXTEEN lanes, two halves of eight: the half that holds the row's
// projection runs its pass; the other half, in the same instructions, takes the step after -- _step from the candidate
// the closing evaluation is about to confirm, and the first Newton pass of its projection.  When the evaluation does
// come back within tolerance (and the rules of _constrained_extend accept the candidate), the half that ran ahead
// holds the row's projection from then on and the halves change roles; when it does not, what ran ahead is dropped.
// A step then costs one pass, not two.  Every value is computed by the statements of the kernel above from the same
// inputs -- running ahead only changes WHEN -- so the candidates are the same bit for bit
// (tests/test_gpu_rrt.py::..._whatever_the_lanes_per_row, 16 and 64).  Launch: ceil(entries / ROWS) waves at most 1 024.
template <class PS, int NP, int ROWS, class PQ = PlanQRuntime>
__global__ void __launch_bounds__(kPoseBlock)
k_rrt_gen_project_ahead(int L, int S, double eps, int par, const int *__restrict__ pi, const double *__restrict__ pd,
                        const int *__restrict__ qidx, const double *__restrict__ qbase, const uint8_t *__restrict__ isplan,
                        const double *__restrict__ lo, const double *__restrict__ hi, const double *__restrict__ Tgt, RrtLanes ln,
                        RrtCand cd, int *__restrict__ ctr) {
  static_assert(NP > 0, "the planning joints of the library's program");
  typedef PoseRows<PS, 8> P;
  constexpr int NQ = PS::kNQ;
  // ROWS rows of a wave, 4 or 1: a row runs ahead only in the pass it expects to close its step in, and the statements of
  // running ahead (the rules, _step: seven float64 divisions) are executed by the whole wave whenever one of its rows
  // does -- with four rows at four different points of their steps that is nearly every pass.  With no more rows than
  // SIMDs a row has its wave to itself (sixteen of its lanes) and pays for them once per step.
  static_assert(ROWS == 4 || ROWS == 1, "four rows of sixteen lanes, or one");
  constexpr int kRows = ROWS;
  // A row's target, the end of its chain, the candidate under judgement and the configuration its projection stands at live
  // in LDS, one copy for both halves: across the Newton pass only the projection itself stays in registers (the kernel
  // above keeps them all there and spills ~1 500 moves' worth into the accumulation registers; here they would be more)
  __shared__ double xch[2 * kRows * kXchStride];  // a half's exchange area (the sines and cosines its eight lanes share out)
  __shared__ double rowmem[kRows * (3 * NP + NQ)];
  const int lane = (int)(threadIdx.x & 63), row = lane >> 4, half = (lane >> 3) & 1, g = lane & 7;
  const bool leader = (lane & 15) == 0;
  const unsigned long long kLeaders = ROWS == 4 ? 0x0001000100010001ull : 0x1ull;
  const int lrow = row < kRows ? row : 0;  // (lanes beyond the wave's rows never hold one: they keep company)
  double *x = xch + (lrow * 2 + half) * kXchStride;
  double *Tl = rowmem + lrow * (3 * NP + NQ), *wl = Tl + NP, *qnl = wl + NP, *qrow = qnl + NP;
  const typename P::Consts c(pi, pd);
  const int n = ctr[RC_LISTN + par];
  const int32_t *__restrict__ list = ln.list[par];
  if (blockIdx.x == 0 && lane == 0) ctr[RC_LISTN + (par ^ 1)] = 0;  // (the acceptance kernel of this chunk fills it)
  const int per = (n + (int)gridDim.x - 1) / (int)gridDim.x;
  const int begin = (int)blockIdx.x * per;
  const int end = begin + per < n ? begin + per : n;
  if (begin >= end) return;
  int base = 0;
  if (lane == 0) {
    base = atomicAdd(&ctr[RC_EDGES], (end - begin) * S);
    atomicAdd(&ctr[RC_ACTIVE], end - begin);
  }
  base = __builtin_amdgcn_readfirstlane(base);
  int next = begin;
  typename P::Row r;  // the row's projection on the half `chalf`; scratch on the other
  bool busy = false;
  int l = 0, first = 0, s = 0, count = 0, lvl0 = 0;
  int chalf = 0;  // the half that holds the row's projection
  int cit = 0;    // Newton updates that projection has had (its r.it)
  int need = 1;   // updates the row's last step took before it closed: the pass this step is expected to close in

  // _step(from, T, eps) (planning/utils.py:167-186) into rr: the statements of begin_step above
  auto step_into = [&](typename P::Row &rr, const double *from) {
    double d[NP], q[NP];
#pragma unroll
    for (int k = 0; k < NP; k++) d[k] = Tl[k] - from[k];
    const double mag = seqnorm(d, NP);
    const double sm = eps < mag ? eps : mag;
    bool reach = true;
#pragma unroll
    for (int k = 0; k < NP; k++) {
      q[k] = from[k] + (d[k] / mag) * sm;
      reach = reach && (q[k] == Tl[k]);
    }
    reach = reach || (mag <= eps);
    if (reach) {
#pragma unroll
      for (int k = 0; k < NP; k++) q[k] = Tl[k];  // a step of at most eps lands on the target
    }
#pragma unroll
    for (int k = 0; k < NQ; k++) { rr.qold[k] = qbase[k]; rr.qv[k] = qbase[k]; }
#pragma unroll
    for (int k = 0; k < NP; k++) {
      const int at = PQ::at(k, qidx);
      const double fk = from[k];
#pragma unroll
      for (int j = 0; j < NQ; j++)
        if (j == at) { rr.qold[j] = fk; rr.qv[j] = q[k]; }
    }
    rr.it = 0;
  };
  // the rules of _constrained_extend on the configuration in `qrow`, were it what the projection returned within tolerance;
  // the candidate (its planning coordinates) goes to `qnl`.  Called by all lanes of the wave; ends with the row's fence.
  auto rules = [&](bool &good, bool &reach) {
    double d[NP], qn[NP];
    good = true;
#pragma unroll
    for (int k = 0; k < NQ; k++)  // a projection that moves a joint outside the planning set is rejected
      if (!isplan[k]) good = good && (qrow[k] == qbase[k]);
    reach = true;
#pragma unroll
    for (int k = 0; k < NP; k++) {
      qn[k] = qrow[PQ::at(k, qidx)];
      reach = reach && (qn[k] == Tl[k]);
      good = good && (qn[k] >= lo[k] && qn[k] <= hi[k]);
    }
#pragma unroll
    for (int k = 0; k < NP; k++) d[k] = qn[k] - wl[k];
    good = good && !(seqnorm(d, NP) < 1e-8);
#pragma unroll
    for (int k = 0; k < NP; k++) d[k] = Tl[k] - qn[k];
    const double after = seqnorm(d, NP);
#pragma unroll
    for (int k = 0; k < NP; k++) d[k] = Tl[k] - wl[k];
    good = good && !(after > seqnorm(d, NP));
    if (leader && busy) {
#pragma unroll
      for (int k = 0; k < NP; k++) qnl[k] = qn[k];
    }
    row_fence();
  };

  for (;;) {
    const unsigned long long free_rows = __ballot(!busy) & kLeaders;
    if (free_rows != 0ull && next < end) {
      bool took = false;
      if (!busy && row < kRows) {
        const int at = next + __popcll(free_rows & ((1ull << (row * 16)) - 1ull));
        if (at < end) {
          l = list[at];
          first = base + (at - begin) * S;
          s = 0; count = 0;
          lvl0 = ln.cnt[l];
          if (first + S > cd.cap) {  // (the host sizes S for the space there is: a lane refused here waits for the next chunk)
            if (leader) {
              atomicOr(&ctr[RC_OVERFLOW], 1);
              for (int slot = first; slot < cd.cap; slot++) {
                for (int k = 0; k < NP; k++) {
                  const double wk = ln.C[(int64_t)k * L + l];
                  cd.A[(int64_t)slot * NP + k] = wk; cd.B[(int64_t)slot * NP + k] = wk;
                }
                cd.lane[slot] = l; cd.level[slot] = 0; cd.rule[slot] = 0; cd.reach[slot] = 0;
              }
              ln.gfirst[l] = 0; ln.gcount[l] = 0; ln.gend[l] = 0;
            }
          } else {
            busy = true;
            took = true;
            chalf = 0; cit = 0; need = 1;
            if (leader) {
#pragma unroll
              for (int k = 0; k < NP; k++) { Tl[k] = Tgt[(int64_t)k * L + l]; wl[k] = ln.C[(int64_t)k * L + l]; }
            }
          }
        }
      }
      next += __popcll(free_rows);
      row_fence();
      if (took && half == 0) step_into(r, wl);
    }
    if (__ballot(busy) == 0ull) {
      if (next < end) continue;  // (every row of this pass was refused candidate space: on to the next entries)
      break;
    }
    // ---- one pass.  The configuration the row's projection stands at, to both halves
    const bool mine = busy && half == chalf;
    if (mine && g == 0) {
#pragma unroll
      for (int k = 0; k < NQ; k++) qrow[k] = r.qv[k];
    }
    row_fence();
    // Run ahead?  Only behind the evaluation that is expected to close its step (the projection has had as many updates as
    // the step before took), with a step left in the chunk, and if the candidate it would confirm passes the rules and is
    // not the target itself
    bool good = false, reach = false, judged = false;
    bool ahead = busy && cit >= need && s + 1 < S;
    if (__ballot(ahead) != 0ull) {
      rules(good, reach);
      judged = true;
      ahead = ahead && good && !reach;
    }
    const bool runs_ahead = ahead && !mine;
    if (runs_ahead) step_into(r, qnl);
    int res = -2;
    if (mine || runs_ahead) res = P::iterate(c, r, x, g);
    const int lead = lane & ~15;
    const int res_row = __shfl(res, lead + 8 * chalf), res_ahead = __shfl(res, lead + 8 * (chalf ^ 1));
    const bool done_step = busy && res_row != -1;
    if (busy && res_row == -1) {
      cit++;  // (another update: whatever ran ahead is dropped)
      if (cit > need) need = cit;
    }
    if (__ballot(done_step) != 0ull) {
      // a projection is through (1: within tolerance at `qrow`; 0 / 2: failed): the candidate and the rules
      if (!judged) rules(good, reach);
      bool restep = false;
      if (done_step) {
        good = good && res_row == 1;
        const int slot = first + s;
        if (leader) {
#pragma unroll
          for (int k = 0; k < NP; k++) {
            cd.A[(int64_t)slot * NP + k] = wl[k];
            cd.B[(int64_t)slot * NP + k] = good ? qnl[k] : wl[k];  // (refused: a harmless edge for the validation launch)
          }
          cd.lane[slot] = l;
          cd.level[slot] = lvl0 + s;
          cd.rule[slot] = good ? 1 : 0;
          cd.reach[slot] = (good && reach) ? 1 : 0;
        }
        count++;
        s++;
        const bool over = !good || reach;  // the lane ends after this chunk's candidates whatever their verdicts
        need = cit > 1 ? cit : 1;
        if (good && leader) {
#pragma unroll
          for (int k = 0; k < NP; k++) wl[k] = qnl[k];
        }
        if (!over && s < S) {
          if (ahead && res_ahead == -1) {
            chalf ^= 1;  // the half that ran ahead holds the next step's projection, one update in
            cit = 1;
          } else {
            cit = 0;
            restep = true;
          }
        } else {
          if (leader) {
            const double *wend = good ? qnl : wl;  // (the chain's end: wl is being rewritten by this lane alone, read it as it will be)
            for (int sl = first + count; sl < first + S; sl++) {  // slots behind the last candidate: a zero-length edge nobody reads
#pragma unroll
              for (int k = 0; k < NP; k++) { cd.A[(int64_t)sl * NP + k] = wend[k]; cd.B[(int64_t)sl * NP + k] = wend[k]; }
              cd.lane[sl] = l; cd.level[sl] = 0; cd.rule[sl] = 0; cd.reach[sl] = 0;
            }
            ln.gfirst[l] = first;
            ln.gcount[l] = count;
            ln.gend[l] = (uint8_t)(over ? 1 : 0);
          }
          busy = false;
        }
      }
      row_fence();  // (the chain's new end, before anybody steps from it)
      if (restep && half == chalf) step_into(r, wl);
    }
  }
}

// ---- row f3: batched IK seeds ---------------------------------------------------------------------------------------
// k_ik_solve (mjpl_project.h) around the generated chain: one seed per row, the Levenberg-Marquardt damping, the locking of
// joints on a limit and the restarts statement for statement; G lanes share the hinges' sines and cosines.
template <class PS, int G>
struct IkRows {
  static constexpr int NQ = PS::kNQ, NJ = PS::kNJ, NJX = NJ > 0 ? NJ : 1;
  struct Consts {
    const double *tail, *jrange, *movable;
    double pos_tol, ori_tol, damp, lm, max_step;
    int maxit, max_restarts;
    uint64_t restart_seed;
    __device__ __forceinline__ Consts(const int *__restrict__ pi, const double *__restrict__ pd, int max_restarts_, uint64_t seed_)
        : tail(pd + pi[PH_OFF_TAIL]), jrange(pd + pi[PH_OFF_JRANGE]), movable(jrange + 2 * NQ), pos_tol(tail[IT_POS_TOL]),
          ori_tol(tail[IT_ORI_TOL]), damp(tail[IT_DAMP]), lm(tail[IT_LM]), max_step(tail[IT_MAX_STEP]), maxit(pi[PH_MAXIT]),
          max_restarts(max_restarts_), restart_seed(seed_) {}
  };
  struct Row {
    double qv[NQ];
    double epos, eori, lam_scale, prev_err2, best_err2;
    int it, best_it, restarts;
    uint64_t rkey;
    __device__ __forceinline__ void start(const Consts &c, int64_t i) {
      epos = 0; eori = 0; lam_scale = 1.0; prev_err2 = 1.0e300; best_err2 = 1.0e300;
      it = 0; best_it = 0; restarts = 0;
      rkey = rrt_key(c.restart_seed, (uint64_t)i, 0x494bull);
    }
  };

  // one pass of k_ik_solve's loop: -1 = go on, 1 = solved, 0 = out of iterations
  static __device__ __forceinline__ int iterate(const Consts &c, Row &r, double *x, int g) {
    const double *tail = c.tail, *jrange = c.jrange, *movable = c.movable;
    double sn[NJX], cs[NJX], jx[NJX][6];
    hinge_sincos<PS, G>(r.qv, sn, cs, x, g);
    PoseChainOut o;
    PS::chain(r.qv, sn, cs, jx, o, tail);  // (the site offset sits at the same place of the IK tail: IT_SITE_* == PT_SITE_*)
    double e[6];
    ik_error(tail, o, e);
    r.epos = sqrt(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]);
    r.eori = sqrt(e[3] * e[3] + e[4] * e[4] + e[5] * e[5]);
    if (r.epos <= c.pos_tol && r.eori <= c.ori_tol) return 1;
    if (r.it >= c.maxit) return 0;
    const double err2 = r.epos * r.epos + r.eori * r.eori;
    if (err2 < 0.98 * r.best_err2) { r.best_err2 = err2; r.best_it = r.it; }
    if (r.it - r.best_it >= 12 && r.restarts < c.max_restarts) {
      r.restarts++;
#pragma unroll
      for (int jk = 0; jk < NJ; jk++) {
        const int jid = PS::jid(jk);
        if (movable[jid] != 0.0) {
          const double u = rrt_u01(r.rkey, (uint64_t)r.restarts * 64u + (uint64_t)(jid & 63));
          r.qv[PS::qadr(jk)] = jrange[2 * jid] + u * (jrange[2 * jid + 1] - jrange[2 * jid]);
        }
      }
      r.lam_scale = 1.0; r.prev_err2 = 1.0e300; r.best_err2 = 1.0e300; r.best_it = r.it;
      r.it++;
      return -1;
    }
    r.lam_scale = err2 > r.prev_err2 ? fmin(r.lam_scale * 4.0, 1.0e4) : fmax(r.lam_scale * 0.5, 1.0 / 64.0);
    r.prev_err2 = err2;
    const double lam = (c.damp + c.lm * err2) * r.lam_scale;
#pragma unroll
    for (int jk = 0; jk < NJ; jk++) {
      const double mv = movable[PS::jid(jk)];
      const double ax[3] = {jx[jk][0], jx[jk][1], jx[jk][2]};
      double col[6];
      if (PS::jtype(jk) == JT_HINGE) {
        const double rr[3] = {o.site_xpos[0] - jx[jk][3], o.site_xpos[1] - jx[jk][4], o.site_xpos[2] - jx[jk][5]};
        col[0] = mv * (ax[1] * rr[2] - ax[2] * rr[1]);
        col[1] = mv * (ax[2] * rr[0] - ax[0] * rr[2]);
        col[2] = mv * (ax[0] * rr[1] - ax[1] * rr[0]);
        col[3] = mv * ax[0]; col[4] = mv * ax[1]; col[5] = mv * ax[2];
      } else {
        col[0] = mv * ax[0]; col[1] = mv * ax[1]; col[2] = mv * ax[2];
        col[3] = 0; col[4] = 0; col[5] = 0;
      }
#pragma unroll
      for (int i = 0; i < 6; i++) jx[jk][i] = col[i];
    }
    unsigned locked = 0;
    double scale = 1.0;
    double dqv[NJX];
    for (int pass = 0; pass < 2; pass++) {
      double A[6][6];
#pragma unroll
      for (int i = 0; i < 6; i++)
#pragma unroll
        for (int j = 0; j < 6; j++) A[i][j] = (i == j) ? lam : 0.0;
#pragma unroll
      for (int k = 0; k < NJ; k++) {
        if ((locked >> (k & 31)) & 1u) continue;
#pragma unroll
        for (int i = 0; i < 6; i++)
#pragma unroll
          for (int j = 0; j < 6; j++) A[i][j] = A[i][j] + jx[k][i] * jx[k][j];
      }
      double y[6];
      chol6_solve(A, e, y);
      double big = 0;
#pragma unroll
      for (int k = 0; k < NJ; k++) {
        double acc = 0;
        if (!((locked >> (k & 31)) & 1u)) {
#pragma unroll
          for (int i = 0; i < 6; i++) acc = acc + jx[k][i] * y[i];
        }
        dqv[k] = acc;
        big = fabs(acc) > big ? fabs(acc) : big;
      }
      scale = big > c.max_step ? c.max_step / big : 1.0;
      if (pass == 1) break;
      unsigned out = 0;
#pragma unroll
      for (int jk = 0; jk < NJ; jk++) {
        const int jid = PS::jid(jk);
        const double dq = dqv[jk], v = r.qv[PS::qadr(jk)];
        const double span = jrange[2 * jid + 1] - jrange[2 * jid];
        const bool at_lo = v <= jrange[2 * jid] + 1e-9 * span, at_hi = v >= jrange[2 * jid + 1] - 1e-9 * span;
        if (movable[jid] != 0.0 && ((at_lo && dq < 0) || (at_hi && dq > 0))) out |= 1u << (jk & 31);
      }
      if (out == 0) break;
      locked = out;
    }
#pragma unroll
    for (int jk = 0; jk < NJ; jk++) {
      const int jid = PS::jid(jk);
      double v = r.qv[PS::qadr(jk)] + scale * dqv[jk];
      if (movable[jid] != 0.0) {
        v = v < jrange[2 * jid] ? jrange[2 * jid] : v;
        v = v > jrange[2 * jid + 1] ? jrange[2 * jid + 1] : v;
        r.qv[PS::qadr(jk)] = v;
      }
    }
    r.it++;
    return -1;
  }
};

template <class PS, int G>
__global__ void __launch_bounds__(kPoseBlock)
k_ik_solve_rows(const int *__restrict__ pi, const double *__restrict__ pd, const double *__restrict__ Q, int64_t N, int64_t per,
                double *__restrict__ Qout, uint8_t *__restrict__ ok, int32_t *__restrict__ iters, double *__restrict__ err,
                int max_restarts, uint64_t restart_seed) {
  typedef IkRows<PS, G> P;
  constexpr int NQ = PS::kNQ;
  __shared__ double xch[G == 1 ? 1 : RowId<G>::kRows * kXchStride];
  const RowId<G> id;
  double *x = xch + (G == 1 ? 0 : id.row * kXchStride);
  const typename P::Consts c(pi, pd, max_restarts, restart_seed);
  int64_t next = (int64_t)blockIdx.x * per;
  const int64_t end = next + per < N ? next + per : N;
  typename P::Row r;
  bool busy = false;
  int64_t i = 0;
  for (;;) {
    const unsigned long long free_rows = id.rows(!busy);
    if (free_rows != 0ull && next < end) {
      if (!busy) {
        i = next + id.rank(free_rows);
        if (i < end) {
#pragma unroll
          for (int k = 0; k < NQ; k++) r.qv[k] = Q[i * NQ + k];
          r.start(c, i);
          busy = true;
        }
      }
      next += __popcll(free_rows);
    }
    if (__ballot(busy) == 0ull) break;
    if (busy) {
      const int res = P::iterate(c, r, x, id.g);
      if (res >= 0) {
        if (id.g == 0) {
#pragma unroll
          for (int k = 0; k < NQ; k++) Qout[i * NQ + k] = r.qv[k];
          ok[i] = res == 1 ? 1 : 0;
          if (iters) iters[i] = r.it;
          if (err) { err[2 * i] = r.epos; err[2 * i + 1] = r.eori; }
        }
        busy = false;
      }
    }
  }
}

}  // namespace mjpl
