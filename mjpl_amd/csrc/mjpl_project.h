// mjpl_project.h -- the PoseConstraint projection (pose_constraint.py:72-171) as device code shared by libmjpl_hip.so
// and the per-model libraries: one lane's Newton iteration, the batched kernel around it (row f1), the planner's
// chunk of projected extension steps (row e) and the batched IK seeds (row f3).  Everything is a template over a pose spec `PS`: void = the
// interpreting statement (the chain program of mjpl_pose.h read from memory), else a generated struct whose `chain`
// is the same statements for ONE (model, site body) as straight-line code (mjpl_amd/specialise.py: generate_pose).
// Structs that cross the library boundary by value live here: this header is part of the source stamp.
#pragma once

#include <type_traits>

#include "mjpl_device.h"
#include "mjpl_pose.h"

namespace mjpl {

// counters shared with the host (one 64-byte block, read back per chunk / per exchange)
// RC_LISTN + p: entries of active-lane list p of a projecting extension (chunk c reads list c & 1, its acceptance
// kernel writes list (c + 1) & 1: the lanes still extending, packed -- the generating kernel's waves hold live rows only)
enum : int { RC_EDGES = 0, RC_ACC, RC_ACTIVE, RC_CONN, RC_CONN_REFA, RC_CONN_REFB, RC_NEWA, RC_NEWB,
             RC_OVERFLOW, RC_LISTN, RC_LISTN1, RC_SIZE = 16 };

struct RrtLanes {
  double *T, *C, *RA;          // [nplan][L] SoA: target, current end of the lane's chain, reach of extend A
  int32_t *near, *refA, *refB; // nearest node; final reference of the lane in tree A / B
  uint8_t *on, *act;           // takes part this round; still extending
  int32_t *cnt, *off;          // accepted nodes of the lane in this extension; exclusive scan
  int32_t *gfirst, *gcount;    // candidates of the lane in this chunk: first slot, how many
  uint8_t *gend;               // the lane ends after this chunk's candidates whatever their verdicts
  int32_t *goal;               // biased lanes: goal index (-1: not biased)
  int32_t *list[2];            // projecting extensions: the active lanes of a chunk, packed (RC_LISTN + parity entries)
};

struct RrtCand {               // candidates of one chunk, AoS rows of nplan
  double *A, *B;
  int32_t *lane, *level;
  uint8_t *valid, *rule, *reach;
  int cap;
};

__device__ __forceinline__ double seqnorm(const double *d, int n) {  // the sum order of the whole path
  double s = 0;
  for (int k = 0; k < n; k++) s = s + d[k] * d[k];
  return sqrt(s);
}

constexpr int kRrtMaxPlan = 16;

// ---- row f1: batched PoseConstraint (pose_constraint.py:72-171), one lane per configuration ----
// LDS per lane: the working qpos [nq] and a [6][njoint] store that holds the chain joints'
// world axis/anchor after FK and is overwritten in place by the RPY-Jacobian columns.
constexpr int kPoseBlock = 64;

constexpr double kPoseMaxCond = 1e6;  // (see k_pose_apply: fast path of the 6x6 pseudo-inverse)

// One lane's PoseConstraint.apply (pose_constraint.py:78-91): qw ([nq] at a stride of kPoseBlock, LDS) holds the
// configuration to project on entry and what the projection made of it on return; qo[k * qos] is q_old; jst the
// lane's [6 * njoint] store.  Every lane of the wave calls it (`active` false: the lane only keeps company).
// Returns 1: within tolerance, 0: left the joint limits or went farther than 2 q_step from q_old, 2: iteration
// bound; *iters counts the Newton steps taken.
// The chain program in (pi, pd) is interpreted.  (A model's library runs the same iteration around its generated chain
// -- the same statements in the same order, the same float64 values -- in the row kernels of mjpl_rows.h.)
__device__ __forceinline__ int pose_project_lane(const int *__restrict__ pi, const double *__restrict__ pd, double *qw,
                                                 double *jst, const double *qo, int qos, bool active, int *iters) {
  constexpr int B = kPoseBlock;
  const int nq = pi[PH_NQ], nj = pi[PH_NJOINT], maxit = pi[PH_MAXIT];
  const double *tail = pd + pi[PH_OFF_TAIL];
  const double *jrange = pd + pi[PH_OFF_JRANGE];
  const double tol = tail[PT_TOL], far_at = 2 * tail[PT_QSTEP];
  // joint ids / types of the chain, in chain order, follow the per-body counts in `pi`
  bool done = !active;
  int result = 0, it = 0;
  while (__ballot(!done) != 0ull) {
    if (!done) {
      PoseChainOut o;
      pose_chain(pi, pd, qw, B, jst, B, o);
      double dx[6], qs[4];
      pose_displacement(tail, o, dx, qs);
      if (norm6(dx) <= tol) {
        done = true; result = 1;
      } else if (it >= maxit) {
        done = true; result = 2;
      } else {
        // _get_jacobian: E_rpy(world rpy of the site) @ [jacp; jacr]
        double rpy[3];
        quat2rpy(rpy, qs);
        const double c_p = cos(rpy[1]), c_y = cos(rpy[2]), s_p = sin(rpy[1]), s_y = sin(rpy[2]);
        const double e33 = c_y / c_p, e34 = s_y / c_p, e43 = -s_y, e44 = c_p;
        const double e53 = c_y * (s_p / c_p), e54 = s_y * (s_p / c_p);
        double A[6][6];
#pragma unroll
        for (int r = 0; r < 6; r++)
#pragma unroll
          for (int c = 0; c < 6; c++) A[r][c] = 0;
        int ic = PH_SIZE, jk = 0;
        for (int b = 0; b < pi[PH_NBODY]; b++) {
          const int njnt = pi[ic++];
          for (int j = 0; j < njnt; j++, jk++, ic += 3) {
            const int jtype = pi[ic];
            const double ax[3] = {jst[(0 * nj + jk) * B], jst[(1 * nj + jk) * B], jst[(2 * nj + jk) * B]};
            double col[6];
            if (jtype == JT_HINGE) {
              const double r[3] = {o.site_xpos[0] - jst[(3 * nj + jk) * B], o.site_xpos[1] - jst[(4 * nj + jk) * B],
                                   o.site_xpos[2] - jst[(5 * nj + jk) * B]};
              col[0] = ax[1] * r[2] - ax[2] * r[1];
              col[1] = ax[2] * r[0] - ax[0] * r[2];
              col[2] = ax[0] * r[1] - ax[1] * r[0];
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
            for (int r = 0; r < 6; r++) jst[(r * nj + jk) * B] = col[r];
#pragma unroll
            for (int r = 0; r < 6; r++)
#pragma unroll
              for (int c = 0; c < 6; c++) A[r][c] = A[r][c] + col[r] * col[c];
          }
        }
        // pinv(J J^T) dx (pose_constraint.py:164-171).  Away from kinematic singularities J J^T is
        // positive definite and modestly conditioned: its inverse by a certified Cholesky solve is
        // pinv's result to ~cond * 2^-53 (<= 1e-10 relative here) at a hundredth of the
        // eigen-decomposition's latency; anything the certificate refuses takes the eigen path,
        // where pinv's singular-value cut-off decides.
        double y[6];
        const bool fast = spd6_solve_certified(A, dx, y, kPoseMaxCond);
        if (__ballot(!fast) != 0ull) {
          if (!fast) {  // (a copy: only this branch needs the matrix in memory, for the call)
            double Ae[6][6];
#pragma unroll
            for (int r = 0; r < 6; r++)
#pragma unroll
              for (int c = 0; c < 6; c++) Ae[r][c] = A[r][c];
            pinv_sym6_apply(Ae, dx, y);
          }
        }
        ic = PH_SIZE; jk = 0;
        for (int b = 0; b < pi[PH_NBODY]; b++) {
          const int njnt = pi[ic++];
          for (int j = 0; j < njnt; j++, jk++, ic += 3) {
            const int qadr = pi[ic + 1];
            double acc = 0;
#pragma unroll
            for (int r = 0; r < 6; r++) acc = acc + jst[(r * nj + jk) * B] * y[r];
            qw[qadr * B] -= acc;
          }
        }
        bool viol = false;
        double s = 0;
        for (int k = 0; k < nq; k++) {
          const double v = qw[k * B];
          viol = viol || !(v >= jrange[2 * k] && v <= jrange[2 * k + 1]);
          const double d = v - qo[k * qos];
          s = s + d * d;
        }
        it++;
        if (viol || sqrt(s) > far_at) { done = true; result = 0; }
      }
    }
  }
  *iters = it;
  return result;
}

__global__ void __launch_bounds__(kPoseBlock)
k_pose_apply(const int *__restrict__ pi, const double *__restrict__ pd, const double *__restrict__ Qold,
             const double *__restrict__ Q, int64_t N, double *__restrict__ Qout,
             uint8_t *__restrict__ ok, int32_t *__restrict__ iters) {
  extern __shared__ double smem[];
  constexpr int B = kPoseBlock;
  const int lane = threadIdx.x;
  const int nq = pi[PH_NQ];
  double *qw = smem + lane;                    // [nq][B]
  double *jst = smem + (size_t)nq * B + lane;  // [6 * nj][B]
  const int64_t i = (int64_t)blockIdx.x * B + lane;
  const bool active = i < N;
  for (int k = 0; k < nq; k++) qw[k * B] = active ? Q[i * nq + k] : 0.0;
  int it = 0;
  const int result = pose_project_lane(pi, pd, qw, jst, Qold + (active ? i : 0) * nq, 1, active, &it);
  if (active) {
    for (int k = 0; k < nq; k++) Qout[i * nq + k] = qw[k * B];
    ok[i] = result == 1 ? 1 : 0;
    if (iters) iters[i] = result == 2 ? -it : it;
  }
}


// One chunk of an extension under a projecting constraint (PoseConstraint): every active lane takes up to
// S steps -- _step towards the target, the projection of that step (pose_project_lane: PoseConstraint.apply),
// the rules of _constrained_extend on what comes back (planning/utils.py:139-164: joint limits, moved >= 1e-8,
// not farther from the target; a projection that moves a joint outside the planning set is rejected) -- each from
// where the step before ended.  The projection depends on the previous step's result, not on its collision
// verdict, so a lane's S candidates are generated on the spot and validated together; k_rrt_accept keeps the
// leading valid ones, which are exactly what S chunks of one step would have kept.  The host takes S > 1 when
// few lanes are left and a chunk costs the latency of its kernels whatever it holds.
// Every emitting lane owns S slots: those behind its last candidate carry a zero-length edge nobody reads.
// One workgroup of 64 lanes; LDS per lane: qw[nq] | jst[6 * njoint] | qo[nq].
// The lanes of a chunk come from the packed list `par` of the extension (RC_LISTN + par entries, written by k_rrt_begin
// for chunk 0 and by the acceptance kernel of the chunk before otherwise): lane i of the grid takes entry i.
// This is the interpreting statement (the chain program read from memory, one lane per row); a model's library runs the
// same chunk around its generated chain in k_rrt_gen_project_rows (mjpl_rows.h).
__global__ void __launch_bounds__(kPoseBlock)
k_rrt_gen_project(int L, int nplan, int S, double eps, int par, const int *__restrict__ pi, const double *__restrict__ pd,
                  const int *__restrict__ qidx, const double *__restrict__ qbase, const uint8_t *__restrict__ isplan,
                  const double *__restrict__ lo, const double *__restrict__ hi, const double *__restrict__ Tgt, RrtLanes ln,
                  RrtCand cd, int *__restrict__ ctr) {
  extern __shared__ double smem[];
  constexpr int B = kPoseBlock;
  const int lane = threadIdx.x;
  const int at = blockIdx.x * B + lane;
  const int nlist = ctr[RC_LISTN + par];
  if (blockIdx.x == 0 && lane == 0) ctr[RC_LISTN + (par ^ 1)] = 0;  // (the acceptance kernel of this chunk fills it)
  const int nq = pi[PH_NQ], nj = pi[PH_NJOINT];
  const bool act = at < nlist;
  const int l = act ? ln.list[par][at] : 0;
  const unsigned long long am = __ballot(act);
  if (am == 0ull) return;
  double *qw = smem + lane;
  double *jst = smem + (size_t)nq * B + lane;
  double *qo = smem + ((size_t)nq + 6 * (size_t)nj) * B + lane;
  // S slots per active lane, one reservation per wave
  int base = 0;
  if (lane == 0) {
    const int nact = __popcll(am);
    base = atomicAdd(&ctr[RC_EDGES], nact * S);
    atomicAdd(&ctr[RC_ACTIVE], nact);
  }
  base = __builtin_amdgcn_readfirstlane(base);
  int first = base + __popcll(am & ((1ull << lane) - 1ull)) * S;
  bool going = act;
  if (going && first + S > cd.cap) {  // (the host sizes S for the space there is: a lane refused here waits for the next chunk)
    atomicOr(&ctr[RC_OVERFLOW], 1);
    for (int slot = first; slot < cd.cap; slot++) {
      for (int c = 0; c < nplan; c++) {
        const double v = ln.C[(int64_t)c * L + l];
        cd.A[(int64_t)slot * nplan + c] = v;
        cd.B[(int64_t)slot * nplan + c] = v;
      }
      cd.lane[slot] = l; cd.level[slot] = 0; cd.rule[slot] = 0; cd.reach[slot] = 0;
    }
    ln.gfirst[l] = 0; ln.gcount[l] = 0; ln.gend[l] = 0;
    going = false;
  }
  const bool mine = going;
  double T[kRrtMaxPlan], w[kRrtMaxPlan], q[kRrtMaxPlan], d[kRrtMaxPlan];
  if (going)
    for (int c = 0; c < nplan; c++) { T[c] = Tgt[(int64_t)c * L + l]; w[c] = ln.C[(int64_t)c * L + l]; }
  const int lvl0 = mine ? ln.cnt[l] : 0;
  int count = 0, end = 0;
  for (int s = 0; s < S && __ballot(going) != 0ull; s++) {
    if (going) {
      // _step(w, T, eps) (planning/utils.py:167-186)
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
      for (int k = 0; k < nq; k++) { qo[k * B] = qbase[k]; qw[k * B] = qbase[k]; }
      for (int c = 0; c < nplan; c++) { qo[qidx[c] * B] = w[c]; qw[qidx[c] * B] = q[c]; }
    }
    int it = 0;
    const int result = pose_project_lane(pi, pd, qw, jst, qo, B, going, &it);
    if (going) {
      bool good = result == 1;
      for (int k = 0; k < nq; k++)  // a projection that moves a joint outside the planning set is rejected
        if (!isplan[k]) good = good && (qw[k * B] == qbase[k]);
      bool reach = true;
      for (int c = 0; c < nplan; c++) {
        q[c] = qw[qidx[c] * B];
        reach = reach && (q[c] == T[c]);
        good = good && (q[c] >= lo[c] && q[c] <= hi[c]);
      }
      for (int c = 0; c < nplan; c++) d[c] = q[c] - w[c];
      good = good && !(seqnorm(d, nplan) < 1e-8);
      for (int c = 0; c < nplan; c++) d[c] = T[c] - q[c];
      const double after = seqnorm(d, nplan);
      for (int c = 0; c < nplan; c++) d[c] = T[c] - w[c];
      good = good && !(after > seqnorm(d, nplan));
      const int slot = first + s;
      for (int c = 0; c < nplan; c++) {
        cd.A[(int64_t)slot * nplan + c] = w[c];
        cd.B[(int64_t)slot * nplan + c] = good ? q[c] : w[c];  // (refused: a harmless edge for the validation launch)
      }
      cd.lane[slot] = l;
      cd.level[slot] = lvl0 + s;
      cd.rule[slot] = good ? 1 : 0;
      cd.reach[slot] = (good && reach) ? 1 : 0;
      count++;
      if (good) {
        for (int c = 0; c < nplan; c++) w[c] = q[c];
        if (reach) { end = 1; going = false; }
      } else {
        end = 1; going = false;
      }
    }
  }
  if (mine) {
    for (int slot = first + count; slot < first + S; slot++) {
      for (int c = 0; c < nplan; c++) {
        cd.A[(int64_t)slot * nplan + c] = w[c];
        cd.B[(int64_t)slot * nplan + c] = w[c];
      }
      cd.lane[slot] = l; cd.level[slot] = 0; cd.rule[slot] = 0; cd.reach[slot] = 0;
    }
    ln.gfirst[l] = first;
    ln.gcount[l] = count;
    ln.gend[l] = (uint8_t)end;
  }
}


// ---- row f3: batched IK seeds (the role of MinkIKSolver.solve_ik, mink_ik_solver.py:72-116) ----
// One lane per seed: damped least squares on the 6-D world-frame pose error of the site,
//   dq = J^T (J J^T + (damp + lm |e|^2) I)^-1 e,   |dq|_inf <= max_step,   q clamped to jnt_range,
// joints outside the solver's joint set are held (their Jacobian columns are zero).  A seed is
// solved when |e_pos| <= pos_tol and |e_ori| <= ori_tol (:100-102).
// The chain program is interpreted here; IkRows (mjpl_rows.h): the same iteration around a model's generated chain,
// configuration / columns / step in registers (statement for statement: the same results).
__global__ void __launch_bounds__(kPoseBlock)
k_ik_solve(const int *__restrict__ pi, const double *__restrict__ pd, const double *__restrict__ Q,
           int64_t N, double *__restrict__ Qout, uint8_t *__restrict__ ok, int32_t *__restrict__ iters,
           double *__restrict__ err, int max_restarts, uint64_t restart_seed) {
  extern __shared__ double smem[];
  constexpr int B = kPoseBlock;
  const int lane = threadIdx.x;
  const int nq = pi[PH_NQ], nj = pi[PH_NJOINT], maxit = pi[PH_MAXIT];
  const double *tail = pd + pi[PH_OFF_TAIL];
  const double *jrange = pd + pi[PH_OFF_JRANGE];
  const double *movable = jrange + 2 * nq;
  double *qw = smem + lane;
  double *jst = smem + (size_t)nq * B + lane;
  const int64_t i = (int64_t)blockIdx.x * B + lane;
  const bool active = i < N;
  for (int k = 0; k < nq; k++) qw[k * B] = active ? Q[i * nq + k] : 0.0;
  const double pos_tol = tail[IT_POS_TOL], ori_tol = tail[IT_ORI_TOL];
  const double damp = tail[IT_DAMP], lm = tail[IT_LM], max_step = tail[IT_MAX_STEP];
  bool done = !active, solved = false;
  int it = 0;
  double epos = 0, eori = 0;
  // Levenberg-Marquardt flavour of the damping: the error-proportional term is scaled up when an
  // iteration made the error grow and relaxed when it shrank; joints that sit on a limit and are
  // pushed further out are taken out of the step (one re-solve), otherwise a clamped joint keeps
  // absorbing the step every iteration and the seed stalls on the boundary.
  double lam_scale = 1.0, prev_err2 = 1.0e300;
  // A seed that has not improved its best error by 1 % for 12 iterations sits in a local minimum
  // (in practice: on joint limits) and more iterations do not move it.  Like the reference, which
  // re-draws the start with random_config when an attempt fails (mink_ik_solver.py:108-115), the
  // row then restarts from a fresh uniform draw of the solver's joints -- inside its iteration
  // budget, up to max_restarts times.
  double best_err2 = 1.0e300;
  int best_it = 0, restarts = 0;
  const uint64_t rkey = rrt_key(restart_seed, (uint64_t)i, 0x494bull);
  while (__ballot(!done) != 0ull) {
    if (!done) {
      PoseChainOut o;
      pose_chain(pi, pd, qw, B, jst, B, o);
      double e[6];
      ik_error(tail, o, e);
      epos = sqrt(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]);
      eori = sqrt(e[3] * e[3] + e[4] * e[4] + e[5] * e[5]);
      if (epos <= pos_tol && eori <= ori_tol) {
        done = true; solved = true;
      } else if (it >= maxit) {
        done = true;
      } else {
        const double err2 = epos * epos + eori * eori;
        if (err2 < 0.98 * best_err2) { best_err2 = err2; best_it = it; }
        if (it - best_it >= 12 && restarts < max_restarts) {
          restarts++;
          int ic2 = PH_SIZE;
          for (int b = 0; b < pi[PH_NBODY]; b++) {
            const int njnt = pi[ic2++];
            for (int j = 0; j < njnt; j++, ic2 += 3) {
              const int qadr = pi[ic2 + 1], jid = pi[ic2 + 2];
              if (movable[jid] != 0.0) {
                const double u = rrt_u01(rkey, (uint64_t)restarts * 64u + (uint64_t)(jid & 63));
                qw[qadr * B] = jrange[2 * jid] + u * (jrange[2 * jid + 1] - jrange[2 * jid]);
              }
            }
          }
          lam_scale = 1.0; prev_err2 = 1.0e300; best_err2 = 1.0e300; best_it = it;
          it++;
          continue;
        }
        lam_scale = err2 > prev_err2 ? fmin(lam_scale * 4.0, 1.0e4) : fmax(lam_scale * 0.5, 1.0 / 64.0);
        prev_err2 = err2;
        const double lam = (damp + lm * err2) * lam_scale;
        // Jacobian columns of the chain joints -> rows 0..5 of the store (held joints: zero)
        int ic = PH_SIZE, jk = 0;
        for (int b = 0; b < pi[PH_NBODY]; b++) {
          const int njnt = pi[ic++];
          for (int j = 0; j < njnt; j++, jk++, ic += 3) {
            const int jtype = pi[ic], jid = pi[ic + 2];
            const double mv = movable[jid];
            const double ax[3] = {jst[(0 * nj + jk) * B], jst[(1 * nj + jk) * B], jst[(2 * nj + jk) * B]};
            double col[6];
            if (jtype == JT_HINGE) {
              const double r[3] = {o.site_xpos[0] - jst[(3 * nj + jk) * B], o.site_xpos[1] - jst[(4 * nj + jk) * B],
                                   o.site_xpos[2] - jst[(5 * nj + jk) * B]};
              col[0] = mv * (ax[1] * r[2] - ax[2] * r[1]);
              col[1] = mv * (ax[2] * r[0] - ax[0] * r[2]);
              col[2] = mv * (ax[0] * r[1] - ax[1] * r[0]);
              col[3] = mv * ax[0]; col[4] = mv * ax[1]; col[5] = mv * ax[2];
            } else {
              col[0] = mv * ax[0]; col[1] = mv * ax[1]; col[2] = mv * ax[2];
              col[3] = 0; col[4] = 0; col[5] = 0;
            }
#pragma unroll
            for (int r = 0; r < 6; r++) jst[(r * nj + jk) * B] = col[r];
          }
        }
        unsigned locked = 0;
        double scale = 1.0;
        for (int pass = 0; pass < 2; pass++) {
          double A[6][6];
#pragma unroll
          for (int r = 0; r < 6; r++)
#pragma unroll
            for (int c = 0; c < 6; c++) A[r][c] = (r == c) ? lam : 0.0;
          for (int k = 0; k < nj; k++) {
            if ((locked >> (k & 31)) & 1u) continue;
            double col[6];
#pragma unroll
            for (int r = 0; r < 6; r++) col[r] = jst[(r * nj + k) * B];
#pragma unroll
            for (int r = 0; r < 6; r++)
#pragma unroll
              for (int c = 0; c < 6; c++) A[r][c] = A[r][c] + col[r] * col[c];
          }
          double y[6];
          chol6_solve(A, e, y);
          // dq of chain joint k -> row 6 of the store; step length limit over the whole update
          double big = 0;
          for (int k = 0; k < nj; k++) {
            double acc = 0;
            if (!((locked >> (k & 31)) & 1u)) {
#pragma unroll
              for (int r = 0; r < 6; r++) acc = acc + jst[(r * nj + k) * B] * y[r];
            }
            jst[(6 * nj + k) * B] = acc;
            big = fabs(acc) > big ? fabs(acc) : big;
          }
          scale = big > max_step ? max_step / big : 1.0;
          if (pass == 1 || nj > 32) break;
          unsigned out = 0;
          ic = PH_SIZE; jk = 0;
          for (int b = 0; b < pi[PH_NBODY]; b++) {
            const int njnt = pi[ic++];
            for (int j = 0; j < njnt; j++, jk++, ic += 3) {
              const int qadr = pi[ic + 1], jid = pi[ic + 2];
              const double dq = jst[(6 * nj + jk) * B], v = qw[qadr * B];
              const double span = jrange[2 * jid + 1] - jrange[2 * jid];
              const bool at_lo = v <= jrange[2 * jid] + 1e-9 * span, at_hi = v >= jrange[2 * jid + 1] - 1e-9 * span;
              if (movable[jid] != 0.0 && ((at_lo && dq < 0) || (at_hi && dq > 0))) out |= 1u << (jk & 31);
            }
          }
          if (out == 0) break;
          locked = out;
        }
        ic = PH_SIZE; jk = 0;
        for (int b = 0; b < pi[PH_NBODY]; b++) {
          const int njnt = pi[ic++];
          for (int j = 0; j < njnt; j++, jk++, ic += 3) {
            const int qadr = pi[ic + 1], jid = pi[ic + 2];
            double v = qw[qadr * B] + scale * jst[(6 * nj + jk) * B];
            if (movable[jid] != 0.0) {
              v = v < jrange[2 * jid] ? jrange[2 * jid] : v;
              v = v > jrange[2 * jid + 1] ? jrange[2 * jid + 1] : v;
              qw[qadr * B] = v;
            }
          }
        }
        it++;
      }
    }
  }
  if (active) {
    for (int k = 0; k < nq; k++) Qout[i * nq + k] = qw[k * B];
    ok[i] = solved ? 1 : 0;
    if (iters) iters[i] = it;
    if (err) { err[2 * i] = epos; err[2 * i + 1] = eori; }
  }
}

}  // namespace mjpl

#include "mjpl_rows.h"
