/*
 * mjpl_oracle_pose.c -- CPU ORACLE (test infrastructure, NOT the product).  See mjpl_oracle.h.
 *
 * Row f1 of SURVEY.md section 8: PoseConstraint (src/mjpl/constraint/pose_constraint.py).
 * Upstream arithmetic: mink.lie (SE3/SO3: [MINK-recalled]) and MuJoCo (mj_kinematics,
 * mj_jacSite, mju_mat2Quat: [MJ-recalled]); neither is vendored under /root/reference.
 */
#include "mjpl_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#include "orc_math.h"

#define ORC_MAXCHAIN 64

/* mju_mat2Quat [MJ-recalled: engine_util_spatial.c] */
static void mat2quat(double *quat, const double *mat) {
  if (mat[0] + mat[4] + mat[8] > 0) {
    quat[0] = 0.5 * sqrt(1 + mat[0] + mat[4] + mat[8]);
    quat[1] = 0.25 * (mat[7] - mat[5]) / quat[0];
    quat[2] = 0.25 * (mat[2] - mat[6]) / quat[0];
    quat[3] = 0.25 * (mat[3] - mat[1]) / quat[0];
  } else if (mat[0] > mat[4] && mat[0] > mat[8]) {
    quat[1] = 0.5 * sqrt(1 + mat[0] - mat[4] - mat[8]);
    quat[0] = 0.25 * (mat[7] - mat[5]) / quat[1];
    quat[2] = 0.25 * (mat[1] + mat[3]) / quat[1];
    quat[3] = 0.25 * (mat[2] + mat[6]) / quat[1];
  } else if (mat[4] > mat[8]) {
    quat[2] = 0.5 * sqrt(1 - mat[0] + mat[4] - mat[8]);
    quat[0] = 0.25 * (mat[2] - mat[6]) / quat[2];
    quat[1] = 0.25 * (mat[1] + mat[3]) / quat[2];
    quat[3] = 0.25 * (mat[5] + mat[7]) / quat[2];
  } else {
    quat[3] = 0.5 * sqrt(1 - mat[0] - mat[4] + mat[8]);
    quat[0] = 0.25 * (mat[3] - mat[1]) / quat[3];
    quat[1] = 0.25 * (mat[2] + mat[6]) / quat[3];
    quat[2] = 0.25 * (mat[5] + mat[7]) / quat[3];
  }
  normalize4(quat);
}

/* SO3.apply [MINK-recalled: lie/so3.py]: (q * (0, v) * q^-1).xyz with mju_mulQuat products */
static void so3_apply(double *res, const double *quat, const double *vec) {
  const double pv[4] = {0, vec[0], vec[1], vec[2]};
  const double qi[4] = {quat[0], -quat[1], -quat[2], -quat[3]};
  double t[4], r[4];
  mul_quat(t, quat, pv);
  mul_quat(r, t, qi);
  res[0] = r[1]; res[1] = r[2]; res[2] = r[3];
}

/* SO3.as_rpy_radians [MINK-recalled: lie/so3.py compute_{roll,pitch,yaw}_radians] */
static void quat2rpy(double *rpy, const double *q) {
  rpy[0] = atan2(2 * (q[0] * q[1] + q[2] * q[3]), 1 - 2 * (q[1] * q[1] + q[2] * q[2]));
  rpy[1] = asin(2 * (q[0] * q[2] - q[3] * q[1]));
  rpy[2] = atan2(2 * (q[0] * q[3] + q[1] * q[2]), 1 - 2 * (q[2] * q[2] + q[3] * q[3]));
}

/* mj_kinematics restricted to the ancestors of the site's body, keeping what mj_jacSite
 * needs: per joint of the chain its world axis and anchor (xaxis, xanchor).
 * Same operation order as orc_kinematics (mjpl_oracle.c). */
typedef struct chain_kin {
  int njoint;
  int jid[ORC_MAXCHAIN];
  double xaxis[ORC_MAXCHAIN][3], xanchor[ORC_MAXCHAIN][3];
  double xpos[3], xquat[4], xmat[9];        /* the site's body */
  double site_xpos[3], site_xmat[9];
} chain_kin;

static int chain_kinematics(const orc_model *m, const orc_pose *ps, const double *qpos, chain_kin *ck) {
  int chain[ORC_MAXCHAIN], n = 0, status = ORC_OK;
  for (int b = ps->site_body; b > 0; b = m->body_parentid[b]) {
    if (n >= ORC_MAXCHAIN) return ORC_E_OVERFLOW;
    chain[n++] = b;
  }
  double p[3] = {0, 0, 0}, q[4] = {1, 0, 0, 0}, mat[9];
  quat2mat(mat, q);
  ck->njoint = 0;
  for (int c = n - 1; c >= 0; c--) {
    const int i = chain[c];
    double np_[3], nq[4];
    mul_mat_vec3(np_, mat, m->body_pos + 3 * i);
    np_[0] += p[0]; np_[1] += p[1]; np_[2] += p[2];
    mul_quat(nq, q, m->body_quat + 4 * i);
    const int jadr = m->body_jntadr[i];
    for (int j = 0; j < m->body_jntnum[i]; j++) {
      const int jid = jadr + j;
      const int qadr = m->jnt_qposadr[jid];
      const int jtype = m->jnt_type[jid];
      if (ck->njoint >= ORC_MAXCHAIN) return ORC_E_OVERFLOW;
      double *xaxis = ck->xaxis[ck->njoint], *xanchor = ck->xanchor[ck->njoint];
      ck->jid[ck->njoint++] = jid;
      rot_vec_quat(xaxis, m->jnt_axis + 3 * jid, nq);
      rot_vec_quat(xanchor, m->jnt_pos + 3 * jid, nq);
      xanchor[0] += np_[0]; xanchor[1] += np_[1]; xanchor[2] += np_[2];
      if (jtype == ORC_JNT_SLIDE) {
        const double d = qpos[qadr] - m->qpos0[qadr];
        np_[0] += xaxis[0] * d; np_[1] += xaxis[1] * d; np_[2] += xaxis[2] * d;
      } else if (jtype == ORC_JNT_HINGE) {
        double qloc[4], vec[3];
        axis_angle2quat(qloc, m->jnt_axis + 3 * jid, qpos[qadr] - m->qpos0[qadr]);
        mul_quat(nq, nq, qloc);
        rot_vec_quat(vec, m->jnt_pos + 3 * jid, nq);
        np_[0] = xanchor[0] - vec[0]; np_[1] = xanchor[1] - vec[1]; np_[2] = xanchor[2] - vec[2];
      } else {
        status = ORC_E_JOINT;
      }
    }
    normalize4(nq);
    memcpy(p, np_, sizeof(p));
    memcpy(q, nq, sizeof(q));
    quat2mat(mat, q);
  }
  memcpy(ck->xpos, p, sizeof(p));
  memcpy(ck->xquat, q, sizeof(q));
  memcpy(ck->xmat, mat, sizeof(mat));
  /* mj_local2Global for the site */
  double sp[3], sq[4];
  mul_mat_vec3(sp, mat, ps->site_pos);
  ck->site_xpos[0] = sp[0] + p[0]; ck->site_xpos[1] = sp[1] + p[1]; ck->site_xpos[2] = sp[2] + p[2];
  mul_quat(sq, q, ps->site_quat);
  quat2mat(ck->site_xmat, sq);
  return status;
}

int orc_site_pose(const orc_model *m, const orc_pose *p, const double *qpos, double *xpos, double *xmat) {
  chain_kin ck;
  int rc = chain_kinematics(m, p, qpos, &ck);
  if (xpos) memcpy(xpos, ck.site_xpos, sizeof(ck.site_xpos));
  if (xmat) memcpy(xmat, ck.site_xmat, sizeof(ck.site_xmat));
  return rc;
}

/* pose_constraint.py:93-123 */
static void displacement(const orc_pose *p, const chain_kin *ck, double *dx) {
  /* world_T_site = SE3.from_rotation_and_translation(SO3.from_matrix(xmat), xpos) (utils.py:70-75) */
  double qs[4], qc[4], t[3], rpy[3], d[6];
  mat2quat(qs, ck->site_xmat);
  /* C_T_site = C_T_world.multiply(world_T_site) [MINK-recalled: lie/se3.py multiply] */
  mul_quat(qc, p->c_quat, qs);
  so3_apply(t, p->c_quat, ck->site_xpos);
  d[0] = t[0] + p->c_pos[0]; d[1] = t[1] + p->c_pos[1]; d[2] = t[2] + p->c_pos[2];
  quat2rpy(rpy, qc);
  d[3] = rpy[0]; d[4] = rpy[1]; d[5] = rpy[2];
  for (int k = 0; k < 6; k++) {
    dx[k] = 0;
    if (d[k] > p->hi[k]) dx[k] = d[k] - p->hi[k];
    if (d[k] < p->lo[k]) dx[k] = d[k] - p->lo[k];
  }
}

int orc_pose_displacement(const orc_model *m, const orc_pose *p, const double *qpos, double *dx) {
  chain_kin ck;
  int rc = chain_kinematics(m, p, qpos, &ck);
  if (rc != ORC_OK) return rc;
  displacement(p, &ck, dx);
  return ORC_OK;
}

/* pose_constraint.py:125-171.  mj_jacSite [MJ-recalled: engine_core_util.c mj_jac]: for every
 * dof above the body, hinge: jacr = axis, jacp = axis x (point - anchor); slide: jacp = axis. */
static void jacobian(const orc_model *m, const chain_kin *ck, double *J) {
  const int nv = m->njnt;
  double qs[4], rpy[3];
  for (int k = 0; k < 6 * nv; k++) J[k] = 0;
  mat2quat(qs, ck->site_xmat);
  quat2rpy(rpy, qs);
  const double c_p = cos(rpy[1]), c_y = cos(rpy[2]), s_p = sin(rpy[1]), s_y = sin(rpy[2]);
  const double e33 = c_y / c_p, e34 = s_y / c_p, e43 = -s_y, e44 = c_p;
  const double e53 = c_y * (s_p / c_p), e54 = s_y * (s_p / c_p);
  for (int k = 0; k < ck->njoint; k++) {
    const int jid = ck->jid[k];
    const double *ax = ck->xaxis[k];
    double jp[3], jr[3] = {0, 0, 0};
    if (m->jnt_type[jid] == ORC_JNT_HINGE) {
      const double r[3] = {ck->site_xpos[0] - ck->xanchor[k][0], ck->site_xpos[1] - ck->xanchor[k][1],
                           ck->site_xpos[2] - ck->xanchor[k][2]};
      jp[0] = ax[1] * r[2] - ax[2] * r[1];
      jp[1] = ax[2] * r[0] - ax[0] * r[2];
      jp[2] = ax[0] * r[1] - ax[1] * r[0];
      jr[0] = ax[0]; jr[1] = ax[1]; jr[2] = ax[2];
    } else {
      jp[0] = ax[0]; jp[1] = ax[1]; jp[2] = ax[2];
    }
    J[0 * nv + jid] = jp[0];
    J[1 * nv + jid] = jp[1];
    J[2 * nv + jid] = jp[2];
    J[3 * nv + jid] = e33 * jr[0] + e34 * jr[1];
    J[4 * nv + jid] = e43 * jr[0] + e44 * jr[1];
    J[5 * nv + jid] = e53 * jr[0] + e54 * jr[1] + jr[2];
  }
}

int orc_pose_jacobian(const orc_model *m, const orc_pose *p, const double *qpos, double *J) {
  chain_kin ck;
  int rc = chain_kinematics(m, p, qpos, &ck);
  if (rc != ORC_OK) return rc;
  jacobian(m, &ck, J);
  return ORC_OK;
}

/* np.linalg.pinv(A) for symmetric A (6x6): A = V diag(w) V^T by cyclic Jacobi rotations (sweeps
 * until the off-diagonal mass is below 1e-40 of the diagonal mass, i.e. |off| <= 1e-20 |diag|;
 * at most 12; a 6x6 converges quadratically in ~6), then V diag(1/w_i if w_i > 1e-15 * max w
 * else 0) V^T -- numpy's default cutoff rcond = 1e-15 on the singular values. */
#define ORC_JACOBI_SWEEPS 12
void orc_pinv_sym6(const double *Ain, double *out) {
  double A[6][6], V[6][6];
  for (int i = 0; i < 6; i++)
    for (int j = 0; j < 6; j++) { A[i][j] = Ain[6 * i + j]; V[i][j] = (i == j) ? 1.0 : 0.0; }
  for (int sweep = 0; sweep < ORC_JACOBI_SWEEPS; sweep++) {
    double off = 0, dia = 0;
    for (int p = 0; p < 6; p++) {
      dia = dia + A[p][p] * A[p][p];
      for (int q = p + 1; q < 6; q++) off = off + A[p][q] * A[p][q];
    }
    if (off <= 1e-40 * dia) break;
    for (int p = 0; p < 5; p++) {
      for (int q = p + 1; q < 6; q++) {
        const double apq = A[p][q];
        if (apq == 0) continue;
        const double theta = (A[q][q] - A[p][p]) / (2 * apq);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1));
        const double c = 1 / sqrt(t * t + 1), s = t * c;
        for (int k = 0; k < 6; k++) {  /* columns p, q of A and V */
          const double akp = A[k][p], akq = A[k][q];
          A[k][p] = c * akp - s * akq;
          A[k][q] = s * akp + c * akq;
          const double vkp = V[k][p], vkq = V[k][q];
          V[k][p] = c * vkp - s * vkq;
          V[k][q] = s * vkp + c * vkq;
        }
        for (int k = 0; k < 6; k++) {  /* rows p, q of A */
          const double apk = A[p][k], aqk = A[q][k];
          A[p][k] = c * apk - s * aqk;
          A[q][k] = s * apk + c * aqk;
        }
        A[p][q] = 0; A[q][p] = 0;
      }
    }
  }
  double wmax = 0, inv[6];
  for (int i = 0; i < 6; i++) if (fabs(A[i][i]) > wmax) wmax = fabs(A[i][i]);
  for (int i = 0; i < 6; i++) inv[i] = (fabs(A[i][i]) > 1e-15 * wmax) ? 1 / A[i][i] : 0;
  for (int i = 0; i < 6; i++)
    for (int j = 0; j < 6; j++) {
      double acc = 0;
      for (int k = 0; k < 6; k++) acc = acc + (V[i][k] * inv[k]) * V[j][k];
      out[6 * i + j] = acc;
    }
}

static double norm_seq(const double *v, int n) {
  double s = 0;
  for (int i = 0; i < n; i++) s = s + v[i] * v[i];
  return sqrt(s);
}

static int within_limits(const orc_model *m, const orc_pose *p, const double *q) {
  for (int j = 0; j < m->njnt; j++)
    if (!(q[j] >= p->jnt_range[2 * j] && q[j] <= p->jnt_range[2 * j + 1])) return 0;
  return 1;
}

int orc_pose_valid(const orc_model *m, const orc_pose *p, const double *qpos) {
  double dx[6];
  if (!within_limits(m, p, qpos)) return 0;
  int rc = orc_pose_displacement(m, p, qpos, dx);
  if (rc != ORC_OK) return rc;
  return norm_seq(dx, 6) <= p->tolerance ? 1 : 0;
}

int orc_pose_apply(const orc_model *m, const orc_pose *p, const double *q_old, const double *q,
                   double *q_out, int32_t *iters) {
  const int nq = m->nq, nv = m->njnt;
  if (nv > ORC_MAXCHAIN || nq != nv) return ORC_E_JOINT;
  double J[6 * ORC_MAXCHAIN], dx[6], A[36], P[36], y[6], diff[ORC_MAXCHAIN];
  chain_kin ck;
  memcpy(q_out, q, sizeof(double) * (size_t)nq);
  const int maxit = p->max_iters > 0 ? p->max_iters : 1000;
  for (int it = 0;; it++) {
    if (iters) *iters = it;
    int rc = chain_kinematics(m, p, q_out, &ck);
    if (rc != ORC_OK) return rc;
    displacement(p, &ck, dx);
    if (norm_seq(dx, 6) <= p->tolerance) return 1;
    if (it >= maxit) return ORC_E_NOCONVERGE;
    jacobian(m, &ck, J);
    for (int i = 0; i < 6; i++)
      for (int k = 0; k < 6; k++) {
        double acc = 0;
        for (int d = 0; d < nv; d++) acc = acc + J[i * nv + d] * J[k * nv + d];
        A[6 * i + k] = acc;
      }
    orc_pinv_sym6(A, P);
    for (int i = 0; i < 6; i++) {
      double acc = 0;
      for (int k = 0; k < 6; k++) acc = acc + P[6 * i + k] * dx[k];
      y[i] = acc;
    }
    for (int d = 0; d < nv; d++) {
      double acc = 0;
      for (int i = 0; i < 6; i++) acc = acc + J[i * nv + d] * y[i];
      q_out[m->jnt_qposadr[d]] -= acc;
    }
    for (int k = 0; k < nq; k++) diff[k] = q_out[k] - q_old[k];
    if (!within_limits(m, p, q_out) || norm_seq(diff, nq) > 2 * p->q_step) {
      if (iters) *iters = it + 1;  /* the rejected step counts as taken */
      return 0;
    }
  }
}

typedef struct pose_job {
  const orc_model *m; const orc_pose *p; const double *Q_old, *Q; double *Q_out;
  uint8_t *ok; int32_t *iters; int64_t lo, hi; int status;
} pose_job;

static void *pose_job_run(void *arg) {
  pose_job *j = (pose_job *)arg;
  const int nq = j->m->nq;
  for (int64_t i = j->lo; i < j->hi; i++) {
    int32_t it = 0;
    int rc = orc_pose_apply(j->m, j->p, j->Q_old + i * nq, j->Q + i * nq, j->Q_out + i * nq, &it);
    j->ok[i] = rc == 1;
    if (j->iters) j->iters[i] = (rc == ORC_E_NOCONVERGE) ? -it : it;  /* negative: gave up after max_iters */
    if (rc < 0 && rc != ORC_E_NOCONVERGE) j->status = rc;
  }
  return NULL;
}

int orc_pose_apply_batch(const orc_model *m, const orc_pose *p, const double *Q_old, const double *Q,
                         int64_t N, int32_t nthreads, double *Q_out, uint8_t *ok, int32_t *iters) {
  if (nthreads < 1) nthreads = 1;
  if (nthreads > 512) nthreads = 512;
  pose_job jobs[512];
  pthread_t th[512];
  const int64_t chunk = (N + nthreads - 1) / nthreads;
  int status = ORC_OK;
  for (int t = 0; t < nthreads; t++) {
    pose_job jb = {m, p, Q_old, Q, Q_out, ok, iters, t * chunk, (t + 1) * chunk < N ? (t + 1) * chunk : N, ORC_OK};
    if (jb.lo > N) jb.lo = N;
    jobs[t] = jb;
    if (nthreads == 1) pose_job_run(&jobs[t]);
    else pthread_create(&th[t], NULL, pose_job_run, &jobs[t]);
  }
  for (int t = 0; t < nthreads; t++) {
    if (nthreads > 1) pthread_join(th[t], NULL);
    if (jobs[t].status != ORC_OK) status = jobs[t].status;
  }
  return status;
}

/* ------------------------------------------------------------------ IK seeds
 * CPU statement of the damped-least-squares iteration of the product's k_ik_solve (the role of
 * MinkIKSolver.solve_ik, src/mjpl/inverse_kinematics/mink_ik_solver.py:72-116, whose own arithmetic
 * is a QP in the un-vendored mink / daqp wheels): world-frame 6-D pose error, J^T (J J^T + lam I)^-1 e
 * with error-proportional, adaptively scaled damping, step-length limit, joint-range clamp with an
 * active set, restarts from a uniform draw when a row stalls.  Test infrastructure and CPU baseline
 * only; parity with the GPU is tolerance-level (both must reach the pose within the tolerances). */
static uint64_t ik_sm64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
static double ik_u01(uint64_t key, uint64_t ctr) {
  return (double)(ik_sm64(key + ctr * 0x9E3779B97F4A7C15ull) >> 11) * 0x1.0p-53;
}

static void ik_pose_error(const orc_ik *d, const chain_kin *ck, double *e) {
  double qs[4], qe[4];
  mat2quat(qs, ck->site_xmat);
  const double qc[4] = {qs[0], -qs[1], -qs[2], -qs[3]};
  mul_quat(qe, d->target_quat, qc);
  const double sgn = qe[0] < 0 ? -1.0 : 1.0;
  const double w = sgn * qe[0], v[3] = {sgn * qe[1], sgn * qe[2], sgn * qe[3]};
  const double sn = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
  const double k = sn > 1e-12 ? 2 * atan2(sn, w) / sn : 2.0;
  for (int c = 0; c < 3; c++) { e[c] = d->target_pos[c] - ck->site_xpos[c]; e[3 + c] = k * v[c]; }
}

static void chol6(double A[6][6], const double *b, double *x) {
  double L[6][6] = {{0}};
  for (int i = 0; i < 6; i++)
    for (int j = 0; j <= i; j++) {
      double s = A[i][j];
      for (int k = 0; k < j; k++) s -= L[i][k] * L[j][k];
      L[i][j] = i == j ? sqrt(s > 1e-300 ? s : 1e-300) : s / L[j][j];
    }
  double y[6];
  for (int i = 0; i < 6; i++) {
    double s = b[i];
    for (int k = 0; k < i; k++) s -= L[i][k] * y[k];
    y[i] = s / L[i][i];
  }
  for (int i = 5; i >= 0; i--) {
    double s = y[i];
    for (int k = i + 1; k < 6; k++) s -= L[k][i] * x[k];
    x[i] = s / L[i][i];
  }
}

int orc_ik_solve(const orc_model *m, const orc_ik *d, const double *q0, int64_t row, double *q_out,
                 int32_t *iters, double *err2) {
  orc_pose ps;
  memset(&ps, 0, sizeof(ps));
  ps.site_body = d->site_body;
  memcpy(ps.site_pos, d->site_pos, sizeof(ps.site_pos));
  memcpy(ps.site_quat, d->site_quat, sizeof(ps.site_quat));
  const int nq = m->nq;
  double q[ORC_MAXCHAIN * 4];
  if (nq > ORC_MAXCHAIN * 4) return ORC_E_OVERFLOW;
  memcpy(q, q0, sizeof(double) * (size_t)nq);
  const double damp = d->damping > 0 ? d->damping : 1e-6, lm = d->lm_damping >= 0 ? d->lm_damping : 0.1;
  const double max_step = d->max_step > 0 ? d->max_step : 0.2;
  double lam_scale = 1.0, prev = 1e300, best = 1e300, epos = 0, eori = 0;
  int best_it = 0, restarts = 0, it = 0, solved = 0;
  const uint64_t key = ik_sm64(ik_sm64(d->restart_seed) ^ ik_sm64(((uint64_t)row << 40) ^ 0x494bull));
  for (;;) {
    chain_kin ck;
    int rc = chain_kinematics(m, &ps, q, &ck);
    if (rc != ORC_OK) return rc;
    double e[6];
    ik_pose_error(d, &ck, e);
    epos = sqrt(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]);
    eori = sqrt(e[3] * e[3] + e[4] * e[4] + e[5] * e[5]);
    if (epos <= d->pos_tolerance && eori <= d->ori_tolerance) { solved = 1; break; }
    if (it >= d->iterations) break;
    const double e2 = epos * epos + eori * eori;
    if (e2 < 0.98 * best) { best = e2; best_it = it; }
    if (it - best_it >= 12 && restarts < d->restarts) {
      restarts++;
      for (int k = 0; k < ck.njoint; k++) {
        const int jid = ck.jid[k];
        if (d->movable[jid]) {
          const double u = ik_u01(key, (uint64_t)restarts * 64u + (uint64_t)(jid & 63));
          q[m->jnt_qposadr[jid]] = d->jnt_range[2 * jid] + u * (d->jnt_range[2 * jid + 1] - d->jnt_range[2 * jid]);
        }
      }
      lam_scale = 1.0; prev = 1e300; best = 1e300; best_it = it;
      it++;
      continue;
    }
    lam_scale = e2 > prev ? fmin(lam_scale * 4.0, 1e4) : fmax(lam_scale * 0.5, 1.0 / 64.0);
    prev = e2;
    const double lam = (damp + lm * e2) * lam_scale;
    double col[ORC_MAXCHAIN][6], dq[ORC_MAXCHAIN];
    for (int k = 0; k < ck.njoint; k++) {
      const int jid = ck.jid[k];
      const double mv = d->movable[jid] ? 1.0 : 0.0, *ax = ck.xaxis[k];
      if (m->jnt_type[jid] == ORC_JNT_HINGE) {
        const double r[3] = {ck.site_xpos[0] - ck.xanchor[k][0], ck.site_xpos[1] - ck.xanchor[k][1],
                             ck.site_xpos[2] - ck.xanchor[k][2]};
        col[k][0] = mv * (ax[1] * r[2] - ax[2] * r[1]);
        col[k][1] = mv * (ax[2] * r[0] - ax[0] * r[2]);
        col[k][2] = mv * (ax[0] * r[1] - ax[1] * r[0]);
        col[k][3] = mv * ax[0]; col[k][4] = mv * ax[1]; col[k][5] = mv * ax[2];
      } else {
        col[k][0] = mv * ax[0]; col[k][1] = mv * ax[1]; col[k][2] = mv * ax[2];
        col[k][3] = col[k][4] = col[k][5] = 0;
      }
    }
    unsigned locked = 0;
    double scale = 1.0;
    for (int pass = 0; pass < 2; pass++) {
      double A[6][6], y[6];
      for (int r = 0; r < 6; r++)
        for (int c = 0; c < 6; c++) A[r][c] = r == c ? lam : 0.0;
      for (int k = 0; k < ck.njoint; k++) {
        if ((locked >> (k & 31)) & 1u) continue;
        for (int r = 0; r < 6; r++)
          for (int c = 0; c < 6; c++) A[r][c] += col[k][r] * col[k][c];
      }
      chol6(A, e, y);
      double big = 0;
      for (int k = 0; k < ck.njoint; k++) {
        double acc = 0;
        if (!((locked >> (k & 31)) & 1u))
          for (int r = 0; r < 6; r++) acc += col[k][r] * y[r];
        dq[k] = acc;
        if (fabs(acc) > big) big = fabs(acc);
      }
      scale = big > max_step ? max_step / big : 1.0;
      if (pass == 1 || ck.njoint > 32) break;
      unsigned out = 0;
      for (int k = 0; k < ck.njoint; k++) {
        const int jid = ck.jid[k];
        const double v = q[m->jnt_qposadr[jid]], lo = d->jnt_range[2 * jid], hi = d->jnt_range[2 * jid + 1];
        const double span = hi - lo;
        if (d->movable[jid] && ((v <= lo + 1e-9 * span && dq[k] < 0) || (v >= hi - 1e-9 * span && dq[k] > 0)))
          out |= 1u << (k & 31);
      }
      if (!out) break;
      locked = out;
    }
    for (int k = 0; k < ck.njoint; k++) {
      const int jid = ck.jid[k];
      if (!d->movable[jid]) continue;
      double v = q[m->jnt_qposadr[jid]] + scale * dq[k];
      v = v < d->jnt_range[2 * jid] ? d->jnt_range[2 * jid] : v;
      v = v > d->jnt_range[2 * jid + 1] ? d->jnt_range[2 * jid + 1] : v;
      q[m->jnt_qposadr[jid]] = v;
    }
    it++;
  }
  memcpy(q_out, q, sizeof(double) * (size_t)nq);
  if (iters) *iters = it;
  if (err2) { err2[0] = epos; err2[1] = eori; }
  return solved;
}

typedef struct ik_job {
  const orc_model *m; const orc_ik *d; const double *Q; double *Q_out; uint8_t *ok; int32_t *iters; double *err;
  int64_t lo, hi; int status;
} ik_job;

static void *ik_job_run(void *arg) {
  ik_job *j = (ik_job *)arg;
  const int nq = j->m->nq;
  for (int64_t i = j->lo; i < j->hi; i++) {
    int32_t it = 0;
    double e2[2] = {0, 0};
    int rc = orc_ik_solve(j->m, j->d, j->Q + i * nq, i, j->Q_out + i * nq, &it, e2);
    j->ok[i] = rc == 1;
    if (j->iters) j->iters[i] = it;
    if (j->err) { j->err[2 * i] = e2[0]; j->err[2 * i + 1] = e2[1]; }
    if (rc < 0) j->status = rc;
  }
  return NULL;
}

int orc_ik_solve_batch(const orc_model *m, const orc_ik *d, const double *Q, int64_t N, int32_t nthreads,
                       double *Q_out, uint8_t *ok, int32_t *iters, double *err) {
  if (nthreads < 1) nthreads = 1;
  if (nthreads > 512) nthreads = 512;
  ik_job jobs[512];
  pthread_t th[512];
  const int64_t chunk = (N + nthreads - 1) / nthreads;
  int status = ORC_OK;
  for (int t = 0; t < nthreads; t++) {
    ik_job jb = {m, d, Q, Q_out, ok, iters, err, t * chunk, (t + 1) * chunk < N ? (t + 1) * chunk : N, ORC_OK};
    if (jb.lo > N) jb.lo = N;
    jobs[t] = jb;
    if (nthreads == 1) ik_job_run(&jobs[t]);
    else pthread_create(&th[t], NULL, ik_job_run, &jobs[t]);
  }
  for (int t = 0; t < nthreads; t++) {
    if (nthreads > 1) pthread_join(th[t], NULL);
    if (jobs[t].status != ORC_OK) status = jobs[t].status;
  }
  return status;
}
