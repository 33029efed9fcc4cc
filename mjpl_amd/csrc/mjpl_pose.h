// mjpl_pose.h -- device code of the batched PoseConstraint:
// PoseConstraint.valid_config / apply (src/mjpl/constraint/pose_constraint.py:72-171) for one
// configuration per lane, float64.  FK runs along the site's ancestor chain only and keeps what
// mj_jacSite needs (world axis + anchor of every joint on the chain); the 6x6 pseudo-inverse of
// J J^T is a fully unrolled cyclic-Jacobi eigen-decomposition held in registers.
//
// Upstream arithmetic restated from published algorithms: mj_kinematics / mj_jacSite /
// mju_mat2Quat [MJ-recalled], mink.lie SE3/SO3 [MINK-recalled], np.linalg.pinv (cutoff 1e-15 x
// largest singular value).  Operation order follows oracle/mjpl_oracle_pose.c.
#pragma once

#include "mjpl_device.h"

namespace mjpl {

// pose program, int words: header, then per chain body {njnt}, per joint {type, qadr, dof}
enum : int { PH_NBODY = 0, PH_NJOINT, PH_NQ, PH_MAXIT, PH_OFF_JRANGE, PH_OFF_TAIL, PH_SIZE };
// doubles: per body pos[3] quat[4]; per joint axis[3] pos[3] qpos0; then at PH_OFF_TAIL:
enum : int { PT_SITE_POS = 0, PT_SITE_QUAT = 3, PT_C_QUAT = 7, PT_C_POS = 11, PT_LO = 14, PT_HI = 20,
             PT_TOL = 26, PT_QSTEP = 27, PT_SIZE = 28 };
// ... and at PH_OFF_JRANGE: jnt_range[njnt][2]

__device__ __forceinline__ void mat2quat(double *quat, const double *mat) {  // mju_mat2Quat
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

__device__ __forceinline__ void so3_apply(double *res, const double *quat, const double *vec) {
  const double pv[4] = {0, vec[0], vec[1], vec[2]};
  const double qi[4] = {quat[0], -quat[1], -quat[2], -quat[3]};
  double t[4], r[4];
  mul_quat(t, quat, pv);
  mul_quat(r, t, qi);
  res[0] = r[1]; res[1] = r[2]; res[2] = r[3];
}

__device__ __forceinline__ void quat2rpy(double *rpy, const double *q) {
  rpy[0] = atan2(2 * (q[0] * q[1] + q[2] * q[3]), 1 - 2 * (q[1] * q[1] + q[2] * q[2]));
  rpy[1] = asin(2 * (q[0] * q[2] - q[3] * q[1]));
  rpy[2] = atan2(2 * (q[0] * q[3] + q[1] * q[2]), 1 - 2 * (q[2] * q[2] + q[3] * q[3]));
}

// One Jacobi rotation in the (P, Q) plane; literal indices keep A and V in registers.
template <int P, int Q>
__device__ __forceinline__ void jacobi_rotate(double (&A)[6][6], double (&V)[6][6]) {
  const double apq = A[P][Q];
  if (apq == 0) return;
  const double theta = (A[Q][Q] - A[P][P]) / (2 * apq);
  const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1));
  const double c = 1 / sqrt(t * t + 1), s = t * c;
#pragma unroll
  for (int k = 0; k < 6; k++) {
    const double akp = A[k][P], akq = A[k][Q];
    A[k][P] = c * akp - s * akq;
    A[k][Q] = s * akp + c * akq;
    const double vkp = V[k][P], vkq = V[k][Q];
    V[k][P] = c * vkp - s * vkq;
    V[k][Q] = s * vkp + c * vkq;
  }
#pragma unroll
  for (int k = 0; k < 6; k++) {
    const double apk = A[P][k], aqk = A[Q][k];
    A[P][k] = c * apk - s * aqk;
    A[Q][k] = s * apk + c * aqk;
  }
  A[P][Q] = 0; A[Q][P] = 0;
}

// y = pinv(A) x for symmetric A (np.linalg.pinv semantics); A is destroyed.  Not inlined: it is the path of
// the rare ill-conditioned step, and its 72 doubles of A and V inside the caller cost the common path its
// registers (k_rrt_gen_project: 328 registers and 272 bytes of scratch with it inlined).
__device__ __attribute__((noinline)) void pinv_sym6_apply(double (&A)[6][6], const double *x, double *y) {
  double V[6][6];
#pragma unroll
  for (int i = 0; i < 6; i++)
#pragma unroll
    for (int j = 0; j < 6; j++) V[i][j] = (i == j) ? 1.0 : 0.0;
  // sweeps until the off-diagonal mass is below 1e-40 of the diagonal mass (the oracle's rule)
#pragma unroll 1
  for (int sweep = 0; sweep < 12; sweep++) {
    double off = 0, dia = 0;
#pragma unroll
    for (int p = 0; p < 6; p++) {
      dia = dia + A[p][p] * A[p][p];
#pragma unroll
      for (int q = p + 1; q < 6; q++) off = off + A[p][q] * A[p][q];
    }
    if (off <= 1e-40 * dia) break;
    jacobi_rotate<0, 1>(A, V); jacobi_rotate<0, 2>(A, V); jacobi_rotate<0, 3>(A, V);
    jacobi_rotate<0, 4>(A, V); jacobi_rotate<0, 5>(A, V); jacobi_rotate<1, 2>(A, V);
    jacobi_rotate<1, 3>(A, V); jacobi_rotate<1, 4>(A, V); jacobi_rotate<1, 5>(A, V);
    jacobi_rotate<2, 3>(A, V); jacobi_rotate<2, 4>(A, V); jacobi_rotate<2, 5>(A, V);
    jacobi_rotate<3, 4>(A, V); jacobi_rotate<3, 5>(A, V); jacobi_rotate<4, 5>(A, V);
  }
  double wmax = 0, inv[6];
#pragma unroll
  for (int i = 0; i < 6; i++) wmax = fabs(A[i][i]) > wmax ? fabs(A[i][i]) : wmax;
#pragma unroll
  for (int i = 0; i < 6; i++) inv[i] = (fabs(A[i][i]) > 1e-15 * wmax) ? 1 / A[i][i] : 0.0;
  // P[i][j] = sum_k (V[i][k] inv[k]) V[j][k];  y[i] = sum_j P[i][j] x[j]   (the oracle's order)
#pragma unroll
  for (int i = 0; i < 6; i++) {
    double yi = 0;
#pragma unroll
    for (int j = 0; j < 6; j++) {
      double pij = 0;
#pragma unroll
      for (int k = 0; k < 6; k++) pij = pij + (V[i][k] * inv[k]) * V[j][k];
      yi = yi + pij * x[j];
    }
    y[i] = yi;
  }
}

struct PoseChainOut {
  double site_xpos[3], site_xmat[9];
};

// mj_kinematics along the chain.  q: this lane's qpos, q[k * qs].  jst (nullable): this lane's
// [6][njoint] store (element (r, k) at jst[(r * nj + k) * js]) receiving xaxis (rows 0..2) and
// xanchor (rows 3..5) of every chain joint.
__device__ __forceinline__ void pose_chain(const int *__restrict__ pi, const double *__restrict__ pd,
                                           const double *q, int qs, double *jst, int js,
                                           PoseChainOut &out) {
  const int nb = pi[PH_NBODY], nj = pi[PH_NJOINT];
  double p[3] = {0, 0, 0}, qt[4] = {1, 0, 0, 0}, R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  int ic = PH_SIZE, dc = 0, jk = 0;
  for (int b = 0; b < nb; b++) {
    const int njnt = pi[ic++];
    double np[3], nq[4];
    {
      const double bpos[3] = {pd[dc], pd[dc + 1], pd[dc + 2]};
      const double bquat[4] = {pd[dc + 3], pd[dc + 4], pd[dc + 5], pd[dc + 6]};
      dc += 7;
      mul_mat_vec3(np, R, bpos);
      np[0] += p[0]; np[1] += p[1]; np[2] += p[2];
      mul_quat(nq, qt, bquat);
    }
    for (int j = 0; j < njnt; j++, jk++) {
      const int jtype = pi[ic], qadr = pi[ic + 1];
      ic += 3;
      const double jaxis[3] = {pd[dc], pd[dc + 1], pd[dc + 2]};
      const double jpos[3] = {pd[dc + 3], pd[dc + 4], pd[dc + 5]};
      const double dq = q[qadr * qs] - pd[dc + 6];
      dc += 7;
      double xaxis[3], xanchor[3];
      rot_vec_quat(xaxis, jaxis, nq);
      rot_vec_quat(xanchor, jpos, nq);
      xanchor[0] += np[0]; xanchor[1] += np[1]; xanchor[2] += np[2];
      if (jst) {
#pragma unroll
        for (int r = 0; r < 3; r++) {
          jst[(r * nj + jk) * js] = xaxis[r];
          jst[((3 + r) * nj + jk) * js] = xanchor[r];
        }
      }
      if (jtype == JT_SLIDE) {
        np[0] += xaxis[0] * dq; np[1] += xaxis[1] * dq; np[2] += xaxis[2] * dq;
      } else {
        double sn, cs, vec[3];
        sincos_half(dq * 0.5, &sn, &cs);
        const double qloc[4] = {cs, jaxis[0] * sn, jaxis[1] * sn, jaxis[2] * sn};
        mul_quat(nq, nq, qloc);
        rot_vec_quat(vec, jpos, nq);
        np[0] = xanchor[0] - vec[0]; np[1] = xanchor[1] - vec[1]; np[2] = xanchor[2] - vec[2];
      }
    }
    normalize4(nq);
#pragma unroll
    for (int k = 0; k < 3; k++) p[k] = np[k];
#pragma unroll
    for (int k = 0; k < 4; k++) qt[k] = nq[k];
    quat2mat(R, qt);
  }
  const double *tail = pd + pi[PH_OFF_TAIL];
  const double spos[3] = {tail[PT_SITE_POS], tail[PT_SITE_POS + 1], tail[PT_SITE_POS + 2]};
  const double squat[4] = {tail[PT_SITE_QUAT], tail[PT_SITE_QUAT + 1], tail[PT_SITE_QUAT + 2],
                           tail[PT_SITE_QUAT + 3]};
  double sp[3], sq[4];
  mul_mat_vec3(sp, R, spos);
  out.site_xpos[0] = sp[0] + p[0]; out.site_xpos[1] = sp[1] + p[1]; out.site_xpos[2] = sp[2] + p[2];
  mul_quat(sq, qt, squat);
  quat2mat(out.site_xmat, sq);
}

// _displacement_from_constraint (pose_constraint.py:93-123); also returns the site's world
// quaternion (SO3.from_matrix) for the Jacobian's E_rpy.
__device__ __forceinline__ void pose_displacement(const double *tail, const PoseChainOut &o, double *dx,
                                                  double *qsite) {
  const double cq[4] = {tail[PT_C_QUAT], tail[PT_C_QUAT + 1], tail[PT_C_QUAT + 2], tail[PT_C_QUAT + 3]};
  double qc[4], t[3], rpy[3], d[6];
  mat2quat(qsite, o.site_xmat);
  mul_quat(qc, cq, qsite);
  so3_apply(t, cq, o.site_xpos);
  d[0] = t[0] + tail[PT_C_POS]; d[1] = t[1] + tail[PT_C_POS + 1]; d[2] = t[2] + tail[PT_C_POS + 2];
  quat2rpy(rpy, qc);
  d[3] = rpy[0]; d[4] = rpy[1]; d[5] = rpy[2];
#pragma unroll
  for (int k = 0; k < 6; k++) {
    double v = 0;
    if (d[k] > tail[PT_HI + k]) v = d[k] - tail[PT_HI + k];
    if (d[k] < tail[PT_LO + k]) v = d[k] - tail[PT_LO + k];
    dx[k] = v;
  }
}

__device__ __forceinline__ double norm6(const double *v) {
  double s = 0;
#pragma unroll
  for (int k = 0; k < 6; k++) s = s + v[k] * v[k];
  return sqrt(s);
}

// ---- row f3: batched damped-least-squares IK seeds -----------------------------------------
// IK tail (doubles at PH_OFF_TAIL of an IK program): site pose offsets as above, then
enum : int { IT_SITE_POS = 0, IT_SITE_QUAT = 3, IT_TGT_POS = 7, IT_TGT_QUAT = 10, IT_POS_TOL = 14,
             IT_ORI_TOL = 15, IT_DAMP = 16, IT_LM = 17, IT_MAX_STEP = 18, IT_SIZE = 19 };
// ... followed at PH_OFF_JRANGE by jnt_range[njnt][2] and then movable[njnt] (1.0 / 0.0)

// Solve (A) y = b for a symmetric positive definite 6x6 by an unrolled Cholesky factorisation.
__device__ __forceinline__ void chol6_solve(double (&A)[6][6], const double *b, double *y) {
  double L[6][6];
#pragma unroll
  for (int j = 0; j < 6; j++) {
    double d = A[j][j];
#pragma unroll
    for (int k = 0; k < j; k++) d = d - L[j][k] * L[j][k];
    d = sqrt(d > 1e-300 ? d : 1e-300);
    L[j][j] = d;
    const double inv = 1 / d;
#pragma unroll
    for (int i = j + 1; i < 6; i++) {
      double v = A[i][j];
#pragma unroll
      for (int k = 0; k < j; k++) v = v - L[i][k] * L[j][k];
      L[i][j] = v * inv;
    }
  }
  double z[6];
#pragma unroll
  for (int i = 0; i < 6; i++) {
    double v = b[i];
#pragma unroll
    for (int k = 0; k < i; k++) v = v - L[i][k] * z[k];
    z[i] = v / L[i][i];
  }
#pragma unroll
  for (int i = 5; i >= 0; i--) {
    double v = z[i];
#pragma unroll
    for (int k = i + 1; k < 6; k++) v = v - L[k][i] * y[k];
    y[i] = v / L[i][i];
  }
}

// y = A^-1 b for a symmetric positive definite, well-conditioned 6x6: Cholesky factor L, its inverse
// M (so that A^-1 = M^T M), and the certificate  cond_2(A) <= tr(A) tr(A^-1) = tr(A) ||M||_F^2.
// Returns false -- y untouched, A intact -- unless that bound is below `max_cond`: then the caller
// takes the eigen-decomposition (np.linalg.pinv's cut-off semantics matter only there).  For a
// matrix that passes, A^-1 b and pinv(A) b are the same vector up to cond * 2^-53 relative.
__device__ __forceinline__ bool spd6_solve_certified(const double (&A)[6][6], const double *b, double *y,
                                                     double max_cond) {
  double L[6][6], M[6][6];
  bool good = true;
  double tr = 0;
#pragma unroll
  for (int j = 0; j < 6; j++) {
    tr = tr + A[j][j];
    double d = A[j][j];
#pragma unroll
    for (int k = 0; k < j; k++) d = d - L[j][k] * L[j][k];
    good = good && (d > 0.0);  // (also false for NaN)
    d = sqrt(good ? d : 1.0);
    L[j][j] = d;
    const double inv = 1 / d;
    M[j][j] = inv;
#pragma unroll
    for (int i = j + 1; i < 6; i++) {
      double v = A[i][j];
#pragma unroll
      for (int k = 0; k < j; k++) v = v - L[i][k] * L[j][k];
      L[i][j] = v * inv;
    }
  }
  // M = L^-1 (lower triangular), column by column
  double fro = 0;
#pragma unroll
  for (int j = 0; j < 6; j++) {
#pragma unroll
    for (int i = j + 1; i < 6; i++) {
      double v = 0;
#pragma unroll
      for (int k = j; k < i; k++) v = v - L[i][k] * M[k][j];
      M[i][j] = v * M[i][i];
    }
#pragma unroll
    for (int i = j; i < 6; i++) fro = fro + M[i][j] * M[i][j];
  }
  good = good && (tr * fro <= max_cond);
  if (!good) return false;
  double z[6];
#pragma unroll
  for (int i = 0; i < 6; i++) {
    double v = 0;
#pragma unroll
    for (int k = 0; k <= i; k++) v = v + M[i][k] * b[k];
    z[i] = v;
  }
#pragma unroll
  for (int i = 0; i < 6; i++) {
    double v = 0;
#pragma unroll
    for (int k = i; k < 6; k++) v = v + M[k][i] * z[k];
    y[i] = v;
  }
  return true;
}

// world-frame pose error of the site against the target: e[0..2] = p_t - p,
// e[3..5] = rotation vector of R_t R^T (quaternion logarithm, shortest way round)
__device__ __forceinline__ void ik_error(const double *tail, const PoseChainOut &o, double *e) {
  double qs[4], qe[4];
  mat2quat(qs, o.site_xmat);
  const double qt[4] = {tail[IT_TGT_QUAT], tail[IT_TGT_QUAT + 1], tail[IT_TGT_QUAT + 2], tail[IT_TGT_QUAT + 3]};
  const double qc[4] = {qs[0], -qs[1], -qs[2], -qs[3]};
  mul_quat(qe, qt, qc);
  const double sgn = qe[0] < 0 ? -1.0 : 1.0;
  const double w = sgn * qe[0], v[3] = {sgn * qe[1], sgn * qe[2], sgn * qe[3]};
  const double sn = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
  const double k = sn > 1e-12 ? 2 * atan2(sn, w) / sn : 2.0;
  e[0] = tail[IT_TGT_POS] - o.site_xpos[0];
  e[1] = tail[IT_TGT_POS + 1] - o.site_xpos[1];
  e[2] = tail[IT_TGT_POS + 2] - o.site_xpos[2];
  e[3] = k * v[0]; e[4] = k * v[1]; e[5] = k * v[2];
}

}  // namespace mjpl
