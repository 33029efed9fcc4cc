/*
 * orc_math.h -- CPU ORACLE (test infrastructure, NOT the product): the small vector /
 * quaternion helpers shared by mjpl_oracle.c and mjpl_oracle_pose.c.
 */
#ifndef ORC_MATH_H
#define ORC_MATH_H

#include <math.h>

#define ORC_MINVAL 1e-15 /* mjMINVAL [MJ-recalled: mjmodel.h] */
#define ORC_UNUSED __attribute__((unused))

/* ------------------------------------------------------------------ small vector helpers
 * [MJ-recalled: engine_util_blas.c / engine_util_spatial.c]; operation order is the contract. */

ORC_UNUSED static double dot3(const double *a, const double *b) {
  return a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
}

ORC_UNUSED static void mul_mat_vec3(double *res, const double *mat, const double *vec) {
  res[0] = mat[0] * vec[0] + mat[1] * vec[1] + mat[2] * vec[2];
  res[1] = mat[3] * vec[0] + mat[4] * vec[1] + mat[5] * vec[2];
  res[2] = mat[6] * vec[0] + mat[7] * vec[1] + mat[8] * vec[2];
}

ORC_UNUSED static void mul_matT_vec3(double *res, const double *mat, const double *vec) {
  res[0] = mat[0] * vec[0] + mat[3] * vec[1] + mat[6] * vec[2];
  res[1] = mat[1] * vec[0] + mat[4] * vec[1] + mat[7] * vec[2];
  res[2] = mat[2] * vec[0] + mat[5] * vec[1] + mat[8] * vec[2];
}

/* mju_mulQuat */
ORC_UNUSED static void mul_quat(double *res, const double *a, const double *b) {
  double t0 = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
  double t1 = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
  double t2 = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
  double t3 = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
  res[0] = t0; res[1] = t1; res[2] = t2; res[3] = t3;
}

/* mju_rotVecQuat (3.x form: v + 2*cross(q_xyz, q_w*v + cross(q_xyz, v))) */
ORC_UNUSED static void rot_vec_quat(double *res, const double *vec, const double *quat) {
  if (vec[0] == 0 && vec[1] == 0 && vec[2] == 0) {
    res[0] = res[1] = res[2] = 0;
  } else if (quat[0] == 1 && quat[1] == 0 && quat[2] == 0 && quat[3] == 0) {
    res[0] = vec[0]; res[1] = vec[1]; res[2] = vec[2];
  } else {
    double t0 = quat[0] * vec[0] + quat[2] * vec[2] - quat[3] * vec[1];
    double t1 = quat[0] * vec[1] + quat[3] * vec[0] - quat[1] * vec[2];
    double t2 = quat[0] * vec[2] + quat[1] * vec[1] - quat[2] * vec[0];
    double r0 = vec[0] + 2 * (quat[2] * t2 - quat[3] * t1);
    double r1 = vec[1] + 2 * (quat[3] * t0 - quat[1] * t2);
    double r2 = vec[2] + 2 * (quat[1] * t1 - quat[2] * t0);
    res[0] = r0; res[1] = r1; res[2] = r2;
  }
}

/* mju_axisAngle2Quat */
ORC_UNUSED static void axis_angle2quat(double *res, const double *axis, double angle) {
  if (angle == 0) {
    res[0] = 1; res[1] = 0; res[2] = 0; res[3] = 0;
  } else {
    double s = sin(angle * 0.5);
    res[0] = cos(angle * 0.5);
    res[1] = axis[0] * s;
    res[2] = axis[1] * s;
    res[3] = axis[2] * s;
  }
}

/* mju_normalize4 */
ORC_UNUSED static void normalize4(double *v) {
  double norm = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3]);
  if (norm < ORC_MINVAL) {
    v[0] = 1; v[1] = 0; v[2] = 0; v[3] = 0;
  } else if (fabs(norm - 1) > ORC_MINVAL) {
    double inv = 1 / norm;
    v[0] *= inv; v[1] *= inv; v[2] *= inv; v[3] *= inv;
  }
}

/* mju_quat2Mat */
ORC_UNUSED static void quat2mat(double *res, const double *q) {
  if (q[0] == 1 && q[1] == 0 && q[2] == 0 && q[3] == 0) {
    res[0] = 1; res[1] = 0; res[2] = 0;
    res[3] = 0; res[4] = 1; res[5] = 0;
    res[6] = 0; res[7] = 0; res[8] = 1;
  } else {
    const double q00 = q[0] * q[0], q01 = q[0] * q[1], q02 = q[0] * q[2], q03 = q[0] * q[3];
    const double q11 = q[1] * q[1], q12 = q[1] * q[2], q13 = q[1] * q[3];
    const double q22 = q[2] * q[2], q23 = q[2] * q[3], q33 = q[3] * q[3];
    res[0] = q00 + q11 - q22 - q33;
    res[4] = q00 - q11 + q22 - q33;
    res[8] = q00 - q11 - q22 + q33;
    res[1] = 2 * (q12 - q03);
    res[2] = 2 * (q13 + q02);
    res[3] = 2 * (q12 + q03);
    res[5] = 2 * (q23 - q01);
    res[6] = 2 * (q13 - q02);
    res[7] = 2 * (q23 + q01);
  }
}

ORC_UNUSED static double clipd(double x, double lo, double hi) { /* mju_clip */
  return x < lo ? lo : (x > hi ? hi : x);
}

#endif
