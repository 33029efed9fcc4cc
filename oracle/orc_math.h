/*
 * orc_math.h -- CPU ORACLE (test infrastructure, NOT the product): the small vector /
 * quaternion helpers shared by mjpl_oracle.c and mjpl_oracle_pose.c.
 */
#ifndef ORC_MATH_H
#define ORC_MATH_H

#include <math.h>

#define ORC_MINVAL 1e-15 /* mjMINVAL [MJ-recalled: mjmodel.h] */
#define ORC_UNUSED __attribute__((unused))

/* Operation count of the float64 arithmetic the oracle executes ("exact static count emitted by
 * cpu_ref").  Built only into libmjpl_oracle_count.so (-DORC_COUNT_FLOPS, oracle/Makefile; tools/count_flops.py):
 * every routine adds the additions / subtractions, multiplications, divisions and square roots of the statements
 * it has just executed -- read off the expressions, in the branch that was taken -- to one global tally; sin/cos
 * pairs are tallied as calls (the fdlibm restatement below executes 27 additions and 28 multiplications per pair).
 * Comparisons, negations, fabs, fmin / fmax and copies are not counted.  Single-threaded use only. */
#ifdef ORC_COUNT_FLOPS
typedef struct { long long add, mul, div, sqrt, sincos; } orc_flops_t;
extern orc_flops_t orc_flops;
#define ORC_FL(a, m, d, s) ((void)(orc_flops.add += (a), orc_flops.mul += (m), orc_flops.div += (d), orc_flops.sqrt += (s)))
#define ORC_FL_SINCOS() ((void)(orc_flops.sincos += 1))
#else
#define ORC_FL(a, m, d, s) ((void)0)
#define ORC_FL_SINCOS() ((void)0)
#endif

/* ------------------------------------------------------------------ small vector helpers
 * [MJ-recalled: engine_util_blas.c / engine_util_spatial.c]; operation order is the contract. */

ORC_UNUSED static double dot3(const double *a, const double *b) {
  ORC_FL(2, 3, 0, 0);
  return a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
}

ORC_UNUSED static void mul_mat_vec3(double *res, const double *mat, const double *vec) {
  ORC_FL(6, 9, 0, 0);
  res[0] = mat[0] * vec[0] + mat[1] * vec[1] + mat[2] * vec[2];
  res[1] = mat[3] * vec[0] + mat[4] * vec[1] + mat[5] * vec[2];
  res[2] = mat[6] * vec[0] + mat[7] * vec[1] + mat[8] * vec[2];
}

ORC_UNUSED static void mul_matT_vec3(double *res, const double *mat, const double *vec) {
  ORC_FL(6, 9, 0, 0);
  res[0] = mat[0] * vec[0] + mat[3] * vec[1] + mat[6] * vec[2];
  res[1] = mat[1] * vec[0] + mat[4] * vec[1] + mat[7] * vec[2];
  res[2] = mat[2] * vec[0] + mat[5] * vec[1] + mat[8] * vec[2];
}

/* mju_mulQuat */
ORC_UNUSED static void mul_quat(double *res, const double *a, const double *b) {
  ORC_FL(12, 16, 0, 0);
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
    ORC_FL(12, 18, 0, 0);
    double t0 = quat[0] * vec[0] + quat[2] * vec[2] - quat[3] * vec[1];
    double t1 = quat[0] * vec[1] + quat[3] * vec[0] - quat[1] * vec[2];
    double t2 = quat[0] * vec[2] + quat[1] * vec[1] - quat[2] * vec[0];
    double r0 = vec[0] + 2 * (quat[2] * t2 - quat[3] * t1);
    double r1 = vec[1] + 2 * (quat[3] * t0 - quat[1] * t2);
    double r2 = vec[2] + 2 * (quat[1] * t1 - quat[2] * t0);
    res[0] = r0; res[1] = r1; res[2] = r2;
  }
}

/* Which sin/cos mju_axisAngle2Quat uses.  0 (default): the C library's, as MuJoCo does -- libm
 * implementations differ from one another in the last bit, and so does the GPU's, so a verdict
 * whose signed distance lies within ~1e-15 of zero may differ between ANY two of them.
 * 1: the fdlibm algorithm restated below (Sun Microsystems' published k_sin.c / k_cos.c /
 * e_rem_pio2.c, medium-argument path), evaluated in plain IEEE double operations in a fixed order
 * without FMA, hence bit-reproducible on any IEEE machine.  The at-threshold parity tests run in
 * mode 1 so that "bit-exact" can be asserted down to the last ulp of the joint value; every other
 * test runs in mode 0.  (orc_set_trig in mjpl_oracle.c.) */
extern int orc_trig_mode;

ORC_UNUSED static double orc_ksin(double x, double y) {
  const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
               S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
               S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
  double z = x * x, w = z * z;
  double r = S2 + z * (S3 + z * S4) + z * w * (S5 + z * S6);
  double v = z * x;
  return x - ((z * (0.5 * y - v * r) - y) - v * S1);
}

ORC_UNUSED static double orc_kcos(double x, double y) {
  const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
               C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
               C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
  double z = x * x, w = z * z;
  double r = z * (C1 + z * (C2 + z * C3)) + (w * w) * (C4 + z * (C5 + z * C6));
  double hz = 0.5 * z;
  w = 1.0 - hz;
  return w + (((1.0 - w) - hz) + (z * r - x * y));
}

ORC_UNUSED static void orc_sincos(double x, double *s, double *c) {
  ORC_FL_SINCOS();
  if (orc_trig_mode == 0 || !(fabs(x) < 8.2e5)) { *s = sin(x); *c = cos(x); return; }
  /* x = n * pi/2 + (y0 + y1), pi/2 in three 33-bit pieces (Cody-Waite) */
  const double invpio2 = 6.36619772367581382433e-01, pio2_1 = 1.57079632673412561417e+00,
               pio2_2 = 6.07710050630396597660e-11, pio2_2t = 2.02226624879595063154e-21;
  double fn = rint(x * invpio2);
  int n = (int)fn;
  double t = x - fn * pio2_1;
  double w = fn * pio2_2;
  double r = t - w;
  w = fn * pio2_2t - ((t - r) - w);
  double y0 = r - w;
  double y1 = (r - y0) - w;
  double sn = orc_ksin(y0, y1), cs = orc_kcos(y0, y1);
  switch (n & 3) {
    case 0: *s = sn; *c = cs; break;
    case 1: *s = cs; *c = -sn; break;
    case 2: *s = -sn; *c = -cs; break;
    default: *s = -cs; *c = sn; break;
  }
}

/* mju_axisAngle2Quat */
ORC_UNUSED static void axis_angle2quat(double *res, const double *axis, double angle) {
  if (angle == 0) {
    res[0] = 1; res[1] = 0; res[2] = 0; res[3] = 0;
  } else {
    double s, c;
    orc_sincos(angle * 0.5, &s, &c);
    ORC_FL(0, 4, 0, 0);
    res[0] = c;
    res[1] = axis[0] * s;
    res[2] = axis[1] * s;
    res[3] = axis[2] * s;
  }
}

/* mju_normalize4 */
ORC_UNUSED static void normalize4(double *v) {
  double norm = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3]);
  ORC_FL(3, 4, 0, 1);
  if (norm < ORC_MINVAL) {
    v[0] = 1; v[1] = 0; v[2] = 0; v[3] = 0;
  } else if ((ORC_FL(1, 0, 0, 0), fabs(norm - 1) > ORC_MINVAL)) {
    ORC_FL(0, 4, 1, 0);
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
    ORC_FL(15, 16, 0, 0);
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
