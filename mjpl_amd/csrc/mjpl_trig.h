// mjpl_trig.h -- float64 sin/cos for the gfx950 kernels.
//
// mju_axisAngle2Quat (the only trig on the hot path; reference call site
// src/mjpl/constraint/collision_constraint.py:28 -> mj_kinematics) needs sin(a/2), cos(a/2)
// for joint angles |a/2| of a few radians.  OCML's sincos carries a Payne-Hanek branch we
// never take; this is the classic Cody-Waite (3-part pi/2) reduction followed by the
// degree-13/14 minimax kernels (fdlibm/msun k_sin, k_cos coefficients), < 1 ulp, written
// without FMA so that host and device agree bit-for-bit under -ffp-contract=off.
// |x| >= 2^19*pi/2 falls back to the platform sin/cos.
#pragma once

#if defined(__HIPCC__) || defined(__HIP__)
#define MJPL_HD __host__ __device__ __forceinline__
#else
#define MJPL_HD static inline
#endif

#include <math.h>

namespace mjpl {

MJPL_HD double k_sin(double x, double y) {
  const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
               S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
               S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
  double z = x * x;
  double w = z * z;
  double r = S2 + z * (S3 + z * S4) + z * w * (S5 + z * S6);
  double v = z * x;
  return x - ((z * (0.5 * y - v * r) - y) - v * S1);
}

MJPL_HD double k_cos(double x, double y) {
  const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
               C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
               C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
  double z = x * x;
  double w = z * z;
  double r = z * (C1 + z * (C2 + z * C3)) + (w * w) * (C4 + z * (C5 + z * C6));
  double hz = 0.5 * z;
  w = 1.0 - hz;
  return w + (((1.0 - w) - hz) + (z * r - x * y));
}

// sin(x) -> *s, cos(x) -> *c
MJPL_HD void sincos_pi2(double x, double *s, double *c) {
  const double invpio2 = 6.36619772367581382433e-01;
  const double pio2_1 = 1.57079632673412561417e+00;   // first 33 bits of pi/2
  const double pio2_2 = 6.07710050630396597660e-11;   // second 33 bits
  const double pio2_2t = 2.02226624879595063154e-21;  // pi/2 - (pio2_1 + pio2_2)
  double ax = fabs(x);
  if (!(ax < 8.2e5)) {  // also catches NaN/inf
    *s = sin(x);
    *c = cos(x);
    return;
  }
  // Branch-free two-stage Cody-Waite reduction: fn*pio2_1 and fn*pio2_2 are exact
  // (33-bit constants, |fn| < 2^20); for |x| <= pi/4 fn = 0 and (y0, y1) = (x, 0) exactly.
  double fn = rint(x * invpio2);
  int n = (int)fn;
  double t = x - fn * pio2_1;
  double w = fn * pio2_2;
  double r = t - w;
  w = fn * pio2_2t - ((t - r) - w);
  double y0 = r - w;
  double y1 = (r - y0) - w;
  double sn = k_sin(y0, y1);
  double cs = k_cos(y0, y1);
  switch (n & 3) {
    case 0: *s = sn; *c = cs; break;
    case 1: *s = cs; *c = -sn; break;
    case 2: *s = -sn; *c = -cs; break;
    default: *s = -cs; *c = sn; break;
  }
}

}  // namespace mjpl
