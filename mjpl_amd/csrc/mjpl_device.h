// mjpl_device.h -- float64 kinematics + primitive narrowphase for gfx950, and the per-lane
// interpreter that walks the body tree for one configuration.
//
// What it replaces: mujoco.mj_kinematics + mujoco.mj_collision as called by
// CollisionConstraint.valid_config (reference src/mjpl/constraint/collision_constraint.py:26-30)
// and the allowed-body-pair filter CollisionRuleset.obeys_ruleset (:66-95), for a whole
// wavefront of configurations at a time.
//
// Numerics contract: every expression below is evaluated in IEEE-754 binary64 with one
// rounding per operation (the translation unit is built with -ffp-contract=off), in the
// operation order of MuJoCo's scalar C routines, so a verdict can differ from the CPU path
// only through sin/cos (mjpl_trig.h, <= 1 ulp from libm).  The null-quaternion / zero-vector
// shortcuts of mju_rotVecQuat / mju_quat2Mat / mju_axisAngle2Quat are value-identical to the
// general formulas (they differ at most in the sign of an exact zero), so the kernels run
// the general formulas branch-free.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mjpl_trig.h"

namespace mjpl {

#define MJPL_MINVAL 1e-15  // mjMINVAL

// Where the compiled model tables live while a kernel runs.
//   default            : global memory read through the constant address space, i.e. wave-uniform
//                        s_load_* into SGPRs via the scalar data cache.  Uniform constants then
//                        never occupy VGPRs and feed the FP64 VALU as scalar operands.
//   -DMJPL_TABLES_LDS=1: staged into LDS by every workgroup (ds_read broadcast into VGPRs);
//                        kept as an A/B build (profiles/ holds the comparison).
#ifndef MJPL_TABLES_LDS
#define MJPL_TABLES_LDS 0
#endif
#if MJPL_TABLES_LDS
typedef const int *IP;
typedef const double *DP;
#else
typedef const __attribute__((address_space(4))) int *IP;
typedef const __attribute__((address_space(4))) double *DP;
#endif

// ----------------------------------------------------------------------------- program layout
// The model is compiled on the host (mjpl_hip.hip: compile_program) into two flat tables that
// every workgroup stages into LDS: `ip` (int32 control words) and `dp` (float64 constants).

enum : int {
  H_NBODYOPS = 0,  // number of moving-body ops
  H_NPLAN,         // planning columns
  H_NSAVE,         // LDS pose-save slots
  H_NSLOTS,        // register slots in use
  H_OFF_BODYOPS,   // int offset of the first body op
  H_OFF_PERM,      // int offset of the ascending-qpos-address column permutation
  H_OFF_WORLD,     // double offset of the static ("world") geom table, W_LEN doubles per row
  H_NWORLD,        // rows in the world table
  H_SIZE
};

// body op: 6 header ints, then joints, then geoms
enum : int { B_PARENT = 0, B_DOFF, B_BODYID, B_NJNT, B_SAVE, B_NGEOM, B_SIZE };
enum : int { PARENT_CUR = 0, PARENT_STATIC = -1 };  // k > 0: restore LDS save slot k-1
// body dp: pos[3] quat[4]; if PARENT_STATIC: + ppos[3] pquat[4] pmat[9]

enum : int { J_TYPE = 0, J_QSRC, J_FLAGS, J_DOFF, J_SIZE };  // qsrc >= 0: planning column
enum : int { JF_POS_NONZERO = 1 };
// joint dp: axis[3] pos[3] qpos0 qconst

enum : int { G_TYPE = 0, G_FLAGS, G_DOFF, G_STORE, G_GEOMID, G_NSTORED, G_WMASK_LO, G_WMASK_HI,
             G_PMASK_LO, G_PMASK_HI, G_SIZE };
enum : int { GF_SAMEPOS = 1, GF_SAMEROT = 2 };
// geom dp: lpos[3] lquat[4] size[3] pad[2] | wbound[nworld] | wmargin[nworld] | stored[nstored][5]
//   wbound[w]  cull bound against static geom w: (r1+r2+margin)^2, or margin + rbound for a plane
//   wmargin[w] pair margin max(margin_cur, margin_w)
//   stored[k]  = bound, margin, psize[3] of the k-th earlier moving partner
enum : int { GD_LPOS = 0, GD_LQUAT = 3, GD_SIZE = 7, GD_WBOUND = 12 };
enum : int { SD_BOUND = 0, SD_MARGIN, SD_SIZE, SD_LEN = 5 };

// static partners: G_WMASK (non-plane) and G_PMASK (plane) are bit masks over the rows of the
// world table (<= 64 static geoms).
// stored partners: one packed int each, right after the geom record:
//   bits 0..5 first slot ; bits 6..11 second slot (boxes, else SLOT_NONE) ; bits 12..15 type
//   bit 17    partner is the FIRST geom of the pair in mj_collision's (g1,g2) order
enum : int { P_FIRST = 1 << 17 };
// world table row (fixed stride so that the next row can be requested before it is needed):
//   [0..2] pos  [3..5] z axis  [6] info word (type | geom id << 8)  [7] pad   <- cull part, 64 B
//   [8..10] x axis  [11..13] y axis  [14..16] size  [17..19] pad              <- narrowphase part
enum : int { W_POS = 0, W_ZAXIS = 3, W_INFO = 6, W_XAXIS = 8, W_YAXIS = 11, W_SIZE = 14, W_LEN = 20 };

enum : int { GT_PLANE = 0, GT_SPHERE = 2, GT_CAPSULE = 3, GT_BOX = 6 };
enum : int { JT_SLIDE = 2, JT_HINGE = 3 };

// ----------------------------------------------------------------------------- small math
// [MJ-recalled: engine_util_blas.c, engine_util_spatial.c]

MJPL_HD double dot3(const double *a, const double *b) {
  return a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
}

MJPL_HD void mul_mat_vec3(double *res, const double *mat, const double *vec) {
  res[0] = mat[0] * vec[0] + mat[1] * vec[1] + mat[2] * vec[2];
  res[1] = mat[3] * vec[0] + mat[4] * vec[1] + mat[5] * vec[2];
  res[2] = mat[6] * vec[0] + mat[7] * vec[1] + mat[8] * vec[2];
}

MJPL_HD void mul_matT_vec3(double *res, const double *mat, const double *vec) {
  res[0] = mat[0] * vec[0] + mat[3] * vec[1] + mat[6] * vec[2];
  res[1] = mat[1] * vec[0] + mat[4] * vec[1] + mat[7] * vec[2];
  res[2] = mat[2] * vec[0] + mat[5] * vec[1] + mat[8] * vec[2];
}

MJPL_HD void mul_quat(double *res, const double *a, const double *b) {
  double t0 = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
  double t1 = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
  double t2 = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
  double t3 = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
  res[0] = t0; res[1] = t1; res[2] = t2; res[3] = t3;
}

MJPL_HD void rot_vec_quat(double *res, const double *vec, const double *quat) {
  double t0 = quat[0] * vec[0] + quat[2] * vec[2] - quat[3] * vec[1];
  double t1 = quat[0] * vec[1] + quat[3] * vec[0] - quat[1] * vec[2];
  double t2 = quat[0] * vec[2] + quat[1] * vec[1] - quat[2] * vec[0];
  double r0 = vec[0] + 2 * (quat[2] * t2 - quat[3] * t1);
  double r1 = vec[1] + 2 * (quat[3] * t0 - quat[1] * t2);
  double r2 = vec[2] + 2 * (quat[1] * t1 - quat[2] * t0);
  res[0] = r0; res[1] = r1; res[2] = r2;
}

MJPL_HD void normalize4(double *v) {
  double norm = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3]);
  bool tiny = norm < MJPL_MINVAL;
  bool scale = fabs(norm - 1) > MJPL_MINVAL;
  double inv = 1 / norm;
  double s0 = v[0] * inv, s1 = v[1] * inv, s2 = v[2] * inv, s3 = v[3] * inv;
  v[0] = tiny ? 1.0 : (scale ? s0 : v[0]);
  v[1] = tiny ? 0.0 : (scale ? s1 : v[1]);
  v[2] = tiny ? 0.0 : (scale ? s2 : v[2]);
  v[3] = tiny ? 0.0 : (scale ? s3 : v[3]);
}

MJPL_HD void quat2mat(double *res, const double *q) {
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

// third column of quat2mat only (capsule axis); same expressions as res[2], res[5], res[8]
MJPL_HD void quat2zaxis(double *m, const double *q) {
  const double q00 = q[0] * q[0], q01 = q[0] * q[1], q02 = q[0] * q[2];
  const double q11 = q[1] * q[1], q13 = q[1] * q[3];
  const double q22 = q[2] * q[2], q23 = q[2] * q[3], q33 = q[3] * q[3];
  m[2] = 2 * (q13 + q02);
  m[5] = 2 * (q23 - q01);
  m[8] = q00 - q11 - q22 + q33;
}

MJPL_HD double clipd(double x, double lo, double hi) { return x < lo ? lo : (x > hi ? hi : x); }

// ----------------------------------------------------------------------------- narrowphase
// Verdict-only ("ncon > 0") primitives.  [MJ-recalled: engine_collision_primitive.c,
// engine_collision_box.c]; capsule-box / box-box follow the oracle's documented deviations.
// A geom is (pos[3], m[9]) with m row-major; capsules use only the z column m[2],m[5],m[8].

struct Geom {
  double pos[3];
  double m[9];
};

MJPL_HD bool sphere_sphere(double margin, const double *pos1, double r1, const double *pos2,
                           double r2) {
  double dif[3] = {pos1[0] - pos2[0], pos1[1] - pos2[1], pos1[2] - pos2[2]};
  double cdist_sqr = dot3(dif, dif);
  double min_dist = margin + r1 + r2;
  return !(cdist_sqr > min_dist * min_dist);
}

MJPL_HD bool plane_sphere(double margin, const Geom &pl, const double *pos2, double r2) {
  double n[3] = {pl.m[2], pl.m[5], pl.m[8]};
  double tmp[3] = {pos2[0] - pl.pos[0], pos2[1] - pl.pos[1], pos2[2] - pl.pos[2]};
  double cdist = dot3(tmp, n);
  return !(cdist > margin + r2);
}

MJPL_HD bool plane_capsule(double margin, const Geom &pl, const Geom &cap, const double *size2) {
  double seg[3] = {size2[1] * cap.m[2], size2[1] * cap.m[5], size2[1] * cap.m[8]};
  double e1[3] = {cap.pos[0] + seg[0], cap.pos[1] + seg[1], cap.pos[2] + seg[2]};
  double e2[3] = {cap.pos[0] - seg[0], cap.pos[1] - seg[1], cap.pos[2] - seg[2]};
  bool n1 = plane_sphere(margin, pl, e1, size2[0]);
  bool n2 = plane_sphere(margin, pl, e2, size2[0]);
  return n1 || n2;
}

MJPL_HD bool plane_box(double margin, const Geom &pl, const Geom &box, const double *size2) {
  double norm[3] = {pl.m[2], pl.m[5], pl.m[8]};
  double dif[3] = {box.pos[0] - pl.pos[0], box.pos[1] - pl.pos[1], box.pos[2] - pl.pos[2]};
  double dist = dot3(dif, norm);
  bool any = false;
  // rolled on purpose: unrolling the 8 corners keeps ~40 extra VGPRs live
#pragma unroll 1
  for (int i = 0; i < 8; i++) {
    double vec[3], corner[3];
    vec[0] = (i & 1) ? size2[0] : -size2[0];
    vec[1] = (i & 2) ? size2[1] : -size2[1];
    vec[2] = (i & 4) ? size2[2] : -size2[2];
    mul_mat_vec3(corner, box.m, vec);
    double ldist = dot3(norm, corner);
    any = any || !(dist + ldist > margin || ldist > 0);
  }
  return any;
}

MJPL_HD bool sphere_capsule(double margin, const double *pos1, double r1, const Geom &cap,
                            const double *size2) {
  double len = size2[1];
  double axis[3] = {cap.m[2], cap.m[5], cap.m[8]};
  double vec[3] = {pos1[0] - cap.pos[0], pos1[1] - cap.pos[1], pos1[2] - cap.pos[2]};
  double x = clipd(dot3(axis, vec), -len, len);
  vec[0] = axis[0] * x + cap.pos[0];
  vec[1] = axis[1] * x + cap.pos[1];
  vec[2] = axis[2] * x + cap.pos[2];
  return sphere_sphere(margin, pos1, r1, vec, size2[0]);
}

__device__ __forceinline__ bool capsule_capsule(double margin, const Geom &c1, const double *size1,
                                                const Geom &c2, const double *size2) {
  double axis1[3] = {c1.m[2] * size1[1], c1.m[5] * size1[1], c1.m[8] * size1[1]};
  double axis2[3] = {c2.m[2] * size2[1], c2.m[5] * size2[1], c2.m[8] * size2[1]};
  double dif[3] = {c1.pos[0] - c2.pos[0], c1.pos[1] - c2.pos[1], c1.pos[2] - c2.pos[2]};
  double ma = dot3(axis1, axis1);
  double mb = -dot3(axis1, axis2);
  double mc = dot3(axis2, axis2);
  double u = -dot3(axis1, dif);
  double v = dot3(axis2, dif);
  double det = ma * mc - mb * mb;
  double vec1[3], vec2[3];
  bool general = fabs(det) >= MJPL_MINVAL;
  bool res = false;

  if (general) {
    // same divisions as the scalar routine, sign-selected numerators instead of branches
    double x1 = (mc * u - mb * v) / det;
    double x2 = (ma * v - mb * u) / det;
    bool hi1 = x1 > 1, lo1 = x1 < -1;
    double x2c = (hi1 ? (v - mb) : (v + mb)) / mc;
    x1 = hi1 ? 1.0 : (lo1 ? -1.0 : x1);
    x2 = (hi1 || lo1) ? x2c : x2;
    bool hi2 = x2 > 1, lo2 = x2 < -1;
    double x1c = clipd((hi2 ? (u - mb) : (u + mb)) / ma, -1, 1);
    x2 = hi2 ? 1.0 : (lo2 ? -1.0 : x2);
    x1 = (hi2 || lo2) ? x1c : x1;
#pragma unroll
    for (int k = 0; k < 3; k++) {
      vec1[k] = c1.pos[k] + axis1[k] * x1;
      vec2[k] = c2.pos[k] + axis2[k] * x2;
    }
    res = sphere_sphere(margin, vec1, size1[0], vec2, size2[0]);
  } else {
    // parallel axes (rare): any of the four end tests
    double x1, x2;
    for (int k = 0; k < 3; k++) vec1[k] = c1.pos[k] + axis1[k];
    x2 = clipd((v - mb) / mc, -1, 1);
    for (int k = 0; k < 3; k++) vec2[k] = c2.pos[k] + axis2[k] * x2;
    res = sphere_sphere(margin, vec1, size1[0], vec2, size2[0]);
    for (int k = 0; k < 3; k++) vec1[k] = c1.pos[k] - axis1[k];
    x2 = clipd((v + mb) / mc, -1, 1);
    for (int k = 0; k < 3; k++) vec2[k] = c2.pos[k] + axis2[k] * x2;
    res = res || sphere_sphere(margin, vec1, size1[0], vec2, size2[0]);
    for (int k = 0; k < 3; k++) vec2[k] = c2.pos[k] + axis2[k];
    x1 = clipd((u - mb) / ma, -1, 1);
    for (int k = 0; k < 3; k++) vec1[k] = c1.pos[k] + axis1[k] * x1;
    res = res || sphere_sphere(margin, vec1, size1[0], vec2, size2[0]);
    for (int k = 0; k < 3; k++) vec2[k] = c2.pos[k] - axis2[k];
    x1 = clipd((u + mb) / ma, -1, 1);
    for (int k = 0; k < 3; k++) vec1[k] = c1.pos[k] + axis1[k] * x1;
    res = res || sphere_sphere(margin, vec1, size1[0], vec2, size2[0]);
  }
  return res;
}

MJPL_HD bool sphere_box_local(double margin, const double *c, double r, const double *size2) {
  double d[3];
#pragma unroll
  for (int k = 0; k < 3; k++) d[k] = clipd(c[k], -size2[k], size2[k]) - c[k];
  double dist = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
  return !(dist - r > margin);
}

MJPL_HD bool sphere_box(double margin, const double *pos1, double r1, const Geom &box,
                        const double *size2) {
  double tmp[3] = {pos1[0] - box.pos[0], pos1[1] - box.pos[1], pos1[2] - box.pos[2]};
  double center[3];
  mul_matT_vec3(center, box.m, tmp);
  return sphere_box_local(margin, center, r1, size2);
}

MJPL_HD double capbox_g(const double *p, const double *h, const double *s, double t) {
  double g = 0;
#pragma unroll
  for (int k = 0; k < 3; k++) {
    double x = p[k] + t * h[k];
    double e = x - fmin(fmax(x, -s[k]), s[k]);
    g = g + h[k] * e;
  }
  return g;
}

// exact 1-D convex minimisation of dist^2(segment point, box); see oracle/mjpl_oracle.c
__device__ __forceinline__ bool capsule_box(double margin, const Geom &cap, const double *size1,
                                            const Geom &box, const double *size2) {
  double tmp[3] = {cap.pos[0] - box.pos[0], cap.pos[1] - box.pos[1], cap.pos[2] - box.pos[2]};
  double axis[3] = {cap.m[2], cap.m[5], cap.m[8]};
  double p[3], a[3], h[3], inv[3];
  mul_matT_vec3(p, box.m, tmp);
  mul_matT_vec3(a, box.m, axis);
#pragma unroll
  for (int k = 0; k < 3; k++) {
    h[k] = a[k] * size1[1];
    inv[k] = 1 / h[k];  // h == 0: +-inf, the breakpoint becomes +-inf or NaN and is skipped
  }

  double lo = -1, hi = 1;
  const double g_m1 = capbox_g(p, h, size2, lo);
  const double g_p1 = capbox_g(p, h, size2, hi);
  double glo = g_m1, ghi = g_p1;
  // six face breakpoints in the scalar routine's order (k = 0,1,2; minus before plus)
#pragma unroll
  for (int k = 0; k < 3; k++) {
#pragma unroll
    for (int sgn = -1; sgn <= 1; sgn += 2) {
      double tb = (sgn * size2[k] - p[k]) * inv[k];
      bool inside = (tb > lo && tb < hi);
      double gb = capbox_g(p, h, size2, tb);
      bool below = inside && gb <= 0;
      bool above = inside && !(gb <= 0);
      lo = below ? tb : lo;
      glo = below ? gb : glo;
      hi = above ? tb : hi;
      ghi = above ? gb : ghi;
    }
  }
  double den = ghi - glo;
  double t = (den > 0) ? lo + (hi - lo) * ((0 - glo) / den) : lo;
  t = (g_m1 >= 0) ? -1.0 : ((g_p1 <= 0) ? 1.0 : t);
  double c[3] = {p[0] + t * h[0], p[1] + t * h[1], p[2] + t * h[2]};
  return sphere_box_local(margin, c, size1[0], size2);
}

// 15-axis separating-axis verdict; see oracle/mjpl_oracle.c box_box
__device__ __forceinline__ bool box_box(double margin, const Geom &b1, const double *size1,
                                        const Geom &b2, const double *size2) {
  double d[3] = {b2.pos[0] - b1.pos[0], b2.pos[1] - b1.pos[1], b2.pos[2] - b1.pos[2]};
  double R[9], A[9], t[3];
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++)
      R[3 * i + j] = b1.m[i] * b2.m[j] + b1.m[3 + i] * b2.m[3 + j] + b1.m[6 + i] * b2.m[6 + j];
  mul_matT_vec3(t, b1.m, d);
#pragma unroll
  for (int k = 0; k < 9; k++) A[k] = fabs(R[k]);
  bool sep = false;
#pragma unroll
  for (int i = 0; i < 3; i++) {
    double rb = size2[0] * A[3 * i] + size2[1] * A[3 * i + 1] + size2[2] * A[3 * i + 2];
    sep = sep || (fabs(t[i]) - (size1[i] + rb) > margin);
  }
#pragma unroll
  for (int j = 0; j < 3; j++) {
    double ra = size1[0] * A[j] + size1[1] * A[3 + j] + size1[2] * A[6 + j];
    double tj = t[0] * R[j] + t[1] * R[3 + j] + t[2] * R[6 + j];
    sep = sep || (fabs(tj) - (ra + size2[j]) > margin);
  }
#pragma unroll
  for (int i = 0; i < 3; i++) {
    const int i1 = (i + 1) % 3, i2 = (i + 2) % 3;
#pragma unroll
    for (int j = 0; j < 3; j++) {
      const int j1 = (j + 1) % 3, j2 = (j + 2) % 3;
      double len2 = 1 - R[3 * i + j] * R[3 * i + j];
      double ra = size1[i1] * A[3 * i2 + j] + size1[i2] * A[3 * i1 + j];
      double rb = size2[j1] * A[3 * i + j2] + size2[j2] * A[3 * i + j1];
      double tl = t[i2] * R[3 * i1 + j] - t[i1] * R[3 * i2 + j];
      bool s = !(len2 < 1e-12) && (fabs(tl) - (ra + rb) > margin * sqrt(len2));
      sep = sep || s;
    }
  }
  return !sep;
}

// Pair dispatch.  mj_collision calls its function table with (g1, g2) ordered by geom type,
// and by geom id when the types are equal; `pfirst` says that the PARTNER is g1.  Mixed-type
// pairs are ordered by type inside each routine's signature, so only the symmetric-type
// routines whose rounding depends on the argument order (sphere-sphere's radius sum,
// capsule-capsule, box-box) look at `pfirst`, with wave-uniform selects.
// WBOX: some static geom is a box.  MBOX: some moving geom is a box.  The flags only remove
// dead narrowphase code (and its registers) from an instantiation.
template <bool WBOX, bool MBOX>
__device__ __forceinline__ bool pair_contact(int tcur, const Geom &cur, const double *scur, int tpar,
                                             const Geom &par, const double *spar, bool pfirst,
                                             double margin) {
  constexpr bool PBOX = WBOX || MBOX;  // the partner may be a box
  bool r = false;
  if (tpar == GT_PLANE) {
    if (tcur == GT_SPHERE) r = plane_sphere(margin, par, cur.pos, scur[0]);
    else if (tcur == GT_CAPSULE) r = plane_capsule(margin, par, cur, scur);
    else if (MBOX) r = plane_box(margin, par, cur, scur);
  } else if (tcur == GT_SPHERE && tpar == GT_SPHERE) {
    const double r1 = pfirst ? spar[0] : scur[0], r2 = pfirst ? scur[0] : spar[0];
    r = sphere_sphere(margin, cur.pos, r1, par.pos, r2);  // (a-b)^2 == (b-a)^2 exactly
  } else if (tcur == GT_SPHERE && tpar == GT_CAPSULE) {
    r = sphere_capsule(margin, cur.pos, scur[0], par, spar);
  } else if (tcur == GT_CAPSULE && tpar == GT_SPHERE) {
    r = sphere_capsule(margin, par.pos, spar[0], cur, scur);
  } else if (PBOX && tcur == GT_SPHERE && tpar == GT_BOX) {
    r = sphere_box(margin, cur.pos, scur[0], par, spar);
  } else if (MBOX && tcur == GT_BOX && tpar == GT_SPHERE) {
    r = sphere_box(margin, par.pos, spar[0], cur, scur);
  } else if (PBOX && tcur == GT_CAPSULE && tpar == GT_BOX) {
    r = capsule_box(margin, cur, scur, par, spar);
  } else if (MBOX && tcur == GT_BOX && tpar == GT_CAPSULE) {
    r = capsule_box(margin, par, spar, cur, scur);
  } else if (tcur == GT_CAPSULE && tpar == GT_CAPSULE) {
    Geom c1, c2;
    double s1[2], s2[2];
#pragma unroll
    for (int k = 0; k < 3; k++) {
      c1.pos[k] = pfirst ? par.pos[k] : cur.pos[k];
      c2.pos[k] = pfirst ? cur.pos[k] : par.pos[k];
      c1.m[2 + 3 * k] = pfirst ? par.m[2 + 3 * k] : cur.m[2 + 3 * k];
      c2.m[2 + 3 * k] = pfirst ? cur.m[2 + 3 * k] : par.m[2 + 3 * k];
    }
#pragma unroll
    for (int k = 0; k < 2; k++) {
      s1[k] = pfirst ? spar[k] : scur[k];
      s2[k] = pfirst ? scur[k] : spar[k];
    }
    r = capsule_capsule(margin, c1, s1, c2, s2);
  } else if (MBOX) {
    Geom b1, b2;
    double s1[3], s2[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
      b1.pos[k] = pfirst ? par.pos[k] : cur.pos[k];
      b2.pos[k] = pfirst ? cur.pos[k] : par.pos[k];
      s1[k] = pfirst ? spar[k] : scur[k];
      s2[k] = pfirst ? scur[k] : spar[k];
    }
#pragma unroll
    for (int k = 0; k < 9; k++) {
      b1.m[k] = pfirst ? par.m[k] : cur.m[k];
      b2.m[k] = pfirst ? cur.m[k] : par.m[k];
    }
    r = box_box(margin, b1, s1, b2, s2);
  }
  return r;
}

// ----------------------------------------------------------------------------- interpreter

struct FkOut {  // global-memory destinations of the FK parity kernel (any may be null)
  double *xpos, *xquat, *geom_xpos, *geom_xmat;
  int nbody, ngeom;
};

template <int MAXS>
struct SlotFile {
  double v[MAXS > 0 ? MAXS : 1][6];  // pos[3], zaxis[3] of earlier moving sphere/capsule geoms
};

// Slot access is expanded in place by macros, with a literal index per slot: the slot file is
// a local of run_config whose every access has a constant index from the start, so the first
// SROA pass promotes it to VGPRs.  (Behind a helper function taking it by reference, or
// indexed by a loop variable, SimplifyCFG merges the per-slot loads into one load with a
// selected address and the file ends up in scratch memory.)
#define MJPL_FOR_SLOTS(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)
#define MJPL_SLOT_PUT(n)                                                          \
  case n:                                                                         \
    if constexpr (MAXS > n) {                                                     \
      sf.v[n][0] = t6[0]; sf.v[n][1] = t6[1]; sf.v[n][2] = t6[2];                 \
      sf.v[n][3] = t6[3]; sf.v[n][4] = t6[4]; sf.v[n][5] = t6[5];                 \
    }                                                                             \
    break;
#define MJPL_SLOT_GET(n)                                                          \
  case n:                                                                         \
    if constexpr (MAXS > n) {                                                     \
      t6[0] = sf.v[n][0]; t6[1] = sf.v[n][1]; t6[2] = sf.v[n][2];                 \
      t6[3] = sf.v[n][3]; t6[4] = sf.v[n][4]; t6[5] = sf.v[n][5];                 \
    }                                                                             \
    break;
// A sphere/capsule occupies one slot (pos, z axis); a box a second one (x and y axes).
// Slot ids are packed as  first | (second << 6), second == SLOT_NONE when unused.
enum : int { SLOT_NONE = 63 };

// Optimisation barrier: makes the compiler treat a per-lane value as redefined here, so that
// expressions of it are not hoisted out of the partner loops (loop-invariant code motion of the
// narrowphase prologues costs tens of VGPRs that stay live across the whole loop).
__device__ __forceinline__ void pin(double &x) { asm volatile("" : "+v"(x)); }
__device__ __forceinline__ void pin_geom(Geom &g) {
  pin(g.pos[0]); pin(g.pos[1]); pin(g.pos[2]); pin(g.m[2]); pin(g.m[5]); pin(g.m[8]);
}

// control words are wave-uniform: pin them to SGPRs so the interpreter's branches are scalar
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

// Walk the moving part of the body tree for this lane's configuration.
//   ip, dp : program tables in LDS            q : this lane's planning columns, q[c*qstride]
//   save   : this lane's LDS pose-save area, save[(slot*7+k)*sstride]
// Returns true iff the configuration has a contact outside the allowed body pairs.
// `active` = false lanes run along (wave-uniform control flow) but never report a hit.
// EMIT: also write body/geom world poses to `out` row `row` (FK parity kernel).
template <int MAXS, bool EMIT, bool WBOX, bool MBOX>
__device__ __forceinline__ bool run_config(IP ip, DP dp, const double *q, int qstride, double *save,
                                           int sstride, bool active, const FkOut &out, int64_t row) {
  SlotFile<MAXS> sf;
  double p[3] = {0, 0, 0}, qt[4] = {1, 0, 0, 0}, R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  bool hit = false;
  const int nbodyops = uni(ip[H_NBODYOPS]);
  DP world = dp + uni(ip[H_OFF_WORLD]);
  int pc = uni(ip[H_OFF_BODYOPS]);

  for (int b = 0; b < nbodyops; b++) {
    // a wave whose every lane is already decided skips the rest of the tree
    if (!EMIT && __ballot(active && !hit) == 0ull) break;

    const int parent = uni(ip[pc + B_PARENT]);
    DP bd = dp + uni(ip[pc + B_DOFF]);
    const int njnt = uni(ip[pc + B_NJNT]);
    const int save_slot = uni(ip[pc + B_SAVE]);
    const int ngeom = uni(ip[pc + B_NGEOM]);
    const int body_id = uni(ip[pc + B_BODYID]);
    pc += B_SIZE;

    double pp[3], pq[4], pR[9];
    if (parent == PARENT_CUR) {
#pragma unroll
      for (int k = 0; k < 3; k++) pp[k] = p[k];
#pragma unroll
      for (int k = 0; k < 4; k++) pq[k] = qt[k];
#pragma unroll
      for (int k = 0; k < 9; k++) pR[k] = R[k];
    } else if (parent == PARENT_STATIC) {
#pragma unroll
      for (int k = 0; k < 3; k++) pp[k] = bd[7 + k];
#pragma unroll
      for (int k = 0; k < 4; k++) pq[k] = bd[10 + k];
#pragma unroll
      for (int k = 0; k < 9; k++) pR[k] = bd[14 + k];
    } else {
      const double *sv = save + (size_t)(parent - 1) * 7 * sstride;
#pragma unroll
      for (int k = 0; k < 3; k++) pp[k] = sv[k * sstride];
#pragma unroll
      for (int k = 0; k < 4; k++) pq[k] = sv[(3 + k) * sstride];
      quat2mat(pR, pq);  // bit-identical to the matrix built when the pose was saved
    }

    // fixed offset relative to the parent
    double np[3], nq[4];
    {
      double bpos[3] = {bd[0], bd[1], bd[2]};
      double bquat[4] = {bd[3], bd[4], bd[5], bd[6]};
      mul_mat_vec3(np, pR, bpos);
      np[0] += pp[0]; np[1] += pp[1]; np[2] += pp[2];
      mul_quat(nq, pq, bquat);
    }

    // joints
    for (int j = 0; j < njnt; j++) {
      const int jtype = uni(ip[pc + J_TYPE]);
      const int qsrc = uni(ip[pc + J_QSRC]);
      const int jflags = uni(ip[pc + J_FLAGS]);
      DP jd = dp + uni(ip[pc + J_DOFF]);
      pc += J_SIZE;
      const double qv = (qsrc >= 0) ? q[qsrc * qstride] : jd[7];
      const double dq = qv - jd[6];
      double jaxis[3] = {jd[0], jd[1], jd[2]};
      double jpos[3] = {jd[3], jd[4], jd[5]};
      if (jtype == JT_SLIDE) {
        double xaxis[3];
        rot_vec_quat(xaxis, jaxis, nq);
        np[0] += xaxis[0] * dq; np[1] += xaxis[1] * dq; np[2] += xaxis[2] * dq;
      } else {
        double xanchor[3] = {np[0], np[1], np[2]};
        if (jflags & JF_POS_NONZERO) {
          rot_vec_quat(xanchor, jpos, nq);
          xanchor[0] += np[0]; xanchor[1] += np[1]; xanchor[2] += np[2];
        }
        double s, c;
        sincos_pi2(dq * 0.5, &s, &c);
        double qloc[4] = {c, jaxis[0] * s, jaxis[1] * s, jaxis[2] * s};
        mul_quat(nq, nq, qloc);
        if (jflags & JF_POS_NONZERO) {
          double vec[3];
          rot_vec_quat(vec, jpos, nq);
          np[0] = xanchor[0] - vec[0]; np[1] = xanchor[1] - vec[1]; np[2] = xanchor[2] - vec[2];
        }
      }
    }

    normalize4(nq);
#pragma unroll
    for (int k = 0; k < 3; k++) p[k] = np[k];
#pragma unroll
    for (int k = 0; k < 4; k++) qt[k] = nq[k];
    quat2mat(R, qt);

    if (save_slot >= 0) {
      double *sv = save + (size_t)save_slot * 7 * sstride;
#pragma unroll
      for (int k = 0; k < 3; k++) sv[k * sstride] = p[k];
#pragma unroll
      for (int k = 0; k < 4; k++) sv[(3 + k) * sstride] = qt[k];
    }
    if (EMIT) {
      if (active && out.xpos)
        for (int k = 0; k < 3; k++) out.xpos[(row * out.nbody + body_id) * 3 + k] = p[k];
      if (active && out.xquat)
        for (int k = 0; k < 4; k++) out.xquat[(row * out.nbody + body_id) * 4 + k] = qt[k];
    }

    // geoms of this body: world pose, then every enabled pair against world geoms and
    // against earlier moving geoms held in register slots
    for (int g = 0; g < ngeom; g++) {
      const int gtype = uni(ip[pc + G_TYPE]);
      const int gflags = uni(ip[pc + G_FLAGS]);
      DP gd = dp + uni(ip[pc + G_DOFF]);
      const int store = uni(ip[pc + G_STORE]);
      const int geom_id = uni(ip[pc + G_GEOMID]);
      const int nstored = uni(ip[pc + G_NSTORED]);
      const unsigned long long wmask_all =
          (unsigned long long)(unsigned)uni(ip[pc + G_WMASK_LO]) |
          ((unsigned long long)(unsigned)uni(ip[pc + G_WMASK_HI]) << 32);
      const unsigned long long pmask_all =
          (unsigned long long)(unsigned)uni(ip[pc + G_PMASK_LO]) |
          ((unsigned long long)(unsigned)uni(ip[pc + G_PMASK_HI]) << 32);
      pc += G_SIZE;

      Geom cur;
      const double gsize[3] = {gd[GD_SIZE], gd[GD_SIZE + 1], gd[GD_SIZE + 2]};
      if (gflags & GF_SAMEPOS) {
        cur.pos[0] = p[0]; cur.pos[1] = p[1]; cur.pos[2] = p[2];
      } else {
        double lpos[3] = {gd[0], gd[1], gd[2]};
        mul_mat_vec3(cur.pos, R, lpos);
        cur.pos[0] += p[0]; cur.pos[1] += p[1]; cur.pos[2] += p[2];
      }
      if (gflags & GF_SAMEROT) {
        if (EMIT || MBOX) {
#pragma unroll
          for (int k = 0; k < 9; k++) cur.m[k] = R[k];
        } else {
          cur.m[2] = R[2]; cur.m[5] = R[5]; cur.m[8] = R[8];
        }
      } else {
        double lq[4] = {gd[3], gd[4], gd[5], gd[6]}, gq[4];
        mul_quat(gq, qt, lq);
        if (EMIT || (MBOX && gtype == GT_BOX)) quat2mat(cur.m, gq);
        else quat2zaxis(cur.m, gq);
      }
      if (EMIT) {
        if (active && out.geom_xpos)
          for (int k = 0; k < 3; k++) out.geom_xpos[(row * out.ngeom + geom_id) * 3 + k] = cur.pos[k];
        if (active && out.geom_xmat)
          for (int k = 0; k < 9; k++) out.geom_xmat[(row * out.ngeom + geom_id) * 9 + k] = cur.m[k];
      }

      if (!EMIT) {
        const int nworld = uni(ip[H_NWORLD]);
        DP wbound = gd + GD_WBOUND;

        // ---- static planes (few): signed-distance cull, then the plane routines
        for (unsigned long long pm = pmask_all; pm; pm &= pm - 1) {
          const int wc = (int)__builtin_ctzll(pm);
          DP r = world + wc * W_LEN;
          Geom par;
          par.pos[0] = r[W_POS]; par.pos[1] = r[W_POS + 1]; par.pos[2] = r[W_POS + 2];
          par.m[2] = r[W_ZAXIS]; par.m[5] = r[W_ZAXIS + 1]; par.m[8] = r[W_ZAXIS + 2];
          par.m[0] = par.m[1] = par.m[3] = par.m[4] = par.m[6] = par.m[7] = 0;
          double dif[3] = {cur.pos[0] - par.pos[0], cur.pos[1] - par.pos[1], cur.pos[2] - par.pos[2]};
          double n[3] = {par.m[2], par.m[5], par.m[8]};
          const bool pass = !(dot3(dif, n) > wbound[wc]) && active && !hit;
          if (__ballot(pass) == 0ull) continue;
          const double psize[3] = {0, 0, 0};
          const bool contact = pair_contact<WBOX, MBOX>(gtype, cur, gsize, GT_PLANE, par, psize, true,
                                                        wbound[nworld + wc]);
          hit = hit || (pass && contact);
        }

        // ---- other static partners: rows of the world table selected by the enable mask.
        // The cull part of the NEXT enabled row (64 B) and its bound are requested, without a
        // branch, before the current row is tested, so that the scalar-load round trip of a
        // pair overlaps the previous pair's arithmetic.
#ifdef MJPL_X_SKIP_WORLD
        unsigned long long wmask = 0;
#else
        unsigned long long wmask = wmask_all;
#endif
        int w = wmask ? (int)__builtin_ctzll(wmask) : 0;
        DP rn = world + w * W_LEN;
        double nx_pos[3] = {rn[W_POS], rn[W_POS + 1], rn[W_POS + 2]};
        double nx_z[3] = {rn[W_ZAXIS], rn[W_ZAXIS + 1], rn[W_ZAXIS + 2]};
        int nx_info = ((IP)(rn + W_INFO))[0];
        double nx_bound = wbound[w];
        while (wmask) {
          Geom par;
          par.pos[0] = nx_pos[0]; par.pos[1] = nx_pos[1]; par.pos[2] = nx_pos[2];
          par.m[2] = nx_z[0]; par.m[5] = nx_z[1]; par.m[8] = nx_z[2];
          const int info = nx_info;
          const double bound = nx_bound;
          const int wc = w;
          wmask &= wmask - 1;
          w = wmask ? (int)__builtin_ctzll(wmask) : 0;
          rn = world + w * W_LEN;
          nx_pos[0] = rn[W_POS]; nx_pos[1] = rn[W_POS + 1]; nx_pos[2] = rn[W_POS + 2];
          nx_z[0] = rn[W_ZAXIS]; nx_z[1] = rn[W_ZAXIS + 1]; nx_z[2] = rn[W_ZAXIS + 2];
          nx_info = ((IP)(rn + W_INFO))[0];
          nx_bound = wbound[w];

          const int ptype = info & 255;
          __builtin_assume(ptype != GT_PLANE);
          pin_geom(cur);
          // bounding cull (mj_collideSphere): squared centre distance; (a-b)^2 == (b-a)^2 exactly,
          // so the pair order does not matter
          double dif[3] = {cur.pos[0] - par.pos[0], cur.pos[1] - par.pos[1], cur.pos[2] - par.pos[2]};
          const bool pass = !(dot3(dif, dif) > bound) && active && !hit;
          if (__ballot(pass) == 0ull) continue;  // nobody in the wave needs the narrowphase
#ifdef MJPL_X_SKIP_NARROW
          hit = hit || (pass && dif[0] == 12345.0);
          continue;
#endif

          DP r = world + wc * W_LEN;
          const double psize[3] = {r[W_SIZE], r[W_SIZE + 1], r[W_SIZE + 2]};
          if (WBOX) {
            par.m[0] = r[W_XAXIS]; par.m[3] = r[W_XAXIS + 1]; par.m[6] = r[W_XAXIS + 2];
            par.m[1] = r[W_YAXIS]; par.m[4] = r[W_YAXIS + 1]; par.m[7] = r[W_YAXIS + 2];
          } else {
            par.m[0] = par.m[1] = par.m[3] = par.m[4] = par.m[6] = par.m[7] = 0;
          }
          // mj_collision order: smaller geom type first, geom id breaks ties
          const int pgid = info >> 8;
          const bool pfirst = (ptype < gtype) || (ptype == gtype && pgid < geom_id);
          const double margin = wbound[nworld + wc];
          const bool contact = pair_contact<WBOX, MBOX>(gtype, cur, gsize, ptype, par, psize, pfirst, margin);
          hit = hit || (pass && contact);
        }

        // ---- earlier moving partners, held in the register slot file
        DP sd = gd + GD_WBOUND + 2 * nworld;
        // same software pipeline as above: word and bound of the next entry are in flight
        // while the current one is tested (entry 0 of the next geom record is harmless to read)
        int nx_pw = ip[pc];
        double nx_sb = sd[SD_BOUND];
#ifdef MJPL_X_SKIP_STORED
        for (int e = 0; e < 0; e++, sd += SD_LEN) {
#else
        for (int e = 0; e < nstored; e++, sd += SD_LEN) {
#endif
          const int pw = uni(nx_pw);
          const double sbound = nx_sb;
          nx_pw = ip[pc + e + 1];
          nx_sb = sd[SD_LEN + SD_BOUND];
          const int ptype = (pw >> 12) & 15;
          const bool pfirst = (pw & P_FIRST) != 0;
          pin_geom(cur);
          Geom par;
          {
            double t6[6] = {0, 0, 0, 0, 0, 0};
            const int slot_ = pw & 63;
            switch (slot_) { MJPL_FOR_SLOTS(MJPL_SLOT_GET) default: break; }
            par.pos[0] = t6[0]; par.pos[1] = t6[1]; par.pos[2] = t6[2];
            par.m[2] = t6[3]; par.m[5] = t6[4]; par.m[8] = t6[5];
          }
          double dif[3] = {cur.pos[0] - par.pos[0], cur.pos[1] - par.pos[1], cur.pos[2] - par.pos[2]};
          const bool pass = !(dot3(dif, dif) > sbound) && active && !hit;
          if (__ballot(pass) == 0ull) continue;
#ifdef MJPL_X_SKIP_NARROW
          hit = hit || (pass && dif[0] == 12345.0);
          continue;
#endif

          const double psize[3] = {sd[SD_SIZE], sd[SD_SIZE + 1], sd[SD_SIZE + 2]};
          {
            double t6[6] = {0, 0, 0, 0, 0, 0};
            if (MBOX && ((pw >> 6) & 63) != SLOT_NONE) {  // stored box: x and y axes
              const int slot_ = (pw >> 6) & 63;
              switch (slot_) { MJPL_FOR_SLOTS(MJPL_SLOT_GET) default: break; }
            }
            par.m[0] = t6[0]; par.m[3] = t6[1]; par.m[6] = t6[2];
            par.m[1] = t6[3]; par.m[4] = t6[4]; par.m[7] = t6[5];
          }
          const bool contact = pair_contact<WBOX, MBOX>(gtype, cur, gsize, ptype, par, psize, pfirst, sd[SD_MARGIN]);
          hit = hit || (pass && contact);
        }
      }
      pc += nstored;

      if (!EMIT && store >= 0) {
        {
          const double t6[6] = {cur.pos[0], cur.pos[1], cur.pos[2], cur.m[2], cur.m[5], cur.m[8]};
          const int slot_ = store & 63;
          switch (slot_) { MJPL_FOR_SLOTS(MJPL_SLOT_PUT) default: break; }
        }
        if (MBOX && ((store >> 6) & 63) != SLOT_NONE) {
          const double t6[6] = {cur.m[0], cur.m[3], cur.m[6], cur.m[1], cur.m[4], cur.m[7]};
          const int slot_ = (store >> 6) & 63;
          switch (slot_) { MJPL_FOR_SLOTS(MJPL_SLOT_PUT) default: break; }
        }
      }
    }
  }
  return hit;
}

}  // namespace mjpl
