/*
 * mjpl_oracle.c -- CPU ORACLE (test infrastructure, NOT the product).  See mjpl_oracle.h.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math -fPIC -shared (oracle/Makefile).
 * -ffp-contract=off matters: MuJoCo's release wheels are built without FMA
 * contraction of these scalar routines being guaranteed either way; we fix the
 * IEEE-754 "one rounding per operation" reading so that the HIP path (also built
 * with -ffp-contract=off) can be compared bit-for-bit.
 *
 * Every engine routine below restates upstream MuJoCo 3.x from its published
 * algorithm ([MJ-recalled]; upstream file named per function).  The reference
 * call sites are src/mjpl/constraint/collision_constraint.py:27-30.
 */
#include "mjpl_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_MAXCON 4096

/* per-thread scratch, grown on demand (a malloc per configuration serialises 256 threads) */
static __thread double *tls_buf[5];
static __thread size_t tls_cap[5];

static double *scratch(int slot, size_t ndoubles) {
  if (tls_cap[slot] < ndoubles) {
    free(tls_buf[slot]);
    tls_buf[slot] = (double *)malloc(sizeof(double) * ndoubles);
    tls_cap[slot] = tls_buf[slot] ? ndoubles : 0;
  }
  return tls_buf[slot];
}

#include "orc_math.h"

#ifdef ORC_COUNT_FLOPS
orc_flops_t orc_flops;
void orc_flops_reset(void) { memset(&orc_flops, 0, sizeof(orc_flops)); }
void orc_flops_get(long long *out5) {
  out5[0] = orc_flops.add; out5[1] = orc_flops.mul; out5[2] = orc_flops.div; out5[3] = orc_flops.sqrt; out5[4] = orc_flops.sincos;
}
#endif

int orc_trig_mode = 0;
void orc_set_trig(int mode) { orc_trig_mode = mode ? 1 : 0; }
int orc_get_trig(void) { return orc_trig_mode; }

/* ------------------------------------------------------------------ a3: mj_kinematics
 * [MJ-recalled: engine_core_smooth.c mj_kinematics + mj_local2Global].
 * Reference call site: collision_constraint.py:28. */
int orc_kinematics(const orc_model *m, const double *qpos,
                   double *xpos_out, double *xquat_out, double *xmat_out,
                   double *geom_xpos, double *geom_xmat) {
  const int nb = m->nbody;
  double *xpos = scratch(0, (size_t)nb * 16);
  if (!xpos) return ORC_E_OVERFLOW;
  double *xquat = xpos + 3 * nb;
  double *xmat = xquat + 4 * nb;
  int status = ORC_OK;

  /* world body */
  xpos[0] = xpos[1] = xpos[2] = 0;
  xquat[0] = 1; xquat[1] = xquat[2] = xquat[3] = 0;
  quat2mat(xmat, xquat);

  for (int i = 1; i < nb; i++) {
    const int pid = m->body_parentid[i];
    const double *bodypos = m->body_pos + 3 * i;
    const double *bodyquat = m->body_quat + 4 * i;
    double p[3], q[4];

    /* fixed translation and rotation relative to the parent */
    mul_mat_vec3(p, xmat + 9 * pid, bodypos);
    p[0] += xpos[3 * pid]; p[1] += xpos[3 * pid + 1]; p[2] += xpos[3 * pid + 2];
    ORC_FL(3, 0, 0, 0);
    mul_quat(q, xquat + 4 * pid, bodyquat);

    /* accumulate joints */
    const int jadr = m->body_jntadr[i];
    for (int j = 0; j < m->body_jntnum[i]; j++) {
      const int jid = jadr + j;
      const int qadr = m->jnt_qposadr[jid];
      const int jtype = m->jnt_type[jid];
      double xaxis[3], xanchor[3];
      rot_vec_quat(xaxis, m->jnt_axis + 3 * jid, q);
      rot_vec_quat(xanchor, m->jnt_pos + 3 * jid, q);
      xanchor[0] += p[0]; xanchor[1] += p[1]; xanchor[2] += p[2];
      ORC_FL(3, 0, 0, 0);

      if (jtype == ORC_JNT_SLIDE) {
        const double d = qpos[qadr] - m->qpos0[qadr];
        p[0] += xaxis[0] * d; p[1] += xaxis[1] * d; p[2] += xaxis[2] * d;
        ORC_FL(4, 3, 0, 0);
      } else if (jtype == ORC_JNT_HINGE) {
        double qloc[4], vec[3];
        axis_angle2quat(qloc, m->jnt_axis + 3 * jid, qpos[qadr] - m->qpos0[qadr]);
        mul_quat(q, q, qloc);
        /* correct for off-centre rotation */
        rot_vec_quat(vec, m->jnt_pos + 3 * jid, q);
        p[0] = xanchor[0] - vec[0]; p[1] = xanchor[1] - vec[1]; p[2] = xanchor[2] - vec[2];
        ORC_FL(4, 0, 0, 0);  /* + the joint value minus its reference */
      } else {
        status = ORC_E_JOINT;
      }
    }

    normalize4(q);
    memcpy(xquat + 4 * i, q, sizeof(q));
    memcpy(xpos + 3 * i, p, sizeof(p));
    quat2mat(xmat + 9 * i, q);
  }

  /* geoms: mj_local2Global (general branch; the sameframe shortcuts are value-identical) */
  for (int g = 0; g < m->ngeom; g++) {
    const int b = m->geom_bodyid[g];
    if (geom_xpos) {
      double p[3];
      mul_mat_vec3(p, xmat + 9 * b, m->geom_pos + 3 * g);
      geom_xpos[3 * g + 0] = p[0] + xpos[3 * b + 0];
      geom_xpos[3 * g + 1] = p[1] + xpos[3 * b + 1];
      geom_xpos[3 * g + 2] = p[2] + xpos[3 * b + 2];
      ORC_FL(3, 0, 0, 0);
    }
    if (geom_xmat) {
      double q[4];
      mul_quat(q, xquat + 4 * b, m->geom_quat + 4 * g);
      quat2mat(geom_xmat + 9 * g, q);
    }
  }

  if (xpos_out) memcpy(xpos_out, xpos, sizeof(double) * 3 * (size_t)nb);
  if (xquat_out) memcpy(xquat_out, xquat, sizeof(double) * 4 * (size_t)nb);
  if (xmat_out) memcpy(xmat_out, xmat, sizeof(double) * 9 * (size_t)nb);
  return status;
}

/* ------------------------------------------------------------------ a5: narrowphase primitives
 * Verdict-only restatements ("ncon > 0"); contact frames are not needed because the
 * reference reads nothing but contact.geom (collision_constraint.py:30).
 * [MJ-recalled: engine_collision_primitive.c, engine_collision_box.c] */

/* mjraw_SphereSphere */
static int raw_sphere_sphere(double margin, const double *pos1, double r1,
                             const double *pos2, double r2) {
  double dif[3] = {pos1[0] - pos2[0], pos1[1] - pos2[1], pos1[2] - pos2[2]};
  double cdist_sqr = dot3(dif, dif);
  double min_dist = margin + r1 + r2;
  ORC_FL(5, 1, 0, 0);
  if (cdist_sqr > min_dist * min_dist) return 0;
  return 1;
}

/* mjc_PlaneSphere: plane normal is the z column of mat1 */
static int plane_sphere(double margin, const double *pos1, const double *mat1,
                        const double *pos2, double r2) {
  double n[3] = {mat1[2], mat1[5], mat1[8]};
  double tmp[3] = {pos2[0] - pos1[0], pos2[1] - pos1[1], pos2[2] - pos1[2]};
  double cdist = dot3(tmp, n);
  ORC_FL(4, 0, 0, 0);
  if (cdist > margin + r2) return 0;
  return 1;
}

/* mjc_PlaneCapsule: sphere-plane test at both segment ends */
static int plane_capsule(double margin, const double *pos1, const double *mat1,
                         const double *pos2, const double *mat2, const double *size2) {
  double axis[3] = {mat2[2], mat2[5], mat2[8]};
  double seg[3] = {size2[1] * axis[0], size2[1] * axis[1], size2[1] * axis[2]};
  double e[3];
  ORC_FL(6, 3, 0, 0);
  e[0] = pos2[0] + seg[0]; e[1] = pos2[1] + seg[1]; e[2] = pos2[2] + seg[2];
  int n1 = plane_sphere(margin, pos1, mat1, e, size2[0]);
  e[0] = pos2[0] - seg[0]; e[1] = pos2[1] - seg[1]; e[2] = pos2[2] - seg[2];
  int n2 = plane_sphere(margin, pos1, mat1, e, size2[0]);
  return n1 + n2;
}

/* mjc_PlaneBox: a corner counts if it is below the centre (ldist <= 0) and within margin */
static int plane_box(double margin, const double *pos1, const double *mat1,
                     const double *pos2, const double *mat2, const double *size2) {
  double norm[3] = {mat1[2], mat1[5], mat1[8]};
  double dif[3] = {pos2[0] - pos1[0], pos2[1] - pos1[1], pos2[2] - pos1[2]};
  double dist = dot3(dif, norm);
  ORC_FL(3, 0, 0, 0);
  int cnt = 0;
  for (int i = 0; i < 8; i++) {
    double vec[3], corner[3];
    vec[0] = (i & 1) ? size2[0] : -size2[0];
    vec[1] = (i & 2) ? size2[1] : -size2[1];
    vec[2] = (i & 4) ? size2[2] : -size2[2];
    mul_mat_vec3(corner, mat2, vec);
    double ldist = dot3(norm, corner);
    ORC_FL(1, 0, 0, 0);
    if (dist + ldist > margin || ldist > 0) continue;
    if (++cnt >= 4) return 4;
  }
  return cnt;
}

/* mjc_SphereCapsule: sphere vs clamped projection on the segment */
static int sphere_capsule(double margin, const double *pos1, double r1,
                          const double *pos2, const double *mat2, const double *size2) {
  double len = size2[1];
  double axis[3] = {mat2[2], mat2[5], mat2[8]};
  double vec[3] = {pos1[0] - pos2[0], pos1[1] - pos2[1], pos1[2] - pos2[2]};
  double x = clipd(dot3(axis, vec), -len, len);
  ORC_FL(6, 3, 0, 0);
  vec[0] = axis[0] * x + pos2[0];
  vec[1] = axis[1] * x + pos2[1];
  vec[2] = axis[2] * x + pos2[2];
  return raw_sphere_sphere(margin, pos1, r1, vec, size2[0]);
}

/* mjc_CapsuleCapsule: segment-segment closest points, parallel special case */
static int capsule_capsule(double margin, const double *pos1, const double *mat1,
                           const double *size1, const double *pos2, const double *mat2,
                           const double *size2) {
  double axis1[3] = {mat1[2] * size1[1], mat1[5] * size1[1], mat1[8] * size1[1]};
  double axis2[3] = {mat2[2] * size2[1], mat2[5] * size2[1], mat2[8] * size2[1]};
  double dif[3] = {pos1[0] - pos2[0], pos1[1] - pos2[1], pos1[2] - pos2[2]};

  double ma = dot3(axis1, axis1);
  double mb = -dot3(axis1, axis2);
  double mc = dot3(axis2, axis2);
  double u = -dot3(axis1, dif);
  double v = dot3(axis2, dif);
  double det = ma * mc - mb * mb;
  double vec1[3], vec2[3];
  ORC_FL(4, 8, 0, 0);  /* the two scaled axes, dif, det (the five dot products count themselves) */

  if (fabs(det) >= ORC_MINVAL) {
    double x1 = (mc * u - mb * v) / det;
    double x2 = (ma * v - mb * u) / det;
    ORC_FL(2, 4, 2, 0);

    if (x1 > 1) {
      x1 = 1;
      x2 = (v - mb) / mc;
      ORC_FL(1, 0, 1, 0);
    } else if (x1 < -1) {
      x1 = -1;
      x2 = (v + mb) / mc;
      ORC_FL(1, 0, 1, 0);
    }
    if (x2 > 1) {
      x2 = 1;
      x1 = clipd((u - mb) / ma, -1, 1);
      ORC_FL(1, 0, 1, 0);
    } else if (x2 < -1) {
      x2 = -1;
      x1 = clipd((u + mb) / ma, -1, 1);
      ORC_FL(1, 0, 1, 0);
    }
    ORC_FL(6, 6, 0, 0);

    for (int k = 0; k < 3; k++) {
      vec1[k] = pos1[k] + axis1[k] * x1;
      vec2[k] = pos2[k] + axis2[k] * x2;
    }
    return raw_sphere_sphere(margin, vec1, size1[0], vec2, size2[0]);
  }

  /* parallel axes: up to two sphere tests per end */
  ORC_FL(4 * 7, 4 * 3, 4, 0);  /* (rare: counted as all four end tests) */
  int n = 0;
  double x1, x2;
  /* x1 = 1 */
  for (int k = 0; k < 3; k++) vec1[k] = pos1[k] + axis1[k];
  x2 = clipd((v - mb) / mc, -1, 1);
  for (int k = 0; k < 3; k++) vec2[k] = pos2[k] + axis2[k] * x2;
  n += raw_sphere_sphere(margin, vec1, size1[0], vec2, size2[0]);
  /* x1 = -1 */
  for (int k = 0; k < 3; k++) vec1[k] = pos1[k] - axis1[k];
  x2 = clipd((v + mb) / mc, -1, 1);
  for (int k = 0; k < 3; k++) vec2[k] = pos2[k] + axis2[k] * x2;
  n += raw_sphere_sphere(margin, vec1, size1[0], vec2, size2[0]);
  if (n >= 2) return n;
  /* x2 = 1 */
  for (int k = 0; k < 3; k++) vec2[k] = pos2[k] + axis2[k];
  x1 = clipd((u - mb) / ma, -1, 1);
  for (int k = 0; k < 3; k++) vec1[k] = pos1[k] + axis1[k] * x1;
  n += raw_sphere_sphere(margin, vec1, size1[0], vec2, size2[0]);
  if (n >= 2) return n;
  /* x2 = -1 */
  for (int k = 0; k < 3; k++) vec2[k] = pos2[k] - axis2[k];
  x1 = clipd((u + mb) / ma, -1, 1);
  for (int k = 0; k < 3; k++) vec1[k] = pos1[k] + axis1[k] * x1;
  n += raw_sphere_sphere(margin, vec1, size1[0], vec2, size2[0]);
  return n;
}

/* verdict core shared by sphere-box and capsule-box: centre c is in the BOX frame.
 * mjraw_SphereBox: clamp the centre to the box, contact iff |clamped-c| - r <= margin
 * (centre inside the box gives dist 0 => contact). */
static int sphere_box_local(double margin, const double *c, double r, const double *size2) {
  double d[3];
  for (int k = 0; k < 3; k++) d[k] = clipd(c[k], -size2[k], size2[k]) - c[k];
  double dist = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
  ORC_FL(3 + 2 + 1, 3, 0, 1);
  if (dist - r > margin) return 0;
  return 1;
}

/* mjc_SphereBox */
static int sphere_box(double margin, const double *pos1, double r1,
                      const double *pos2, const double *mat2, const double *size2) {
  double tmp[3] = {pos1[0] - pos2[0], pos1[1] - pos2[1], pos1[2] - pos2[2]};
  double center[3];
  ORC_FL(3, 0, 0, 0);
  mul_matT_vec3(center, mat2, tmp);
  return sphere_box_local(margin, center, r1, size2);
}

/* mjc_CapsuleBox -- DEVIATION (documented in DESIGN.md): upstream finds the segment
 * point closest to the box with a face/edge/corner case analysis and then runs
 * sphere-box tests there; that routine cannot be restated op-for-op from memory.
 * We compute the same quantity exactly: t* = argmin_{t in [-1,1]} dist^2(p + t*h, box)
 * (convex, piecewise quadratic; its derivative g(t) = sum_k h_k * excess_k(t) is
 * monotone piecewise linear, so we bracket the root between breakpoints and solve
 * the linear piece), then apply the sphere-box verdict at p + t*h.  "parity unpinned". */
static double capbox_g(const double *p, const double *h, const double *s, double t) {
  double g = 0;
  for (int k = 0; k < 3; k++) {
    double x = p[k] + t * h[k];
    double e = x - fmin(fmax(x, -s[k]), s[k]);
    g = g + h[k] * e;
    ORC_FL(3, 2, 0, 0);
  }
  return g;
}

static int capsule_box(double margin, const double *pos1, const double *mat1,
                       const double *size1, const double *pos2, const double *mat2,
                       const double *size2) {
  double tmp[3] = {pos1[0] - pos2[0], pos1[1] - pos2[1], pos1[2] - pos2[2]};
  double p[3], a[3], h[3], inv[3];
  double axis[3] = {mat1[2], mat1[5], mat1[8]};
  mul_matT_vec3(p, mat2, tmp);
  mul_matT_vec3(a, mat2, axis);
  for (int k = 0; k < 3; k++) {
    h[k] = a[k] * size1[1];
    inv[k] = 1 / h[k]; /* h == 0: +-inf, the breakpoint becomes +-inf or NaN and is skipped */
    ORC_FL(1, 1, 1, 0); /* (+ one subtraction of tmp per axis) */
  }

  double lo = -1, hi = 1;
  double glo = capbox_g(p, h, size2, lo);
  double ghi = capbox_g(p, h, size2, hi);
  /* the six face breakpoints t = (+-s_k - p_k)/h_k, k = 0,1,2, minus before plus */
  for (int k = 0; k < 3; k++) {
    for (int sgn = -1; sgn <= 1; sgn += 2) {
      double tb = (sgn * size2[k] - p[k]) * inv[k];
      ORC_FL(1, 2, 0, 0);
      int inside = (tb > lo && tb < hi);
      double gb = capbox_g(p, h, size2, tb);
      if (inside && gb <= 0) { lo = tb; glo = gb; }
      if (inside && !(gb <= 0)) { hi = tb; ghi = gb; }
    }
  }
  double den = ghi - glo;
  double t = (den > 0) ? lo + (hi - lo) * ((0 - glo) / den) : lo;
  ORC_FL(1 + 3 + 3, 1 + 3, 1, 0);  /* den, the linear piece, and the point p + t h */
  /* minimum at an end of the segment (decided on the values at -1 and +1) */
  double g_m1 = capbox_g(p, h, size2, -1), g_p1 = capbox_g(p, h, size2, 1);
  if (g_m1 >= 0) t = -1;
  else if (g_p1 <= 0) t = 1;
  double c[3] = {p[0] + t * h[0], p[1] + t * h[1], p[2] + t * h[2]};
  return sphere_box_local(margin, c, size1[0], size2);
}

/* mjc_BoxBox -- DEVIATION: verdict by the 15-axis separating-axis test with margin
 * (upstream runs the same SAT to find the penetration axis and then clips faces to
 * produce contacts).  "parity unpinned". */
static int box_box(double margin, const double *pos1, const double *mat1, const double *size1,
                   const double *pos2, const double *mat2, const double *size2) {
  double d[3] = {pos2[0] - pos1[0], pos2[1] - pos1[1], pos2[2] - pos1[2]};
  double R[9], A[9], t[3];
  /* R = mat1^T * mat2, t = mat1^T * d */
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++)
      R[3 * i + j] = mat1[i] * mat2[j] + mat1[3 + i] * mat2[3 + j] + mat1[6 + i] * mat2[6 + j];
  mul_matT_vec3(t, mat1, d);
  ORC_FL(3 + 18, 27, 0, 0);
  for (int k = 0; k < 9; k++) A[k] = fabs(R[k]);
  /* face axes of box 1 */
  for (int i = 0; i < 3; i++) {
    double rb = size2[0] * A[3 * i] + size2[1] * A[3 * i + 1] + size2[2] * A[3 * i + 2];
    ORC_FL(4, 3, 0, 0);
    if (fabs(t[i]) - (size1[i] + rb) > margin) return 0;
  }
  /* face axes of box 2 */
  for (int j = 0; j < 3; j++) {
    double ra = size1[0] * A[j] + size1[1] * A[3 + j] + size1[2] * A[6 + j];
    double tj = t[0] * R[j] + t[1] * R[3 + j] + t[2] * R[6 + j];
    ORC_FL(6, 6, 0, 0);
    if (fabs(tj) - (ra + size2[j]) > margin) return 0;
  }
  /* edge x edge axes, normalised so that margin keeps its metric meaning */
  for (int i = 0; i < 3; i++) {
    const int i1 = (i + 1) % 3, i2 = (i + 2) % 3;
    for (int j = 0; j < 3; j++) {
      const int j1 = (j + 1) % 3, j2 = (j + 2) % 3;
      double len2 = 1 - R[3 * i + j] * R[3 * i + j];
      ORC_FL(1, 1, 0, 0);
      if (len2 < 1e-12) continue; /* near-parallel edges: covered by the face axes */
      double ra = size1[i1] * A[3 * i2 + j] + size1[i2] * A[3 * i1 + j];
      double rb = size2[j1] * A[3 * i + j2] + size2[j2] * A[3 * i + j1];
      double tl = t[i2] * R[3 * i1 + j] - t[i1] * R[3 * i2 + j];
      ORC_FL(5, 7, 0, 1);
      if (fabs(tl) - (ra + rb) > margin * sqrt(len2)) return 0;
    }
  }
  return 1;
}

/* dispatch with type1 <= type2, as mj_collision's function table is indexed */
static int pair_dispatch(int t1, const double *p1, const double *m1, const double *s1,
                         int t2, const double *p2, const double *m2, const double *s2,
                         double margin) {
  if (t1 > t2) return pair_dispatch(t2, p2, m2, s2, t1, p1, m1, s1, margin);
  switch (t1) {
    case ORC_GEOM_PLANE:
      if (t2 == ORC_GEOM_SPHERE) return plane_sphere(margin, p1, m1, p2, s2[0]) > 0;
      if (t2 == ORC_GEOM_CAPSULE) return plane_capsule(margin, p1, m1, p2, m2, s2) > 0;
      if (t2 == ORC_GEOM_BOX) return plane_box(margin, p1, m1, p2, m2, s2) > 0;
      return ORC_E_PAIRTYPE;
    case ORC_GEOM_SPHERE:
      if (t2 == ORC_GEOM_SPHERE) return raw_sphere_sphere(margin, p1, s1[0], p2, s2[0]) > 0;
      if (t2 == ORC_GEOM_CAPSULE) return sphere_capsule(margin, p1, s1[0], p2, m2, s2) > 0;
      if (t2 == ORC_GEOM_BOX) return sphere_box(margin, p1, s1[0], p2, m2, s2) > 0;
      return ORC_E_PAIRTYPE;
    case ORC_GEOM_CAPSULE:
      if (t2 == ORC_GEOM_CAPSULE) return capsule_capsule(margin, p1, m1, s1, p2, m2, s2) > 0;
      if (t2 == ORC_GEOM_BOX) return capsule_box(margin, p1, m1, s1, p2, m2, s2) > 0;
      return ORC_E_PAIRTYPE;
    case ORC_GEOM_BOX:
      if (t2 == ORC_GEOM_BOX) return box_box(margin, p1, m1, s1, p2, m2, s2) > 0;
      return ORC_E_PAIRTYPE;
    default:
      return ORC_E_PAIRTYPE;
  }
}

int orc_pair_test(int32_t type1, const double *pos1, const double *mat1, const double *size1,
                  int32_t type2, const double *pos2, const double *mat2, const double *size2,
                  double margin) {
  return pair_dispatch(type1, pos1, mat1, size1, type2, pos2, mat2, size2, margin);
}

/* ------------------------------------------------------------------ a4: mj_collision
 * [MJ-recalled: engine_collision_driver.c].  The broadphase (body AABB sweep-and-prune)
 * and midphase (BVH) only prune pairs whose bounding volumes are disjoint, so the set of
 * contacts equals "all geom pairs that pass the filters and whose narrowphase reports a
 * contact"; we enumerate that set directly.  Reference call site collision_constraint.py:29. */

/* filterBitmask + filterBodyPair.  returns 1 if the pair is filtered OUT. */
static int pair_filtered(const orc_model *m, int g1, int g2) {
  const int ct1 = m->geom_contype[g1], ca1 = m->geom_conaffinity[g1];
  const int ct2 = m->geom_contype[g2], ca2 = m->geom_conaffinity[g2];
  if (!(ct1 & ca2) && !(ct2 & ca1)) return 1;
  const int b1 = m->geom_bodyid[g1], b2 = m->geom_bodyid[g2];
  const int w1 = m->body_weldid[b1], w2 = m->body_weldid[b2];
  if (w1 == w2) return 1;
  const int wp1 = m->body_weldid[m->body_parentid[w1]];
  const int wp2 = m->body_weldid[m->body_parentid[w2]];
  if (w1 != 0 && w2 != 0 && (w1 == wp2 || w2 == wp1)) return 1;
  return 0;
}

/* mj_collideSphere-style bounding test: returns 1 if the pair can be skipped. */
static int bound_skip(const orc_model *m, const double *gx, const double *gm, int g1, int g2,
                      double margin) {
  const double r1 = m->geom_rbound[g1], r2 = m->geom_rbound[g2];
  if (r1 > 0 && r2 > 0) {
    const double *a = gx + 3 * g1, *b = gx + 3 * g2;
    double dif[3] = {a[0] - b[0], a[1] - b[1], a[2] - b[2]};
    double bound = r1 + r2 + margin;
    ORC_FL(5, 1, 0, 0);
    if (dot3(dif, dif) > bound * bound) return 1;
    return 0;
  }
  /* plane vs bounded geom */
  int gp = -1, gs = -1;
  if (m->geom_type[g1] == ORC_GEOM_PLANE && r2 > 0) { gp = g1; gs = g2; }
  if (m->geom_type[g2] == ORC_GEOM_PLANE && r1 > 0) { gp = g2; gs = g1; }
  if (gp >= 0) {
    const double *pm = gm + 9 * gp;
    double n[3] = {pm[2], pm[5], pm[8]};
    double dif[3] = {gx[3 * gs] - gx[3 * gp], gx[3 * gs + 1] - gx[3 * gp + 1],
                     gx[3 * gs + 2] - gx[3 * gp + 2]};
    ORC_FL(4, 0, 0, 0);
    if (dot3(dif, n) > margin + m->geom_rbound[gs]) return 1;
  }
  return 0;
}

static int has_collision_fn(int t1, int t2) {
  if (t1 > t2) { int t = t1; t1 = t2; t2 = t; }
  if (t1 == ORC_GEOM_PLANE && (t2 == ORC_GEOM_PLANE || t2 == ORC_GEOM_HFIELD)) return 0;
  return 1;
}

int orc_collision(const orc_model *m, const double *gx, const double *gm,
                  int32_t *contact_geom, int32_t maxcon, int32_t *ncon_out) {
  int ncon = 0, status = ORC_OK;
  for (int g1 = 0; g1 < m->ngeom; g1++) {
    for (int g2 = g1 + 1; g2 < m->ngeom; g2++) {
      if (pair_filtered(m, g1, g2)) continue;
      if (!has_collision_fn(m->geom_type[g1], m->geom_type[g2])) continue;
      double margin = fmax(m->geom_margin[g1], m->geom_margin[g2]);
      if (bound_skip(m, gx, gm, g1, g2, margin)) continue;
      int hit = pair_dispatch(m->geom_type[g1], gx + 3 * g1, gm + 9 * g1, m->geom_size + 3 * g1,
                              m->geom_type[g2], gx + 3 * g2, gm + 9 * g2, m->geom_size + 3 * g2,
                              margin);
      if (hit < 0) { status = hit; continue; }
      if (hit) {
        if (ncon >= maxcon) { *ncon_out = ncon; return ORC_E_OVERFLOW; }
        /* table order: smaller geom type first */
        int a = g1, b = g2;
        if (m->geom_type[g1] > m->geom_type[g2]) { a = g2; b = g1; }
        contact_geom[2 * ncon] = a;
        contact_geom[2 * ncon + 1] = b;
        ncon++;
      }
    }
  }
  *ncon_out = ncon;
  return status;
}

/* ------------------------------------------------------------------ a6: CollisionRuleset.obeys_ruleset
 * collision_constraint.py:66-95.  `allowed` holds body-id pairs already sorted per row
 * (np.sort(body_ids, axis=1), collision_constraint.py:64). */
int orc_obeys_ruleset(const orc_model *m, const int32_t *contact_geom, int32_t ncon,
                      const int32_t *allowed, int32_t nallowed) {
  if (ncon == 0) return 1;          /* :83-85 no collisions */
  if (nallowed == 0) return 0;      /* :86-88 collisions, none allowed */
  for (int c = 0; c < ncon; c++) {  /* :93-95 every contact must match an allowed pair */
    int b1 = m->geom_bodyid[contact_geom[2 * c]];
    int b2 = m->geom_bodyid[contact_geom[2 * c + 1]];
    if (b1 > b2) { int t = b1; b1 = b2; b2 = t; }
    int ok = 0;
    for (int a = 0; a < nallowed && !ok; a++)
      ok = (allowed[2 * a] == b1 && allowed[2 * a + 1] == b2);
    if (!ok) return 0;
  }
  return 1;
}

/* ------------------------------------------------------------------ a1: valid_config
 * collision_constraint.py:26-30: qpos <- q; mj_kinematics; mj_collision; ruleset. */
int orc_valid_config(const orc_model *m, const int32_t *allowed, int32_t nallowed,
                     const double *qpos) {
  double *buf = scratch(1, (size_t)m->ngeom * 12);
  int32_t *con = (int32_t *)scratch(2, ORC_MAXCON);
  if (!buf || !con) return ORC_E_OVERFLOW;
  double *gx = buf, *gm = buf + 3 * m->ngeom;
  int32_t ncon = 0;
  int st = orc_kinematics(m, qpos, NULL, NULL, NULL, gx, gm);
  if (st == ORC_OK) st = orc_collision(m, gx, gm, con, ORC_MAXCON, &ncon);
  int res = (st == ORC_OK) ? orc_obeys_ruleset(m, con, ncon, allowed, nallowed) : st;
  return res;
}

/* ------------------------------------------------------------------ a8: _step
 * planning/utils.py:167-185.  np.linalg.norm on a 1-D float64 array is sqrt(x.dot(x));
 * BLAS ddot's summation order is unspecified, we FIX it to the sequential left-to-right
 * sum without FMA (DESIGN.md "waypoint semantics"). */
static double norm_seq(const double *v, int n) {
  double s = 0;
  for (int k = 0; k < n; k++) s = s + v[k] * v[k];
  ORC_FL(n, n, 0, 1);
  return sqrt(s);
}

static int array_equal(const double *a, const double *b, int n) {
  for (int k = 0; k < n; k++)
    if (!(a[k] == b[k])) return 0;
  return 1;
}

void orc_step(const double *start, const double *target, int32_t n, double max_step,
              double *out) {
  if (array_equal(start, target, n)) { /* :180-181 */
    for (int k = 0; k < n; k++) out[k] = start[k];
    return;
  }
  double dir[64];
  double *d = n <= 64 ? dir : (double *)malloc(sizeof(double) * (size_t)n);
  for (int k = 0; k < n; k++) d[k] = target[k] - start[k];     /* :182 */
  double mag = norm_seq(d, n);                                 /* :183 */
  double stepmag = max_step < mag ? max_step : mag;            /* min(max_step_dist, magnitude) */
  for (int k = 0; k < n; k++) out[k] = start[k] + (d[k] / mag) * stepmag; /* :184-185 */
  ORC_FL(2 * n, n, n, 0);
  if (d != dir) free(d);
}

/* ------------------------------------------------------------------ a9: _valid_collision_interval
 * planning/utils.py:188-216.  Waypoints are generated by repeated _step until one is
 * array_equal to `end`; first and last are dropped; all() short-circuits. */
#define ORC_MAX_WAYPOINTS (1 << 22)

int orc_valid_collision_interval(const orc_model *m, const int32_t *allowed, int32_t nallowed,
                                 const double *start, const double *end, double step_dist,
                                 int32_t *nwaypoints, int32_t *first_bad) {
  const int n = m->nq;
  double *w = scratch(3, 2 * (size_t)n);  /* per-thread, grown once: no malloc per edge */
  if (!w) return ORC_E_OVERFLOW;
  double *nx = w + n;
  int res = 1, idx = 0, bad = 0;
  for (int k = 0; k < n; k++) {
    if (!isfinite(start[k]) || !isfinite(end[k])) return ORC_E_NONFINITE;
    w[k] = start[k];
  }
  /* w walks start -> end; every w that is neither start nor end is an interior waypoint. */
  while (!array_equal(w, end, n)) {
    orc_step(w, end, n, step_dist, nx);
    memcpy(w, nx, sizeof(double) * (size_t)n);
    if (array_equal(w, end, n)) break;   /* that was the last element: dropped by [1:-1] */
    idx++;
    if (idx > ORC_MAX_WAYPOINTS) { res = ORC_E_NONFINITE; break; }
    if (res == 1) {                      /* all(...) stops evaluating after the first False */
      int v = orc_valid_config(m, allowed, nallowed, w);
      if (v < 0) { res = v; break; }
      if (!v) { res = 0; bad = idx; }
    }
    /* keep counting waypoints (list is fully built before all() runs, :210-214) */
  }
  if (nwaypoints) *nwaypoints = idx;
  if (first_bad) *first_bad = bad;
  return res;
}

/* ------------------------------------------------------------------ batched drivers (pthreads) */

static void gather_q(const orc_model *m, const orc_batch *b, const double *Q, int64_t N,
                     int64_t i, double *qpos) {
  for (int k = 0; k < m->nq; k++) qpos[k] = b->qpos_base[k];
  for (int c = 0; c < b->nplan; c++)
    qpos[b->qidx[c]] = b->layout == 0 ? Q[(int64_t)c * N + i] : Q[i * b->nplan + c];
}

typedef struct {
  const orc_model *m; const orc_batch *b;
  const double *QA, *QB; int64_t N; double step;
  uint8_t *valid; int32_t *first_bad, *ncheck; int mode;
} job_t;

/* items [lo, hi) of a batch job; returns the last negative status seen (or ORC_OK) */
static int job_range(const job_t *j, int64_t lo, int64_t hi) {
  const orc_model *m = j->m;
  int status = ORC_OK;
  double *qa = scratch(4, 2 * (size_t)m->nq);
  if (!qa) return ORC_E_OVERFLOW;
  double *qb = qa + m->nq;
  for (int64_t i = lo; i < hi; i++) {
    if (j->mode == 0) {
      gather_q(m, j->b, j->QA, j->N, i, qa);
      int v = orc_valid_config(m, j->b->allowed, j->b->nallowed, qa);
      if (v < 0) { status = v; v = 0; }
      j->valid[i] = (uint8_t)v;
    } else {
      gather_q(m, j->b, j->QA, j->N, i, qa);
      gather_q(m, j->b, j->QB, j->N, i, qb);
      int32_t nchk = 1, fb = -1;
      int v = orc_valid_config(m, j->b->allowed, j->b->nallowed, qb);
      if (v < 0) { status = v; v = 0; }
      if (!v) {
        fb = 0;
      } else {
        int32_t nwp = 0, bad = 0;
        v = orc_valid_collision_interval(m, j->b->allowed, j->b->nallowed, qa, qb, j->step,
                                         &nwp, &bad);
        if (v < 0) { status = v; v = 0; bad = 0; }
        if (!v) { fb = bad; nchk += bad; } else { nchk += nwp; }
      }
      j->valid[i] = (uint8_t)v;
      if (j->first_bad) j->first_bad[i] = fb;
      if (j->ncheck) j->ncheck[i] = nchk;
    }
  }
  return status;
}

/* Persistent worker pool.  Threads are created once (grown on demand up to ORC_MAX_THREADS) and
 * sleep on a condition variable between batches; a batch is cut into chunks that the workers --
 * and the calling thread -- take from a shared atomic cursor, so a pass over a few hundred
 * thousand edges (milliseconds of work per core) pays neither 256 pthread_create calls nor the
 * imbalance of a static partition (edges that end in an obstacle cost a fifth of a free one). */
#define ORC_MAX_THREADS 512
#define ORC_CHUNK 64

static struct {
  pthread_mutex_t mu;
  pthread_cond_t wake, done;
  pthread_t th[ORC_MAX_THREADS];
  int nthreads;           /* workers created so far */
  unsigned long long gen; /* batch generation: a worker runs each generation at most once */
  int want;               /* workers that should take part in the current generation */
  int running;            /* workers still inside the current generation */
  const job_t *job;
  int64_t cursor, N;      /* next unclaimed item (atomic) */
  int status;
} pool = {PTHREAD_MUTEX_INITIALIZER, PTHREAD_COND_INITIALIZER, PTHREAD_COND_INITIALIZER,
          {0}, 0, 0, 0, 0, NULL, 0, 0, ORC_OK};

static int pool_drain(const job_t *j) {
  int status = ORC_OK;
  for (;;) {
    const int64_t lo = __atomic_fetch_add(&pool.cursor, (int64_t)ORC_CHUNK, __ATOMIC_RELAXED);
    if (lo >= pool.N) break;
    const int64_t hi = lo + ORC_CHUNK < pool.N ? lo + ORC_CHUNK : pool.N;
    const int st = job_range(j, lo, hi);
    if (st != ORC_OK) status = st;
  }
  return status;
}

static void *pool_worker(void *arg) {
  const int id = (int)(intptr_t)arg;
  unsigned long long seen = 0;
  pthread_mutex_lock(&pool.mu);
  for (;;) {
    while (pool.gen == seen || id >= pool.want) {
      if (pool.gen != seen && id >= pool.want) seen = pool.gen;  /* not invited to this batch */
      pthread_cond_wait(&pool.wake, &pool.mu);
    }
    seen = pool.gen;
    const job_t *j = pool.job;
    pthread_mutex_unlock(&pool.mu);
    const int st = pool_drain(j);
    pthread_mutex_lock(&pool.mu);
    if (st != ORC_OK) pool.status = st;
    if (--pool.running == 0) pthread_cond_signal(&pool.done);
  }
  return NULL;
}

/* one batch at a time (callers are the tests and bench.py: single-threaded hosts) */
static pthread_mutex_t pool_batch_mu = PTHREAD_MUTEX_INITIALIZER;

static int run_jobs(job_t proto, int64_t N, int nthreads) {
  if (nthreads < 1) nthreads = 1;
  if (nthreads > ORC_MAX_THREADS) nthreads = ORC_MAX_THREADS;
  const int64_t nchunks = (N + ORC_CHUNK - 1) / ORC_CHUNK;
  if ((int64_t)nthreads > nchunks) nthreads = nchunks > 0 ? (int)nchunks : 1;
  if (nthreads == 1) return N > 0 ? job_range(&proto, 0, N) : ORC_OK;

  pthread_mutex_lock(&pool_batch_mu);
  pthread_mutex_lock(&pool.mu);
  const int helpers = nthreads - 1;  /* the caller works too */
  while (pool.nthreads < helpers) {
    if (pthread_create(&pool.th[pool.nthreads], NULL, pool_worker, (void *)(intptr_t)pool.nthreads) != 0) break;
    pthread_detach(pool.th[pool.nthreads]);
    pool.nthreads++;
  }
  const int use = helpers < pool.nthreads ? helpers : pool.nthreads;
  pool.job = &proto;
  pool.N = N;
  __atomic_store_n(&pool.cursor, (int64_t)0, __ATOMIC_RELAXED);
  pool.status = ORC_OK;
  pool.want = use;
  pool.running = use;
  pool.gen++;
  pthread_cond_broadcast(&pool.wake);
  pthread_mutex_unlock(&pool.mu);

  const int mine = pool_drain(&proto);

  pthread_mutex_lock(&pool.mu);
  while (pool.running > 0) pthread_cond_wait(&pool.done, &pool.mu);
  int status = pool.status;
  pool.want = 0;
  pthread_mutex_unlock(&pool.mu);
  pthread_mutex_unlock(&pool_batch_mu);
  if (mine != ORC_OK) status = mine;
  return status;
}

int orc_valid_configs(const orc_model *m, const orc_batch *b, const double *Q, int64_t N,
                      int32_t nthreads, uint8_t *valid) {
  job_t j;
  memset(&j, 0, sizeof(j));
  j.m = m; j.b = b; j.QA = Q; j.N = N; j.valid = valid; j.mode = 0;
  return run_jobs(j, N, nthreads);
}

int orc_valid_edges(const orc_model *m, const orc_batch *b, const double *QA, const double *QB,
                    int64_t E, double step_dist, int32_t nthreads,
                    uint8_t *valid, int32_t *first_bad, int32_t *ncheck) {
  if (!(step_dist > 0)) return ORC_E_NONFINITE;
  job_t j;
  memset(&j, 0, sizeof(j));
  j.m = m; j.b = b; j.QA = QA; j.QB = QB; j.N = E; j.step = step_dist;
  j.valid = valid; j.first_bad = first_bad; j.ncheck = ncheck; j.mode = 1;
  return run_jobs(j, E, nthreads);
}

int orc_fk_batch(const orc_model *m, const orc_batch *b, const double *Q, int64_t N,
                 double *xpos, double *xquat, double *geom_xpos, double *geom_xmat) {
  double *q = (double *)malloc(sizeof(double) * (size_t)m->nq);
  int status = ORC_OK;
  for (int64_t i = 0; i < N; i++) {
    gather_q(m, b, Q, N, i, q);
    int st = orc_kinematics(m, q, xpos ? xpos + i * 3 * m->nbody : NULL,
                            xquat ? xquat + i * 4 * m->nbody : NULL, NULL,
                            geom_xpos ? geom_xpos + i * 3 * m->ngeom : NULL,
                            geom_xmat ? geom_xmat + i * 9 * m->ngeom : NULL);
    if (st != ORC_OK) status = st;
  }
  free(q);
  return status;
}
