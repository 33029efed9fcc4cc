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
typedef const float *FP;
#else
typedef const __attribute__((address_space(4))) int *IP;
typedef const __attribute__((address_space(4))) double *DP;
typedef const __attribute__((address_space(4))) float *FP;
#endif

// Scalar type of an instantiation.  double = the exact path (MuJoCo's arithmetic, one rounding
// per operation; verdicts are final).  float = the FILTER path: the same algorithms in binary32,
// every comparison against a contact threshold classified with a tolerance into
// certain-no-contact / certain-contact / uncertain; uncertain items are re-run on the exact path.
template <class T> struct Real;
template <> struct Real<double> { static constexpr bool exact = true; typedef DP Tab; };
template <> struct Real<float> { static constexpr bool exact = false; typedef FP Tab; };

// verdict codes of the narrowphase (the exact path only produces 0 and 1)
enum : int { V_NONE = 0, V_CONTACT = 1, V_UNSURE = 2,
              V_CLEAR = 3 };  // (a generated check asked for the edge certificate: no contact, and clear of it by the edge's motion)
// The float32 filter decides a configuration only while its float32 world poses are provably
// within tol/2 of the float64 ones (DESIGN.md section 5.1b derives the bound E = A + B * C per
// model): every moving body origin within FC_MAXCOORD metres of the world origin (C), every
// hinge angle within FC_MAXANGLE radians of its reference.  Anything else goes to the exact path.
// The limits are per model: mjpl_create derives them from the kinematic chain and the tolerance.
enum : int { FC_MAXCOORD = 0, FC_MAXANGLE = 1, FC_SIZE = 4 };
constexpr float kFilterMaxAngle = 6.5f;

// ----------------------------------------------------------------------------- counter-based RNG
// splitmix64 finaliser as a stateless generator: u01(key, counter).  The frontier planner's sampler
// (mjpl_rrt.h; mirrored in NumPy by mjpl_amd/planning/parallel_rrt.py) and the IK restarts use it.
MJPL_HD uint64_t sm64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
MJPL_HD uint64_t rrt_key(uint64_t seed, uint64_t rank, uint64_t round) {
  return sm64(sm64(seed) ^ sm64((rank << 40) ^ round));
}
MJPL_HD double rrt_u01(uint64_t key, uint64_t ctr) {
  return (double)(sm64(key + ctr * 0x9E3779B97F4A7C15ull) >> 11) * 0x1.0p-53;
}

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
  H_OFF_WCULL,     // constant-table offset of the world cull table
  H_OFF_WNARROW,   // constant-table offset of the world narrowphase table
  H_NWORLD,        // static geoms
  H_NWPAD,         // ... rounded up to a multiple of 4
  H_OFF_FCONST,    // constant-table offset of the filter's limits (FC_*)
  H_SIZE
};

// body op: 6 header ints, then joints, then geoms
enum : int { B_PARENT = 0, B_DOFF, B_BODYID, B_NJNT, B_SAVE, B_NGEOM, B_SIZE };
enum : int { PARENT_CUR = 0, PARENT_STATIC = -1 };  // k > 0: restore LDS save slot k-1
// body dp: pos[3] quat[4]; if PARENT_STATIC: + ppos[3] pquat[4] pmat[9]

enum : int { J_TYPE = 0, J_QSRC, J_FLAGS, J_DOFF, J_SIZE };  // qsrc >= 0: planning column
enum : int { JF_POS_NONZERO = 1 };
// joint dp: axis[3] pos[3] qpos0 qconst

// geom record: 10 header ints, then MAX_SLOTS stored-partner words (0 where the slot is no partner)
enum : int { G_TYPE = 0, G_FLAGS, G_DOFF, G_STORE, G_GEOMID, G_SMASK, G_WMASK_LO, G_WMASK_HI,
             G_PMASK_LO, G_PMASK_HI, G_SIZE };
enum : int { GF_SAMEPOS = 1, GF_SAMEROT = 2 };
enum : int { MAX_SLOTS = 32 };
// geom constants: lpos[3] lquat[4] size[3] pad[2] | wbound[nwpad] | wmargin[nwpad]
//                 | sbound[MAX_SLOTS] | smargin[MAX_SLOTS] | ssize[MAX_SLOTS][3]
//   wbound[w]  cull bound against static geom w: (r1+r2+margin)^2, or margin + rbound for a
//              plane; +inf where the pair is disabled.  nwpad = rows rounded up to a multiple of 4
//   wmargin[w] pair margin max(margin_cur, margin_w)
//   sbound[s]  cull bound against the earlier moving geom held in register slot s (+inf if none)
enum : int { GD_LPOS = 0, GD_LQUAT = 3, GD_SIZE = 7, GD_GEOMID = 10, GD_WBOUND = 12 };
// ... | sgeom[16]: model geom id of the earlier moving geom held in slot s (as a number of the
//                  table's scalar type; exact below 2^24); GD_GEOMID: this geom's model id
enum : int { GS_BOUND = 0, GS_MARGIN = MAX_SLOTS, GS_SIZE = 2 * MAX_SLOTS, GS_GEOMID = 5 * MAX_SLOTS };

// static partners: G_WMASK (non-plane) and G_PMASK (plane) are bit masks over the rows of the
// world tables (<= 64 static geoms); G_SMASK is a bit mask over register slots.
// stored-partner word s:  bits 0..5 second slot (boxes, else SLOT_NONE) ; bits 12..15 type ;
//   bit 17 = the partner is the FIRST geom of the pair in mj_collision's (g1,g2) order
enum : int { P_FIRST = 1 << 17 };
// world tables (fixed strides so that a chunk of four rows is one wide scalar load):
//   cull table   [nwpad / 4][4 fields][4 rows]: x, y, z, info word (type | geom id << 8) of four
//                rows side by side, so that the packed binary32 culls of two rows take their
//                operands from adjacent scalar registers (row-major cost 12 s_mov per chunk)
//   narrow table [nworld][12]: z axis[3], x axis[3], y axis[3], size[3]
enum : int { WC_POS = 0, WC_INFO = 3, WC_LEN = 4 };
// table index of field f of world row w
MJPL_HD int wc_at(int w, int f) { return ((w >> 2) << 4) + (f << 2) + (w & 3); }
// candidate kinds in the queued kernels' records
enum : int { EK_PLANE = 0, EK_STATIC = 1, EK_SLOT = 2 };
enum : int { WN_ZAXIS = 0, WN_XAXIS = 3, WN_YAXIS = 6, WN_SIZE = 9, WN_LEN = 12 };

enum : int { GT_PLANE = 0, GT_SPHERE = 2, GT_CAPSULE = 3, GT_BOX = 6 };
enum : int { JT_SLIDE = 2, JT_HINGE = 3 };

// ----------------------------------------------------------------------------- small math
// [MJ-recalled: engine_util_blas.c, engine_util_spatial.c]

// The exact path (T = double) keeps one rounding per operation in MuJoCo's order; the filter
// path (T = float) lets the compiler contract a*b+c into fused multiply-adds: fewer
// instructions and a smaller error, which its tolerance band covers either way.
#define MJPL_FILTER_FMA _Pragma("clang fp contract(fast)")

template <class T>
MJPL_HD T dot3(const T *a, const T *b) {
  if constexpr (Real<T>::exact) {
    return a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
  } else {
    MJPL_FILTER_FMA
    return a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
  }
}

template <class T>
MJPL_HD T sqnorm3(T x, T y, T z) {
  if constexpr (Real<T>::exact) {
    return x * x + y * y + z * z;
  } else {
    MJPL_FILTER_FMA
    return x * x + y * y + z * z;
  }
}

template <class T>
MJPL_HD void mul_mat_vec3(T *res, const T *mat, const T *vec) {
  if constexpr (Real<T>::exact) {
    res[0] = mat[0] * vec[0] + mat[1] * vec[1] + mat[2] * vec[2];
    res[1] = mat[3] * vec[0] + mat[4] * vec[1] + mat[5] * vec[2];
    res[2] = mat[6] * vec[0] + mat[7] * vec[1] + mat[8] * vec[2];
  } else {
    MJPL_FILTER_FMA
    res[0] = mat[0] * vec[0] + mat[1] * vec[1] + mat[2] * vec[2];
    res[1] = mat[3] * vec[0] + mat[4] * vec[1] + mat[5] * vec[2];
    res[2] = mat[6] * vec[0] + mat[7] * vec[1] + mat[8] * vec[2];
  }
}

template <class T>
MJPL_HD void mul_matT_vec3(T *res, const T *mat, const T *vec) {
  res[0] = mat[0] * vec[0] + mat[3] * vec[1] + mat[6] * vec[2];
  res[1] = mat[1] * vec[0] + mat[4] * vec[1] + mat[7] * vec[2];
  res[2] = mat[2] * vec[0] + mat[5] * vec[1] + mat[8] * vec[2];
}

template <class T>
MJPL_HD void mul_quat(T *res, const T *a, const T *b) {
  T t0, t1, t2, t3;
  if constexpr (Real<T>::exact) {
    t0 = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
    t1 = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
    t2 = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
    t3 = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
  } else {
    MJPL_FILTER_FMA
    t0 = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
    t1 = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
    t2 = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
    t3 = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
  }
  res[0] = t0; res[1] = t1; res[2] = t2; res[3] = t3;
}

template <class T>
MJPL_HD void rot_vec_quat(T *res, const T *vec, const T *quat) {
  T r0, r1, r2;
  if constexpr (Real<T>::exact) {
    T t0 = quat[0] * vec[0] + quat[2] * vec[2] - quat[3] * vec[1];
    T t1 = quat[0] * vec[1] + quat[3] * vec[0] - quat[1] * vec[2];
    T t2 = quat[0] * vec[2] + quat[1] * vec[1] - quat[2] * vec[0];
    r0 = vec[0] + 2 * (quat[2] * t2 - quat[3] * t1);
    r1 = vec[1] + 2 * (quat[3] * t0 - quat[1] * t2);
    r2 = vec[2] + 2 * (quat[1] * t1 - quat[2] * t0);
  } else {
    MJPL_FILTER_FMA
    T t0 = quat[0] * vec[0] + quat[2] * vec[2] - quat[3] * vec[1];
    T t1 = quat[0] * vec[1] + quat[3] * vec[0] - quat[1] * vec[2];
    T t2 = quat[0] * vec[2] + quat[1] * vec[1] - quat[2] * vec[0];
    r0 = vec[0] + 2 * (quat[2] * t2 - quat[3] * t1);
    r1 = vec[1] + 2 * (quat[3] * t0 - quat[1] * t2);
    r2 = vec[2] + 2 * (quat[1] * t1 - quat[2] * t0);
  }
  res[0] = r0; res[1] = r1; res[2] = r2;
}

// a / b: IEEE division on the exact path, reciprocal-multiply (v_rcp_f32, 1 ulp) on the filter
template <class T>
MJPL_HD T rdiv(T a, T b) {
#if defined(__HIP_DEVICE_COMPILE__)
  if constexpr (!Real<T>::exact) return a * __builtin_amdgcn_rcpf(b);
#endif
  return a / b;
}

template <class T>
MJPL_HD T rsqrt_val(T x) {  // sqrt: correctly rounded on the exact path, v_sqrt_f32 on the filter
#if defined(__HIP_DEVICE_COMPILE__)
  if constexpr (!Real<T>::exact) return __builtin_amdgcn_sqrtf(x);
#endif
  return sqrt(x);
}

template <class T>
MJPL_HD void normalize4(T *v) {
  T norm = rsqrt_val(v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3]);
  if constexpr (Real<T>::exact) {
    bool tiny = norm < MJPL_MINVAL;
    bool scale = fabs(norm - 1) > MJPL_MINVAL;
    T inv = 1 / norm;
    T s0 = v[0] * inv, s1 = v[1] * inv, s2 = v[2] * inv, s3 = v[3] * inv;
    v[0] = tiny ? T(1) : (scale ? s0 : v[0]);
    v[1] = tiny ? T(0) : (scale ? s1 : v[1]);
    v[2] = tiny ? T(0) : (scale ? s2 : v[2]);
    v[3] = tiny ? T(0) : (scale ? s3 : v[3]);
  } else {
    T inv = rdiv(T(1), norm);  // always rescale: the two thresholds are far below the tolerance
    v[0] *= inv; v[1] *= inv; v[2] *= inv; v[3] *= inv;
  }
}

template <class T>
MJPL_HD void quat2mat(T *res, const T *q) {
  if constexpr (Real<T>::exact) {
    const T q00 = q[0] * q[0], q01 = q[0] * q[1], q02 = q[0] * q[2], q03 = q[0] * q[3];
    const T q11 = q[1] * q[1], q12 = q[1] * q[2], q13 = q[1] * q[3];
    const T q22 = q[2] * q[2], q23 = q[2] * q[3], q33 = q[3] * q[3];
    res[0] = q00 + q11 - q22 - q33;
    res[4] = q00 - q11 + q22 - q33;
    res[8] = q00 - q11 - q22 + q33;
    res[1] = 2 * (q12 - q03);
    res[2] = 2 * (q13 + q02);
    res[3] = 2 * (q12 + q03);
    res[5] = 2 * (q23 - q01);
    res[6] = 2 * (q13 - q02);
    res[7] = 2 * (q23 + q01);
  } else {
    MJPL_FILTER_FMA
    res[0] = q[0] * q[0] + q[1] * q[1] - q[2] * q[2] - q[3] * q[3];
    res[4] = q[0] * q[0] - q[1] * q[1] + q[2] * q[2] - q[3] * q[3];
    res[8] = q[0] * q[0] - q[1] * q[1] - q[2] * q[2] + q[3] * q[3];
    res[1] = 2 * (q[1] * q[2] - q[0] * q[3]);
    res[2] = 2 * (q[1] * q[3] + q[0] * q[2]);
    res[3] = 2 * (q[1] * q[2] + q[0] * q[3]);
    res[5] = 2 * (q[2] * q[3] - q[0] * q[1]);
    res[6] = 2 * (q[1] * q[3] - q[0] * q[2]);
    res[7] = 2 * (q[2] * q[3] + q[0] * q[1]);
  }
}

// third column of quat2mat only (capsule axis); same expressions as res[2], res[5], res[8]
template <class T>
MJPL_HD void quat2zaxis(T *m, const T *q) {
  if constexpr (Real<T>::exact) {
    const T q00 = q[0] * q[0], q01 = q[0] * q[1], q02 = q[0] * q[2];
    const T q11 = q[1] * q[1], q13 = q[1] * q[3];
    const T q22 = q[2] * q[2], q23 = q[2] * q[3], q33 = q[3] * q[3];
    m[2] = 2 * (q13 + q02);
    m[5] = 2 * (q23 - q01);
    m[8] = q00 - q11 - q22 + q33;
  } else {
    MJPL_FILTER_FMA
    m[2] = 2 * (q[1] * q[3] + q[0] * q[2]);
    m[5] = 2 * (q[2] * q[3] - q[0] * q[1]);
    m[8] = q[0] * q[0] - q[1] * q[1] - q[2] * q[2] + q[3] * q[3];
  }
}

template <class T>
MJPL_HD T clipd(T x, T lo, T hi) { return x < lo ? lo : (x > hi ? hi : x); }

// classify a surplus s = (distance) - (contact threshold): contact iff s <= 0 on the exact path
template <class T>
MJPL_HD int classify(T s, T tol) {
  if constexpr (Real<T>::exact) return !(s > 0) ? V_CONTACT : V_NONE;
  return s > tol ? V_NONE : (s < -tol ? V_CONTACT : V_UNSURE);
}
// The edge certificate of the fused filter kernel (mjpl_fused.h: an edge whose END configuration keeps every enabled pair
// farther apart than the pair can move along the edge needs no waypoint checks): a narrowphase routine called with an
// accumulator also says whether its pair is clear of contact by `extra` metres MORE than the verdict V_NONE needs -- the
// same quantities, one more compare.  `clear` only ever goes from true to false; a routine that cannot tell leaves it false.
template <class T>
struct CertAcc {
  T extra;
  bool clear;
};
template <class T>
MJPL_HD int classify_c(T s, T tol, CertAcc<T> *ca) {
  if (ca) ca->clear = ca->clear && (s - ca->extra > tol);
  return classify(s, tol);
}
MJPL_HD int v_or(int a, int b) {  // "any contact" over several tests
  return (a == V_CONTACT || b == V_CONTACT) ? V_CONTACT : ((a == V_UNSURE || b == V_UNSURE) ? V_UNSURE : V_NONE);
}

// ----------------------------------------------------------------------------- narrowphase
// Verdict-only ("ncon > 0") primitives.  [MJ-recalled: engine_collision_primitive.c,
// engine_collision_box.c]; capsule-box / box-box follow the oracle's documented deviations.
// A geom is (pos[3], m[9]) with m row-major; capsules use only the z column m[2],m[5],m[8].
// Every routine returns V_NONE / V_CONTACT, plus V_UNSURE on the filter path (T = float) when
// the decision lies inside the tolerance band or the computation is ill-conditioned.

template <class T>
struct GeomT {
  T pos[3];
  T m[9];
};

template <class T>
MJPL_HD int sphere_sphere(T margin, const T *pos1, T r1, const T *pos2, T r2, T tol, CertAcc<T> *ca = nullptr) {
  T dif[3] = {pos1[0] - pos2[0], pos1[1] - pos2[1], pos1[2] - pos2[2]};
  T cdist_sqr = dot3(dif, dif);
  T min_dist = margin + r1 + r2;
  if constexpr (Real<T>::exact) return !(cdist_sqr > min_dist * min_dist) ? V_CONTACT : V_NONE;
  return classify_c(rsqrt_val(cdist_sqr) - min_dist, tol, ca);
}

template <class T>
MJPL_HD int plane_sphere(T margin, const GeomT<T> &pl, const T *pos2, T r2, T tol, CertAcc<T> *ca = nullptr) {
  T n[3] = {pl.m[2], pl.m[5], pl.m[8]};
  T tmp[3] = {pos2[0] - pl.pos[0], pos2[1] - pl.pos[1], pos2[2] - pl.pos[2]};
  T cdist = dot3(tmp, n);
  if constexpr (Real<T>::exact) return !(cdist > margin + r2) ? V_CONTACT : V_NONE;
  return classify_c(cdist - (margin + r2), tol, ca);
}

template <class T>
MJPL_HD int plane_capsule(T margin, const GeomT<T> &pl, const GeomT<T> &cap, const T *size2, T tol, CertAcc<T> *ca = nullptr) {
  T seg[3] = {size2[1] * cap.m[2], size2[1] * cap.m[5], size2[1] * cap.m[8]};
  T e1[3] = {cap.pos[0] + seg[0], cap.pos[1] + seg[1], cap.pos[2] + seg[2]};
  T e2[3] = {cap.pos[0] - seg[0], cap.pos[1] - seg[1], cap.pos[2] - seg[2]};
  return v_or(plane_sphere(margin, pl, e1, size2[0], tol, ca), plane_sphere(margin, pl, e2, size2[0], tol, ca));
}

template <class T>
MJPL_HD int plane_box(T margin, const GeomT<T> &pl, const GeomT<T> &box, const T *size2, T tol, CertAcc<T> *ca = nullptr) {
  T norm[3] = {pl.m[2], pl.m[5], pl.m[8]};
  T dif[3] = {box.pos[0] - pl.pos[0], box.pos[1] - pl.pos[1], box.pos[2] - pl.pos[2]};
  T dist = dot3(dif, norm);
  int res = V_NONE;
  // rolled on purpose: unrolling the 8 corners keeps ~40 extra VGPRs live
#pragma unroll 1
  for (int i = 0; i < 8; i++) {
    T vec[3], corner[3];
    vec[0] = (i & 1) ? size2[0] : -size2[0];
    vec[1] = (i & 2) ? size2[1] : -size2[1];
    vec[2] = (i & 4) ? size2[2] : -size2[2];
    mul_mat_vec3(corner, box.m, vec);
    T ldist = dot3(norm, corner);
    if constexpr (Real<T>::exact) {
      res = (res == V_CONTACT || !(dist + ldist > margin || ldist > 0)) ? V_CONTACT : V_NONE;
    } else {
      // a corner counts iff it is below the centre (ldist <= 0) and within the margin
      if (ca) ca->clear = ca->clear && (dist + ldist - margin - ca->extra > tol);  // (every corner that far above the plane)
      const bool sure_out = (dist + ldist - margin > tol) || (ldist > tol);
      const bool sure_in = (dist + ldist - margin < -tol) && (ldist < -tol);
      res = v_or(res, sure_in ? V_CONTACT : (sure_out ? V_NONE : V_UNSURE));
    }
  }
  return res;
}

template <class T>
MJPL_HD int sphere_capsule(T margin, const T *pos1, T r1, const GeomT<T> &cap, const T *size2, T tol, CertAcc<T> *ca = nullptr) {
  T len = size2[1];
  T axis[3] = {cap.m[2], cap.m[5], cap.m[8]};
  T vec[3] = {pos1[0] - cap.pos[0], pos1[1] - cap.pos[1], pos1[2] - cap.pos[2]};
  T x = clipd(dot3(axis, vec), -len, len);
  vec[0] = axis[0] * x + cap.pos[0];
  vec[1] = axis[1] * x + cap.pos[1];
  vec[2] = axis[2] * x + cap.pos[2];
  return sphere_sphere(margin, pos1, r1, vec, size2[0], tol, ca);
}

template <class T>
__device__ __forceinline__ int capsule_capsule(T margin, const GeomT<T> &c1, const T *size1,
                                               const GeomT<T> &c2, const T *size2, T tol, CertAcc<T> *ca = nullptr) {
  T axis1[3] = {c1.m[2] * size1[1], c1.m[5] * size1[1], c1.m[8] * size1[1]};
  T axis2[3] = {c2.m[2] * size2[1], c2.m[5] * size2[1], c2.m[8] * size2[1]};
  T dif[3] = {c1.pos[0] - c2.pos[0], c1.pos[1] - c2.pos[1], c1.pos[2] - c2.pos[2]};
  T ma = dot3(axis1, axis1);
  T mb = -dot3(axis1, axis2);
  T mc = dot3(axis2, axis2);
  T u = -dot3(axis1, dif);
  T v = dot3(axis2, dif);
  T det = ma * mc - mb * mb;
  T vec1[3], vec2[3];
  int res = V_NONE;

  if constexpr (!Real<T>::exact) {
    // Leave to the exact path: nearly parallel axes (sin^2 < 1e-4: binary32 cannot form det, and
    // the float64 routine switches to its parallel branch at the ABSOLUTE threshold |det| < 1e-15,
    // which tiny capsules reach at any angle -- ma * mc itself is below it for half-lengths under
    // ~2e-4 m), and degenerate capsules.  NaN fails both tests.
    if (!(det > T(1e-4) * ma * mc) || !(det > T(4e-15))) {
      if (ca) ca->clear = false;
      return V_UNSURE;
    }
  }
  const bool general = Real<T>::exact ? (fabs(det) >= T(MJPL_MINVAL)) : true;
  if (general) {
    // same divisions as the scalar routine, sign-selected numerators instead of branches
    T x1 = rdiv(mc * u - mb * v, det);
    T x2 = rdiv(ma * v - mb * u, det);
    bool hi1 = x1 > 1, lo1 = x1 < -1;
    T x2c = rdiv(hi1 ? (v - mb) : (v + mb), mc);
    x1 = hi1 ? T(1) : (lo1 ? T(-1) : x1);
    x2 = (hi1 || lo1) ? x2c : x2;
    bool hi2 = x2 > 1, lo2 = x2 < -1;
    T x1c = clipd(rdiv(hi2 ? (u - mb) : (u + mb), ma), T(-1), T(1));
    x2 = hi2 ? T(1) : (lo2 ? T(-1) : x2);
    x1 = (hi2 || lo2) ? x1c : x1;
#pragma unroll
    for (int k = 0; k < 3; k++) {
      vec1[k] = c1.pos[k] + axis1[k] * x1;
      vec2[k] = c2.pos[k] + axis2[k] * x2;
    }
    if constexpr (Real<T>::exact) {
      res = sphere_sphere(margin, vec1, size1[0], vec2, size2[0], tol);
    } else {
      // The parameters above carry binary32 rounding amplified by 1 / sin^2, so the verdict does
      // not rest on them being the minimiser:
      //  * CONTACT needs only two points ON the segments (x1, x2 in [-1, 1] by construction) that
      //    are closer than the threshold;
      //  * NO CONTACT needs a lower bound of the minimum.  f = |vec1 - vec2|^2 / 2 is convex in
      //    (x1, x2), so f(x*) >= f(x) + grad f . (x* - x) >= f(x) - 2 (|g1|' + |g2|'), where a
      //    component whose parameter sits on its bound and whose gradient points outward
      //    contributes nothing.  g_k is itself evaluated in binary32: |a_k| * tol / 8 covers that
      //    (its inputs are accurate to tol / 2 * 4 eps / B).
      // Whatever is neither goes to the exact path.
      const T dd[3] = {vec1[0] - vec2[0], vec1[1] - vec2[1], vec1[2] - vec2[2]};
      const T D2 = dot3(dd, dd);
      const T D = rsqrt_val(D2);
      const T rs = margin + size1[0] + size2[0];
      const T g1 = dot3(axis1, dd), g2 = -dot3(axis2, dd);
      const T e1 = x1 >= 1 ? fmax(g1, T(0)) : (x1 <= -1 ? fmax(-g1, T(0)) : fabs(g1));
      const T e2 = x2 >= 1 ? fmax(g2, T(0)) : (x2 <= -1 ? fmax(-g2, T(0)) : fabs(g2));
      const T alen = T(1.001) * (size1[1] + size2[1]);  // |a1| + |a2| (the axes are unit vectors)
      const T lo2 = D2 - T(4) * (e1 + e2) - alen * (T(0.5) * tol + T(2e-6) * D) - T(1e-6) * D2;
      if (ca) ca->clear = ca->clear && (lo2 > (rs + ca->extra + tol) * (rs + ca->extra + tol));
      res = (D - rs < -tol) ? V_CONTACT : ((lo2 > (rs + tol) * (rs + tol)) ? V_NONE : V_UNSURE);
    }
  } else {
    // parallel axes (rare, exact path only): any of the four end tests
    T x1, x2;
    for (int k = 0; k < 3; k++) vec1[k] = c1.pos[k] + axis1[k];
    x2 = clipd((v - mb) / mc, T(-1), T(1));
    for (int k = 0; k < 3; k++) vec2[k] = c2.pos[k] + axis2[k] * x2;
    res = sphere_sphere(margin, vec1, size1[0], vec2, size2[0], tol);
    for (int k = 0; k < 3; k++) vec1[k] = c1.pos[k] - axis1[k];
    x2 = clipd((v + mb) / mc, T(-1), T(1));
    for (int k = 0; k < 3; k++) vec2[k] = c2.pos[k] + axis2[k] * x2;
    res = v_or(res, sphere_sphere(margin, vec1, size1[0], vec2, size2[0], tol));
    for (int k = 0; k < 3; k++) vec2[k] = c2.pos[k] + axis2[k];
    x1 = clipd((u - mb) / ma, T(-1), T(1));
    for (int k = 0; k < 3; k++) vec1[k] = c1.pos[k] + axis1[k] * x1;
    res = v_or(res, sphere_sphere(margin, vec1, size1[0], vec2, size2[0], tol));
    for (int k = 0; k < 3; k++) vec2[k] = c2.pos[k] - axis2[k];
    x1 = clipd((u + mb) / ma, T(-1), T(1));
    for (int k = 0; k < 3; k++) vec1[k] = c1.pos[k] + axis1[k] * x1;
    res = v_or(res, sphere_sphere(margin, vec1, size1[0], vec2, size2[0], tol));
  }
  return res;
}

template <class T>
MJPL_HD int sphere_box_local(T margin, const T *c, T r, const T *size2, T tol, CertAcc<T> *ca = nullptr) {
  T d[3];
#pragma unroll
  for (int k = 0; k < 3; k++) d[k] = clipd(c[k], -size2[k], size2[k]) - c[k];
  T dist = rsqrt_val(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
  if constexpr (Real<T>::exact) return !(dist - r > margin) ? V_CONTACT : V_NONE;
  return classify_c(dist - r - margin, tol, ca);
}

template <class T>
MJPL_HD int sphere_box(T margin, const T *pos1, T r1, const GeomT<T> &box, const T *size2, T tol, CertAcc<T> *ca = nullptr) {
  T tmp[3] = {pos1[0] - box.pos[0], pos1[1] - box.pos[1], pos1[2] - box.pos[2]};
  T center[3];
  mul_matT_vec3(center, box.m, tmp);
  return sphere_box_local(margin, center, r1, size2, tol, ca);
}

template <class T>
MJPL_HD T capbox_g(const T *p, const T *h, const T *s, T t) {
  T g = 0;
#pragma unroll
  for (int k = 0; k < 3; k++) {
    T x = p[k] + t * h[k];
    T e = x - fmin(fmax(x, -s[k]), s[k]);
    g = g + h[k] * e;
  }
  return g;
}

// exact 1-D convex minimisation of dist^2(segment point, box); see oracle/mjpl_oracle.c
template <class T>
__device__ __forceinline__ int capsule_box(T margin, const GeomT<T> &cap, const T *size1,
                                           const GeomT<T> &box, const T *size2, T tol, CertAcc<T> *ca = nullptr) {
  T tmp[3] = {cap.pos[0] - box.pos[0], cap.pos[1] - box.pos[1], cap.pos[2] - box.pos[2]};
  T axis[3] = {cap.m[2], cap.m[5], cap.m[8]};
  T p[3], a[3], h[3], inv[3];
  mul_matT_vec3(p, box.m, tmp);
  mul_matT_vec3(a, box.m, axis);
#pragma unroll
  for (int k = 0; k < 3; k++) {
    h[k] = a[k] * size1[1];
    inv[k] = rdiv(T(1), h[k]);  // h == 0: +-inf, the breakpoint becomes +-inf or NaN and is skipped
  }

  T lo = -1, hi = 1;
  const T g_m1 = capbox_g(p, h, size2, lo);
  const T g_p1 = capbox_g(p, h, size2, hi);
  T glo = g_m1, ghi = g_p1;
  // six face breakpoints in the scalar routine's order (k = 0,1,2; minus before plus)
#pragma unroll
  for (int k = 0; k < 3; k++) {
#pragma unroll
    for (int sgn = -1; sgn <= 1; sgn += 2) {
      T tb = (sgn * size2[k] - p[k]) * inv[k];
      bool inside = (tb > lo && tb < hi);
      T gb = capbox_g(p, h, size2, tb);
      bool below = inside && gb <= 0;
      bool above = inside && !(gb <= 0);
      lo = below ? tb : lo;
      glo = below ? gb : glo;
      hi = above ? tb : hi;
      ghi = above ? gb : ghi;
    }
  }
  T den = ghi - glo;
  T t = (den > 0) ? lo + (hi - lo) * rdiv(0 - glo, den) : lo;
  t = (g_m1 >= 0) ? T(-1) : ((g_p1 <= 0) ? T(1) : t);
  T c[3] = {p[0] + t * h[0], p[1] + t * h[1], p[2] + t * h[2]};
  if constexpr (Real<T>::exact) {
    return sphere_box_local(margin, c, size1[0], size2, tol);
  } else {
    // As in capsule_capsule the verdict does not rest on t being the minimiser (the breakpoints
    // divide by components of h that may be tiny): CONTACT needs one point of the segment within
    // the threshold of the box; NO CONTACT needs a lower bound of the minimum, and dist^2 / 2 is
    // convex in t with derivative g(t): dist^2(t*) >= dist^2(t) - 4 |g(t)|', the one-sided part
    // only when t sits on an end of [-1, 1].
    t = clipd(t, T(-1), T(1));
    c[0] = p[0] + t * h[0]; c[1] = p[1] + t * h[1]; c[2] = p[2] + t * h[2];
    T d[3];
#pragma unroll
    for (int k = 0; k < 3; k++) d[k] = clipd(c[k], -size2[k], size2[k]) - c[k];
    const T D2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
    const T D = rsqrt_val(D2);
    const T rs = margin + size1[0];
    const T gt = -(h[0] * d[0] + h[1] * d[1] + h[2] * d[2]);  // g(t) = sum h_k e_k, e = -d
    const T ge = t >= 1 ? fmax(gt, T(0)) : (t <= -1 ? fmax(-gt, T(0)) : fabs(gt));
    const T hl = T(1.001) * size1[1];
    const T lo2 = D2 - T(4) * ge - hl * (T(0.5) * tol + T(2e-6) * D) - T(1e-6) * D2;
    if (ca) ca->clear = ca->clear && (lo2 > (rs + ca->extra + tol) * (rs + ca->extra + tol));
    return (D - rs < -tol) ? V_CONTACT : ((lo2 > (rs + tol) * (rs + tol)) ? V_NONE : V_UNSURE);
  }
}

// 15-axis separating-axis verdict; see oracle/mjpl_oracle.c box_box
template <class T>
__device__ __forceinline__ int box_box(T margin, const GeomT<T> &b1, const T *size1,
                                       const GeomT<T> &b2, const T *size2, T tol) {
  T d[3] = {b2.pos[0] - b1.pos[0], b2.pos[1] - b1.pos[1], b2.pos[2] - b1.pos[2]};
  T R[9], A[9], t[3];
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++)
      R[3 * i + j] = b1.m[i] * b2.m[j] + b1.m[3 + i] * b2.m[3 + j] + b1.m[6 + i] * b2.m[6 + j];
  mul_matT_vec3(t, b1.m, d);
#pragma unroll
  for (int k = 0; k < 9; k++) A[k] = fabs(R[k]);
  // exact path: separated iff some axis has gap > margin.  filter: gaps classified with tol
  bool sep = false, near = false;
#pragma unroll
  for (int i = 0; i < 3; i++) {
    T rb = size2[0] * A[3 * i] + size2[1] * A[3 * i + 1] + size2[2] * A[3 * i + 2];
    T gap = fabs(t[i]) - (size1[i] + rb);
    if constexpr (Real<T>::exact) sep = sep || (gap > margin);
    else { sep = sep || (gap - margin > tol); near = near || !(gap - margin < -tol); }
  }
#pragma unroll
  for (int j = 0; j < 3; j++) {
    T ra = size1[0] * A[j] + size1[1] * A[3 + j] + size1[2] * A[6 + j];
    T tj = t[0] * R[j] + t[1] * R[3 + j] + t[2] * R[6 + j];
    T gap = fabs(tj) - (ra + size2[j]);
    if constexpr (Real<T>::exact) sep = sep || (gap > margin);
    else { sep = sep || (gap - margin > tol); near = near || !(gap - margin < -tol); }
  }
#pragma unroll
  for (int i = 0; i < 3; i++) {
    const int i1 = (i + 1) % 3, i2 = (i + 2) % 3;
#pragma unroll
    for (int j = 0; j < 3; j++) {
      const int j1 = (j + 1) % 3, j2 = (j + 2) % 3;
      T len2 = 1 - R[3 * i + j] * R[3 * i + j];
      T ra = size1[i1] * A[3 * i2 + j] + size1[i2] * A[3 * i1 + j];
      T rb = size2[j1] * A[3 * i + j2] + size2[j2] * A[3 * i + j1];
      T tl = t[i2] * R[3 * i1 + j] - t[i1] * R[3 * i2 + j];
      if constexpr (Real<T>::exact) {
        bool s = !(len2 < T(1e-12)) && (fabs(tl) - (ra + rb) > margin * sqrt(len2));
        sep = sep || s;
      } else {
        // nearly parallel edges: the exact path skips this axis below 1e-12, which binary32
        // cannot resolve -> such a pair can only be certain through another axis
        const bool skew = len2 > T(1e-3);
        T gap = fabs(tl) - (ra + rb) - margin * rsqrt_val(fmax(len2, T(0)));
        sep = sep || (skew && gap > tol);
        near = near || !skew || !(gap < -tol);
      }
    }
  }
  if constexpr (Real<T>::exact) return sep ? V_NONE : V_CONTACT;
  return sep ? V_NONE : (near ? V_UNSURE : V_CONTACT);
}

// Pair dispatch.  mj_collision calls its function table with (g1, g2) ordered by geom type,
// and by geom id when the types are equal; `pfirst` says that the PARTNER is g1.  Mixed-type
// pairs are ordered by type inside each routine's signature, so only the symmetric-type
// routines whose rounding depends on the argument order (sphere-sphere's radius sum,
// capsule-capsule, box-box) look at `pfirst`, with wave-uniform selects.
// WBOX: some static geom is a box.  MBOX: some moving geom is a box.  The flags only remove
// dead narrowphase code (and its registers) from an instantiation.
template <class T, bool WBOX, bool MBOX>
__device__ __forceinline__ int pair_contact(int tcur, const GeomT<T> &cur, const T *scur, int tpar,
                                            const GeomT<T> &par, const T *spar, bool pfirst, T margin,
                                            T tol, CertAcc<T> *ca = nullptr) {
  constexpr bool PBOX = WBOX || MBOX;  // the partner may be a box
  int r = V_NONE;
  if (tpar == GT_PLANE) {
    if (tcur == GT_SPHERE) r = plane_sphere(margin, par, cur.pos, scur[0], tol, ca);
    else if (tcur == GT_CAPSULE) r = plane_capsule(margin, par, cur, scur, tol, ca);
    else if (MBOX) r = plane_box(margin, par, cur, scur, tol, ca);
  } else if (tcur == GT_SPHERE && tpar == GT_SPHERE) {
    const T r1 = pfirst ? spar[0] : scur[0], r2 = pfirst ? scur[0] : spar[0];
    r = sphere_sphere(margin, cur.pos, r1, par.pos, r2, tol, ca);  // (a-b)^2 == (b-a)^2 exactly
  } else if (tcur == GT_SPHERE && tpar == GT_CAPSULE) {
    r = sphere_capsule(margin, cur.pos, scur[0], par, spar, tol, ca);
  } else if (tcur == GT_CAPSULE && tpar == GT_SPHERE) {
    r = sphere_capsule(margin, par.pos, spar[0], cur, scur, tol, ca);
  } else if (PBOX && tcur == GT_SPHERE && tpar == GT_BOX) {
    r = sphere_box(margin, cur.pos, scur[0], par, spar, tol, ca);
  } else if (MBOX && tcur == GT_BOX && tpar == GT_SPHERE) {
    r = sphere_box(margin, par.pos, spar[0], cur, scur, tol, ca);
  } else if (PBOX && tcur == GT_CAPSULE && tpar == GT_BOX) {
    r = capsule_box(margin, cur, scur, par, spar, tol, ca);
  } else if (MBOX && tcur == GT_BOX && tpar == GT_CAPSULE) {
    r = capsule_box(margin, par, spar, cur, scur, tol, ca);
  } else if (tcur == GT_CAPSULE && tpar == GT_CAPSULE) {
    GeomT<T> c1, c2;
    T s1[2], s2[2];
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
    r = capsule_capsule(margin, c1, s1, c2, s2, tol, ca);
  } else if (MBOX) {
    GeomT<T> b1, b2;
    T s1[3], s2[3];
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
    if (ca) ca->clear = false;  // (the separating-axis verdict carries no distance: such a pair never certifies)
    r = box_box(margin, b1, s1, b2, s2, tol);
  }
  return r;
}

// ----------------------------------------------------------------------------- interpreter

struct FkOut {  // global-memory destinations of the FK parity kernel (any may be null)
  double *xpos, *xquat, *geom_xpos, *geom_xmat;
  int nbody, ngeom;
};

// Register slot file: six MAXS-wide vectors (pos x/y/z, z-axis x/y/z) held in VGPRs.  A slot is
// selected with a wave-uniform index, which the compiler turns into indirect VGPR addressing
// (s_set_gpr_idx_on + v_mov): a handful of instructions, against a compare tree for a switch.
template <class T, int MAXS>
struct SlotFile {
  typedef T Vec __attribute__((ext_vector_type(MAXS)));
  Vec f[6];
};

template <class T, int MAXS>
__device__ __forceinline__ void slot_put6(SlotFile<T, MAXS> &sf, int slot, const T *t6) {
#pragma unroll
  for (int k = 0; k < 6; k++) sf.f[k][slot] = t6[k];
}

template <class T, int MAXS>
__device__ __forceinline__ void slot_get6(const SlotFile<T, MAXS> &sf, int slot, T *t6) {
#pragma unroll
  for (int k = 0; k < 6; k++) t6[k] = sf.f[k][slot];
}

// A sphere/capsule occupies one slot (pos, z axis); a box a second one (x and y axes).
// Slot ids are packed as  first | (second << 6), second == SLOT_NONE when unused.
enum : int { SLOT_NONE = 63 };

// Optimisation barrier: makes the compiler treat a per-lane value as redefined here, so that
// expressions of it are not hoisted out of the partner loops (loop-invariant code motion of the
// narrowphase prologues costs tens of VGPRs that stay live across the whole loop).
__device__ __forceinline__ void pin(double &x) { asm volatile("" : "+v"(x)); }
__device__ __forceinline__ void pin(float &x) { asm volatile("" : "+v"(x)); }
__device__ __forceinline__ void pin(int &x) { asm volatile("" : "+v"(x)); }
template <class T>
__device__ __forceinline__ void pin_geom(GeomT<T> &g) {
  pin(g.pos[0]); pin(g.pos[1]); pin(g.pos[2]); pin(g.m[2]); pin(g.m[5]); pin(g.m[8]);
}

// the int stored in the first four bytes of a table scalar
__device__ __forceinline__ int info_bits(float v) { return __builtin_bit_cast(int, v); }
__device__ __forceinline__ int info_bits(double v) { return (int)(unsigned)__builtin_bit_cast(unsigned long long, v); }

// value of `v` in lane `l` (wave-uniform l) as a scalar: v_readlane_b32
__device__ __forceinline__ float bcast(float v, int l) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}
__device__ __forceinline__ double bcast(double v, int l) {
  const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, l);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), l);
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

// Park a 64-bit wave mask in lane LANE (a literal) of the register pair (lo, hi): two
// v_writelane_b32.  This clang has the readlane builtin but not the writelane one, and the hazard
// recogniser does not look inside inline asm: a VALU-written SGPR / VCC (the v_cmp that produced the
// mask) needs two wait states before another VALU instruction reads it as data on gfx94x/gfx950 --
// without them the low half of the mask is sometimes stale -- hence the s_nop.
template <int LANE>
__device__ __forceinline__ void park_mask(int &lo, int &hi, unsigned long long m) {
  asm("s_nop 1\n\tv_writelane_b32 %0, %2, %4\n\tv_writelane_b32 %1, %3, %4"
      : "+v"(lo), "+v"(hi)
      : "s"((int)(unsigned)m), "s"((int)(unsigned)(m >> 32)), "n"(LANE));
}
// two masks behind one wait state
template <int LANE_A, int LANE_B>
__device__ __forceinline__ void park_mask2(int &lo, int &hi, unsigned long long ma, unsigned long long mb) {
  asm("s_nop 1\n\tv_writelane_b32 %0, %2, %6\n\tv_writelane_b32 %1, %3, %6\n\t"
      "v_writelane_b32 %0, %4, %7\n\tv_writelane_b32 %1, %5, %7"
      : "+v"(lo), "+v"(hi)
      : "s"((int)(unsigned)ma), "s"((int)(unsigned)(ma >> 32)), "s"((int)(unsigned)mb), "s"((int)(unsigned)(mb >> 32)),
        "n"(LANE_A), "n"(LANE_B));
}

// ... two masks in lanes `lane` and `lane + 1` chosen at run time (wave-uniform).  v_writelane takes its
// value from the scalar side already, so the lane select cannot be a second SGPR (one constant-bus read
// per instruction on gfx9): it travels in M0, whose old value is put back -- the compiler does not treat
// M0 as clobbered by inline assembly.
__device__ __forceinline__ void park_mask_at2(int &lo, int &hi, unsigned long long ma, unsigned long long mb, int lane) {
  int keep;
  asm("s_mov_b32 %2, m0\n\ts_mov_b32 m0, %7\n\ts_nop 1\n\t"
      "v_writelane_b32 %0, %3, m0\n\tv_writelane_b32 %1, %4, m0\n\t"
      "s_add_u32 m0, m0, 1\n\ts_nop 0\n\t"
      "v_writelane_b32 %0, %5, m0\n\tv_writelane_b32 %1, %6, m0\n\t"
      "s_mov_b32 m0, %2"
      : "+v"(lo), "+v"(hi), "=&s"(keep)
      : "s"((int)(unsigned)ma), "s"((int)(unsigned)(ma >> 32)), "s"((int)(unsigned)mb), "s"((int)(unsigned)(mb >> 32)), "s"(lane)
      : "scc");
}

// control words are wave-uniform: pin them to SGPRs so the interpreter's branches are scalar
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

template <class T>
__device__ __forceinline__ void sincos_half(T x, T *s, T *c) {
  if constexpr (Real<T>::exact) sincos_pi2(x, s, c);
#ifdef MJPL_X_FAST_SINCOS
  else { *s = __sinf(x); *c = __cosf(x); }
#else
  else sincosf(x, s, c);
#endif
}

// Items the float32 filter could not decide, handed to the exact float64 kernels on the device.
// ga >= 0: one geom pair (ga = the later moving geom, gb = its partner) of the configuration;
// ga < 0: the whole configuration.
struct UndecidedConfigs {
  double *q;    // [cap][nplan] configuration (waypoint), row-major
  int *edge;    // [cap] edge it belongs to
  int *idx;     // [cap] its check index inside that edge
  int *ga, *gb; // [cap] model geom ids of the undecided pair, or -1
  int *count;   // entries written (may exceed cap: the overflow went to the edge-level list)
  int cap;
};

// Where a drain reports candidates it cannot decide (count == nullptr: flag the owning lane).
// Where the edges of a launch live, for kernels that rebuild a waypoint instead of reading it.
struct EdgeSource {
  const double *QA, *QB;  // null: not an edge launch
  long long E;
  int layout;
  double step;
  const double *ckpt;     // [item slot][nplan] exact waypoints of the items with idx % kCkptEvery == 0, or null
  int idx0_end;           // != 0: check index 0 of an item is the edge's END, QB (launches that make the endpoint an item, too)
};
constexpr int kCkptEvery = 32;

// Waypoint `idx` (1-based) of edge i, EXACTLY as the reference's recurrence produces it
// (planning/utils.py:182-185, the statements of edge_body): steps from QA, each recomputing
// direction and distance from the previous waypoint.  `out[0..nplan)` doubles as the working row.
// Meant for the few waypoints that go to the exact re-check -- one lane walks alone here, at most
// kCkptEvery - 1 steps when the launch keeps checkpoints (`slot` = the waypoint's item slot; an
// edge's items are consecutive slots).
template <class Perm>
__device__ inline void exact_waypoint(const EdgeSource &src, Perm perm, int nplan, long long i, int idx,
                                      double *out, long long slot = -1) {
  auto at = [&](const double *Q, int k) -> double {
    return (src.layout == MJPL_SOA) ? Q[(long long)k * src.E + i] : Q[i * nplan + k];
  };
  int from = 0;
  if (src.idx0_end && idx == 0) {
    for (int k = 0; k < nplan; k++) out[k] = at(src.QB, k);
    return;
  }
  if (src.ckpt && slot >= 0 && idx >= kCkptEvery) {
    from = idx - idx % kCkptEvery;
    const double *row = src.ckpt + (size_t)(slot - (idx - from)) * nplan;
    for (int k = 0; k < nplan; k++) out[k] = row[k];
  } else {
    for (int k = 0; k < nplan; k++) out[k] = at(src.QA, k);
  }
  for (int n = from; n < idx; n++) {
    double s = 0;
    for (int k = 0; k < nplan; k++) {
      const int col = perm[k];
      const double d = at(src.QB, col) - out[col];
      s = s + d * d;
    }
    const double mag = sqrt(s);
    const double sm = src.step < mag ? src.step : mag;
    for (int k = 0; k < nplan; k++) {
      const double d = at(src.QB, k) - out[k];
      out[k] = out[k] + (d / mag) * sm;
    }
  }
}

struct PatchSink {
  UndecidedConfigs uc;
  const _Float16 *adq = nullptr;  // edge certificate (generated checks): this lane's |dq| per planning column at adq[k * 64]; null: none
  const double *qcol;  // configurations of this wave's lanes: q[k] of lane l at qcol[k * B + l * L]
  int B, L, nplan;     // (LDS columns: B = block size, L = 1)
  int idx;             // check index of the configurations under test (wave-uniform) ...
  const int *item_edge, *item_idx;  // ... or, lane-per-waypoint kernels: (edge, index) of item i
  // lane-per-waypoint kernels test waypoints in closed form (see k_filter_items): what goes to the
  // exact re-check is rebuilt by the recurrence, not copied from qcol
  EdgeSource src;
  IP perm;
};

// A lane hands ONE undecided geom pair of its own configuration to the exact pair kernel
// (immediate interpreter; the queued one does the same from its drains, for the candidate's
// owner).  `item`: the lane's row / item index.  Returns false if there is nowhere to hand it.
__device__ __forceinline__ bool hand_over_own(const PatchSink &ps, int item, int ga, int gb) {
  if (!ps.uc.count) return false;
  const int u = atomicAdd(ps.uc.count, 1);
  if (u >= ps.uc.cap) return false;
  const int lane = threadIdx.x & 63;
  const int ed = ps.item_edge ? ps.item_edge[item] : item;
  const int ix = ps.item_idx ? ps.item_idx[item] : ps.idx;
  if (ps.src.QA)
    exact_waypoint(ps.src, ps.perm, ps.nplan, ed, ix, ps.uc.q + (size_t)u * ps.nplan, ps.item_idx ? item : -1);
  else
    for (int k = 0; k < ps.nplan; k++) ps.uc.q[(size_t)u * ps.nplan + k] = ps.qcol[k * ps.B + lane * ps.L];
  ps.uc.edge[u] = ed;
  ps.uc.idx[u] = ix;
  ps.uc.ga[u] = ga;
  ps.uc.gb[u] = gb;
  return true;
}

// Walk the moving part of the body tree for this lane's configuration.
//   ip, tp : program tables (control words; constants in the instantiation's scalar type)
//   q      : this lane's planning columns, q[c*qstride] (float64; the float32 filter also takes
//            them already rounded to binary32)
//   save   : this lane's LDS pose-save area, save[(slot*7+k)*sstride]
// Returns V_CONTACT iff the configuration has a contact outside the allowed body pairs,
// V_NONE if it has none, and -- filter path only -- V_UNSURE if that cannot be told within
// `tol` (the caller then re-runs the configuration on the exact path).
// `active` = false lanes run along (wave-uniform control flow) but never report anything.
// EMIT: also write body/geom world poses to `out` row `row` (FK parity kernel, exact path).
template <class T, int MAXS, bool EMIT, bool WBOX, bool MBOX, class QT>
__device__ __forceinline__ int run_config(IP ip, typename Real<T>::Tab tp, const QT *q, int qstride,
                                          T *save, int sstride, bool active, T tol, const FkOut &out,
                                          int64_t row, const PatchSink &ps = PatchSink{}) {
  // float32 filter: a pair the narrowphase cannot decide goes to the exact pair kernel by itself
  // (`ps`), and the lane carries on with its other pairs; only what cannot be handed over makes
  // the whole configuration undecided
  auto undecided = [&](bool un, int ga, int gb) -> bool {
    if constexpr (Real<T>::exact) return un;
    else return un && !hand_over_own(ps, (int)row, ga, gb);
  };
  typedef typename Real<T>::Tab Tab;
  typedef GeomT<T> Geom;
  SlotFile<T, MAXS> sf;
  T p[3] = {0, 0, 0}, qt[4] = {1, 0, 0, 0}, R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  bool hit = false, unsure = false;
  const int nbodyops = uni(ip[H_NBODYOPS]);
  Tab wcull = tp + uni(ip[H_OFF_WCULL]);
  Tab wnarrow = tp + uni(ip[H_OFF_WNARROW]);
  const int nwpad = uni(ip[H_NWPAD]);
  int pc = uni(ip[H_OFF_BODYOPS]);
  // filter path: the per-model limits inside which binary32 poses are within tol / 2
  const T maxcoord = Real<T>::exact ? T(0) : tp[uni(ip[H_OFF_FCONST]) + FC_MAXCOORD];
  const T maxangle = Real<T>::exact ? T(0) : tp[uni(ip[H_OFF_FCONST]) + FC_MAXANGLE];

  for (int b = 0; b < nbodyops; b++) {
    // a wave whose every lane is already decided skips the rest of the tree
    if (!EMIT && __ballot(active && !hit && !unsure) == 0ull) break;

    const int parent = uni(ip[pc + B_PARENT]);
    Tab bd = tp + uni(ip[pc + B_DOFF]);
    const int njnt = uni(ip[pc + B_NJNT]);
    const int save_slot = uni(ip[pc + B_SAVE]);
    const int ngeom = uni(ip[pc + B_NGEOM]);
    const int body_id = uni(ip[pc + B_BODYID]);
    pc += B_SIZE;

    T pp[3], pq[4], pR[9];
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
      const T *sv = save + (size_t)(parent - 1) * 7 * sstride;
#pragma unroll
      for (int k = 0; k < 3; k++) pp[k] = sv[k * sstride];
#pragma unroll
      for (int k = 0; k < 4; k++) pq[k] = sv[(3 + k) * sstride];
      quat2mat(pR, pq);  // bit-identical to the matrix built when the pose was saved
    }

    // fixed offset relative to the parent
    T np[3], nq[4];
    {
      T bpos[3] = {bd[0], bd[1], bd[2]};
      T bquat[4] = {bd[3], bd[4], bd[5], bd[6]};
      mul_mat_vec3(np, pR, bpos);
      np[0] += pp[0]; np[1] += pp[1]; np[2] += pp[2];
      mul_quat(nq, pq, bquat);
    }

    // joints
    for (int j = 0; j < njnt; j++) {
      const int jtype = uni(ip[pc + J_TYPE]);
      const int qsrc = uni(ip[pc + J_QSRC]);
      const int jflags = uni(ip[pc + J_FLAGS]);
      Tab jd = tp + uni(ip[pc + J_DOFF]);
      pc += J_SIZE;
      const T qv = (qsrc >= 0) ? (T)q[qsrc * qstride] : jd[7];
      const T dq = qv - jd[6];
      T jaxis[3] = {jd[0], jd[1], jd[2]};
      T jpos[3] = {jd[3], jd[4], jd[5]};
      if (jtype == JT_SLIDE) {
        T xaxis[3];
        rot_vec_quat(xaxis, jaxis, nq);
        np[0] += xaxis[0] * dq; np[1] += xaxis[1] * dq; np[2] += xaxis[2] * dq;
      } else {
        T xanchor[3] = {np[0], np[1], np[2]};
        if (jflags & JF_POS_NONZERO) {
          rot_vec_quat(xanchor, jpos, nq);
          xanchor[0] += np[0]; xanchor[1] += np[1]; xanchor[2] += np[2];
        }
        if constexpr (!Real<T>::exact)  // binary32(q) is off by eps |q|: only modest angles are decided here
          unsure = unsure || (active && !(fabs(dq) <= maxangle));
        T sn, cs;
        sincos_half(dq * T(0.5), &sn, &cs);
        T qloc[4] = {cs, jaxis[0] * sn, jaxis[1] * sn, jaxis[2] * sn};
        mul_quat(nq, nq, qloc);
        if (jflags & JF_POS_NONZERO) {
          T vec[3];
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
    if constexpr (!Real<T>::exact)  // see run_config_queued: too far out for the float32 tolerance
      unsure = unsure || (active && !(fmax(fabs(p[0]), fmax(fabs(p[1]), fabs(p[2]))) <= maxcoord));

    if (save_slot >= 0) {
      T *sv = save + (size_t)save_slot * 7 * sstride;
#pragma unroll
      for (int k = 0; k < 3; k++) sv[k * sstride] = p[k];
#pragma unroll
      for (int k = 0; k < 4; k++) sv[(3 + k) * sstride] = qt[k];
    }
    if constexpr (EMIT) {
      if (active && out.xpos)
        for (int k = 0; k < 3; k++) out.xpos[(row * out.nbody + body_id) * 3 + k] = p[k];
      if (active && out.xquat)
        for (int k = 0; k < 4; k++) out.xquat[(row * out.nbody + body_id) * 4 + k] = qt[k];
    }

    // geoms of this body: world pose, then every enabled pair against static geoms and
    // against earlier moving geoms held in register slots
    for (int g = 0; g < ngeom; g++) {
      const int gtype = uni(ip[pc + G_TYPE]);
      const int gflags = uni(ip[pc + G_FLAGS]);
      Tab gd = tp + uni(ip[pc + G_DOFF]);
      const int store = uni(ip[pc + G_STORE]);
      const int geom_id = uni(ip[pc + G_GEOMID]);
      const unsigned smask = (unsigned)uni(ip[pc + G_SMASK]);
      const unsigned long long wmask_all =
          (unsigned long long)(unsigned)uni(ip[pc + G_WMASK_LO]) |
          ((unsigned long long)(unsigned)uni(ip[pc + G_WMASK_HI]) << 32);
      const unsigned long long pmask_all =
          (unsigned long long)(unsigned)uni(ip[pc + G_PMASK_LO]) |
          ((unsigned long long)(unsigned)uni(ip[pc + G_PMASK_HI]) << 32);
      IP swords = ip + pc + G_SIZE;  // MAX_SLOTS stored-partner words
      pc += G_SIZE + MAX_SLOTS;

      Geom cur;
      const T gsize[3] = {gd[GD_SIZE], gd[GD_SIZE + 1], gd[GD_SIZE + 2]};
      if (gflags & GF_SAMEPOS) {
        cur.pos[0] = p[0]; cur.pos[1] = p[1]; cur.pos[2] = p[2];
      } else {
        T lpos[3] = {gd[0], gd[1], gd[2]};
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
        T lq[4] = {gd[3], gd[4], gd[5], gd[6]}, gq[4];
        mul_quat(gq, qt, lq);
        if (EMIT || (MBOX && gtype == GT_BOX)) quat2mat(cur.m, gq);
        else quat2zaxis(cur.m, gq);
      }
      if constexpr (EMIT) {
        if (active && out.geom_xpos)
          for (int k = 0; k < 3; k++) out.geom_xpos[(row * out.ngeom + geom_id) * 3 + k] = cur.pos[k];
        if (active && out.geom_xmat)
          for (int k = 0; k < 9; k++) out.geom_xmat[(row * out.ngeom + geom_id) * 9 + k] = cur.m[k];
      }

      if (!EMIT) {
        Tab wbound = gd + GD_WBOUND;            // [nwpad] bounds, then [nwpad] margins
        Tab sbound = wbound + 2 * nwpad;        // [16] bounds, [16] margins, [16][3] sizes
        bool live = active && !hit && !unsure;  // lanes that still need an answer

        // ---- static planes (few): signed-distance cull, then the plane routines
        for (unsigned long long pm = pmask_all; pm; pm &= pm - 1) {
          const int wc = (int)__builtin_ctzll(pm);
          Tab rw = wnarrow + wc * WN_LEN;
          Geom par;
          par.pos[0] = wcull[wc_at(wc, 0)]; par.pos[1] = wcull[wc_at(wc, 1)]; par.pos[2] = wcull[wc_at(wc, 2)];
          par.m[2] = rw[WN_ZAXIS]; par.m[5] = rw[WN_ZAXIS + 1]; par.m[8] = rw[WN_ZAXIS + 2];
          par.m[0] = par.m[1] = par.m[3] = par.m[4] = par.m[6] = par.m[7] = 0;
          T dif[3] = {cur.pos[0] - par.pos[0], cur.pos[1] - par.pos[1], cur.pos[2] - par.pos[2]};
          T n[3] = {par.m[2], par.m[5], par.m[8]};
          const bool pass = !(dot3(dif, n) > wbound[wc]) && live;
          if (__builtin_amdgcn_ballot_w64(pass) == 0ull) continue;
          const T psize[3] = {0, 0, 0};
          const int code = pair_contact<T, WBOX, MBOX>(gtype, cur, gsize, GT_PLANE, par, psize, true,
                                                       wbound[nwpad + wc], tol);
          hit = hit || (pass && code == V_CONTACT);
          const bool un = undecided(pass && code == V_UNSURE, geom_id, info_bits(wcull[wc_at(wc, WC_INFO)]) >> 8);
          unsure = unsure || un;
          live = live && !(pass && code == V_CONTACT) && !un;
        }

        // ---- other static partners, four rows of the world cull table at a time.  The four
        // bounding culls (mj_collideSphere: squared centre distance; (a-b)^2 == (b-a)^2 exactly,
        // so the pair order does not matter; the filter's bounds are widened by tol) are
        // unrolled with their operands in SGPRs; the narrowphase is a rolled loop over the
        // rows some lane still needs, so there is one copy of it in the instruction stream.
#ifdef MJPL_X_SKIP_WORLD
        const unsigned long long wmask = 0;
#else
        const unsigned long long wmask = wmask_all;
#endif
        for (int base = 0; base < nwpad; base += 4) {
          const unsigned bits = (unsigned)(wmask >> base) & 15u;
          if (bits == 0) continue;
          Tab rc = wcull + base * WC_LEN;
          Tab bc = wbound + base;
          pin_geom(cur);
          bool ps0, ps1, ps2, ps3;
#define MJPL_WCULL(k, out)                                                              \
          {                                                                             \
            T dx = cur.pos[0] - rc[k], dy = cur.pos[1] - rc[4 + (k)],                   \
              dz = cur.pos[2] - rc[8 + (k)];                                            \
            out = !(dx * dx + dy * dy + dz * dz > bc[k]) && live && ((bits >> (k)) & 1u); \
          }
          MJPL_WCULL(0, ps0) MJPL_WCULL(1, ps1) MJPL_WCULL(2, ps2) MJPL_WCULL(3, ps3)
#undef MJPL_WCULL
          if (__builtin_amdgcn_ballot_w64(ps0 || ps1 || ps2 || ps3) == 0ull) continue;
#ifdef MJPL_X_SKIP_NARROW
          hit = hit || ((ps0 || ps1 || ps2 || ps3) && cur.pos[0] == T(12345.0));
          continue;
#endif
#pragma unroll 1
          for (int k = 0; k < 4; k++) {
            const bool pk = (k == 0 ? ps0 : (k == 1 ? ps1 : (k == 2 ? ps2 : ps3))) && live;
            if (__builtin_amdgcn_ballot_w64(pk) == 0ull) continue;
            const int wc = base + k;
            Tab rw = wnarrow + wc * WN_LEN;
            const int info = info_bits(wcull[wc_at(wc, WC_INFO)]);
            const int ptype = info & 255;
            __builtin_assume(ptype != GT_PLANE);
            Geom par;
            par.pos[0] = wcull[wc_at(wc, 0)]; par.pos[1] = wcull[wc_at(wc, 1)]; par.pos[2] = wcull[wc_at(wc, 2)];
            par.m[2] = rw[WN_ZAXIS]; par.m[5] = rw[WN_ZAXIS + 1]; par.m[8] = rw[WN_ZAXIS + 2];
            if (WBOX) {
              par.m[0] = rw[WN_XAXIS]; par.m[3] = rw[WN_XAXIS + 1]; par.m[6] = rw[WN_XAXIS + 2];
              par.m[1] = rw[WN_YAXIS]; par.m[4] = rw[WN_YAXIS + 1]; par.m[7] = rw[WN_YAXIS + 2];
            } else {
              par.m[0] = par.m[1] = par.m[3] = par.m[4] = par.m[6] = par.m[7] = 0;
            }
            const T psize[3] = {rw[WN_SIZE], rw[WN_SIZE + 1], rw[WN_SIZE + 2]};
            // mj_collision order: smaller geom type first, geom id breaks ties
            const int pgid = info >> 8;
            const bool pfirst = (ptype < gtype) || (ptype == gtype && pgid < geom_id);
            const int code = pair_contact<T, WBOX, MBOX>(gtype, cur, gsize, ptype, par, psize, pfirst,
                                                         wbound[nwpad + wc], tol);
            hit = hit || (pk && code == V_CONTACT);
            const bool un = undecided(pk && code == V_UNSURE, geom_id, pgid);
            unsure = unsure || un;
            live = live && !(pk && code == V_CONTACT) && !un;
          }
        }

        // ---- earlier moving partners, held in the register slot file: same scheme, four
        // slots at a time, the slot registers addressed by literal index
#ifdef MJPL_X_SKIP_STORED
        const unsigned smask_use = 0;
#else
        const unsigned smask_use = smask;
#endif
        // culls are expanded per slot (literal register index); each lane records the slots
        // it passed in `lanebits`, the wave records them in `anybits`
        unsigned lanebits = 0, anybits = 0;
        if (smask_use != 0) {
          pin_geom(cur);
#pragma unroll
          for (int n = 0; n < MAXS; n++) {  // literal register operands after unrolling
            if ((smask_use >> n) & 1u) {
              T dx = cur.pos[0] - sf.f[0][n], dy = cur.pos[1] - sf.f[1][n], dz = cur.pos[2] - sf.f[2][n];
              const bool ps = !(dx * dx + dy * dy + dz * dz > sbound[n]) && live;
              lanebits |= ps ? (1u << n) : 0u;
              anybits |= (__builtin_amdgcn_ballot_w64(ps) != 0ull) ? (1u << n) : 0u;
            }
          }
        }
#ifdef MJPL_X_SKIP_NARROW
        hit = hit || (lanebits != 0 && cur.pos[0] == T(12345.0));
        anybits = 0;
#endif
        for (unsigned ab = anybits; ab; ab &= ab - 1) {  // rolled: one copy of the narrowphase
          const int slot = (int)__builtin_ctz(ab);
          const bool pk = ((lanebits >> slot) & 1u) && live;
          if (__builtin_amdgcn_ballot_w64(pk) == 0ull) continue;
          const int pw = uni(swords[slot]);
          const int ptype = (pw >> 12) & 15;
          const bool pfirst = (pw & P_FIRST) != 0;
          Geom par;
          {
            T t6[6];
            slot_get6(sf, slot, t6);
            par.pos[0] = t6[0]; par.pos[1] = t6[1]; par.pos[2] = t6[2];
            par.m[2] = t6[3]; par.m[5] = t6[4]; par.m[8] = t6[5];
          }
          {
            T t6[6] = {0, 0, 0, 0, 0, 0};
            if (MBOX && (pw & 63) != SLOT_NONE) slot_get6(sf, pw & 63, t6);  // stored box: x and y axes
            par.m[0] = t6[0]; par.m[3] = t6[1]; par.m[6] = t6[2];
            par.m[1] = t6[3]; par.m[4] = t6[4]; par.m[7] = t6[5];
          }
          Tab sz = sbound + 2 * MAX_SLOTS + 3 * slot;
          const T psize[3] = {sz[0], sz[1], sz[2]};
          const int code = pair_contact<T, WBOX, MBOX>(gtype, cur, gsize, ptype, par, psize, pfirst,
                                                       sbound[MAX_SLOTS + slot], tol);
          hit = hit || (pk && code == V_CONTACT);
          const bool un = undecided(pk && code == V_UNSURE, geom_id, (int)sbound[GS_GEOMID + slot]);
          unsure = unsure || un;
          live = live && !(pk && code == V_CONTACT) && !un;
        }
      }

      if (!EMIT && store >= 0) {
        {
          const T t6[6] = {cur.pos[0], cur.pos[1], cur.pos[2], cur.m[2], cur.m[5], cur.m[8]};
          slot_put6(sf, store & 63, t6);
        }
        if (MBOX && ((store >> 6) & 63) != SLOT_NONE) {
          const T t6[6] = {cur.m[0], cur.m[3], cur.m[6], cur.m[1], cur.m[4], cur.m[7]};
          slot_put6(sf, (store >> 6) & 63, t6);
        }
      }
    }
  }
  return hit ? V_CONTACT : (unsure ? V_UNSURE : V_NONE);
}


// ----------------------------------------------------------------------------- diagnostics
// -DMJPL_STAMPS: per-wave s_memtime accounting of the queued interpreter's phases, summed into
// g_stamps by lane 0 (a timing-only build; see tools/stamps.py).  Never defined in the product.
#ifdef MJPL_STAMPS
__device__ unsigned long long g_stamps[16];
__device__ __forceinline__ unsigned long long stamp() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}
#define MJPL_T0(var) unsigned long long var = stamp()
#define MJPL_ACC(slot, var) { unsigned long long t_ = stamp(); acc[slot] += t_ - var; var = t_; }
#else
#define MJPL_T0(var)
#define MJPL_ACC(slot, var)
#endif

// ----------------------------------------------------------------------------- queued narrowphase
// The immediate interpreter above runs a narrowphase routine as soon as ANY lane of the wave
// passes a bounding cull: with 64 unrelated configurations per wave that is the case for ~28 %
// of the 213 Franka pairs although only ~2 % of the (lane, pair) tests pass, so ~95 % of the
// lanes idle through every narrowphase call.  The queued interpreter separates the two: culls
// run lane-per-configuration; every (lane, pair) that passes is PUSHED into a per-wave LDS
// queue (both geoms' poses + what identifies the pair); when 64 candidates are waiting (and
// at the end of the configuration) the wave DRAINS the queue with one candidate per lane, so
// the narrowphase runs with full lanes, and reports contacts back to the owning lanes through
// a per-wave flag word.  Arithmetic per pair is unchanged, only which lane executes it.

// Two queues per wave, so that a drain runs ONE family of narrowphase routines with full lanes:
//   N: slot partners (pose travels in the record), static spheres / capsules and planes
//   B: static boxes (sphere-box / capsule-box, the expensive routines)
// record: cur pos/axis, (N only: slot partner pos/axis).  A push that would not fit first drains
// one batch of up to 64 off the top, so mid-configuration drains run with (nearly) full lanes.
enum : int { QN_CAP = 64, QN_FIELDS = 12, QB_CAP = 64, QB_FIELDS = 6 };
// Models with MOVING boxes (MBOX builds): every candidate with a box on either side goes to the
// B queue, whose records then carry full frames -- [0..5] cur pos, z axis; [6..11] cur x, y axes;
// [12..17] slot partner pos, z axis; [18..23] slot partner x, y axes -- so the N drains keep to the
// cheap sphere / capsule / plane routines.
enum : int { QB_FIELDS_MBOX = 24 };
// i0: bits 0..5 owner lane, 6..9 cur type, 10..13 partner type, 14 pfirst, 15..16 kind, 17..24 index
// i1: constant-table offset of the cur geom's block (sizes; slot sizes)

template <class T, bool MBOX = false>
struct WaveQueue {
  static constexpr int kBFields = MBOX ? QB_FIELDS_MBOX : QB_FIELDS;
  T *nf;          // [QN_FIELDS][QN_CAP]
  int *ni0, *ni1; // [QN_CAP] each: packed pair id, constant-table offset of the cur geom's block
  T *bf;          // [kBFields][QB_CAP]
  int *bi0, *bi1; // [QB_CAP]
  int *flags;     // [64] : bit0 contact, bit1 unsure, bit2 a candidate closer than its certificate margin; bits 3..: the owner's item
  static __host__ __device__ constexpr size_t bytes() {
    return (size_t)QN_FIELDS * QN_CAP * sizeof(T) + 2 * QN_CAP * sizeof(int) +
           (size_t)kBFields * QB_CAP * sizeof(T) + 2 * QB_CAP * sizeof(int) + 64 * sizeof(int);
  }
  __device__ __forceinline__ void carve(char *base) {
    nf = reinterpret_cast<T *>(base);
    ni0 = reinterpret_cast<int *>(nf + QN_FIELDS * QN_CAP);
    ni1 = ni0 + QN_CAP;
    bf = reinterpret_cast<T *>(ni1 + QN_CAP);
    bi0 = reinterpret_cast<int *>(bf + kBFields * QB_CAP);
    bi1 = bi0 + QB_CAP;
    flags = bi1 + QB_CAP;
  }
};

// BOXQ selects the queue.  ALL: empty it (end of a configuration); otherwise one batch.
template <class T, bool BOXQ, bool ALL, bool MBOX = false>
__device__ __forceinline__ void queue_drain(const WaveQueue<T, MBOX> &wq, int &qn, const T *tp, const T *wcull,
                                            const T *wnarrow, int nwpad, T tol, const PatchSink &ps,
                                            unsigned long long *dacc = nullptr) {
  // tp / wcull / wnarrow point into the workgroup's LDS copy of the constant table: the drain
  // reads them with per-lane addresses (every lane has its own candidate), which from global
  // memory costs a ~1-2 us dependent gather per drain.
  typedef const T *Tab;
  typedef GeomT<T> Geom;
  constexpr int CAP = BOXQ ? QB_CAP : QN_CAP;
#ifdef MJPL_X_NODRAIN  // timing-only build: the item kernel's candidates are queued and thrown away
  if (ps.item_idx) {
    qn = 0;
    return;
  }
#endif
  const T *qf = BOXQ ? wq.bf : wq.nf;
  const int *qi0 = BOXQ ? wq.bi0 : wq.ni0, *qi1 = BOXQ ? wq.bi1 : wq.ni1;
  const int lane = threadIdx.x & 63;
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  do {  // wave-uniform
    const int n = qn < 64 ? qn : 64;
#ifdef MJPL_STAMPS
    unsigned long long ts0 = stamp();
#endif
    const int j = qn - n + lane;
    const bool on = lane < n;
    qn -= n;
    const int jj = on ? j : 0;
    const int i0 = qi0[jj];
    const int i1 = qi1[jj];
    Tab gd = tp + (i1 & 0xffff);
    // the certificate margin of the candidate (metres; 0: none asked for) -- float32 filter builds only
    CertAcc<T> cacc = {T(0), true};
    CertAcc<T> *ca = nullptr;
    if constexpr (!Real<T>::exact) {
      cacc.extra = (T)__builtin_bit_cast(_Float16, (unsigned short)((unsigned)i1 >> 16));
      ca = &cacc;
    }
    const int owner = i0 & 63, gtype = (i0 >> 6) & 15, ptype = (i0 >> 10) & 15;
    const bool pfirst = (i0 >> 14) & 1;
    const int kind = (i0 >> 15) & 3, index = (i0 >> 17) & 255;
    Geom cur, par;
    cur.pos[0] = qf[0 * CAP + jj]; cur.pos[1] = qf[1 * CAP + jj]; cur.pos[2] = qf[2 * CAP + jj];
    cur.m[2] = qf[3 * CAP + jj]; cur.m[5] = qf[4 * CAP + jj]; cur.m[8] = qf[5 * CAP + jj];
    cur.m[0] = cur.m[1] = cur.m[3] = cur.m[4] = cur.m[6] = cur.m[7] = 0;
    par.m[0] = par.m[1] = par.m[3] = par.m[4] = par.m[6] = par.m[7] = 0;
    // Per-lane gathers: the cur geom's size, and for a static partner its whole pose and size
    // from the world tables (a slot partner's pose travelled in the record).
    const T gsize[3] = {gd[GD_SIZE], gd[GD_SIZE + 1], gd[GD_SIZE + 2]};
    T psize[3], margin;
    int code = V_NONE;
    if constexpr (BOXQ && MBOX) {
      // a box on either side: full frames (see QB_FIELDS_MBOX); static partners from the tables
      cur.m[0] = qf[6 * CAP + jj]; cur.m[3] = qf[7 * CAP + jj]; cur.m[6] = qf[8 * CAP + jj];
      cur.m[1] = qf[9 * CAP + jj]; cur.m[4] = qf[10 * CAP + jj]; cur.m[7] = qf[11 * CAP + jj];
      if (kind == EK_SLOT) {
        par.pos[0] = qf[12 * CAP + jj]; par.pos[1] = qf[13 * CAP + jj]; par.pos[2] = qf[14 * CAP + jj];
        par.m[2] = qf[15 * CAP + jj]; par.m[5] = qf[16 * CAP + jj]; par.m[8] = qf[17 * CAP + jj];
        par.m[0] = qf[18 * CAP + jj]; par.m[3] = qf[19 * CAP + jj]; par.m[6] = qf[20 * CAP + jj];
        par.m[1] = qf[21 * CAP + jj]; par.m[4] = qf[22 * CAP + jj]; par.m[7] = qf[23 * CAP + jj];
        Tab sb = gd + GD_WBOUND + 2 * nwpad;
        margin = sb[MAX_SLOTS + index];
        psize[0] = sb[2 * MAX_SLOTS + 3 * index]; psize[1] = sb[2 * MAX_SLOTS + 3 * index + 1];
        psize[2] = sb[2 * MAX_SLOTS + 3 * index + 2];
      } else {  // EK_STATIC (any type), EK_PLANE
        Tab rw = wnarrow + index * WN_LEN;
        par.pos[0] = wcull[wc_at(index, 0)]; par.pos[1] = wcull[wc_at(index, 1)]; par.pos[2] = wcull[wc_at(index, 2)];
        par.m[2] = rw[WN_ZAXIS]; par.m[5] = rw[WN_ZAXIS + 1]; par.m[8] = rw[WN_ZAXIS + 2];
        par.m[0] = rw[WN_XAXIS]; par.m[3] = rw[WN_XAXIS + 1]; par.m[6] = rw[WN_XAXIS + 2];
        par.m[1] = rw[WN_YAXIS]; par.m[4] = rw[WN_YAXIS + 1]; par.m[7] = rw[WN_YAXIS + 2];
        const bool plane = kind == EK_PLANE;  // (the interpreter passes no size for a plane)
        psize[0] = plane ? T(0) : rw[WN_SIZE]; psize[1] = plane ? T(0) : rw[WN_SIZE + 1];
        psize[2] = plane ? T(0) : rw[WN_SIZE + 2];
        margin = gd[GD_WBOUND + nwpad + index];
      }
      if (on) code = pair_contact<T, true, true>(gtype, cur, gsize, ptype, par, psize, pfirst, margin, tol, ca);
    } else if constexpr (BOXQ) {
      Tab rw = wnarrow + index * WN_LEN;
      par.pos[0] = wcull[wc_at(index, 0)]; par.pos[1] = wcull[wc_at(index, 1)]; par.pos[2] = wcull[wc_at(index, 2)];
      par.m[2] = rw[WN_ZAXIS]; par.m[5] = rw[WN_ZAXIS + 1]; par.m[8] = rw[WN_ZAXIS + 2];
      par.m[0] = rw[WN_XAXIS]; par.m[3] = rw[WN_XAXIS + 1]; par.m[6] = rw[WN_XAXIS + 2];
      par.m[1] = rw[WN_YAXIS]; par.m[4] = rw[WN_YAXIS + 1]; par.m[7] = rw[WN_YAXIS + 2];
      psize[0] = rw[WN_SIZE]; psize[1] = rw[WN_SIZE + 1]; psize[2] = rw[WN_SIZE + 2];
      margin = gd[GD_WBOUND + nwpad + index];
#ifdef MJPL_STAMPS
      pin(margin); pin(psize[0]); pin(par.m[0]);
      unsigned long long ts1 = stamp();
#endif
#ifndef MJPL_X_DRAIN_NONARROW
      if (on) code = pair_contact<T, true, false>(gtype, cur, gsize, GT_BOX, par, psize, pfirst, margin, tol, ca);
#endif
#ifdef MJPL_STAMPS
      pin(code);
      unsigned long long ts2 = stamp();
      dacc[4] += 1ull; dacc[5] += (unsigned long long)n; dacc[6] += ts1 - ts0; dacc[7] += ts2 - ts1;
#endif
    } else {
      if (kind == EK_SLOT) {
        par.pos[0] = qf[6 * CAP + jj]; par.pos[1] = qf[7 * CAP + jj]; par.pos[2] = qf[8 * CAP + jj];
        par.m[2] = qf[9 * CAP + jj]; par.m[5] = qf[10 * CAP + jj]; par.m[8] = qf[11 * CAP + jj];
        Tab sb = gd + GD_WBOUND + 2 * nwpad;
        margin = sb[MAX_SLOTS + index];
        psize[0] = sb[2 * MAX_SLOTS + 3 * index]; psize[1] = sb[2 * MAX_SLOTS + 3 * index + 1];
        psize[2] = sb[2 * MAX_SLOTS + 3 * index + 2];
      } else {
        Tab rw = wnarrow + index * WN_LEN;
        par.pos[0] = wcull[wc_at(index, 0)]; par.pos[1] = wcull[wc_at(index, 1)]; par.pos[2] = wcull[wc_at(index, 2)];
        par.m[2] = rw[WN_ZAXIS]; par.m[5] = rw[WN_ZAXIS + 1]; par.m[8] = rw[WN_ZAXIS + 2];
        psize[0] = rw[WN_SIZE]; psize[1] = rw[WN_SIZE + 1]; psize[2] = rw[WN_SIZE + 2];
        margin = gd[GD_WBOUND + nwpad + index];
      }
#ifdef MJPL_STAMPS
      pin(margin); pin(psize[0]); pin(par.m[2]);
      unsigned long long ts1 = stamp();
#endif
#ifndef MJPL_X_DRAIN_NONARROW
      if (on) code = pair_contact<T, false, false>(gtype, cur, gsize, ptype, par, psize, pfirst, margin, tol, ca);
#endif
#ifdef MJPL_STAMPS
      pin(code);
      unsigned long long ts2 = stamp();
      dacc[0] += 1ull; dacc[1] += (unsigned long long)n; dacc[2] += ts1 - ts0; dacc[3] += ts2 - ts1;
#endif
    }
#ifdef MJPL_X_DRAIN_NONARROW  // timing-only build: pop + gather, no narrowphase
    code = (cur.pos[0] + par.pos[0] + psize[0] + gsize[0] + margin == T(12345.0)) ? V_CONTACT : V_NONE;
#endif
    if (on && code == V_CONTACT) atomicOr(&wq.flags[owner], 1);
    if constexpr (!Real<T>::exact) {  // (a candidate that is not clear by its margin: its owner's edge is not certified)
      if (on && cacc.extra > T(0) && !(code == V_NONE && cacc.clear)) atomicOr(&wq.flags[owner], 4);
    }
    // (an owner known to be in contact -- from this batch, the line above, or an earlier one -- needs no exact check
    //  of another pair: on uniformly random configurations half of the hand-offs were of this kind)
    if (on && code == V_UNSURE && !(wq.flags[owner] & 1)) {
      // hand this one pair of the owner's configuration to the exact kernel; the owner walks on
      // as if it were free (the patch pass clears valid / lowers first_bad if it is not)
      bool handed = false;
      if (ps.uc.count) {
        const int u = atomicAdd(ps.uc.count, 1);
        if (u < ps.uc.cap) {
          const int item = (int)((unsigned)wq.flags[owner] >> 3);
          const int ed = ps.item_edge ? ps.item_edge[item] : item;
          const int ix = ps.item_idx ? ps.item_idx[item] : ps.idx;
          if (ps.src.QA)  // row `ed` of the caller's configurations (ix = 0), or waypoint ix of edge `ed`
            exact_waypoint(ps.src, ps.perm, ps.nplan, ed, ix, ps.uc.q + (size_t)u * ps.nplan, ps.item_idx ? item : -1);
          else
            for (int k = 0; k < ps.nplan; k++) ps.uc.q[(size_t)u * ps.nplan + k] = ps.qcol[k * ps.B + owner * ps.L];
          ps.uc.edge[u] = ed;
          ps.uc.idx[u] = ix;
          ps.uc.ga[u] = (int)gd[GD_GEOMID];
          int gb;
          if ((!BOXQ || MBOX) && kind == EK_SLOT) gb = (int)gd[GD_WBOUND + 2 * nwpad + GS_GEOMID + index];
          else gb = info_bits(wcull[wc_at(index, WC_INFO)]) >> 8;
          ps.uc.gb[u] = gb;
          handed = true;
        }
      }
      if (!handed) atomicOr(&wq.flags[owner], 2);
    }
  } while (ALL && qn > 0);
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
}

// Append the lanes of `pm` (each with its own cur pose cur6 = pos + z axis; a slot partner's pose in
// t6) to a wave's candidate queue, first draining one batch if they would not fit.  The interpreter
// below carries this as a lambda; generated per-model code (mjpl_amd/specialise.py) calls it.
// MBOX builds (models with moving boxes): a candidate of the box queue carries full frames -- cur6b = the
// moving geom's x and y axes, t6b = a slot partner's (zeros where that geom is no box).
template <class T, bool BOXQ, bool MBOX = false>
__device__ __forceinline__ void queue_push(const WaveQueue<T, MBOX> &wq, int &fill, T &dead, int &fl, bool active, bool far,
                                           const T *ltab, const T *lwcull, const T *lwnarrow, int nwpad, T tol,
                                           const PatchSink &ps, unsigned long long pm, int kind, int index, int gtype,
                                           int ptype, bool pfirst, int gdoff, const T *cur6, const T *t6,
                                           const T *cur6b = nullptr, const T *t6b = nullptr, T mcert = T(0)) {
  constexpr int CAP = BOXQ ? QB_CAP : QN_CAP;
  const int lane = threadIdx.x & 63;
#ifdef MJPL_X_NOPUSH  // timing-only build: the item kernel culls, but queues nothing (the masks stay live through fl)
  if (ps.item_idx) {
    fl ^= (int)(pm >> (lane & 31)) & 4;
    return;
  }
#endif
  const int cnt = (int)__builtin_popcountll(pm);
  if (fill + cnt > CAP) {  // make room: one batch leaves the top of the queue
    queue_drain<T, BOXQ, false, MBOX>(wq, fill, ltab, lwcull, lwnarrow, nwpad, tol, ps);
    fl = wq.flags[lane] & 3;
    dead = (fl != 0 || !active || far) ? T(__builtin_inff()) : T(0);
  }
  T *qf = BOXQ ? wq.bf : wq.nf;
  int *qi0 = BOXQ ? wq.bi0 : wq.ni0, *qi1 = BOXQ ? wq.bi1 : wq.ni1;
  const bool mine = (pm >> lane) & 1ull;
  const int off = fill + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(pm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)pm, 0u));
  if (mine) {
#pragma unroll
    for (int k = 0; k < 6; k++) qf[k * CAP + off] = cur6[k];
    if (!BOXQ && kind == EK_SLOT) {  // a static partner's pose is read from the world tables at the drain
#pragma unroll
      for (int k = 0; k < 6; k++) qf[(6 + k) * CAP + off] = t6[k];
    }
    if constexpr (BOXQ && MBOX) {  // (the record layout of run_config_queued's own push)
#pragma unroll
      for (int k = 0; k < 6; k++) qf[(6 + k) * CAP + off] = cur6b ? cur6b[k] : T(0);
      if (kind == EK_SLOT) {
#pragma unroll
        for (int k = 0; k < 6; k++) qf[(12 + k) * CAP + off] = t6[k];
#pragma unroll
        for (int k = 0; k < 6; k++) qf[(18 + k) * CAP + off] = t6b ? t6b[k] : T(0);
      }
    }
    qi0[off] = lane | (gtype << 6) | (ptype << 10) | ((pfirst ? 1 : 0) << 14) | (kind << 15) | (index << 17);
    // the candidate's certificate margin (0: none) as a binary16 in the high half, rounded UP: x (1 + 2^-9) rounds to no less than x
    const unsigned short mh = __builtin_bit_cast(unsigned short, (_Float16)((float)mcert * 1.001953125f));
    qi1[off] = gdoff | ((int)mh << 16);
  }
  fill += cnt;
}

// Queued version of run_config for models without moving boxes (slots hold pos + z axis).
template <class T, int MAXS, bool WBOX, bool MBOX, class QT>
__device__ __forceinline__ int run_config_queued(IP ip, typename Real<T>::Tab tp, const T *ltab,
                                                 const QT *q, int qstride, T *save, int sstride,
                                                 bool active, T tol, const WaveQueue<T, MBOX> &wq, int item,
                                                 const PatchSink &ps) {
  typedef typename Real<T>::Tab Tab;
  typedef GeomT<T> Geom;
  SlotFile<T, MAXS> sf;
  T p[3] = {0, 0, 0}, qt[4] = {1, 0, 0, 0}, R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  const int lane = threadIdx.x & 63;
  // Lane state lives in VGPR values, not in bools: loop-carried lane masks cost three scalar
  // instructions per mask per iteration.  `dead` is +inf for lanes that need no more tests; it is
  // added to the cull measure so that one compare yields the pass mask.  `fl` mirrors flags[lane].
#ifdef MJPL_STAMPS
  unsigned long long acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
  MJPL_T0(tt);
  const T kInf = __builtin_inff();
  T dead = active ? T(0) : kInf;
  bool far = false;
  int fl = 0;
  int qn = 0, qb = 0;  // wave-uniform queue fills (general / static boxes)
#ifdef MJPL_X_Q_NOPUSH
  unsigned long long sink = 0;
#endif
  wq.flags[lane] = (int)((unsigned)item << 3);  // bits 0..2 flags, the rest: whose item this is
  const int nbodyops = uni(ip[H_NBODYOPS]);
  Tab wcull = tp + uni(ip[H_OFF_WCULL]);
  const int nwpad = uni(ip[H_NWPAD]);
  const T *lwcull = ltab + uni(ip[H_OFF_WCULL]);
  const T *lwnarrow = ltab + uni(ip[H_OFF_WNARROW]);
  int pc = uni(ip[H_OFF_BODYOPS]);
  // the per-model limits inside which binary32 poses are within tol / 2 of the binary64 ones (T = double: the EXACT
  // check through the same queues -- k_edges_fused_f64 -- has no such limits, no tolerance and never an UNSURE)
  const T maxcoord = Real<T>::exact ? T(0) : tp[uni(ip[H_OFF_FCONST]) + FC_MAXCOORD];
  const T maxangle = Real<T>::exact ? T(0) : tp[uni(ip[H_OFF_FCONST]) + FC_MAXANGLE];

  for (int b = 0; b < nbodyops; b++) {
    if (__builtin_amdgcn_ballot_w64(dead == T(0)) == 0ull && qn == 0 && qb == 0) break;  // every lane decided

    const int parent = uni(ip[pc + B_PARENT]);
    Tab bd = tp + uni(ip[pc + B_DOFF]);
    const int njnt = uni(ip[pc + B_NJNT]);
    const int save_slot = uni(ip[pc + B_SAVE]);
    const int ngeom = uni(ip[pc + B_NGEOM]);
    pc += B_SIZE;

    T pp[3], pq[4], pR[9];
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
      const T *sv = save + (size_t)(parent - 1) * 7 * sstride;
#pragma unroll
      for (int k = 0; k < 3; k++) pp[k] = sv[k * sstride];
#pragma unroll
      for (int k = 0; k < 4; k++) pq[k] = sv[(3 + k) * sstride];
      quat2mat(pR, pq);
    }
    T np[3], nq[4];
    {
      T bpos[3] = {bd[0], bd[1], bd[2]};
      T bquat[4] = {bd[3], bd[4], bd[5], bd[6]};
      mul_mat_vec3(np, pR, bpos);
      np[0] += pp[0]; np[1] += pp[1]; np[2] += pp[2];
      mul_quat(nq, pq, bquat);
    }
    for (int j = 0; j < njnt; j++) {
      const int jtype = uni(ip[pc + J_TYPE]);
      const int qsrc = uni(ip[pc + J_QSRC]);
      const int jflags = uni(ip[pc + J_FLAGS]);
      Tab jd = tp + uni(ip[pc + J_DOFF]);
      pc += J_SIZE;
      const T qv = (qsrc >= 0) ? (T)q[qsrc * qstride] : jd[7];
      const T dq = qv - jd[6];
      T jaxis[3] = {jd[0], jd[1], jd[2]};
      T jpos[3] = {jd[3], jd[4], jd[5]};
      if (jtype == JT_SLIDE) {
        T xaxis[3];
        rot_vec_quat(xaxis, jaxis, nq);
        np[0] += xaxis[0] * dq; np[1] += xaxis[1] * dq; np[2] += xaxis[2] * dq;
      } else {
        T xanchor[3] = {np[0], np[1], np[2]};
        if (jflags & JF_POS_NONZERO) {
          rot_vec_quat(xanchor, jpos, nq);
          xanchor[0] += np[0]; xanchor[1] += np[1]; xanchor[2] += np[2];
        }
        if constexpr (!Real<T>::exact) far = far || !(fabs(dq) <= maxangle);  // binary32(q) is off by eps |q|
        T sn, cs;
        sincos_half(dq * T(0.5), &sn, &cs);
        T qloc[4] = {cs, jaxis[0] * sn, jaxis[1] * sn, jaxis[2] * sn};
        mul_quat(nq, nq, qloc);
        if (jflags & JF_POS_NONZERO) {
          T vec[3];
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
    // float32 positions lose absolute accuracy with distance from the origin: beyond the model's
    // limit (FC_MAXCOORD) the tolerance band no longer covers the rounding error, so the whole
    // configuration goes to the exact path (this also catches NaN)
    if constexpr (!Real<T>::exact) {
      far = far || !(fmax(fabs(p[0]), fmax(fabs(p[1]), fabs(p[2]))) <= maxcoord);
      if (far) dead = kInf;
    }
    if (save_slot >= 0) {
      T *sv = save + (size_t)save_slot * 7 * sstride;
#pragma unroll
      for (int k = 0; k < 3; k++) sv[k * sstride] = p[k];
#pragma unroll
      for (int k = 0; k < 4; k++) sv[(3 + k) * sstride] = qt[k];
    }

    MJPL_ACC(0, tt);  // FK of the body
    for (int g = 0; g < ngeom; g++) {
      const int gtype = uni(ip[pc + G_TYPE]);
      const int gflags = uni(ip[pc + G_FLAGS]);
      const int gdoff = uni(ip[pc + G_DOFF]);
      Tab gd = tp + gdoff;
      const int store = uni(ip[pc + G_STORE]);
      const int geom_id = uni(ip[pc + G_GEOMID]);
      const unsigned smask = (unsigned)uni(ip[pc + G_SMASK]);
      const unsigned long long wmask_all =
          (unsigned long long)(unsigned)uni(ip[pc + G_WMASK_LO]) |
          ((unsigned long long)(unsigned)uni(ip[pc + G_WMASK_HI]) << 32);
      const unsigned long long pmask_all =
          (unsigned long long)(unsigned)uni(ip[pc + G_PMASK_LO]) |
          ((unsigned long long)(unsigned)uni(ip[pc + G_PMASK_HI]) << 32);
      IP swords = ip + pc + G_SIZE;
      pc += G_SIZE + MAX_SLOTS;

      Geom cur;
      if (gflags & GF_SAMEPOS) {
        cur.pos[0] = p[0]; cur.pos[1] = p[1]; cur.pos[2] = p[2];
      } else {
        T lpos[3] = {gd[0], gd[1], gd[2]};
        mul_mat_vec3(cur.pos, R, lpos);
        cur.pos[0] += p[0]; cur.pos[1] += p[1]; cur.pos[2] += p[2];
      }
      const bool curbox = MBOX && gtype == GT_BOX;  // (wave-uniform) a moving box: the whole frame
      if (gflags & GF_SAMEROT) {
        if (curbox) {
#pragma unroll
          for (int k = 0; k < 9; k++) cur.m[k] = R[k];
        } else {
          cur.m[2] = R[2]; cur.m[5] = R[5]; cur.m[8] = R[8];
        }
      } else {
        T lq[4] = {gd[3], gd[4], gd[5], gd[6]}, gq[4];
        mul_quat(gq, qt, lq);
        if (curbox) quat2mat(cur.m, gq);
        else quat2zaxis(cur.m, gq);
      }
      if (!curbox) cur.m[0] = cur.m[1] = cur.m[3] = cur.m[4] = cur.m[6] = cur.m[7] = T(0);

      MJPL_ACC(1, tt);  // geom record + pose
      // ---- culls: straight-line code, four static rows (then four register slots) at a time;
      // a pair costs ~9 VALU and no branch.  Pushes are a rolled loop over the few rows of a
      // chunk that some lane passed.  (A rolled per-pair loop spends most of its time in scalar
      // control flow: ~5 branches and ~35 instructions per pair.)
      Tab wbound = gd + GD_WBOUND;            // [nwpad] bounds, then [nwpad] margins
      Tab sbound = wbound + 2 * nwpad;        // [16] bounds, [16] margins, [16][3] sizes
      // t6: a slot partner's pos + z axis; t6b: its x, y axes (MBOX builds, partner a box)
      auto push = [&](auto boxq, unsigned long long pm, int kind, int index, int ptype, bool pfirst,
                      const T *t6, const T *t6b = nullptr) {
        constexpr bool BOXQ = decltype(boxq)::value;
        constexpr int CAP = BOXQ ? QB_CAP : QN_CAP;
        int &fill = BOXQ ? qb : qn;
        const int cnt = (int)__builtin_popcountll(pm);
        if (fill + cnt > CAP) {  // make room: one batch leaves the top of the queue
          MJPL_ACC(2, tt);
#ifdef MJPL_STAMPS
          queue_drain<T, BOXQ, false, MBOX>(wq, fill, ltab, lwcull, lwnarrow, nwpad, tol, ps, acc + 8);
#else
          queue_drain<T, BOXQ, false, MBOX>(wq, fill, ltab, lwcull, lwnarrow, nwpad, tol, ps);
#endif
          fl = wq.flags[lane] & 3;
          dead = (fl != 0 || !active || far) ? kInf : T(0);
          MJPL_ACC(3, tt);  // drains
        }
        T *qf = BOXQ ? wq.bf : wq.nf;
        int *qi0 = BOXQ ? wq.bi0 : wq.ni0, *qi1 = BOXQ ? wq.bi1 : wq.ni1;
        const bool mine = (pm >> lane) & 1ull;
        const int off = fill + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(pm >> 32),
                                                              __builtin_amdgcn_mbcnt_lo((unsigned)pm, 0u));
        if (mine) {
          qf[0 * CAP + off] = cur.pos[0]; qf[1 * CAP + off] = cur.pos[1]; qf[2 * CAP + off] = cur.pos[2];
          qf[3 * CAP + off] = cur.m[2]; qf[4 * CAP + off] = cur.m[5]; qf[5 * CAP + off] = cur.m[8];
          if (!BOXQ && kind == EK_SLOT) {  // a static partner's pose is read from the world tables at the drain
            qf[6 * CAP + off] = t6[0]; qf[7 * CAP + off] = t6[1]; qf[8 * CAP + off] = t6[2];
            qf[9 * CAP + off] = t6[3]; qf[10 * CAP + off] = t6[4]; qf[11 * CAP + off] = t6[5];
          }
          if constexpr (BOXQ && MBOX) {
            qf[6 * CAP + off] = cur.m[0]; qf[7 * CAP + off] = cur.m[3]; qf[8 * CAP + off] = cur.m[6];
            qf[9 * CAP + off] = cur.m[1]; qf[10 * CAP + off] = cur.m[4]; qf[11 * CAP + off] = cur.m[7];
            if (kind == EK_SLOT) {
#pragma unroll
              for (int k = 0; k < 6; k++) qf[(12 + k) * CAP + off] = t6[k];
#pragma unroll
              for (int k = 0; k < 6; k++) qf[(18 + k) * CAP + off] = t6b ? t6b[k] : T(0);
            }
          }
          qi0[off] = lane | (gtype << 6) | (ptype << 10) | ((pfirst ? 1 : 0) << 14) | (kind << 15) | (index << 17);
          qi1[off] = gdoff;
        }
        fill += cnt;
      };
      const std::integral_constant<bool, false> kGeneral{};
      const std::integral_constant<bool, true> kBoxes{};
#ifdef MJPL_X_Q_FKONLY
      const unsigned long long pmask_use = 0, wmask_use = 0;
      const unsigned smask_use = 0;
#else
      const unsigned long long pmask_use = pmask_all, wmask_use = wmask_all;
      const unsigned smask_use = smask;
#endif
      // static planes (few)
      for (unsigned long long pm_ = pmask_use; pm_; pm_ &= pm_ - 1) {
        const int wc = (int)__builtin_ctzll(pm_);
        Tab rw = tp + uni(ip[H_OFF_WNARROW]) + wc * WN_LEN;
        const T ppos[3] = {wcull[wc_at(wc, 0)], wcull[wc_at(wc, 1)], wcull[wc_at(wc, 2)]};
        const T pz[3] = {rw[WN_ZAXIS], rw[WN_ZAXIS + 1], rw[WN_ZAXIS + 2]};
        T dif[3] = {cur.pos[0] - ppos[0], cur.pos[1] - ppos[1], cur.pos[2] - ppos[2]};
        const unsigned long long pm = __builtin_amdgcn_ballot_w64(!(dot3(dif, pz) + dead > wbound[wc]));
        if (pm == 0ull) continue;
        if (curbox) push(kBoxes, pm, EK_PLANE, wc, GT_PLANE, true, ppos);
        else push(kGeneral, pm, EK_PLANE, wc, GT_PLANE, true, ppos);
      }
      // other static geoms: one wide scalar load of four rows (+ their four bounds) per chunk
#ifdef MJPL_CHUNK_PREFETCH
      T nrc[16], nbc[4];
#pragma unroll
      for (int k = 0; k < 16; k++) nrc[k] = wcull[k];
#pragma unroll
      for (int k = 0; k < 4; k++) nbc[k] = wbound[k];
#endif
      for (int base = 0; base < nwpad; base += 4) {
        const unsigned bits = (unsigned)(wmask_use >> base) & 15u;
        T rcv[16], bcv[4];
#ifdef MJPL_CHUNK_PREFETCH
#pragma unroll
        for (int k = 0; k < 16; k++) rcv[k] = nrc[k];
#pragma unroll
        for (int k = 0; k < 4; k++) bcv[k] = nbc[k];
        {
          Tab rn = wcull + (base + 4) * WC_LEN;
          Tab bn = wbound + (base + 4);
#pragma unroll
          for (int k = 0; k < 16; k++) nrc[k] = rn[k];
#pragma unroll
          for (int k = 0; k < 4; k++) nbc[k] = bn[k];
        }
#else
        if (bits == 0) continue;
        {
          Tab rn = wcull + base * WC_LEN;
          Tab bn = wbound + base;
#pragma unroll
          for (int k = 0; k < 16; k++) rcv[k] = rn[k];
#pragma unroll
          for (int k = 0; k < 4; k++) bcv[k] = bn[k];
        }
#endif
        if (bits == 0) continue;
        pin_geom(cur);
        unsigned long long m0, m1, m2, m3;
#define MJPL_QCULL(k, out)                                                              \
        {                                                                               \
          T dx = cur.pos[0] - rcv[k], dy = cur.pos[1] - rcv[4 + (k)],                     \
            dz = cur.pos[2] - rcv[8 + (k)];                                              \
          out = __builtin_amdgcn_ballot_w64(!(sqnorm3(dx, dy, dz) + dead > bcv[k]));     \
        }
        MJPL_QCULL(0, m0) MJPL_QCULL(1, m1) MJPL_QCULL(2, m2) MJPL_QCULL(3, m3)
#undef MJPL_QCULL
#ifdef MJPL_X_Q_NOPUSH  // timing-only build: culls, no pushes (the masks stay live through `sink`)
        sink ^= m0 ^ (m1 << 1) ^ (m2 << 2) ^ (m3 << 3);
        continue;
#endif
        if ((m0 | m1 | m2 | m3) == 0ull) continue;
#pragma unroll 1
        for (int k = 0; k < 4; k++) {
          const unsigned long long pm = k == 0 ? m0 : (k == 1 ? m1 : (k == 2 ? m2 : m3));
          // (a row that is no partner of this geom was culled with the chunk all the same, and a PLANE's row holds the
          //  bound of the plane test: near the plane's origin it passed, and the pair was examined a second time)
          if (pm == 0ull || !((bits >> k) & 1u)) continue;
          const int wc = base + k;
          // the row's info word came with the chunk (first 4 bytes of its 4th scalar)
          const T iw = k == 0 ? rcv[12] : (k == 1 ? rcv[13] : (k == 2 ? rcv[14] : rcv[15]));
          const int info = uni(info_bits(iw));
          const int ptype = info & 255, pgid = info >> 8;
          // mj_collision order: smaller geom type first, geom id breaks ties
          const bool pfirst = (ptype < gtype) || (ptype == gtype && pgid < geom_id);
          if ((WBOX && ptype == GT_BOX) || curbox) push(kBoxes, pm, EK_STATIC, wc, ptype, pfirst, cur.pos);
          else push(kGeneral, pm, EK_STATIC, wc, ptype, pfirst, cur.pos);
        }
      }
      // earlier moving geoms in the register slots
      if (smask_use != 0) {
        pin_geom(cur);
        unsigned anybits = 0;
        unsigned lanebits = 0;
#pragma unroll
        for (int n = 0; n < MAXS; n++) {  // literal register operands after unrolling
          if ((smask_use >> n) & 1u) {
            T dx = cur.pos[0] - sf.f[0][n], dy = cur.pos[1] - sf.f[1][n], dz = cur.pos[2] - sf.f[2][n];
            const bool ps = !(sqnorm3(dx, dy, dz) + dead > sbound[n]);
            lanebits |= ps ? (1u << n) : 0u;
            anybits |= (__builtin_amdgcn_ballot_w64(ps) != 0ull) ? (1u << n) : 0u;
          }
        }
#ifdef MJPL_X_Q_NOPUSH
        sink ^= (unsigned long long)anybits * 0x9E3779B97F4A7C15ull ^ (unsigned long long)lanebits;
        anybits = 0;
#endif
        for (unsigned ab = anybits; ab; ab &= ab - 1) {
          const int slot = (int)__builtin_ctz(ab);
          const unsigned long long pm = __builtin_amdgcn_ballot_w64((lanebits >> slot) & 1u);
          const int pw = uni(swords[slot]);
          const int ptype = (pw >> 12) & 15;
          T t6[6];
          slot_get6(sf, slot, t6);
          if (MBOX && (curbox || ptype == GT_BOX)) {
            T t6b[6] = {0, 0, 0, 0, 0, 0};
            if ((pw & 63) != SLOT_NONE) slot_get6(sf, pw & 63, t6b);  // stored box: x and y axes
            push(kBoxes, pm, EK_SLOT, slot, ptype, (pw & P_FIRST) != 0, t6, t6b);
          } else {
            push(kGeneral, pm, EK_SLOT, slot, ptype, (pw & P_FIRST) != 0, t6);
          }
        }
      }

      MJPL_ACC(2, tt);  // culls + pushes
      if (store >= 0) {
        const T t6[6] = {cur.pos[0], cur.pos[1], cur.pos[2], cur.m[2], cur.m[5], cur.m[8]};
        slot_put6(sf, store & 63, t6);
        if (MBOX && ((store >> 6) & 63) != SLOT_NONE) {
          const T t6b[6] = {cur.m[0], cur.m[3], cur.m[6], cur.m[1], cur.m[4], cur.m[7]};
          slot_put6(sf, (store >> 6) & 63, t6b);
        }
      }
      MJPL_ACC(4, tt);  // slot store
    }
  }
  MJPL_ACC(5, tt);
#ifdef MJPL_STAMPS
  if (qn > 0) queue_drain<T, false, true, MBOX>(wq, qn, ltab, lwcull, lwnarrow, nwpad, tol, ps, acc + 8);
  if ((WBOX || MBOX) && qb > 0) queue_drain<T, true, true, MBOX>(wq, qb, ltab, lwcull, lwnarrow, nwpad, tol, ps, acc + 8);
#else
  if (qn > 0) queue_drain<T, false, true, MBOX>(wq, qn, ltab, lwcull, lwnarrow, nwpad, tol, ps);
  if ((WBOX || MBOX) && qb > 0) queue_drain<T, true, true, MBOX>(wq, qb, ltab, lwcull, lwnarrow, nwpad, tol, ps);
#endif
  fl = wq.flags[lane] & 3;
  MJPL_ACC(3, tt);
#ifdef MJPL_STAMPS
  if (lane == 0) {
    for (int k = 0; k < 6; k++) atomicAdd(&g_stamps[k], acc[k]);
    for (int k = 8; k < 16; k++) atomicAdd(&g_stamps[k], acc[k]);
    atomicAdd(&g_stamps[7], 1ull);
  }
#endif
#ifdef MJPL_X_Q_NOPUSH
  if (sink == 0x123456789ull) fl |= 1;
#endif
  if (active && far) return V_UNSURE;  // nothing this lane's candidates said can be trusted
  return !active ? V_NONE : ((fl & 1) ? V_CONTACT : ((fl & 2) ? V_UNSURE : V_NONE));
}

}  // namespace mjpl
