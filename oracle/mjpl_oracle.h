/*
 * mjpl_oracle.h -- CPU ORACLE (test infrastructure, NOT the product).
 *
 * Plain-C float64 restatement of the reference hot path
 *     CollisionConstraint.valid_config          (src/mjpl/constraint/collision_constraint.py:26-30)
 *     CollisionRuleset.obeys_ruleset            (src/mjpl/constraint/collision_constraint.py:66-95)
 *     _step / _valid_collision_interval         (src/mjpl/planning/utils.py:167-216)
 * whose arithmetic lives in the un-vendored third-party wheel `mujoco >= 3`
 * (pyproject.toml:12; call sites collision_constraint.py:28-29).  MuJoCo's C
 * sources are not present in /root/reference nor installed, so the engine
 * routines (mj_kinematics, mj_collision, the primitive narrowphase) are restated
 * from their published algorithm -- every such function is tagged [MJ-recalled].
 *
 * PARITY PINNING: pinned against the analytic known-answer tests the reference's
 * own test-suite holds for this path (tests/golden/kat_reference.json, from
 * test/test_collision_constraint.py:16-33 and test/test_planning_utils.py:207-344)
 * and against independent closed-form FK fixtures (tests/golden/fk_*.json).
 * Capsule/box narrowphase and 6/7-DoF verdicts are NOT pinned by any reference
 * test: for those this oracle is "parity unpinned".
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  The product (mjpl_amd/, libmjpl_hip.so) never does.
 */
#ifndef MJPL_ORACLE_H
#define MJPL_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* MuJoCo enums (mjtJoint / mjtGeom) -- values as in mujoco/mjmodel.h [MJ-recalled] */
enum { ORC_JNT_FREE = 0, ORC_JNT_BALL = 1, ORC_JNT_SLIDE = 2, ORC_JNT_HINGE = 3 };
enum {
  ORC_GEOM_PLANE = 0, ORC_GEOM_HFIELD = 1, ORC_GEOM_SPHERE = 2, ORC_GEOM_CAPSULE = 3,
  ORC_GEOM_ELLIPSOID = 4, ORC_GEOM_CYLINDER = 5, ORC_GEOM_BOX = 6, ORC_GEOM_MESH = 7
};

/* The subset of mjModel the path reads.  All arrays are borrowed. */
typedef struct orc_model {
  int32_t nq, njnt, nbody, ngeom;
  const int32_t *body_parentid;   /* [nbody] */
  const int32_t *body_weldid;     /* [nbody] */
  const int32_t *body_jntadr;     /* [nbody] (-1 if none) */
  const int32_t *body_jntnum;     /* [nbody] */
  const double  *body_pos;        /* [nbody*3] */
  const double  *body_quat;       /* [nbody*4] */
  const int32_t *jnt_type;        /* [njnt] */
  const int32_t *jnt_qposadr;     /* [njnt] */
  const double  *jnt_axis;        /* [njnt*3] */
  const double  *jnt_pos;         /* [njnt*3] */
  const double  *qpos0;           /* [nq] */
  const int32_t *geom_type;       /* [ngeom] */
  const int32_t *geom_bodyid;     /* [ngeom] */
  const int32_t *geom_contype;    /* [ngeom] */
  const int32_t *geom_conaffinity;/* [ngeom] */
  const double  *geom_size;       /* [ngeom*3] */
  const double  *geom_pos;        /* [ngeom*3] */
  const double  *geom_quat;       /* [ngeom*4] */
  const double  *geom_rbound;     /* [ngeom] */
  const double  *geom_margin;     /* [ngeom] */
} orc_model;

/* A batch request: planning columns are scattered into a full-qpos template. */
typedef struct orc_batch {
  const double  *qpos_base;   /* [nq]   values of the non-planning joints      */
  const int32_t *qidx;        /* [nplan] qpos index of each planning column    */
  int32_t        nplan;
  int32_t        layout;      /* 0: SoA [nplan][N]   1: AoS [N][nplan]         */
  const int32_t *allowed;     /* [nallowed*2] sorted body-id pairs (a6)        */
  int32_t        nallowed;
} orc_batch;

/* status codes */
#define ORC_OK            0
#define ORC_E_JOINT      -2   /* joint type outside {slide,hinge}               */
#define ORC_E_PAIRTYPE   -3   /* geom pair with no restated narrowphase routine */
#define ORC_E_OVERFLOW   -4   /* contact buffer too small                       */
#define ORC_E_NONFINITE  -5   /* NaN/inf in an edge (reference would spin)      */

/* a3: mj_kinematics -- body and geom world poses for one qpos. Outputs may be NULL. */
int orc_kinematics(const orc_model *m, const double *qpos,
                   double *xpos, double *xquat, double *xmat,
                   double *geom_xpos, double *geom_xmat);

/* a4+a5: mj_collision -- fills contact_geom[ncon][2]; returns status. */
int orc_collision(const orc_model *m, const double *geom_xpos, const double *geom_xmat,
                  int32_t *contact_geom, int32_t maxcon, int32_t *ncon);

/* a6: CollisionRuleset.obeys_ruleset on a contact list. returns 1/0. */
int orc_obeys_ruleset(const orc_model *m, const int32_t *contact_geom, int32_t ncon,
                      const int32_t *allowed, int32_t nallowed);

/* a1: CollisionConstraint.valid_config(qpos[nq]) -> 1 valid, 0 in collision, <0 error */
int orc_valid_config(const orc_model *m, const int32_t *allowed, int32_t nallowed,
                     const double *qpos);

/* a8: _step(start, target, max_step_dist) -> out[n]  (sequential-sum 2-norm) */
void orc_step(const double *start, const double *target, int32_t n, double max_step,
              double *out);

/* a9: _valid_collision_interval on full-nq vectors: 1/0, <0 error.
 * nwaypoints (nullable) receives the number of interior waypoints generated,
 * first_bad (nullable) the 1-based index of the first colliding one (0 if none). */
int orc_valid_collision_interval(const orc_model *m, const int32_t *allowed, int32_t nallowed,
                                 const double *start, const double *end, double step_dist,
                                 int32_t *nwaypoints, int32_t *first_bad);

/* batched a1 over N configurations (nthreads >= 1: static partition, pthreads) */
int orc_valid_configs(const orc_model *m, const orc_batch *b, const double *Q, int64_t N,
                      int32_t nthreads, uint8_t *valid);

/* batched "validated edge" (SURVEY 8d): endpoint QB checked first (index 0), then the
 * interior waypoints of _valid_collision_interval(QA,QB,step) in order (1..K).
 * valid[e] = AND.  first_bad (nullable): -1 if valid else index of first failing check.
 * ncheck (nullable): number of configuration checks actually performed per edge. */
int orc_valid_edges(const orc_model *m, const orc_batch *b, const double *QA, const double *QB,
                    int64_t E, double step_dist, int32_t nthreads,
                    uint8_t *valid, int32_t *first_bad, int32_t *ncheck);

/* batched a3 for FK parity: outputs are [N][nbody*3], [N][nbody*4], [N][ngeom*3], [N][ngeom*9] */
int orc_fk_batch(const orc_model *m, const orc_batch *b, const double *Q, int64_t N,
                 double *xpos, double *xquat, double *geom_xpos, double *geom_xmat);

/* single primitive pair test, for unit tests: returns 1 contact / 0 none / <0 unsupported.
 * pos[3], mat[9] row-major, size[3]; types as ORC_GEOM_*; argument order free. */
int orc_pair_test(int32_t type1, const double *pos1, const double *mat1, const double *size1,
                  int32_t type2, const double *pos2, const double *mat2, const double *size2,
                  double margin);


/* ------------------------------------------------------------------ row f1: PoseConstraint
 * (src/mjpl/constraint/pose_constraint.py:11-171).  The SE3 / SO3 arithmetic of the reference
 * lives in the un-vendored `mink >= 0.0.8` wheel (pyproject.toml:11) on top of mju_mat2Quat /
 * mju_mulQuat, restated here from the published algorithms [MINK-recalled], [MJ-recalled];
 * np.linalg.pinv of the 6x6 symmetric J J^T is restated as a cyclic-Jacobi eigen-decomposition
 * with numpy's cutoff (eigenvalues <= 1e-15 * largest are dropped).  Pinned by the reference's
 * own analytic test (test/test_pose_constraint.py:16-50) and by a numpy restatement that uses
 * np.linalg.pinv itself (tests/test_pose_oracle.py); otherwise tolerance-level parity. */
typedef struct orc_pose {
  int32_t site_body;          /* model.site_bodyid[site]                         */
  double  site_pos[3];        /* model.site_pos[site]                            */
  double  site_quat[4];       /* model.site_quat[site]                           */
  double  c_quat[4];          /* C_T_world = reference_frame.inverse(): rotation */
  double  c_pos[3];           /*                                    translation  */
  double  lo[6], hi[6];       /* x y z roll pitch yaw bounds (pose_constraint.py:57-59) */
  double  tolerance, q_step;  /* :29-30                                          */
  const double *jnt_range;    /* [njnt*2] JointLimitConstraint (joint_limit_constraint.py:16-17) */
  int32_t max_iters;          /* the reference loops without a bound; the oracle gives up here   */
} orc_pose;

#define ORC_E_NOCONVERGE -6   /* orc_pose_apply ran max_iters projections */

/* data.site(name).xpos / xmat after mj_kinematics (utils.site_pose, src/mjpl/utils.py:60-75) */
int orc_site_pose(const orc_model *m, const orc_pose *p, const double *qpos, double *xpos, double *xmat);
/* _displacement_from_constraint (pose_constraint.py:93-123) -> dx[6] */
int orc_pose_displacement(const orc_model *m, const orc_pose *p, const double *qpos, double *dx);
/* _get_jacobian (pose_constraint.py:125-147): E_rpy @ [jacp; jacr], row-major [6][njnt] */
int orc_pose_jacobian(const orc_model *m, const orc_pose *p, const double *qpos, double *J);
/* np.linalg.pinv of a symmetric 6x6 (row-major) */
void orc_pinv_sym6(const double *A, double *out);
/* PoseConstraint.valid_config (pose_constraint.py:72-76): 1/0, <0 error */
int orc_pose_valid(const orc_model *m, const orc_pose *p, const double *qpos);
/* PoseConstraint.apply (pose_constraint.py:78-91): 1 -> q_out holds the projection, 0 -> None,
 * <0 error.  iters (nullable) receives the number of projection steps taken. */
int orc_pose_apply(const orc_model *m, const orc_pose *p, const double *q_old, const double *q,
                   double *q_out, int32_t *iters);
/* batched apply over N rows of full-nq vectors [N][nq]; ok[i] = 1/0; iters[i] = projection steps
 * taken, negative if the row gave up after max_iters */
int orc_pose_apply_batch(const orc_model *m, const orc_pose *p, const double *Q_old, const double *Q,
                         int64_t N, int32_t nthreads, double *Q_out, uint8_t *ok, int32_t *iters);

/* IK seeds: the CPU statement of the product's damped-least-squares iteration (mjpl_oracle_pose.c);
 * fields as mjpl_ik_desc in include/mjpl_hip.h */
typedef struct orc_ik {
  int32_t site_body;
  double  site_pos[3], site_quat[4], target_pos[3], target_quat[4];
  double  pos_tolerance, ori_tolerance;
  int32_t iterations;
  double  damping, lm_damping, max_step;
  const double  *jnt_range;
  const uint8_t *movable;
  int32_t restarts;
  uint64_t restart_seed;
} orc_ik;
/* one row: 1 solved / 0 not / < 0 error; `row` keys the restart draws */
int orc_ik_solve(const orc_model *m, const orc_ik *d, const double *q0, int64_t row, double *q_out,
                 int32_t *iters, double *err2);
int orc_ik_solve_batch(const orc_model *m, const orc_ik *d, const double *Q, int64_t N, int32_t nthreads,
                       double *Q_out, uint8_t *ok, int32_t *iters, double *err);

/* sin/cos used by mju_axisAngle2Quat: 0 = libm (default, as MuJoCo), 1 = the fdlibm algorithm in
 * fixed-order IEEE double operations (bit-reproducible; see orc_math.h).  Process-wide. */
void orc_set_trig(int mode);
int orc_get_trig(void);

#ifdef __cplusplus
}
#endif
#endif
