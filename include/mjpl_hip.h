/*
 * mjpl_hip.h -- C ABI of libmjpl_hip.so, the MI355X (gfx950) batched collision-validation
 * engine that sits behind mjpl's Constraint plug-in surface.
 *
 * The reference (adlarkin/mjpl, pure Python) has no FFI for this path: its
 * CollisionConstraint calls the third-party MuJoCo engine through pybind, one
 * configuration at a time.  Each entry point below names the reference interface
 * (file:line under /root/reference) whose work it replaces; INTEGRATION.md shows the
 * ctypes binding a maintainer adds on the reference side.
 *
 * Conventions: plain pointers and sizes, caller owns every buffer, every function returns
 * an int status (MJPL_OK = 0) and never throws; mjpl_last_error() returns a thread-local
 * message for the last non-zero status.  One engine per thread/stream.  There is NO CPU
 * fallback: without a gfx950 device mjpl_create fails with MJPL_E_NODEVICE.
 *
 * Batch layout: a batch holds only the nplan PLANNING columns of qpos (the joints a
 * planner samples, rrt.py:162-206); the remaining qpos entries come from the template
 * given to mjpl_set_planning.  layout = MJPL_SOA: Q[c*N + i] (coalesced column reads,
 * the native layout); layout = MJPL_AOS: Q[i*nplan + c] (numpy row-major, transposed
 * through LDS inside the kernel).
 */
#ifndef MJPL_HIP_H
#define MJPL_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MJPL_OK            0
#define MJPL_E_ARG        -1   /* bad argument (NULL, negative size, step_dist <= 0 ...)      */
#define MJPL_E_JOINT      -2   /* joint type outside {slide, hinge} (reference README.md:20)  */
#define MJPL_E_PAIRTYPE   -3   /* a colliding geom pair has no primitive narrowphase routine  */
#define MJPL_E_CAPACITY   -4   /* model exceeds a compiled-in limit (slots, LDS)              */
#define MJPL_E_NODEVICE   -5   /* no usable HIP device                                        */
#define MJPL_E_HIP        -6   /* HIP runtime error (message in mjpl_last_error)              */
#define MJPL_E_NONFINITE  -7   /* NaN/inf edge (the reference's waypoint loop would not end)  */

#define MJPL_SOA 0
#define MJPL_AOS 1

/* mjpl_check_edges flags */
#define MJPL_EDGE_INTERIOR_ONLY 1  /* skip check 0 (the endpoint): exactly _valid_collision_interval */

/* mjtJoint / mjtGeom values */
#define MJPL_JNT_SLIDE 2
#define MJPL_JNT_HINGE 3
#define MJPL_GEOM_PLANE 0
#define MJPL_GEOM_SPHERE 2
#define MJPL_GEOM_CAPSULE 3
#define MJPL_GEOM_BOX 6

/* The mjModel subset the path reads; field names as in mjModel.  Replaces the
 * `model: mujoco.MjModel` argument of CollisionConstraint.__init__
 * (src/mjpl/constraint/collision_constraint.py:10-24).  Arrays are borrowed for the
 * duration of mjpl_create only. */
typedef struct mjpl_model_desc {
  int32_t nq, njnt, nbody, ngeom;
  const int32_t *body_parentid;    /* [nbody]   */
  const int32_t *body_weldid;      /* [nbody]   */
  const int32_t *body_jntadr;      /* [nbody]   */
  const int32_t *body_jntnum;      /* [nbody]   */
  const double  *body_pos;         /* [nbody*3] */
  const double  *body_quat;        /* [nbody*4] */
  const int32_t *jnt_type;         /* [njnt]    */
  const int32_t *jnt_qposadr;      /* [njnt]    */
  const double  *jnt_axis;         /* [njnt*3]  */
  const double  *jnt_pos;          /* [njnt*3]  */
  const double  *qpos0;            /* [nq]      */
  const int32_t *geom_type;        /* [ngeom]   */
  const int32_t *geom_bodyid;      /* [ngeom]   */
  const int32_t *geom_contype;     /* [ngeom]   */
  const int32_t *geom_conaffinity; /* [ngeom]   */
  const double  *geom_size;        /* [ngeom*3] */
  const double  *geom_pos;         /* [ngeom*3] */
  const double  *geom_quat;        /* [ngeom*4] */
  const double  *geom_rbound;      /* [ngeom]   */
  const double  *geom_margin;      /* [ngeom]   */
} mjpl_model_desc;

typedef struct mjpl_engine mjpl_engine;

typedef struct mjpl_info {
  int32_t device;            /* HIP device ordinal                                   */
  int32_t nplan;             /* planning columns per configuration                    */
  int32_t nmoving_geoms;     /* geoms on bodies below a joint                         */
  int32_t nstatic_geoms;     /* geoms welded to the world (poses folded at create)    */
  int32_t npairs;            /* geom pairs left after the mj_collision filters + a6   */
  int32_t npairs_world;      /* ... of which moving-vs-static                         */
  int32_t nslots;            /* register slots holding earlier moving geoms           */
  int32_t nsaves;            /* LDS pose saves for branching bodies                   */
  int32_t lds_bytes_configs; /* dynamic LDS per block, configs kernel                 */
  int32_t lds_bytes_edges;   /* dynamic LDS per block, edges kernel                   */
  int32_t block_threads;
  int32_t compute_units;
  int32_t filter_enabled;    /* float32 filter in front of the exact kernels              */
  float   filter_tol;        /* its tolerance band, metres                                */
  int32_t lds_bytes_filter;  /* dynamic LDS per block, float32 filter kernels             */
  int32_t filter_block_threads;
  char    arch[32];          /* gcnArchName, e.g. "gfx950:sramecc+:xnack-"            */
  /* the filter's binary32 error bound for this model: a float32 signed distance is
   * within filter_err_a + filter_err_b * C of the float64 one while every moving coordinate
   * magnitude is <= C; configurations beyond filter_max_coord go to the float64 kernels */
  float   filter_max_coord;
  float   filter_err_a;
  float   filter_err_b;
  int32_t filter_poisoned_geoms; /* static geoms too large / far for binary32: always re-checked */
  /* which float32 interpreter this model runs: 0 = queued narrowphase, small builds (no moving
   * boxes, <= 16 stored geoms); 1 = queued, the general 24-slot build with moving boxes in the
   * box queue; 2 = immediate narrowphase (more than 24 stored geoms, or MJPL_FORCE_IMMEDIATE=1
   * at creation: tests).  A specialised library can replace 0 and 1 (mjpl_spec_loaded), not 2. */
  int32_t filter_interpreter;
  /* how an edge launch is laid out on the device (defaults; MJPL_PERSIST / MJPL_TAIL at creation):
   * persistent_kernels = 1: endpoint and item kernels run as persistent grids of waves with tile queues;
   * fused_tail = 1: walking kernel, pair re-check and exact edge kernel are roles of one launch (k_tail) */
  int32_t persistent_kernels;
  int32_t fused_tail;
  /* fused_edges = 1 (default where the model runs the queued float32 interpreter or its own kernels; MJPL_FUSED=0
   * at creation restores the two persistent kernels): ONE filter kernel per edge launch -- endpoint tiles and
   * waypoint tiles of a batch served by the same resident workgroups from a work pool in LDS (k_edges_fused) --
   * followed by k_tail.  With the filter off (filter_enabled = 0) the float64 checks take the same route
   * (k_edges_fused_f64, then k_check_edges over the edges too long for the pool).  fused_waves: wavefronts per
   * workgroup of that kernel. */
  int32_t fused_edges;
  int32_t fused_waves;
} mjpl_info;

/* ---- lifetime ------------------------------------------------------------------ */

/* CollisionConstraint.__init__ + CollisionRuleset.__init__
 * (collision_constraint.py:10-24, :42-64).  allowed_bodies: nallowed body-id pairs
 * (any order inside a pair).  The mj_collision pair filters (contype/conaffinity,
 * same weld body, weld parent-child) and the allowed-body-pair ruleset
 * (collision_constraint.py:66-95) are folded into a static pair list here. */
int mjpl_create(const mjpl_model_desc *model, const int32_t *allowed_bodies, int32_t nallowed,
                int32_t device, mjpl_engine **out);
void mjpl_destroy(mjpl_engine *e);

/* Which qpos entries a batch column feeds, and the template for all others
 * (planners keep non-planning joints at q_init, rrt.py:205-206; :162-172).
 * Default after create: nplan = nq, qidx = 0..nq-1, template = qpos0. */
int mjpl_set_planning(mjpl_engine *e, const int32_t *qidx, int32_t nplan, const double *qpos_base);

int mjpl_get_info(const mjpl_engine *e, mjpl_info *out);

/* Verdicts are always those of the exact float64 kernels.  By default a float32 FILTER kernel
 * runs first: the same algorithms in binary32, every comparison against a contact threshold
 * classified with a tolerance `tol` (metres) as certain / uncertain; only the items it cannot
 * decide are re-run by the float64 kernel (same stream, no host round trip).  enable = 0
 * sends everything through the float64 kernels.  Environment overrides at create:
 * MJPL_FILTER=0|1, MJPL_FILTER_TOL=<metres>.
 * Half of `tol` is the budget for the binary32 rounding error of this model's poses, which
 * mjpl_create bounds from the kinematic chain (mjpl_info.filter_err_a / _b).  The default tolerance
 * is max(1e-4, 2.5 * filter_err_a); a tolerance set here is kept as given, and if the model's
 * error floor does not fit under it the filter is switched off for this engine
 * (mjpl_info.filter_enabled = 0: everything runs on the float64 kernels). */
int mjpl_set_filter(mjpl_engine *e, int32_t enable, double tol);
/* how many items of the most recent launch went to the float64 kernel (synchronises) */
int64_t mjpl_filter_last_undecided(mjpl_engine *e);
/* diagnostic: the (edge, check index, geom a, geom b) records the last filter launch handed to the exact pair kernel;
 * returns how many there were (the arrays, each of `cap` entries or NULL, receive up to cap of them), or -1 */
int64_t mjpl_filter_undecided_pairs(mjpl_engine *e, int32_t *edge, int32_t *idx, int32_t *ga, int32_t *gb, int64_t cap);
/* The filter validates edges in two passes: the endpoints of all edges, then the interior
 * waypoints of the edges whose endpoint passed (MJPL_TWO_PASS=0 at create: one pass).  Returns how
 * many edges the interior pass of the most recent mjpl_check_edges* took (synchronises); 0 for a
 * one-pass launch, -1 if the filter is off. */
int64_t mjpl_filter_last_interior_edges(mjpl_engine *e);
/* The interior pass runs one lane per waypoint: a small kernel walks the waypoint recurrence of
 * every surviving edge once and writes the waypoints out as work items (MJPL_EXPAND=0 at create:
 * one lane per edge walks them; edges too long for the item space always do).  Returns how many
 * waypoint items the most recent mjpl_check_edges* checked (synchronises); -1 if never used. */
int64_t mjpl_filter_last_items(mjpl_engine *e);
/* The fused filter kernel spares an edge its waypoint checks when the END configuration keeps every enabled pair
 * farther apart than the pair can move while the planning joints travel the edge (a certificate that only ever says
 * "free": the verdict _valid_collision_interval, planning/utils.py:188-216, would reach by checking every waypoint).
 * Returns how many surviving edges of the most recent mjpl_check_edges* were spared (synchronises); -1 if the filter is off. */
int64_t mjpl_filter_last_certified(mjpl_engine *e);

/* ---- host-buffer entry points (stage H2D, run, copy back, synchronise) ---------- */

/* Batched CollisionConstraint.valid_config (collision_constraint.py:26-30):
 * valid[i] = 1 iff configuration i has no contact outside the allowed body pairs. */
int mjpl_check_configs(mjpl_engine *e, const double *Q, int64_t N, int32_t layout,
                       uint8_t *valid);

/* Batched "validated edge" of _constrained_extend (planning/utils.py:143-158): the endpoint
 * QB is collision-checked (check index 0), then the interior waypoints of
 * _valid_collision_interval(QA, QB, step_dist) (planning/utils.py:188-216) in order
 * (check index 1..K).  valid[e] = AND of all checks.  first_bad (nullable): -1 if valid,
 * else the index of the first failing check.  flags = MJPL_EDGE_INTERIOR_ONLY drops check 0,
 * which is _valid_collision_interval itself.  step_dist <= 0 -> MJPL_E_ARG
 * (the reference raises ValueError("`step_dist` must be > 0"), utils.py:207-208). */
int mjpl_check_edges(mjpl_engine *e, const double *QA, const double *QB, int64_t E,
                     double step_dist, int32_t layout, int32_t flags, uint8_t *valid,
                     int32_t *first_bad);

/* Batched mj_kinematics (call site collision_constraint.py:28) for the FK parity check:
 * xpos [N][nbody*3], xquat [N][nbody*4], geom_xpos [N][ngeom*3], geom_xmat [N][ngeom*9];
 * any output may be NULL. */
int mjpl_fk(mjpl_engine *e, const double *Q, int64_t N, int32_t layout,
            double *xpos, double *xquat, double *geom_xpos, double *geom_xmat);

/* ---- device-resident entry points (asynchronous on the engine's stream) --------- */

int mjpl_check_configs_dev(mjpl_engine *e, const double *dQ, int64_t N, int32_t layout,
                           uint8_t *dvalid);
int mjpl_check_edges_dev(mjpl_engine *e, const double *dQA, const double *dQB, int64_t E,
                         double step_dist, int32_t layout, int32_t flags, uint8_t *dvalid,
                         int32_t *dfirst_bad);
/* The device-pointer edge entry point cannot return MJPL_E_NONFINITE (it does not synchronise):
 * a NaN/inf or absurdly long edge gets valid = 0 and first_bad = -2, and a sticky status bit is
 * set on the device.  mjpl_take_status synchronises the engine's stream, stores MJPL_OK or
 * MJPL_E_NONFINITE (the status every mjpl_check_edges_dev call since the last take would have
 * returned) in *status and clears the bit.  The host-buffer mjpl_check_edges clears the bit
 * before it launches and reports it itself. */
int mjpl_take_status(mjpl_engine *e, int32_t *status);
/* as mjpl_check_configs_dev but one bit per configuration, packed by wavefront ballot:
 * bit (i & 63) of dbits[i >> 6]. */
int mjpl_check_configs_bits_dev(mjpl_engine *e, const double *dQ, int64_t N, int32_t layout,
                                uint64_t *dbits);

/* Brute-force nearest neighbour of each query among tree nodes (Tree.nearest_neighbor,
 * planning/tree.py:57-66): nodes SoA [nplan][cap] with n valid, queries SoA [nplan][M];
 * out_idx[j] = argmin_i ||node_i - query_j||, ties to the lowest index. */
int mjpl_nearest_dev(mjpl_engine *e, const double *dnodes, int64_t n, int64_t cap,
                     const double *dqueries, int64_t M, int32_t *dout_idx, double *dout_dist2);

/* The same over the node range [n0, n) only, behind an answer for the nodes below n0 (dprev_idx, dprev_dist2: what an
 * earlier mjpl_nearest_dev over the first n0 nodes returned for these queries; NULL, NULL: none): the result is the
 * whole scan's -- a node of the range wins only if it is strictly nearer, so ties still go to the lowest index
 * (Tree.nearest_neighbor, planning/tree.py:57-66, over a tree that has grown since it was last scanned). */
int mjpl_nearest_range_dev(mjpl_engine *e, const double *dnodes, int64_t n0, int64_t n, int64_t cap,
                           const double *dqueries, int64_t M, int32_t *dout_idx, double *dout_dist2,
                           const int32_t *dprev_idx, const double *dprev_dist2);

/* Which screen the last mjpl_nearest_dev ran in front of its float64 distances: 0 none (plain float64 scan),
 * 1 binary32 on the vector units, 2 binary16 operands on the matrix cores (large trees and query sets whose
 * coordinates are all below 256 in magnitude and whose nodes' squared norms are below 65504, the range of
 * binary16).  The result is the float64 scan's either way.  Synchronises. */
int32_t mjpl_nearest_last_screen(mjpl_engine *e);

/* ---- options: the switches tests and tools need, through the ABI instead of the caller's environment ----
 * mjpl_set_option(e, name, value): MJPL_OK, or MJPL_E_ARG for an unknown name / a value out of range.  Options
 * that shape the compiled model (e.g. "filter") take effect at the next mjpl_set_planning / first launch; the rest
 * at the next call.  Names: the table in tools/README.md ("nn_cells", "nn_cells_min_nodes", "nn_mfma",
 * "nn_sample", "filter", "fused", ...).  mjpl_get_option reads one back. */
int mjpl_set_option(mjpl_engine *e, const char *name, double value);
int mjpl_get_option(mjpl_engine *e, const char *name, double *value);
/* the table itself: how many options there are, and option `index`'s name (NULL beyond the table; *writable = 0 for a
 * read-only one).  The library reads NO switch from the environment -- except that with MJPL_DEBUG=1 set when an engine
 * is made, every writable option NAME is also taken from the variable MJPL_<NAME> (shell tools, A/B scripts). */
int32_t mjpl_option_count(void);
const char *mjpl_option_name(int32_t index, int32_t *writable);
/* per-model libraries (mjpl_amd/specialise.py) are looked for in `dir` instead of the directory `spec` beside this
 * library; NULL or "": the default again.  Process-wide; takes effect at the next mjpl_set_planning / mjpl_set_spec. */
int mjpl_set_spec_dir(const char *dir);

/* ---- device memory and stream helpers (so that hosts need no other GPU runtime) -- */

int mjpl_dev_alloc(mjpl_engine *e, size_t bytes, void **out);
int mjpl_dev_free(mjpl_engine *e, void *p);
int mjpl_h2d(mjpl_engine *e, void *dst, const void *src, size_t bytes);  /* async */
int mjpl_d2h(mjpl_engine *e, void *dst, const void *src, size_t bytes);  /* async */
int mjpl_sync(mjpl_engine *e);
/* raw hipStream_t of the engine, for interop with other runtimes */
void *mjpl_stream(mjpl_engine *e);

/* ---- measurement ---------------------------------------------------------------- */

/* Stages of one mjpl_check_edges* launch, in stream order.  With the float32 filter on:
 * endpoints of all edges (+ emission of the survivors' interior waypoints as work items), one lane
 * per waypoint item, the walking kernel for what was not expanded, the float64 re-check of the
 * undecided pairs / configurations, the float64 edge kernel for undecided whole edges.  With the
 * filter off the float64 edge kernel is the whole launch. */
#define MJPL_STAGE_ENDPOINTS 0  /* k_filter_endpoints (incl. the counter memset); with mjpl_info.fused_edges:
                                  * k_edges_fused, the whole float32 filter of the launch (stage 1 is then empty) */
#define MJPL_STAGE_ITEMS     1  /* k_filter_items                                          */
#define MJPL_STAGE_WALK      2  /* k_filter_edges                                          */
#define MJPL_STAGE_PATCH     3  /* k_patch_pairs (immediate interpreter: k_check_configs, patch mode); with mjpl_info.fused_tail:
                                  * k_tail, the one launch that holds the walking, pair and exact-edge roles (stages
                                  * 2 and 4 are then empty) */
#define MJPL_STAGE_EXACT     4  /* k_check_edges                                           */
#define MJPL_NSTAGES         5

/* Run mjpl_check_edges_dev `iters` times back to back on the engine's stream, timed with HIP
 * events recorded on that stream: two around the whole run (*ms_mean = mean duration of a call,
 * all kernels) and, on every `sample_every`-th call, one after each stage, so that
 * stage_ms[MJPL_NSTAGES] (nullable) receives each stage's mean duration over *nsamples (nullable)
 * sampled calls.  Sampling keeps the instrumentation below one percent of the run it measures;
 * sample_every > iters: no call is sampled (stage_ms untouched, *nsamples = 0).
 * Inputs and outputs are device-resident.  Used by bench.py for roofline.achieved. */
int mjpl_time_edges_stages_dev(mjpl_engine *e, const double *dQA, const double *dQB, int64_t E,
                               double step_dist, int32_t layout, uint8_t *dvalid, int32_t iters,
                               int32_t sample_every, float *ms_mean, float *stage_ms,
                               int32_t *nsamples);
/* The same with every call sampled: ms[k] = the mean call, ms_first[k] (nullable) = the mean
 * duration of the filter's item pass (filter off: of the float64 edge kernel). */
int mjpl_time_edges_dev(mjpl_engine *e, const double *dQA, const double *dQB, int64_t E,
                        double step_dist, int32_t layout, uint8_t *dvalid, int32_t iters,
                        float *ms, float *ms_first);
int mjpl_time_configs_dev(mjpl_engine *e, const double *dQ, int64_t N, int32_t layout,
                          uint8_t *dvalid, int32_t iters, float *ms);

/* ---- misc ------------------------------------------------------------------------ */

int mjpl_device_count(void);
const char *mjpl_last_error(void);
const char *mjpl_version(void);

/* ---- PoseConstraint (src/mjpl/constraint/pose_constraint.py) ---
 * One handle per (engine, site, constraint frame).  Batches are rows of FULL qpos vectors,
 * [N][nq] (apply() receives and returns full configurations, pose_constraint.py:78-91). */
typedef struct mjpl_pose_desc {
  int32_t site_body;          /* model.site_bodyid[site] (pose_constraint.py:70)              */
  double  site_pos[3];        /* model.site_pos[site]                                          */
  double  site_quat[4];       /* model.site_quat[site], wxyz                                   */
  double  c_quat[4];          /* C_T_world = reference_frame.inverse() (:63): rotation, wxyz   */
  double  c_pos[3];           /*                                        translation           */
  double  lo[6], hi[6];       /* x y z roll pitch yaw bounds (:57-59)                          */
  double  tolerance;          /* :29 (>= 0)                                                    */
  double  q_step;             /* :30 (> 0; may be +inf)                                        */
  const double *jnt_range;    /* [njnt*2] JointLimitConstraint (joint_limit_constraint.py:16-17) */
  int32_t max_iters;          /* the reference loops without a bound (:80); <= 0 -> 1000       */
} mjpl_pose_desc;

typedef struct mjpl_pose mjpl_pose;

/* PoseConstraint.__init__ (pose_constraint.py:18-70); ValueError cases -> MJPL_E_ARG */
int mjpl_pose_create(mjpl_engine *e, const mjpl_pose_desc *desc, mjpl_pose **out);
/* 1: a generated projection of the engine's per-model library serves this handle (the chain from the world to the
 * site's body as straight-line code: mjpl_amd/specialise.py); 0: the interpreting kernels.
 * Same results either way, bit for bit.  MJPL_POSE_SPEC=0 at creation keeps a handle on the interpreting kernels. */
int mjpl_pose_spec_loaded(mjpl_pose *p);
/* Host only, no device needed: the chain program of (model, site body) that mjpl_pose_create compiles -- int words
 * [header | per chain body {njnt}, per joint {type, qadr, joint id}], doubles [per body pos[3] quat[4]; per joint
 * axis[3] pos[3] qpos0] -- and the hash a generated projection is looked up by.  pi = pd = NULL returns the sizes. */
int mjpl_pose_chain_dump(const mjpl_model_desc *model, int32_t site_body, int32_t *pi, int32_t *npi, double *pd, int32_t *npd,
                         uint64_t *hash);
void mjpl_pose_destroy(mjpl_pose *p);
/* `pose_constraint.q_step = ...` (examples/franka_constrained_move_to_pose.py:72-75) */
int mjpl_pose_set_q_step(mjpl_pose *p, double q_step);

/* Batched PoseConstraint.apply (pose_constraint.py:78-91): row i of Q is projected onto the
 * constraint starting from itself; ok[i] = 1 and Q_out row i = the projection, or ok[i] = 0
 * (the reference's None: joint limits violated or farther than 2*q_step from Q_old row i).
 * iters (nullable): projection steps taken; negative = gave up after max_iters (ok = 0). */
int mjpl_pose_apply(mjpl_pose *p, const double *Q_old, const double *Q, int64_t N, double *Q_out,
                    uint8_t *ok, int32_t *iters);
/* Batched PoseConstraint.valid_config (pose_constraint.py:72-76); xpos [N][3] / xmat [N][9]
 * (nullable) receive the site's world pose (utils.site_pose, src/mjpl/utils.py:60-75). */
int mjpl_pose_valid(mjpl_pose *p, const double *Q, int64_t N, uint8_t *valid, double *xpos, double *xmat);
/* device-pointer variants: asynchronous on the engine's stream */
int mjpl_pose_apply_dev(mjpl_pose *p, const double *dQ_old, const double *dQ, int64_t N, double *dQ_out,
                        uint8_t *dok, int32_t *diters);
int mjpl_pose_valid_dev(mjpl_pose *p, const double *dQ, int64_t N, uint8_t *dvalid, double *dxpos, double *dxmat);

/* ---- IK seeds (the role of MinkIKSolver.solve_ik,
 * src/mjpl/inverse_kinematics/mink_ik_solver.py:72-116).  The reference's arithmetic is a QP in
 * the un-vendored mink / daqp wheels; parity is tolerance-level (pose error within the
 * tolerances, constraints obeyed -- what test/test_mink_ik_solver.py:64-70 asserts).  Every row
 * of Q [N][nq] is one start configuration (q_init_guess or a random_config restart, :108-115)
 * iterated with damped least squares; joints with movable[j] == 0 are held (:63-70). */
typedef struct mjpl_ik_desc {
  int32_t site_body;          /* model.site_bodyid[site]                                       */
  double  site_pos[3];
  double  site_quat[4];       /* wxyz                                                          */
  double  target_pos[3];      /* pose: SE3 in the world frame (ik_solver_interface.py:13-19)   */
  double  target_quat[4];     /* wxyz                                                          */
  double  pos_tolerance;      /* mink_ik_solver.py:19 (metres)                                 */
  double  ori_tolerance;      /* :20 (radians)                                                 */
  int32_t iterations;         /* :23 iterations per attempt (>= 1)                             */
  double  damping;            /* Tikhonov term (reference: damping=1e-3 of the QP, :106); <= 0 -> 1e-6 */
  double  lm_damping;         /* error-proportional term (FrameTask lm_damping=0.1, :80); < 0 -> 0.1   */
  double  max_step;           /* |dq|_inf limit per iteration, radians / metres; <= 0 -> 0.2   */
  const double  *jnt_range;   /* [njnt*2] mink.ConfigurationLimit (:86)                        */
  const uint8_t *movable;     /* [njnt] 1 for the solver's `joints`, 0 for held joints         */
  int32_t restarts;           /* a stalled row re-draws its joints uniformly in their ranges, up to this
                               * many times within `iterations` (the reference restarts a failed attempt
                               * from random_config, :108-115); 0: none                           */
  uint64_t restart_seed;      /* seed of those draws                                           */
} mjpl_ik_desc;

/* ok[i] = 1 iff row i reached the target within the tolerances; Q_out row i is its final
 * iterate either way; iters (nullable) the iterations used; err (nullable) [N][2] the final
 * position / orientation error norms. */
int mjpl_ik_solve(mjpl_engine *e, const mjpl_ik_desc *desc, const double *Q, int64_t N, double *Q_out,
                  uint8_t *ok, int32_t *iters, double *err);
int mjpl_ik_solve_dev(mjpl_engine *e, const mjpl_ik_desc *desc, const double *dQ, int64_t N, double *dQ_out,
                      uint8_t *dok, int32_t *diters, double *derr);

/* ---- per-model specialised filter kernels ------------------------------
 * mjpl_create compiles a model into a program (control words + constant tables) that generic kernels
 * interpret.  mjpl_amd/specialise.py turns one such program into straight-line code, builds the
 * float32 filter kernels around it as libmjpl_spec_<hash>.so, and an engine whose program has that
 * hash launches them instead of the interpreting ones (same verdicts; MJPL_SPEC=0 disables,
 * MJPL_SPEC_DIR overrides the directory `spec` next to the library).  mjpl_program_dump compiles a
 * model on the host only -- no device needed -- and returns the program and its hash: call it with
 * ip = fp = dp = NULL for the sizes, then with buffers. */
typedef struct mjpl_program_info {
  uint64_t hash;            /* FNV-1a of (ip, fp, dp, kernel variant, MJPL_SPEC_ABI, digest of the shared headers) */
  int32_t maxs, wbox, mbox; /* kernel variant: slot-file width, static / moving boxes present */
  int32_t immediate;        /* 1: the model runs the immediate interpreter (more than 24 stored geoms: not specialisable) */
  int32_t filter_usable;
  float   filter_tol;
  int32_t nslots, nsave;
  int32_t spec_abi;
  /* scene-generic libraries (one per ROBOT: static geoms may change without a compiler): the hash of what
   * such a library carries as literals -- moving bodies, their geoms and self pairs, planning set,
   * tolerance -- the number of cull rows per moving geom it reads from the engine's scene table, and
   * whether this program can run on one (<= scene_rows static geoms, <= 24 moving geoms numbered
   * consecutively; not the immediate interpreter's programs) */
  uint64_t robot_hash;
  int32_t scene_rows;
  int32_t scene_ok;
} mjpl_program_info;
int mjpl_program_dump(const mjpl_model_desc *model, const int32_t *allowed_bodies, int32_t nallowed,
                      const int32_t *qidx, int32_t nplan, const double *qpos_base, double filter_tol,
                      int32_t *ip, int32_t *nip, float *fp, double *dp, int32_t *ntab,
                      mjpl_program_info *info);
/* Host-only look-up, no device needed: is there a loadable library for this hash (mjpl_program_info.hash with
 * generic = 0, .robot_hash with generic = 1) that was built for this version of the engine?  1 / 0.  A
 * deployment step calls it after mjpl_amd/specialise.py; it is the look-up mjpl_create performs, and like it
 * may be called from any number of threads at once (the table of loaded libraries is the process's). */
int mjpl_spec_probe(uint64_t hash, int32_t generic);
/* 1 if the engine's current program runs on its own specialised library (hash of the whole program), 2 if
 * on the robot's scene-generic one (obstacles from a table), 0: the interpreting kernels */
int mjpl_spec_loaded(const mjpl_engine *e);
/* enable = 0: this engine runs the interpreting kernels whatever libraries exist (A/B measurements,
 * bench.py's "interpreter" variant); 1: look the libraries up again (the program's own first, then the
 * robot's scene-generic one); 2: the scene-generic one only (bench.py's "scene_generic" variant).
 * Synchronises. */
int mjpl_set_spec(mjpl_engine *e, int32_t enable);

/* ---- frontier bi-RRT, device-resident (BASELINE configs[3]) -------------
 * RRT.plan_to_configs' sample / extend / connect loop (src/mjpl/planning/rrt.py:190-235) for `lanes`
 * samples per round, with _constrained_extend's per-step rules (src/mjpl/planning/utils.py:139-164):
 * step <= epsilon towards the target, PoseConstraint projection first if a pose handle is given
 * (constraint order of examples/franka_constrained_move_to_pose.py:60-64), joint limits [lo, hi],
 * stop when a step moves less than 1e-8, leads away from the target, or fails the collision check
 * (endpoint + interval waypoints at interval_step; <= 0: endpoint only).  Both trees stay in HBM
 * as SoA slabs [nplan][capacity]; batches are over the engine's planning columns
 * (mjpl_set_planning).  The goal tree's roots are the goal configurations (the reference's sink
 * node, rrt.py:179-188, is implicit).  Rank k of W draws its own targets (counter-based generator
 * keyed by seed, k and the round); after every round all ranks' new nodes are all-gathered and
 * appended in rank order, so all ranks hold identical trees. */
typedef struct mjpl_rrt mjpl_rrt;

typedef struct mjpl_rrt_desc {
  int32_t lanes;               /* samples per rank per round                                   */
  int64_t capacity;            /* node capacity of each tree                                    */
  double  epsilon;             /* rrt.py:30                                                     */
  double  interval_step;       /* collision_interval_check step (rrt.py:28); <= 0: none         */
  double  goal_bias;           /* rrt.py:32                                                     */
  uint64_t seed;
  const double *lo, *hi;       /* [nplan] sampling box = JointLimitConstraint ranges            */
  mjpl_pose *pose;             /* nullable: PoseConstraint applied first                        */
  int64_t max_new_per_round;   /* slab rows per tree per rank per round; <= 0: max(128 lanes, 65536) */
  int32_t max_steps_per_round; /* most nodes a lane adds per extension; 0: no cap (the reference's
                                * _constrained_extend runs a chain to its end, planning/utils.py:139-164).
                                * > 0: a lane of the growing tree still under way after that many nodes is
                                * CARRIED -- it sits the connect phase out and, the next time its tree grows,
                                * goes on from the node it reached towards the same target instead of drawing
                                * a new one; a capped connect-phase lane just stops.  Deterministic per (lane,
                                * round): all ranks still hold identical trees. */
} mjpl_rrt_desc;

typedef struct mjpl_rrt_round_info {
  int32_t round;
  int32_t new_nodes[2];        /* appended this round, all ranks: start tree, goal tree         */
  int32_t nodes[2];            /* tree sizes after the round                                    */
  int32_t connected;           /* 1: a lane's two extensions met                                */
  int32_t conn_start, conn_goal; /* node ids of the junction in the start / goal tree           */
  int32_t conn_rank;           /* the rank whose lane connected (lowest rank, lowest lane wins) */
  int32_t stop_requested;      /* some rank passed request_stop != 0 this round                 */
} mjpl_rrt_round_info;

int mjpl_rrt_create(mjpl_engine *e, const mjpl_rrt_desc *desc, mjpl_rrt **out);
void mjpl_rrt_destroy(mjpl_rrt *r);
/* new query: start tree = {q_init}, goal tree = the ngoal rows of q_goals [ngoal][nplan] */
int mjpl_rrt_reset(mjpl_rrt *r, const double *q_init, const double *q_goals, int32_t ngoal, uint64_t seed);
/* one round: sample, extend the growing tree, extend the other towards what was reached, exchange.
 * Synchronises (the host needs the connection flag and the node counts).  request_stop != 0 (a
 * rank's time limit has passed) travels with the exchange, so that all ranks stop after the same
 * round: info->stop_requested.  An error that only one rank runs into (its slab of new nodes is
 * full, a HIP error) travels the same way: that rank still takes part in the exchange, and every
 * rank returns the error from the same round. */
int mjpl_rrt_round(mjpl_rrt *r, int32_t request_stop, mjpl_rrt_round_info *info);
/* The round split at its exchange step, for a launcher that moves the slabs itself (another
 * transport than RCCL; the tests, which run two ranks on one GPU): mjpl_rrt_round is
 *   round_begin -> all-gather of the headers -> all-gather of the slabs -> round_finish.
 * mjpl_rrt_set_world gives a planner its rank identity without a communicator (the sampler is keyed
 * by it, rrt.py:195-203 per rank); with mjpl_comm_init on the engine and no set_world the
 * communicator's rank / world are used.
 * round_begin: sample, extend, connect on this rank; synchronises; head[8] (host) receives this
 *   rank's exchange header: [0] new nodes of the tree that grew this round, [1] of the other tree,
 *   [2] connecting lane (INT32_MAX: none), [3] / [4] its node in the start / goal tree (>= 0: node id;
 *   < 0: -1 - index into this rank's slab), [5] stop request, [6] status of this rank's half
 *   (MJPL_OK, or the error every rank must return from round_finish), [7] 0.
 * round_slabs: device pointers of this rank's new nodes of pass p (0: the tree that grew, 1: the
 *   other): rows [head[p]][nplan] float64 and parents [head[p]] int32 (>= 0 node id, < 0 slab-relative).
 * round_finish: heads = all ranks' headers in rank order [world][8] (host); drows_all[p] /
 *   dparents_all[p] = device buffers holding every rank's slab of pass p, rank k at row offset
 *   k * stride_rows[p] (stride_rows[p] >= the largest head[p]; the pointers may be NULL when every
 *   count of the pass is 0).  Appends all slabs in rank order, resolves the winner (lowest rank),
 *   fills info.  The buffers must stay valid until the engine's stream has been synchronised. */
int mjpl_rrt_set_world(mjpl_rrt *r, int32_t rank, int32_t world);
int mjpl_rrt_round_begin(mjpl_rrt *r, int32_t request_stop, int32_t *head);
int mjpl_rrt_round_slabs(mjpl_rrt *r, int32_t pass, const void **drows, const void **dparents);
int mjpl_rrt_round_finish(mjpl_rrt *r, const int32_t *heads, const void *const *drows_all,
                          const void *const *dparents_all, const int32_t *stride_rows, mjpl_rrt_round_info *info);
/* after a round with connected = 1: the path q_init ... goal, rows [len][nplan] */
int mjpl_rrt_path(mjpl_rrt *r, double *path, int32_t maxlen, int32_t *len);
/* download a tree (0 = start, 1 = goal): rows Q [n][nplan] and parent ids (-1 = root); either may
 * be NULL; *n always receives the node count */
int mjpl_rrt_get_tree(mjpl_rrt *r, int32_t tree, double *Q, int32_t *parent, int64_t maxn, int64_t *n);
/* the targets [lanes][nplan] and participation flags of the most recent round (tests) */
int mjpl_rrt_get_lanes(mjpl_rrt *r, double *targets, uint8_t *on);

/* The exchange step runs on RCCL, called from this library on the engine's stream; a launcher only
 * ferries the 128-byte ncclUniqueId from rank 0 to the others (any transport).  Without a
 * communicator an engine is a world of one. */
int mjpl_comm_unique_id(void *id128);
int mjpl_comm_init(mjpl_engine *e, const void *id128, int32_t rank, int32_t world);
int mjpl_comm_destroy(mjpl_engine *e);
/* ncclAllGather of bytes_per_rank bytes per rank on the engine's stream (in place allowed:
 * dsend == drecv + rank * bytes_per_rank) */
int mjpl_allgather_dev(mjpl_engine *e, const void *dsend, void *drecv, size_t bytes_per_rank);

#ifdef __cplusplus
}
#endif
#endif
