/* mjpl_cpu_ref.c -- the CPU oracle behind the SAME C ABI as libmjpl_hip.so.
 *
 * TEST INFRASTRUCTURE, like everything under oracle/: only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library.  SURVEY.md section 8b asks for "identical entry
 * points in the cpu_ref library (same header) for oracle + baseline": this file implements the
 * host-pointer entry points of include/mjpl_hip.h for rows a1 / a9 / a3 -- mjpl_create / destroy /
 * set_planning / check_configs / check_edges / fk / last_error / version -- over the float64
 * restatement in mjpl_oracle.c, so that one binding (and one set of known-answer tests) runs against
 * either library.  It deliberately exports nothing else: mjpl_amd.engine.load_library() binds every
 * symbol the header declares and therefore refuses this library -- the product cannot run on it.
 *
 * What each entry point follows is cited in include/mjpl_hip.h and in mjpl_oracle.h. */
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/mjpl_hip.h"
#include "mjpl_oracle.h"

struct mjpl_engine {
  orc_model m;          /* arrays below are owned copies */
  void *owned[32];
  int nowned;
  int32_t *allowed;     /* [nallowed*2], each pair sorted (collision_constraint.py:60-64) */
  int32_t nallowed;
  int32_t *qidx;
  int32_t nplan;
  double *qbase;
  int32_t nthreads;
};

static __thread char g_err[512];

static int fail(int code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

const char *mjpl_last_error(void) { return g_err; }
const char *mjpl_version(void) { return "mjpl cpu_ref (float64 oracle behind include/mjpl_hip.h)"; }

static void *keep(mjpl_engine *e, const void *src, size_t bytes) {
  void *p = malloc(bytes ? bytes : 1);
  if (p && src && bytes) memcpy(p, src, bytes);
  if (p && e->nowned < 32) e->owned[e->nowned++] = p;
  return p;
}

static int map_status(int st) {
  switch (st) {
    case ORC_OK: return MJPL_OK;
    case ORC_E_JOINT: return fail(MJPL_E_JOINT, "joint type outside {slide, hinge}");
    case ORC_E_PAIRTYPE: return fail(MJPL_E_PAIRTYPE, "a colliding geom pair has no primitive narrowphase routine");
    case ORC_E_NONFINITE: return fail(MJPL_E_NONFINITE, "non-finite edge");
    default: return fail(MJPL_E_CAPACITY, "oracle status %d", st);
  }
}

int mjpl_create(const mjpl_model_desc *d, const int32_t *allowed_bodies, int32_t nallowed, int32_t device,
                mjpl_engine **out) {
  (void)device;
  if (!d || !out || nallowed < 0 || (nallowed && !allowed_bodies)) return fail(MJPL_E_ARG, "mjpl_create: bad argument");
  mjpl_engine *e = (mjpl_engine *)calloc(1, sizeof(*e));
  if (!e) return fail(MJPL_E_CAPACITY, "out of memory");
  orc_model *m = &e->m;
  m->nq = d->nq; m->njnt = d->njnt; m->nbody = d->nbody; m->ngeom = d->ngeom;
#define KEEP_I(f, n) m->f = (const int32_t *)keep(e, d->f, sizeof(int32_t) * (size_t)(n))
#define KEEP_D(f, n) m->f = (const double *)keep(e, d->f, sizeof(double) * (size_t)(n))
  KEEP_I(body_parentid, d->nbody); KEEP_I(body_weldid, d->nbody); KEEP_I(body_jntadr, d->nbody);
  KEEP_I(body_jntnum, d->nbody); KEEP_D(body_pos, 3 * d->nbody); KEEP_D(body_quat, 4 * d->nbody);
  KEEP_I(jnt_type, d->njnt); KEEP_I(jnt_qposadr, d->njnt); KEEP_D(jnt_axis, 3 * d->njnt);
  KEEP_D(jnt_pos, 3 * d->njnt); KEEP_D(qpos0, d->nq);
  KEEP_I(geom_type, d->ngeom); KEEP_I(geom_bodyid, d->ngeom); KEEP_I(geom_contype, d->ngeom);
  KEEP_I(geom_conaffinity, d->ngeom); KEEP_D(geom_size, 3 * d->ngeom); KEEP_D(geom_pos, 3 * d->ngeom);
  KEEP_D(geom_quat, 4 * d->ngeom); KEEP_D(geom_rbound, d->ngeom); KEEP_D(geom_margin, d->ngeom);
#undef KEEP_I
#undef KEEP_D
  e->allowed = (int32_t *)keep(e, allowed_bodies, sizeof(int32_t) * 2 * (size_t)nallowed);
  e->nallowed = nallowed;
  for (int a = 0; a < nallowed; a++) {
    int32_t *p = e->allowed + 2 * a;
    if (p[0] < 0 || p[1] < 0 || p[0] >= d->nbody || p[1] >= d->nbody) {
      mjpl_destroy(e);
      return fail(MJPL_E_ARG, "allowed body pair %d out of range", a);
    }
    if (p[0] > p[1]) { const int32_t t = p[0]; p[0] = p[1]; p[1] = t; }
  }
  e->nplan = d->nq;
  e->qidx = (int32_t *)malloc(sizeof(int32_t) * (size_t)(d->nq ? d->nq : 1));
  e->qbase = (double *)malloc(sizeof(double) * (size_t)(d->nq ? d->nq : 1));
  for (int k = 0; k < d->nq; k++) { e->qidx[k] = k; e->qbase[k] = d->qpos0[k]; }
  const char *t = getenv("MJPL_CPU_REF_THREADS");
  e->nthreads = t ? atoi(t) : 4;
  if (e->nthreads < 1) e->nthreads = 1;
  *out = e;
  return MJPL_OK;
}

void mjpl_destroy(mjpl_engine *e) {
  if (!e) return;
  for (int k = 0; k < e->nowned; k++) free(e->owned[k]);
  free(e->qidx);
  free(e->qbase);
  free(e);
}

int mjpl_set_planning(mjpl_engine *e, const int32_t *qidx, int32_t nplan, const double *qpos_base) {
  if (!e || !qidx || !qpos_base || nplan < 1 || nplan > e->m.nq) return fail(MJPL_E_ARG, "mjpl_set_planning: bad argument");
  for (int k = 0; k < nplan; k++)
    if (qidx[k] < 0 || qidx[k] >= e->m.nq) return fail(MJPL_E_ARG, "mjpl_set_planning: qpos index out of range");
  memcpy(e->qidx, qidx, sizeof(int32_t) * (size_t)nplan);
  memcpy(e->qbase, qpos_base, sizeof(double) * (size_t)e->m.nq);
  e->nplan = nplan;
  return MJPL_OK;
}

static orc_batch batch_of(const mjpl_engine *e, int32_t layout) {
  orc_batch b;
  b.qpos_base = e->qbase;
  b.qidx = e->qidx;
  b.nplan = e->nplan;
  b.layout = layout == MJPL_SOA ? 0 : 1;
  b.allowed = e->allowed;
  b.nallowed = e->nallowed;
  return b;
}

int mjpl_check_configs(mjpl_engine *e, const double *Q, int64_t N, int32_t layout, uint8_t *valid) {
  if (!e || N < 0 || (N && (!Q || !valid))) return fail(MJPL_E_ARG, "mjpl_check_configs: bad argument");
  if (N == 0) return MJPL_OK;
  const orc_batch b = batch_of(e, layout);
  return map_status(orc_valid_configs(&e->m, &b, Q, N, e->nthreads, valid));
}

int mjpl_check_edges(mjpl_engine *e, const double *QA, const double *QB, int64_t E, double step_dist, int32_t layout,
                     int32_t flags, uint8_t *valid, int32_t *first_bad) {
  if (!e || E < 0 || (E && (!QA || !QB || !valid))) return fail(MJPL_E_ARG, "mjpl_check_edges: bad argument");
  if (!(step_dist > 0.0)) return fail(MJPL_E_ARG, "`step_dist` must be > 0");  /* utils.py:207-208 */
  if (E == 0) return MJPL_OK;
  const orc_batch b = batch_of(e, layout);
  if (!(flags & MJPL_EDGE_INTERIOR_ONLY))
    return map_status(orc_valid_edges(&e->m, &b, QA, QB, E, step_dist, e->nthreads, valid, first_bad, NULL));
  /* _valid_collision_interval alone (utils.py:188-216), edge by edge on full qpos vectors */
  double *a = (double *)malloc(sizeof(double) * 2 * (size_t)e->m.nq), *q = a + e->m.nq;
  int worst = MJPL_OK;
  for (int64_t i = 0; i < E; i++) {
    memcpy(a, e->qbase, sizeof(double) * (size_t)e->m.nq);
    memcpy(q, e->qbase, sizeof(double) * (size_t)e->m.nq);
    for (int c = 0; c < e->nplan; c++) {
      const int64_t at = layout == MJPL_SOA ? (int64_t)c * E + i : i * e->nplan + c;
      a[e->qidx[c]] = QA[at];
      q[e->qidx[c]] = QB[at];
    }
    int32_t fb = 0;
    const int v = orc_valid_collision_interval(&e->m, e->allowed, e->nallowed, a, q, step_dist, NULL, &fb);
    if (v < 0) {
      valid[i] = 0;
      if (first_bad) first_bad[i] = -2;
      worst = map_status(v);
      continue;
    }
    valid[i] = (uint8_t)v;
    if (first_bad) first_bad[i] = v ? -1 : fb;  /* (1-based interior index = the check index) */
  }
  free(a);
  return worst;
}

int mjpl_fk(mjpl_engine *e, const double *Q, int64_t N, int32_t layout, double *xpos, double *xquat, double *geom_xpos,
            double *geom_xmat) {
  if (!e || N < 0 || (N && !Q)) return fail(MJPL_E_ARG, "mjpl_fk: bad argument");
  if (N == 0) return MJPL_OK;
  const orc_batch b = batch_of(e, layout);
  /* the oracle writes all four outputs: give it scratch for the ones the caller does not want */
  double *sx = xpos ? NULL : (double *)malloc(sizeof(double) * (size_t)N * 3 * (size_t)e->m.nbody);
  double *sq = xquat ? NULL : (double *)malloc(sizeof(double) * (size_t)N * 4 * (size_t)e->m.nbody);
  double *gp = geom_xpos ? NULL : (double *)malloc(sizeof(double) * (size_t)N * 3 * (size_t)e->m.ngeom);
  double *gm = geom_xmat ? NULL : (double *)malloc(sizeof(double) * (size_t)N * 9 * (size_t)e->m.ngeom);
  const int st = orc_fk_batch(&e->m, &b, Q, N, xpos ? xpos : sx, xquat ? xquat : sq, geom_xpos ? geom_xpos : gp,
                              geom_xmat ? geom_xmat : gm);
  free(sx); free(sq); free(gp); free(gm);
  return map_status(st);
}
