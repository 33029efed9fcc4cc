/* capsule_box_mj.c -- TEST / ANALYSIS INFRASTRUCTURE, not part of the product and not the oracle's verdict.
 *
 * The oracle's capsule-box routine (mjpl_oracle.c: capsule_box) is a declared deviation from MuJoCo's
 * mjc_CapsuleBox: it minimises the distance between the capsule's segment and the box EXACTLY and runs the
 * sphere-box verdict at the minimiser, where upstream searches a finite set of closest-FEATURE candidates and
 * runs its sphere-box routine at the best one.  This file holds a second routine that follows the published
 * STRUCTURE of mjc_CapsuleBox as closely as it can be stated without its source [MJ-recalled:
 * engine_collision_box.c]:
 *   1. the capsule's centre and half axis are moved into the box frame;
 *   2. candidates for the point of the segment nearest to the box:
 *      (a) either END of the segment, kept only if at most ONE of its coordinates lies outside the box (the end
 *          is over a face, or inside): squared distance to its clamp;
 *      (b) each of the TWELVE box edges against the segment: the closest points of two segments by the 2x2
 *          system mjc_CapsuleCapsule solves (skipped when |det| < mjMINVAL = 1e-15: parallel), both parameters
 *          clamped to [-1, 1] with the re-projection of the clamped one's partner;
 *      the candidate with the smallest squared distance gives the segment parameter;
 *   3. mjraw_SphereBox at that point of the axis with the capsule's radius; contact iff its distance <= margin.
 *      (Upstream may add a SECOND sphere further along the axis so that a capsule lying on a face gets two
 *      contacts.  Every point of the axis is a point of the capsule, so a second sphere can only report a contact
 *      the exact routine reports as well; it is left out here, which can only make this routine report FEWER
 *      contacts than upstream, i.e. overstate the disagreement measured below.)
 *
 * orc_capsule_box_compare runs both on the same poses and returns both verdicts, so that
 * tools/capsule_box_deviation.py can count how often, and in which direction, they differ -- the bound DESIGN.md
 * section 4 states on how many verdicts of the headline batch could differ from MuJoCo's because of this routine.
 * Since the exact minimum is never farther than any candidate, "structured: contact, exact: free" cannot happen up
 * to rounding; what is measured is "exact: contact, structured: free".
 */
#include "mjpl_oracle.c"

static double seg_box_clamp_d2(const double *e, const double *s, int *nout) {
  double d2 = 0;
  int n = 0;
  for (int k = 0; k < 3; k++) {
    double c = e[k];
    if (c < -s[k]) { c = -s[k]; n++; }
    else if (c > s[k]) { c = s[k]; n++; }
    d2 = d2 + (e[k] - c) * (e[k] - c);
  }
  *nout = n;
  return d2;
}

/* verdict of the structured routine; *tbest receives the segment parameter it tests at */
static int capsule_box_mjstruct(double margin, const double *pos1, const double *mat1, const double *size1,
                                const double *pos2, const double *mat2, const double *size2, double *tbest) {
  double tmp[3] = {pos1[0] - pos2[0], pos1[1] - pos2[1], pos1[2] - pos2[2]};
  double p[3], a[3], h[3];
  double axis[3] = {mat1[2], mat1[5], mat1[8]};
  mul_matT_vec3(p, mat2, tmp);
  mul_matT_vec3(a, mat2, axis);
  for (int k = 0; k < 3; k++) h[k] = a[k] * size1[1];

  double best = 1e300, x = 0;
  /* (a) the ends of the segment, where they are over a face or inside */
  for (int sgn = -1; sgn <= 1; sgn += 2) {
    double e[3] = {p[0] + sgn * h[0], p[1] + sgn * h[1], p[2] + sgn * h[2]};
    int nout;
    double d2 = seg_box_clamp_d2(e, size2, &nout);
    if (nout <= 1 && d2 < best) { best = d2; x = sgn; }
  }
  /* (b) the twelve edges: edge along box axis `ax` through the corner (s1, s2) of the other two */
  for (int ax = 0; ax < 3; ax++) {
    const int a1 = (ax + 1) % 3, a2 = (ax + 2) % 3;
    for (int s1 = -1; s1 <= 1; s1 += 2)
      for (int s2 = -1; s2 <= 1; s2 += 2) {
        double ce[3] = {0, 0, 0}, he[3] = {0, 0, 0};
        ce[a1] = s1 * size2[a1];
        ce[a2] = s2 * size2[a2];
        he[ax] = size2[ax];
        /* closest points of segment p + x1 h and edge ce + x2 he (the statements of mjc_CapsuleCapsule's
         * non-parallel branch: dif = p - ce; ma = h.h; mb = -h.he; mc = he.he; u = -h.dif; v = he.dif) */
        double dif[3] = {p[0] - ce[0], p[1] - ce[1], p[2] - ce[2]};
        double ma = h[0] * h[0] + h[1] * h[1] + h[2] * h[2];
        double mb = -(h[0] * he[0] + h[1] * he[1] + h[2] * he[2]);
        double mc = he[0] * he[0] + he[1] * he[1] + he[2] * he[2];
        double u = -(h[0] * dif[0] + h[1] * dif[1] + h[2] * dif[2]);
        double v = he[0] * dif[0] + he[1] * dif[1] + he[2] * dif[2];
        double det = ma * mc - mb * mb;
        if (fabs(det) < 1e-15) continue;
        double x1 = (mc * u - mb * v) / det, x2 = (ma * v - mb * u) / det;
        if (x1 > 1) { x1 = 1; x2 = (v - mb) / mc; }
        else if (x1 < -1) { x1 = -1; x2 = (v + mb) / mc; }
        if (x2 > 1) { x2 = 1; x1 = clipd((u - mb) / ma, -1, 1); }
        else if (x2 < -1) { x2 = -1; x1 = clipd((u + mb) / ma, -1, 1); }
        double d2 = 0;
        for (int k = 0; k < 3; k++) {
          double d = (p[k] + x1 * h[k]) - (ce[k] + x2 * he[k]);
          d2 = d2 + d * d;
        }
        if (d2 < best) { best = d2; x = x1; }
      }
  }
  double c[3] = {p[0] + x * h[0], p[1] + x * h[1], p[2] + x * h[2]};
  if (tbest) *tbest = x;
  return sphere_box_local(margin, c, size1[0], size2);
}

/* n pose pairs: capsule (pos [3], mat [9] row-major, size [2] = radius, half length), box (pos, mat, half sizes [3]).
 * v_exact / v_struct receive the two verdicts (1 = contact). */
int orc_capsule_box_compare(long n, const double *cpos, const double *cmat, const double *csize, const double *bpos,
                            const double *bmat, const double *bsize, double margin, unsigned char *v_exact,
                            unsigned char *v_struct, double *t_struct) {
  for (long i = 0; i < n; i++) {
    double t = 0;
    v_exact[i] = (unsigned char)(capsule_box(margin, cpos + 3 * i, cmat + 9 * i, csize + 2 * i, bpos + 3 * i, bmat + 9 * i,
                                             bsize + 3 * i) != 0);
    v_struct[i] = (unsigned char)(capsule_box_mjstruct(margin, cpos + 3 * i, cmat + 9 * i, csize + 2 * i, bpos + 3 * i,
                                                        bmat + 9 * i, bsize + 3 * i, &t) != 0);
    if (t_struct) t_struct[i] = t;
  }
  return 0;
}

/* signed distance-like quantity of the exact routine, for placing poses by bisection: distance of the segment's
 * nearest point to the box minus the radius (negative inside) */
double orc_capsule_box_gap(const double *cpos, const double *cmat, const double *csize, const double *bpos,
                           const double *bmat, const double *bsize) {
  /* dense scan of the convex function along the segment, then golden-section refinement */
  double tmp[3] = {cpos[0] - bpos[0], cpos[1] - bpos[1], cpos[2] - bpos[2]};
  double p[3], a[3];
  double axis[3] = {cmat[2], cmat[5], cmat[8]};
  mul_matT_vec3(p, bmat, tmp);
  mul_matT_vec3(a, bmat, axis);
  double lo = -1, hi = 1;
  for (int it = 0; it < 200; it++) {
    double m1 = lo + (hi - lo) * 0.381966011250105, m2 = lo + (hi - lo) * 0.618033988749895;
    double e1[3], e2[3];
    int dummy;
    for (int k = 0; k < 3; k++) { e1[k] = p[k] + m1 * a[k] * csize[1]; e2[k] = p[k] + m2 * a[k] * csize[1]; }
    if (seg_box_clamp_d2(e1, bsize, &dummy) <= seg_box_clamp_d2(e2, bsize, &dummy)) hi = m2; else lo = m1;
  }
  double e[3];
  int dummy;
  double t = 0.5 * (lo + hi);
  for (int k = 0; k < 3; k++) e[k] = p[k] + t * a[k] * csize[1];
  return sqrt(seg_box_clamp_d2(e, bsize, &dummy)) - csize[0];
}

void orc_capsule_box_gap_batch(long n, const double *cpos, const double *cmat, const double *csize, const double *bpos,
                               const double *bmat, const double *bsize, double *gap) {
  for (long i = 0; i < n; i++)
    gap[i] = orc_capsule_box_gap(cpos + 3 * i, cmat + 9 * i, csize + 2 * i, bpos + 3 * i, bmat + 9 * i, bsize + 3 * i);
}
