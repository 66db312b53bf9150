/* srt_oracle_scattered.c -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of modelnum=4: scattered ln(N_s) samples in a kd-tree, moving-least-squares
 * interpolation (scattered_interp_dens_model_adapter.f95:63-235 setup, :249-312 density part of
 * funcPlasmaParams; kdtree_mod.f95; lsinterp_mod.f95:175-221, :244-449; blas.f95:40-62,96-137 ->
 * LAPACK dposv / BLAS dgemm, dgemv, restated for the tiny sizes used here).
 *
 * Not bit-comparable with the reference by construction: the reference inserts the points in
 * `randperm` order drawn from the compiler's RNG (scattered_..:137-165, util.f95:7-23; SURVEY A-12), so
 * its tree shape -- hence the ORDER in which neighbours are summed -- cannot be reproduced.  The set of
 * neighbours and every formula are the same; agreement is at rounding level times the conditioning
 * of the normal equations (checked in tests/test_oracle_vs_ref.py and the golden vectors).
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "srt_oracle.h"
#include "srt_oracle_internal.h"

static const double PI = 3.141592653589793238462643;
static const double R_E = 6371.2e3;

/* kdtree_add (kdtree_mod.f95:25-55), iterative on an index-linked array */
static void kd_add(struct so_scattered *sc, int idx) {
  kdnode *nd = sc->nodes;
  nd[idx].left = nd[idx].right = -1;
  if (sc->root < 0) {
    nd[idx].dim = 0;
    sc->root = idx;
    return;
  }
  int cur = sc->root, depth = 0;
  for (;;) {
    int d = nd[cur].dim;
    int *next = (nd[cur].p[d] < nd[idx].p[d]) ? &nd[cur].right : &nd[cur].left;
    depth++;
    if (*next < 0) {
      nd[idx].dim = depth % 3;
      *next = idx;
      return;
    }
    cur = *next;
  }
}

static double dist2(const double a[3], const double b[3]) {
  double d0 = a[0] - b[0], d1 = a[1] - b[1], d2 = a[2] - b[2];
  return (d0 * d0 + d1 * d1) + d2 * d2;
}

/* kdtree_nearest (kdtree_mod.f95:386-444) */
static void kd_nearest(const struct so_scattered *sc, int t, const double p[3], int excludeself, int *best) {
  if (t < 0) return;
  const kdnode *nd = sc->nodes;
  if (*best < 0) *best = t;
  double dist_here = dist2(nd[t].p, p), dist_best = dist2(nd[*best].p, p);
  if (dist_here < dist_best)
    if (!(dist_here == 0.0 && excludeself == 1)) *best = t;
  int d = nd[t].dim;
  int near_ = (nd[t].p[d] < p[d]) ? nd[t].right : nd[t].left;
  int far_ = (nd[t].p[d] < p[d]) ? nd[t].left : nd[t].right;
  kd_nearest(sc, near_, p, excludeself, best);
  dist_best = dist2(nd[*best].p, p);
  double da = (nd[t].p[d] - p[d]) * (nd[t].p[d] - p[d]);
  if (da < dist_best) kd_nearest(sc, far_, p, excludeself, best);
}
/* Note: like the reference, `best` starts as the first node visited (the root), which with excludeself=1
 * may be the query point itself at distance 0 and can then never be displaced; the adapter only calls it
 * that way for points already in the tree, where the root is the query for exactly one point. */

/* kdtree_search_core (kdtree_mod.f95:135-189): right, then left, then self; strict inequalities */
static void kd_search(struct so_scattered *sc, int t, const double p[3], double radius, int *count) {
  if (t < 0) return;
  const kdnode *nd = sc->nodes;
  int d = nd[t].dim;
  if (nd[t].p[d] < p[d] + radius) kd_search(sc, nd[t].right, p, radius, count);
  if (nd[t].p[d] > p[d] - radius) kd_search(sc, nd[t].left, p, radius, count);
  if (dist2(nd[t].p, p) < radius * radius) {
    if (*count == sc->cap) {
      sc->cap = sc->cap ? 2 * sc->cap : 1024;
      sc->found = (int *)realloc(sc->found, sizeof(int) * sc->cap);
    }
    sc->found[(*count)++] = t;
  }
}

/* tabular_monomials for 3 dimensions (lsinterp_mod.f95:70-99) */
#define SO_MAXORDER 6
#define SO_MAXJ 84 /* (6+3 choose 3) */
/* generate_monomials for 3 dimensions (lsinterp_mod.f95:114-164): every triple of 0..degree, last dimension fastest, kept when its
 * sum is <= degree -- the same sequence as the tables above for degrees 0..3 */
static int monomials3_general(int degree, int ex[SO_MAXJ][3]) {
  int J = 0;
  for (int a = 0; a <= degree; a++)
    for (int b = 0; b <= degree; b++)
      for (int c = 0; c <= degree; c++)
        if (a + b + c <= degree) {
          ex[J][0] = a;
          ex[J][1] = b;
          ex[J][2] = c;
          J++;
        }
  return J;
}
static int monomials3(int degree, int ex[SO_MAXJ][3]) {
  if (degree > 3) return monomials3_general(degree, ex); /* lsinterp_mod.f95:273-281 */
  static const int e1[3][20] = {{0}, {0}, {0}};
  (void)e1;
  static const int x0[1] = {0}, y0[1] = {0}, z0[1] = {0};
  static const int x1[4] = {0, 0, 0, 1}, y1[4] = {0, 0, 1, 0}, z1[4] = {0, 1, 0, 0};
  static const int x2[10] = {0, 0, 0, 0, 0, 0, 1, 1, 1, 2}, y2[10] = {0, 0, 0, 1, 1, 2, 0, 0, 1, 0},
                   z2[10] = {0, 1, 2, 0, 1, 0, 0, 1, 0, 0};
  static const int x3[20] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 2, 2, 2, 3},
                   y3[20] = {0, 0, 0, 0, 1, 1, 1, 2, 2, 3, 0, 0, 0, 1, 1, 2, 0, 0, 1, 0},
                   z3[20] = {0, 1, 2, 3, 0, 1, 2, 0, 1, 0, 0, 1, 2, 0, 1, 0, 0, 1, 0, 0};
  const int *xs, *ys, *zs;
  int J;
  switch (degree) {
  case 0: xs = x0; ys = y0; zs = z0; J = 1; break;
  case 1: xs = x1; ys = y1; zs = z1; J = 4; break;
  case 2: xs = x2; ys = y2; zs = z2; J = 10; break;
  default: xs = x3; ys = y3; zs = z3; J = 20; break;
  }
  for (int j = 0; j < J; j++) {
    ex[j][0] = xs[j];
    ex[j][1] = ys[j];
    ex[j][2] = zs[j];
  }
  return J;
}

/* etainv (lsinterp_mod.f95:175-209) */
static double etainv(double r, double radius, double hin, int exact) {
  const double eps = 5.0e-16;
  if (exact == 1) {
    double h = hin, s = radius;
    return ((1.0 + eps) / (exp((r / h) * (r / h)) - 1.0 + eps)) * (0.5 + 0.5 * cos(r * 2.0 * PI / s / 2.0));
  }
  double h = hin / 4.0, s = radius, alpha = 1.1;
  return exp(-pow((r + radius * eps) / h, alpha)) * (0.5 + 0.5 * cos(r * 2.0 * PI / s / 2.0));
}
/* coswindow (:215-221) */
static double coswindow(double r, double radius) { return 0.5 + 0.5 * cos(r * 2.0 * PI / radius / 2.0); }

static double ipow(double b, int e) {
  double r = 1.0;
  for (int i = 0; i < e; i++) r = r * b;
  return r;
}

/* dposv('U') for n <= 84: dpotf2 + dpotrs (LAPACK 3.2.1; dpotrf takes its unblocked path below the block size of 64 -- orders
 * 0..5; order 6 (n = 84) would run the blocked form there, whose sums are the same in another order), b overwritten by the solution */
static int dposv_upper(int n, double A[SO_MAXJ][SO_MAXJ], double b[SO_MAXJ]) {
  for (int j = 0; j < n; j++) {
    double s = 0.0;
    for (int l = 0; l < j; l++) s = s + A[l][j] * A[l][j];
    double ajj = A[j][j] - s;
    if (ajj <= 0.0 || ajj != ajj) return j + 1;
    ajj = sqrt(ajj);
    A[j][j] = ajj;
    for (int c = j + 1; c < n; c++) {
      double t = 0.0;
      for (int l = 0; l < j; l++) t = t + A[l][c] * A[l][j];
      A[j][c] = (A[j][c] - t) * (1.0 / ajj);
    }
  }
  /* U^T y = b */
  for (int i = 0; i < n; i++) {
    double t = b[i];
    for (int l = 0; l < i; l++) t = t - A[l][i] * b[l];
    b[i] = t / A[i][i];
  }
  /* U x = y */
  for (int i = n - 1; i >= 0; i--) {
    double t = b[i];
    for (int l = i + 1; l < n; l++) t = t - A[i][l] * b[l];
    b[i] = t / A[i][i];
  }
  return 0;
}

/* lsinterp (lsinterp_mod.f95:244-449), scaled = 0 */
static int lsinterp(struct so_scattered *sc, const double rin[3], double radius, double fi[4]) {
  int ex[SO_MAXJ][3];
  int J = monomials3(sc->order, ex);
  int I = 0;
  kd_search(sc, sc->root, rin, radius, &I);
  const kdnode *nd = sc->nodes;
  for (int s = 0; s < 4; s++) fi[s] = 0.0;
  if (I < J) return 2;
  double *r = (double *)malloc(sizeof(double) * I);
  double sw = 0.0, swv = 0.0;
  for (int i = 0; i < I; i++) {
    const double *q = nd[sc->found[i]].p;
    double acc = 0.0;
    for (int k = 0; k < 3; k++) acc = acc + (rin[k] - q[k]) * (rin[k] - q[k]);
    r[i] = sqrt(acc);
  }
  for (int i = 0; i < I; i++) {
    double cw = coswindow(r[i], radius);
    swv = swv + cw * nd[sc->found[i]].val[sc->nspec];
    sw = sw + cw;
  }
  double avgdist = swv / sw;
  double hin = sc->local_window_scale * avgdist;
  int *keep = (int *)malloc(sizeof(int) * I);
  int Ik = 0;
  for (int i = 0; i < I; i++)
    if (etainv(r[i], radius, hin, sc->exact) > 1.0e-16) keep[Ik++] = i;
  if (Ik < J) {
    Ik = 0;
    for (int i = 0; i < I; i++) keep[Ik++] = i;
  }
  double *dinv = (double *)malloc(sizeof(double) * Ik);
  double *E = (double *)malloc(sizeof(double) * Ik * J);
  for (int i = 0; i < Ik; i++) dinv[i] = sqrt(0.5 * etainv(r[keep[i]], radius, hin, sc->exact));
  for (int j = 0; j < J; j++)
    for (int i = 0; i < Ik; i++) {
      const double *q = nd[sc->found[keep[i]]].p;
      double e = 1.0;
      for (int k = 0; k < 3; k++)
        if (ex[j][k] != 0) e = e * ipow(q[k] - rin[k], ex[j][k]);
      E[(size_t)j * Ik + i] = dinv[i] * e;
    }
  static __thread double A[SO_MAXJ][SO_MAXJ];
  double c[SO_MAXJ];
  for (int a = 0; a < J; a++)
    for (int b = 0; b < J; b++) {
      double t = 0.0;
      for (int l = 0; l < Ik; l++) t = t + E[(size_t)a * Ik + l] * E[(size_t)b * Ik + l];
      A[a][b] = t;
    }
  for (int j = 0; j < J; j++) c[j] = 0.0;
  c[0] = 1.0;
  int info = dposv_upper(J, A, c);
  int status = 0;
  if (info != 0) {
    status = 1;
  } else {
    /* tmp = E*aa (dgemv 'N', column sweep); aa = tmp*dinv; fi = dot(aa, vals) */
    for (int s = 0; s < sc->nspec; s++) {
      double acc = 0.0;
      for (int i = 0; i < Ik; i++) {
        double tmp = 0.0;
        for (int j = 0; j < J; j++) tmp = tmp + c[j] * E[(size_t)j * Ik + i];
        acc = acc + (tmp * dinv[i]) * nd[sc->found[keep[i]]].val[s];
      }
      fi[s] = acc;
    }
  }
  free(r);
  free(keep);
  free(dinv);
  free(E);
  return status;
}

void so_scattered_params(struct so_model *m, const double x[3], double qs[4], double Ns[4], double ms[4],
                         double nus[4]) {
  struct so_scattered *sc = m->sc;
  for (int s = 0; s < 4; s++) qs[s] = Ns[s] = ms[s] = nus[s] = 0.0;
  double fi[4];
  if (((x[0] * x[0] + x[1] * x[1]) + x[2] * x[2]) > R_E * R_E) {
    lsinterp(sc, x, sc->maxnearest * sc->window_scale, fi);
    for (int s = 0; s < sc->nspec; s++) Ns[s] = exp(fi[s]); /* status 1/2 -> fi = 0 -> Ns = 1 */
  }
  for (int s = 0; s < sc->nspec; s++) {
    qs[s] = sc->qs[s];
    ms[s] = sc->ms[s];
  }
}

void so_scattered_free(struct so_model *m) {
  if (!m->sc) return;
  free(m->sc->nodes);
  free(m->sc->found);
  free(m->sc);
  m->sc = NULL;
}

static int read_numbers(FILE *f, int n, double *out) {
  for (int i = 0; i < n; i++)
    if (fscanf(f, " %lf", &out[i]) != 1) {
      int ch = fgetc(f);
      if (ch == ',') {
        i--;
        continue;
      }
      return i;
    }
  return n;
}

/* setup (scattered_interp_dens_model_adapter.f95:63-235) */
so_model *so_model_create_scattered_file(const char *ptsfile, int yearday, int msec, double window_scale,
                                         int order, int exact, double local_window_scale, unsigned perm_seed) {
  FILE *f = fopen(ptsfile, "r");
  if (!f) return NULL;
  double hdr[7];
  if (read_numbers(f, 7, hdr) != 7) {
    fclose(f);
    return NULL;
  }
  int nspec = (int)hdr[0];
  if (nspec < 1 || nspec > SO_MAXSPEC || order < 0 || order > SO_MAXORDER) {
    fclose(f);
    return NULL;
  }
  struct so_scattered *sc = (struct so_scattered *)calloc(1, sizeof *sc);
  sc->nspec = nspec;
  sc->window_scale = window_scale;
  sc->local_window_scale = local_window_scale;
  sc->order = order;
  sc->exact = exact;
  sc->root = -1;
  read_numbers(f, nspec, sc->qs);
  read_numbers(f, nspec, sc->ms);
  int cap = 65536, n = 0;
  double *raw = (double *)malloc(sizeof(double) * cap * (3 + nspec));
  double row[8];
  while (read_numbers(f, 3 + nspec, row) == 3 + nspec) {
    if (n == cap) {
      cap *= 2;
      raw = (double *)realloc(raw, sizeof(double) * cap * (3 + nspec));
    }
    memcpy(raw + (size_t)n * (3 + nspec), row, sizeof(double) * (3 + nspec));
    n++;
  }
  fclose(f);
  /* randperm (util.f95:7-23) with our own generator */
  int *idx = (int *)malloc(sizeof(int) * n);
  for (int i = 0; i < n; i++) idx[i] = i;
  unsigned long long st = 0x9E3779B97F4A7C15ull ^ perm_seed;
  for (int k = n; k >= 2; k--) {
    st = st * 6364136223846793005ull + 1442695040888963407ull;
    double u = (double)(st >> 11) / 9007199254740992.0;
    int j = (int)floor((double)k * u);
    int t = idx[j];
    idx[j] = idx[k - 1];
    idx[k - 1] = t;
  }
  sc->nodes = (kdnode *)malloc(sizeof(kdnode) * (n ? n : 1));
  int *node_of = (int *)malloc(sizeof(int) * (n ? n : 1));
  for (int j = 0; j < n; j++) {
    int i = idx[j];
    const double *src = raw + (size_t)i * (3 + nspec);
    node_of[i] = -1;
    int best = -1;
    kd_nearest(sc, sc->root, src, 0, &best);
    if (best >= 0 && dist2(sc->nodes[best].p, src) == 0.0) continue; /* duplicate: ignored (:160-163) */
    kdnode *nd = &sc->nodes[sc->n];
    memcpy(nd->p, src, sizeof(double) * 3);
    for (int s = 0; s < nspec; s++) nd->val[s] = src[3 + s];
    nd->val[nspec] = 1.0; /* placeholder for the nearest-neighbour distance (:152) */
    node_of[i] = sc->n;
    kd_add(sc, sc->n);
    sc->n++;
  }
  /* nearest-neighbour distance of every sample outside the Earth (:167-203) */
  sc->maxnearest = 0.0;
  for (int i = 0; i < n; i++) {
    const double *src = raw + (size_t)i * (3 + nspec);
    if ((src[0] * src[0] + src[1] * src[1]) + src[2] * src[2] >= R_E * R_E) {
      int best = -1;
      kd_nearest(sc, sc->root, src, 1, &best);
      double dd[3] = {src[0] - sc->nodes[best].p[0], src[1] - sc->nodes[best].p[1], src[2] - sc->nodes[best].p[2]};
      double dist = sqrt((dd[0] * dd[0] + dd[1] * dd[1]) + dd[2] * dd[2]);
      /* kdtree_find_ptr: the node holding exactly this point (first inserted copy for duplicates) */
      int nodei = node_of[i];
      if (nodei < 0) {
        int b2 = -1;
        kd_nearest(sc, sc->root, src, 0, &b2);
        nodei = b2;
      }
      sc->nodes[nodei].val[nspec] = dist;
      if (dist > sc->maxnearest) sc->maxnearest = dist;
    }
  }
  /* The reference's kdtree_nearest seeds `best` with the root; for the ONE sample that is the root this makes
     "nearest other sample" = itself (distance 0).  Which sample that is depends on the reference's RNG, so
     it cannot be matched; perm_seed with bit 31 set stores the true distance there instead (what the HIP
     path does for every sample). */
  if ((perm_seed & 0x80000000u) && sc->root >= 0) {
    const double *rp = sc->nodes[sc->root].p;
    if ((rp[0] * rp[0] + rp[1] * rp[1]) + rp[2] * rp[2] >= R_E * R_E) {
      double bestd = -1.0;
      for (int i = 0; i < sc->n; i++) {
        if (i == sc->root) continue;
        double d2 = dist2(sc->nodes[i].p, rp);
        if (bestd < 0 || d2 < bestd) bestd = d2;
      }
      if (bestd >= 0) {
        double dd = sqrt(bestd);
        sc->nodes[sc->root].val[nspec] = dd;
        if (dd > sc->maxnearest) sc->maxnearest = dd;
      }
    }
  }
  free(raw);
  free(idx);
  free(node_of);
  so_model *m = (so_model *)calloc(1, sizeof *m);
  m->kind = 4;
  m->nspec = nspec;
  m->sc = sc;
  so_dipole_tilt(yearday, msec, &m->mu);
  return m;
}

/* Test helper: overwrite the stored nearest-sample distance of the sample at exactly p (val[nspec]) -- e.g. 0 for the
 * sample that is the root of the REFERENCE's kd-tree (it seeds kdtree_nearest's `best` with the root, so the root's own
 * "nearest other sample" is itself; which sample that is comes from the reference build, tests/golden/
 * scattered_o3_golden.npz: ref_root_index).  Returns the node index, or -1 if no sample sits at p. */
int so_scattered_set_spacing(so_model *m, const double p[3], double value) {
  if (!m || m->kind != 4) return -1;
  struct so_scattered *sc = m->sc;
  for (int i = 0; i < sc->n; i++)
    if (sc->nodes[i].p[0] == p[0] && sc->nodes[i].p[1] == p[1] && sc->nodes[i].p[2] == p[2]) {
      sc->nodes[i].val[sc->nspec] = value;
      return i;
    }
  return -1;
}
double so_scattered_radius(const so_model *m) { return (m && m->kind == 4) ? m->sc->maxnearest * m->sc->window_scale : 0.0; }
