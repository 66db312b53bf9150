/* srt_oracle_sampler.c -- TEST INFRASTRUCTURE ONLY (see srt_oracle.h).
 *
 * CPU restatement of the reference's random / adaptive sample-set builder:
 *   fortran/gcpm_dens_model_buildgrid_random.f95:228-407   stages (radial, uniform, adaptive, zero altitude, iri shell)
 *   fortran/randomsampling_mod.f95:27-200                  recursivesampler, DEPTH-FIRST exactly as written there
 *   fortran/kdtree_mod.f95:203-304                         kdtree_search_rect == strict box test (done by brute force)
 *   fortran/gcpm_dens_model_buildgrid_random_helpermod.f95:28-46   f(x) = log(Ns)
 *   fortran/util.f95:26-49                                 normal()
 * with any oracle model in place of GCPM.
 *
 * Parity status of THIS file: the control flow is the reference's; the random numbers are not and cannot be (the
 * reference seeds random_number from the clock, `init_random_seed`, so no two runs of it agree).  The uniforms here
 * are counter-based and keyed by (stage, sample/pass, half-box, draw) -- the same keying the device builder uses --
 * so that the depth-first order of this file and the level-by-level order of the device produce the same SET of
 * samples, which is what tests/test_gpu_sampler.py checks.  "Parity unpinned" against the reference's own output
 * for that reason; f(x) itself is the pinned so_plasma_params.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "srt_oracle.h"

#define R_E 6371.2e3 /* constants.f95:8 */

static uint64_t mix(uint64_t z) {
  z += 0x9e3779b97f4a7c15ULL;
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
  return z ^ (z >> 31);
}
static double uniform(uint64_t seed, uint64_t stream, uint64_t a, uint64_t b, uint64_t c) {
  uint64_t h = mix(seed + stream);
  h = mix(h ^ a);
  h = mix(h ^ b);
  h = mix(h ^ c);
  return (double)(h >> 11) * 0x1.0p-53;
}
/* util.f95:26-49 */
static double normal(uint64_t seed, uint64_t stream, uint64_t a, uint64_t b, uint64_t c0) {
  for (unsigned t = 0;; ++t) {
    double u = 2.0 * uniform(seed, stream, a, b, c0 + 2 * t) - 1.0;
    double v = 2.0 * uniform(seed, stream, a, b, c0 + 2 * t + 1) - 1.0;
    double r = u * u + v * v;
    if (r <= 0.0 || r > 1.0) continue;
    return u * sqrt(-2.0 * log(r) / r);
  }
}

typedef struct {
  so_model *m;
  int nspec;
  uint64_t seed;
  double *rec; /* [n][3+4] */
  long n, cap;
  long nsamples; /* helpermod's counter */
} pool_t;

static void pool_add(pool_t *P, const double x[3]) { /* f(x) then kdtree_add */
  if (P->n == P->cap) {
    P->cap = P->cap ? 2 * P->cap : 1024;
    P->rec = (double *)realloc(P->rec, (size_t)P->cap * 7 * sizeof(double));
  }
  double qs[4], Ns[4], ms[4], nus[4], B0[3];
  so_plasma_params(P->m, x, qs, Ns, ms, nus, B0);
  double *r = P->rec + (size_t)P->n * 7;
  r[0] = x[0];
  r[1] = x[1];
  r[2] = x[2];
  for (int s = 0; s < 4; ++s) r[3 + s] = s < P->nspec ? log(Ns[s]) : 0.0;
  P->n++;
  P->nsamples++;
}

static void shell_point(uint64_t seed, uint64_t stream, uint64_t i, uint64_t try_, double rmin, double rmax, double p[3]) {
  double xd = normal(seed, stream, i, try_, 0);
  double yd = normal(seed, stream, i, try_, 1ULL << 20);
  double zd = normal(seed, stream, i, try_, 2ULL << 20);
  double nrm = sqrt(xd * xd + yd * yd + zd * zd);
  xd = xd / nrm;
  yd = yd / nrm;
  zd = zd / nrm;
  double r = uniform(seed, stream, i, try_, 3ULL << 20);
  r = rmin + (rmax - rmin) * r;
  p[0] = r * xd;
  p[1] = r * yd;
  p[2] = r * zd;
}
static int inside(const double p[3], const double lo[3], const double hi[3]) {
  return p[0] > lo[0] && p[0] < hi[0] && p[1] > lo[1] && p[1] < hi[1] && p[2] > lo[2] && p[2] < hi[2];
}
/* Deliberate divergence shared with the device builder: a sample whose ln N is not finite (log(0) inside the Earth for
 * the in-scope models) is left out of the box statistics -- in the reference one such sample turns the variance of every
 * enclosing box into NaN, and `NaN > alpha` being false ends all refinement from the root down. */
static int counted(const pool_t *P, const double *r, const double rlo[3], const double rhi[3]) {
  if (!inside(r, rlo, rhi)) return 0;
  for (int s = 0; s < P->nspec; ++s)
    if (!isfinite(r[3 + s])) return 0;
  return 1;
}

/* one side of randomsampling_mod.f95:76-134 (lower) / :136-193 (upper) */
static void recursivesampler(pool_t *P, const double limit_min[3], const double limit_max[3], double alpha, int depth,
                             int maxdepth, int numincrease, uint64_t pass, uint64_t callid);

static void one_side(pool_t *P, const double lmin[3], const double lmax[3], double alpha, int depth, int maxdepth,
                     int numincrease, uint64_t pass, uint64_t hid) {
  double center[3], lower[3], upper[3], rlo[3], rhi[3];
  for (int c = 0; c < 3; ++c) {
    center[c] = lmin[c] + 0.5 * (lmax[c] - lmin[c]);
    lower[c] = center[c] - lmin[c];
    upper[c] = lmax[c] - center[c];
    rlo[c] = center[c] - lower[c];
    rhi[c] = center[c] + upper[c];
  }
  long cnt = 0;
  for (long i = 0; i < P->n; ++i) cnt += counted(P, P->rec + (size_t)i * 7, rlo, rhi);
  int j = 0; /* draw counter of this half */
  if (cnt <= 2) {
    for (int i = 0; i < numincrease; ++i, ++j) {
      double rt[3];
      for (int c = 0; c < 3; ++c) rt[c] = uniform(P->seed, 3, pass, hid, (uint64_t)(3 * j + c)) * (lmax[c] - lmin[c]) + lmin[c];
      pool_add(P, rt);
    }
  } else {
    j = numincrease; /* the second batch always uses draws numincrease.. (the device evaluates both batches up front) */
  }
  /* search again, mean, variance */
  cnt = 0;
  double sum[4] = {0, 0, 0, 0};
  for (long i = 0; i < P->n; ++i) {
    const double *r = P->rec + (size_t)i * 7;
    if (!counted(P, r, rlo, rhi)) continue;
    cnt++;
    for (int s = 0; s < P->nspec; ++s) sum[s] += r[3 + s];
  }
  double mean[4];
  for (int s = 0; s < P->nspec; ++s) mean[s] = sum[s] / (double)cnt;
  double var = 0.0;
  for (int s = 0; s < P->nspec; ++s) {
    double sq = 0.0;
    for (long i = 0; i < P->n; ++i) {
      const double *r = P->rec + (size_t)i * 7;
      if (!counted(P, r, rlo, rhi)) continue;
      sq += (r[3 + s] - mean[s]) * (r[3 + s] - mean[s]);
    }
    var = var + 1.0 / (double)(cnt - 1) * sq;
  }
  double vol = ((lmax[0] - lmin[0]) / R_E) * ((lmax[1] - lmin[1]) / R_E) * ((lmax[2] - lmin[2]) / R_E);
  double var1 = vol * vol * var / (double)cnt;
  if (sqrt(fabs(var1)) > alpha) {
    for (int i = 0; i < numincrease; ++i, ++j) {
      double rt[3];
      for (int c = 0; c < 3; ++c) rt[c] = uniform(P->seed, 3, pass, hid, (uint64_t)(3 * j + c)) * (lmax[c] - lmin[c]) + lmin[c];
      pool_add(P, rt);
    }
    recursivesampler(P, lmin, lmax, alpha, depth + 1, maxdepth, numincrease, pass, hid);
  }
}

static void recursivesampler(pool_t *P, const double limit_min[3], const double limit_max[3], double alpha, int depth,
                             int maxdepth, int numincrease, uint64_t pass, uint64_t callid) {
  if (depth > maxdepth) return; /* :65-68 */
  int dim = depth % 3;          /* :73 (0-based) */
  double lmin[3], lmax[3];
  memcpy(lmin, limit_min, sizeof lmin);
  memcpy(lmax, limit_max, sizeof lmax);
  lmax[dim] = lmin[dim] + 0.5 * (lmax[dim] - lmin[dim]); /* :81-82 */
  one_side(P, lmin, lmax, alpha, depth, maxdepth, numincrease, pass, 2 * callid);
  memcpy(lmin, limit_min, sizeof lmin);
  memcpy(lmax, limit_max, sizeof lmax);
  lmin[dim] = lmin[dim] + 0.5 * (lmax[dim] - lmin[dim]); /* :140-141 */
  one_side(P, lmin, lmax, alpha, depth, maxdepth, numincrease, pass, 2 * callid + 1);
}

/* The driver program's stages.  counts = {n_zero_altitude, n_iri_pad, n_initial_radial, n_initial_uniform,
 * adaptive_nmax}.  Returns malloc'd [n][3+nspec]; stage_counts[6] as the library's. */
double *so_build_samples(so_model *m, const double bounds[6], const long counts[5], double initial_tol, int max_recursion,
                         int numincrease, int max_passes, uint64_t seed, long n_in, const double *in_pts, long *n_out,
                         long stage_counts[6]) {
  pool_t P;
  memset(&P, 0, sizeof P);
  P.m = m;
  P.nspec = so_model_nspec(m);
  P.seed = seed;
  const int w = 3 + P.nspec;
  const double lo[3] = {bounds[0], bounds[2], bounds[4]}, hi[3] = {bounds[1], bounds[3], bounds[5]};
  if (!numincrease) numincrease = 5;
  if (max_passes <= 0) max_passes = 64;
  for (int q = 0; q < 6; ++q) stage_counts[q] = 0;
  for (long i = 0; i < n_in; ++i) { /* :210-227: kdtree_add without f() */
    if (P.n == P.cap) {
      P.cap = P.cap ? 2 * P.cap : 1024;
      P.rec = (double *)realloc(P.rec, (size_t)P.cap * 7 * sizeof(double));
    }
    double *r = P.rec + (size_t)P.n * 7;
    memset(r, 0, 7 * sizeof(double));
    memcpy(r, in_pts + (size_t)i * w, (size_t)w * sizeof(double));
    P.n++;
  }
  stage_counts[0] = n_in;
  /* radial :228-272 */
  if (counts[2] > 0) {
    double r2 = 0.0;
    for (int q = 0; q < 8; ++q) {
      double x = bounds[q & 4 ? 1 : 0], y = bounds[q & 2 ? 3 : 2], z = bounds[q & 1 ? 5 : 4];
      double v = x * x + y * y + z * z;
      if (v > r2) r2 = v;
    }
    double rmax = sqrt(r2);
    for (long i = 0; i < counts[2]; ++i) {
      double p[3];
      for (uint64_t t = 0;; ++t) {
        shell_point(seed, 1, (uint64_t)i, t, R_E, rmax, p);
        if (inside(p, lo, hi)) break;
      }
      pool_add(&P, p);
    }
    stage_counts[1] = counts[2];
  }
  /* uniform :275-296 */
  for (long i = 0; i < counts[3]; ++i) {
    double p[3];
    for (int c = 0; c < 3; ++c) p[c] = lo[c] + uniform(seed, 2, (uint64_t)i, 0, (uint64_t)c) * (hi[c] - lo[c]);
    pool_add(&P, p);
  }
  stage_counts[2] = counts[3];
  /* adaptive :298-347 */
  if (counts[4] > 0) {
    P.nsamples = 0;
    double tol = initial_tol;
    for (int pass = 0; P.nsamples < counts[4] && pass < max_passes; ++pass) {
      recursivesampler(&P, lo, hi, tol, 0, max_recursion, numincrease, (uint64_t)pass, 1);
      tol = tol / 2.0;
    }
    stage_counts[3] = P.nsamples;
  }
  /* zero altitude :349-377, iri :379-405 */
  for (int stg = 4; stg <= 5; ++stg) {
    long n = stg == 4 ? counts[0] : counts[1];
    double rmax = stg == 4 ? R_E : R_E + 2000000.0;
    for (long i = 0; i < n; ++i) {
      double p[3];
      shell_point(seed, (uint64_t)stg, (uint64_t)i, 0, R_E, rmax, p);
      if (inside(p, lo, hi)) {
        pool_add(&P, p);
        stage_counts[stg]++;
      }
    }
  }
  double *out = (double *)malloc(((size_t)P.n + 1) * (size_t)w * sizeof(double));
  for (long i = 0; i < P.n; ++i) memcpy(out + (size_t)i * w, P.rec + (size_t)i * 7, (size_t)w * sizeof(double));
  free(P.rec);
  *n_out = P.n;
  return out;
}

void so_free(void *p) { free(p); }
