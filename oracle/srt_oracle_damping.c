/* srt_oracle_damping.c -- TEST INFRASTRUCTURE ONLY (see srt_oracle.h).
 *
 * CPU restatement (plain C, scalar, one function per MATLAB function) of the reference's hot-plasma damping
 * post-processor, /root/reference/matlab/damping/:
 *   test_dampray.m:24-99, test_compare_time_and_spatial_damping.m:47-85   per-row driver, running magnitude
 *   spatialdamping.m, temporaldamping.m, hot_dispersion_imag.m, hot_dispersion_real.m, integrand.m, fG1.m, fG2.m,
 *   suprathermal.m, maxwellboltzmann.m, quadva.m (Vadapt + f1 + check_spacing), ../stix_parameters.m, const.m,
 *   ../physconst.m
 *
 * PARITY UNPINNED: the reference for this row is MATLAB source; there is no MATLAB or Octave in the image, the ray
 * files its test scripts read (test_4000Hz.txt, test_400Hz.txt) are not in the repository, and it ships no expected
 * outputs.  What pins this file instead: closed-form checks in tests/test_oracle_damping.py (quadva on integrals with
 * known values; Maxwellian Landau damping of a parallel whistler against the textbook rate; the scripts' own
 * consistency check, spatial rate = temporal rate / group speed).  besselj -> libm j0/j1/jn.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define EPSM 2.220446049250313e-16 /* matlab eps */
static const double EPS0 = 8.854187817e-12;
static const double PI_ = 3.141592653589793;
#define ROW 20

typedef struct {
  int dist, mode, nres;
  int m[8];
  double Ne_h, kT, tol;
} sod_params;

/* const.m */
static const double Q_ = 1.60217646e-19, ME_ = 9.10938188e-31, CLIGHT_CONST = 299792458.0;
static double clight_physconst(void) { return sqrt(1.0 / EPS0 / (PI_ * 4e-7)); } /* physconst.m */

/* suprathermal.m / maxwellboltzmann.m */
static double dist_f(const sod_params *p, double vperp, double vpar) {
  if (p->dist == 0) {
    double v = 100.0 * sqrt(vperp * vperp + vpar * vpar + 1.0);
    double v2 = v * v, v4 = v2 * v2;
    double f = 4.9e5 / v4 - 8.3e14 / (v4 * v) + 5.4e23 / (v4 * v2);
    return f * 1.0e12;
  }
  double c = ME_ / (2.0 * PI_ * p->kT);
  return p->Ne_h * (c * sqrt(c)) * exp(-ME_ * (vperp * vperp + vpar * vpar) / 2.0 / p->kT);
}

static double besselj(int n, double x) {
  double sg = 1.0;
  if (n < 0) {
    n = -n;
    if (n & 1) sg = -sg;
  }
  if (x < 0.0) {
    x = -x;
    if (n & 1) sg = -sg;
  }
  return sg * (n == 0 ? j0(x) : n == 1 ? j1(x) : jn(n, x));
}

/* ../stix_parameters.m, nus = 0 */
static void stix(double w, int nspec, const double *qs, const double *Ns, const double *ms, double B0, double *S, double *D,
                 double *P, double *R, double *L) {
  double sr = 0, sl = 0, sp = 0;
  for (int s = 0; s < nspec; ++s) {
    double wps2 = Ns[s] * (qs[s] * qs[s]) / ms[s] / EPS0;
    double wcs = (qs[s] * B0) / ms[s];
    sr += wps2 / (w * (w + wcs));
    sl += wps2 / (w * (w - wcs));
    sp += wps2 / (w * w);
  }
  *R = 1 - sr;
  *L = 1 - sl;
  *P = 1 - sp;
  *S = 0.5 * (*R + *L);
  *D = 0.5 * (*R - *L);
}

typedef struct {
  const sod_params *p;
  double kperp, kpar, w, wch, qh, mh, R, L, P, S;
} integ_ctx;

/* fG1.m / fG2.m derivative pieces */
static void dfs(const sod_params *p, double vperp, double vpar, double *dfperp, double *dfpar) {
  double d = 1e-8 * fabs(vperp);
  if (d < 10 * EPSM) d = 10 * EPSM;
  *dfperp = (dist_f(p, vperp + d, vpar) - dist_f(p, vperp - d, vpar)) / (2 * d);
  d = 1e-8 * fabs(vpar);
  if (d < 10 * EPSM) d = 10 * EPSM;
  *dfpar = (dist_f(p, vperp, vpar + d) - dist_f(p, vperp, vpar - d)) / (2 * d);
}

/* integrand.m (scalar vperp) */
static double integrand(const integ_ctx *c, double vperp) {
  double cl = clight_physconst();
  double theta = atan2(c->kperp, c->kpar);
  double n = sqrt((cl * cl / (c->w * c->w)) * (c->kperp * c->kperp + c->kpar * c->kpar));
  double ct = cos(theta), st = sin(theta), n2 = n * n;
  double sum = 0.0;
  double x = c->kperp * vperp / c->wch;
  for (int mi = 0; mi < c->p->nres; ++mi) {
    int m = c->p->m[mi];
    double Jm = besselj(m, x), Jm1 = besselj(m - 1, x), Jp1 = besselj(m + 1, x);
    double vpar = (c->w - m * c->wch) / c->kpar;
    double dfperp, dfpar;
    dfs(c->p, vperp, vpar, &dfperp, &dfpar);
    double cross = vpar * dfperp - vperp * dfpar;
    double G1 = dfperp - (c->kpar / c->w) * cross;                                             /* fG1.m:23 */
    double G2 = Jm * (dfpar - (m * c->wch + EPSM) / (c->w * vperp + EPSM) * cross);            /* fG2.m:28 */
    double Rn = c->R - n2, Ln = c->L - n2, dJ = Jp1 - Jm1;
    sum = sum + (G1 * ((c->P - n2 * st * st) * (2 * Ln * vperp * Jp1 * Jp1 + 2 * vperp * Rn * Jm1 * Jm1 + n2 * st * st * vperp * dJ * dJ) -
                       n2 * ct * st * (2 * vpar * Jm * (Jp1 * Rn + Jm1 * Ln) + n2 * ct * st * vperp * dJ * dJ)) +
                 G2 * (4 * vpar * Jm * (Ln * Rn + n2 * st * st * (c->S - n2)) - 2 * n2 * ct * st * (Rn * vperp * Jm1 + Ln * vperp * Jp1)));
  }
  return -2 * PI_ * PI_ * ((c->qh * c->qh / c->mh / EPS0) / (c->w * fabs(c->kpar))) * sum * vperp;
}

/* hot_dispersion_imag.m:37-44: integrand_t */
static double integrand_t(void *ctx, double t) {
  const integ_ctx *c = (const integ_ctx *)ctx;
  double k = clight_physconst();
  return ((1 + EPSM) / (t * t + EPSM)) * (k * integrand(c, k * ((1 - t + EPSM) / (t + EPSM))));
}

/* quadva.m for a finite interval [a,b] without interior break points.  Returns Ifx; *ok = OK; *fail = 1 for
 * 'Difficulty evaluating integrand.' */
typedef double (*qfun)(void *, double);
double sod_quadva(qfun f, void *ctx, double a, double b, double reltol, double abstol, int *ok, int *fail, double *errbnd_out,
                  int *nevals) {
  static const double pn[7] = {0.2077849550078985, 0.4058451513773972, 0.5860872354676911, 0.7415311855993944,
                               0.8648644233597691, 0.9491079123427585, 0.9914553711208126};
  static const double pw[7] = {0.2044329400752989, 0.1903505780647854, 0.1690047266392679, 0.1406532597155259,
                               0.1047900103222502, 0.06309209262997855, 0.02293532201052922};
  static const double pw7[7] = {0, 0.3818300505051189, 0, 0.2797053914892767, 0, 0.1294849661688697, 0};
  double nodes[15], wt[15], ewt[15];
  for (int i = 0; i < 7; ++i) {
    nodes[i] = -pn[6 - i];
    nodes[8 + i] = pn[i];
    wt[i] = pw[6 - i];
    wt[8 + i] = pw[i];
    ewt[i] = pw[6 - i] - pw7[6 - i];
    ewt[8 + i] = pw[i] - pw7[i];
  }
  nodes[7] = 0;
  wt[7] = 0.2094821410847278;
  ewt[7] = 0.2094821410847278 - 0.4179591836734694;
  double rtol = reltol <= 0 ? 0 : fmax(reltol, 100 * EPSM), atol = fmax(abstol, 0);
  if (atol + rtol == 0) {
    rtol = 1e-5;
    atol = 1e-10;
  }
  enum { MAXS = 1400 };
  double *lo = malloc(sizeof(double) * MAXS * 2), *hi = lo + MAXS;
  double *q = malloc(sizeof(double) * MAXS * 2), *e = q + MAXS;
  int nsub = 10;
  for (int i = 0; i < 10; ++i) { /* linspace(-1,1,11) */
    lo[i] = -1.0 + i * (2.0 / 10.0);
    hi[i] = i == 9 ? 1.0 : -1.0 + (i + 1) * (2.0 / 10.0);
  }
  double tbma = 2.0, IfxOK = 0, errOK = 0, Ifx = NAN, errbnd = NAN;
  int first = 1;
  *ok = 1;
  *fail = 0;
  if (nevals) *nevals = 0;
  for (;;) {
    int bad = 0;
    double prev = -INFINITY;
    for (int s = 0; s < nsub; ++s) {
      double mid = (lo[s] + hi[s]) / 2, hh = (hi[s] - lo[s]) / 2;
      double qs = 0, es = 0;
      for (int i = 0; i < 15; ++i) {
        double t = nodes[i] * hh + mid;
        /* f1 (quadva.m:126-136) */
        double Tt = 0.25 * (b - a) * t * (3 - t * t) + 0.5 * (b + a);
        if (!(s == 0 && i == 0) && (Tt - prev) <= 100 * EPSM * fmax(fabs(prev), fabs(Tt))) bad = 1; /* check_spacing */
        prev = Tt;
        double y = f(ctx, Tt);
        y = 0.75 * (b - a) * y * (1 - t * t);
        if (!isfinite(y)) bad = 1;
        qs += wt[i] * y;
        es += ewt[i] * y;
        if (nevals) ++*nevals;
      }
      q[s] = qs * hh;
      e[s] = es * hh;
    }
    if (bad) break;
    double sq = 0, se = 0;
    for (int s = 0; s < nsub; ++s) {
      sq += q[s];
      se += e[s];
    }
    Ifx = sq + IfxOK;
    errbnd = fabs(se + errOK);
    double tol = fmax(atol, rtol * fabs(Ifx));
    if (errbnd <= tol) goto done;
    int nkeep = 0;
    double accE = 0, accQ = 0;
    double *nlo = malloc(sizeof(double) * MAXS * 2), *nhi = nlo + MAXS;
    for (int s = 0; s < nsub; ++s) {
      double hh = (hi[s] - lo[s]) / 2;
      if (fabs(e[s]) <= (2 / tbma) * hh * tol) {
        accE += e[s];
        accQ += q[s];
      } else {
        if (2 * nkeep + 1 < MAXS) {
          double mid = (lo[s] + hi[s]) / 2;
          nlo[2 * nkeep] = lo[s];
          nhi[2 * nkeep] = mid;
          nlo[2 * nkeep + 1] = mid;
          nhi[2 * nkeep + 1] = hi[s];
        }
        ++nkeep;
      }
    }
    errOK = errOK + accE;
    IfxOK = IfxOK + accQ;
    if (nkeep == 0) {
      free(nlo);
      goto done;
    }
    if (2 * nkeep > 650) {
      free(nlo);
      break;
    }
    free(lo);
    lo = nlo;
    hi = nhi;
    nsub = 2 * nkeep;
    first = 0;
  }
  *ok = 0;
  if (first) *fail = 1;
done:
  free(lo);
  free(q);
  if (errbnd_out) *errbnd_out = errbnd;
  return Ifx;
}

/* hot_dispersion_imag.m */
static double hot_dispersion_imag(const sod_params *p, double kperp, double kpar, double w, double wch, double qh, double mh, int nspec,
                                  const double *qs, const double *Ns, const double *ms, double B0, double TOL, int *ok, int *fail) {
  integ_ctx c;
  double D;
  c.p = p;
  c.kperp = kperp;
  c.kpar = kpar;
  c.w = w;
  c.wch = wch;
  c.qh = qh;
  c.mh = mh;
  stix(w, nspec, qs, Ns, ms, B0, &c.S, &D, &c.P, &c.R, &c.L);
  return sod_quadva(integrand_t, &c, 0.0, 1.0, TOL, EPSM, ok, fail, NULL, NULL);
}

/* hot_dispersion_real.m */
static double hot_dispersion_real(double kperp, double kpar, double w, int nspec, const double *qs, const double *Ns, const double *ms,
                                  double B0) {
  double S, D, P, R, L, cl = clight_physconst();
  stix(w, nspec, qs, Ns, ms, B0, &S, &D, &P, &R, &L);
  double theta = atan2(kperp, kpar);
  double n = cl / w * sqrt(kperp * kperp + kpar * kpar);
  double s2 = sin(theta) * sin(theta), c2 = cos(theta) * cos(theta);
  double A = S * s2 + P * c2, B = R * L * s2 + P * S * (1 + c2), C = R * L * P;
  double nn = n * n;
  return 4 * (A * nn * nn - B * nn + C);
}

/* spatialdamping.m (one hot species) */
double sod_spatialdamping(const sod_params *p, double kperp, double kpar, double w, double wch, double qh, double mh, int nspec,
                          const double *qs, const double *Ns, const double *ms, double B0, int *ok, int *fail) {
  double cl = clight_physconst();
  double theta = atan2(kperp, kpar);
  double n = sqrt((cl * cl / (w * w)) * (kperp * kperp + kpar * kpar));
  double S, D, P, R, L;
  stix(w, nspec, qs, Ns, ms, B0, &S, &D, &P, &R, &L);
  double A = S * sin(theta) * sin(theta) + P * cos(theta) * cos(theta);
  double B = R * L * sin(theta) * sin(theta) + P * S * (1 + cos(theta) * cos(theta));
  double Di = hot_dispersion_imag(p, kperp, kpar, w, wch, qh, mh, nspec, qs, Ns, ms, B0, p->tol, ok, fail);
  return 0 + -(w / cl) * (1.0 / 2) * (1 / (4 * n * (2 * A * n * n - B))) * Di;
}

/* temporaldamping.m */
double sod_temporaldamping(const sod_params *p, double kperp, double kpar, double w, double wch, double qh, double mh, int nspec,
                           const double *qs, const double *Ns, const double *ms, double B0, int *ok, int *fail) {
  double d = 1e-8 * fabs(w);
  if (d < 10 * EPSM) d = 10 * EPSM;
  double dD0dw = (hot_dispersion_real(kperp, kpar, w + d, nspec, qs, Ns, ms, B0) - hot_dispersion_real(kperp, kpar, w - d, nspec, qs, Ns, ms, B0)) /
                 (2 * d);
  double Di = hot_dispersion_imag(p, kperp, kpar, w, wch, qh, mh, nspec, qs, Ns, ms, B0, p->tol, ok, fail);
  return 0 + -Di / dD0dw;
}

/* test_dampray.m:50-93 over the kept rows of a batch of rays (rows [nrays][slots][20] as the library's). */
void sod_damping(const sod_params *pin, int nspec, const double *qs, const double *ms, int slots, int outputper, long nrays,
                 const double *rows, const int *nrows, const double *w0, double *rate, double *magnitude, int *flag) {
  sod_params p = *pin;
  if (p.nres == 0) {
    p.nres = 3;
    p.m[0] = -1;
    p.m[1] = 0;
    p.m[2] = 1;
  }
  if (!(p.tol > 0)) p.tol = 1e-3;
  for (long ray = 0; ray < nrays; ++ray) {
    int T = nrows[ray], kept = T > 0 ? (T - 1) / outputper + 1 : 0;
    double mag = 1.0;
    for (int r = 0; r < slots; ++r) {
      size_t idx = (size_t)ray * slots + r;
      rate[idx] = 0;
      if (flag) flag[idx] = 0;
      if (r >= kept) {
        if (magnitude) magnitude[idx] = 0;
        continue;
      }
      if (r > 0) {
        const double *row = rows + idx * ROW, *prev = row - ROW;
        double w = w0[ray];
        const double *vg = row + 7, *n = row + 10, *B0 = row + 13, *Ns = row + 16;
        double Bmag = sqrt(B0[0] * B0[0] + B0[1] * B0[1] + B0[2] * B0[2]);
        double wce_h = (-Q_ * Bmag) / ME_;
        double k[3], kk = 0, kpar = 0, Bhat[3], kp2 = 0;
        for (int c = 0; c < 3; ++c) k[c] = n[c] * w / CLIGHT_CONST;
        for (int c = 0; c < 3; ++c) kk += k[c] * k[c];
        double kmag = sqrt(kk);
        for (int c = 0; c < 3; ++c) Bhat[c] = B0[c] / Bmag;
        for (int c = 0; c < 3; ++c) kpar += k[c] * Bhat[c];
        for (int c = 0; c < 3; ++c) {
          double v = k[c] - kpar * Bhat[c];
          kp2 += v * v;
        }
        double kperp = sqrt(kp2);
        if (kmag != 0) {
          int ok, fail;
          if (p.mode == 0) {
            double ki = sod_spatialdamping(&p, kperp, kpar, w, wce_h, -Q_, ME_, nspec, qs, Ns, ms, Bmag, &ok, &fail);
            double kv = 0, vv = 0;
            for (int c = 0; c < 3; ++c) {
              kv += k[c] * vg[c];
              vv += vg[c] * vg[c];
            }
            double ka = ki * kv / (kmag * sqrt(vv));
            if (fail) ka = NAN;
            rate[idx] = ka;
            double dx = row[1] - prev[1], dy = row[2] - prev[2], dz = row[3] - prev[3];
            mag = mag * exp(-sqrt(dx * dx + dy * dy + dz * dz) * ka);
          } else {
            double g = sod_temporaldamping(&p, kperp, kpar, w, wce_h, -Q_, ME_, nspec, qs, Ns, ms, Bmag, &ok, &fail);
            if (fail) g = NAN;
            rate[idx] = g;
            mag = mag * exp(g * (row[0] - prev[0]));
          }
          if (flag) flag[idx] = fail ? 2 : ok ? 0 : 1;
        } else {
          mag = 0; /* magnitude(ii) is never assigned */
          if (flag) flag[idx] = 3;
        }
      }
      if (magnitude) magnitude[idx] = mag;
    }
  }
}

/* quadva on test integrands, for tests/test_oracle_damping.py: kind 0 exp(x), 1 1/sqrt(x) (end-point singularity),
 * 2 cos(50 x), 3 1/(1e-4+(x-0.3)^2) (sharp peak) */
static double test_fun(void *ctx, double x) {
  switch (*(int *)ctx) {
    case 0: return exp(x);
    case 1: return 1.0 / sqrt(x);
    case 2: return cos(50.0 * x);
    default: return 1.0 / (1e-4 + (x - 0.3) * (x - 0.3));
  }
}
double sod_quadva_test(int kind, double a, double b, double reltol, double abstol, int *ok, double *errbnd, int *nevals) {
  int fail;
  return sod_quadva(test_fun, &kind, a, b, reltol, abstol, ok, &fail, errbnd, nevals);
}
