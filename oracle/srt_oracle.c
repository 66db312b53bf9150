/* srt_oracle.c -- TEST INFRASTRUCTURE ONLY (see srt_oracle.h).
 *
 * Scalar fp64 restatement of the reference's hot path, written to follow the Fortran's operation
 * order so that, compiled without FMA contraction (-ffp-contract=off, baseline x86-64), it agrees
 * with the reference (flang -O3 build, oracle/_ref) to the last bit wherever libm agrees.
 *
 * Compile: gcc -O2 -ffp-contract=off -fno-fast-math -shared -fPIC srt_oracle.c srt_oracle_scattered.c -lm -lpthread
 */
#include "srt_oracle.h"

#include <complex.h>
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "srt_oracle_internal.h"
#include "tricubic_matrix.h"

/* ------------------------------------------------------------------ constants.f95:4-12 */
static const double EPS0 = 8.854187817e-12;
static const double PI = 3.141592653589793238462643;
static const double R_E = 6371.2e3;
static double c_light(void) {
  /* C = sqrt(1/EPS0/MU0), MU0 = PI*4e-7 (constants.f95:6-7) */
  const double MU0 = PI * 4e-7;
  return sqrt(1.0 / EPS0 / MU0);
}
double so_speed_of_light(void) { return c_light(); }
#define R2D (180.0 / PI)

static double dot3(const double a[3], const double b[3]) {
  return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2];
}
/* flang's MAX(x,y) lowers to (x > y) ? x : y  (SURVEY Appendix A-1 probe) */
static double fmax_f(double x, double y) { return (x > y) ? x : y; }

/* Complex arithmetic as flang lowers it (established by bisecting against oracle/_ref):
 * a real operand is promoted to (r, 0) and the full complex operation is performed; division is
 * compiler-rt's __divdc3 (scale the divisor by 2^-ilogb, then the textbook formula). */
typedef struct { double re, im; } zc;
static zc zmul(zc a, zc b) { zc r = {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; return r; }
static zc zdiv(zc a, zc b) {
  int ilogbw = 0;
  double c = b.re, d = b.im;
  double logbw = logb(fmax(fabs(c), fabs(d)));
  if (isfinite(logbw)) {
    ilogbw = (int)logbw;
    c = scalbn(c, -ilogbw);
    d = scalbn(d, -ilogbw);
  }
  double denom = c * c + d * d;
  zc r = {scalbn((a.re * c + a.im * d) / denom, -ilogbw), scalbn((a.im * c - a.re * d) / denom, -ilogbw)};
  return r;
}
static zc zsqrt(zc a) {
  double complex v = csqrt(a.re + a.im * I);
  zc r = {creal(v), cimag(v)};
  return r;
}
static zc zreal(double x) { zc r = {x, 0.0}; return r; }

/* ------------------------------------------------------------------ util.f95:109-122 */
void so_cartesian_to_spherical(const double x[3], double p[3]) {
  p[0] = sqrt((x[0] * x[0] + x[1] * x[1]) + x[2] * x[2]);
  p[1] = atan2(x[1], x[0]);
  if (p[0] != 0.0)
    p[2] = acos(x[2] / p[0]);
  else
    p[2] = 0.0;
}
/* util.f95:126-142  matmul(transpose(A), p) */
static void spherical_to_cartesian_vec(const double p[3], double theta, double phi, double out[3]) {
  double ct = cos(theta), st = sin(theta), cp = cos(phi), sp = sin(phi);
  out[0] = ((ct * sp) * p[0] + (-st) * p[1]) + (ct * cp) * p[2];
  out[1] = ((st * sp) * p[0] + ct * p[1]) + (st * cp) * p[2];
  out[2] = (cp * p[0] + 0.0 * p[1]) + (-sp) * p[2];
}
/* util.f95:148-162  matmul(A, p) */
static void cartesian_to_spherical_vec(const double p[3], double theta, double phi, double out[3]) {
  double ct = cos(theta), st = sin(theta), cp = cos(phi), sp = sin(phi);
  out[0] = ((ct * sp) * p[0] + (st * sp) * p[1]) + cp * p[2];
  out[1] = ((-st) * p[0] + ct * p[1]) + 0.0 * p[2];
  out[2] = ((ct * cp) * p[0] + (st * cp) * p[1]) + (-sp) * p[2];
}

/* ------------------------------------------------------------------ bmodel_dipole.f95:20-48 */
static void bmodel_cartesian(const double x[3], double B[3]) {
  double p[3];
  so_cartesian_to_spherical(x, p);
  double R = p[0] / R_E;
  double Bo = .312 / 10000.0;
  double Bor3 = Bo * pow(R, -3.0);
  double Brad = -2.0 * Bor3 * cos(p[2]);
  double Btheta = -Bor3 * sin(p[2]);
  double v[3] = {Brad, 0.0, Btheta};
  spherical_to_cartesian_vec(v, p[1], p[2], B);
}

/* ------------------------------------------------------------------ xform_double chain */
static void rotate_x(double a, const double in[3], double out[3]) { /* Rotate_x.f95 */
  double c = cos(a), s = sin(a);
  out[0] = in[0];
  out[1] = in[1] * c + in[2] * s;
  out[2] = in[2] * c - in[1] * s;
}
static void rotate_y(double a, const double in[3], double out[3]) { /* Rotate_y.f95:2-12 */
  double c = cos(a), s = sin(a);
  out[0] = in[0] * c + in[2] * s;
  out[1] = in[1];
  out[2] = in[2] * c - in[0] * s;
}
static void rotate_z(double a, const double in[3], double out[3]) { /* Rotate_z.f95 */
  double c = cos(a), s = sin(a);
  out[0] = in[0] * c + in[1] * s;
  out[1] = in[1] * c - in[0] * s;
  out[2] = in[2];
}
static const double DEGRAD = 3.141592653589793238462643 / 180.0;
/* T0.f95:7-24 */
static double t0_d(const int itime[2], double *ut) {
  int iyr = itime[0] / 1000;
  int iday = itime[0] - iyr * 1000;
  *ut = itime[1] / 3600000.0;
  double fracday = *ut / 24.0;
  double rmjd = 45.0 + (double)(float)(iyr - 1859) * 365.0 + ((double)(float)((iyr - 1861) / 4) + 1.0) +
                (double)(float)iday - 1.0 + fracday;
  return (rmjd - 51544.5) / 36525.0;
}
/* Get_q_c.f95:4-30 (+ POL_TO_CART.f95, T1.f95:7-21, T2.f95:7-31) */
static void get_q_c(const int itime[2], double q_c[3]) {
  int iyr = itime[0] / 1000;
  int iday = itime[0] - iyr * 1000;
  double ut = itime[1] / 3600000.0;
  double fracday = ut / 24.0;
  double rmjd = 45.0 + (double)(iyr - 1859) * 365.0 + ((double)((iyr - 1861) / 4) + 1.0) + (double)iday -
                1.0 + fracday;
  double factor = (rmjd - 46066.0) / 365.25;
  double phi = (78.8 + 4.283e-2 * factor) * DEGRAD;
  double lamda = (289.1 - 1.413e-2 * factor) * DEGRAD;
  double q_g[3], temp[3];
  double coslat = cos(phi);
  q_g[0] = 1.0 * coslat * cos(lamda);
  q_g[1] = 1.0 * coslat * sin(lamda);
  q_g[2] = 1.0 * sin(phi);
  /* t1_d(itime, q_g, temp, -1) */
  {
    double ut1;
    double t0 = t0_d(itime, &ut1);
    double theta = (100.461 + 36000.770 * t0 + 15.04107 * ut1) * DEGRAD;
    rotate_z((double)(-1.0f) * theta, q_g, temp);
  }
  /* t2_d(itime, temp, q_c, 1) */
  {
    double ut2;
    double tt0 = t0_d(itime, &ut2);
    double epsilon = (23.439 - 0.013 * tt0) * DEGRAD;
    double m = (357.528 + 35999.05 * tt0 + 0.04107 * ut2) * DEGRAD;
    double cgamma = 280.46 + 36000.772 * tt0 + 0.04107 * ut2;
    double lamdas = (cgamma + (1.915 - 0.0048 * tt0) * sin(m) + 0.02 * sin(2.0 * m)) * DEGRAD;
    double t[3];
    rotate_x(epsilon, temp, t);
    rotate_z(lamdas, t, q_c);
  }
}
/* T4.f95:7-18: mu */
void so_dipole_tilt(int yearday, int msec, double *mu) {
  int itime[2] = {yearday, msec};
  double q[3];
  get_q_c(itime, q);
  *mu = -atan(q[0] / sqrt(q[1] * q[1] + q[2] * q[2]));
}

/* Dipole-B tail shared by all three adapters (ngo_..adapter.f95:144-202, interp_..:184-267,
 * scattered_..:283-370) with use_igrf = use_tsyganenko = 0. */
void so_bfield(so_model *m, const double x[3], double B0[3]) {
  double Bsm[3], Bgsm[3], Bt[3];
  float base[3];
  if (m->use_igrf) {
    /* IGRF_GSM(real(x_gsm/R_E)), x_gsm = SM_TO_GSM_d(itime, x)   (interp_dens_model_adapter.f95:186,236-241) */
    double xg[3];
    rotate_y(-1 * m->mu, x, xg);
    so_igrf_gsw(m->igrf_G, m->igrf_H, m->igrf_REC, m->igrf_A, (float)(xg[0] / R_E), (float)(xg[1] / R_E), (float)(xg[2] / R_E),
                &base[0], &base[1], &base[2]);
  } else {
    bmodel_cartesian(x, Bsm);
    /* SM_TO_GSM_d: t4_d(..., -1) => rotate_y(-mu) */
    rotate_y(-1 * m->mu, Bsm, Bgsm);
    for (int i = 0; i < 3; i++) base[i] = (float)(1.0e9 * Bgsm[i]); /* B0xBASE = real(1.0e9_DP*B0tmp2(1)) */
  }
  for (int i = 0; i < 3; i++) {
    float tsy = 0.0f;
    Bt[i] = (double)(base[i] + tsy) * 1.0e-9;
  }
  /* GSM_TO_SM_d: rotate_y(+mu) */
  rotate_y(1 * m->mu, Bt, B0);
}

/* ================================================================== Ngo model */
/* list-directed reader: every READ starts on a new record; items separated by blanks/commas;
 * a READ continues onto following records until its list is satisfied (ngo_dens_model.f95:52-118) */
typedef struct {
  FILE *f;
} ldr;
static int ldr_read(ldr *r, int n, double *out) {
  int got = 0;
  char line[4096];
  while (got < n) {
    if (!fgets(line, sizeof line, r->f)) return got;
    char *s = line;
    while (got < n) {
      while (*s == ' ' || *s == '\t' || *s == ',' || *s == '\r' || *s == '\n') s++;
      if (!*s) break;
      char tok[128];
      int k = 0;
      while (*s && *s != ' ' && *s != '\t' && *s != ',' && *s != '\r' && *s != '\n' && k < 127) {
        char ch = *s++;
        if (ch == 'd' || ch == 'D') ch = 'e';
        tok[k++] = ch;
      }
      tok[k] = 0;
      out[got++] = strtod(tok, NULL);
    }
  }
  return got;
}

/* ngo_dens_model.f95:165-353 `dens` -- only what determines ani(1:4) */
static void ngo_dens(so_ngo *g) {
  double exnor[5], qi[5], sh[5];
  double *z = g->z;
  double cosz2 = cos(z[2]);
  double sinz2 = sin(z[2]);
  double sinz22 = sinz2 * sinz2;
  /* init == -1 forever (:57): scale heights recomputed every call (:180-186) */
  double rb7370 = g->rbase / 7370.;
  sh[2] = (double)1.150600f * g->therm * rb7370 * rb7370;
  sh[3] = sh[2] / 4.;
  sh[4] = sh[3] / 4.;
  double alpha[5] = {0, 0, 0, 0, 0};
  double gph = g->rbase * (1.0 - g->rbase / z[1]);
  exnor[2] = exp(-gph / sh[2]);
  exnor[3] = exnor[2] * exnor[2] * exnor[2] * exnor[2];
  exnor[4] = exnor[3] * exnor[3] * exnor[3] * exnor[3];
  double q = 0.0;
  for (int i = 2; i <= g->num; i++) {
    qi[i] = g->alpha0[i] * exnor[i];
    q = q + qi[i];
  }
  for (int i = 2; i <= g->num; i++) alpha[i] = qi[i] / q;
  double anr = sqrt(q);
  double arg = (z[1] - g->rzero) / g->scbot;
  if (!(arg < 13.0)) arg = 13.0;
  double exarg = exp(-arg * arg);
  double anli = 1.0 - exarg;
  double l = z[1] / (g->r0 * sinz22);
  double ani1 = g->ane0 * anr;
  ani1 = ani1 * anli;
  if (g->kducts != 0) {
    /* plasmapause (:218-239) */
    double deltal = l - g->lk;
    if (!(deltal < 0.0)) {
      double d2 = g->ddk * g->ddk;
      double argl = deltal * deltal / (d2 * 2.0);
      if (!(argl < 80.00)) argl = 80.00;
      double f = exp(-argl);
      double trm = pow(g->rconsn / z[1], g->expk);
      double argr = (z[1] - g->rconsn) / g->scr;
      if (!(argr < 12.50)) argr = 12.5;
      double fr = exp(-argr * argr);
      double trmodl = trm + (1. - trm) * fr;
      double anlk = f + trmodl * (1.0 - f);
      ani1 = ani1 * anlk;
    }
    if (g->kducts != 1) {
      double latitu = g->latitu;
      int skip_ducts = 0;
      if (!(g->l0[2] > 0.0)) {
        /* sinusoidal density perturbation (:241-288) */
        g->kinit = 3;
        double dl = l + g->l0[2];
        if (!(dl * g->sidedu[2] >= 0.0)) dl = 0;
        double delk = -g->l0[2] - (g->lk + g->ddk) + g->dd[2] / 2;
        double critl = (g->lk + g->ddk) + fmod(delk, g->dd[2]);
        if (!(l <= critl)) {
          double argl = 2.0 * g->pi * dl / g->dd[2];
          double delnl = (g->def[2] / 2.) * (1. + cos(argl));
          double delr = 0.0, arglr = 0.0, frduct, anl;
          int done = 0, lower = 0;
          if (latitu <= 0 && z[1] <= g->rducus[2]) lower = 1;
          if (!lower && latitu >= 0 && z[1] <= g->rducun[2]) lower = 1;
          if (!lower) {
            if (latitu >= 0) delr = z[1] - g->rducun[2];
            if (latitu <= 0) delr = z[1] - g->rducus[2];
            if (latitu <= 0) arglr = delr * delr / g->hu2s[2];
            if (latitu >= 0) arglr = delr * delr / g->hu2n[2];
            if (arglr >= 75.0) {
              done = 1; /* goto 990 */
            } else {
              frduct = exp(-arglr);
              delnl = delnl * frduct;
              anl = 1.0 + delnl;
              ani1 = ani1 * anl;
              done = 1;
            }
          }
          if (!done) {
            int at970 = 0;
            if (latitu <= 0 && z[1] >= g->rducls[2]) at970 = 1;
            if (!at970 && latitu >= 0 && z[1] >= g->rducln[2]) at970 = 1;
            if (at970) {
              anl = 1.0 + delnl;
              ani1 = ani1 * anl;
            } else {
              if (latitu >= 0) delr = z[1] - g->rducln[2];
              if (latitu <= 0) delr = z[1] - g->rducls[2];
              if (latitu <= 0) arglr = delr * delr / g->hl2s[2];
              if (latitu >= 0) arglr = delr * delr / g->hl2n[2];
              if (!(arglr >= 75.0)) {
                frduct = exp(-arglr);
                delnl = delnl * frduct;
                anl = 1.0 + delnl;
                ani1 = ani1 * anl;
              }
            }
          }
        }
        if (g->kducts == 2) skip_ducts = 1; /* 990 */
      }
      if (!skip_ducts) {
        /* duct(s) (:289-336) */
        for (int kd = g->kinit; kd <= g->kducts; kd++) {
          double dl = l - g->l0[kd];
          if (!(dl * g->sidedu[kd] >= 0.0)) dl = 0;
          double d2 = g->dd[kd] * g->dd[kd];
          double argl = dl * dl / (d2 * 2.0);
          if (argl > 80.0) continue;
          double delnl = g->def[kd] * exp(-argl);
          double delr = 0.0, arglr = 0.0, frduct, anl;
          int lower = 0;
          if (latitu >= 0 && z[1] <= g->rducun[kd]) lower = 1;
          if (!lower && latitu <= 0 && z[1] <= g->rducus[kd]) lower = 1;
          if (!lower) {
            if (latitu >= 0) delr = z[1] - g->rducun[kd];
            if (latitu <= 0) delr = z[1] - g->rducus[kd];
            if (latitu >= 0) arglr = delr * delr / g->hu2n[kd];
            if (latitu <= 0) arglr = delr * delr / g->hu2s[kd];
            if (arglr >= 75.0) continue;
            frduct = exp(-arglr);
            delnl = delnl * frduct;
            anl = 1.0 + delnl;
          } else {
            int at235 = 0;
            if (latitu >= 0 && z[1] >= g->rducln[kd]) at235 = 1;
            if (!at235 && latitu <= 0 && z[1] >= g->rducls[kd]) at235 = 1;
            if (at235) {
              anl = 1.0 + delnl;
            } else {
              if (latitu >= 0) delr = z[1] - g->rducln[kd];
              if (latitu <= 0) delr = z[1] - g->rducls[kd];
              if (latitu >= 0) arglr = delr * delr / g->hl2n[kd];
              if (latitu <= 0) arglr = delr * delr / g->hl2s[kd];
              if (arglr >= 75.0) continue;
              frduct = exp(-arglr);
              delnl = delnl * frduct;
              anl = 1.0 + delnl;
            }
          }
          ani1 = ani1 * anl;
        }
      }
    }
  }
  g->ani[1] = ani1;
  for (int i = 2; i <= g->num; i++) g->ani[i] = ani1 * alpha[i];
}

/* ngo_dens_model.f95:29-160 readinput */
static int ngo_readinput(so_ngo *g, const char *filename) {
  memset(g, 0, sizeof *g);
  g->pi = (double)3.141592653589793f;
  g->r0 = 6370.f;
  double radgra = (double)180.f / g->pi;
  double grarad = (double)1.f / radgra;
  ldr r;
  r.f = fopen(filename, "r");
  if (!r.f) return -1;
  double v[16];
  if (ldr_read(&r, 4, v) != 4) goto bad; /* intera numres nsuppr spelat */
  g->kinit = 2;
  for (;;) {
    if (ldr_read(&r, 2, v) != 2) goto done; /* distre latitu ; end=9501 */
    g->latitu = v[1];
    if (v[0] <= -1.) break;
  }
  if (ldr_read(&r, 10, v) != 10) goto done;
  g->num = (int)v[0];
  g->kducts = (int)v[4];
  double dsrrng = v[7], dsrlat = v[8], dsdens = v[9];
  if (g->num == 0) goto done;
  if (g->num > 4) goto bad;
  if (ldr_read(&r, 5, v) != 5) goto done; /* egfeq therm hm absb relb */
  g->therm = v[1];
  if (ldr_read(&r, 5, v) != 5) goto done; /* rbase ane0 alpha0(2:4) */
  g->rbase = v[0];
  g->ane0 = v[1];
  g->alpha0[2] = v[2];
  g->alpha0[3] = v[3];
  g->alpha0[4] = v[4];
  if (ldr_read(&r, 5, v) != 5) goto done; /* rzero scbot rstop rdiv hmin */
  g->rzero = v[0];
  g->scbot = v[1];
  if (g->kducts != 0) {
    if (g->kducts > 9) goto bad;
    if (ldr_read(&r, 5, v) != 5) goto done; /* lk expk ddk rconsn scr */
    g->lk = v[0];
    g->expk = v[1];
    g->ddk = v[2];
    g->rconsn = v[3];
    g->scr = v[4];
    for (int k = 2; k <= g->kducts; k++) {
      if (ldr_read(&r, 12, v) != 12) goto done;
      g->l0[k] = v[0];
      g->def[k] = v[1];
      g->dd[k] = v[2];
      g->rducln[k] = v[3];
      g->rducun[k] = v[5];
      g->rducls[k] = v[7];
      g->rducus[k] = v[9];
      g->sidedu[k] = v[11];
      g->hl2n[k] = v[4] * v[4];
      g->hl2s[k] = v[8] * v[8];
      g->hu2n[k] = v[6] * v[6];
      g->hu2s[k] = v[10] * v[10];
    }
  }
  if (ldr_read(&r, 8, v) != 8) goto done; /* pstalt ... (profiles: no effect on later state) */
  /* setting ane0 to desired value at dsrrng,dsrlat (:120-123) */
  g->z[2] = ((double)90.00f - dsrlat) * grarad;
  g->z[1] = dsrrng * g->r0;
  ngo_dens(g);
  g->ane0 = g->ane0 * dsdens / g->ani[1];
done:
  fclose(r.f);
  return 0;
bad:
  fclose(r.f);
  return -1;
}

/* ngo_dens_model_adapter.f95:63-142 (density head of funcPlasmaParams) */
static void ngo_params(so_model *m, const double x[3], double qs[4], double Ns[4], double ms[4],
                       double nus[4]) {
  so_ngo *g = &m->ngo;
  double p[3];
  double d2r = 2.0 * PI / 360.0;
  so_cartesian_to_spherical(x, p);
  double sp = sin(p[2]);
  double L;
  if (R_E * (sp * sp) != 0.0)
    L = p[0] / (R_E * (sp * sp));
  else
    L = 0.0;
  double lam = 90.0 - (p[2] * 360.0 / 2.0 / PI);
  double lamr = d2r * lam;
  double cl = cos(lamr);
  double r = g->r0 * L * (cl * cl);
  g->z[1] = r;
  g->z[2] = d2r * (90.0 - lam);
  g->latitu = lam;
  ngo_dens(g);
  const double e = 1.602e-19;
  qs[0] = e * -1.0;
  qs[1] = e * 1.0;
  qs[2] = e * 1.0;
  qs[3] = e * 1.0;
  ms[0] = 9.10938188e-31;
  ms[1] = 1.6726e-27;
  ms[2] = 4.0 * 1.6726e-27;
  ms[3] = 16.0 * 1.6726e-27;
  Ns[0] = 1.0e6 * g->ani[1];
  Ns[1] = 1.0e6 * g->ani[2];
  Ns[2] = 1.0e6 * g->ani[3];
  Ns[3] = 1.0e6 * g->ani[4];
  nus[0] = nus[1] = nus[2] = nus[3] = 0.0;
}

so_model *so_model_create_ngo(const char *configfile, int yearday, int msec) {
  so_model *m = (so_model *)calloc(1, sizeof *m);
  m->kind = 1;
  m->nspec = 4;
  if (ngo_readinput(&m->ngo, configfile) != 0) {
    free(m);
    return NULL;
  }
  so_dipole_tilt(yearday, msec, &m->mu);
  return m;
}

/* ================================================================== interp (tricubic) model */
#define IDX(g, s, i, j, k) ((((size_t)(k) * (g)->ny + (j)) * (g)->nx + (i)) * (g)->nspec + (s))

/* libtricubic.f95:722-793, applied per species (interp_dens_model_adapter.f95:119-131) */
static void fd_axis(const so_grid *g, const double *src, double *dst, int axis, double h) {
  int n[3] = {g->nx, g->ny, g->nz};
  int na = n[axis];
  for (int s = 0; s < g->nspec; s++)
    for (int k = 0; k < g->nz; k++)
      for (int j = 0; j < g->ny; j++)
        for (int i = 0; i < g->nx; i++) {
          int c[3] = {i, j, k};
          int a = c[axis];
          int lo[3] = {i, j, k}, hi[3] = {i, j, k};
          double v;
          if (a == 0) {
            hi[axis] = 1;
            v = (src[IDX(g, s, hi[0], hi[1], hi[2])] - src[IDX(g, s, lo[0], lo[1], lo[2])]) / h;
          } else if (a == na - 1) {
            lo[axis] = na - 2;
            v = (src[IDX(g, s, hi[0], hi[1], hi[2])] - src[IDX(g, s, lo[0], lo[1], lo[2])]) / h;
          } else {
            lo[axis] = a - 1;
            hi[axis] = a + 1;
            v = (src[IDX(g, s, hi[0], hi[1], hi[2])] - src[IDX(g, s, lo[0], lo[1], lo[2])]) / 2.0 / h;
          }
          dst[IDX(g, s, i, j, k)] = v;
        }
}

static int grid_finish(so_grid *g) {
  size_t n = (size_t)g->nspec * g->nx * g->ny * g->nz;
  /* interp_dens_model_adapter.f95:87-95 */
  g->delx = (g->maxx - g->minx) / (g->nx - 1.0);
  g->dely = (g->maxy - g->miny) / (g->ny - 1.0);
  g->delz = (g->maxz - g->minz) / (g->nz - 1.0);
  g->x = (double *)malloc(sizeof(double) * g->nx);
  g->y = (double *)malloc(sizeof(double) * g->ny);
  g->z = (double *)malloc(sizeof(double) * g->nz);
  for (int i = 0; i < g->nx; i++) g->x[i] = (double)i * g->delx + g->minx;
  for (int i = 0; i < g->ny; i++) g->y[i] = (double)i * g->dely + g->miny;
  for (int i = 0; i < g->nz; i++) g->z[i] = (double)i * g->delz + g->minz;
  if (!g->have_derivs) {
    for (int a = 1; a < 8; a++) {
      g->arr[a] = (double *)malloc(sizeof(double) * n);
      memset(g->arr[a], 0, sizeof(double) * n);
    }
    int nx = g->nx, ny = g->ny, nz = g->nz;
    /* order of libtricubic.f95:736-790: dfdx, dfdy, dfdz, d2fdxdy=d/dx(dfdy), d2fdxdz=d/dx(dfdz),
       d2fdydz=d/dy(dfdz), d3=d/dx(d2fdydz) */
    if (nx > 2) fd_axis(g, g->arr[0], g->arr[1], 0, g->delx);
    if (ny > 2) fd_axis(g, g->arr[0], g->arr[2], 1, g->dely);
    if (nz > 2) fd_axis(g, g->arr[0], g->arr[3], 2, g->delz);
    if (nx > 2 && ny > 2) fd_axis(g, g->arr[2], g->arr[4], 0, g->delx);
    if (nx > 2 && nz > 2) fd_axis(g, g->arr[3], g->arr[5], 0, g->delx);
    if (ny > 2 && nz > 2) fd_axis(g, g->arr[3], g->arr[6], 1, g->dely);
    if (nx > 2 && ny > 2 && nz > 2) fd_axis(g, g->arr[6], g->arr[7], 0, g->delx);
  }
  return 0;
}

so_model *so_model_create_interp(int nspec, int nx, int ny, int nz, const double bounds[6],
                                 const double *qs, const double *ms, const double *F, int yearday,
                                 int msec) {
  if (nspec < 1 || nspec > SO_MAXSPEC) return NULL;
  so_model *m = (so_model *)calloc(1, sizeof *m);
  m->kind = 3;
  m->nspec = nspec;
  so_grid *g = &m->grid;
  g->nspec = nspec;
  g->nx = nx;
  g->ny = ny;
  g->nz = nz;
  g->minx = bounds[0];
  g->maxx = bounds[1];
  g->miny = bounds[2];
  g->maxy = bounds[3];
  g->minz = bounds[4];
  g->maxz = bounds[5];
  for (int s = 0; s < nspec; s++) {
    g->qs[s] = qs[s];
    g->ms[s] = ms[s];
  }
  size_t n = (size_t)nspec * nx * ny * nz;
  g->arr[0] = (double *)malloc(sizeof(double) * n);
  memcpy(g->arr[0], F, sizeof(double) * n);
  g->have_derivs = 0;
  grid_finish(g);
  so_dipole_tilt(yearday, msec, &m->mu);
  return m;
}

/* interp_dens_model_adapter.f95:52-134 setup */
so_model *so_model_create_interp_file(const char *gridfile, int yearday, int msec) {
  ldr r;
  r.f = fopen(gridfile, "r");
  if (!r.f) return NULL;
  double v[16];
  if (ldr_read(&r, 5, v) != 5) {
    fclose(r.f);
    return NULL;
  }
  int compder = (int)v[0], nspec = (int)v[1], nx = (int)v[2], ny = (int)v[3], nz = (int)v[4];
  if (nspec < 1 || nspec > SO_MAXSPEC) {
    fclose(r.f);
    return NULL;
  }
  so_model *m = (so_model *)calloc(1, sizeof *m);
  m->kind = 3;
  m->nspec = nspec;
  so_grid *g = &m->grid;
  g->nspec = nspec;
  g->nx = nx;
  g->ny = ny;
  g->nz = nz;
  ldr_read(&r, 6, v);
  g->minx = v[0];
  g->maxx = v[1];
  g->miny = v[2];
  g->maxy = v[3];
  g->minz = v[4];
  g->maxz = v[5];
  ldr_read(&r, nspec, g->qs);
  ldr_read(&r, nspec, g->ms);
  size_t ncell = (size_t)nx * ny * nz, n = ncell * nspec;
  g->arr[0] = (double *)malloc(sizeof(double) * n);
  for (size_t c = 0; c < ncell; c++) ldr_read(&r, nspec, g->arr[0] + c * nspec); /* one record per node */
  g->have_derivs = 0;
  if (compder == 1) {
    g->have_derivs = 1;
    for (int a = 1; a < 8; a++) {
      g->arr[a] = (double *)malloc(sizeof(double) * n);
      ldr_read(&r, (int)n, g->arr[a]); /* read(infile,*) dat%dfdx : whole array, list-directed */
    }
  }
  fclose(r.f);
  grid_finish(g);
  so_dipole_tilt(yearday, msec, &m->mu);
  return m;
}

/* maxloc((/1..n/), mask = 0 <= (xi - x))  (libtricubic.f95:835-840): last 1-based index whose
   node is <= xi, 0 when none */
static int cell_index(const double *x, int n, double xi) {
  int best = 0;
  for (int i = 0; i < n; i++)
    if (0.0 <= (xi - x[i])) best = i + 1;
  return best;
}

/* libtricubic.f95:796-933 tricubic_interpolate_at with derx=dery=derz=0, one species */
static double tricubic_interpolate_at(const so_grid *g, int s, double xi, double yi, double zi) {
  int flagi = 0, flagj = 0, flagk = 0;
  int nx = g->nx, ny = g->ny, nz = g->nz;
  double dx = g->delx, dy = g->dely, dz = g->delz;
  int is0 = cell_index(g->x, nx, xi);
  int js0 = cell_index(g->y, ny, yi);
  int ks0 = cell_index(g->z, nz, zi);
  double xil, yil, zil;
  if (is0 >= 1 && is0 < nx) xil = (xi - g->x[is0 - 1]) / dx; else xil = 0.0;
  if (js0 >= 1 && js0 < ny) yil = (yi - g->y[js0 - 1]) / dy; else yil = 0.0;
  if (ks0 >= 1 && ks0 < nz) zil = (zi - g->z[ks0 - 1]) / dz; else zil = 0.0;
  double b[64];
  for (int l = 0; l < 8; l++) {
    int io = l & 1, jo = (l >> 1) & 1, ko = (l >> 2) & 1; /* point2xyz :592-636 */
    int it = is0 + io, jt = js0 + jo, kt = ks0 + ko;
    if (it < 1) { it = 1; flagi = 1; }
    if (it > nx) { it = nx; flagi = 1; }
    if (jt < 1) { jt = 1; flagj = 1; }
    if (jt > ny) { jt = ny; flagj = 1; }
    if (kt < 1) { kt = 1; flagk = 1; }
    if (kt > nz) { kt = nz; flagk = 1; }
    size_t id = IDX(g, s, it - 1, jt - 1, kt - 1);
    double f = g->arr[0][id];
    double fx = g->arr[1][id] * dx;
    double fy = g->arr[2][id] * dy;
    double fz = g->arr[3][id] * dz;
    double fxy = g->arr[4][id] * dx * dy;
    double fxz = g->arr[5][id] * dx * dz;
    double fyz = g->arr[6][id] * dy * dz;
    double fxyz = g->arr[7][id] * dx * dy * dz;
    /* sticky flags: never reset inside the corner loop (Appendix A-7) */
    if (flagi == 1) { fx = 0.0; fxy = 0.0; fxz = 0.0; fxyz = 0.0; }
    if (flagj == 1) { fy = 0.0; fxy = 0.0; fyz = 0.0; fxyz = 0.0; }
    if (flagk == 1) { fz = 0.0; fxz = 0.0; fyz = 0.0; fxyz = 0.0; }
    b[0 + l] = f;
    b[8 + l] = fx;
    b[16 + l] = fy;
    b[24 + l] = fz;
    b[32 + l] = fxy;
    b[40 + l] = fxz;
    b[48 + l] = fyz;
    b[56 + l] = fxyz;
  }
  /* a = matmul(Amat, b) :715-720 */
  double a[64];
  for (int r = 0; r < 64; r++) {
    double acc = 0.0;
    for (int e = TRI_PTR[r]; e < TRI_PTR[r + 1]; e++) acc = acc + (double)TRI_VAL[e] * b[(int)TRI_COL[e]];
    a[r] = acc;
  }
  /* tricubic_eval :658-695 with derx=dery=derz=0 */
  double px[4] = {1.0, xil, xil * xil, xil * (xil * xil)};
  double py[4] = {1.0, yil, yil * yil, yil * (yil * yil)};
  double pz[4] = {1.0, zil, zil * zil, zil * (zil * zil)};
  double val = 0.0;
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++)
      for (int k = 0; k < 4; k++) {
        double cont = a[i + 4 * j + 16 * k] * px[i] * py[j] * pz[k];
        val = val + cont;
      }
  return val;
}

static void interp_params(so_model *m, const double x[3], double qs[4], double Ns[4], double ms[4],
                          double nus[4]) {
  so_grid *g = &m->grid;
  for (int s = 0; s < 4; s++) qs[s] = Ns[s] = ms[s] = nus[s] = 0.0;
  for (int s = 0; s < g->nspec; s++) {
    double v = tricubic_interpolate_at(g, s, x[0], x[1], x[2]);
    Ns[s] = exp(v); /* interpolated in log scale (:206) */
    qs[s] = g->qs[s];
    ms[s] = g->ms[s];
    nus[s] = 0.0;
  }
}

/* ================================================================== plugin dispatch */
void so_plasma_params(so_model *m, const double x[3], double qs[4], double Ns[4], double ms[4],
                      double nus[4], double B0[3]) {
  switch (m->kind) {
  case 1: ngo_params(m, x, qs, Ns, ms, nus); break;
  case 3: interp_params(m, x, qs, Ns, ms, nus); break;
  case 4: so_scattered_params(m, x, qs, Ns, ms, nus); break;
  }
  so_bfield(m, x, B0);
}
int so_model_nspec(const so_model *m) { return m->nspec; }
int so_model_kind(const so_model *m) { return m->kind; }
void so_model_destroy(so_model *m) {
  if (!m) return;
  if (m->kind == 3) {
    for (int a = 0; a < 8; a++) free(m->grid.arr[a]);
    free(m->grid.x);
    free(m->grid.y);
    free(m->grid.z);
  }
  if (m->kind == 4) so_scattered_free(m);
  free(m);
}

/* ================================================================== dispersion physics */
/* raytracer.f95:81-102 */
void so_stix_parameters(double w, int nspec, const double *qs, const double *Ns, const double *ms,
                        double B0mag, double *S, double *D, double *P, double *R, double *L) {
  double sr = 0.0, sl = 0.0, sp = 0.0;
  for (int s = 0; s < nspec; s++) {
    double wps2 = (Ns[s] * (qs[s] * qs[s]) / ms[s] / EPS0);
    double wcs = ((qs[s] * B0mag) / ms[s]);
    sr = sr + wps2 / (w * (w + wcs));
    sl = sl + wps2 / (w * (w - wcs));
    sp = sp + wps2 / (w * w);
  }
  *R = 1.0 - sr;
  *L = 1.0 - sl;
  *P = 1.0 - sp;
  *S = 1.0 / 2.0 * (*R + *L);
  *D = 1.0 / 2.0 * (*R - *L);
}

/* raytracer.f95:41-72 */
double so_dispersion_relation(const double n[3], double w, int nspec, const double *qs,
                              const double *Ns, const double *ms, const double B0[3]) {
  double nmag2 = dot3(n, n);
  double cos2phi = (dot3(n, B0) * dot3(n, B0)) / (dot3(n, n) * dot3(B0, B0));
  double sin2phi = 1.0 - cos2phi;
  double S, D, P, R, L;
  so_stix_parameters(w, nspec, qs, Ns, ms, sqrt(dot3(B0, B0)), &S, &D, &P, &R, &L);
  double A = S * sin2phi + P * cos2phi;
  double B = R * L * sin2phi + P * S * (1.0 + cos2phi);
  /* :65 (mis-parenthesised threshold, Appendix A-5) */
  double maxN = Ns[0], maxq = fabs(qs[0]), minm = ms[0];
  for (int s = 1; s < nspec; s++) {
    if (Ns[s] > maxN) maxN = Ns[s];
    if (fabs(qs[s]) > maxq) maxq = fabs(qs[s]);
    if (ms[s] < minm) minm = ms[s];
  }
  if (w > 100.0 * sqrt((maxN * (maxq * maxq))) / (minm * EPS0)) return -nmag2 + 1.0;
  return A * (nmag2 * nmag2) - B * nmag2 + R * L * P;
}

/* raytracer.f95:118-155 with plasma parameters already evaluated at x */
static void dfdk_at(const double k[3], double w, double del, int nspec, const double *qs,
                    const double *Ns, const double *ms, const double B0[3], double out[3]) {
  for (int c = 0; c < 3; c++) {
    double d = fmax_f(del * fabs(k[c]), del);
    double np[3], nm[3];
    /* dkx = d*(/1,0,0/): the other components see k +- 0 */
    double C = c_light();
    for (int i = 0; i < 3; i++) {
      double dk = (i == c) ? d : d * 0.0;
      np[i] = (k[i] + dk) * C / w;
      nm[i] = (k[i] - dk) * C / w;
    }
    out[c] = (so_dispersion_relation(np, w, nspec, qs, Ns, ms, B0) -
              so_dispersion_relation(nm, w, nspec, qs, Ns, ms, B0)) / d / 2.0;
  }
}
void so_dfdk(so_model *m, const double k[3], double w, const double x[3], double del, double out[3]) {
  double qs[4], Ns[4], ms[4], nus[4], B0[3];
  so_plasma_params(m, x, qs, Ns, ms, nus, B0);
  dfdk_at(k, w, del, m->nspec, qs, Ns, ms, B0, out);
}
/* raytracer.f95:172-198 */
double so_dfdw(so_model *m, const double k[3], double w, const double x[3], double del) {
  double qs[4], Ns[4], ms[4], nus[4], B0[3];
  double C = c_light();
  so_plasma_params(m, x, qs, Ns, ms, nus, B0);
  double d = fmax_f(del * fabs(w), del);
  double np[3], nm[3];
  for (int i = 0; i < 3; i++) {
    np[i] = k[i] * C / (w + d);
    nm[i] = k[i] * C / (w - d);
  }
  return (so_dispersion_relation(np, (w + d), m->nspec, qs, Ns, ms, B0) -
          so_dispersion_relation(nm, (w - d), m->nspec, qs, Ns, ms, B0)) / d / 2.0;
}
/* raytracer.f95:215-263 */
void so_dfdx(so_model *m, const double k[3], double w, const double x[3], double del, double out[3]) {
  double qs[4], Ns[4], ms[4], nus[4], B0[3];
  double C = c_light();
  double n[3];
  for (int i = 0; i < 3; i++) n[i] = k[i] * C / w;
  for (int c = 0; c < 3; c++) {
    double d = fmax_f(del * fabs(x[c]), del);
    double xp[3], xm[3];
    for (int i = 0; i < 3; i++) {
      double dd = (i == c) ? d : d * 0.0;
      xp[i] = x[i] + dd;
      xm[i] = x[i] - dd;
    }
    so_plasma_params(m, xp, qs, Ns, ms, nus, B0);
    double Fp = so_dispersion_relation(n, w, m->nspec, qs, Ns, ms, B0);
    so_plasma_params(m, xm, qs, Ns, ms, nus, B0);
    double Fn = so_dispersion_relation(n, w, m->nspec, qs, Ns, ms, B0);
    out[c] = (Fp - Fn) / d / 2.0;
  }
}
/* raytracer.f95:282-314 */
void so_evalrhs(so_model *m, const double args[7], double del, double rhs[7]) {
  const double *x = args, *k = args + 3;
  double w = args[6];
  double dfdk[3], dfdx[3];
  so_dfdk(m, k, w, x, 1.0e-8, dfdk);
  double dfdw = so_dfdw(m, k, w, x, 1.0e-8);
  so_dfdx(m, k, w, x, del, dfdx);
  for (int i = 0; i < 3; i++) {
    rhs[i] = -(dfdk[i] / dfdw);
    rhs[3 + i] = dfdx[i] / dfdw;
  }
  rhs[6] = 0.0;
}

/* ------------------------------------------------------------------ is_right_handed
 * raytracer.f95:355-405 builds the 3x3 complex wave matrix (entries rounded through default-kind
 * cmplx => float32; phi arrives in DEGREES but is fed to cos/sin as is -- :450, Appendix A-4),
 * takes E = row 3 of V^H from zgesvd (smallest singular value) and tests the sense of rotation
 * from Re(E) to Re(iE) in the x-y plane.
 *
 * M = [[a,-iD,b],[iD,c,0],[b,0,d]] is Hermitian and unitarily similar (diag(1,i,1)) to the real
 * symmetric T = [[a,D,b],[D,c,0],[b,0,d]]; its singular values are |eigenvalues of T| and the
 * right singular vector of the smallest one is the eigenvector v of the least-|lambda| eigenvalue.
 * Row 2 of (M - lambda I) v = 0 gives E2/E1 = i D/(c - lambda) for E = conj(v), whence the x-y
 * rotation from Re(E) to Re(iE) has the sign of -D/(c-lambda), independent of the arbitrary phase
 * zgesvd leaves on E.  angle >= 0  <=>  not( D/(c-lambda) > 0 ).  Checked against the reference's
 * own is_right_handed (LAPACK path) on random and on-trajectory tuples in tests/test_oracle_vs_ref.py.
 */
static void sym3_eigenvalues(double a, double D, double b, double c, double d, double ev[3]) {
  /* cyclic Jacobi on T; entries are O(1..1e6), 3x3 => converges in a few sweeps */
  double A[3][3] = {{a, D, b}, {D, c, 0.0}, {b, 0.0, d}};
  for (int sweep = 0; sweep < 60; sweep++) {
    double off = fabs(A[0][1]) + fabs(A[0][2]) + fabs(A[1][2]);
    double diag = fabs(A[0][0]) + fabs(A[1][1]) + fabs(A[2][2]);
    if (off <= 1e-300 || off <= 1e-18 * diag) break;
    for (int p = 0; p < 2; p++)
      for (int q = p + 1; q < 3; q++) {
        if (A[p][q] == 0.0) continue;
        double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
        double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        double cs = 1.0 / sqrt(t * t + 1.0), sn = t * cs;
        for (int r = 0; r < 3; r++) { /* A <- A J */
          double arp = A[r][p], arq = A[r][q];
          A[r][p] = cs * arp - sn * arq;
          A[r][q] = sn * arp + cs * arq;
        }
        for (int r = 0; r < 3; r++) { /* A <- J^T A */
          double apr = A[p][r], aqr = A[q][r];
          A[p][r] = cs * apr - sn * aqr;
          A[q][r] = sn * apr + cs * aqr;
        }
      }
  }
  ev[0] = A[0][0];
  ev[1] = A[1][1];
  ev[2] = A[2][2];
}
int so_is_right_handed(double n2, double phi, double S, double D, double P) {
  double cp = cos(phi), sp = sin(phi);
  /* form_dispersion_matrix :355-371, float32 rounding from default-kind cmplx */
  double a = (double)(float)(S - n2 * (cp * cp));
  double Dm = (double)(float)(D);
  double b = (double)(float)(n2 * cp * sp);
  double c = (double)(float)(S - n2);
  double d = (double)(float)(P - n2 * (sp * sp));
  double ev[3];
  sym3_eigenvalues(a, Dm, b, c, d, ev);
  double lam = ev[0];
  if (fabs(ev[1]) < fabs(lam)) lam = ev[1];
  if (fabs(ev[2]) < fabs(lam)) lam = ev[2];
  return !(Dm / (c - lam) > 0.0);
}

/* raytracer.f95:408-502 */
void so_solve_dispersion_relation(so_model *m, const double k[3], double w, const double x[3],
                                  double k1[2], double k2[2]) {
  double qs[4], Ns[4], ms[4], nus[4], B0[3];
  double C = c_light();
  so_plasma_params(m, x, qs, Ns, ms, nus, B0);
  double cos2phi = (dot3(k, B0) * dot3(k, B0)) / (dot3(k, k) * dot3(B0, B0));
  double sin2phi = 1.0 - cos2phi;
  double phi = acos(sqrt(cos2phi)) * R2D;
  double B0mag = sqrt(dot3(B0, B0));
  double S, D, P, R, L;
  so_stix_parameters(w, m->nspec, qs, Ns, ms, B0mag, &S, &D, &P, &R, &L);
  double A = S * sin2phi + P * cos2phi;
  double B = R * L * sin2phi + P * S * (1.0 + cos2phi);
  zc discriminant = zreal(B * B - 4.0 * A * R * L * P);
  zc sq = zsqrt(discriminant);
  zc bp = {B + sq.re, 0.0 + sq.im}, bm = {B - sq.re, 0.0 - sq.im};
  zc nsq1 = zdiv(bp, zreal(2.0 * A));
  zc nsq2 = zdiv(bm, zreal(2.0 * A));
  zc n1 = zsqrt(nsq1);
  zc n2 = zsqrt(nsq2);
  zc kk1 = zdiv(zmul(zreal(w), n1), zreal(C));
  zc kk2 = zdiv(zmul(zreal(w), n2), zreal(C));
  if (n1.re > 0.0 && so_is_right_handed(nsq1.re, phi, S, D, P)) {
    kk1 = zdiv(zmul(zreal(w), n2), zreal(C));
    kk2 = zdiv(zmul(zreal(w), n1), zreal(C));
  }
  k1[0] = kk1.re;
  k1[1] = kk1.im;
  k2[0] = kk2.re;
  k2[1] = kk2.im;
}

/* ================================================================== integrator */
/* raytracer.f95:504-532 */
void so_rk4(so_model *m, const double x[7], double del, double dt, double out[7]) {
  double k1[7], k2[7], k3[7], k4[7], tmp[7], r[7];
  so_evalrhs(m, x, del, r);
  for (int i = 0; i < 7; i++) k1[i] = dt * r[i];
  for (int i = 0; i < 7; i++) tmp[i] = x[i] + 1.0 / 2.0 * k1[i];
  so_evalrhs(m, tmp, del, r);
  for (int i = 0; i < 7; i++) k2[i] = dt * r[i];
  for (int i = 0; i < 7; i++) tmp[i] = x[i] + 1.0 / 2.0 * k2[i];
  so_evalrhs(m, tmp, del, r);
  for (int i = 0; i < 7; i++) k3[i] = dt * r[i];
  for (int i = 0; i < 7; i++) tmp[i] = x[i] + k3[i];
  so_evalrhs(m, tmp, del, r);
  for (int i = 0; i < 7; i++) k4[i] = dt * r[i];
  for (int i = 0; i < 7; i++)
    out[i] = x[i] + 1.0 / 6.0 * (k1[i] + 2.0 * k2[i] + 2.0 * k3[i] + k4[i]);
}

/* raytracer.f95:8-27 tableau, :534-596 rk45 */
void so_rk45(so_model *m, const double x[7], double del, double dt, double out4[7], double out5[7]) {
  static const double a2[1] = {1.0 / 4.0};
  static const double a3[2] = {3.0 / 32.0, 9.0 / 32.0};
  static const double a4[3] = {1932.0 / 2197.0, -7200.0 / 2197.0, 7296.0 / 2197.0};
  static const double a5[4] = {439.0 / 216.0, -8.0, 3680.0 / 513.0, -845.0 / 4104.0};
  static const double a6[5] = {-8.0 / 27.0, 2.0, -3544.0 / 2565.0, 1859.0 / 4104.0, -11.0 / 40.0};
  static const double b4[6] = {25.0 / 216.0, 0.0, 1408.0 / 2565.0, 2197.0 / 4104.0, -1.0 / 5.0, 0.0};
  static const double b5[6] = {16.0 / 135.0, 0.0, 6656.0 / 12825.0, 28561.0 / 56430.0, -9.0 / 50.0,
                               2.0 / 55.0};
  double k1[7], k2[7], k3[7], k4[7], k5[7], k6[7], tmp[7], r[7];
  so_evalrhs(m, x, del, r);
  for (int i = 0; i < 7; i++) k1[i] = dt * r[i];
  for (int i = 0; i < 7; i++) tmp[i] = x[i] + (a2[0] * k1[i]);
  so_evalrhs(m, tmp, del, r);
  for (int i = 0; i < 7; i++) k2[i] = dt * r[i];
  for (int i = 0; i < 7; i++) tmp[i] = x[i] + (a3[0] * k1[i] + a3[1] * k2[i]);
  so_evalrhs(m, tmp, del, r);
  for (int i = 0; i < 7; i++) k3[i] = dt * r[i];
  for (int i = 0; i < 7; i++) tmp[i] = x[i] + (a4[0] * k1[i] + a4[1] * k2[i] + a4[2] * k3[i]);
  so_evalrhs(m, tmp, del, r);
  for (int i = 0; i < 7; i++) k4[i] = dt * r[i];
  for (int i = 0; i < 7; i++)
    tmp[i] = x[i] + (a5[0] * k1[i] + a5[1] * k2[i] + a5[2] * k3[i] + a5[3] * k4[i]);
  so_evalrhs(m, tmp, del, r);
  for (int i = 0; i < 7; i++) k5[i] = dt * r[i];
  for (int i = 0; i < 7; i++)
    tmp[i] = x[i] + (a6[0] * k1[i] + a6[1] * k2[i] + a6[2] * k3[i] + a6[3] * k4[i] + a6[4] * k5[i]);
  so_evalrhs(m, tmp, del, r);
  for (int i = 0; i < 7; i++) k6[i] = dt * r[i];
  for (int i = 0; i < 7; i++) {
    out4[i] = x[i] + (b4[0] * k1[i] + b4[1] * k2[i] + b4[2] * k3[i] + b4[3] * k4[i] + b4[4] * k5[i] +
                      b4[5] * k6[i]);
    out5[i] = x[i] + (b5[0] * k1[i] + b5[1] * k2[i] + b5[2] * k3[i] + b5[3] * k4[i] + b5[4] * k5[i] +
                      b5[5] * k6[i]);
  }
}

/* raytracer.f95:324-353 */
static int stopconditions(const double pos[3], const double k[3], const double vgrel[3], double dt,
                          int nstep, int maxsteps, double minalt) {
  if (sqrt(dot3(pos, pos)) < minalt) return 1;
  if (sqrt(dot3(k, k)) == 0.0) return 2;
  if (sqrt(dot3(vgrel, vgrel)) > 1.0 + 1e-2) return 3;
  if (dt < (double)1e-14f) return 5; /* default-real literal (:343) */
  if (nstep >= maxsteps) return 6;
  return 0;
}

static void emit_row(so_model *m, double *rows, int capacity, int *nrows, double t, const double x[7],
                     double w, int first, double vg_out[3]) {
  double C = c_light();
  double dfdk[3], qs[4], Ns[4], ms[4], nus[4], B0[3];
  so_dfdk(m, x + 3, w, x, 1.0e-8, dfdk);
  double dfdw = so_dfdw(m, x + 3, w, x, 1.0e-8);
  so_plasma_params(m, x, qs, Ns, ms, nus, B0);
  double n[3], vp[3], vg[3];
  for (int i = 0; i < 3; i++) n[i] = x[3 + i] * C / w;
  double nn = dot3(n, n);
  if (!first || nn > 0) {
    for (int i = 0; i < 3; i++) {
      vp[i] = n[i] / nn;
      vg[i] = -(dfdk[i] / dfdw) / C;
    }
  } else {
    for (int i = 0; i < 3; i++) vp[i] = vg[i] = 0.0;
  }
  for (int i = 0; i < 3; i++) vg_out[i] = vg[i];
  if (*nrows < capacity) {
    double *r = rows + (size_t)(*nrows) * SO_ROW;
    r[0] = t;
    for (int i = 0; i < 3; i++) {
      r[1 + i] = x[i];
      r[4 + i] = vp[i];
      r[7 + i] = vg[i];
      r[10 + i] = n[i];
      r[13 + i] = B0[i];
    }
    for (int s = 0; s < 4; s++) r[16 + s] = Ns[s];
  }
  (*nrows)++;
}

/* raytracer.f95:609-995 */
int so_raytracer_run(so_model *m, const so_params *p, const double pos0[3], const double dir0_in[3],
                     double w0, double *rows, int capacity, int *nrows_total, int *stopcond_out) {
  double dir0[3] = {dir0_in[0], dir0_in[1], dir0_in[2]};
  double last_vg[3] = {0, 0, 0};
  int nrows = 0;
  /* field-aligned start :661-674 */
  if (dir0[0] == 0 && dir0[1] == 0 && dir0[2] == 0) {
    double qs[4], Ns[4], ms[4], nus[4], B0[3], sp[3], Bs[3];
    so_plasma_params(m, pos0, qs, Ns, ms, nus, B0);
    so_cartesian_to_spherical(pos0, sp);
    cartesian_to_spherical_vec(B0, sp[1], sp[2], Bs);
    Bs[0] = fabs(Bs[0]);
    spherical_to_cartesian_vec(Bs, sp[1], sp[2], B0);
    double nb = sqrt(dot3(B0, B0));
    for (int i = 0; i < 3; i++) dir0[i] = B0[i] / nb;
  }
  double k1[2], k2[2];
  so_solve_dispersion_relation(m, dir0, w0, pos0, k1, k2);
  const double *k0mag = (p->root == 1) ? k1 : k2;
  double x[7];
  for (int i = 0; i < 3; i++) {
    x[i] = pos0[i];
    /* real(k0mag*dir0) */
    zc km0 = {k0mag[0], k0mag[1]};
    x[3 + i] = zmul(km0, zreal(dir0[i])).re;
  }
  x[6] = w0;
  double dt = p->dt0, t = 0.0;
  int lastrefinedown = 0;
  emit_row(m, rows, capacity, &nrows, t, x, w0, 1, last_vg);
  int stopcond = 0;
  int nstep = 1;
  double w = 0.0; /* -finit-local-zero (Makefile:10); Appendix A-1 */
  int w_assigned = 0;
  for (;;) {
    if (t >= p->tmax) {
      stopcond = 0;
      break;
    }
    stopcond = stopconditions(x, x + 3, last_vg, dt, nstep, p->maxsteps, p->minalt);
    if (stopcond != 0) break;
    double est1[7], est2[7], dtincr;
    if (p->fixedstep == 0) {
      so_rk45(m, x, p->del, dt, est1, est2);
      dtincr = dt;
      double err;
      double s1 = 0, s2 = 0;
      for (int i = 3; i < 6; i++) {
        s1 = s1 + fabs(est1[i] - est2[i]);
        s2 = s2 + fabs(est2[i]);
      }
      double kterm = s1 / s2;
      if (!w_assigned && p->first_attempt_policy == 1) {
        err = kterm;
      } else {
        double d1[3], d2[3];
        so_dfdk(m, est1 + 3, w, est1, 1.0e-8, d1);
        so_dfdk(m, est2 + 3, w, est2, 1.0e-8, d2);
        double t1 = 0, t2 = 0;
        for (int i = 0; i < 3; i++) {
          t1 = t1 + fabs(d1[i] - d2[i]);
          t2 = t2 + fabs(d2[i]);
        }
        err = fmax_f(kterm, t1 / t2);
      }
      if (err > p->maxerr) {
        dt = 0.8 * dt;
        lastrefinedown = 1;
        continue;
      }
      if (lastrefinedown == 0 && err < p->maxerr / 100.0 && dt * 1.25 < p->dtmax) {
        dt = dt * 1.25;
        lastrefinedown = 0;
      }
    } else {
      so_rk4(m, x, p->del, dt, est2);
      dtincr = dt;
    }
    double cur_pos[3] = {est2[0], est2[1], est2[2]};
    double kr[3] = {est2[3], est2[4], est2[5]};
    w = est2[6];
    w_assigned = 1;
    so_solve_dispersion_relation(m, kr, w, cur_pos, k1, k2);
    const double *km = (p->root == 1) ? k1 : k2;
    /* k = kmag*(k/sqrt(dot_product(k,k))) in complex arithmetic; k is real here */
    zc kn = zsqrt(zreal(dot3(kr, kr)));
    zc kmz = {km[0], km[1]};
    double kre[3], kim[3], imsum = 0.0;
    for (int i = 0; i < 3; i++) {
      zc u = zmul(kmz, zdiv(zreal(kr[i]), kn));
      kre[i] = u.re;
      kim[i] = u.im;
    }
    imsum = (kim[0] * kim[0] + kim[1] * kim[1]) + kim[2] * kim[2];
    if (imsum > 0.0) {
      if (p->fixedstep == 0) {
        dt = dt / 2.0;
        lastrefinedown = 1;
        continue;
      } else {
        /* `return` with stopcond still 0 (:900-905) */
        stopcond = 0;
        break;
      }
    }
    for (int i = 0; i < 7; i++) x[i] = est2[i];
    for (int i = 0; i < 3; i++) x[3 + i] = kre[i];
    lastrefinedown = 0;
    t = t + dtincr;
    nstep = nstep + 1;
    {
      /* a non-finite state makes the reference `stop` the whole process inside csvd (blas.f95:208-211,
         SURVEY A-3); the batch API ends only this ray, with code 9, before the bad row is emitted */
      int finite = 1;
      for (int i = 0; i < 6; i++)
        if (!isfinite(x[i])) finite = 0;
      if (!finite) {
        stopcond = 9;
        break;
      }
    }
    emit_row(m, rows, capacity, &nrows, t, x, w, 0, last_vg);
  }
  *nrows_total = nrows;
  *stopcond_out = stopcond;
  return nrows < capacity ? nrows : capacity;
}

/* ------------------------------------------------------------------ batch driver (pthreads) */
typedef struct {
  so_model *m;
  const so_params *p;
  long lo, hi;
  const double *pos0, *dir0, *w0;
  double *rows;
  int capacity;
  int *nrows, *stopcond;
  long steps;
} batch_job;
static void *batch_worker(void *arg) {
  batch_job *j = (batch_job *)arg;
  so_model *m = j->m;
  so_model local;
  struct so_scattered lsc;
  if (m->kind == 1) { /* Ngo model has per-call scratch state: give each thread its own copy */
    local = *m;
    m = &local;
  } else if (m->kind == 4) { /* scattered model: per-thread neighbour scratch buffer */
    local = *m;
    lsc = *m->sc;
    lsc.found = NULL;
    lsc.cap = 0;
    local.sc = &lsc;
    m = &local;
  }
  j->steps = 0;
  for (long r = j->lo; r < j->hi; r++) {
    int nt, sc;
    so_raytracer_run(m, j->p, j->pos0 + 3 * r, j->dir0 + 3 * r, j->w0[r],
                     j->rows ? j->rows + (size_t)r * j->capacity * SO_ROW : NULL,
                     j->rows ? j->capacity : 0, &nt, &sc);
    j->nrows[r] = nt;
    j->stopcond[r] = sc;
    j->steps += nt - 1;
  }
  if (m->kind == 4) free(lsc.found);
  return NULL;
}
long so_trace_batch(so_model *m, const so_params *p, long nrays, const double *pos0,
                    const double *dir0, const double *w0, double *rows, int capacity, int *nrows,
                    int *stopcond, int nthreads) {
  if (nthreads < 1) nthreads = 1;
  if (nthreads > 256) nthreads = 256;
  pthread_t th[256];
  batch_job jobs[256];
  long per = (nrays + nthreads - 1) / nthreads;
  int used = 0;
  for (int i = 0; i < nthreads; i++) {
    long lo = i * per, hi = lo + per;
    if (lo >= nrays) break;
    if (hi > nrays) hi = nrays;
    jobs[i] = (batch_job){m, p, lo, hi, pos0, dir0, w0, rows, capacity, nrows, stopcond, 0};
    used++;
  }
  if (used == 1) {
    batch_worker(&jobs[0]);
  } else {
    for (int i = 0; i < used; i++) pthread_create(&th[i], NULL, batch_worker, &jobs[i]);
    for (int i = 0; i < used; i++) pthread_join(th[i], NULL);
  }
  long steps = 0;
  for (int i = 0; i < used; i++) steps += jobs[i].steps;
  return steps;
}
