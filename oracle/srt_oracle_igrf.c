/* srt_oracle_igrf.c -- TEST INFRASTRUCTURE ONLY (see srt_oracle.h).
 *
 * CPU restatement of the use_igrf = 1 branch of the adapters' field tail (interp_dens_model_adapter.f95:214-241 and
 * its twins in the ngo / scattered adapters): tsy_recalc -> RECALC_08 (tsyganenko/geopack0508_adapter.for:21-30,
 * geopack2008.for:486-1196) with the solar-wind velocity (-400,0,0) (GSW == GSM), SUN_08 (:333-381), and
 * IGRF_GSM -> IGRF_GSW_08 (:55-185) with GEOGSW_08 (:1421-1457).  Default REAL (fp32) throughout, like the Fortran;
 * DOUBLE PRECISION only where SUN_08 declares it.  The Gauss coefficients are DATA (published IAGA numbers), read
 * from a table file (stanford_raytracer_amd/data/igrf_coeffs.txt).
 * Parity: PINNED against the reference build (oracle/_ref/ref_harness --use_igrf=1; tests/golden/igrf_golden.npz).
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "srt_oracle.h"
#include "srt_oracle_internal.h"

/* table: 12 epochs (1965..2020) + secular variation, entries 1..105 */
typedef struct {
  float g[13][105], h[13][105];
} igrf_table;

static int read_table(const char *path, igrf_table *t) {
  FILE *f = fopen(path, "r");
  if (!f) return -1;
  char line[1024];
  int seen = 0;
  memset(t, 0, sizeof *t);
  while (fgets(line, sizeof line, f)) {
    if (line[0] != 'g' && line[0] != 'h') continue;
    char *s = line + 1;
    int mn = (int)strtol(s, &s, 10);
    if (mn < 1 || mn > 105) continue;
    for (int e = 0; e < 13; ++e) {
      float v = strtof(s, &s);
      if (line[0] == 'g') t->g[e][mn - 1] = v;
      else t->h[e][mn - 1] = v;
    }
    ++seen;
  }
  fclose(f);
  return seen == 210 ? 0 : -1;
}

/* SUN_08, geopack2008.for:333-381 (only GST, SRASN, SDEC are used by RECALC_08) */
static void sun_08(int iyear, int iday, int ihour, int min, int isec, float *gst, float *slong, float *srasn, float *sdec) {
  const float RAD = 57.295779513f;
  if (iyear < 1901 || iyear > 2099) return;
  double fday = (double)(ihour * 3600 + min * 60 + isec) / 86400.0;
  double dj = 365 * (iyear - 1900) + (iyear - 1901) / 4 + iday - 0.5 + fday;
  float t = (float)(dj / (double)36525.f);
  float vl = (float)fmod((double)279.696678f + (double)0.9856473354f * dj, 360.0);
  *gst = (float)(fmod((double)279.690983f + (double).9856473354f * dj + (double)360.f * fday + (double)180.f, 360.0) / (double)RAD);
  float g = (float)(fmod((double)358.475845f + (double)0.985600267f * dj, 360.0) / (double)RAD);
  *slong = (vl + (1.91946f - 0.004789f * t) * sinf(g) + 0.020094f * sinf(2.f * g)) / RAD;
  if (*slong > 6.2831853f) *slong = *slong - 6.2831853f;
  if (*slong < 0.f) *slong = *slong + 6.2831853f;
  float obliq = (23.45229f - 0.0130125f * t) / RAD;
  float sob = sinf(obliq);
  float slp = *slong - 9.924e-5f;
  float sind = sob * sinf(slp);
  float cosd = sqrtf(1.f - sind * sind);
  float sc = sind / cosd;
  *sdec = atanf(sc);
  *srasn = 3.141592654f - atan2f(cosf(obliq) / sob * sc, -cosf(slp) / cosd);
}

/* RECALC_08 with VGSE = (-400,0,0): G, H, REC and the GEO->GSW matrix A (row-major A11 A12 A13 A21 ...) */
void so_igrf_recalc(const igrf_table *tab, int iyear, int iday, int ihour, int min, int isec, float *G, float *H, float *REC,
                    float A[9]) {
  int iy = iyear;
  if (iy < 1965) iy = 1965;
  if (iy > 2025) iy = 2025;
  for (int n = 1; n <= 14; ++n) {
    int n2 = 2 * n - 1;
    n2 = n2 * (n2 - 2);
    for (int m = 1; m <= n; ++m) {
      int mn = n * (n - 1) / 2 + m;
      REC[mn - 1] = (float)((n - m) * (n + m - 2)) / (float)n2;
    }
  }
  if (iy >= 2020) { /* extrapolate with the secular variation, degrees <= 8 (entries <= 45) */
    float dt = (float)iy + (float)(iday - 1) / 365.25f - 2020.f;
    for (int n = 0; n < 105; ++n) {
      G[n] = tab->g[11][n];
      H[n] = tab->h[11][n];
      if (n + 1 > 45) continue;
      G[n] = G[n] + tab->g[12][n] * dt;
      H[n] = H[n] + tab->h[12][n] * dt;
    }
  } else {
    int e = (iy - 1965) / 5; /* interpolate between epoch e and e+1 */
    float f2 = ((float)iy + (float)(iday - 1) / 365.25f - (float)(1965 + 5 * e)) / 5.f;
    float f1 = 1.f - f2;
    for (int n = 0; n < 105; ++n) {
      G[n] = tab->g[e][n] * f1 + tab->g[e + 1][n] * f2;
      H[n] = tab->h[e][n] * f1 + tab->h[e + 1][n] * f2;
    }
  }
  /* Schmidt normalisation, :1012-1029 */
  float s = 1.f;
  for (int n = 2; n <= 14; ++n) {
    int mn = n * (n - 1) / 2 + 1;
    s = s * (float)(2 * n - 3) / (float)(n - 1);
    G[mn - 1] = G[mn - 1] * s;
    H[mn - 1] = H[mn - 1] * s;
    float p = s;
    for (int m = 2; m <= n; ++m) {
      float aa = 1.f;
      if (m == 2) aa = 2.f;
      p = p * sqrtf(aa * (float)(n - m + 1) / (float)(n + m - 2));
      int mnn = mn + m - 1;
      G[mnn - 1] = G[mnn - 1] * p;
      H[mnn - 1] = H[mnn - 1] * p;
    }
  }
  float g10 = -G[1], g11 = G[2], h11 = H[2];
  float sq = g11 * g11 + h11 * h11;
  float sqq = sqrtf(sq);
  float sqr = sqrtf(g10 * g10 + sq);
  float sl0 = -h11 / sqq, cl0 = -g11 / sqq, st0 = sqq / sqr, ct0 = g10 / sqr;
  float stcl = st0 * cl0, stsl = st0 * sl0;
  float gst = 0, slong = 0, srasn = 0, sdec = 0;
  sun_08(iy, iday, ihour, min, isec, &gst, &slong, &srasn, &sdec);
  float s1 = cosf(srasn) * cosf(sdec), s2 = sinf(srasn) * cosf(sdec), s3 = sinf(sdec);
  float dj = (float)(365 * (iy - 1900) + (iy - 1901) / 4 + iday) - 0.5f + (float)(ihour * 3600 + min * 60 + isec) / 86400.f;
  float t = dj / 36525.f;
  float obliq = (23.45229f - 0.0130125f * t) / 57.2957795f;
  float dz1 = 0.f, dz2 = -sinf(obliq), dz3 = cosf(obliq);
  float dy1 = dz2 * s3 - dz3 * s2, dy2 = dz3 * s1 - dz1 * s3, dy3 = dz1 * s2 - dz2 * s1;
  const float vx = -400.f, vy = 0.f, vz = 0.f;
  float v = sqrtf(vx * vx + vy * vy + vz * vz);
  float dx1 = -vx / v, dx2 = -vy / v, dx3 = -vz / v;
  float x1 = dx1 * s1 + dx2 * dy1 + dx3 * dz1;
  float x2 = dx1 * s2 + dx2 * dy2 + dx3 * dz2;
  float x3 = dx1 * s3 + dx2 * dy3 + dx3 * dz3;
  float cgst = cosf(gst), sgst = sinf(gst);
  float dip1 = stcl * cgst - stsl * sgst, dip2 = stcl * sgst + stsl * cgst, dip3 = ct0;
  float y1 = dip2 * x3 - dip3 * x2, y2 = dip3 * x1 - dip1 * x3, y3 = dip1 * x2 - dip2 * x1;
  float y = sqrtf(y1 * y1 + y2 * y2 + y3 * y3);
  y1 = y1 / y;
  y2 = y2 / y;
  y3 = y3 / y;
  float z1 = x2 * y3 - x3 * y2, z2 = x3 * y1 - x1 * y3, z3 = x1 * y2 - x2 * y1;
  A[0] = x1 * cgst + x2 * sgst;  /* A11 */
  A[1] = -x1 * sgst + x2 * cgst; /* A12 */
  A[2] = x3;                     /* A13 */
  A[3] = y1 * cgst + y2 * sgst;  /* A21 */
  A[4] = -y1 * sgst + y2 * cgst; /* A22 */
  A[5] = y3;                     /* A23 */
  A[6] = z1 * cgst + z2 * sgst;  /* A31 */
  A[7] = -z1 * sgst + z2 * cgst; /* A32 */
  A[8] = z3;                     /* A33 */
}

/* IGRF_GSW_08, geopack2008.for:55-185 */
void so_igrf_gsw(const float *G, const float *H, const float *REC, const float A[9], float xgsw, float ygsw, float zgsw, float *hx,
                 float *hy, float *hz) {
  float a[15], b[15];
  /* GEOGSW_08 J = -1 */
  float xgeo = A[0] * xgsw + A[3] * ygsw + A[6] * zgsw;
  float ygeo = A[1] * xgsw + A[4] * ygsw + A[7] * zgsw;
  float zgeo = A[2] * xgsw + A[5] * ygsw + A[8] * zgsw;
  float rho2 = xgeo * xgeo + ygeo * ygeo;
  float r = sqrtf(rho2 + zgeo * zgeo);
  float c = zgeo / r;
  float rho = sqrtf(rho2);
  float s = rho / r;
  float cf, sf;
  if (s < 1.e-5f) {
    cf = 1.f;
    sf = 0.f;
  } else {
    cf = xgeo / rho;
    sf = ygeo / rho;
  }
  float pp = 1.f / r;
  float p = pp;
  int irp3 = (int)(r + 2);
  int nm = 3 + 30 / irp3;
  if (nm > 13) nm = 13;
  int k = nm + 1;
  for (int n = 1; n <= k; ++n) {
    p = p * pp;
    a[n] = p;
    b[n] = p * n;
  }
  p = 1.f;
  float d = 0.f, bbr = 0.f, bbt = 0.f, bbf = 0.f;
  float x = 0.f, y = 0.f;
  for (int m = 1; m <= k; ++m) {
    int mm = 0;
    if (m == 1) {
      x = 0.f;
      y = 1.f;
    } else {
      mm = m - 1;
      float w = x;
      x = w * cf + y * sf;
      y = y * cf - w * sf;
    }
    float q = p, z = d, bi = 0.f, p2 = 0.f, d2 = 0.f;
    for (int n = m; n <= k; ++n) {
      float an = a[n];
      int mn = n * (n - 1) / 2 + m;
      float e = G[mn - 1], hh = H[mn - 1];
      float w = e * y + hh * x;
      bbr = bbr + b[n] * w * q;
      bbt = bbt - an * w * z;
      if (m != 1) {
        float qq = q;
        if (s < 1.e-5f) qq = z;
        bi = bi + an * (e * x - hh * y) * qq;
      }
      float xk = REC[mn - 1];
      float dp = c * z - s * q - xk * d2;
      float pm = c * q - xk * p2;
      d2 = z;
      p2 = q;
      z = dp;
      q = pm;
    }
    d = s * d + c * p;
    p = s * p;
    if (m == 1) continue;
    bi = bi * mm;
    bbf = bbf + bi;
  }
  float br = bbr, bt = bbt, bf;
  if (s < 1.e-5f) {
    if (c < 0.f) bbf = -bbf;
    bf = bbf;
  } else {
    bf = bbf / s;
  }
  float he = br * s + bt * c;
  float hxgeo = he * cf - bf * sf, hygeo = he * sf + bf * cf, hzgeo = br * c - bt * s;
  /* GEOGSW_08 J = +1 */
  *hx = A[0] * hxgeo + A[1] * hygeo + A[2] * hzgeo;
  *hy = A[3] * hxgeo + A[4] * hygeo + A[5] * hzgeo;
  *hz = A[6] * hxgeo + A[7] * hygeo + A[8] * hzgeo;
}

/* use_igrf = 1 for this model (itime as the adapters decode it, interp_dens_model_adapter.f95:217-221) */
int so_model_set_igrf(so_model *m, int yearday, int msec, const char *coeff_file) {
  igrf_table *t = (igrf_table *)malloc(sizeof *t);
  if (!t || read_table(coeff_file, t)) {
    free(t);
    return -1;
  }
  int year = yearday / 1000, day = yearday % 1000;
  int hour = msec / (1000 * 60 * 60);
  int min = (msec - hour * (1000 * 60 * 60)) / (1000 * 60);
  int sec = (msec - hour * (1000 * 60 * 60) - min * (1000 * 60)) / 1000;
  so_igrf_recalc(t, year, day, hour, min, sec, m->igrf_G, m->igrf_H, m->igrf_REC, m->igrf_A);
  m->use_igrf = 1;
  free(t);
  return 0;
}
