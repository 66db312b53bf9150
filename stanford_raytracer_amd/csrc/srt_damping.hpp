// srt_damping.hpp -- the step after the path: hot-plasma (Landau / cyclotron) damping along the kept
// rows of every ray, on the device, from the row buffer the trace kernel left in HBM.
//
// Reference (MATLAB post-processor, /root/reference/matlab/damping/):
//   test_dampray.m:24-99          per-row driver: k = n w/c, kpar/kperp about B0, ki along vg, running magnitude
//   spatialdamping.m:26-44        ki  = sum_f -(w/c) 1/2 1/(4 n (2 A n^2 - B)) Di
//   temporaldamping.m:27-42       gamma = sum_f -Di / (dD0/dw), central difference, DEL = 1e-8
//   hot_dispersion_imag.m:29-48   Di = quadva( ((1+eps)/(t^2+eps)) c I(c (1-t+eps)/(t+eps)), [0,1], TOL, eps )
//   hot_dispersion_real.m:14-26   D0 = 4 (A n^4 - B n^2 + R L P)
//   integrand.m:20-72             I(vperp): sum over resonances m of the Kennel/Chen integrand at vpar = (w - m wch)/kpar
//   fG1.m, fG2.m                  G1, G2 from central differences of the distribution (DEL = 1e-8)
//   suprathermal.m, maxwellboltzmann.m   the two distributions the scripts use
//   quadva.m:49-122, :171-182     Shampine's vectorised adaptive Gauss-Kronrod (7,15): 10 initial panels on [-1,1], the
//                                 f1 change of variable, per-panel acceptance against the running total, all rejected
//                                 panels bisected at once, at most 650 panels
//   ../stix_parameters.m          cold Stix parameters (nus = 0 as the scripts set it)
//
// Mapping: one wave per kept row.  quadva is *already* a data-parallel algorithm -- every iteration evaluates 15 nodes
// on each live panel, then decides about all panels together -- so the wave walks the panel list four panels (60
// lanes) at a time with the list in LDS, and takes the same per-iteration decisions in the same order as the
// reference's vectorised code (sequential sums, so the CPU oracle's restatement agrees to rounding of the integrand).
#pragma once
#include "srt_device.hpp"

namespace srt {

constexpr int DMP_MAXSUB = 650;  // quadva.m:111
constexpr int DMP_MAXRES = 8;

struct DampParams {
  int dist;   // 0 suprathermal (Bell 2002), 1 Maxwell-Boltzmann
  int mode;   // 0 spatial (ki along vg), 1 temporal (gamma)
  int nres;
  int m[DMP_MAXRES];
  double Ne_h, kT; // Maxwellian: hot density (m^-3) and temperature (J)
  double tol;      // TOL of the scripts (relative), 1e-3
  double qh, mh;   // hot species: -Q, ME of const.m
};

struct DampArgs {
  const double *rows;   // [nrays][slots][ROW]
  const int *nrows;     // [nrays] total rows of the ray
  const double *w0;     // [nrays]
  long long nrays;
  int slots, outputper, nspec;
  double q[MAXSPEC], ms[MAXSPEC];
  DampParams p;
  double *rate;         // [nrays][slots]
  int *flag;            // [nrays][slots]: 0 ok, 1 error test not met (quadva's OK = false), 2 integrand not finite, 3 k = 0
};

// per-row constants every lane holds
struct DampRow {
  double w, kperp, kpar, wch, R, L, P, S, n2, st, ct, pref;
};

__device__ __forceinline__ double damp_f(const DampParams &p, double vperp, double vpar) {
  if (p.dist == 0) { // suprathermal.m
    const double v2c = vperp * vperp + vpar * vpar + 1.0; // v0 = 1
    const double v = 100.0 * sqrt(v2c);
    const double v2 = v * v, v4 = v2 * v2;
    const double f = 4.9e5 / v4 - 8.3e14 / (v4 * v) + 5.4e23 / (v4 * v2);
    return f * 1.0e12;
  }
  // Ne_h * maxwellboltzmann(vperp, vpar, m, kT)
  const double c = p.mh / (2.0 * PI * p.kT);
  return p.Ne_h * (c * sqrt(c)) * exp(-p.mh * (vperp * vperp + vpar * vpar) / 2.0 / p.kT);
}

// besselj(n, x) for integer n of either sign and real x of either sign
__device__ __forceinline__ double damp_besselj(int n, double x) {
  double sg = 1.0;
  if (n < 0) {
    n = -n;
    if (n & 1) sg = -sg;
  }
  if (x < 0.0) {
    x = -x;
    if (n & 1) sg = -sg;
  }
  const double v = n == 0 ? j0(x) : n == 1 ? j1(x) : jn(n, x);
  return sg * v;
}

// integrand.m for one vperp
__device__ __noinline__ double damp_integrand(const DampParams &p, const DampRow &r, double vperp) {
  const double EPSM = 2.220446049250313e-16; // matlab eps
  const double DEL = 1e-8;
  const double n2 = r.n2, st = r.st, ct = r.ct;
  const double x = r.kperp * vperp / r.wch;
  double sum = 0.0;
  for (int mi = 0; mi < p.nres; ++mi) {
    const int m = p.m[mi];
    const double Jm = damp_besselj(m, x), Jm1 = damp_besselj(m - 1, x), Jp1 = damp_besselj(m + 1, x);
    const double vpar = (r.w - (double)m * r.wch) / r.kpar;
    // fG1.m / fG2.m: the same four samples of f serve both
    double d = DEL * fabs(vperp);
    if (d < 10.0 * EPSM) d = 10.0 * EPSM;
    const double dfperp = (damp_f(p, vperp + d, vpar) - damp_f(p, vperp - d, vpar)) / (2.0 * d);
    d = DEL * fabs(vpar);
    if (d < 10.0 * EPSM) d = 10.0 * EPSM;
    const double dfpar = (damp_f(p, vperp, vpar + d) - damp_f(p, vperp, vpar - d)) / (2.0 * d);
    const double cross = vpar * dfperp - vperp * dfpar;
    const double G1 = dfperp - (r.kpar / r.w) * cross;
    const double G2 = Jm * (dfpar - ((double)m * r.wch + EPSM) / (r.w * vperp + EPSM) * cross);
    const double Rn = r.R - n2, Ln = r.L - n2, dJ = Jp1 - Jm1;
    sum = sum + (G1 * ((r.P - n2 * st * st) * (2.0 * Ln * vperp * Jp1 * Jp1 + 2.0 * vperp * Rn * Jm1 * Jm1 + n2 * st * st * vperp * dJ * dJ) -
                       n2 * ct * st * (2.0 * vpar * Jm * (Jp1 * Rn + Jm1 * Ln) + n2 * ct * st * vperp * dJ * dJ)) +
                 G2 * (4.0 * vpar * Jm * (Ln * Rn + n2 * st * st * (r.S - n2)) - 2.0 * n2 * ct * st * (Rn * vperp * Jm1 + Ln * vperp * Jp1)));
  }
  return r.pref * sum * vperp; // pref = -2 pi^2 ((qh^2/mh/EPS0)/(w |kpar|))
}

// hot_dispersion_imag.m:37-44 composed with quadva's f1 (quadva.m:126-136, a = 0, b = 1): value at GK abscissa s in [-1,1]
__device__ __forceinline__ double damp_node(const DampParams &p, const DampRow &r, double cl, double s, double &Tt) {
  const double EPSM = 2.220446049250313e-16;
  Tt = 0.25 * s * (3.0 - s * s) + 0.5;
  const double vn = (1.0 - Tt + EPSM) / (Tt + EPSM);
  double y = ((1.0 + EPSM) / (Tt * Tt + EPSM)) * (cl * damp_integrand(p, r, vn * cl));
  return 0.75 * y * (1.0 - s * s);
}

// stix_parameters.m with nus = 0
__device__ inline void damp_stix(const DampArgs &a, const double *Ns, double w, double Bmag, double &S, double &D, double &P, double &R,
                                 double &L) {
  double sr = 0.0, sl = 0.0, sp = 0.0;
  for (int s = 0; s < a.nspec; ++s) {
    const double wps2 = Ns[s] * (a.q[s] * a.q[s]) / a.ms[s] / EPS0;
    const double wcs = (a.q[s] * Bmag) / a.ms[s];
    sr += wps2 / (w * (w + wcs));
    sl += wps2 / (w * (w - wcs));
    sp += wps2 / (w * w);
  }
  R = 1.0 - sr;
  L = 1.0 - sl;
  P = 1.0 - sp;
  S = 0.5 * (R + L);
  D = 0.5 * (R - L);
}

// hot_dispersion_real.m
__device__ inline double damp_d0(const DampArgs &a, const double *Ns, double kperp, double kpar, double w, double Bmag, double cl) {
  double S, D, P, R, L;
  damp_stix(a, Ns, w, Bmag, S, D, P, R, L);
  const double th = atan2(kperp, kpar);
  const double n = cl / w * sqrt(kperp * kperp + kpar * kpar);
  const double s2 = sin(th) * sin(th), c2 = cos(th) * cos(th);
  const double A = S * s2 + P * c2, B = R * L * s2 + P * S * (1.0 + c2), C = R * L * P;
  const double nn = n * n;
  return 4.0 * (A * nn * nn - B * nn + C);
}

__global__ __launch_bounds__(64) void damping_rate_kernel(DampArgs a) {
  __shared__ double lo[2][DMP_MAXSUB + 2], hi[2][DMP_MAXSUB + 2], qsub[DMP_MAXSUB + 2], esub[DMP_MAXSUB + 2], fxs[64];
  const int lane = threadIdx.x;
  const double EPSM = 2.220446049250313e-16;
  const double MU0 = PI * 4e-7;
  const double cl = sqrt(1.0 / EPS0 / MU0); // physconst.m (spatialdamping.m, integrand.m)
  const double CL_CONST = 299792458.0;      // const.m (test_dampray.m: k = n w / clight)
  // GK(7,15), quadva.m:49-61
  const double pn[7] = {0.2077849550078985, 0.4058451513773972, 0.5860872354676911, 0.7415311855993944,
                        0.8648644233597691, 0.9491079123427585, 0.9914553711208126};
  const double pw[7] = {0.2044329400752989, 0.1903505780647854, 0.1690047266392679, 0.1406532597155259,
                        0.1047900103222502, 0.06309209262997855, 0.02293532201052922};
  const double pw7[7] = {0.0, 0.3818300505051189, 0.0, 0.2797053914892767, 0.0, 0.1294849661688697, 0.0};
  auto node_of = [&](int i) { return i < 7 ? -pn[6 - i] : i == 7 ? 0.0 : pn[i - 8]; };
  auto wt_of = [&](int i) { return i < 7 ? pw[6 - i] : i == 7 ? 0.2094821410847278 : pw[i - 8]; };
  auto ewt_of = [&](int i) { return wt_of(i) - (i < 7 ? pw7[6 - i] : i == 7 ? 0.4179591836734694 : pw7[i - 8]); };

  const long long total = a.nrays * a.slots;
  for (long long idx = blockIdx.x; idx < total; idx += gridDim.x) {
    const long long ray = idx / a.slots;
    const int rr = (int)(idx % a.slots);
    const int T = a.nrows[ray];
    const int kept = T > 0 ? (T - 1) / a.outputper + 1 : 0;
    if (rr == 0 || rr >= kept) { // row 0: magnitude 1 by definition; beyond the ray: nothing
      if (lane == 0) {
        a.rate[idx] = 0.0;
        a.flag[idx] = 0;
      }
      continue;
    }
    const double *row = a.rows + (size_t)idx * ROW;
    const double w = a.w0[ray];
    const double vg[3] = {row[7], row[8], row[9]}, nv[3] = {row[10], row[11], row[12]}, B0[3] = {row[13], row[14], row[15]};
    double Ns[MAXSPEC];
    for (int s = 0; s < a.nspec; ++s) Ns[s] = row[16 + s];
    const double Bmag = sqrt(B0[0] * B0[0] + B0[1] * B0[1] + B0[2] * B0[2]);
    // test_dampray.m:64-75
    double k[3], kdotB = 0.0, kk = 0.0;
    for (int c = 0; c < 3; ++c) k[c] = nv[c] * w / CL_CONST;
    for (int c = 0; c < 3; ++c) kk += k[c] * k[c];
    const double kmag = sqrt(kk);
    double Bhat[3];
    for (int c = 0; c < 3; ++c) Bhat[c] = B0[c] / Bmag;
    for (int c = 0; c < 3; ++c) kdotB += k[c] * Bhat[c];
    const double kpar = kdotB;
    double kp2 = 0.0;
    for (int c = 0; c < 3; ++c) {
      const double v = k[c] - kpar * Bhat[c];
      kp2 += v * v;
    }
    const double kperp = sqrt(kp2);
    if (!(kmag != 0.0)) { // 'Re{n} = 0, not solving evanescent mode': magnitude stays 0 from here on
      if (lane == 0) {
        a.rate[idx] = 0.0;
        a.flag[idx] = 3;
      }
      continue;
    }
    DampRow r;
    r.w = w;
    r.kperp = kperp;
    r.kpar = kpar;
    r.wch = (a.p.qh * Bmag) / a.p.mh; // hot gyrofrequency, signed
    double D_;
    damp_stix(a, Ns, w, Bmag, r.S, D_, r.P, r.R, r.L);
    const double th = atan2(kperp, kpar);
    r.st = sin(th);
    r.ct = cos(th);
    const double nref = sqrt((cl * cl / (w * w)) * (kperp * kperp + kpar * kpar));
    r.n2 = nref * nref;
    r.pref = -2.0 * PI * PI * ((a.p.qh * a.p.qh / a.p.mh / EPS0) / (w * fabs(kpar)));

    // ---- quadva.m Vadapt, tinterval = linspace(-1,1,11) ----
    const double rtol = a.p.tol > 100.0 * EPSM ? a.p.tol : 100.0 * EPSM, atol = EPSM;
    const double tbma = 2.0;
    int cur = 0, nsub = 10;
    if (lane < 10) {
      // linspace(-1,1,11): a + i*(b-a)/10, end point exact
      lo[0][lane] = -1.0 + (double)lane * (2.0 / 10.0);
      hi[0][lane] = lane == 9 ? 1.0 : -1.0 + (double)(lane + 1) * (2.0 / 10.0);
    }
    __syncthreads();
    double IfxOK = 0.0, errOK = 0.0, Ifx = NAN, errbnd = NAN;
    int status = 1; // left by `break`: OK = false
    bool first = true;
    while (true) {
      // all nodes of all live panels, four panels per trip
      bool bad = false;
      double prev_last = -INFINITY; // check_spacing runs over the whole row vector of transformed abscissae
      for (int base = 0; base < nsub; base += 4) {
        const int sp = lane / 15, nd = lane % 15;
        const int sub = base + sp;
        const bool act = lane < 60 && sub < nsub;
        double fx = 0.0, Tt = 0.0;
        if (act) {
          const double l = lo[cur][sub], h = hi[cur][sub];
          const double mid = (l + h) / 2.0, hh = (h - l) / 2.0;
          const double s = node_of(nd) * hh + mid;
          fx = damp_node(a.p, r, cl, s, Tt);
        }
        // check_spacing(Tt): diff(x) <= 100 eps max(|x_i|,|x_{i+1}|)
        const double Tn = __shfl_down(Tt, 1, 64);
        const int nact = (nsub - base >= 4 ? 4 : nsub - base) * 15;
        bool close = false;
        if (act && lane + 1 < nact) close = (Tn - Tt) <= 100.0 * EPSM * fmax(fabs(Tt), fabs(Tn));
        if (lane == 0 && base > 0) close = close || (Tt - prev_last) <= 100.0 * EPSM * fmax(fabs(Tt), fabs(prev_last));
        prev_last = __shfl(Tt, nact - 1, 64);
        if (__ballot(close || (act && !isfinite(fx)))) bad = true;
        fxs[lane] = fx;
        __syncthreads();
        if (lane < 4 && base + lane < nsub) {
          const double l = lo[cur][base + lane], h = hi[cur][base + lane], hh = (h - l) / 2.0;
          double q = 0.0, e = 0.0;
          for (int i = 0; i < 15; ++i) {
            q += wt_of(i) * fxs[lane * 15 + i];
            e += ewt_of(i) * fxs[lane * 15 + i];
          }
          qsub[base + lane] = q * hh;
          esub[base + lane] = e * hh;
        }
        __syncthreads();
      }
      if (bad) { // too_close || any(~isfinite(fx)): break with the previous iteration's Ifx
        if (first) status = 2; // 'Difficulty evaluating integrand.'
        break;
      }
      double sq = 0.0, se = 0.0;
      for (int i = 0; i < nsub; ++i) {
        sq += qsub[i];
        se += esub[i];
      }
      Ifx = sq + IfxOK;
      errbnd = fabs(se + errOK);
      const double tol = fmax(atol, rtol * fabs(Ifx));
      if (errbnd <= tol) {
        status = 0;
        break;
      }
      // accept panels whose error is small for their length; bisect the others (all lanes walk the list together)
      int nkeep = 0;
      double accE = 0.0, accQ = 0.0; // sum(errsubs(ndx)), sum(Ifxsubs(ndx)) (quadva.m:104-106)
      for (int i = 0; i < nsub; ++i) {
        const double l = lo[cur][i], h = hi[cur][i], hh = (h - l) / 2.0;
        if (fabs(esub[i]) <= (2.0 / tbma) * hh * tol) {
          accE += esub[i];
          accQ += qsub[i];
        } else {
          if (lane == 0 && 2 * nkeep + 1 < DMP_MAXSUB + 2) {
            const double mid = (l + h) / 2.0;
            lo[cur ^ 1][2 * nkeep] = l;
            hi[cur ^ 1][2 * nkeep] = mid;
            lo[cur ^ 1][2 * nkeep + 1] = mid;
            hi[cur ^ 1][2 * nkeep + 1] = h;
          }
          ++nkeep;
        }
      }
      errOK = errOK + accE;
      IfxOK = IfxOK + accQ;
      if (nkeep == 0) {
        status = 0;
        break;
      }
      if (2 * nkeep > DMP_MAXSUB) break; // OK = false
      __syncthreads();
      cur ^= 1;
      nsub = 2 * nkeep;
      first = false;
    }
    __syncthreads();
    double out;
    if (status == 2) out = NAN;
    else {
      const double Di = Ifx;
      if (a.p.mode == 0) {
        const double A = r.S * r.st * r.st + r.P * r.ct * r.ct;
        const double B = r.R * r.L * r.st * r.st + r.P * r.S * (1.0 + r.ct * r.ct);
        const double ki = 0.0 + -(w / cl) * (1.0 / 2.0) * (1.0 / (4.0 * nref * (2.0 * A * nref * nref - B))) * Di;
        double kv = 0.0, vv = 0.0;
        for (int c = 0; c < 3; ++c) {
          kv += k[c] * vg[c];
          vv += vg[c] * vg[c];
        }
        out = ki * kv / (kmag * sqrt(vv));
      } else {
        double d = 1e-8 * fabs(w);
        if (d < 10.0 * EPSM) d = 10.0 * EPSM;
        const double dD = (damp_d0(a, Ns, kperp, kpar, w + d, Bmag, cl) - damp_d0(a, Ns, kperp, kpar, w - d, Bmag, cl)) / (2.0 * d);
        out = 0.0 + -Di / dD;
      }
    }
    if (lane == 0) {
      a.rate[idx] = out;
      a.flag[idx] = status;
    }
  }
}

// test_dampray.m:87-93 / test_compare_time_and_spatial_damping.m:77-80: running magnitude along each ray
__global__ void damping_magnitude_kernel(DampArgs a, double *mag) {
  const long long ray = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (ray >= a.nrays) return;
  const int T = a.nrows[ray];
  const int kept = T > 0 ? (T - 1) / a.outputper + 1 : 0;
  double m = 1.0;
  for (int r = 0; r < a.slots; ++r) {
    const size_t idx = (size_t)ray * a.slots + r;
    if (r >= kept) {
      mag[idx] = 0.0;
      continue;
    }
    if (r > 0) {
      if (a.flag[idx] == 3) m = 0.0; // magnitude(ii) is never assigned: stays at its zeros() value
      else {
        const double *p1 = a.rows + idx * ROW, *p0 = p1 - ROW;
        if (a.p.mode == 0) {
          const double dx = p1[1] - p0[1], dy = p1[2] - p0[2], dz = p1[3] - p0[3];
          m = m * exp(-sqrt(dx * dx + dy * dy + dz * dz) * a.rate[idx]);
        } else {
          m = m * exp(a.rate[idx] * (p1[0] - p0[0]));
        }
      }
    }
    mag[idx] = m;
  }
}

} // namespace srt
