// srt_fastmath.hpp -- the three elementary functions the scattered model's weights are made of, on the argument ranges
// that model guarantees, at about a quarter of the device library's instruction count (no special cases, no
// large-argument reduction), accurate to ~1 ulp.  450 of the 650 instructions per (stencil point, neighbour) of
// lsinterp's weight (lsinterp_mod.f95:175-221: coswindow, etainv) were cos / log / exp / exp of the general library.
#pragma once
#include <hip/hip_runtime.h>

namespace srt {

// exp(t) for t <= 709 (underflows to 0 through ldexp; NaN stays NaN).
__device__ __forceinline__ double exp_fast(double t_in) {
  const double L2E = 1.4426950408889634074, LN2HI = 6.93147180369123816490e-01, LN2LO = 1.90821492927058770002e-10;
  const double t = fmax(t_in, -800.0);
  const double n = rint(t * L2E);
  double r = fma(-n, LN2HI, t);
  r = fma(-n, LN2LO, r); // |r| <= ln2/2
  // Taylor to r^13: remainder (ln2/2)^14/14! = 4e-18
  double p = 1.0 / 6227020800.0;
  p = fma(p, r, 1.0 / 479001600.0);
  p = fma(p, r, 1.0 / 39916800.0);
  p = fma(p, r, 1.0 / 3628800.0);
  p = fma(p, r, 1.0 / 362880.0);
  p = fma(p, r, 1.0 / 40320.0);
  p = fma(p, r, 1.0 / 5040.0);
  p = fma(p, r, 1.0 / 720.0);
  p = fma(p, r, 1.0 / 120.0);
  p = fma(p, r, 1.0 / 24.0);
  p = fma(p, r, 1.0 / 6.0);
  p = fma(p, r, 0.5);
  p = fma(p, r, 1.0);
  p = fma(p, r, 1.0);
  return ldexp(p, (int)n);
}

// ln(x) for finite x > 0 (normal or subnormal)
__device__ __forceinline__ double log_pos(double x) {
  const double LN2HI = 6.93147180369123816490e-01, LN2LO = 1.90821492927058770002e-10;
  int e = __builtin_amdgcn_frexp_exp(x);
  double m = __builtin_amdgcn_frexp_mant(x); // [0.5, 1)
  const bool lowhalf = m < 0.70710678118654752440;
  m = lowhalf ? m + m : m;
  e = lowhalf ? e - 1 : e; // m in [sqrt(1/2), sqrt(2))
  const double num = m - 1.0, den = m + 1.0;
  // s = num/den by one reciprocal estimate + two Newton steps + residual correction (den in [1.7, 2.42))
  double rc = __builtin_amdgcn_rcp(den);
  rc = fma(fma(-den, rc, 1.0), rc, rc);
  rc = fma(fma(-den, rc, 1.0), rc, rc);
  double s = num * rc;
  s = fma(fma(-den, s, num), rc, s);
  const double z = s * s; // <= 0.02944
  // 2 atanh(s) = 2 s (1 + z/3 + z^2/5 + ... + z^11/23): remainder z^12/25 = 2e-20
  double p = 1.0 / 23.0;
  p = fma(p, z, 1.0 / 21.0);
  p = fma(p, z, 1.0 / 19.0);
  p = fma(p, z, 1.0 / 17.0);
  p = fma(p, z, 1.0 / 15.0);
  p = fma(p, z, 1.0 / 13.0);
  p = fma(p, z, 1.0 / 11.0);
  p = fma(p, z, 1.0 / 9.0);
  p = fma(p, z, 1.0 / 7.0);
  p = fma(p, z, 1.0 / 5.0);
  p = fma(p, z, 1.0 / 3.0);
  // ln m = 2s + 2s*z*p, the small part first
  const double s2 = s + s;
  const double fe = (double)e;
  const double small = fma(s2 * z, p, fe * LN2LO);
  return fma(fe, LN2HI, s2 + small);
}

// cos(x) for 0 <= x <= pi (slightly beyond is fine): sin(pi/2 - x) with one odd polynomial on [-pi/2, pi/2]
__device__ __forceinline__ double cos_0pi(double x) {
  const double PIO2HI = 1.57079632679489655800e+00, PIO2LO = 6.12323399573676603587e-17;
  const double u = (PIO2HI - x) + PIO2LO;
  const double z = u * u;
  // sin u = u (1 - z/3! + z^2/5! - ... + z^12/25!): remainder (pi/2)^27/27! = 2e-23
  double p = 1.0 / 15511210043330985984000000.0;
  p = fma(p, z, -1.0 / 25852016738884976640000.0);
  p = fma(p, z, 1.0 / 51090942171709440000.0);
  p = fma(p, z, -1.0 / 121645100408832000.0);
  p = fma(p, z, 1.0 / 355687428096000.0);
  p = fma(p, z, -1.0 / 1307674368000.0);
  p = fma(p, z, 1.0 / 6227020800.0);
  p = fma(p, z, -1.0 / 39916800.0);
  p = fma(p, z, 1.0 / 362880.0);
  p = fma(p, z, -1.0 / 5040.0);
  p = fma(p, z, 1.0 / 120.0);
  p = fma(p, z, -1.0 / 6.0);
  return fma(u * z, p, u);
}

} // namespace srt
