// srt_fastmath.hpp -- device-only elementary functions shared by the kernels: the hot path's division and the
// range-specialised sqrt / sincos / log / exp / pow of the scattered model and the T04_s field.
#pragma once
#include <hip/hip_runtime.h>

namespace srt {

constexpr double FM_PI = 3.14159265358979323846;

// a/b for operands well inside the exponent range (every division of the hot path: frequencies, densities,
// field magnitudes, grid spacings).  This is the compiler's own fp64 division sequence -- v_rcp_f64, two Newton
// steps on the reciprocal, quotient, one residual correction -- without the v_div_scale / v_div_fmas /
// v_div_fixup wrapper that only matters for operands or quotients near the ends of the exponent range, so the
// result is bit-identical to a/b wherever that wrapper would not have scaled (8 instructions instead of 11).
__device__ __forceinline__ double fdiv(double a, double b) {
  double r = __builtin_amdgcn_rcp(b);
  r = fma(fma(-b, r, 1.0), r, r);
  r = fma(fma(-b, r, 1.0), r, r);
  double q = a * r;
  return fma(fma(-b, q, a), r, q);
}

// Elementary functions on the argument ranges of the scattered model's per-sample passes (srt_scattered.hpp) and of the
// T04_s field (srt_t04.hpp): each is the textbook (fdlibm) kernel without the library's range handling -- arguments there
// are never denormal, huge or NaN-by-construction -- and agrees with the library to <= 1-2 ulp, at a third to a half of
// its instructions.
namespace fm {
// sqrt for 0 <= x, neither denormal nor near overflow: the compiler's own sequence (v_rsq_f64, one Goldschmidt step, two
// residual corrections) without its range scaling
__device__ __forceinline__ double sqrt_pos(double x) {
  const double y = __builtin_amdgcn_rsq(x);
  double g = x * y, h = 0.5 * y;
  const double r = fma(-h, g, 0.5);
  g = fma(g, r, g);
  h = fma(h, r, h);
  double d = fma(-g, g, x);
  g = fma(d, h, g);
  d = fma(-g, g, x);
  g = fma(d, h, g);
  return x == 0.0 ? 0.0 : g;
}
// g = sqrt(x) and inv = 1 / g for a positive normal x from ONE v_rsq_f64: the Goldschmidt pair (g, h = 1 / (2 g)) of sqrt_pos,
// then one Newton step on 2 h against the finished g (<= 1 ulp each; 13 operations against sqrt_pos + fdiv's 20)
__device__ __forceinline__ void sqrt_and_inv_pos(double x, double &g_out, double &inv_out) {
  const double y = __builtin_amdgcn_rsq(x);
  double g = x * y, h = 0.5 * y;
  const double r = fma(-h, g, 0.5);
  g = fma(g, r, g);
  h = fma(h, r, h);
  double d = fma(-g, g, x);
  g = fma(d, h, g);
  d = fma(-g, g, x);
  g = fma(d, h, g);
  double inv = h + h;
  inv = fma(fma(-g, inv, 1.0), inv, inv);
  g_out = g;
  inv_out = inv;
}
// sin and cos of a in [0, pi (1 + 2e-3)]: quadrant k = 0, 1, 2, t = a - k pi/2 in about [-pi/4, pi/4]
__device__ __forceinline__ void sincos_0pi(double a, double &s, double &c) {
  const double PIO2_HI = 1.57079632673412561417e+00, PIO2_LO = 6.07710050650619224932e-11; // k * hi is exact (33 bits)
  const double kf = a > 0.75 * FM_PI ? 2.0 : (a > 0.25 * FM_PI ? 1.0 : 0.0);
  const double t = fma(-kf, PIO2_LO, fma(-kf, PIO2_HI, a));
  const double z = t * t;
  const double rs = 8.33333333332248946124e-03 + z * (-1.98412698298579493134e-04 + z * (2.75573137070700676789e-06 + z * (-2.50507602534068634195e-08 + z * 1.58969099521155010221e-10)));
  const double st = t + (z * t) * (-1.66666666666666324348e-01 + z * rs);
  const double rc = z * (4.16666666666666019037e-02 + z * (-1.38888888888741095749e-03 + z * (2.48015872894767294178e-05 + z * (-2.75573143513906633035e-07 + z * (2.08757232129817482790e-09 + z * -1.13596475577881948265e-11)))));
  const double hz = 0.5 * z, w = 1.0 - hz;
  const double ct = w + (((1.0 - w) - hz) + z * rc);
  s = kf == 1.0 ? ct : (kf == 2.0 ? -st : st);
  c = kf == 1.0 ? -st : (kf == 2.0 ? -ct : ct);
}
// ln x for a positive normal x
__device__ __forceinline__ double log_pos(double x) {
  double m = __builtin_amdgcn_frexp_mant(x); // [0.5, 1)
  int e = __builtin_amdgcn_frexp_exp(x);
  const bool low = m < 0.70710678118654752440;
  m = low ? m + m : m; // [sqrt(1/2), sqrt 2)
  e = low ? e - 1 : e;
  const double f = m - 1.0, k = (double)e;
  const double sq = fdiv(f, 2.0 + f), z = sq * sq, w = z * z;
  const double t1 = w * (3.999999999940941908e-01 + w * (2.222219843214978396e-01 + w * 1.531383769920937332e-01));
  const double t2 = z * (6.666666666666735130e-01 + w * (2.857142874366239149e-01 + w * (1.818357216161805012e-01 + w * 1.479819860511658591e-01)));
  const double R = t2 + t1, hfsq = 0.5 * f * f;
  return k * 6.93147180369123816490e-01 - ((hfsq - (sq * (hfsq + R) + k * 1.90821492927058770002e-10)) - f);
}
// e^y for y <= ~700 (underflows to 0 below -745)
__device__ __forceinline__ double exp_any(double y) {
  y = fmax(y, -800.0);
  const double k = rint(y * 1.44269504088896338700e+00);
  const double r = fma(-k, 1.90821492927058770002e-10, fma(-k, 6.93147180369123816490e-01, y)); // |r| <= 0.3466
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
  return ldexp(p, (int)k);
}
// sin and cos of |x| up to ~1e5 (T04_s: positions in Earth radii over scale lengths, tilt angles): quadrant k = rint(x 2/pi),
// t = x - k pi/2 by a two-part pi/2 (k * hi exact, error ~k * 1e-21)
__device__ __forceinline__ void sincos_mod(double x, double &s, double &c) {
  const double k = rint(x * 0.63661977236758134308);
  double t = fma(-k, 1.57079632673412561417e+00, x);
  t = fma(-k, 6.07710050650619224932e-11, t);
  const double z = t * t;
  const double rs = 8.33333333332248946124e-03 + z * (-1.98412698298579493134e-04 + z * (2.75573137070700676789e-06 + z * (-2.50507602534068634195e-08 + z * 1.58969099521155010221e-10)));
  const double st = t + (z * t) * (-1.66666666666666324348e-01 + z * rs);
  const double rc = z * (4.16666666666666019037e-02 + z * (-1.38888888888741095749e-03 + z * (2.48015872894767294178e-05 + z * (-2.75573143513906633035e-07 + z * (2.08757232129817482790e-09 + z * -1.13596475577881948265e-11)))));
  const double hz = 0.5 * z, w = 1.0 - hz;
  const double ct = w + (((1.0 - w) - hz) + z * rc);
  const int q = (int)k;
  const double s1 = (q & 1) ? ct : st, c1 = (q & 1) ? -st : ct;
  s = (q & 2) ? -s1 : s1;
  c = (q & 2) ? -c1 : c1;
}
__device__ __forceinline__ double sin_mod(double x) {
  double s, c;
  sincos_mod(x, s, c);
  return s;
}
__device__ __forceinline__ double cos_mod(double x) {
  double s, c;
  sincos_mod(x, s, c);
  return c;
}
// x**y for x > 0 (relative error ~|y ln x| 2^-52); x <= 0 or not finite: the library's
__device__ __forceinline__ double pow_pos(double x, double y) {
  if (!(x > 0.0) || !(x < 1.0e300)) return pow(x, y);
  return exp_any(y * log_pos(x));
}
} // namespace fm

} // namespace srt
