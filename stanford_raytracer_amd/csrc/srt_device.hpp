// srt_device.hpp -- device-side physics of the many-ray Haselgrove integrator (gfx950 / CDNA4).
//
// One ray per lane.  Everything here is fp64 VALU work; there is no dense contraction on this path,
// so no MFMA.  The functions restate WHAT the reference computes (file:line cited per function) but
// are organised for the GPU: common sub-expressions that the Fortran re-evaluates per call
// (Stix parameters shared by the six F evaluations of dF/dk, the plasma state shared by dF/dk and
// dF/dw, trigonometry in the dipole field) are evaluated once, and divisions are merged.  Results
// differ from the reference at rounding level only; the parity ladder in DESIGN.md bounds that.
#pragma once
#include <type_traits>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "srt_fastmath.hpp"
#include "srt_t04.hpp"

namespace srt {

constexpr double EPS0 = 8.854187817e-12;          // constants.f95:4
constexpr double PI = 3.141592653589793238462643; // constants.f95:5
constexpr double R_E = 6371.2e3;                  // constants.f95:8
constexpr int MAXSPEC = 4;
constexpr int ROW = 20;

// Per-model species constants (qs, ms of funcPlasmaParams; constant in all three adapters).
struct Species {
  double q[MAXSPEC], m[MAXSPEC];
  double c[MAXSPEC]; // q^2/(m eps0):   wps2 = Ns * c      (raytracer.f95:92)
  double g[MAXSPEC]; // q/m:            wcs  = g * |B0|    (raytracer.f95:93)
  double maxq2;      // maxval(abs(qs))**2                  (raytracer.f95:65)
  double minm_eps0;  // minval(ms)*EPS0
  int nspec;
};

// Dipole field + SM<->GSM round trip constants (bmodel_dipole.f95, xform_double/T4.f95).
struct FieldConst {
  double bo_re3; // Bo * R_E^3, Bo = .312/10000 T
  double cm, sm; // cos(mu), sin(mu), mu = dipole tilt for the run's itime
  // use_igrf = 1 (interp_dens_model_adapter.f95:236-241): Schmidt-normalised Gauss coefficients for the run's date,
  // recursion constants and the GEO->GSM matrix, as RECALC_08 leaves them (host: srt_host::igrf_setup)
  int use_igrf, yearday, msec, use_tsy;
  // g, h, rec per (m, n) term in the ORDER THE SYNTHESIS VISITS THEM (m = 1..14 outer, n = m..14 inner): entry
  // (m, n) sits at igrf_off(m) + n - m (rows 1..5 in 0..59, rows 6..14 from 64).  128 entries: lane l of a wave keeps entries l
  // and l + 64 in registers.
  float Gv[128], Hv[128], Rv[128];
  float A[9];
  // use_tsyganenko = 1 (interp_dens_model_adapter.f95:223-258): T04_s(iopt, real(parmod), real(psi), real(x_gsm/R_E));
  // parmod = Pdyn, Dst, ByIMF, BzIMF, W1..W6 (driver flags --tsyganenko_*), psi = the adapters' COMMON /GEOPACK1/ PSI,
  // which aliases geopack's ST0 (see srt_host::igrf_setup)
  float parmod[10], psi;
};
// (rows m >= 6 start at 64: no row straddles the two register halves, so the half is chosen once per m, not per term)
__host__ __device__ inline int igrf_off(int m) { return (m - 1) * 15 - (m - 1) * m / 2 + (m >= 6 ? 4 : 0); }

struct Common {
  Species sp;
  FieldConst fld;
  double C; // speed of light as constants.f95:7 computes it
};
// The same constants with the field option fixed at compile time: the trace kernel is instantiated once for the plain
// dipole and once for everything else (IGRF main field and / or T04_s external field, chosen at run time inside), so
// that the dipole kernel carries none of their registers or code (sharing one kernel cost it 11 %).
// Plain `Common` means "look at fld.use_igrf at run time" (the layered kernels).
struct CommonDipole : Common {};
struct CommonIgrf : Common {};     // the adapters' general field tail: IGRF or dipole base, T04_s on top if use_tsy (run time)
struct CommonIgrfOnly : Common {}; // use_igrf = 1, use_tsyganenko = 0: no T04 call sites (seven or eight per stencil) in the kernel
template <class CM>
__device__ __forceinline__ bool field_is_igrf(const CM &cm) {
  if constexpr (std::is_same<CM, CommonDipole>::value) return false;
  else if constexpr (std::is_same<CM, CommonIgrf>::value || std::is_same<CM, CommonIgrfOnly>::value) return true;
  else return cm.fld.use_igrf != 0 || cm.fld.use_tsy != 0;
}
template <class CM>
constexpr bool field_igrf_only() { return std::is_same<CM, CommonIgrfOnly>::value; }

// ---------------------------------------------------------------------------------------------
// IGRF_GSM -> IGRF_GSW_08 (geopack2008.for:55-185, GEOGSW_08 :1421-1457): spherical-harmonic synthesis of the main
// field at GSM positions in Earth radii, nT, default REAL (fp32) like the Fortran, for NP points per lane at once.
//  * The expansion is truncated at a degree that falls with distance (NM = 3 + 30/int(r+2), at most 13); the loops run
//    to the wave's largest degree with per-point predicates, so the term index is wave-uniform.
//  * The 105 (g, h, rec) terms live in six registers spread over the wave (lane l: terms l and l + 64 in visiting
//    order), loaded by two coalesced reads per call; term i is a v_readlane with a uniform lane number: no memory
//    access and no wait inside the loops (a load per term -- scalar or vector -- stalls the wave once per term).
//    => CALL IN WAVE-UNIFORM CONTROL FLOW ONLY: v_readlane reads a lane's register whether or not the lane is active,
//    but an inactive lane has not loaded its terms.  (The trace kernel used to synthesise the field at the second
//    end-point estimate inside `if (!first_attempt)`: stale terms, a wrong error estimate, wrong accept / reject
//    decisions -- tests/test_igrf.py::test_gpu_igrf_adaptive_step_control_matches_the_oracle.)
//  * One point per call is a 105-step chain of dependent fp32 operations with one wave per SIMD to hide it behind:
//    the seven stencil points of a right-hand side are therefore synthesised together (independent chains).
//  * The Fortran's A(N), B(N) arrays are the running products r^-(n+1), n r^-(n+1): registers, same multiplication chain.
typedef float f2_t __attribute__((ext_vector_type(2)));
template <int NP>
__device__ __forceinline__ void igrf_core(const FieldConst &f, const float (&xg)[NP], const float (&yg)[NP], const float (&zg)[NP],
                                          float (&hx)[NP], float (&hy)[NP], float (&hz)[NP]) {
#pragma clang fp contract(off)
  const int lane = (int)__lane_id();
  const int g0 = __builtin_bit_cast(int, f.Gv[lane]), g1 = __builtin_bit_cast(int, f.Gv[lane + 64]);
  const int h0 = __builtin_bit_cast(int, f.Hv[lane]), h1 = __builtin_bit_cast(int, f.Hv[lane + 64]);
  const int r0 = __builtin_bit_cast(int, f.Rv[lane]), r1 = __builtin_bit_cast(int, f.Rv[lane + 64]);
  const float a11 = f.A[0], a12 = f.A[1], a13 = f.A[2], a21 = f.A[3], a22 = f.A[4], a23 = f.A[5], a31 = f.A[6], a32 = f.A[7], a33 = f.A[8];
  float c[NP], s[NP], cf[NP], sf[NP], pp[NP], p[NP], d[NP], bbr[NP], bbt[NP], bbf[NP], x[NP], y[NP], am[NP];
  int k[NP];
  bool pole[NP];
  int kmax = 0;
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const float xgeo = a11 * xg[i] + a21 * yg[i] + a31 * zg[i];
    const float ygeo = a12 * xg[i] + a22 * yg[i] + a32 * zg[i];
    const float zgeo = a13 * xg[i] + a23 * yg[i] + a33 * zg[i];
    const float rho2 = xgeo * xgeo + ygeo * ygeo;
    const float r = sqrtf(rho2 + zgeo * zgeo);
    c[i] = zgeo / r;
    const float rho = sqrtf(rho2);
    s[i] = rho / r;
    pole[i] = s[i] < 1.e-5f;
    cf[i] = pole[i] ? 1.f : xgeo / rho;
    sf[i] = pole[i] ? 0.f : ygeo / rho;
    pp[i] = 1.f / r;
    const int irp3 = (int)(r + 2.f);
    int nm = 3 + 30 / (irp3 < 1 ? 1 : irp3);
    if (nm > 13) nm = 13;
    k[i] = nm + 1;
    kmax = k[i] > kmax ? k[i] : kmax;
    p[i] = 1.f;
    d[i] = bbr[i] = bbt[i] = bbf[i] = x[i] = 0.f;
    y[i] = 1.f;
    am[i] = pp[i] * pp[i]; // A(m) = pp^(m+1)
  }
  for (int off = 32; off > 0; off >>= 1) {
    const int o = __shfl_xor(kmax, off, 64);
    kmax = o > kmax ? o : kmax;
  }
  kmax = __builtin_amdgcn_readfirstlane(kmax);
#ifdef SRT_IGRF_PROBE_KMAX
  kmax = kmax < SRT_IGRF_PROBE_KMAX ? kmax : SRT_IGRF_PROBE_KMAX; // (timing probe only: wrong fields)
#endif
  // The points of one stencil lie within 1e-6 of each other: they share the truncation degree unless r + 2 crosses an
  // integer between them.  Then ONE predicate per lane and trip covers all NP chains (the per-point form below costs
  // an exec-mask save/restore per point and trip).
  // A point on the polar axis (sin(colatitude) < 1e-5: `pole`) sums the phi component with dP in place of P; a wave that
  // holds such a point takes the per-point form too (practically never: the paired form then carries no select for it).
  bool samek = !pole[0];
#pragma unroll
  for (int i = 1; i < NP; ++i) samek = samek && k[i] == k[0] && !pole[i];
  bool paired = false;
  if constexpr (NP > 2) if (__all(samek)) {
    paired = true;
    // Two points per register pair: every multiplication and addition below is one v_pk_mul_f32 / v_pk_add_f32 for two
    // chains (IEEE operations, unfused as in the Fortran: the same bits as the one-point form, half the instructions).
    // An odd NP repeats its last point in the spare half.  The n loop is unrolled twice with the roles of (q, z) and
    // (p2, d2) exchanged, so the recurrence's hand-down costs no register moves.
    constexpr int NQ = (NP + 1) / 2;
    const int kl = k[0];
    f2_t C[NQ], S[NQ], CF[NQ], SF[NQ], PP[NQ], P[NQ], D[NQ], BR[NQ], BT[NQ], BF[NQ], X[NQ], Y[NQ], AM[NQ];
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
      const int a = 2 * j, b = 2 * j + 1 < NP ? 2 * j + 1 : NP - 1;
      C[j] = f2_t{c[a], c[b]};
      S[j] = f2_t{s[a], s[b]};
      CF[j] = f2_t{cf[a], cf[b]};
      SF[j] = f2_t{sf[a], sf[b]};
      PP[j] = f2_t{pp[a], pp[b]};
      P[j] = Y[j] = f2_t{1.f, 1.f};
      D[j] = BR[j] = BT[j] = BF[j] = X[j] = f2_t{0.f, 0.f};
      AM[j] = PP[j] * PP[j];
    }
    for (int m = 1; m <= kmax; ++m) {
      f2_t Q[NQ], Z[NQ], BI[NQ], P2[NQ], D2[NQ], AN[NQ];
      const bool mlive = m <= kl;
#pragma unroll
      for (int j = 0; j < NQ; ++j) {
        if (mlive && m > 1) {
          const f2_t w = X[j];
          X[j] = w * CF[j] + Y[j] * SF[j];
          Y[j] = Y[j] * CF[j] - w * SF[j];
        }
        Q[j] = P[j];
        Z[j] = D[j];
        BI[j] = P2[j] = D2[j] = f2_t{0.f, 0.f};
        AN[j] = AM[j];
      }
      const int base = igrf_off(m) - m;
      const int gm = m >= 6 ? g1 : g0, hm = m >= 6 ? h1 : h0, rm = m >= 6 ? r1 : r0; // the row's half of the table
      // one (m, n) term for all chains; the new (q, z) are left in (p2, d2) and the old (q, z) are the next (p2, d2)
      auto term = [&](int n, f2_t (&q)[NQ], f2_t (&z)[NQ], f2_t (&p2)[NQ], f2_t (&d2)[NQ]) {
#pragma clang fp contract(off)
        const int j6 = (base + n) & 63; // wave-uniform
        const float e = __builtin_bit_cast(float, __builtin_amdgcn_readlane(gm, j6));
        const float hh = __builtin_bit_cast(float, __builtin_amdgcn_readlane(hm, j6));
        const float xk = __builtin_bit_cast(float, __builtin_amdgcn_readlane(rm, j6));
        const float fn = (float)n;
        if (n <= kl) {
          // Written row by row over the chains, each row's results passed through an empty volatile asm (no instruction;
          // it pins the order: the row's operations come before it, their consumers after).  A packed operation issued
          // right behind its producer waits ~11 cycles instead of 4 and needs a wait state, and left to itself the
          // scheduler puts a chain's operations next to each other -- with one wave per SIMD there is nothing else to
          // issue meanwhile.
          f2_t t1[NQ], t2[NQ], w[NQ], u[NQ];
#define IGRF_ROW_(dst, expr)                                                                                           \
  _Pragma("unroll") for (int j = 0; j < NQ; ++j) dst[j] = expr;                                                        \
  _Pragma("unroll") for (int j = 0; j < NQ; ++j) asm volatile("" : "+v"(dst[j]))
          IGRF_ROW_(t1, e * Y[j]);
          IGRF_ROW_(t2, hh * X[j]);
          IGRF_ROW_(w, t1[j] + t2[j]);
          IGRF_ROW_(u, AN[j] * fn);
          IGRF_ROW_(u, u[j] * w[j]);
          IGRF_ROW_(u, u[j] * q[j]);
          IGRF_ROW_(BR, BR[j] + u[j]);
          IGRF_ROW_(u, AN[j] * w[j]);
          IGRF_ROW_(u, u[j] * z[j]);
          IGRF_ROW_(BT, BT[j] - u[j]);
          if (m != 1) { // (a compile-time flag for it -- two copies of the loop -- measured: no faster)
            IGRF_ROW_(t1, e * X[j]);
            IGRF_ROW_(t2, hh * Y[j]);
            IGRF_ROW_(w, t1[j] - t2[j]);
            IGRF_ROW_(w, AN[j] * w[j]);
            IGRF_ROW_(u, w[j] * q[j]);
            IGRF_ROW_(BI, BI[j] + u[j]);
          }
          IGRF_ROW_(t1, C[j] * z[j]);
          IGRF_ROW_(t2, S[j] * q[j]);
          IGRF_ROW_(w, t1[j] - t2[j]);
          IGRF_ROW_(t1, xk * d2[j]);
          IGRF_ROW_(d2, w[j] - t1[j]);
          IGRF_ROW_(t1, C[j] * q[j]);
          IGRF_ROW_(t2, xk * p2[j]);
          IGRF_ROW_(p2, t1[j] - t2[j]);
          IGRF_ROW_(AN, AN[j] * PP[j]);
#undef IGRF_ROW_
        }
      };
      for (int n = m; n <= kmax; n += 2) {
        term(n, Q, Z, P2, D2);
        if (n + 1 <= kmax) term(n + 1, P2, D2, Q, Z);
      }
      if (mlive) {
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
          D[j] = S[j] * D[j] + C[j] * P[j];
          P[j] = S[j] * P[j];
          if (m != 1) BF[j] = BF[j] + BI[j] * (float)(m - 1);
          AM[j] = AM[j] * PP[j];
        }
      }
    }
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      bbr[i] = (i & 1) ? BR[i / 2].y : BR[i / 2].x;
      bbt[i] = (i & 1) ? BT[i / 2].y : BT[i / 2].x;
      bbf[i] = (i & 1) ? BF[i / 2].y : BF[i / 2].x;
    }
  }
  if (!paired)
  for (int m = 1; m <= kmax; ++m) {
    float q[NP], z[NP], bi[NP], p2[NP], d2[NP], an[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      if (m <= k[i] && m > 1) {
        const float w = x[i];
        x[i] = w * cf[i] + y[i] * sf[i];
        y[i] = y[i] * cf[i] - w * sf[i];
      }
      q[i] = p[i];
      z[i] = d[i];
      bi[i] = p2[i] = d2[i] = 0.f;
      an[i] = am[i];
    }
    const int base = igrf_off(m) - m;
    const int gm = m >= 6 ? g1 : g0, hm = m >= 6 ? h1 : h0, rm = m >= 6 ? r1 : r0;
    for (int n = m; n <= kmax; ++n) {
      const int j = (base + n) & 63; // wave-uniform
      const float e = __builtin_bit_cast(float, __builtin_amdgcn_readlane(gm, j));
      const float hh = __builtin_bit_cast(float, __builtin_amdgcn_readlane(hm, j));
      const float xk = __builtin_bit_cast(float, __builtin_amdgcn_readlane(rm, j));
      const float fn = (float)n;
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        if (n <= k[i]) { // implies m <= k
          const float w = e * y[i] + hh * x[i];
          bbr[i] = bbr[i] + (an[i] * fn) * w * q[i];
          bbt[i] = bbt[i] - an[i] * w * z[i];
          if (m != 1) bi[i] = bi[i] + an[i] * (e * x[i] - hh * y[i]) * (pole[i] ? z[i] : q[i]);
          const float dp = c[i] * z[i] - s[i] * q[i] - xk * d2[i];
          const float pm = c[i] * q[i] - xk * p2[i];
          d2[i] = z[i];
          p2[i] = q[i];
          z[i] = dp;
          q[i] = pm;
          an[i] = an[i] * pp[i];
        }
      }
    }
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      if (m <= k[i]) {
        d[i] = s[i] * d[i] + c[i] * p[i];
        p[i] = s[i] * p[i];
        if (m != 1) bbf[i] = bbf[i] + bi[i] * (float)(m - 1);
        am[i] = am[i] * pp[i];
      }
    }
  }
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    float bf;
    if (pole[i]) bf = c[i] < 0.f ? -bbf[i] : bbf[i];
    else bf = bbf[i] / s[i];
    const float he = bbr[i] * s[i] + bbt[i] * c[i];
    const float hxgeo = he * cf[i] - bf * sf[i], hygeo = he * sf[i] + bf * cf[i], hzgeo = bbr[i] * c[i] - bbt[i] * s[i];
    hx[i] = a11 * hxgeo + a12 * hygeo + a13 * hzgeo;
    hy[i] = a21 * hxgeo + a22 * hygeo + a23 * hzgeo;
    hz[i] = a31 * hxgeo + a32 * hygeo + a33 * hzgeo;
  }
}

// T04_s on the device: one compiled body (srt_t04.hpp is ~1 000 lines of formulae; it must not be inlined per call site)
__device__ __noinline__ void t04_device(const FieldConst &f, float xg, float yg, float zg, float &tx, float &ty, float &tz) {
  t04::t04_s(f.parmod, f.psi, xg, yg, zg, tx, ty, tz);
}

// The adapters' field tail in full (interp_dens_model_adapter.f95:186,214-267): x_gsm = SM_TO_GSM_d(x); base field in
// GSM nT as REAL -- IGRF_GSM(real(x_gsm/R_E)), or the dipole rotated to GSM; plus T04_s(real(parmod), real(psi),
// real(x_gsm/R_E)) when use_tsyganenko; (base + tsy)*1e-9; GSM_TO_SM_d.
template <int NP, bool IGRF_ONLY = false>
__device__ __forceinline__ void bfield_igrf(const FieldConst &f, const double (&pt)[NP][3], double (&B)[NP][3]) {
  float xg[NP], yg[NP], zg[NP], hx[NP], hy[NP], hz[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    xg[i] = (float)((pt[i][0] * f.cm - pt[i][2] * f.sm) / R_E);
    yg[i] = (float)(pt[i][1] / R_E);
    zg[i] = (float)((pt[i][2] * f.cm + pt[i][0] * f.sm) / R_E);
  }
  if (IGRF_ONLY || f.use_igrf) { // wave-uniform
    igrf_core<NP>(f, xg, yg, zg, hx, hy, hz);
  } else {
#pragma unroll
    for (int i = 0; i < NP; ++i) { // the dipole, as in bfield() below
      const double x = pt[i][0], y = pt[i][1], z = pt[i][2];
      const double rho2 = x * x + y * y, r2 = rho2 + z * z, r = sqrt(r2);
      const double k = fdiv(f.bo_re3, r2 * r2 * r);
      const double bx = -3.0 * k * x * z, by = -3.0 * k * y * z, bz = k * (rho2 - 2.0 * z * z);
      hx[i] = (float)(1.0e9 * (bx * f.cm - bz * f.sm));
      hy[i] = (float)(1.0e9 * by);
      hz[i] = (float)(1.0e9 * (bz * f.cm + bx * f.sm));
    }
  }
  float tx[NP], ty[NP], tz[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) tx[i] = ty[i] = tz[i] = 0.0f;
  if constexpr (!IGRF_ONLY) {
    if (f.use_tsy) {
#pragma unroll
      for (int i = 0; i < NP; ++i) t04_device(f, xg[i], yg[i], zg[i], tx[i], ty[i], tz[i]);
    }
  }
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const double gx = (double)(hx[i] + tx[i]) * 1.0e-9, gy = (double)(hy[i] + ty[i]) * 1.0e-9, gz = (double)(hz[i] + tz[i]) * 1.0e-9;
    B[i][0] = gx * f.cm + gz * f.sm;
    B[i][1] = gy;
    B[i][2] = gz * f.cm - gx * f.sm;
  }
}
__device__ __noinline__ void bfield_igrf1(const FieldConst &f, double x, double y, double z, double B[3]) {
  const double pt[1][3] = {{x, y, z}};
  double b[1][3];
  bfield_igrf<1>(f, pt, b);
  B[0] = b[0][0];
  B[1] = b[0][1];
  B[2] = b[0][2];
}

// ---------------------------------------------------------------------------------------------
// Dipole B in SM coordinates, then the adapters' GSM round trip through float32 nT
// (bmodel_dipole.f95:20-48; interp_dens_model_adapter.f95:243-267 and its twins; SURVEY A-8).
// Trig-free form: B = Bo R_E^3 / r^5 * (-3xz, -3yz, x^2 + y^2 - 2 z^2).
template <class CM>
__device__ __forceinline__ void bfield(const CM &cm, double x, double y, double z, double B[3]) {
  const FieldConst &f = cm.fld;
  if (field_is_igrf(cm)) { // wave-uniform, or a compile-time constant
    bfield_igrf1(f, x, y, z, B);
    return;
  }
  double rho2 = x * x + y * y;
  double r2 = rho2 + z * z;
  double r = sqrt(r2);
  double k = fdiv(f.bo_re3, r2 * r2 * r);
  double bx = -3.0 * k * x * z;
  double by = -3.0 * k * y * z;
  double bz = k * (rho2 - 2.0 * z * z);
  // SM -> GSM: rotate about y by -mu
  double gx = bx * f.cm - bz * f.sm;
  double gz = bz * f.cm + bx * f.sm;
  // B0xBASE = real(1.0e9_DP*B0tmp2(1)); B0tmp = (B0xBASE + 0.0)*1.0e-9_DP
  gx = (double)((float)(1.0e9 * gx)) * 1.0e-9;
  double gy = (double)((float)(1.0e9 * by)) * 1.0e-9;
  gz = (double)((float)(1.0e9 * gz)) * 1.0e-9;
  // GSM -> SM: rotate about y by +mu
  B[0] = gx * f.cm + gz * f.sm;
  B[1] = gy;
  B[2] = gz * f.cm - gx * f.sm;
}

// ---------------------------------------------------------------------------------------------
// stix_parameters (raytracer.f95:81-102) + the products dispersion_relation needs.
struct Stix {
  double S, D, P, R, L;
  double RL, PS, RLP;
  bool freespace; // raytracer.f95:65 (mis-parenthesised threshold, SURVEY A-5)
};

__device__ __forceinline__ Stix stix_parameters(const Species &sp, double w, const double Ns[MAXSPEC],
                                                double Bmag) {
  double sr = 0.0, sl = 0.0, sw = 0.0, maxN = 0.0;
#pragma unroll
  for (int s = 0; s < MAXSPEC; ++s) {
    double wps2 = Ns[s] * sp.c[s];
    double wcs = sp.g[s] * Bmag;
    double a = w + wcs, b = w - wcs;
    double t = fdiv(wps2, w * a * b); // one division serves both the R and the L term
    sr += t * b;                   // wps2/(w*(w+wcs))
    sl += t * a;                   // wps2/(w*(w-wcs))
    sw += wps2;
    maxN = fmax(maxN, Ns[s]);
  }
  Stix st;
  st.R = 1.0 - sr;
  st.L = 1.0 - sl;
  st.P = 1.0 - fdiv(sw, w * w);
  st.S = 0.5 * (st.R + st.L);
  st.D = 0.5 * (st.R - st.L);
  st.RL = st.R * st.L;
  st.PS = st.P * st.S;
  st.RLP = st.RL * st.P;
  // w > 100*sqrt(maxN*maxq^2)/(minm*eps0)
  double lhs = w * sp.minm_eps0 * 0.01;
  st.freespace = (lhs * lhs > maxN * sp.maxq2) && lhs > 0.0;
  return st;
}

// dispersion_relation (raytracer.f95:41-72) for refractive index n, given the Stix parameters.
__device__ __forceinline__ double dispersion_F(const Stix &st, const double n[3], const double B[3],
                                               double B2) {
  double nmag2 = n[0] * n[0] + n[1] * n[1] + n[2] * n[2];
  double nb = n[0] * B[0] + n[1] * B[1] + n[2] * B[2];
  double cos2 = fdiv(nb * nb, nmag2 * B2);
  double sin2 = 1.0 - cos2;
  double A = st.S * sin2 + st.P * cos2;
  double Bq = st.RL * sin2 + st.PS * (1.0 + cos2);
  double F = A * (nmag2 * nmag2) - Bq * nmag2 + st.RLP;
  return st.freespace ? (1.0 - nmag2) : F;
}

__device__ __forceinline__ double fd_step(double del, double v) {
  double a = del * fabs(v);
  return (a > del) ? a : del; // max(del*abs(v), del)  (raytracer.f95:139,192,239)
}

// dispersion_relation_dFdk with del = 1e-8 (raytracer.f95:118-155), plasma state given.
__device__ __forceinline__ void dFdk(const Stix &st, const double k[3], double cw /* C/w */,
                                     const double B[3], double B2, double out[3]) {
  const double del = 1.0e-8;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    double d = fd_step(del, k[c]);
    double np[3] = {k[0] * cw, k[1] * cw, k[2] * cw};
    double nm[3] = {np[0], np[1], np[2]};
    np[c] = (k[c] + d) * cw;
    nm[c] = (k[c] - d) * cw;
    out[c] = fdiv(dispersion_F(st, np, B, B2) - dispersion_F(st, nm, B, B2), d) * 0.5;
  }
}

// dispersion_relation_dFdw with del = 1e-8 (raytracer.f95:172-198).
__device__ __forceinline__ double dFdw(const Species &sp, const double k[3], double w, double C,
                                       const double Ns[MAXSPEC], const double B[3], double B2,
                                       double Bmag) {
  const double del = 1.0e-8;
  double d = fd_step(del, w);
  double wp = w + d, wm = w - d;
  Stix sp_ = stix_parameters(sp, wp, Ns, Bmag);
  Stix sm_ = stix_parameters(sp, wm, Ns, Bmag);
  double cp = fdiv(C, wp), cm = fdiv(C, wm);
  double np[3] = {k[0] * cp, k[1] * cp, k[2] * cp};
  double nm[3] = {k[0] * cm, k[1] * cm, k[2] * cm};
  return fdiv(dispersion_F(sp_, np, B, B2) - dispersion_F(sm_, nm, B, B2), d) * 0.5;
}

// ---------------------------------------------------------------------------------------------
// is_right_handed (raytracer.f95:355-405) without the complex SVD.
//
// The reference builds M = [[a,-iD,b],[iD,c,0],[b,0,d]] (entries rounded to float32 by default-kind
// cmplx, phi passed in DEGREES to cos/sin -- SURVEY A-4), takes E = row 3 of V^H from zgesvd and tests
// the x-y rotation sense from Re(E) to Re(iE).  M is unitarily similar to the real symmetric
// T = [[a,D,b],[D,c,0],[b,0,d]]; with lambda = eigenvalue of T of least magnitude the test equals
// .not.( D/(c-lambda) > 0 ).  Eigenvalues by fixed-sweep cyclic Jacobi, fully unrolled on scalars.
__device__ __forceinline__ void jacobi_rot(double &app, double &aqq, double &apq, double &arp, double &arq) {
  // annihilate apq; r is the third index
  if (apq != 0.0) {
    double theta = (aqq - app) / (2.0 * apq);
    double t = copysign(1.0, theta) / (fabs(theta) + sqrt(theta * theta + 1.0));
    double cs = 1.0 / sqrt(t * t + 1.0), sn = t * cs;
    app = app - t * apq;
    aqq = aqq + t * apq;
    apq = 0.0;
    double nrp = cs * arp - sn * arq;
    double nrq = sn * arp + cs * arq;
    arp = nrp;
    arq = nrq;
  }
}

__device__ inline bool is_right_handed(double n2, double phi_deg, double S, double D, double P) {
  double sp, cp;
  sincos(phi_deg, &sp, &cp); // degrees fed to cos/sin, as the reference does (raytracer.f95:450 -> :361)
  double a00 = (double)(float)(S - n2 * (cp * cp));
  double a01 = (double)(float)(D);
  double a02 = (double)(float)(n2 * cp * sp);
  double a11 = (double)(float)(S - n2);
  double a12 = 0.0;
  double a22 = (double)(float)(P - n2 * (sp * sp));
  const double c = a11, Dm = a01;
#pragma unroll 1
  for (int sweep = 0; sweep < 8; ++sweep) {
    jacobi_rot(a00, a11, a01, a02, a12); // (p,q)=(0,1), r=2
    jacobi_rot(a00, a22, a02, a01, a12); // (0,2), r=1
    jacobi_rot(a11, a22, a12, a01, a02); // (1,2), r=0
  }
  double lam = a00;
  if (fabs(a11) < fabs(lam)) lam = a11;
  if (fabs(a22) < fabs(lam)) lam = a22;
  return !(Dm / (c - lam) > 0.0);
}

// solve_dispersion_relation (raytracer.f95:408-502): complex roots k1, k2 (re, im) along direction k.
struct Roots {
  double k1re, k1im, k2re, k2im;
};
__device__ __forceinline__ void csqrt_real_or_imag(double re, double im, double &ore, double &oim) {
  // principal square root of re + i*im
  if (im == 0.0) {
    if (re >= 0.0) {
      ore = sqrt(re);
      oim = 0.0;
    } else {
      ore = 0.0;
      oim = sqrt(-re);
    }
    return;
  }
  double m = hypot(re, im);
  double a = sqrt(0.5 * (m + fabs(re)));
  double b = im / (2.0 * a);
  if (re >= 0.0) {
    ore = a;
    oim = b;
  } else {
    ore = fabs(b);
    oim = copysign(a, im);
  }
}
__device__ inline Roots solve_dispersion(const Common &cm, const double kdir[3], double w,
                                         const double Ns[MAXSPEC], const double B[3]) {
  double B2 = B[0] * B[0] + B[1] * B[1] + B[2] * B[2];
  double kb = kdir[0] * B[0] + kdir[1] * B[1] + kdir[2] * B[2];
  double kk = kdir[0] * kdir[0] + kdir[1] * kdir[1] + kdir[2] * kdir[2];
  double cos2 = (kb * kb) / (kk * B2);
  double sin2 = 1.0 - cos2;
  // The reference lets cos2 exceed 1 by rounding and dies in zgesvd on the NaN (SURVEY A-2); clamp.
  double phi = acos(sqrt(fmin(cos2, 1.0))) * (180.0 / PI);
  Stix st = stix_parameters(cm.sp, w, Ns, sqrt(B2));
  double A = st.S * sin2 + st.P * cos2;
  double Bq = st.RL * sin2 + st.PS * (1.0 + cos2);
  double disc = Bq * Bq - 4.0 * A * st.RLP;
  double sre, sim;
  csqrt_real_or_imag(disc, 0.0, sre, sim);
  double inv2A = 1.0 / (2.0 * A);
  double q1re = (Bq + sre) * inv2A, q1im = sim * inv2A;
  double q2re = (Bq - sre) * inv2A, q2im = -sim * inv2A;
  double n1re, n1im, n2re, n2im;
  csqrt_real_or_imag(q1re, q1im, n1re, n1im);
  csqrt_real_or_imag(q2re, q2im, n2re, n2im);
  double wc = w / cm.C;
  Roots r;
  bool swap = (n1re > 0.0) && is_right_handed(q1re, phi, st.S, st.D, st.P);
  if (swap) {
    r.k1re = wc * n2re; r.k1im = wc * n2im; r.k2re = wc * n1re; r.k2im = wc * n1im;
  } else {
    r.k1re = wc * n1re; r.k1im = wc * n1im; r.k2re = wc * n2re; r.k2im = wc * n2im;
  }
  return r;
}

// raytracer_stopconditions (raytracer.f95:324-353)
__device__ __forceinline__ int stop_conditions(const double x[6], const double vg[3], double dt, int nstep,
                                               int maxsteps, double minalt) {
  if (sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]) < minalt) return 1;
  if (sqrt(x[3] * x[3] + x[4] * x[4] + x[5] * x[5]) == 0.0) return 2;
  if (sqrt(vg[0] * vg[0] + vg[1] * vg[1] + vg[2] * vg[2]) > 1.0 + 1e-2) return 3;
  if (dt < (double)1e-14f) return 5;
  if (nstep >= maxsteps) return 6;
  return 0;
}

// RKF45 tableau (raytracer.f95:8-27)
namespace rkf {
constexpr double a21 = 1.0 / 4.0;
constexpr double a31 = 3.0 / 32.0, a32 = 9.0 / 32.0;
constexpr double a41 = 1932.0 / 2197.0, a42 = -7200.0 / 2197.0, a43 = 7296.0 / 2197.0;
constexpr double a51 = 439.0 / 216.0, a52 = -8.0, a53 = 3680.0 / 513.0, a54 = -845.0 / 4104.0;
constexpr double a61 = -8.0 / 27.0, a62 = 2.0, a63 = -3544.0 / 2565.0, a64 = 1859.0 / 4104.0, a65 = -11.0 / 40.0;
constexpr double b41 = 25.0 / 216.0, b43 = 1408.0 / 2565.0, b44 = 2197.0 / 4104.0, b45 = -1.0 / 5.0;
constexpr double b51 = 16.0 / 135.0, b53 = 6656.0 / 12825.0, b54 = 28561.0 / 56430.0, b55 = -9.0 / 50.0,
                 b56 = 2.0 / 55.0;
} // namespace rkf

} // namespace srt
