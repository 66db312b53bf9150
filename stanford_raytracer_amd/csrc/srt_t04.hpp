// srt_t04.hpp -- the T04_s (Tsyganenko & Sitnov 2005, "TS05") external magnetic field, use_tsyganenko = 1 of the
// adapters' field tail (interp_dens_model_adapter.f95:223-258: B0xTsy from T04_s(iopt, real(parmod), real(psi),
// real(x_gsm/R_E))).  Reference source: tsyganenko/TS05_aka_TS04.for (routine:line cited per function).
//
// The model is a sum of six modules (Chapman-Ferraro shielding of the dipole, two tail-current modes, symmetric and
// partial ring current, Region 1 and Region 2 Birkeland currents, IMF penetration), each an analytic expression in
// position, dipole tilt and the fitted coefficients (DATA: srt_t04_tables.h, generated from the published tables).
// Internally REAL*8 like the Fortran (IMPLICIT REAL*8), the interface is REAL.  Where the Fortran spells out 3 x 3 or
// 5 x 5 harmonics term by term, loops are used here with the same order of summation.
// Compiles for the device (hipcc) and for the host (tests/test_t04_host.py builds it with g++).
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define T04_HD __host__ __device__
#else
#define T04_HD
#endif

#if defined(__HIP_DEVICE_COMPILE__)
#include "srt_fastmath.hpp"
#endif

namespace srt {
namespace t04 {
#include "srt_t04_tables.h"

// Elementary functions.  Host (tests, the CLI's host-side checks): libm, as the Fortran's run-time library.  Device: the
// range-specialised kernels of srt_fastmath.hpp -- one evaluation of T04_s makes ~440 sin, ~380 cos, ~580 sqrt, ~200 exp and
// ~120 pow calls, which with the library's full-range versions were ~85 % of its 1.3e5 instructions.  Arguments are positions
// in Earth radii over fitted scale lengths and tilt angles (|x| < 1e3); results agree with libm to <= 2 ulp, far inside the
// REAL interface of T04_s (tests/test_t04.py holds the device field to the reference's at 3e-6).
#if defined(__HIP_DEVICE_COMPILE__)
T04_HD static inline double t_sin(double x) { return ::srt::fm::sin_mod(x); }
T04_HD static inline double t_cos(double x) { return ::srt::fm::cos_mod(x); }
T04_HD static inline double t_exp(double x) { return ::srt::fm::exp_any(x); }
T04_HD static inline double t_pow(double x, double y) { return ::srt::fm::pow_pos(x, y); }
T04_HD static inline double t_sqrt(double x) { return ::srt::fm::sqrt_pos(x); }
T04_HD static inline double t_log(double x) { return x > 0.0 ? ::srt::fm::log_pos(x) : ::log(x); }
#else
T04_HD static inline double t_sin(double x) { return ::sin(x); }
T04_HD static inline double t_cos(double x) { return ::cos(x); }
T04_HD static inline double t_exp(double x) { return ::exp(x); }
T04_HD static inline double t_pow(double x, double y) { return ::pow(x, y); }
T04_HD static inline double t_sqrt(double x) { return ::sqrt(x); }
T04_HD static inline double t_log(double x) { return ::log(x); }
#endif

struct V3 {
  double x, y, z;
};
T04_HD static inline double sq(double v) { return v * v; }

// SHLCAR3X3 (:362-692): shielding field of the tilted dipole, 9 "perpendicular" + 9 "parallel" box harmonics
T04_HD static inline V3 shlcar3x3(double X, double Y, double Z, double PS) {
  const double *A = T04D_SHLCAR3X3_A;
  const double P[3] = {A[36], A[37], A[38]}, R[3] = {A[39], A[40], A[41]}, Q[3] = {A[42], A[43], A[44]},
               S[3] = {A[45], A[46], A[47]};
  const double T1 = A[48], T2 = A[49];
  const double CPS = t_cos(PS), SPS = t_sin(PS), S2PS = 2.0 * CPS;
  const double ST1 = t_sin(PS * T1), CT1 = t_cos(PS * T1), ST2 = t_sin(PS * T2), CT2 = t_cos(PS * T2);
  const double X1 = X * CT1 - Z * ST1, Z1 = X * ST1 + Z * CT1, X2 = X * CT2 - Z * ST2, Z2 = X * ST2 + Z * CT2;
  V3 B = {0.0, 0.0, 0.0};
  int l = 0;
  for (int i = 0; i < 3; ++i)
    for (int k = 0; k < 3; ++k, l += 2) {
      const double SQPR = t_sqrt(1.0 / sq(P[i]) + 1.0 / sq(R[k]));
      const double CYP = t_cos(Y / P[i]), SYP = t_sin(Y / P[i]), CZR = t_cos(Z1 / R[k]), SZR = t_sin(Z1 / R[k]);
      const double EXPR = t_exp(SQPR * X1);
      double FX, HY, FZ;
      if (k < 2) {
        FX = -SQPR * EXPR * CYP * SZR;
        HY = EXPR / P[i] * SYP * SZR;
        FZ = -EXPR * CYP / R[k] * CZR;
      } else {
        FX = -EXPR * CYP * (SQPR * Z1 * CZR + SZR / R[k] * (X1 + 1.0 / SQPR));
        HY = EXPR / P[i] * SYP * (Z1 * CZR + X1 / R[k] * SZR / SQPR);
        FZ = -EXPR * CYP * (CZR * (1.0 + X1 / sq(R[k]) / SQPR) - Z1 / R[k] * SZR);
      }
      const double HX = FX * CT1 + FZ * ST1, HZ = -FX * ST1 + FZ * CT1;
      const double a = A[l] + A[l + 1] * CPS;
      B.x += a * HX;
      B.y += a * HY;
      B.z += a * HZ;
    }
  for (int i = 0; i < 3; ++i)
    for (int k = 0; k < 3; ++k, l += 2) {
      const double SQQS = t_sqrt(1.0 / sq(Q[i]) + 1.0 / sq(S[k]));
      const double CYQ = t_cos(Y / Q[i]), SYQ = t_sin(Y / Q[i]), CZS = t_cos(Z2 / S[k]), SZS = t_sin(Z2 / S[k]);
      const double EXQS = t_exp(SQQS * X2);
      const double FX = -SQQS * EXQS * CYQ * CZS * SPS;
      const double HY = EXQS / Q[i] * SYQ * CZS * SPS;
      const double FZ = EXQS * CYQ / S[k] * SZS * SPS;
      const double HX = FX * CT2 + FZ * ST2, HZ = -FX * ST2 + FZ * CT2;
      const double a = A[l] + A[l + 1] * S2PS;
      B.x += a * HX;
      B.y += a * HY;
      B.z += a * HZ;
    }
  return B;
}

// TAILDISK (:933-1022): field of a current disk of variable thickness, 5 terms
T04_HD static inline V3 taildisk(double D0, double DELTADX, double DELTADY, double X, double Y, double Z) {
  const double *F = T04D_TAILDISK_F, *Bc = T04D_TAILDISK_B, *C = T04D_TAILDISK_C;
  const double RHO = t_sqrt(X * X + Y * Y);
  const double DRHODX = X / RHO, DRHODY = Y / RHO;
  const double DEX = t_exp(X / 7.0);
  const double D = D0 + DELTADY * sq(Y / 20.0) + DELTADX * DEX;
  const double DDDY = DELTADY * Y * 0.005, DDDX = DELTADX / 7.0 * DEX;
  const double DZETA = t_sqrt(Z * Z + D * D);
  const double DDZETADX = D * DDDX / DZETA, DDZETADY = D * DDDY / DZETA, DDZETADZ = Z / DZETA;
  V3 B = {0.0, 0.0, 0.0};
  for (int i = 0; i < 5; ++i) {
    const double BI = Bc[i], CI = C[i];
    const double S1 = t_sqrt(sq(RHO + BI) + sq(DZETA + CI)), S2 = t_sqrt(sq(RHO - BI) + sq(DZETA + CI));
    const double DS1DRHO = (RHO + BI) / S1, DS2DRHO = (RHO - BI) / S2, DS1DDZ = (DZETA + CI) / S1, DS2DDZ = (DZETA + CI) / S2;
    const double DS1DX = DS1DRHO * DRHODX + DS1DDZ * DDZETADX, DS1DY = DS1DRHO * DRHODY + DS1DDZ * DDZETADY, DS1DZ = DS1DDZ * DDZETADZ;
    const double DS2DX = DS2DRHO * DRHODX + DS2DDZ * DDZETADX, DS2DY = DS2DRHO * DRHODY + DS2DDZ * DDZETADY, DS2DZ = DS2DDZ * DDZETADZ;
    const double S1TS2 = S1 * S2, S1PS2 = S1 + S2, S1PS2SQ = S1PS2 * S1PS2;
    const double FAC1 = t_sqrt(S1PS2SQ - sq(2.0 * BI));
    const double AS = FAC1 / (S1TS2 * S1PS2SQ);
    const double DASDS1 = (1.0 / (FAC1 * S2) - AS / S1PS2 * (S2 * S2 + S1 * (3.0 * S1 + 4.0 * S2))) / (S1 * S1PS2);
    const double DASDS2 = (1.0 / (FAC1 * S1) - AS / S1PS2 * (S1 * S1 + S2 * (3.0 * S2 + 4.0 * S1))) / (S2 * S1PS2);
    const double DASDX = DASDS1 * DS1DX + DASDS2 * DS2DX, DASDY = DASDS1 * DS1DY + DASDS2 * DS2DY, DASDZ = DASDS1 * DS1DZ + DASDS2 * DS2DZ;
    B.x = B.x - F[i] * X * DASDZ;
    B.y = B.y - F[i] * Y * DASDZ;
    B.z = B.z + F[i] * (2.0 * AS + X * DASDX + Y * DASDY);
  }
  return B;
}

// SHLCAR5X5 (:1024-1076): shielding field of a tail mode, 5 x 5 box harmonics, coefficients linear in the shift
T04_HD static inline V3 shlcar5x5(const double *A, double X, double Y, double Z, double DSHIFT) {
  V3 H = {0.0, 0.0, 0.0};
  int l = 0;
  for (int i = 0; i < 5; ++i) {
    const double RP = 1.0 / A[50 + i];
    const double CYPI = t_cos(Y * RP), SYPI = t_sin(Y * RP);
    for (int k = 0; k < 5; ++k, l += 2) {
      const double RR = 1.0 / A[55 + k];
      const double SZRK = t_sin(Z * RR), CZRK = t_cos(Z * RR);
      const double SQPR = t_sqrt(RP * RP + RR * RR);
      const double EPR = t_exp(X * SQPR);
      const double DBX = -SQPR * EPR * CYPI * SZRK, DBY = RP * EPR * SYPI * SZRK, DBZ = -RR * EPR * CYPI * CZRK;
      const double COEF = A[l] + A[l + 1] * DSHIFT;
      H.x += COEF * DBX;
      H.y += COEF * DBY;
      H.z += COEF * DBZ;
    }
  }
  return H;
}

struct TailPar { // COMMON /TAIL/
  double DXSHIFT1, DXSHIFT2, D, DELTADY;
};

// UNWARPED (:837-931), IOPT = 0: both tail modes in unwarped coordinates
T04_HD static inline void unwarped(const TailPar &tp, double X, double Y, double Z, V3 &B1, V3 &B2) {
  const double DELTADX1 = T04D_UNWARPED_DELTADX1[0], ALPHA1 = T04D_UNWARPED_DELTADX1[1], XSHIFT1 = T04D_UNWARPED_DELTADX1[2];
  const double DELTADX2 = T04D_UNWARPED_DELTADX2[0], ALPHA2 = T04D_UNWARPED_DELTADX2[1], XSHIFT2 = T04D_UNWARPED_DELTADX2[2];
  const double XM1 = T04D_UNWARPED_XM1[0], XM2 = T04D_UNWARPED_XM1[1];
  {
    const double XSC = (X - XSHIFT1 - tp.DXSHIFT1) * ALPHA1 - XM1 * (ALPHA1 - 1.0);
    const V3 F = taildisk(tp.D * ALPHA1, DELTADX1, tp.DELTADY, XSC, Y * ALPHA1, Z * ALPHA1);
    const V3 H = shlcar5x5(T04D_UNWARPED_A1, X, Y, Z, tp.DXSHIFT1);
    B1 = {F.x + H.x, F.y + H.y, F.z + H.z};
  }
  {
    const double XSC = (X - XSHIFT2 - tp.DXSHIFT2) * ALPHA2 - XM2 * (ALPHA2 - 1.0);
    const V3 F = taildisk(tp.D * ALPHA2, DELTADX2, tp.DELTADY, XSC, Y * ALPHA2, Z * ALPHA2);
    const V3 H = shlcar5x5(T04D_UNWARPED_A2, X, Y, Z, tp.DXSHIFT2);
    B2 = {F.x + H.x, F.y + H.y, F.z + H.z};
  }
}

// WARPED (:764-835): twisting of the tail current sheet about the x axis (G = /Gblock/)
T04_HD static inline void warped(const TailPar &tp, double G, double PS, double X, double Y, double Z, V3 &B1, V3 &B2) {
  const double DGDX = 0.0, XL = 20.0, DXLDX = 0.0;
  const double SPS = t_sin(PS);
  const double RHO2 = Y * Y + Z * Z, RHO = t_sqrt(RHO2);
  double PHI, CPHI, SPHI;
  if (Y == 0.0 && Z == 0.0) {
    PHI = 0.0;
    CPHI = 1.0;
    SPHI = 0.0;
  } else {
    PHI = atan2(Z, Y);
    CPHI = Y / RHO;
    SPHI = Z / RHO;
  }
  const double XL4 = XL * XL * XL * XL;
  const double RR4L4 = RHO / (RHO2 * RHO2 + XL4);
  const double F = PHI + G * RHO2 * RR4L4 * CPHI * SPS;
  const double DFDPHI = 1.0 - G * RHO2 * RR4L4 * SPHI * SPS;
  const double DFDRHO = G * RR4L4 * RR4L4 * (3.0 * XL4 - RHO2 * RHO2) * CPHI * SPS;
  const double DFDX = RR4L4 * CPHI * SPS * (DGDX * RHO2 - G * RHO * RR4L4 * 4.0 * XL * XL * XL * DXLDX);
  const double CF = t_cos(F), SF = t_sin(F);
  V3 A1, A2;
  unwarped(tp, X, RHO * CF, RHO * SF, A1, A2);
  auto deform = [&](const V3 &A) {
    const double BRHO_AS = A.y * CF + A.z * SF, BPHI_AS = -A.y * SF + A.z * CF;
    const double BRHO_S = BRHO_AS * DFDPHI, BPHI_S = BPHI_AS - RHO * (A.x * DFDX + BRHO_AS * DFDRHO);
    return V3{A.x * DFDPHI, BRHO_S * CPHI - BPHI_S * SPHI, BRHO_S * SPHI + BPHI_S * CPHI};
  };
  B1 = deform(A1);
  B2 = deform(A2);
}

// DEFORMED (:694-762): tilt-dependent bending of the tail current sheet (RH0 = /RH0block/)
T04_HD static inline void deformed(const TailPar &tp, double RH0, double G, double PS, double X, double Y, double Z, V3 &B1, V3 &B2) {
  const double RH2 = T04D_DEFORMED_RH2[0];
  const int IEPS = 3;
  const double SPS = t_sin(PS);
  const double R2 = X * X + Y * Y + Z * Z, R = t_sqrt(R2), ZR = Z / R;
  const double RH = RH0 + RH2 * ZR * ZR;
  const double DRHDR = -ZR / R * 2.0 * RH2 * ZR, DRHDZ = 2.0 * RH2 * ZR / R;
  const double RRH = R / RH;
  const double F = 1.0 / t_pow(1.0 + RRH * RRH * RRH, 1.0 / IEPS);
  const double F2 = F * F;
  const double DFDR = -((RRH * RRH) * (F2 * F2)) / RH;
  const double DFDRH = -RRH * DFDR;
  const double SPSAS = SPS * F, CPSAS = t_sqrt(1.0 - SPSAS * SPSAS);
  const double XAS = X * CPSAS - Z * SPSAS, ZAS = X * SPSAS + Z * CPSAS;
  const double FACPS = SPS / CPSAS * (DFDR + DFDRH * DRHDR) / R;
  const double PSASX = FACPS * X, PSASY = FACPS * Y, PSASZ = FACPS * Z + SPS / CPSAS * DFDRH * DRHDZ;
  const double DXASDX = CPSAS - ZAS * PSASX, DXASDY = -ZAS * PSASY, DXASDZ = -SPSAS - ZAS * PSASZ;
  const double DZASDX = SPSAS + XAS * PSASX, DZASDY = XAS * PSASY, DZASDZ = CPSAS + XAS * PSASZ;
  const double FAC1 = DXASDZ * DZASDY - DXASDY * DZASDZ, FAC2 = DXASDX * DZASDZ - DXASDZ * DZASDX,
               FAC3 = DZASDX * DXASDY - DXASDX * DZASDY;
  V3 A1, A2;
  warped(tp, G, PS, XAS, Y, ZAS, A1, A2);
  auto back = [&](const V3 &A) {
    return V3{A.x * DZASDZ - A.z * DXASDZ + A.y * FAC1, A.y * FAC2, A.z * DXASDX - A.x * DZASDX + A.y * FAC3};
  };
  B1 = back(A1);
  B2 = back(A2);
}

// ---- modules still to come are declared here so that EXTERN reads top-down ----
struct BirkOut {
  V3 r11, r12, r21, r22;
};
T04_HD static inline BirkOut birk_tot(double XKAPPA1, double XKAPPA2, double PS, double X, double Y, double Z);
T04_HD static inline void full_rc(double SC_SY, double SC_AS, double PHI, double PS, double X, double Y, double Z, V3 &SRC, V3 &PRC);

// DIPOLE (:2514-2543): the geodipole in GSM for tilt PS (subtracted outside the magnetopause)
T04_HD static inline V3 dipole(double PS, double X, double Y, double Z) {
  const double SPS = t_sin(PS), CPS = t_cos(PS);
  const double P = X * X, U = Z * Z, V = 3.0 * Z * X, T = Y * Y;
  const double Q = 30115.0 / t_pow(t_sqrt(P + T + U), 5);
  return V3{Q * ((T + U - 2.0 * P) * SPS - V * CPS), -3.0 * Y * Q * (X * SPS + Z * CPS), Q * ((P + T - 2.0 * U) * CPS - V * SPS)};
}

struct Components { // EXTERN's module outputs, for tests
  V3 cf, t1, t2, src, prc, r11, r12, r21, r22, himf, total;
};

// EXTERN (:118-360) with IOPGEN = IOPT = IOPB = IOPR = 0, as T04_s calls it
T04_HD static inline Components external_field(const double *A, double PDYN, double DST, double BYIMF, double BZIMF, double W1, double W2,
                                               double W3, double W4, double W5, double W6, double PS, double X, double Y, double Z) {
  const double A0_A = T04D_EXTERN_A0_A[0], A0_S0 = T04D_EXTERN_A0_A[1], A0_X0 = T04D_EXTERN_A0_A[2];
  const double DSIG = T04D_EXTERN_DSIG[0], RH2 = T04D_EXTERN_RH0[1];
  Components o = {};
  const double XAPPA = t_pow(PDYN / 2.0, A[22]);
  const double RH0 = 7.5, G = 35.0;
  const double XAPPA3 = XAPPA * XAPPA * XAPPA;
  const double XX = X * XAPPA, YY = Y * XAPPA, ZZ = Z * XAPPA;
  const double SPS = t_sin(PS);
  const double X0 = A0_X0 / XAPPA, AM = A0_A / XAPPA, S0 = A0_S0;
  const double FACTIMF = A[19];
  const double OIMFX = 0.0, OIMFY = BYIMF * FACTIMF, OIMFZ = BZIMF * FACTIMF;
  const double R = t_sqrt(X * X + Y * Y + Z * Z);
  double XSS = X, ZSS = Z, DD;
  do { // iterative search of the unwarped coordinates (to find SIGMA)
    const double XSOLD = XSS, ZSOLD = ZSS;
    const double RH = RH0 + RH2 * sq(ZSS / R);
    const double SINPSAS = SPS / t_pow(1.0 + (R / RH) * (R / RH) * (R / RH), 0.33333333);
    const double COSPSAS = t_sqrt(1.0 - SINPSAS * SINPSAS);
    ZSS = X * SINPSAS + Z * COSPSAS;
    XSS = X * COSPSAS - Z * SINPSAS;
    DD = fabs(XSS - XSOLD) + fabs(ZSS - ZSOLD);
  } while (DD > 1.0e-6);
  const double RHO2 = Y * Y + ZSS * ZSS;
  const double ASQ = AM * AM;
  double XMXM = AM + XSS - X0;
  if (XMXM < 0.0) XMXM = 0.0;
  const double AXX0 = XMXM * XMXM, ARO = ASQ + RHO2;
  const double SIGMA = t_sqrt((ARO + AXX0 + t_sqrt(sq(ARO + AXX0) - 4.0 * ASQ * AXX0)) / (2.0 * ASQ));
  if (SIGMA < S0 + DSIG) {
    const V3 CF = shlcar3x3(XX, YY, ZZ, PS);
    o.cf = {CF.x * XAPPA3, CF.y * XAPPA3, CF.z * XAPPA3};
    {
      double DSTT = -20.0;
      if (DST < DSTT) DSTT = DST;
      const double ZNAM = t_pow(fabs(DSTT), (double)0.37f); // `**0.37`: a default-REAL literal
      TailPar tp;
      tp.DXSHIFT1 = A[23] - A[24] / ZNAM;
      tp.DXSHIFT2 = A[25] - A[26] / ZNAM;
      tp.D = A[35] * t_exp(-W1 / A[36]) + A[68];
      tp.DELTADY = (double)4.7f; // `DELTADY=4.7`: a default-REAL literal
      deformed(tp, RH0, G, PS, XX, YY, ZZ, o.t1, o.t2);
    }
    {
      double ZNAM = fabs(DST);
      if (DST >= -20.0) ZNAM = 20.0;
      const double XKAPPA1 = A[31] * t_pow(ZNAM / 20.0, A[32]), XKAPPA2 = A[33] * t_pow(ZNAM / 20.0, A[34]);
      const BirkOut b = birk_tot(XKAPPA1, XKAPPA2, PS, XX, YY, ZZ);
      o.r11 = b.r11;
      o.r12 = b.r12;
      o.r21 = b.r21;
      o.r22 = b.r22;
    }
    {
      const double PHI = A[37];
      double ZNAM = fabs(DST);
      if (DST >= -20.0) ZNAM = 20.0;
      const double SC_SY = A[27] * t_pow(20.0 / ZNAM, A[28]) * XAPPA, SC_AS = A[29] * t_pow(20.0 / ZNAM, A[30]) * XAPPA;
      full_rc(SC_SY, SC_AS, PHI, PS, XX, YY, ZZ, o.src, o.prc);
    }
    o.himf = {0.0, BYIMF, BZIMF};
    const double DLP1 = t_pow(PDYN / 2.0, A[20]), DLP2 = t_pow(PDYN / 2.0, A[21]);
    const double TAMP1 = A[1] + A[2] * DLP1 + A[3] * A[38] * W1 / t_sqrt(W1 * W1 + A[38] * A[38]) + A[4] * DST;
    const double TAMP2 = A[5] + A[6] * DLP2 + A[7] * A[39] * W2 / t_sqrt(W2 * W2 + A[39] * A[39]) + A[8] * DST;
    const double A_SRC = A[9] + A[10] * A[40] * W3 / t_sqrt(W3 * W3 + A[40] * A[40]) + A[11] * DST;
    const double A_PRC = A[12] + A[13] * A[41] * W4 / t_sqrt(W4 * W4 + A[41] * A[41]) + A[14] * DST;
    const double A_R11 = A[15] + A[16] * A[42] * W5 / t_sqrt(W5 * W5 + A[42] * A[42]);
    const double A_R21 = A[17] + A[18] * A[43] * W6 / t_sqrt(W6 * W6 + A[43] * A[43]);
    const double BBX = A[0] * o.cf.x + TAMP1 * o.t1.x + TAMP2 * o.t2.x + A_SRC * o.src.x + A_PRC * o.prc.x + A_R11 * o.r11.x + A_R21 * o.r21.x + A[19] * o.himf.x;
    const double BBY = A[0] * o.cf.y + TAMP1 * o.t1.y + TAMP2 * o.t2.y + A_SRC * o.src.y + A_PRC * o.prc.y + A_R11 * o.r11.y + A_R21 * o.r21.y + A[19] * o.himf.y;
    const double BBZ = A[0] * o.cf.z + TAMP1 * o.t1.z + TAMP2 * o.t2.z + A_SRC * o.src.z + A_PRC * o.prc.z + A_R11 * o.r11.z + A_R21 * o.r21.z + A[19] * o.himf.z;
    if (SIGMA < S0 - DSIG) {
      o.total = {BBX, BBY, BBZ};
    } else { // the boundary layer: interpolation between the inside field and the IMF
      const double FINT = 0.5 * (1.0 - (SIGMA - S0) / DSIG), FEXT = 0.5 * (1.0 + (SIGMA - S0) / DSIG);
      const V3 Qd = dipole(PS, X, Y, Z);
      o.total = {(BBX + Qd.x) * FINT + OIMFX * FEXT - Qd.x, (BBY + Qd.y) * FINT + OIMFY * FEXT - Qd.y, (BBZ + Qd.z) * FINT + OIMFZ * FEXT - Qd.z};
    }
  } else { // outside the magnetopause
    const V3 Qd = dipole(PS, X, Y, Z);
    o.total = {OIMFX - Qd.x, OIMFY - Qd.y, OIMFZ - Qd.z};
  }
  return o;
}


// ---------------------------------------------------------------------------------------------------------------
// Region 1 / Region 2 Birkeland currents
// box-harmonic shielding field shared by BIRK_SHL (:1532-1667) and RC_SHIELD (:2376-2512): two sums ("perpendicular"
// and "parallel" symmetry), 3 x 3 harmonics each, every coefficient split into 4 parts (1, X_SC, f(tilt), f(tilt) X_SC)
T04_HD static inline V3 shield_86(const double *A, double PS, double X_SC, double X, double Y, double Z, double FAC_SC) {
  // (the Fortran evaluates the harmonics of BOTH sums inside each of the two M loops and uses half of them; only the
  // needed ones are evaluated here -- same values, same order of summation)
  const double CPS = t_cos(PS), SPS = t_sin(PS), S3PS = 2.0 * CPS;
  const double PST1 = PS * A[84], PST2 = PS * A[85];
  const double ST1 = t_sin(PST1), CT1 = t_cos(PST1), ST2 = t_sin(PST2), CT2 = t_cos(PST2);
  const double X1 = X * CT1 - Z * ST1, Z1 = X * ST1 + Z * CT1, X2 = X * CT2 - Z * ST2, Z2 = X * ST2 + Z * CT2;
  int L = 0;
  V3 Gv = {0.0, 0.0, 0.0};
  for (int M = 1; M <= 2; ++M) {
    const double XM = M == 1 ? X1 : X2, ZM = M == 1 ? Z1 : Z2, CT = M == 1 ? CT1 : CT2, ST = M == 1 ? ST1 : ST2;
    const double tilt = M == 1 ? CPS : S3PS;
    for (int I = 0; I < 3; ++I) {
      const double P = M == 1 ? A[72 + I] : A[78 + I]; // P (perpendicular sum) or Q (parallel sum)
      const double CY = t_cos(Y / P), SY = t_sin(Y / P);
      for (int K = 0; K < 3; ++K) {
        const double R = M == 1 ? A[75 + K] : A[81 + K]; // R or S
        const double SZ = t_sin(ZM / R), CZ = t_cos(ZM / R);
        const double SQ = t_sqrt(1.0 / (P * P) + 1.0 / (R * R));
        const double E = t_exp(XM * SQ);
        double FX, FY, FZ;
        if (M == 1) {
          FX = -SQ * E * CY * SZ * FAC_SC;
          FY = E * SY * SZ / P * FAC_SC;
          FZ = -E * CY * CZ / R * FAC_SC;
        } else {
          FX = -SPS * SQ * E * CY * CZ * FAC_SC;
          FY = SPS / P * E * SY * CZ * FAC_SC;
          FZ = SPS / R * E * CY * SZ * FAC_SC;
        }
        for (int N = 1; N <= 2; ++N)
          for (int NN = 1; NN <= 2; ++NN) {
            double HX = FX, HY = FY, HZ = FZ;
            if (N == 2) {
              HX = HX * tilt;
              HY = HY * tilt;
              HZ = HZ * tilt;
            }
            if (NN == 2) {
              HX = HX * X_SC;
              HY = HY * X_SC;
              HZ = HZ * X_SC;
            }
            const double HXR = HX * CT + HZ * ST, HZR = -HX * ST + HZ * CT;
            Gv.x += HXR * A[L];
            Gv.y += HY * A[L];
            Gv.z += HZR * A[L];
            ++L;
          }
      }
    }
  }
  return Gv;
}

// R_S (:1424-1437) and THETA_S (:1439-1452): the deformed spherical coordinates of the conical current sheets
T04_HD static inline double r_s(const double *A, double R, double THETA) {
  const double R2 = R * R;
  return R + A[1] / R + A[2] * R / t_sqrt(R2 + A[10] * A[10]) + A[3] * R / (R2 + A[11] * A[11]) +
         (A[4] + A[5] / R + A[6] * R / t_sqrt(R2 + A[12] * A[12]) + A[7] * R / (R2 + A[13] * A[13])) * t_cos(THETA) +
         (A[8] * R / t_sqrt(R2 + A[14] * A[14]) + A[9] * R / sq(R2 + A[15] * A[15])) * t_cos(2.0 * THETA);
}
T04_HD static inline double theta_s(const double *A, double R, double THETA) {
  const double R2 = R * R;
  return THETA + (A[16] + A[17] / R + A[18] / R2 + A[19] * R / t_sqrt(R2 + A[26] * A[26])) * t_sin(THETA) +
         (A[20] + A[21] * R / t_sqrt(R2 + A[27] * A[27]) + A[22] * R / (R2 + A[28] * A[28])) * t_sin(2.0 * THETA) +
         (A[23] + A[24] / R + A[25] * R / (R2 + A[29] * A[29])) * t_sin(3.0 * THETA);
}

// FIALCOS (:1454-1530): field of the N-th harmonic of a conical current sheet of half-thickness DT about THETA0
T04_HD static inline void fialcos(double R, double THETA, double PHI, int N, double THETA0, double DT, double &BTHETA, double &BPHI) {
  const double SINTE = t_sin(THETA), RO = R * SINTE, COSTE = t_cos(THETA), SINFI = t_sin(PHI), COSFI = t_cos(PHI);
  const double TG = SINTE / (1.0 + COSTE), CTG = SINTE / (1.0 - COSTE);
  const double TETANP = THETA0 + DT, TETANM = THETA0 - DT;
  double TGP = 0.0, TGM = 0.0, TGM2 = 0.0, TGP2 = 0.0;
  if (!(THETA < TETANM)) {
    TGP = tan(TETANP * 0.5);
    TGM = tan(TETANM * 0.5);
    TGM2 = TGM * TGM;
    TGP2 = TGP * TGP;
  }
  double COSM1 = 1.0, SINM1 = 0.0, TM = 1.0, TGM2M = 1.0, TGP2M = 1.0, BT = 0.0, BP = 0.0;
  for (int M = 1; M <= N; ++M) {
    TM = TM * TG;
    const double CC = COSM1 * COSFI - SINM1 * SINFI, SS = SINM1 * COSFI + COSM1 * SINFI;
    COSM1 = CC;
    SINM1 = SS;
    double T, DTT;
    if (THETA < TETANM) {
      T = TM;
      DTT = 0.5 * M * TM * (TG + CTG);
    } else if (THETA < TETANP) {
      TGM2M = TGM2M * TGM2;
      const double FC = 1.0 / (TGP - TGM), FC1 = 1.0 / (2 * M + 1);
      const double TGM2M1 = TGM2M * TGM, TG21 = 1.0 + TG * TG;
      T = FC * (TM * (TGP - TG) + FC1 * (TM * TG - TGM2M1 / TM));
      DTT = 0.5 * M * FC * TG21 * (TM / TG * (TGP - TG) - FC1 * (TM - TGM2M1 / (TM * TG)));
    } else {
      TGP2M = TGP2M * TGP2;
      TGM2M = TGM2M * TGM2;
      const double FC = 1.0 / (TGP - TGM), FC1 = 1.0 / (2 * M + 1);
      T = FC * FC1 * (TGP2M * TGP - TGM2M * TGM) / TM;
      DTT = -T * M * 0.5 * (TG + CTG);
    }
    BT = M * T * CC / RO;
    BP = -DTT * SS / R;
  }
  BTHETA = BT * 800.0;
  BPHI = BP * 800.0;
}

// ONE_CONE (:1361-1422): one cone of field-aligned current, by numerical differentiation of the deformed coordinates
T04_HD static inline V3 one_cone(const double *A, int MODE, double DTHETA, double X, double Y, double Z) {
  const double DR = T04D_ONE_CONE_DR[0], DT = T04D_ONE_CONE_DR[1];
  const double THETA0 = A[30];
  const double RHO2 = X * X + Y * Y, RHO = t_sqrt(RHO2), R = t_sqrt(RHO2 + Z * Z);
  const double THETA = atan2(RHO, Z), PHI = atan2(Y, X);
  const double RS = r_s(A, R, THETA), THETAS = theta_s(A, R, THETA);
  double BTAST, BFAST;
  fialcos(RS, THETAS, PHI, MODE, THETA0, DTHETA, BTAST, BFAST);
  const double DRSDR = (r_s(A, R + DR, THETA) - r_s(A, R - DR, THETA)) / (2.0 * DR);
  const double DRSDT = (r_s(A, R, THETA + DT) - r_s(A, R, THETA - DT)) / (2.0 * DT);
  const double DTSDR = (theta_s(A, R + DR, THETA) - theta_s(A, R - DR, THETA)) / (2.0 * DR);
  const double DTSDT = (theta_s(A, R, THETA + DT) - theta_s(A, R, THETA - DT)) / (2.0 * DT);
  const double STSST = t_sin(THETAS) / t_sin(THETA), RSR = RS / R;
  const double BR = -RSR / R * STSST * BTAST * DRSDT, BTHETA = RSR * STSST * BTAST * DRSDR,
               BPHI = RSR * BFAST * (DRSDR * DTSDT - DRSDT * DTSDR);
  const double S = RHO / R, C = Z / R, SF = Y / RHO, CF = X / RHO;
  const double BE = BR * S + BTHETA * C;
  return V3{A[0] * (BE * CF - BPHI * SF), A[0] * (BE * SF + BPHI * CF), A[0] * (BR * C - BTHETA * S)};
}

// TWOCONES (:1341-1359): northern + southern cone
T04_HD static inline V3 twocones(const double *A, int MODE, double DTHETA, double X, double Y, double Z) {
  const V3 N = one_cone(A, MODE, DTHETA, X, Y, Z), S = one_cone(A, MODE, DTHETA, X, -Y, -Z);
  return V3{N.x - S.x, N.y + S.y, N.z + S.z};
}

// BIRK_1N2 (:1211-1339): Region NUMB (1, 2), mode MODE (1, 2), with the day-night asymmetry and tilt deformation
T04_HD static inline V3 birk_1n2(int NUMB, int MODE, double XKAPPA, double PS, double X, double Y, double Z) {
  const double BETA = T04D_BIRK_1N2_BETA[0], RH = T04D_BIRK_1N2_BETA[1], EPS = T04D_BIRK_1N2_BETA[2];
  const double B = 0.5, RHO_0 = 7.0;
  const double DPHI = NUMB == 1 ? 0.055 : 0.030, DTHETA = NUMB == 1 ? 0.06 : 0.09;
  const double Xsc = X * XKAPPA, Ysc = Y * XKAPPA, Zsc = Z * XKAPPA;
  const double RHO = t_sqrt(Xsc * Xsc + Zsc * Zsc), Rsc = t_sqrt(Xsc * Xsc + Ysc * Ysc + Zsc * Zsc);
  const double RHO2 = RHO_0 * RHO_0;
  const double PHI = (Xsc == 0.0 && Zsc == 0.0) ? 0.0 : atan2(-Zsc, Xsc);
  const double SPHIC = t_sin(PHI), CPHIC = t_cos(PHI);
  const double BRACK = DPHI + B * RHO2 / (RHO2 + 1.0) * (RHO * RHO - 1.0) / (RHO2 + RHO * RHO);
  const double R1RH = (Rsc - 1.0) / RH;
  const double PW = t_pow(R1RH, EPS);
  const double PSIAS = BETA * PS / t_pow(1.0 + PW, 1.0 / EPS);
  const double PHIS = PHI - BRACK * t_sin(PHI) - PSIAS;
  const double DPHISPHI = 1.0 - BRACK * t_cos(PHI);
  const double DEN = RH * Rsc * t_pow(1.0 + PW, 1.0 / EPS + 1.0);
  const double DPHISRHO = -2.0 * B * RHO2 * RHO / sq(RHO2 + RHO * RHO) * t_sin(PHI) + BETA * PS * t_pow(R1RH, EPS - 1.0) * RHO / DEN;
  const double DPHISDY = BETA * PS * t_pow(R1RH, EPS - 1.0) * Ysc / DEN;
  const double SPHICS = t_sin(PHIS), CPHICS = t_cos(PHIS);
  const double XS = RHO * CPHICS, ZS = -RHO * SPHICS;
  const double *A = NUMB == 1 ? (MODE == 1 ? T04D_BIRK_1N2_A11 : T04D_BIRK_1N2_A12) : (MODE == 1 ? T04D_BIRK_1N2_A21 : T04D_BIRK_1N2_A22);
  const V3 T = twocones(A, MODE, DTHETA, XS, Ysc, ZS);
  const double BRHOAS = T.x * CPHICS - T.z * SPHICS, BPHIAS = -T.x * SPHICS - T.z * CPHICS;
  const double BRHO_S = BRHOAS * DPHISPHI * XKAPPA;
  const double BPHI_S = (BPHIAS - RHO * (T.y * DPHISDY + BRHOAS * DPHISRHO)) * XKAPPA;
  const double BY_S = T.y * DPHISPHI * XKAPPA;
  return V3{BRHO_S * CPHIC - BPHI_S * SPHIC, BY_S, -BRHO_S * SPHIC - BPHI_S * CPHIC};
}

// BIRK_TOT (:1078-1209), IOPB = 0
T04_HD static inline BirkOut birk_tot(double XKAPPA1, double XKAPPA2, double PS, double X, double Y, double Z) {
  BirkOut o;
  auto add = [](const V3 &a, const V3 &b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; };
  double X_SC = XKAPPA1 - 1.1;
  o.r11 = add(birk_1n2(1, 1, XKAPPA1, PS, X, Y, Z), shield_86(T04D_BIRK_TOT_SH11, PS, X_SC, X, Y, Z, 1.0));
  o.r12 = add(birk_1n2(1, 2, XKAPPA1, PS, X, Y, Z), shield_86(T04D_BIRK_TOT_SH12, PS, X_SC, X, Y, Z, 1.0));
  X_SC = XKAPPA2 - 1.0;
  o.r21 = add(birk_1n2(2, 1, XKAPPA2, PS, X, Y, Z), shield_86(T04D_BIRK_TOT_SH21, PS, X_SC, X, Y, Z, 1.0));
  o.r22 = add(birk_1n2(2, 2, XKAPPA2, PS, X, Y, Z), shield_86(T04D_BIRK_TOT_SH22, PS, X_SC, X, Y, Z, 1.0));
  return o;
}


// ---------------------------------------------------------------------------------------------------------------
// Ring current: symmetric (SRC) and partial (PRC)
// vector potential of two circular current loops at the deformed position (RS, SINTS, COSTS): the complete elliptic
// integrals by the polynomial fits the Fortran spells out twice in AP and twice in APPRC (:1956-1980, :2130-2160);
// three of ELK's constants are default-REAL literals there
T04_HD static inline double loop_aphi(double RRC, double DD, double RHOS, double ZS) {
  const double P = sq(RRC + RHOS) + ZS * ZS + DD * DD;
  const double XK2 = 4.0 * RRC * RHOS / P;
  const double XK = t_sqrt(XK2);
  const double XKRHO12 = XK * t_sqrt(RHOS);
  const double XK2S = 1.0 - XK2;
  const double DL = t_log(1.0 / XK2S);
  const double ELK = 1.38629436112 + XK2S * (0.09666344259 + XK2S * ((double)0.03590092383f + XK2S * ((double)0.03742563713f + XK2S * (double)0.01451196212f))) +
                     DL * (0.5 + XK2S * (0.12498593597 + XK2S * (0.06880248576 + XK2S * (0.03328355346 + XK2S * 0.00441787012))));
  const double ELE = 1.0 + XK2S * (0.44325141463 + XK2S * (0.0626060122 + XK2S * (0.04757383546 + XK2S * 0.01736506451))) +
                     DL * XK2S * (0.2499836831 + XK2S * (0.09200180037 + XK2S * (0.04069697526 + XK2S * 0.00526449639)));
  return ((1.0 - XK2 * 0.5) * ELK - ELE) / XKRHO12;
}
// (ALPHA_S, GAMMA_S) -> (RHOS, ZS): the closed-form inversion shared by AP and APPRC (:1946-1955, :2120-2129)
T04_HD static inline void deformed_to_rz(double ALPHA_S, double GAMMA_S, double &RHOS, double &ZS) {
  const double GAMMAS2 = GAMMA_S * GAMMA_S, ALSQH = ALPHA_S * ALPHA_S / 2.0;
  const double F = 64.0 / 27.0 * GAMMAS2 + ALSQH * ALSQH;
  const double Q = t_pow(t_sqrt(F) + ALSQH, 1.0 / 3.0);
  const double G13 = t_pow(GAMMAS2, 1.0 / 3.0);
  double C = Q - 4.0 * G13 / (3.0 * Q);
  if (C < 0.0) C = 0.0;
  const double G = t_sqrt(C * C + 4.0 * G13);
  const double RS = 4.0 / ((t_sqrt(2.0 * G - C) + t_sqrt(C)) * (G + C));
  const double COSTS = GAMMA_S * RS * RS, SINTS = t_sqrt(1.0 - COSTS * COSTS);
  RHOS = RS * SINTS;
  ZS = RS * COSTS;
}
T04_HD static inline double exp_guard(double arg) { return arg < -500.0 ? 0.0 : t_exp(arg); }

// AP (:1891-2006): azimuthal vector potential of the symmetric ring current
T04_HD static inline double ap(double R, double SINT, double COST) {
  const double *D = T04D_AP_A1;
  const double A1 = D[0], A2 = D[1], RRC1 = D[2], DD1 = D[3], RRC2 = D[4], DD2 = D[5], P1 = D[6], R1 = D[7], DR1 = D[8], DLA1 = D[9],
               P2 = D[10], R2 = D[11], DR2 = D[12], DLA2 = D[13], P3 = D[14], R3 = D[15], DR3 = D[16];
  bool PROX = false;
  double SINT1 = SINT, COST1 = COST;
  if (SINT1 < 1.0e-2) {
    SINT1 = 1.0e-2;
    COST1 = (double).99994999875f;
    PROX = true;
  }
  const double ALPHA = SINT1 * SINT1 / R, GAMMA = COST1 / (R * R);
  const double DEXP1 = exp_guard(-sq((R - R1) / DR1) - sq(COST1 / DLA1));
  const double DEXP2 = exp_guard(-sq((R - R2) / DR2) - sq(COST1 / DLA2));
  const double DEXP3 = exp_guard(-sq((R - R3) / DR3));
  const double ALPHA_S = ALPHA * (1.0 + P1 * DEXP1 + P2 * DEXP2 + P3 * DEXP3);
  double RHOS, ZS;
  deformed_to_rz(ALPHA_S, GAMMA, RHOS, ZS);
  double v = A1 * loop_aphi(RRC1, DD1, RHOS, ZS) + A2 * loop_aphi(RRC2, DD2, RHOS, ZS);
  if (PROX) v = v * SINT / SINT1;
  return v;
}

// APPRC (:2054-2171): the same for the axially symmetric part of the partial ring current
T04_HD static inline double apprc(double R, double SINT, double COST) {
  const double *D = T04D_APPRC_A1;
  const double A1 = D[0], A2 = D[1], RRC1 = D[2], DD1 = D[3], RRC2 = D[4], DD2 = D[5], P1 = D[6], ALPHA1 = D[7], DAL1 = D[8], BETA1 = D[9],
               DG1 = D[10], P2 = D[11], ALPHA2 = D[12], DAL2 = D[13], BETA2 = D[14], DG2 = D[15], BETA3 = D[16], P3 = D[17], ALPHA3 = D[18],
               DAL3 = D[19], BETA4 = D[20], DG3 = D[21], BETA5 = D[22], Q0 = D[23], Q1 = D[24], ALPHA4 = D[25], DAL4 = D[26], DG4 = D[27],
               Q2 = D[28], ALPHA5 = D[29], DAL5 = D[30], DG5 = D[31], BETA6 = D[32], BETA7 = D[33];
  bool PROX = false;
  double SINT1 = SINT, COST1 = COST;
  if (SINT1 < 1.0e-2) {
    SINT1 = 1.0e-2;
    COST1 = (double).99994999875f;
    PROX = true;
  }
  const double ALPHA = SINT1 * SINT1 / R, GAMMA = COST1 / (R * R);
  const double DEXP1 = exp_guard(-sq(GAMMA / DG1));
  const double DEXP2 = exp_guard(-sq((ALPHA - ALPHA4) / DAL4) - sq(GAMMA / DG4));
  const double ALPHA_S =
      ALPHA * (1.0 + P1 / t_pow(1.0 + sq((ALPHA - ALPHA1) / DAL1), BETA1) * DEXP1 +
               P2 * (ALPHA - ALPHA2) / t_pow(1.0 + sq((ALPHA - ALPHA2) / DAL2), BETA2) / t_pow(1.0 + sq(GAMMA / DG2), BETA3) +
               P3 * sq(ALPHA - ALPHA3) / t_pow(1.0 + sq((ALPHA - ALPHA3) / DAL3), BETA4) / t_pow(1.0 + sq(GAMMA / DG3), BETA5));
  const double GAMMA_S = GAMMA * (1.0 + Q0 + Q1 * (ALPHA - ALPHA4) * DEXP2 +
                                  Q2 * (ALPHA - ALPHA5) / t_pow(1.0 + sq((ALPHA - ALPHA5) / DAL5), BETA6) / t_pow(1.0 + sq(GAMMA / DG5), BETA7));
  double RHOS, ZS;
  deformed_to_rz(ALPHA_S, GAMMA_S, RHOS, ZS);
  double v = A1 * loop_aphi(RRC1, DD1, RHOS, ZS) + A2 * loop_aphi(RRC2, DD2, RHOS, ZS);
  if (PROX) v = v * SINT / SINT1;
  return v;
}

// RC_SYMM (:1846-1889) and PRC_SYMM (:2008-2052): B = curl (A_phi e_phi) by numerical differentiation
template <class APF>
T04_HD static inline V3 curl_aphi(APF AP, double X, double Y, double Z) {
  const double DS = 1.0e-2, DC = 0.99994999875, D = 1.0e-4, DRD = 5.0e3;
  const double RHO2 = X * X + Y * Y, R2 = RHO2 + Z * Z, R = t_sqrt(R2);
  const double RP = R + D, RM = R - D;
  const double SINT = t_sqrt(RHO2) / R, COST = Z / R;
  if (SINT < DS) { // too close to the z axis: A_phi ~ t_sin(theta)
    const double A = AP(R, DS, DC) / DS;
    const double DARDR = (RP * AP(RP, DS, DC) - RM * AP(RM, DS, DC)) * DRD;
    const double FXY = Z * (2.0 * A - DARDR) / (R * R2);
    return V3{FXY * X, FXY * Y, (2.0 * A * COST * COST + DARDR * SINT * SINT) / R};
  }
  const double THETA = atan2(SINT, COST), TP = THETA + D, TM = THETA - D;
  const double SINTP = t_sin(TP), SINTM = t_sin(TM), COSTP = t_cos(TP), COSTM = t_cos(TM);
  const double BR = (SINTP * AP(R, SINTP, COSTP) - SINTM * AP(R, SINTM, COSTM)) / (R * SINT) * DRD;
  const double BT = (RM * AP(RM, SINT, COST) - RP * AP(RP, SINT, COST)) / R * DRD;
  const double FXY = (BR + BT * COST / SINT) / R;
  return V3{FXY * X, FXY * Y, BR * COST - BT * SINT};
}

// FFS (:2361-2374)
T04_HD static inline void ffs(double A, double A0, double DA, double &F, double &FA, double &FS) {
  const double SQ1 = t_sqrt(sq(A + A0) + DA * DA), SQ2 = t_sqrt(sq(A - A0) + DA * DA);
  FA = 2.0 / (SQ1 + SQ2);
  F = FA * A;
  FS = 0.5 * (SQ1 + SQ2) / (SQ1 * SQ2) * (1.0 - F * F);
}

// BR_PRC_Q (:2230-2298), BT_PRC_Q (:2300-2359): radial / polar field of the quadrupole part of the partial ring current
T04_HD static inline double br_prc_q(double R, double SINT, double COST) {
  const double *P = T04D_BR_PRC_Q_A1;
  const double XK1 = P[18], AL1 = P[19], DAL1 = P[20], B1 = P[21], BE1 = P[22], XK2 = P[23], AL2 = P[24], DAL2 = P[25], B2 = P[26], BE2 = P[27],
               XK3 = P[28], XK4 = P[29], AL3 = P[30], DAL3 = P[31], B3 = P[32], BE3 = P[33], AL4 = P[34], DAL4 = P[35], DG1 = P[36], AL5 = P[37],
               DAL5 = P[38], DG2 = P[39], C1 = P[40], C2 = P[41], C3 = P[42], AL6 = P[43], DAL6 = P[44], DRM = P[45];
  const double SINT2 = SINT * SINT, COST2 = COST * COST, SC = SINT * COST;
  const double ALPHA = SINT2 / R, GAMMA = COST / (R * R);
  double F, FA, FS, Dv[18];
  ffs(ALPHA, AL1, DAL1, F, FA, FS);
  Dv[0] = SC * t_pow(F, XK1) / (t_pow(R / B1, BE1) + 1.0);
  Dv[1] = Dv[0] * COST2;
  ffs(ALPHA, AL2, DAL2, F, FA, FS);
  Dv[2] = SC * t_pow(FS, XK2) / (t_pow(R / B2, BE2) + 1.0);
  Dv[3] = Dv[2] * COST2;
  ffs(ALPHA, AL3, DAL3, F, FA, FS);
  Dv[4] = SC * t_pow(ALPHA, XK3) * t_pow(FS, XK4) / (t_pow(R / B3, BE3) + 1.0);
  Dv[5] = Dv[4] * COST2;
  double ARGA = sq((ALPHA - AL4) / DAL4) + 1.0, ARGG = 1.0 + sq(GAMMA / DG1);
  Dv[6] = SC / ARGA / ARGG;
  Dv[7] = Dv[6] / ARGA;
  Dv[8] = Dv[7] / ARGA;
  Dv[9] = Dv[8] / ARGA;
  ARGA = sq((ALPHA - AL5) / DAL5) + 1.0;
  ARGG = 1.0 + sq(GAMMA / DG2);
  Dv[10] = SC / ARGA / ARGG;
  Dv[11] = Dv[10] / ARGA;
  Dv[12] = Dv[11] / ARGA;
  Dv[13] = Dv[12] / ARGA;
  const double R4 = sq(R * R);
  Dv[14] = SC / (R4 + sq(C1 * C1));
  Dv[15] = SC / (R4 + sq(C2 * C2)) * COST2;
  Dv[16] = SC / (R4 + sq(C3 * C3)) * (COST2 * COST2);
  ffs(ALPHA, AL6, DAL6, F, FA, FS);
  Dv[17] = SC * FS / (1.0 + sq((R - 1.2) / DRM));
  double v = P[0] * Dv[0];
  for (int i = 1; i < 18; ++i) v += P[i] * Dv[i];
  return v;
}
T04_HD static inline double bt_prc_q(double R, double SINT, double COST) {
  const double *P = T04D_BT_PRC_Q_A1;
  const double XK1 = P[17], AL1 = P[18], DAL1 = P[19], B1 = P[20], BE1 = P[21], XK2 = P[22], AL2 = P[23], DAL2 = P[24], BE2 = P[25], XK3 = P[26],
               XK4 = P[27], AL3 = P[28], DAL3 = P[29], B3 = P[30], BE3 = P[31], AL4 = P[32], DAL4 = P[33], DG1 = P[34], AL5 = P[35], DAL5 = P[36],
               DG2 = P[37], C1 = P[38], C2 = P[39], C3 = P[40];
  const double SINT2 = SINT * SINT, COST2 = COST * COST;
  const double ALPHA = SINT2 / R, GAMMA = COST / (R * R);
  double F, FA, FS, Dv[17];
  ffs(ALPHA, AL1, DAL1, F, FA, FS);
  Dv[0] = t_pow(F, XK1) / (t_pow(R / B1, BE1) + 1.0);
  Dv[1] = Dv[0] * COST2;
  ffs(ALPHA, AL2, DAL2, F, FA, FS);
  Dv[2] = t_pow(FA, XK2) / t_pow(R, BE2);
  Dv[3] = Dv[2] * COST2;
  ffs(ALPHA, AL3, DAL3, F, FA, FS);
  Dv[4] = t_pow(FS, XK3) * t_pow(ALPHA, XK4) / (t_pow(R / B3, BE3) + 1.0);
  Dv[5] = Dv[4] * COST2;
  ffs(GAMMA, 0.0, DG1, F, FA, FS);
  const double FCC = 1.0 + sq((ALPHA - AL4) / DAL4);
  Dv[6] = 1.0 / FCC * FS;
  Dv[7] = Dv[6] / FCC;
  Dv[8] = Dv[7] / FCC;
  Dv[9] = Dv[8] / FCC;
  const double ARG = 1.0 + sq((ALPHA - AL5) / DAL5);
  Dv[10] = 1.0 / ARG / (1.0 + sq(GAMMA / DG2));
  Dv[11] = Dv[10] / ARG;
  Dv[12] = Dv[11] / ARG;
  Dv[13] = Dv[12] / ARG;
  const double R4 = sq(R * R);
  Dv[14] = 1.0 / (R4 + C1 * C1);
  Dv[15] = COST2 / (R4 + C2 * C2);
  Dv[16] = COST2 * COST2 / (R4 + C3 * C3);
  double v = P[0] * Dv[0];
  for (int i = 1; i < 17; ++i) v += P[i] * Dv[i];
  return v;
}

// PRC_QUAD (:2173-2228): field of the quadrupole part of the partial ring current
T04_HD static inline V3 prc_quad(double X, double Y, double Z) {
  const double D = 1.0e-4, DD = 2.0e-4, DS = 1.0e-2, DC = 0.99994999875;
  const double RHO2 = X * X + Y * Y, R = t_sqrt(RHO2 + Z * Z), RHO = t_sqrt(RHO2);
  const double SINT = RHO / R, COST = Z / R, RP = R + D, RM = R - D;
  if (SINT > DS) {
    const double CPHI = X / RHO, SPHI = Y / RHO;
    const double BR = br_prc_q(R, SINT, COST), BT = bt_prc_q(R, SINT, COST);
    const double DBRR = (br_prc_q(RP, SINT, COST) - br_prc_q(RM, SINT, COST)) / DD;
    const double THETA = atan2(SINT, COST), TP = THETA + D, TM = THETA - D;
    const double DBTT = (bt_prc_q(R, t_sin(TP), t_cos(TP)) - bt_prc_q(R, t_sin(TM), t_cos(TM))) / DD;
    return V3{SINT * (BR + (BR + R * DBRR + DBTT) * SPHI * SPHI) + COST * BT, -SINT * SPHI * CPHI * (BR + R * DBRR + DBTT),
              (BR * COST - BT * SINT) * CPHI};
  }
  const double ST = DS;
  double CT = DC;
  if (Z < 0.0) CT = -DC;
  const double THETA = atan2(ST, CT), TP = THETA + D, TM = THETA - D;
  const double BR = br_prc_q(R, ST, CT), BT = bt_prc_q(R, ST, CT);
  const double DBRR = (br_prc_q(RP, ST, CT) - br_prc_q(RM, ST, CT)) / DD;
  const double DBTT = (bt_prc_q(R, t_sin(TP), t_cos(TP)) - bt_prc_q(R, t_sin(TM), t_cos(TM))) / DD;
  const double FCXY = R * DBRR + DBTT;
  return V3{(BR * (X * X + 2.0 * Y * Y) + FCXY * Y * Y) / sq(R * ST) + BT * COST, -(BR + FCXY) * X * Y / sq(R * ST), (BR * COST / ST - BT) * X / R};
}

// SRC_PRC (:1762-1844) + FULL_RC (:1669-1760), IOPR = 0
T04_HD static inline void full_rc(double SC_SY, double SC_PR, double PHI, double PS, double X, double Y, double Z, V3 &SRC, V3 &PRC) {
  const double CPS = t_cos(PS), SPS = t_sin(PS);
  const double XT = X * CPS - Z * SPS, ZT = Z * CPS + X * SPS;
  const double XTS = XT / SC_SY, YTS = Y / SC_SY, ZTS = ZT / SC_SY, XTA = XT / SC_PR, YTA = Y / SC_PR, ZTA = ZT / SC_PR;
  const V3 BS = curl_aphi([](double r, double s, double c) { return ap(r, s, c); }, XTS, YTS, ZTS);
  const V3 BA = curl_aphi([](double r, double s, double c) { return apprc(r, s, c); }, XTA, YTA, ZTA);
  const double CP = t_cos(PHI), SP = t_sin(PHI);
  const double XR = XTA * CP - YTA * SP, YR = XTA * SP + YTA * CP;
  const V3 BQ = prc_quad(XR, YR, ZTA);
  const double BXA_Q = BQ.x * CP + BQ.y * SP, BYA_Q = -BQ.x * SP + BQ.y * CP;
  const double BXP = BA.x + BXA_Q, BYP = BA.y + BYA_Q, BZP = BA.z + BQ.z;
  const V3 HS = {BS.x * CPS + BS.z * SPS, BS.y, BS.z * CPS - BS.x * SPS};
  const V3 HP = {BXP * CPS + BZP * SPS, BYP, BZP * CPS - BXP * SPS};
  double X_SC = SC_SY - 1.0;
  const V3 FS = shield_86(T04D_FULL_RC_C_SY, PS, X_SC, X, Y, Z, (X_SC + 1.0) * (X_SC + 1.0) * (X_SC + 1.0));
  X_SC = SC_PR - 1.0;
  const V3 FP = shield_86(T04D_FULL_RC_C_PR, PS, X_SC, X, Y, Z, (X_SC + 1.0) * (X_SC + 1.0) * (X_SC + 1.0));
  SRC = V3{HS.x + FS.x, HS.y + FS.y, HS.z + FS.z};
  PRC = V3{HP.x + FP.x, HP.y + FP.y, HP.z + FP.z};
}

// T04_s (:5-116): REAL interface over the REAL*8 model
T04_HD static inline void t04_s(const float *PARMOD, float PS, float X, float Y, float Z, float &BX, float &BY, float &BZ) {
  const double PDYN = PARMOD[0];
  const double DST_AST = (double)(PARMOD[1] * 0.8f) - (double)13.f * t_sqrt(PDYN);
  const Components c = external_field(T04D_T04_S_A, PDYN, DST_AST, PARMOD[2], PARMOD[3], PARMOD[4], PARMOD[5], PARMOD[6], PARMOD[7],
                                      PARMOD[8], PARMOD[9], PS, X, Y, Z);
  BX = (float)c.total.x;
  BY = (float)c.total.y;
  BZ = (float)c.total.z;
}


} // namespace t04
} // namespace srt
