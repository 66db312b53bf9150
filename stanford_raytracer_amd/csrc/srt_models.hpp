// srt_models.hpp -- device-side density models (the reference's funcPlasmaParams adapters).
//
// Every model exposes
//     template <int NP> __device__ void density(const double (&p)[NP][3], double (&Ns)[NP][4], double *lds)
// which must be called by ALL 64 lanes of the wave together (the interp model stages coefficient
// tiles through LDS cooperatively).  Blocks are exactly one wave (64 threads).
#pragma once
#include "srt_device.hpp"

namespace srt {

constexpr int WAVE = 64;
// LDS tile of the interp model: a ring of RING units; one unit = one k-plane (16 coefficients = 128 B) of one
// species for all 64 lanes = 8 KiB.  TILE_PAD_BYTES of head room in front: the immediate offset of an LDS-DMA load
// shifts the LDS address as well as the global one (probed: tools/probes/dma_probe.hip), so destinations are
// biased by -imm.  34 KiB per wave, 4 waves per CU.
constexpr int UNIT_BYTES = WAVE * 128;
constexpr int RING = 4;
constexpr int TILE_PAD_BYTES = 2048;
constexpr int TILE_DOUBLES = (RING * UNIT_BYTES + TILE_PAD_BYTES) / 8;

typedef double d2_t __attribute__((ext_vector_type(2)));
#ifdef SRT_TRIP_TIMING
__device__ unsigned long long srt_trip_cycles[16]; // see srt_kernels.hpp
// per-wave sums live in LDS and are flushed once at the end of the kernel: a global atomic per mark would itself show up in
// whatever waits on memory next (every wave adding to the same 16 words)
__device__ __forceinline__ unsigned long long *srt_tt_lds() {
  __shared__ unsigned long long acc[16];
  return acc;
}
#endif

#ifdef SRT_PHASE_TIMING
// (timing builds) how the persistent waves of a trace launch ran: [0] waves that claimed at least one ray, [1] sum of their
// spans in timer ticks, [2] earliest start, [3] latest end -- [3] - [2] over the launch's HIP-event time is the timer's rate,
// [1] / ([0] ([3] - [2])) the share of the launch a working wave was resident for
__device__ unsigned long long srt_wave_stats[4] = {0ull, 0ull, ~0ull, 0ull};
#endif
// Per-launch device scratch of a model (only the scattered model has any: srt_scattered.hpp overloads this).
template <class M>
__device__ __forceinline__ void bind_scratch(const M &, double *, double *, double * = nullptr) {}
// A lane of the trace kernel has just been given a new ray (the scattered model forgets the lane's candidate block).
template <class M>
__device__ __forceinline__ void new_ray_hook(const M &, double *, bool) {}
// What the cooperative kernels (trace, gradients, RK step) reserve per wave for a model: LDS (in doubles) and the waves per
// SIMD the kernel is compiled for (1: all 512 registers of a lane; the scattered model specialises this: srt_scattered.hpp).
template <class M>
struct WaveBudget {
  static constexpr int LDS_DOUBLES = TILE_DOUBLES;
  static constexpr int WAVES_PER_EU = 1;
};
#define SRT_AS1 __attribute__((address_space(1)))
#define SRT_AS3 __attribute__((address_space(3)))

// =============================================================================================
// modelnum = 1: Ngo diffusive-equilibrium model.
// ngo_dens_model_adapter.f95:105-140 (adapter head) + ngo_dens_model.f95:165-353 (dens).
// Trig-free restatement: with phi the colatitude the adapter computes L = r/(R_E sin^2 phi),
// lam = 90 - phi (deg) and z(1) = r0*L*cos^2(lam) = r0*r/R_E, z(2) = phi; dens() needs only z(1),
// sin^2 z(2) = rho^2/r^2 and the sign of the latitude.
struct NgoModel {
  // state read by dens(): see ngo_dens_model.f95:8-24 and readinput :29-160
  double r0, pi32;
  double therm, rbase, ane0, alpha0[5], rzero, scbot;
  double lk, expk, ddk, rconsn, scr;
  double l0[10], def[10], dd[10], rducln[10], rducun[10], rducls[10], rducus[10], sidedu[10];
  double hl2n[10], hl2s[10], hu2n[10], hu2s[10];
  int num, kducts, kinit;

  // noinline: ONE compiled body, so that a point gets the same arithmetic whichever path (and lane) evaluates it
  __device__ __noinline__ void dens_point(double x, double y, double z, double Ns[4]) const {
    double rho2 = x * x + y * y;
    double r2 = rho2 + z * z;
    double r = sqrt(r2);
    // z(1) = r0*L*cos^2(lam) = r0*r/R_E ; sin^2 z(2) = rho^2/r^2 ; only the sign of the latitude is used
    dens_core(r0 * r / R_E, rho2 / r2, z, Ns);
  }

  // dens (ngo_dens_model.f95:165-353) entered with z(1), sin^2 z(2) and the latitude
  __device__ inline void dens_core(double z1, double sinz22, double latitu, double Ns[4]) const {
    // scale heights (:180-186); 1.150600 is a default-real literal (SURVEY A-6)
    double rb7370 = rbase / 7370.0;
    double sh2 = (double)1.150600f * therm * rb7370 * rb7370;
    double gph = rbase * (1.0 - rbase / z1);
    double e2 = exp(-gph / sh2);
    double e3 = e2 * e2 * e2 * e2;
    double e4 = e3 * e3 * e3 * e3;
    double q2 = alpha0[2] * e2, q3 = (num >= 3) ? alpha0[3] * e3 : 0.0, q4 = (num >= 4) ? alpha0[4] * e4 : 0.0;
    double q = q2 + q3 + q4;
    double anr = sqrt(q);
    double arg = (z1 - rzero) / scbot;
    if (!(arg < 13.0)) arg = 13.0;
    double anli = 1.0 - exp(-arg * arg);
    double l = z1 / (r0 * sinz22);
    double ani1 = ane0 * anr * anli;
    if (kducts != 0) {
      double deltal = l - lk;
      if (!(deltal < 0.0)) { // plasmapause (:218-239)
        double d2 = ddk * ddk;
        double argl = deltal * deltal / (d2 * 2.0);
        if (!(argl < 80.0)) argl = 80.0;
        double f = exp(-argl);
        double trm = pow(rconsn / z1, expk);
        double argr = (z1 - rconsn) / scr;
        if (!(argr < 12.5)) argr = 12.5;
        double fr = exp(-argr * argr);
        double trmodl = trm + (1.0 - trm) * fr;
        ani1 *= f + trmodl * (1.0 - f);
      }
      if (kducts != 1) ani1 *= ducts(l, z1, latitu);
    }
    double invq = 1.0 / q;
    Ns[0] = 1.0e6 * ani1;
    Ns[1] = 1.0e6 * (ani1 * (q2 * invq));
    Ns[2] = 1.0e6 * (ani1 * (q3 * invq));
    Ns[3] = 1.0e6 * (ani1 * (q4 * invq));
  }

  // altitude taper shared by the sinusoidal perturbation and the ducts (:250-276, :299-327).
  // Returns false when the contribution is skipped entirely (arglr >= 75).
  __device__ inline bool taper(int kd, double z1, double latitu, bool north_first, double &delnl) const {
    bool lower;
    if (north_first)
      lower = (latitu >= 0 && z1 <= rducun[kd]) || (latitu <= 0 && z1 <= rducus[kd]);
    else
      lower = (latitu <= 0 && z1 <= rducus[kd]) || (latitu >= 0 && z1 <= rducun[kd]);
    double delr = 0.0, arglr = 0.0;
    if (!lower) { // above the upper edge
      if (north_first) {
        if (latitu >= 0) delr = z1 - rducun[kd];
        if (latitu <= 0) delr = z1 - rducus[kd];
        if (latitu >= 0) arglr = delr * delr / hu2n[kd];
        if (latitu <= 0) arglr = delr * delr / hu2s[kd];
      } else {
        if (latitu >= 0) delr = z1 - rducun[kd];
        if (latitu <= 0) delr = z1 - rducus[kd];
        if (latitu <= 0) arglr = delr * delr / hu2s[kd];
        if (latitu >= 0) arglr = delr * delr / hu2n[kd];
      }
      if (arglr >= 75.0) return false;
      delnl *= exp(-arglr);
      return true;
    }
    bool inside;
    if (north_first)
      inside = (latitu >= 0 && z1 >= rducln[kd]) || (latitu <= 0 && z1 >= rducls[kd]);
    else
      inside = (latitu <= 0 && z1 >= rducls[kd]) || (latitu >= 0 && z1 >= rducln[kd]);
    if (inside) return true;
    if (latitu >= 0) delr = z1 - rducln[kd];
    if (latitu <= 0) delr = z1 - rducls[kd];
    if (north_first) {
      if (latitu >= 0) arglr = delr * delr / hl2n[kd];
      if (latitu <= 0) arglr = delr * delr / hl2s[kd];
    } else {
      if (latitu <= 0) arglr = delr * delr / hl2s[kd];
      if (latitu >= 0) arglr = delr * delr / hl2n[kd];
    }
    if (arglr >= 75.0) return false;
    delnl *= exp(-arglr);
    return true;
  }

  __device__ __noinline__ double ducts(double l, double z1, double latitu) const {
    double fac = 1.0;
    int kstart = kinit;
    if (!(l0[2] > 0.0)) { // sinusoidal density perturbation (:241-288)
      kstart = 3;
      double dl = l + l0[2];
      if (!(dl * sidedu[2] >= 0.0)) dl = 0.0;
      double delk = -l0[2] - (lk + ddk) + dd[2] / 2;
      double critl = (lk + ddk) + fmod(delk, dd[2]);
      if (!(l <= critl)) {
        double argl = 2.0 * pi32 * dl / dd[2];
        double delnl = (def[2] / 2.0) * (1.0 + cos(argl));
        if (taper(2, z1, latitu, false, delnl)) fac *= 1.0 + delnl;
      }
      if (kducts == 2) return fac;
    }
    for (int kd = kstart; kd <= kducts; ++kd) { // gaussian ducts (:289-336)
      double dl = l - l0[kd];
      if (!(dl * sidedu[kd] >= 0.0)) dl = 0.0;
      double d2 = dd[kd] * dd[kd];
      double argl = dl * dl / (d2 * 2.0);
      if (argl > 80.0) continue;
      double delnl = def[kd] * exp(-argl);
      if (taper(kd, z1, latitu, true, delnl)) fac *= 1.0 + delnl;
    }
    return fac;
  }

  template <int NP>
  __device__ __forceinline__ void density(const double (&p)[NP][3], double (&Ns)[NP][4], double *) const {
#pragma unroll
    for (int i = 0; i < NP; ++i) dens_point(p[i][0], p[i][1], p[i][2], Ns[i]);
  }
  // Two points per call, ONE compiled body (noinline) for every path of the trace kernel: a single density evaluation is one
  // dependent chain of exp / pow / division steps (measured 7.7 cycles per instruction with one wave per SIMD); two
  // independent chains in one body interleave.  Point i of a stencil always goes through half (i & 1) of the pair, whichever
  // path and lane evaluates it, so a ray's arithmetic does not depend on either.
  struct Dens2 {
    double a[4], b[4];
  };
  __device__ __noinline__ Dens2 dens_pair(double ax, double ay, double az, double bx, double by, double bz) const {
    Dens2 r;
    const double rhoa = ax * ax + ay * ay, ra2 = rhoa + az * az, ra = sqrt(ra2);
    const double rhob = bx * bx + by * by, rb2 = rhob + bz * bz, rb = sqrt(rb2);
    const double z1[2] = {r0 * ra / R_E, r0 * rb / R_E}, s2[2] = {rhoa / ra2, rhob / rb2}, lat[2] = {az, bz};
    double N[2][4];
    dens_core2(z1, s2, lat, N);
#pragma unroll
    for (int s = 0; s < 4; ++s) r.a[s] = N[0][s], r.b[s] = N[1][s];
    return r;
  }
  // dens_core for two points, statement by statement, so that the two dependent chains sit in the same basic blocks and
  // interleave: the plasmapause term (taken when deltal >= 0 in the Fortran, :218-239) is evaluated for both points and
  // selected -- the same values as the branch gives.  The operations and their order per point are those of dens_core.
  __device__ __forceinline__ void dens_core2(const double (&z1)[2], const double (&sinz22)[2], const double (&latitu)[2], double (&Ns)[2][4]) const {
    const double rb7370 = rbase / 7370.0;
    const double sh2 = (double)1.150600f * therm * rb7370 * rb7370;
    double q2[2], q3[2], q4[2], q[2], ani1[2], l[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const double gph = rbase * (1.0 - rbase / z1[p]);
      const double e2 = exp(-gph / sh2);
      const double e3 = e2 * e2 * e2 * e2;
      const double e4 = e3 * e3 * e3 * e3;
      q2[p] = alpha0[2] * e2, q3[p] = (num >= 3) ? alpha0[3] * e3 : 0.0, q4[p] = (num >= 4) ? alpha0[4] * e4 : 0.0;
      q[p] = q2[p] + q3[p] + q4[p];
      const double anr = sqrt(q[p]);
      double arg = (z1[p] - rzero) / scbot;
      if (!(arg < 13.0)) arg = 13.0;
      const double anli = 1.0 - exp(-arg * arg);
      l[p] = z1[p] / (r0 * sinz22[p]);
      ani1[p] = ane0 * anr * anli;
    }
    if (kducts != 0) { // wave-uniform
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const double deltal = l[p] - lk;
        const double d2 = ddk * ddk;
        double argl = deltal * deltal / (d2 * 2.0);
        if (!(argl < 80.0)) argl = 80.0;
        const double f = exp(-argl);
        const double trm = pow(rconsn / z1[p], expk);
        double argr = (z1[p] - rconsn) / scr;
        if (!(argr < 12.5)) argr = 12.5;
        const double fr = exp(-argr * argr);
        const double trmodl = trm + (1.0 - trm) * fr;
        const double with_pp = ani1[p] * (f + trmodl * (1.0 - f));
        ani1[p] = !(deltal < 0.0) ? with_pp : ani1[p]; // plasmapause (:218-239)
      }
      if (kducts != 1) {
#pragma unroll
        for (int p = 0; p < 2; ++p) ani1[p] *= ducts(l[p], z1[p], latitu[p]);
      }
    }
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const double invq = 1.0 / q[p];
      Ns[p][0] = 1.0e6 * ani1[p];
      Ns[p][1] = 1.0e6 * (ani1[p] * (q2[p] * invq));
      Ns[p][2] = 1.0e6 * (ani1[p] * (q3[p] * invq));
      Ns[p][3] = 1.0e6 * (ani1[p] * (q4[p] * invq));
    }
  }
  // point i of the stencil (0 centre, 1 + 2a / 2 + 2a = +- d_a along axis a, 7 = the free point or, without one, the centre again)
  template <int NE>
  __device__ __forceinline__ static void stencil_point(int i, const double oc[3], const double od[3], const double oe[3], double q[3]) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      double v = oc[a];
      v = (i == 1 + 2 * a) ? oc[a] + od[a] : v;
      v = (i == 2 + 2 * a) ? oc[a] - od[a] : v;
      v = (NE && i == 7) ? oe[a] : v;
      q[a] = v;
    }
  }
  // Stencil of one right-hand side: centre, centre +- d_a e_a, optionally one free point.  All 64 lanes call together.
  // Tail mode: a launch ends with a few long rays (config[1]: 27 of 100 k rays run to maxsteps while the mean ray
  // stops after 29 steps), each doing its 7-8 density evaluations one after the other while most lanes idle.  When at most
  // 16 lanes of the wave need a stencil, the k-th needy lane's points are spread over lanes 4k .. 4k+3 -- one dens_pair per lane
  // instead of four in a row -- and handed back through LDS.  Same function, same arguments, another lane: bit-identical
  // results.
  template <int NE>
  __device__ __forceinline__ void density_stencil(const double c[3], const double d[3], const double *extra,
                                                  double (&Ns)[7 + NE][4], double *, bool need = true) const {
    const unsigned long long needy = __ballot(need);
    const int nneedy = __popcll(needy);
    if (nneedy >= 1 && nneedy <= 16) { // wave-uniform
#ifdef SRT_TRIP_TIMING
      __builtin_amdgcn_sched_barrier(0); unsigned long long tt0_ = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0);
#define SRT_TTM(slot_) do { __builtin_amdgcn_sched_barrier(0); unsigned long long t1_ = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); if (threadIdx.x == 0) srt_tt_lds()[slot_] += t1_ - tt0_; tt0_ = t1_; } while (0)
#else
#define SRT_TTM(slot_) do {} while (0)
#endif
      const int lane = (int)__lane_id(), slot = lane >> 2, sub = lane & 3;
      unsigned long long mask = needy;
      for (int k = 0; k < slot && mask; ++k) mask &= mask - 1; // drop the `slot` lowest needy lanes
      const int owner = (slot < nneedy) ? __builtin_ctzll(mask) : lane;
      double oc[3], od[3], oe[3], qa[3], qb[3];
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        oc[a] = __shfl(c[a], owner);
        od[a] = __shfl(d[a], owner);
        oe[a] = NE ? __shfl(extra[a], owner) : 0.0;
      }
      stencil_point<NE>(2 * sub, oc, od, oe, qa);
      stencil_point<NE>(2 * sub + 1, oc, od, oe, qb);
      SRT_TTM(11);
      const Dens2 mine = dens_pair(qa[0], qa[1], qa[2], qb[0], qb[1], qb[2]);
      SRT_TTM(12);
      // hand-off through LDS: every lane parks its two points' densities (64 B), the owners read their 7-8 points back as
      // 128-bit words -- 16 LDS reads; per-lane-indexed cross-lane reads (64 ds_bpermute) measured 7 % slower on the launch,
      // and behind a conditional assignment (one exec-mask branch per value) 15 % slower
      const int rank = __popcll(needy & ((1ull << lane) - 1ull)); // this lane's slot if it is an owner
      __shared__ __attribute__((aligned(16))) double park[WAVE * 8];
      {
        d2_t *mp_ = (d2_t *)(park + 8 * lane);
        mp_[0] = d2_t{mine.a[0], mine.a[1]};
        mp_[1] = d2_t{mine.a[2], mine.a[3]};
        mp_[2] = d2_t{mine.b[0], mine.b[1]};
        mp_[3] = d2_t{mine.b[2], mine.b[3]};
      }
      __syncthreads(); // block == one wave
#pragma unroll
      for (int i = 0; i < 7 + NE; ++i) {
        const d2_t *src = (const d2_t *)(park + 8 * ((4 * rank + (i >> 1)) & 63) + 4 * (i & 1));
        const d2_t u = src[0], v = src[1];
        Ns[i][0] = u.x, Ns[i][1] = u.y, Ns[i][2] = v.x, Ns[i][3] = v.y;
      }
      __syncthreads(); // the next round overwrites the slots
      SRT_TTM(13);
      return;
    }
    const double oe[3] = {NE ? extra[0] : 0.0, NE ? extra[1] : 0.0, NE ? extra[2] : 0.0};
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      double qa[3], qb[3];
      stencil_point<NE>(2 * h, c, d, oe, qa);
      stencil_point<NE>(2 * h + 1, c, d, oe, qb);
      const Dens2 r = dens_pair(qa[0], qa[1], qa[2], qb[0], qb[1], qb[2]);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        Ns[2 * h][s] = r.a[s];
        if (2 * h + 1 < 7 + NE) Ns[2 * h + 1][s] = r.b[s];
      }
    }
  }
};

// =============================================================================================
// modelnum = 3: tricubic interpolation of ln N_s on a regular grid
// (interp_dens_model_adapter.f95:148-210 + libtricubic.f95:796-933).
//
// HBM layout: the 64 Lekien-Marsden polynomial coefficients of every cell are expanded ONCE at model
// creation (srt_build_coeffs kernel = tricubic_get_coeff applied to all cells, including the clamped
// out-of-range cells and their sticky-flag quirk, SURVEY A-7) into
//     coef[cell][species][64]  doubles, cell = (ck*(ny+1)+cj)*(nx+1)+ci, ci = #nodes <= x (0..nx)
// i.e. one contiguous 512 B block per (cell, species), 2 KiB per cell at nspec = 4 -- exactly the
// 2048 B per lookup that the reference gathers from 8 arrays x 8 corners.  A lookup then is one
// contiguous block read plus a polynomial evaluation.  Per wave the blocks of all 64 lanes stream through a
// ring of four 8-KiB LDS buffers, one k-plane (128 B per lane) of one species at a time, by LDS-DMA with a
// lookahead of three units, and every lane evaluates all its stencil points from each unit (see
// InterpModel::density_stencil).
struct Axis {
  double min, del, rdel; // rdel = 1/del: only for the first guess of the cell search
  int n;
  // node i (0-based): real(i)*del + min, separately rounded like interp_dens_model_adapter.f95:93-95
  // (the product must not be fused into the sum: the Fortran rounds it separately)
  __device__ __forceinline__ double node(int i) const {
#pragma clang fp contract(off)
    double prod = (double)i * del;
    return prod + min;
  }
  // maxloc(.., mask = 0 <= (xi - x)) (libtricubic.f95:835-840): number of nodes <= xi, and the local
  // coordinate (:842-856)
  __device__ __forceinline__ int locate(double xi, double &xl) const {
    double f = (xi - min) * rdel;
    int g;
    if (!(f >= 0.0)) g = 0;
    else if (f >= (double)n) g = n;
    else g = (int)f + 1;
    // the quotient is within one node of the exact answer: one correction each way, no loops
    g += (g < n && node(g) <= xi) ? 1 : 0;
    g -= (g > 0 && node(g - 1) > xi) ? 1 : 0;
    xl = (g >= 1 && g < n) ? fdiv(xi - node(g - 1), del) : 0.0;
    return g;
  }
};

struct InterpModel {
  const double *coef; // [ncell][nspec][64]
  Axis ax, ay, az;
  int nspec;

  __device__ __forceinline__ int cell_id(int ci, int cj, int ck) const {
    return (ck * (ay.n + 1) + cj) * (ax.n + 1) + ci;
  }

  // tricubic_eval with derx=dery=derz=0 (libtricubic.f95:658-695) for NP points that share one cell:
  // nested Horner, coefficient-outer / point-inner so each coefficient is read once.
  // chunk(q) returns coefficients (2q, 2q+1).
  template <int NP, class Chunk>
  __device__ __forceinline__ static void eval(Chunk chunk, const double (&x)[NP], const double (&y)[NP],
                                              const double (&z)[NP], double (&out)[NP]) {
    double vz[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) vz[i] = 0.0;
#pragma unroll
    for (int k = 3; k >= 0; --k) {
      double vy[NP];
#pragma unroll
      for (int i = 0; i < NP; ++i) vy[i] = 0.0;
#pragma unroll
      for (int j = 3; j >= 0; --j) {
        const int q = 2 * j + 8 * k;
        double2 lo = chunk(q), hi = chunk(q + 1);
#pragma unroll
        for (int i = 0; i < NP; ++i) {
          double vx = fma(fma(fma(hi.y, x[i], hi.x), x[i], lo.y), x[i], lo.x);
          vy[i] = fma(vy[i], y[i], vx);
        }
      }
#pragma unroll
      for (int i = 0; i < NP; ++i) vz[i] = fma(vz[i], z[i], vy[i]);
    }
#pragma unroll
    for (int i = 0; i < NP; ++i) out[i] = vz[i];
  }

  // rare path: one point, straight from global memory (lane-divergent)
  __device__ __noinline__ void point_direct(double x, double y, double z, double lnN[4]) const {
    double xl, yl, zl;
    int ci = ax.locate(x, xl), cj = ay.locate(y, yl), ck = az.locate(z, zl);
    const double *base = coef + (size_t)cell_id(ci, cj, ck) * (size_t)(nspec * 64);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      if (s < nspec) {
        const double2 *a = reinterpret_cast<const double2 *>(base + s * 64);
        double x1[1] = {xl}, y1[1] = {yl}, z1[1] = {zl}, o1[1];
        eval<1>([&](int q) { return a[q]; }, x1, y1, z1, o1);
        lnN[s] = o1[0];
      }
    }
  }

  // ------------------------------------------------------------------------------------------
  // Staging of coefficient blocks through the LDS ring.
  //
  // Unit (s, k) = coefficients a[16k .. 16k+15] of species s (8 chunks of 16 B = one aligned 128-B line per lane).
  // LDS image of a unit: 64 rows of 128 B; lane j owns row slot 8*(j&7) + (j>>3) and finds logical chunk q at
  // physical chunk (q + j) & 7, so the 16-lane groups of ds_read_b128 ({0-3,12-15,20-27}, ...) touch 16 different
  // 16-B bank slots (row parity = (j>>3)&1 selects the half of the 256-B bank row): conflict-free.
  // Filled by LDS-DMA: instruction t of a unit writes 1 KiB = row slots 8t..8t+7; its lane L therefore carries
  // chunk (L&7) of the row of lane (L&56)+t.  The per-lane source addresses depend only on the cells, so they are
  // built ONCE per density call (two ds_bpermute per instruction) and serve all 4*nspec units: the unit is
  // selected by the 12-bit immediate offset (k*128, +512 to reach into the next species).
  __device__ __forceinline__ void stage_prepare(int cell, unsigned long long (&a)[8]) const {
    const int lane = threadIdx.x;
    const unsigned long long base =
        reinterpret_cast<unsigned long long>(coef) + (unsigned long long)(unsigned)cell * (unsigned long long)(nspec * 512);
    const int blo = (int)(unsigned)base, bhi = (int)(unsigned)(base >> 32);
    const int p = lane & 7, g = lane & 56;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int src = (g + t) << 2;
      const unsigned lo = (unsigned)__builtin_amdgcn_ds_bpermute(src, blo);
      const unsigned hi = (unsigned)__builtin_amdgcn_ds_bpermute(src, bhi);
      a[t] = (((unsigned long long)hi << 32) | (unsigned long long)lo) + (unsigned long long)(((p - t) & 7) << 4);
    }
  }
  // LDS byte addresses of this lane's 8 logical chunks inside ring buffer 0
  __device__ __forceinline__ static void read_addrs(double *lds, unsigned (&ra)[8]) {
    const int lane = threadIdx.x;
    const unsigned base = (unsigned)(unsigned long long)((SRT_AS3 char *)lds) + TILE_PAD_BYTES +
                          (unsigned)((8 * (lane & 7) + (lane >> 3)) * 128);
#pragma unroll
    for (int q = 0; q < 8; ++q) ra[q] = base + (unsigned)(((q + lane) & 7) << 4);
  }
  // DMA of one unit into ring buffer J; IMM = byte offset of the unit relative to the addresses in a[]
  // AUX = cache policy bits of the load.  (Measured: nt on the species >= 1 units, hoping to keep species 0
  // resident in L2, made the kernel 9 % slower -- default policy everywhere.)
  template <int IMM, int J, int AUX = 0>
  __device__ __forceinline__ static void issue_unit(const unsigned long long (&a)[8], double *lds) {
    SRT_AS3 char *ring = (SRT_AS3 char *)lds + TILE_PAD_BYTES;
#pragma unroll
    for (int t = 0; t < 8; ++t)
      __builtin_amdgcn_global_load_lds((const SRT_AS1 void *)a[t], (SRT_AS3 void *)(ring + (J * UNIT_BYTES + t * 1024 - IMM)),
                                       16, IMM, AUX);
  }
  template <int N>
  __device__ __forceinline__ static void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
  }
  // All 8 chunks of a unit from ring buffer J into registers.  Inline asm on purpose: the compiler's wait-count
  // pass makes every LDS load it can see wait for ALL LDS-DMA in flight, which would serialise the ring.
  template <int J>
  __device__ __forceinline__ static void read_unit(const unsigned (&ra)[8], d2_t (&c)[8]) {
    asm volatile("ds_read_b128 %0, %8 offset:%16\n\t"
                 "ds_read_b128 %1, %9 offset:%16\n\t"
                 "ds_read_b128 %2, %10 offset:%16\n\t"
                 "ds_read_b128 %3, %11 offset:%16\n\t"
                 "ds_read_b128 %4, %12 offset:%16\n\t"
                 "ds_read_b128 %5, %13 offset:%16\n\t"
                 "ds_read_b128 %6, %14 offset:%16\n\t"
                 "ds_read_b128 %7, %15 offset:%16\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(c[0]), "=&v"(c[1]), "=&v"(c[2]), "=&v"(c[3]), "=&v"(c[4]), "=&v"(c[5]), "=&v"(c[6]), "=&v"(c[7])
                 : "v"(ra[0]), "v"(ra[1]), "v"(ra[2]), "v"(ra[3]), "v"(ra[4]), "v"(ra[5]), "v"(ra[6]), "v"(ra[7]),
                   "n"(J * UNIT_BYTES)
                 : "memory");
  }

  // One k-plane of tricubic_eval (libtricubic.f95:658-695, derx=dery=derz=0) for the 7-point stencil of evalrhs
  // (centre, x+-, y+-, z+-) plus NE (0/1) free points, all in one cell.  c[2j], c[2j+1] = a(0..3 + 4j + 16k).
  // Points that share a coordinate share the partial Horner sums; per point the operations and their order are
  // exactly those of a plain nested Horner evaluation (k outer, j middle, i inner).
  template <int NE>
  __device__ __forceinline__ static void plane_stencil(const d2_t (&c)[8], const double (&X)[3], const double (&Y)[3],
                                                       const double (&Z)[3], const double (&E)[3],
                                                       double (&vz)[7 + NE]) {
    double vc = 0.0, vxp = 0.0, vxm = 0.0, vyp = 0.0, vym = 0.0, ve = 0.0;
#pragma unroll
    for (int j = 3; j >= 0; --j) {
      const d2_t lo = c[2 * j], hi = c[2 * j + 1];
      double h0 = fma(fma(fma(hi.y, X[0], hi.x), X[0], lo.y), X[0], lo.x);
      double hp = fma(fma(fma(hi.y, X[1], hi.x), X[1], lo.y), X[1], lo.x);
      double hm = fma(fma(fma(hi.y, X[2], hi.x), X[2], lo.y), X[2], lo.x);
      vc = fma(vc, Y[0], h0);
      vxp = fma(vxp, Y[0], hp);
      vxm = fma(vxm, Y[0], hm);
      vyp = fma(vyp, Y[1], h0);
      vym = fma(vym, Y[2], h0);
      if (NE) {
        double he = fma(fma(fma(hi.y, E[0], hi.x), E[0], lo.y), E[0], lo.x);
        ve = fma(ve, E[1], he);
      }
    }
    vz[0] = fma(vz[0], Z[0], vc);
    vz[1] = fma(vz[1], Z[0], vxp);
    vz[2] = fma(vz[2], Z[0], vxm);
    vz[3] = fma(vz[3], Z[0], vyp);
    vz[4] = fma(vz[4], Z[0], vym);
    vz[5] = fma(vz[5], Z[1], vc);
    vz[6] = fma(vz[6], Z[2], vc);
    if (NE) vz[7 + NE - 1] = fma(vz[7 + NE - 1], E[2], ve);
  }
  // the same plane for NP unrelated points of one cell
  template <int NP>
  __device__ __forceinline__ static void plane_points(const d2_t (&c)[8], const double (&x)[NP], const double (&y)[NP],
                                                      const double (&z)[NP], double (&vz)[NP]) {
    double vy[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) vy[i] = 0.0;
#pragma unroll
    for (int j = 3; j >= 0; --j) {
      const d2_t lo = c[2 * j], hi = c[2 * j + 1];
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        double vx = fma(fma(fma(hi.y, x[i], hi.x), x[i], lo.y), x[i], lo.x);
        vy[i] = fma(vy[i], y[i], vx);
      }
    }
#pragma unroll
    for (int i = 0; i < NP; ++i) vz[i] = fma(vz[i], z[i], vy[i]);
  }

  // Densities at the evalrhs stencil around c (offsets d) and, when NE == 1, at one more point `extra`.
  // Ns[0] centre, Ns[1+2a] = c + d_a e_a, Ns[2+2a] = c - d_a e_a, Ns[7] = extra.
  //
  // The 4*nspec units of the call stream through the ring with a lookahead of three: while the lane evaluates
  // plane k of species s from buffer (3-k), the DMA of the next three units is in flight (the loads of one wave
  // retire in issue order, so s_waitcnt vmcnt(24) = "all but the youngest three units have landed").
  template <int NE>
  __device__ __forceinline__ void density_stencil(const double c[3], const double d[3], const double *extra,
                                                  double (&Ns)[7 + NE][4], double *lds, bool = true) const {
    double X[3], Y[3], Z[3], E[3] = {0.0, 0.0, 0.0};
    const int ci = ax.locate(c[0], X[0]), cj = ay.locate(c[1], Y[0]), ck = az.locate(c[2], Z[0]);
    bool same = true;
    same = same && (ax.locate(c[0] + d[0], X[1]) == ci) && (ax.locate(c[0] - d[0], X[2]) == ci);
    same = same && (ay.locate(c[1] + d[1], Y[1]) == cj) && (ay.locate(c[1] - d[1], Y[2]) == cj);
    same = same && (az.locate(c[2] + d[2], Z[1]) == ck) && (az.locate(c[2] - d[2], Z[2]) == ck);
    bool extra_same = true;
    if (NE) {
      extra_same = (ax.locate(extra[0], E[0]) == ci) && (ay.locate(extra[1], E[1]) == cj) && (az.locate(extra[2], E[2]) == ck);
    }
    unsigned long long a[8];
    unsigned ra[8];
    stage_prepare(cell_id(ci, cj, ck), a);
    read_addrs(lds, ra);
    issue_unit<384, 0>(a, lds); // species 0: planes 3, 2, 1 -> buffers 0, 1, 2
    issue_unit<256, 1>(a, lds);
    issue_unit<128, 2>(a, lds);
    double acc[7 + NE][4];
#pragma unroll
    for (int i = 0; i < 7 + NE; ++i)
#pragma unroll
      for (int k = 0; k < 4; ++k) acc[i][k] = 0.0;
#pragma unroll 1
    for (int s = 0; s < nspec; ++s) {
      const bool last = s + 1 >= nspec; // wave-uniform
      double vz[7 + NE];
#pragma unroll
      for (int i = 0; i < 7 + NE; ++i) vz[i] = 0.0;
      d2_t cf[8];
      issue_unit<0, 3>(a, lds); // plane 0 of this species
      wait_vm<24>();
      read_unit<0>(ra, cf);
      plane_stencil<NE>(cf, X, Y, Z, E, vz); // k = 3
      if (!last) {
        issue_unit<512 + 384, 0>(a, lds); // plane 3 of the next species
        wait_vm<24>();
      } else wait_vm<16>();
      read_unit<1>(ra, cf);
      plane_stencil<NE>(cf, X, Y, Z, E, vz); // k = 2
      if (!last) {
        issue_unit<512 + 256, 1>(a, lds);
        wait_vm<24>();
      } else wait_vm<8>();
      read_unit<2>(ra, cf);
      plane_stencil<NE>(cf, X, Y, Z, E, vz); // k = 1
      if (!last) {
        issue_unit<512 + 128, 2>(a, lds);
        wait_vm<24>();
      } else wait_vm<0>();
      read_unit<3>(ra, cf);
      plane_stencil<NE>(cf, X, Y, Z, E, vz); // k = 0
      // s is a run-time loop index: select statically so that acc stays in registers (no scratch)
#pragma unroll
      for (int i = 0; i < 7 + NE; ++i)
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[i][k] = (k == s) ? vz[i] : acc[i][k];
#pragma unroll
      for (int t = 0; t < 8; ++t) a[t] += 512;
    }
    // rare: a stencil point (offsets ~1e-6 |x| against cells of ~1e5..1e6 m) or the free point lies in another cell
    if (__any(!same || !extra_same)) {
      if (!same) {
#pragma unroll
        for (int ax_ = 0; ax_ < 3; ++ax_)
#pragma unroll
          for (int sg = 0; sg < 2; ++sg) {
            double pt[3] = {c[0], c[1], c[2]}, t[4] = {0.0, 0.0, 0.0, 0.0};
            pt[ax_] = sg ? c[ax_] - d[ax_] : c[ax_] + d[ax_];
            point_direct(pt[0], pt[1], pt[2], t);
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[1 + 2 * ax_ + sg][k] = t[k];
          }
      }
      if (NE && !extra_same) {
        double t[4] = {0.0, 0.0, 0.0, 0.0};
        point_direct(extra[0], extra[1], extra[2], t);
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[7 + NE - 1][k] = t[k];
      }
    }
#pragma unroll
    for (int i = 0; i < 7 + NE; ++i)
#pragma unroll
      for (int k = 0; k < 4; ++k) Ns[i][k] = (k < nspec) ? exp(acc[i][k]) : 0.0; // Ns = exp(Ns) (:206)
  }

  // NP unrelated points (layered kernels; not performance-critical): same units, no lookahead.
  template <int NP>
  __device__ __forceinline__ void density(const double (&p)[NP][3], double (&Ns)[NP][4], double *lds) const {
    // cell of point 0; the other points normally share it
    double xl[NP], yl[NP], zl[NP];
    int ci = ax.locate(p[0][0], xl[0]);
    int cj = ay.locate(p[0][1], yl[0]);
    int ck = az.locate(p[0][2], zl[0]);
    unsigned strag = 0; // bit i: point i lies in another cell
#pragma unroll
    for (int i = 1; i < NP; ++i) {
      int c1 = ax.locate(p[i][0], xl[i]);
      int c2 = ay.locate(p[i][1], yl[i]);
      int c3 = az.locate(p[i][2], zl[i]);
      if (c1 != ci || c2 != cj || c3 != ck) strag |= 1u << i;
    }
    unsigned long long a[8];
    unsigned ra[8];
    stage_prepare(cell_id(ci, cj, ck), a);
    read_addrs(lds, ra);
    double acc[NP][4];
#pragma unroll
    for (int i = 0; i < NP; ++i)
#pragma unroll
      for (int k = 0; k < 4; ++k) acc[i][k] = 0.0;
#pragma unroll 1
    for (int s = 0; s < nspec; ++s) {
      double vz[NP];
#pragma unroll
      for (int i = 0; i < NP; ++i) vz[i] = 0.0;
      d2_t cf[8];
      issue_unit<384, 0>(a, lds);
      issue_unit<256, 1>(a, lds);
      issue_unit<128, 2>(a, lds);
      issue_unit<0, 3>(a, lds);
      wait_vm<0>();
      read_unit<0>(ra, cf);
      plane_points<NP>(cf, xl, yl, zl, vz);
      read_unit<1>(ra, cf);
      plane_points<NP>(cf, xl, yl, zl, vz);
      read_unit<2>(ra, cf);
      plane_points<NP>(cf, xl, yl, zl, vz);
      read_unit<3>(ra, cf);
      plane_points<NP>(cf, xl, yl, zl, vz);
#pragma unroll
      for (int i = 0; i < NP; ++i)
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[i][k] = (k == s) ? vz[i] : acc[i][k];
#pragma unroll
      for (int t = 0; t < 8; ++t) a[t] += 512;
    }
    if (__any(strag != 0)) {
#pragma unroll
      for (int i = 1; i < NP; ++i)
        if (strag & (1u << i)) {
          double t[4] = {0.0, 0.0, 0.0, 0.0};
          point_direct(p[i][0], p[i][1], p[i][2], t);
#pragma unroll
          for (int k = 0; k < 4; ++k) acc[i][k] = t[k];
        }
    }
#pragma unroll
    for (int i = 0; i < NP; ++i)
#pragma unroll
      for (int s = 0; s < 4; ++s) Ns[i][s] = (s < nspec) ? exp(acc[i][s]) : 0.0; // Ns = exp(Ns) (:206)
  }
};

} // namespace srt
