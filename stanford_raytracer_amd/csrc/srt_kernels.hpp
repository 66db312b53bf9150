// srt_kernels.hpp -- kernels of the many-ray Haselgrove integrator (templates over the density model).
//
// Launch shape: blocks of exactly one wave (64 threads); one ray per lane.  The trace kernel is
// persistent: each wave pulls rays from a global queue, integrates all its lanes in lock step (one
// RK attempt per loop trip) and refills lanes whose rays have stopped (wavefront ballot + one atomic
// per refill; a new ray's launch point is evaluated in the running rays' end-point slot of that trip)
// until the queue is empty -- every wave reaches that exit.
#pragma once
#include "srt_models.hpp"

namespace srt {

// Build with -DSRT_TRIP_TIMING for a cycle breakdown of one attempt trip in block 0 (srt_trip_cycles[], printed after a trace
// launch when the environment variable SRT_TRIP_TIMING is set): 0 stencil densities inside the stages, 1 rest of evalrhs,
// 2 end-point stencil densities, 3 error term, 4 root re-projection, 5 end-point right-hand side, 6 bookkeeping, 7 trips,
// 8 trips with <= 8 active lanes, 9 loop top (stop tests, refill), 10 stage bookkeeping.
#ifdef SRT_TRIP_TIMING
#define SRT_TT_BEGIN() __builtin_amdgcn_sched_barrier(0); unsigned long long tt0_ = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0)
#define SRT_TT(slot)                                                                                   \
  do {                                                                                                 \
    __builtin_amdgcn_sched_barrier(0);                                                                 \
    unsigned long long tt1_ = __builtin_readcyclecounter();                                            \
    __builtin_amdgcn_sched_barrier(0);                                                                 \
    if (threadIdx.x == 0) srt_tt_lds()[slot] += tt1_ - tt0_;                        \
    tt0_ = tt1_;                                                                                       \
  } while (0)
#define SRT_TT_COUNT(slot, v)                                                                          \
  do {                                                                                                 \
    if (threadIdx.x == 0) srt_tt_lds()[slot] += (unsigned long long)(v);              \
  } while (0)
#else
#define SRT_TT_BEGIN() do {} while (0)
#define SRT_TT(slot) do {} while (0)
#define SRT_TT_COUNT(slot, v) do {} while (0)
#endif

struct TraceParams {
  double dt0, dtmax, tmax, maxerr, minalt, del;
  int maxsteps, root, fixedstep, outputper, first_attempt_policy, refill_threshold;
  int slots;
  int wave_cap; // rays a wave holds at most (64 = every lane; 8 = every trip runs in the Ngo model's tail mode, see srt_api.hip)
};

struct TraceArgs {
  const double *pos0; // SoA [3][nrays]
  const double *dir0; // SoA [3][nrays]
  const double *w0;
  const int *order;   // optional: queue position -> ray id (srt_params.ray_order); nullptr = identity
  long long nrays;
  double *rows;       // [nrays][slots][ROW]
  int *nrows;         // [nrays]
  int *stopcond;      // [nrays]
  unsigned long long *counters; // [0] queue head, [1] accepted steps, [2] attempts, [3] wave-attempts (loop trips of all waves)
  double *scratch;    // scattered model: staging records, ScatteredModel::REC_CAP * REC doubles per block (or nullptr)
  double *scratch2;   // scattered model: candidate blocks, ScatteredModel::BLOCK_DOUBLES doubles per block (or nullptr)
  TraceParams p;
};

// Stencil of raytracer_evalrhs: centre + the six points x +- d e_c of dispersion_relation_dFdx
// (raytracer.f95:239-262).  p[0] centre, p[1+2c] = x + d_c e_c, p[2+2c] = x - d_c e_c.
template <int NP>
__device__ __forceinline__ void stencil_points(const double x[3], double del, double (&p)[NP][3], double d[3]) {
#pragma unroll
  for (int c = 0; c < 3; ++c) d[c] = fd_step(del, x[c]);
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    p[i][0] = x[0];
    p[i][1] = x[1];
    p[i][2] = x[2];
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    p[1 + 2 * c][c] = x[c] + d[c];
    p[2 + 2 * c][c] = x[c] - d[c];
  }
}

// One offset point of dispersion_relation_dFdx (raytracer.f95:239-262): the field at x +- d e_c, the Stix parameters of the
// plasma there, the dispersion function.  ONE compiled body: in tail mode (below) another lane evaluates it for the ray's
// lane and has to produce that lane's bits.
template <class CM>
__device__ __noinline__ double offset_F(const CM &cm, double px, double py, double pz, double N0, double N1, double N2, double N3,
                                        double w, double n0, double n1, double n2) {
  double Bp[3];
  bfield(cm, px, py, pz, Bp);
  const double Bp2 = Bp[0] * Bp[0] + Bp[1] * Bp[1] + Bp[2] * Bp[2];
  const double Ns[4] = {N0, N1, N2, N3};
  const Stix st = stix_parameters(cm.sp, w, Ns, sqrt(Bp2));
  const double n[3] = {n0, n1, n2};
  return dispersion_F(st, n, Bp, Bp2);
}
// Models whose launches end in a long tail of a few rays (config[1]: 27 of 100 k rays run to maxsteps, the mean ray stops
// after 29 steps) spread the six offset evaluations of a right-hand side over idle lanes when at most 8 lanes of the wave
// carry a ray: the launch then runs at one ray's sequential trip time, and those six chains in a row were a quarter of it.
#ifndef SRT_NGO_SPREAD
#define SRT_NGO_SPREAD 1 // (0: every right-hand side in the ray's own lane, for A/B)
#endif
#ifndef SRT_INTERP_SPREAD
#define SRT_INTERP_SPREAD 0 // (1: the same for the interp model's tail -- measured, HISTORY section 12.3)
#endif
template <class M>
constexpr bool spread_rhs() {
  return (SRT_NGO_SPREAD != 0 && std::is_same<M, NgoModel>::value) || (SRT_INTERP_SPREAD != 0 && std::is_same<M, InterpModel>::value);
}
// the exchange area of the spread right-hand sides: ONE allocation per kernel, whichever call sites use it
struct SpreadLds {
  double xch[8][48]; // per ray: 6 x {p(3), Ns(4)}, w, n(3)
  double xchF[8][8];
};
__device__ __forceinline__ SpreadLds &spread_lds() {
  __shared__ __attribute__((aligned(16))) SpreadLds area;
  return area;
}

// raytracer_evalrhs (raytracer.f95:282-314) given the densities at the 7 stencil points: 7 dipole-field
// evaluations, 3 + 6 Stix evaluations, 14 dispersion-function evaluations.  Also returns dF/dk and dF/dw
// at the centre (the group-velocity terms of raytracer.f95:916-919 are the same numbers).
// SPREAD (dipole field only): the six offset evaluations go through offset_F -- one after the other in the ray's lane, or, when
// __ballot(need) has at most 8 lanes, point i of the r-th such lane in lane 8 r + i (inputs and results through LDS).
template <int NP, bool SPREAD = false, class CM>
__device__ __forceinline__ void rhs_from_plasma(const CM &cm, const double x[3], const double k[3], double w,
                                                const double d[3], const double (&p)[NP][3],
                                                const double (&Ns)[NP][4], double rhs[6], double dk[3], double &dw,
                                                double B[3], const double (*Bpre)[3] = nullptr, bool need = true) {
  // use_igrf (wave-uniform): the seven fields are synthesised together (srt_device.hpp igrf_core); dipole: one by one
  const bool igrf = field_is_igrf(cm);
  double Ball[7][3];
  if (igrf) {
    if (Bpre != nullptr) { // the caller has synthesised the stencil's fields already (trace kernel, end-point stencil)
#pragma unroll
      for (int i = 0; i < 7; ++i)
        for (int c = 0; c < 3; ++c) Ball[i][c] = Bpre[i][c];
    } else {
      double p7[7][3];
#pragma unroll
      for (int i = 0; i < 7; ++i)
        for (int c = 0; c < 3; ++c) p7[i][c] = p[i][c];
      bfield_igrf<7, field_igrf_only<CM>()>(cm.fld, p7, Ball);
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) B[c] = Ball[0][c];
  } else {
    bfield(cm, x[0], x[1], x[2], B);
  }
  double B2 = B[0] * B[0] + B[1] * B[1] + B[2] * B[2];
  double Bmag = sqrt(B2);
  Stix st0 = stix_parameters(cm.sp, w, Ns[0], Bmag);
  double cw = fdiv(cm.C, w);
  dFdk(st0, k, cw, B, B2, dk);
  dw = dFdw(cm.sp, k, w, cm.C, Ns[0], B, B2, Bmag);
  double n[3] = {k[0] * cw, k[1] * cw, k[2] * cw};
  double dx[3];
  bool spread_done = false;
  if constexpr (SPREAD) {
    if (!igrf) { // wave-uniform
      double F6[7] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
      const unsigned long long needy = __ballot(need);
      const int nneedy = __popcll(needy);
      if (nneedy >= 1 && nneedy <= 8) { // wave-uniform: tail mode
        SpreadLds &sl_ = spread_lds();
        double (&xch)[8][48] = sl_.xch;
        double (&xchF)[8][8] = sl_.xchF;
        const int lane = (int)__lane_id();
        const int rank = __popcll(needy & ((1ull << lane) - 1ull));
        if (need) {
#pragma unroll
          for (int i = 1; i <= 6; ++i) {
            double *q = &xch[rank][7 * (i - 1)];
            q[0] = p[i][0], q[1] = p[i][1], q[2] = p[i][2];
            q[3] = Ns[i][0], q[4] = Ns[i][1], q[5] = Ns[i][2], q[6] = Ns[i][3];
          }
          xch[rank][42] = w, xch[rank][43] = n[0], xch[rank][44] = n[1], xch[rank][45] = n[2];
        }
        __syncthreads(); // block == one wave
        const int slot = lane >> 3, i = lane & 7;
        const bool work = slot < nneedy && i >= 1 && i <= 6;
        const double *src = xch[work ? slot : 0]; // (idle lanes ride along on the first ray's first point: one call site, uniform control flow)
        const double *q = src + 7 * ((work ? i : 1) - 1);
        const double F = offset_F(cm, q[0], q[1], q[2], q[3], q[4], q[5], q[6], src[42], src[43], src[44], src[45]);
        if (work) xchF[slot][i] = F;
        __syncthreads();
#pragma unroll
        for (int i2 = 1; i2 <= 6; ++i2) F6[i2] = xchF[need ? rank : 0][i2];
        __syncthreads(); // (the next right-hand side overwrites both arrays)
      } else {
#pragma unroll 1
        for (int i = 1; i <= 6; ++i) {
          double px = p[1][0], py = p[1][1], pz = p[1][2], a0 = Ns[1][0], a1 = Ns[1][1], a2 = Ns[1][2], a3 = Ns[1][3];
#pragma unroll
          for (int t = 2; t <= 6; ++t) { // (static indices: the arrays stay in registers)
            px = i == t ? p[t][0] : px, py = i == t ? p[t][1] : py, pz = i == t ? p[t][2] : pz;
            a0 = i == t ? Ns[t][0] : a0, a1 = i == t ? Ns[t][1] : a1, a2 = i == t ? Ns[t][2] : a2, a3 = i == t ? Ns[t][3] : a3;
          }
          const double F = offset_F(cm, px, py, pz, a0, a1, a2, a3, w, n[0], n[1], n[2]);
#pragma unroll
          for (int t = 1; t <= 6; ++t) F6[t] = i == t ? F : F6[t];
        }
      }
#pragma unroll
      for (int c = 0; c < 3; ++c) dx[c] = fdiv(F6[1 + 2 * c] - F6[2 + 2 * c], d[c]) * 0.5;
      spread_done = true;
    }
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    if (spread_done) break;
    double Fpm[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int i = 1 + 2 * c + s;
      double Bp[3];
      if (igrf) {
#pragma unroll
        for (int q = 0; q < 3; ++q) Bp[q] = Ball[i][q];
      } else {
        bfield(cm, p[i][0], p[i][1], p[i][2], Bp);
      }
      double Bp2 = Bp[0] * Bp[0] + Bp[1] * Bp[1] + Bp[2] * Bp[2];
      Stix st = stix_parameters(cm.sp, w, Ns[i], sqrt(Bp2));
      Fpm[s] = dispersion_F(st, n, Bp, Bp2);
    }
    dx[c] = fdiv(Fpm[0] - Fpm[1], d[c]) * 0.5;
  }
  double idw = fdiv(1.0, dw);
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    rhs[c] = -(dk[c] * idw);
    rhs[3 + c] = dx[c] * idw;
  }
}

template <class M, class CM>
__device__ __forceinline__ void evalrhs(const M &m, const CM &cm, const double x[6], double w, double del,
                                        double rhs[6], double *lds, bool need = true) {
  double p[7][3], d[3], Ns[7][4], dk[3], dw, B[3];
  SRT_TT_BEGIN();
  stencil_points<7>(x, del, p, d);
  m.template density_stencil<0>(x, d, nullptr, Ns, lds, need);
  SRT_TT(0);
  rhs_from_plasma<7, spread_rhs<M>()>(cm, x, x + 3, w, d, p, Ns, rhs, dk, dw, B, nullptr, need);
  SRT_TT(1);
}

// Explicit RK stage loop shared by rk4 (raytracer.f95:504-532) and rk45 (:534-596).
// Stage vectors live in registers with static indices; the stage loop is not unrolled so that a
// single copy of evalrhs sits in the instruction cache.
struct Tableau {
  double a[6][5];
  int stages;
};
__constant__ Tableau TAB_RKF45 = {{{0, 0, 0, 0, 0},
                                   {rkf::a21, 0, 0, 0, 0},
                                   {rkf::a31, rkf::a32, 0, 0, 0},
                                   {rkf::a41, rkf::a42, rkf::a43, 0, 0},
                                   {rkf::a51, rkf::a52, rkf::a53, rkf::a54, 0},
                                   {rkf::a61, rkf::a62, rkf::a63, rkf::a64, rkf::a65}},
                                  6};
__constant__ Tableau TAB_RK4 = {{{0, 0, 0, 0, 0}, {0.5, 0, 0, 0, 0}, {0, 0.5, 0, 0, 0}, {0, 0, 1.0, 0, 0},
                                 {0, 0, 0, 0, 0}, {0, 0, 0, 0, 0}},
                                4};

// r1 = evalrhs at x when the caller already has it (first stage; it does not depend on dt, so it is carried
// from the previous accepted step and across rejected attempts), else nullptr.
template <class M, class CM>
__device__ __forceinline__ void rk_stages(const M &m, const CM &cm, const Tableau &tab, const double x[6],
                                          double w, double del, double dt, double (&ks)[6][6], double *lds,
                                          const double *r1 = nullptr, bool need = true) {
#pragma unroll
  for (int j = 0; j < 6; ++j)
#pragma unroll
    for (int c = 0; c < 6; ++c) ks[j][c] = 0.0;
  if (r1) {
#pragma unroll
    for (int c = 0; c < 6; ++c) ks[0][c] = dt * r1[c];
  }
#pragma unroll 1
  for (int s = (r1 ? 1 : 0); s < tab.stages; ++s) {
    double tmp[6], arow[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) arow[j] = tab.a[s][j]; // wave-uniform: one wide scalar load per stage
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      double acc = 0.0;
#pragma unroll
      for (int j = 0; j < 5; ++j)
        if (j < s) acc += arow[j] * ks[j][c];
      tmp[c] = x[c] + acc;
    }
    double r[6];
    evalrhs(m, cm, tmp, w, del, r, lds, need);
#pragma unroll
    for (int j = 0; j < 6; ++j)
      if (j == s) {
#pragma unroll
        for (int c = 0; c < 6; ++c) ks[j][c] = dt * r[c];
      }
  }
}

__device__ __forceinline__ void rk4_combine(const double x[6], const double (&ks)[6][6], double out[6]) {
#pragma unroll
  for (int c = 0; c < 6; ++c)
    out[c] = x[c] + (1.0 / 6.0) * (ks[0][c] + 2.0 * ks[1][c] + 2.0 * ks[2][c] + ks[3][c]);
}
__device__ __forceinline__ void rk45_combine(const double x[6], const double (&ks)[6][6], double o4[6],
                                             double o5[6]) {
#pragma unroll
  for (int c = 0; c < 6; ++c) {
    o4[c] = x[c] + (rkf::b41 * ks[0][c] + rkf::b43 * ks[2][c] + rkf::b44 * ks[3][c] + rkf::b45 * ks[4][c]);
    o5[c] = x[c] + (rkf::b51 * ks[0][c] + rkf::b53 * ks[2][c] + rkf::b54 * ks[3][c] + rkf::b55 * ks[4][c] +
                    rkf::b56 * ks[5][c]);
  }
}

// Plasma state at one position plus the derived quantities a trajectory row needs.
struct PointState {
  double Ns[4];
  double B[3], B2, Bmag;
};

// group velocity pieces at (x, k, w): dfdk, dfdw (raytracer.f95:916-919 / :700-703)
__device__ __forceinline__ void group_terms(const Common &cm, const PointState &ps, const double k[3], double w,
                                            double dk[3], double &dw) {
  Stix st = stix_parameters(cm.sp, w, ps.Ns, ps.Bmag);
  dFdk(st, k, cm.C / w, ps.B, ps.B2, dk);
  dw = dFdw(cm.sp, k, w, cm.C, ps.Ns, ps.B, ps.B2, ps.Bmag);
}

__device__ __forceinline__ void store_row(double *row, double t, const double x[6], const double vp[3],
                                          const double vg[3], const double n[3], const PointState &ps) {
  row[0] = t;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    row[1 + c] = x[c];
    row[4 + c] = vp[c];
    row[7 + c] = vg[c];
    row[10 + c] = n[c];
    row[13 + c] = ps.B[c];
  }
#pragma unroll
  for (int s = 0; s < 4; ++s) row[16 + s] = ps.Ns[s];
}

// =============================================================================================
// raytracer_run for a whole launch set (raytracer.f95:609-995 x the driver loop :1144-1232).
// FOPT: field option fixed at compile time -- 0 dipole, 1 IGRF main field alone, 2 the general tail (T04_s on an IGRF or
// dipole base, chosen at run time inside)
template <class M, bool FIXED, bool USE_LDS, int FOPT = 0>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WaveBudget<M>::WAVES_PER_EU))) void trace_kernel(const M *__restrict__ mp, const Common *__restrict__ cp, TraceArgs a) {
  const M &m = *mp;
  constexpr bool IGRF = FOPT != 0;
  typedef typename std::conditional<FOPT == 0, CommonDipole, typename std::conditional<FOPT == 1, CommonIgrfOnly, CommonIgrf>::type>::type CM;
  const CM &cm = *static_cast<const CM *>(cp);
  __shared__ __attribute__((aligned(16))) double tile[USE_LDS ? WaveBudget<M>::LDS_DOUBLES : 2];
  double *lds = USE_LDS ? tile : nullptr;
  bind_scratch(m, lds, a.scratch, a.scratch2);
  const TraceParams &P = a.p;
  const int lane = threadIdx.x;
  const Tableau &tab = FIXED ? TAB_RK4 : TAB_RKF45;

  // per-lane ray state (registers for the whole life of the ray)
  long long ray = -1;
  bool active = false, needinit = false;
  bool queue_empty = false; // wave-uniform
  double x[6] = {2.0 * R_E, 0.0, 0.5 * R_E, 1e-3, 0.0, 1e-4}; // benign state for lanes without a ray
  double w = 2.0e4, t = 0.0, dt = P.dt0;
  double vg[3] = {0, 0, 0};
  double r1[6] = {0, 0, 0, 0, 0, 0}; // evalrhs at the current state (first RK stage), carried across attempts
  double dirv[3] = {0, 0, 0};
  int nstep = 1, lastrefinedown = 0;
  bool first_attempt = true;
  unsigned long long acc_steps = 0, acc_attempts = 0, wave_trips = 0;
  const int threshold = P.refill_threshold > 0 ? P.refill_threshold : 1;
#ifdef SRT_PHASE_TIMING
  const unsigned long long srt_wave_t0 = __builtin_readcyclecounter();
#endif
#ifdef SRT_TRIP_TIMING
  if (threadIdx.x < 16) srt_tt_lds()[threadIdx.x] = 0ull;
  __syncthreads();
#endif
  SRT_TT_BEGIN();

  for (;;) {
    // ---- A. loop-top tests of raytracer_run (:749-763)
    if (active) {
      int stop = -1;
      if (t >= P.tmax) stop = 0;
      else {
        int sc = stop_conditions(x, vg, dt, nstep, P.maxsteps, P.minalt);
        if (sc != 0) stop = sc;
      }
      if (stop >= 0) {
        a.nrows[ray] = nstep;
        a.stopcond[ray] = stop;
        acc_steps += (unsigned long long)(nstep - 1);
        active = false;
      }
    }
    // ---- B. refill free lanes from the queue (ballot compaction)
    unsigned long long freemask = __ballot(!active);
    int nfree = __popcll(freemask);
    const int room = P.wave_cap - (WAVE - nfree);      // a wave holds at most wave_cap rays
    const int take = nfree < room ? nfree : (room > 0 ? room : 0);
    if (!queue_empty && take > 0 && (nfree >= threshold || nfree == WAVE)) {
      unsigned long long base = 0;
      if (lane == 0) base = atomicAdd(a.counters, (unsigned long long)take);
      base = __shfl(base, 0);
      if (!active) {
        int rank = __popcll(freemask & ((1ull << lane) - 1ull));
        long long id = (long long)base + rank;
        if (rank < take && id < a.nrays) {
          if (a.order) id = a.order[id];
          ray = id;
          needinit = true;
          new_ray_hook(m, lds, true);
          x[0] = a.pos0[id];
          x[1] = a.pos0[a.nrays + id];
          x[2] = a.pos0[2 * a.nrays + id];
          dirv[0] = a.dir0[id];
          dirv[1] = a.dir0[a.nrays + id];
          dirv[2] = a.dir0[2 * a.nrays + id];
          w = a.w0[id];
        }
      }
      if ((long long)base + take >= a.nrays) queue_empty = true;
    }
    // ---- C. (newly claimed rays have no step of their own this trip: they ride along with the running ones and
    // their launch point takes the place of the step's end point in E/F/G, see there; :661-742.  Every ray is
    // initialised by that one code path, so its arithmetic does not depend on when it was claimed.)
    // ---- D. exit when nothing is left (every wave reaches this)
    if (!__any(active || needinit)) {
      if (queue_empty) break;
      continue;
    }

    // ---- E. one attempt for every lane (:770-817).  Stage 1 = r1 (carried); stages 2.. are evaluated here.
    acc_attempts += active ? 1ull : 0ull;
    ++wave_trips;
    SRT_TT(9);
    SRT_TT_COUNT(7, 1);
#ifdef SRT_TRIP_TIMING
    const int tt_live_ = __popcll(__ballot(active || needinit)); // (a ballot inside the macro's lane-0 branch would see lane 0 only)
    SRT_TT_COUNT(8, tt_live_ <= 8 ? 1 : 0);
#endif
    double est1[6], est2[6];
    {
      double ks[6][6];
      rk_stages(m, cm, tab, x, w, P.del, dt, ks, lds, r1, active);
      SRT_TT(10); // all stages (slots 0 and 1 are inside it)
      if (FIXED) {
        rk4_combine(x, ks, est2);
#pragma unroll
        for (int c = 0; c < 6; ++c) est1[c] = est2[c];
      } else {
        rk45_combine(x, ks, est1, est2);
      }
    }
    if (needinit) { // a ray claimed this trip: its launch point is evaluated where the others evaluate their end point
#pragma unroll
      for (int c = 0; c < 3; ++c) est1[c] = est2[c] = x[c];
    }
    bool reject = false;
    const double dtincr = dt;
    // One staging for everything evaluated at the step's end points: the plasma at est2 (error term, root
    // re-projection, output row -- the reference evaluates it 4 times), its 6-point stencil (= first RK
    // stage of the NEXT attempt), and the plasma at est1 (error term, adaptive only).
    constexpr int NPOST = FIXED ? 7 : 8;
    double pp[NPOST][3], dpost[3], NP_[NPOST][4];
    stencil_points<NPOST>(est2, P.del, pp, dpost);
    if (!FIXED) {
      pp[NPOST - 1][0] = est1[0];
      pp[NPOST - 1][1] = est1[1];
      pp[NPOST - 1][2] = est1[2];
    }
    m.template density_stencil<NPOST - 7>(est2, dpost, FIXED ? nullptr : est1, NP_, lds, active || needinit);
    SRT_TT(2);
    PointState ps2;
#pragma unroll
    for (int s = 0; s < 4; ++s) ps2.Ns[s] = NP_[0][s];
    // IGRF / T04 field options: the fields of the whole end-point stencil (its centre est2, the six offsets and est1)
    // are synthesised together, once -- est2 alone and est1 alone are serial chains with nothing to hide behind, and
    // the stencil's centre IS est2 (rhs_from_plasma below takes the fields from here)
    double Bpost[IGRF ? NPOST : 1][3];
    if constexpr (IGRF) {
      bfield_igrf<NPOST, FOPT == 1>(cm.fld, pp, Bpost);
#pragma unroll
      for (int c = 0; c < 3; ++c) ps2.B[c] = Bpost[0][c];
    } else {
      bfield(cm, est2[0], est2[1], est2[2], ps2.B);
    }
    ps2.B2 = ps2.B[0] * ps2.B[0] + ps2.B[1] * ps2.B[1] + ps2.B[2] * ps2.B[2];
    ps2.Bmag = sqrt(ps2.B2);
    if (!FIXED) {
      // error term (:778-788): max of the relative differences in k and in dF/dk
      double s1 = 0.0, s2 = 0.0;
#pragma unroll
      for (int c = 3; c < 6; ++c) {
        s1 += fabs(est1[c] - est2[c]);
        s2 += fabs(est2[c]);
      }
      double kterm = s1 / s2;
      double err;
      if (first_attempt) {
        // the reference reads its local w before assigning it (SURVEY A-1)
        err = (P.first_attempt_policy == 1) ? kterm : __builtin_nan("");
      } else {
        PointState ps1;
#pragma unroll
        for (int s = 0; s < 4; ++s) ps1.Ns[s] = NP_[NPOST - 1][s];
        if constexpr (IGRF) {
#pragma unroll
          for (int c = 0; c < 3; ++c) ps1.B[c] = Bpost[NPOST - 1][c];
        } else {
          bfield(cm, est1[0], est1[1], est1[2], ps1.B);
        }
        ps1.B2 = ps1.B[0] * ps1.B[0] + ps1.B[1] * ps1.B[1] + ps1.B[2] * ps1.B[2];
        ps1.Bmag = sqrt(ps1.B2);
        double d1[3], d2[3];
        Stix st1 = stix_parameters(cm.sp, w, ps1.Ns, ps1.Bmag);
        dFdk(st1, est1 + 3, cm.C / w, ps1.B, ps1.B2, d1);
        Stix st2 = stix_parameters(cm.sp, w, ps2.Ns, ps2.Bmag);
        dFdk(st2, est2 + 3, cm.C / w, ps2.B, ps2.B2, d2);
        double t1 = 0.0, t2 = 0.0;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          t1 += fabs(d1[c] - d2[c]);
          t2 += fabs(d2[c]);
        }
        double dterm = t1 / t2;
        err = (kterm > dterm) ? kterm : dterm; // flang's MAX(x,y)
      }
      if (err > P.maxerr) {
        reject = true;
        if (active) {
          dt = 0.8 * dt;
          lastrefinedown = 1;
        }
      } else if (active && lastrefinedown == 0 && err < P.maxerr / 100.0 && dt * 1.25 < P.dtmax) {
        dt = dt * 1.25;
      }
    }
    SRT_TT(3);
    // ---- F. re-project |k| on the chosen root, keep direction (:819-836)
    double kdir[3] = {est2[3], est2[4], est2[5]};
    if (needinit) { // launch direction (:661-674)
      kdir[0] = dirv[0];
      kdir[1] = dirv[1];
      kdir[2] = dirv[2];
      if (kdir[0] == 0.0 && kdir[1] == 0.0 && kdir[2] == 0.0) {
        // field-aligned start: B/|B| with the radial component made positive
        double rr = sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
        double er[3] = {x[0] / rr, x[1] / rr, x[2] / rr};
        double br = ps2.B[0] * er[0] + ps2.B[1] * er[1] + ps2.B[2] * er[2];
        double adj = fabs(br) - br; // flip the radial part when negative
#pragma unroll
        for (int c = 0; c < 3; ++c) kdir[c] = ps2.B[c] + adj * er[c];
        double nb = sqrt(kdir[0] * kdir[0] + kdir[1] * kdir[1] + kdir[2] * kdir[2]);
#pragma unroll
        for (int c = 0; c < 3; ++c) kdir[c] /= nb;
      }
    }
    Roots rt = solve_dispersion(cm, kdir, w, ps2.Ns, ps2.B);
    double kmre = (P.root == 1) ? rt.k1re : rt.k2re;
    double kmim = (P.root == 1) ? rt.k1im : rt.k2im;
    double kn = sqrt(est2[3] * est2[3] + est2[4] * est2[4] + est2[5] * est2[5]);
    double knew[3], imsum = 0.0;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      double u = est2[3 + c] / kn;
      knew[c] = needinit ? kmre * kdir[c] : kmre * u; // launch: real(k0mag*dir0) (:683-690)
      double im = kmim * u;
      imsum += im * im;
    }
    // first-stage right-hand side at the would-be new state (also yields its group-velocity terms)
    double rn[6], dk[3], dw, Bn[3];
    SRT_TT(4);
    rhs_from_plasma<NPOST, spread_rhs<M>()>(cm, est2, knew, w, dpost, pp, NP_, rn, dk, dw, Bn, IGRF ? Bpost : nullptr, active || needinit);
    SRT_TT(5);
    if (needinit) {
      // launch state and row 0 (:693-742)
#pragma unroll
      for (int c = 0; c < 3; ++c) x[3 + c] = knew[c];
#pragma unroll
      for (int c = 0; c < 6; ++c) r1[c] = rn[c];
      t = 0.0;
      dt = P.dt0;
      lastrefinedown = 0;
      nstep = 1;
      first_attempt = true;
      double cw = cm.C / w;
      double n[3] = {x[3] * cw, x[4] * cw, x[5] * cw};
      double nn = n[0] * n[0] + n[1] * n[1] + n[2] * n[2];
      double vp[3];
      if (nn > 0.0) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          vp[c] = n[c] / nn;
          vg[c] = -(dk[c] / dw) / cm.C;
        }
      } else {
#pragma unroll
        for (int c = 0; c < 3; ++c) vp[c] = vg[c] = 0.0;
      }
      store_row(a.rows + (size_t)ray * (size_t)P.slots * ROW, t, x, vp, vg, n, ps2);
      active = true;
      needinit = false;
    } else if (active && !reject) {
      first_attempt = false; // w = est2(7) is assigned from here on (:821)
      if (imsum > 0.0) {
        // outside the resonance cone (:891-906)
        if (!FIXED) {
          dt = dt / 2.0;
          lastrefinedown = 1;
        } else {
          a.nrows[ray] = nstep; // `return` with stopcond 0 and a truncated trajectory
          a.stopcond[ray] = 0;
          acc_steps += (unsigned long long)(nstep - 1);
          active = false;
        }
      } else {
        // ---- G. accept (:908-986)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          x[c] = est2[c];
          x[3 + c] = knew[c];
        }
#pragma unroll
        for (int c = 0; c < 6; ++c) r1[c] = rn[c];
        lastrefinedown = 0;
        t = t + dtincr;
        nstep = nstep + 1;
        // a non-finite state makes the reference die in zgesvd (`stop`, blas.f95:208-211); end this ray only
        bool finite = true;
#pragma unroll
        for (int c = 0; c < 6; ++c) finite = finite && isfinite(x[c]);
        if (!finite) {
          a.nrows[ray] = nstep - 1;
          a.stopcond[ray] = 9;
          acc_steps += (unsigned long long)(nstep - 2);
          active = false;
        }
        double cw = cm.C / w;
        double n[3] = {x[3] * cw, x[4] * cw, x[5] * cw};
        double nn = n[0] * n[0] + n[1] * n[1] + n[2] * n[2];
        double vp[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          vp[c] = n[c] / nn;
          vg[c] = -(dk[c] / dw) / cm.C;
        }
        const int row = nstep - 1;
        if (finite && row % P.outputper == 0) {
          int slot = row / P.outputper;
          if (slot < P.slots)
            store_row(a.rows + ((size_t)ray * (size_t)P.slots + (size_t)slot) * ROW, t, x, vp, vg, n, ps2);
        }
      }
    }
    SRT_TT(6);
  }
#ifdef SRT_TRIP_TIMING
  __syncthreads();
  if (threadIdx.x < 16) atomicAdd(&srt_trip_cycles[threadIdx.x], srt_tt_lds()[threadIdx.x]);
#endif
#ifdef SRT_PHASE_TIMING
  if (lane == 0) {
    const unsigned long long t_end = __builtin_readcyclecounter();
    if (wave_trips > 0) {
      atomicAdd(&srt_wave_stats[0], 1ull);
      atomicAdd(&srt_wave_stats[1], t_end - srt_wave_t0);
    }
    atomicMin(&srt_wave_stats[2], srt_wave_t0);
    atomicMax(&srt_wave_stats[3], t_end);
  }
#endif
  // per-wave totals
  for (int off = 32; off > 0; off >>= 1) {
    acc_steps += __shfl_down(acc_steps, off);
    acc_attempts += __shfl_down(acc_attempts, off);
  }
  if (lane == 0) {
    atomicAdd(a.counters + 1, acc_steps);
    atomicAdd(a.counters + 2, acc_attempts);
    atomicAdd(a.counters + 3, wave_trips);
  }
}

// =============================================================================================
// Layered kernels for the parity ladder (one item per lane; all lanes of a wave participate).

// funcPlasmaParams: out[n][19] = qs(4) Ns(4) ms(4) nus(4) B0(3)
template <class M, bool USE_LDS>
__global__ __launch_bounds__(64) void params_kernel(const M *__restrict__ mp, const Common *__restrict__ cp, long long n, const double *x, double *out) {
  const M &m = *mp;
  const Common &cm = *cp;
  __shared__ __attribute__((aligned(16))) double tile[USE_LDS ? TILE_DOUBLES : 2];
  long long i = (long long)blockIdx.x * WAVE + threadIdx.x;
  long long j = i < n ? i : n - 1;
  double p[1][3] = {{x[3 * j], x[3 * j + 1], x[3 * j + 2]}};
  double Ns[1][4];
  m.template density<1>(p, Ns, tile);
  double B[3];
  bfield(cm, p[0][0], p[0][1], p[0][2], B);
  if (i < n) {
    double *o = out + 19 * i;
    for (int s = 0; s < 4; ++s) {
      o[s] = cm.sp.q[s];
      o[4 + s] = Ns[0][s];
      o[8 + s] = cm.sp.m[s];
      o[12 + s] = 0.0;
    }
    for (int c = 0; c < 3; ++c) o[16 + c] = B[c];
  }
}

// out[n][10] = F, S, D, P, R, L, Re k1, Im k1, Re k2, Im k2
template <class M, bool USE_LDS>
__global__ __launch_bounds__(64) void dispersion_kernel(const M *__restrict__ mp, const Common *__restrict__ cp, long long n, const double *x,
                                                        const double *k, const double *w, double *out) {
  const M &m = *mp;
  const Common &cm = *cp;
  __shared__ __attribute__((aligned(16))) double tile[USE_LDS ? TILE_DOUBLES : 2];
  long long i = (long long)blockIdx.x * WAVE + threadIdx.x;
  long long j = i < n ? i : n - 1;
  double p[1][3] = {{x[3 * j], x[3 * j + 1], x[3 * j + 2]}};
  double kk[3] = {k[3 * j], k[3 * j + 1], k[3 * j + 2]};
  double ww = w[j];
  double Ns[1][4];
  m.template density<1>(p, Ns, tile);
  double B[3];
  bfield(cm, p[0][0], p[0][1], p[0][2], B);
  double B2 = B[0] * B[0] + B[1] * B[1] + B[2] * B[2];
  Stix st = stix_parameters(cm.sp, ww, Ns[0], sqrt(B2));
  double cw = cm.C / ww;
  double nv[3] = {kk[0] * cw, kk[1] * cw, kk[2] * cw};
  double F = dispersion_F(st, nv, B, B2);
  Roots rt = solve_dispersion(cm, kk, ww, Ns[0], B);
  if (i < n) {
    double *o = out + 10 * i;
    o[0] = F; o[1] = st.S; o[2] = st.D; o[3] = st.P; o[4] = st.R; o[5] = st.L;
    o[6] = rt.k1re; o[7] = rt.k1im; o[8] = rt.k2re; o[9] = rt.k2im;
  }
}

__global__ void handedness_kernel(long long n, const double *in, int *out) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = is_right_handed(in[5 * i], in[5 * i + 1], in[5 * i + 2], in[5 * i + 3], in[5 * i + 4]) ? 1 : 0;
}

// out[n][14] = dFdk(3), dFdw, dFdx(3), rhs(7)
template <class M, bool USE_LDS>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WaveBudget<M>::WAVES_PER_EU))) void gradients_kernel(const M *__restrict__ mp, const Common *__restrict__ cp, long long n, const double *x,
                                                       const double *k, const double *w, double del,
                                                       double *out, double *scratch) {
  const M &m = *mp;
  const Common &cm = *cp;
  __shared__ __attribute__((aligned(16))) double tile[USE_LDS ? WaveBudget<M>::LDS_DOUBLES : 2];
  bind_scratch(m, USE_LDS ? tile : nullptr, scratch);
  long long i = (long long)blockIdx.x * WAVE + threadIdx.x;
  long long j = i < n ? i : n - 1;
  double st[6] = {x[3 * j], x[3 * j + 1], x[3 * j + 2], k[3 * j], k[3 * j + 1], k[3 * j + 2]};
  double ww = w[j];
  double rhs[6];
  evalrhs(m, cm, st, ww, del, rhs, USE_LDS ? tile : nullptr, i < n);
  // recover the gradients themselves: dfdw from a second (cheap) evaluation at the centre
  double p[1][3] = {{st[0], st[1], st[2]}};
  double Ns[1][4];
  m.template density<1>(p, Ns, tile);
  PointState ps;
  for (int s = 0; s < 4; ++s) ps.Ns[s] = Ns[0][s];
  bfield(cm, st[0], st[1], st[2], ps.B);
  ps.B2 = ps.B[0] * ps.B[0] + ps.B[1] * ps.B[1] + ps.B[2] * ps.B[2];
  ps.Bmag = sqrt(ps.B2);
  double dk[3], dw;
  group_terms(cm, ps, st + 3, ww, dk, dw);
  if (i < n) {
    double *o = out + 14 * i;
    for (int c = 0; c < 3; ++c) {
      o[c] = dk[c];
      o[4 + c] = rhs[3 + c] * dw; // dfdx = rhs*dfdw
      o[7 + c] = rhs[c];
      o[10 + c] = rhs[3 + c];
    }
    o[3] = dw;
    o[13] = 0.0;
  }
}

// out[n][21] = rk4(7), rk45 4th(7), rk45 5th(7)
template <class M, bool USE_LDS>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WaveBudget<M>::WAVES_PER_EU))) void rkstep_kernel(const M *__restrict__ mp, const Common *__restrict__ cp, long long n, const double *args,
                                                    const double *dtv, double del, double *out, double *scratch) {
  const M &m = *mp;
  const Common &cm = *cp;
  __shared__ __attribute__((aligned(16))) double tile[USE_LDS ? WaveBudget<M>::LDS_DOUBLES : 2];
  bind_scratch(m, USE_LDS ? tile : nullptr, scratch);
  long long i = (long long)blockIdx.x * WAVE + threadIdx.x;
  long long j = i < n ? i : n - 1;
  double st[6];
  for (int c = 0; c < 6; ++c) st[c] = args[7 * j + c];
  double ww = args[7 * j + 6], dt = dtv[j];
  double ks[6][6];
  double r4[6], o4[6], o5[6];
  rk_stages(m, cm, TAB_RK4, st, ww, del, dt, ks, USE_LDS ? tile : nullptr, nullptr, i < n);
  rk4_combine(st, ks, r4);
  rk_stages(m, cm, TAB_RKF45, st, ww, del, dt, ks, USE_LDS ? tile : nullptr, nullptr, i < n);
  rk45_combine(st, ks, o4, o5);
  if (i < n) {
    double *o = out + 21 * i;
    for (int c = 0; c < 6; ++c) {
      o[c] = r4[c];
      o[7 + c] = o4[c];
      o[14 + c] = o5[c];
    }
    o[6] = o[13] = o[20] = ww;
  }
}

} // namespace srt
