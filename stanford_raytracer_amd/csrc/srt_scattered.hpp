// srt_scattered.hpp -- modelnum = 4 on the device: scattered ln N_s samples, moving-least-squares interpolation
// (scattered_interp_dens_model_adapter.f95:284-312 + lsinterp_mod.f95:244-449, etainv :175-209, coswindow :215-221).
//
// MI355X-first data structure: the reference's pointer-linked kd-tree (kdtree_mod.f95) only serves a fixed-radius
// neighbour query (radius = maxnearest * window_scale), so the samples are binned once into a uniform grid with
// cell edge = that radius, sorted by cell (CSR), 64 B per sample {x,y,z, lnN_1..4, nearest-sample distance}.
// A query scans the 27 surrounding cells.  The neighbour SET is the reference's; only the order of summation
// differs (it is RNG-dependent in the reference anyway, SURVEY A-12).
//
// The normal equations are accumulated on the fly, so no neighbour list is stored:
//   A = sum_i w_i m_i m_i^T,  b_s = sum_i w_i m_i lnN_s(i),  w_i = 0.5*etainv(r_i) (= dinv_i^2), m_i = monomials(x_i - x)
//   solve A y = e_1 (Cholesky, dposv 'U'), ln N_s(x) = y . b_s           [== dot(aa, vals) with aa = (E y) * dinv]
#pragma once
#include <type_traits>
#include <utility>
#include "srt_models.hpp"

namespace srt {

#define SRT_LDS __attribute__((address_space(3)))
#define SRT_PRIV __attribute__((address_space(5)))
typedef float f4_t __attribute__((ext_vector_type(4)));

// Build with -DSRT_PHASE_TIMING for a cycle breakdown of coop_stencil (srt_phase_cycles[], printed after every trace
// launch when the environment variable SRT_PHASE_TIMING is set): 0 scan, 1 pass 1, 2 its reductions, 3 base pass,
// 4 pass 2, 5 combine + solve, 6 hand-off, 7 own-list path, 8 stencils served.
#ifdef SRT_PHASE_TIMING
__device__ unsigned long long srt_phase_cycles[16];
// sf_weights' choice of tier, per stencil: 0 stencils, 1 second-order tier, then why not: 2 no usable centre, 3 a sample within 100
// stencil widths, 4 the L dr bound, 5 the h bound, 6 the free point (any of its conditions)
__device__ unsigned long long srt_tier_stats[8];
// per-wave counters in LDS (behind the lists, the staging pointer and the hand-off area), flushed to the global ones
// once per coop_stencil call: an atomic per phase and stencil would itself show up in whatever waits on memory next
#define SRT_PHASE_LDS ((SRT_LDS unsigned long long *)srt_lds_base_ + ScatteredModel::LDS_PHASE)
#define SRT_PHASE_BEGIN(ldsbase)                                                                     \
  SRT_LDS char *srt_lds_base_ = (SRT_LDS char *)(ldsbase);                                           \
  unsigned long long srt_t0_ = __builtin_readcyclecounter()
#define SRT_PHASE(slot)                                                                              \
  do {                                                                                               \
    unsigned long long t1_ = __builtin_readcyclecounter();                                           \
    if (threadIdx.x == 0) SRT_PHASE_LDS[slot] += t1_ - srt_t0_;                                      \
    srt_t0_ = t1_;                                                                                   \
  } while (0)
#define SRT_PHASE_COUNT(slot) do { if (threadIdx.x == 0) SRT_PHASE_LDS[slot] += 1ull; } while (0)
#define SRT_PHASE_ADD(slot, v) do { if (threadIdx.x == 0) SRT_PHASE_LDS[slot] += (unsigned long long)(v); } while (0)
#define SRT_PHASE_ZERO(ldsbase)                                                                      \
  do {                                                                                               \
    SRT_LDS char *srt_lds_base_ = (SRT_LDS char *)(ldsbase);                                         \
    if (threadIdx.x < 16) SRT_PHASE_LDS[threadIdx.x] = 0ull;                                         \
  } while (0)
#define SRT_PHASE_FLUSH(ldsbase)                                                                     \
  do {                                                                                               \
    SRT_LDS char *srt_lds_base_ = (SRT_LDS char *)(ldsbase);                                         \
    if (threadIdx.x < 16) {                                                                          \
      atomicAdd(&srt_phase_cycles[threadIdx.x], SRT_PHASE_LDS[threadIdx.x]);                         \
      SRT_PHASE_LDS[threadIdx.x] = 0ull;                                                             \
    }                                                                                                \
  } while (0)
#else
#define SRT_PHASE_ADD(slot, v) do {} while (0)
#define SRT_PHASE_BEGIN(ldsbase) do {} while (0)
#define SRT_PHASE(slot) do {} while (0)
#define SRT_PHASE_COUNT(slot) do {} while (0)
#define SRT_PHASE_ZERO(ldsbase) do {} while (0)
#define SRT_PHASE_FLUSH(ldsbase) do {} while (0)
#endif

// Orders 2 and 3: A = sum w m m^T has J (J + 1) / 2 entries (55 / 210) but only N different sums -- the moments
// sum w x^a y^b z^c, a + b + c <= 2 * order (m_a m_c is a monomial of that degree): N = 35 / 84.  35 + 40 running sums fit the 256
// vector registers next to the record being folded in; 55 + 40 do not (the rest would sit in accumulation registers and be
// copied in and out per neighbour); for order 3, 84 + 80 sums fit the unified 512 registers, 210 + 80 fit nothing.
namespace mom {
constexpr int count(int deg) { return (deg + 1) * (deg + 2) * (deg + 3) / 6; } // monomials of degree <= deg in 3 variables
// moment order: by degree; within a degree x descending, then y descending
constexpr int find(int a, int b, int c) {
  const int d = a + b + c;
  int idx = d > 0 ? count(d - 1) : 0;
  for (int aa = d; aa >= 0; --aa)
    for (int bb = d - aa; bb >= 0; --bb) {
      if (aa == a && bb == b) return idx;
      ++idx;
    }
  return -1;
}
constexpr int exponent(int i, int ax) { // exponent of axis ax (0 x, 1 y, 2 z) of moment i
  int idx = 0;
  for (int d = 0; d < 16; ++d)
    for (int aa = d; aa >= 0; --aa)
      for (int bb = d - aa; bb >= 0; --bb) {
        if (idx == i) return ax == 0 ? aa : (ax == 1 ? bb : d - aa - bb);
        ++idx;
      }
  return -1;
}
// tabular_monomials, 3 dimensions (lsinterp_mod.f95:76-99): lexicographic in (x, y, z), total degree <= order
constexpr int tab_exponent(int order, int j, int ax) {
  int idx = 0;
  for (int x = 0; x <= order; ++x)
    for (int y = 0; y <= order - x; ++y)
      for (int z = 0; z <= order - x - y; ++z) {
        if (idx == j) return ax == 0 ? x : (ax == 1 ? y : z);
        ++idx;
      }
  return -1;
}
} // namespace mom

template <int ORDER>
struct MomentsT {
  static constexpr int J = mom::count(ORDER);
  static constexpr int N = mom::count(2 * ORDER);
  static constexpr int NT = J * (J + 1) / 2;
  // the monomial is its parent times one coordinate: the last non-zero exponent (z, else y, else x) reduced by one
  static constexpr int axis(int i) { return mom::exponent(i, 2) > 0 ? 2 : (mom::exponent(i, 1) > 0 ? 1 : 0); }
  static constexpr int parent(int i) {
    return mom::find(mom::exponent(i, 0) - (axis(i) == 0), mom::exponent(i, 1) - (axis(i) == 1), mom::exponent(i, 2) - (axis(i) == 2));
  }
  static constexpr int of_m(int a) { return mom::find(mom::tab_exponent(ORDER, a, 0), mom::tab_exponent(ORDER, a, 1), mom::tab_exponent(ORDER, a, 2)); }
  static constexpr int of_pair(int a, int c) {
    return mom::find(mom::tab_exponent(ORDER, a, 0) + mom::tab_exponent(ORDER, c, 0), mom::tab_exponent(ORDER, a, 1) + mom::tab_exponent(ORDER, c, 1),
                     mom::tab_exponent(ORDER, a, 2) + mom::tab_exponent(ORDER, c, 2));
  }
  // packed upper triangle t -> (a, c), row a holds A[a][a..J-1]
  static constexpr int row_of(int t) {
    int a = 0;
    while (t >= J - a) {
      t -= J - a;
      ++a;
    }
    return a;
  }
  static constexpr int col_of(int t) {
    int a = 0;
    while (t >= J - a) {
      t -= J - a;
      ++a;
    }
    return a + t;
  }
  template <int I>
  struct C { // compile-time constants of moment I / packed entry I
    static constexpr int par = parent(I), ax = axis(I);
  };
  template <int T>
  struct P {
    static constexpr int mom = of_pair(row_of(T), col_of(T));
  };
  template <int A>
  struct M1 {
    static constexpr int mom = of_m(A);
  };
  // mono[0 .. sizeof...(I)]: the monomials in moment order, each its parent times one coordinate
  template <int NM, int... I>
  __device__ __forceinline__ static void monomials(double (&mono)[NM], const double (&d)[3], std::integer_sequence<int, I...>) {
    mono[0] = 1.0;
    ((mono[I + 1] = mono[C<I + 1>::par] * d[C<I + 1>::ax]), ...);
  }
  // the same chain with mono[0] given (a weight: every entry then carries it)
  template <int NM, int... I>
  __device__ __forceinline__ static void monomials_from(double (&mono)[NM], const double (&d)[3], std::integer_sequence<int, I...>) {
    ((mono[I + 1] = mono[C<I + 1>::par] * d[C<I + 1>::ax]), ...);
  }
  // the moments of the top degree (indices LOW .. N - 1; their parents have degree 2 ORDER - 1 < LOW): sum += parent * coordinate
  static constexpr int LOW = mom::count(2 * ORDER - 1);
  template <int NM, int... I>
  __device__ __forceinline__ static void fold_leaves(double (&M)[N], const double (&mono)[NM], const double (&d)[3], std::integer_sequence<int, I...>) {
    static_assert(NM >= LOW, "the products up to degree 2 ORDER - 1");
    ((M[LOW + I] = fma(mono[C<LOW + I>::par], d[C<LOW + I>::ax], M[LOW + I])), ...);
  }
  template <int... T>
  __device__ __forceinline__ static void expand(const double (&M)[N], double (&A)[NT], std::integer_sequence<int, T...>) {
    ((A[T] = M[P<T>::mom]), ...);
  }
  template <int NM, int... A>
  __device__ __forceinline__ static void firsts(const double (&mono)[NM], double (&m)[J], std::integer_sequence<int, A...>) {
    ((m[A] = mono[M1<A>::mom]), ...);
  }
};
using Moments = MomentsT<2>;
static_assert(Moments::LOW == 20 && Moments::parent(20) >= 10 && Moments::parent(34) < 20 && Moments::of_m(9) < 10, "degree-4 moments are leaves");
static_assert(Moments::N == 35 && Moments::J == 10 && MomentsT<3>::N == 84 && MomentsT<3>::J == 20, "moment counts");
static_assert(Moments::of_m(4) == mom::find(0, 1, 1) && Moments::of_m(9) == mom::find(2, 0, 0) && MomentsT<3>::of_m(19) == mom::find(3, 0, 0) &&
                  MomentsT<3>::of_m(12) == mom::find(1, 0, 2),
              "tabular_monomials order");


struct ScatteredModel {
  const double *pts;     // [npts][8]: x, y, z, lnN[4], nearest-sample distance
  const double *xyz;     // [3][npts]: the positions once more, SoA (candidate scans: 24 coalesced bytes per sample, not 64)
  const int *cell_start; // [ncells + 1]
  double origin[3], inv_cell, radius, lws;
  double u11, inv_radius; // 1.1 (radius (1 + 5e-16))**1.1 and 1 / radius: per-model constants of the second-order tier's test (sf_weights)
  double bmargin; // candidate blocks (coop_stencil) hold the samples within radius * (1 + bmargin) of their centre; the grid's cell edge is that too
  int dims[3];
  int nspec, order, exact, npts;

  // The out-of-line functions below receive `this` in vector registers, so every member access through it would be a
  // per-lane flat load followed by a full wait (the cell_start lookups of one stencil: nine of them, one after the
  // other).  The model is wave-uniform: a copy fetched through the scalar cache (address from lane 0) lives in SGPRs.
  __device__ __forceinline__ static int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
  __device__ __forceinline__ static double uni(double v) {
    return __hiloint2double(uni(__double2hiint(v)), uni(__double2loint(v)));
  }
  // the value lane `l` holds (l: compile-time constant), as a wave-uniform number: v_readlane, no LDS round trip
  __device__ __forceinline__ static double from_lane(double v, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
  }
  template <class T>
  __device__ __forceinline__ static T *uni(T *v) {
    const unsigned long long a = (unsigned long long)v;
    const unsigned long long lo = (unsigned)uni((int)(unsigned)a), hi = (unsigned)uni((int)(unsigned)(a >> 32));
    return (T *)((hi << 32) | lo);
  }
  __device__ __forceinline__ ScatteredModel uniform_copy() const {
    const ScatteredModel *self = uni(this);
    ScatteredModel M;
    M.pts = uni(self->pts);
    M.xyz = uni(self->xyz);
    M.cell_start = uni(self->cell_start);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      M.origin[k] = uni(self->origin[k]);
      M.dims[k] = uni(self->dims[k]);
    }
    M.inv_cell = uni(self->inv_cell);
    M.radius = uni(self->radius);
    M.lws = uni(self->lws);
    M.bmargin = uni(self->bmargin);
    M.u11 = uni(self->u11);
    M.inv_radius = uni(self->inv_radius);
    M.nspec = uni(self->nspec);
    M.order = uni(self->order);
    M.exact = uni(self->exact);
    M.npts = uni(self->npts);
    return M;
  }
  __device__ __forceinline__ const SRT_AS1 double *gpts() const { return (const SRT_AS1 double *)pts; }
  __device__ __forceinline__ const SRT_AS1 double *gxyz() const { return (const SRT_AS1 double *)xyz; }
  // (immutable for the model's life: the constant address space lets uniform lookups go through the scalar cache)
  __device__ __forceinline__ const __attribute__((address_space(4))) int *gcells() const {
    return (const __attribute__((address_space(4))) int *)cell_start;
  }

  __device__ __forceinline__ double etainv(double r, double hin) const {
    const double eps = 5.0e-16;
    double win = 0.5 + 0.5 * cos(r * 2.0 * PI / radius / 2.0);
    if (exact == 1) {
      double q = r / hin;
      return ((1.0 + eps) / (exp(q * q) - 1.0 + eps)) * win;
    }
    double h = hin / 4.0;
    // ((r + radius*eps)/h)**1.1 as x * exp(0.1 ln x): x > 0 always; within ~2 ulp of pow() (the exponent 0.1 ln x is
    // small, so the logarithm's rounding is damped), at a third of its instruction count
    double x = (r + radius * eps) / h;
    return exp(-(x * exp(0.1 * log(x)))) * win;
  }

  // visit every sample within `radius` of x (strictly inside, kdtree_mod.f95:171)
  template <class F>
  __device__ __forceinline__ void for_neighbours(const double x[3], F f) const {
    int c[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      double t = floor((x[k] - origin[k]) * inv_cell);
      t = fmin(fmax(t, -2.0), (double)dims[k] + 1.0);
      c[k] = (int)t;
    }
    const double r2 = radius * radius;
    for (int dz = -1; dz <= 1; ++dz) {
      int cz = c[2] + dz;
      if (cz < 0 || cz >= dims[2]) continue;
      for (int dy = -1; dy <= 1; ++dy) {
        int cy = c[1] + dy;
        if (cy < 0 || cy >= dims[1]) continue;
        int x0 = c[0] - 1 < 0 ? 0 : c[0] - 1;
        int x1 = c[0] + 1 >= dims[0] ? dims[0] - 1 : c[0] + 1;
        if (x0 > x1) continue;
        int row = (cz * dims[1] + cy) * dims[0];
        int lo = cell_start[row + x0], hi = cell_start[row + x1 + 1]; // x-adjacent cells are contiguous
        for (int i = lo; i < hi; ++i) {
          const double *q = pts + (size_t)i * 8;
          double d0 = q[0] - x[0], d1 = q[1] - x[1], d2 = q[2] - x[2];
          double s = d0 * d0 + d1 * d1 + d2 * d2;
          if (s < r2) f(q, d0, d1, d2, sqrt(s));
        }
      }
    }
  }

  template <int J>
  __device__ __forceinline__ static void monomials(double dx, double dy, double dz, double (&m)[J]) {
    // tabular_monomials, 3 dimensions (lsinterp_mod.f95:70-99): exponent triples in the reference's order
    m[0] = 1.0;
    if (J >= 4) {
      m[1] = dz;
      m[2] = dy;
      m[3] = dx;
    }
    if (J == 10) {
      // order 2: (0,0,0)(0,0,1)(0,0,2)(0,1,0)(0,1,1)(0,2,0)(1,0,0)(1,0,1)(1,1,0)(2,0,0)
      m[1] = dz;
      m[2] = dz * dz;
      m[3] = dy;
      m[4] = dy * dz;
      m[5] = dy * dy;
      m[6] = dx;
      m[7] = dx * dz;
      m[8] = dx * dy;
      m[9] = dx * dx;
    }
    if constexpr (J == 20) { // order 3: the 20 monomials of degree <= 3 in the reference's (lexicographic) order
      const double dd[3] = {dx, dy, dz};
      double mono[20];
      MomentsT<3>::monomials(mono, dd, std::make_integer_sequence<int, 19>{});
      MomentsT<3>::firsts(mono, m, std::make_integer_sequence<int, 20>{});
    }
  }

  // Normal equations of order 3 (J = 20): the 84 moments + 80 right-hand sums, then A (210 entries) is laid out in
  // private memory and factorised by rolled loops (dposv 'U' as solve_fit): once per fit, against hundreds of folds.
  struct Sums20 {
    double Mm[MomentsT<3>::N], b[20][4];
    __device__ __forceinline__ void zero() {
#pragma unroll
      for (int t = 0; t < MomentsT<3>::N; ++t) Mm[t] = 0.0;
#pragma unroll
      for (int a = 0; a < 20; ++a)
#pragma unroll
        for (int s = 0; s < 4; ++s) b[a][s] = 0.0;
    }
    __device__ __forceinline__ void fold(double w2, double d0, double d1, double d2, const double (&ln)[4]) {
      using MT = MomentsT<3>;
      const double dd[3] = {d0, d1, d2};
      double mono[MT::N], m[20];
      MT::monomials(mono, dd, std::make_integer_sequence<int, MT::N - 1>{});
      Mm[0] += w2;
#pragma unroll
      for (int i = 1; i < MT::N; ++i) Mm[i] += w2 * mono[i];
      MT::firsts(mono, m, std::make_integer_sequence<int, 20>{});
#pragma unroll
      for (int a = 0; a < 20; ++a) {
        const double wa = w2 * m[a];
#pragma unroll
        for (int s = 0; s < 4; ++s) b[a][s] += wa * ln[s];
      }
    }
    template <class F>
    __device__ __forceinline__ void reduce(F sum) {
#pragma unroll
      for (int t = 0; t < MomentsT<3>::N; ++t) Mm[t] = sum(Mm[t]);
#pragma unroll
      for (int a = 0; a < 20; ++a)
#pragma unroll
        for (int s = 0; s < 4; ++s) b[a][s] = sum(b[a][s]);
    }
    __device__ __noinline__ int solve(double fi[4]) const {
      using MT = MomentsT<3>;
      constexpr int J = 20;
      double A[MT::NT], y[J], bb[J][4]; // private memory: indexed by loop variables below
      MT::expand(Mm, A, std::make_integer_sequence<int, MT::NT>{});
#pragma unroll
      for (int a = 0; a < J; ++a)
#pragma unroll
        for (int s = 0; s < 4; ++s) bb[a][s] = b[a][s];
      auto at = [&](int r, int c) -> double & { return A[r * J - r * (r - 1) / 2 + (c - r)]; };
#pragma unroll 1
      for (int j = 0; j < J; ++j) {
        double s = at(j, j);
#pragma unroll 1
        for (int l = 0; l < j; ++l) s -= at(l, j) * at(l, j);
        if (!(s > 0.0)) return 1;
        const double ujj = sqrt(s), inv = 1.0 / ujj;
        at(j, j) = ujj;
#pragma unroll 1
        for (int c = j + 1; c < J; ++c) {
          double t = at(j, c);
#pragma unroll 1
          for (int l = 0; l < j; ++l) t -= at(l, c) * at(l, j);
          at(j, c) = t * inv;
        }
      }
#pragma unroll 1
      for (int i = 0; i < J; ++i) { // U^T z = e_1
        double t = (i == 0) ? 1.0 : 0.0;
#pragma unroll 1
        for (int l = 0; l < i; ++l) t -= at(l, i) * y[l];
        y[i] = t / at(i, i);
      }
#pragma unroll 1
      for (int i = J - 1; i >= 0; --i) { // U y = z
        double t = y[i];
#pragma unroll 1
        for (int l = i + 1; l < J; ++l) t -= at(i, l) * y[l];
        y[i] = t / at(i, i);
      }
      double acc[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll 1
      for (int j = 0; j < J; ++j)
#pragma unroll
        for (int s = 0; s < 4; ++s) acc[s] += y[j] * bb[j][s];
#pragma unroll
      for (int s = 0; s < 4; ++s) fi[s] = acc[s];
      return 0;
    }
  };

  // order 3 (J = 20) per lane: as interpolate<J> below, with the moments standing in for A
  __device__ __noinline__ int interpolate_o3(const double x[3], double fi[4]) const {
    constexpr int J = 20;
    int count = 0;
    double sw = 0.0, swv = 0.0;
    for_neighbours(x, [&](const double *q, double, double, double, double r) {
      double cw = 0.5 + 0.5 * cos(r * 2.0 * PI / radius / 2.0);
      sw += cw;
      swv += cw * q[7];
      ++count;
    });
    fi[0] = fi[1] = fi[2] = fi[3] = 0.0;
    if (count < J) return 2;
    const double hin = lws * (swv / sw);
    Sums20 S;
    int kept = 0;
    bool usemask = true;
    for (int attempt = 0; attempt < 2; ++attempt) {
      S.zero();
      kept = 0;
      for_neighbours(x, [&](const double *q, double d0, double d1, double d2, double r) {
        double e = etainv(r, hin);
        if (usemask && !(e > 1.0e-16)) return; // :316-317
        ++kept;
        const double ln[4] = {q[3], q[4], q[5], q[6]};
        S.fold(0.5 * e, d0, d1, d2, ln);
      });
      if (kept >= J) break;
      usemask = false; // threw out too many samples: use them all (:319-323)
    }
    double f4[4];
    if (S.solve(f4) != 0) return 1;
    fi[0] = f4[0], fi[1] = f4[1], fi[2] = f4[2], fi[3] = f4[3];
    return 0;
  }

  // returns status: 0 ok, 1 solve failed, 2 too few samples (lsinterp_mod.f95:262-264)
  template <int J>
  __device__ __noinline__ int interpolate(const double x[3], double fi[4]) const {
    // pass 1: count, cosine-window-weighted mean of the samples' nearest-neighbour distances (:296-303)
    int count = 0;
    double sw = 0.0, swv = 0.0;
    for_neighbours(x, [&](const double *q, double, double, double, double r) {
      double cw = 0.5 + 0.5 * cos(r * 2.0 * PI / radius / 2.0);
      sw += cw;
      swv += cw * q[7];
      ++count;
    });
    fi[0] = fi[1] = fi[2] = fi[3] = 0.0;
    if (count < J) return 2;
    const double hin = lws * (swv / sw);
    constexpr int NT = J * (J + 1) / 2;
    double A[NT], b[J][4];
    int kept = 0;
    bool usemask = true;
    for (int attempt = 0; attempt < 2; ++attempt) {
#pragma unroll
      for (int t = 0; t < NT; ++t) A[t] = 0.0;
#pragma unroll
      for (int j = 0; j < J; ++j)
#pragma unroll
        for (int s = 0; s < 4; ++s) b[j][s] = 0.0;
      kept = 0;
      for_neighbours(x, [&](const double *q, double d0, double d1, double d2, double r) {
        double e = etainv(r, hin);
        if (usemask && !(e > 1.0e-16)) return; // :316-317
        ++kept;
        double w2 = 0.5 * e;
        double m[J];
        monomials<J>(d0, d1, d2, m);
        int t = 0;
#pragma unroll
        for (int a = 0; a < J; ++a) {
          double wa = w2 * m[a];
#pragma unroll
          for (int c = a; c < J; ++c) A[t++] += wa * m[c];
#pragma unroll
          for (int s = 0; s < 4; ++s) b[a][s] += wa * q[3 + s];
        }
      });
      if (kept >= J) break;
      usemask = false; // threw out too many samples: use them all (:319-323)
    }
    // dposv 'U': A = U^T U, packed upper triangle, row a holds A[a][a..J-1]
    auto at = [&](int r, int c) -> double & { return A[r * J - r * (r - 1) / 2 + (c - r)]; };
#pragma unroll
    for (int j = 0; j < J; ++j) {
      double s = at(j, j);
#pragma unroll
      for (int l = 0; l < j; ++l) s -= at(l, j) * at(l, j);
      if (!(s > 0.0)) return 1;
      double ujj = sqrt(s), inv = 1.0 / ujj;
      at(j, j) = ujj;
#pragma unroll
      for (int c = j + 1; c < J; ++c) {
        double t = at(j, c);
#pragma unroll
        for (int l = 0; l < j; ++l) t -= at(l, c) * at(l, j);
        at(j, c) = t * inv;
      }
    }
    double y[J];
#pragma unroll
    for (int i = 0; i < J; ++i) { // U^T z = e_1
      double t = (i == 0) ? 1.0 : 0.0;
#pragma unroll
      for (int l = 0; l < i; ++l) t -= at(l, i) * y[l];
      y[i] = t / at(i, i);
    }
#pragma unroll
    for (int i = J - 1; i >= 0; --i) { // U y = z
      double t = y[i];
#pragma unroll
      for (int l = i + 1; l < J; ++l) t -= at(i, l) * y[l];
      y[i] = t / at(i, i);
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      double acc = 0.0;
#pragma unroll
      for (int j = 0; j < J; ++j) acc += y[j] * b[j][s];
      fi[s] = acc;
    }
    return 0;
  }

  // ------------------------------------------------------------------------------------------
  // Cooperative evaluation of one lane's whole evalrhs stencil (the trace kernel's path).
  //
  // A per-lane neighbour loop is the wrong shape for a wave: neighbour counts differ by 40x between lanes
  // (tens in the outer cube, >1000 near the Earth) and only ~16 % of the scanned candidates lie inside the radius,
  // so most lanes idle most of the time.  Instead the wave serves ONE owner lane at a time: its <= 8 stencil points
  // (offsets of ~1e-6 |x|, so practically the same neighbour set) go to the 8 groups of 8 lanes; the candidates are
  // compacted into an index list in LDS (ballot + popcount), and the groups walk the dense list for the
  // per-neighbour work (weights, the 55 + 40 normal-equation terms), 8 lanes splitting the samples.  The partial sums
  // of the 8 lanes are combined by DPP within the group.  Per point the terms are those of interpolate<J>; only the
  // order of summation differs (it is RNG-dependent in the reference anyway, SURVEY A-12).
  // LDS of a wave, in doubles (L = LIST_DOUBLES):
  //   [0, L)  the list area -- one shared candidate list of SHARED_CAP entries, or 8 own lists of LIST_CAP, or (pass 2) the
  //           ring of NBUF 64-record buffers;   L staging-buffer pointer;  L + 1 candidate-block pointer;
  //   [L + 2, L + 34) the groups' results on their way to the owner;  [L + 34, L + 50) cycle counters (timing builds);
  //   [L + 64, L + 320) the 64 lanes' candidate-block headers {centre x, y, z, entry count (int; -1 = none)}
#ifndef SRT_SCAT_WAVES
// Waves per SIMD the cooperative kernels of this model are built for: 1 = 34.5 KiB of LDS per wave, all 512 registers, a ring of
// four record buffers; 2 = 18.5 KiB, <= 256 registers (every phase spill-free), two buffers.  Measured at BASELINE config[4]'s
// full size, same box: 18.8 s per launch with 2, 20.1 s with 1.  (Before the staging buffer became chunk-major the second wave
// bought nothing -- 24.3 s against 21.6 s: a CU's memory pipeline takes requests per cache LINE, and 16-byte pieces of 64
// different records per instruction saturated it with four waves already.)
#define SRT_SCAT_WAVES 2
#endif
  static constexpr int WAVES_PER_EU = SRT_SCAT_WAVES;
  static constexpr int NBUF = WAVES_PER_EU == 2 ? 2 : 4;  // 64-record buffers of pass 2's ring
  static constexpr int LIST_DOUBLES = NBUF * 1024;          // the list area: NBUF x 8 KiB
  static constexpr int LIST_CAP = LIST_DOUBLES / 4;         // entries per group (8 groups x 4 bytes)
  static constexpr int LDS_SCRATCH_SLOT = LIST_DOUBLES, LDS_BLOCK_SLOT = LDS_SCRATCH_SLOT + 1, LDS_PARK = LDS_SCRATCH_SLOT + 2,
                       LDS_PHASE = LDS_PARK + 32, LDS_HDR = LDS_SCRATCH_SLOT + 64;
  static constexpr int LDS_DOUBLES = LDS_HDR + 4 * 64;
#ifndef SRT_SCAT_TAYLOR
// 1 (round 4): the seven windows of a sample in pass 1 as a cubic in the angle difference about the centre's window (instead of cos / sin
// series of the difference and the addition theorem), and -- for a stencil all of whose samples are far enough -- the seven weights
// from a second-order expansion of the weight about the centre (sf_weights); 0: round 3's series everywhere.
#define SRT_SCAT_TAYLOR 1
#endif
#ifndef SRT_SCAT_REC_AHEAD
// 1: the weights pass reads the staging records of the samples beyond the LDS side arrays one trip ahead (second-order tier)
#define SRT_SCAT_REC_AHEAD 1
#endif
#ifndef SRT_SCAT_FUSED
// 1: the weights and the pair loop of the shared path run FUSED, tile by tile (sf_fused): the 64 records of a tile --
// {x, y, z, ln N_s, the eight half-weights} -- exist only in LDS, between the lanes that weigh them (lane = sample) and the
// groups that sum them (lane = (point, 1 of 8)).  Nothing of a record goes to device memory any more except {r_c, cos, sin} of
// the samples beyond the LDS side arrays; the sample itself is gathered twice (pass 1, fused pass) from the sample array.
// 0 (default until measured faster): the three-phase form with the staging records written twice and read back through the LDS-DMA ring.
#define SRT_SCAT_FUSED 0
#endif
  static constexpr bool FUSED = SRT_SCAT_FUSED != 0;
  static constexpr int TILE_BYTES = 8192; // 64 records x 128 B, [16-byte chunk][record], at the END of the list area
  // candidate block of a lane: BLOCK_CAP entries {float dx, dy, dz (from the block's centre), int sample index}
  static constexpr int BLOCK_CAP = 4096, BLOCK_DOUBLES = 64 * BLOCK_CAP * 2;
  static constexpr int TRIP_MAX = 9 * 64; // entries one trip of a 27-cell scan can add

  // block == one wave, and the LDS accesses of a wave complete in issue order: ordering LDS writes before other lanes' reads
  // needs no s_barrier and -- unlike __syncthreads(), which waits for vmcnt(0) too -- no wait for the global / scratch stores in
  // flight (the owner's 32 results on their way to scratch at every hand-off: 8 k cycles of a stencil's 95 k sat there); only the
  // compiler must not move LDS accesses across.  NOT for LDS-DMA landings or for global data handed between lanes.
  __device__ __forceinline__ static void wave_lds_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
  template <int CTRL>
  __device__ __forceinline__ static double dpp_move(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
  }
  // sum over the 8 lanes of a group (every lane gets the total): xor 1, xor 2, mirror within 8
  __device__ __forceinline__ static double group_sum(double v) {
    v += dpp_move<0xB1>(v);  // quad_perm [1,0,3,2]
    v += dpp_move<0x4E>(v);  // quad_perm [2,3,0,1]
    v += dpp_move<0x141>(v); // row_half_mirror
    return v;
  }
  // two sums over the 8 lanes of a group at once: the total of `a` lands in the group's lanes 0..3, that of `b` in lanes 4..7
  // (the halves swap what they do not keep through row_half_mirror -- lane i takes lane 7 - i's -- then xor 1, xor 2 within
  // the quad)
  __device__ __forceinline__ static double group_sum2(double a, double b) {
    const bool low = (threadIdx.x & 4) == 0;
    const double keep = low ? a : b, send = low ? b : a;
    double v = keep + dpp_move<0x141>(send);
    v += dpp_move<0xB1>(v);
    v += dpp_move<0x4E>(v);
    return v;
  }
  // Eight sums over the 8 lanes of a group at once, transposed: lane `sub` of the group ends with the group's total of
  // x[rev3(sub)] (rev3: the three bits of sub reversed).  Three halving levels -- the halves / quad halves / neighbours swap what
  // they do not keep (row_half_mirror, quad_perm [2,3,0,1], quad_perm [1,0,3,2]) -- 4 + 2 + 1 exchanges of 7 operations.
  __device__ __forceinline__ static double group_transpose_sum(const double (&x)[8]) {
    const int sub = threadIdx.x & 7;
    const bool b2 = (sub & 4) != 0, b1 = (sub & 2) != 0, b0 = (sub & 1) != 0;
    double y[4], z[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) y[i] = (b2 ? x[2 * i + 1] : x[2 * i]) + dpp_move<0x141>(b2 ? x[2 * i] : x[2 * i + 1]);
#pragma unroll
    for (int i = 0; i < 2; ++i) z[i] = (b1 ? y[2 * i + 1] : y[2 * i]) + dpp_move<0x4E>(b1 ? y[2 * i] : y[2 * i + 1]);
    return (b0 ? z[1] : z[0]) + dpp_move<0xB1>(b0 ? z[0] : z[1]);
  }
  // sum over the 8 groups of what each lane holds (lanes with the same `sub`), every lane gets its total
  __device__ __forceinline__ static double across_groups(double v) {
    v += dpp_move<0x128>(v); // row_ror:8 -- the other group of the row
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    return v;
  }
  __device__ __forceinline__ static int group_sum(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, false);
    return v;
  }

  // Scan the candidate rows of point p; accepted sample indices (|q - p| < radius) are appended to the group's
  // list; whenever a list could overflow, every group runs `body` over what it has and starts over.  Returns true
  // if that happened (the final list is then only the tail).  The caller runs `body` over the final list.
  struct Rows { // the candidate rows of one point: 3 x 3 (z, y) rows of 3 x-adjacent cells (contiguous in the CSR order)
    int cy, cz, x0, x1;
    bool live;
  };
  __device__ __forceinline__ void row_range(const Rows &R, int r, int &lo, int &hi) const {
    const int cz = R.cz + r / 3 - 1, cy = R.cy + r % 3 - 1;
    const bool ok = R.live && cz >= 0 && cz < dims[2] && cy >= 0 && cy < dims[1] && R.x0 <= R.x1;
    // always a valid lookup (no branch around the loads: the nine rows' lookups are in flight together)
    const int czc = cz < 0 ? 0 : (cz >= dims[2] ? dims[2] - 1 : cz), cyc = cy < 0 ? 0 : (cy >= dims[1] ? dims[1] - 1 : cy);
    const int row = (czc * dims[1] + cyc) * dims[0];
    const int x0 = R.x0 < 0 ? 0 : (R.x0 >= dims[0] ? dims[0] - 1 : R.x0), x1 = R.x1 < 0 ? 0 : (R.x1 >= dims[0] ? dims[0] - 1 : R.x1);
    const int l = gcells()[row + x0], h = gcells()[row + x1 + 1];
    lo = ok ? l : 0;
    hi = ok ? h : 0;
  }
  template <class Body>
  __device__ __forceinline__ bool scan_rows(const double (&p)[3], const Rows &R, SRT_LDS int *list, int &n_list,
                                            Body body) const {
    const int lane = threadIdx.x, g = lane >> 3, sub = lane & 7;
    const double r2 = radius * radius;
    bool flushed = false;
    n_list = 0;
#pragma unroll 1
    for (int r = 0; r < 9; ++r) {
      int lo, hi;
      row_range(R, r, lo, hi);
#pragma unroll 1
      for (int i = lo + sub; __any(i < hi); i += 8) {
        bool acc = false;
        if (i < hi) {
          const SRT_AS1 double *q = gxyz() + i;
          double d0 = q[0] - p[0], d1 = q[npts] - p[1], d2 = q[2 * (size_t)npts] - p[2];
          acc = d0 * d0 + d1 * d1 + d2 * d2 < r2;
        }
        const unsigned gm = (unsigned)(__ballot(acc) >> (8 * g)) & 0xffu;
        if (acc) list[n_list + __popc(gm & ((1u << sub) - 1u))] = i;
        n_list += __popc(gm);
        if (__any(n_list > LIST_CAP - 8)) {
          __syncthreads(); // block == one wave: orders the list writes before the reads below
#pragma unroll 1
          for (int k = sub; k < n_list; k += 8) body(list[k]);
          __syncthreads();
          n_list = 0;
          flushed = true;
        }
      }
    }
    __syncthreads();
    return flushed;
  }

  // dposv 'U' on the packed upper triangle (row a holds A[a][a..J-1]) with right-hand side e_1: A = U^T U, U^T z = e_1, U y = z.
  // The reciprocals of the pivots are kept: the two substitutions multiply by them where dtrsv divides (<= 1 ulp per entry,
  // against a summation order that differs from the reference's anyway) -- 2 J divisions less on one dependent chain.
  template <int J>
  __device__ __forceinline__ static int chol_y(double (&A)[J * (J + 1) / 2], double (&y)[J]) {
    auto at = [&](int r, int c) -> double & { return A[r * J - r * (r - 1) / 2 + (c - r)]; };
    double rinv[J];
#pragma unroll
    for (int j = 0; j < J; ++j) {
      double s = at(j, j);
#pragma unroll
      for (int l = 0; l < j; ++l) s -= at(l, j) * at(l, j);
      if (!(s > 0.0)) return 1;
#ifdef SRT_SCAT_RSQ
      double ujj, inv; // (a positive pivot of sums of squares of O(1) offsets: normal range)
      fm::sqrt_and_inv_pos(s, ujj, inv);
#else
      const double ujj = fm::sqrt_pos(s), inv = fdiv(1.0, ujj); // (a positive pivot of sums of squares of O(1) offsets: normal range)
#endif
      at(j, j) = ujj;
      rinv[j] = inv;
#pragma unroll
      for (int c = j + 1; c < J; ++c) {
        double t = at(j, c);
#pragma unroll
        for (int l = 0; l < j; ++l) t -= at(l, c) * at(l, j);
        at(j, c) = t * inv;
      }
    }
#pragma unroll
    for (int i = 0; i < J; ++i) { // U^T z = e_1
      double t = (i == 0) ? 1.0 : 0.0;
#pragma unroll
      for (int l = 0; l < i; ++l) t -= at(l, i) * y[l];
      y[i] = t * rinv[i];
    }
#pragma unroll
    for (int i = J - 1; i >= 0; --i) { // U y = z
      double t = y[i];
#pragma unroll
      for (int l = i + 1; l < J; ++l) t -= at(i, l) * y[l];
      y[i] = t * rinv[i];
    }
    return 0;
  }
  template <int J>
  __device__ __forceinline__ static int solve_fit(double (&A)[J * (J + 1) / 2], const double (&b)[J][4], double fi[4]) {
    double y[J];
    if (chol_y<J>(A, y) != 0) return 1;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      double acc = 0.0;
#pragma unroll
      for (int j = 0; j < J; ++j) acc += y[j] * b[j][s];
      fi[s] = acc;
    }
    return 0;
  }
  // Order 2 on the shared path: the group's 75 totals are parked in LDS (area[0..34] the moments, area[35 + 4 a + s] the
  // right-hand sums) so that no accumulator is alive during the factorisation -- at two waves per SIMD (256 registers) the
  // solve otherwise runs out of scratch memory, one dependent reload after the other.
  template <int... T>
  __device__ __forceinline__ static void expand_parked(SRT_LDS const double *area, double (&A)[55], std::integer_sequence<int, T...>) {
    ((A[T] = area[Moments::P<T>::mom]), ...);
  }
  // (returns the fit of ONE species, `s`: the hand-off to the owner takes species lane & 3 from each lane)
  __device__ __forceinline__ static int solve10_parked(SRT_LDS const double *area, int s, double &fi) {
    double A[55], y[10];
    expand_parked(area, A, std::make_integer_sequence<int, 55>{});
    if (chol_y<10>(A, y) != 0) return 1;
    double acc = 0.0;
#pragma unroll
    for (int j = 0; j < 10; ++j) acc += y[j] * area[35 + 4 * j + s];
    fi = acc;
    return 0;
  }

  // Wave-wide reductions in registers, returned wave-uniform (DPP: within the group of 8, row_mirror for the row of 16,
  // row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3 -- lane 63 then holds the result).
  template <int CTRL, int ROWS>
  __device__ __forceinline__ static double dpp_move_rows(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROWS, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROWS, 0xF, false);
    return __hiloint2double(hi, lo);
  }
  // largest of 64 non-negative numbers (the rows a masked step leaves out see 0), wave-uniform
  __device__ __forceinline__ static double wave_max_nonneg(double v) {
    v = fmax(v, dpp_move<0xB1>(v));
    v = fmax(v, dpp_move<0x4E>(v));
    v = fmax(v, dpp_move<0x141>(v));
    v = fmax(v, dpp_move<0x140>(v));
    v = fmax(v, dpp_move_rows<0x142, 0xA>(v));
    v = fmax(v, dpp_move_rows<0x143, 0xC>(v));
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
  }
  __device__ __forceinline__ static int wave_total(int v) {
    v = group_sum(v);
    v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);
    return __builtin_amdgcn_readlane(v, 63);
  }

  // the LDS list area: 8 own lists of LIST_CAP entries, or one shared list
  struct Fit4 { // ln N_s of a fit, returned in registers
    double v[4];
  };
  static constexpr int SHARED_CAP = 8 * LIST_CAP;
  // ---- the shared-list path: staged neighbour records ------------------------------------------------------------
  // The stencil points 1..6 sit within ~1e-6 |x| of the centre (and the free point 7, the other end-point estimate of
  // the same step, within some tens of metres), so for one sample the arguments of the weight's transcendental functions
  // (cos of the window, x**1.1 and exp of etainv) differ between the eight points by ~1e-6 relative.  The path has three
  // phases, each its own out-of-line function (its own register allocation: at two waves per SIMD there are 256):
  //
  //  sf_pass1    lane = sample (64 per trip): the sample is gathered, r_c, cos a_c, sin a_c at the centre are evaluated
  //              once, the cosine windows of the other points follow by the addition theorem from
  //              dr = r_g - r_c (|pi dr / R| <= 3.2e-3, series to da^4 / da^5: remainder 1e-18), and the eight
  //              window-weighted mean spacings h_g (lsinterp_mod.f95:296-303) are reduced over the wave.  The sample goes
  //              to a per-wave staging buffer in device memory (REC doubles per sample) with r_c, cos, sin.
  //  sf_weights  lane = sample again, now with the h_g known: ALL EIGHT weights of the sample, sharing what they can --
  //              with t_g = r_g^2 - r_c^2 = delta (delta -+ 2 d_a) taken from the offsets directly (no cancellation, no
  //              square root), eps = t_g / r_c^2, dr = (t_g / 2 r_c)(1 - eps/4 + eps^2/8 - 5 eps^3/64 + 7 eps^4/128)
  //              (|eps| <= 2e-3: remainder 1e-15 dr), tau = (1 + dr/(r_c + R eps0)) (h_c/h_g) - 1:
  //                  x_g**1.1 = u_c (1 + tau)**1.1            binomial series to tau^4     (|tau| <= 1e-3: remainder 5e-18)
  //                  exp(-u_g) = E_c exp(-du), du = u_g - u_c  exponential series to du^5   (|du| <= 1.2e-3: remainder 4e-21)
  //                  cos(a_c + da) = ca cos da - sa sin da
  //              i.e. the same numbers as etainv() to within its own rounding (both carry ~u_c * 2^-53 from the rounding
  //              of r).  u_c = a sh with a = (r_c + R eps0)**1.1 and sh = (h_c / 4)**-1.1.  Samples for which the bounds
  //              do not hold (within ~1e3 stencil widths of the centre; exact == 1; no usable centre) are "direct": their
  //              weights come from etainv() itself.  The eight half-weights 0.5 eta (0 = not a neighbour of that point,
  //              or masked, :316-317) replace r_c, cos, sin .. in the record.
  //  sf_sums     lane = (point g, 1 of 8): group g walks the records -- through a ring of 64-record buffers in LDS filled by
  //              LDS-DMA -- and accumulates the normal equations of its point from {x, y, z, ln N_s, weight g}: no
  //              transcendental function, no series and no branch in the loop.
  // Record: [0..2] x y z, [3..6] ln N_s, [7] -, after pass 1: [8] r_c [10] cos a_c [11] sin a_c; after the weights:
  // [8 + g] half-weight at point g (g < 7), [15] at the free point.
  static constexpr int REC = 16, REC_CAP = 4096;
  // The staging buffer of a wave is chunk-major: 16-byte chunk t (0..7) of record k sits at (t * REC_CAP + k) * 16 bytes, so that
  // the 64 lanes of a wave -- 64 consecutive records -- write and read 1 KiB of consecutive bytes per instruction (8 cache
  // lines) instead of one 16-byte piece of 64 different lines: the memory pipeline of a CU takes requests per LINE.
  __device__ __forceinline__ static SRT_AS1 d2_t *chunk(SRT_AS1 double *rec, int t, int k) {
    return (SRT_AS1 d2_t *)rec + ((size_t)t * REC_CAP + (size_t)k);
  }

  __device__ __forceinline__ double etainv_at(double ss, double hin) const { return etainv(sqrt(ss), hin); }

  // what pass 1 hands on through LDS (the park area is free until the hand-off at the end of the stencil)
  struct Pass1Out {
    double hin8[8];
    int cnt8[8];
    double rmin; // the list's smallest distance from the centre (sf_weights' choice of tier)
    int kept8[8]; // written by sf_weights: samples that keep a non-zero weight at point g (the mask rule's count, :316-323)
  };
  __device__ __forceinline__ static SRT_LDS Pass1Out *pass1_out(SRT_LDS const int *list) {
    return (SRT_LDS Pass1Out *)((SRT_LDS double *)const_cast<SRT_LDS int *>(list) + LDS_PARK);
  }
  static_assert(sizeof(Pass1Out) <= 32 * 8, "Pass1Out must fit the park area");

  // What the weights need of a sample -- {x, y}, {z, r_c}, {cos a_c, sin a_c} -- waits for them in the list area's free part
  // behind the list (three arrays of `cap` 16-byte entries) instead of coming back from device memory; samples beyond
  // the capacity are read back from their records.
  struct Side {
    SRT_LDS d2_t *a, *b, *c;
    int cap;
  };
  __device__ __forceinline__ static Side side_of(SRT_LDS const int *list, int n_list) {
    const int used = (n_list + 3) >> 2; // 16-byte units the list occupies
    Side S;
    S.cap = (LIST_DOUBLES / 2 - used) / 3;
    S.a = (SRT_LDS d2_t *)const_cast<SRT_LDS int *>(list) + used;
    S.b = S.a + S.cap;
    S.c = S.b + S.cap;
    return S;
  }
  // The fused form: the list area is [list | {cos, sin} of cap samples | r_c of cap samples | .. | tile]; x, y, z come with the
  // sample's second gather.  A shared list must leave the tile its 8 KiB: FUSED_LIST_CAP entries at most.
  struct Side2 {
    SRT_LDS d2_t *cs;
    SRT_LDS double *rc;
    int cap;
  };
  static constexpr int FUSED_LIST_CAP = (LIST_DOUBLES * 8 - TILE_BYTES) / 4 - 64;
  __device__ __forceinline__ static Side2 side2_of(SRT_LDS const int *list, int n_list) {
    const int used = (n_list + 3) >> 2; // 16-byte units the list occupies
    Side2 S;
    const int room = LIST_DOUBLES * 8 - TILE_BYTES - used * 16;
    S.cap = room > 0 ? room / 24 : 0;
    S.cs = (SRT_LDS d2_t *)const_cast<SRT_LDS int *>(list) + used;
    S.rc = (SRT_LDS double *)(S.cs + S.cap);
    return S;
  }
  __device__ __forceinline__ static SRT_LDS char *tile_of(SRT_LDS const int *list) {
    return (SRT_LDS char *)const_cast<SRT_LDS int *>(list) + (LIST_DOUBLES * 8 - TILE_BYTES);
  }
  // p7near: the free point 7 lies as close to the centre as the six offset points may (<= 1e-3 radius) and takes the
  // addition theorem / the series like them; else its window is evaluated by cos() and its weight by etainv().
  __device__ __forceinline__ void sf_pass1(const double (&p_in)[3], unsigned long long livemask, int npts, int n_list,
                                        SRT_LDS const int *list, double *rec_flat, double dmax6, double d7, bool p7near) const {
    const ScatteredModel M = uniform_copy();
    const double p[3] = {p_in[0], p_in[1], p_in[2]};
    const double radius = M.radius, lws = M.lws;
    SRT_AS1 double *const rec = (SRT_AS1 double *)uni(rec_flat); // (wave-uniform: in scalar registers -- as an argument it arrives in vector registers, was spilled and came back from scratch every trip) // device memory: global loads / stores, not flat ones
    const int lane = threadIdx.x;
    const double r2 = radius * radius;
    const double pi_R = PI / radius;
    double pg[8][3]; // the 8 points, wave-uniform
#pragma unroll
    for (int gg = 0; gg < 8; ++gg)
#pragma unroll
      for (int k = 0; k < 3; ++k) pg[gg][k] = from_lane(p[k], 8 * gg);
    SRT_PHASE_BEGIN(list);
    double s8[8], v8[8];
    int c8[8];
#pragma unroll
    for (int gg = 0; gg < 8; ++gg) {
      s8[gg] = v8[gg] = 0.0;
      c8[gg] = 0;
    }
    bool lv8[8];
#pragma unroll
    for (int gg = 0; gg < 8; ++gg) lv8[gg] = (livemask >> (8 * gg)) & 1ull; // wave-uniform
    const Side side = side_of(list, n_list);
    const Side2 side2 = side2_of(list, n_list);
    // the points' offsets from the centre (as in sf_weights)
    const double reps = radius * 5.0e-16;
    const double da3[3] = {pg[1][0] - pg[0][0], pg[3][1] - pg[0][1], pg[5][2] - pg[0][2]};
    const double mda3[3] = {pg[2][0] - pg[0][0], pg[4][1] - pg[0][1], pg[6][2] - pg[0][2]};
    const double o7[3] = {pg[7][0] - pg[0][0], pg[7][1] - pg[0][1], pg[7][2] - pg[0][2]};
    const double o7sq = o7[0] * o7[0] + o7[1] * o7[1] + o7[2] * o7[2];
    double rmin_l = 1.0e300; // (this lane's samples)
    // (one sample ahead: the next gather is in flight while this sample is worked on; issuing the first one before the points are
    // handed round above was measured: no gain)
    d2_t na = {0.0, 0.0}, nb = na, nc = na, nd = na;
    if (lane < n_list) {
      const SRT_AS1 d2_t *q = (const SRT_AS1 d2_t *)(M.gpts() + (size_t)list[lane] * 8);
      na = q[0], nb = q[1], nc = q[2], nd = q[3];
    }
#pragma unroll 1
    for (int k = lane; k < n_list; k += 64) {
      const d2_t qa = na, qb = nb, qc = nc, qd = nd;
      if (k + 64 < n_list) {
        const SRT_AS1 d2_t *q = (const SRT_AS1 d2_t *)(M.gpts() + (size_t)list[k + 64] * 8);
        na = q[0], nb = q[1], nc = q[2], nd = q[3];
      }
      const double q0 = qa.x, q1 = qa.y, q2 = qb.x, q7 = qd.y;
      double rc, ca, sa;
      const double dc[3] = {q0 - pg[0][0], q1 - pg[0][1], q2 - pg[0][2]};
      const double ssc = dc[0] * dc[0] + dc[1] * dc[1] + dc[2] * dc[2];
      {
        const double ss = ssc;
        rc = fm::sqrt_pos(ss);
        rmin_l = fmin(rmin_l, rc);
        fm::sincos_0pi(rc * pi_R, sa, ca);
        const bool in = lv8[0] && ss < r2;
        const double cw = in ? 0.5 + 0.5 * ca : 0.0;
        s8[0] += cw;
        v8[0] += cw * q7;
        c8[0] += in ? 1 : 0;
      }
      // The other points' windows.  da_g = (r_g - r_c) pi / R from t_g = r_g^2 - r_c^2 = delta (delta -+ 2 d_a), taken from the
      // offsets directly, by the series sf_weights uses (same dr_g there: the windows and the weights of a sample see the same
      // number) -- no square root per point; ROW BY ROW over the points with the order pinned (see sf_weights); chain j is
      // point j + 1, the free point's only where it is live and near.  A sample too close to the centre for the series
      // (|eps| <= 2e-3 needs the offsets within 1e-3 r_c) takes the square roots (per lane; rare).
      const double inv = fdiv(1.0, rc + reps), inv2 = inv * inv, hinv = 0.5 * inv;
      auto chains = [&](auto nq) {
        constexpr int NQ = decltype(nq)::value;
#define SF_ROW(dst, expr)                                                                                              \
  _Pragma("unroll") for (int j = 0; j < NQ; ++j) dst[j] = expr;                                                        \
  _Pragma("unroll") for (int j = 0; j < NQ; ++j) asm volatile("" : "+v"(dst[j]))
        double t[NQ], tin[NQ], eps[NQ], pl[NQ], da[NQ], y[NQ], h[NQ], r[NQ];
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
          if (j < 6) {
            const double dl = (j & 1) ? mda3[j >> 1] : da3[j >> 1];
            const double x = fma(-2.0, dc[j >> 1], dl);
            t[j] = dl * x;
            tin[j] = fma(dl, x, ssc);
          } else {
            t[j] = o7sq - 2.0 * (o7[0] * dc[0] + o7[1] * dc[1] + o7[2] * dc[2]);
            tin[j] = ssc + t[j];
          }
        }
        SF_ROW(eps, t[j] * inv2);
        if constexpr (SRT_SCAT_TAYLOR != 0) { // (to eps^2: the next term, 5 eps^3 / 64 <= 6e-10 of an angle difference <= 3e-3)
          SF_ROW(pl, fma(eps[j], 0.125, -0.25));
        } else {
          SF_ROW(pl, fma(eps[j], 0.0546875, -0.078125));
          SF_ROW(pl, fma(eps[j], pl[j], 0.125));
          SF_ROW(pl, fma(eps[j], pl[j], -0.25));
        }
        SF_ROW(pl, fma(eps[j], pl[j], 1.0));
        SF_ROW(da, t[j] * hinv);
        SF_ROW(da, da[j] * pl[j]);
        SF_ROW(da, da[j] * pi_R);
        if (!(dmax6 * inv <= 1.0e-3) || (NQ == 7 && !(d7 * inv <= 1.0e-3))) { // per lane, rare
#pragma unroll
          for (int j = 0; j < NQ; ++j) {
            const double e0 = q0 - pg[j + 1][0], e1 = q1 - pg[j + 1][1], e2 = q2 - pg[j + 1][2];
            tin[j] = e0 * e0 + e1 * e1 + e2 * e2;
            da[j] = (fm::sqrt_pos(tin[j]) - rc) * pi_R;
          }
        }
        if constexpr (SRT_SCAT_TAYLOR != 0) {
          // the window W(a) = 1/2 + cos(a)/2 at a_c + da as a cubic in da about a_c: derivatives -sin/2, -cos/2, +sin/2;
          // |da| <= pi 1e-3 (the free point; 1e-5 for the six offset points): remainder da^4 / 48 <= 2e-12 (1e-21) --
          // 3 rows instead of 9
          const double w1 = -0.5 * sa, w2 = -0.25 * ca, w3 = (1.0 / 12.0) * sa, w0c = 0.5 + 0.5 * ca;
          SF_ROW(h, fma(w3, da[j], w2));
          SF_ROW(h, fma(h[j], da[j], w1));
          SF_ROW(h, fma(h[j], da[j], w0c));
        } else {
        SF_ROW(y, da[j] * da[j]);
        SF_ROW(h, fma(y[j], 1.0 / 24.0, -0.5));
        SF_ROW(r, fma(y[j], 1.0 / 120.0, -1.0 / 6.0));
        SF_ROW(h, fma(y[j], h[j], 1.0)); // cos da
        SF_ROW(r, fma(y[j], r[j], 1.0));
        SF_ROW(r, da[j] * r[j]); // sin da
        SF_ROW(r, sa * r[j]);
        SF_ROW(h, fma(ca, h[j], -r[j]));
        SF_ROW(h, fma(0.5, h[j], 0.5));
        }
#undef SF_ROW
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
          const bool in = lv8[j + 1] && tin[j] < r2; // strictly inside (kdtree_mod.f95:171)
          const double cw = in ? h[j] : 0.0;
          s8[j + 1] += cw;
          v8[j + 1] += cw * q7;
          c8[j + 1] += in ? 1 : 0;
        }
      };
      if (p7near && lv8[7]) chains(std::integral_constant<int, 7>{}); // (wave-uniform)
      else chains(std::integral_constant<int, 6>{});
      if (npts > 7 && !p7near) { // a far free point: its own cosine
        double d0 = q0 - pg[7][0], d1 = q1 - pg[7][1], d2 = q2 - pg[7][2];
        double ss = d0 * d0 + d1 * d1 + d2 * d2;
        if (lv8[7] && ss < r2) {
          double cw = 0.5 + 0.5 * cos(sqrt(ss) * 2.0 * PI / radius / 2.0);
          s8[7] += cw;
          v8[7] += cw * q7;
          c8[7] += 1;
        }
      }
      if constexpr (FUSED) {
        if (k < side2.cap) {
          side2.cs[k] = d2_t{ca, sa};
          side2.rc[k] = rc;
        } else { // a list longer than the side arrays: through the staging buffer
          *chunk(rec, 4, k) = d2_t{rc, 0.0};
          *chunk(rec, 5, k) = d2_t{ca, sa};
        }
        continue;
      }
      *chunk(rec, 0, k) = qa;
      *chunk(rec, 1, k) = qb;
      *chunk(rec, 2, k) = qc;
      *chunk(rec, 3, k) = d2_t{qd.x, 0.0};
#ifdef SRT_PROBE_EXTRA_STORES // (timing probe: the same 64 bytes once more, into chunks the weights overwrite later)
      *chunk(rec, 6, k) = qa;
      *chunk(rec, 7, k) = qb;
      if (k < side.cap) {
        *chunk(rec, 4, k) = qc;
        *chunk(rec, 5, k) = qd;
      }
#endif
      if (k < side.cap) { // what the weights need waits in LDS ..
        side.a[k] = qa;
        side.b[k] = d2_t{q2, rc};
        side.c[k] = d2_t{ca, sa};
      } else { // .. or, for a list longer than the side arrays, in the record
        *chunk(rec, 4, k) = d2_t{rc, 0.0};
        *chunk(rec, 5, k) = d2_t{ca, sa};
      }
    }
    SRT_PHASE(1);
    SRT_LDS Pass1Out *o = pass1_out(list);
    // The kernel is bound by the vector-instruction issue rate, so the 16 sums are reduced TRANSPOSED (group_transpose_sum:
    // 8 sums -> one register, lane `sub` of every group holding the group's total of sum rev3(sub): 49 operations for 8 sums),
    // then over the 8 groups (row_ror:8 in registers, xor 16 and xor 32 through ds_bpermute, which does not take a vector
    // issue slot): every lane ends with the wave totals of ITS sum, and one division per lane replaces eight.
    int n8[8];
    const double ts = across_groups(group_transpose_sum(s8)), tv = across_groups(group_transpose_sum(v8));
    const double hmine = lws * (tv / ts);
    const int rev3 = ((lane & 1) << 2) | (lane & 2) | ((lane >> 2) & 1);
#pragma unroll
    for (int gg = 0; gg < 8; gg += 2) { // (a lane counts at most REC_CAP / 64 samples, the wave REC_CAP: 16 bits each)
      const int two = wave_total(c8[gg] | (c8[gg + 1] << 16));
      n8[gg] = two & 0xFFFF, n8[gg + 1] = (int)((unsigned)two >> 16);
    }
    if (lane < 8) o->hin8[rev3] = hmine;
    // (smallest distance = 1 / the largest reciprocal; lanes without a sample hold 1e300; the wave maximum is wave-uniform)
    const double rmin_w = fdiv(1.0, wave_max_nonneg(fdiv(1.0, rmin_l)));
    if (lane == 0) {
#pragma unroll
      for (int gg = 0; gg < 8; ++gg) o->cnt8[gg] = n8[gg];
      o->rmin = rmin_w;
    }
    // (block == one wave) the sums and the side arrays written above are read by other lanes next: LDS only -- unless the list is
    // longer than the side arrays, whose rest the weights pass reads back from the records (global memory: the stores must have landed)
    if (n_list > (FUSED ? side2.cap : side.cap)) __syncthreads();
    else wave_lds_sync();
    SRT_PHASE(2);
  }

  // All eight half-weights of every sample of the list (see above); usemask: the reference's weight > 1e-16 mask (:316-317)
  template <int J>
  __device__ __forceinline__ void sf_weights(const double (&p_in)[3], bool live, unsigned long long livemask, int npts, int n_list,
                                          SRT_LDS const int *list, double *rec_flat, double dmax6, double d7, bool p7near,
                                          bool usemask) const {
    const ScatteredModel M = uniform_copy();
    const double p[3] = {p_in[0], p_in[1], p_in[2]};
    const double radius = M.radius;
    SRT_AS1 double *const rec = (SRT_AS1 double *)uni(rec_flat);
    const int lane = threadIdx.x, g = lane >> 3;
    const double r2 = radius * radius, pi_R = PI / radius, reps = radius * 5.0e-16;
    SRT_PHASE_BEGIN(list);
    SRT_LDS Pass1Out *o = pass1_out(list);
    double hin8[8], pg[8][3];
    bool fit8[8], lv8[8];
    int kept_c[8] = {0, 0, 0, 0, 0, 0, 0, 0}; // wave-uniform: samples that keep a non-zero weight at point g
#pragma unroll
    for (int gg = 0; gg < 8; ++gg) {
      hin8[gg] = uni(o->hin8[gg]);
      lv8[gg] = (livemask >> (8 * gg)) & 1ull;
      fit8[gg] = lv8[gg] && gg < npts && uni(o->cnt8[gg]) >= J;
#pragma unroll
      for (int k = 0; k < 3; ++k) pg[gg][k] = from_lane(p[k], 8 * gg);
    }
    // the centre's h against each point's: eta_g = h_c / h_g - 1
    double eta8[8], etamax6 = 0.0;
#pragma unroll
    for (int gg = 0; gg < 8; ++gg) {
      eta8[gg] = fdiv(hin8[0] - hin8[gg], hin8[gg]);
      if (gg < 7 && fit8[gg]) etamax6 = fmax(etamax6, fabs(eta8[gg]));
    }
    const double eta7 = fit8[7] ? fabs(eta8[7]) : 0.0;
    const bool base_ok = M.exact != 1 && uni(o->cnt8[0]) >= 1 && hin8[0] > 0.0 && etamax6 <= 1.0e-3;
    const double sh = base_ok ? fm::exp_any(-1.1 * fm::log_pos(hin8[0] * 0.25)) : 0.0; // (h_c / 4)**-1.1: u_c = a sh
    // the points' offsets from the centre: point 1 + 2a / 2 + 2a = centre +- da e_a, point 7 = centre + (o7x, o7y, o7z)
    const double da3[3] = {pg[1][0] - pg[0][0], pg[3][1] - pg[0][1], pg[5][2] - pg[0][2]};
    const double mda3[3] = {pg[2][0] - pg[0][0], pg[4][1] - pg[0][1], pg[6][2] - pg[0][2]};
    const double o7[3] = {pg[7][0] - pg[0][0], pg[7][1] - pg[0][1], pg[7][2] - pg[0][2]};
    const double o7sq = o7[0] * o7[0] + o7[1] * o7[1] + o7[2] * o7[2];
    const Side side = side_of(list, n_list);
    // ---- second-order tier (round 4, SRT_SCAT_TAYLOR).  The weight of a sample at a point is expanded about the centre in
    // dr = r_g - r_c and dh = h_g - h_c:  w + dr (w_r + w_rh dh + (w_rr / 2) dr) + w_h dh, with (E = exp(-u), u = x**1.1, W the window)
    //   w = E W / 2,  w_r = E (L W + W') / 2,  w_rr = E ((L' + L^2) W + 2 L W' + W'') / 2,  L = -1.1 u / r,  L' = -0.11 u / r^2,
    //   w_h = w 1.1 u / h,  w_rh = w_r 1.1 u / h + w 1.21 u / (h r).
    // (and (w_hh / 2) dh^2, w_hh = w ((1.1 u / h)^2 - 2.31 u / h^2)).
    // Remainder (L dr)^3 / 6 with |L dr| <= 1.1 (u / r) d, and u / r grows like r^0.1: largest at the search radius, whatever the
    // sample.  Taken for a stencil (one decision, wave-uniform) when 1.1 u(R) d / R <= 2e-4 (remainder <= 2e-12 of a weight),
    // 1.1 u(R) |h_c / h_g - 1| <= 2e-4 (its cube / 6: 1e-12) and every sample is at least 1e3 stencil widths away (the dr series'
    // own condition).  Timing builds count the outcome per stencil (srt_tier_stats): at BASELINE config[4] 99.4 % of the stencils
    // take this tier, the rest fail on the free point
    // (pass 1's smallest distance: the dr series' own condition) -- tools/scattered_taylor_prototype.py: ln N at the seven points
    // to 4e-13, its central-difference gradient to 3e-8 median against the exact weights.  11 rows per point instead of 34;
    // every other stencil takes the series below, as in round 3.
    bool tay_all = false;
    double tc1 = 0.0, tc2 = 0.0, ihc = 0.0;
    double dh8[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    if constexpr (SRT_SCAT_TAYLOR != 0) {
      const double rmin = uni(o->rmin);
      if (base_ok && rmin > 0.0) {
        const double irm = fdiv(1.0, rmin + reps), iR = M.inv_radius;
        const double umax = M.u11 * sh; // 1.1 u at the radius (u11: the model's constant 1.1 (radius + reps)**1.1 -- a log + exp chain per stencil before)
        tay_all = dmax6 * irm <= 1.0e-3 && umax * (dmax6 * iR) <= 2.0e-4 && umax * etamax6 <= 2.0e-4 &&
                  (!fit8[7] || (p7near && d7 * irm <= 1.0e-3 && umax * (d7 * iR) <= 2.0e-4 && umax * eta7 <= 2.0e-4));
#ifdef SRT_PHASE_TIMING
        if (lane == 0) {
          const bool f7 = fit8[7] && !(p7near && d7 * irm <= 1.0e-3 && umax * (d7 * iR) <= 2.0e-4 && umax * eta7 <= 2.0e-4);
          atomicAdd(&srt_tier_stats[tay_all ? 1 : (!(dmax6 * irm <= 1.0e-3) ? 3 : (!(umax * (dmax6 * iR) <= 2.0e-4) ? 4 : (!(umax * etamax6 <= 2.0e-4) ? 5 : (f7 ? 6 : 7))))], 1ull);
        }
#endif
      }
#ifdef SRT_PHASE_TIMING
      if (lane == 0) {
        atomicAdd(&srt_tier_stats[0], 1ull);
        if (!(base_ok && rmin > 0.0)) atomicAdd(&srt_tier_stats[2], 1ull);
      }
#endif
      tc1 = -0.5 * pi_R, tc2 = -0.5 * pi_R * pi_R;
      ihc = tay_all ? fdiv(1.0, hin8[0]) : 0.0;
#pragma unroll
      for (int gg = 0; gg < 8; ++gg) dh8[gg] = hin8[gg] - hin8[0];
    }
    auto weigh_taylor = [&](int k, const d2_t s0, const d2_t s1, const d2_t s2) {
      const double q0 = s0.x, q1 = s0.y, q2 = s1.x, rc = s1.y, ca = s2.x, sa = s2.y;
      const double dc[3] = {q0 - pg[0][0], q1 - pg[0][1], q2 - pg[0][2]};
      const double ssc = dc[0] * dc[0] + dc[1] * dc[1] + dc[2] * dc[2];
      const double xr = rc + reps;
      const double a11 = xr * fm::exp_any(0.1 * fm::log_pos(xr));
      const double inv = fdiv(1.0, xr), inv2 = inv * inv, hinv = 0.5 * inv;
      const double u = a11 * sh;
      const double Eh = 0.5 * fm::exp_any(-u), thr = 0.5 * 1.0e-16;
      const double f0 = 0.5 + 0.5 * ca, fp = tc1 * sa, fpp = tc2 * ca;
      const double w0 = Eh * f0;
      const double L1 = -1.1 * u * inv, L2h = -0.055 * u * inv2;
      const double wr = Eh * fma(L1, f0, fp);
      const double hrr = Eh * fma(fma(0.5 * L1, L1, L2h), f0, fma(L1, fp, 0.5 * fpp));
      const double kh = 1.1 * u * ihc;
      const double wh = w0 * kh, wrh = fma(wr, kh, w0 * (1.1 * kh * inv));
      const double hhh = 0.5 * w0 * fma(kh, kh, -2.31 * u * (ihc * ihc)); // w_hh / 2
      double w8[8];
      w8[0] = (fit8[0] && ssc < r2 && (w0 > thr || !usemask)) ? w0 : 0.0;
      w8[7] = 0.0;
      auto chains = [&](auto nq) {
        constexpr int NQ = decltype(nq)::value;
#define SF_ROW(dst, expr)                                                                                              \
  _Pragma("unroll") for (int j = 0; j < NQ; ++j) dst[j] = expr;                                                        \
  _Pragma("unroll") for (int j = 0; j < NQ; ++j) asm volatile("" : "+v"(dst[j]))
        double t[NQ], tin[NQ], au[NQ], dr[NQ], X[NQ];
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
          if (j < 6) {
            const double dl = (j & 1) ? mda3[j >> 1] : da3[j >> 1];
            const double x = fma(-2.0, dc[j >> 1], dl);
            t[j] = dl * x;
            tin[j] = fma(dl, x, ssc);
          } else {
            t[j] = o7sq - 2.0 * (o7[0] * dc[0] + o7[1] * dc[1] + o7[2] * dc[2]);
            tin[j] = ssc + t[j];
          }
        }
        SF_ROW(au, t[j] * inv2); // eps; dr = (t / 2r)(1 - eps/4 + eps^2/8): the next term, 5 eps^3 / 64 <= 6e-10, times |L dr| <= 2e-4
        SF_ROW(X, fma(au[j], 0.125, -0.25));
        SF_ROW(X, fma(au[j], X[j], 1.0));
        SF_ROW(dr, t[j] * hinv);
        SF_ROW(dr, dr[j] * X[j]);
        SF_ROW(au, fma(wrh, dh8[j + 1], wr));
        SF_ROW(au, fma(hrr, dr[j], au[j]));
        SF_ROW(X, fma(hhh, dh8[j + 1], wh));
        SF_ROW(X, fma(X[j], dh8[j + 1], w0));
        SF_ROW(X, fma(dr[j], au[j], X[j]));
#undef SF_ROW
#pragma unroll
        for (int j = 0; j < NQ; ++j) w8[j + 1] = (fit8[j + 1] && tin[j] < r2 && (X[j] > thr || !usemask)) ? X[j] : 0.0;
      };
      if (fit8[7]) chains(std::integral_constant<int, 7>{}); // (wave-uniform)
      else chains(std::integral_constant<int, 6>{});
      *chunk(rec, 4, k) = d2_t{w8[0], w8[1]};
      *chunk(rec, 5, k) = d2_t{w8[2], w8[3]};
      *chunk(rec, 6, k) = d2_t{w8[4], w8[5]};
      *chunk(rec, 7, k) = d2_t{w8[6], w8[7]};
      // (one compare per point and sample here instead of a select, a compare and an add per (point, neighbour) in the pair
      // loop; the count itself is scalar arithmetic)
#pragma unroll
      for (int gg = 0; gg < 8; ++gg) kept_c[gg] += __popcll(__ballot(w8[gg] != 0.0));
    };
    auto weigh = [&](int k, const d2_t s0, const d2_t s1, const d2_t s2) {
      const double q0 = s0.x, q1 = s0.y, q2 = s1.x, rc = s1.y, ca = s2.x, sa = s2.y;
      const double dc[3] = {q0 - pg[0][0], q1 - pg[0][1], q2 - pg[0][2]};
      const double ssc = dc[0] * dc[0] + dc[1] * dc[1] + dc[2] * dc[2];
      const double xr = rc + reps;
      const double a11 = xr * fm::exp_any(0.1 * fm::log_pos(xr));
      const double inv = fdiv(1.0, xr), inv2 = inv * inv, hinv = 0.5 * inv;
      const double u = a11 * sh;
      const double E = fm::exp_any(-u);
      double tb6 = dmax6 * inv, tb7 = d7 * inv;
      tb6 = tb6 + etamax6 * (1.0 + tb6);
      tb7 = tb7 + eta7 * (1.0 + tb7);
      const bool dir6 = !(base_ok && tb6 <= 1.0e-3 && u * 1.2 * tb6 <= 1.0e-3);
      const bool dir7 = dir6 || !(p7near && tb7 <= 1.0e-3 && u * 1.2 * tb7 <= 1.0e-3);
      double w8[8];
      // (HALF-weights from here on: 0.5 E is exact, so 0.5 E X W are the bits of 0.5 (E X W); the reference's mask
      // weight > 1e-16 (:316-317) is the same test on the halves)
      const double Eh = 0.5 * E, thr = 0.5 * 1.0e-16;
      const double w0 = Eh * (0.5 + 0.5 * ca);
      w8[0] = (fit8[0] && ssc < r2 && (w0 > thr || !usemask)) ? w0 : 0.0; // strictly inside (kdtree_mod.f95:171)
      w8[7] = 0.0;
      // The series of the other points, written ROW BY ROW over the points with the order pinned (an empty volatile asm per
      // result, as in igrf_core, srt_device.hpp): each point's ~40 operations are one dependent chain (three Horner forms in
      // a row), an instruction issued right behind its producer waits ~5 ns instead of 2, and left to itself the scheduler
      // emits the chains one after the other.  Chain j is point j + 1; the free point's chain only where it has a fit.
      auto chains = [&](auto nq) {
        constexpr int NQ = decltype(nq)::value;
#define SF_ROW(dst, expr)                                                                                              \
  _Pragma("unroll") for (int j = 0; j < NQ; ++j) dst[j] = expr;                                                        \
  _Pragma("unroll") for (int j = 0; j < NQ; ++j) asm volatile("" : "+v"(dst[j]))
        // t = r_g^2 - r_c^2 from the offset itself
        double t[NQ], tin[NQ], pl[NQ], dr[NQ], au[NQ], X[NQ], cd[NQ], sd[NQ];
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
          if (j < 6) {
            const double dl = (j & 1) ? mda3[j >> 1] : da3[j >> 1];
            const double x = fma(-2.0, dc[j >> 1], dl);
            t[j] = dl * x;
            tin[j] = fma(dl, x, ssc);
          } else {
            t[j] = o7sq - 2.0 * (o7[0] * dc[0] + o7[1] * dc[1] + o7[2] * dc[2]);
            tin[j] = ssc + t[j];
          }
        }
        SF_ROW(au, t[j] * inv2); // eps
        SF_ROW(pl, fma(au[j], 0.0546875, -0.078125));
        SF_ROW(pl, fma(au[j], pl[j], 0.125));
        SF_ROW(pl, fma(au[j], pl[j], -0.25));
        SF_ROW(pl, fma(au[j], pl[j], 1.0));
        SF_ROW(dr, t[j] * hinv); // (= (0.5 t) inv to the bit)
        SF_ROW(dr, dr[j] * pl[j]);
        SF_ROW(au, dr[j] * inv); // tau
        SF_ROW(au, fma(au[j], 1.0 + eta8[j + 1], eta8[j + 1]));
        SF_ROW(pl, fma(au[j], 0.0078375, -0.0165));
        SF_ROW(pl, fma(au[j], pl[j], 0.055));
        SF_ROW(pl, fma(au[j], pl[j], 1.1));
        SF_ROW(au, au[j] * pl[j]); // (1 + tau)**1.1 - 1
        SF_ROW(au, u * au[j]);     // du
        SF_ROW(pl, fma(au[j], -1.0 / 120.0, 1.0 / 24.0));
        SF_ROW(pl, fma(au[j], pl[j], -1.0 / 6.0));
        SF_ROW(pl, fma(au[j], pl[j], 0.5));
        SF_ROW(pl, fma(au[j], pl[j], -1.0));
        SF_ROW(X, fma(au[j], pl[j], 1.0)); // exp(-du)
        SF_ROW(dr, dr[j] * pi_R);          // da
        SF_ROW(au, dr[j] * dr[j]);
        SF_ROW(cd, fma(au[j], 1.0 / 24.0, -0.5));
        SF_ROW(sd, fma(au[j], 1.0 / 120.0, -1.0 / 6.0));
        SF_ROW(cd, fma(au[j], cd[j], 1.0));
        SF_ROW(sd, fma(au[j], sd[j], 1.0));
        SF_ROW(sd, dr[j] * sd[j]);
        SF_ROW(sd, sa * sd[j]);
        SF_ROW(X, Eh * X[j]);
        SF_ROW(cd, fma(ca, cd[j], -sd[j]));
        SF_ROW(cd, fma(cd[j], 0.5, 0.5));
        SF_ROW(X, X[j] * cd[j]);
#undef SF_ROW
#pragma unroll
        for (int j = 0; j < NQ; ++j) w8[j + 1] = (fit8[j + 1] && tin[j] < r2 && (X[j] > thr || !usemask)) ? X[j] : 0.0;
      };
      if (fit8[7]) chains(std::integral_constant<int, 7>{}); // (wave-uniform)
      else chains(std::integral_constant<int, 6>{});
      if (dir6 || (dir7 && fit8[7])) { // rare, per lane: etainv() itself at the points that cannot take the series
#pragma unroll 1
        for (int gg = 0; gg < 8; ++gg) {
          if (gg < 7 && !dir6) continue;
          double px = pg[0][0], py = pg[0][1], pz = pg[0][2], hg = hin8[0];
          bool fg = fit8[0];
#pragma unroll
          for (int t = 1; t < 8; ++t) {
            px = gg == t ? pg[t][0] : px, py = gg == t ? pg[t][1] : py, pz = gg == t ? pg[t][2] : pz;
            hg = gg == t ? hin8[t] : hg;
            fg = gg == t ? fit8[t] : fg;
          }
          const double e0 = q0 - px, e1 = q1 - py, e2 = q2 - pz;
          const double ss = e0 * e0 + e1 * e1 + e2 * e2;
          double e = (fg && ss < r2) ? 0.5 * M.etainv_at(ss, hg) : 0.0;
          e = (usemask && !(e > thr)) ? 0.0 : e; // :316-317
#pragma unroll
          for (int t = 0; t < 8; ++t) w8[t] = gg == t ? e : w8[t];
        }
      }
      *chunk(rec, 4, k) = d2_t{w8[0], w8[1]};
      *chunk(rec, 5, k) = d2_t{w8[2], w8[3]};
      *chunk(rec, 6, k) = d2_t{w8[4], w8[5]};
      *chunk(rec, 7, k) = d2_t{w8[6], w8[7]};
      // (one compare per point and sample here instead of a select, a compare and an add per (point, neighbour) in the pair
      // loop; the count itself is scalar arithmetic)
#pragma unroll
      for (int gg = 0; gg < 8; ++gg) kept_c[gg] += __popcll(__ballot(w8[gg] != 0.0));
    };
    // the samples whose {x, y, z, r_c, cos, sin} wait in LDS: no load from device memory in this loop, so nothing in it ever
    // waits for the previous trip's stores
    const int nlds = n_list < side.cap ? n_list : side.cap;
    if (tay_all) { // (wave-uniform)
#pragma unroll 1
      for (int k = lane; k < nlds; k += 64) weigh_taylor(k, side.a[k], side.b[k], side.c[k]);
#if SRT_SCAT_REC_AHEAD
      // (a list longer than the side arrays -- most are: 423 samples on average against 306 slots: the rest comes back from the staging
      // records, one trip AHEAD, so that a trip's loads are not what its first instruction waits for)
      {
        int k = nlds + lane;
        const d2_t z2 = d2_t{0.0, 0.0};
        d2_t n0 = z2, n1 = z2, n4 = z2, n5 = z2;
        if (k < n_list) n0 = *chunk(rec, 0, k), n1 = *chunk(rec, 1, k), n4 = *chunk(rec, 4, k), n5 = *chunk(rec, 5, k);
#pragma unroll 1
        for (; k < n_list; k += 64) {
          const d2_t c0 = n0, c1 = n1, c4 = n4, c5 = n5;
          const int kn = k + 64;
          if (kn < n_list) n0 = *chunk(rec, 0, kn), n1 = *chunk(rec, 1, kn), n4 = *chunk(rec, 4, kn), n5 = *chunk(rec, 5, kn);
          weigh_taylor(k, c0, d2_t{c1.x, c4.x}, c5);
        }
      }
#else
#pragma unroll 1
      for (int k = nlds + lane; k < n_list; k += 64) {
        const d2_t c0 = *chunk(rec, 0, k), c1 = *chunk(rec, 1, k), c4 = *chunk(rec, 4, k), c5 = *chunk(rec, 5, k);
        weigh_taylor(k, c0, d2_t{c1.x, c4.x}, c5);
      }
#endif
    } else {
#pragma unroll 1
    for (int k = lane; k < nlds; k += 64) weigh(k, side.a[k], side.b[k], side.c[k]);
    // a list longer than the side arrays: the rest back from their records
#pragma unroll 1
    for (int k = nlds + lane; k < n_list; k += 64) {
      const d2_t c0 = *chunk(rec, 0, k), c1 = *chunk(rec, 1, k), c4 = *chunk(rec, 4, k), c5 = *chunk(rec, 5, k);
      weigh(k, c0, d2_t{c1.x, c4.x}, c5);
    }
    }
    // the records behind the list's end, up to the pair loop's last whole 64-record chunk, carry zero weights: the loop tests no index
#pragma unroll 1
    for (int k = n_list + lane; k < ((n_list + 63) & ~63); k += 64) {
      const d2_t z = {0.0, 0.0};
      *chunk(rec, 4, k) = z;
      *chunk(rec, 5, k) = z;
      *chunk(rec, 6, k) = z;
      *chunk(rec, 7, k) = z;
    }
    if (lane == 0) {
#pragma unroll
      for (int gg = 0; gg < 8; ++gg) o->kept8[gg] = kept_c[gg];
    }
    __syncthreads(); // block == one wave: the weights written above are read by other lanes next
    SRT_PHASE(3);
  }

  // Normal equations of this group's point from the finished records, and the solve
  template <int J>
  __device__ __noinline__ Fit4 sf_sums(const double (&p_in)[3], bool fit, int n_list, SRT_LDS const int *list, double *rec_flat,
                                       int &kept_out) const {
    Fit4 fi;
    const double p[3] = {p_in[0], p_in[1], p_in[2]}; // in registers: the asm statements below clobber memory
    SRT_AS1 double *const rec = (SRT_AS1 double *)uni(rec_flat);
    const int lane = threadIdx.x, g = lane >> 3, sub = lane & 7;
    constexpr int NT = J * (J + 1) / 2;
    SRT_PHASE_BEGIN(list);
    constexpr bool O3 = J == 20; // order 3: all sums live in S20 (84 moments + 80), A is only formed inside its solve
    double A[(O3 || J == 10) ? 1 : NT], b[O3 ? 1 : J][4];
    double Mm[J == 10 ? Moments::N : 1]; // order 2: the 35 moments stand in for the 55 entries of A while summing
    typename std::conditional<O3, Sums20, int>::type S20;
    kept_out = pass1_out(list)->kept8[g]; // (counted where the weights were made; the park area is outside the ring)
    // The records come back through a ring of NBUF 64-record buffers in LDS (the list area: the list is dead by now),
    // filled by LDS-DMA NBUF - 1 buffers ahead -- the staging buffers of the CU's waves do not stay in L2 (the scans of
    // the other waves stream through it).
    // Buffer layout: [16-byte chunk t of the record][record] -- DMA instruction t moves chunk t of 64 records (lane =
    // record), and the 8 lanes of a group read 8 neighbouring 16-byte slots (all 8 groups the same ones: broadcast).
    const int nchunk = (n_list + 63) >> 6; // wave-uniform
    SRT_AS3 char *const ring = (SRT_AS3 char *)list;
    auto issue = [&](int c) {
      const int r = c * 64 + lane;
      const int rc = r < n_list ? r : n_list - 1; // (x, y, z, ln N of the last sample behind the end: finite; the weights there are zero)
      SRT_AS3 char *dst = ring + (c % NBUF) * 8192;
#pragma unroll
      for (int t = 0; t < 8; ++t) // (chunk-major staging: instruction t reads 1 KiB of consecutive bytes)
        __builtin_amdgcn_global_load_lds((const SRT_AS1 void *)chunk(rec, t, t < 4 ? rc : r), (SRT_AS3 void *)(dst + 1024 * t), 16, 0, 0);
    };
    const bool any = __any(fit);
    if (any) { // (first thing: the first buffers are on their way while the sums are zeroed)
      for (int c = 0; c < NBUF - 1 && c < nchunk; ++c) issue(c);
    }
    if constexpr (O3) S20.zero();
#pragma unroll
    for (int t = 0; t < ((O3 || J == 10) ? 1 : NT); ++t) A[t] = 0.0;
#pragma unroll
    for (int t = 0; t < (J == 10 ? Moments::N : 1); ++t) Mm[t] = 0.0;
#pragma unroll
    for (int a = 0; a < (O3 ? 1 : J); ++a)
#pragma unroll
      for (int s = 0; s < 4; ++s) b[a][s] = 0.0;
    auto fold = [&](double w2, double d0, double d1, double d2, const double (&ln)[4]) {
      if constexpr (O3) {
        S20.fold(w2, d0, d1, d2, ln);
      } else if constexpr (J == 10) {
        // the weight is folded into the monomials as they are built (w, w x, w x^2 .. : the same 34 products), so the
        // moments take an addition each and the right-hand sums find their w m_a ready made
        // The 15 moments of degree 4 are leaves of the monomial chain (nothing is built from them, the right-hand sums need
        // degree <= 2): each is ONE fused multiply-add of its parent's product into its sum instead of a product and an addition.
        const double dd[3] = {d0, d1, d2};
        double wm[Moments::LOW], m[10];
        wm[0] = w2;
        Moments::monomials_from(wm, dd, std::make_integer_sequence<int, Moments::LOW - 1>{});
#pragma unroll
        for (int i = 0; i < Moments::LOW; ++i) Mm[i] += wm[i];
        Moments::fold_leaves(Mm, wm, dd, std::make_integer_sequence<int, Moments::N - Moments::LOW>{});
        Moments::firsts(wm, m, std::make_integer_sequence<int, 10>{}); // w m_a
#pragma unroll
        for (int a = 0; a < 10; ++a)
#pragma unroll
          for (int s = 0; s < 4; ++s) b[a][s] += m[a] * ln[s];
      } else {
        double m[J];
        monomials<J>(d0, d1, d2, m);
        int t = 0;
#pragma unroll
        for (int a = 0; a < J; ++a) {
          double wa = w2 * m[a];
#pragma unroll
          for (int cI = a; cI < J; ++cI) A[t++] += wa * m[cI];
#pragma unroll
          for (int s = 0; s < 4; ++s) b[a][s] += wa * ln[s];
        }
      }
    };
    {
      const int slot = (g == 7) ? 15 : 8 + g;
      const unsigned ring0 = (unsigned)(unsigned long long)ring + (unsigned)(sub * 16);
      const unsigned slot_off = (unsigned)((slot >> 1) * 1024 + (slot & 1) * 8);
#pragma unroll 1
      for (int c = 0; c < nchunk && any; ++c) {
        if (c + NBUF - 1 < nchunk) issue(c + NBUF - 1); // (into the buffer read on the previous trip: its reads have all returned)
        const int ahead = min(nchunk - 1 - c, NBUF - 1); // buffers that may still be in flight while this one is read
        if (ahead >= 3) InterpModel::wait_vm<24>();
        else if (ahead == 2) InterpModel::wait_vm<16>();
        else if (ahead == 1) InterpModel::wait_vm<8>();
        else InterpModel::wait_vm<0>();
        if (fit) {
          const unsigned cbase = ring0 + (unsigned)((c % NBUF) * 8192);
          // (one record ahead, two register sets in turn: a record's LDS reads are in flight while the previous one is folded
          // in; inline asm: the compiler's wait-count pass would make an LDS load it can see wait for ALL DMA in flight)
          struct RecRegs {
            d2_t c0, c1, c2;
            double ln3, w2;
          };
          auto rd = [&](RecRegs &R, int i) {
            const unsigned ra = cbase + (unsigned)(i * 128);
            asm volatile("ds_read_b128 %0, %5\n\t"
                         "ds_read_b128 %1, %5 offset:1024\n\t"
                         "ds_read_b128 %2, %5 offset:2048\n\t"
                         "ds_read_b64 %3, %5 offset:3072\n\t"
                         "ds_read_b64 %4, %6"
                         : "=&v"(R.c0), "=&v"(R.c1), "=&v"(R.c2), "=&v"(R.ln3), "=&v"(R.w2)
                         : "v"(ra), "v"(ra + slot_off)
                         : "memory");
          };
          auto landed = [&](RecRegs &R) {
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(R.c0), "+v"(R.c1), "+v"(R.c2), "+v"(R.ln3), "+v"(R.w2) : : "memory");
          };
          auto use = [&](const RecRegs &R, int) {
            const double ln[4] = {R.c1.y, R.c2.x, R.c2.y, R.ln3};
            fold(R.w2, R.c0.x - p[0], R.c0.y - p[1], R.c1.x - p[2], ln);
          };
          RecRegs Ra, Rb;
          rd(Ra, 0);
          landed(Ra);
#pragma unroll 1
          for (int i = 0; i < 8; i += 2) {
            rd(Rb, i + 1);
            use(Ra, i);
            landed(Rb);
            rd(Ra, i + 2 < 8 ? i + 2 : 7); // (the last one is read for nothing)
            use(Rb, i + 1);
            landed(Ra);
          }
        }
      }
      InterpModel::wait_vm<0>();
    }
    SRT_PHASE(4);
    // combine the 8 lanes' partial sums and solve
    fi.v[0] = fi.v[1] = fi.v[2] = fi.v[3] = 0.0;
    if (__any(fit)) {
      if constexpr (O3) {
        S20.reduce([](double v) { return group_sum(v); });
        if (fit) {
          double f4[4];
          if (S20.solve(f4) == 0) {
#pragma unroll
            for (int s = 0; s < 4; ++s) fi.v[s] = f4[s];
          }
        }
      } else if constexpr (J == 10) {
        SRT_LDS double *area = (SRT_LDS double *)const_cast<SRT_LDS int *>(list) + 80 * g; // (the ring is dead by now)
        // the 75 sums two at a time (group_sum2: 13 instead of 18 operations per two sums); sum t's total lands in
        // sub-lanes 0..3, sum t + 1's in 4..7, and one lane of each half parks it
        static_assert(Moments::N == 35, "area layout: 35 moments, then 10 x 4 right-hand sums");
        auto sumof = [&](auto ic) -> double {
          constexpr int t = decltype(ic)::value;
          if constexpr (t < 35) return Mm[t];
          else if constexpr (t < 75) return b[(t - 35) >> 2][(t - 35) & 3];
          else return 0.0;
        };
        auto pairs = [&](auto self, auto ic) -> void {
          constexpr int t = decltype(ic)::value;
          if constexpr (t < 75) {
            const double v = group_sum2(sumof(std::integral_constant<int, t>{}), sumof(std::integral_constant<int, t + 1>{}));
            if ((sub & 3) == ((t >> 1) & 3) && t + (sub >> 2) < 75) area[t + (sub >> 2)] = v;
            self(self, std::integral_constant<int, t + 2>{});
          }
        };
        pairs(pairs, std::integral_constant<int, 0>{});
        wave_lds_sync(); // block == one wave: the totals written above are read by the group's other lanes below
        if (fit) {
          double f1;
          if (solve10_parked(area, sub & 3, f1) == 0) fi.v[0] = fi.v[1] = fi.v[2] = fi.v[3] = f1; // (this lane's species only)
        }
        wave_lds_sync(); // (the area is list space again)
      } else {
#pragma unroll
        for (int t = 0; t < NT; ++t) A[t] = group_sum(A[t]);
#pragma unroll
        for (int a = 0; a < J; ++a)
#pragma unroll
          for (int s = 0; s < 4; ++s) b[a][s] = group_sum(b[a][s]);
        if (fit) {
          double f4[4];
          if (solve_fit<J>(A, b, f4) == 0) {
#pragma unroll
            for (int s = 0; s < 4; ++s) fi.v[s] = f4[s];
          }
        }
      }
    }
    SRT_PHASE(5);
    return fi;
  }

  // ---- the fused form of sf_weights + sf_sums (SRT_SCAT_FUSED) -------------------------------------------------------------
  // One pass over the list, 64 samples (a tile) at a time:
  //   weigh   lane = sample: the sample is gathered again from the sample array (one trip ahead of its use), {r_c, cos, sin}
  //           come from pass 1 (LDS side arrays; staging buffer for a long list), the eight half-weights as in sf_weights (the same
  //           series, row by row, the same fallbacks) -- and the finished 128-byte record goes into the tile in LDS;
  //   sum     lane = (point g, 1 of 8): the group folds the tile's records into its normal equations, as sf_sums does from its ring.
  // Records behind the list's end carry zero weights (and the last sample's coordinates: finite), so the sums test no index; the
  // samples a point keeps above the weight mask are counted where the weights are made (one compare per point and sample, the
  // count itself scalar).  The running sums stay in registers across the weighing code: at two waves per SIMD that is 150 of the
  // 256, so the series run over the points in two halves there (SRT_SCAT_CHAIN_SPLIT).
#ifndef SRT_SCAT_CHAIN_SPLIT
#define SRT_SCAT_CHAIN_SPLIT (SRT_SCAT_WAVES == 2)
#endif
  template <int J>
  __device__ __noinline__ Fit4 sf_fused(const double (&p_in)[3], bool live, unsigned long long livemask, int npts, int n_list,
                                        SRT_LDS const int *list, double *rec_flat, double dmax6, double d7, bool p7near,
                                        int &kept_out) const {
    const ScatteredModel M = uniform_copy();
    Fit4 fi;
    const double p[3] = {p_in[0], p_in[1], p_in[2]};
    const double radius = M.radius;
    SRT_AS1 double *const rec = (SRT_AS1 double *)rec_flat;
    const int lane = threadIdx.x, g = lane >> 3, sub = lane & 7;
    const double r2 = radius * radius, pi_R = PI / radius, reps = radius * 5.0e-16;
    constexpr bool usemask = true;
    SRT_PHASE_BEGIN(list);
    SRT_LDS const Pass1Out *o = pass1_out(list);
    double hin8[8], pg[8][3];
    bool fit8[8], lv8[8];
    int kept_c[8] = {0, 0, 0, 0, 0, 0, 0, 0}; // wave-uniform
#pragma unroll
    for (int gg = 0; gg < 8; ++gg) {
      hin8[gg] = uni(o->hin8[gg]);
      lv8[gg] = (livemask >> (8 * gg)) & 1ull;
      fit8[gg] = lv8[gg] && gg < npts && uni(o->cnt8[gg]) >= J;
#pragma unroll
      for (int k = 0; k < 3; ++k) pg[gg][k] = from_lane(p[k], 8 * gg);
    }
    const bool fit = live && o->cnt8[g] >= J;
    double eta8[8], etamax6 = 0.0;
#pragma unroll
    for (int gg = 0; gg < 8; ++gg) {
      eta8[gg] = fdiv(hin8[0] - hin8[gg], hin8[gg]);
      if (gg < 7 && fit8[gg]) etamax6 = fmax(etamax6, fabs(eta8[gg]));
    }
    const double eta7 = fit8[7] ? fabs(eta8[7]) : 0.0;
    const bool base_ok = M.exact != 1 && uni(o->cnt8[0]) >= 1 && hin8[0] > 0.0 && etamax6 <= 1.0e-3;
    const double sh = base_ok ? fm::exp_any(-1.1 * fm::log_pos(hin8[0] * 0.25)) : 0.0; // (h_c / 4)**-1.1: u_c = a sh
    const double da3[3] = {pg[1][0] - pg[0][0], pg[3][1] - pg[0][1], pg[5][2] - pg[0][2]};
    const double mda3[3] = {pg[2][0] - pg[0][0], pg[4][1] - pg[0][1], pg[6][2] - pg[0][2]};
    const double o7[3] = {pg[7][0] - pg[0][0], pg[7][1] - pg[0][1], pg[7][2] - pg[0][2]};
    const double o7sq = o7[0] * o7[0] + o7[1] * o7[1] + o7[2] * o7[2];
    const double pc[3] = {pg[0][0], pg[0][1], pg[0][2]};
    const Side2 side = side2_of(list, n_list);
    // ---- the running sums (as sf_sums)
    constexpr int NT = J * (J + 1) / 2;
    constexpr bool O3 = J == 20;
    double A[(O3 || J == 10) ? 1 : NT], b[O3 ? 1 : J][4];
    double Mm[J == 10 ? Moments::N : 1];
    typename std::conditional<O3, Sums20, int>::type S20;
    if constexpr (O3) S20.zero();
#pragma unroll
    for (int t = 0; t < ((O3 || J == 10) ? 1 : NT); ++t) A[t] = 0.0;
#pragma unroll
    for (int t = 0; t < (J == 10 ? Moments::N : 1); ++t) Mm[t] = 0.0;
#pragma unroll
    for (int a = 0; a < (O3 ? 1 : J); ++a)
#pragma unroll
      for (int s = 0; s < 4; ++s) b[a][s] = 0.0;
    auto fold = [&](double w2, double d0, double d1, double d2, const double (&ln)[4]) {
      if constexpr (O3) {
        S20.fold(w2, d0, d1, d2, ln);
      } else if constexpr (J == 10) {
        const double dd[3] = {d0, d1, d2};
        double wm[Moments::LOW], m[10];
        wm[0] = w2;
        Moments::monomials_from(wm, dd, std::make_integer_sequence<int, Moments::LOW - 1>{});
#pragma unroll
        for (int i = 0; i < Moments::LOW; ++i) Mm[i] += wm[i];
        Moments::fold_leaves(Mm, wm, dd, std::make_integer_sequence<int, Moments::N - Moments::LOW>{});
        Moments::firsts(wm, m, std::make_integer_sequence<int, 10>{}); // w m_a
#pragma unroll
        for (int a = 0; a < 10; ++a)
#pragma unroll
          for (int s = 0; s < 4; ++s) b[a][s] += m[a] * ln[s];
      } else {
        double m[J];
        monomials<J>(d0, d1, d2, m);
        int t = 0;
#pragma unroll
        for (int a = 0; a < J; ++a) {
          double wa = w2 * m[a];
#pragma unroll
          for (int cI = a; cI < J; ++cI) A[t++] += wa * m[cI];
#pragma unroll
          for (int s = 0; s < 4; ++s) b[a][s] += wa * ln[s];
        }
      }
    };
    // ---- the eight half-weights of one sample (sf_weights' arithmetic; w8[] out)
    auto weigh = [&](const double q0, const double q1, const double q2, const double rc, const double ca, const double sa, double (&w8)[8]) {
      const double dc[3] = {q0 - pc[0], q1 - pc[1], q2 - pc[2]};
      const double ssc = dc[0] * dc[0] + dc[1] * dc[1] + dc[2] * dc[2];
      const double xr = rc + reps;
      const double a11 = xr * fm::exp_any(0.1 * fm::log_pos(xr));
      const double inv = fdiv(1.0, xr), inv2 = inv * inv, hinv = 0.5 * inv;
      const double u = a11 * sh;
      const double E = fm::exp_any(-u);
      double tb6 = dmax6 * inv, tb7 = d7 * inv;
      tb6 = tb6 + etamax6 * (1.0 + tb6);
      tb7 = tb7 + eta7 * (1.0 + tb7);
      const bool dir6 = !(base_ok && tb6 <= 1.0e-3 && u * 1.2 * tb6 <= 1.0e-3);
      const bool dir7 = dir6 || !(p7near && tb7 <= 1.0e-3 && u * 1.2 * tb7 <= 1.0e-3);
      const double Eh = 0.5 * E, thr = 0.5 * 1.0e-16;
      const double w0 = Eh * (0.5 + 0.5 * ca);
      w8[0] = (fit8[0] && ssc < r2 && (w0 > thr || !usemask)) ? w0 : 0.0; // strictly inside (kdtree_mod.f95:171)
      w8[7] = 0.0;
      // chains FIRST .. FIRST + NQ - 1 (chain j = point j + 1), row by row with the order pinned (see sf_weights)
      auto chains = [&](auto first, auto nq) {
        constexpr int F0 = decltype(first)::value, NQ = decltype(nq)::value;
#define SF_ROW(dst, expr)                                                                                              \
  _Pragma("unroll") for (int j = 0; j < NQ; ++j) dst[j] = expr;                                                        \
  _Pragma("unroll") for (int j = 0; j < NQ; ++j) asm volatile("" : "+v"(dst[j]))
        double t[NQ], tin[NQ], pl[NQ], dr[NQ], au[NQ], X[NQ], cd[NQ], sd[NQ];
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
          const int jj = F0 + j;
          if (jj < 6) {
            const double dl = (jj & 1) ? mda3[jj >> 1] : da3[jj >> 1];
            const double x = fma(-2.0, dc[jj >> 1], dl);
            t[j] = dl * x;
            tin[j] = fma(dl, x, ssc);
          } else {
            t[j] = o7sq - 2.0 * (o7[0] * dc[0] + o7[1] * dc[1] + o7[2] * dc[2]);
            tin[j] = ssc + t[j];
          }
        }
        SF_ROW(au, t[j] * inv2); // eps
        SF_ROW(pl, fma(au[j], 0.0546875, -0.078125));
        SF_ROW(pl, fma(au[j], pl[j], 0.125));
        SF_ROW(pl, fma(au[j], pl[j], -0.25));
        SF_ROW(pl, fma(au[j], pl[j], 1.0));
        SF_ROW(dr, t[j] * hinv); // (= (0.5 t) inv to the bit)
        SF_ROW(dr, dr[j] * pl[j]);
        SF_ROW(au, dr[j] * inv); // tau
        SF_ROW(au, fma(au[j], 1.0 + eta8[F0 + j + 1], eta8[F0 + j + 1]));
        SF_ROW(pl, fma(au[j], 0.0078375, -0.0165));
        SF_ROW(pl, fma(au[j], pl[j], 0.055));
        SF_ROW(pl, fma(au[j], pl[j], 1.1));
        SF_ROW(au, au[j] * pl[j]); // (1 + tau)**1.1 - 1
        SF_ROW(au, u * au[j]);     // du
        SF_ROW(pl, fma(au[j], -1.0 / 120.0, 1.0 / 24.0));
        SF_ROW(pl, fma(au[j], pl[j], -1.0 / 6.0));
        SF_ROW(pl, fma(au[j], pl[j], 0.5));
        SF_ROW(pl, fma(au[j], pl[j], -1.0));
        SF_ROW(X, fma(au[j], pl[j], 1.0)); // exp(-du)
        SF_ROW(dr, dr[j] * pi_R);          // da
        SF_ROW(au, dr[j] * dr[j]);
        SF_ROW(cd, fma(au[j], 1.0 / 24.0, -0.5));
        SF_ROW(sd, fma(au[j], 1.0 / 120.0, -1.0 / 6.0));
        SF_ROW(cd, fma(au[j], cd[j], 1.0));
        SF_ROW(sd, fma(au[j], sd[j], 1.0));
        SF_ROW(sd, dr[j] * sd[j]);
        SF_ROW(sd, sa * sd[j]);
        SF_ROW(X, Eh * X[j]);
        SF_ROW(cd, fma(ca, cd[j], -sd[j]));
        SF_ROW(cd, fma(cd[j], 0.5, 0.5));
        SF_ROW(X, X[j] * cd[j]);
#undef SF_ROW
#pragma unroll
        for (int j = 0; j < NQ; ++j) w8[F0 + j + 1] = (fit8[F0 + j + 1] && tin[j] < r2 && (X[j] > thr || !usemask)) ? X[j] : 0.0;
      };
      using I0 = std::integral_constant<int, 0>;
      using I3 = std::integral_constant<int, 3>;
      using I4 = std::integral_constant<int, 4>;
      if constexpr (SRT_SCAT_CHAIN_SPLIT != 0) {
        chains(I0{}, I3{});
        if (fit8[7]) chains(I3{}, I4{}); // (wave-uniform)
        else chains(I3{}, I3{});
      } else {
        if (fit8[7]) chains(I0{}, std::integral_constant<int, 7>{}); // (wave-uniform)
        else chains(I0{}, std::integral_constant<int, 6>{});
      }
      if (dir6 || (dir7 && fit8[7])) { // rare, per lane: etainv() itself at the points that cannot take the series
#pragma unroll 1
        for (int gg = 0; gg < 8; ++gg) {
          if (gg < 7 && !dir6) continue;
          double px = pg[0][0], py = pg[0][1], pz = pg[0][2], hg = hin8[0];
          bool fg = fit8[0];
#pragma unroll
          for (int t = 1; t < 8; ++t) {
            px = gg == t ? pg[t][0] : px, py = gg == t ? pg[t][1] : py, pz = gg == t ? pg[t][2] : pz;
            hg = gg == t ? hin8[t] : hg;
            fg = gg == t ? fit8[t] : fg;
          }
          const double e0 = q0 - px, e1 = q1 - py, e2 = q2 - pz;
          const double ss = e0 * e0 + e1 * e1 + e2 * e2;
          double e = (fg && ss < r2) ? 0.5 * M.etainv_at(ss, hg) : 0.0;
          e = (usemask && !(e > thr)) ? 0.0 : e; // :316-317
#pragma unroll
          for (int t = 0; t < 8; ++t) w8[t] = gg == t ? e : w8[t];
        }
      }
    };
    // ---- the pass
    SRT_LDS char *const tile = tile_of(list);
    const int nchunk = (n_list + 63) >> 6; // wave-uniform
    const bool any = __any(fit);
    // (one tile ahead: the next tile's gather is in flight while this one is weighed and summed)
    d2_t na = {0.0, 0.0}, nb = na, nc = na, nd = na, n4 = na, n5 = na;
    auto fetch = [&](int k) {
      const int kc = k < n_list ? k : n_list - 1;
      const SRT_AS1 d2_t *q = (const SRT_AS1 d2_t *)(M.gpts() + (size_t)list[kc] * 8);
      na = q[0], nb = q[1], nc = q[2], nd = q[3];
      if (kc >= side.cap) {
        n4 = *chunk(rec, 4, kc);
        n5 = *chunk(rec, 5, kc);
      }
    };
    if (any && nchunk > 0) fetch(lane);
    const int slot = (g == 7) ? 15 : 8 + g;
    const unsigned tile0 = (unsigned)(unsigned long long)tile + (unsigned)(sub * 16);
    const unsigned slot_off = (unsigned)((slot >> 1) * 1024 + (slot & 1) * 8);
#pragma unroll 1
    for (int c = 0; c < nchunk && any; ++c) {
      const int k = c * 64 + lane;
      const int kc = k < n_list ? k : n_list - 1;
      const d2_t qa = na, qb = nb, qc = nc, qd = nd, q4 = n4, q5 = n5;
      double rc, ca, sa;
      if (kc < side.cap) {
        const d2_t cs = side.cs[kc];
        ca = cs.x, sa = cs.y;
        rc = side.rc[kc];
      } else {
        rc = q4.x, ca = q5.x, sa = q5.y;
      }
      if (c + 1 < nchunk) fetch(k + 64);
      double w8[8];
      weigh(qa.x, qa.y, qb.x, rc, ca, sa, w8);
      if (k >= n_list) { // behind the list's end: no weight at any point
#pragma unroll
        for (int t = 0; t < 8; ++t) w8[t] = 0.0;
      }
#pragma unroll
      for (int gg = 0; gg < 8; ++gg) kept_c[gg] += __popcll(__ballot(w8[gg] != 0.0));
      {
        SRT_LDS d2_t *tl = (SRT_LDS d2_t *)tile + lane;
        tl[0] = qa;
        tl[64] = qb;
        tl[128] = qc;
        tl[192] = d2_t{qd.x, 0.0};
        tl[256] = d2_t{w8[0], w8[1]};
        tl[320] = d2_t{w8[2], w8[3]};
        tl[384] = d2_t{w8[4], w8[5]};
        tl[448] = d2_t{w8[6], w8[7]};
      }
      __syncthreads(); // block == one wave: the tile written above is read by other lanes below
      if (fit) {
        struct RecRegs {
          d2_t c0, c1, c2;
          double ln3, w2;
        };
        auto rd = [&](RecRegs &R, int i) {
          const unsigned ra = tile0 + (unsigned)(i * 128);
          asm volatile("ds_read_b128 %0, %5\n\t"
                       "ds_read_b128 %1, %5 offset:1024\n\t"
                       "ds_read_b128 %2, %5 offset:2048\n\t"
                       "ds_read_b64 %3, %5 offset:3072\n\t"
                       "ds_read_b64 %4, %6"
                       : "=&v"(R.c0), "=&v"(R.c1), "=&v"(R.c2), "=&v"(R.ln3), "=&v"(R.w2)
                       : "v"(ra), "v"(ra + slot_off)
                       : "memory");
        };
        auto landed = [&](RecRegs &R) {
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(R.c0), "+v"(R.c1), "+v"(R.c2), "+v"(R.ln3), "+v"(R.w2) : : "memory");
        };
        auto use = [&](const RecRegs &R) {
          const double ln[4] = {R.c1.y, R.c2.x, R.c2.y, R.ln3};
          fold(R.w2, R.c0.x - p[0], R.c0.y - p[1], R.c1.x - p[2], ln);
        };
        RecRegs Ra, Rb;
        rd(Ra, 0);
        landed(Ra);
#pragma unroll 1
        for (int i = 0; i < 8; i += 2) {
          rd(Rb, i + 1);
          use(Ra);
          landed(Rb);
          rd(Ra, i + 2 < 8 ? i + 2 : 7); // (the last one is read for nothing)
          use(Rb);
          landed(Ra);
        }
      }
      __syncthreads(); // (the tile is rewritten next trip)
    }
    {
      int kk = kept_c[0];
#pragma unroll
      for (int t = 1; t < 8; ++t) kk = g == t ? kept_c[t] : kk;
      kept_out = kk;
    }
    SRT_PHASE(4);
    // combine the 8 lanes' partial sums and solve (as sf_sums)
    fi.v[0] = fi.v[1] = fi.v[2] = fi.v[3] = 0.0;
    if (any) {
      if constexpr (O3) {
        S20.reduce([](double v) { return group_sum(v); });
        if (fit) {
          double f4[4];
          if (S20.solve(f4) == 0) {
#pragma unroll
            for (int s = 0; s < 4; ++s) fi.v[s] = f4[s];
          }
        }
      } else if constexpr (J == 10) {
        SRT_LDS double *area = (SRT_LDS double *)const_cast<SRT_LDS int *>(list) + 80 * g; // (the list and the side arrays are dead by now)
        static_assert(Moments::N == 35, "area layout: 35 moments, then 10 x 4 right-hand sums");
        auto sumof = [&](auto ic) -> double {
          constexpr int t = decltype(ic)::value;
          if constexpr (t < 35) return Mm[t];
          else if constexpr (t < 75) return b[(t - 35) >> 2][(t - 35) & 3];
          else return 0.0;
        };
        auto pairs = [&](auto self, auto ic) -> void {
          constexpr int t = decltype(ic)::value;
          if constexpr (t < 75) {
            const double v = group_sum2(sumof(std::integral_constant<int, t>{}), sumof(std::integral_constant<int, t + 1>{}));
            if ((sub & 3) == ((t >> 1) & 3) && t + (sub >> 2) < 75) area[t + (sub >> 2)] = v;
            self(self, std::integral_constant<int, t + 2>{});
          }
        };
        pairs(pairs, std::integral_constant<int, 0>{});
        __syncthreads(); // block == one wave: the totals written above are read by the group's other lanes below
        if (fit) {
          double f1;
          if (solve10_parked(area, sub & 3, f1) == 0) fi.v[0] = fi.v[1] = fi.v[2] = fi.v[3] = f1; // (this lane's species only)
        }
        __syncthreads(); // (the area is list space again)
      } else {
#pragma unroll
        for (int t = 0; t < NT; ++t) A[t] = group_sum(A[t]);
#pragma unroll
        for (int a = 0; a < J; ++a)
#pragma unroll
          for (int s = 0; s < 4; ++s) b[a][s] = group_sum(b[a][s]);
        if (fit) {
          double f4[4];
          if (solve_fit<J>(A, b, f4) == 0) {
#pragma unroll
            for (int s = 0; s < 4; ++s) fi.v[s] = f4[s];
          }
        }
      }
    }
    SRT_PHASE(5);
    return fi;
  }

  // pass 1 + weights in one out-of-line body (one call, one set of registers saved around it); returns whether this group's
  // fit exists: its point has at least J samples (else status 2: too few samples, lsinterp_mod.f95:262-264)
  template <int J>
  __device__ __noinline__ bool sf_prepare(const double (&p)[3], bool live, unsigned long long livemask, int npts, int n_list,
                                          SRT_LDS const int *list, double *rec, double dmax6, double d7, bool p7near) const {
    sf_pass1(p, livemask, npts, n_list, list, rec, dmax6, d7, p7near);
    if constexpr (!FUSED) sf_weights<J>(p, live, livemask, npts, n_list, list, rec, dmax6, d7, p7near, true);
    return live && pass1_out(list)->cnt8[threadIdx.x >> 3] >= J;
  }
  // `redo`: some point threw out too many samples (fewer than J kept: "use them all", lsinterp_mod.f95:319-323).  The list, the
  // side arrays and pass 1's results are gone by then (the pair loop's ring and the parked totals have overwritten them), so
  // the caller serves this stencil on the own-list path, which carries that rule itself (wave-uniform; rare).
  template <int J>
  __device__ __forceinline__ Fit4 shared_fit(const double (&p)[3], bool live, unsigned long long livemask, int npts, int n_list,
                                             SRT_LDS const int *list, double *rec, double dmax6, double d7, bool p7near,
                                             bool &redo) const {
    if constexpr (FUSED) {
      if (n_list > FUSED_LIST_CAP) { // (wave-uniform) the list leaves the tile no room: the own-list path serves this stencil
        redo = true;
        Fit4 none;
        none.v[0] = none.v[1] = none.v[2] = none.v[3] = 0.0;
        return none;
      }
    }
    const bool fit = sf_prepare<J>(p, live, livemask, npts, n_list, list, rec, dmax6, d7, p7near);
    int kept = 0;
    Fit4 fi;
    if constexpr (FUSED) fi = sf_fused<J>(p, live, livemask, npts, n_list, list, rec, dmax6, d7, p7near, kept);
    else fi = sf_sums<J>(p, fit, n_list, list, rec, kept);
    redo = __any(fit && kept < J);
    return fi;
  }
  // The own-list path: the stencil straddles a grid cell, is too wide for the series, or the shared list would not
  // fit: every group scans the rows of its own point (8 lists of LIST_CAP, processed in pieces if they overflow) and
  // evaluates etainv() per (point, neighbour).
  template <int J>
  __device__ __noinline__ Fit4 own_fit(const double (&p)[3], bool live, const Rows &R, SRT_LDS int *lists) const {
    Fit4 fi;
    const ScatteredModel M = uniform_copy();
    const double radius = M.radius, lws = M.lws;
    const int lane = threadIdx.x, g = lane >> 3, sub = lane & 7;
    const double r2 = radius * radius;
    constexpr int NT = J * (J + 1) / 2;
    SRT_LDS int *list = lists + g * LIST_CAP;
    // ---- pass 1: count, cosine-window-weighted mean of the samples' nearest-neighbour distances (:296-303)
    int count = 0, n_list = 0;
    double sw = 0.0, swv = 0.0;
    auto body1 = [&](int i) {
      const SRT_AS1 double *q = M.gpts() + (size_t)i * 8;
      double d0 = q[0] - p[0], d1 = q[1] - p[1], d2 = q[2] - p[2];
      double r = sqrt(d0 * d0 + d1 * d1 + d2 * d2);
      double cw = 0.5 + 0.5 * cos(r * 2.0 * PI / radius / 2.0);
      sw += cw;
      swv += cw * q[7];
      ++count;
    };
    const bool partial = M.scan_rows(p, R, list, n_list, body1);
#pragma unroll 1
    for (int k = sub; k < n_list; k += 8) body1(list[k]);
    count = group_sum(count);
    sw = group_sum(sw);
    swv = group_sum(swv);
    const bool fit = live && count >= J; // else status 2: too few samples (lsinterp_mod.f95:262-264)
    const double hin = lws * (swv / sw);
    // ---- pass 2: normal equations
    constexpr bool O3 = J == 20; // order 3: all sums live in S20 (84 moments + 80), A is only formed inside its solve
    double A[O3 ? 1 : NT], b[O3 ? 1 : J][4];
    typename std::conditional<O3, Sums20, int>::type S20;
    int kept = 0;
    bool usemask = true, todo = fit;
    auto body2q = [&](const double (&q)[8]) {
      double d0 = q[0] - p[0], d1 = q[1] - p[1], d2 = q[2] - p[2];
      double ss = d0 * d0 + d1 * d1 + d2 * d2;
      double e = M.etainv(sqrt(ss), hin);
      if (usemask && !(e > 1.0e-16)) return; // :316-317
      ++kept;
      double w2 = 0.5 * e;
      if constexpr (O3) {
        const double ln[4] = {q[3], q[4], q[5], q[6]};
        S20.fold(w2, d0, d1, d2, ln);
      } else {
        double m[J];
        monomials<J>(d0, d1, d2, m);
        int t = 0;
#pragma unroll
        for (int a = 0; a < J; ++a) {
          double wa = w2 * m[a];
#pragma unroll
          for (int cI = a; cI < J; ++cI) A[t++] += wa * m[cI];
#pragma unroll
          for (int s = 0; s < 4; ++s) b[a][s] += wa * q[3 + s];
        }
      }
    };
    auto guarded = [&](int i) {
      if (todo) {
        const SRT_AS1 double *q = M.gpts() + (size_t)i * 8;
        double qq[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) qq[t] = q[t];
        body2q(qq);
      }
    };
#pragma unroll 1
    for (int attempt = 0; attempt < 2; ++attempt) {
      if (!__any(todo)) break; // wave-uniform
      if (todo) {
        if constexpr (O3) S20.zero();
#pragma unroll
        for (int t = 0; t < (O3 ? 1 : NT); ++t) A[t] = 0.0;
#pragma unroll
        for (int a = 0; a < (O3 ? 1 : J); ++a)
#pragma unroll
          for (int s = 0; s < 4; ++s) b[a][s] = 0.0;
        kept = 0;
      }
      int nfin = n_list; // the complete list of pass 1, unless it had to be processed in pieces
      if (partial) M.scan_rows(p, R, list, nfin, guarded);
#pragma unroll 1
      for (int k = sub; k < nfin; k += 8) guarded(list[k]);
      const int kept_all = group_sum(kept);
      // threw out too many samples: use them all (:319-323)
      todo = todo && kept_all < J;
      if (todo) usemask = false;
    }
    // combine the 8 lanes' partial sums and solve
    fi.v[0] = fi.v[1] = fi.v[2] = fi.v[3] = 0.0;
    if (__any(fit)) {
      if constexpr (O3) {
        S20.reduce([](double v) { return group_sum(v); });
        if (fit) {
          double f4[4];
          if (S20.solve(f4) == 0) {
#pragma unroll
            for (int s = 0; s < 4; ++s) fi.v[s] = f4[s];
          }
        }
      } else {
#pragma unroll
        for (int t = 0; t < NT; ++t) A[t] = group_sum(A[t]);
#pragma unroll
        for (int a = 0; a < J; ++a)
#pragma unroll
          for (int s = 0; s < 4; ++s) b[a][s] = group_sum(b[a][s]);
        if (fit) {
          double f4[4];
          if (solve_fit<J>(A, b, f4) == 0) {
#pragma unroll
            for (int s = 0; s < 4; ++s) fi.v[s] = f4[s];
          }
        }
      }
    }
    return fi;
  }

  // ---- candidate blocks ----------------------------------------------------------------------------------------
  // A ray moves ~4 % of the search radius per step (BASELINE config[4]), so consecutive stencils of a lane -- the six of
  // one attempt and those of the next steps -- see almost the same samples.  Each lane of the trace kernel therefore owns
  // a candidate block in device memory: every sample within radius * (1 + bmargin) of the block's centre C (fp64 test, made
  // when the block is built by a scan of the 27 cells around C: the grid's cell edge is radius * (1 + bmargin)), stored as
  // {float(q - C), sample index} = 16 bytes.  A stencil whose points all lie within bmargin * radius of C (less a slack for
  // the float32 coordinates) finds its candidates by filtering the block -- one coalesced 16-byte load per candidate and a
  // float32 distance test against radius + stencil extent + slack, a superset of what every point's own exact test
  // (shared_fit, passes 1 and 2) accepts -- instead of scanning the cells: ~1.4 x the neighbour count instead of ~6.5 x.
  // A block is a function of its centre alone; the trace kernel forgets a lane's block when the lane is given a new ray
  // (new_ray_hook), so a ray's arithmetic depends on its own history only.
  struct Block {
    double C[3];
    int count; // -1: none
  };
  __device__ __forceinline__ static Block load_header(SRT_LDS const int *lists, int j) {
    SRT_LDS const double *h = (SRT_LDS const double *)lists + LDS_HDR + 4 * j;
    Block b;
    b.C[0] = h[0], b.C[1] = h[1], b.C[2] = h[2];
    b.count = ((SRT_LDS const int *)(h + 3))[0];
    return b;
  }
  // scan the 27 cells around the centre pc (all lanes) into `blk`; returns the entry count, -1 if the block cannot hold them
  // (out of line: one stencil in twenty rebuilds its block; the nine rows' prefetch registers stay out of coop_stencil's allocation)
  __device__ __noinline__ int build_block(const double (&pc)[3], const Rows &Rc, SRT_AS1 f4_t *blk) const {
    const ScatteredModel M = uniform_copy();
    const int lane = threadIdx.x;
    const double Rb = M.radius * (1.0 + M.bmargin), Rb2 = Rb * Rb;
    int lo9[9], hi9[9], maxlen = 0;
#pragma unroll
    for (int r = 0; r < 9; ++r) {
      M.row_range(Rc, r, lo9[r], hi9[r]);
      maxlen = max(maxlen, hi9[r] - lo9[r]);
    }
    const SRT_AS1 double *xs = M.gxyz(), *ys = xs + M.npts, *zs = ys + M.npts;
    int idxn[9];
    double qxn[9], qyn[9], qzn[9];
    auto fetch = [&](int t0) {
#pragma unroll
      for (int r = 0; r < 9; ++r) {
        const int i = lo9[r] + t0 + lane;
        idxn[r] = i < hi9[r] ? i : -1;
        const int ic = idxn[r] < 0 ? 0 : idxn[r];
        qxn[r] = xs[ic], qyn[r] = ys[ic], qzn[r] = zs[ic];
      }
    };
    int n = 0;
    if (maxlen > 0) fetch(0);
#pragma unroll 1
    for (int t0 = 0; t0 < maxlen; t0 += 64) {
      if (n > BLOCK_CAP - TRIP_MAX) return -1;
      int idx[9];
      double qx[9], qy[9], qz[9];
#pragma unroll
      for (int r = 0; r < 9; ++r) idx[r] = idxn[r], qx[r] = qxn[r], qy[r] = qyn[r], qz[r] = qzn[r];
      if (t0 + 64 < maxlen) fetch(t0 + 64);
#pragma unroll
      for (int r = 0; r < 9; ++r) {
        const double d0 = qx[r] - pc[0], d1 = qy[r] - pc[1], d2 = qz[r] - pc[2];
        const bool acc = idx[r] >= 0 && d0 * d0 + d1 * d1 + d2 * d2 < Rb2;
        const unsigned long long m = __ballot(acc);
        if (acc) blk[n + __popcll(m & ((1ull << lane) - 1ull))] = f4_t{(float)d0, (float)d1, (float)d2, __int_as_float(idx[r])};
        n = __builtin_amdgcn_readfirstlane(n + __popcll(m));
      }
    }
    return n;
  }
  // candidates of a stencil centred at pc from a block: entries within rs of pc (float32 test, rs includes the slack)
  __device__ __forceinline__ int filter_block(const Block &B, const double (&pc)[3], double rs, const SRT_AS1 f4_t *blk,
                                              SRT_LDS int *lists) const {
    const int lane = threadIdx.x;
    const float px = (float)(pc[0] - B.C[0]), py = (float)(pc[1] - B.C[1]), pz = (float)(pc[2] - B.C[2]);
    const float rs2 = (float)rs * (float)rs;
    int n_list = 0;
    const int last = B.count - 1;
    // (the whole of a typical block -- ~1.4 x the neighbour count, a few hundred entries -- is in flight at once)
#pragma unroll 1
    for (int t0 = 0; t0 < B.count; t0 += 1024) {
      f4_t e[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int i = t0 + 64 * u + lane;
        e[u] = blk[i < last ? i : last];
      }
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        if (t0 + 64 * u > last) break; // wave-uniform
        const int i = t0 + 64 * u + lane;
        const float d0 = e[u].x - px, d1 = e[u].y - py, d2 = e[u].z - pz;
        const bool acc = i <= last && d0 * d0 + d1 * d1 + d2 * d2 < rs2;
        const unsigned long long m = __ballot(acc);
        if (acc) lists[n_list + __popcll(m & ((1ull << lane) - 1ull))] = __float_as_int(e[u].w);
        n_list = __builtin_amdgcn_readfirstlane(n_list + __popcll(m));
      }
    }
    return n_list;
  }

  // Candidates of a stencil centred at pc by a scan of the centre's 27 cells (all lanes; rs = radius widened by the largest
  // distance of a stencil point from the centre: a superset of every point's neighbour set, each point applies its own exact
  // test later).  All nine candidate rows advance together, 64 samples of each per trip: 27 coalesced loads in flight at once, no
  // index arithmetic beyond row start + lane.  List order: trip, row, lane.  Returns the list length, -1 if the list would not
  // fit the list area or the staging buffer (a trip adds at most TRIP_MAX entries: tested once per trip).
  __device__ __noinline__ int scan_cells(const double (&pc)[3], double rs, const Rows &Rc, SRT_LDS int *lists) const {
    const ScatteredModel M = uniform_copy();
    const int lane = threadIdx.x;
    const double rs2 = rs * rs * (1.0 + 1.0e-12);
    int lo9[9], hi9[9], maxlen = 0;
#pragma unroll
    for (int r = 0; r < 9; ++r) {
      M.row_range(Rc, r, lo9[r], hi9[r]);
      maxlen = max(maxlen, hi9[r] - lo9[r]);
    }
    const SRT_AS1 double *xs = M.gxyz(), *ys = xs + M.npts, *zs = ys + M.npts;
    // (one trip ahead: the next 27 loads are in flight while this trip's samples are tested and compacted)
    int idxn[9];
    double qxn[9], qyn[9], qzn[9];
    auto fetch = [&](int t0) {
#pragma unroll
      for (int r = 0; r < 9; ++r) {
        const int i = lo9[r] + t0 + lane;
        idxn[r] = i < hi9[r] ? i : -1;
        const int ic = idxn[r] < 0 ? 0 : idxn[r];
        qxn[r] = xs[ic], qyn[r] = ys[ic], qzn[r] = zs[ic];
      }
    };
    int n_list = 0;
    if (maxlen > 0) fetch(0);
#pragma unroll 1
    for (int t0 = 0; t0 < maxlen; t0 += 64) {
      if (n_list > (SHARED_CAP < REC_CAP ? SHARED_CAP : REC_CAP) - TRIP_MAX) return -1;
      int idx[9];
      double qx[9], qy[9], qz[9];
#pragma unroll
      for (int r = 0; r < 9; ++r) idx[r] = idxn[r], qx[r] = qxn[r], qy[r] = qyn[r], qz[r] = qzn[r];
      if (t0 + 64 < maxlen) fetch(t0 + 64);
#pragma unroll
      for (int r = 0; r < 9; ++r) {
        const double d0 = qx[r] - pc[0], d1 = qy[r] - pc[1], d2 = qz[r] - pc[2];
        const bool acc = idx[r] >= 0 && d0 * d0 + d1 * d1 + d2 * d2 < rs2;
        const unsigned long long m = __ballot(acc);
        if (acc) lists[n_list + __popcll(m & ((1ull << lane) - 1ull))] = idx[r];
        n_list = __builtin_amdgcn_readfirstlane(n_list + __popcll(m));
      }
    }
    __syncthreads(); // block == one wave: orders the list writes before the reads of the caller
    return n_list;
  }

  // ------------------------------------------------------------------------------------------
  // Orders >= 4: the reference leaves its tables there and generates the exponent triples (lsinterp_mod.f95:114-164, 273-281);
  // J = 35 monomials at order 4, 56 at order 5.  Nobody ships an input that asks for them, so this path is built to ANSWER, not
  // to be fast: the whole wave fits ONE point at a time.  The J (J + 1) / 2 entries of E^T E and the 4 J right-hand sums are
  // dealt out over the lanes (entry p to lane p % 64: T = 13 / 29 accumulators per lane, in registers); the candidates of the 27 cells
  // are weighed 64 at a time (lane = candidate), and for every sample that stays the wave lays its row {dinv m_0 .. dinv m_J-1,
  // dinv ln N_0..3} into LDS -- lane a computes monomial a -- from where each lane reads the two factors of each of its entries.
  // The products are the reference's (E = dinv * monomial, A = sum E_a E_b); the sums run in cell order.  dposv 'U' and the two
  // substitutions then run in LDS with lane c on column c, every sum over l in the reference's order.
  // LDS: area[0, NT + 4 J) the matrix and the right-hand sums (<= 1 820 doubles), area[1888, 1888 + J) y, area[1984, 2048) the row.
  static constexpr int GEN_MAXORDER = 5, GEN_Y = 1888, GEN_ROW = 1984;
  __device__ __forceinline__ static double wave_sum(double v) { return across_groups(group_sum(v)); }
  template <int N>
  __device__ __forceinline__ static double pick(const double (&tab)[N], int e) {
    double v = tab[0];
#pragma unroll
    for (int k = 1; k < N; ++k) v = (e == k) ? tab[k] : v;
    return v;
  }
  // fi[0..3] = ln N_s at the wave-uniform point pt; all 64 lanes call together.  A failed fit gives zeros (the reference's fi = 0).
  // (Point and result go through the caller's private arrays on purpose: a call that touches none of its caller's stack is marked
  // `tail`, which switches the no-callee-saved-registers convention off for the callee -- 24-32 registers saved to AGPRs then, and the
  // two-waves-per-SIMD kernels that call this function fall to one wave.)
  template <int T>
  __device__ __noinline__ void gen_point(const double *pt, SRT_LDS double *area, double *fi) const {
    const ScatteredModel M = uniform_copy();
    const double p0 = uni(pt[0]), p1 = uni(pt[1]), p2 = uni(pt[2]);
    const int lane = threadIdx.x;
    const int order = M.order, J = (order + 1) * (order + 2) * (order + 3) / 6, NT = J * (J + 1) / 2, NE = NT + 4 * J;
    const double r2 = M.radius * M.radius;
#pragma unroll
    for (int s = 0; s < 4; ++s) fi[s] = 0.0;
    Rows R;
    {
      const double pp[3] = {p0, p1, p2};
      int cc[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        double t = floor((pp[k] - M.origin[k]) * M.inv_cell);
        t = fmin(fmax(t, -2.0), (double)M.dims[k] + 1.0);
        cc[k] = uni((int)t);
      }
      R.cy = cc[1];
      R.cz = cc[2];
      R.x0 = cc[0] - 1 < 0 ? 0 : cc[0] - 1;
      R.x1 = cc[0] + 1 >= M.dims[0] ? M.dims[0] - 1 : cc[0] + 1;
      R.live = true;
    }
    // ---- pass 1 (:296-303): how many samples, and the window-weighted mean of their spacings
    int count = 0;
    double sw = 0.0, swv = 0.0;
#pragma unroll 1
    for (int r = 0; r < 9; ++r) {
      int lo, hi;
      M.row_range(R, r, lo, hi);
#pragma unroll 1
      for (int i = lo + lane; i < hi; i += 64) {
        const SRT_AS1 double *q = M.gpts() + (size_t)i * 8;
        const double d0 = q[0] - p0, d1 = q[1] - p1, d2 = q[2] - p2;
        const double ss = d0 * d0 + d1 * d1 + d2 * d2;
        if (ss < r2) {
          const double cw = 0.5 + 0.5 * cos(sqrt(ss) * 2.0 * PI / M.radius / 2.0);
          sw += cw;
          swv += cw * q[7];
          ++count;
        }
      }
    }
    count = wave_total(count);
    if (count < J) return; // status 2
    sw = wave_sum(sw);
    swv = wave_sum(swv);
    const double hin = M.lws * (swv / sw);
    // ---- pass 2 (:316-323): how many stay above the weight mask; fewer than J: all of them are used
    int kept = 0;
#pragma unroll 1
    for (int r = 0; r < 9; ++r) {
      int lo, hi;
      M.row_range(R, r, lo, hi);
#pragma unroll 1
      for (int i = lo + lane; i < hi; i += 64) {
        const SRT_AS1 double *q = M.gpts() + (size_t)i * 8;
        const double d0 = q[0] - p0, d1 = q[1] - p1, d2 = q[2] - p2;
        const double ss = d0 * d0 + d1 * d1 + d2 * d2;
        if (ss < r2 && M.etainv(sqrt(ss), hin) > 1.0e-16) ++kept;
      }
    }
    kept = wave_total(kept);
    const bool usemask = kept >= J;
    // ---- this lane's entries (p = lane + 64 t) as the LDS slots of their two factors, and this lane's monomial
    int ij[T];
    {
      int row = 0, rem = lane; // entry p of the packed upper triangle = (row, row + rem)
#pragma unroll
      for (int t = 0; t < T; ++t) {
        const int p = lane + 64 * t;
        while (row < J && rem >= J - row) {
          rem -= J - row;
          ++row;
        }
        int ia = row, ib = row + rem;
        if (p >= NT) {
          const int q = p - NT;
          ia = q >> 2;
          ib = J + (q & 3);
        }
        if (p >= NE) ia = ib = 0;
        ij[t] = ia | (ib << 8);
        rem += 64;
      }
    }
    int ex = 0, ey = 0, ez = 0;
    {
      int n = 0;
#pragma unroll 1
      for (int a = 0; a <= order; ++a)
#pragma unroll 1
        for (int b = 0; a + b <= order; ++b)
#pragma unroll 1
          for (int c = 0; a + b + c <= order; ++c) {
            if (n == lane) ex = a, ey = b, ez = c;
            ++n;
          }
    }
    double acc[T];
#pragma unroll
    for (int t = 0; t < T; ++t) acc[t] = 0.0;
    SRT_LDS double *row = area + GEN_ROW;
    wave_lds_sync();
    // ---- pass 3: the sums
#pragma unroll 1
    for (int r = 0; r < 9; ++r) {
      int lo, hi;
      M.row_range(R, r, lo, hi);
#pragma unroll 1
      for (int base = lo; base < hi; base += 64) {
        const int i = base + lane;
        bool use = false;
        double d0 = 0.0, d1 = 0.0, d2 = 0.0, dinv = 0.0, f0 = 0.0, f1 = 0.0, f2 = 0.0, f3 = 0.0;
        if (i < hi) {
          const SRT_AS1 double *q = M.gpts() + (size_t)i * 8;
          d0 = q[0] - p0, d1 = q[1] - p1, d2 = q[2] - p2;
          const double ss = d0 * d0 + d1 * d1 + d2 * d2;
          if (ss < r2) {
            const double e = M.etainv(sqrt(ss), hin);
            use = !usemask || e > 1.0e-16;
            dinv = sqrt(0.5 * e); // :331-341 (scaled = 0)
            f0 = q[3], f1 = q[4], f2 = q[5], f3 = q[6];
          }
        }
        unsigned long long todo = __ballot(use);
#pragma unroll 1
        while (todo != 0ull) {
          const int l = __builtin_ctzll(todo);
          todo &= todo - 1ull;
          const double D0 = from_lane(d0, l), D1 = from_lane(d1, l), D2 = from_lane(d2, l), W = from_lane(dinv, l);
          double X[GEN_MAXORDER + 1], Y[GEN_MAXORDER + 1], Z[GEN_MAXORDER + 1];
          X[0] = Y[0] = Z[0] = 1.0;
#pragma unroll
          for (int k = 1; k <= GEN_MAXORDER; ++k) X[k] = X[k - 1] * D0, Y[k] = Y[k - 1] * D1, Z[k] = Z[k - 1] * D2;
          double val = W * ((pick(X, ex) * pick(Y, ey)) * pick(Z, ez)); // E(i, a) = dinv * monomial (:343-353)
          const int s = lane - J;
          const double fs = s == 0 ? from_lane(f0, l) : (s == 1 ? from_lane(f1, l) : (s == 2 ? from_lane(f2, l) : from_lane(f3, l)));
          val = s >= 0 ? W * fs : val;
          if (lane < J + 4) row[lane] = val;
          wave_lds_sync();
#pragma unroll
          for (int t = 0; t < T; ++t) {
            acc[t] = fma(row[ij[t] & 0xFF], row[ij[t] >> 8], acc[t]);
            // (eight entries' reads in flight at a time: left alone, the scheduler hoists all 2 T of them, and this function -- which the
            // two-waves-per-SIMD trace kernels call -- must stay inside 256 registers)
            if ((t & 7) == 7) asm volatile("" ::: "memory");
          }
          wave_lds_sync();
        }
      }
    }
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const int p = lane + 64 * t;
      if (p < NE) area[p] = acc[t];
    }
    wave_lds_sync();
    // ---- dposv 'U' (dpotf2 + dpotrs) on the packed upper triangle; lane c owns column c
    auto at = [&](int rr, int c) -> SRT_LDS double & { return area[rr * J - rr * (rr - 1) / 2 + (c - rr)]; };
    const int c = lane;
#pragma unroll 1
    for (int j = 0; j < J; ++j) {
      double t = 0.0;
      if (c >= j && c < J) {
#pragma unroll 1
        for (int l = 0; l < j; ++l) t = t + at(l, c) * at(l, j);
        t = at(j, c) - t;
      }
      const double ajj = from_lane(t, j);
      if (!(ajj > 0.0)) return; // status 1
      const double u = sqrt(ajj), inv = 1.0 / u;
      if (c == j) at(j, c) = u;
      else if (c > j && c < J) at(j, c) = t * inv;
      wave_lds_sync();
    }
    SRT_LDS double *y = area + GEN_Y;
#pragma unroll 1
    for (int i = 0; i < J; ++i) { // U^T z = e_1 (every lane the same sums: the reads are broadcasts)
      double t = (i == 0) ? 1.0 : 0.0;
#pragma unroll 1
      for (int l = 0; l < i; ++l) t = t - at(l, i) * y[l];
      t = t / at(i, i);
      if (lane == 0) y[i] = t;
      wave_lds_sync();
    }
#pragma unroll 1
    for (int i = J - 1; i >= 0; --i) { // U y = z
      double t = y[i];
#pragma unroll 1
      for (int l = i + 1; l < J; ++l) t = t - at(i, l) * y[l];
      t = t / at(i, i);
      if (lane == 0) y[i] = t;
      wave_lds_sync();
    }
    double fs = 0.0; // lane s < 4: species s
#pragma unroll 1
    for (int j = 0; j < J; ++j) fs = fs + y[j] * area[NT + 4 * j + (lane & 3)];
#pragma unroll
    for (int s = 0; s < 4; ++s) fi[s] = from_lane(fs, s);
    wave_lds_sync();
  }
  __device__ __forceinline__ Fit4 gen_fit(double p0, double p1, double p2, SRT_LDS double *area) const {
    double pt[3] = {p0, p1, p2}, fi[4];
    if (order == 4) gen_point<13>(pt, area, fi);
    else gen_point<29>(pt, area, fi);
    Fit4 out;
#pragma unroll
    for (int s = 0; s < 4; ++s) out.v[s] = fi[s];
    return out;
  }
  // densities of the stencils of the lanes with need == true, one point after the other (out as coop_stencil's)
  __device__ __noinline__ void gen_stencil(const double *c, const double *d, const double *extra, int npts, bool need, double *out,
                                           SRT_LDS double *area) const {
    const int lane = threadIdx.x;
    const int ns = uni(nspec);
    const unsigned long long needmask = __ballot(need);
#pragma unroll 1
    for (int j = 0; j < 64; ++j) {
      if (!((needmask >> j) & 1ull)) continue; // wave-uniform
      double oc[3], od[3], oe[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        oc[k] = from_lane(c[k], j);
        od[k] = from_lane(d[k], j);
        oe[k] = (npts > 7) ? from_lane(extra[k], j) : 0.0;
      }
#pragma unroll 1
      for (int g = 0; g < npts; ++g) {
        double p[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          double v = oc[k];
          v = (g == 1 + 2 * k) ? oc[k] + od[k] : v;
          v = (g == 2 + 2 * k) ? oc[k] - od[k] : v;
          v = (g == 7) ? oe[k] : v;
          p[k] = v;
        }
        const bool outside = p[0] * p[0] + p[1] * p[1] + p[2] * p[2] > R_E * R_E; // (:307-309)
        Fit4 fi;
#pragma unroll
        for (int s = 0; s < 4; ++s) fi.v[s] = 0.0;
        if (outside) fi = gen_fit(p[0], p[1], p[2], area);
        if (lane == j) {
#pragma unroll
          for (int s = 0; s < 4; ++s) out[4 * g + s] = (outside && s < ns) ? exp(fi.v[s]) : 0.0;
        }
      }
    }
  }

  __device__ __noinline__ static double exp_handoff(double y) { return fm::exp_any(y); }
  // out[8][4] (per lane): densities at the lane's stencil points 0..npts-1 (point 7 = extra).  All 64 lanes call
  // together; lanes with need == false are not served (their out is left untouched).
  //
  // Per owner lane: its <= 8 points go to the 8 groups of 8 lanes.  Normally the six offsets are tiny against the radius:
  // then ONE candidate list for the whole stencil (radius widened by the stencil's extent) is made in LDS -- from the
  // lane's candidate block (trace kernel), else by a scan of the centre's 27 cells by all 64 lanes when the points share a
  // grid cell -- and shared_fit() does the rest; otherwise own_fit().
  template <int J>
  __device__ __noinline__ void coop_stencil(const double *c, const double *d, const double *extra, int npts, bool need,
                                            double *out, SRT_LDS int *lists) const {
    const ScatteredModel M = uniform_copy();
    const double radius = M.radius;
    const int nspec = M.nspec;
    const int lane = threadIdx.x, g = lane >> 3;
    const unsigned long long needmask = __ballot(need);
    double *const rec = (double *)((SRT_LDS unsigned long long *)lists)[LDS_SCRATCH_SLOT]; // nullptr: no staging buffer
    SRT_AS1 f4_t *const blocks = (SRT_AS1 f4_t *)((SRT_LDS unsigned long long *)lists)[LDS_BLOCK_SLOT]; // nullptr: none
    // c, d, extra and out are arrays in the CALLER's private memory (density_stencil): addressed as such -- scratch loads / stores --
    // not through the generic pointers they arrive as.  A flat access counts in lgkmcnt too, so the LDS-only syncs of the hand-off
    // (s_waitcnt lgkmcnt(0)) waited for the owner's 32 result stores to complete after all.
    const SRT_PRIV double *const cpv = (const SRT_PRIV double *)c, *const dpv = (const SRT_PRIV double *)d, *const epv = (const SRT_PRIV double *)extra;
    SRT_PRIV double *const outp = (SRT_PRIV double *)out;
#pragma unroll 1
    for (int j = 0; j < 64; ++j) {
      if (!((needmask >> j) & 1ull)) continue; // wave-uniform
      // the owner's stencil, and this group's point of it
      double p[3];
      {
        // this lane's stencil, all nine numbers fetched TOGETHER: c, d, extra point into the caller's private memory (flat loads),
        // and read where they are used -- each behind the previous one's v_readlane -- they were six round trips in a row per
        // stencil.  (Kept in registers across the owner loop instead they are spilled and come back one by one: measured in the
        // listing, 161 scratch operations against 140.)
        double mc[3], md[3], me[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) mc[k] = cpv[k], md[k] = dpv[k], me[k] = epv[k];
#pragma unroll
        for (int k = 0; k < 3; ++k) asm volatile("" : "+v"(mc[k]), "+v"(md[k]), "+v"(me[k]));
        double oc[3], od[3], oe[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          oc[k] = from_lane(mc[k], j); // (j is wave-uniform: v_readlane, no LDS round trip)
          od[k] = from_lane(md[k], j);
          oe[k] = (npts > 7) ? from_lane(me[k], j) : 0.0;
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          double v = oc[k];
          v = (g == 1 + 2 * k) ? oc[k] + od[k] : v;
          v = (g == 2 + 2 * k) ? oc[k] - od[k] : v;
          v = (g == 7) ? oe[k] : v;
          p[k] = v;
        }
      }
      const bool live = g < npts && (p[0] * p[0] + p[1] * p[1] + p[2] * p[2] > R_E * R_E);
      Rows R;
      int cx;
      {
        int cc[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          double t = floor((p[k] - M.origin[k]) * M.inv_cell);
          t = fmin(fmax(t, -2.0), (double)M.dims[k] + 1.0);
          cc[k] = (int)t;
        }
        cx = cc[0];
        R.cy = cc[1];
        R.cz = cc[2];
        R.x0 = cc[0] - 1 < 0 ? 0 : cc[0] - 1;
        R.x1 = cc[0] + 1 >= M.dims[0] ? M.dims[0] - 1 : cc[0] + 1;
        R.live = live;
      }
      const unsigned long long livemask = __ballot(live);
      if (livemask == 0ull) { // the whole stencil is inside the Earth
        if (lane == j)
          for (int t = 0; t < 4 * npts; ++t) outp[t] = 0.0;
        continue;
      }
      SRT_PHASE_BEGIN(lists);
      SRT_PHASE_COUNT(8);
      // ---- candidates of the shared path
      double pc[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) pc[k] = from_lane(p[k], 0);
      // distance of this group's point from the centre; dmax: of the six offset points, d7: of the free point
      const double e2own = (g < npts) ? (p[0] - pc[0]) * (p[0] - pc[0]) + (p[1] - pc[1]) * (p[1] - pc[1]) + (p[2] - pc[2]) * (p[2] - pc[2]) : 0.0;
      double e2s = (g < 7) ? e2own : 0.0;
      e2s = wave_max_nonneg(e2s);
      const double e27 = from_lane(e2own, 56); // 0 when there is no free point
      const double dmax = sqrt(e2s), d7 = sqrt(e27), ext = fmax(dmax, d7);
      const bool p7near = npts > 7 && d7 <= 1.0e-3 * radius;
      const bool geom = rec != nullptr && dmax <= 1.0e-3 * radius;
      bool shared = false;
      int n_list = 0;
      SRT_PHASE_ADD(14, (unsigned long long)d7);
      if (geom && blocks != nullptr) {
        // ---- from the owner's candidate block (rebuilt about this centre if the stencil has left it)
        Block B = load_header(lists, j);
        const double Rb = radius * (1.0 + M.bmargin), slack = 2.0e-6 * Rb;
        const double dC = sqrt((pc[0] - B.C[0]) * (pc[0] - B.C[0]) + (pc[1] - B.C[1]) * (pc[1] - B.C[1]) + (pc[2] - B.C[2]) * (pc[2] - B.C[2]));
        SRT_AS1 f4_t *blk = blocks + (size_t)j * BLOCK_CAP;
        if (!(B.count >= 0 && dC + ext + slack <= M.bmargin * radius)) {
          Rows Rc;
          Rc.cy = __builtin_amdgcn_readlane(R.cy, 0);
          Rc.cz = __builtin_amdgcn_readlane(R.cz, 0);
          Rc.x0 = __builtin_amdgcn_readlane(R.x0, 0);
          Rc.x1 = __builtin_amdgcn_readlane(R.x1, 0);
          Rc.live = true;
          B.count = M.build_block(pc, Rc, blk);
#pragma unroll
          for (int k = 0; k < 3; ++k) B.C[k] = pc[k];
          if (lane == 0) {
            SRT_LDS double *h = (SRT_LDS double *)lists + LDS_HDR + 4 * j;
            h[0] = pc[0], h[1] = pc[1], h[2] = pc[2];
            ((SRT_LDS int *)(h + 3))[0] = B.count;
          }
          __syncthreads(); // block == one wave: the entries written above are read by other lanes below
          SRT_PHASE_COUNT(11);
          SRT_PHASE(15); // (timing builds: slot 11 counts the rebuilt blocks, slot 15 their cycles; the filter's are in slot 0)
        }
        SRT_PHASE_ADD(12, B.count > 0 ? B.count : 0);
        if (B.count >= 0) {
          n_list = M.filter_block(B, pc, radius + ext + slack, blk, lists);
          shared = true;
          wave_lds_sync(); // orders the list writes before the reads below
        }
      }
      if (!shared && geom &&
          !__any(g < npts && (cx != __builtin_amdgcn_readlane(cx, 0) || R.cy != __builtin_amdgcn_readlane(R.cy, 0) ||
                              R.cz != __builtin_amdgcn_readlane(R.cz, 0)))) {
        // ---- no block (layered kernels; block overflow): scan the centre's 27 cells
        Rows Rc;
        Rc.cy = __builtin_amdgcn_readlane(R.cy, 0);
        Rc.cz = __builtin_amdgcn_readlane(R.cz, 0);
        Rc.x0 = __builtin_amdgcn_readlane(R.x0, 0);
        Rc.x1 = __builtin_amdgcn_readlane(R.x1, 0);
        Rc.live = true;
        n_list = M.scan_cells(pc, radius + ext, Rc, lists);
        shared = n_list >= 0; // (-1: would not fit the list area / the staging buffer: every group scans for itself instead)
        n_list = shared ? n_list : 0;
      }
      SRT_PHASE(0);
      SRT_PHASE_ADD(10, n_list);
      Fit4 fi;
      bool own = !shared;
      if (shared) fi = shared_fit<J>(p, live, livemask, npts, n_list, lists, rec, dmax, d7, p7near, own);
      if (own) {
        __syncthreads(); // (after a shared attempt: its LDS traffic is over before the lists are rebuilt)
        fi = own_fit<J>(p, live, R, lists);
      }
      SRT_PHASE(shared ? 9 : 7);
      // hand the results to the owner through LDS (behind the lists and the buffer pointers): the group leaders
      // park their four densities, the owner collects the 8 x 4
      {
        SRT_LDS d2_t *park = (SRT_LDS d2_t *)((SRT_LDS double *)lists + LDS_PARK);
        // (lane (g, sub < 4) hands over species sub: one exponential per lane; sf_sums<10> fills only that one)
        const int sm = lane & 3;
        const double fm_ = sm == 0 ? fi.v[0] : (sm == 1 ? fi.v[1] : (sm == 2 ? fi.v[2] : fi.v[3]));
        // (out of line on purpose: inlined here, the exponential's eleven constants were hoisted out of the owner loop, spilled, and came
        // back from scratch one at a time inside its dependent chain -- eight round trips per stencil, most of the hand-off's 7 k cycles)
        const double val = (live && sm < nspec) ? exp_handoff(fm_) : 0.0; // failed fit: fi = 0 -> Ns = 1
        if ((lane & 7) < 4) ((SRT_LDS double *)park)[4 * g + sm] = val;
        wave_lds_sync();
        if (lane == j) {
#pragma unroll
          for (int gg = 0; gg < 8; ++gg) {
            const d2_t a = park[2 * gg], b = park[2 * gg + 1];
            if (gg < npts) {
              outp[gg * 4 + 0] = a.x;
              outp[gg * 4 + 1] = a.y;
              outp[gg * 4 + 2] = b.x;
              outp[gg * 4 + 3] = b.y;
            }
          }
        }
      }
      wave_lds_sync(); // the lists are reused by the next owner
      SRT_PHASE(6);
    }
    SRT_PHASE_FLUSH(lists);
  }

  __device__ __forceinline__ void dens_point(const double x[3], double Ns[4]) const {
    if (x[0] * x[0] + x[1] * x[1] + x[2] * x[2] > R_E * R_E) {
      double fi[4];
      if (order == 0) interpolate<1>(x, fi);
      else if (order == 1) interpolate<4>(x, fi);
      else if (order == 2) interpolate<10>(x, fi);
      else interpolate_o3(x, fi);
#pragma unroll
      for (int s = 0; s < 4; ++s) Ns[s] = (s < nspec) ? exp(fi[s]) : 0.0; // failed fit: fi = 0 -> Ns = 1
    } else {
#pragma unroll
      for (int s = 0; s < 4; ++s) Ns[s] = 0.0; // inside the Earth (scattered_..adapter.f95:307-309)
    }
  }

  template <int NE>
  __device__ __forceinline__ void density_stencil(const double c[3], const double d[3], const double *extra,
                                                  double (&Ns)[7 + NE][4], double *lds, bool need = true) const {
    // trace / gradient / RK-step kernels: the wave serves its lanes one at a time (coop_stencil).  Those kernels always pass
    // their LDS (USE_LDS): there is no per-lane form of the stencil here, so that they carry none of interpolate<J>()'s code
    // and registers.
    double out[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) out[i] = 0.0;
    double cc[3] = {c[0], c[1], c[2]}, dd[3] = {d[0], d[1], d[2]};
    double ee[3] = {NE ? extra[0] : 0.0, NE ? extra[1] : 0.0, NE ? extra[2] : 0.0};
    SRT_LDS int *lists = (SRT_LDS int *)lds;
    if (order > 3) gen_stencil(cc, dd, ee, 7 + NE, need, out, (SRT_LDS double *)lds);
    else if (order == 0) coop_stencil<1>(cc, dd, ee, 7 + NE, need, out, lists);
    else if (order == 1) coop_stencil<4>(cc, dd, ee, 7 + NE, need, out, lists);
    else if (order == 2) coop_stencil<10>(cc, dd, ee, 7 + NE, need, out, lists);
    else coop_stencil<20>(cc, dd, ee, 7 + NE, need, out, lists);
#pragma unroll
    for (int i = 0; i < 7 + NE; ++i)
#pragma unroll
      for (int s = 0; s < 4; ++s) Ns[i][s] = out[i * 4 + s];
  }

  template <int NP>
  __device__ __forceinline__ void density(const double (&p)[NP][3], double (&Ns)[NP][4], double *lds) const {
    if (order > 3) { // wave-uniform; all 64 lanes are here (the layered kernels clamp their item index)
      const int lane = threadIdx.x;
      const int ns = uni(nspec);
#pragma unroll 1
      for (int i = 0; i < NP; ++i)
#pragma unroll 1
        for (int j = 0; j < 64; ++j) {
          double x[3];
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            double v = p[0][k];
#pragma unroll
            for (int q = 1; q < NP; ++q) v = (i == q) ? p[q][k] : v;
            x[k] = from_lane(v, j);
          }
          const bool outside = x[0] * x[0] + x[1] * x[1] + x[2] * x[2] > R_E * R_E;
          Fit4 fi;
#pragma unroll
          for (int s = 0; s < 4; ++s) fi.v[s] = 0.0;
          if (outside) fi = gen_fit(x[0], x[1], x[2], (SRT_LDS double *)lds);
          if (lane == j) {
#pragma unroll
            for (int q = 0; q < NP; ++q)
#pragma unroll
              for (int s = 0; s < 4; ++s)
                if (q == i) Ns[q][s] = (outside && s < ns) ? exp(fi.v[s]) : 0.0;
          }
        }
      return;
    }
#pragma unroll
    for (int i = 0; i < NP; ++i) { // static indices; the interpolator itself is one out-of-line copy per order
      double x[3] = {p[i][0], p[i][1], p[i][2]}, n4[4];
      dens_point(x, n4);
#pragma unroll
      for (int s = 0; s < 4; ++s) Ns[i][s] = n4[s];
    }
  }
};

// The wave's slices of the launch's staging buffer (ScatteredModel::REC_CAP records per one-wave block) and of its
// candidate blocks (BLOCK_DOUBLES per one-wave block), parked in LDS behind the lists for coop_stencil; nullptr = none
// (no staging: every stencil takes the own-list path; no blocks: every shared-path stencil scans the cells).
__device__ __forceinline__ void bind_scratch(const ScatteredModel &, double *lds, double *scratch, double *scratch2 = nullptr) {
  if (lds == nullptr) return;
  double *mine = scratch ? scratch + (size_t)blockIdx.x * ScatteredModel::REC_CAP * ScatteredModel::REC : nullptr;
  double *blks = scratch2 ? scratch2 + (size_t)blockIdx.x * ScatteredModel::BLOCK_DOUBLES : nullptr;
  ((SRT_LDS unsigned long long *)lds)[ScatteredModel::LDS_SCRATCH_SLOT] = (unsigned long long)mine;
  ((SRT_LDS unsigned long long *)lds)[ScatteredModel::LDS_BLOCK_SLOT] = (unsigned long long)blks;
  ((SRT_LDS int *)((SRT_LDS double *)lds + ScatteredModel::LDS_HDR + 4 * threadIdx.x + 3))[0] = -1;
  SRT_PHASE_ZERO((SRT_LDS double *)lds);
  __syncthreads();
}
__device__ __forceinline__ void new_ray_hook(const ScatteredModel &, double *lds, bool fresh) {
  if (lds != nullptr && fresh) ((SRT_LDS int *)((SRT_LDS double *)lds + ScatteredModel::LDS_HDR + 4 * threadIdx.x + 3))[0] = -1;
}
template <>
struct WaveBudget<ScatteredModel> {
  static constexpr int LDS_DOUBLES = ScatteredModel::LDS_DOUBLES;
  static constexpr int WAVES_PER_EU = ScatteredModel::WAVES_PER_EU;
};

} // namespace srt
