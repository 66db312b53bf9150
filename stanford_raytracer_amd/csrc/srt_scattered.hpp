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

// Build with -DSRT_PHASE_TIMING for a cycle breakdown of coop_stencil (srt_phase_cycles[], printed after every trace
// launch when the environment variable SRT_PHASE_TIMING is set): 0 scan, 1 pass 1, 2 its reductions, 3 base pass,
// 4 pass 2, 5 combine + solve, 6 hand-off, 7 own-list path, 8 stencils served.
#ifdef SRT_PHASE_TIMING
__device__ unsigned long long srt_phase_cycles[16];
// per-wave counters in LDS (behind the lists, the staging pointer and the hand-off area), flushed to the global ones
// once per coop_stencil call: an atomic per phase and stencil would itself show up in whatever waits on memory next
#define SRT_PHASE_LDS ((SRT_LDS unsigned long long *)srt_lds_base_ + (32768 + 512) / 8)
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
static_assert(Moments::N == 35 && Moments::J == 10 && MomentsT<3>::N == 84 && MomentsT<3>::J == 20, "moment counts");
static_assert(Moments::of_m(4) == mom::find(0, 1, 1) && Moments::of_m(9) == mom::find(2, 0, 0) && MomentsT<3>::of_m(19) == mom::find(3, 0, 0) &&
                  MomentsT<3>::of_m(12) == mom::find(1, 0, 2),
              "tabular_monomials order");

struct ScatteredModel {
  const double *pts;     // [npts][8]: x, y, z, lnN[4], nearest-sample distance
  const double *xyz;     // [3][npts]: the positions once more, SoA (candidate scans: 24 coalesced bytes per sample, not 64)
  const int *cell_start; // [ncells + 1]
  double origin[3], inv_cell, radius, lws;
  int dims[3];
  int nspec, order, exact, npts;

  // The out-of-line functions below receive `this` in vector registers, so every member access through it would be a
  // per-lane flat load followed by a full wait (the cell_start lookups of one stencil: nine of them, one after the
  // other).  The model is wave-uniform: a copy fetched through the scalar cache (address from lane 0) lives in SGPRs.
  __device__ __forceinline__ static int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
  __device__ __forceinline__ static double uni(double v) {
    return __hiloint2double(uni(__double2hiint(v)), uni(__double2loint(v)));
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
  static constexpr int LIST_CAP = 1024; // entries per group: 8 groups x 4 KiB = 32 KiB of the wave's LDS

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

  template <int J>
  __device__ __forceinline__ static int solve_fit(double (&A)[J * (J + 1) / 2], const double (&b)[J][4], double fi[4]) {
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

  // sum over all 64 lanes (every lane gets the total)
  __device__ __forceinline__ static double wave_sum(double v) {
    v = group_sum(v);
    v += __shfl_xor(v, 8);
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    return v;
  }

  // the LDS list area: 8 own lists of LIST_CAP entries, or one shared list
  struct Fit4 { // ln N_s of a fit, returned in registers
    double v[4];
  };
  static constexpr int SHARED_CAP = 8 * LIST_CAP;
  // ---- the shared-list path: staged neighbour records ------------------------------------------------------------
  // The stencil points 1..6 sit within ~1e-6 |x| of the centre, so for one sample the arguments of the weight's
  // transcendental functions (cos of the window, x**1.1 and exp of etainv) differ between those points by ~1e-6
  // relative.  They are therefore evaluated ONCE per sample, at the centre, by a lane that owns the sample (64 samples
  // per trip, no accumulators alive), and parked with the sample in a per-wave staging buffer in device memory
  // (REC doubles per sample).  Pass 2 then gets the weight at its own point from the centre's values by short series in
  // the differences -- with dr = r_g - r_c, tau = (1 + dr/(r_c + R eps)) (h_c/h_g) - 1:
  //     x_g**1.1 = u_c (1 + tau)**1.1            binomial series to tau^5        (|tau| <= 1e-3: remainder 3e-21)
  //     exp(-u_g) = E_c exp(-du), du = u_g - u_c  exponential series to du^6      (|du|  <= 1.2e-3: remainder 7e-25)
  //     cos(a_c + da) = ca cos da - sa sin da     da = pi dr / R <= 3.2e-3, series to da^4 / da^5 (remainder 1e-18)
  // i.e. the same numbers as etainv() to within its own rounding (both carry ~u_c * 2^-53 from the rounding of r).
  // Samples for which the bounds do not hold (a sample within ~1e3 stencil widths of the centre; exact == 1; no usable
  // centre) are "direct": the owning lane evaluates etainv() itself for the points 0..6 and parks the seven weights.
  // The free point 7 (the other end-point estimate of the step, up to maxerr |x| away) is always direct.
  // Record: [0..2] x y z, [3..6] ln N_s, [7] 0 = series / 1 = direct, series: [8] r_c [9] 1/(r_c + R eps) [10] cos a_c
  // [11] sin a_c [12] u_c [13] E_c; direct: [8 + g] weight at point g; [15] weight at point 7.
  static constexpr int REC = 16, REC_CAP = 4096;
  static constexpr int LDS_SCRATCH_SLOT = SHARED_CAP / 2; // (in doubles) the wave's staging-buffer pointer sits behind the lists

  __device__ __forceinline__ double etainv_at(double ss, double hin) const { return etainv(sqrt(ss), hin); }

  template <int J>
  __device__ __noinline__ Fit4 shared_fit(const double (&p_in)[3], bool live, unsigned long long livemask, int npts, int n_list,
                                          SRT_LDS const int *list, double *rec_flat, double dmax) const {
    Fit4 fi;
    const ScatteredModel M = uniform_copy();
    const double p[3] = {p_in[0], p_in[1], p_in[2]}; // in registers: the asm statements of pass 2 clobber memory
    const double radius = M.radius, lws = M.lws;
    const int exact = M.exact;
    SRT_AS1 double *const rec = (SRT_AS1 double *)rec_flat; // device memory: global loads / stores, not flat ones
    const int lane = threadIdx.x, g = lane >> 3, sub = lane & 7;
    const double r2 = radius * radius;
    const double pi_R = PI / radius;
    constexpr int NT = J * (J + 1) / 2;
    double pg[8][3]; // the 8 points, wave-uniform
#pragma unroll
    for (int gg = 0; gg < 8; ++gg)
#pragma unroll
      for (int k = 0; k < 3; ++k) pg[gg][k] = __shfl(p[k], 8 * gg);
    SRT_PHASE_BEGIN(list);
    // ---- pass 1: count, cosine-window-weighted mean of the samples' nearest-neighbour distances (:296-303), for all
    // 8 points from each sample; the sample and the centre's r, cos, sin go to the staging buffer
    double hin8[8];
    int cnt8[8];
    {
      double s8[8], v8[8];
      int c8[8];
#pragma unroll
      for (int gg = 0; gg < 8; ++gg) {
        s8[gg] = v8[gg] = 0.0;
        c8[gg] = 0;
      }
      // (one sample ahead: the next gather is in flight while this sample is worked on)
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
        {
          double d0 = q0 - pg[0][0], d1 = q1 - pg[0][1], d2 = q2 - pg[0][2];
          double ss = d0 * d0 + d1 * d1 + d2 * d2;
          rc = sqrt(ss);
          sincos(rc * 2.0 * PI / radius / 2.0, &sa, &ca);
          if ((livemask & 1ull) && ss < r2) {
            double cw = 0.5 + 0.5 * ca;
            s8[0] += cw;
            v8[0] += cw * q7;
            c8[0] += 1;
          }
        }
#pragma unroll
        for (int gg = 1; gg < 7; ++gg) {
          double d0 = q0 - pg[gg][0], d1 = q1 - pg[gg][1], d2 = q2 - pg[gg][2];
          double ss = d0 * d0 + d1 * d1 + d2 * d2;
          if (((livemask >> (8 * gg)) & 1ull) && ss < r2) {
            double da = (sqrt(ss) - rc) * pi_R, da2 = da * da;
            double cd = 1.0 + da2 * (-0.5 + da2 * (1.0 / 24.0));
            double sd = da * (1.0 + da2 * (-1.0 / 6.0 + da2 * (1.0 / 120.0)));
            double cw = 0.5 + 0.5 * (ca * cd - sa * sd);
            s8[gg] += cw;
            v8[gg] += cw * q7;
            c8[gg] += 1;
          }
        }
        if (npts > 7) {
          double d0 = q0 - pg[7][0], d1 = q1 - pg[7][1], d2 = q2 - pg[7][2];
          double ss = d0 * d0 + d1 * d1 + d2 * d2;
          if (((livemask >> 56) & 1ull) && ss < r2) {
            double cw = 0.5 + 0.5 * cos(sqrt(ss) * 2.0 * PI / radius / 2.0);
            s8[7] += cw;
            v8[7] += cw * q7;
            c8[7] += 1;
          }
        }
        SRT_AS1 d2_t *r = (SRT_AS1 d2_t *)(rec + (size_t)k * REC);
        r[0] = qa;
        r[1] = qb;
        r[2] = qc;
        r[3] = d2_t{qd.x, 0.0};
        r[4] = d2_t{rc, 0.0};
        r[5] = d2_t{ca, sa};
      }
      SRT_PHASE(1);
#pragma unroll
      for (int gg = 0; gg < 8; ++gg) {
        double ts = wave_sum(s8[gg]), tv = wave_sum(v8[gg]);
        cnt8[gg] = (int)wave_sum((double)c8[gg]);
        hin8[gg] = lws * (tv / ts);
      }
    }
    double hin = hin8[0];
    int count = cnt8[0];
#pragma unroll
    for (int gg = 1; gg < 8; ++gg) {
      hin = (g == gg) ? hin8[gg] : hin;
      count = (g == gg) ? cnt8[gg] : count;
    }
    const bool fit = live && count >= J; // else status 2: too few samples (lsinterp_mod.f95:262-264)
    // the centre's h against this point's: eta = h_c / h_g - 1
    const double eta = (hin8[0] - hin) / hin;
    double etamax = (fit && g < 7) ? fabs(eta) : 0.0;
    etamax = fmax(etamax, __shfl_xor(etamax, 8));
    etamax = fmax(etamax, __shfl_xor(etamax, 16));
    etamax = fmax(etamax, __shfl_xor(etamax, 32));
    SRT_PHASE(2);
    const bool base_ok = exact != 1 && cnt8[0] >= 1 && hin8[0] > 0.0 && etamax <= 1.0e-3;
    // ---- base pass: the centre's x**1.1 and exp; which samples are direct; the free point's weight
    {
      const double reps = radius * 5.0e-16;
      const double hc = hin8[0] / 4.0;
      bool any_direct = false;
      double nrc = 0.0, nq0 = 0.0, nq1 = 0.0, nq2 = 0.0; // (one record ahead, as in pass 1)
      if (lane < n_list) {
        const SRT_AS1 double *r = rec + (size_t)lane * REC;
        nrc = r[8], nq0 = r[0], nq1 = r[1], nq2 = r[2];
      }
#pragma unroll 1
      for (int k = lane; k < n_list; k += 64) {
        SRT_AS1 double *r = rec + (size_t)k * REC;
        const double rc = nrc, q0 = nq0, q1 = nq1, q2 = nq2;
        if (k + 64 < n_list) {
          const SRT_AS1 double *rn = rec + (size_t)(k + 64) * REC;
          nrc = rn[8], nq0 = rn[0], nq1 = rn[1], nq2 = rn[2];
        }
        const double xr = rc + reps;
        const double x = xr / hc;
        const double u = x * exp(0.1 * log(x));
        const double E = exp(-u);
        const double inv = 1.0 / xr;
        double tb = dmax * inv;
        tb = tb + etamax * (1.0 + tb);
        const bool series = base_ok && tb <= 1.0e-3 && u * 1.2 * tb <= 1.0e-3;
        any_direct = any_direct || !series;
        r[7] = series ? 0.0 : 1.0;
        r[9] = inv;
        r[12] = u;
        r[13] = E;
        if (npts > 7) {
          double d0 = q0 - pg[7][0], d1 = q1 - pg[7][1], d2 = q2 - pg[7][2];
          double ss = d0 * d0 + d1 * d1 + d2 * d2;
          r[15] = (((livemask >> 56) & 1ull) && ss < r2) ? M.etainv_at(ss, hin8[7]) : 0.0;
        }
      }
      if (__any(any_direct)) { // rare: wave-uniform trips so that the shuffles see every lane
#pragma unroll 1
        for (int k0 = 0; k0 < n_list; k0 += 64) {
          const int k = k0 + lane;
          SRT_AS1 double *r = rec + (size_t)(k < n_list ? k : 0) * REC;
          const bool direct = k < n_list && r[7] != 0.0;
          if (!__any(direct)) continue;
          const double q0 = r[0], q1 = r[1], q2 = r[2];
#pragma unroll 1
          for (int gg = 0; gg < 7; ++gg) {
            const double px = __shfl(p[0], 8 * gg), py = __shfl(p[1], 8 * gg), pz = __shfl(p[2], 8 * gg);
            const double hg = __shfl(hin, 8 * gg);
            const bool lv = (livemask >> (8 * gg)) & 1ull;
            if (direct) {
              double d0 = q0 - px, d1 = q1 - py, d2 = q2 - pz;
              double ss = d0 * d0 + d1 * d1 + d2 * d2;
              r[8 + gg] = (lv && ss < r2) ? M.etainv_at(ss, hg) : 0.0;
            }
          }
        }
      }
    }
    __syncthreads(); // block == one wave: the records written above are read by other lanes below
    SRT_PHASE(3);
    // ---- pass 2: normal equations.  Group g walks the records for point g, its 8 lanes splitting them; no
    // transcendental function and no branch in the loop
    constexpr bool O3 = J == 20; // order 3: all sums live in S20 (84 moments + 80), A is only formed inside its solve
    double A[O3 ? 1 : NT], b[O3 ? 1 : J][4];
    double Mm[J == 10 ? Moments::N : 1]; // order 2: the 35 moments stand in for the 55 entries of A while summing
    typename std::conditional<O3, Sums20, int>::type S20;
    int kept = 0;
    bool usemask = true, todo = fit;
    const double eta1 = 1.0 + eta;
    const int slot = (g == 7) ? 15 : 8 + g;
    const bool group7 = g == 7;
#pragma unroll 1
    for (int attempt = 0; attempt < 2; ++attempt) {
      if (!__any(todo)) break; // wave-uniform
      if (todo) {
        if constexpr (O3) S20.zero();
#pragma unroll
        for (int t = 0; t < (O3 ? 1 : NT); ++t) A[t] = 0.0;
#pragma unroll
        for (int t = 0; t < (J == 10 ? Moments::N : 1); ++t) Mm[t] = 0.0;
#pragma unroll
        for (int a = 0; a < (O3 ? 1 : J); ++a)
#pragma unroll
          for (int s = 0; s < 4; ++s) b[a][s] = 0.0;
        kept = 0;
      }
      double pw2 = 0.0, pd0 = 0.0, pd1 = 0.0, pd2 = 0.0, pln[4] = {0.0, 0.0, 0.0, 0.0}; // the record waiting to be folded in
      auto fold = [&](double w2, double d0, double d1, double d2, const double (&ln)[4]) {
        if constexpr (O3) {
          S20.fold(w2, d0, d1, d2, ln);
        } else if constexpr (J == 10) {
          const double dd[3] = {d0, d1, d2};
          double mono[Moments::N], m[10];
          Moments::monomials(mono, dd, std::make_integer_sequence<int, Moments::N - 1>{});
          Mm[0] += w2;
#pragma unroll
          for (int i = 1; i < Moments::N; ++i) Mm[i] += w2 * mono[i];
          Moments::firsts(mono, m, std::make_integer_sequence<int, 10>{}); // m_a: the same products as monomials<10>()
#pragma unroll
          for (int a = 0; a < 10; ++a) {
            double wa = w2 * m[a];
#pragma unroll
            for (int s = 0; s < 4; ++s) b[a][s] += wa * ln[s];
          }
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
      // The records come back through a ring of four 64-record buffers in LDS (the list area: the list is dead by now),
      // filled by LDS-DMA three buffers ahead -- the staging buffers of the CU's waves do not stay in L2 (the scans of
      // the other waves stream through it), and one wave per SIMD has nothing else to hide that latency behind.
      // Buffer layout: [16-byte chunk t of the record][record] -- DMA instruction t moves chunk t of 64 records (lane =
      // record), and the 8 lanes of a group read 8 neighbouring 16-byte slots (all 8 groups the same ones: broadcast).
      {
        const int nchunk = (n_list + 63) >> 6; // wave-uniform
        SRT_AS3 char *const ring = (SRT_AS3 char *)list;
        auto issue = [&](int c) {
          int r = c * 64 + lane;
          r = r < n_list ? r : n_list - 1;
          const SRT_AS1 char *src = (const SRT_AS1 char *)(rec + (size_t)r * REC);
          SRT_AS3 char *dst = ring + (c & 3) * 8192;
#pragma unroll
          for (int t = 0; t < 8; ++t)
            __builtin_amdgcn_global_load_lds((const SRT_AS1 void *)(src + 16 * t), (SRT_AS3 void *)(dst + 1024 * t), 16, 0, 0);
        };
        if (__any(todo)) {
          for (int c = 0; c < 3 && c < nchunk; ++c) issue(c);
        }
        const unsigned ring0 = (unsigned)(unsigned long long)ring + (unsigned)(sub * 16);
        const unsigned slot_off = (unsigned)((slot >> 1) * 1024 + (slot & 1) * 8);
#pragma unroll 1
        for (int c = 0; c < nchunk && __any(todo); ++c) {
          if (c + 3 < nchunk) issue(c + 3);
          const int ahead = nchunk - 1 - c; // buffers that may still be in flight while this one is read
          if (ahead >= 3) InterpModel::wait_vm<24>();
          else if (ahead == 2) InterpModel::wait_vm<16>();
          else if (ahead == 1) InterpModel::wait_vm<8>();
          else InterpModel::wait_vm<0>();
          if (todo) {
            const unsigned cbase = ring0 + (unsigned)((c & 3) * 8192);
#pragma unroll 1
            for (int i = 0; i < 8; ++i) {
              const int k = c * 64 + 8 * i + sub;
              const unsigned ra = cbase + (unsigned)(i * 128);
              d2_t c0, c1, c2, c3, c4, c5, c6;
              double wdir;
              // (inline asm: the compiler's wait-count pass would make an LDS load it can see wait for ALL DMA in flight)
              asm volatile("ds_read_b128 %0, %8\n\t"
                           "ds_read_b128 %1, %8 offset:1024\n\t"
                           "ds_read_b128 %2, %8 offset:2048\n\t"
                           "ds_read_b128 %3, %8 offset:3072\n\t"
                           "ds_read_b128 %4, %8 offset:4096\n\t"
                           "ds_read_b128 %5, %8 offset:5120\n\t"
                           "ds_read_b128 %6, %8 offset:6144\n\t"
                           "ds_read_b64 %7, %9\n\t"
                           "s_waitcnt lgkmcnt(0)"
                           : "=&v"(c0), "=&v"(c1), "=&v"(c2), "=&v"(c3), "=&v"(c4), "=&v"(c5), "=&v"(c6), "=&v"(wdir)
                           : "v"(ra), "v"(ra + slot_off)
                           : "memory");
              // the sums of the PREVIOUS record (independent of this record's weight, which is one long dependent chain)
              fold(pw2, pd0, pd1, pd2, pln);
              const double d0 = c0.x - p[0], d1 = c0.y - p[1], d2 = c1.x - p[2];
              const double ss = d0 * d0 + d1 * d1 + d2 * d2;
              // series from the centre's values ([8] r_c [9] inv [10] ca [11] sa [12] u_c [13] E_c)
              const double dr = sqrt(ss) - c4.x;
              const double ti = dr * c4.y;
              const double tau = ti * eta1 + eta;
              const double sp = tau * (1.1 + tau * (0.055 + tau * (-0.0165 + tau * (0.0078375 + tau * -0.00454575))));
              const double du = c6.x * sp;
              const double X = 1.0 + du * (-1.0 + du * (0.5 + du * (-1.0 / 6.0 + du * (1.0 / 24.0 + du * (-1.0 / 120.0 + du * (1.0 / 720.0))))));
              const double da = dr * pi_R, da2 = da * da;
              const double cd = 1.0 + da2 * (-0.5 + da2 * (1.0 / 24.0));
              const double sd = da * (1.0 + da2 * (-1.0 / 6.0 + da2 * (1.0 / 120.0)));
              const double win = 0.5 + 0.5 * (c5.x * cd - c5.y * sd);
              double e = c6.y * X * win;
              e = (group7 || c3.y != 0.0) ? wdir : e;
              // strictly inside (kdtree_mod.f95:171; the shared list holds a superset); :316-317
              const bool use = k < n_list && ss < r2 && !(usemask && !(e > 1.0e-16));
              kept += use ? 1 : 0;
              // (this record waits for the next trip's reads: same records in the same order, the sums do not change)
              pw2 = use ? 0.5 * e : 0.0;
              pd0 = d0, pd1 = d1, pd2 = d2;
              pln[0] = c1.y, pln[1] = c2.x, pln[2] = c2.y, pln[3] = c3.x;
            }
          }
        }
        if (todo) {
          fold(pw2, pd0, pd1, pd2, pln); // the last record
          pw2 = 0.0;
        }
        InterpModel::wait_vm<0>();
      }
      const int kept_all = group_sum(kept);
      // threw out too many samples: use them all (:319-323)
      todo = todo && kept_all < J;
      if (todo) usemask = false;
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
      } else {
        if constexpr (J == 10) {
#pragma unroll
          for (int t = 0; t < Moments::N; ++t) Mm[t] = group_sum(Mm[t]);
          Moments::expand(Mm, A, std::make_integer_sequence<int, 55>{});
        } else {
#pragma unroll
          for (int t = 0; t < NT; ++t) A[t] = group_sum(A[t]);
        }
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

  // out[8][4] (per lane): densities at the lane's stencil points 0..npts-1 (point 7 = extra).  All 64 lanes call
  // together; lanes with need == false are not served (their out is left untouched).
  //
  // Per owner lane: its <= 8 points go to the 8 groups of 8 lanes.  Normally all points share the centre's grid cell
  // and the six offsets are tiny against the radius: then ONE scan of the centre's 27 cells by all 64 lanes (radius
  // widened by the stencil's extent) yields a superset list in LDS and shared_fit() does the rest; otherwise own_fit().
  template <int J>
  __device__ __noinline__ void coop_stencil(const double *c, const double *d, const double *extra, int npts, bool need,
                                            double *out, SRT_LDS int *lists) const {
    const ScatteredModel M = uniform_copy();
    const double radius = M.radius;
    const int nspec = M.nspec;
    const int lane = threadIdx.x, g = lane >> 3;
    const unsigned long long needmask = __ballot(need);
    double *const rec = (double *)((SRT_LDS unsigned long long *)lists)[LDS_SCRATCH_SLOT]; // nullptr: no staging buffer
#pragma unroll 1
    for (int j = 0; j < 64; ++j) {
      if (!((needmask >> j) & 1ull)) continue; // wave-uniform
      // the owner's stencil, and this group's point of it
      double p[3];
      {
        double oc[3], od[3], oe[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          oc[k] = __shfl(c[k], j);
          od[k] = __shfl(d[k], j);
          oe[k] = (npts > 7) ? __shfl(extra[k], j) : 0.0;
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
          for (int t = 0; t < 4 * npts; ++t) out[t] = 0.0;
        continue;
      }
      SRT_PHASE_BEGIN(lists);
      SRT_PHASE_COUNT(8);
      // ---- candidates of the shared path
      double pc[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) pc[k] = __shfl(p[k], 0);
      // distance of this group's point from the centre; dmax: of the six offset points, e2max: of all (free point too)
      double e2 = (g < npts) ? (p[0] - pc[0]) * (p[0] - pc[0]) + (p[1] - pc[1]) * (p[1] - pc[1]) + (p[2] - pc[2]) * (p[2] - pc[2]) : 0.0;
      double e2s = (g < 7) ? e2 : 0.0;
      e2 = fmax(e2, __shfl_xor(e2, 8));
      e2 = fmax(e2, __shfl_xor(e2, 16));
      e2 = fmax(e2, __shfl_xor(e2, 32));
      e2s = fmax(e2s, __shfl_xor(e2s, 8));
      e2s = fmax(e2s, __shfl_xor(e2s, 16));
      e2s = fmax(e2s, __shfl_xor(e2s, 32));
      const double dmax = sqrt(e2s);
      bool shared = rec != nullptr && dmax <= 1.0e-3 * radius &&
                    !__any(g < npts && (cx != __shfl(cx, 0) || R.cy != __shfl(R.cy, 0) || R.cz != __shfl(R.cz, 0)));
      int n_list = 0;
      SRT_PHASE(14);
      if (shared) {
        // widen by the largest distance of a stencil point from the centre (plus rounding slack): a superset of
        // every point's neighbour set; each point applies its own exact test later
        const double rs = radius + sqrt(e2);
        const double rs2 = rs * rs * (1.0 + 1.0e-12);
        Rows Rc;
        Rc.cy = __builtin_amdgcn_readlane(R.cy, 0);
        Rc.cz = __builtin_amdgcn_readlane(R.cz, 0);
        Rc.x0 = __builtin_amdgcn_readlane(R.x0, 0);
        Rc.x1 = __builtin_amdgcn_readlane(R.x1, 0);
        Rc.live = true;
        // All nine candidate rows advance together, 64 samples of each per trip: 27 coalesced loads in flight at once
        // (one wave per SIMD: nothing else hides their latency), no index arithmetic beyond row start + lane.  List
        // order: trip, row, lane.  The list area holds SHARED_CAP entries and a trip adds at most 9 x 64, so the
        // overflow test (does it fit the staging buffer?) is made once per trip.
        int lo9[9], hi9[9], maxlen = 0;
#pragma unroll
        for (int r = 0; r < 9; ++r) {
          M.row_range(Rc, r, lo9[r], hi9[r]);
          maxlen = max(maxlen, hi9[r] - lo9[r]);
        }
        SRT_PHASE_ADD(11, hi9[0] - lo9[0] + hi9[1] - lo9[1] + hi9[2] - lo9[2] + hi9[3] - lo9[3] + hi9[4] - lo9[4] + hi9[5] - lo9[5] + hi9[6] - lo9[6] + hi9[7] - lo9[7] + hi9[8] - lo9[8]);
        SRT_PHASE(15);
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
        if (maxlen > 0) fetch(0);
#pragma unroll 1
        for (int t0 = 0; t0 < maxlen; t0 += 64) {
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
          if (n_list > REC_CAP) { // does not fit: every group scans for itself instead
            shared = false;
            break;
          }
        }
        __syncthreads(); // block == one wave: orders the list writes before the reads below
      }
      SRT_PHASE(0);
      SRT_PHASE_ADD(10, n_list);
      const Fit4 fi = shared ? shared_fit<J>(p, live, livemask, npts, n_list, lists, rec, dmax) : own_fit<J>(p, live, R, lists);
      SRT_PHASE(shared ? 9 : 7);
      // hand the results to the owner through LDS (behind the lists and the staging-buffer pointer): the group leaders
      // park their four densities, the owner collects the 8 x 4
      {
        SRT_LDS d2_t *park = (SRT_LDS d2_t *)((SRT_LDS double *)lists + LDS_SCRATCH_SLOT + 2);
        double val[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) val[s] = (live && s < nspec) ? exp(fi.v[s]) : 0.0; // failed fit: fi = 0 -> Ns = 1
        SRT_PHASE(12);
        if ((lane & 7) == 0) {
          park[2 * g] = d2_t{val[0], val[1]};
          park[2 * g + 1] = d2_t{val[2], val[3]};
        }
        __syncthreads();
        if (lane == j) {
#pragma unroll
          for (int gg = 0; gg < 8; ++gg) {
            const d2_t a = park[2 * gg], b = park[2 * gg + 1];
            if (gg < npts) {
              out[gg * 4 + 0] = a.x;
              out[gg * 4 + 1] = a.y;
              out[gg * 4 + 2] = b.x;
              out[gg * 4 + 3] = b.y;
            }
          }
        }
        SRT_PHASE(13);
      }
      __syncthreads(); // the lists are reused by the next owner
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
    if (lds != nullptr) {
      // trace / gradient / RK-step kernels: the wave serves its lanes one at a time (coop_stencil)
      double out[32];
#pragma unroll
      for (int i = 0; i < 32; ++i) out[i] = 0.0;
      double cc[3] = {c[0], c[1], c[2]}, dd[3] = {d[0], d[1], d[2]};
      double ee[3] = {NE ? extra[0] : 0.0, NE ? extra[1] : 0.0, NE ? extra[2] : 0.0};
      SRT_LDS int *lists = (SRT_LDS int *)lds;
      if (order == 0) coop_stencil<1>(cc, dd, ee, 7 + NE, need, out, lists);
      else if (order == 1) coop_stencil<4>(cc, dd, ee, 7 + NE, need, out, lists);
      else if (order == 2) coop_stencil<10>(cc, dd, ee, 7 + NE, need, out, lists);
      else coop_stencil<20>(cc, dd, ee, 7 + NE, need, out, lists);
#pragma unroll
      for (int i = 0; i < 7 + NE; ++i)
#pragma unroll
        for (int s = 0; s < 4; ++s) Ns[i][s] = out[i * 4 + s];
      return;
    }
    double p[7 + NE][3];
#pragma unroll
    for (int i = 0; i < 7 + NE; ++i) {
      p[i][0] = c[0];
      p[i][1] = c[1];
      p[i][2] = c[2];
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      p[1 + 2 * a][a] = c[a] + d[a];
      p[2 + 2 * a][a] = c[a] - d[a];
    }
    if (NE) {
      p[7 + NE - 1][0] = extra[0];
      p[7 + NE - 1][1] = extra[1];
      p[7 + NE - 1][2] = extra[2];
    }
    density<7 + NE>(p, Ns, nullptr);
  }

  template <int NP>
  __device__ __forceinline__ void density(const double (&p)[NP][3], double (&Ns)[NP][4], double *) const {
#pragma unroll
    for (int i = 0; i < NP; ++i) { // static indices; the interpolator itself is one out-of-line copy per order
      double x[3] = {p[i][0], p[i][1], p[i][2]}, n4[4];
      dens_point(x, n4);
#pragma unroll
      for (int s = 0; s < 4; ++s) Ns[i][s] = n4[s];
    }
  }
};

// The wave's slice of the launch's staging buffer (ScatteredModel::REC_CAP records per one-wave block), parked in LDS
// behind the lists for coop_stencil; nullptr = none (every stencil then takes the own-list path).
__device__ __forceinline__ void bind_scratch(const ScatteredModel &, double *lds, double *scratch) {
  if (lds == nullptr) return;
  double *mine = scratch ? scratch + (size_t)blockIdx.x * ScatteredModel::REC_CAP * ScatteredModel::REC : nullptr;
  ((SRT_LDS unsigned long long *)lds)[ScatteredModel::LDS_SCRATCH_SLOT] = (unsigned long long)mine;
  SRT_PHASE_ZERO((SRT_LDS double *)lds);
  __syncthreads();
}

} // namespace srt
