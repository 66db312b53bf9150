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
#include "srt_device.hpp"

namespace srt {

#define SRT_LDS __attribute__((address_space(3)))

struct ScatteredModel {
  const double *pts;     // [npts][8]: x, y, z, lnN[4], nearest-sample distance
  const int *cell_start; // [ncells + 1]
  double origin[3], inv_cell, radius, lws;
  int dims[3];
  int nspec, order, exact, npts;

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
    if (J >= 10) {
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
  // (offsets of ~1e-6 |x|, so practically the same neighbour set) go to the 8 groups of 8 lanes; the lanes of a
  // group split the candidate rows, compact the accepted samples into a per-group index list in LDS (ballot +
  // popcount), and then walk that dense list for the expensive per-neighbour work (cos / pow / exp weights, the
  // 55 + 40 normal-equation terms).  The partial sums of the 8 lanes are combined by DPP within the group.
  // Per point the terms are those of interpolate<J>; only the order of summation differs (it is RNG-dependent in
  // the reference anyway, SURVEY A-12).
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
    const int row = (cz * dims[1] + cy) * dims[0];
    lo = ok ? cell_start[row + R.x0] : 0;
    hi = ok ? cell_start[row + R.x1 + 1] : 0;
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
          const double *q = pts + (size_t)i * 8;
          double d0 = q[0] - p[0], d1 = q[1] - p[1], d2 = q[2] - p[2];
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

  // out[8][4] (per lane): densities at the lane's stencil points 0..npts-1 (point 7 = extra).  All 64 lanes call
  // together; lanes with need == false are not served (their out is left untouched).
  //
  // Per owner lane:  A. candidate scan -> index list in LDS.  Normally all stencil points share the centre's grid
  //                     cell: then ONE scan by all 64 lanes (centre point, radius widened by the stencil's extent)
  //                     yields a superset list shared by the 8 groups (SHARED_CAP entries); otherwise every group
  //                     scans the rows of its own point (8 lists of LIST_CAP, processed in pieces if they overflow).
  //                  B. pass 1 (window-weighted mean spacing): shared list -> each lane takes every 64th sample
  //                     for all 8 points, wave reduction; own lists -> group by group.
  //                  C. pass 2 (normal equations): group g walks the list for point g with the exact |q - p| < radius
  //                     test, 8 lanes splitting the samples; DPP reduction; Cholesky; result to the owner lane.
  static constexpr int SHARED_CAP = 8 * LIST_CAP;
  template <int J>
  __device__ __noinline__ void coop_stencil(const double *c, const double *d, const double *extra, int npts, bool need,
                                            double *out, SRT_LDS int *lists) const {
    const int lane = threadIdx.x, g = lane >> 3, sub = lane & 7;
    const unsigned long long needmask = __ballot(need);
    const double r2 = radius * radius;
    constexpr int NT = J * (J + 1) / 2;
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
          double t = floor((p[k] - origin[k]) * inv_cell);
          t = fmin(fmax(t, -2.0), (double)dims[k] + 1.0);
          cc[k] = (int)t;
        }
        cx = cc[0];
        R.cy = cc[1];
        R.cz = cc[2];
        R.x0 = cc[0] - 1 < 0 ? 0 : cc[0] - 1;
        R.x1 = cc[0] + 1 >= dims[0] ? dims[0] - 1 : cc[0] + 1;
        R.live = live;
      }
      const unsigned long long livemask = __ballot(live);
      if (livemask == 0ull) { // the whole stencil is inside the Earth
        if (lane == j)
          for (int t = 0; t < 4 * npts; ++t) out[t] = 0.0;
        continue;
      }
      // ---- A. candidates
      double pc[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) pc[k] = __shfl(p[k], 0);
      bool shared = !__any(g < npts && (cx != __shfl(cx, 0) || R.cy != __shfl(R.cy, 0) || R.cz != __shfl(R.cz, 0)));
      int n_list = 0;
      SRT_LDS int *list = lists; // the shared list, or this group's own
      if (shared) {
        // widen by the largest distance of a stencil point from the centre (plus rounding slack): a superset of
        // every point's neighbour set; each point applies its own exact test later
        double e2 = (g < npts) ? (p[0] - pc[0]) * (p[0] - pc[0]) + (p[1] - pc[1]) * (p[1] - pc[1]) + (p[2] - pc[2]) * (p[2] - pc[2]) : 0.0;
        e2 = fmax(e2, __shfl_xor(e2, 8));
        e2 = fmax(e2, __shfl_xor(e2, 16));
        e2 = fmax(e2, __shfl_xor(e2, 32));
        const double rs = radius + sqrt(e2);
        const double rs2 = rs * rs * (1.0 + 1.0e-12);
        Rows Rc;
        Rc.cy = __shfl(R.cy, 0);
        Rc.cz = __shfl(R.cz, 0);
        Rc.x0 = __shfl(R.x0, 0);
        Rc.x1 = __shfl(R.x1, 0);
        Rc.live = true;
#pragma unroll 1
        for (int r = 0; r < 9 && shared; ++r) {
          int lo, hi;
          row_range(Rc, r, lo, hi);
#pragma unroll 1
          for (int i = lo + lane; __any(i < hi); i += 64) {
            bool acc = false;
            if (i < hi) {
              const double *q = pts + (size_t)i * 8;
              double d0 = q[0] - pc[0], d1 = q[1] - pc[1], d2 = q[2] - pc[2];
              acc = d0 * d0 + d1 * d1 + d2 * d2 < rs2;
            }
            const unsigned long long m = __ballot(acc);
            if (n_list + __popcll(m) > SHARED_CAP) { // does not fit: every group scans for itself instead
              shared = false;
              break;
            }
            if (acc) list[n_list + __popcll(m & ((1ull << lane) - 1ull))] = i;
            n_list += __popcll(m);
          }
        }
        __syncthreads(); // block == one wave: orders the list writes before the reads below
      }
      // ---- B. pass 1: count, cosine-window-weighted mean of the samples' nearest-neighbour distances (:296-303)
      int count = 0;
      double sw = 0.0, swv = 0.0;
      bool partial = false;
      if (shared) {
        double pg[8][3]; // the 8 points, wave-uniform
#pragma unroll
        for (int gg = 0; gg < 8; ++gg)
#pragma unroll
          for (int k = 0; k < 3; ++k) pg[gg][k] = __shfl(p[k], 8 * gg);
        double s8[8], v8[8];
        int c8[8];
#pragma unroll
        for (int gg = 0; gg < 8; ++gg) {
          s8[gg] = v8[gg] = 0.0;
          c8[gg] = 0;
        }
#pragma unroll 1
        for (int k = lane; k < n_list; k += 64) {
          const double *q = pts + (size_t)list[k] * 8;
          const double q0 = q[0], q1 = q[1], q2 = q[2], q7 = q[7];
#pragma unroll
          for (int gg = 0; gg < 8; ++gg) {
            double d0 = q0 - pg[gg][0], d1 = q1 - pg[gg][1], d2 = q2 - pg[gg][2];
            double ss = d0 * d0 + d1 * d1 + d2 * d2;
            if (((livemask >> (8 * gg)) & 1ull) && ss < r2) {
              double cw = 0.5 + 0.5 * cos(sqrt(ss) * 2.0 * PI / radius / 2.0);
              s8[gg] += cw;
              v8[gg] += cw * q7;
              c8[gg] += 1;
            }
          }
        }
#pragma unroll
        for (int gg = 0; gg < 8; ++gg) {
          double ts = wave_sum(s8[gg]), tv = wave_sum(v8[gg]);
          int tc = (int)wave_sum((double)c8[gg]);
          if (g == gg) {
            sw = ts;
            swv = tv;
            count = tc;
          }
        }
      } else {
        list = lists + g * LIST_CAP;
        auto body1 = [&](int i) {
          const double *q = pts + (size_t)i * 8;
          double d0 = q[0] - p[0], d1 = q[1] - p[1], d2 = q[2] - p[2];
          double r = sqrt(d0 * d0 + d1 * d1 + d2 * d2);
          double cw = 0.5 + 0.5 * cos(r * 2.0 * PI / radius / 2.0);
          sw += cw;
          swv += cw * q[7];
          ++count;
        };
        partial = scan_rows(p, R, list, n_list, body1);
#pragma unroll 1
        for (int k = sub; k < n_list; k += 8) body1(list[k]);
        count = group_sum(count);
        sw = group_sum(sw);
        swv = group_sum(swv);
      }
      const bool fit = live && count >= J; // else status 2: too few samples (lsinterp_mod.f95:262-264)
      const double hin = lws * (swv / sw);
      // ---- C. pass 2: normal equations
      double A[NT], b[J][4];
      int kept = 0;
      bool usemask = true, todo = fit;
      auto body2q = [&](const double (&q)[8]) {
        double d0 = q[0] - p[0], d1 = q[1] - p[1], d2 = q[2] - p[2];
        double ss = d0 * d0 + d1 * d1 + d2 * d2;
        if (!(ss < r2)) return; // strictly inside (kdtree_mod.f95:171); only the shared list holds others
        double e = etainv(sqrt(ss), hin);
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
          for (int cI = a; cI < J; ++cI) A[t++] += wa * m[cI];
#pragma unroll
          for (int s = 0; s < 4; ++s) b[a][s] += wa * q[3 + s];
        }
      };
#pragma unroll 1
      for (int attempt = 0; attempt < 2; ++attempt) {
        if (!__any(todo)) break; // wave-uniform
        if (todo) {
#pragma unroll
          for (int t = 0; t < NT; ++t) A[t] = 0.0;
#pragma unroll
          for (int a = 0; a < J; ++a)
#pragma unroll
            for (int s = 0; s < 4; ++s) b[a][s] = 0.0;
          kept = 0;
        }
        auto guarded = [&](int i) { // (the rare piecewise path)
          if (todo) {
            const double *q = pts + (size_t)i * 8;
            double qq[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) qq[t] = q[t];
            body2q(qq);
          }
        };
        int nfin = n_list; // the complete list of pass 1, unless it had to be processed in pieces
        if (partial) scan_rows(p, R, list, nfin, guarded);
        if (todo) {
          // walk the list one sample ahead: the next record is in flight while this one is folded in
          int k = sub;
          double qn[8];
          if (k < nfin) {
            const double *q = pts + (size_t)list[k] * 8;
#pragma unroll
            for (int t = 0; t < 8; ++t) qn[t] = q[t];
          }
#pragma unroll 1
          while (k < nfin) {
            double qc[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) qc[t] = qn[t];
            k += 8;
            if (k < nfin) {
              const double *q = pts + (size_t)list[k] * 8;
#pragma unroll
              for (int t = 0; t < 8; ++t) qn[t] = q[t];
            }
            body2q(qc);
          }
        }
        if (partial) n_list = nfin;
        const int kept_all = group_sum(kept);
        // threw out too many samples: use them all (:319-323)
        todo = todo && kept_all < J;
        if (todo) usemask = false;
      }
      // combine the 8 lanes' partial sums, solve, and hand the result to the owner
      double fi[4] = {0.0, 0.0, 0.0, 0.0};
      if (__any(fit)) {
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
            for (int s = 0; s < 4; ++s) fi[s] = f4[s];
          }
        }
      }
      double val[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) val[s] = (live && s < nspec) ? exp(fi[s]) : 0.0; // failed fit: fi = 0 -> Ns = 1
#pragma unroll
      for (int gg = 0; gg < 8; ++gg)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          double v = __shfl(val[s], 8 * gg);
          if (lane == j && gg < npts) out[gg * 4 + s] = v;
        }
      __syncthreads(); // the lists are reused by the next owner
    }
  }

  __device__ __forceinline__ void dens_point(const double x[3], double Ns[4]) const {
    if (x[0] * x[0] + x[1] * x[1] + x[2] * x[2] > R_E * R_E) {
      double fi[4];
      if (order == 0) interpolate<1>(x, fi);
      else if (order == 1) interpolate<4>(x, fi);
      else interpolate<10>(x, fi);
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
      else coop_stencil<10>(cc, dd, ee, 7 + NE, need, out, lists);
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

} // namespace srt
