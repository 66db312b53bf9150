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
    return exp(-pow((r + radius * eps) / h, 1.1)) * win;
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
                                                  double (&Ns)[7 + NE][4], double *) const {
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
