// srt_scattered_host.cpp -- host-side preparation of the scattered-sample model (no device code).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <numeric>
#include <thread>

#include "srt_host.hpp"

namespace srt_host {

static const double R_E = 6371.2e3;

namespace {
struct Grid {
  double origin[3], inv;
  int dims[3];
  std::vector<int> start, order; // CSR over cells; order = sample indices sorted by cell
  int cell_of(const double *p, int k) const {
    int c = (int)std::floor((p[k] - origin[k]) * inv);
    return c < 0 ? 0 : (c >= dims[k] ? dims[k] - 1 : c);
  }
  size_t id(int cx, int cy, int cz) const { return ((size_t)cz * dims[1] + cy) * dims[0] + cx; }
};

void build_grid(const std::vector<double> &xyz, int n, double cell, const double lo[3], const double hi[3], Grid &g) {
  g.inv = 1.0 / cell;
  size_t ncell = 1;
  for (int k = 0; k < 3; ++k) {
    g.origin[k] = lo[k];
    g.dims[k] = std::max(1, (int)std::floor((hi[k] - lo[k]) * g.inv) + 1);
    ncell *= (size_t)g.dims[k];
  }
  g.start.assign(ncell + 1, 0);
  std::vector<size_t> cid(n);
  for (int i = 0; i < n; ++i) {
    const double *p = &xyz[3 * (size_t)i];
    cid[i] = g.id(g.cell_of(p, 0), g.cell_of(p, 1), g.cell_of(p, 2));
    g.start[cid[i] + 1]++;
  }
  for (size_t c = 0; c < ncell; ++c) g.start[c + 1] += g.start[c];
  g.order.resize(n);
  std::vector<int> fill(g.start.begin(), g.start.end() - 1);
  for (int i = 0; i < n; ++i) g.order[fill[cid[i]]++] = i;
}
} // namespace

// ---- binary side-format of the model-4 sample file: "SRTPTS01", int32 nspec, int32 0, int64 npts,
// bounds[6], qs[4], ms[4] (136-byte header), then npts records of 3 + nspec little-endian doubles ----
static const char PTS_MAGIC[8] = {'S', 'R', 'T', 'P', 'T', 'S', '0', '1'};
struct PtsHeader {
  char magic[8];
  int32_t nspec, zero;
  int64_t npts;
  double bounds[6], qs[4], ms[4];
};
bool is_binary_points(const char *path) {
  FILE *f = fopen(path, "rb");
  if (!f) return false;
  char m[8];
  const bool ok = fread(m, 1, 8, f) == 8 && memcmp(m, PTS_MAGIC, 8) == 0;
  fclose(f);
  return ok;
}
static bool read_points_binary(const char *path, PtsHeader &h, std::vector<double> &raw, std::string &err) {
  FILE *f = fopen(path, "rb");
  if (!f) { err = "cannot open"; return false; }
  bool ok = fread(&h, sizeof h, 1, f) == 1 && memcmp(h.magic, PTS_MAGIC, 8) == 0 && h.nspec >= 1 && h.nspec <= 4 && h.npts >= 0;
  if (ok) {
    raw.resize((size_t)h.npts * (size_t)(3 + h.nspec));
    ok = fread(raw.data(), sizeof(double), raw.size(), f) == raw.size();
  }
  fclose(f);
  if (!ok) err = "truncated or malformed binary sample file";
  return ok;
}
bool write_points_file(const char *path, bool binary, int nspec, int64_t npts, const double bounds[6], const double *qs, const double *ms,
                       const double *rec, std::string &err) {
  FILE *f = fopen(path, binary ? "wb" : "w");
  if (!f) { err = "cannot open for writing"; return false; }
  bool ok = true;
  if (binary) {
    PtsHeader h;
    memset(&h, 0, sizeof h);
    memcpy(h.magic, PTS_MAGIC, 8);
    h.nspec = nspec;
    h.npts = npts;
    for (int k = 0; k < 6; ++k) h.bounds[k] = bounds[k];
    for (int k = 0; k < nspec; ++k) { h.qs[k] = qs[k]; h.ms[k] = ms[k]; }
    ok = fwrite(&h, sizeof h, 1, f) == 1 && fwrite(rec, sizeof(double), (size_t)npts * (3 + nspec), f) == (size_t)npts * (3 + nspec);
  } else {
    // gcpm_dens_model_buildgrid_random.f95:196-208 (header) and helpermod :37-43 (one sample per record, es24.15e3)
    char b[25];
    fprintf(f, "%10d\n", nspec);
    for (int k = 0; k < 6; ++k) { format_es24(bounds[k], b); fputs(b, f); }
    fputc('\n', f);
    for (int k = 0; k < nspec; ++k) { format_es24(qs[k], b); fputs(b, f); }
    fputc('\n', f);
    for (int k = 0; k < nspec; ++k) { format_es24(ms[k], b); fputs(b, f); }
    fputc('\n', f);
    std::string line;
    for (int64_t i = 0; i < npts && ok; ++i) {
      line.clear();
      for (int k = 0; k < 3 + nspec; ++k) { format_es24(rec[(size_t)i * (3 + nspec) + k], b); line += b; }
      line += '\n';
      ok = fputs(line.c_str(), f) >= 0;
    }
  }
  ok = (fclose(f) == 0) && ok;
  if (!ok) err = "write failed";
  return ok;
}

bool build_scattered(const char *path, double window_scale, ScatteredHost &out, std::string &err, double cell_scale, long long root_file_index) {
  std::vector<double> raw;
  int nspec = 0;
  if (is_binary_points(path)) {
    PtsHeader h;
    if (!read_points_binary(path, h, raw, err)) return false;
    nspec = h.nspec;
    out.nspec = nspec;
    for (int k = 0; k < nspec; ++k) { out.qs[k] = h.qs[k]; out.ms[k] = h.ms[k]; }
  } else {
    // fast path: the whole file tokenised by all cores; valid when every record holds exactly what its READ takes
    // (7 header numbers, nspec charges, nspec masses, then 3 + nspec numbers per sample)
    std::vector<double> all;
    std::string e2;
    bool fast = read_all_numbers(path, all, e2) && all.size() >= 7;
    if (fast) {
      nspec = (int)all[0];
      fast = nspec >= 1 && nspec <= 4 && all.size() >= (size_t)(7 + 2 * nspec) &&
             (all.size() - 7 - 2 * nspec) % (size_t)(3 + nspec) == 0;
    }
    if (fast) {
      out.nspec = nspec;
      for (int k = 0; k < nspec; ++k) {
        out.qs[k] = all[7 + k];
        out.ms[k] = all[7 + nspec + k];
      }
      raw.assign(all.begin() + 7 + 2 * nspec, all.end());
    } else {
      // the Fortran's record semantics, one READ at a time
      ListReader r(path);
      if (!r.ok()) {
        err = "cannot open";
        return false;
      }
      double hdr[7];
      if (r.read(7, hdr) != 7) { err = "header (nspec + bounds) incomplete"; return false; }
      nspec = (int)hdr[0];
      if (nspec < 1 || nspec > 4) { err = "nspec must be 1..4 (SRT_MAXSPEC: this library carries at most four species; see INTEGRATION.md section 1)"; return false; }
      out.nspec = nspec;
      if (r.read(nspec, out.qs) != nspec || r.read(nspec, out.ms) != nspec) { err = "charges/masses incomplete"; return false; }
      double row[8];
      while (r.read(3 + nspec, row) == 3 + nspec) raw.insert(raw.end(), row, row + 3 + nspec);
    }
  }
  const int w = 3 + nspec;
  int n0 = (int)(raw.size() / w);
  if (n0 == 0) { err = "no samples"; return false; }
  // drop exact duplicates (the reference ignores a sample whose position is already in the tree)
  std::vector<int> idx(n0);
  std::iota(idx.begin(), idx.end(), 0);
  std::sort(idx.begin(), idx.end(), [&](int a, int b) {
    const double *pa = &raw[(size_t)a * w], *pb = &raw[(size_t)b * w];
    if (pa[0] != pb[0]) return pa[0] < pb[0];
    if (pa[1] != pb[1]) return pa[1] < pb[1];
    if (pa[2] != pb[2]) return pa[2] < pb[2];
    return a < b;
  });
  std::vector<int> keep;
  for (int j = 0; j < n0; ++j) {
    if (j > 0) {
      const double *pa = &raw[(size_t)idx[j - 1] * w], *pb = &raw[(size_t)idx[j] * w];
      if (pa[0] == pb[0] && pa[1] == pb[1] && pa[2] == pb[2]) continue;
    }
    keep.push_back(idx[j]);
  }
  std::sort(keep.begin(), keep.end());
  const int n = (int)keep.size();
  std::vector<double> xyz(3 * (size_t)n);
  double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
  for (int i = 0; i < n; ++i)
    for (int k = 0; k < 3; ++k) {
      double v = raw[(size_t)keep[i] * w + k];
      xyz[3 * (size_t)i + k] = v;
      lo[k] = std::min(lo[k], v);
      hi[k] = std::max(hi[k], v);
    }
  // nearest other sample for every sample outside the Earth: uniform search grid with ~4 samples per cell
  double vol = 1.0;
  for (int k = 0; k < 3; ++k) vol *= std::max(hi[k] - lo[k], 1.0);
  double cell = std::cbrt(vol / std::max(n, 1) * 4.0);
  Grid g;
  build_grid(xyz, n, cell, lo, hi, g);
  std::vector<double> nn(n, 1.0); // placeholder 1.0 for samples inside the Earth (scattered_..adapter.f95:152)
  unsigned nthr = std::thread::hardware_concurrency();
  nthr = nthr ? (nthr > 32 ? 32 : nthr) : 1;
  if (n < 20000) nthr = 1;
  std::vector<double> maxn(nthr, 0.0);
  auto nn_range = [&](unsigned tid, int i0, int i1) {
  double maxnearest = 0.0;
  for (int i = i0; i < i1; ++i) {
    const double *p = &xyz[3 * (size_t)i];
    if (p[0] * p[0] + p[1] * p[1] + p[2] * p[2] < R_E * R_E) continue;
    int c[3] = {g.cell_of(p, 0), g.cell_of(p, 1), g.cell_of(p, 2)};
    double best = -1.0;
    int maxring = std::max(g.dims[0], std::max(g.dims[1], g.dims[2]));
    for (int ring = 1; ring <= maxring; ++ring) {
      for (int cz = std::max(0, c[2] - ring); cz <= std::min(g.dims[2] - 1, c[2] + ring); ++cz)
        for (int cy = std::max(0, c[1] - ring); cy <= std::min(g.dims[1] - 1, c[1] + ring); ++cy)
          for (int cx = std::max(0, c[0] - ring); cx <= std::min(g.dims[0] - 1, c[0] + ring); ++cx) {
            // only the shell of this ring (inner cells were scanned at smaller rings)
            if (ring > 1 && std::abs(cx - c[0]) < ring && std::abs(cy - c[1]) < ring && std::abs(cz - c[2]) < ring) continue;
            size_t id = g.id(cx, cy, cz);
            for (int s = g.start[id]; s < g.start[id + 1]; ++s) {
              int j = g.order[s];
              if (j == i) continue;
              const double *q = &xyz[3 * (size_t)j];
              double d2 = (q[0] - p[0]) * (q[0] - p[0]) + (q[1] - p[1]) * (q[1] - p[1]) + (q[2] - p[2]) * (q[2] - p[2]);
              if (best < 0 || d2 < best) best = d2;
            }
          }
      // the scanned cube extends at least ring*cell beyond p in every direction: nothing closer can remain
      if (best >= 0 && std::sqrt(best) <= (double)ring * cell) break;
    }
    if (best >= 0) {
      nn[i] = std::sqrt(best);
      maxnearest = std::max(maxnearest, nn[i]);
    }
  }
  maxn[tid] = maxnearest;
  };
  {
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nthr; ++t)
      th.emplace_back(nn_range, t, (int)((long long)n * t / nthr), (int)((long long)n * (t + 1) / nthr));
    for (auto &x : th) x.join();
  }
  double maxnearest = 0.0;
  for (double v : maxn) maxnearest = std::max(maxnearest, v);
  if (root_file_index >= 0) {
    // opt-in: the reference's kd-tree root.  Its kdtree_nearest starts from the root, so for the ONE sample that is the root
    // "nearest other sample" is the sample itself: its stored spacing stays 0 and never enters maxnearest
    // (kdtree_mod.f95:386-444, scattered_interp_dens_model_adapter.f95:167-203; SURVEY A-12).  Which sample that is comes from
    // the reference build's RNG (randperm); the caller names it by its position in the file.
    int ri = -1;
    for (int i = 0; i < n; ++i)
      if (keep[i] == root_file_index) ri = i;
    if (ri < 0) { err = "scattered root sample: no such record in the file (or it is a dropped duplicate)"; return false; }
    const double *p = &xyz[3 * (size_t)ri];
    if (p[0] * p[0] + p[1] * p[1] + p[2] * p[2] >= R_E * R_E) {
      nn[ri] = 0.0;
      maxnearest = 0.0;
      for (int i = 0; i < n; ++i) {
        const double *q = &xyz[3 * (size_t)i];
        if (q[0] * q[0] + q[1] * q[1] + q[2] * q[2] >= R_E * R_E) maxnearest = std::max(maxnearest, nn[i]);
      }
    }
  }
  out.maxnearest = maxnearest;
  out.radius = maxnearest * window_scale;
  if (!(out.radius > 0)) { err = "degenerate sample set (max nearest distance is zero)"; return false; }
  // query grid: cell edge = radius * cell_scale (>= the radius), samples sorted by cell
  Grid q;
  build_grid(xyz, n, out.radius * (cell_scale >= 1.0 ? cell_scale : 1.0), lo, hi, q);
  out.npts = n;
  out.inv_cell = q.inv;
  for (int k = 0; k < 3; ++k) {
    out.origin[k] = q.origin[k];
    out.dims[k] = q.dims[k];
  }
  out.cell_start = q.start;
  out.pts.assign((size_t)n * 8, 0.0);
  for (int s = 0; s < n; ++s) {
    int i = q.order[s];
    double *dst = &out.pts[(size_t)s * 8];
    dst[0] = xyz[3 * (size_t)i];
    dst[1] = xyz[3 * (size_t)i + 1];
    dst[2] = xyz[3 * (size_t)i + 2];
    for (int k = 0; k < nspec; ++k) dst[3 + k] = raw[(size_t)keep[i] * w + 3 + k];
    dst[7] = nn[i];
  }
  return true;
}

} // namespace srt_host

// ---- C ABI: model-4 sample files (include/srt.h) ----
#include "../../include/srt.h"
int srt_set_error(int code, const char *fmt, ...);
extern "C" int srt_points_file_write(const char *path, int binary, int nspec, int64_t npts, const double bounds[6], const double *qs,
                                     const double *ms, const double *records) {
  if (!path || !bounds || !qs || !ms || (npts > 0 && !records) || npts < 0) return srt_set_error(SRT_EINVAL, "bad argument");
  if (nspec < 1 || nspec > SRT_MAXSPEC) return srt_set_error(SRT_EINVAL, "nspec=%d unsupported", nspec);
  std::string err;
  if (!srt_host::write_points_file(path, binary != 0, nspec, npts, bounds, qs, ms, records, err)) return srt_set_error(SRT_EIO, "%s: %s", path, err.c_str());
  return SRT_OK;
}
extern "C" int srt_points_file_is_binary(const char *path) { return path && srt_host::is_binary_points(path) ? 1 : 0; }
extern "C" int srt_points_file_convert(const char *in, const char *out_binary) {
  if (!in || !out_binary) return srt_set_error(SRT_EINVAL, "null argument");
  std::vector<double> all;
  std::string err;
  if (srt_host::is_binary_points(in)) return srt_set_error(SRT_EINVAL, "%s is already binary", in);
  if (!srt_host::read_all_numbers(in, all, err) || all.size() < 7) return srt_set_error(SRT_EIO, "%s: %s", in, err.empty() ? "too short" : err.c_str());
  const int nspec = (int)all[0];
  if (nspec < 1 || nspec > 4 || all.size() < (size_t)(7 + 2 * nspec) || (all.size() - 7 - 2 * nspec) % (size_t)(3 + nspec) != 0)
    return srt_set_error(SRT_EIO, "%s: not a model-4 sample file (or records do not match their READs)", in);
  const int64_t npts = (int64_t)((all.size() - 7 - 2 * nspec) / (size_t)(3 + nspec));
  if (!srt_host::write_points_file(out_binary, true, nspec, npts, &all[1], &all[7], &all[7 + nspec], all.data() + 7 + 2 * nspec, err))
    return srt_set_error(SRT_EIO, "%s: %s", out_binary, err.c_str());
  return SRT_OK;
}
