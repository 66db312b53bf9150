// srt_host.cpp -- host-side file formats of the drop-in boundary (no device code).
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <thread>

#include "srt_host.hpp"

#include <charconv>
#include <cerrno>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../include/srt.h"

int srt_set_error(int code, const char *fmt, ...);

namespace srt_host {

// ---------------------------------------------------------------------------------------------
// xform_double: get_q_c_d (Get_q_c.f95:4-30) -> t1_d inverse (T1.f95) -> t2_d (T2.f95) -> mu (T4.f95)
static void rot_x(double a, const double in[3], double out[3]) {
  double c = cos(a), s = sin(a);
  out[0] = in[0]; out[1] = in[1] * c + in[2] * s; out[2] = in[2] * c - in[1] * s;
}
static void rot_z(double a, const double in[3], double out[3]) {
  double c = cos(a), s = sin(a);
  out[0] = in[0] * c + in[1] * s; out[1] = in[1] * c - in[0] * s; out[2] = in[2];
}
double dipole_tilt(int yearday, int msec) {
  const double degrad = 3.141592653589793238462643 / 180.0;
  int iyr = yearday / 1000, iday = yearday - iyr * 1000;
  double ut = msec / 3600000.0, fracday = ut / 24.0;
  double rmjd = 45.0 + (double)(iyr - 1859) * 365.0 + ((double)((iyr - 1861) / 4) + 1.0) + (double)iday - 1.0 + fracday;
  double factor = (rmjd - 46066.0) / 365.25;
  double phi = (78.8 + 4.283e-2 * factor) * degrad;
  double lamda = (289.1 - 1.413e-2 * factor) * degrad;
  double qg[3] = {cos(phi) * cos(lamda), cos(phi) * sin(lamda), sin(phi)};
  double t0 = (rmjd - 51544.5) / 36525.0; // T0.f95
  double theta = (100.461 + 36000.770 * t0 + 15.04107 * ut) * degrad;
  double tmp[3], tmp2[3], qc[3];
  rot_z(-theta, qg, tmp);
  double eps = (23.439 - 0.013 * t0) * degrad;
  double m = (357.528 + 35999.05 * t0 + 0.04107 * ut) * degrad;
  double cg = 280.46 + 36000.772 * t0 + 0.04107 * ut;
  double lamdas = (cg + (1.915 - 0.0048 * t0) * sin(m) + 0.02 * sin(2.0 * m)) * degrad;
  rot_x(eps, tmp, tmp2);
  rot_z(lamdas, tmp2, qc);
  return -atan(qc[0] / sqrt(qc[1] * qc[1] + qc[2] * qc[2]));
}

// ---------------------------------------------------------------------------------------------
// IGRF set-up for one date (host, once per model): what RECALC_08 leaves in COMMON /GEOPACK1/ and /GEOPACK2/ that
// IGRF_GSW_08 reads.  Everything is default REAL in the Fortran, so everything is float here; SUN_08's two DOUBLE
// PRECISION variables are double (geopack2008.for:333-381, :486-1196).
namespace {
struct SunAngles {
  float gst = 0.f, srasn = 0.f, sdec = 0.f;
};
SunAngles sun_angles(int iyear, int iday, int ihour, int min, int isec) {
  SunAngles o;
  const float RAD = 57.295779513f;
  if (iyear < 1901 || iyear > 2099) return o;
  const double fday = (double)(ihour * 3600 + min * 60 + isec) / 86400.0;
  const double dj = 365 * (iyear - 1900) + (iyear - 1901) / 4 + iday - 0.5 + fday;
  const float t = (float)(dj / (double)36525.f);
  const float vl = (float)std::fmod((double)279.696678f + (double)0.9856473354f * dj, 360.0);
  o.gst = (float)(std::fmod((double)279.690983f + (double).9856473354f * dj + (double)360.f * fday + (double)180.f, 360.0) / (double)RAD);
  const float g = (float)(std::fmod((double)358.475845f + (double)0.985600267f * dj, 360.0) / (double)RAD);
  float slong = (vl + (1.91946f - 0.004789f * t) * sinf(g) + 0.020094f * sinf(2.f * g)) / RAD;
  if (slong > 6.2831853f) slong = slong - 6.2831853f;
  if (slong < 0.f) slong = slong + 6.2831853f;
  const float obliq = (23.45229f - 0.0130125f * t) / RAD;
  const float sob = sinf(obliq), slp = slong - 9.924e-5f;
  const float sind = sob * sinf(slp);
  const float cosd = sqrtf(1.f - sind * sind);
  const float sc = sind / cosd;
  o.sdec = atanf(sc);
  o.srasn = 3.141592654f - atan2f(cosf(obliq) / sob * sc, -cosf(slp) / cosd);
  return o;
}
} // namespace

bool igrf_setup(const char *coeff_file, int yearday, int msec, float G[105], float H[105], float REC[105], float A[9],
                float *psi, std::string &err) {
  // table: "g|h mn v1965 v1970 ... v2020 sv"
  static const int NEP = 12;
  std::vector<float> tg(13 * 105, 0.f), th(13 * 105, 0.f);
  FILE *f = fopen(coeff_file, "r");
  if (!f) {
    err = std::string("cannot open IGRF coefficient table ") + coeff_file;
    return false;
  }
  char line[2048];
  int seen = 0;
  while (fgets(line, sizeof line, f)) {
    if (line[0] != 'g' && line[0] != 'h') continue;
    char *s = line + 1;
    const long mn = strtol(s, &s, 10);
    if (mn < 1 || mn > 105) continue;
    for (int e = 0; e <= NEP; ++e) (line[0] == 'g' ? tg : th)[e * 105 + (mn - 1)] = strtof(s, &s);
    ++seen;
  }
  fclose(f);
  if (seen != 210) {
    err = std::string("IGRF coefficient table has the wrong number of rows: ") + coeff_file;
    return false;
  }
  // itime -> calendar fields as the adapters do (interp_dens_model_adapter.f95:217-221)
  const int iyear = yearday / 1000, iday = yearday % 1000;
  const int ihour = msec / (1000 * 60 * 60);
  const int min = (msec - ihour * (1000 * 60 * 60)) / (1000 * 60);
  const int isec = (msec - ihour * (1000 * 60 * 60) - min * (1000 * 60)) / 1000;
  int iy = iyear < 1965 ? 1965 : iyear > 2025 ? 2025 : iyear;
  for (int n = 1; n <= 14; ++n) {
    const int n2 = (2 * n - 1) * (2 * n - 3);
    for (int m = 1; m <= n; ++m) REC[n * (n - 1) / 2 + m - 1] = (float)((n - m) * (n + m - 2)) / (float)n2;
  }
  const float yfrac = (float)iy + (float)(iday - 1) / 365.25f;
  if (iy >= 2020) { // extrapolate: secular variation for degrees <= 8 (the first 45 entries)
    const float dt = yfrac - 2020.f;
    for (int n = 0; n < 105; ++n) {
      G[n] = tg[11 * 105 + n];
      H[n] = th[11 * 105 + n];
      if (n < 45) {
        G[n] = G[n] + tg[12 * 105 + n] * dt;
        H[n] = H[n] + th[12 * 105 + n] * dt;
      }
    }
  } else { // interpolate between the two bracketing epochs
    const int e = (iy - 1965) / 5;
    const float f2 = (yfrac - (float)(1965 + 5 * e)) / 5.f, f1 = 1.f - f2;
    for (int n = 0; n < 105; ++n) {
      G[n] = tg[e * 105 + n] * f1 + tg[(e + 1) * 105 + n] * f2;
      H[n] = th[e * 105 + n] * f1 + th[(e + 1) * 105 + n] * f2;
    }
  }
  // Gauss -> Schmidt quasi-normalised
  float s = 1.f;
  for (int n = 2; n <= 14; ++n) {
    const int mn = n * (n - 1) / 2 + 1;
    s = s * (float)(2 * n - 3) / (float)(n - 1);
    G[mn - 1] = G[mn - 1] * s;
    H[mn - 1] = H[mn - 1] * s;
    float p = s;
    for (int m = 2; m <= n; ++m) {
      const float aa = m == 2 ? 2.f : 1.f;
      p = p * sqrtf(aa * (float)(n - m + 1) / (float)(n + m - 2));
      G[mn + m - 2] = G[mn + m - 2] * p;
      H[mn + m - 2] = H[mn + m - 2] * p;
    }
  }
  // dipole axis in GEO, Sun direction, GSE/GSW axes, GEO -> GSW rotation
  const float g10 = -G[1], g11 = G[2], h11 = H[2];
  const float sq = g11 * g11 + h11 * h11, sqq = sqrtf(sq), sqr = sqrtf(g10 * g10 + sq);
  const float sl0 = -h11 / sqq, cl0 = -g11 / sqq, st0 = sqq / sqr, ct0 = g10 / sqr;
  const float stcl = st0 * cl0, stsl = st0 * sl0;
  const SunAngles sun = sun_angles(iy, iday, ihour, min, isec);
  const float s1 = cosf(sun.srasn) * cosf(sun.sdec), s2 = sinf(sun.srasn) * cosf(sun.sdec), s3 = sinf(sun.sdec);
  const float dj = (float)(365 * (iy - 1900) + (iy - 1901) / 4 + iday) - 0.5f + (float)(ihour * 3600 + min * 60 + isec) / 86400.f;
  const float t = dj / 36525.f;
  const float obliq = (23.45229f - 0.0130125f * t) / 57.2957795f;
  const float dz1 = 0.f, dz2 = -sinf(obliq), dz3 = cosf(obliq);
  const float dy1 = dz2 * s3 - dz3 * s2, dy2 = dz3 * s1 - dz1 * s3, dy3 = dz1 * s2 - dz2 * s1;
  const float vx = -400.f, vy = 0.f, vz = 0.f; // tsy_recalc's solar wind: GSW coincides with GSM
  const float v = sqrtf(vx * vx + vy * vy + vz * vz);
  const float dx1 = -vx / v, dx2 = -vy / v, dx3 = -vz / v;
  const float x1 = dx1 * s1 + dx2 * dy1 + dx3 * dz1, x2 = dx1 * s2 + dx2 * dy2 + dx3 * dz2, x3 = dx1 * s3 + dx2 * dy3 + dx3 * dz3;
  const float cgst = cosf(sun.gst), sgst = sinf(sun.gst);
  const float dip1 = stcl * cgst - stsl * sgst, dip2 = stcl * sgst + stsl * cgst, dip3 = ct0;
  float y1 = dip2 * x3 - dip3 * x2, y2 = dip3 * x1 - dip1 * x3, y3 = dip1 * x2 - dip2 * x1;
  const float y = sqrtf(y1 * y1 + y2 * y2 + y3 * y3);
  y1 = y1 / y;
  y2 = y2 / y;
  y3 = y3 / y;
  const float z1 = x2 * y3 - x3 * y2, z2 = x3 * y1 - x1 * y3, z3 = x1 * y2 - x2 * y1;
  A[0] = x1 * cgst + x2 * sgst;
  A[1] = -x1 * sgst + x2 * cgst;
  A[2] = x3;
  A[3] = y1 * cgst + y2 * sgst;
  A[4] = -y1 * sgst + y2 * cgst;
  A[5] = y3;
  A[6] = z1 * cgst + z2 * sgst;
  A[7] = -z1 * sgst + z2 * cgst;
  A[8] = z3;
  if (psi) {
    // What the adapters pass to T04_s as the tilt angle: they declare `real(kind=SP) :: PSI; COMMON /GEOPACK1/ PSI`
    // (interp_dens_model_adapter.f95:45-46), i.e. PSI is the FIRST word of geopack's block -- and that word is ST0, the
    // sine of the dipole axis' colatitude (geopack2008.for:569), not RECALC_08's PSI (16th word).  The reference
    // therefore runs T04_s with PS = ST0 (~0.17 in 2010) whatever the date and hour; parity means doing the same.
    *psi = st0;
  }
  return true;
}

// ---------------------------------------------------------------------------------------------
ListReader::ListReader(const char *path) : f_(fopen(path, "r")), line_(1 << 16) {}
ListReader::~ListReader() {
  if (f_) fclose((FILE *)f_);
}
int64_t ListReader::read(int64_t n, double *out) {
  int64_t got = 0;
  FILE *f = (FILE *)f_;
  while (got < n) {
    if (!fgets(line_.data(), (int)line_.size(), f)) return got;
    char *s = line_.data();
    while (got < n) {
      while (*s == ' ' || *s == '\t' || *s == ',' || *s == '\r' || *s == '\n') ++s;
      if (!*s) break;
      char *e = s;
      // accept Fortran 'd' exponents
      char tok[64];
      int k = 0;
      while (*e && *e != ' ' && *e != '\t' && *e != ',' && *e != '\r' && *e != '\n' && k < 63) {
        char ch = *e++;
        tok[k++] = (ch == 'd' || ch == 'D') ? 'e' : ch;
      }
      tok[k] = 0;
      char *endp = nullptr;
      double v = strtod(tok, &endp);
      if (endp == tok) return got; // not a number
      out[got++] = v;
      s = e;
    }
  }
  return got;
}

bool read_newray(const char *path, NgoConfig &c, std::string &err) {
  ListReader r(path);
  if (!r.ok()) {
    err = "cannot open";
    return false;
  }
  double v[16];
  if (r.read(4, v) != 4) { err = "card 1 (intera numres nsuppr spelat) incomplete"; return false; }
  for (;;) { // satellite coordinates until distre <= -1 (:64-80)
    if (r.read(2, v) != 2) { err = "unterminated satellite-coordinate list"; return false; }
    c.last_latitu = v[1];
    if (v[0] <= -1.0) break;
  }
  if (r.read(10, v) != 10) { err = "model card (num kskip mode kount kducts ktape refalt dsrrng dsrlat dsdens) incomplete"; return false; }
  c.num = (int)v[0];
  c.kducts = (int)v[4];
  c.dsrrng = v[7]; c.dsrlat = v[8]; c.dsdens = v[9];
  if (c.num < 2 || c.num > 4) { err = "num must be 2..4"; return false; }
  if (c.kducts < 0 || c.kducts > 9) { err = "kducts must be 0..9"; return false; }
  if (r.read(5, v) != 5) { err = "card egfeq therm hm absb relb incomplete"; return false; }
  c.therm = v[1];
  if (r.read(5, v) != 5) { err = "card rbase ane0 alpha0(2:4) incomplete"; return false; }
  c.rbase = v[0]; c.ane0 = v[1]; c.alpha0[2] = v[2]; c.alpha0[3] = v[3]; c.alpha0[4] = v[4];
  if (r.read(5, v) != 5) { err = "card rzero scbot rstop rdiv hmin incomplete"; return false; }
  c.rzero = v[0]; c.scbot = v[1];
  if (c.kducts >= 1) {
    if (r.read(5, v) != 5) { err = "plasmapause card (lk expk ddk rconsn scr) incomplete"; return false; }
    c.lk = v[0]; c.expk = v[1]; c.ddk = v[2]; c.rconsn = v[3]; c.scr = v[4];
    for (int k = 2; k <= c.kducts; ++k) {
      if (r.read(12, v) != 12) { err = "duct card incomplete"; return false; }
      c.l0[k] = v[0]; c.def[k] = v[1]; c.dd[k] = v[2];
      c.rducln[k] = v[3]; c.hducln[k] = v[4]; c.rducun[k] = v[5]; c.hducun[k] = v[6];
      c.rducls[k] = v[7]; c.hducls[k] = v[8]; c.rducus[k] = v[9]; c.hducus[k] = v[10];
      c.sidedu[k] = v[11];
    }
  }
  return true;
}

// --- the record-by-record reader (exactly the Fortran's READ semantics; slow: one strtod at a time)
static bool read_grid_file_records(const char *path, GridFile &g, std::string &err) {
  ListReader r(path);
  if (!r.ok()) { err = "cannot open"; return false; }
  double v[8];
  if (r.read(5, v) != 5) { err = "size header incomplete"; return false; }
  g.compder = (int)v[0]; g.nspec = (int)v[1]; g.nx = (int)v[2]; g.ny = (int)v[3]; g.nz = (int)v[4];
  if (g.nspec < 1 || g.nspec > 4) { err = "nspec must be 1..4 (SRT_MAXSPEC: this library carries at most four species; see INTEGRATION.md section 1)"; return false; }
  if (g.nx < 2 || g.ny < 2 || g.nz < 2) { err = "grid needs >= 2 nodes per axis"; return false; }
  if (r.read(6, g.bounds) != 6) { err = "bounds incomplete"; return false; }
  if (r.read(g.nspec, g.qs) != g.nspec) { err = "charges incomplete"; return false; }
  if (r.read(g.nspec, g.ms) != g.nspec) { err = "masses incomplete"; return false; }
  size_t nnode = (size_t)g.nx * g.ny * g.nz, n = nnode * g.nspec;
  g.F.resize(n);
  for (size_t c = 0; c < nnode; ++c) // one record per node (:100-106)
    if (r.read(g.nspec, g.F.data() + c * g.nspec) != g.nspec) { err = "grid values truncated"; return false; }
  g.have_derivs = (g.compder == 1);
  if (g.have_derivs)
    for (int a = 0; a < 7; ++a) {
      g.derivs[a].resize(n);
      if (r.read((int64_t)n, g.derivs[a].data()) != (int64_t)n) { err = "derivative block truncated"; return false; }
    }
  return true;
}

// --- whole-file tokeniser, value block parsed by all host cores (a 256^3 x 4 grid is 67 M numbers, 1.7 GB of text)
namespace {
inline bool is_sep(char c) { return c == ' ' || c == '\t' || c == ',' || c == '\r' || c == '\n'; }
// parse every token of [b, e) as a double ('d' exponents accepted); false on a non-numeric token
bool parse_tokens(const char *b, const char *e, std::vector<double> &out) {
  char tok[72];
  while (b < e) {
    while (b < e && is_sep(*b)) ++b;
    if (b >= e) break;
    int k = 0;
    while (b < e && !is_sep(*b) && k < 70) {
      char ch = *b++;
      tok[k++] = (ch == 'd' || ch == 'D') ? 'e' : ch;
    }
    tok[k] = 0;
    char *endp = nullptr;
    double v = strtod(tok, &endp);
    if (endp == tok || *endp) return false;
    out.push_back(v);
  }
  return true;
}
struct MappedFile {
  const char *p = nullptr;
  size_t n = 0;
  int fd = -1;
  explicit MappedFile(const char *path) {
    fd = open(path, O_RDONLY);
    if (fd < 0) return;
    struct stat st;
    if (fstat(fd, &st) != 0 || st.st_size == 0) return;
    void *m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    if (m == MAP_FAILED) return;
    p = (const char *)m;
    n = (size_t)st.st_size;
  }
  ~MappedFile() {
    if (p) munmap((void *)p, n);
    if (fd >= 0) close(fd);
  }
};
const char GRID_MAGIC[8] = {'S', 'R', 'T', 'G', 'R', 'I', 'D', '1'};
struct GridBinHeader { // 8 + 5*4 + 4 (pad) + 14*8 = 144 bytes, then F, then 7 derivative blocks if compder == 1
  char magic[8];
  int32_t compder, nspec, nx, ny, nz, pad;
  double bounds[6], qs[4], ms[4];
};
} // namespace

bool is_binary_grid(const char *path) {
  FILE *f = fopen(path, "rb");
  if (!f) return false;
  char m[8] = {0};
  size_t k = fread(m, 1, 8, f);
  fclose(f);
  return k == 8 && memcmp(m, GRID_MAGIC, 8) == 0;
}

static bool read_grid_binary(const char *path, GridFile &g, std::string &err) {
  MappedFile mf(path);
  if (!mf.p) { err = "cannot open"; return false; }
  if (mf.n < sizeof(GridBinHeader)) { err = "binary grid: header truncated"; return false; }
  GridBinHeader h;
  memcpy(&h, mf.p, sizeof h);
  g.compder = h.compder; g.nspec = h.nspec; g.nx = h.nx; g.ny = h.ny; g.nz = h.nz;
  if (g.nspec < 1 || g.nspec > 4) { err = "nspec must be 1..4 (SRT_MAXSPEC: this library carries at most four species; see INTEGRATION.md section 1)"; return false; }
  if (g.nx < 2 || g.ny < 2 || g.nz < 2) { err = "grid needs >= 2 nodes per axis"; return false; }
  memcpy(g.bounds, h.bounds, sizeof g.bounds);
  memcpy(g.qs, h.qs, sizeof g.qs);
  memcpy(g.ms, h.ms, sizeof g.ms);
  size_t n = (size_t)g.nx * g.ny * g.nz * g.nspec;
  g.have_derivs = (g.compder == 1);
  size_t need = sizeof h + n * sizeof(double) * (g.have_derivs ? 8 : 1);
  if (mf.n < need) { err = "binary grid: value block truncated"; return false; }
  const char *q = mf.p + sizeof h;
  g.F.resize(n);
  memcpy(g.F.data(), q, n * sizeof(double));
  if (g.have_derivs)
    for (int a = 0; a < 7; ++a) {
      g.derivs[a].resize(n);
      memcpy(g.derivs[a].data(), q + (size_t)(a + 1) * n * sizeof(double), n * sizeof(double));
    }
  return true;
}

bool write_grid_binary(const char *path, const GridFile &g, std::string &err) {
  FILE *f = fopen(path, "wb");
  if (!f) { err = "cannot open for writing"; return false; }
  GridBinHeader h;
  memset(&h, 0, sizeof h);
  memcpy(h.magic, GRID_MAGIC, 8);
  h.compder = g.have_derivs ? 1 : 0; h.nspec = g.nspec; h.nx = g.nx; h.ny = g.ny; h.nz = g.nz;
  memcpy(h.bounds, g.bounds, sizeof h.bounds);
  memcpy(h.qs, g.qs, sizeof h.qs);
  memcpy(h.ms, g.ms, sizeof h.ms);
  bool ok = fwrite(&h, sizeof h, 1, f) == 1 && fwrite(g.F.data(), sizeof(double), g.F.size(), f) == g.F.size();
  if (g.have_derivs)
    for (int a = 0; a < 7 && ok; ++a) ok = fwrite(g.derivs[a].data(), sizeof(double), g.derivs[a].size(), f) == g.derivs[a].size();
  ok = (fclose(f) == 0) && ok;
  if (!ok) err = "write failed";
  return ok;
}

// every numeric token of a text file, in order, parsed by all host cores; false if the file cannot be opened or holds
// a non-numeric token
bool read_all_numbers(const char *path, std::vector<double> &all, std::string &err) {
  MappedFile mf(path);
  if (!mf.p) { err = "cannot open"; return false; }
  const char *b = mf.p, *e = mf.p + mf.n;
  unsigned nt = std::thread::hardware_concurrency();
  nt = nt ? (nt > 32 ? 32 : nt) : 1;
  if ((size_t)(e - b) < (size_t)nt * 4096) nt = 1;
  std::vector<const char *> cut(nt + 1);
  cut[0] = b;
  cut[nt] = e;
  for (unsigned t = 1; t < nt; ++t) {
    const char *c = b + (size_t)(e - b) * t / nt;
    const char *nl = (const char *)memchr(c, '\n', (size_t)(e - c));
    cut[t] = nl ? nl + 1 : e;
  }
  std::vector<std::vector<double>> part(nt);
  std::vector<char> good(nt, 1);
  std::vector<std::thread> th;
  for (unsigned t = 0; t < nt; ++t)
    th.emplace_back([&, t] {
      part[t].reserve((size_t)(cut[t + 1] - cut[t]) / 16 + 16);
      good[t] = parse_tokens(cut[t], cut[t + 1], part[t]) ? 1 : 0;
    });
  for (auto &x : th) x.join();
  size_t total = 0;
  for (unsigned t = 0; t < nt; ++t) {
    if (!good[t]) { err = "non-numeric token"; return false; }
    total += part[t].size();
  }
  all.clear();
  all.reserve(total);
  for (unsigned t = 0; t < nt; ++t) all.insert(all.end(), part[t].begin(), part[t].end());
  return true;
}

bool read_grid_file(const char *path, GridFile &g, std::string &err) {
  if (is_binary_grid(path)) return read_grid_binary(path, g, err);
  MappedFile mf(path);
  if (!mf.p) { err = "cannot open"; return false; }
  // header: the first four records (sizes; bounds; charges; masses)
  const char *b = mf.p, *e = mf.p + mf.n;
  std::vector<double> hv;
  auto record = [&](int want) -> bool { // one READ: starts on a new record, may continue over following ones
    size_t got0 = hv.size();
    while ((int)(hv.size() - got0) < want && b < e) {
      const char *nl = (const char *)memchr(b, '\n', (size_t)(e - b));
      const char *le = nl ? nl : e;
      std::vector<double> t;
      if (!parse_tokens(b, le, t)) return false;
      for (double v : t)
        if ((int)(hv.size() - got0) < want) hv.push_back(v);
      b = nl ? nl + 1 : e;
    }
    return (int)(hv.size() - got0) == want;
  };
  if (!record(5)) { err = "size header incomplete"; return false; }
  g.compder = (int)hv[0]; g.nspec = (int)hv[1]; g.nx = (int)hv[2]; g.ny = (int)hv[3]; g.nz = (int)hv[4];
  if (g.nspec < 1 || g.nspec > 4) { err = "nspec must be 1..4 (SRT_MAXSPEC: this library carries at most four species; see INTEGRATION.md section 1)"; return false; }
  if (g.nx < 2 || g.ny < 2 || g.nz < 2) { err = "grid needs >= 2 nodes per axis"; return false; }
  if (!record(6)) { err = "bounds incomplete"; return false; }
  if (!record(g.nspec)) { err = "charges incomplete"; return false; }
  if (!record(g.nspec)) { err = "masses incomplete"; return false; }
  for (int k = 0; k < 6; ++k) g.bounds[k] = hv[5 + k];
  for (int k = 0; k < g.nspec; ++k) { g.qs[k] = hv[11 + k]; g.ms[k] = hv[11 + g.nspec + k]; }
  const size_t nnode = (size_t)g.nx * g.ny * g.nz, n = nnode * g.nspec;
  g.have_derivs = (g.compder == 1);
  const size_t want = n * (g.have_derivs ? 8 : 1);
  // value block: cut at record boundaries, one piece per thread
  unsigned nt = std::thread::hardware_concurrency();
  nt = nt ? (nt > 32 ? 32 : nt) : 1;
  if ((size_t)(e - b) < (size_t)nt * 4096) nt = 1;
  std::vector<const char *> cut(nt + 1);
  cut[0] = b;
  cut[nt] = e;
  for (unsigned t = 1; t < nt; ++t) {
    const char *c = b + (size_t)(e - b) * t / nt;
    const char *nl = (const char *)memchr(c, '\n', (size_t)(e - c));
    cut[t] = nl ? nl + 1 : e;
  }
  std::vector<std::vector<double>> part(nt);
  std::vector<char> good(nt, 1);
  std::vector<std::thread> th;
  for (unsigned t = 0; t < nt; ++t)
    th.emplace_back([&, t] {
      part[t].reserve((size_t)(cut[t + 1] - cut[t]) / 20 + 16);
      good[t] = parse_tokens(cut[t], cut[t + 1], part[t]) ? 1 : 0;
    });
  for (auto &x : th) x.join();
  size_t total = 0;
  bool allgood = true;
  for (unsigned t = 0; t < nt; ++t) { total += part[t].size(); allgood = allgood && good[t]; }
  if (!allgood || total != want) {
    // not the plain one-record-per-node layout (extra fields on a line, a stray token, ...): re-read with the
    // Fortran's record semantics
    g = GridFile();
    return read_grid_file_records(path, g, err);
  }
  g.F.resize(n);
  if (g.have_derivs)
    for (int a = 0; a < 7; ++a) g.derivs[a].resize(n);
  size_t pos = 0;
  for (unsigned t = 0; t < nt; ++t) {
    for (double v : part[t]) {
      size_t blk = pos / n, off = pos % n;
      (blk == 0 ? g.F : g.derivs[blk - 1])[off] = v;
      ++pos;
    }
    std::vector<double>().swap(part[t]);
  }
  return true;
}

// ---------------------------------------------------------------------------------------------
void format_es24(double v, char out[25]) {
  char tmp[64];
  if (std::isnan(v)) {
    snprintf(out, 25, "%24s", "NaN");
    return;
  }
  if (std::isinf(v)) {
    snprintf(out, 25, "%24s", v > 0 ? "Inf" : "-Inf");
    return;
  }
  snprintf(tmp, sizeof tmp, "%.15E", v); // d.dddddddddddddddE+XX
  char *e = strchr(tmp, 'E');
  int ex = atoi(e + 1);
  *e = 0;
  char buf[64];
  snprintf(buf, sizeof buf, "%sE%c%03d", tmp, ex < 0 ? '-' : '+', ex < 0 ? -ex : ex);
  snprintf(out, 25, "%24s", buf);
}

} // namespace srt_host

// ================================================================================ C ABI (file I/O)
extern "C" void srt_free(void *p) { free(p); }

// ---- model-3 grid files: text of gcpm_dens_model_buildgrid.f95:302-327 <-> binary side-format
extern "C" int srt_grid_file_read(const char *path, int32_t dims[5], double bounds[6], double qs[4], double ms[4],
                                  double **F, double **derivs) {
  if (!path || !dims || !bounds || !qs || !ms || !F) return srt_set_error(SRT_EINVAL, "null argument");
  srt_host::GridFile g;
  std::string err;
  if (!srt_host::read_grid_file(path, g, err)) return srt_set_error(SRT_EIO, "%s: %s", path, err.c_str());
  dims[0] = g.have_derivs ? 1 : 0; dims[1] = g.nspec; dims[2] = g.nx; dims[3] = g.ny; dims[4] = g.nz;
  memcpy(bounds, g.bounds, 6 * sizeof(double));
  memcpy(qs, g.qs, 4 * sizeof(double));
  memcpy(ms, g.ms, 4 * sizeof(double));
  const size_t n = g.F.size();
  *F = (double *)malloc(n * sizeof(double));
  if (!*F) return srt_set_error(SRT_ENOMEM, "grid of %zu values", n);
  memcpy(*F, g.F.data(), n * sizeof(double));
  if (derivs) {
    *derivs = nullptr;
    if (g.have_derivs) {
      *derivs = (double *)malloc(7 * n * sizeof(double));
      if (!*derivs) return srt_set_error(SRT_ENOMEM, "derivative blocks of %zu values", 7 * n);
      for (int a = 0; a < 7; ++a) memcpy(*derivs + (size_t)a * n, g.derivs[a].data(), n * sizeof(double));
    }
  }
  return SRT_OK;
}

extern "C" int srt_grid_file_write(const char *path, int binary, int nspec, int nx, int ny, int nz,
                                   const double bounds[6], const double *qs, const double *ms, const double *F,
                                   const double *derivs) {
  if (!path || !bounds || !qs || !ms || !F) return srt_set_error(SRT_EINVAL, "null argument");
  if (nspec < 1 || nspec > SRT_MAXSPEC || nx < 2 || ny < 2 || nz < 2) return srt_set_error(SRT_EINVAL, "bad grid shape");
  const size_t n = (size_t)nx * ny * nz * nspec;
  if (binary) {
    srt_host::GridFile g;
    g.nspec = nspec; g.nx = nx; g.ny = ny; g.nz = nz;
    g.have_derivs = derivs != nullptr;
    memcpy(g.bounds, bounds, sizeof g.bounds);
    for (int k = 0; k < nspec; ++k) { g.qs[k] = qs[k]; g.ms[k] = ms[k]; }
    g.F.assign(F, F + n);
    if (derivs)
      for (int a = 0; a < 7; ++a) g.derivs[a].assign(derivs + (size_t)a * n, derivs + (size_t)(a + 1) * n);
    std::string err;
    if (!srt_host::write_grid_binary(path, g, err)) return srt_set_error(SRT_EIO, "%s: %s", path, err.c_str());
    return SRT_OK;
  }
  // text, byte for byte what gcpm_dens_model_buildgrid.f95:302-327 writes: (5i10); (6es24.15e3); charges; masses;
  // then every value on its own record in es24.15e3 (16 significant digits -- the format's, not ours: use the
  // binary form where the last bit matters)
  FILE *f = fopen(path, "w");
  if (!f) return srt_set_error(SRT_EIO, "%s: cannot open for writing", path);
  std::vector<char> big(1 << 22);
  setvbuf(f, big.data(), _IOFBF, big.size());
  char num[25];
  fprintf(f, "%10d%10d%10d%10d%10d\n", derivs ? 1 : 0, nspec, nx, ny, nz);
  for (int k = 0; k < 6; ++k) { srt_host::format_es24(bounds[k], num); fputs(num, f); }
  fputc('\n', f);
  for (int k = 0; k < nspec; ++k) { srt_host::format_es24(qs[k], num); fputs(num, f); }
  fputc('\n', f);
  for (int k = 0; k < nspec; ++k) { srt_host::format_es24(ms[k], num); fputs(num, f); }
  fputc('\n', f);
  for (int blk = 0; blk < (derivs ? 8 : 1); ++blk) {
    const double *v = blk == 0 ? F : derivs + (size_t)(blk - 1) * n;
    for (size_t c = 0; c < n; ++c) {
      srt_host::format_es24(v[c], num);
      num[24] = '\n';
      fwrite(num, 1, 25, f);
    }
  }
  if (fclose(f) != 0) return srt_set_error(SRT_EIO, "%s: write failed", path);
  return SRT_OK;
}

extern "C" int srt_grid_file_convert(const char *in, const char *out_binary) {
  if (!in || !out_binary) return srt_set_error(SRT_EINVAL, "null argument");
  srt_host::GridFile g;
  std::string err;
  if (!srt_host::read_grid_file(in, g, err)) return srt_set_error(SRT_EIO, "%s: %s", in, err.c_str());
  if (!srt_host::write_grid_binary(out_binary, g, err)) return srt_set_error(SRT_EIO, "%s: %s", out_binary, err.c_str());
  return SRT_OK;
}

extern "C" int srt_grid_file_is_binary(const char *path) { return path && srt_host::is_binary_grid(path) ? 1 : 0; }

extern "C" int64_t srt_read_rays_file(const char *path, double **pos0, double **dir0, double **w0) {
  if (!path || !pos0 || !dir0 || !w0) return srt_set_error(SRT_EINVAL, "null argument");
  srt_host::ListReader r(path);
  if (!r.ok()) return srt_set_error(SRT_EIO, "%s: cannot open", path);
  std::vector<double> all;
  double v[7];
  while (r.read(7, v) == 7) all.insert(all.end(), v, v + 7); // read(infile,*) pos0, dir0, w ; EOF ends (driver:1146)
  int64_t n = (int64_t)(all.size() / 7);
  *pos0 = (double *)malloc(sizeof(double) * 3 * (n ? n : 1));
  *dir0 = (double *)malloc(sizeof(double) * 3 * (n ? n : 1));
  *w0 = (double *)malloc(sizeof(double) * (n ? n : 1));
  for (int64_t i = 0; i < n; ++i) {
    for (int c = 0; c < 3; ++c) {
      (*pos0)[3 * i + c] = all[7 * i + c];
      (*dir0)[3 * i + c] = all[7 * i + 3 + c];
    }
    (*w0)[i] = all[7 * i + 6];
  }
  return n;
}

// .ray reader: the inverse of srt_write_ray_file (what matlab/readrayoutput.m does for the damping scripts).  Every record
// is a run of blank-separated numbers -- 2 + 17 + 1 + 4 nspec of them -- so the whole file is parsed as one token stream by
// all host cores and cut into records; the records of a ray are consecutive.
extern "C" int64_t srt_read_ray_file(const char *path, int32_t *nspec_out, double qs[4], double ms[4], int64_t *nrecords,
                                     int64_t **raynum, int32_t **stopcond, int32_t **kept, double **w0, double **rows) {
  if (!path || !nspec_out || !qs || !ms || !nrecords || !raynum || !stopcond || !kept || !w0 || !rows)
    return srt_set_error(SRT_EINVAL, "null argument");
  *nspec_out = 0;
  *nrecords = 0;
  *raynum = nullptr, *stopcond = nullptr, *kept = nullptr, *w0 = nullptr, *rows = nullptr;
  struct stat st;
  if (stat(path, &st) != 0) return srt_set_error(SRT_EIO, "%s: cannot open", path);
  std::vector<double> all;
  std::string err;
  if (st.st_size > 0 && !srt_host::read_all_numbers(path, all, err)) return srt_set_error(SRT_EIO, "%s: %s", path, err.c_str());
  int64_t nrec = 0, nray = 0;
  int nspec = 0;
  if (!all.empty()) {
    if (all.size() < 20) return srt_set_error(SRT_EIO, "%s: truncated record", path);
    nspec = (int)all[19];
    if (nspec < 1 || nspec > SRT_MAXSPEC || (double)nspec != all[19]) return srt_set_error(SRT_EIO, "%s: nspec = %g in the first record", path, all[19]);
    const size_t per = 20 + (size_t)4 * nspec;
    if (all.size() % per) return srt_set_error(SRT_EIO, "%s: %zu numbers are not a whole number of %zu-number records", path, all.size(), per);
    nrec = (int64_t)(all.size() / per);
    for (int64_t r = 0; r < nrec; ++r) {
      const double *v = all.data() + (size_t)r * per;
      if ((int)v[19] != nspec) return srt_set_error(SRT_EIO, "%s: record %lld has nspec %g", path, (long long)r + 1, v[19]);
      if (r == 0 || v[0] != all[(size_t)(r - 1) * per] || v[2] == 0.0) ++nray; // (a ray's first record is its row 0: t = 0)
    }
  }
  const int64_t na = nray ? nray : 1, nr = nrec ? nrec : 1;
  *raynum = (int64_t *)malloc(sizeof(int64_t) * na);
  *stopcond = (int32_t *)malloc(sizeof(int32_t) * na);
  *kept = (int32_t *)malloc(sizeof(int32_t) * na);
  *w0 = (double *)malloc(sizeof(double) * na);
  *rows = (double *)malloc(sizeof(double) * SRT_ROW * nr);
  if (!*raynum || !*stopcond || !*kept || !*w0 || !*rows) {
    free(*raynum), free(*stopcond), free(*kept), free(*w0), free(*rows);
    *raynum = nullptr, *stopcond = nullptr, *kept = nullptr, *w0 = nullptr, *rows = nullptr;
    return srt_set_error(SRT_ENOMEM, "%s: %lld records", path, (long long)nrec);
  }
  for (int s = 0; s < 4; ++s) qs[s] = ms[s] = 0.0;
  const size_t per = 20 + (size_t)4 * (nspec ? nspec : 1);
  int64_t ray = -1;
  for (int64_t r = 0; r < nrec; ++r) {
    const double *v = all.data() + (size_t)r * per;
    if (r == 0 || v[0] != all[(size_t)(r - 1) * per] || v[2] == 0.0) { // (so files appended to, which repeat ray numbers, keep their rays apart)
      ++ray;
      (*raynum)[ray] = (int64_t)v[0];
      (*stopcond)[ray] = (int32_t)v[1];
      (*kept)[ray] = 0;
      (*w0)[ray] = v[18];
    }
    (*kept)[ray] += 1;
    double *row = *rows + (size_t)r * SRT_ROW;
    for (int c = 0; c < 16; ++c) row[c] = v[2 + c]; // t, pos, vprel, vgrel, n, B0
    for (int k = 0; k < 4; ++k) row[16 + k] = k < nspec ? v[20 + 2 * nspec + k] : 0.0; // Ns
    if (r == 0)
      for (int k = 0; k < nspec; ++k) {
        qs[k] = v[20 + k];
        ms[k] = v[20 + nspec + k];
      }
  }
  *nspec_out = nspec;
  *nrecords = nrec;
  return nray;
}

// record format of raytracer_driver.f95:1197-1217:
//   (i10, i10, 17es24.15e3, i10) raynum, stopcond, t, pos, vprel, vgrel, n, B0, w, nspec
//   then nspec x es24.15e3 for each of qs, ms, Ns, nus
// es24.15e3 into exactly 24 characters (no terminator): one snprintf, exponent widened to three digits by hand
static inline void put_es24(double v, char *out) {
  char tmp[40];
  if (std::isnan(v) || std::isinf(v)) {
    char t[25];
    srt_host::format_es24(v, t);
    memcpy(out, t, 24);
    return;
  }
  // [-]d.ddddddddddddddde+XX[X]: std::to_chars with a precision is correctly rounded like printf's %.15E (and 5x faster)
  const std::to_chars_result tr = std::to_chars(tmp, tmp + sizeof tmp - 1, v, std::chars_format::scientific, 15);
  const int n = (int)(tr.ptr - tmp);
  tmp[n] = 0;
  const char *e = (const char *)memchr(tmp, 'e', (size_t)n);
  const int mant = (int)(e - tmp);
  int ex = atoi(e + 2);
  const int len = mant + 5;
  char *q = out;
  for (int k = len; k < 24; ++k) *q++ = ' ';
  memcpy(q, tmp, (size_t)mant);
  q += mant;
  *q++ = 'E';
  *q++ = e[1];
  *q++ = (char)('0' + (ex / 100) % 10);
  *q++ = (char)('0' + (ex / 10) % 10);
  *q++ = (char)('0' + ex % 10);
}

// Records are fixed-length (raytracer_driver.f95:1197-1217: i10, i10, 17 es24.15e3, i10, 4 nspec es24.15e3, newline), so a
// ray's file offset is known from the kept-row counts before it: all host cores format disjoint ray ranges and pwrite them
// in place (SRT_IO_THREADS=1: one thread).  At BASELINE config[2] (1 M rays, 13 kept rows each) the file is 10.7 GB of text.
extern "C" int srt_write_ray_file(const char *path, int append, int64_t raynum0, int64_t nrays, const srt_params *p,
                                  int nspec, const double *qs, const double *ms, const double *w0,
                                  const double *rows, const int32_t *nrows, const int32_t *stopcond) {
  if (!path || !p || !qs || !ms || !w0 || !rows || !nrows || !stopcond) return srt_set_error(SRT_EINVAL, "null argument");
  if (nspec < 1 || nspec > SRT_MAXSPEC) return srt_set_error(SRT_EINVAL, "nspec out of range");
  // the record's ray number is an i10 field (raytracer_driver.f95:1197): beyond ten digits Fortran prints asterisks and the
  // fixed-length record could not hold the number -- refused rather than shifted or truncated
  if (raynum0 < 0 || nrays < 0 || raynum0 + nrays - 1 > 9999999999ll)
    return srt_set_error(SRT_EINVAL, "ray numbers %lld .. %lld do not fit the record's i10 field", (long long)raynum0, (long long)(raynum0 + nrays - 1));
  const int slots = srt_rows_per_ray(p);
  const int per = p->outputper < 1 ? 1 : p->outputper;
  const size_t L = 20 + 17 * 24 + 10 + (size_t)4 * nspec * 24 + 1;
  const int fd = open(path, O_WRONLY | O_CREAT | (append ? 0 : O_TRUNC), 0644);
  if (fd < 0) return srt_set_error(SRT_EIO, "%s: cannot open for writing", path);
  off_t base = 0;
  if (append) {
    struct stat st;
    if (fstat(fd, &st) == 0) base = st.st_size;
  }
  auto kept_of = [&](int64_t r) {
    int k = (nrows[r] + per - 1) / per;
    return k > slots ? slots : (k < 0 ? 0 : k);
  };
  int nth = (int)std::thread::hardware_concurrency();
  if (const char *e = getenv("SRT_IO_THREADS")) nth = atoi(e);
  if (nth < 1) nth = 1;
  if (nth > 64) nth = 64;
  if ((int64_t)nth > nrays) nth = nrays > 0 ? (int)nrays : 1;
  // contiguous ray ranges, one per thread; prefix[r] = records before ray r
  std::vector<int64_t> prefix((size_t)nrays + 1, 0);
  for (int64_t r = 0; r < nrays; ++r) prefix[r + 1] = prefix[r] + kept_of(r);
  std::vector<int64_t> first(nth + 1);
  for (int t = 0; t <= nth; ++t) first[t] = (nrays * t) / nth;
  std::vector<char> consts((size_t)8 * 24); // qs and ms of the record, formatted once
  for (int s = 0; s < nspec; ++s) {
    put_es24(qs[s], consts.data() + 24 * s);
    put_es24(ms[s], consts.data() + 24 * (nspec + s));
  }
  char zero[24];
  put_es24(0.0, zero);
  std::vector<int> fail(nth, 0);
  auto work = [&](int t) {
    const size_t CH = (size_t)1 << 22;
    std::vector<char> buf(CH + L);
    size_t fill = 0;
    off_t off = base + (off_t)((size_t)prefix[first[t]] * L);
    auto flush = [&]() {
      size_t done = 0;
      while (done < fill) {
        ssize_t w = pwrite(fd, buf.data() + done, fill - done, off + (off_t)done);
        if (w < 0 && errno == EINTR) continue; // interrupted before anything was written: again
        if (w <= 0) {
          fail[t] = 1;
          return;
        }
        done += (size_t)w;
      }
      off += (off_t)fill;
      fill = 0;
    };
    for (int64_t r = first[t]; r < first[t + 1] && !fail[t]; ++r) {
      const int kept = kept_of(r);
      char head[24], wbuf[24];
      snprintf(head, sizeof head, "%10lld%10d", (long long)(raynum0 + r), (int)stopcond[r]);
      put_es24(w0[r], wbuf);
      for (int s = 0; s < kept; ++s) {
        const double *row = rows + ((size_t)r * slots + s) * SRT_ROW;
        char *q = buf.data() + fill;
        memcpy(q, head, 20);
        q += 20;
        for (int c = 0; c < 16; ++c, q += 24) put_es24(row[c], q);
        memcpy(q, wbuf, 24);
        q += 24;
        char ns[12];
        snprintf(ns, sizeof ns, "%10d", nspec);
        memcpy(q, ns, 10);
        q += 10;
        memcpy(q, consts.data(), (size_t)2 * nspec * 24);
        q += (size_t)2 * nspec * 24;
        for (int k = 0; k < nspec; ++k, q += 24) put_es24(row[16 + k], q);
        for (int k = 0; k < nspec; ++k, q += 24) memcpy(q, zero, 24);
        *q++ = '\n';
        fill += L;
        if (fill >= CH) flush();
      }
    }
    if (fill && !fail[t]) flush();
  };
  if (nth == 1) work(0);
  else {
    std::vector<std::thread> th;
    for (int t = 0; t < nth; ++t) th.emplace_back(work, t);
    for (auto &x : th) x.join();
  }
  const int rc = close(fd);
  for (int t = 0; t < nth; ++t)
    if (fail[t]) return srt_set_error(SRT_EIO, "%s: write failed", path);
  if (rc != 0) return srt_set_error(SRT_EIO, "%s: close failed", path);
  return SRT_OK;
}
