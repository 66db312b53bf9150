// srt_host.hpp -- host-side helpers of libsrt_hip: file formats of the boundary and run constants.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace srt_host {

// dipole tilt angle mu of xform_double/T4.f95:7-18 for itime = (yearday, msec)
double dipole_tilt(int yearday, int msec);
// use_igrf = 1: Gauss coefficients for the run's date, Schmidt-normalised, the recursion constants and the GEO->GSM
// matrix (geopack2008.for RECALC_08 + SUN_08 with V_sw = (-400,0,0), i.e. tsy_recalc of geopack0508_adapter.for:21-30),
// default REAL arithmetic like the Fortran.  coeff_file = table of IAGA coefficients (data/igrf_coeffs.txt).
bool igrf_setup(const char *coeff_file, int yearday, int msec, float G[105], float H[105], float REC[105], float A[9],
                float *psi, std::string &err); // psi: what the adapters hand to T04_s as PS (first word of COMMON /GEOPACK1/ = ST0)

// newray.in card file (ngo_dens_model.f95:45-118; field names per matlab/unused/parse_newray_cards.m)
struct NgoConfig {
  int num = 0, kducts = 0;
  double dsrrng = 0, dsrlat = 0, dsdens = 0, last_latitu = 0;
  double therm = 0, rbase = 0, ane0 = 0, alpha0[5] = {0, 0, 0, 0, 0}, rzero = 0, scbot = 0;
  double lk = 0, expk = 0, ddk = 0, rconsn = 0, scr = 0;
  double l0[10] = {0}, def[10] = {0}, dd[10] = {0}, rducln[10] = {0}, hducln[10] = {0}, rducun[10] = {0},
         hducun[10] = {0}, rducls[10] = {0}, hducls[10] = {0}, rducus[10] = {0}, hducus[10] = {0}, sidedu[10] = {0};
};
bool read_newray(const char *path, NgoConfig &cfg, std::string &err);

// model-3 grid file (interp_dens_model_adapter.f95:58-117)
struct GridFile {
  int compder = 0, nspec = 0, nx = 0, ny = 0, nz = 0;
  bool have_derivs = false;
  double bounds[6] = {0}, qs[4] = {0}, ms[4] = {0};
  std::vector<double> F;
  std::vector<double> derivs[7];
};
// text (parsed by all host cores) or the binary side-format "SRTGRID1" (detected by its magic)
bool read_grid_file(const char *path, GridFile &g, std::string &err);
bool is_binary_grid(const char *path);
// binary side-format: 144-byte header {magic, compder, nspec, nx, ny, nz, bounds[6], qs[4], ms[4]} followed by the
// value blocks of the text format as raw little-endian doubles in the same order
bool write_grid_binary(const char *path, const GridFile &g, std::string &err);

// Fortran list-directed numeric reader: a READ starts on a new record and continues over following
// records until its list is satisfied; blanks and commas separate; 'd' exponents accepted.
class ListReader {
public:
  explicit ListReader(const char *path);
  ~ListReader();
  bool ok() const { return f_ != nullptr; }
  // reads n values for one READ statement; returns how many were obtained
  int64_t read(int64_t n, double *out);
private:
  void *f_;
  std::vector<char> line_;
};

// every numeric token of a text file in order, parsed by all host cores
bool read_all_numbers(const char *path, std::vector<double> &all, std::string &err);

// es24.15e3 edit descriptor
void format_es24(double v, char out[25]);

} // namespace srt_host

namespace srt_host {
// model-4 scattered sample file (scattered_interp_dens_model_adapter.f95:85-133) prepared for the device:
// duplicates dropped (:160-163), nearest-sample distance per sample outside the Earth (:167-203),
// samples binned into a uniform grid with cell edge = maxnearest*window_scale and sorted by cell.
bool is_binary_points(const char *path);
bool write_points_file(const char *path, bool binary, int nspec, int64_t npts, const double bounds[6], const double *qs, const double *ms,
                       const double *rec, std::string &err);
struct ScatteredHost {
  int nspec = 0, npts = 0;
  double qs[4] = {0}, ms[4] = {0};
  double maxnearest = 0, radius = 0;
  double origin[3] = {0, 0, 0}, inv_cell = 0;
  int dims[3] = {1, 1, 1};
  std::vector<double> pts;       // [npts][8]
  std::vector<int> cell_start;   // [ncells+1]
};
// cell_scale: edge of the query grid's cells in units of the search radius (>= 1; the device's candidate blocks reach
// radius * cell_scale from their centre and are built from the 27 cells around it)
bool build_scattered(const char *path, double window_scale, ScatteredHost &out, std::string &err, double cell_scale = 1.0,
                     long long root_file_index = -1);
} // namespace srt_host
