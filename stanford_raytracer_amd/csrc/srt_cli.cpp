// srt_cli.cpp -- `raytracer`: command-line front end with the reference's flag grammar, ray-input format and
// .ray output format (fortran/raytracer_driver.f95), driving the batched HIP path through the C ABI.
//
// Grammar (fortran/util.f95:53-84 getopt_named): every argument is --name=value; the FIRST argument whose
// text between column 3 and the first '=' equals the name wins; unknown flags are ignored.  Numeric integer
// flags are read as reals and floored (driver:195-196).  Where the reference leaves an unset flag
// uninitialised we fail with a message instead (documented divergence).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/srt.h"

static int g_argc;
static char **g_argv;

static bool getopt_named(const char *name, std::string &out) {
  for (int i = 1; i < g_argc; ++i) {
    const char *a = g_argv[i];
    const char *eq = strchr(a, '=');
    if (!eq || eq - a < 2) continue;
    std::string key(a + 2, eq - (a + 2));
    if (key == name) {
      out = eq + 1;
      return true;
    }
  }
  return false;
}
static bool get_real(const char *name, double &v) {
  std::string s;
  if (!getopt_named(name, s)) return false;
  for (auto &ch : s)
    if (ch == 'd' || ch == 'D') ch = 'e';
  v = strtod(s.c_str(), nullptr);
  return true;
}
static bool get_int(const char *name, int &v) {
  double d;
  if (!get_real(name, d)) return false;
  v = (int)floor(d);
  return true;
}
static void need(bool ok, const char *name) {
  if (!ok) {
    fprintf(stderr, "raytracer: required flag --%s=... is missing\n", name);
    exit(2);
  }
}
#define CHECK(call)                                                      \
  do {                                                                   \
    int rc_ = (call);                                                    \
    if (rc_ != 0) {                                                      \
      fprintf(stderr, "raytracer: %s failed: %s\n", #call, srt_last_error()); \
      return 1;                                                          \
    }                                                                    \
  } while (0)

int main(int argc, char **argv) {
  g_argc = argc;
  g_argv = argv;
  if (argc == 1) {
    puts("Usage:\n  raytracer --param1=value1 --param2=value2 ...\n"
         "  --dt0 --dtmax --tmax --root --fixedstep --maxerr --maxsteps --minalt\n"
         "  --inputraysfile --outputfile --outputper\n"
         "  --modelnum  (1) Ngo model  (3) interpolated model (gridded)  (4) interpolated model (scattered)\n"
         "  model 1: --ngo_configfile --yearday --milliseconds_day --use_tsyganenko=0 --use_igrf=0|1\n"
         "  model 3: --interp_interpfile --yearday --milliseconds_day --use_tsyganenko=0 --use_igrf=0|1\n"
         "  model 4: model 3 flags + --scattered_interp_window_scale --scattered_interp_order\n"
         "           --scattered_interp_exact --scattered_interp_local_window_scale\n"
         "  extra:   --device=N  --first_attempt_policy=0|1  --chunk_rays=N  --ray_order=0|1\n"
         "  tools:   --grid2bin_in=<text grid> --grid2bin_out=<binary grid>   (convert and exit; --interp_interpfile\n"
         "           accepts either form)");
    return 0;
  }
  {
    std::string gin, gout; // model-3 grid: text -> binary side-format (no GPU needed)
    if (getopt_named("grid2bin_in", gin)) {
      need(getopt_named("grid2bin_out", gout), "grid2bin_out");
      CHECK(srt_grid_file_convert(gin.c_str(), gout.c_str()));
      return 0;
    }
  }
  srt_params p;
  memset(&p, 0, sizeof p);
  int modelnum = 0, device = 0, chunk = 0;
  std::string rays_path, out_path, file;
  need(get_real("dt0", p.dt0), "dt0");
  need(get_real("tmax", p.tmax), "tmax");
  need(get_int("root", p.root), "root");
  need(get_int("fixedstep", p.fixedstep), "fixedstep");
  need(get_int("maxsteps", p.maxsteps), "maxsteps");
  need(get_real("minalt", p.minalt), "minalt");
  need(getopt_named("inputraysfile", rays_path), "inputraysfile");
  need(getopt_named("outputfile", out_path), "outputfile");
  need(get_int("modelnum", modelnum), "modelnum");
  if (p.fixedstep == 0) {
    need(get_real("dtmax", p.dtmax), "dtmax");
    need(get_real("maxerr", p.maxerr), "maxerr");
  } else {
    get_real("dtmax", p.dtmax);
    get_real("maxerr", p.maxerr);
  }
  p.outputper = 1;
  get_int("outputper", p.outputper);
  get_int("device", device);
  get_int("first_attempt_policy", p.first_attempt_policy);
  get_int("ray_order", p.ray_order);
  get_int("chunk_rays", chunk);
  int yearday = 0, msec = 0, use_tsy = 0, use_igrf = 0;
  need(get_int("yearday", yearday), "yearday");
  need(get_int("milliseconds_day", msec), "milliseconds_day");
  get_int("use_tsyganenko", use_tsy);
  get_int("use_igrf", use_igrf);
  if (use_tsy != 0) {
    fprintf(stderr, "raytracer: --use_tsyganenko=1 (T04_s external field) is outside the accelerated path\n");
    return 2;
  }
  CHECK(srt_init(device));
  srt_model *m = nullptr;
  // FD step for dF/dx: delSP for the single-precision models, delDP otherwise (driver:251-252, :1158-1176)
  if (modelnum == 1) {
    need(getopt_named("ngo_configfile", file), "ngo_configfile");
    p.del = 1.0e-4;
    CHECK(srt_model_create_ngo(file.c_str(), yearday, msec, &m));
  } else if (modelnum == 3) {
    need(getopt_named("interp_interpfile", file), "interp_interpfile");
    p.del = 1.0e-6;
    printf(" Reading input file\n");
    CHECK(srt_model_create_interp_file(file.c_str(), yearday, msec, &m));
    printf(" Done\n");
  } else if (modelnum == 4) {
    need(getopt_named("interp_interpfile", file), "interp_interpfile");
    double ws = 0, lws = 0;
    int order = 0, exact = 0;
    need(get_real("scattered_interp_window_scale", ws), "scattered_interp_window_scale");
    need(get_int("scattered_interp_order", order), "scattered_interp_order");
    need(get_int("scattered_interp_exact", exact), "scattered_interp_exact");
    need(get_real("scattered_interp_local_window_scale", lws), "scattered_interp_local_window_scale");
    p.del = 1.0e-6;
    CHECK(srt_model_create_scattered_file(file.c_str(), yearday, msec, ws, order, exact, lws, &m));
  } else {
    fprintf(stderr, "raytracer: --modelnum=%d is not on the accelerated path (1, 3, 4 are)\n", modelnum);
    return 2;
  }
  if (use_igrf != 0) {
    std::string coeffs;
    getopt_named("igrf_coeffs", coeffs); // ours: table of Gauss coefficients (default: shipped beside the library)
    CHECK(srt_model_set_field(m, 1, 0, coeffs.empty() ? nullptr : coeffs.c_str()));
  }
  double *pos0 = nullptr, *dir0 = nullptr, *w0 = nullptr;
  int64_t nrays = srt_read_rays_file(rays_path.c_str(), &pos0, &dir0, &w0);
  if (nrays < 0) {
    fprintf(stderr, "raytracer: %s\n", srt_last_error());
    return 1;
  }
  const int slots = srt_rows_per_ray(&p);
  double qs[SRT_MAXSPEC], ms[SRT_MAXSPEC];
  srt_model_species(m, qs, ms);
  const int nspec = srt_model_nspec(m);
  // bound host/device memory: at most ~2 GiB of trajectory rows per launch
  int64_t per = (int64_t)slots * SRT_ROW * 8;
  int64_t maxchunk = chunk > 0 ? chunk : ((int64_t)2 << 30) / (per > 0 ? per : 1);
  if (maxchunk < 64) maxchunk = 64;
  int64_t total_steps = 0;
  // the reference opens the output with status="replace" even when there are no rays
  {
    FILE *f = fopen(out_path.c_str(), "w");
    if (!f) {
      fprintf(stderr, "raytracer: cannot open %s\n", out_path.c_str());
      return 1;
    }
    fclose(f);
  }
  for (int64_t lo = 0; lo < nrays; lo += maxchunk) {
    int64_t n = nrays - lo < maxchunk ? nrays - lo : maxchunk;
    std::vector<double> rows((size_t)n * slots * SRT_ROW);
    std::vector<int32_t> nrows(n), stop(n);
    int64_t steps = 0;
    CHECK(srt_trace_batch(m, &p, n, pos0 + 3 * lo, dir0 + 3 * lo, w0 + lo, rows.data(), nrows.data(), stop.data(), &steps));
    total_steps += steps;
    CHECK(srt_write_ray_file(out_path.c_str(), 1, lo + 1, n, &p, nspec, qs, ms, w0 + lo, rows.data(), nrows.data(), stop.data()));
  }
  printf(" %lld rays, %lld accepted steps\n", (long long)nrays, (long long)total_steps);
  srt_free(pos0);
  srt_free(dir0);
  srt_free(w0);
  srt_model_destroy(m);
  return 0;
}
