// srt_cli.cpp -- `raytracer`: command-line front end with the reference's flag grammar, ray-input format and
// .ray output format (fortran/raytracer_driver.f95), driving the batched HIP path through the C ABI.
//
// Grammar (fortran/util.f95:53-84 getopt_named): every argument is --name=value; the FIRST argument whose
// text between column 3 and the first '=' equals the name wins; unknown flags are ignored.  Numeric integer
// flags are read as reals and floored (driver:195-196).  Where the reference leaves an unset flag
// uninitialised we fail with a message instead (documented divergence).
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <future>
#include <memory>
#include <string>
#include <thread>
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

static double now_s() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// model flags of the driver (raytracer_driver.f95:256-770) -> model handle; del = the driver's FD step for the model
static int make_model(int device, srt_model **out, double *del) {
  int modelnum = 0;
  std::string file;
  need(get_int("modelnum", modelnum), "modelnum");
  srt_params p; // only p.del is set here
  memset(&p, 0, sizeof p);
  int yearday = 0, msec = 0, use_tsy = 0, use_igrf = 0;
  need(get_int("yearday", yearday), "yearday");
  need(get_int("milliseconds_day", msec), "milliseconds_day");
  get_int("use_tsyganenko", use_tsy);
  get_int("use_igrf", use_igrf);
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
    // ours (opt-in): --scattered_interp_root_sample=<record number, 1 = the file's first sample>: the sample at the root of the
    // reference's kd-tree keeps a stored spacing of 0 there (kdtree_mod.f95:386-444); default: none
    double root = 0;
    get_real("scattered_interp_root_sample", root);
    CHECK(srt_model_create_scattered_file_root(file.c_str(), yearday, msec, ws, order, exact, lws, (int64_t)floor(root) - 1, &m));
  } else {
    fprintf(stderr, "raytracer: --modelnum=%d is not on the accelerated path (1, 3, 4 are)\n", modelnum);
    return 2;
  }
  if (use_igrf != 0 || use_tsy != 0) {
    std::string coeffs;
    getopt_named("igrf_coeffs", coeffs); // ours: table of Gauss coefficients (default: shipped beside the library)
    if (use_tsy != 0) { // raytracer_driver.f95:292-341
      const char *names[10] = {"tsyganenko_Pdyn", "tsyganenko_Dst", "tsyganenko_ByIMF", "tsyganenko_BzIMF", "tsyganenko_W1",
                               "tsyganenko_W2",   "tsyganenko_W3",  "tsyganenko_W4",    "tsyganenko_W5",    "tsyganenko_W6"};
      double parmod[10];
      for (int k = 0; k < 10; ++k) need(get_real(names[k], parmod[k]), names[k]);
      CHECK(srt_model_set_tsyganenko_params(m, parmod));
    }
    CHECK(srt_model_set_field(m, use_igrf != 0, use_tsy != 0, coeffs.empty() ? nullptr : coeffs.c_str()));
  }
  *out = m;
  *del = p.del;
  return 0;
}

int main(int argc, char **argv) {
  g_argc = argc;
  g_argv = argv;
  if (argc == 1) {
    puts("Usage:\n  raytracer --param1=value1 --param2=value2 ...\n"
         "  --dt0 --dtmax --tmax --root --fixedstep --maxerr --maxsteps --minalt\n"
         "  --inputraysfile --outputfile --outputper\n"
         "  --modelnum  (1) Ngo model  (3) interpolated model (gridded)  (4) interpolated model (scattered)\n"
         "  model 1: --ngo_configfile --yearday --milliseconds_day --use_tsyganenko=0|1 --use_igrf=0|1 --tsyganenko_Pdyn .. _W6\n"
         "  model 3: --interp_interpfile --yearday --milliseconds_day --use_tsyganenko=0|1 --use_igrf=0|1\n"
         "  model 4: model 3 flags + --scattered_interp_window_scale --scattered_interp_order\n"
         "           --scattered_interp_exact --scattered_interp_local_window_scale\n"
         "           [--scattered_interp_root_sample=N: record N of the file is the root of the reference's kd-tree (spacing 0)]\n"
         "  extra:   --device=N | --devices=0,1,..  --chunk_rays=N  --ray_order=0|1  --timing=1 (wall clock per phase)\n"
         "           --first_attempt_policy=1|0: error estimate of a ray's first adaptive attempt, where the reference reads an\n"
         "             unset variable: 1 (default) = from the k term alone, as the reference's gfortran build behaves;\n"
         "             0 = NaN => accepted at dt0, dt not grown, as a flang build behaves (the goldens of this repository)\n"
         "  tools:   --grid2bin_in=<text grid> --grid2bin_out=<binary grid>   (convert and exit; --interp_interpfile\n"
         "           accepts either form);  --pts2bin_in / --pts2bin_out: the same for model-4 sample files\n"
         "           --buildgrid=1 --filename=<out> --minx .. --maxz --nx --ny --nz --compder=0|1 (the reference's regular grid\n"
         "           builder gcpm_dens_model_buildgrid with the model of --modelnum in place of GCPM; --binary=1)\n"
         "           --buildsamples=1 --filename=<out> --minx .. --maxz --n_initial_uniform ... (the reference's random grid\n"
         "           builder with the model of --modelnum in place of GCPM; --seed, --binary=1)\n"
         "           --damping_out=<file>: hot-plasma damping along the kept rows (raynum, row, t, rate, magnitude, flag);\n"
         "           with --damping_in=<.ray file> instead of tracing: along the records of an existing file");
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
  {
    std::string pin, pout; // model-4 sample file: text -> binary side-format (no GPU needed)
    if (getopt_named("pts2bin_in", pin)) {
      need(getopt_named("pts2bin_out", pout), "pts2bin_out");
      CHECK(srt_points_file_convert(pin.c_str(), pout.c_str()));
      return 0;
    }
  }
  {
    // the reference's gcpm_dens_model_buildgrid with the model of --modelnum in place of GCPM: same flags
    // (gcpm_dens_model_buildgrid.f95:42-160: --minx .. --maxz, --nx --ny --nz (reals, floored), --compder, --filename) and
    // the builder's exact text layout (:302-327, srt_grid_file_write); ours: --binary=1 (SRTGRID1), --device
    std::string flag;
    if (getopt_named("buildgrid", flag) && flag != "0") {
      std::string out;
      need(getopt_named("filename", out), "filename");
      double b[6], v = 0;
      const char *bn[6] = {"minx", "maxx", "miny", "maxy", "minz", "maxz"};
      for (int k = 0; k < 6; ++k) need(get_real(bn[k], b[k]), bn[k]);
      int n3[3] = {0, 0, 0}, compder = 0, device = 0, binary = 0;
      const char *nn[3] = {"nx", "ny", "nz"};
      for (int k = 0; k < 3; ++k) {
        need(get_real(nn[k], v), nn[k]);
        n3[k] = (int)floor(v);
        if (n3[k] < 2) {
          fprintf(stderr, "raytracer: --%s must be >= 2 (the interpolated model needs two nodes per axis)\n", nn[k]);
          return 2;
        }
      }
      if (get_real("compder", v)) compder = (int)floor(v);
      get_int("device", device);
      get_int("binary", binary);
      srt_model *m = nullptr;
      double del = 0.0;
      int rc = make_model(device, &m, &del);
      if (rc) return rc;
      const int nspec = srt_model_nspec(m);
      const size_t nval = (size_t)n3[0] * n3[1] * n3[2] * nspec;
      std::vector<double> F(nval), D(compder ? 7 * nval : 0);
      CHECK(srt_build_grid(m, compder ? 1 : 0, n3[0], n3[1], n3[2], b, F.data(), compder ? D.data() : nullptr));
      double qs[SRT_MAXSPEC], ms[SRT_MAXSPEC];
      srt_model_species(m, qs, ms);
      CHECK(srt_grid_file_write(out.c_str(), binary, nspec, n3[0], n3[1], n3[2], b, qs, ms, F.data(), compder ? D.data() : nullptr));
      printf(" %d x %d x %d nodes, %d species%s\n", n3[0], n3[1], n3[2], nspec, compder ? ", 7 derivative blocks" : "");
      srt_model_destroy(m);
      return 0;
    }
  }
  {
    // the reference's gcpm_dens_model_buildgrid_random with the model of --modelnum in place of GCPM: same flags
    // (gcpm_dens_model_buildgrid_random.f95:56-90), --filename = output; ours: --seed, --binary=1
    std::string fname;
    if (getopt_named("buildsamples", fname) && fname != "0") {
      std::string out;
      need(getopt_named("filename", out), "filename");
      srt_sampler_params sp;
      memset(&sp, 0, sizeof sp);
      const char *bn[6] = {"minx", "maxx", "miny", "maxy", "minz", "maxz"};
      for (int k = 0; k < 6; ++k) need(get_real(bn[k], sp.bounds[k]), bn[k]);
      double v = 0;
      if (get_real("n_zero_altitude", v)) sp.n_zero_altitude = (int64_t)floor(v);
      if (get_real("n_iri_pad", v)) sp.n_iri_pad = (int64_t)floor(v);
      if (get_real("n_initial_radial", v)) sp.n_initial_radial = (int64_t)floor(v);
      if (get_real("n_initial_uniform", v)) sp.n_initial_uniform = (int64_t)floor(v);
      sp.adaptive_nmax = 100000; // :92
      if (get_real("adaptive_nmax", v)) sp.adaptive_nmax = (int64_t)floor(v);
      get_real("initial_tol", sp.initial_tol);
      get_int("max_recursion", sp.max_recursion);
      if (get_real("seed", v)) sp.seed = (uint64_t)v;
      int device = 0, binary = 0;
      get_int("device", device);
      get_int("binary", binary);
      srt_model *m = nullptr;
      double del = 0.0;
      int rc = make_model(device, &m, &del);
      if (rc) return rc;
      int64_t n = 0, counts[6];
      double *rec = nullptr;
      CHECK(srt_build_samples(m, &sp, 0, nullptr, &n, &rec, counts));
      double qs[SRT_MAXSPEC], ms[SRT_MAXSPEC];
      srt_model_species(m, qs, ms);
      CHECK(srt_points_file_write(out.c_str(), binary, srt_model_nspec(m), n, sp.bounds, qs, ms, rec));
      printf(" %lld samples: %lld radial, %lld uniform, %lld adaptive, %lld zero-altitude, %lld ionosphere\n", (long long)n,
             (long long)counts[1], (long long)counts[2], (long long)counts[3], (long long)counts[4], (long long)counts[5]);
      srt_free(rec);
      srt_model_destroy(m);
      return 0;
    }
  }
  {
    // ours: --damping_in=<existing .ray file> --damping_out=<file>: the MATLAB post-processor (matlab/damping/test_dampray.m)
    // on a file this program or the reference's driver wrote, without tracing anything
    std::string din;
    if (getopt_named("damping_in", din)) {
      std::string dout;
      need(getopt_named("damping_out", dout), "damping_out");
      srt_damping_params dpar;
      memset(&dpar, 0, sizeof dpar);
      get_int("damping_dist", dpar.dist);
      get_int("damping_mode", dpar.mode);
      get_real("damping_Ne_h", dpar.Ne_h);
      double kTeV = 0;
      if (get_real("damping_kT_eV", kTeV)) dpar.kT = kTeV * 1.60217646e-19;
      get_real("damping_tol", dpar.tol);
      int device = 0;
      get_int("device", device);
      int32_t nspec = 0;
      double qs[4], ms[4], *w0 = nullptr, *packed = nullptr;
      int64_t nrec = 0, *raynum = nullptr;
      int32_t *stop = nullptr, *kept = nullptr;
      const int64_t n = srt_read_ray_file(din.c_str(), &nspec, qs, ms, &nrec, &raynum, &stop, &kept, &w0, &packed);
      if (n < 0) {
        fprintf(stderr, "raytracer: %s\n", srt_last_error());
        return 1;
      }
      FILE *f = fopen(dout.c_str(), "w");
      if (!f) {
        fprintf(stderr, "raytracer: cannot open %s\n", dout.c_str());
        return 1;
      }
      if (n > 0) {
        CHECK(srt_init(device));
        int slots = 1;
        for (int64_t i = 0; i < n; ++i) slots = kept[i] > slots ? kept[i] : slots;
        // the file's records are every row it kept: outputper = 1 with nrows = records of the ray
        std::vector<double> rows((size_t)n * slots * SRT_ROW, 0.0), rate((size_t)n * slots), mag((size_t)n * slots);
        std::vector<int32_t> flag((size_t)n * slots);
        int64_t off = 0;
        for (int64_t i = 0; i < n; ++i) {
          memcpy(rows.data() + (size_t)i * slots * SRT_ROW, packed + (size_t)off * SRT_ROW, sizeof(double) * SRT_ROW * (size_t)kept[i]);
          off += kept[i];
        }
        CHECK(srt_damping(&dpar, nspec, qs, ms, slots, 1, n, rows.data(), kept, w0, rate.data(), mag.data(), flag.data()));
        for (int64_t i = 0; i < n; ++i)
          for (int r = 0; r < kept[i]; ++r) {
            const size_t idx = (size_t)i * slots + r;
            fprintf(f, "%10lld%10d%25.15E%25.15E%25.15E%10d\n", (long long)raynum[i], r + 1, rows[idx * SRT_ROW], rate[idx], mag[idx], (int)flag[idx]);
          }
      }
      fclose(f);
      printf(" %lld rays, %lld records\n", (long long)n, (long long)nrec);
      srt_free(raynum), srt_free(stop), srt_free(kept), srt_free(w0), srt_free(packed);
      return 0;
    }
  }
  srt_params p;
  memset(&p, 0, sizeof p);
  int device = 0, chunk = 0;
  std::string rays_path, out_path;
  need(get_real("dt0", p.dt0), "dt0");
  need(get_real("tmax", p.tmax), "tmax");
  need(get_int("root", p.root), "root");
  need(get_int("fixedstep", p.fixedstep), "fixedstep");
  need(get_int("maxsteps", p.maxsteps), "maxsteps");
  need(get_real("minalt", p.minalt), "minalt");
  need(getopt_named("inputraysfile", rays_path), "inputraysfile");
  need(getopt_named("outputfile", out_path), "outputfile");
  if (p.fixedstep == 0) {
    need(get_real("dtmax", p.dtmax), "dtmax");
    need(get_real("maxerr", p.maxerr), "maxerr");
  } else {
    get_real("dtmax", p.dtmax);
    get_real("maxerr", p.maxerr);
  }
  p.outputper = 1;
  get_int("outputper", p.outputper);
  if (p.outputper < 1) p.outputper = 1; // the reference's mod(i-1, outputper) with outputper <= 0 is undefined; the library clamps too
  get_int("device", device);
  // The first adaptive attempt (SURVEY A-1).  Default 1 = the step pattern of the reference's own
  // toolchain (gfortran: MAX(k_term, NaN) = k_term, Makefile:3,10); 0 = a flang build's (NaN: accept, no growth).
  p.first_attempt_policy = 1;
  get_int("first_attempt_policy", p.first_attempt_policy);
  if (p.first_attempt_policy != 0 && p.first_attempt_policy != 1) {
    fprintf(stderr, "raytracer: --first_attempt_policy must be 0 or 1\n");
    return 2;
  }
  get_int("ray_order", p.ray_order);
  get_int("chunk_rays", chunk);
  // ours: --devices=0,1,.. = one host thread and one model replica per GPU, contiguous shards of the ray file
  // (ceil(n/ndev) rays each), records written in ray order.  Default: the one device of --device.
  std::vector<int> devices;
  {
    std::string dl;
    if (getopt_named("devices", dl)) {
      const char *c = dl.c_str();
      while (*c) {
        char *end = nullptr;
        long v = strtol(c, &end, 10);
        if (end == c) break;
        devices.push_back((int)v);
        c = (*end == ',') ? end + 1 : end;
        if (end && *end && *end != ',') break;
      }
    }
    if (devices.empty()) devices.push_back(device);
  }
  const int ndev = (int)devices.size();
  int timing = 0;
  get_int("timing", timing);
  const double t_start = now_s();
  double *pos0 = nullptr, *dir0 = nullptr, *w0 = nullptr;
  int64_t nrays = srt_read_rays_file(rays_path.c_str(), &pos0, &dir0, &w0);
  const double parse_s = now_s() - t_start;
  if (nrays < 0) {
    fprintf(stderr, "raytracer: %s\n", srt_last_error());
    return 1;
  }
  // ours: --damping_out=<file> [--damping_dist=0|1 --damping_mode=0|1 --damping_Ne_h --damping_kT_eV --damping_tol]
  std::string damp_path;
  srt_damping_params dpar;
  memset(&dpar, 0, sizeof dpar);
  if (getopt_named("damping_out", damp_path)) {
    get_int("damping_dist", dpar.dist);
    get_int("damping_mode", dpar.mode);
    get_real("damping_Ne_h", dpar.Ne_h);
    double kTeV = 0;
    if (get_real("damping_kT_eV", kTeV)) dpar.kT = kTeV * 1.60217646e-19;
    get_real("damping_tol", dpar.tol);
  }
  // the reference opens the output with status="replace" even when there are no rays
  {
    FILE *f = fopen(out_path.c_str(), "w");
    if (!f) {
      fprintf(stderr, "raytracer: cannot open %s\n", out_path.c_str());
      return 1;
    }
    fclose(f);
  }
  // one shard per device; shard k writes <out>.part<k> (or <out> itself when there is one device), concatenated below
  struct Shard {
    int device = 0;
    int64_t lo = 0, hi = 0, steps = 0;
    std::string out, damp;
    int rc = 0;
    double model_s = 0, trace_s = 0, write_s = 0; // --timing=1: wall clock of the phases (write_s runs beside the next trace)
    int64_t out_bytes = 0;
  };
  std::vector<Shard> shards(ndev);
  const int64_t per_dev = (nrays + ndev - 1) / ndev;
  for (int k = 0; k < ndev; ++k) {
    Shard &sh = shards[k];
    sh.device = devices[k];
    sh.lo = std::min<int64_t>((int64_t)k * per_dev, nrays);
    sh.hi = std::min<int64_t>(sh.lo + per_dev, nrays);
    sh.out = ndev == 1 ? out_path : out_path + ".part" + std::to_string(k);
    sh.damp = damp_path.empty() ? std::string() : (ndev == 1 ? damp_path : damp_path + ".part" + std::to_string(k));
  }
  auto run_shard = [&](Shard &sh) -> int {
    srt_model *m = nullptr;
    double del = 0.0;
    const double tm0 = now_s();
    int rc = make_model(sh.device, &m, &del);
    if (rc) return rc;
    sh.model_s = now_s() - tm0;
    srt_params q = p;
    q.del = del;
    const int slots = srt_rows_per_ray(&q);
    double qs[SRT_MAXSPEC], ms[SRT_MAXSPEC];
    srt_model_species(m, qs, ms);
    const int nspec = srt_model_nspec(m);
    // bound host/device memory: at most ~2 GiB of trajectory rows per launch
    int64_t per = (int64_t)slots * SRT_ROW * 8;
    int64_t maxchunk = chunk > 0 ? chunk : ((int64_t)2 << 30) / (per > 0 ? per : 1);
    if (maxchunk < 64) maxchunk = 64;
    if (ndev > 1) {
      FILE *f = fopen(sh.out.c_str(), "w");
      if (!f) {
        fprintf(stderr, "raytracer: cannot open %s\n", sh.out.c_str());
        return 1;
      }
      fclose(f);
    }
    // chunk k's records are formatted and written (all host cores, srt_write_ray_file) while chunk k+1 is on the GPU
    struct Chunk {
      std::vector<double> rows, rate, mag;
      std::vector<int32_t> nrows, stop, flag;
    };
    std::future<int> pending;
    auto finish = [&]() -> int {
      if (!pending.valid()) return 0;
      const int rc = pending.get();
      if (rc) fprintf(stderr, "raytracer: writing %s failed\n", sh.out.c_str());
      return rc;
    };
    for (int64_t lo = sh.lo; lo < sh.hi; lo += maxchunk) {
      int64_t n = sh.hi - lo < maxchunk ? sh.hi - lo : maxchunk;
      auto c = std::make_shared<Chunk>();
      c->rows.resize((size_t)n * slots * SRT_ROW);
      c->nrows.resize(n);
      c->stop.resize(n);
      int64_t steps = 0;
      const double tt0 = now_s();
      CHECK(srt_trace_batch(m, &q, n, pos0 + 3 * lo, dir0 + 3 * lo, w0 + lo, c->rows.data(), c->nrows.data(), c->stop.data(), &steps));
      sh.trace_s += now_s() - tt0;
      sh.steps += steps;
      const bool damp = !sh.damp.empty();
      if (damp) {
        // the MATLAB post-processor (matlab/damping/test_dampray.m) on the rows just traced: one record per kept row
        c->rate.resize((size_t)n * slots);
        c->mag.resize((size_t)n * slots);
        c->flag.resize((size_t)n * slots);
        CHECK(srt_damping(&dpar, nspec, qs, ms, slots, q.outputper, n, c->rows.data(), c->nrows.data(), w0 + lo, c->rate.data(), c->mag.data(), c->flag.data()));
      }
      if (int rc = finish()) return rc;
      const bool first = lo == sh.lo;
      pending = std::async(std::launch::async, [=, &sh, &q]() -> int {
        const double tw0 = now_s();
        if (srt_write_ray_file(sh.out.c_str(), 1, lo + 1, n, &q, nspec, qs, ms, w0 + lo, c->rows.data(), c->nrows.data(), c->stop.data())) return 1;
        sh.write_s += now_s() - tw0;
        if (damp) {
          FILE *f = fopen(sh.damp.c_str(), first ? "w" : "a");
          if (!f) return 1;
          for (int64_t i = 0; i < n; ++i) {
            const int kept = c->nrows[i] > 0 ? (c->nrows[i] - 1) / q.outputper + 1 : 0;
            for (int r = 0; r < kept && r < slots; ++r) {
              const size_t idx = (size_t)i * slots + r;
              fprintf(f, "%10lld%10d%25.15E%25.15E%25.15E%10d\n", (long long)(lo + i + 1), r * q.outputper + 1, c->rows[idx * SRT_ROW], c->rate[idx],
                      c->mag[idx], (int)c->flag[idx]);
            }
          }
          if (fclose(f) != 0) return 1;
        }
        return 0;
      });
    }
    if (int rc = finish()) return rc;
    srt_model_destroy(m);
    return 0;
  };
  if (ndev == 1) {
    shards[0].rc = run_shard(shards[0]);
  } else {
    std::vector<std::thread> th;
    for (int k = 0; k < ndev; ++k) th.emplace_back([&, k] { shards[k].rc = run_shard(shards[k]); });
    for (auto &t : th) t.join();
  }
  int64_t total_steps = 0;
  for (auto &sh : shards) {
    if (sh.rc) return sh.rc;
    total_steps += sh.steps;
  }
  if (ndev > 1) { // records in ray order: shard 0's file, then shard 1's, ...
    auto cat = [](const std::string &dst, const std::vector<std::string> &parts) -> bool {
      FILE *o = fopen(dst.c_str(), "w");
      if (!o) return false;
      std::vector<char> buf(1 << 22);
      for (auto &pn : parts) {
        FILE *i = fopen(pn.c_str(), "r");
        if (!i) continue; // an empty shard (fewer rays than devices) wrote nothing
        size_t got;
        while ((got = fread(buf.data(), 1, buf.size(), i)) > 0) fwrite(buf.data(), 1, got, o);
        fclose(i);
        remove(pn.c_str());
      }
      return fclose(o) == 0;
    };
    std::vector<std::string> parts, dparts;
    for (auto &sh : shards) {
      parts.push_back(sh.out);
      if (!sh.damp.empty()) dparts.push_back(sh.damp);
    }
    if (!cat(out_path, parts) || (!damp_path.empty() && !cat(damp_path, dparts))) {
      fprintf(stderr, "raytracer: cannot assemble %s\n", out_path.c_str());
      return 1;
    }
  }
  printf(" %lld rays, %lld accepted steps\n", (long long)nrays, (long long)total_steps);
  if (timing) {
    // ours (--timing=1): wall clock per phase, max over the shards (they run side by side); the writer of chunk k runs beside
    // the trace of chunk k + 1, so trace_s + write_s may exceed wall_s
    double model_s = 0, trace_s = 0, write_s = 0;
    for (auto &sh : shards) {
      model_s = std::max(model_s, sh.model_s);
      trace_s = std::max(trace_s, sh.trace_s);
      write_s = std::max(write_s, sh.write_s);
    }
    long long bytes = 0;
    if (FILE *f = fopen(out_path.c_str(), "rb")) {
      if (fseeko(f, 0, SEEK_END) == 0) bytes = (long long)ftello(f);
      fclose(f);
    }
    printf(" timing: {\"parse_s\": %.3f, \"model_s\": %.3f, \"trace_s\": %.3f, \"write_s\": %.3f, \"wall_s\": %.3f, \"out_bytes\": %lld, "
           "\"first_attempt_policy\": %d, \"devices\": %d}\n", parse_s, model_s, trace_s, write_s, now_s() - t_start, bytes,
           (int)p.first_attempt_policy, ndev);
  }
  srt_free(pos0);
  srt_free(dir0);
  srt_free(w0);
  return 0;
}
