/* srt.h -- C ABI of the MI355X-native many-ray Haselgrove integrator (libsrt_hip.so).
 *
 * Drop-in boundary for ONE path of rareid2/Stanford_Raytracer: the ray loop of
 * fortran/raytracer_driver.f95:1144-1232, i.e. the per-ray call
 *     call raytracer_run(pos,time,vprel,vgrel,n,B0,qs,ms,Ns,nus,stopcond, pos0,dir0,w,dt0,dtmax,
 *                        maxerr,maxsteps,minalt,root,tmax,fixedstep,del,funcPlasmaParams,data,
 *                        raytracer_stopconditions)            (signature fortran/raytracer.f95:609-642)
 * becomes one batched call, srt_trace_batch().  The reference's plugin callback
 *     subroutine funcPlasmaParams(x, qs, Ns, ms, nus, B0, funcPlasmaParamsData)   (raytracer.f95:121-129)
 * cannot be a host callback on a GPU path; its three in-scope implementations are selected by the
 * model handle instead (modelnum 1 / 3 / 4 of raytracer_driver.f95:256-770) and are exposed for
 * point queries through srt_plasma_params().
 *
 * Conventions: plain pointers and sizes, no C++ or torch types.  All arrays are HOST memory unless a
 * function name ends in _device.  Vectors are AoS: pos0[i*3+c].  Every function returns 0 on success
 * or a negative SRT_E* code; srt_last_error() gives the text.  Per-ray failures never abort a batch:
 * they are reported through stopcond (same codes as raytracer.f95:324-353, plus SRT_STOP_NUMERIC for
 * the reference's process-killing `stop` on an SVD failure, blas.f95:208-211).
 * The library needs a gfx950 GPU; there is no CPU fallback.
 */
#ifndef SRT_H
#define SRT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SRT_VERSION 1
#define SRT_MAXSPEC 4
/* one trajectory row: t, pos[3], vprel[3], vgrel[3], n[3], B0[3], Ns[4]  (the per-step outputs of
 * raytracer_run; qs/ms/nus are per-model constants, see srt_model_species) */
#define SRT_ROW 20

enum {
  SRT_OK = 0,
  SRT_EINVAL = -1,  /* bad argument */
  SRT_EIO = -2,     /* file could not be read / written / parsed */
  SRT_EDEVICE = -3, /* HIP error or no gfx950 device */
  SRT_ENOMEM = -4
};

/* stop codes, raytracer.f95:324-353 and :749-754 */
enum {
  SRT_STOP_TMAX = 0,    /* t >= tmax (also: fixed-step ray left the resonance cone, :900-905) */
  SRT_STOP_MINALT = 1,  /* |pos| < minalt */
  SRT_STOP_KZERO = 2,   /* |k| == 0 */
  SRT_STOP_VGROUP = 3,  /* |vgrel| > 1.01 */
  SRT_STOP_DT = 5,      /* dt < 1e-14 */
  SRT_STOP_MAXSTEPS = 6,
  SRT_STOP_NUMERIC = 9  /* ours: non-finite state where the reference would `stop` the process */
};

/* integrator parameters = the scalar arguments of raytracer_run + the driver's outputper */
typedef struct srt_params {
  double dt0, dtmax, tmax, maxerr, minalt;
  double del;        /* FD step for dF/dx: 1e-4 for model 1, 1e-6 for models 3,4 (driver:251-252) */
  int32_t maxsteps;
  int32_t root;      /* 1 or 2 (2 = whistler) */
  int32_t fixedstep; /* 1 = RK4, 0 = adaptive RKF45 */
  int32_t outputper; /* keep rows 0, outputper, 2*outputper, ... (driver:1197) */
  int32_t first_attempt_policy; /* a ray's FIRST adaptive attempt, where raytracer.f95:778 reads an unset local (SURVEY A-1):
                                   0: NaN error term => accepted at dt0, dt not grown (what a flang build of the reference does,
                                      hence the goldens of this repository, api.make_params and bench.py);
                                   1: error from the k term alone (what the reference's own gfortran toolchain does; the
                                      `raytracer` executable's default, --first_attempt_policy).
                                   THE LIBRARY HAS NO DEFAULT: this member is read as given (a zeroed struct = 0); callers
                                   that replace a gfortran-built reference set 1.  INTEGRATION.md section 3. */
  int32_t refill_threshold;     /* free lanes per wave before new rays are claimed (0 = default) */
  int32_t ray_order;            /* order in which the launch set is WORKED ON (results and their order are unchanged):
                                   0 = as given; 1 = sorted on the device by the Morton code of the launch cell, so
                                   that the lanes of a wave start in neighbouring cells of the interp grid and share
                                   coefficient lines (SURVEY.md 8d allows this input permutation; other models ignore it);
                                   2 = as 1 with the rays that are likely to stop early (above 6 kHz, launched inwards)
                                   behind all others -- an experiment switch for the launch's tail, not a gain (HISTORY.md section 9) */
} srt_params;

typedef struct srt_model srt_model; /* opaque; owns device copies of all model data */

/* ---- library ---- */
/* Binds the CALLING THREAD to a HIP device (and makes it the default for threads that never call srt_init).  Models are
 * created on the calling thread's device and remember it: one host thread per GPU can drive its own model replica
 * (the CLI's --devices=0,1,..; the reference has no parallel mode, raytracer_driver.f95:1144-1232 is a serial loop). */
int srt_init(int device);
const char *srt_last_error(void);
int srt_device_info(char *name, int name_len, int *cu_count, int64_t *hbm_bytes);

/* ---- models (replace ngosetup / interpsetup / scatteredinterpsetup + the TRANSFER'd state blob) ---- */
/* modelnum=1: Ngo diffusive-equilibrium model; configfile is the legacy newray.in card file
 * (ngo_dens_model.f95:29-160).  yearday/msec = itime of the adapters (dipole tilt). */
int srt_model_create_ngo(const char *configfile, int yearday, int msec, srt_model **out);
/* modelnum=3: regular grid of ln(N_s), text file in the format of
 * gcpm_dens_model_buildgrid.f95:302-327 as read by interp_dens_model_adapter.f95:58-117 */
int srt_model_create_interp_file(const char *gridfile, int yearday, int msec, srt_model **out);
/* same, from host arrays: F[nz][ny][nx][nspec] (file order: species fastest, then x, y, z);
 * derivs = NULL (finite differences as libtricubic.f95:722-793) or 7 arrays of F's shape in the
 * order dfdx, dfdy, dfdz, d2fdxdy, d2fdxdz, d2fdydz, d3fdxdydz */
int srt_model_create_interp(int nspec, int nx, int ny, int nz, const double bounds[6],
                            const double *qs, const double *ms, const double *F,
                            const double *const *derivs, int yearday, int msec, srt_model **out);
/* modelnum=4: scattered ln(N_s) samples, text file of gcpm_dens_model_buildgrid_random.f95:210-225;
 * parameters are the --scattered_interp_* flags of raytracer_driver.f95:690-728.  order = degree of the fitted monomials: 0..3 (the
 * reference's tabular_monomials, lsinterp_mod.f95:70-99) on the kernels built for speed, 4 and 5 (its generate_monomials, :114-164:
 * 35 / 56 monomials) on one cooperative path that answers at ~1 % of order 2's rate; order >= 6 is refused with SRT_EINVAL */
int srt_model_create_scattered_file(const char *ptsfile, int yearday, int msec, double window_scale,
                                    int order, int exact, double local_window_scale, srt_model **out);
/* The same with the reference's kd-tree ROOT reproduced (opt-in, for diffing against a reference binary): the reference's
 * kdtree_nearest starts its search at the tree's root (kdtree_mod.f95:386-444), so for the ONE sample that is the root
 * "the nearest other sample" is the sample itself -- its stored spacing stays 0 and does not enter maxnearest
 * (scattered_interp_dens_model_adapter.f95:167-203).  Every lookup whose search window holds that sample sees
 * it.  Which sample it is comes from the reference build's RNG (randperm, :137-140), so the caller names it:
 * root_sample = its 0-based record number in the file (the reference prints nothing; a harness linked against its modules can
 * ask the tree -- oracle/ref_harness.f95 --mode=scatroot), -1 = none = srt_model_create_scattered_file, which stores the true
 * distance for every sample (the documented default). */
int srt_model_create_scattered_file_root(const char *ptsfile, int yearday, int msec, double window_scale,
                                         int order, int exact, double local_window_scale, int64_t root_sample,
                                         srt_model **out);
/* The step before the path: sample a model's funcPlasmaParams on a regular nx x ny x nz grid in
 * log space ON THE DEVICE -- gcpm_dens_model_buildgrid.f95:160-300 with any model handle in place of GCPM.
 * compder = 1 adds the seven explicit finite-difference blocks (d = 1e-3*|pos|, :197-296); compder = 0 leaves the
 * derivatives to the interp model's own finite differences, as the adapter does.
 *   srt_build_grid: host outputs F[nz][ny][nx][nspec] and derivs (7 blocks concatenated; may be NULL if !compder),
 *                   ready for srt_grid_file_write.
 *   srt_model_create_interp_from_model: the same grid becomes a modelnum=3 model without leaving the device. */
int srt_build_grid(srt_model *src, int compder, int nx, int ny, int nz, const double bounds[6], double *F,
                   double *derivs);
int srt_model_create_interp_from_model(srt_model *src, int compder, int nx, int ny, int nz,
                                       const double bounds[6], int yearday, int msec, srt_model **out);
/* The other builder of the step before the path: the random / adaptive sample set of
 * gcpm_dens_model_buildgrid_random.f95:228-407 + randomsampling_mod.f95:27-200 (recursivesampler), with any model
 * handle in place of GCPM, generated and refined ON THE DEVICE.  Stage order and per-stage rules are the reference's:
 * points of an existing file first (in_pts, kept and pushed to the output), n_initial_radial shell samples retried
 * until inside the box, n_initial_uniform samples, adaptive passes with tol = initial_tol, tol/2, ... until
 * adaptive_nmax adaptive samples exist, n_zero_altitude samples on the sphere r = R_E and n_iri_pad samples in the
 * shell R_E .. R_E+2000 km (both kept only when inside the box).  The recursion of one pass is evaluated level by
 * level (all half-boxes of one depth at once) from counter-based random numbers, so the set is a pure function of
 * `seed` (the reference seeds from the clock).  max_passes bounds the tolerance-halving loop, which the reference
 * leaves unbounded (0 = 64).  out = malloc'd [n_out][3+nspec] records "x y z lnN_1..lnN_nspec", the lines of the
 * model-4 file (srt_free); stage_counts[6] = samples from {input, radial, uniform, adaptive, zero altitude, iri}. */
typedef struct srt_sampler_params {
  double bounds[6];          /* minx maxx miny maxy minz maxz */
  int64_t n_zero_altitude, n_iri_pad, n_initial_radial, n_initial_uniform, adaptive_nmax;
  double initial_tol;
  int32_t max_recursion;
  int32_t numincrease;       /* the reference passes 5 (gcpm_dens_model_buildgrid_random.f95:338); 0 = 5 */
  int32_t max_passes;
  int32_t reserved;
  uint64_t seed;
} srt_sampler_params;
int srt_build_samples(srt_model *src, const srt_sampler_params *sp, int64_t n_in, const double *in_pts,
                      int64_t *n_out, double **out, int64_t stage_counts[6]);
/* Field options of the adapters (driver flags --use_igrf / --use_tsyganenko; interp_dens_model_adapter.f95:214-267
 * and its twins), evaluated on the device at every lookup, in the arithmetic types of the Fortran:
 *   use_igrf = 1: IGRF main field instead of the dipole (geopack2008.for IGRF_GSW_08 after RECALC_08 for the model's
 *     yearday / milliseconds_day, REAL);
 *   use_tsyganenko = 1: the T04_s (Tsyganenko & Sitnov 2005) external field added to it (TS05_aka_TS04.for, REAL*8
 *     inside, REAL interface), with PARMOD from srt_model_set_tsyganenko_params and geopack's dipole tilt for the date.
 * igrf_coeff_file = table of the published Gauss coefficients, needed by either option (NULL: $SRT_IGRF_COEFFS, else the
 * data/igrf_coeffs.txt shipped beside the library). */
int srt_model_set_field(srt_model *m, int use_igrf, int use_tsyganenko, const char *igrf_coeff_file);
/* parmod[10] = Pdyn (nPa), Dst (nT), ByIMF, BzIMF (nT), W1 .. W6: the driver's --tsyganenko_* flags */
int srt_model_set_tsyganenko_params(srt_model *m, const double parmod[10]);
void srt_model_destroy(srt_model *m);
int srt_model_kind(const srt_model *m);  /* 1, 3 or 4 */
int srt_model_nspec(const srt_model *m);
int srt_model_species(const srt_model *m, double qs[SRT_MAXSPEC], double ms[SRT_MAXSPEC]);
int64_t srt_model_device_bytes(const srt_model *m);

/* ---- layered entry points (each mirrors one reference procedure, batched over n items) ---- */
/* funcPlasmaParams: x[n][3] -> qs,Ns,ms,nus [n][4], B0[n][3] */
int srt_plasma_params(srt_model *m, int64_t n, const double *x, double *qs, double *Ns, double *ms,
                      double *nus, double *B0);
/* dispersion_relation + stix_parameters + solve_dispersion_relation at (x,k,w):
 * out[n][10] = F, S, D, P, R, L, Re k1, Im k1, Re k2, Im k2   (raytracer.f95:41-102, 408-502) */
int srt_dispersion(srt_model *m, int64_t n, const double *x, const double *k, const double *w,
                   double *out);
/* is_right_handed(n2, phi[deg, as the reference passes it], S, D, P) for in[n][5] -> out[n] (0/1)
 * (raytracer.f95:373-405; model-independent) */
int srt_is_right_handed(int64_t n, const double *in, int32_t *out);
/* dFdk(del=1e-8), dFdw(1e-8), dFdx(del), raytracer_evalrhs: out[n][14]  (raytracer.f95:118-314) */
int srt_gradients(srt_model *m, int64_t n, const double *x, const double *k, const double *w,
                  double del, double *out);
/* one rk4 and one rk45 step from args[n][7] with dt[n]: out[n][21] = rk4(7), out4(7), out5(7)
 * (raytracer.f95:504-596) */
int srt_rk_step(srt_model *m, int64_t n, const double *args, const double *dt, double del,
                double *out);

/* ---- the hot path: replaces the driver's whole ray loop ---- */
/* slots per ray = ceil(maxsteps/outputper) */
int32_t srt_rows_per_ray(const srt_params *p);
/* Inputs pos0/dir0 [nrays][3], w0[nrays].  Outputs (caller-allocated host arrays):
 *   rows     [nrays][slots][SRT_ROW]   kept rows (row index r*outputper is stored in slot r)
 *   nrows    [nrays]                   total rows T the ray produced (= accepted steps + 1)
 *   stopcond [nrays]
 * accepted_steps (optional) receives sum(nrows-1).
 * One call per model at a time: the device staging of this entry point is owned by the model handle and reused from call
 * to call (grow-only; srt_model_trim() gives it back).  A second host thread calling it on the SAME model waits for the
 * first (per-model lock); different models run concurrently. */
int srt_trace_batch(srt_model *m, const srt_params *p, int64_t nrays, const double *pos0,
                    const double *dir0, const double *w0, double *rows, int32_t *nrows,
                    int32_t *stopcond, int64_t *accepted_steps);
/* Frees the grow-only device scratch a model has accumulated (host-buffer staging of srt_trace_batch, per-launch sort and
 * staging buffers); the tables stay.  The next call allocates what it needs again. */
int srt_model_trim(srt_model *m);
/* Same with every buffer already resident in device memory (what bench.py times).
 * d_pos0/d_dir0 are SoA on the device: [3][nrays].  stream = hipStream_t (NULL = default).
 * d_counters: 4 x int64 scratch/outputs: [0] queue head (zeroed by the call), [1] accepted steps,
 * [2] attempts, [3] wave-attempts (lane occupancy = [2] / (64 * [3])).  Asynchronous: returns after enqueue. */
int srt_trace_batch_device(srt_model *m, const srt_params *p, int64_t nrays, const double *d_pos0,
                           const double *d_dir0, const double *d_w0, double *d_rows,
                           int32_t *d_nrows, int32_t *d_stopcond, int64_t *d_counters, void *stream);
/* Packed form of a row buffer for transport (multi-GPU gather; the driver's own loop writes exactly
 * these rows, raytracer_driver.f95:1197): a ray keeps rows 0, outputper, 2*outputper, .. < nrows, i.e.
 * kept = (nrows-1)/outputper + 1 of its `slots` slots.  d_offsets[nrays+1] receives the exclusive prefix sums of the
 * kept counts (d_offsets[nrays] = total), d_packed[total][SRT_ROW] the kept rows of ray 0, ray 1, ... back to back.
 * capacity_rows = rows d_packed can hold (nrays*slots always suffices; 0 = offsets only).  If the total exceeds the
 * capacity, the rays that do not fit are left out: compare d_offsets[nrays] with the capacity after synchronising.
 * All pointers are device memory of ONE device -- the call finds that device from the buffers themselves (not from the
 * calling thread's binding), works there and leaves the thread's device as it found it; buffers on different devices or
 * host pointers fail with SRT_EINVAL.  `stream` must belong to the same device.  Asynchronous on `stream`. */
int srt_pack_rows_device(int32_t slots, int32_t outputper, int64_t nrays, const double *d_rows,
                         const int32_t *d_nrows, int64_t *d_offsets, double *d_packed, int64_t capacity_rows,
                         void *stream);
/* duration in ms of the most recent trace kernel on this model, measured with HIP events on the
 * stream it ran on (synchronises that stream) */
int srt_last_kernel_ms(srt_model *m, float *ms);
/* the same for the launch `back` launches before the most recent one (0 .. 3): lets a caller that keeps several
 * launches in flight on different streams read their durations afterwards */
int srt_launch_ms(srt_model *m, int back, float *ms);

/* ---- the step after the path: hot-plasma damping along the kept rows ----
 * Replaces the MATLAB post-processor matlab/damping/: test_dampray.m:24-99 (per-row driver, running magnitude),
 * spatialdamping.m / temporaldamping.m, hot_dispersion_imag.m (adaptive Gauss-Kronrod quadrature, quadva.m, of
 * integrand.m over vperp), fG1.m, fG2.m, suprathermal.m, maxwellboltzmann.m.  It consumes exactly the columns a kept
 * row holds (n, vgrel, pos, B0, Ns) plus the ray's w and the model's qs, ms; collisions are ignored as the scripts do.
 * One hot species (electrons: qh = -1.60217646e-19 C, mh = 9.10938188e-31 kg, const.m). */
typedef struct srt_damping_params {
  int32_t dist;    /* 0: suprathermal.m (Bell 2002); 1: Ne_h * maxwellboltzmann(vperp,vpar,ME,kT) */
  int32_t mode;    /* 0: spatial rate ki along vg (test_dampray.m); 1: temporal rate gamma (temporaldamping.m) */
  int32_t nres;    /* number of resonances, 1..8; 0 = the script's m = [-1 0 1] */
  int32_t m[8];    /* resonance orders (0 = Landau, +-1 = cyclotron) */
  double Ne_h, kT; /* dist 1: hot density in m^-3, temperature in J */
  double tol;      /* relative tolerance of the quadrature; 0 = the scripts' 1e-3 */
} srt_damping_params;
/* rows/nrows/w0 as returned by srt_trace_batch with the same slots = srt_rows_per_ray(p) and outputper.
 * Outputs [nrays][slots]: rate (ki along vg in 1/m, or gamma in 1/s; 0 in slot 0), magnitude (1 in slot 0, then
 * magnitude(i-1)*exp(-|pos_i - pos_{i-1}| ki_i) or *exp(gamma_i (t_i - t_{i-1})); 0 beyond the ray), flag (0 ok,
 * 1 quadrature stopped before its error test was met, 2 integrand not finite (rate = NaN), 3 k = 0 (the scripts'
 * "not solving evanescent mode": magnitude is 0 from there on)).  magnitude and flag may be NULL. */
int srt_damping(const srt_damping_params *dp, int nspec, const double *qs, const double *ms, int32_t slots,
                int32_t outputper, int64_t nrays, const double *rows, const int32_t *nrows, const double *w0,
                double *rate, double *magnitude, int32_t *flag);
/* the same on buffers already resident in device memory (the row buffer srt_trace_batch_device filled);
 * d_magnitude may be NULL, d_flag may not.  Asynchronous on `stream`. */
int srt_damping_device(const srt_damping_params *dp, int nspec, const double *qs, const double *ms, int32_t slots,
                       int32_t outputper, int64_t nrays, const double *d_rows, const int32_t *d_nrows,
                       const double *d_w0, double *d_rate, double *d_magnitude, int32_t *d_flag, void *stream);

/* ---- file formats of the boundary ---- */
/* ray input file: 7 list-directed reals per line (raytracer_driver.f95:1146); returns count, fills
 * malloc'd arrays the caller frees with srt_free */
int64_t srt_read_rays_file(const char *path, double **pos0, double **dir0, double **w0);
/* .ray writer, record format of raytracer_driver.f95:1197-1217; raynum0 = number of first ray (1).
 * qs/ms: the model's species constants (srt_model_species); pure host code, needs no GPU. */
int srt_write_ray_file(const char *path, int append, int64_t raynum0, int64_t nrays,
                       const srt_params *p, int nspec, const double *qs, const double *ms,
                       const double *w0, const double *rows, const int32_t *nrows,
                       const int32_t *stopcond);
/* .ray reader: the inverse of srt_write_ray_file, for files written by this library or by the reference's driver (what
 * matlab/readrayoutput.m is to the MATLAB damping scripts): returns the number of rays (< 0: error) and malloc'd arrays
 * (srt_free): raynum / stopcond / kept (records of the ray) / w0 per ray, rows[nrecords][SRT_ROW] = the kept rows of all rays
 * back to back (the packed layout of srt_pack_rows_device), plus the record's constants nspec, qs, ms.  Host code, no GPU. */
int64_t srt_read_ray_file(const char *path, int32_t *nspec, double qs[4], double ms[4], int64_t *nrecords,
                          int64_t **raynum, int32_t **stopcond, int32_t **kept, double **w0, double **rows);
void srt_free(void *p);

/* model-3 grid files, the step before the path: the text format written by
 * gcpm_dens_model_buildgrid.f95:302-327 and read by interp_dens_model_adapter.f95:58-117, and a binary
 * side-format ("SRTGRID1": 144-byte header {magic, compder, nspec, nx, ny, nz, bounds[6], qs[4], ms[4]} + the
 * same value blocks as raw little-endian doubles) for grids whose text takes longer to parse than to trace.
 * srt_model_create_interp_file() accepts either (detected by the magic).  Pure host code, needs no GPU.
 *   dims = {compder, nspec, nx, ny, nz}; F[nz][ny][nx][nspec]; derivs = 7 blocks of F's shape, concatenated
 *   (dfdx dfdy dfdz d2fdxdy d2fdxdz d2fdydz d3fdxdydz), or NULL.  *F / *derivs are malloc'd: srt_free(). */
int srt_grid_file_read(const char *path, int32_t dims[5], double bounds[6], double qs[4], double ms[4],
                       double **F, double **derivs);
int srt_grid_file_write(const char *path, int binary, int nspec, int nx, int ny, int nz,
                        const double bounds[6], const double *qs, const double *ms, const double *F,
                        const double *derivs);
int srt_grid_file_convert(const char *in_text_or_binary, const char *out_binary);
int srt_grid_file_is_binary(const char *path);

/* model-4 sample files: the text format written by gcpm_dens_model_buildgrid_random.f95:196-225 (+ helper
 * module :37-43) and read by scattered_interp_dens_model_adapter.f95:85-133, and a binary side-format ("SRTPTS01":
 * 136-byte header {magic, nspec, 0, npts, bounds[6], qs[4], ms[4]} + npts records of 3 + nspec raw doubles).
 * srt_model_create_scattered_file() accepts either (detected by the magic).  records = [npts][3 + nspec]
 * "x y z lnN_1 .. lnN_nspec", e.g. the output of srt_build_samples.  Pure host code, needs no GPU. */
int srt_points_file_write(const char *path, int binary, int nspec, int64_t npts, const double bounds[6],
                          const double *qs, const double *ms, const double *records);
int srt_points_file_convert(const char *in_text, const char *out_binary);
int srt_points_file_is_binary(const char *path);

#ifdef __cplusplus
}
#endif
#endif
