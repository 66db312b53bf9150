/* srt_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C, scalar, fp64, no FMA contraction) of the reference's hot path:
 *   fortran/raytracer.f95 (integrator + dispersion physics), bmodel_dipole.f95, util.f95,
 *   xform_double/ (SM<->GSM chain), ngo_dens_model(+_adapter).f95, interp_dens_model_adapter.f95,
 *   tricubic-for/libtricubic.f95, scattered_interp_dens_model_adapter.f95 + kdtree_mod + lsinterp_mod.
 * Every function cites the reference file:line it follows.  It exists so that tests/, smoke() and
 * bench.py's cpu_baseline leg have something to check the HIP path against on the GPU box (where
 * /root/reference does not exist).  The product (stanford_raytracer_amd/) never links, imports or
 * calls it.
 *
 * Parity status: PINNED -- validated against the reference itself (oracle/_ref/ref_harness, built
 * from /root/reference by oracle/build_ref.py) and against the committed golden vectors generated
 * from it (tests/golden/, tests/golden/make_golden.py).  The reference ships no tests/fixtures of
 * its own for this path.
 */
#ifndef SRT_ORACLE_H
#define SRT_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

#define SO_MAXSPEC 4
#define SO_ROW 20 /* t, pos[3], vprel[3], vgrel[3], n[3], B0[3], Ns[4] */

typedef struct so_model so_model;

typedef struct {
  double dt0, dtmax, tmax, maxerr, minalt, del;
  int maxsteps, root, fixedstep;
  int first_attempt_policy; /* 0 = NaN error term => accept, no growth (flang, Appendix A-1);
                               1 = error from the k term alone (gfortran<=8 MAX semantics) */
} so_params;

/* model construction; return NULL on error */
so_model *so_model_create_ngo(const char *configfile, int yearday, int msec);
so_model *so_model_create_interp_file(const char *gridfile, int yearday, int msec);
/* F in the file's order: species fastest, then x, y, z (interp_dens_model_adapter.f95:100-106) */
so_model *so_model_create_interp(int nspec, int nx, int ny, int nz, const double bounds[6],
                                 const double *qs, const double *ms, const double *F, int yearday,
                                 int msec);
so_model *so_model_create_scattered_file(const char *ptsfile, int yearday, int msec,
                                         double window_scale, int order, int exact,
                                         double local_window_scale, unsigned perm_seed);
/* use_igrf = 1 (interp_dens_model_adapter.f95:236-241): IGRF via geopack's RECALC_08 / IGRF_GSW_08 restated in
 * srt_oracle_igrf.c; coeff_file = the Gauss-coefficient table (stanford_raytracer_amd/data/igrf_coeffs.txt). */
int so_model_set_igrf(so_model *m, int yearday, int msec, const char *coeff_file);
void so_model_destroy(so_model *m);
int so_model_nspec(const so_model *m);
int so_model_kind(const so_model *m); /* 1 ngo, 3 interp, 4 scattered */

/* L1/L0: funcPlasmaParams */
void so_plasma_params(so_model *m, const double x[3], double qs[4], double Ns[4], double ms[4],
                      double nus[4], double B0[3]);
/* L3 */
double so_dispersion_relation(const double n[3], double w, int nspec, const double *qs,
                              const double *Ns, const double *ms, const double B0[3]);
void so_stix_parameters(double w, int nspec, const double *qs, const double *Ns, const double *ms,
                        double B0mag, double *S, double *D, double *P, double *R, double *L);
int so_is_right_handed(double n2, double phi, double S, double D, double P);
void so_solve_dispersion_relation(so_model *m, const double k[3], double w, const double x[3],
                                  double k1[2], double k2[2]);
void so_dfdk(so_model *m, const double k[3], double w, const double x[3], double del, double out[3]);
double so_dfdw(so_model *m, const double k[3], double w, const double x[3], double del);
void so_dfdx(so_model *m, const double k[3], double w, const double x[3], double del, double out[3]);
void so_evalrhs(so_model *m, const double args[7], double del, double rhs[7]);
/* L4 */
void so_rk4(so_model *m, const double x[7], double del, double dt, double out[7]);
void so_rk45(so_model *m, const double x[7], double del, double dt, double out4[7], double out5[7]);
/* raytracer_run for one ray.  rows: capacity*SO_ROW doubles; returns number of rows written
 * (<= capacity; rows beyond capacity are counted in *nrows_total but not stored). */
int so_raytracer_run(so_model *m, const so_params *p, const double pos0[3], const double dir0_in[3],
                     double w0, double *rows, int capacity, int *nrows_total, int *stopcond);
/* batch over rays, nthreads >= 1 (pthreads; each ray independent). rows: nrays*capacity*SO_ROW */
long so_trace_batch(so_model *m, const so_params *p, long nrays, const double *pos0,
                    const double *dir0, const double *w0, double *rows, int capacity,
                    int *nrows, int *stopcond, int nthreads);

/* helpers exposed for tests */
int so_scattered_set_spacing(so_model *m, const double p[3], double value); /* srt_oracle_scattered.c */
double so_scattered_radius(const so_model *m);                              /* maxnearest * window_scale */
void so_dipole_tilt(int yearday, int msec, double *mu);
void so_bfield(so_model *m, const double x[3], double B0[3]);
double so_speed_of_light(void);

#ifdef __cplusplus
}
#endif
#endif
