/* srt_oracle_internal.h -- TEST INFRASTRUCTURE ONLY: shared structs of the CPU oracle. */
#ifndef SRT_ORACLE_INTERNAL_H
#define SRT_ORACLE_INTERNAL_H
#include <stddef.h>

/* module state of ngo_dens_model.f95:8-24 that `dens` reads (1-based like the Fortran) */
typedef struct {
  double pi, r0;
  int num, kducts, kinit;
  double therm, rbase, ane0, alpha0[5], rzero, scbot;
  double lk, expk, ddk, rconsn, scr;
  double l0[10], def[10], dd[10], rducln[10], rducun[10], rducls[10], rducus[10], sidedu[10];
  double hl2n[10], hl2s[10], hu2n[10], hu2s[10];
  double latitu;
  double z[3];   /* z(1), z(2) */
  double ani[5]; /* ani(1:4) */
} so_ngo;

/* interpStateData (interp_dens_model_adapter.f95:16-38); arrays in file order: species fastest, x, y, z */
typedef struct {
  int nspec, nx, ny, nz, have_derivs;
  double minx, maxx, miny, maxy, minz, maxz, delx, dely, delz;
  double qs[4], ms[4];
  double *x, *y, *z;
  double *arr[8]; /* F, dfdx, dfdy, dfdz, d2fdxdy, d2fdxdz, d2fdydz, d3fdxdydz */
} so_grid;

#ifndef SO_MAXSPEC
#define SO_MAXSPEC 4
#endif
typedef struct {
  double p[3];
  double val[SO_MAXSPEC + 1]; /* ln N_s ..., then distance to the nearest other sample */
  int dim, left, right;
} kdnode;

struct so_scattered {
  kdnode *nodes;
  int n, root, nspec;
  double qs[4], ms[4];
  double window_scale, local_window_scale, maxnearest;
  int order, exact;
  /* scratch for searches */
  int *found, cap;
};


struct so_model {
  int kind, nspec;
  double mu; /* dipole tilt (T4.f95) */
  int use_igrf; /* srt_oracle_igrf.c */
  float igrf_G[105], igrf_H[105], igrf_REC[105], igrf_A[9];
  so_ngo ngo;
  so_grid grid;
  struct so_scattered *sc;
};

void so_igrf_gsw(const float *G, const float *H, const float *REC, const float A[9], float xgsw, float ygsw, float zgsw, float *hx,
                 float *hy, float *hz);
void so_cartesian_to_spherical(const double x[3], double p[3]);
void so_scattered_params(struct so_model *m, const double x[3], double qs[4], double Ns[4],
                         double ms[4], double nus[4]);
void so_scattered_free(struct so_model *m);
#endif
