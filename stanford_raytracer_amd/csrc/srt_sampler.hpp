// srt_sampler.hpp -- the random / adaptive sample-set builder on the device (second half).
//
// Reference: fortran/gcpm_dens_model_buildgrid_random.f95:228-407 (stages: radial, uniform, adaptive, zero altitude,
// ionosphere shell) and fortran/randomsampling_mod.f95:27-200 (recursivesampler), with any in-scope model in place
// of GCPM.  The reference refines depth-first, one kd-tree insertion at a time, from a clock-seeded random_number
// stream; none of that is reproducible, and none of it is parallel.  What IS defined is the decision rule of one
// half-box: count the samples strictly inside it; with <= 2 add `numincrease`; sum the per-species sample variances
// of ln N over the samples inside; refine (add `numincrease` more, recurse) iff sqrt(|vol^2 var / count|) > alpha.
// A half-box's decision depends only on the samples inside it, which only its ancestors add, so the recursion is
// evaluated LEVEL BY LEVEL: all calls of one depth at once, samples sorted by half-box (device radix sort), one wave
// per half-box for the statistics.  Random numbers are counter-based (keyed by stage / pass / half-box / draw), so
// the sample set is a pure function of the seed and identical to the depth-first order of the CPU oracle.
#pragma once
#include "srt_models.hpp"

namespace srt {

// ---- counter-based uniforms: splitmix64 finaliser chained over the key words --------------------------------
__host__ __device__ inline unsigned long long smp_mix(unsigned long long z) {
  z += 0x9e3779b97f4a7c15ULL;
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
  return z ^ (z >> 31);
}
__host__ __device__ inline double smp_uniform(unsigned long long seed, unsigned long long stream, unsigned long long a,
                                              unsigned long long b, unsigned long long c) {
  unsigned long long h = smp_mix(seed + stream);
  h = smp_mix(h ^ a);
  h = smp_mix(h ^ b);
  h = smp_mix(h ^ c);
  return (double)(h >> 11) * 0x1.0p-53; // [0,1)
}
// util.f95:26-49 normal(): polar method, retried until 0 < r <= 1.  Draws c0+2t, c0+2t+1 for retry t.
__host__ __device__ inline double smp_normal(unsigned long long seed, unsigned long long stream, unsigned long long a,
                                             unsigned long long b, unsigned long long c0) {
  for (unsigned t = 0;; ++t) {
    const double u = 2.0 * smp_uniform(seed, stream, a, b, c0 + 2 * t) - 1.0;
    const double v = 2.0 * smp_uniform(seed, stream, a, b, c0 + 2 * t + 1) - 1.0;
    const double r = u * u + v * v;
    if (r <= 0.0 || r > 1.0) continue;
    return u * sqrt(-2.0 * log(r) / r);
  }
}

enum { SMP_RADIAL = 1, SMP_UNIFORM = 2, SMP_ADAPT = 3, SMP_ZEROALT = 4, SMP_IRI = 5 };
constexpr int SMP_REC = 8;          // x y z lnN1..4 pad: 64 B, the record of the scattered model's table
constexpr int SMP_MAXTRY = 100000;  // radial stage: attempts per sample before the request is refused

struct SmpBox {
  double lo[3], hi[3];
  unsigned long long id; // heap number of the call: root 1, half = 2*call + side, child call = its half
};

__device__ inline bool smp_inside(const double p[3], const double lo[3], const double hi[3]) {
  return p[0] > lo[0] && p[0] < hi[0] && p[1] > lo[1] && p[1] < hi[1] && p[2] > lo[2] && p[2] < hi[2];
}

// One shell sample: direction = three normals normalised, radius uniform in [rmin,rmax]
// (gcpm_dens_model_buildgrid_random.f95:247-257, :357-367, :385-395).  try_ = attempt number.
__device__ inline void smp_shell_point(unsigned long long seed, unsigned long long stream, unsigned long long i,
                                       unsigned long long try_, double rmin, double rmax, double p[3]) {
#pragma clang fp contract(off)
  double xd = smp_normal(seed, stream, i, try_, 0);
  double yd = smp_normal(seed, stream, i, try_, 1ULL << 20);
  double zd = smp_normal(seed, stream, i, try_, 2ULL << 20);
  const double nrm = sqrt(xd * xd + yd * yd + zd * zd);
  xd = xd / nrm;
  yd = yd / nrm;
  zd = zd / nrm;
  double r = smp_uniform(seed, stream, i, try_, 3ULL << 20);
  r = rmin + (rmax - rmin) * r;
  p[0] = r * xd;
  p[1] = r * yd;
  p[2] = r * zd;
}

// stage positions -> rec[i][0..2], valid[i].  Invalid records get the box centre (evaluated, then dropped).
__global__ void smp_stage_kernel(int stage, long long n, unsigned long long seed, SmpBox box, double rmin, double rmax,
                                 double *rec, int *valid) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double p[3];
  int ok = 1;
  if (stage == SMP_UNIFORM) { // :281-290
#pragma clang fp contract(off)
    for (int c = 0; c < 3; ++c) {
      const double u = smp_uniform(seed, SMP_UNIFORM, (unsigned long long)i, 0, (unsigned long long)c);
      p[c] = box.lo[c] + u * (box.hi[c] - box.lo[c]);
    }
  } else if (stage == SMP_RADIAL) { // :246-268: retried until the point falls strictly inside the box
    ok = 0;
    for (int t = 0; t < SMP_MAXTRY && !ok; ++t) {
      smp_shell_point(seed, SMP_RADIAL, (unsigned long long)i, (unsigned long long)t, rmin, rmax, p);
      ok = smp_inside(p, box.lo, box.hi);
    }
  } else { // zero altitude / ionosphere shell: one draw, kept only when inside (:356-376, :384-404)
    smp_shell_point(seed, (unsigned long long)stage, (unsigned long long)i, 0, rmin, rmax, p);
    ok = smp_inside(p, box.lo, box.hi);
  }
  if (!ok)
    for (int c = 0; c < 3; ++c) p[c] = 0.5 * (box.lo[c] + box.hi[c]);
  for (int c = 0; c < 3; ++c) rec[i * SMP_REC + c] = p[c];
  valid[i] = ok;
}

// f(x) of the helper module (gcpm_dens_model_buildgrid_random_helpermod.f95:28-46): Ns = log(Ns)
template <class M, bool USE_LDS>
__global__ __launch_bounds__(64) void smp_eval_kernel(const M *__restrict__ mp, long long n, double *rec) {
  const M &m = *mp;
  __shared__ __attribute__((aligned(16))) double tile[USE_LDS ? TILE_DOUBLES : 2];
  const long long id = (long long)blockIdx.x * WAVE + threadIdx.x;
  const bool live = id < n;
  const long long i = live ? id : n - 1; // every lane takes part in the (cooperative) lookups
  double p[1][3] = {{rec[i * SMP_REC + 0], rec[i * SMP_REC + 1], rec[i * SMP_REC + 2]}};
  double Ns[1][4];
  m.template density<1>(p, Ns, tile);
  if (live) {
#pragma unroll
    for (int s = 0; s < 4; ++s) rec[i * SMP_REC + 3 + s] = log(Ns[0][s]);
    rec[i * SMP_REC + 7] = 0.0;
  }
}

// bounds of half `side` of a call at depth d (randomsampling_mod.f95:79-86 lower, :138-145 upper) and the rectangle
// kdtree_search_rect is asked for (center, lower, upper; strict inequalities, kdtree_mod.f95:278-279)
__host__ __device__ inline void smp_half(const SmpBox &b, int dim, int side, double lo[3], double hi[3], double rlo[3],
                                         double rhi[3]) {
#pragma clang fp contract(off)
  for (int c = 0; c < 3; ++c) {
    lo[c] = b.lo[c];
    hi[c] = b.hi[c];
  }
  const double mid = b.lo[dim] + 0.5 * (b.hi[dim] - b.lo[dim]);
  if (side == 0) hi[dim] = mid;
  else lo[dim] = mid;
  for (int c = 0; c < 3; ++c) {
    const double center = lo[c] + 0.5 * (hi[c] - lo[c]);
    const double lower = center - lo[c], upper = hi[c] - center;
    rlo[c] = center - lower;
    rhi[c] = center + upper;
  }
}

// per pooled sample: which half of its call it lies in -> sort key (`none` = number of halves = not in an active half)
__global__ void smp_keys_kernel(long long npool, const double *rec, const int *slot, const SmpBox *calls, int dim,
                                unsigned none, unsigned *keys, int *idx) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npool) return;
  unsigned key = none;
  const int c = slot[i];
  if (c >= 0) {
    const double p[3] = {rec[i * SMP_REC], rec[i * SMP_REC + 1], rec[i * SMP_REC + 2]};
    for (int side = 0; side < 2; ++side) {
      double lo[3], hi[3], rlo[3], rhi[3];
      smp_half(calls[c], dim, side, lo, hi, rlo, rhi);
      if (smp_inside(p, rlo, rhi)) key = 2u * (unsigned)c + (unsigned)side;
    }
  }
  keys[i] = key;
  idx[i] = (int)i;
}

// per half: its box and 2*ninc candidate positions (the first ninc are used when the half holds <= 2 samples, the
// second ninc when it is refined; randomsampling_mod.f95:98-103, :126-131: rt = randnum*(max-min)+min)
__global__ void smp_cand_kernel(int nhalf, const SmpBox *calls, int dim, unsigned long long seed, unsigned long long pass,
                                int ninc, SmpBox *halves, double *cand) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nhalf * 2 * ninc) return;
  const int h = t / (2 * ninc), j = t % (2 * ninc);
  double lo[3], hi[3], rlo[3], rhi[3];
  smp_half(calls[h >> 1], dim, h & 1, lo, hi, rlo, rhi);
  const unsigned long long hid = 2ULL * calls[h >> 1].id + (unsigned long long)(h & 1);
  if (j == 0) {
    SmpBox hb;
    for (int c = 0; c < 3; ++c) {
      hb.lo[c] = lo[c];
      hb.hi[c] = hi[c];
    }
    hb.id = hid;
    halves[h] = hb;
  }
  {
#pragma clang fp contract(off)
    for (int c = 0; c < 3; ++c) {
      const double u = smp_uniform(seed, SMP_ADAPT, pass, hid, (unsigned long long)(3 * j + c));
      cand[(size_t)t * SMP_REC + c] = u * (hi[c] - lo[c]) + lo[c];
    }
  }
}

__device__ inline double smp_wave_sum(double v) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// one wave per half: statistics and the two decisions (randomsampling_mod.f95:94-134 / :153-193)
__global__ __launch_bounds__(64) void smp_stats_kernel(int nhalf, long long npool, const unsigned *skeys, const int *sidx,
                                                       const double *rec, const double *cand, const SmpBox *halves,
                                                       int dim, int nspec, int ninc, double alpha, int *nadd, int *refine,
                                                       int *addA) {
  const int h = blockIdx.x;
  if (h >= nhalf) return;
  const int lane = threadIdx.x;
  // segment of this half in the sorted key array
  long long lo = 0, hi = npool;
  while (lo < hi) {
    const long long mid = (lo + hi) >> 1;
    if (skeys[mid] < (unsigned)h) lo = mid + 1;
    else hi = mid;
  }
  const long long beg = lo;
  hi = npool;
  while (lo < hi) {
    const long long mid = (lo + hi) >> 1;
    if (skeys[mid] <= (unsigned)h) lo = mid + 1;
    else hi = mid;
  }
  const long long end = lo;
  // Samples whose ln N is not finite (a model with zero density inside the Earth gives log(0)) are left out of the
  // statistics: one of them makes every enclosing box's variance NaN and `NaN > alpha` is false, which silently ends
  // all refinement from the root down (deliberate divergence; they stay in the output for the caller).
  auto finite4 = [&](const double *r) {
    bool ok = true;
    for (int s = 0; s < nspec; ++s) ok = ok && isfinite(r[s]);
    return ok;
  };
  long long mycnt = 0;
  for (long long q = beg + lane; q < end; q += WAVE) mycnt += finite4(rec + (size_t)sidx[q] * SMP_REC + 3) ? 1 : 0;
  long long cnt = (long long)smp_wave_sum((double)mycnt);
  const bool a = cnt <= 2; // "If we have too few points, then add some"
  // The candidates lie inside the half by construction, but kdtree_search_rect's strict test is applied to them
  // like to any other sample (a draw of exactly 0 would sit on the face).
  const SmpBox hb = halves[h];
  double rlo[3], rhi[3];
  {
#pragma clang fp contract(off)
    for (int c = 0; c < 3; ++c) {
      const double center = hb.lo[c] + 0.5 * (hb.hi[c] - hb.lo[c]);
      const double lower = center - hb.lo[c], upper = hb.hi[c] - center;
      rlo[c] = center - lower;
      rhi[c] = center + upper;
    }
  }
  const double *cd = cand + (size_t)h * 2 * ninc * SMP_REC;
  bool cand_in = false;
  if (a && lane < ninc) {
    const double p[3] = {cd[lane * SMP_REC], cd[lane * SMP_REC + 1], cd[lane * SMP_REC + 2]};
    cand_in = smp_inside(p, rlo, rhi) && finite4(cd + lane * SMP_REC + 3);
  }
  const int ncand = __popcll(__ballot(cand_in));
  cnt += ncand;
  // mean, then the sum of squared deviations (two passes, as the reference)
  double sum[4] = {0, 0, 0, 0};
  for (long long q = beg + lane; q < end; q += WAVE) {
    const double *r = rec + (size_t)sidx[q] * SMP_REC + 3;
    if (!finite4(r)) continue;
    for (int s = 0; s < nspec; ++s) sum[s] += r[s];
  }
  if (cand_in)
    for (int s = 0; s < nspec; ++s) sum[s] += cd[lane * SMP_REC + 3 + s];
  double mean[4];
  for (int s = 0; s < nspec; ++s) mean[s] = smp_wave_sum(sum[s]) / (double)cnt;
  double sq[4] = {0, 0, 0, 0};
  for (long long q = beg + lane; q < end; q += WAVE) {
    const double *r = rec + (size_t)sidx[q] * SMP_REC + 3;
    if (!finite4(r)) continue;
    for (int s = 0; s < nspec; ++s) sq[s] += (r[s] - mean[s]) * (r[s] - mean[s]);
  }
  if (cand_in)
    for (int s = 0; s < nspec; ++s) {
      const double v = cd[lane * SMP_REC + 3 + s];
      sq[s] += (v - mean[s]) * (v - mean[s]);
    }
  double var = 0.0;
  for (int s = 0; s < nspec; ++s) var = var + 1.0 / (double)(cnt - 1) * smp_wave_sum(sq[s]);
  const double vol = ((hb.hi[0] - hb.lo[0]) / R_E) * ((hb.hi[1] - hb.lo[1]) / R_E) * ((hb.hi[2] - hb.lo[2]) / R_E);
  const double var1 = vol * vol * var / (double)cnt;
  const bool ref = sqrt(fabs(var1)) > alpha; // NaN (a -inf sample, a single sample) => no refinement
  if (lane == 0) {
    addA[h] = a ? 1 : 0;
    refine[h] = ref ? 1 : 0;
    nadd[h] = (a ? ninc : 0) + (ref ? ninc : 0);
  }
  (void)dim;
}

// per half: committed candidates go to the pool (first batch, then second), child calls are created
__global__ void smp_commit_kernel(int nhalf, int ninc, const int *addA, const int *refine, const int *addoff,
                                  const int *childoff, int make_children, const double *cand, const SmpBox *halves,
                                  long long pool_base, double *rec, int *slot, SmpBox *children) {
  const int h = blockIdx.x * blockDim.x + threadIdx.x;
  if (h >= nhalf) return;
  const int child = (refine[h] && make_children) ? childoff[h] : -1;
  long long dst = pool_base + addoff[h];
  for (int batch = 0; batch < 2; ++batch) {
    if (!(batch == 0 ? addA[h] : refine[h])) continue;
    for (int j = 0; j < ninc; ++j, ++dst) {
      const double *src = cand + ((size_t)h * 2 * ninc + batch * ninc + j) * SMP_REC;
      for (int c = 0; c < SMP_REC; ++c) rec[dst * SMP_REC + c] = src[c];
      slot[dst] = child;
    }
  }
  if (child >= 0) children[child] = halves[h];
}

// per pooled sample: follow its half into the child call, or retire
__global__ void smp_reslot_kernel(long long npool, const unsigned *keys, unsigned none, const int *refine,
                                  const int *childoff, int make_children, int *slot) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npool) return;
  const unsigned k = keys[i];
  slot[i] = (k < none && refine[k] && make_children) ? childoff[k] : -1;
}

__global__ void smp_fill_int(long long n, int *a, int v) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) a[i] = v;
}

} // namespace srt
