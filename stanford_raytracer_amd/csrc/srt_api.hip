// srt_api.hip -- C ABI (include/srt.h) of the MI355X-native many-ray Haselgrove integrator.
// Host side: model construction (device-resident tables), kernel launches, HIP-event timing.
// There is no CPU fallback: every entry point fails with SRT_EDEVICE when no GPU is usable.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/srt.h"
#include "srt_host.hpp"
#include "srt_kernels.hpp"
#include "srt_scattered.hpp"
#include "srt_sampler.hpp"
#include "srt_damping.hpp"
#include <hipcub/hipcub.hpp>
#include "tricubic_matrix.h"

using namespace srt;

// ------------------------------------------------------------------------------------------ errors
static thread_local std::string g_err;
int srt_set_error(int code, const char *fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}
extern "C" const char *srt_last_error(void) { return g_err.c_str(); }

#define HIP_OK(expr)                                                                          \
  do {                                                                                        \
    hipError_t e_ = (expr);                                                                   \
    if (e_ != hipSuccess)                                                                     \
      return srt_set_error(e_ == hipErrorOutOfMemory ? SRT_ENOMEM : SRT_EDEVICE, "%s: %s", #expr, \
                           hipGetErrorString(e_));                                            \
  } while (0)

// Device selection.  srt_init(device) binds the CALLING THREAD to a device (and makes it the process default for
// threads that never called srt_init): one host thread per GPU may drive its own models concurrently (the CLI's
// --devices=0,1,..).  A model remembers the device it was created on, and every entry point that takes a model
// switches to that device first, whichever thread calls it.
#include <atomic>
#include <mutex>
static std::atomic<int> g_default_device{-1};
static thread_local int t_device = -1;
extern "C" int srt_init(int device) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
    return srt_set_error(SRT_EDEVICE, "no HIP device available (this library has no CPU path)");
  if (device < 0 || device >= n) return srt_set_error(SRT_EINVAL, "device %d out of range (%d devices)", device, n);
  HIP_OK(hipSetDevice(device));
  t_device = device;
  g_default_device.store(device);
  return SRT_OK;
}
// the device the calling thread is bound to right now (srt_init's, or a DeviceScope's)
static int current_device() {
  int d = -1;
  if (hipGetDevice(&d) != hipSuccess) {
    (void)hipGetLastError();
    d = t_device >= 0 ? t_device : 0;
  }
  return d;
}
// An entry point that works on a model (or on buffers that live on some device) binds the calling thread to THAT device
// for its own duration only: the thread's srt_init() binding (t_device) and whatever device the caller's own HIP / torch
// code had selected are back in place when the call returns.
struct DeviceScope {
  int prev = -1;
  int enter(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) {
      prev = -1;
      (void)hipGetLastError();
    }
    if (hipSetDevice(dev) != hipSuccess) return srt_set_error(SRT_EDEVICE, "hipSetDevice(%d) failed", dev);
    return SRT_OK;
  }
  // entry points without a model: the thread's srt_init() device (the process default for threads that never called it;
  // device 0 when nobody did), for the duration of the call only
  int enter_default() {
    int dev = t_device >= 0 ? t_device : g_default_device.load();
    if (dev < 0) {
      int rc = srt_init(0);
      if (rc) return rc;
      dev = 0;
    }
    return enter(dev);
  }
  ~DeviceScope() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
};
// the one device that owns every (non-null) buffer of a device-buffer entry point; -1 + error if one is not device memory or
// they live on different devices
static int device_of(const char *who, const void *const *ptrs, int n, int *dev_out) {
  int dev = -1;
  for (int i = 0; i < n; ++i) {
    const void *q = ptrs[i];
    if (!q) continue;
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, q) != hipSuccess || at.type != hipMemoryTypeDevice) {
      (void)hipGetLastError();
      return srt_set_error(SRT_EINVAL, "%s: %p is not device memory", who, q);
    }
    if (dev < 0) dev = at.device;
    else if (at.device != dev) return srt_set_error(SRT_EINVAL, "%s: buffers live on devices %d and %d", who, dev, at.device);
  }
  *dev_out = dev;
  return SRT_OK;
}
extern "C" int srt_device_info(char *name, int name_len, int *cu_count, int64_t *hbm_bytes) {
  DeviceScope srt_iscope_;
  int rc = srt_iscope_.enter_default();
  if (rc) return rc;
  hipDeviceProp_t p;
  HIP_OK(hipGetDeviceProperties(&p, current_device()));
  if (name && name_len > 0) snprintf(name, name_len, "%s (%s)", p.name, p.gcnArchName);
  if (cu_count) *cu_count = p.multiProcessorCount;
  if (hbm_bytes) *hbm_bytes = (int64_t)p.totalGlobalMem;
  return SRT_OK;
}

// ------------------------------------------------------------------------------------------ model
struct srt_model {
  int kind = 0, nspec = 0;
  int device = current_device(); // the HIP device the tables live on (the creating thread's)
  Common cm{};
  NgoModel ngo{};
  InterpModel interp{};
  ScatteredModel scat{};
  double *d_pts = nullptr;
  double *d_xyz = nullptr; // scattered model: the sample positions once more, SoA [3][npts] (the candidate scans read only these)
  int *d_cells = nullptr;
  double *d_coef = nullptr;
  void *d_model = nullptr;   // device copy of ngo / interp (kernels read it through scalar loads)
  Common *d_common = nullptr;
  int64_t device_bytes = 0;
  int cu_count = 256;
  // Per-launch scratch in NSLOT slots, so that launches on different streams may overlap.  A launch takes a slot whose previous
  // launch has FINISHED (its `done` event has fired) before it opens a new one: a caller that launches from one stream, one
  // launch after the other, lives in a single slot (scattered model: 9.7 GB of candidate blocks + staging per slot at grid 2048),
  // and only launches that really overlap occupy more.  When every slot is busy the next one in turn is waited for.
  struct LaunchSlot {
    hipEvent_t done = nullptr; // recorded behind the launch that used this slot's scratch last
    bool used = false;
    // ray_order option (grow-only): keys in/out, ids in/out, radix-sort workspace
    unsigned *d_keys[2] = {nullptr, nullptr};
    int *d_ids[2] = {nullptr, nullptr};
    void *d_sorttmp = nullptr;
    size_t sort_cap = 0, sorttmp_bytes = 0;
    // scattered model (grow-only): staging records of coop_stencil, REC_CAP * REC doubles per one-wave block
    double *d_stage = nullptr;
    long long stage_blocks = 0, stage_failed = 0; // (failed: the grid size whose allocation was refused -- not retried per launch)
    // scattered model (grow-only): the lanes' candidate blocks, BLOCK_DOUBLES per one-wave block
    double *d_blocks = nullptr;
    long long cand_blocks = 0, cand_failed = 0;
  };
  static constexpr int NSLOT = 4;
  LaunchSlot slot[NSLOT];
  int rr_slot = 0;
  // timing history of the last NHIST launches (srt_last_kernel_ms, srt_launch_ms), its own ring: slots are reused out of turn
  struct LaunchTimes {
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool used = false;
  };
  static constexpr int NHIST = 4;
  LaunchTimes hist[NHIST];
  int next_hist = 0, last_hist = -1;
  bool warned_scratch = false;
  // device staging of the host-buffer entry point srt_trace_batch (grow-only, freed with the model): the CLI calls it
  // once per chunk of the ray file, and a fresh hipMalloc / hipFree of gigabytes per call costs as much as a small launch
  struct HostIO {
    void *p[9] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    size_t cap[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  } io;
  std::mutex io_lock; // one srt_trace_batch (host-buffer call) per model at a time: the staging above is shared
};
static int io_reserve(srt_model *m, int k, size_t bytes, void **out) {
  if (bytes > m->io.cap[k]) {
    if (m->io.p[k]) (void)hipFree(m->io.p[k]);
    m->io.p[k] = nullptr;
    m->io.cap[k] = 0;
    if (hipMalloc(&m->io.p[k], bytes) != hipSuccess) {
      (void)hipGetLastError();
      return srt_set_error(SRT_ENOMEM, "hipMalloc of %zu bytes failed", bytes);
    }
    m->io.cap[k] = bytes;
  }
  *out = m->io.p[k];
  return SRT_OK;
}

// switch the calling thread to the model's device until the enclosing entry point returns
#define ensure_model(m) (srt_dscope_.enter((m)->device))
#define SRT_MODEL_SCOPE DeviceScope srt_dscope_

static void fill_common(Common &cm, int nspec, const double *qs, const double *ms, int yearday, int msec) {
  memset(&cm, 0, sizeof cm);
  cm.sp.nspec = nspec;
  double maxq = 0.0, minm = 0.0;
  for (int s = 0; s < nspec; ++s) {
    cm.sp.q[s] = qs[s];
    cm.sp.m[s] = ms[s];
    cm.sp.c[s] = qs[s] * qs[s] / ms[s] / EPS0;
    cm.sp.g[s] = qs[s] / ms[s];
    if (fabs(qs[s]) > maxq) maxq = fabs(qs[s]);
    if (s == 0 || ms[s] < minm) minm = ms[s];
  }
  cm.sp.maxq2 = maxq * maxq;
  cm.sp.minm_eps0 = minm * EPS0;
  const double MU0 = PI * 4e-7; // constants.f95:6
  cm.C = sqrt(1.0 / EPS0 / MU0); // constants.f95:7
  cm.fld.yearday = yearday;
  cm.fld.msec = msec;
  double mu = srt_host::dipole_tilt(yearday, msec);
  cm.fld.cm = cos(mu);
  cm.fld.sm = sin(mu);
  cm.fld.bo_re3 = (.312 / 10000.0) * R_E * R_E * R_E;
}

static int model_finish(srt_model *m) {
  HIP_OK(hipMalloc(&m->d_common, sizeof(Common)));
  HIP_OK(hipMemcpy(m->d_common, &m->cm, sizeof(Common), hipMemcpyHostToDevice));
  if (m->kind == 1) {
    HIP_OK(hipMalloc(&m->d_model, sizeof(NgoModel)));
    HIP_OK(hipMemcpy(m->d_model, &m->ngo, sizeof(NgoModel), hipMemcpyHostToDevice));
  } else if (m->kind == 3) {
    HIP_OK(hipMalloc(&m->d_model, sizeof(InterpModel)));
    HIP_OK(hipMemcpy(m->d_model, &m->interp, sizeof(InterpModel), hipMemcpyHostToDevice));
  } else if (m->kind == 4) {
    HIP_OK(hipMalloc(&m->d_model, sizeof(ScatteredModel)));
    HIP_OK(hipMemcpy(m->d_model, &m->scat, sizeof(ScatteredModel), hipMemcpyHostToDevice));
  }
  hipDeviceProp_t p;
  m->device = current_device();
  HIP_OK(hipGetDeviceProperties(&p, m->device));
  m->cu_count = p.multiProcessorCount;
  for (auto &sl : m->slot) HIP_OK(hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
  for (auto &h : m->hist) {
    HIP_OK(hipEventCreate(&h.ev0));
    HIP_OK(hipEventCreate(&h.ev1));
  }
  return SRT_OK;
}

extern "C" void srt_model_destroy(srt_model *m) {
  if (!m) return;
  SRT_MODEL_SCOPE; // its tables and events live on m->device; the caller's device is back when this returns
  (void)ensure_model(m);
  if (m->d_coef) (void)hipFree(m->d_coef);
  if (m->d_pts) (void)hipFree(m->d_pts);
  if (m->d_xyz) (void)hipFree(m->d_xyz);
  if (m->d_cells) (void)hipFree(m->d_cells);
  if (m->d_model) (void)hipFree(m->d_model);
  for (auto &sl : m->slot) {
    for (int k = 0; k < 2; ++k) {
      if (sl.d_keys[k]) (void)hipFree(sl.d_keys[k]);
      if (sl.d_ids[k]) (void)hipFree(sl.d_ids[k]);
    }
    if (sl.d_sorttmp) (void)hipFree(sl.d_sorttmp);
    if (sl.d_stage) (void)hipFree(sl.d_stage);
    if (sl.d_blocks) (void)hipFree(sl.d_blocks);
    if (sl.done) (void)hipEventDestroy(sl.done);
  }
  for (auto &h : m->hist) {
    if (h.ev0) (void)hipEventDestroy(h.ev0);
    if (h.ev1) (void)hipEventDestroy(h.ev1);
  }
  if (m->d_common) (void)hipFree(m->d_common);
  for (void *q : m->io.p)
    if (q) (void)hipFree(q);
  delete m;
}
extern "C" int srt_model_trim(srt_model *m) {
  if (!m) return srt_set_error(SRT_EINVAL, "null model");
  SRT_MODEL_SCOPE;
  int rc = ensure_model(m);
  if (rc) return rc;
  std::lock_guard<std::mutex> hold(m->io_lock);
  for (auto &sl : m->slot) {
    if (sl.used) HIP_OK(hipEventSynchronize(sl.done)); // the launch that used this slot's scratch is over
    for (int k = 0; k < 2; ++k) {
      if (sl.d_keys[k]) (void)hipFree(sl.d_keys[k]);
      if (sl.d_ids[k]) (void)hipFree(sl.d_ids[k]);
      sl.d_keys[k] = nullptr;
      sl.d_ids[k] = nullptr;
    }
    if (sl.d_sorttmp) (void)hipFree(sl.d_sorttmp);
    if (sl.d_stage) (void)hipFree(sl.d_stage);
    if (sl.d_blocks) (void)hipFree(sl.d_blocks);
    sl.d_sorttmp = nullptr;
    sl.d_stage = nullptr;
    sl.d_blocks = nullptr;
    sl.sort_cap = sl.sorttmp_bytes = 0;
    sl.stage_blocks = sl.cand_blocks = 0;
    sl.stage_failed = sl.cand_failed = 0;
  }
  for (int k = 0; k < 9; ++k) {
    if (m->io.p[k]) (void)hipFree(m->io.p[k]);
    m->io.p[k] = nullptr;
    m->io.cap[k] = 0;
  }
  return SRT_OK;
}
// use_igrf (raytracer_driver.f95 --use_igrf; interp_dens_model_adapter.f95:236-241 and twins)
#include <dlfcn.h>
extern "C" int srt_model_set_field(srt_model *m, int use_igrf, int use_tsyganenko, const char *igrf_coeff_file) {
  if (!m) return srt_set_error(SRT_EINVAL, "null model");
  if (use_tsyganenko != 0 && use_tsyganenko != 1) return srt_set_error(SRT_EINVAL, "use_tsyganenko must be 0 or 1");
  if (use_igrf != 0 && use_igrf != 1) return srt_set_error(SRT_EINVAL, "use_igrf must be 0 or 1");
  SRT_MODEL_SCOPE;
  int rc = ensure_model(m);
  if (rc) return rc;
  FieldConst &f = m->cm.fld;
  if (use_igrf || use_tsyganenko) { // both need geopack's RECALC_08 for the date: coefficients, GEO->GSM matrix, dipole tilt
    std::string path;
    if (igrf_coeff_file && *igrf_coeff_file) path = igrf_coeff_file;
    else if (const char *e = getenv("SRT_IGRF_COEFFS")) path = e;
    else { // the table shipped beside the library: <pkg>/lib/libsrt_hip.so -> <pkg>/data/igrf_coeffs.txt
      Dl_info info;
      if (dladdr((const void *)&srt_model_set_field, &info) && info.dli_fname) {
        path = info.dli_fname;
        size_t k = path.rfind('/');
        path = (k == std::string::npos ? std::string(".") : path.substr(0, k)) + "/../data/igrf_coeffs.txt";
      }
    }
    std::string err;
    float G[105], H[105], REC[105];
    if (!srt_host::igrf_setup(path.c_str(), f.yearday, f.msec, G, H, REC, f.A, &f.psi, err)) return srt_set_error(SRT_EIO, "%s", err.c_str());
    for (int mm = 1; mm <= 14; ++mm)
      for (int n = mm; n <= 14; ++n) { // geopack's index n(n-1)/2 + m  ->  visiting order
        const int mn = n * (n - 1) / 2 + mm - 1, e = igrf_off(mm) + n - mm;
        f.Gv[e] = G[mn];
        f.Hv[e] = H[mn];
        f.Rv[e] = REC[mn];
      }
  }
  f.use_igrf = use_igrf;
  f.use_tsy = use_tsyganenko;
  HIP_OK(hipMemcpy(m->d_common, &m->cm, sizeof(Common), hipMemcpyHostToDevice));
  return SRT_OK;
}
// T04_s's PARMOD (driver flags --tsyganenko_Pdyn, _Dst, _ByIMF, _BzIMF, _W1 .. _W6; raytracer_driver.f95:292-341)
extern "C" int srt_model_set_tsyganenko_params(srt_model *m, const double parmod[10]) {
  if (!m || !parmod) return srt_set_error(SRT_EINVAL, "null argument");
  SRT_MODEL_SCOPE;
  int rc = ensure_model(m);
  if (rc) return rc;
  for (int i = 0; i < 10; ++i) m->cm.fld.parmod[i] = (float)parmod[i]; // real(parmod)
  HIP_OK(hipMemcpy(m->d_common, &m->cm, sizeof(Common), hipMemcpyHostToDevice));
  return SRT_OK;
}
extern "C" int srt_model_kind(const srt_model *m) { return m ? m->kind : 0; }
extern "C" int srt_model_nspec(const srt_model *m) { return m ? m->nspec : 0; }
extern "C" int64_t srt_model_device_bytes(const srt_model *m) { return m ? m->device_bytes : 0; }
extern "C" int srt_model_species(const srt_model *m, double qs[SRT_MAXSPEC], double ms[SRT_MAXSPEC]) {
  if (!m) return srt_set_error(SRT_EINVAL, "null model");
  for (int s = 0; s < SRT_MAXSPEC; ++s) {
    qs[s] = s < m->nspec ? m->cm.sp.q[s] : 0.0;
    ms[s] = s < m->nspec ? m->cm.sp.m[s] : 0.0;
  }
  return SRT_OK;
}

// ---- Ngo: normalisation of ane0 (ngo_dens_model.f95:120-123) needs one evaluation of dens() -------
__global__ void ngo_norm_kernel(NgoModel g, double z1, double sinz22, double latitu, double *out) {
  double Ns[4];
  g.dens_core(z1, sinz22, latitu, Ns);
  out[0] = Ns[0] * 1.0e-6; // ani(1)
}

extern "C" int srt_model_create_ngo(const char *configfile, int yearday, int msec, srt_model **out) {
  if (!configfile || !out) return srt_set_error(SRT_EINVAL, "null argument");
  DeviceScope srt_iscope_;
  int rc = srt_iscope_.enter_default();
  if (rc) return rc;
  srt_host::NgoConfig cfg;
  std::string err;
  if (!srt_host::read_newray(configfile, cfg, err)) return srt_set_error(SRT_EIO, "%s: %s", configfile, err.c_str());
  srt_model *m = new srt_model;
  m->kind = 1;
  m->nspec = 4;
  NgoModel &g = m->ngo;
  memset(&g, 0, sizeof g);
  g.r0 = 6370.0;
  g.pi32 = (double)3.141592653589793f; // default-real literal, ngo_dens_model.f95:36 (SURVEY A-6)
  g.num = cfg.num;
  g.kducts = cfg.kducts;
  g.kinit = 2;
  g.therm = cfg.therm;
  g.rbase = cfg.rbase;
  g.ane0 = cfg.ane0;
  for (int i = 0; i < 5; ++i) g.alpha0[i] = cfg.alpha0[i];
  g.rzero = cfg.rzero;
  g.scbot = cfg.scbot;
  g.lk = cfg.lk;
  g.expk = cfg.expk;
  g.ddk = cfg.ddk;
  g.rconsn = cfg.rconsn;
  g.scr = cfg.scr;
  for (int k = 0; k < 10; ++k) {
    g.l0[k] = cfg.l0[k]; g.def[k] = cfg.def[k]; g.dd[k] = cfg.dd[k];
    g.rducln[k] = cfg.rducln[k]; g.rducun[k] = cfg.rducun[k];
    g.rducls[k] = cfg.rducls[k]; g.rducus[k] = cfg.rducus[k];
    g.sidedu[k] = cfg.sidedu[k];
    g.hl2n[k] = cfg.hducln[k] * cfg.hducln[k]; g.hl2s[k] = cfg.hducls[k] * cfg.hducls[k];
    g.hu2n[k] = cfg.hducun[k] * cfg.hducun[k]; g.hu2s[k] = cfg.hducus[k] * cfg.hducus[k];
  }
  // ane0 <- ane0*dsdens/ani(1) at (dsrrng, dsrlat)  (:120-123); grarad is built on the float32 pi
  double radgra = 180.0 / g.pi32, grarad = 1.0 / radgra;
  double z2 = (90.0 - cfg.dsrlat) * grarad;
  double z1 = cfg.dsrrng * g.r0;
  double s2 = sin(z2);
  double *d_out = nullptr;
  double ani1 = 0.0;
  if (hipMalloc(&d_out, sizeof(double)) != hipSuccess) {
    delete m;
    return srt_set_error(SRT_ENOMEM, "hipMalloc failed");
  }
  // the latitude dens() sees during this call is the last satellite latitude read (:64)
  hipLaunchKernelGGL(ngo_norm_kernel, dim3(1), dim3(1), 0, 0, g, z1, s2 * s2, cfg.last_latitu, d_out);
  hipError_t e = hipMemcpy(&ani1, d_out, sizeof(double), hipMemcpyDeviceToHost);
  (void)hipFree(d_out);
  if (e != hipSuccess) {
    delete m;
    return srt_set_error(SRT_EDEVICE, "ngo normalisation kernel failed: %s", hipGetErrorString(e));
  }
  g.ane0 = g.ane0 * cfg.dsdens / ani1;
  const double e_ = 1.602e-19;
  double qs[4] = {e_ * -1.0, e_, e_, e_};
  double ms[4] = {9.10938188e-31, 1.6726e-27, 4.0 * 1.6726e-27, 16.0 * 1.6726e-27};
  fill_common(m->cm, 4, qs, ms, yearday, msec);
  rc = model_finish(m);
  if (rc) {
    srt_model_destroy(m);
    return rc;
  }
  *out = m;
  return SRT_OK;
}

// ---- interp: finite-difference derivatives + coefficient expansion on the device --------------------
struct GridDims {
  int nspec, nx, ny, nz;
};
__device__ __forceinline__ size_t gidx(const GridDims &g, int s, int i, int j, int k) {
  return (((size_t)k * g.ny + j) * g.nx + i) * g.nspec + s;
}
// tricubic_compute_finite_difference_derivatives, one axis (libtricubic.f95:736-790)
__global__ void fd_axis_kernel(GridDims g, const double *src, double *dst, int axis, double h) {
  size_t total = (size_t)g.nspec * g.nx * g.ny * g.nz;
  for (size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (size_t)gridDim.x * blockDim.x) {
    int s = (int)(id % g.nspec);
    size_t r = id / g.nspec;
    int i = (int)(r % g.nx);
    r /= g.nx;
    int j = (int)(r % g.ny);
    int k = (int)(r / g.ny);
    int c[3] = {i, j, k};
    int n[3] = {g.nx, g.ny, g.nz};
    int lo[3] = {i, j, k}, hi[3] = {i, j, k};
    int a = c[axis], na = n[axis];
    double v;
    if (a == 0) {
      hi[axis] = 1;
      v = (src[gidx(g, s, hi[0], hi[1], hi[2])] - src[gidx(g, s, lo[0], lo[1], lo[2])]) / h;
    } else if (a == na - 1) {
      lo[axis] = na - 2;
      v = (src[gidx(g, s, hi[0], hi[1], hi[2])] - src[gidx(g, s, lo[0], lo[1], lo[2])]) / h;
    } else {
      lo[axis] = a - 1;
      hi[axis] = a + 1;
      v = (src[gidx(g, s, hi[0], hi[1], hi[2])] - src[gidx(g, s, lo[0], lo[1], lo[2])]) / 2.0 / h;
    }
    dst[id] = v;
  }
}

struct ArrPtrs {
  const double *a[8];
};
__constant__ short c_tri_ptr[65];
__constant__ signed char c_tri_col[TRI_NNZ];
__constant__ signed char c_tri_val[TRI_NNZ];

// tricubic_get_coeff for every cell (libtricubic.f95:638-656,715-720) with the corner gathering,
// clamping and sticky derivative-zeroing flags of tricubic_interpolate_at (:859-921; SURVEY A-7).
// One 64-thread block per (cell, species): thread t first gathers constraint b[t] (t = 8*family +
// corner), then computes coefficient row t of the sparse 64x64 product.
__global__ __launch_bounds__(64) void build_coeffs_kernel(GridDims g, ArrPtrs arrs, double dx, double dy, double dz,
                                                          double *coef, long long npairs) {
  __shared__ double b[64];
  const int t = threadIdx.x;
  const int q = t >> 3, l = t & 7;
  for (long long pair = blockIdx.x; pair < npairs; pair += gridDim.x) {
    long long cell = pair / g.nspec;
    int s = (int)(pair % g.nspec);
    int ci = (int)(cell % (g.nx + 1));
    long long r = cell / (g.nx + 1);
    int cj = (int)(r % (g.ny + 1));
    int ck = (int)(r / (g.ny + 1));
    int it = ci + (l & 1), jt = cj + ((l >> 1) & 1), kt = ck + (l >> 2); // 1-based corner indices
    it = it < 1 ? 1 : (it > g.nx ? g.nx : it);
    jt = jt < 1 ? 1 : (jt > g.ny ? g.ny : jt);
    kt = kt < 1 ? 1 : (kt > g.nz ? g.nz : kt);
    // flags are set by the first clamped corner and never reset inside the corner loop
    bool fi = (ci == 0) || (ci == g.nx && l >= 1);
    bool fj = (cj == 0) || (cj == g.ny && l >= 2);
    bool fk = (ck == 0) || (ck == g.nz && l >= 4);
    double v = arrs.a[q][gidx(g, s, it - 1, jt - 1, kt - 1)];
    bool zero = false;
    switch (q) {
    case 1: v = v * dx; zero = fi; break;
    case 2: v = v * dy; zero = fj; break;
    case 3: v = v * dz; zero = fk; break;
    case 4: v = v * dx * dy; zero = fi || fj; break;
    case 5: v = v * dx * dz; zero = fi || fk; break;
    case 6: v = v * dy * dz; zero = fj || fk; break;
    case 7: v = v * dx * dy * dz; zero = fi || fj || fk; break;
    default: break;
    }
    __syncthreads();
    b[t] = zero ? 0.0 : v;
    __syncthreads();
    double acc = 0.0;
    for (int e = c_tri_ptr[t]; e < c_tri_ptr[t + 1]; ++e) acc += (double)c_tri_val[e] * b[(int)c_tri_col[e]];
    coef[(size_t)pair * 64 + t] = acc;
  }
}

// FD derivatives (unless given) + coefficient expansion + model object, from the 8 arrays already on the device.
// d_arr[0] = F; d_arr[1..7] = derivative blocks (contents ignored when !have_derivs).  The arrays are freed here.
static int interp_from_device_arrays(int nspec, int nx, int ny, int nz, const double bounds[6], const double *qs,
                                     const double *ms, double *d_arr[8], bool have_derivs, int yearday, int msec,
                                     srt_model **out) {
  const size_t nnode = (size_t)nx * ny * nz, n = nnode * nspec;
  const size_t ncell = (size_t)(nx + 1) * (ny + 1) * (nz + 1);
  GridDims g{nspec, nx, ny, nz};
  // interp_dens_model_adapter.f95:87-89
  double dx = (bounds[1] - bounds[0]) / (nx - 1.0);
  double dy = (bounds[3] - bounds[2]) / (ny - 1.0);
  double dz = (bounds[5] - bounds[4]) / (nz - 1.0);
  double *d_coef = nullptr;
  auto cleanup = [&]() {
    for (int a = 0; a < 8; ++a)
      if (d_arr[a]) {
        (void)hipFree(d_arr[a]);
        d_arr[a] = nullptr;
      }
  };
  hipError_t e = hipSuccess;
  if (!have_derivs) {
    for (int a = 1; a < 8 && e == hipSuccess; ++a) e = hipMemset(d_arr[a], 0, n * sizeof(double));
    int blocks = (int)((n + 255) / 256 < 65535 * 4 ? (n + 255) / 256 : 65535 * 4);
    auto fd = [&](int src, int dst, int axis, double h) {
      hipLaunchKernelGGL(fd_axis_kernel, dim3(blocks), dim3(256), 0, 0, g, (const double *)d_arr[src], d_arr[dst], axis, h);
    };
    // order and guards of libtricubic.f95:736-790
    if (nx > 2) fd(0, 1, 0, dx);
    if (ny > 2) fd(0, 2, 1, dy);
    if (nz > 2) fd(0, 3, 2, dz);
    if (nx > 2 && ny > 2) fd(2, 4, 0, dx);
    if (nx > 2 && nz > 2) fd(3, 5, 0, dx);
    if (ny > 2 && nz > 2) fd(3, 6, 1, dy);
    if (nx > 2 && ny > 2 && nz > 2) fd(6, 7, 0, dx);
  }
  if (e == hipSuccess) e = hipMalloc(&d_coef, ncell * nspec * 64 * sizeof(double));
  if (e != hipSuccess) {
    cleanup();
    return srt_set_error(e == hipErrorOutOfMemory ? SRT_ENOMEM : SRT_EDEVICE, "interp model setup: %s", hipGetErrorString(e));
  }
  (void)hipMemcpyToSymbol(HIP_SYMBOL(c_tri_ptr), TRI_PTR, sizeof TRI_PTR);
  (void)hipMemcpyToSymbol(HIP_SYMBOL(c_tri_col), TRI_COL, sizeof TRI_COL);
  (void)hipMemcpyToSymbol(HIP_SYMBOL(c_tri_val), TRI_VAL, sizeof TRI_VAL);
  ArrPtrs ap;
  for (int a = 0; a < 8; ++a) ap.a[a] = d_arr[a];
  long long npairs = (long long)ncell * nspec;
  int blocks = (int)(npairs < 262144 ? npairs : 262144);
  hipLaunchKernelGGL(build_coeffs_kernel, dim3(blocks), dim3(64), 0, 0, g, ap, dx, dy, dz, d_coef, npairs);
  e = hipDeviceSynchronize();
  cleanup();
  if (e != hipSuccess) {
    (void)hipFree(d_coef);
    return srt_set_error(SRT_EDEVICE, "coefficient build failed: %s", hipGetErrorString(e));
  }
  srt_model *m = new srt_model;
  m->kind = 3;
  m->nspec = nspec;
  m->d_coef = d_coef;
  m->device_bytes = (int64_t)(ncell * nspec * 64 * sizeof(double));
  m->interp.coef = d_coef;
  m->interp.nspec = nspec;
  m->interp.ax = Axis{bounds[0], dx, 1.0 / dx, nx};
  m->interp.ay = Axis{bounds[2], dy, 1.0 / dy, ny};
  m->interp.az = Axis{bounds[4], dz, 1.0 / dz, nz};
  fill_common(m->cm, nspec, qs, ms, yearday, msec);
  int rc = model_finish(m);
  if (rc) {
    srt_model_destroy(m);
    return rc;
  }
  *out = m;
  return SRT_OK;
}

static int alloc_grid_arrays(size_t n, double *d_arr[8]) {
  for (int a = 0; a < 8; ++a) d_arr[a] = nullptr;
  for (int a = 0; a < 8; ++a) {
    if (hipMalloc(&d_arr[a], n * sizeof(double)) != hipSuccess) {
      for (int b = 0; b < a; ++b) (void)hipFree(d_arr[b]);
      return srt_set_error(SRT_ENOMEM, "hipMalloc of grid arrays failed (%zu bytes each)", n * sizeof(double));
    }
  }
  return SRT_OK;
}

extern "C" int srt_model_create_interp(int nspec, int nx, int ny, int nz, const double bounds[6],
                                       const double *qs, const double *ms, const double *F,
                                       const double *const *derivs, int yearday, int msec, srt_model **out) {
  if (!bounds || !qs || !ms || !F || !out) return srt_set_error(SRT_EINVAL, "null argument");
  if (nspec < 1 || nspec > SRT_MAXSPEC)
    return srt_set_error(SRT_EINVAL, "nspec=%d unsupported (1..%d species: SRT_MAXSPEC, include/srt.h)", nspec, SRT_MAXSPEC);
  if (nx < 2 || ny < 2 || nz < 2) return srt_set_error(SRT_EINVAL, "grid must have >= 2 nodes per axis");
  DeviceScope srt_iscope_;
  int rc = srt_iscope_.enter_default();
  if (rc) return rc;
  const size_t n = (size_t)nx * ny * nz * nspec;
  if ((size_t)(nx + 1) * (ny + 1) * (nz + 1) >= (size_t)1 << 31) return srt_set_error(SRT_EINVAL, "grid too large");
  double *d_arr[8];
  rc = alloc_grid_arrays(n, d_arr);
  if (rc) return rc;
  hipError_t e = hipMemcpy(d_arr[0], F, n * sizeof(double), hipMemcpyHostToDevice);
  if (derivs)
    for (int a = 1; a < 8 && e == hipSuccess; ++a)
      e = hipMemcpy(d_arr[a], derivs[a - 1], n * sizeof(double), hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    for (int a = 0; a < 8; ++a) (void)hipFree(d_arr[a]);
    return srt_set_error(SRT_EDEVICE, "grid upload: %s", hipGetErrorString(e));
  }
  return interp_from_device_arrays(nspec, nx, ny, nz, bounds, qs, ms, d_arr, derivs != nullptr, yearday, msec, out);
}

// ---- the step before the path: sample a model on a regular grid, in log space, on the device ----
// gcpm_dens_model_buildgrid.f95:160-300 with any in-scope model in place of GCPM: node coordinates
// (/ (ind) /)*del + min (:163-179), f = log(Ns) (:212-214), and, for compder = 1, the seven explicit
// finite-difference blocks with d = 1e-3*|pos| (:197-201, :219-296) in the reference's own order of operations.
struct SampleArgs {
  double min[3], del[3];
  double *a[8];
  long long nnode;
  int compder;
};
template <class M, bool USE_LDS>
__global__ __launch_bounds__(64) void sample_grid_kernel(const M *__restrict__ mp, GridDims g, SampleArgs A) {
  const M &m = *mp;
  __shared__ __attribute__((aligned(16))) double tile[USE_LDS ? TILE_DOUBLES : 2];
  const long long id = (long long)blockIdx.x * WAVE + threadIdx.x;
  const bool live = id < A.nnode;
  const long long node = live ? id : A.nnode - 1; // every lane takes part in the (cooperative) lookups
  const int i = (int)(node % g.nx), j = (int)((node / g.nx) % g.ny), k = (int)(node / ((long long)g.nx * g.ny));
  double pos[3];
  {
#pragma clang fp contract(off)
    double px = (double)i * A.del[0], py = (double)j * A.del[1], pz = (double)k * A.del[2];
    pos[0] = px + A.min[0];
    pos[1] = py + A.min[1];
    pos[2] = pz + A.min[2];
  }
  double d = 1.0e-3 * sqrt(pos[0] * pos[0] + pos[1] * pos[1] + pos[2] * pos[2]);
  if (d == 0.0) d = 1.0e-3;
  auto lnN = [&](double sx, double sy, double sz, double (&L)[4]) { // log(Ns) at (pos + sx*dx) + sy*dy) + sz*dz
    double p[1][3] = {{(pos[0] + sx * d) + 0.0, (pos[1] + 0.0) + sy * d, ((pos[2] + 0.0) + 0.0) + sz * d}};
    double Ns[1][4];
    m.template density<1>(p, Ns, tile);
#pragma unroll
    for (int s = 0; s < 4; ++s) L[s] = log(Ns[0][s]);
  };
  auto put = [&](int blk, const double (&v)[4]) {
    if (live)
      for (int s = 0; s < g.nspec; ++s) A.a[blk][(size_t)node * g.nspec + s] = v[s];
  };
  double f[4], t[4], u[4];
  lnN(0, 0, 0, f);
  put(0, f);
  if (!A.compder) return;
  // first derivatives: tmp = log(N+) ; tmp = tmp - log(N-) ; tmp = tmp/d/2
  for (int ax = 0; ax < 3; ++ax) {
    lnN(ax == 0, ax == 1, ax == 2, t);
    lnN(-(ax == 0), -(ax == 1), -(ax == 2), u);
#pragma unroll
    for (int s = 0; s < 4; ++s) t[s] = (t[s] - u[s]) / d / 2.0;
    put(1 + ax, t);
  }
  // mixed second derivatives: (+,+) - (-,+) - (+,-) + (-,-) ; /d/d/4        pairs (x,y) (x,z) (y,z)
  for (int pr = 0; pr < 3; ++pr) {
    const int a0 = pr == 2 ? 1 : 0, a1 = pr == 0 ? 1 : 2;
    double sg[3];
    auto at = [&](double s0, double s1, double (&L)[4]) {
      sg[0] = sg[1] = sg[2] = 0.0;
      sg[a0] = s0;
      sg[a1] = s1;
      lnN(sg[0], sg[1], sg[2], L);
    };
    at(1, 1, t);
    at(-1, 1, u);
#pragma unroll
    for (int s = 0; s < 4; ++s) t[s] = t[s] - u[s];
    at(1, -1, u);
#pragma unroll
    for (int s = 0; s < 4; ++s) t[s] = t[s] - u[s];
    at(-1, -1, u);
#pragma unroll
    for (int s = 0; s < 4; ++s) t[s] = (t[s] + u[s]) / d / d / 4.0;
    put(4 + pr, t);
  }
  // third derivative: +++ -(-++) -(+-+) +(--+) -(++-) +(-+-) +(+--) -(---) ; /d/d/d/8
  {
    const double sx[8] = {1, -1, 1, -1, 1, -1, 1, -1}, sy[8] = {1, 1, -1, -1, 1, 1, -1, -1}, sz[8] = {1, 1, 1, 1, -1, -1, -1, -1};
    const double sign[8] = {1, -1, -1, 1, -1, 1, 1, -1};
    lnN(sx[0], sy[0], sz[0], t);
    for (int q = 1; q < 8; ++q) {
      lnN(sx[q], sy[q], sz[q], u);
#pragma unroll
      for (int s = 0; s < 4; ++s) t[s] = sign[q] > 0 ? t[s] + u[s] : t[s] - u[s];
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) t[s] = t[s] / d / d / d / 8.0;
    put(7, t);
  }
}

// fills d_arr[0] (and d_arr[1..7] when compder) on the device
static int sample_model_on_grid(srt_model *src, int compder, int nx, int ny, int nz, const double bounds[6],
                                double *d_arr[8]) {
  GridDims g{src->nspec, nx, ny, nz};
  SampleArgs A;
  // gcpm_dens_model_buildgrid.f95:161-163
  A.del[0] = (bounds[1] - bounds[0]) / (nx - 1.0);
  A.del[1] = (bounds[3] - bounds[2]) / (ny - 1.0);
  A.del[2] = (bounds[5] - bounds[4]) / (nz - 1.0);
  A.min[0] = bounds[0];
  A.min[1] = bounds[2];
  A.min[2] = bounds[4];
  for (int a = 0; a < 8; ++a) A.a[a] = d_arr[a];
  A.nnode = (long long)nx * ny * nz;
  A.compder = compder;
  const unsigned blocks = (unsigned)((A.nnode + WAVE - 1) / WAVE);
  if (src->kind == 1)
    hipLaunchKernelGGL((sample_grid_kernel<NgoModel, false>), dim3(blocks), dim3(WAVE), 0, 0, (const NgoModel *)src->d_model, g, A);
  else if (src->kind == 3)
    hipLaunchKernelGGL((sample_grid_kernel<InterpModel, true>), dim3(blocks), dim3(WAVE), 0, 0, (const InterpModel *)src->d_model, g, A);
  else
    hipLaunchKernelGGL((sample_grid_kernel<ScatteredModel, true>), dim3(blocks), dim3(WAVE), 0, 0, (const ScatteredModel *)src->d_model, g, A);
  hipError_t e = hipDeviceSynchronize();
  if (e != hipSuccess) return srt_set_error(SRT_EDEVICE, "grid sampling failed: %s", hipGetErrorString(e));
  return SRT_OK;
}

static int check_grid_request(srt_model *src, int nx, int ny, int nz, const double bounds[6]) {
  if (!src || !bounds) return srt_set_error(SRT_EINVAL, "null argument");
  if (nx < 2 || ny < 2 || nz < 2) return srt_set_error(SRT_EINVAL, "grid must have >= 2 nodes per axis");
  if ((size_t)(nx + 1) * (ny + 1) * (nz + 1) >= (size_t)1 << 31) return srt_set_error(SRT_EINVAL, "grid too large");
  if (!(bounds[1] > bounds[0]) || !(bounds[3] > bounds[2]) || !(bounds[5] > bounds[4])) return srt_set_error(SRT_EINVAL, "empty bounds");
  return SRT_OK;
}

extern "C" int srt_build_grid(srt_model *src, int compder, int nx, int ny, int nz, const double bounds[6], double *F,
                              double *derivs) {
  int rc = check_grid_request(src, nx, ny, nz, bounds);
  if (rc) return rc;
  SRT_MODEL_SCOPE;
  if ((rc = ensure_model(src))) return rc;
  if (!F || (compder && !derivs)) return srt_set_error(SRT_EINVAL, "null output");
  const size_t n = (size_t)nx * ny * nz * src->nspec;
  double *d_arr[8];
  rc = alloc_grid_arrays(n, d_arr);
  if (rc) return rc;
  rc = sample_model_on_grid(src, compder ? 1 : 0, nx, ny, nz, bounds, d_arr);
  hipError_t e = hipSuccess;
  if (!rc) {
    e = hipMemcpy(F, d_arr[0], n * sizeof(double), hipMemcpyDeviceToHost);
    if (compder)
      for (int a = 1; a < 8 && e == hipSuccess; ++a)
        e = hipMemcpy(derivs + (size_t)(a - 1) * n, d_arr[a], n * sizeof(double), hipMemcpyDeviceToHost);
  }
  for (int a = 0; a < 8; ++a) (void)hipFree(d_arr[a]);
  if (rc) return rc;
  if (e != hipSuccess) return srt_set_error(SRT_EDEVICE, "grid download: %s", hipGetErrorString(e));
  return SRT_OK;
}

extern "C" int srt_model_create_interp_from_model(srt_model *src, int compder, int nx, int ny, int nz,
                                                  const double bounds[6], int yearday, int msec, srt_model **out) {
  int rc = check_grid_request(src, nx, ny, nz, bounds);
  if (rc) return rc;
  if (!out) return srt_set_error(SRT_EINVAL, "null argument");
  SRT_MODEL_SCOPE; // the new model is built beside its source, on the source's device
  if ((rc = ensure_model(src))) return rc;
  const size_t n = (size_t)nx * ny * nz * src->nspec;
  double *d_arr[8];
  rc = alloc_grid_arrays(n, d_arr);
  if (rc) return rc;
  rc = sample_model_on_grid(src, compder ? 1 : 0, nx, ny, nz, bounds, d_arr);
  if (rc) {
    for (int a = 0; a < 8; ++a) (void)hipFree(d_arr[a]);
    return rc;
  }
  return interp_from_device_arrays(src->nspec, nx, ny, nz, bounds, src->cm.sp.q, src->cm.sp.m, d_arr, compder != 0,
                                   yearday, msec, out);
}

extern "C" int srt_model_create_interp_file(const char *gridfile, int yearday, int msec, srt_model **out) {
  if (!gridfile || !out) return srt_set_error(SRT_EINVAL, "null argument");
  srt_host::GridFile gf;
  std::string err;
  if (!srt_host::read_grid_file(gridfile, gf, err)) return srt_set_error(SRT_EIO, "%s: %s", gridfile, err.c_str());
  const double *dptr[7];
  for (int a = 0; a < 7; ++a) dptr[a] = gf.have_derivs ? gf.derivs[a].data() : nullptr;
  return srt_model_create_interp(gf.nspec, gf.nx, gf.ny, gf.nz, gf.bounds, gf.qs, gf.ms, gf.F.data(),
                                 gf.have_derivs ? dptr : nullptr, yearday, msec, out);
}

extern "C" int srt_model_create_scattered_file(const char *ptsfile, int yearday, int msec, double window_scale,
                                               int order, int exact, double local_window_scale, srt_model **out) {
  return srt_model_create_scattered_file_root(ptsfile, yearday, msec, window_scale, order, exact, local_window_scale, -1, out);
}

extern "C" int srt_model_create_scattered_file_root(const char *ptsfile, int yearday, int msec, double window_scale,
                                                    int order, int exact, double local_window_scale, int64_t root_sample,
                                                    srt_model **out) {
  if (!ptsfile || !out) return srt_set_error(SRT_EINVAL, "null argument");
  if (root_sample < -1) return srt_set_error(SRT_EINVAL, "root_sample must be -1 (none) or a 0-based record number");
  // orders 0..3: tabular_monomials (lsinterp_mod.f95:76-99), their own kernels; 4 and 5: generate_monomials (:114-164), one
  // cooperative path built to answer (srt_scattered.hpp gen_point); beyond: 84+ monomials, no room in a wave's LDS
  if (order < 0 || order > 5)
    return srt_set_error(SRT_EINVAL, "scattered_interp_order=%d: orders 0..5 are supported (the reference's generate_monomials orders "
                                     "N >= 6 are not built: stay on the Fortran path for them)", order);
  if (!(window_scale > 0) || !(local_window_scale > 0)) return srt_set_error(SRT_EINVAL, "window scales must be > 0");
  DeviceScope srt_iscope_;
  int rc = srt_iscope_.enter_default();
  if (rc) return rc;
  srt_host::ScatteredHost h;
  std::string err;
  // reach of the trace kernel's candidate blocks beyond the search radius (srt_scattered.hpp; SRT_SCATTERED_MARGIN overrides
  // the default for experiments; results do not depend on it beyond the order of summation)
  double bmargin = 0.125;
  if (const char *e = getenv("SRT_SCATTERED_MARGIN")) {
    const double v = atof(e);
    if (v >= 0.01 && v <= 1.0) bmargin = v;
  }
  if (!srt_host::build_scattered(ptsfile, window_scale, h, err, 1.0 + bmargin, (long long)root_sample))
    return srt_set_error(SRT_EIO, "%s: %s", ptsfile, err.c_str());
  srt_model *m = new srt_model;
  m->kind = 4;
  m->nspec = h.nspec;
  std::vector<double> xyz(3 * (size_t)h.npts);
  for (size_t i = 0; i < (size_t)h.npts; ++i)
    for (int c = 0; c < 3; ++c) xyz[(size_t)c * h.npts + i] = h.pts[8 * i + c];
  hipError_t e = hipMalloc(&m->d_pts, h.pts.size() * sizeof(double));
  if (e == hipSuccess) e = hipMalloc(&m->d_cells, h.cell_start.size() * sizeof(int));
  if (e == hipSuccess) e = hipMalloc(&m->d_xyz, xyz.size() * sizeof(double));
  if (e == hipSuccess) e = hipMemcpy(m->d_xyz, xyz.data(), xyz.size() * sizeof(double), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(m->d_pts, h.pts.data(), h.pts.size() * sizeof(double), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(m->d_cells, h.cell_start.data(), h.cell_start.size() * sizeof(int), hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    srt_model_destroy(m);
    return srt_set_error(e == hipErrorOutOfMemory ? SRT_ENOMEM : SRT_EDEVICE, "scattered model upload: %s", hipGetErrorString(e));
  }
  m->device_bytes = (int64_t)((h.pts.size() + xyz.size()) * sizeof(double) + h.cell_start.size() * sizeof(int));
  ScatteredModel &s = m->scat;
  s.pts = m->d_pts;
  s.xyz = m->d_xyz;
  s.cell_start = m->d_cells;
  for (int k = 0; k < 3; ++k) {
    s.origin[k] = h.origin[k];
    s.dims[k] = h.dims[k];
  }
  s.inv_cell = h.inv_cell;
  s.radius = h.radius;
  s.lws = local_window_scale;
  s.bmargin = bmargin;
  s.u11 = 1.1 * std::pow(h.radius * (1.0 + 5.0e-16), 1.1);
  s.inv_radius = 1.0 / h.radius;
  s.nspec = h.nspec;
  s.order = order;
  s.exact = exact;
  s.npts = h.npts;
  fill_common(m->cm, h.nspec, h.qs, h.ms, yearday, msec);
  rc = model_finish(m);
  if (rc) {
    srt_model_destroy(m);
    return rc;
  }
  *out = m;
  return SRT_OK;
}

// ------------------------------------------------------------------------------------------ launches
template <class K, class... Args>
static void launch_wave_blocks(K kernel, long long n, hipStream_t st, Args... args) {
  long long blocks = (n + WAVE - 1) / WAVE;
  hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(WAVE), 0, st, args...);
}

struct DevBuf {
  double *p = nullptr;
  ~DevBuf() {
    if (p) (void)hipFree(p);
  }
  int alloc(size_t n) { return hipMalloc(&p, n * sizeof(double)) == hipSuccess ? 0 : -1; }
};

// a scratch allocation of the scattered model was refused: say so ONCE per model (the launch still runs -- without staging
// every stencil takes the own-list path, without blocks every stencil scans its cells -- only slower), and do not try that
// size again at every launch (LaunchSlot::*_failed; srt_model_trim forgets it)
static void scratch_refused(srt_model *m, const char *what, size_t bytes) {
  if (m->warned_scratch) return;
  m->warned_scratch = true;
  fprintf(stderr, "libsrt_hip: hipMalloc of %.2f GB for the scattered model's %s was refused; tracing without them (slower). "
                  "srt_model_trim() on this or other models releases launch scratch.\n", (double)bytes / 1e9, what);
}
// SRT_SCATTERED_STAGING=0 in the environment: no staging buffer, i.e. the own-list path for every stencil (tests hold
// the two paths against each other)
static bool staging_enabled() {
  const char *e = getenv("SRT_SCATTERED_STAGING");
  return !(e && e[0] == '0');
}

// SRT_SCATTERED_BLOCKS=0: no candidate blocks, i.e. every shared-path stencil of the trace kernel scans the cells (A/B, tests)
static bool blocks_enabled() {
  const char *e = getenv("SRT_SCATTERED_BLOCKS");
  return !(e && e[0] == '0');
}

// scattered model: staging records for the one-wave blocks serving n items (srt_scattered.hpp shared_fit); beyond
// 4096 blocks (2 GiB) the kernels run without (own-list path)
static void stage_alloc(DevBuf &b, int64_t n) {
  const int64_t blocks = (n + WAVE - 1) / WAVE;
  if (!staging_enabled() || blocks > 4096 || b.alloc((size_t)blocks * ScatteredModel::REC_CAP * ScatteredModel::REC)) {
    b.p = nullptr;
    (void)hipGetLastError();
  }
}

static int upload(DevBuf &b, const double *h, size_t n) {
  if (b.alloc(n)) return srt_set_error(SRT_ENOMEM, "hipMalloc failed");
  HIP_OK(hipMemcpy(b.p, h, n * sizeof(double), hipMemcpyHostToDevice));
  return SRT_OK;
}

extern "C" int srt_plasma_params(srt_model *m, int64_t n, const double *x, double *qs, double *Ns, double *ms,
                                 double *nus, double *B0) {
  if (!m || !x || n < 0) return srt_set_error(SRT_EINVAL, "bad argument");
  if (n == 0) return SRT_OK;
  SRT_MODEL_SCOPE;
  int rc = ensure_model(m);
  if (rc) return rc;
  DevBuf dx, dout;
  if ((rc = upload(dx, x, 3 * n))) return rc;
  if (dout.alloc(19 * n)) return srt_set_error(SRT_ENOMEM, "hipMalloc failed");
  if (m->kind == 1) launch_wave_blocks(params_kernel<NgoModel, false>, n, 0, (const NgoModel *)m->d_model, (const Common *)m->d_common, (long long)n, (const double *)dx.p, dout.p);
  else if (m->kind == 3) launch_wave_blocks(params_kernel<InterpModel, true>, n, 0, (const InterpModel *)m->d_model, (const Common *)m->d_common, (long long)n, (const double *)dx.p, dout.p);
  else if (m->kind == 4)
    launch_wave_blocks(params_kernel<ScatteredModel, true>, n, 0, (const ScatteredModel *)m->d_model, (const Common *)m->d_common, (long long)n, (const double *)dx.p, dout.p);
  else return srt_set_error(SRT_EINVAL, "model kind %d unsupported", m->kind);
  std::vector<double> h(19 * n);
  HIP_OK(hipMemcpy(h.data(), dout.p, h.size() * sizeof(double), hipMemcpyDeviceToHost));
  for (int64_t i = 0; i < n; ++i) {
    for (int s = 0; s < 4; ++s) {
      if (qs) qs[4 * i + s] = h[19 * i + s];
      if (Ns) Ns[4 * i + s] = h[19 * i + 4 + s];
      if (ms) ms[4 * i + s] = h[19 * i + 8 + s];
      if (nus) nus[4 * i + s] = h[19 * i + 12 + s];
    }
    if (B0)
      for (int c = 0; c < 3; ++c) B0[3 * i + c] = h[19 * i + 16 + c];
  }
  return SRT_OK;
}

extern "C" int srt_dispersion(srt_model *m, int64_t n, const double *x, const double *k, const double *w, double *out) {
  if (!m || !x || !k || !w || !out || n < 0) return srt_set_error(SRT_EINVAL, "bad argument");
  if (n == 0) return SRT_OK;
  SRT_MODEL_SCOPE;
  int rc = ensure_model(m);
  if (rc) return rc;
  DevBuf dx, dk, dw, dout;
  if ((rc = upload(dx, x, 3 * n)) || (rc = upload(dk, k, 3 * n)) || (rc = upload(dw, w, n))) return rc;
  if (dout.alloc(10 * n)) return srt_set_error(SRT_ENOMEM, "hipMalloc failed");
  if (m->kind == 1)
    launch_wave_blocks(dispersion_kernel<NgoModel, false>, n, 0, (const NgoModel *)m->d_model, (const Common *)m->d_common, (long long)n, (const double *)dx.p, (const double *)dk.p, (const double *)dw.p, dout.p);
  else if (m->kind == 3)
    launch_wave_blocks(dispersion_kernel<InterpModel, true>, n, 0, (const InterpModel *)m->d_model, (const Common *)m->d_common, (long long)n, (const double *)dx.p, (const double *)dk.p, (const double *)dw.p, dout.p);
  else if (m->kind == 4)
    launch_wave_blocks(dispersion_kernel<ScatteredModel, true>, n, 0, (const ScatteredModel *)m->d_model, (const Common *)m->d_common, (long long)n, (const double *)dx.p, (const double *)dk.p, (const double *)dw.p, dout.p);
  else return srt_set_error(SRT_EINVAL, "model kind %d unsupported", m->kind);
  HIP_OK(hipMemcpy(out, dout.p, 10 * n * sizeof(double), hipMemcpyDeviceToHost));
  return SRT_OK;
}

extern "C" int srt_is_right_handed(int64_t n, const double *in, int32_t *out) {
  if (!in || !out || n < 0) return srt_set_error(SRT_EINVAL, "bad argument");
  if (n == 0) return SRT_OK;
  DeviceScope srt_iscope_;
  int rc = srt_iscope_.enter_default();
  if (rc) return rc;
  DevBuf din;
  if ((rc = upload(din, in, 5 * n))) return rc;
  int *d_out = nullptr;
  HIP_OK(hipMalloc(&d_out, n * sizeof(int)));
  hipLaunchKernelGGL(handedness_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, (long long)n, (const double *)din.p, d_out);
  hipError_t e = hipMemcpy(out, d_out, n * sizeof(int), hipMemcpyDeviceToHost);
  (void)hipFree(d_out);
  if (e != hipSuccess) return srt_set_error(SRT_EDEVICE, "handedness kernel: %s", hipGetErrorString(e));
  return SRT_OK;
}

extern "C" int srt_gradients(srt_model *m, int64_t n, const double *x, const double *k, const double *w, double del, double *out) {
  if (!m || !x || !k || !w || !out || n < 0) return srt_set_error(SRT_EINVAL, "bad argument");
  if (n == 0) return SRT_OK;
  SRT_MODEL_SCOPE;
  int rc = ensure_model(m);
  if (rc) return rc;
  DevBuf dx, dk, dw, dout;
  if ((rc = upload(dx, x, 3 * n)) || (rc = upload(dk, k, 3 * n)) || (rc = upload(dw, w, n))) return rc;
  if (dout.alloc(14 * n)) return srt_set_error(SRT_ENOMEM, "hipMalloc failed");
  if (m->kind == 1)
    launch_wave_blocks(gradients_kernel<NgoModel, false>, n, 0, (const NgoModel *)m->d_model, (const Common *)m->d_common, (long long)n, (const double *)dx.p, (const double *)dk.p, (const double *)dw.p, del, dout.p, (double *)nullptr);
  else if (m->kind == 3)
    launch_wave_blocks(gradients_kernel<InterpModel, true>, n, 0, (const InterpModel *)m->d_model, (const Common *)m->d_common, (long long)n, (const double *)dx.p, (const double *)dk.p, (const double *)dw.p, del, dout.p, (double *)nullptr);
  else if (m->kind == 4) {
    DevBuf stage;
    stage_alloc(stage, n);
    launch_wave_blocks(gradients_kernel<ScatteredModel, true>, n, 0, (const ScatteredModel *)m->d_model, (const Common *)m->d_common, (long long)n, (const double *)dx.p, (const double *)dk.p, (const double *)dw.p, del, dout.p, stage.p);
    HIP_OK(hipDeviceSynchronize()); // `stage` is freed at the end of this scope
  }
  else return srt_set_error(SRT_EINVAL, "model kind %d unsupported", m->kind);
  HIP_OK(hipMemcpy(out, dout.p, 14 * n * sizeof(double), hipMemcpyDeviceToHost));
  return SRT_OK;
}

extern "C" int srt_rk_step(srt_model *m, int64_t n, const double *args, const double *dt, double del, double *out) {
  if (!m || !args || !dt || !out || n < 0) return srt_set_error(SRT_EINVAL, "bad argument");
  if (n == 0) return SRT_OK;
  SRT_MODEL_SCOPE;
  int rc = ensure_model(m);
  if (rc) return rc;
  DevBuf da, dd, dout;
  if ((rc = upload(da, args, 7 * n)) || (rc = upload(dd, dt, n))) return rc;
  if (dout.alloc(21 * n)) return srt_set_error(SRT_ENOMEM, "hipMalloc failed");
  if (m->kind == 1)
    launch_wave_blocks(rkstep_kernel<NgoModel, false>, n, 0, (const NgoModel *)m->d_model, (const Common *)m->d_common, (long long)n, (const double *)da.p, (const double *)dd.p, del, dout.p, (double *)nullptr);
  else if (m->kind == 3)
    launch_wave_blocks(rkstep_kernel<InterpModel, true>, n, 0, (const InterpModel *)m->d_model, (const Common *)m->d_common, (long long)n, (const double *)da.p, (const double *)dd.p, del, dout.p, (double *)nullptr);
  else if (m->kind == 4) {
    DevBuf stage;
    stage_alloc(stage, n);
    launch_wave_blocks(rkstep_kernel<ScatteredModel, true>, n, 0, (const ScatteredModel *)m->d_model, (const Common *)m->d_common, (long long)n, (const double *)da.p, (const double *)dd.p, del, dout.p, stage.p);
    HIP_OK(hipDeviceSynchronize());
  }
  else return srt_set_error(SRT_EINVAL, "model kind %d unsupported", m->kind);
  HIP_OK(hipMemcpy(out, dout.p, 21 * n * sizeof(double), hipMemcpyDeviceToHost));
  return SRT_OK;
}


// ---- the random / adaptive sample-set builder (kernels and the level-by-level scheme: srt_sampler.hpp) ----
namespace {
struct DevMem { // grow-only device buffer
  void *p = nullptr;
  size_t cap = 0;
  ~DevMem() {
    if (p) (void)hipFree(p);
  }
  // at least `bytes`; the first `keep` bytes survive a reallocation
  int reserve(size_t bytes, size_t keep = 0) {
    if (bytes <= cap) return 0;
    size_t want = bytes + bytes / 2 + 4096;
    void *q = nullptr;
    if (hipMalloc(&q, want) != hipSuccess) return -1;
    if (p && keep && hipMemcpy(q, p, keep, hipMemcpyDeviceToDevice) != hipSuccess) {
      (void)hipFree(q);
      return -1;
    }
    if (p) (void)hipFree(p);
    p = q;
    cap = want;
    return 0;
  }
  template <class T> T *as() { return (T *)p; }
};
} // namespace

static void smp_eval(srt_model *src, long long n, double *rec) {
  if (n <= 0) return;
  if (src->kind == 1) launch_wave_blocks(smp_eval_kernel<NgoModel, false>, n, 0, (const NgoModel *)src->d_model, n, rec);
  else if (src->kind == 3) launch_wave_blocks(smp_eval_kernel<InterpModel, true>, n, 0, (const InterpModel *)src->d_model, n, rec);
  else launch_wave_blocks(smp_eval_kernel<ScatteredModel, true>, n, 0, (const ScatteredModel *)src->d_model, n, rec);
}

extern "C" int srt_build_samples(srt_model *src, const srt_sampler_params *sp, int64_t n_in, const double *in_pts,
                                 int64_t *n_out, double **out, int64_t stage_counts[6]) {
  if (!src || !sp || !n_out || !out || n_in < 0 || (n_in > 0 && !in_pts)) return srt_set_error(SRT_EINVAL, "bad argument");
  if (src->kind != 1 && src->kind != 3 && src->kind != 4) return srt_set_error(SRT_EINVAL, "model kind %d unsupported", src->kind);
  const double *bd = sp->bounds;
  if (!(bd[1] > bd[0]) || !(bd[3] > bd[2]) || !(bd[5] > bd[4])) return srt_set_error(SRT_EINVAL, "empty bounds");
  if (sp->n_zero_altitude < 0 || sp->n_iri_pad < 0 || sp->n_initial_radial < 0 || sp->n_initial_uniform < 0 || sp->max_recursion < 0 ||
      sp->numincrease < 0)
    return srt_set_error(SRT_EINVAL, "negative count");
  SRT_MODEL_SCOPE;
  int rc = ensure_model(src);
  if (rc) return rc;
  const int nspec = src->nspec, ninc = sp->numincrease ? sp->numincrease : 5;
  if (ninc > WAVE) return srt_set_error(SRT_EINVAL, "numincrease > %d", WAVE);
  const int max_passes = sp->max_passes > 0 ? sp->max_passes : 64;
  SmpBox root;
  for (int c = 0; c < 3; ++c) {
    root.lo[c] = bd[2 * c];
    root.hi[c] = bd[2 * c + 1];
  }
  root.id = 1;
  int64_t counts[6] = {n_in, 0, 0, 0, 0, 0};
  const size_t RB = SMP_REC * sizeof(double);
  DevMem pool, slot, stage, valid;
  long long npool = 0;
#define SMP_OK(expr)                                                                        \
  do {                                                                                      \
    hipError_t e_ = (expr);                                                                 \
    if (e_ != hipSuccess) return srt_set_error(SRT_EDEVICE, "%s: %s", #expr, hipGetErrorString(e_)); \
  } while (0)
#define SMP_MEM(expr)                                                           \
  do {                                                                          \
    if ((expr)) return srt_set_error(SRT_ENOMEM, "device allocation failed (sampler)"); \
  } while (0)

  // (0) points of an existing file: part of the initial set and of the output (:210-227)
  if (n_in > 0) {
    std::vector<double> h((size_t)n_in * SMP_REC, 0.0);
    for (int64_t i = 0; i < n_in; ++i)
      for (int c = 0; c < 3 + nspec; ++c) h[(size_t)i * SMP_REC + c] = in_pts[(size_t)i * (3 + nspec) + c];
    SMP_MEM(pool.reserve(h.size() * sizeof(double)));
    SMP_OK(hipMemcpy(pool.p, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice));
    npool = n_in;
  }
  // shell and uniform stages share one routine: positions, f(x), then append (device) or compact (host)
  std::vector<double> tail; // zero-altitude and ionosphere samples, appended to the output after the pool
  auto run_stage = [&](int stg, long long n, double rmin, double rmax, bool to_pool, int64_t &count) -> int {
    if (n <= 0) return SRT_OK;
    SMP_MEM(stage.reserve((size_t)n * RB));
    SMP_MEM(valid.reserve((size_t)n * sizeof(int)));
    hipLaunchKernelGGL(smp_stage_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, stg, n, (unsigned long long)sp->seed, root, rmin,
                       rmax, stage.as<double>(), valid.as<int>());
    smp_eval(src, n, stage.as<double>());
    SMP_OK(hipDeviceSynchronize());
    if (to_pool) {
      std::vector<int> hv((size_t)n);
      SMP_OK(hipMemcpy(hv.data(), valid.p, (size_t)n * sizeof(int), hipMemcpyDeviceToHost));
      for (long long i = 0; i < n; ++i)
        if (!hv[i])
          return srt_set_error(SRT_EINVAL, "radial stage: sample %lld found no point of the shell inside the box in %d attempts", i, SMP_MAXTRY);
      SMP_MEM(pool.reserve((size_t)(npool + n) * RB, (size_t)npool * RB));
      SMP_OK(hipMemcpy(pool.as<double>() + (size_t)npool * SMP_REC, stage.p, (size_t)n * RB, hipMemcpyDeviceToDevice));
      npool += n;
      count = n;
    } else {
      std::vector<int> hv((size_t)n);
      std::vector<double> hr((size_t)n * SMP_REC);
      SMP_OK(hipMemcpy(hv.data(), valid.p, (size_t)n * sizeof(int), hipMemcpyDeviceToHost));
      SMP_OK(hipMemcpy(hr.data(), stage.p, (size_t)n * RB, hipMemcpyDeviceToHost));
      for (long long i = 0; i < n; ++i)
        if (hv[i]) {
          for (int c = 0; c < 3 + nspec; ++c) tail.push_back(hr[(size_t)i * SMP_REC + c]);
          ++count;
        }
    }
    return SRT_OK;
  };
  // (1) radial (:228-272): rmin = R_E, rmax = the farthest corner of the box
  {
    double r2 = 0.0;
    for (int q = 0; q < 8; ++q) {
      const double x = bd[q & 4 ? 1 : 0], y = bd[q & 2 ? 3 : 2], z = bd[q & 1 ? 5 : 4];
      const double v = x * x + y * y + z * z;
      if (v > r2) r2 = v;
    }
    if ((rc = run_stage(SMP_RADIAL, sp->n_initial_radial, R_E, sqrt(r2), true, counts[1]))) return rc;
  }
  // (2) uniform (:275-296)
  if ((rc = run_stage(SMP_UNIFORM, sp->n_initial_uniform, 0.0, 0.0, true, counts[2]))) return rc;

  // (3) adaptive (:298-347): tol halves until adaptive_nmax adaptive samples exist
  if (sp->adaptive_nmax > 0) {
    DevMem keys0, keys1, idx0, idx1, sorttmp, callsA, callsB, halves, cand, nadd, refine, addA, addoff, childoff, scantmp;
    DevMem *calls = &callsA, *children = &callsB;
    int64_t nsamples = 0;
    double tol = sp->initial_tol;
    for (int pass = 0; nsamples < sp->adaptive_nmax && pass < max_passes; ++pass, tol = tol / 2.0) {
      if (npool >= (1ll << 31) - (1ll << 26)) return srt_set_error(SRT_EINVAL, "sample pool too large");
      SMP_MEM(slot.reserve((size_t)npool * sizeof(int)));
      if (npool > 0) hipLaunchKernelGGL(smp_fill_int, dim3((unsigned)((npool + 255) / 256)), dim3(256), 0, 0, npool, slot.as<int>(), 0);
      SMP_MEM(calls->reserve(sizeof(SmpBox)));
      SMP_OK(hipMemcpy(calls->p, &root, sizeof(SmpBox), hipMemcpyHostToDevice));
      long long ncalls = 1;
      for (int depth = 0; depth <= sp->max_recursion && ncalls > 0; ++depth) {
        const int dim = depth % 3;
        const long long nhalf = 2 * ncalls;
        if (nhalf > (1ll << 26)) return srt_set_error(SRT_EINVAL, "adaptive sampler: %lld half-boxes at depth %d (tolerance too small)", nhalf, depth);
        const long long ncand = nhalf * 2 * ninc;
        SMP_MEM(keys0.reserve((size_t)npool * 4 + 4) || keys1.reserve((size_t)npool * 4 + 4) || idx0.reserve((size_t)npool * 4 + 4) ||
                idx1.reserve((size_t)npool * 4 + 4));
        SMP_MEM(halves.reserve((size_t)nhalf * sizeof(SmpBox)) || cand.reserve((size_t)ncand * RB));
        SMP_MEM(nadd.reserve((size_t)nhalf * 4) || refine.reserve((size_t)nhalf * 4) || addA.reserve((size_t)nhalf * 4) ||
                addoff.reserve((size_t)nhalf * 4) || childoff.reserve((size_t)nhalf * 4));
        if (npool > 0) {
          hipLaunchKernelGGL(smp_keys_kernel, dim3((unsigned)((npool + 255) / 256)), dim3(256), 0, 0, npool, pool.as<double>(), slot.as<int>(),
                             calls->as<SmpBox>(), dim, (unsigned)nhalf, keys0.as<unsigned>(), idx0.as<int>());
          int bits = 1;
          while ((1ll << bits) <= nhalf) ++bits;
          size_t need = 0;
          SMP_OK(hipcub::DeviceRadixSort::SortPairs(nullptr, need, keys0.as<unsigned>(), keys1.as<unsigned>(), idx0.as<int>(), idx1.as<int>(),
                                                    (int)npool, 0, bits, 0));
          SMP_MEM(sorttmp.reserve(need));
          SMP_OK(hipcub::DeviceRadixSort::SortPairs(sorttmp.p, need, keys0.as<unsigned>(), keys1.as<unsigned>(), idx0.as<int>(), idx1.as<int>(),
                                                    (int)npool, 0, bits, 0));
        }
        hipLaunchKernelGGL(smp_cand_kernel, dim3((unsigned)((ncand + 255) / 256)), dim3(256), 0, 0, (int)nhalf, calls->as<SmpBox>(), dim,
                           (unsigned long long)sp->seed, (unsigned long long)pass, ninc, halves.as<SmpBox>(), cand.as<double>());
        smp_eval(src, ncand, cand.as<double>());
        hipLaunchKernelGGL(smp_stats_kernel, dim3((unsigned)nhalf), dim3(WAVE), 0, 0, (int)nhalf, npool, keys1.as<unsigned>(), idx1.as<int>(),
                           pool.as<double>(), cand.as<double>(), halves.as<SmpBox>(), dim, nspec, ninc, tol, nadd.as<int>(), refine.as<int>(),
                           addA.as<int>());
        size_t need = 0;
        SMP_OK(hipcub::DeviceScan::ExclusiveSum(nullptr, need, nadd.as<int>(), addoff.as<int>(), (int)nhalf, 0));
        SMP_MEM(scantmp.reserve(need));
        SMP_OK(hipcub::DeviceScan::ExclusiveSum(scantmp.p, need, nadd.as<int>(), addoff.as<int>(), (int)nhalf, 0));
        SMP_OK(hipcub::DeviceScan::ExclusiveSum(scantmp.p, need, refine.as<int>(), childoff.as<int>(), (int)nhalf, 0));
        int last[4];
        SMP_OK(hipMemcpy(&last[0], addoff.as<int>() + nhalf - 1, 4, hipMemcpyDeviceToHost));
        SMP_OK(hipMemcpy(&last[1], nadd.as<int>() + nhalf - 1, 4, hipMemcpyDeviceToHost));
        SMP_OK(hipMemcpy(&last[2], childoff.as<int>() + nhalf - 1, 4, hipMemcpyDeviceToHost));
        SMP_OK(hipMemcpy(&last[3], refine.as<int>() + nhalf - 1, 4, hipMemcpyDeviceToHost));
        const long long total_add = (long long)last[0] + last[1], nchild = (long long)last[2] + last[3];
        const int make_children = depth + 1 <= sp->max_recursion ? 1 : 0; // a call beyond maxdepth returns at once (:65-68)
        if (npool + total_add >= (1ll << 31) - (1ll << 26)) return srt_set_error(SRT_EINVAL, "sample pool too large");
        SMP_MEM(pool.reserve((size_t)(npool + total_add) * RB, (size_t)npool * RB));
        SMP_MEM(slot.reserve((size_t)(npool + total_add) * sizeof(int), (size_t)npool * sizeof(int)));
        SMP_MEM(children->reserve((size_t)(nchild > 0 ? nchild : 1) * sizeof(SmpBox)));
        if (npool > 0)
          hipLaunchKernelGGL(smp_reslot_kernel, dim3((unsigned)((npool + 255) / 256)), dim3(256), 0, 0, npool, keys0.as<unsigned>(),
                             (unsigned)nhalf, refine.as<int>(), childoff.as<int>(), make_children, slot.as<int>());
        hipLaunchKernelGGL(smp_commit_kernel, dim3((unsigned)((nhalf + 255) / 256)), dim3(256), 0, 0, (int)nhalf, ninc, addA.as<int>(),
                           refine.as<int>(), addoff.as<int>(), childoff.as<int>(), make_children, cand.as<double>(), halves.as<SmpBox>(),
                           npool, pool.as<double>(), slot.as<int>(), children->as<SmpBox>());
        SMP_OK(hipDeviceSynchronize());
        npool += total_add;
        nsamples += total_add;
        std::swap(calls, children);
        ncalls = make_children ? nchild : 0;
      }
    }
    counts[3] = nsamples;
  }
  // (4) zero altitude (:349-377), (5) ionosphere shell R_E .. R_E + 2000 km (:379-405): one draw each, kept when inside
  if ((rc = run_stage(SMP_ZEROALT, sp->n_zero_altitude, R_E, R_E, false, counts[4]))) return rc;
  if ((rc = run_stage(SMP_IRI, sp->n_iri_pad, R_E, R_E + 2000000.0, false, counts[5]))) return rc;

  const size_t w = 3 + (size_t)nspec, ntail = tail.size() / w;
  double *res = (double *)malloc(((size_t)npool + ntail + 1) * w * sizeof(double));
  if (!res) return srt_set_error(SRT_ENOMEM, "malloc failed");
  if (npool > 0) {
    std::vector<double> h((size_t)npool * SMP_REC);
    hipError_t e = hipMemcpy(h.data(), pool.p, h.size() * sizeof(double), hipMemcpyDeviceToHost);
    if (e != hipSuccess) {
      free(res);
      return srt_set_error(SRT_EDEVICE, "sample download: %s", hipGetErrorString(e));
    }
    for (long long i = 0; i < npool; ++i)
      for (size_t c = 0; c < w; ++c) res[(size_t)i * w + c] = h[(size_t)i * SMP_REC + c];
  }
  if (ntail) memcpy(res + (size_t)npool * w, tail.data(), tail.size() * sizeof(double));
  *out = res;
  *n_out = npool + (int64_t)ntail;
  if (stage_counts)
    for (int q = 0; q < 6; ++q) stage_counts[q] = counts[q];
  return SRT_OK;
#undef SMP_OK
#undef SMP_MEM
}

// ------------------------------------------------------------------------------------------ hot path
extern "C" int32_t srt_rows_per_ray(const srt_params *p) {
  if (!p || p->maxsteps < 1) return 0;
  int per = p->outputper < 1 ? 1 : p->outputper;
  return (p->maxsteps + per - 1) / per;
}

static int check_params(const srt_params *p) {
  if (!p) return srt_set_error(SRT_EINVAL, "null params");
  if (p->maxsteps < 1) return srt_set_error(SRT_EINVAL, "maxsteps must be >= 1");
  if (p->root != 1 && p->root != 2) return srt_set_error(SRT_EINVAL, "root must be 1 or 2");
  if (!(p->del > 0.0)) return srt_set_error(SRT_EINVAL, "del must be > 0");
  if (p->ray_order < 0 || p->ray_order > 2) return srt_set_error(SRT_EINVAL, "ray_order must be 0, 1 or 2");
  return SRT_OK;
}

// ray_order = 1: Morton code of the launch cell (9 bits per axis) per ray
__device__ __forceinline__ unsigned spread3(unsigned v) { // 0b abc -> 0b a00b00c
  v &= 0x3ffu;
  v = (v | (v << 16)) & 0x030000ffu;
  v = (v | (v << 8)) & 0x0300f00fu;
  v = (v | (v << 4)) & 0x030c30c3u;
  v = (v | (v << 2)) & 0x09249249u;
  return v;
}
// two_class (ray_order = 2, an experiment switch: HISTORY.md section 9): rays that are likely to stop early -- above 6 kHz and
// launched inwards: 8.5 % of the BASELINE launch set, mean 75 rows against 197 -- sort behind all others (key bit 30: above the Morton code, 10 bits per axis = bits 0..29)
__global__ void ray_keys_kernel(const InterpModel *mp, const double *pos0 /* SoA [3][n] */, const double *dir0, const double *w0,
                                int two_class, long long n, unsigned *keys, int *ids) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const InterpModel &m = *mp;
  double xl;
  const double x = pos0[i], y = pos0[n + i], z = pos0[2 * n + i];
  unsigned ci = (unsigned)m.ax.locate(x, xl), cj = (unsigned)m.ay.locate(y, xl), ck = (unsigned)m.az.locate(z, xl);
  unsigned key = spread3(ci) | (spread3(cj) << 1) | (spread3(ck) << 2);
  if (two_class) {
    const bool inward = dir0[i] * x + dir0[n + i] * y + dir0[2 * n + i] * z < 0.0;
    if (inward && w0[i] > 2.0 * PI * 6.0e3) key |= 1u << 30;
  }
  keys[i] = key;
  ids[i] = (int)i;
}

extern "C" int srt_trace_batch_device(srt_model *m, const srt_params *p, int64_t nrays, const double *d_pos0,
                                      const double *d_dir0, const double *d_w0, double *d_rows, int32_t *d_nrows,
                                      int32_t *d_stopcond, int64_t *d_counters, void *stream) {
  if (!m || !d_pos0 || !d_dir0 || !d_w0 || !d_rows || !d_nrows || !d_stopcond || !d_counters || nrays < 0)
    return srt_set_error(SRT_EINVAL, "bad argument");
  int rc = check_params(p);
  if (rc) return rc;
  SRT_MODEL_SCOPE;
  if ((rc = ensure_model(m))) return rc;
  hipStream_t st = (hipStream_t)stream;
  TraceArgs a;
  a.pos0 = d_pos0;
  a.dir0 = d_dir0;
  a.w0 = d_w0;
  a.nrays = nrays;
  a.rows = d_rows;
  a.nrows = d_nrows;
  a.stopcond = d_stopcond;
  a.counters = (unsigned long long *)d_counters;
  a.p.dt0 = p->dt0; a.p.dtmax = p->dtmax; a.p.tmax = p->tmax; a.p.maxerr = p->maxerr;
  a.p.minalt = p->minalt; a.p.del = p->del;
  a.p.maxsteps = p->maxsteps; a.p.root = p->root; a.p.fixedstep = p->fixedstep;
  a.p.outputper = p->outputper < 1 ? 1 : p->outputper;
  a.p.first_attempt_policy = p->first_attempt_policy;
  a.p.refill_threshold = p->refill_threshold;
  a.p.slots = srt_rows_per_ray(p);
  a.order = nullptr;
  HIP_OK(hipMemsetAsync(d_counters, 0, 4 * sizeof(int64_t), st));
  // persistent grid: enough one-wave blocks to fill the chip, never more than the rays need
  // interp: 34 KiB of LDS per wave, 512 registers per lane: one wave per SIMD; scattered: 18.5 KiB, <= 256 registers: two
  int per_cu = m->kind == 3 ? 4 : (m->kind == 4 ? 4 * ScatteredModel::WAVES_PER_EU : 8);
  if (const char *e = getenv("SRT_WAVES_PER_CU")) {
    const int v = atoi(e);
    if (v >= 1 && v <= per_cu) per_cu = v;
  }
  long long grid = (long long)m->cu_count * per_cu;
  // Rays a wave holds at most.  64 = every lane.  (Measured for the Ngo model, whose launch at BASELINE config[1] is as long as
  // its longest ray: with at most 8 rays per wave every trip runs in tail mode -- the 8 stencil points of a ray on 8 lanes,
  // srt_models.hpp -- but a tail-mode trip is only 2.2x shorter than a full one while serving 8x fewer rays: 47.5 -> 51.6 ms.
  // SRT_WAVE_CAP / SRT_WAVES_PER_CU remain as experiment switches; a ray's arithmetic does not depend on either.)
  int cap = WAVE;
  if (const char *e = getenv("SRT_WAVE_CAP")) {
    const int v = atoi(e);
    if (v >= 1 && v <= WAVE) cap = v;
  }
  a.p.wave_cap = cap;
  long long want = (nrays + cap - 1) / cap;
  if (grid > want) grid = want;
  if (grid < 1) grid = 1;
  // scratch slot: one whose last launch is over (preferring one that holds buffers already), else a fresh one, else -- all
  // busy -- the next in turn, which this stream then waits for
  int si = -1;
  for (int k = 0; k < srt_model::NSLOT && si < 0; ++k)
    if (m->slot[k].used) {
      const hipError_t q = hipEventQuery(m->slot[k].done);
      if (q == hipSuccess) si = k;
      else if (q != hipErrorNotReady) HIP_OK(q);
      else (void)hipGetLastError();
    }
  for (int k = 0; k < srt_model::NSLOT && si < 0; ++k)
    if (!m->slot[k].used) si = k;
  if (si < 0) {
    si = m->rr_slot;
    m->rr_slot = (m->rr_slot + 1) % srt_model::NSLOT;
  }
  srt_model::LaunchSlot &sl = m->slot[si];
  if (sl.used) HIP_OK(hipStreamWaitEvent(st, sl.done, 0)); // the slot's previous launch (maybe on another stream) is over
  srt_model::LaunchTimes &lt = m->hist[m->next_hist];
  HIP_OK(hipEventRecord(lt.ev0, st));
  if (p->ray_order >= 1 && m->kind == 3 && nrays > WAVE && nrays < (1ll << 31) && m->interp.ax.n < 1023 &&
      m->interp.ay.n < 1023 && m->interp.az.n < 1023) {
    // work through the launch set in the order of the rays' launch cells (inside the timed region)
    if ((size_t)nrays > sl.sort_cap) {
      if (sl.used) HIP_OK(hipEventSynchronize(sl.done)); // about to free what that launch may still read
      for (int k = 0; k < 2; ++k) {
        if (sl.d_keys[k]) (void)hipFree(sl.d_keys[k]);
        if (sl.d_ids[k]) (void)hipFree(sl.d_ids[k]);
        sl.d_keys[k] = nullptr;
        sl.d_ids[k] = nullptr;
      }
      sl.sort_cap = 0;
      for (int k = 0; k < 2; ++k) {
        HIP_OK(hipMalloc(&sl.d_keys[k], (size_t)nrays * sizeof(unsigned)));
        HIP_OK(hipMalloc(&sl.d_ids[k], (size_t)nrays * sizeof(int)));
      }
      sl.sort_cap = (size_t)nrays;
    }
    size_t need = 0;
    HIP_OK(hipcub::DeviceRadixSort::SortPairs(nullptr, need, sl.d_keys[0], sl.d_keys[1], sl.d_ids[0], sl.d_ids[1], (int)nrays, 0, 31, st));
    if (need > sl.sorttmp_bytes) {
      if (sl.used) HIP_OK(hipEventSynchronize(sl.done));
      if (sl.d_sorttmp) (void)hipFree(sl.d_sorttmp);
      sl.d_sorttmp = nullptr;
      sl.sorttmp_bytes = 0;
      HIP_OK(hipMalloc(&sl.d_sorttmp, need));
      sl.sorttmp_bytes = need;
    }
    hipLaunchKernelGGL(ray_keys_kernel, dim3((unsigned)((nrays + 255) / 256)), dim3(256), 0, st, (const InterpModel *)m->d_model,
                       d_pos0, d_dir0, d_w0, p->ray_order == 2 ? 1 : 0, (long long)nrays, sl.d_keys[0], sl.d_ids[0]);
    size_t tb = sl.sorttmp_bytes;
    HIP_OK(hipcub::DeviceRadixSort::SortPairs(sl.d_sorttmp, tb, sl.d_keys[0], sl.d_keys[1], sl.d_ids[0], sl.d_ids[1], (int)nrays, 0, 31, st));
    a.order = sl.d_ids[1];
  }
  a.scratch = nullptr;
  if (m->kind == 4 && staging_enabled()) { // without the buffer the kernel still runs (own-list path everywhere), only slower
    if (grid > sl.stage_blocks && !(sl.stage_failed > 0 && grid >= sl.stage_failed)) {
      if (sl.used) HIP_OK(hipEventSynchronize(sl.done));
      if (sl.d_stage) (void)hipFree(sl.d_stage);
      sl.d_stage = nullptr;
      sl.stage_blocks = 0;
      const size_t bytes = (size_t)grid * ScatteredModel::REC_CAP * ScatteredModel::REC * sizeof(double);
      if (hipMalloc(&sl.d_stage, bytes) == hipSuccess) sl.stage_blocks = grid;
      else {
        (void)hipGetLastError();
        sl.stage_failed = grid;
        scratch_refused(m, "staging records", bytes);
      }
    }
    a.scratch = grid <= sl.stage_blocks ? sl.d_stage : nullptr;
  }
  a.scratch2 = nullptr;
  if (m->kind == 4 && a.scratch != nullptr && blocks_enabled()) { // without them the kernel scans the cells for every stencil
    if (grid > sl.cand_blocks && !(sl.cand_failed > 0 && grid >= sl.cand_failed)) {
      if (sl.used) HIP_OK(hipEventSynchronize(sl.done));
      if (sl.d_blocks) (void)hipFree(sl.d_blocks);
      sl.d_blocks = nullptr;
      sl.cand_blocks = 0;
      const size_t bytes = (size_t)grid * ScatteredModel::BLOCK_DOUBLES * sizeof(double);
      if (hipMalloc(&sl.d_blocks, bytes) == hipSuccess) sl.cand_blocks = grid;
      else {
        (void)hipGetLastError();
        sl.cand_failed = grid;
        scratch_refused(m, "candidate blocks", bytes);
      }
    }
    a.scratch2 = grid <= sl.cand_blocks ? sl.d_blocks : nullptr;
  }
  const bool fixed = p->fixedstep != 0;
  const int fopt = m->cm.fld.use_tsy != 0 ? 2 : (m->cm.fld.use_igrf != 0 ? 1 : 0);
  // one instantiation per (model, integrator, field option): the dipole kernels carry none of the IGRF code, the
  // IGRF-alone kernels none of the T04 call sites
#define SRT_LAUNCH_TRACE1(MODEL, LDS, FIX)                                                                                       \
  do {                                                                                                                           \
    if (fopt == 2) hipLaunchKernelGGL((trace_kernel<MODEL, FIX, LDS, 2>), dim3((unsigned)grid), dim3(WAVE), 0, st, dm, dc, a);   \
    else if (fopt == 1) hipLaunchKernelGGL((trace_kernel<MODEL, FIX, LDS, 1>), dim3((unsigned)grid), dim3(WAVE), 0, st, dm, dc, a); \
    else hipLaunchKernelGGL((trace_kernel<MODEL, FIX, LDS, 0>), dim3((unsigned)grid), dim3(WAVE), 0, st, dm, dc, a);             \
  } while (0)
#define SRT_LAUNCH_TRACE(MODEL, LDS)                                                                                             \
  do {                                                                                                                           \
    const MODEL *dm = (const MODEL *)m->d_model;                                                                                 \
    const Common *dc = (const Common *)m->d_common;                                                                              \
    if (fixed) SRT_LAUNCH_TRACE1(MODEL, LDS, true);                                                                              \
    else SRT_LAUNCH_TRACE1(MODEL, LDS, false);                                                                                   \
  } while (0)
  if (m->kind == 1) SRT_LAUNCH_TRACE(NgoModel, false);
  else if (m->kind == 3) SRT_LAUNCH_TRACE(InterpModel, true);
  else if (m->kind == 4) SRT_LAUNCH_TRACE(ScatteredModel, true);
  else return srt_set_error(SRT_EINVAL, "model kind %d unsupported", m->kind);
#undef SRT_LAUNCH_TRACE
#undef SRT_LAUNCH_TRACE1
  HIP_OK(hipGetLastError());
  HIP_OK(hipEventRecord(lt.ev1, st));
  HIP_OK(hipEventRecord(sl.done, st));
#ifdef SRT_TRIP_TIMING
  if (getenv("SRT_TRIP_TIMING")) {
    unsigned long long h[16];
    HIP_OK(hipDeviceSynchronize());
    HIP_OK(hipMemcpyFromSymbol(h, HIP_SYMBOL(srt_trip_cycles), sizeof h));
    fprintf(stderr, "srt trip cycles:");
    for (int i = 0; i < 16; ++i) fprintf(stderr, " %llu", h[i]);
    fprintf(stderr, "\n");
    memset(h, 0, sizeof h);
    HIP_OK(hipMemcpyToSymbol(HIP_SYMBOL(srt_trip_cycles), h, sizeof h));
  }
#endif
#ifdef SRT_PHASE_TIMING
  if (m->kind == 4 && getenv("SRT_PHASE_TIMING")) {
    unsigned long long h[16];
    HIP_OK(hipDeviceSynchronize());
    HIP_OK(hipMemcpyFromSymbol(h, HIP_SYMBOL(srt_phase_cycles), sizeof h));
    fprintf(stderr, "srt phase cycles:");
    for (int i = 0; i < 16; ++i) fprintf(stderr, " %llu", h[i]);
    fprintf(stderr, "\n");
    memset(h, 0, sizeof h);
    HIP_OK(hipMemcpyToSymbol(HIP_SYMBOL(srt_phase_cycles), h, sizeof h));
    {
      unsigned long long ts[8];
      HIP_OK(hipMemcpyFromSymbol(ts, HIP_SYMBOL(srt_tier_stats), sizeof ts));
      fprintf(stderr, "srt tier stats (stencils, second-order, no centre, near sample, L bound, h bound, free point, other):");
      for (int i = 0; i < 8; ++i) fprintf(stderr, " %llu", ts[i]);
      fprintf(stderr, "\n");
      memset(ts, 0, sizeof ts);
      HIP_OK(hipMemcpyToSymbol(HIP_SYMBOL(srt_tier_stats), ts, sizeof ts));
    }
    unsigned long long ws[4];
    float ms = 0.f;
    HIP_OK(hipMemcpyFromSymbol(ws, HIP_SYMBOL(srt_wave_stats), sizeof ws));
    HIP_OK(hipEventElapsedTime(&ms, lt.ev0, lt.ev1));
    // (the timers of the eight XCDs are not aligned with each other: only a wave's own span means something)
    fprintf(stderr, "srt wave stats: grid %lld, working waves %llu, launch %.1f ms, mean span of a working wave %.4g ticks = %.1f MHz if it ran for the whole launch\n",
            grid, ws[0], ms, ws[0] ? (double)ws[1] / (double)ws[0] : 0.0, ws[0] ? (double)ws[1] / (double)ws[0] / (ms * 1e3) : 0.0);
    const unsigned long long z[4] = {0ull, 0ull, ~0ull, 0ull};
    HIP_OK(hipMemcpyToSymbol(HIP_SYMBOL(srt_wave_stats), z, sizeof z));
  }
#endif
  sl.used = true;
  lt.used = true;
  m->last_hist = m->next_hist;
  m->next_hist = (m->next_hist + 1) % srt_model::NHIST;
  return SRT_OK;
}

extern "C" int srt_launch_ms(srt_model *m, int back, float *ms) {
  if (!m || !ms) return srt_set_error(SRT_EINVAL, "null argument");
  if (back < 0 || back >= srt_model::NHIST || m->last_hist < 0) return srt_set_error(SRT_EINVAL, "no such launch");
  srt_model::LaunchTimes &lt = m->hist[(m->last_hist - back + srt_model::NHIST) % srt_model::NHIST];
  if (!lt.used) return srt_set_error(SRT_EINVAL, "no such launch");
  SRT_MODEL_SCOPE;
  int rc = ensure_model(m);
  if (rc) return rc;
  HIP_OK(hipEventSynchronize(lt.ev1));
  HIP_OK(hipEventElapsedTime(ms, lt.ev0, lt.ev1));
  return SRT_OK;
}

extern "C" int srt_last_kernel_ms(srt_model *m, float *ms) { return srt_launch_ms(m, 0, ms); }


// ---- the step after the path: hot-plasma damping along the kept rows (kernels: srt_damping.hpp) ----
extern "C" int srt_damping_device(const srt_damping_params *dp, int nspec, const double *qs, const double *ms, int32_t slots,
                                  int32_t outputper, int64_t nrays, const double *d_rows, const int32_t *d_nrows,
                                  const double *d_w0, double *d_rate, double *d_magnitude, int32_t *d_flag, void *stream) {
  if (!dp || !qs || !ms || !d_rows || !d_nrows || !d_w0 || !d_rate || !d_flag || nrays < 0)
    return srt_set_error(SRT_EINVAL, "bad argument");
  if (nspec < 1 || nspec > SRT_MAXSPEC) return srt_set_error(SRT_EINVAL, "nspec=%d unsupported", nspec);
  if (slots < 1 || outputper < 1) return srt_set_error(SRT_EINVAL, "slots and outputper must be >= 1");
  if (dp->dist != 0 && dp->dist != 1) return srt_set_error(SRT_EINVAL, "dist must be 0 (suprathermal) or 1 (Maxwell-Boltzmann)");
  if (dp->mode != 0 && dp->mode != 1) return srt_set_error(SRT_EINVAL, "mode must be 0 (spatial) or 1 (temporal)");
  if (dp->nres < 0 || dp->nres > DMP_MAXRES) return srt_set_error(SRT_EINVAL, "nres out of range (0..%d)", DMP_MAXRES);
  if (dp->dist == 1 && (!(dp->kT > 0.0) || !(dp->Ne_h >= 0.0))) return srt_set_error(SRT_EINVAL, "Maxwell-Boltzmann needs kT > 0 and Ne_h >= 0");
  if (!(dp->tol >= 0.0)) return srt_set_error(SRT_EINVAL, "tol must be >= 0");
  if (nrays == 0) return SRT_OK;
  // the work goes to the device that owns the buffers (as srt_pack_rows_device), for the duration of the call
  int dev = -1, rc;
  {
    const void *ptrs[6] = {d_rows, d_nrows, d_w0, d_rate, d_magnitude, d_flag};
    if ((rc = device_of("srt_damping_device", ptrs, 6, &dev))) return rc;
  }
  DeviceScope scope;
  if ((rc = scope.enter(dev))) return rc;
  if (nrays * (long long)slots >= (1ll << 40)) return srt_set_error(SRT_EINVAL, "too many rows");
  DampArgs a;
  a.rows = d_rows;
  a.nrows = d_nrows;
  a.w0 = d_w0;
  a.nrays = nrays;
  a.slots = slots;
  a.outputper = outputper;
  a.nspec = nspec;
  for (int s = 0; s < MAXSPEC; ++s) {
    a.q[s] = s < nspec ? qs[s] : 0.0;
    a.ms[s] = s < nspec ? ms[s] : 1.0;
  }
  a.p.dist = dp->dist;
  a.p.mode = dp->mode;
  a.p.nres = dp->nres ? dp->nres : 3;
  for (int i = 0; i < DMP_MAXRES; ++i) a.p.m[i] = dp->nres ? dp->m[i] : (i < 3 ? i - 1 : 0);
  for (int i = 0; i < a.p.nres; ++i)
    if (a.p.m[i] < -16 || a.p.m[i] > 16) return srt_set_error(SRT_EINVAL, "resonance order %d out of range", a.p.m[i]);
  a.p.Ne_h = dp->Ne_h;
  a.p.kT = dp->kT;
  a.p.tol = dp->tol > 0.0 ? dp->tol : 1e-3;
  a.p.qh = -1.60217646e-19; // const.m
  a.p.mh = 9.10938188e-31;
  a.rate = d_rate;
  a.flag = d_flag;
  hipStream_t st = (hipStream_t)stream;
  const long long total = nrays * slots;
  const long long grid = total < (1ll << 30) ? total : (1ll << 30);
  hipLaunchKernelGGL(damping_rate_kernel, dim3((unsigned)grid), dim3(WAVE), 0, st, a);
  if (d_magnitude) hipLaunchKernelGGL(damping_magnitude_kernel, dim3((unsigned)((nrays + 63) / 64)), dim3(64), 0, st, a, d_magnitude);
  HIP_OK(hipGetLastError());
  return SRT_OK;
}

extern "C" int srt_damping(const srt_damping_params *dp, int nspec, const double *qs, const double *ms, int32_t slots,
                           int32_t outputper, int64_t nrays, const double *rows, const int32_t *nrows, const double *w0,
                           double *rate, double *magnitude, int32_t *flag) {
  if (!rows || !nrows || !w0 || !rate || nrays < 0 || slots < 1) return srt_set_error(SRT_EINVAL, "bad argument");
  DeviceScope srt_iscope_;
  int rc = srt_iscope_.enter_default();
  if (rc) return rc;
  if (nrays == 0) return SRT_OK;
  const size_t nr = (size_t)nrays * slots;
  DevBuf d_rows, d_w0, d_rate, d_mag;
  int *d_nrows = nullptr, *d_flag = nullptr;
  if ((rc = upload(d_rows, rows, nr * SRT_ROW)) || (rc = upload(d_w0, w0, nrays))) return rc;
  if (d_rate.alloc(nr) || d_mag.alloc(nr)) return srt_set_error(SRT_ENOMEM, "hipMalloc failed");
  hipError_t e = hipMalloc(&d_nrows, nrays * sizeof(int));
  if (e == hipSuccess) e = hipMalloc(&d_flag, nr * sizeof(int));
  if (e == hipSuccess) e = hipMemcpy(d_nrows, nrows, nrays * sizeof(int), hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    rc = srt_damping_device(dp, nspec, qs, ms, slots, outputper, nrays, d_rows.p, d_nrows, d_w0.p, d_rate.p, d_mag.p, d_flag, nullptr);
    if (!rc) e = hipDeviceSynchronize();
    if (!rc && e == hipSuccess) e = hipMemcpy(rate, d_rate.p, nr * sizeof(double), hipMemcpyDeviceToHost);
    if (!rc && e == hipSuccess && magnitude) e = hipMemcpy(magnitude, d_mag.p, nr * sizeof(double), hipMemcpyDeviceToHost);
    if (!rc && e == hipSuccess && flag) e = hipMemcpy(flag, d_flag, nr * sizeof(int), hipMemcpyDeviceToHost);
  }
  if (d_nrows) (void)hipFree(d_nrows);
  if (d_flag) (void)hipFree(d_flag);
  if (rc) return rc;
  if (e != hipSuccess) return srt_set_error(SRT_EDEVICE, "damping: %s", hipGetErrorString(e));
  return SRT_OK;
}


// ---- packed trajectory rows (multi-GPU gather, SURVEY 8e: only the rows a ray produced travel) ----
// kept rows of a ray = rows 0, outputper, 2*outputper, .. < nrows  (raytracer_driver.f95:1197)
struct KeptRows {
  const int32_t *nrows;
  int32_t outputper, slots;
  __host__ __device__ long long operator()(long long i) const {
    const int32_t n = nrows[i];
    int32_t k = n > 0 ? (n - 1) / outputper + 1 : 0;
    return k < slots ? k : slots;
  }
};
__global__ void pack_total_kernel(KeptRows kr, long long nrays, long long *offsets) {
  offsets[nrays] = offsets[nrays - 1] + kr(nrays - 1);
}
// one wave per ray: the ray's kept rows are one contiguous run of 160-B records in both buffers
__global__ __launch_bounds__(256) void pack_rows_kernel(long long nrays, int slots, const double *__restrict__ rows,
                                                        const long long *__restrict__ offsets, double *__restrict__ packed,
                                                        long long capacity) {
  const int lane = threadIdx.x & 63;
  for (long long ray = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); ray < nrays; ray += (long long)gridDim.x * 4) {
    const long long o0 = offsets[ray], o1 = offsets[ray + 1];
    if (o1 > capacity) continue; // the caller sees offsets[nrays] > capacity and fails the call
    const long long ndbl = (o1 - o0) * SRT_ROW;
    const double2 *src = (const double2 *)(rows + (size_t)ray * slots * SRT_ROW);
    double2 *dst = (double2 *)(packed + (size_t)o0 * SRT_ROW);
    for (long long q = lane; q < ndbl / 2; q += 64) dst[q] = src[q];
  }
}

extern "C" int srt_pack_rows_device(int32_t slots, int32_t outputper, int64_t nrays, const double *d_rows,
                                    const int32_t *d_nrows, int64_t *d_offsets, double *d_packed,
                                    int64_t capacity_rows, void *stream) {
  if (!d_offsets || nrays < 0 || slots < 1 || outputper < 1 || capacity_rows < 0 ||
      (nrays > 0 && (!d_rows || !d_nrows || (!d_packed && capacity_rows > 0))))
    return srt_set_error(SRT_EINVAL, "bad argument");
  // The work goes to the device that OWNS the buffers (not to whatever device the calling thread happens to be bound to):
  // a process may drive several GPUs from one thread.  All buffers must live on one device.
  int dev = -1, rc;
  {
    const void *ptrs[4] = {d_offsets, nrays > 0 ? (const void *)d_rows : nullptr, nrays > 0 ? (const void *)d_nrows : nullptr,
                           nrays > 0 && capacity_rows > 0 ? (const void *)d_packed : nullptr};
    if ((rc = device_of("srt_pack_rows_device", ptrs, 4, &dev))) return rc;
  }
  DeviceScope scope;
  if ((rc = scope.enter(dev))) return rc;
  hipStream_t st = (hipStream_t)stream;
  if (nrays == 0) {
    HIP_OK(hipMemsetAsync(d_offsets, 0, sizeof(int64_t), st));
    return SRT_OK;
  }
  if (nrays >= (1ll << 31) - 1) return srt_set_error(SRT_EINVAL, "too many rays for one pack call");
  // offsets[0 .. nrays] = exclusive prefix sums of the kept-row counts (one extra item so the total lands in [nrays])
  static_assert(sizeof(long long) == sizeof(int64_t), "int64");
  auto counts = hipcub::TransformInputIterator<long long, KeptRows, hipcub::CountingInputIterator<long long>>(
      hipcub::CountingInputIterator<long long>(0), KeptRows{d_nrows, outputper, slots});
  size_t need = 0;
  HIP_OK(hipcub::DeviceScan::ExclusiveSum(nullptr, need, counts, (long long *)d_offsets, (int)nrays, st));
  void *tmp = nullptr;
  HIP_OK(hipMallocAsync(&tmp, need ? need : 16, st));
  hipError_t e = hipcub::DeviceScan::ExclusiveSum(tmp, need, counts, (long long *)d_offsets, (int)nrays, st);
  const hipError_t ef = hipFreeAsync(tmp, st); // (also on the error path)
  HIP_OK(e);
  HIP_OK(ef);
  // offsets[nrays] = offsets[nrays-1] + kept(nrays-1)
  hipLaunchKernelGGL(pack_total_kernel, dim3(1), dim3(1), 0, st, KeptRows{d_nrows, outputper, slots}, (long long)nrays, (long long *)d_offsets);
  if (capacity_rows > 0) {
    long long blocks = (nrays + 3) / 4;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(pack_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (long long)nrays, (int)slots, d_rows,
                       (const long long *)d_offsets, d_packed, (long long)capacity_rows);
  }
  HIP_OK(hipGetLastError());
  return SRT_OK;
}

// AoS [n][3] -> SoA [3][n]
__global__ void aos_to_soa3(const double *in, double *out, long long n) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    out[i] = in[3 * i];
    out[n + i] = in[3 * i + 1];
    out[2 * n + i] = in[3 * i + 2];
  }
}

extern "C" int srt_trace_batch(srt_model *m, const srt_params *p, int64_t nrays, const double *pos0,
                               const double *dir0, const double *w0, double *rows, int32_t *nrows,
                               int32_t *stopcond, int64_t *accepted_steps) {
  if (!m || !pos0 || !dir0 || !w0 || !rows || !nrows || !stopcond || nrays < 0) return srt_set_error(SRT_EINVAL, "bad argument");
  int rc = check_params(p);
  if (rc) return rc;
  if (nrays == 0) {
    if (accepted_steps) *accepted_steps = 0;
    return SRT_OK;
  }
  SRT_MODEL_SCOPE;
  if ((rc = ensure_model(m))) return rc;
  std::lock_guard<std::mutex> hold(m->io_lock); // the staging below belongs to the model
  const int slots = srt_rows_per_ray(p);
  const size_t nrow_d = (size_t)nrays * slots * SRT_ROW;
  double *a_pos, *a_dir, *s_pos, *s_dir, *dw, *drows;
  int32_t *d_n, *d_s;
  int64_t *d_c;
  const size_t v3 = (size_t)3 * nrays * sizeof(double);
  if ((rc = io_reserve(m, 0, v3, (void **)&a_pos)) || (rc = io_reserve(m, 1, v3, (void **)&a_dir)) || (rc = io_reserve(m, 2, v3, (void **)&s_pos)) ||
      (rc = io_reserve(m, 3, v3, (void **)&s_dir)) || (rc = io_reserve(m, 4, (size_t)nrays * sizeof(double), (void **)&dw)) ||
      (rc = io_reserve(m, 5, nrow_d * sizeof(double), (void **)&drows)) || (rc = io_reserve(m, 6, (size_t)nrays * sizeof(int32_t), (void **)&d_n)) ||
      (rc = io_reserve(m, 7, (size_t)nrays * sizeof(int32_t), (void **)&d_s)) || (rc = io_reserve(m, 8, 4 * sizeof(int64_t), (void **)&d_c)))
    return rc;
  HIP_OK(hipMemcpy(a_pos, pos0, v3, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(a_dir, dir0, v3, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(dw, w0, (size_t)nrays * sizeof(double), hipMemcpyHostToDevice));
  unsigned blocks = (unsigned)((nrays + 255) / 256);
  hipLaunchKernelGGL(aos_to_soa3, dim3(blocks), dim3(256), 0, 0, (const double *)a_pos, s_pos, (long long)nrays);
  hipLaunchKernelGGL(aos_to_soa3, dim3(blocks), dim3(256), 0, 0, (const double *)a_dir, s_dir, (long long)nrays);
  (void)hipMemsetAsync(drows, 0, nrow_d * sizeof(double), 0);
  rc = srt_trace_batch_device(m, p, nrays, s_pos, s_dir, dw, drows, d_n, d_s, d_c, nullptr);
  if (rc == SRT_OK) {
    hipError_t e = hipMemcpy(rows, drows, nrow_d * sizeof(double), hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(nrows, d_n, nrays * sizeof(int32_t), hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(stopcond, d_s, nrays * sizeof(int32_t), hipMemcpyDeviceToHost);
    int64_t c[4] = {0, 0, 0, 0};
    if (e == hipSuccess) e = hipMemcpy(c, d_c, sizeof c, hipMemcpyDeviceToHost);
    if (e != hipSuccess) rc = srt_set_error(SRT_EDEVICE, "trace kernel failed: %s", hipGetErrorString(e));
    else if (accepted_steps) *accepted_steps = c[1];
  }
  return rc;
}
