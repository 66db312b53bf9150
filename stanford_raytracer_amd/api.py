"""Host-side mirror of the reference's interface for the hot path, over the C ABI (include/srt.h).

Names follow the reference: a *model* is what `--modelnum` selects and `setup()` builds
(raytracer_driver.f95:256-770); `plasma_params` is `funcPlasmaParams` (raytracer.f95:121-129);
`trace` is the driver's ray loop around `raytracer_run` (raytracer_driver.f95:1144-1232).
The library is HIP-only: importing works anywhere, but every call needs an MI355X and raises
`SrtError` otherwise -- there is no CPU fallback.
"""
import ctypes as C
import os

import numpy as np

from . import build as _build

ROW = 20
STOP_NAMES = {0: "tmax", 1: "minalt", 2: "k=0", 3: "vgroup", 5: "dt", 6: "maxsteps", 9: "numeric"}


class SrtError(RuntimeError):
    pass


class Params(C.Structure):
    """srt_params: scalar arguments of raytracer_run + the driver's outputper."""
    _fields_ = [("dt0", C.c_double), ("dtmax", C.c_double), ("tmax", C.c_double), ("maxerr", C.c_double),
                ("minalt", C.c_double), ("del_", C.c_double), ("maxsteps", C.c_int32), ("root", C.c_int32),
                ("fixedstep", C.c_int32), ("outputper", C.c_int32), ("first_attempt_policy", C.c_int32),
                ("refill_threshold", C.c_int32), ("ray_order", C.c_int32)]


def make_params(dt0=1e-3, dtmax=0.1, tmax=1.0, maxerr=5e-4, minalt=6371.2e3 + 100e3, del_=1e-6, maxsteps=2000,
                root=2, fixedstep=0, outputper=1, first_attempt_policy=0, refill_threshold=0, ray_order=0):
    return Params(dt0, dtmax, tmax, maxerr, minalt, del_, maxsteps, root, fixedstep, outputper,
                  first_attempt_policy, refill_threshold, ray_order)


_lib = None
dp = C.POINTER(C.c_double)
ip = C.POINTER(C.c_int32)


class SamplerParams(C.Structure):
    """srt_sampler_params of include/srt.h (the flags of gcpm_dens_model_buildgrid_random.f95:56-90)."""
    _fields_ = [("bounds", C.c_double * 6), ("n_zero_altitude", C.c_int64), ("n_iri_pad", C.c_int64),
                ("n_initial_radial", C.c_int64), ("n_initial_uniform", C.c_int64), ("adaptive_nmax", C.c_int64),
                ("initial_tol", C.c_double), ("max_recursion", C.c_int32), ("numincrease", C.c_int32),
                ("max_passes", C.c_int32), ("reserved", C.c_int32), ("seed", C.c_uint64)]


class DampingParams(C.Structure):
    """srt_damping_params of include/srt.h."""
    _fields_ = [("dist", C.c_int32), ("mode", C.c_int32), ("nres", C.c_int32), ("m", C.c_int32 * 8),
                ("Ne_h", C.c_double), ("kT", C.c_double), ("tol", C.c_double)]


def library_path():
    return _build.LIB


def lib():
    """Load libsrt_hip.so (building it if the sources are newer).  Fails loudly when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("SRT_LIB_OVERRIDE") or _build.LIB  # override: A/B builds of the same sources (tools/ab.sh)
    if not os.path.exists(path):
        _build.build()
    if not os.path.exists(path):
        raise SrtError("libsrt_hip.so is missing and could not be built; the HIP path is the only path")
    L = C.CDLL(path)
    vp = C.c_void_p
    L.srt_last_error.restype = C.c_char_p
    L.srt_init.argtypes = [C.c_int]
    L.srt_device_info.argtypes = [C.c_char_p, C.c_int, ip, C.POINTER(C.c_int64)]
    L.srt_model_create_ngo.argtypes = [C.c_char_p, C.c_int, C.c_int, C.POINTER(vp)]
    L.srt_model_create_interp_file.argtypes = [C.c_char_p, C.c_int, C.c_int, C.POINTER(vp)]
    L.srt_model_create_interp.argtypes = [C.c_int] * 4 + [dp, dp, dp, dp, C.POINTER(dp), C.c_int, C.c_int,
                                                          C.POINTER(vp)]
    L.srt_model_create_scattered_file.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int,
                                                  C.c_double, C.POINTER(vp)]
    L.srt_model_create_scattered_file_root.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int,
                                                       C.c_double, C.c_int64, C.POINTER(vp)]
    L.srt_model_set_field.argtypes = [vp, C.c_int, C.c_int, C.c_char_p]
    L.srt_model_set_tsyganenko_params.argtypes = [vp, dp]
    L.srt_model_destroy.argtypes = [vp]
    L.srt_model_destroy.restype = None
    L.srt_model_trim.argtypes = [vp]
    L.srt_model_kind.argtypes = [vp]
    L.srt_model_nspec.argtypes = [vp]
    L.srt_model_species.argtypes = [vp, dp, dp]
    L.srt_model_device_bytes.argtypes = [vp]
    L.srt_model_device_bytes.restype = C.c_int64
    L.srt_plasma_params.argtypes = [vp, C.c_int64, dp, dp, dp, dp, dp, dp]
    L.srt_dispersion.argtypes = [vp, C.c_int64, dp, dp, dp, dp]
    L.srt_is_right_handed.argtypes = [C.c_int64, dp, ip]
    L.srt_gradients.argtypes = [vp, C.c_int64, dp, dp, dp, C.c_double, dp]
    L.srt_rk_step.argtypes = [vp, C.c_int64, dp, dp, C.c_double, dp]
    L.srt_rows_per_ray.argtypes = [C.POINTER(Params)]
    L.srt_rows_per_ray.restype = C.c_int32
    L.srt_trace_batch.argtypes = [vp, C.POINTER(Params), C.c_int64, dp, dp, dp, dp, ip, ip, C.POINTER(C.c_int64)]
    L.srt_trace_batch_device.argtypes = [vp, C.POINTER(Params), C.c_int64, vp, vp, vp, vp, vp, vp, vp, vp]
    L.srt_pack_rows_device.argtypes = [C.c_int32, C.c_int32, C.c_int64, vp, vp, vp, vp, C.c_int64, vp]
    L.srt_last_kernel_ms.argtypes = [vp, C.POINTER(C.c_float)]
    L.srt_launch_ms.argtypes = [vp, C.c_int, C.POINTER(C.c_float)]
    L.srt_read_rays_file.argtypes = [C.c_char_p, C.POINTER(dp), C.POINTER(dp), C.POINTER(dp)]
    L.srt_read_rays_file.restype = C.c_int64
    L.srt_write_ray_file.argtypes = [C.c_char_p, C.c_int, C.c_int64, C.c_int64, C.POINTER(Params), C.c_int, dp, dp,
                                     dp, dp, ip, ip]
    L.srt_read_ray_file.argtypes = [C.c_char_p, ip, dp, dp, C.POINTER(C.c_int64), C.POINTER(C.POINTER(C.c_int64)), C.POINTER(ip),
                                    C.POINTER(ip), C.POINTER(dp), C.POINTER(dp)]
    L.srt_read_ray_file.restype = C.c_int64
    L.srt_free.argtypes = [vp]
    L.srt_free.restype = None
    L.srt_build_grid.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, dp, dp, dp]
    L.srt_model_create_interp_from_model.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, dp, C.c_int, C.c_int,
                                                     C.POINTER(vp)]
    L.srt_build_samples.argtypes = [vp, C.POINTER(SamplerParams), C.c_int64, dp, C.POINTER(C.c_int64), C.POINTER(dp),
                                    C.POINTER(C.c_int64)]
    L.srt_damping.argtypes = [C.POINTER(DampingParams), C.c_int, dp, dp, C.c_int32, C.c_int32, C.c_int64, dp, ip, dp, dp,
                              dp, ip]
    L.srt_damping_device.argtypes = [C.POINTER(DampingParams), C.c_int, dp, dp, C.c_int32, C.c_int32, C.c_int64, vp, vp,
                                     vp, vp, vp, vp, vp]
    i32p = C.POINTER(C.c_int32)
    L.srt_grid_file_read.argtypes = [C.c_char_p, i32p, dp, dp, dp, C.POINTER(dp), C.POINTER(dp)]
    L.srt_grid_file_write.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, dp, dp, dp, dp, dp]
    L.srt_grid_file_convert.argtypes = [C.c_char_p, C.c_char_p]
    L.srt_grid_file_is_binary.argtypes = [C.c_char_p]
    L.srt_points_file_write.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int64, dp, dp, dp, dp]
    L.srt_points_file_convert.argtypes = [C.c_char_p, C.c_char_p]
    L.srt_points_file_is_binary.argtypes = [C.c_char_p]
    _lib = L
    return L


# error codes of include/srt.h
SRT_OK, SRT_EINVAL, SRT_EIO, SRT_EDEVICE, SRT_ENOMEM = 0, -1, -2, -3, -4


def _check(rc):
    if rc != 0:
        raise SrtError("srt error %d: %s" % (rc, (lib().srt_last_error() or b"").decode()))


def _f64(a, shape=None):
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float64))
    if shape is not None:
        a = a.reshape(shape)
    return a


def _dp(a):
    return a.ctypes.data_as(dp)


def init(device=0):
    _check(lib().srt_init(device))


def device_info():
    name = C.create_string_buffer(256)
    cu = C.c_int32()
    mem = C.c_int64()
    _check(lib().srt_device_info(name, 256, C.byref(cu), C.byref(mem)))
    return {"name": name.value.decode(), "cu_count": cu.value, "hbm_bytes": mem.value}


class Model:
    """Opaque density/field model living on the device (replaces the adapters' state blob)."""

    def __init__(self, handle):
        self.h = handle
        self.kind = lib().srt_model_kind(handle)
        self.nspec = lib().srt_model_nspec(handle)

    @classmethod
    def ngo(cls, configfile, yearday=2010001, msec=0):
        h = C.c_void_p()
        _check(lib().srt_model_create_ngo(os.fsencode(configfile), yearday, msec, C.byref(h)))
        return cls(h)

    @classmethod
    def interp_file(cls, gridfile, yearday=2010001, msec=0):
        h = C.c_void_p()
        _check(lib().srt_model_create_interp_file(os.fsencode(gridfile), yearday, msec, C.byref(h)))
        return cls(h)

    @classmethod
    def interp(cls, F, bounds, qs, ms, derivs=None, yearday=2010001, msec=0):
        """F[nz, ny, nx, nspec] = ln N_s in the grid file's order (species fastest, then x, y, z)."""
        F = _f64(F)
        nz, ny, nx, ns = F.shape
        b, q, m = _f64(bounds, 6), _f64(qs, ns), _f64(ms, ns)
        dptr = None
        keep = []
        if derivs is not None:
            keep = [_f64(d, F.shape) for d in derivs]
            dptr = (dp * 7)(*[_dp(d) for d in keep])
        h = C.c_void_p()
        _check(lib().srt_model_create_interp(ns, nx, ny, nz, _dp(b), _dp(q), _dp(m), _dp(F), dptr, yearday, msec,
                                             C.byref(h)))
        return cls(h)

    @classmethod
    def scattered_file(cls, ptsfile, yearday=2010001, msec=0, window_scale=1.5, order=2, exact=0,
                       local_window_scale=5.0, root_sample=-1):
        """root_sample: 0-based record number of the sample at the root of the REFERENCE's kd-tree (its stored spacing stays
        0 there, include/srt.h srt_model_create_scattered_file_root); -1 = the true distance for every sample."""
        h = C.c_void_p()
        _check(lib().srt_model_create_scattered_file_root(os.fsencode(ptsfile), yearday, msec, window_scale, order,
                                                          exact, local_window_scale, int(root_sample), C.byref(h)))
        return cls(h)

    def build_grid(self, nx, ny, nz, bounds, compder=False):
        """Sample this model on a regular grid in log space on the device (the reference's grid builder with this
        model in place of GCPM).  -> F[nz,ny,nx,nspec], derivs (list of 7 arrays or None)."""
        ns = self.nspec
        F = np.empty((nz, ny, nx, ns))
        D = np.empty((7, nz, ny, nx, ns)) if compder else None
        _check(lib().srt_build_grid(self.h, int(bool(compder)), nx, ny, nz, _dp(_f64(bounds, (6,))), _dp(F),
                                    _dp(D) if compder else None))
        return F, (list(D) if compder else None)

    def build_samples(self, bounds, n_zero_altitude=0, n_iri_pad=0, n_initial_radial=0, n_initial_uniform=0,
                      adaptive_nmax=0, initial_tol=1.0, max_recursion=20, numincrease=5, max_passes=0, seed=0,
                      input_points=None):
        """The reference's random / adaptive sample-set builder with this model in place of GCPM, on the device.
        -> (samples[n, 3+nspec] = the lines "x y z lnN_1.." of a model-4 file, stage_counts[6])."""
        sp = SamplerParams()
        sp.bounds[:] = [float(b) for b in bounds]
        sp.n_zero_altitude, sp.n_iri_pad = int(n_zero_altitude), int(n_iri_pad)
        sp.n_initial_radial, sp.n_initial_uniform = int(n_initial_radial), int(n_initial_uniform)
        sp.adaptive_nmax, sp.initial_tol, sp.max_recursion = int(adaptive_nmax), float(initial_tol), int(max_recursion)
        sp.numincrease, sp.max_passes, sp.seed = int(numincrease), int(max_passes), int(seed)
        w = 3 + self.nspec
        n_in, inp = 0, None
        if input_points is not None:
            inp = _f64(input_points).reshape(-1, w)
            n_in = inp.shape[0]
        n_out = C.c_int64()
        out = dp()
        counts = (C.c_int64 * 6)()
        _check(lib().srt_build_samples(self.h, C.byref(sp), n_in, _dp(inp) if n_in else None, C.byref(n_out),
                                       C.byref(out), counts))
        try:
            res = np.ctypeslib.as_array(out, shape=(n_out.value, w)).copy() if n_out.value else np.empty((0, w))
        finally:
            lib().srt_free(out)
        return res, list(counts)

    def to_interp(self, nx, ny, nz, bounds, compder=False, yearday=2010001, msec=0):
        """A modelnum=3 model tabulating this one, built without leaving the device."""
        h = C.c_void_p()
        _check(lib().srt_model_create_interp_from_model(self.h, int(bool(compder)), nx, ny, nz,
                                                        _dp(_f64(bounds, (6,))), yearday, msec, C.byref(h)))
        return Model(h)

    def set_field(self, use_igrf=0, use_tsyganenko=0, igrf_coeff_file=None, parmod=None):
        """--use_igrf / --use_tsyganenko of the driver; parmod = Pdyn, Dst, ByIMF, BzIMF, W1..W6 (--tsyganenko_*)."""
        if parmod is not None:
            _check(lib().srt_model_set_tsyganenko_params(self.h, _dp(_f64(parmod, (10,)))))
        _check(lib().srt_model_set_field(self.h, int(use_igrf), int(use_tsyganenko),
                                         os.fsencode(igrf_coeff_file) if igrf_coeff_file else None))
        return self

    def trim(self):
        """Give back the grow-only device scratch (host-buffer staging, per-launch buffers); the tables stay."""
        _check(lib().srt_model_trim(self.h))
        return self

    def close(self):
        if self.h:
            lib().srt_model_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def device_bytes(self):
        return lib().srt_model_device_bytes(self.h)

    def species(self):
        qs, ms = np.zeros(4), np.zeros(4)
        _check(lib().srt_model_species(self.h, _dp(qs), _dp(ms)))
        return qs, ms

    # ---- layered entry points
    def plasma_params(self, x):
        """funcPlasmaParams at x[n,3] -> [n,19] = qs(4) Ns(4) ms(4) nus(4) B0(3)."""
        x = _f64(x, (-1, 3))
        n = x.shape[0]
        qs, Ns, ms, nus, B0 = (np.zeros((n, 4)) for _ in range(4)) if False else \
            (np.zeros((n, 4)), np.zeros((n, 4)), np.zeros((n, 4)), np.zeros((n, 4)), np.zeros((n, 3)))
        _check(lib().srt_plasma_params(self.h, n, _dp(x), _dp(qs), _dp(Ns), _dp(ms), _dp(nus), _dp(B0)))
        return np.concatenate([qs, Ns, ms, nus, B0], axis=1)

    def dispersion(self, x, k, w):
        x, k, w = _f64(x, (-1, 3)), _f64(k, (-1, 3)), _f64(w, (-1,))
        out = np.zeros((x.shape[0], 10))
        _check(lib().srt_dispersion(self.h, x.shape[0], _dp(x), _dp(k), _dp(w), _dp(out)))
        return out

    def gradients(self, x, k, w, del_):
        x, k, w = _f64(x, (-1, 3)), _f64(k, (-1, 3)), _f64(w, (-1,))
        out = np.zeros((x.shape[0], 14))
        _check(lib().srt_gradients(self.h, x.shape[0], _dp(x), _dp(k), _dp(w), del_, _dp(out)))
        return out

    def rk_step(self, args, dt, del_):
        args = _f64(args, (-1, 7))
        dt = _f64(np.broadcast_to(dt, (args.shape[0],)))
        out = np.zeros((args.shape[0], 21))
        _check(lib().srt_rk_step(self.h, args.shape[0], _dp(args), _dp(dt), del_, _dp(out)))
        return out

    # ---- the hot path
    def trace(self, pos0, dir0, w0, params=None, **kw):
        """Batched raytracer_run.  Returns rows[n, slots, 20], nrows[n], stopcond[n], accepted_steps."""
        p = params if params is not None else make_params(**kw)
        pos0, dir0, w0 = _f64(pos0, (-1, 3)), _f64(dir0, (-1, 3)), _f64(w0, (-1,))
        n = pos0.shape[0]
        slots = lib().srt_rows_per_ray(C.byref(p))
        rows = np.zeros((n, slots, ROW))
        nrows = np.zeros(n, dtype=np.int32)
        stop = np.zeros(n, dtype=np.int32)
        steps = C.c_int64()
        _check(lib().srt_trace_batch(self.h, C.byref(p), n, _dp(pos0), _dp(dir0), _dp(w0), _dp(rows),
                                     nrows.ctypes.data_as(ip), stop.ctypes.data_as(ip), C.byref(steps)))
        return rows, nrows, stop, steps.value

    def launch_ms(self, back=0):
        ms = C.c_float()
        _check(lib().srt_launch_ms(self.h, back, C.byref(ms)))
        return float(ms.value)

    def last_kernel_ms(self):
        ms = C.c_float()
        _check(lib().srt_last_kernel_ms(self.h, C.byref(ms)))
        return ms.value


def damping_params(dist=0, mode=0, m=(), Ne_h=0.0, kT=0.0, tol=0.0):
    p = DampingParams()
    p.dist, p.mode, p.nres, p.Ne_h, p.kT, p.tol = dist, mode, len(m), Ne_h, kT, tol
    for i, v in enumerate(m):
        p.m[i] = int(v)
    return p


def damping(species, outputper, rows, nrows, w0, **kw):
    """Hot-plasma damping along the kept rows (the reference's matlab/damping post-processor, on the device).
    rows [nrays, slots, 20], nrows [nrays], w0 [nrays] as Model.trace returns them -> rate, magnitude, flag."""
    qs, ms = species
    rows = _f64(rows)
    nrays, slots, _ = rows.shape
    nrows = np.ascontiguousarray(nrows, dtype=np.int32)
    qs, ms, w0 = _f64(qs), _f64(ms), _f64(w0, (nrays,))
    rate, mag = np.zeros((nrays, slots)), np.zeros((nrays, slots))
    flag = np.zeros((nrays, slots), dtype=np.int32)
    p = damping_params(**kw)
    _check(lib().srt_damping(C.byref(p), qs.size, _dp(qs), _dp(ms), slots, outputper, nrays, _dp(rows),
                             nrows.ctypes.data_as(ip), _dp(w0), _dp(rate), _dp(mag), flag.ctypes.data_as(ip)))
    return rate, mag, flag


def is_right_handed(rows):
    """rows[n,5] = n2, phi (degrees, as the reference passes it), S, D, P -> bool[n]."""
    rows = _f64(rows, (-1, 5))
    out = np.zeros(rows.shape[0], dtype=np.int32)
    _check(lib().srt_is_right_handed(rows.shape[0], _dp(rows), out.ctypes.data_as(ip)))
    return out.astype(bool)


def read_rays_file(path):
    a, b, c = dp(), dp(), dp()
    n = lib().srt_read_rays_file(os.fsencode(path), C.byref(a), C.byref(b), C.byref(c))
    if n < 0:
        _check(int(n))
    pos0 = np.ctypeslib.as_array(a, (n, 3)).copy() if n else np.zeros((0, 3))
    dir0 = np.ctypeslib.as_array(b, (n, 3)).copy() if n else np.zeros((0, 3))
    w0 = np.ctypeslib.as_array(c, (n,)).copy() if n else np.zeros((0,))
    for ptr in (a, b, c):
        lib().srt_free(ptr)
    return pos0, dir0, w0


def write_ray_file(path, species, params, w0, rows, nrows, stopcond, raynum0=1, append=False):
    """species: a Model, or (nspec, qs, ms)."""
    if isinstance(species, Model):
        qs, ms = species.species()
        nspec = species.nspec
    else:
        nspec, qs, ms = species
        qs, ms = _f64(qs), _f64(ms)
    w0 = _f64(w0, (-1,))
    rows = _f64(rows)
    nrows = np.ascontiguousarray(nrows, dtype=np.int32)
    stopcond = np.ascontiguousarray(stopcond, dtype=np.int32)
    _check(lib().srt_write_ray_file(os.fsencode(path), int(append), raynum0, w0.shape[0], C.byref(params), nspec,
                                    _dp(qs), _dp(ms), _dp(w0), _dp(rows), nrows.ctypes.data_as(ip),
                                    stopcond.ctypes.data_as(ip)))


def read_ray_file(path):
    """A .ray file (this library's or the reference driver's) -> dict(raynum, stopcond, kept, w0 per ray; rows[nrecords, 20] =
    the kept rows of all rays back to back; nspec, qs, ms).  `padded(slots)` rebuilds the [nrays, slots, 20] layout."""
    nspec, nrec = C.c_int32(), C.c_int64()
    qs, ms = np.zeros(4), np.zeros(4)
    pr, ps, pk, pw, prow = C.POINTER(C.c_int64)(), ip(), ip(), dp(), dp()
    n = lib().srt_read_ray_file(os.fsencode(path), C.byref(nspec), _dp(qs), _dp(ms), C.byref(nrec), C.byref(pr), C.byref(ps),
                                C.byref(pk), C.byref(pw), C.byref(prow))
    if n < 0:
        _check(int(n))
    try:
        out = {"raynum": np.ctypeslib.as_array(pr, (n,)).copy() if n else np.zeros(0, dtype=np.int64),
               "stopcond": np.ctypeslib.as_array(ps, (n,)).copy() if n else np.zeros(0, dtype=np.int32),
               "kept": np.ctypeslib.as_array(pk, (n,)).copy() if n else np.zeros(0, dtype=np.int32),
               "w0": np.ctypeslib.as_array(pw, (n,)).copy() if n else np.zeros(0),
               "rows": np.ctypeslib.as_array(prow, (nrec.value, ROW)).copy() if nrec.value else np.zeros((0, ROW)),
               "nspec": nspec.value, "qs": qs[:nspec.value].copy(), "ms": ms[:nspec.value].copy()}
    finally:
        for ptr in (pr, ps, pk, pw, prow):
            lib().srt_free(ptr)
    return out


def padded_rows(ray):
    """read_ray_file's packed rows -> (rows[nrays, slots, 20], nrows[nrays]) with slots = the longest ray (outputper = 1)."""
    kept = ray["kept"]
    n, slots = len(kept), int(kept.max()) if len(kept) else 1
    rows = np.zeros((n, slots, ROW))
    off = np.concatenate([[0], np.cumsum(kept)])
    for i in range(n):
        rows[i, :kept[i]] = ray["rows"][off[i]:off[i + 1]]
    return rows, kept.astype(np.int32)


# ---- model-3 grid files: text of the reference's grid builder <-> binary side-format (host code, no GPU) -------
def read_grid_file(path):
    """-> dict(F[nz,ny,nx,nspec], bounds[6], qs, ms, derivs = None | [7] arrays of F's shape)."""
    dims = (C.c_int32 * 5)()
    bounds, qs, ms = np.zeros(6), np.zeros(4), np.zeros(4)
    pF, pD = C.POINTER(C.c_double)(), C.POINTER(C.c_double)()
    _check(lib().srt_grid_file_read(path.encode(), dims, _dp(bounds), _dp(qs), _dp(ms), C.byref(pF), C.byref(pD)))
    compder, nspec, nx, ny, nz = (int(v) for v in dims)
    n = nspec * nx * ny * nz
    try:
        F = np.ctypeslib.as_array(pF, shape=(n,)).copy().reshape(nz, ny, nx, nspec)
        derivs = None
        if compder:
            derivs = list(np.ctypeslib.as_array(pD, shape=(7 * n,)).copy().reshape(7, nz, ny, nx, nspec))
    finally:
        lib().srt_free(pF)
        if pD:
            lib().srt_free(pD)
    return {"F": F, "bounds": bounds, "qs": qs[:nspec], "ms": ms[:nspec], "derivs": derivs}


def write_grid_file(path, F, bounds, qs, ms, derivs=None, binary=False):
    F = _f64(F)
    nz, ny, nx, nspec = F.shape
    d = None
    if derivs is not None:
        d = _f64(np.stack([np.asarray(a, dtype=np.float64) for a in derivs]), (7, nz, ny, nx, nspec))
    q4, m4 = np.zeros(4), np.zeros(4)
    q4[:nspec], m4[:nspec] = qs[:nspec], ms[:nspec]
    _check(lib().srt_grid_file_write(path.encode(), int(binary), nspec, nx, ny, nz, _dp(_f64(bounds, (6,))), _dp(q4),
                                     _dp(m4), _dp(F), _dp(d) if d is not None else None))


def convert_grid_file(src, dst_binary):
    _check(lib().srt_grid_file_convert(src.encode(), dst_binary.encode()))


def grid_file_is_binary(path):
    return bool(lib().srt_grid_file_is_binary(path.encode()))


def write_points_file(path, records, bounds, qs, ms, binary=False):
    """Model-4 sample file from records[n, 3+nspec] (e.g. Model.build_samples' output): the reference builder's text
    layout, or the binary side-format."""
    rec = _f64(records)
    n, w = rec.shape
    _check(lib().srt_points_file_write(os.fsencode(path), int(bool(binary)), w - 3, n, _dp(_f64(bounds, (6,))),
                                       _dp(_f64(qs)), _dp(_f64(ms)), _dp(rec)))


def convert_points_file(src_text, dst_binary):
    _check(lib().srt_points_file_convert(os.fsencode(src_text), os.fsencode(dst_binary)))


def points_file_is_binary(path):
    return bool(lib().srt_points_file_is_binary(os.fsencode(path)))
