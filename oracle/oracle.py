"""ctypes binding of the CPU oracle (oracle/libsrt_oracle.so) -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product package stanford_raytracer_amd never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libsrt_oracle.so")
SRCS = ["srt_oracle.c", "srt_oracle_scattered.c", "srt_oracle_sampler.c", "srt_oracle_damping.c", "srt_oracle_igrf.c"]
ROW = 20


def build(force=False):
    srcs = [os.path.join(HERE, s) for s in SRCS]
    deps = srcs + [os.path.join(HERE, h) for h in ("srt_oracle.h", "srt_oracle_internal.h", "tricubic_matrix.h")]
    if not force and os.path.exists(LIB) and all(os.path.getmtime(d) <= os.path.getmtime(LIB) for d in deps):
        return LIB
    cmd = ["gcc", "-O2", "-ffp-contract=off", "-fno-fast-math", "-shared", "-fPIC", "-o", LIB, *srcs, "-lm", "-lpthread"]
    subprocess.check_call(cmd)
    return LIB


class DampingParams(C.Structure):
    _fields_ = [("dist", C.c_int), ("mode", C.c_int), ("nres", C.c_int), ("m", C.c_int * 8), ("Ne_h", C.c_double),
                ("kT", C.c_double), ("tol", C.c_double)]


class Params(C.Structure):
    _fields_ = [("dt0", C.c_double), ("dtmax", C.c_double), ("tmax", C.c_double), ("maxerr", C.c_double),
                ("minalt", C.c_double), ("del_", C.c_double), ("maxsteps", C.c_int), ("root", C.c_int),
                ("fixedstep", C.c_int), ("first_attempt_policy", C.c_int)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB)
        dp = C.POINTER(C.c_double)
        ip = C.POINTER(C.c_int)
        L.so_model_create_ngo.restype = C.c_void_p
        L.so_model_create_ngo.argtypes = [C.c_char_p, C.c_int, C.c_int]
        L.so_model_create_interp_file.restype = C.c_void_p
        L.so_model_create_interp_file.argtypes = [C.c_char_p, C.c_int, C.c_int]
        L.so_model_create_interp.restype = C.c_void_p
        L.so_model_create_interp.argtypes = [C.c_int] * 4 + [dp, dp, dp, dp, C.c_int, C.c_int]
        L.so_model_create_scattered_file.restype = C.c_void_p
        L.so_model_create_scattered_file.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int,
                                                     C.c_double, C.c_uint]
        L.so_model_destroy.argtypes = [C.c_void_p]
        L.so_model_set_igrf.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_char_p]
        L.so_model_nspec.argtypes = [C.c_void_p]
        L.so_scattered_set_spacing.argtypes = [C.c_void_p, dp, C.c_double]
        L.so_scattered_radius.restype = C.c_double
        L.so_scattered_radius.argtypes = [C.c_void_p]
        L.so_plasma_params.argtypes = [C.c_void_p, dp, dp, dp, dp, dp, dp]
        L.so_dispersion_relation.restype = C.c_double
        L.so_dispersion_relation.argtypes = [dp, C.c_double, C.c_int, dp, dp, dp, dp]
        L.so_stix_parameters.argtypes = [C.c_double, C.c_int, dp, dp, dp, C.c_double, dp, dp, dp, dp, dp]
        L.so_is_right_handed.argtypes = [C.c_double] * 5
        L.so_solve_dispersion_relation.argtypes = [C.c_void_p, dp, C.c_double, dp, dp, dp]
        L.so_dfdk.argtypes = [C.c_void_p, dp, C.c_double, dp, C.c_double, dp]
        L.so_dfdw.restype = C.c_double
        L.so_dfdw.argtypes = [C.c_void_p, dp, C.c_double, dp, C.c_double]
        L.so_dfdx.argtypes = [C.c_void_p, dp, C.c_double, dp, C.c_double, dp]
        L.so_evalrhs.argtypes = [C.c_void_p, dp, C.c_double, dp]
        L.so_rk4.argtypes = [C.c_void_p, dp, C.c_double, C.c_double, dp]
        L.so_rk45.argtypes = [C.c_void_p, dp, C.c_double, C.c_double, dp, dp]
        L.so_raytracer_run.argtypes = [C.c_void_p, C.POINTER(Params), dp, dp, C.c_double, dp, C.c_int, ip, ip]
        L.so_trace_batch.restype = C.c_long
        L.so_trace_batch.argtypes = [C.c_void_p, C.POINTER(Params), C.c_long, dp, dp, dp, dp, C.c_int, ip, ip,
                                     C.c_int]
        lp = C.POINTER(C.c_long)
        L.so_build_samples.restype = C.c_void_p
        L.so_build_samples.argtypes = [C.c_void_p, dp, lp, C.c_double, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_long, dp,
                                       lp, lp]
        L.so_free.argtypes = [C.c_void_p]
        L.sod_damping.argtypes = [C.POINTER(DampingParams), C.c_int, dp, dp, C.c_int, C.c_int, C.c_long, dp, ip, dp, dp, dp, ip]
        L.sod_quadva_test.restype = C.c_double
        L.sod_quadva_test.argtypes = [C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, ip, dp, ip]
        L.so_dipole_tilt.argtypes = [C.c_int, C.c_int, dp]
        L.so_speed_of_light.restype = C.c_double
        _lib = L
    return _lib


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


def _arr(v, n=None):
    a = np.ascontiguousarray(np.asarray(v, dtype=np.float64))
    if n is not None:
        assert a.size == n
    return a


class Model:
    def __init__(self, handle):
        if not handle:
            raise RuntimeError("oracle model creation failed")
        self.h = C.c_void_p(handle)
        self.nspec = lib().so_model_nspec(self.h)

    @classmethod
    def ngo(cls, configfile, yearday=2010001, msec=0):
        return cls(lib().so_model_create_ngo(os.fsencode(configfile), yearday, msec))

    @classmethod
    def interp_file(cls, gridfile, yearday=2010001, msec=0):
        return cls(lib().so_model_create_interp_file(os.fsencode(gridfile), yearday, msec))

    @classmethod
    def interp(cls, F, bounds, qs, ms, yearday=2010001, msec=0):
        """F: array [nz, ny, nx, nspec] (C order) == file order species fastest, then x, y, z."""
        F = _arr(F)
        nz, ny, nx, nspec = F.shape
        b, q, m = _arr(bounds, 6), _arr(qs, nspec), _arr(ms, nspec)
        return cls(lib().so_model_create_interp(nspec, nx, ny, nz, _dp(b), _dp(q), _dp(m), _dp(F), yearday, msec))

    @classmethod
    def scattered_file(cls, ptsfile, yearday=2010001, msec=0, window_scale=1.5, order=2, exact=0,
                       local_window_scale=5.0, perm_seed=1):
        return cls(lib().so_model_create_scattered_file(os.fsencode(ptsfile), yearday, msec, window_scale, order,
                                                        exact, local_window_scale, perm_seed))

    def set_spacing(self, point, value):
        """Scattered model: overwrite the stored nearest-sample distance of the sample at exactly `point`."""
        p = np.ascontiguousarray(point, dtype=np.float64)
        i = lib().so_scattered_set_spacing(self.h, p.ctypes.data_as(C.POINTER(C.c_double)), float(value))
        if i < 0:
            raise ValueError("no sample at %r" % (point,))
        return i

    def search_radius(self):
        return lib().so_scattered_radius(self.h)

    def set_igrf(self, yearday=2010001, msec=0, coeff_file=None):
        """use_igrf = 1 for this model."""
        path = coeff_file or os.path.join(os.path.dirname(HERE), "stanford_raytracer_amd", "data", "igrf_coeffs.txt")
        if lib().so_model_set_igrf(self.h, yearday, msec, os.fsencode(path)) != 0:
            raise RuntimeError("IGRF coefficient table unreadable: %s" % path)
        return self

    def __del__(self):
        try:
            lib().so_model_destroy(self.h)
        except Exception:
            pass

    # ---- layers
    def build_samples(self, bounds, n_zero_altitude=0, n_iri_pad=0, n_initial_radial=0, n_initial_uniform=0,
                      adaptive_nmax=0, initial_tol=1.0, max_recursion=20, numincrease=5, max_passes=0, seed=0,
                      input_points=None):
        """Depth-first restatement of the reference's sample-set builder (srt_oracle_sampler.c)."""
        L = lib()
        w = 3 + self.nspec
        counts = (C.c_long * 5)(n_zero_altitude, n_iri_pad, n_initial_radial, n_initial_uniform, adaptive_nmax)
        n_in, inp = 0, None
        if input_points is not None:
            inp = _arr(input_points).reshape(-1, w)
            n_in = inp.shape[0]
        n_out = C.c_long()
        sc = (C.c_long * 6)()
        p = L.so_build_samples(self.h, _dp(_arr(bounds, 6)), counts, initial_tol, max_recursion, numincrease, max_passes,
                               seed, n_in, _dp(inp) if n_in else None, C.byref(n_out), sc)
        try:
            a = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_double)), shape=(n_out.value, w)).copy() if n_out.value \
                else np.empty((0, w))
        finally:
            L.so_free(p)
        return a, list(sc)

    def plasma_params(self, x):
        x = _arr(x, 3)
        qs, Ns, ms, nus, B0 = (np.zeros(4), np.zeros(4), np.zeros(4), np.zeros(4), np.zeros(3))
        lib().so_plasma_params(self.h, _dp(x), _dp(qs), _dp(Ns), _dp(ms), _dp(nus), _dp(B0))
        return qs, Ns, ms, nus, B0

    def disp(self, x, k, w):
        """Returns [F, S, D, P, R, L, re k1, im k1, re k2, im k2] like ref_harness --mode=disp."""
        x, k = _arr(x, 3), _arr(k, 3)
        qs, Ns, ms, nus, B0 = self.plasma_params(x)
        c = lib().so_speed_of_light()
        n = _arr(k * c / w)
        F = lib().so_dispersion_relation(_dp(n), w, self.nspec, _dp(qs), _dp(Ns), _dp(ms), _dp(B0))
        o = [C.c_double() for _ in range(5)]
        # sqrt(dot_product(B0,B0)) with the Fortran's association
        b0mag = float(np.sqrt((B0[0] * B0[0] + B0[1] * B0[1]) + B0[2] * B0[2]))
        lib().so_stix_parameters(w, self.nspec, _dp(qs), _dp(Ns), _dp(ms), b0mag, *[C.byref(v) for v in o])
        k1, k2 = np.zeros(2), np.zeros(2)
        lib().so_solve_dispersion_relation(self.h, _dp(k), w, _dp(x), _dp(k1), _dp(k2))
        return np.array([F] + [v.value for v in o] + [k1[0], k1[1], k2[0], k2[1]])

    def grad(self, x, k, w, del_):
        """[dFdk(3), dFdw, dFdx(3), rhs(7)] like ref_harness --mode=grad."""
        x, k = _arr(x, 3), _arr(k, 3)
        dk, dx, rhs = np.zeros(3), np.zeros(3), np.zeros(7)
        lib().so_dfdk(self.h, _dp(k), w, _dp(x), 1.0e-8, _dp(dk))
        dw = lib().so_dfdw(self.h, _dp(k), w, _dp(x), 1.0e-8)
        lib().so_dfdx(self.h, _dp(k), w, _dp(x), del_, _dp(dx))
        args = _arr(np.concatenate([x, k, [w]]))
        lib().so_evalrhs(self.h, _dp(args), del_, _dp(rhs))
        return np.concatenate([dk, [dw], dx, rhs])

    def step(self, args, dt, del_):
        """[rk4(7), rk45 4th(7), rk45 5th(7)] like ref_harness --mode=step."""
        args = _arr(args, 7)
        r4, o4, o5 = np.zeros(7), np.zeros(7), np.zeros(7)
        lib().so_rk4(self.h, _dp(args), del_, dt, _dp(r4))
        lib().so_rk45(self.h, _dp(args), del_, dt, _dp(o4), _dp(o5))
        return np.concatenate([r4, o4, o5])

    def trace(self, pos0, dir0, w0, capacity=None, nthreads=1, **kw):
        """Batch raytracer_run.  Returns (rows[nrays, capacity, 20], nrows[nrays], stopcond[nrays], steps)."""
        p = make_params(**kw)
        pos0 = _arr(pos0).reshape(-1, 3)
        dir0 = _arr(dir0).reshape(-1, 3)
        w0 = _arr(w0).reshape(-1)
        n = pos0.shape[0]
        if capacity is None:
            capacity = p.maxsteps
        rows = np.zeros((n, capacity, ROW)) if capacity > 0 else np.zeros((0,))
        nrows = np.zeros(n, dtype=np.int32)
        stop = np.zeros(n, dtype=np.int32)
        steps = lib().so_trace_batch(self.h, C.byref(p), n, _dp(pos0), _dp(dir0), _dp(w0),
                                     _dp(rows) if capacity > 0 else None, capacity, _ip(nrows), _ip(stop), nthreads)
        return rows, nrows, stop, steps


def make_params(dt0=1e-3, dtmax=0.1, tmax=1.0, maxerr=5e-4, minalt=6.4712e6, del_=1e-6, maxsteps=2000, root=2,
                fixedstep=0, first_attempt_policy=0):
    return Params(dt0, dtmax, tmax, maxerr, minalt, del_, maxsteps, root, fixedstep, first_attempt_policy)


def is_right_handed(n2, phi, S, D, P):
    return bool(lib().so_is_right_handed(n2, phi, S, D, P))


def dipole_tilt(yearday, msec):
    v = C.c_double()
    lib().so_dipole_tilt(yearday, msec, C.byref(v))
    return v.value


def damping_params(dist=0, mode=0, m=(), Ne_h=0.0, kT=0.0, tol=0.0):
    p = DampingParams()
    p.dist, p.mode, p.nres, p.Ne_h, p.kT, p.tol = dist, mode, len(m), Ne_h, kT, tol
    for i, v in enumerate(m):
        p.m[i] = int(v)
    return p


def damping(qs, ms, outputper, rows, nrows, w0, **kw):
    """matlab/damping restated (srt_oracle_damping.c): rows [nrays, slots, 20] -> rate, magnitude, flag [nrays, slots]."""
    rows = _arr(rows)
    nrays, slots, _ = rows.shape
    nrows = np.ascontiguousarray(nrows, dtype=np.int32)
    qs, ms, w0 = _arr(qs), _arr(ms), _arr(w0, nrays)
    rate, mag = np.zeros((nrays, slots)), np.zeros((nrays, slots))
    flag = np.zeros((nrays, slots), dtype=np.int32)
    p = damping_params(**kw)
    lib().sod_damping(C.byref(p), qs.size, _dp(qs), _dp(ms), slots, outputper, nrays, _dp(rows), _ip(nrows), _dp(w0),
                      _dp(rate), _dp(mag), _ip(flag))
    return rate, mag, flag


def quadva_test(kind, a, b, reltol, abstol):
    ok, nev, err = C.c_int(), C.c_int(), C.c_double()
    v = lib().sod_quadva_test(kind, a, b, reltol, abstol, C.byref(ok), C.byref(err), C.byref(nev))
    return v, bool(ok.value), err.value, nev.value
