"""Synthetic workloads and on-disk formats for the hot path.

Everything here is input *generation* (numpy only): seeded launch sets, the analytic plasmasphere
used to fill grids / point clouds, and writers for the reference's text formats
(model-3 grid file: gcpm_dens_model_buildgrid.f95:302-327 / interp_dens_model_adapter.f95:58-117;
model-4 scattered file: gcpm_dens_model_buildgrid_random.f95:210-225;
newray.in: ngo_dens_model.f95:45-118; ray input: raytracer_driver.f95:1146).
"""
import numpy as np

R_E = 6371.2e3
Q_E = 1.602e-19
QS = Q_E * np.array([-1.0, 1.0, 1.0, 1.0])
MS = np.array([9.10938188e-31, 1.6726e-27, 4.0 * 1.6726e-27, 16.0 * 1.6726e-27])
MINALT = R_E + 100e3

# Appendix-B style Ngo configuration: plasmapause only, Ne = 3000 cm^-3 at L=2 equator
NEWRAY_PLASMAPAUSE = """0 0 0 0.0
22000.0 0.0
-1.0 0.0
4 1 1 2 1 0 100.0 2.0 0.0 3000.0
0.0 1600.0 1000.0 1.0e-3 1.0e-3
7370.0 1.0 0.08 0.02 0.90
6460.0 140.0 6470.0 6873.0 1.0e-6
5.55 3.0 0.07 5.0e4 5.0e4
0.0 0.0 0.0 0.0 0.0 0.0 0.0 0.0
"""
# plasmapause + one sinusoidal perturbation (l0(2) <= 0) + two gaussian ducts
NEWRAY_DUCTS = """0 0 0 0.0
22000.0 10.0
-1.0 0.0
4 1 1 2 4 0 100.0 2.0 0.0 3000.0
0.0 1600.0 1000.0 1.0e-3 1.0e-3
7370.0 1.0 0.08 0.02 0.90
6460.0 140.0 6470.0 6873.0 1.0e-6
4.5 3.0 0.10 5.0e4 5.0e4
-0.5 0.2 0.4 6900.0 300.0 20000.0 2000.0 6900.0 300.0 20000.0 2000.0 1.0
2.5 0.3 0.05 7000.0 400.0 15000.0 3000.0 7000.0 400.0 15000.0 3000.0 0.0
3.5 -0.2 0.08 7000.0 400.0 25000.0 3000.0 7000.0 400.0 25000.0 3000.0 1.0
0.0 0.0 0.0 0.0 0.0 0.0 0.0 0.0
"""


def dipole_b(pos):
    """Centred dipole in SM coordinates (bmodel_dipole.f95), vectorised; used only to aim rays."""
    pos = np.asarray(pos, dtype=np.float64)
    x, y, z = pos[..., 0], pos[..., 1], pos[..., 2]
    r2 = x * x + y * y + z * z
    r = np.sqrt(r2)
    k = 0.312e-4 * R_E ** 3 / (r2 * r2 * r)
    return np.stack([-3.0 * k * x * z, -3.0 * k * y * z, k * (r2 - 3.0 * z * z)], axis=-1)


def launch_set(nrays, seed):
    """Seeded launch set of SURVEY.md 8(d): returns pos0[n,3], dir0[n,3], w0[n]."""
    rng = np.random.default_rng(seed)
    lam = np.deg2rad(rng.uniform(-50.0, 50.0, nrays))
    lon = rng.uniform(0.0, 2.0 * np.pi, nrays)
    r = R_E + rng.uniform(500e3, 4.0 * R_E, nrays)
    pos = np.stack([r * np.cos(lam) * np.cos(lon), r * np.cos(lam) * np.sin(lon), r * np.sin(lam)], axis=-1)
    b = dipole_b(pos)
    bh = b / np.linalg.norm(b, axis=-1, keepdims=True)
    # orthonormal frame around bh
    ref = np.where(np.abs(bh[:, 2:3]) < 0.9, np.array([[0.0, 0.0, 1.0]]), np.array([[1.0, 0.0, 0.0]]))
    e1 = np.cross(bh, ref)
    e1 /= np.linalg.norm(e1, axis=-1, keepdims=True)
    e2 = np.cross(bh, e1)
    ang = np.deg2rad(rng.uniform(10.0, 70.0, nrays))[:, None]
    az = rng.uniform(0.0, 2.0 * np.pi, nrays)[:, None]
    d = np.cos(ang) * bh + np.sin(ang) * (np.cos(az) * e1 + np.sin(az) * e2)
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    f = np.exp(rng.uniform(np.log(0.5e3), np.log(10e3), nrays))
    return np.ascontiguousarray(pos), np.ascontiguousarray(d), 2.0 * np.pi * f


def appendix_b_rays(n=16):
    """The 16 probe rays of SURVEY.md Appendix B (explicit, non-field-aligned directions)."""
    rows = []
    for i in range(n):
        r = R_E + 1000e3 + 50e3 * i
        lam = np.deg2rad(-30.0 + 4.0 * i)
        a = lam + np.deg2rad(20.0)
        rows.append([r * np.cos(lam), 0.0, r * np.sin(lam), np.cos(a), 1e-2, np.sin(a), 2 * np.pi * (1000 + 200 * i)])
    rows = np.array(rows)
    return rows[:, 0:3].copy(), rows[:, 3:6].copy(), rows[:, 6].copy()


def analytic_lnN(pos):
    """ln N_s (m^-3) of the smooth synthetic plasmasphere of SURVEY.md 8(d) config 3; pos[...,3] -> [...,4]."""
    pos = np.asarray(pos, dtype=np.float64)
    r = np.sqrt((pos ** 2).sum(-1))
    rr = np.maximum(r, R_E)
    sin2 = np.where(r > 0, (pos[..., 0] ** 2 + pos[..., 1] ** 2) / np.maximum(r * r, 1e-300), 1.0)
    L = rr / (R_E * np.maximum(sin2, 1e-6))
    ne = 1e9 * (R_E / rr) ** 4 * (1.0 + 0.5 * np.tanh((4.0 - L) / 0.3))
    frac = np.array([1.0, 0.9, 0.08, 0.02])
    return np.log(ne[..., None] * frac)


def make_grid(n, half_width=10.0 * R_E, dtype=np.float64):
    """Regular n^3 grid of ln N_s over +-half_width.  Returns (F[nz,ny,nx,4], bounds[6])."""
    ax = np.arange(n) * ((2.0 * half_width) / (n - 1.0)) + (-half_width)
    F = np.empty((n, n, n, 4), dtype=dtype)
    X, Y = np.meshgrid(ax, ax, indexing="xy")  # [ny, nx]
    for k in range(n):
        P = np.stack([X, Y, np.full_like(X, ax[k])], axis=-1)
        F[k] = analytic_lnN(P)
    b = np.array([-half_width, half_width] * 3)
    return F, b


def write_grid_file(path, F, bounds, qs=QS, ms=MS, derivs=None):
    """Model-3 grid file.  F[nz,ny,nx,nspec]; optional derivs = list of 7 arrays of the same shape."""
    nz, ny, nx, ns = F.shape
    with open(path, "w") as f:
        f.write("%10d%10d%10d%10d%10d\n" % (1 if derivs is not None else 0, ns, nx, ny, nz))
        f.write("".join("%24.15E" % v for v in bounds) + "\n")
        f.write("".join("%24.15E" % v for v in qs) + "\n")
        f.write("".join("%24.15E" % v for v in ms) + "\n")
        flat = F.reshape(-1, ns)
        for row in flat:
            f.write(" ".join("%.17g" % v for v in row) + "\n")
        if derivs is not None:
            for d in derivs:
                for row in np.asarray(d).reshape(-1, ns):
                    f.write(" ".join("%.17g" % v for v in row) + "\n")


def make_points(n_uniform, n_shell, seed, half_width=5.0 * R_E):
    """Scattered sample set: uniform in the cube plus a near-Earth shell.  Returns pts[n,3], lnN[n,4]."""
    rng = np.random.default_rng(seed)
    pu = rng.uniform(-half_width, half_width, (n_uniform, 3))
    v = rng.normal(size=(n_shell, 3))
    v /= np.linalg.norm(v, axis=-1, keepdims=True)
    ps = v * (R_E + rng.uniform(0.0, 2000e3, (n_shell, 1)))
    pts = np.concatenate([pu, ps], axis=0)
    return pts, analytic_lnN(pts)


def write_points_file(path, pts, lnN, bounds, qs=QS, ms=MS):
    """Model-4 scattered file: nspec + bounds; qs; ms; then x y z lnN_1..lnN_nspec per line."""
    ns = lnN.shape[1]
    with open(path, "w") as f:
        f.write("%d " % ns + " ".join("%.17g" % v for v in bounds) + "\n")
        f.write(" ".join("%.17g" % v for v in qs) + "\n")
        f.write(" ".join("%.17g" % v for v in ms) + "\n")
        for p, v in zip(pts, lnN):
            f.write(" ".join("%.17g" % x for x in np.concatenate([p, v])) + "\n")


def write_rays_file(path, pos0, dir0, w0):
    with open(path, "w") as f:
        for p, d, w in zip(pos0, dir0, w0):
            f.write(" ".join("%.17g" % v for v in (*p, *d, w)) + "\n")


def morton_order(pos, half_width, n):
    """Permutation that sorts launch points by the Morton code of their cell on an n^3 grid over +-half_width."""
    c = np.clip(((np.asarray(pos) + half_width) / (2.0 * half_width) * (n - 1)).astype(np.int64), 0, n - 1)

    def spread(v):
        v = v & 0x1FFFFF
        v = (v | (v << 32)) & 0x1F00000000FFFF
        v = (v | (v << 16)) & 0x1F0000FF0000FF
        v = (v | (v << 8)) & 0x100F00F00F00F00F
        v = (v | (v << 4)) & 0x10C30C30C30C30C3
        v = (v | (v << 2)) & 0x1249249249249249
        return v

    code = spread(c[:, 0]) | (spread(c[:, 1]) << 1) | (spread(c[:, 2]) << 2)
    return np.argsort(code, kind="stable")


def make_points_config5(seed, n_uniform=200_000, n_importance=600_000, n_shell=25_000, half_width=10.0 * R_E):
    """The sample set of SURVEY.md 8(d) config 5 (825 k samples, the size of the manual's example): uniform in the
    +-10 R_E cube, importance-sampled with density proportional to |grad ln N_e| of the same analytic plasmasphere
    (rejection from the uniform proposal), and a shell between R_E and R_E + 2000 km.  Returns pts[n,3], lnN[n,4]."""
    rng = np.random.default_rng(seed)
    pu = rng.uniform(-half_width, half_width, (n_uniform, 3))

    def gradmag(p):
        h = 1.0e3
        g = np.zeros(len(p))
        for a in range(3):
            e = np.zeros(3)
            e[a] = h
            g += ((analytic_lnN(p + e)[:, 0] - analytic_lnN(p - e)[:, 0]) / (2 * h)) ** 2
        return np.sqrt(g)

    probe = rng.uniform(-half_width, half_width, (200_000, 3))
    cap = np.percentile(gradmag(probe), 99.5)  # acceptance = min(1, |grad| / cap)
    chunks, have = [], 0
    while have < n_importance:
        c = rng.uniform(-half_width, half_width, (400_000, 3))
        keep = c[rng.uniform(0.0, 1.0, len(c)) < np.minimum(gradmag(c) / cap, 1.0)]
        chunks.append(keep)
        have += len(keep)
    pi = np.concatenate(chunks)[:n_importance]
    v = rng.normal(size=(n_shell, 3))
    v /= np.linalg.norm(v, axis=-1, keepdims=True)
    ps = v * (R_E + rng.uniform(0.0, 2000e3, (n_shell, 1)))
    pts = np.concatenate([pu, pi, ps], axis=0)
    return pts, analytic_lnN(pts)


def synthetic_derivs(shape):
    """Seven file-supplied derivative blocks (computederivatives = 1, interp_dens_model_adapter.f95:107-116) for a grid of
    `shape` = (nz, ny, nx, nspec): exactly reproducible numbers (small integers / 8 times a power of ten, no libm), of the
    size ln N's derivatives have per metre.  Order: dfdx dfdy dfdz d2fdxdy d2fdxdz d2fdydz d3fdxdydz."""
    nz, ny, nx, ns = shape
    k, j, i, s = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), np.arange(ns), indexing="ij")
    scale = [1e-7, 1e-7, 1e-7, 1e-14, 1e-14, 1e-14, 1e-21]
    return [scale[a] * (((i * 7 + j * 13 + k * 29 + s * 5 + a * 3) % 17) - 8).astype(np.float64) / 8.0 for a in range(7)]
