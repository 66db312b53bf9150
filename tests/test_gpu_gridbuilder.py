"""GPU (-m gpu): the step before the path -- a model sampled on a regular grid in log space on the
device, i.e. the reference's grid builder (gcpm_dens_model_buildgrid.f95:160-300) with an in-scope model in place of
GCPM.  Checked against the CPU oracle's funcPlasmaParams evaluated at the same nodes and stencils."""
import numpy as np
import pytest

from stanford_raytracer_amd import api, workloads as wl

pytestmark = pytest.mark.gpu
NX, NY, NZ = 9, 8, 7
BOUNDS = np.array([1.2, 4.0, -2.0, 2.5, -1.5, 1.8]) * wl.R_E


def nodes():
    x = np.arange(NX) * ((BOUNDS[1] - BOUNDS[0]) / (NX - 1.0)) + BOUNDS[0]
    y = np.arange(NY) * ((BOUNDS[3] - BOUNDS[2]) / (NY - 1.0)) + BOUNDS[2]
    z = np.arange(NZ) * ((BOUNDS[5] - BOUNDS[4]) / (NZ - 1.0)) + BOUNDS[4]
    Z, Y, X = np.meshgrid(z, y, x, indexing="ij")
    return np.stack([X, Y, Z], axis=-1)  # [nz, ny, nx, 3]


def oracle_lnN(om, P):
    flat = P.reshape(-1, 3)
    return np.log(np.array([om.plasma_params(p)[1] for p in flat])).reshape(P.shape[:-1] + (4,))


def test_ngo_grid_values_and_derivative_blocks(gpu_models, oracle_models):
    g, o = gpu_models["ngo"], oracle_models["ngo"]
    F, D = g.build_grid(NX, NY, NZ, BOUNDS, compder=True)
    P = nodes()
    # f = log(Ns) at the nodes: the same numbers the plasma-params entry point gives, and the oracle's to rounding
    mine = np.log(g.plasma_params(P.reshape(-1, 3))[:, 4:8]).reshape(F.shape)
    assert np.abs(F - mine).max() <= 2e-14        # device log vs host log of the same densities
    ref = oracle_lnN(o, P)
    assert np.abs(F - ref).max() <= 1e-12
    # the seven finite-difference blocks, restated with the oracle (d = 1e-3 |pos|, reference's stencil signs)
    d = 1e-3 * np.linalg.norm(P, axis=-1)[..., None]
    e = np.eye(3)

    def L(sx, sy, sz):
        return oracle_lnN(o, P + d * (sx * e[0] + sy * e[1] + sz * e[2]))

    want = [(L(1, 0, 0) - L(-1, 0, 0)) / d / 2, (L(0, 1, 0) - L(0, -1, 0)) / d / 2, (L(0, 0, 1) - L(0, 0, -1)) / d / 2,
            (L(1, 1, 0) - L(-1, 1, 0) - L(1, -1, 0) + L(-1, -1, 0)) / d / d / 4,
            (L(1, 0, 1) - L(-1, 0, 1) - L(1, 0, -1) + L(-1, 0, -1)) / d / d / 4,
            (L(0, 1, 1) - L(0, -1, 1) - L(0, 1, -1) + L(0, -1, -1)) / d / d / 4,
            (L(1, 1, 1) - L(-1, 1, 1) - L(1, -1, 1) + L(-1, -1, 1) - L(1, 1, -1) + L(-1, 1, -1) + L(1, -1, -1)
             - L(-1, -1, -1)) / d / d / d / 8]
    for blk, (got, w) in enumerate(zip(D, want)):
        # differences of logs that agree to 1e-13, divided by d^order (d ~ 1e4 .. 3e4 m)
        scale = np.abs(w).max()
        tol = 4e-13 / d.min() ** (1 if blk < 3 else 2 if blk < 6 else 3)
        assert np.abs(got - w).max() <= tol + 1e-9 * scale, "block %d" % blk


def test_tabulated_ngo_is_a_model3_model(gpu_models, tmp_path):
    """Ngo -> grid -> interp model, all on the device; the same grid written to a file and read back gives the same
    model; and between the nodes the table follows the analytic model (config 3 cross-checked against config 2)."""
    g = gpu_models["ngo"]
    n, b = 48, np.array([1.5, 3.5, -1.0, 1.0, -1.0, 1.0]) * wl.R_E   # inside the plasmasphere, above the ionosphere
    tab = g.to_interp(n, n, n, b)
    F, _ = g.build_grid(n, n, n, b)
    qs, ms = g.species()
    path = str(tmp_path / "ngo.bin")
    api.write_grid_file(path, F, b, qs, ms, binary=True)
    rng = np.random.default_rng(3)
    x = rng.uniform([b[0], b[2], b[4]], [b[1], b[3], b[5]], (2000, 3))
    a, c = tab.plasma_params(x), api.Model.interp_file(path).plasma_params(x)
    assert np.array_equal(a, c)
    exact = g.plasma_params(x)
    rel = np.abs(a[:, 4:8] - exact[:, 4:8]) / exact[:, 4:8]
    assert np.median(rel) <= 1e-4 and np.percentile(rel, 99) <= 2e-2
    assert np.array_equal(a[:, 16:19], exact[:, 16:19])       # the field does not come from the table


def test_grid_builder_accepts_every_model_kind(gpu_models):
    """An interp model re-tabulated on its own nodes reproduces its ln N there to rounding (exp/log round trip)."""
    m = gpu_models["interp"]
    F16 = None
    from conftest import GOLDEN_DIR
    import os
    gg = np.load(os.path.join(GOLDEN_DIR, "grid16.npz"))
    F16, b = gg["F"], gg["bounds"]
    F, D = m.build_grid(16, 16, 16, b)
    assert D is None and F.shape == F16.shape
    assert np.abs(F - F16).max() <= 1e-12 * np.abs(F16).max()
