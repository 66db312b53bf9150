import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")

# One RK step of the interp model, GPU vs reference goldens (tests/test_gpu_parity.py::test_g3_single_steps).
# Each bar is a small multiple of what the reference itself does under a 1e-13 relative perturbation of the step's
# inputs (pinned by tests/test_oracle_golden.py::test_g3_interp_self_sensitivity).
G3_INTERP_BARS = {"pos_median": 1e-7, "pos_max": 3e-5, "k_median": 1e-4, "k_p90": 1e-3, "k_max": 1e-2,
                  "k_frac_tight": 0.15}


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs an MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session")
def golden():
    return np.load(os.path.join(GOLDEN_DIR, "golden.npz"))


@pytest.fixture(scope="session")
def grid16():
    g = np.load(os.path.join(GOLDEN_DIR, "grid16.npz"))
    return g["F"], g["bounds"], g["qs"], g["ms"]


@pytest.fixture(scope="session")
def pointsfile(tmp_path_factory):
    """Model-4 sample file written from the committed fixture (the same bytes the goldens were made with)."""
    from stanford_raytracer_amd import workloads as wl

    g = np.load(os.path.join(GOLDEN_DIR, "points5500.npz"))
    path = str(tmp_path_factory.mktemp("pts") / "points.txt")
    wl.write_points_file(path, g["pts"], g["lnN"], g["bounds"], g["qs"], g["ms"])
    return path


@pytest.fixture(scope="session")
def cfgfiles(tmp_path_factory):
    from stanford_raytracer_amd import workloads as wl

    d = tmp_path_factory.mktemp("cfg")
    out = {}
    for name, text in (("ngo", wl.NEWRAY_PLASMAPAUSE), ("ngoducts", wl.NEWRAY_DUCTS)):
        p = d / ("newray_%s.in" % name)
        p.write_text(text)
        out[name] = str(p)
    return out


@pytest.fixture(scope="session")
def oracle_models(cfgfiles, grid16):
    from oracle import oracle

    F, b, qs, ms = grid16
    return {"ngo": oracle.Model.ngo(cfgfiles["ngo"]), "ngoducts": oracle.Model.ngo(cfgfiles["ngoducts"]),
            "interp": oracle.Model.interp(F, b, qs, ms)}


@pytest.fixture(scope="session")
def gpu_models(cfgfiles, grid16, pointsfile):
    from stanford_raytracer_amd import api

    api.init(0)
    F, b, qs, ms = grid16
    return {"ngo": api.Model.ngo(cfgfiles["ngo"]), "ngoducts": api.Model.ngo(cfgfiles["ngoducts"]),
            "interp": api.Model.interp(F, b, qs, ms), "scattered": api.Model.scattered_file(pointsfile)}


@pytest.fixture(scope="session")
def oracle_scattered(pointsfile):
    from oracle import oracle

    # bit 31 of perm_seed: true nearest-sample distance for the tree root too (what the HIP path stores)
    return oracle.Model.scattered_file(pointsfile, perm_seed=2 | 0x80000000)


DELS = {"ngo": 1e-4, "ngoducts": 1e-4, "interp": 1e-6, "scattered": 1e-6}


def vrel(a, b):
    """|a-b| / |b| over the last axis (vector-relative error)."""
    a, b = np.asarray(a), np.asarray(b)
    return np.linalg.norm(a - b, axis=-1) / np.maximum(np.linalg.norm(b, axis=-1), 1e-300)


def parse_ray_file(path):
    """A .ray file in the driver's record format (raytracer_driver.f95:1197-1217, nspec = 4, 822 characters):
    -> array [records, 36] = raynum, stopcond, t, pos3, vprel3, vgrel3, n3, B03, w, nspec, qs4, ms4, Ns4, nus4."""
    rows = []
    for line in open(path):
        assert len(line.rstrip("\n")) == 10 + 10 + 17 * 24 + 10 + 16 * 24
        head = [int(line[0:10]), int(line[10:20])]
        vals = [float(line[20 + 24 * i:44 + 24 * i]) for i in range(17)]
        nspec = int(line[428:438])
        tail = [float(line[438 + 24 * i:462 + 24 * i]) for i in range(16)]
        rows.append(head + vals + [nspec] + tail)
    return np.array(rows)


def oracle_grad_sensitivity(om, x, k, w, del_, eps=(4.0e-16, 1.5e-15)):
    """Per-sample yardstick for the finite-difference gradients: what the ORACLE's own dFdk / dFdw / dFdx / right-hand side do
    when k or x move by `eps` relative (two and seven ulps) -- the level at which any other summation order differs from the reference's.
    Central differences with a 1e-8 relative step amplify that by 2^-53 / 1e-8 times the cancellation in F, which on real data
    (ionospheric densities, states near a resonance where dF/dw passes through zero) is unbounded; a fixed bar would either
    fail there or be loose everywhere.  Returns (base[n,14], yard[n,5]): yard columns = vector-relative change of dFdk, relative
    change of dFdw, vector-relative change of dFdx, of dx/dt, of dk/dt; each the largest over eight perturbations."""
    base = np.array([om.grad(a, b, c, del_) for a, b, c in zip(x, k, w)])
    yard = np.zeros((len(w), 5))
    for sx, sk in [q for e in eps for q in ((0.0, e), (0.0, -e), (e, 0.0), (-e, 0.0))]:
        g = np.array([om.grad(a * (1.0 + sx), b * (1.0 + sk), c, del_) for a, b, c in zip(x, k, w)])
        e = np.stack([vrel(g[:, 0:3], base[:, 0:3]), np.abs(g[:, 3] - base[:, 3]) / np.maximum(np.abs(base[:, 3]), 1e-300),
                      vrel(g[:, 4:7], base[:, 4:7]), vrel(g[:, 7:10], base[:, 7:10]), vrel(g[:, 10:13], base[:, 10:13])], axis=1)
        yard = np.maximum(yard, e)
    return base, yard


def grad_errors(g, ref):
    """The five error columns of oracle_grad_sensitivity for outputs g against ref."""
    return np.stack([vrel(g[:, 0:3], ref[:, 0:3]), np.abs(g[:, 3] - ref[:, 3]) / np.maximum(np.abs(ref[:, 3]), 1e-300),
                     vrel(g[:, 4:7], ref[:, 4:7]), vrel(g[:, 7:10], ref[:, 7:10]), vrel(g[:, 10:13], ref[:, 10:13])], axis=1)


def within_sensitivity(err, yard, floor, factor=10.0, outliers=0.01):
    """err[n] <= factor * max(yard[n], floor) for all but `outliers` of the samples (the yardstick is four random draws per
    sample: a handful of random draws).  Returns (ok, text)."""
    bound = factor * np.maximum(yard, floor)
    bad = err > bound
    txt = "%d / %d over their bound (worst ratio %.3g; err max %.3g, yard max %.3g, floor %.3g)" % (
        bad.sum(), len(err), float(np.max(err / bound)), err.max(), yard.max(), floor)
    return bad.mean() <= outliers, txt


def oracle_step_sensitivity(om, args, dt, del_, eps=(4.0e-16, 1.5e-15, 6.0e-15)):
    """Per-sample yardstick for one RK step (the three outputs rk4 / rkf45 4th / 5th order, 7 numbers each): what the oracle's own
    step does when the state moves by `eps` relative.  Returns (base[n,21], yard[n,3,2]): vector-relative change of position and of
    k for each of the three outputs, the largest over twelve perturbations (a step that crosses a discontinuity of the table is
    chaotic: one draw may miss what the next one shows)."""
    base = np.array([om.step(a, d, del_) for a, d in zip(args, dt)])
    yard = np.zeros((len(dt), 3, 2))
    for sx, sk in [q for e in eps for q in ((0.0, e), (0.0, -e), (e, 0.0), (-e, 0.0))]:
        a2 = np.array(args, dtype=np.float64, copy=True)
        a2[:, 0:3] *= (1.0 + sx)
        a2[:, 3:6] *= (1.0 + sk)
        g = np.array([om.step(a, d, del_) for a, d in zip(a2, dt)])
        for i, o in enumerate((0, 7, 14)):
            yard[:, i, 0] = np.maximum(yard[:, i, 0], vrel(g[:, o:o + 3], base[:, o:o + 3]))
            yard[:, i, 1] = np.maximum(yard[:, i, 1], vrel(g[:, o + 3:o + 6], base[:, o + 3:o + 6]))
    return base, yard
