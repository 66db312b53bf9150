"""CPU: the oracle against the reference itself on fresh seeded inputs (only where oracle/_ref exists)."""
import numpy as np
import pytest

from conftest import DELS
from oracle import refharness
from stanford_raytracer_amd import workloads as wl

pytestmark = pytest.mark.skipif(not refharness.available(), reason="oracle/_ref/ref_harness not built")


@pytest.fixture(scope="module")
def ref_models(cfgfiles, grid16, tmp_path_factory):
    F, b, qs, ms = grid16
    gf = str(tmp_path_factory.mktemp("grid") / "grid16.txt")
    wl.write_grid_file(gf, F, b, qs, ms)
    return {"ngo": {"kind": 1, "file": cfgfiles["ngo"]}, "ngoducts": {"kind": 1, "file": cfgfiles["ngoducts"]},
            "interp": {"kind": 3, "file": gf}}


@pytest.mark.parametrize("name", ["ngo", "ngoducts", "interp"])
def test_params_and_gradients_fresh(oracle_models, ref_models, name):
    m, mdl = oracle_models[name], ref_models[name]
    pos, d, w = wl.launch_set(120, 9001)
    ref = refharness.run_mode("params", pos, mdl)
    mine = np.array([np.concatenate(m.plasma_params(p)) for p in pos])
    assert np.allclose(mine, ref, rtol=1e-13, atol=0)
    od = refharness.run_mode("disp", np.concatenate([pos, d, w[:, None]], axis=1), mdl)
    ok = od[:, 8] > 0
    g_in = np.concatenate([pos[ok], od[ok, 8:9] * d[ok], w[ok, None], np.full((ok.sum(), 1), DELS[name])], axis=1)[:40]
    ref = refharness.run_mode("grad", g_in, mdl)
    mine = np.array([m.grad(r[0:3], r[3:6], r[6], r[7]) for r in g_in])
    assert np.allclose(mine, ref, rtol=1e-12, atol=0)


def test_grid_file_reader_matches_arrays(oracle_models, ref_models):
    from oracle import oracle

    m_file = oracle.Model.interp_file(ref_models["interp"]["file"])
    pos, _, _ = wl.launch_set(50, 9002)
    a = np.array([np.concatenate(m_file.plasma_params(p)) for p in pos])
    b = np.array([np.concatenate(oracle_models["interp"].plasma_params(p)) for p in pos])
    assert np.array_equal(a, b)


def test_adaptive_run_fresh(oracle_models, ref_models):
    m, mdl = oracle_models["ngo"], ref_models["ngo"]
    pos, d, w = wl.launch_set(12, 9003)
    out, _ = refharness.run_rays(mdl, np.concatenate([pos, d, w[:, None]], axis=1), fixedstep=0, dt0=1e-3, dtmax=0.1,
                                 tmax=0.05, maxerr=5e-4, maxsteps=300, minalt=wl.MINALT)
    rows, nrows, stop, _ = m.trace(pos, d, w, capacity=300, fixedstep=0, dt0=1e-3, dtmax=0.1, tmax=0.05, maxerr=5e-4,
                                   maxsteps=300, minalt=wl.MINALT, del_=1e-4)
    assert [o["rows"].shape[0] for o in out] == nrows.tolist()
    assert [o["stopcond"] for o in out] == stop.tolist()


# ---- the reference's OWN program (oracle/_ref/raytracer = fortran/raytracer_driver.f95, built by oracle/build_ref.py) ----
import os
import subprocess
import sys

from conftest import GOLDEN_DIR, ROOT, parse_ray_file

DRIVER = os.path.join(ROOT, "oracle", "_ref", "raytracer")
needs_driver = pytest.mark.skipif(not os.path.exists(DRIVER), reason="oracle/_ref/raytracer not built")


@needs_driver
def test_reference_driver_reproduces_the_committed_ray_files(tmp_path, cfgfiles, grid16):
    """The committed .ray goldens ARE what the reference's own flag parsing (raytracer_driver.f95:181-228) and record
    writer (:1197-1217) produce: rerun it here, byte for byte."""
    sys.path.insert(0, GOLDEN_DIR)
    import make_driver_golden as mk

    p0, d0, w0 = wl.appendix_b_rays()
    rays = str(tmp_path / "rays.txt")
    wl.write_rays_file(rays, p0, d0, w0)
    out1 = str(tmp_path / "c1.ray")
    subprocess.run([DRIVER] + mk.config1_flags(rays, out1, cfgfiles["ngo"]), check=True, stdout=subprocess.DEVNULL)
    assert open(out1, "rb").read() == open(os.path.join(GOLDEN_DIR, "config1_outputper25.ray"), "rb").read()
    F, b, qs, ms = grid16
    gf = str(tmp_path / "grid16.txt")
    wl.write_grid_file(gf, F, b, qs, ms)
    out3 = str(tmp_path / "c3.ray")
    subprocess.run([DRIVER] + mk.interp_flags(rays, out3, gf), check=True, stdout=subprocess.DEVNULL)
    assert open(out3, "rb").read() == open(os.path.join(GOLDEN_DIR, "driver_interp_adaptive.ray"), "rb").read()


def test_oracle_reproduces_the_drivers_adaptive_file(oracle_models):
    """The C oracle traced with the driver's parameters gives the driver's own adaptive model-3 file: same records,
    same stop codes, and the numbers to the 16 significant digits the record format holds (no reference build needed:
    the file is committed)."""
    ref = parse_ray_file(os.path.join(GOLDEN_DIR, "driver_interp_adaptive.ray"))
    p0, d0, w0 = wl.appendix_b_rays()
    rows, nrows, stop, _ = oracle_models["interp"].trace(p0, d0, w0, capacity=2000, fixedstep=0, dt0=1e-3, dtmax=0.1, tmax=0.2,
                                                         maxerr=5e-4, maxsteps=2000, minalt=wl.MINALT, del_=1e-6)
    for ray in range(16):
        r = ref[ref[:, 0] == ray + 1]
        kept = rows[ray, 0:nrows[ray]:16]
        assert len(kept) == len(r) and np.all(r[:, 1] == stop[ray])
        assert np.allclose(kept[:, 0:16], r[:, 2:18], rtol=2e-15, atol=0)     # t, pos, vprel, vgrel, n, B0
        assert np.allclose(kept[:, 16:20], r[:, 28:32], rtol=2e-15, atol=0)   # Ns


def test_scattered_weight_mask_exception_against_the_reference(tmp_path):
    """lsinterp_mod.f95:316-323 on the reference itself: query points in a void next to a densely sampled ball keep NO
    sample above the 1e-16 weight mask, so the reference "uses them all".  The oracle's restatement of that rule gives the
    reference's densities (the same scenario drives the HIP path in
    tests/test_gpu_scattered_paths.py::test_too_few_samples_above_the_weight_mask_goes_through_the_own_list_rule)."""
    import os

    from conftest import GOLDEN_DIR
    from oracle import oracle

    g0 = np.load(os.path.join(GOLDEN_DIR, "points5500.npz"))
    h0 = 3.0e4
    c0 = np.array([3.2 * wl.R_E, 0.4 * wl.R_E, 0.3 * wl.R_E])
    ax = np.arange(-6, 7) * h0
    L = np.stack(np.meshgrid(ax, ax, ax, indexing="ij"), -1).reshape(-1, 3)
    L = L[np.linalg.norm(L, axis=1) <= 6.2 * h0]
    rng = np.random.default_rng(5)
    ball = c0 + L + rng.uniform(-0.05, 0.05, L.shape) * h0
    lone = np.array([[-4.0 * wl.R_E, 0.1 * wl.R_E, 0.2 * wl.R_E], [-4.0 * wl.R_E + 2.0e6, 0.1 * wl.R_E, 0.2 * wl.R_E]])
    pts = np.concatenate([ball, lone])
    s = (pts - c0) / wl.R_E
    lnN = np.array([13.5, 13.4, 11.0, 9.6]) + s @ np.array([[-2.0, -2.0, -1.5, -1.0], [0.3, 0.3, 0.2, 0.1], [-0.2, -0.2, -0.1, -0.1]]) \
        + 0.5 * (s[:, :1] ** 2) * np.array([0.4, 0.4, 0.3, 0.2])
    path = str(tmp_path / "void.txt")
    wl.write_points_file(path, pts, lnN, g0["bounds"], g0["qs"], g0["ms"])
    mdl = {"kind": 4, "file": path}
    u = rng.normal(size=(64, 3))
    u /= np.linalg.norm(u, axis=1, keepdims=True)
    x = c0 + u * (6.2 * h0 + rng.uniform(1.1e6, 1.7e6, (64, 1)))
    x = x[np.linalg.norm(x, axis=1) > 1.5 * wl.R_E]
    # every weight of every query point is below the mask (h = 5 x the 30 km spacing; r / (h/4) > 29)
    r = np.linalg.norm(ball[None, :, :] - x[:, None, :], axis=2)
    assert (r < 3.0e6).sum(axis=1).min() >= 10 and np.exp(-(r.min() / (5.0 * h0 * 1.06 / 4.0)) ** 1.1) < 1e-16
    ref = refharness.run_mode("params", x, mdl)
    root, _, _ = refharness.scattered_root(mdl)
    m = oracle.Model.scattered_file(path, perm_seed=2)
    m.set_spacing(root, 0.0)   # the reference's tree root keeps spacing 0 (SURVEY A-12)
    mine = np.array([np.concatenate(m.plasma_params(p)) for p in x])
    assert np.isfinite(ref).all() and (ref[:, 4:8] > 0).all()
    # the fit extrapolates over 1 000 km from a 190 km ball: condition ~1e9, the summation order differs (A-12)
    assert np.allclose(mine[:, 4:8], ref[:, 4:8], rtol=1e-5, atol=0)
    # and it is the rule that matters: without it (all weights masked) the reference would return exp(0) = 1
    assert (np.abs(np.log(ref[:, 4])) > 5).all()
