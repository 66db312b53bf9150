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
