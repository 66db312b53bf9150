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
