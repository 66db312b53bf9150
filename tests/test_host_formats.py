"""CPU: file formats of the drop-in boundary (pure host code of libsrt_hip.so and the workload writers)."""
import os

import numpy as np

from conftest import GOLDEN_DIR
from stanford_raytracer_amd import api, workloads as wl


def test_ray_file_writer_matches_reference_bytes(golden, tmp_path):
    """srt_write_ray_file reproduces the reference's text records byte for byte when fed the reference's
    own rows (record format raytracer_driver.f95:1197-1217; golden = config 1, outputper=25)."""
    rows_all = golden["g4_ngo_fixed_rows"]          # [16, 101, 20], every row
    nrows, stop = golden["g4_ngo_fixed_nrows"], golden["g4_ngo_fixed_stop"]
    rays = golden["g4_rays"]
    p = api.make_params(maxsteps=2000, outputper=25)
    slots = api.lib().srt_rows_per_ray(__import__("ctypes").byref(p))
    rows = np.zeros((16, slots, 20))
    kept = rows_all[:, ::25]
    rows[:, :kept.shape[1]] = kept
    out = tmp_path / "mine.ray"
    api.write_ray_file(str(out), (4, wl.QS, wl.MS), p, rays[:, 6], rows, nrows, stop)
    ref = open(os.path.join(GOLDEN_DIR, "config1_outputper25.ray")).read()
    assert out.read_text() == ref


def test_es24_special_values(tmp_path):
    p = api.make_params(maxsteps=1, outputper=1)
    rows = np.zeros((1, 1, 20))
    rows[0, 0, 0] = -0.0
    rows[0, 0, 1] = 1e-310  # subnormal
    rows[0, 0, 2] = -1.5e300
    rows[0, 0, 3] = np.nan
    out = tmp_path / "x.ray"
    api.write_ray_file(str(out), (4, wl.QS, wl.MS), p, np.array([1.0]), rows, np.array([1], dtype=np.int32),
                       np.array([2], dtype=np.int32))
    line = out.read_text().splitlines()[0]
    assert len(line) == 10 + 10 + 17 * 24 + 10 + 16 * 24
    assert line[:20] == "         1         2"
    assert "-1.500000000000000E+300" in line and "NaN" in line


def test_rays_file_roundtrip(tmp_path):
    pos, d, w = wl.launch_set(37, 5)
    f = tmp_path / "rays.txt"
    wl.write_rays_file(str(f), pos, d, w)
    p2, d2, w2 = api.read_rays_file(str(f))
    assert np.array_equal(p2, pos) and np.array_equal(d2, d) and np.array_equal(w2, w)
    # ragged tail: an incomplete last record ends the file like the reference's iostat /= 0 (driver:1146-1150)
    with open(f, "a") as fh:
        fh.write("1.0 2.0 3.0\n")
    p3, _, w3 = api.read_rays_file(str(f))
    assert len(w3) == 37
    empty = tmp_path / "empty.txt"
    empty.write_text("")
    assert len(api.read_rays_file(str(empty))[2]) == 0


def test_grid_file_writer_readable_by_oracle(tmp_path):
    from oracle import oracle

    F, b = wl.make_grid(6, half_width=3 * wl.R_E)
    gf = tmp_path / "g.txt"
    wl.write_grid_file(str(gf), F, b)
    a = oracle.Model.interp_file(str(gf))
    c = oracle.Model.interp(F, b, wl.QS, wl.MS)
    x = np.array([1.3 * wl.R_E, -0.4 * wl.R_E, 0.9 * wl.R_E])
    assert np.array_equal(np.concatenate(a.plasma_params(x)), np.concatenate(c.plasma_params(x)))


def test_launch_set_is_seeded_and_sane():
    p1, d1, w1 = wl.launch_set(1000, 3)
    p2, d2, w2 = wl.launch_set(1000, 3)
    assert np.array_equal(p1, p2) and np.array_equal(d1, d2) and np.array_equal(w1, w2)
    r = np.linalg.norm(p1, axis=1)
    assert r.min() >= wl.R_E + 500e3 and r.max() <= 5 * wl.R_E
    assert np.allclose(np.linalg.norm(d1, axis=1), 1.0)
    ang = np.degrees(np.arccos(np.abs(np.sum(d1 * wl.dipole_b(p1), axis=1)) / np.linalg.norm(wl.dipole_b(p1), axis=1)))
    assert ang.min() >= 10 - 1e-6 and ang.max() <= 70 + 1e-6
    f = w1 / (2 * np.pi)
    assert f.min() >= 500 and f.max() <= 10000
