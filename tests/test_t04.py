"""use_tsyganenko = 1 (SURVEY 8f-4, T04_s half): the Tsyganenko & Sitnov 2005 external field.
CPU: stanford_raytracer_amd/csrc/srt_t04.hpp -- the very source the device compiles -- built for the host with g++
(tests/native/t04_host.cpp) and held against goldens captured from the reference's own T04_s / EXTERN
(tests/golden/t04_golden.npz, make_t04_golden.py).  GPU: the adapters' whole field tail on the device against the
reference's funcPlasmaParams with use_tsyganenko=1 (with the dipole and with IGRF as the base field)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN_DIR, ROOT
from stanford_raytracer_amd import workloads as wl


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GOLDEN_DIR, "t04_golden.npz"))


@pytest.fixture(scope="module")
def hostlib(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("t04") / "libt04h.so")
    subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-shared", "-fPIC", "-o", so,
                           os.path.join(ROOT, "tests", "native", "t04_host.cpp")])
    return C.CDLL(so)


def test_t04s_host_build_reproduces_the_reference(gold, hostlib):
    rows, want = gold["t04_in"], gold["t04_out"]
    got = np.zeros((len(rows), 3), dtype=np.float32)
    for i, r in enumerate(rows):
        pm = (C.c_float * 10)(*r[:10])
        out = (C.c_float * 3)()
        hostlib.t04h_t04s(pm, C.c_float(r[10]), C.c_float(r[11]), C.c_float(r[12]), C.c_float(r[13]), out)
        got[i] = list(out)
    exact = np.all(got.astype(np.float64) == want, axis=1)
    err = np.abs(got - want).max(axis=1) / np.linalg.norm(want, axis=1)
    assert exact.mean() >= 0.99 and err.max() <= 1.2e-7          # REAL outputs: identical but for a rare last-bit flip
    # the points cover inside, boundary layer and outside of the model magnetopause
    assert np.linalg.norm(want, axis=1).max() > 20 * np.linalg.norm(want, axis=1).min()


def test_extern_modules_match_the_reference(gold, hostlib):
    """Every module of EXTERN on its own: Chapman-Ferraro, the two tail modes, the four Birkeland modes and the IMF term
    bit for bit; the ring-current modules to 1e-9 (they differentiate a vector potential numerically with a 1e-4 step,
    which multiplies last-bit differences of pow/exp by 1e4 .. 1e6)."""
    rows, want = gold["ext_in"], gold["ext_out"]
    got = np.zeros_like(want)
    for i, r in enumerate(rows):
        inp = (C.c_double * 14)(*r)
        out = (C.c_double * 33)()
        hostlib.t04h_components(inp, out)
        got[i] = list(out)
    names = ["cf", "t1", "t2", "src", "prc", "r11", "r12", "r21", "r22", "himf", "total"]
    for k, nm in enumerate(names):
        a, b = got[:, 3 * k:3 * k + 3], want[:, 3 * k:3 * k + 3]
        scale = np.abs(b).max()
        if nm in ("cf", "r11", "r12", "r22", "himf"):
            assert np.array_equal(a, b), nm
        elif nm == "r21":
            assert np.mean(np.all(a == b, axis=1)) >= 0.99 and np.abs(a - b).max() <= 1e-15 * scale
        elif nm in ("t1", "t2"):
            assert np.abs(a - b).max() <= 1e-11 * scale, nm
        else:
            assert np.abs(a - b).max() <= 1e-9 * scale, nm


@pytest.mark.gpu
def test_gpu_field_tail_with_t04(gold, cfgfiles, grid16):
    from stanford_raytracer_amd import api
    api.init(0)
    x, parmod = gold["x"], gold["parmod"]
    F, b, qs, ms_ = grid16
    for tag in ("a", "b"):
        yd, ms, igrf = (int(v) for v in gold["date_" + tag])
        want = gold["B_" + tag]
        for m in (api.Model.ngo(cfgfiles["ngo"], yd, ms), api.Model.interp(F, b, qs, ms_, yearday=yd, msec=ms)):
            m.set_field(use_igrf=igrf, use_tsyganenko=1, parmod=parmod)
            got = m.plasma_params(x)[:, 16:19]
            err = np.abs(got - want).max(axis=1) / np.linalg.norm(want, axis=1)
            # REAL (fp32) base field + REAL T04 output: agreement to a few ulp32 of the total
            assert err.max() <= 3e-6, (tag, err.max())
            assert np.mean(err <= 1e-14) >= 0.8
            m.set_field(use_igrf=igrf, use_tsyganenko=0)
            assert np.abs(m.plasma_params(x)[:, 16:19] - want).max() > 1e-10     # the external field is tens of nT


@pytest.mark.gpu
def test_gpu_t04_trajectories_against_the_reference(gold, cfgfiles):
    """Fixed-step Ngo rays in dipole + T04 against the reference's own rows (raytracer_run with use_tsyganenko=1)."""
    from stanford_raytracer_amd import api
    api.init(0)
    m = api.Model.ngo(cfgfiles["ngo"]).set_field(use_tsyganenko=1, parmod=gold["parmod"])
    p = api.make_params(dt0=1e-3, dtmax=0.1, maxerr=5e-4, maxsteps=40, minalt=wl.MINALT, tmax=0.03, fixedstep=1, del_=1e-4)
    rows, nrows, stop, _ = m.trace(gold["run_pos0"], gold["run_dir0"], gold["run_w0"], params=p)
    assert np.array_equal(nrows, gold["run_nrows"]) and np.array_equal(stop, gold["run_stop"])
    ref = gold["run_rows"]
    for r in range(ref.shape[0]):
        T = nrows[r]
        pos_err = np.linalg.norm(rows[r, :T, 1:4] - ref[r, :T, 1:4], axis=1) / np.linalg.norm(ref[r, :T, 1:4], axis=1)
        B_err = np.linalg.norm(rows[r, :T, 13:16] - ref[r, :T, 13:16], axis=1) / np.linalg.norm(ref[r, :T, 13:16], axis=1)
        # row 0: the field itself (fp32 tail); then the drift of a trajectory whose dF/dx sees an fp32 staircase in B
        # (tests/test_igrf.py::test_gpu_igrf_trajectories has the argument)
        assert B_err[0] <= 3e-6 and pos_err[1] <= 1e-8
        assert pos_err.max() <= 5e-3 and B_err.max() <= 5e-3


@pytest.mark.gpu
def test_gpu_trace_with_t04_runs_and_differs_from_dipole(cfgfiles):
    """Rays traced in dipole + T04 finish with sane stop codes; near the Earth the path barely moves (the external
    field is < 1 % of the main field there), and it is not identical."""
    from stanford_raytracer_amd import api
    api.init(0)
    parmod = [4.0, -20.0, 2.0, -5.0, 0.5, 0.5, 0.3, 0.3, 0.4, 0.4]
    pos0, dir0, w0 = wl.launch_set(64, 17)
    p = api.make_params(dt0=1e-3, dtmax=0.02, tmax=0.05, maxerr=5e-4, maxsteps=64, minalt=wl.MINALT, outputper=4, del_=1e-4)
    d = api.Model.ngo(cfgfiles["ngo"])
    t = api.Model.ngo(cfgfiles["ngo"]).set_field(use_tsyganenko=1, parmod=parmod)
    rd, nd, sd, _ = d.trace(pos0, dir0, w0, params=p)
    rt, nt, st, _ = t.trace(pos0, dir0, w0, params=p)
    assert set(np.unique(st).tolist()) <= {0, 1, 2, 3, 5, 6, 9} and np.mean(st == 9) < 0.05
    both = (nd > 8) & (nt > 8)
    assert both.sum() > 20
    rel = np.linalg.norm(rt[both, 1, 1:4] - rd[both, 1, 1:4], axis=1) / np.linalg.norm(rd[both, 1, 1:4], axis=1)
    assert 0 < rel.max() < 1e-2
    dB = np.linalg.norm(rt[both, 0, 13:16] - rd[both, 0, 13:16], axis=1) / np.linalg.norm(rd[both, 0, 13:16], axis=1)
    assert dB.max() > 1e-5 and np.median(dB) < 0.2


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["interp", "scattered"])
def test_gpu_t04_with_the_table_models(gold, grid16, pointsfile, name):
    """use_tsyganenko = 1 with modelnum 3 and 4: the general field tail inside the trace kernels of the table models.
    There is no T04 in the CPU oracle (it would be srt_t04.hpp a second time); what can be held: every row's field equals
    funcPlasmaParams' field at the row's position (the layered kernel, itself held to the reference's goldens above), fixed
    steps give identical row counts with and without the external field where it is negligible, run-to-run bit identity."""
    from stanford_raytracer_amd import api
    api.init(0)
    F, b, qs, ms_ = grid16
    g = api.Model.interp(F, b, qs, ms_) if name == "interp" else api.Model.scattered_file(pointsfile)
    g.set_field(use_tsyganenko=1, parmod=gold["parmod"])
    pos, d, w = wl.launch_set(96, 23)
    pos = pos * 0.9
    kw = dict(fixedstep=1, dt0=1e-3, dtmax=0.1, tmax=0.008, maxerr=5e-4, maxsteps=10, del_=1e-6)
    rows, nrows, stop, steps = g.trace(pos, d, w, outputper=1, **kw)
    assert set(np.unique(stop).tolist()) <= {0, 1, 2, 3, 5, 6, 9} and np.mean(stop == 9) < 0.05
    live = nrows > 3
    assert live.sum() >= 40
    for r in (0, 1, 3):
        want = g.plasma_params(rows[live, r, 1:4])[:, 16:19]
        assert np.array_equal(rows[live, r, 13:16], want), r       # same device functions, same arguments
    rows2, nrows2, stop2, _ = g.trace(pos, d, w, outputper=1, **kw)
    assert np.array_equal(nrows, nrows2) and np.array_equal(np.nan_to_num(rows), np.nan_to_num(rows2))
    g.set_field(use_tsyganenko=0)
    rd, nd, _, _ = g.trace(pos, d, w, outputper=1, **kw)
    dB = np.linalg.norm(rows[live, 0, 13:16] - rd[live, 0, 13:16], axis=1) / np.linalg.norm(rd[live, 0, 13:16], axis=1)
    assert dB.max() > 1e-6 and np.median(dB) < 0.2
    assert np.mean(nd == nrows) >= 0.9
