"""root = 1 (raytracer.f95:685-690, 828-836) and modelnum 4 under the fixed-step integrator, against goldens made by the
reference itself (tests/golden/make_root1_golden.py -> root1_golden.npz).

CPU part: the C oracle is held to the reference's rows (bit for bit for the Ngo / interp models, as on the other G4
fixtures).  GPU part (-m gpu): the HIP path against the same goldens -- fixed step: row counts, stop codes and time grids
EQUAL, positions within 10x max(survey ladder, the oracle's own divergence under a 1e-9 shift of the launch points);
adaptive: the bars of tests/test_gpu_trace.py::test_adaptive_trajectories.
"""
import os

import numpy as np
import pytest

from conftest import DELS, GOLDEN_DIR, vrel

HF_KW = dict(dt0=2e-5, dtmax=1e-3, maxerr=5e-4)


@pytest.fixture(scope="module")
def g1():
    return np.load(os.path.join(GOLDEN_DIR, "root1_golden.npz"))


def _kw(prm, name):
    return dict(dt0=prm[0], dtmax=prm[1], tmax=prm[2], maxerr=prm[3], minalt=prm[4], maxsteps=int(prm[5]), root=int(prm[6]),
                fixedstep=int(prm[7]), del_=DELS[name])


def test_the_two_roots_are_different_modes(g1):
    """The fixture is meaningful: on the same HF rays root 1 and root 2 give different wave normals and paths."""
    a, b = g1["ngo_fixed_root1_rows"], g1["ngo_fixed_root2_rows"]
    assert np.array_equal(a[:, 0, 1:4], b[:, 0, 1:4])                      # same launch points
    assert vrel(a[:, 0, 10:13], b[:, 0, 10:13]).min() > 1e-4               # different refractive-index vectors on row 0
    assert vrel(a[:, 100, 1:4], b[:, 100, 1:4]).max() > 1e-6               # and different paths


@pytest.mark.parametrize("name,tag", [("ngo", "ngo_fixed_root1"), ("ngo", "ngo_adaptive_root1"), ("ngo", "ngo_fixed_root2"),
                                      ("interp", "interp_fixed_root1"), ("interp", "interp_adaptive_root1")])
def test_oracle_matches_the_reference_root1(g1, oracle_models, name, tag):
    rays, prm = g1["hf_rays"], g1[tag + "_params"]
    ref_rows, ref_n, ref_stop = g1[tag + "_rows"], g1[tag + "_nrows"], g1[tag + "_stop"]
    rows, nrows, stop, _ = oracle_models[name].trace(rays[:, :3], rays[:, 3:6], rays[:, 6], capacity=int(ref_rows.shape[1]),
                                                     **_kw(prm, name))
    assert np.array_equal(nrows, ref_n) and np.array_equal(stop, ref_stop)
    for i in range(len(nrows)):
        a, b = rows[i, :nrows[i]], ref_rows[i, :nrows[i]]
        assert np.array_equal(a[:, 0], b[:, 0])                            # the same time grid, adaptive included
        assert np.allclose(a, b, rtol=1e-12, atol=0.0)
    assert np.mean(rows[:, :, 1:4] == ref_rows[:, :, 1:4]) > 0.99          # (bit-identical but for a handful of last bits)


def test_oracle_scattered_fixed_step(g1, pointsfile):
    from oracle import oracle

    m = oracle.Model.scattered_file(pointsfile, perm_seed=2)
    rays, prm = g1["scattered_rays"], g1["scattered_fixed_params"]
    ref_rows, ref_n, ref_stop = g1["scattered_fixed_rows"], g1["scattered_fixed_nrows"], g1["scattered_fixed_stop"]
    rows, nrows, stop, _ = m.trace(rays[:, :3], rays[:, 3:6], rays[:, 6], capacity=int(ref_rows.shape[1]), **_kw(prm, "scattered"))
    assert np.array_equal(nrows, ref_n) and np.array_equal(stop, ref_stop)
    assert np.allclose(rows[:, :, 0], ref_rows[:, :, 0], rtol=1e-12)       # fixed step: the same time grid
    assert np.array_equal(rows[:, 0, 1:4], ref_rows[:, 0, 1:4])
    d = vrel(rows[:, 1, 1:4], ref_rows[:, 1, 1:4])
    assert np.median(d) <= 1e-6                                            # dF/dx amplifies the summation-order noise of ln N
    last = int(ref_n.min()) - 1
    assert np.median(vrel(rows[:, last, 1:4], ref_rows[:, last, 1:4])) <= 1e-4


# ------------------------------------------------------------------------------------------------------------ GPU
LADDER = {1: 1.2e-10, 10: 7e-8, 100: 4e-5}


def _divergence(ra, na, rb, nb, r):
    sel = (na > r) & (nb > r)
    return float(vrel(ra[sel, r, 1:4], rb[sel, r, 1:4]).max()) if sel.any() else 0.0


def _yardstick(om, rays, kw, rows_at, cap):
    base = om.trace(rays[:, :3], rays[:, 3:6], rays[:, 6], capacity=cap, **kw)
    out = {r: 0.0 for r in rows_at}
    for eps in (1e-9, -1e-9):
        pert = om.trace(rays[:, :3] * (1 + eps), rays[:, 3:6], rays[:, 6], capacity=cap, **kw)
        for r in rows_at:
            out[r] = max(out[r], _divergence(pert[0], pert[1], base[0], base[1], r))
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("name,tag,rows_at", [("ngo", "ngo_fixed_root1", (1, 10, 100)), ("interp", "interp_fixed_root1", (1, 10, 100)),
                                              ("ngo", "ngo_fixed_root2", (1, 10, 100))])
def test_gpu_fixed_step_root1(g1, gpu_models, oracle_models, name, tag, rows_at):
    rays, prm = g1["hf_rays"], g1[tag + "_params"]
    ref_rows, ref_n, ref_stop = g1[tag + "_rows"], g1[tag + "_nrows"], g1[tag + "_stop"]
    kw = _kw(prm, name)
    cap = int(ref_rows.shape[1])
    rows, nrows, stop, steps = gpu_models[name].trace(rays[:, :3], rays[:, 3:6], rays[:, 6], outputper=1, **dict(kw, maxsteps=cap + 1))
    assert np.array_equal(nrows, ref_n) and np.array_equal(stop, ref_stop) and steps == int((ref_n - 1).sum())
    assert np.array_equal(rows[:, 0, 0:4], ref_rows[:, 0, 0:4])
    assert vrel(rows[:, 0, 10:13], ref_rows[:, 0, 10:13]).max() <= 1e-10   # the launch row's n: the chosen root itself
    yard = _yardstick(oracle_models[name], rays, dict(kw, maxsteps=cap + 1), rows_at, cap)
    for r in rows_at:
        d = _divergence(rows, nrows, ref_rows, ref_n, r)
        bound = 10 * max(yard[r], LADDER[r])
        assert d <= bound, "row %d: position divergence %.2e > %.2e" % (r, d, bound)
        assert np.allclose(rows[:, r, 0], ref_rows[:, r, 0], rtol=1e-12)   # same time grid


@pytest.mark.gpu
@pytest.mark.parametrize("name,tag", [("ngo", "ngo_adaptive_root1"), ("interp", "interp_adaptive_root1")])
def test_gpu_adaptive_root1(g1, gpu_models, oracle_models, name, tag):
    from test_gpu_trace import curve_distance

    rays, prm = g1["hf_rays"], g1[tag + "_params"]
    ref_rows, ref_n, ref_stop = g1[tag + "_rows"], g1[tag + "_nrows"], g1[tag + "_stop"]
    kw = _kw(prm, name)
    rows, nrows, stop, _ = gpu_models[name].trace(rays[:, :3], rays[:, 3:6], rays[:, 6], outputper=1, **kw)
    om = oracle_models[name]
    cap = int(ref_rows.shape[1])
    base = om.trace(rays[:, :3], rays[:, 3:6], rays[:, 6], capacity=cap, **kw)
    yard_curve, yard_rows = 0.0, 0
    for eps in (1e-9, -1e-9):
        pert = om.trace(rays[:, :3] * (1 + eps), rays[:, 3:6], rays[:, 6], capacity=cap, **kw)
        yard_curve = max(yard_curve, curve_distance(pert[0], pert[1], base[0], base[1], prm[2]))
        yard_rows = max(yard_rows, abs(int(pert[1].sum()) - int(base[1].sum())))
    assert np.array_equal(rows[:, 0, 1:4], ref_rows[:, 0, 1:4])
    assert vrel(rows[:, 0, 10:13], ref_rows[:, 0, 10:13]).max() <= 1e-10
    assert np.mean(stop == ref_stop) >= 0.9
    assert curve_distance(rows, nrows, ref_rows, ref_n, prm[2]) <= 10 * max(yard_curve, 1e-7)
    assert abs(int(nrows.sum()) - int(ref_n.sum())) <= max(3 * yard_rows, 0.05 * ref_n.sum())


@pytest.mark.gpu
def test_gpu_scattered_fixed_step(g1, gpu_models, oracle_scattered):
    """modelnum 4, fixed-step RK4: row counts, stop codes and the time grid EQUAL to the reference's; positions within 10x the
    oracle's own divergence under a 1e-9 shift (the summation order of the fit is not the reference's: SURVEY A-12)."""
    rays, prm = g1["scattered_rays"], g1["scattered_fixed_params"]
    ref_rows, ref_n, ref_stop = g1["scattered_fixed_rows"], g1["scattered_fixed_nrows"], g1["scattered_fixed_stop"]
    kw = _kw(prm, "scattered")
    cap = int(ref_rows.shape[1])
    rows, nrows, stop, steps = gpu_models["scattered"].trace(rays[:, :3], rays[:, 3:6], rays[:, 6], outputper=1, **dict(kw, maxsteps=cap + 1))
    assert np.array_equal(nrows, ref_n) and np.array_equal(stop, ref_stop) and steps == int((ref_n - 1).sum())
    assert np.array_equal(rows[:, 0, 0:4], ref_rows[:, 0, 0:4])
    e0 = np.abs(rows[:, 0, 16:20] - ref_rows[:, 0, 16:20]) / ref_rows[:, 0, 16:20]
    assert np.percentile(e0, 90) <= 1e-9
    rows_at = (1, 10, int(ref_n.min()) - 1)
    yard = _yardstick(oracle_scattered, rays, dict(kw, maxsteps=cap + 1), rows_at, cap)
    for r in rows_at:
        assert np.allclose(rows[:, r, 0], ref_rows[:, r, 0], rtol=1e-12)   # same time grid
        d = _divergence(rows, nrows, ref_rows, ref_n, r)
        bound = 10 * max(yard[r], LADDER.get(r, 7e-8), 1e-6)               # 1e-6: dF/dx over a 10 m stencil (test_oracle_golden G4)
        assert d <= bound, "row %d: position divergence %.2e > %.2e" % (r, d, bound)


# ---- the first adaptive attempt: the one trajectory file the reference's tree holds --------------
def _growth_prefix(times):
    """leading time stamps that follow t_{n+1} = t_n + 1e-3 * 1.25**n exactly (the controller growing dt from the first step)"""
    t, dt, n = 0.0, 1e-3, 0
    while n + 1 < len(times) and abs(times[n + 1] - (t + dt)) <= 2e-15 * (t + dt):  # (the file prints 16 digits)
        t, dt, n = t + dt, dt * 1.25, n + 1
    return n


def test_first_attempt_policy_1_is_what_the_references_own_output_shows(oracle_models):
    """gcpm/output.ray of the reference tree (its gfortran-built binary): dt grows by 1.25 from the FIRST accepted step --
    policy 1.  The oracle under policy 1 produces exactly those time stamps on rays whose error estimate stays below
    maxerr / 100; under policy 0 (flang's MAX) the first step is never grown."""
    ref_t = np.loadtxt(os.path.join(GOLDEN_DIR, "reference_gcpm_output_times.txt"))
    nref = _growth_prefix(ref_t)
    assert nref >= 8 and np.allclose(ref_t[1:4], [1e-3, 2.25e-3, 3.8125e-3], rtol=1e-15, atol=0)
    from stanford_raytracer_amd import workloads as wl

    pos, d, w = wl.launch_set(64, 11)
    kw = dict(fixedstep=0, dt0=1e-3, dtmax=0.1, tmax=0.1, maxerr=5e-4, maxsteps=40, del_=DELS["ngo"])
    r1, n1, _, _ = oracle_models["ngo"].trace(pos, d, w, capacity=40, first_attempt_policy=1, **kw)
    r0, n0, _, _ = oracle_models["ngo"].trace(pos, d, w, capacity=40, first_attempt_policy=0, **kw)
    grow1 = np.array([_growth_prefix(r1[i, :n1[i], 0]) for i in range(len(w))])
    grow0 = np.array([_growth_prefix(r0[i, :n0[i], 0]) for i in range(len(w))])
    assert (grow1 >= 3).mean() >= 0.25            # rays with a small error estimate reproduce the file's 0, 1e-3, 2.25e-3, 3.8125e-3
    assert grow0.max() <= 1                        # policy 0: row 2 sits at 2e-3 at the earliest, never at 2.25e-3
    sel = np.where(grow1 >= 3)[0]
    assert np.allclose(r1[sel[0], :4, 0], ref_t[:4], rtol=2e-15, atol=0)


@pytest.mark.gpu
def test_cli_default_first_attempt_follows_the_references_own_output(tmp_path, cfgfiles):
    """The CLI's default (no --first_attempt_policy flag) is policy 1: its .ray file shows the time stamps of the reference's
    own gcpm/output.ray on rays with a small error estimate."""
    import subprocess

    from conftest import parse_ray_file
    from stanford_raytracer_amd import build, workloads as wl

    ref_t = np.loadtxt(os.path.join(GOLDEN_DIR, "reference_gcpm_output_times.txt"))
    pos, d, w = wl.launch_set(64, 11)
    rays, out = str(tmp_path / "rays.txt"), str(tmp_path / "out.ray")
    wl.write_rays_file(rays, pos, d, w)
    cmd = [build.CLI, "--outputper=1", "--dt0=1e-3", "--dtmax=0.1", "--tmax=0.1", "--root=2", "--fixedstep=0", "--maxerr=5e-4",
           "--maxsteps=40", "--minalt=%r" % wl.MINALT, "--inputraysfile=" + rays, "--outputfile=" + out, "--modelnum=1",
           "--yearday=2010001", "--milliseconds_day=0", "--use_tsyganenko=0", "--use_igrf=0", "--ngo_configfile=" + cfgfiles["ngo"]]
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL)
    rec = parse_ray_file(out)
    hits = 0
    for r in np.unique(rec[:, 0]):
        t = rec[rec[:, 0] == r, 2]
        hits += int(len(t) >= 4 and np.allclose(t[:4], ref_t[:4], rtol=2e-15, atol=0))
    assert hits >= 16
