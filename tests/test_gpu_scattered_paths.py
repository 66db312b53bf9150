"""modelnum 4 on the device: the two cooperative stencil paths held against each other and against the CPU oracle.

shared path  = one candidate scan for the whole stencil, staged neighbour records, the weights of the six offset points
               from the centre's transcendental values by series (srt_scattered.hpp shared_fit)
own-list path = every stencil point scans for itself and evaluates etainv() per (point, neighbour) (own_fit); taken
               when there is no staging buffer (SRT_SCATTERED_STAGING=0), the stencil straddles a grid cell, ...
"direct" samples = samples so close to the centre that the series' bounds do not hold: their weights are evaluated by
               etainv() itself on the shared path.
"""
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR, vrel

pytestmark = pytest.mark.gpu


def _own_list(fn):
    """Run fn() with the staging buffer switched off (own-list path for every stencil)."""
    os.environ["SRT_SCATTERED_STAGING"] = "0"
    try:
        return fn()
    finally:
        del os.environ["SRT_SCATTERED_STAGING"]


def _states(oracle_model, n, seed, scale=0.9):
    from stanford_raytracer_amd import workloads as wl

    pos, d, w = wl.launch_set(n, seed)
    pos = pos * scale
    od = np.array([oracle_model.disp(p, dd, ww) for p, dd, ww in zip(pos, d, w)])
    ok = od[:, 8] > 0
    return pos[ok], od[ok, 8:9] * d[ok], w[ok]


def test_shared_and_own_list_paths_agree(gpu_models, oracle_scattered):
    """Same stencils through both paths: the right-hand sides agree far inside the amplification of the 10 m stencil
    (differences of ln N at 1e-16 relative are divided by 1e-6 |x|)."""
    g = gpu_models["scattered"]
    x, k, w = _states(oracle_scattered, 400, 4242)
    a = g.gradients(x, k, w, 1e-6)
    b = _own_list(lambda: g.gradients(x, k, w, 1e-6))
    assert np.isfinite(a).all() and np.isfinite(b).all()
    assert vrel(a[:, 0:3], b[:, 0:3]).max() <= 1e-9                      # dF/dk: densities at the centre only
    e = vrel(a[:, 4:7], b[:, 4:7])
    assert np.median(e) <= 1e-6 and np.percentile(e, 90) <= 1e-4         # dF/dx: d(ln N) over the stencil
    # one RK45 step, positions
    args = np.concatenate([x, k, w[:, None]], axis=1)[:120]
    dt = np.full(len(args), 1e-3)
    sa = g.rk_step(args, dt, 1e-6)
    sb = _own_list(lambda: g.rk_step(args, dt, 1e-6))
    es = vrel(sa[:, 14:17], sb[:, 14:17])            # the stages see k advanced by dt * (dF/dx)/(dF/dw): same yardstick as G3
    assert np.median(es) <= 1e-8 and np.percentile(es, 90) <= 1e-6


@pytest.mark.parametrize("offset_m", [30.0, 300.0, 3.0e4])
def test_samples_next_to_the_query_point(gpu_models, oracle_scattered, offset_m):
    """Query points a few stencil widths away from a sample: that sample's weight cannot come from the series (its
    distance changes by a large fraction across the stencil) and is evaluated directly; the others still use the
    series.  Held against the oracle and against the own-list path."""
    from stanford_raytracer_amd import workloads as wl

    g = gpu_models["scattered"]
    pts = np.load(os.path.join(GOLDEN_DIR, "points5500.npz"))["pts"]
    r = np.linalg.norm(pts, axis=1)
    sel = pts[(r > 1.3 * wl.R_E) & (r < 4.0 * wl.R_E)][:150]
    rng = np.random.default_rng(7)
    u = rng.normal(size=sel.shape)
    x = sel + offset_m * u / np.linalg.norm(u, axis=1, keepdims=True)
    _, d, w = wl.launch_set(len(x), 77)
    od = np.array([oracle_scattered.disp(p, dd, ww) for p, dd, ww in zip(x, d, w)])
    ok = od[:, 8] > 0
    x, k, w = x[ok][:80], (od[ok, 8:9] * d[ok])[:80], w[ok][:80]
    assert len(x) >= 40
    a = g.gradients(x, k, w, 1e-6)
    b = _own_list(lambda: g.gradients(x, k, w, 1e-6))
    o = np.array([oracle_scattered.grad(p, kk, ww, 1e-6) for p, kk, ww in zip(x, k, w)])
    assert vrel(a[:, 0:3], o[:, 0:3]).max() <= 1e-7 and vrel(a[:, 0:3], b[:, 0:3]).max() <= 1e-9
    for other in (b, o):
        e = vrel(a[:, 4:7], other[:, 4:7])
        assert np.median(e) <= 1e-5 and np.percentile(e, 90) <= 1e-3


def test_exact_window_goes_through_the_direct_weights(pointsfile):
    """scattered_interp_exact=1 has no series: every sample of the shared path is direct.  Same answers as the
    own-list path, and the per-lane evaluation of funcPlasmaParams."""
    from oracle import oracle
    from stanford_raytracer_amd import api

    kw = dict(exact=1, local_window_scale=2.0)
    g = api.Model.scattered_file(pointsfile, **kw)
    o = oracle.Model.scattered_file(pointsfile, perm_seed=2 | 0x80000000, **kw)
    x, k, w = _states(o, 200, 99)
    x, k, w = x[:60], k[:60], w[:60]
    a = g.gradients(x, k, w, 1e-6)
    b = _own_list(lambda: g.gradients(x, k, w, 1e-6))
    og = np.array([o.grad(p, kk, ww, 1e-6) for p, kk, ww in zip(x, k, w)])
    assert vrel(a[:, 0:3], b[:, 0:3]).max() <= 1e-9 and vrel(a[:, 0:3], og[:, 0:3]).max() <= 1e-7
    e = vrel(a[:, 4:7], b[:, 4:7])
    assert np.median(e) <= 1e-6 and np.percentile(e, 90) <= 1e-4


def test_trace_is_the_same_with_and_without_staging(gpu_models):
    """A short adaptive launch through both paths: launch rows identical, fates and row totals agree."""
    from stanford_raytracer_amd import workloads as wl

    g = gpu_models["scattered"]
    pos, d, w = wl.launch_set(1024, 31)
    pos = pos * 0.9
    kw = dict(fixedstep=0, dt0=1e-3, dtmax=0.1, tmax=0.05, maxerr=5e-4, maxsteps=60, del_=1e-6, outputper=1)
    ra, na, sa, _ = g.trace(pos, d, w, **kw)
    rb, nb, sb, _ = _own_list(lambda: g.trace(pos, d, w, **kw))
    both = (na > 1) & (nb > 1)
    assert both.sum() >= 400
    assert np.array_equal(ra[both, 0, 1:4], rb[both, 0, 1:4])
    assert vrel(ra[both, 0, 16:20], rb[both, 0, 16:20]).max() <= 1e-12      # densities on the launch row
    assert np.median(vrel(ra[both, 1, 1:4], rb[both, 1, 1:4])) <= 1e-8       # after the first step
    # Fates and row totals: an adaptive launch amplifies rounding (the density gradient is a difference over 1e-6 |x|), so
    # the yardstick is ONE path against itself with the launch points moved by 1e-13 of their length -- the two paths may
    # disagree as often as that (twice, plus three standard deviations of a count of 1 024), not by a constant.
    rc, nc, sc, _ = _own_list(lambda: g.trace(pos * (1.0 + 1e-13), d, w, **kw))
    yard_fate = float(np.mean(sb != sc))
    yard_rows = abs(int(nb.sum()) - int(nc.sum())) / nb.sum()
    n = len(sa)
    sigma = np.sqrt(max(yard_fate, 1.0 / n) * (1.0 - yard_fate) / n)
    assert np.mean(sa != sb) <= 2.0 * yard_fate + 3.0 * sigma, (np.mean(sa != sb), yard_fate)
    assert abs(int(na.sum()) - int(nb.sum())) / nb.sum() <= 2.0 * yard_rows + 0.02, (int(na.sum()), int(nb.sum()), yard_rows)


def test_paths_agree_on_the_full_size_sample_set(tmp_path):
    """BASELINE config[4]'s own sample set (825 k samples, up to ~2 000 neighbours per lookup: dozens of ring buffers
    per stencil, lists near the staging capacity): shared path against own-list path on launch points of that workload,
    and against the per-lane evaluation of funcPlasmaParams at the centre."""
    from stanford_raytracer_amd import api, workloads as wl

    pts, lnN = wl.make_points_config5(5)
    pfile = str(tmp_path / "points825k.txt")
    wl.write_points_file(pfile, pts, lnN, np.array([-10.0 * wl.R_E, 10.0 * wl.R_E] * 3))
    g = api.Model.scattered_file(pfile, window_scale=1.5, order=2, exact=0, local_window_scale=5.0)
    pos, d, w = wl.launch_set(1536, 5)
    disp = g.dispersion(pos, d * 1e-3, w)                    # columns 6.. : the roots' k
    kmag = np.where(disp[:, 8] > 0, disp[:, 8], disp[:, 6])
    ok = np.isfinite(kmag) & (kmag > 0)
    x, k, w = pos[ok], kmag[ok, None] * d[ok], w[ok]
    assert len(x) >= 1000
    a = g.gradients(x, k, w, 1e-6)
    assert np.array_equal(a, g.gradients(x, k, w, 1e-6), equal_nan=True)   # run to run: bit for bit
    b = _own_list(lambda: g.gradients(x, k, w, 1e-6))
    fin = np.isfinite(a).all(axis=1) & np.isfinite(b).all(axis=1)
    assert fin.mean() >= 0.99
    assert vrel(a[fin, 0:3], b[fin, 0:3]).max() <= 1e-8
    e = vrel(a[fin, 4:7], b[fin, 4:7])
    assert np.median(e) <= 1e-5 and np.percentile(e, 90) <= 1e-3
    # a short launch through both paths: same fates
    kw = dict(fixedstep=0, dt0=1e-3, dtmax=0.1, tmax=0.5, maxerr=5e-4, maxsteps=24, del_=1e-6, outputper=8)
    ra, na, sa, _ = g.trace(pos[:1024], d[:1024], w[:1024], **kw)
    rb, nb, sb, _ = _own_list(lambda: g.trace(pos[:1024], d[:1024], w[:1024], **kw))
    assert np.mean(sa == sb) >= 0.97
    both = (na > 1) & (nb > 1)
    assert vrel(ra[both, 0, 16:20], rb[both, 0, 16:20]).max() <= 1e-11


def test_order3_both_paths_and_a_short_trace(pointsfile):
    """--scattered_interp_order=3 (J = 20: tabular_monomials to degree 3, lsinterp_mod.f95:91-99, 244-273): the shared
    path (84 moments + 80 sums per lane, 20 x 20 Cholesky in private memory), the own-list path and the CPU oracle on
    the same stencils, then whole adaptive trajectories against the oracle."""
    from oracle import oracle
    from stanford_raytracer_amd import api, workloads as wl

    g = api.Model.scattered_file(pointsfile, order=3)
    o = oracle.Model.scattered_file(pointsfile, perm_seed=2 | 0x80000000, order=3)
    x, k, w = _states(o, 300, 1234)
    x, k, w = x[:100], k[:100], w[:100]
    a = g.gradients(x, k, w, 1e-6)
    b = _own_list(lambda: g.gradients(x, k, w, 1e-6))
    og = np.array([o.grad(p, kk, ww, 1e-6) for p, kk, ww in zip(x, k, w)])
    assert np.isfinite(a).all() and np.isfinite(b).all()
    assert vrel(a[:, 0:3], og[:, 0:3]).max() <= 1e-7 and vrel(a[:, 0:3], b[:, 0:3]).max() <= 1e-9
    for other in (b, og):
        e = vrel(a[:, 4:7], other[:, 4:7])
        assert np.median(e) <= 1e-5 and np.percentile(e, 90) <= 1e-3
    # trajectories: launch rows identical, fates and lengths as for order 2 (tests/test_gpu_parity.py)
    pos, d, ww = wl.launch_set(64, 515)
    pos = pos * 0.9
    kw = dict(fixedstep=0, dt0=1e-3, dtmax=0.1, tmax=0.01, maxerr=5e-4, maxsteps=60, del_=1e-6)
    rows, nrows, stop, _ = g.trace(pos, d, ww, outputper=1, **kw)
    orows, onrows, ostop, _ = o.trace(pos, d, ww, capacity=60, **kw)
    both = (nrows > 1) & (onrows > 1)
    assert both.sum() >= 32
    assert np.array_equal(rows[both, 0, 1:4], orows[both, 0, 1:4])
    e = np.abs(rows[both, 0, 16:20] - orows[both, 0, 16:20]) / orows[both, 0, 16:20]
    assert e.max() <= 1e-9
    assert np.mean(stop == ostop) >= 0.9
    assert np.median(vrel(rows[both, 1, 1:4], orows[both, 1, 1:4])) <= 1e-6
    assert abs(int(nrows.sum()) - int(onrows.sum())) <= 0.25 * onrows.sum()


@pytest.mark.parametrize("order", [0, 1])
def test_orders_0_and_1_through_both_paths(pointsfile, order):
    """--scattered_interp_order=0 / 1 (J = 1 / 4): the cooperative stencil's shared and own-list paths and the per-lane
    evaluation against the CPU oracle (order 0 is a weighted mean: its gradient comes from the weights alone)."""
    from oracle import oracle
    from stanford_raytracer_amd import api, workloads as wl

    g = api.Model.scattered_file(pointsfile, order=order)
    o = oracle.Model.scattered_file(pointsfile, perm_seed=2 | 0x80000000, order=order)
    pos, _, _ = wl.launch_set(200, 321)
    pos = pos * 0.9
    gp = g.plasma_params(pos)
    op = np.array([np.concatenate(o.plasma_params(p)) for p in pos])
    ok = op[:, 4] > 0
    assert (np.abs(gp[ok, 4:8] - op[ok, 4:8]) / op[ok, 4:8]).max() <= 1e-9
    x, k, w = _states(o, 300, 4321)
    x, k, w = x[:80], k[:80], w[:80]
    a = g.gradients(x, k, w, 1e-6)
    b = _own_list(lambda: g.gradients(x, k, w, 1e-6))
    og = np.array([o.grad(p, kk, ww, 1e-6) for p, kk, ww in zip(x, k, w)])
    assert np.isfinite(a).all() and np.isfinite(b).all()
    assert vrel(a[:, 0:3], og[:, 0:3]).max() <= 1e-7 and vrel(a[:, 0:3], b[:, 0:3]).max() <= 1e-9
    for other in (b, og):
        e = vrel(a[:, 4:7], other[:, 4:7])
        assert np.median(e) <= 1e-5 and np.percentile(e, 90) <= 1e-3


def test_too_few_samples_above_the_weight_mask_goes_through_the_own_list_rule(tmp_path):
    """lsinterp_mod.f95:316-323: weights <= 1e-16 are masked, and a point that keeps fewer than J samples "uses them all".
    Query points in a void next to a densely sampled ball: >= J samples inside the search radius (which the sparsest sample of
    the set makes large), every one of them tens of local window widths away, so every weight is below the mask.  The shared
    path cannot redo its weights (its list and side arrays are gone by the time the count is known): it must hand the stencil
    to the own-list path, which carries the rule -- the two runs are then the same arithmetic, bit for bit -- and the
    answers are the oracle's."""
    from oracle import oracle
    from stanford_raytracer_amd import api, workloads as wl

    g0 = np.load(os.path.join(GOLDEN_DIR, "points5500.npz"))
    h0 = 3.0e4                                         # lattice spacing inside the ball
    c0 = np.array([3.2 * wl.R_E, 0.4 * wl.R_E, 0.3 * wl.R_E])
    ax = np.arange(-6, 7) * h0
    L = np.stack(np.meshgrid(ax, ax, ax, indexing="ij"), -1).reshape(-1, 3)
    L = L[np.linalg.norm(L, axis=1) <= 6.2 * h0]
    rng = np.random.default_rng(5)
    ball = c0 + L + rng.uniform(-0.05, 0.05, L.shape) * h0
    lone = np.array([[-4.0 * wl.R_E, 0.1 * wl.R_E, 0.2 * wl.R_E], [-4.0 * wl.R_E + 2.0e6, 0.1 * wl.R_E, 0.2 * wl.R_E]])
    pts = np.concatenate([ball, lone])                 # the pair's 2 000 km spacing sets the search radius: 3 000 km
    s = (pts - c0) / wl.R_E
    base = np.array([13.5, 13.4, 11.0, 9.6])
    lnN = base + s @ np.array([[-2.0, -2.0, -1.5, -1.0], [0.3, 0.3, 0.2, 0.1], [-0.2, -0.2, -0.1, -0.1]]) \
        + 0.5 * (s[:, :1] ** 2) * np.array([0.4, 0.4, 0.3, 0.2])
    path = str(tmp_path / "void.txt")
    wl.write_points_file(path, pts, lnN, g0["bounds"], g0["qs"], g0["ms"])
    g = api.Model.scattered_file(path)
    o = oracle.Model.scattered_file(path, perm_seed=2 | 0x80000000)
    # query points 1 100 - 1 700 km outside the ball's surface: h = local_window_scale (5) x the 30 km spacing, so
    # r / (h/4) > 29 for every sample and every weight is below 2e-18
    u = rng.normal(size=(64, 3))
    u /= np.linalg.norm(u, axis=1, keepdims=True)
    x = c0 + u * (6.2 * h0 + rng.uniform(1.1e6, 1.7e6, (64, 1)))
    x = x[np.linalg.norm(x, axis=1) > 1.5 * wl.R_E]
    _, d, w = wl.launch_set(len(x), 12)
    od = np.array([o.disp(p, dd, ww) for p, dd, ww in zip(x, d, w)])
    ok = od[:, 8] > 0
    x, k, w = x[ok], od[ok, 8:9] * d[ok], w[ok]
    assert len(x) >= 16
    a = g.gradients(x, k, w, 1e-6)
    b = _own_list(lambda: g.gradients(x, k, w, 1e-6))
    og = np.array([o.grad(p, kk, ww, 1e-6) for p, kk, ww in zip(x, k, w)])
    assert np.isfinite(a).all() and np.isfinite(og).all()
    assert np.array_equal(a, b)                        # both runs end in own_fit: identical arithmetic
    assert vrel(a[:, 0:3], og[:, 0:3]).max() <= 1e-5   # dF/dk: the densities at the centre (an extrapolation over 1 000 km from a
                                                       # 190 km ball: the fit itself is good to 1e-7; measured 9e-7)
