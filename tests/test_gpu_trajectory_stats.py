"""GPU (-m gpu): adaptive trajectories against the CPU oracle with the ORACLE'S OWN SENSITIVITY as the yardstick, on
enough rays for the statistics to mean something.

Adaptive step sequences are chaotic at rounding level (SURVEY A-9: the reference splits from its own FMA rebuild after
2-22 accepted steps, 1 of 16 rays changes fate).  A fixed constant cannot tell "as close as the arithmetic allows" from
"subtly wrong", so every bar below is a small multiple of what the oracle does to ITSELF when the launch points are
shifted by 1e-9 relative (~1 cm) -- measured in the same test, on the same rays:
  * per-ray distance between the position curves (cubic-Hermite resampled on a common time grid),
  * agreement of stop codes, of the first controller decisions (time stamps of rows 1-3) and of the row totals;
and, on 10 k rays of each BASELINE workload (1 k for the scattered one: its oracle does 4e3 steps/s), the distributions:
stop-code histogram, distribution of the final radial distance (Kolmogorov-Smirnov distance; ray-by-ray closeness), total
accepted steps.
"""
import os

import numpy as np
import pytest

from conftest import DELS, vrel
from stanford_raytracer_amd import workloads as wl

pytestmark = pytest.mark.gpu
C_LIGHT = 299792458.0
NTH = max(1, min(16, os.cpu_count() or 1))


def hermite(tt, t, y, v):
    idx = np.clip(np.searchsorted(t, tt, side="right") - 1, 0, len(t) - 2)
    h = t[idx + 1] - t[idx]
    u = (tt - t[idx]) / h
    h00, h10, h01, h11 = 2 * u**3 - 3 * u**2 + 1, u**3 - 2 * u**2 + u, -2 * u**3 + 3 * u**2, u**3 - u**2
    return h00[:, None] * y[idx] + (h10 * h)[:, None] * v[idx] + h01[:, None] * y[idx + 1] + (h11 * h)[:, None] * v[idx + 1]


def curve_distances(ra, na, rb, nb, tmax):
    """per ray: max relative distance between the two position curves on a common time grid (NaN: too short)."""
    out = np.full(ra.shape[0], np.nan)
    for i in range(ra.shape[0]):
        ta, tb = ra[i, :na[i], 0], rb[i, :nb[i], 0]
        if len(ta) < 3 or len(tb) < 3:
            continue
        tt = np.linspace(0, min(ta[-1], tb[-1], tmax), 20)
        pa = hermite(tt, ta, ra[i, :na[i], 1:4], ra[i, :na[i], 7:10] * C_LIGHT)
        pb = hermite(tt, tb, rb[i, :nb[i], 1:4], rb[i, :nb[i], 7:10] * C_LIGHT)
        out[i] = float(vrel(pa, pb).max())
    return out


def compare(a, b, tmax):
    """statistics of run a against run b (each = rows, nrows, stop)."""
    (ra, na, sa), (rb, nb, sb) = a, b
    d = curve_distances(ra, na, rb, nb, tmax)
    d = d[np.isfinite(d)]
    both = (na > 4) & (nb > 4)
    same_t = np.all(ra[both, 1:4, 0] == rb[both, 1:4, 0], axis=1)
    return {"curve_median": float(np.median(d)), "curve_p90": float(np.percentile(d, 90)), "curve_p99": float(np.percentile(d, 99)),
            "stop_agree": float(np.mean(sa == sb)), "same_t": float(same_t.mean()), "n_curves": int(len(d)),
            "rows_rel": abs(int(na.sum()) - int(nb.sum())) / max(int(nb.sum()), 1)}


CASES = {"ngo": dict(n=1024, scale=1.0, tmax=0.2, maxsteps=120), "interp": dict(n=1024, scale=0.9, tmax=0.2, maxsteps=120),
         "scattered": dict(n=1024, scale=0.9, tmax=0.02, maxsteps=40)}


@pytest.mark.parametrize("name", ["ngo", "interp", "scattered"])
def test_adaptive_trajectories_within_the_oracles_own_sensitivity(gpu_models, oracle_models, oracle_scattered, name):
    c = CASES[name]
    g = gpu_models[name]
    o = oracle_scattered if name == "scattered" else oracle_models[name]
    pos, d, w = wl.launch_set(c["n"], 4321)
    pos = pos * c["scale"]
    kw = dict(fixedstep=0, dt0=1e-3, dtmax=0.1, tmax=c["tmax"], maxerr=5e-4, maxsteps=c["maxsteps"], del_=DELS[name])
    gr = g.trace(pos, d, w, outputper=1, **kw)[:3]
    base = o.trace(pos, d, w, capacity=c["maxsteps"], nthreads=NTH, **kw)[:3]
    yard = None
    for eps in (1e-9, -1e-9):
        pert = o.trace(pos * (1 + eps), d, w, capacity=c["maxsteps"], nthreads=NTH, **kw)[:3]
        s = compare(pert, base, c["tmax"])
        yard = s if yard is None else {k: (min if k in ("stop_agree", "same_t") else max)(yard[k], s[k]) for k in s}
    mine = compare(gr, base, c["tmax"])
    msg = "\n%s  GPU vs oracle: %s\n%s  oracle vs oracle(launch shifted 1e-9): %s" % (name, mine, name, yard)
    print(msg)
    assert mine["n_curves"] >= 0.5 * c["n"]
    # row 0 is a pure function of the inputs
    live = (gr[1] > 1) & (base[1] > 1)
    assert np.array_equal(gr[0][live, 0, 1:4], base[0][live, 0, 1:4])
    # curves: no further from the oracle than 3x the oracle is from itself (measured: 0.9-1.4x; floors: the survey ladder at
    # 10 / 100 steps)
    assert mine["curve_median"] <= 3 * max(yard["curve_median"], 7e-8), msg
    assert mine["curve_p90"] <= 3 * max(yard["curve_p90"], 4e-5), msg
    assert mine["curve_p99"] <= 3 * max(yard["curve_p99"], 4e-5), msg
    # decisions: at least as good as the oracle against itself, less a 3-sigma binomial margin for n rays
    marg = lambda p, n: 3.0 * np.sqrt(max(p * (1 - p), 1e-4) / n)
    assert mine["stop_agree"] >= yard["stop_agree"] - marg(yard["stop_agree"], c["n"]) - 0.01, msg
    assert mine["same_t"] >= yard["same_t"] - marg(yard["same_t"], c["n"]) - 0.02, msg
    assert mine["rows_rel"] <= max(3 * yard["rows_rel"], 0.02), msg


def test_first_attempt_policy_1_matches_the_oracle(gpu_models, oracle_models):
    """srt_params.first_attempt_policy = 1: the first attempt's error comes from the k term alone (MAX with a NaN
    operand as gfortran <= 8 -- the reference's own toolchain, Makefile:3,10 / Dockerfile:1 -- evaluates it; SURVEY A-1),
    so step 1 takes part in accept / grow / reject like any other.  The kernel branch against the oracle's, and against
    policy 0 (flang: accepted at dt0, never grown)."""
    for name in ("ngo", "interp"):
        g, o = gpu_models[name], oracle_models[name]
        pos, d, w = wl.launch_set(512, 777)
        if name != "ngo":
            pos = pos * 0.9
        kw = dict(fixedstep=0, dt0=1e-3, dtmax=0.1, tmax=0.05, maxerr=5e-4, maxsteps=30, del_=DELS[name])
        r1, n1, s1, _ = g.trace(pos, d, w, outputper=1, first_attempt_policy=1, **kw)
        r0, n0, s0, _ = g.trace(pos, d, w, outputper=1, first_attempt_policy=0, **kw)
        o1, on1, os1, _ = o.trace(pos, d, w, capacity=30, first_attempt_policy=1, nthreads=NTH, **kw)
        o0, on0, _, _ = o.trace(pos, d, w, capacity=30, first_attempt_policy=0, nthreads=NTH, **kw)
        both = (n1 > 4) & (on1 > 4)
        assert both.sum() >= 200
        # the policy shows: under policy 0 row 1 sits at dt0 on every ray; under policy 1 the first step is rejected on
        # some rays (row 1 before dt0) and the second step is grown on others (row 2 beyond 2 dt0)
        b0 = (n0 > 4) & (on0 > 4)
        assert np.all(r0[b0, 1, 0] == 1e-3) and np.all(r0[b0, 2, 0] <= 2e-3 * (1 + 1e-12))
        moved = (r1[both, 1, 0] != 1e-3) | (r1[both, 2, 0] > 2e-3 * (1 + 1e-12))
        omoved = (o1[both, 1, 0] != 1e-3) | (o1[both, 2, 0] > 2e-3 * (1 + 1e-12))
        assert omoved.mean() > 0.2 and abs(moved.mean() - omoved.mean()) <= 0.05
        # the very first decision depends on the launch state alone: same time stamp of row 1 on (nearly) every ray
        assert np.mean(r1[both, 1, 0] == o1[both, 1, 0]) >= 0.97
        same_t = np.all(r1[both, 1:4, 0] == o1[both, 1:4, 0], axis=1)
        assert same_t.mean() >= (0.9 if name == "ngo" else 0.75)
        assert np.mean(s1 == os1) >= 0.9


# ---- distributions on the BASELINE workloads -------------------------------------------------------------------------
def final_radius(rows, nrows, outputper):
    kept = (np.maximum(nrows, 1) - 1) // outputper
    last = rows[np.arange(len(nrows)), np.minimum(kept, rows.shape[1] - 1), 1:4]
    return np.linalg.norm(last, axis=1)


def hist(stop):
    return np.array([np.mean(stop == c) for c in (0, 1, 2, 3, 5, 6, 9)])


def distribution_check(tag, g, o, pos, d, w, kw, outputper):
    n = len(w)
    rows, nrows, stop, steps = g.trace(pos, d, w, outputper=outputper, **kw)
    # the oracle keeps every row: capacity = maxsteps; resample its kept rows the same way
    orows, onrows, ostop, osteps = o.trace(pos, d, w, capacity=kw["maxsteps"], nthreads=NTH, **kw)
    prows, pnrows, pstop, psteps = o.trace(pos * (1 + 1e-9), d, w, capacity=kw["maxsteps"], nthreads=NTH, **kw)
    fr = final_radius(rows, nrows, outputper)
    fo = final_radius(orows[:, ::outputper], onrows, outputper)
    fp = final_radius(prows[:, ::outputper], pnrows, outputper)
    hg, ho, hp = hist(stop), hist(ostop), hist(pstop)
    # distribution of the final radial distance: Kolmogorov-Smirnov distance between the empirical CDFs (percentiles are
    # ill-conditioned here: the CDF is flat between the rays that came down to minalt and the rest), and, ray by ray, the
    # share of rays that end within 1 % of where the oracle's end
    ks_g, ks_p = ks_distance(fr, fo), ks_distance(fp, fo)
    close_g, close_p = np.mean(np.abs(fr - fo) <= 1e-2 * fo), np.mean(np.abs(fp - fo) <= 1e-2 * fo)
    msg = ("\n%s: %d rays\n  stop histogram GPU    %s\n                 oracle %s\n                 oracle' %s\n"
           "  accepted steps GPU %d oracle %d oracle' %d\n  final |pos|: KS distance to the oracle GPU %.4f oracle' %.4f; within 1 %% of "
           "the oracle's GPU %.4f oracle' %.4f; same stop code GPU %.4f oracle' %.4f"
           % (tag, n, np.round(hg, 4), np.round(ho, 4), np.round(hp, 4), steps, osteps, psteps, ks_g, ks_p, close_g, close_p,
              np.mean(stop == ostop), np.mean(pstop == ostop)))
    print(msg)
    sigma = np.sqrt(np.maximum(ho * (1 - ho), 1e-4) / n)
    # histogram: each code's share within 3 binomial sigma + 3x what the oracle moves under the 1 cm shift
    assert np.all(np.abs(hg - ho) <= 3 * sigma + 3 * np.abs(hp - ho) + 2e-3), msg
    assert abs(steps - osteps) <= max(0.02 * osteps, 3 * abs(psteps - osteps)), msg
    assert ks_g <= 3 * ks_p + 3.0 / np.sqrt(n), msg
    # (shares of n rays: three binomial sigma of the oracle's own share, as for the histogram -- 0.02 at 10 k rays is 4 sigma, at the
    # scattered model's 1 000 rays it was 1.3 sigma: a build that beat the oracle's self-comparison on every other line failed it
    # by one ray, 0.401 against 0.422 - 0.02)
    sig = lambda q: 3.0 * np.sqrt(max(q * (1.0 - q), 1e-4) / n)
    assert close_g >= close_p - max(0.02, sig(close_p)), msg
    assert np.mean(stop == ostop) >= np.mean(pstop == ostop) - max(0.02, sig(np.mean(pstop == ostop))), msg


def ks_distance(a, b):
    a, b = np.sort(a), np.sort(b)
    x = np.concatenate([a, b])
    return float(np.max(np.abs(np.searchsorted(a, x, side="right") / len(a) - np.searchsorted(b, x, side="right") / len(b))))


def test_distribution_config1_ngo(cfgfiles, oracle_models):
    """BASELINE config[1] (Ngo, adaptive RK45, maxsteps 512, outputper 8), first 10 k rays of its launch set."""
    from stanford_raytracer_amd import api

    pos, d, w = wl.launch_set(100_000, 2)
    pos, d, w = pos[:10_000], d[:10_000], w[:10_000]
    kw = dict(fixedstep=0, dt0=1e-3, dtmax=0.1, maxerr=5e-4, tmax=0.5, maxsteps=512, del_=1e-4, minalt=wl.MINALT)
    distribution_check("config[1] ngo", api.Model.ngo(cfgfiles["ngo"]), oracle_models["ngo"], pos, d, w, kw, 8)


def test_distribution_config2_interp():
    """BASELINE config[2] (interp model, adaptive RK45, maxsteps 256, outputper 16), first 10 k rays of its launch set, on a
    64^3 grid of the same plasmasphere over the same +-10 R_E cube (the oracle needs 4.3 GB and minutes for 256^3; the
    kernel and its data path are the same, tests/test_gpu_trace.py runs the 256^3 grid at full size for its properties)."""
    from oracle import oracle
    from stanford_raytracer_amd import api

    F, b = wl.make_grid(64, half_width=10.0 * wl.R_E)
    g, o = api.Model.interp(F, b, wl.QS, wl.MS), oracle.Model.interp(F, b, wl.QS, wl.MS)
    pos, d, w = wl.launch_set(1_000_000, 3)
    pos, d, w = pos[:10_000], d[:10_000], w[:10_000]
    kw = dict(fixedstep=0, dt0=1e-3, dtmax=0.1, maxerr=5e-4, tmax=0.5, maxsteps=256, del_=1e-6, minalt=wl.MINALT)
    distribution_check("config[2] interp", g, o, pos, d, w, kw, 16)


def test_distribution_config4_scattered(tmp_path):
    """BASELINE config[4] (scattered model, maxsteps 64, outputper 8) on a 60 k-sample set of the same plasmasphere over
    the same cube (the oracle's kd-tree set-up for 825 k samples alone takes minutes), first 1 000 rays of its launch set."""
    from oracle import oracle
    from stanford_raytracer_amd import api

    pts, lnN = wl.make_points(58_000, 2_000, 5, half_width=10.0 * wl.R_E)
    pf = str(tmp_path / "pts60k.txt")
    wl.write_points_file(pf, pts, lnN, np.array([-10.0 * wl.R_E, 10.0 * wl.R_E] * 3))
    g = api.Model.scattered_file(pf, window_scale=1.5, order=2, exact=0, local_window_scale=5.0)
    o = oracle.Model.scattered_file(pf, window_scale=1.5, order=2, exact=0, local_window_scale=5.0, perm_seed=2 | 0x80000000)
    pos, d, w = wl.launch_set(1_000_000, 5)
    pos, d, w = pos[:1000], d[:1000], w[:1000]
    kw = dict(fixedstep=0, dt0=1e-3, dtmax=0.1, maxerr=5e-4, tmax=0.5, maxsteps=64, del_=1e-6, minalt=wl.MINALT)
    distribution_check("config[4] scattered", g, o, pos, d, w, kw, 8)
