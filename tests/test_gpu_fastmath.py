"""GPU (-m gpu): the scattered model's own exp / ln / cos (csrc/srt_fastmath.hpp) against numpy's libm."""
import numpy as np
import pytest

from stanford_raytracer_amd import api

pytestmark = pytest.mark.gpu


def ulps(a, b):
    return np.abs(a - b) / np.spacing(np.abs(b))


def test_exp_ln_cos_within_two_ulp():
    api.init(0)
    rng = np.random.default_rng(5)
    t = np.concatenate([rng.uniform(-745, 700, 200000), rng.uniform(-3, 3, 200000), [0.0, -0.0, -1e-300, 709.0]])
    e = api.selftest_math(0, t)
    ref = np.exp(t)
    ok = ref > 1e-300                                    # normal results: relative accuracy
    assert ulps(e[ok], ref[ok]).max() <= 2.0
    assert np.all(np.abs(e[~ok] - ref[~ok]) <= 1e-300) and api.selftest_math(0, [-5000.0])[0] == 0.0
    x = np.concatenate([np.exp(rng.uniform(-700, 700, 200000)), rng.uniform(0.5, 2.0, 200000), [1.0, 5e-324, 1e-310, 1.7e308]])
    l = api.selftest_math(1, x)
    ref = np.log(x)
    assert l[len(x) - 4] == 0.0
    nz = ref != 0
    assert ulps(l[nz], ref[nz]).max() <= 2.0
    c = np.concatenate([rng.uniform(0, np.pi, 400000), [0.0, np.pi / 2, np.pi, np.pi * (1 + 1e-12)]])
    v = api.selftest_math(2, c)
    ref = np.cos(c)
    assert np.abs(v - ref).max() <= 2.3e-16                # one ulp of 1: what the window 0.5 + 0.5 cos sees
    big = np.abs(ref) > 1e-3
    assert ulps(v[big], ref[big]).max() <= 2.0
