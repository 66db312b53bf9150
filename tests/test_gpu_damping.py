"""GPU (-m gpu): hot-plasma damping along the kept rows -- the reference's MATLAB post-processor
(matlab/damping/) as a wave-per-row kernel over the trace kernel's row buffer, against the CPU oracle's restatement of
the same scripts on the same rows."""
import numpy as np
import pytest

from oracle import oracle
from stanford_raytracer_amd import api, workloads as wl

pytestmark = pytest.mark.gpu
Q = 1.60217646e-19


@pytest.fixture(scope="module")
def traced(gpu_models):
    g = gpu_models["ngo"]
    pos0, dir0, w0 = wl.launch_set(160, 21)
    p = api.make_params(dt0=1e-3, dtmax=0.02, tmax=0.3, maxerr=5e-4, maxsteps=400, del_=1e-4, minalt=wl.MINALT, outputper=8)
    rows, nrows, stop, _ = g.trace(pos0, dir0, w0, params=p)
    return g, rows, nrows, w0, 8


def compare(gk, gm, gf, ok_, om, of, kept):
    n_rows = bad_flag = 0
    rel = []
    for r in range(gk.shape[0]):
        k = kept[r]
        assert gm[r, 0] == 1.0 and gk[r, 0] == 0.0 and np.all(gm[r, k:] == 0.0)
        for i in range(1, k):
            n_rows += 1
            if gf[r, i] != of[r, i]:
                bad_flag += 1
                continue
            if of[r, i] >= 2:
                continue
            d = abs(gk[r, i] - ok_[r, i]) / max(abs(ok_[r, i]), 1e-300)
            rel.append(d)
    rel = np.array(rel)
    return n_rows, bad_flag, rel


@pytest.mark.parametrize("kw", [dict(), dict(dist=1, m=(0,), Ne_h=0.1e6, kT=5e3 * Q),
                                dict(dist=1, m=(-1, 0, 1), Ne_h=0.1e6, kT=2e3 * Q, mode=1)])
def test_rates_and_magnitudes_match_the_oracle(traced, kw):
    g, rows, nrows, w0, per = traced
    gk, gm, gf = api.damping(g.species(), per, rows, nrows, w0, **kw)
    ok_, om, of = oracle.damping(*g.species(), per, rows, nrows, w0, **kw)
    kept = (nrows - 1) // per + 1
    n_rows, bad_flag, rel = compare(gk, gm, gf, ok_, om, of, kept)
    assert n_rows > 300
    assert bad_flag <= 0.02 * n_rows           # a panel decision at its threshold may differ (libm vs device Bessel/exp)
    # Same panels, same sums: the two agree far inside the quadrature's own tolerance (1e-3).  The integrand takes
    # central differences of the distribution with a 1e-8 relative step, which amplifies last-bit differences of
    # pow/exp/sqrt by 1e8 -- hence 1e-6, not 1e-14.
    assert np.median(rel) <= 1e-6
    assert np.percentile(rel, 99) <= 2e-3 and rel.max() <= 5e-2
    same = (gf == of) & (of < 2)
    fin = np.isfinite(om) & np.isfinite(gm) & (np.cumsum(~same, axis=1) == 0)
    assert np.allclose(gm[fin], om[fin], rtol=2e-3, atol=1e-12)


DEVICE_ENTRY = r"""
import ctypes as C, sys
import numpy as np
import torch                       # first: torch brings its own HIP runtime and must initialise it before ours
torch.cuda.init()
from stanford_raytracer_amd import api
d = np.load(sys.argv[1])
rows, nrows, w0, qs, ms, per = d["rows"], d["nrows"], d["w0"], d["qs"], d["ms"], int(d["per"])
want = api.damping((qs, ms), per, rows, nrows, w0)
dev = torch.device("cuda:0")
d_rows, d_nrows, d_w0 = (torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (rows, nrows, w0))
nr, slots = rows.shape[:2]
d_rate = torch.empty((nr, slots), dtype=torch.float64, device=dev)
d_mag = torch.empty_like(d_rate)
d_flag = torch.empty((nr, slots), dtype=torch.int32, device=dev)
p = api.damping_params()
st = torch.cuda.Stream()
torch.cuda.synchronize()
rc = api.lib().srt_damping_device(C.byref(p), 4, api._dp(api._f64(qs)), api._dp(api._f64(ms)), slots, per, nr,
                                  d_rows.data_ptr(), d_nrows.data_ptr(), d_w0.data_ptr(), d_rate.data_ptr(),
                                  d_mag.data_ptr(), d_flag.data_ptr(), st.cuda_stream)
assert rc == 0
st.synchronize()
assert np.array_equal(d_rate.cpu().numpy(), want[0], equal_nan=True)
assert np.array_equal(d_mag.cpu().numpy(), want[1], equal_nan=True)
assert np.array_equal(d_flag.cpu().numpy(), want[2])
print("DEVICE_ENTRY_OK")
"""


def test_device_entry_on_resident_rows(traced, tmp_path):
    """srt_damping_device on torch-owned device buffers and a torch stream == the host-buffer entry (own process:
    torch's HIP runtime has to come up before the library's)."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    g, rows, nrows, w0, per = traced
    qs, ms = g.species()
    f = str(tmp_path / "rows.npz")
    np.savez(f, rows=rows, nrows=nrows, w0=w0, qs=qs, ms=ms, per=per)
    out = subprocess.run([sys.executable, "-c", DEVICE_ENTRY, f], cwd=ROOT, capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, PYTHONPATH=ROOT))
    assert out.returncode == 0 and "DEVICE_ENTRY_OK" in out.stdout, out.stderr[-2000:]


def test_bad_arguments(traced):
    g, rows, nrows, w0, per = traced
    with pytest.raises(api.SrtError):
        api.damping(g.species(), per, rows, nrows, w0, dist=2)
    with pytest.raises(api.SrtError):
        api.damping(g.species(), per, rows, nrows, w0, dist=1, kT=0.0)
    with pytest.raises(api.SrtError):
        api.damping(g.species(), 0, rows, nrows, w0)


def test_rays_without_rows_and_empty_batches(traced):
    """A ray with a single row has nothing to damp (magnitude 1 in slot 0, 0 beyond); an empty batch is a no-op."""
    g, rows, nrows, w0, per = traced
    one = nrows.copy()
    one[:5] = 1
    k, m, f = api.damping(g.species(), per, rows, one, w0)
    assert np.all(k[:5] == 0) and np.all(m[:5, 0] == 1.0) and np.all(m[:5, 1:] == 0.0) and np.all(f[:5] == 0)
    k0, m0, f0 = api.damping(g.species(), per, rows[:0], nrows[:0], w0[:0])
    assert k0.shape == (0, rows.shape[1])
