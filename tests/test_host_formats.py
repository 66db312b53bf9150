"""CPU: file formats of the drop-in boundary (pure host code of libsrt_hip.so and the workload writers)."""
import os

import numpy as np
import pytest

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


def test_ray_file_writer_adaptive_golden_threads_and_append(oracle_models, tmp_path, monkeypatch):
    """The writer formats disjoint ray ranges on all host cores and pwrites the fixed-length records in place.  Fed the
    oracle's rows (bit-identical to the reference's on this run), it must give the reference DRIVER's own adaptive model-3
    file byte for byte -- for any thread count, and when the file is built by appending chunks as the CLI does."""
    ref = open(os.path.join(GOLDEN_DIR, "driver_interp_adaptive.ray"), "rb").read()
    p0, d0, w0 = wl.appendix_b_rays()
    rows_all, nrows, stop, _ = oracle_models["interp"].trace(p0, d0, w0, capacity=2000, fixedstep=0, dt0=1e-3, dtmax=0.1, tmax=0.2,
                                                             maxerr=5e-4, maxsteps=2000, minalt=wl.MINALT, del_=1e-6)
    p = api.make_params(maxsteps=2000, outputper=16)
    rows = np.ascontiguousarray(rows_all[:, ::16])
    assert rows.shape[1] == api.lib().srt_rows_per_ray(__import__("ctypes").byref(p))
    for nth in ("1", "3", "16", "64"):      # more threads than rays included
        monkeypatch.setenv("SRT_IO_THREADS", nth)
        out = tmp_path / ("t%s.ray" % nth)
        api.write_ray_file(str(out), (4, wl.QS, wl.MS), p, w0, rows, nrows, stop)
        assert out.read_bytes() == ref, nth
    monkeypatch.setenv("SRT_IO_THREADS", "4")
    out = tmp_path / "chunks.ray"
    for lo, hi in ((0, 5), (5, 6), (6, 16)):
        api.write_ray_file(str(out), (4, wl.QS, wl.MS), p, w0[lo:hi], rows[lo:hi], nrows[lo:hi], stop[lo:hi], raynum0=lo + 1,
                           append=lo > 0)
    assert out.read_bytes() == ref
    # no rays: an empty file, as the reference's open(status="replace") leaves it
    api.write_ray_file(str(out), (4, wl.QS, wl.MS), p, w0[:0], rows[:0], nrows[:0], stop[:0])
    assert out.read_bytes() == b""


def test_es24_matches_printf_on_many_values(tmp_path):
    """es24.15e3 = 16 significant digits, correctly rounded, three exponent digits: against Python's own %.15E on
    320 k values over the whole double range (incl. subnormals, halfway cases, the largest double)."""
    rng = np.random.default_rng(0)
    vals = np.concatenate([rng.normal(size=20000 * 16) * 10.0 ** rng.integers(-308, 308, size=20000 * 16),
                           [0.0, -0.0, 5e-324, 1.7976931348623157e308, 0.1, 0.5, 1.0000000000000005, 9.9999999999999995e22, 1e23,
                            2.5e-5, 1.5, 2.5, 3.5e15, 2.2250738585072014e-308, -4.9e-324, 1e-310]])
    R = np.zeros((len(vals) // 16, 1, 20))
    R[:, 0, :16] = vals.reshape(-1, 16)
    p = api.make_params(maxsteps=1, outputper=1)
    out = tmp_path / "v.ray"
    n = len(R)
    api.write_ray_file(str(out), (4, wl.QS, wl.MS), p, np.ones(n), R, np.ones(n, dtype=np.int32), np.zeros(n, dtype=np.int32))

    def es24(v):
        m, e = ("%.15E" % v).split("E")
        return ("%sE%s%03d" % (m, e[0], abs(int(e)))).rjust(24)

    for i, ln in enumerate(open(out)):
        assert len(ln) == 823
        got = [ln[20 + 24 * c:44 + 24 * c] for c in range(16)]
        assert got == [es24(v) for v in R[i, 0, :16]], i


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


# ---- model-3 grid files: the step before the path ------------------------------------------------
def _small_grid(with_derivs):
    rng = np.random.default_rng(11)
    F = rng.normal(20.0, 3.0, (5, 6, 7, 4))      # [nz, ny, nx, nspec]
    b = np.array([-3.0e7, 3.0e7, -2.0e7, 2.0e7, -1.0e7, 1.5e7])
    d = [rng.normal(0.0, 1e-6, F.shape) for _ in range(7)] if with_derivs else None
    return F, b, d


def test_grid_text_reader_matches_the_writers(tmp_path):
    """The parallel text reader returns exactly what the workload writer (free-format %.17g, one node per record)
    put on disk, and the library's own text writer produces the reference's layout: (5i10), es24.15e3 header
    records, one VALUE per record in es24.15e3 (16 significant digits)."""
    for with_derivs in (False, True):
        F, b, d = _small_grid(with_derivs)
        p1, p2 = str(tmp_path / "a.txt"), str(tmp_path / "b.txt")
        wl.write_grid_file(p1, F, b, derivs=d)
        api.write_grid_file(p2, F, b, wl.QS, wl.MS, derivs=d)
        lines = open(p2).read().split("\n")
        assert lines[0] == "%10d%10d%10d%10d%10d" % (1 if with_derivs else 0, 4, 7, 6, 5)
        assert len(lines[1]) == 6 * 24 and len(lines[2]) == 4 * 24 and all(len(x) == 24 for x in lines[4:-1])
        assert len(lines) - 5 == F.size * (8 if with_derivs else 1)
        assert lines[4] == "%24s" % ("%.15E" % F.flat[0]).replace("E+", "E+0").replace("E-", "E-0")
        for path, exact in ((p1, True), (p2, False)):
            assert not api.grid_file_is_binary(path)
            g = api.read_grid_file(path)
            same = np.array_equal if exact else (lambda x, y: np.allclose(x, y, rtol=6e-16, atol=0))
            assert same(g["F"], F) and same(g["bounds"], b)
            assert same(g["qs"], wl.QS) and same(g["ms"], wl.MS)
            if with_derivs:
                assert all(same(x, y) for x, y in zip(g["derivs"], d))
            else:
                assert g["derivs"] is None


def test_grid_text_reader_record_semantics_and_errors(tmp_path):
    """Fortran 'd' exponents and commas parse; a record with extra fields falls back to the Fortran's record
    semantics (read(..,*) takes nspec values and skips the rest of the record); truncated files are refused."""
    F, b, _ = _small_grid(False)
    lines = open(_write(tmp_path, F, b)).read().splitlines()
    odd = tmp_path / "odd.txt"
    body = [ln.replace("e+", "d+").replace("e-", "d-") for ln in lines[4:]]
    body[3] = body[3].replace(" ", ",") + " 99.0 98.0"       # extra fields on one node's record
    odd.write_text("\n".join(lines[:4] + body) + "\n")
    g = api.read_grid_file(str(odd))
    assert np.array_equal(g["F"], F)
    cut = tmp_path / "cut.txt"
    cut.write_text("\n".join(lines[:50]) + "\n")
    try:
        api.read_grid_file(str(cut))
        assert False, "truncated grid accepted"
    except api.SrtError as e:
        assert "truncated" in str(e)


def _write(tmp_path, F, b):
    p = str(tmp_path / "g.txt")
    wl.write_grid_file(p, F, b)
    return p


def test_grid_binary_side_format_round_trip(tmp_path):
    for with_derivs in (False, True):
        F, b, d = _small_grid(with_derivs)
        txt, bin1, bin2 = str(tmp_path / "g.txt"), str(tmp_path / "g1.bin"), str(tmp_path / "g2.bin")
        wl.write_grid_file(txt, F, b, derivs=d)
        api.convert_grid_file(txt, bin1)                       # text -> binary
        api.write_grid_file(bin2, F, b, wl.QS, wl.MS, derivs=d, binary=True)
        assert open(bin1, "rb").read() == open(bin2, "rb").read()
        assert api.grid_file_is_binary(bin1)
        assert os.path.getsize(bin1) == 144 + F.size * 8 * (8 if with_derivs else 1)
        g = api.read_grid_file(bin1)
        assert np.array_equal(g["F"], F) and np.array_equal(g["bounds"], b) and np.array_equal(g["ms"], wl.MS)
        assert (g["derivs"] is None) == (not with_derivs)
        if with_derivs:
            assert all(np.array_equal(x, y) for x, y in zip(g["derivs"], d))
    trunc = tmp_path / "t.bin"
    trunc.write_bytes(open(bin1, "rb").read()[:1000])
    try:
        api.read_grid_file(str(trunc))
        assert False, "truncated binary grid accepted"
    except api.SrtError as e:
        assert "truncated" in str(e)


def test_points_file_text_layout_and_binary_side_format(tmp_path):
    """Model-4 sample files: the reference builder's record layout (one es24.15e3 field per value,
    gcpm_dens_model_buildgrid_random.f95:196-208 + helper module :37-43), readable by the oracle's reader; the binary
    side-format holds the same numbers bit for bit; conversion of a text file gives the same binary."""
    from oracle import oracle
    rng = np.random.default_rng(4)
    n = 300
    rec = np.concatenate([rng.normal(size=(n, 3)) * 3 * wl.R_E, rng.uniform(10, 25, size=(n, 4))], axis=1)
    b = np.array([-9.0, 9.0] * 3) * wl.R_E
    txt, binf, conv = (str(tmp_path / f) for f in ("p.txt", "p.bin", "c.bin"))
    api.write_points_file(txt, rec, b, wl.QS, wl.MS)
    api.write_points_file(binf, rec, b, wl.QS, wl.MS, binary=True)
    lines = open(txt).read().split("\n")
    assert lines[0] == "%10d" % 4 and len(lines[1]) == 6 * 24 and len(lines[2]) == 4 * 24 and len(lines[4]) == 7 * 24
    assert lines[4][:24] == "%24s" % ("%.15E" % rec[0, 0]).replace("E+", "E+0").replace("E-", "E-0")
    vals = np.array([[float(lines[4 + i][24 * k:24 * k + 24]) for k in range(7)] for i in range(n)])
    assert np.allclose(vals, rec, rtol=2e-15, atol=0)                    # 16 significant digits
    assert not api.points_file_is_binary(txt) and api.points_file_is_binary(binf)
    raw = np.fromfile(binf, dtype=np.float64, offset=136)
    assert np.array_equal(raw.reshape(n, 7), rec)
    api.convert_points_file(txt, conv)
    assert np.array_equal(np.fromfile(conv, dtype=np.float64, offset=136).reshape(n, 7), vals)
    # the oracle's reader (the adapter's READ statements restated) takes the text file
    o = oracle.Model.scattered_file(txt, order=1)
    assert o.nspec == 4


def test_ray_file_reader(tmp_path, monkeypatch):
    """srt_read_ray_file (the post-processor's way into an existing .ray file, matlab/readrayoutput.m's role): the reference
    DRIVER's own files against a plain Python parse of the fixed-width records; then write -> read round trips exactly
    what 16 significant digits hold, special values and an empty file included."""
    from conftest import parse_ray_file

    for name in ("driver_interp_adaptive.ray", "config1_outputper25.ray"):
        path = os.path.join(GOLDEN_DIR, name)
        ref = parse_ray_file(path)
        r = api.read_ray_file(path)
        assert r["nspec"] == 4 and len(r["rows"]) == len(ref) and r["kept"].sum() == len(ref)
        assert np.array_equal(r["raynum"], np.arange(1, 17)) and np.array_equal(r["qs"], ref[0, 20:24]) and np.array_equal(r["ms"], ref[0, 24:28])
        assert np.array_equal(r["rows"][:, 0:16], ref[:, 2:18]) and np.array_equal(r["rows"][:, 16:20], ref[:, 28:32])
        first = np.concatenate([[0], np.cumsum(r["kept"])[:-1]])
        assert np.array_equal(r["stopcond"], ref[first, 1].astype(np.int32)) and np.array_equal(r["w0"], ref[first, 18])
        rows, nrows = api.padded_rows(r)
        assert rows.shape == (16, r["kept"].max(), 20) and np.array_equal(rows[3, :nrows[3]], r["rows"][first[3]:first[3] + nrows[3]])
    # round trip through the writer
    rng = np.random.default_rng(3)
    n, slots = 40, 6
    p = api.make_params(maxsteps=6, outputper=1)
    rows = rng.normal(size=(n, slots, 20)) * 10.0 ** rng.integers(-20, 20, size=(n, slots, 20))
    rows[0, 0, 5], rows[1, 1, 6], rows[2, 0, 0] = np.nan, -np.inf, -0.0
    nrows = rng.integers(1, 7, size=n).astype(np.int32)
    stop = rng.integers(0, 7, size=n).astype(np.int32)
    w0 = rng.uniform(1e3, 1e5, n)
    out = tmp_path / "rt.ray"
    api.write_ray_file(str(out), (4, wl.QS, wl.MS), p, w0, rows, nrows, stop, raynum0=7)
    r = api.read_ray_file(str(out))
    assert np.array_equal(r["raynum"], np.arange(7, 7 + n)) and np.array_equal(r["kept"], nrows) and np.array_equal(r["stopcond"], stop)
    back, _ = api.padded_rows(r)
    mask = np.arange(slots)[None, :] < nrows[:, None]
    a, b = back[mask], rows[mask]
    fin = np.isfinite(b)
    assert np.allclose(a[fin], b[fin], rtol=6e-16, atol=0) and np.array_equal(np.isnan(a), np.isnan(b))
    assert np.array_equal(a[~fin & ~np.isnan(b)], b[~fin & ~np.isnan(b)])
    assert np.allclose(r["w0"], w0, rtol=6e-16, atol=0) and np.allclose(r["qs"], wl.QS, rtol=6e-16) and np.allclose(r["ms"], wl.MS, rtol=6e-16)
    api.write_ray_file(str(out), (4, wl.QS, wl.MS), p, w0[:0], rows[:0], nrows[:0], stop[:0])
    e = api.read_ray_file(str(out))
    assert len(e["raynum"]) == 0 and e["rows"].shape == (0, 20)
    bad = tmp_path / "bad.ray"
    bad.write_text(open(os.path.join(GOLDEN_DIR, "config1_outputper25.ray")).read()[:-400])     # a truncated last record
    with pytest.raises(api.SrtError):
        api.read_ray_file(str(bad))


def test_ray_file_edge_cases_of_the_record_head(tmp_path):
    """The i10 ray-number field cannot hold more than ten digits (Fortran prints asterisks): refused, not shifted; a file that
    was appended to and repeats ray numbers keeps its rays apart (a ray's first record is its row 0, t = 0)."""
    from stanford_raytracer_amd import api, workloads as wl

    p = api.make_params(maxsteps=3, outputper=1)
    rows = np.zeros((2, 3, 20))
    rows[:, 1, 0], rows[:, 2, 0] = 1e-3, 2e-3        # t = 0, 1e-3, 2e-3
    rows[:, :, 1:] = 1.5
    nrows, stop, w0 = np.array([3, 2], dtype=np.int32), np.array([0, 6], dtype=np.int32), np.array([1e4, 2e4])
    out = tmp_path / "big.ray"
    with pytest.raises(api.SrtError, match="i10"):
        api.write_ray_file(str(out), (4, wl.QS, wl.MS), p, w0, rows, nrows, stop, raynum0=9_999_999_999)
    api.write_ray_file(str(out), (4, wl.QS, wl.MS), p, w0, rows, nrows, stop, raynum0=9_999_999_998)   # ..98 and ..99 still fit
    r = api.read_ray_file(str(out))
    assert r["raynum"].tolist() == [9_999_999_998, 9_999_999_999] and r["kept"].tolist() == [3, 2] and r["stopcond"].tolist() == [0, 6]
    # the same two rays appended under the SAME numbers: four rays, not two
    ap = tmp_path / "appended.ray"
    api.write_ray_file(str(ap), (4, wl.QS, wl.MS), p, w0, rows, nrows, stop, raynum0=1)
    api.write_ray_file(str(ap), (4, wl.QS, wl.MS), p, w0[:1], rows[:1], nrows[:1], stop[:1], raynum0=2, append=True)
    r = api.read_ray_file(str(ap))
    assert r["raynum"].tolist() == [1, 2, 2] and r["kept"].tolist() == [3, 2, 3] and r["w0"].tolist() == [1e4, 2e4, 1e4]


def test_writers_reproduce_the_reference_builders_files_byte_for_byte(tmp_path):
    """What `raytracer --buildgrid=1` / `--buildsamples=1` write (srt_grid_file_write, srt_points_file_write) against files the
    reference's OWN producers wrote (tests/golden/make_builder_layout_golden.py: gcpm_dens_model_buildgrid with --compder=1 on
    3 x 4 x 5 nodes, gcpm_dens_model_buildgrid_random with 60 samples): read by the library's readers, written back by its writers,
    the bytes must be the same -- header records, field widths, exponent digits, one value per record, block order."""
    ref = os.path.join(GOLDEN_DIR, "builder_grid_3x4x5_compder1.txt")
    g = api.read_grid_file(ref)
    assert g["F"].shape == (5, 4, 3, 4) and g["derivs"] is not None and len(g["derivs"]) == 7
    out = str(tmp_path / "grid.txt")
    api.write_grid_file(out, g["F"], g["bounds"], g["qs"], g["ms"], derivs=g["derivs"])
    assert open(out, "rb").read() == open(ref, "rb").read()
    # ... and the same file with compder = 0 is the reference's first 4 + 240 records
    api.write_grid_file(out, g["F"], g["bounds"], g["qs"], g["ms"])
    want = open(ref).read().split("\n")[:4 + 240]
    want[0] = "%10d" % 0 + want[0][10:]
    assert open(out).read() == "\n".join(want) + "\n"
    ref = os.path.join(GOLDEN_DIR, "builder_samples_60.txt")
    tok = open(ref).read().split()
    rec = np.array([float(t) for t in tok[15:]]).reshape(-1, 7)
    assert len(rec) == 60
    out = str(tmp_path / "pts.txt")
    api.write_points_file(out, rec, np.array([float(t) for t in tok[1:7]]), np.array([float(t) for t in tok[7:11]]),
                          np.array([float(t) for t in tok[11:15]]))
    assert open(out, "rb").read() == open(ref, "rb").read()
