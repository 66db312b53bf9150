"""GPU (-m gpu): the raytracer-compatible CLI and the Fortran-callable shim reproduce BASELINE config[0]
(16 rays, Ngo + dipole B, fixed RK4) -- the reference's own CPU-runnable case -- in the reference's .ray format."""
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN_DIR, ROOT, parse_ray_file, vrel
from stanford_raytracer_amd import workloads as wl

pytestmark = pytest.mark.gpu
BIN = os.path.join(ROOT, "stanford_raytracer_amd", "bin")


def test_cli_config1_matches_reference_ray_file(tmp_path, cfgfiles, golden):
    rays = golden["g4_rays"]
    rf = tmp_path / "rays.txt"
    wl.write_rays_file(str(rf), rays[:, :3], rays[:, 3:6], rays[:, 6])
    out = tmp_path / "out.ray"
    cmd = [os.path.join(BIN, "raytracer"), "--outputper=25", "--dt0=0.001", "--dtmax=0.1", "--tmax=0.1", "--root=2",
           "--fixedstep=1", "--maxerr=5e-4", "--maxsteps=2000", "--minalt=%r" % wl.MINALT,
           "--inputraysfile=%s" % rf, "--outputfile=%s" % out, "--modelnum=1", "--ngo_configfile=%s" % cfgfiles["ngo"],
           "--yearday=2010001", "--milliseconds_day=0", "--use_tsyganenko=0", "--use_igrf=0", "--tsyganenko_Pdyn=4"]
    subprocess.run(cmd, check=True)
    mine = parse_ray_file(str(out))
    ref = parse_ray_file(os.path.join(GOLDEN_DIR, "config1_outputper25.ray"))
    assert mine.shape == ref.shape
    assert np.array_equal(mine[:, 0:2], ref[:, 0:2])            # raynum, stopcond
    assert np.array_equal(mine[:, 2], ref[:, 2])                # time grid (fixed step)
    assert np.array_equal(mine[:, 18:20], ref[:, 18:20])        # w, nspec
    assert np.array_equal(mine[:, 20:28], ref[:, 20:28]) and np.array_equal(mine[:, 32:36], ref[:, 32:36])  # qs ms nus
    r0 = ref[:, 2] == 0
    assert np.array_equal(mine[r0, 3:6], ref[r0, 3:6])
    assert vrel(mine[:, 3:6], ref[:, 3:6]).max() <= 1e-3         # row 100 of the ladder x10 (4e-5 -> 4e-4) with margin
    # rows 0 and 25 only: the survey ladder (reference vs its own FMA rebuild) gives 7e-8 after 10 steps and 4e-5
    # after 100; single rays sit near 1e-5 at step 25 and move with any 1-ulp change.  This test is about the CLI and
    # the .ray format; the parity bars proper (against the oracle's own sensitivity) are in test_gpu_trace.py
    assert vrel(mine[ref[:, 2] <= 0.0251, 3:6], ref[ref[:, 2] <= 0.0251, 3:6]).max() <= 4e-5
    assert np.max(np.abs(mine[:, 28:32] - ref[:, 28:32]) / ref[:, 28:32]) <= 1e-2  # Ns along the diverging paths

    # the Fortran program over the bind(C) shim writes the same file through the same library
    fdrv = os.path.join(BIN, "srt_fortran_driver")
    if os.path.exists(fdrv):
        out2 = tmp_path / "out_fortran.ray"
        subprocess.run([fdrv, cfgfiles["ngo"], str(rf), str(out2)], check=True)
        assert out2.read_text() == out.read_text()


def test_cli_adaptive_model3_matches_the_reference_drivers_file(tmp_path, grid16):
    """An adaptive modelnum-3 run against the .ray file the reference's OWN program wrote (oracle/_ref/raytracer =
    fortran/raytracer_driver.f95 compiled where it lies; tests/golden/make_driver_golden.py): same flags, same input
    files.  Every integer / format column and every row 0 must be equal; the adaptive rows are compared per ray over
    the records both files hold (step sequences may split after a few steps: SURVEY A-9)."""
    F, b, qs, ms = grid16
    gf = tmp_path / "grid16.txt"
    wl.write_grid_file(str(gf), F, b, qs, ms)
    p0, d0, w0 = wl.appendix_b_rays()
    rf = tmp_path / "rays.txt"
    wl.write_rays_file(str(rf), p0, d0, w0)
    out = tmp_path / "out.ray"
    cmd = [os.path.join(BIN, "raytracer"), "--outputper=16", "--tmax=0.2", "--fixedstep=0", "--modelnum=3",
           "--first_attempt_policy=0",   # the golden comes from a flang build of the reference
           "--interp_interpfile=%s" % gf, "--dt0=0.001", "--dtmax=0.1", "--root=2", "--maxerr=5e-4", "--maxsteps=2000",
           "--minalt=%r" % wl.MINALT, "--inputraysfile=%s" % rf, "--outputfile=%s" % out, "--yearday=2010001",
           "--milliseconds_day=0", "--use_tsyganenko=0", "--use_igrf=0"]
    subprocess.run(cmd, check=True)
    mine = parse_ray_file(str(out))
    ref = parse_ray_file(os.path.join(GOLDEN_DIR, "driver_interp_adaptive.ray"))
    assert set(mine[:, 0]) == set(ref[:, 0]) == set(range(1, 17))
    stop_agree, rows_mine, rows_ref = 0, 0, 0
    for ray in range(1, 17):
        a, r = mine[mine[:, 0] == ray], ref[ref[:, 0] == ray]
        rows_mine += len(a)
        rows_ref += len(r)
        stop_agree += int(a[0, 1] == r[0, 1])
        # row 0: the launch point, k0 on the whistler root, plasma and field there
        assert a[0, 2] == r[0, 2] == 0.0 and np.array_equal(a[0, 3:6], r[0, 3:6])
        assert vrel(a[0, 12:15], r[0, 12:15]) <= 1e-9                      # n
        assert vrel(a[0, 15:18], r[0, 15:18]) <= 2e-7                      # B0 (float32 round trip)
        assert np.max(np.abs(a[0, 28:32] - r[0, 28:32]) / r[0, 28:32]) <= 1e-11   # Ns
        assert vrel(a[0, 6:9], r[0, 6:9]) <= 1e-6 and vrel(a[0, 9:12], r[0, 9:12]) <= 1e-6   # vprel, vgrel (finite differences)
        # constants of the record
        assert np.array_equal(a[:, 18], np.full(len(a), r[0, 18])) and np.all(a[:, 19] == 4)
        assert np.array_equal(a[:, 20:28], np.tile(r[0, 20:28], (len(a), 1))) and np.all(a[:, 32:36] == 0.0)
        assert np.all(a[:, 1] == a[0, 1])                                   # the final stop code on every row
        # first kept row after the launch (16 accepted steps): same time stamp unless the step sequences already split
        # (bar: SURVEY A-9's curve tolerance; the oracle-yardstick trajectory bars are in tests/test_gpu_trajectory_stats.py)
        if len(a) > 1 and len(r) > 1 and a[1, 2] == r[1, 2]:
            assert vrel(a[1, 3:6], r[1, 3:6]) <= 1e-3
    assert stop_agree >= 14
    assert abs(rows_mine - rows_ref) <= 0.1 * rows_ref
    # rays whose step sequences do not split: the WHOLE time-stamp column is equal (every kept row), and so are the record
    # counts; positions then differ only by what 16 .. 2000 steps amplify
    whole, first = 0, 0
    for ray in range(1, 17):
        a, r = mine[mine[:, 0] == ray], ref[ref[:, 0] == ray]
        n = min(len(a), len(r))
        same = int(np.argmax(a[:n, 2] != r[:n, 2])) if np.any(a[:n, 2] != r[:n, 2]) else n   # kept rows before the split
        first += int(same >= 2)
        if same == n and len(a) == len(r):
            whole += 1
            assert np.array_equal(a[:, 2], r[:, 2]) and a[0, 1] == r[0, 1]
            assert vrel(a[1:, 3:6], r[1:, 3:6]).max() <= 1e-3
        elif same >= 2:  # up to the split the curves coincide (same knots): SURVEY A-9's ladder at <= 100 steps
            assert vrel(a[1:same, 3:6], r[1:same, 3:6]).max() <= 1e-3
    # (no floor on these counts: with outputper = 16 the first kept row already lies 16 adaptive steps after the launch, where
    # the interp model's step sequences have split for most rays -- measured: 1 of 16 keeps it; the time-grid agreement of the
    # first steps is asserted at outputper = 1, on 1 024 rays, in tests/test_gpu_trajectory_stats.py)
    print("driver golden: %d of 16 rays keep the reference's whole time grid, %d its first kept row" % (whole, first))


def test_cli_devices_flag_shards_and_keeps_ray_order(tmp_path, cfgfiles):
    """--devices=0,0 (two host threads, two model replicas, contiguous shards -- both on the one card of the GPU box): the
    .ray file is byte-identical to the single-device run; so is a run with more devices than rays."""
    pos, d, w = wl.launch_set(37, 41)
    rf = tmp_path / "rays.txt"
    wl.write_rays_file(str(rf), pos, d, w)
    base = [os.path.join(BIN, "raytracer"), "--outputper=3", "--dt0=0.001", "--dtmax=0.1", "--tmax=0.05", "--root=2",
            "--fixedstep=0", "--maxerr=5e-4", "--maxsteps=80", "--minalt=%r" % wl.MINALT, "--inputraysfile=%s" % rf,
            "--modelnum=1", "--ngo_configfile=%s" % cfgfiles["ngo"], "--yearday=2010001", "--milliseconds_day=0"]
    outs = {}
    for tag, extra in (("one", []), ("two", ["--devices=0,0"]), ("three", ["--devices=0,0,0", "--chunk_rays=64"])):
        o = tmp_path / (tag + ".ray")
        subprocess.run(base + ["--outputfile=%s" % o] + extra, check=True)
        outs[tag] = o.read_bytes()
        assert not list(tmp_path.glob(tag + ".ray.part*"))
    assert len(outs["one"]) > 0 and outs["one"] == outs["two"] == outs["three"]
    one = tmp_path / "single.txt"
    wl.write_rays_file(str(one), pos[:1], d[:1], w[:1])
    o1, o2 = tmp_path / "s1.ray", tmp_path / "s2.ray"
    args = [a for a in base if not a.startswith("--inputraysfile")] + ["--inputraysfile=%s" % one]
    subprocess.run(args + ["--outputfile=%s" % o1], check=True)
    subprocess.run(args + ["--outputfile=%s" % o2, "--devices=0,0"], check=True)   # second shard is empty
    assert o1.read_bytes() == o2.read_bytes()


def test_cli_rejects_out_of_scope_requests(tmp_path, cfgfiles):
    rf = tmp_path / "rays.txt"
    rf.write_text("7e6 0 0 1 0 0 1e4\n")
    base = [os.path.join(BIN, "raytracer"), "--dt0=1e-3", "--tmax=0.01", "--root=2", "--fixedstep=1", "--maxsteps=10",
            "--minalt=6.4712e6", "--inputraysfile=%s" % rf, "--outputfile=%s" % (tmp_path / "o.ray"), "--yearday=2010001",
            "--milliseconds_day=0", "--ngo_configfile=%s" % cfgfiles["ngo"]]
    assert subprocess.run(base + ["--modelnum=1", "--use_tsyganenko=1"]).returncode == 2     # its --tsyganenko_* flags are missing
    assert subprocess.run(base + ["--modelnum=2"]).returncode == 2
    assert subprocess.run(base + ["--modelnum=1"]).returncode == 0


def test_cli_use_igrf(tmp_path, cfgfiles):
    """--use_igrf=1: the .ray rows carry the IGRF field of the library's own evaluation; an unreadable
    coefficient table is an error, not a silent dipole."""
    from stanford_raytracer_amd import api
    rf = tmp_path / "rays.txt"
    rf.write_text("7e6 0 0 1 0 0 1e4\n")
    out = tmp_path / "o.ray"
    base = [os.path.join(BIN, "raytracer"), "--dt0=1e-3", "--tmax=0.01", "--root=2", "--fixedstep=1", "--maxsteps=10",
            "--minalt=6.4712e6", "--inputraysfile=%s" % rf, "--outputfile=%s" % out, "--yearday=2010001",
            "--milliseconds_day=0", "--ngo_configfile=%s" % cfgfiles["ngo"], "--modelnum=1", "--use_igrf=1"]
    assert subprocess.run(base).returncode == 0
    row0 = [float(v) for v in out.read_text().splitlines()[0].split()]
    api.init(0)
    want = api.Model.ngo(cfgfiles["ngo"]).set_field(use_igrf=1).plasma_params([[7e6, 0, 0]])[0, 16:19]
    dip = api.Model.ngo(cfgfiles["ngo"]).plasma_params([[7e6, 0, 0]])[0, 16:19]
    got = np.array(row0[15:18])
    assert np.allclose(got, want, rtol=1e-14, atol=0) and not np.allclose(got, dip, rtol=1e-3)
    assert subprocess.run(base + ["--igrf_coeffs=/nonexistent"]).returncode != 0


def test_binary_grid_is_the_same_model(tmp_path, grid16):
    """a grid converted to the binary side-format by the CLI gives bit-identical plasma parameters to
    the same grid read from text and to the model built from host arrays."""
    from stanford_raytracer_amd import api

    F, b, qs, ms = grid16
    txt, binf = str(tmp_path / "g.txt"), str(tmp_path / "g.bin")
    wl.write_grid_file(txt, F, b, qs, ms)
    subprocess.run([os.path.join(BIN, "raytracer"), "--grid2bin_in=%s" % txt, "--grid2bin_out=%s" % binf], check=True)
    assert api.grid_file_is_binary(binf)
    pos, _, _ = wl.launch_set(500, 21)
    ref = api.Model.interp(F, b, qs, ms).plasma_params(pos)
    for path in (txt, binf):
        assert np.array_equal(api.Model.interp_file(path).plasma_params(pos), ref)


@pytest.mark.parametrize("modelnum", [3, 4])
def test_cli_models_3_and_4_write_what_the_library_computes(tmp_path, grid16, pointsfile, modelnum):
    """modelnum 3 / 4 through the CLI (adaptive RK45, outputper, chunked launches, --ray_order): the .ray file holds
    exactly the rows the library returns for the same launch set, in the reference's record format."""
    from stanford_raytracer_amd import api

    pos, d, w = wl.launch_set(300, 17)
    rf = tmp_path / "rays.txt"
    wl.write_rays_file(str(rf), pos, d, w)
    out = tmp_path / "out.ray"
    common = ["--first_attempt_policy=0",   # what api.make_params defaults to (the CLI's default is 1)
              "--outputper=5", "--dt0=0.001", "--dtmax=0.1", "--tmax=0.05", "--root=2", "--fixedstep=0", "--maxerr=5e-4",
              "--maxsteps=60", "--minalt=%r" % wl.MINALT, "--inputraysfile=%s" % rf, "--outputfile=%s" % out,
              "--yearday=2010001", "--milliseconds_day=0", "--use_tsyganenko=0", "--use_igrf=0", "--chunk_rays=128",
              "--ray_order=1", "--modelnum=%d" % modelnum]
    if modelnum == 3:
        F, b, qs, ms = grid16
        gf = tmp_path / "grid.txt"
        wl.write_grid_file(str(gf), F, b, qs, ms)
        flags = ["--interp_interpfile=%s" % gf]
        m = api.Model.interp(F, b, qs, ms)
    else:
        flags = ["--interp_interpfile=%s" % pointsfile, "--scattered_interp_window_scale=1.5", "--scattered_interp_order=2",
                 "--scattered_interp_exact=0", "--scattered_interp_local_window_scale=5"]
        m = api.Model.scattered_file(pointsfile)
    subprocess.run([os.path.join(BIN, "raytracer")] + common + flags, check=True)
    got = parse_ray_file(str(out))
    rows, nrows, stop, _ = m.trace(pos, d, w, fixedstep=0, dt0=1e-3, dtmax=0.1, tmax=0.05, maxerr=5e-4, maxsteps=60,
                                   minalt=wl.MINALT, del_=1e-6, outputper=5)
    kept = (nrows + 4) // 5
    assert len(got) == int(kept.sum())
    at = 0
    for r in range(len(w)):
        blk = got[at:at + kept[r]]
        at += kept[r]
        assert np.all(blk[:, 0] == r + 1) and np.all(blk[:, 1] == stop[r])
        want = rows[r, :kept[r]]
        # es24.15e3 keeps 16 significant digits
        assert np.allclose(blk[:, 2:18], want[:, 0:16], rtol=6e-16, atol=0)
        assert np.allclose(blk[:, 28:32], want[:, 16:20], rtol=6e-16, atol=0)
        assert np.allclose(blk[:, 18], w[r], rtol=6e-16)


def test_cli_buildsamples_pts2bin_and_damping(tmp_path, cfgfiles):
    """The tools either side of the path from the command line: the reference's random-grid-builder flags with the
    model of --modelnum as the source, the binary side-format of the sample file, tracing on the result, and the damping
    post-pass written next to the .ray file; each checked against the library called directly."""
    from stanford_raytracer_amd import api
    exe = os.path.join(BIN, "raytracer")
    b = np.array([-4.0, 4.0, -4.0, 4.0, -4.0, 4.0]) * wl.R_E
    model = ["--modelnum=1", "--ngo_configfile=%s" % cfgfiles["ngo"], "--yearday=2010001", "--milliseconds_day=0"]
    pts, binf = str(tmp_path / "pts.txt"), str(tmp_path / "pts.bin")
    bflags = ["--%s=%r" % (n, float(v)) for n, v in zip(["minx", "maxx", "miny", "maxy", "minz", "maxz"], b)]
    r = subprocess.run([exe, "--buildsamples=1", "--filename=%s" % pts, "--n_initial_uniform=3000", "--n_iri_pad=2000",
                        "--adaptive_nmax=3000", "--initial_tol=1.0", "--max_recursion=14", "--seed=9"] + bflags + model,
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    api.init(0)
    g = api.Model.ngo(cfgfiles["ngo"])
    want, counts = g.build_samples(b, n_initial_uniform=3000, n_iri_pad=2000, adaptive_nmax=3000, initial_tol=1.0,
                                   max_recursion=14, seed=9)
    got = np.loadtxt(pts, skiprows=4)
    assert got.shape == want.shape and "%d samples" % len(want) in r.stdout
    fin = np.isfinite(want)
    assert np.allclose(got[fin], want[fin], rtol=2e-15, atol=0)
    # only finite samples make a usable model-4 file
    keep = np.all(np.isfinite(want), axis=1)
    qs, ms = g.species()
    api.write_points_file(pts, want[keep], b, qs, ms)
    assert subprocess.run([exe, "--pts2bin_in=%s" % pts, "--pts2bin_out=%s" % binf]).returncode == 0
    x = want[keep][::50, :3] * (1 + 1e-9)
    a = api.Model.scattered_file(pts, order=1).plasma_params(x)
    c = api.Model.scattered_file(binf, order=1).plasma_params(x)
    assert np.array_equal(a, c, equal_nan=True)
    # trace with the damping post-pass
    rf, out, dmp = tmp_path / "rays.txt", tmp_path / "o.ray", tmp_path / "o.damp"
    pos0, dir0, w0 = wl.launch_set(6, 3)
    wl.write_rays_file(str(rf), pos0, dir0, w0)
    run = [exe, "--first_attempt_policy=0", "--dt0=1e-3", "--dtmax=0.02", "--tmax=0.1", "--root=2", "--fixedstep=0", "--maxerr=5e-4",
           "--maxsteps=200", "--minalt=%r" % wl.MINALT, "--outputper=4", "--inputraysfile=%s" % rf, "--outputfile=%s" % out,
           "--damping_out=%s" % dmp] + model
    assert subprocess.run(run).returncode == 0
    p = api.make_params(dt0=1e-3, dtmax=0.02, tmax=0.1, maxerr=5e-4, maxsteps=200, minalt=wl.MINALT, outputper=4, del_=1e-4)
    rows, nrows, stop, _ = g.trace(pos0, dir0, w0, params=p)
    k, m, f = api.damping(g.species(), 4, rows, nrows, w0)
    rec = np.loadtxt(str(dmp)).reshape(-1, 6)
    kept = (nrows - 1) // 4 + 1
    assert len(rec) == kept.sum()
    i = 0
    for ray in range(6):
        for s in range(kept[ray]):
            assert rec[i, 0] == ray + 1 and rec[i, 1] == 4 * s + 1 and rec[i, 5] == f[ray, s]
            assert np.isclose(rec[i, 3], k[ray, s], rtol=1e-13, atol=0, equal_nan=True)
            assert np.isclose(rec[i, 4], m[ray, s], rtol=1e-13, atol=0, equal_nan=True)
            i += 1


def test_cli_use_tsyganenko(tmp_path, cfgfiles):
    """--use_tsyganenko=1 with the driver's --tsyganenko_* flags: row 0 carries dipole + T04_s."""
    from stanford_raytracer_amd import api
    rf = tmp_path / "rays.txt"
    rf.write_text("1.5e7 2e6 1e6 1 0 0 1e4\n")
    out = tmp_path / "o.ray"
    parmod = [4.0, -30.0, 1.0, -5.0, 0.132, 0.303, 0.083, 0.07, 0.211, 0.308]
    names = ["Pdyn", "Dst", "ByIMF", "BzIMF", "W1", "W2", "W3", "W4", "W5", "W6"]
    run = [os.path.join(BIN, "raytracer"), "--dt0=1e-3", "--tmax=0.01", "--root=2", "--fixedstep=1", "--maxsteps=10",
           "--minalt=6.4712e6", "--inputraysfile=%s" % rf, "--outputfile=%s" % out, "--yearday=2010001",
           "--milliseconds_day=0", "--ngo_configfile=%s" % cfgfiles["ngo"], "--modelnum=1", "--use_tsyganenko=1"]
    run += ["--tsyganenko_%s=%r" % (n, v) for n, v in zip(names, parmod)]
    assert subprocess.run(run).returncode == 0
    row0 = [float(v) for v in out.read_text().splitlines()[0].split()]
    api.init(0)
    x = [[1.5e7, 2e6, 1e6]]
    want = api.Model.ngo(cfgfiles["ngo"]).set_field(use_tsyganenko=1, parmod=parmod).plasma_params(x)[0, 16:19]
    dip = api.Model.ngo(cfgfiles["ngo"]).plasma_params(x)[0, 16:19]
    got = np.array(row0[15:18])
    assert np.allclose(got, want, rtol=1e-14, atol=0) and np.abs(got - dip).max() > 1e-9


def test_cli_damping_on_an_existing_ray_file(tmp_path):
    """--damping_in=<.ray> --damping_out=<file>: the post-processor on a file the reference's own driver wrote
    (tests/golden/driver_interp_adaptive.ray), no tracing: the same numbers as api.damping on the file's rows."""
    from stanford_raytracer_amd import api

    src = os.path.join(GOLDEN_DIR, "driver_interp_adaptive.ray")
    out = tmp_path / "d.txt"
    subprocess.run([os.path.join(BIN, "raytracer"), "--damping_in=%s" % src, "--damping_out=%s" % out], check=True)
    r = api.read_ray_file(src)
    rows, nrows = api.padded_rows(r)
    k, m, f = api.damping((r["qs"], r["ms"]), 1, rows, nrows, r["w0"])
    rec = np.loadtxt(str(out)).reshape(-1, 6)
    assert len(rec) == nrows.sum()
    i = 0
    for ray in range(len(nrows)):
        for s_ in range(nrows[ray]):
            assert rec[i, 0] == r["raynum"][ray] and rec[i, 1] == s_ + 1 and rec[i, 5] == f[ray, s_]
            assert rec[i, 2] == rows[ray, s_, 0]
            assert np.isclose(rec[i, 3], k[ray, s_], rtol=1e-13, atol=0, equal_nan=True)
            assert np.isclose(rec[i, 4], m[ray, s_], rtol=1e-13, atol=0, equal_nan=True)
            i += 1
    assert np.all(m[:, 0] == 1.0) and np.nanmin(m[np.arange(len(nrows)), nrows - 1]) < 1.0   # the rays do damp


def test_cli_buildgrid_is_the_reference_builders_command_line(tmp_path, cfgfiles):
    """`raytracer --buildgrid=1` takes gcpm_dens_model_buildgrid's flags (gcpm_dens_model_buildgrid.f95:42-160: bounds, --nx
    --ny --nz read as reals and floored, --compder, --filename) with the model of --modelnum in place of GCPM and writes the
    builder's text layout (byte-for-byte against the reference's own file: tests/test_host_formats.py).  Against the library
    called directly, with and without the seven derivative blocks; the file then serves as a modelnum-3 input."""
    from stanford_raytracer_amd import api
    exe = os.path.join(BIN, "raytracer")
    b = np.array([-4.0, 4.5, -5.0, 4.0, -3.5, 4.2]) * wl.R_E
    model = ["--modelnum=1", "--ngo_configfile=%s" % cfgfiles["ngo"], "--yearday=2010001", "--milliseconds_day=0"]
    bflags = ["--%s=%r" % (n, float(v)) for n, v in zip(["minx", "maxx", "miny", "maxy", "minz", "maxz"], b)]
    api.init(0)
    g = api.Model.ngo(cfgfiles["ngo"])
    qs, ms = g.species()
    for compder in (0, 1):
        out, want = str(tmp_path / ("g%d.txt" % compder)), str(tmp_path / ("w%d.txt" % compder))
        r = subprocess.run([exe, "--buildgrid=1", "--filename=%s" % out, "--nx=9.0", "--ny=11", "--nz=7.9", "--compder=%d" % compder]
                           + bflags + model, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        assert "9 x 11 x 7 nodes" in r.stdout
        F, D = g.build_grid(9, 11, 7, b, compder=bool(compder))
        api.write_grid_file(want, F, b, qs, ms, derivs=D)
        assert open(out, "rb").read() == open(want, "rb").read()
    m = api.Model.interp_file(str(tmp_path / "g0.txt"))
    pos, _, _ = wl.launch_set(64, 5)
    a = m.plasma_params(pos * 0.6)
    c = api.Model.interp(F if False else g.build_grid(9, 11, 7, b)[0], b, qs, ms).plasma_params(pos * 0.6)
    assert np.allclose(a[:, 4:8], c[:, 4:8], rtol=1e-12, atol=0)       # the text keeps 16 significant digits of ln N
    bad = subprocess.run([exe, "--buildgrid=1", "--filename=%s" % out, "--nx=1", "--ny=4", "--nz=4"] + bflags + model,
                         capture_output=True, text=True)
    assert bad.returncode == 2 and "--nx must be >= 2" in bad.stderr
