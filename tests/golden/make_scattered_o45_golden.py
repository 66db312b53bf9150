#!/usr/bin/env python3
"""Golden vectors for --scattered_interp_order=4 and 5 (J = 35 / 56 monomials from generate_monomials, lsinterp_mod.f95:114-164,
273-281) from the REAL reference (oracle/_ref/ref_harness, built by oracle/build_ref.py).  Run in the build container only.

Inputs: the committed 5 500-sample set (tests/golden/points5500.npz) with --scattered_interp_window_scale=2.5, so that most
query points see more samples than monomials (at the fixtures' 1.5 a quarter of the order-4 fits and most order-5 fits end in
"too few samples" -- those cases are here too: the first rows of g0_x use 1.5).  Outputs -> tests/golden/scattered_o45_golden.npz:
  g0_x, g0_o{4,5}_out, g0_o{4,5}_narrow_out   funcPlasmaParams at g0_x (window scale 2.5; the first 16 points also at 1.5)
  g2_in, g2_o{4,5}_out                          (x, k on the whistler root, w, del) and the reference's dFdk, dFdw, dFdx, evalrhs
  rays, g4_o4_rows / _nrows / _stop             8 fixed-step rays of 6 steps at order 4, every row
  ref_root_index / ref_root_point               the sample at the root of the reference's kd-tree (stored spacing 0, SURVEY A-12)

    python tests/golden/make_scattered_o45_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import refharness  # noqa: E402
from stanford_raytracer_amd import workloads as wl  # noqa: E402

WS = 2.5


def main():
    if not refharness.available():
        raise SystemExit("oracle/_ref/ref_harness missing: run python oracle/build_ref.py first")
    info = open(os.path.join(ROOT, "oracle", "_ref", "BUILD_INFO.txt")).read()
    work = os.path.join(HERE, "_work")
    os.makedirs(work, exist_ok=True)
    g = np.load(os.path.join(HERE, "points5500.npz"))
    ptsfile = os.path.join(work, "points.txt")
    wl.write_points_file(ptsfile, g["pts"], g["lnN"], g["bounds"], g["qs"], g["ms"])
    base = {"kind": 4, "file": ptsfile, "window_scale": WS, "order": 4, "exact": 0, "local_window_scale": 5.0}
    gold = np.load(os.path.join(HERE, "golden.npz"))
    x = gold["g0_scattered_x"][:96]
    store = {"build_info": np.array(info), "g0_x": x, "window_scale": np.array(WS)}
    for order in (4, 5):
        m = dict(base, order=order)
        store["g0_o%d_out" % order] = refharness.run_mode("params", x, m)
        store["g0_o%d_narrow_out" % order] = refharness.run_mode("params", x[:16], dict(m, window_scale=1.5))
    pos, d, w = wl.launch_set(120, 909)
    pos = pos * 0.9
    rows = np.concatenate([pos, d, w[:, None]], axis=1)
    out = refharness.run_mode("disp", rows, base)
    ok = out[:, 8] > 0
    xs, ks, ws = pos[ok][:24], (out[ok, 8:9] * d[ok])[:24], w[ok][:24]
    gin = np.concatenate([xs, ks, ws[:, None], np.full((len(ws), 1), 1e-6)], axis=1)
    store["g2_in"] = gin
    for order in (4, 5):
        store["g2_o%d_out" % order] = refharness.run_mode("grad", gin, dict(base, order=order))
    rays = np.concatenate([pos[ok][:8], d[ok][:8], w[ok][:8, None]], axis=1)
    store["rays"] = rays
    kw = dict(dt0=1e-4, dtmax=1e-4, tmax=5.5e-4, maxerr=5e-4, maxsteps=6, minalt=wl.MINALT, root=2, fixedstep=1)
    out, _ = refharness.run_rays(base, rays, **kw)
    T = max(o["rows"].shape[0] for o in out)
    rws = np.zeros((len(out), T, 20))
    nrows = np.zeros(len(out), dtype=np.int32)
    stop = np.zeros(len(out), dtype=np.int32)
    for i, o in enumerate(out):
        r = np.concatenate([o["rows"][:, :16], o["rows"][:, 24:28]], axis=1)  # t, pos, vprel, vgrel, n, B0 | Ns
        rws[i, :r.shape[0]] = r
        nrows[i] = r.shape[0]
        stop[i] = o["stopcond"]
    store["g4_o4_rows"], store["g4_o4_nrows"], store["g4_o4_stop"] = rws, nrows, stop
    store["g4_o4_params"] = np.array([kw["dt0"], kw["dtmax"], kw["tmax"], kw["maxerr"], kw["minalt"], kw["maxsteps"], kw["root"], 1])
    print("g4 order 4: rows %s, stop codes %s" % (nrows.tolist(), stop.tolist()))
    rp, rv, mx = refharness.scattered_root(base)
    dd = np.linalg.norm(g["pts"] - rp, axis=1)
    assert dd.min() == 0.0 and rv[-1] == 0.0
    store["ref_root_index"] = np.array(int(np.argmin(dd)))
    store["ref_root_point"] = rp
    store["ref_maxnearest"] = np.array(mx)
    np.savez_compressed(os.path.join(HERE, "scattered_o45_golden.npz"), **store)
    for order in (4, 5):
        o = store["g0_o%d_out" % order]
        n = store["g0_o%d_narrow_out" % order]
        print("order %d: %d / %d query points with a fit (Ns not 0 or 1); window 1.5: %d / 16" % (
            order, int(((o[:, 4] > 0) & (o[:, 4] != 1)).sum()), len(o), int(((n[:, 4] > 0) & (n[:, 4] != 1)).sum())))


if __name__ == "__main__":
    main()
