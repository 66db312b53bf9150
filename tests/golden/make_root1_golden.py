#!/usr/bin/env python3
"""Golden trajectories of the REAL reference (oracle/_ref/ref_harness, see make_golden.py) for two corners the main
fixture does not hold:

  * root = 1 (raytracer.f95:685-690, 828-836: the `k1` root of solve_dispersion_relation is followed instead of `k2`):
    HF rays (2 .. 5 MHz, above the plasma frequency, where BOTH roots propagate), fixed-step RK4 and adaptive RKF45,
    Ngo and interp models;
  * modelnum 4 (scattered samples) with the fixed-step integrator (scattered_interp_dens_model_adapter.f95:249-372 under
    raytracer.f95:504-532): the rung where row counts and time grids must be EQUAL.

    python tests/golden/make_root1_golden.py        (build container only)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from make_golden import run_set  # noqa: E402
from oracle import refharness  # noqa: E402
from stanford_raytracer_amd import workloads as wl  # noqa: E402


def main():
    if not refharness.available():
        raise SystemExit("oracle/_ref/ref_harness missing: run python oracle/build_ref.py first")
    info = open(os.path.join(ROOT, "oracle", "_ref", "BUILD_INFO.txt")).read()
    work = os.path.join(HERE, "_work")
    os.makedirs(work, exist_ok=True)
    cfg_pp = os.path.join(work, "newray_plasmapause.in")
    open(cfg_pp, "w").write(wl.NEWRAY_PLASMAPAUSE)
    g = np.load(os.path.join(HERE, "grid16.npz"))
    gridfile = os.path.join(work, "grid16.txt")
    wl.write_grid_file(gridfile, g["F"], g["bounds"])
    pz = np.load(os.path.join(HERE, "points5500.npz"))
    ptsfile = os.path.join(work, "points.txt")
    wl.write_points_file(ptsfile, pz["pts"], pz["lnN"], pz["bounds"])
    ngo = {"kind": 1, "file": cfg_pp}
    interp = {"kind": 3, "file": gridfile}
    scat = {"kind": 4, "file": ptsfile, "window_scale": 1.5, "order": 2, "exact": 0, "local_window_scale": 5.0}
    store = {"build_info": np.array(info)}

    # ---- root = 1: HF rays on the Appendix-B launch points / directions
    p0, d0, _ = wl.appendix_b_rays()
    w0 = 2.0 * np.pi * (2.0e6 + 2.0e5 * np.arange(len(p0)))
    store["hf_rays"] = np.concatenate([p0, d0, w0[:, None]], axis=1)
    hf = dict(dt0=2e-5, dtmax=1e-3, maxsteps=400)
    for root in (1, 2):  # root 2 on the same rays: the two modes must come out different
        run_set(ngo, p0, d0, w0, "ngo_fixed_root%d" % root, store, fixedstep=1, tmax=3e-3, root=root, **hf)
        run_set(ngo, p0, d0, w0, "ngo_adaptive_root%d" % root, store, fixedstep=0, tmax=3e-3, root=root, **hf)
    run_set(interp, p0, d0, w0, "interp_fixed_root1", store, fixedstep=1, tmax=2e-3, root=1, **hf)
    run_set(interp, p0, d0, w0, "interp_adaptive_root1", store, fixedstep=0, tmax=2e-3, root=1, **hf)

    # ---- modelnum 4, fixed-step RK4 (whistler-mode rays of the main fixture's scattered launch set)
    lp, ld, lw = wl.launch_set(24, 404)
    store["scattered_rays"] = np.concatenate([lp * 0.9, ld, lw[:, None]], axis=1)
    run_set(scat, lp * 0.9, ld, lw, "scattered_fixed", store, fixedstep=1, dt0=5e-4, tmax=0.0125, maxsteps=60)

    out = os.path.join(HERE, "root1_golden.npz")
    np.savez_compressed(out, **store)
    print("wrote", out, "%.1f KB" % (os.path.getsize(out) / 1e3))
    for k in sorted(store):
        if k.endswith("_nrows"):
            print(k, store[k].tolist(), store[k.replace("_nrows", "_stop")].tolist())
    for f in os.listdir(work):
        os.remove(os.path.join(work, f))
    os.rmdir(work)


if __name__ == "__main__":
    main()
