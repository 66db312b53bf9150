#!/usr/bin/env python3
"""Golden vectors for the use_igrf = 1 branch of the adapters' field tail, from the REAL reference
(oracle/_ref/ref_harness --use_igrf=1: geopack2008's RECALC_08 / IGRF_GSW_08 compiled from /root/reference with flang).
Run in the build container only.  Writes tests/golden/igrf_golden.npz (inputs we authored + the reference's outputs).

    python tests/golden/make_igrf_golden.py
"""
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import refharness  # noqa: E402
from stanford_raytracer_amd import workloads as wl  # noqa: E402

DATES = [(2010001, 0), (2022150, 37000000), (1987200, 5000000), (1968100, 1000)]


def main():
    assert refharness.available()
    td = tempfile.mkdtemp()
    cfg = os.path.join(td, "newray.in")
    open(cfg, "w").write(wl.NEWRAY_PLASMAPAUSE)
    rng = np.random.default_rng(20241)
    x = rng.normal(size=(300, 3))
    x /= np.linalg.norm(x, axis=1)[:, None]
    x *= wl.R_E * rng.uniform(1.02, 9.0, size=(300, 1))
    x[0] = [0, 0, 2 * wl.R_E]                      # geographic-pole branch is in GEO, not SM; still worth a polar point
    x[1] = [0, 0, -3 * wl.R_E]
    store = {"x": x, "dates": np.array(DATES)}
    for i, (yd, ms) in enumerate(DATES):
        ref = refharness.run_mode("params", x, {"kind": 1, "file": cfg, "use_igrf": 1, "yearday": yd, "msec": ms})
        store["B_%d" % i] = ref[:, 16:19]
    # G4: fixed-step Ngo trajectories in the IGRF field (config-1 style), first date
    pos0, dir0, w0 = wl.appendix_b_rays(8)
    rays = np.concatenate([pos0, dir0, w0[:, None]], axis=1)
    kw = dict(dt0=1e-3, dtmax=0.1, maxerr=5e-4, maxsteps=60, minalt=wl.MINALT, root=2, tmax=0.05, fixedstep=1)
    kw["del"] = 1e-4
    out, _ = refharness.run_rays({"kind": 1, "file": cfg, "use_igrf": 1, "yearday": DATES[0][0], "msec": DATES[0][1]}, rays, **kw)
    T = max(len(o["rows"]) for o in out)
    rows = np.zeros((len(out), T, 20))
    for i, o in enumerate(out):
        r = o["rows"]
        rows[i, :len(r)] = np.concatenate([r[:, :16], r[:, 24:28]], axis=1)
    store.update(run_pos0=pos0, run_dir0=dir0, run_w0=w0, run_rows=rows,
                 run_nrows=np.array([len(o["rows"]) for o in out]), run_stop=np.array([o["stopcond"] for o in out]))
    store["provenance"] = np.array("oracle/_ref/ref_harness (flang -O3, x86-64 baseline), geopack2008.for + geopack0508_adapter.for")
    np.savez_compressed(os.path.join(HERE, "igrf_golden.npz"), **store)
    print("wrote igrf_golden.npz", {k: getattr(v, "shape", None) for k, v in store.items()})


if __name__ == "__main__":
    main()
