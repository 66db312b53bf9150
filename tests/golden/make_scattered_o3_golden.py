#!/usr/bin/env python3
"""Golden vectors for --scattered_interp_order=3 (J = 20 monomials, lsinterp_mod.f95:91-99, 244-273) from the REAL
reference (oracle/_ref/ref_harness, built by oracle/build_ref.py).  Run in the build container only.

Inputs: the committed 5 500-sample set (tests/golden/points5500.npz, written with the bytes make_golden.py used) and the
query points of golden.npz's g0_scattered_x.  Outputs -> tests/golden/scattered_o3_golden.npz:
  g0_o3_out        funcPlasmaParams at g0_scattered_x, order 3
  g2_o3_in / _out  (x, k on the whistler root, w) and the reference's dFdk, dFdw, dFdx, evalrhs there (order 3, del = 1e-6)
  ref_root_index / ref_root_point / ref_maxnearest   the sample at the root of the reference's kd-tree (ref_harness
                   --mode=scatroot), whose nearest-sample distance the reference stores as 0, and its maxnearest

    python tests/golden/make_scattered_o3_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import refharness  # noqa: E402
from stanford_raytracer_amd import workloads as wl  # noqa: E402


def main():
    if not refharness.available():
        raise SystemExit("oracle/_ref/ref_harness missing: run python oracle/build_ref.py first")
    info = open(os.path.join(ROOT, "oracle", "_ref", "BUILD_INFO.txt")).read()
    work = os.path.join(HERE, "_work")
    os.makedirs(work, exist_ok=True)
    g = np.load(os.path.join(HERE, "points5500.npz"))
    ptsfile = os.path.join(work, "points.txt")
    wl.write_points_file(ptsfile, g["pts"], g["lnN"], g["bounds"], g["qs"], g["ms"])
    scat3 = {"kind": 4, "file": ptsfile, "window_scale": 1.5, "order": 3, "exact": 0, "local_window_scale": 5.0}
    gold = np.load(os.path.join(HERE, "golden.npz"))
    x = gold["g0_scattered_x"]
    store = {"build_info": np.array(info), "g0_o3_out": refharness.run_mode("params", x, scat3)}
    # states on the whistler root (the reference's own solve_dispersion_relation), then its gradients
    pos, d, w = wl.launch_set(120, 909)
    pos = pos * 0.9
    rows = np.concatenate([pos, d, w[:, None]], axis=1)
    out = refharness.run_mode("disp", rows, scat3)
    ok = out[:, 8] > 0
    xs, ks, ws = pos[ok][:64], (out[ok, 8:9] * d[ok])[:64], w[ok][:64]
    gin = np.concatenate([xs, ks, ws[:, None], np.full((len(ws), 1), 1e-6)], axis=1)
    store["g2_o3_in"] = gin
    store["g2_o3_out"] = refharness.run_mode("grad", gin, scat3)
    # the sample at the root of the REFERENCE's kd-tree (it depends on the compiler's RNG through randperm,
    # scattered_..adapter.f95:137-165): the one sample whose stored nearest-sample distance the reference leaves at 0
    rp, rv, mx = refharness.scattered_root(scat3)
    d = np.linalg.norm(g["pts"] - rp, axis=1)
    assert d.min() == 0.0 and rv[-1] == 0.0
    store["ref_root_index"] = np.array(int(np.argmin(d)))
    store["ref_root_point"] = rp
    store["ref_maxnearest"] = np.array(mx)
    np.savez_compressed(os.path.join(HERE, "scattered_o3_golden.npz"), **store)
    o = store["g0_o3_out"]
    print("order 3: %d / %d query points with a fit (Ns not 0 or 1)" % (int(((o[:, 4] > 0) & (o[:, 4] != 1)).sum()), len(o)))


if __name__ == "__main__":
    main()
