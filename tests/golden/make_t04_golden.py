#!/usr/bin/env python3
"""Golden vectors for use_tsyganenko = 1 (SURVEY 8f-4, T04_s half), from the REAL reference (oracle/_ref/ref_harness built
from /root/reference with flang): T04_s itself, the module outputs of EXTERN, and the adapters' full field tail
(funcPlasmaParams with use_tsyganenko=1, with and without use_igrf).  Run in the build container only.
Writes tests/golden/t04_golden.npz.

    python tests/golden/make_t04_golden.py
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import refharness  # noqa: E402
from stanford_raytracer_amd import workloads as wl  # noqa: E402

# T04_s's model coefficients A(69) as the Fortran stores them (default-REAL literals in a REAL*8 array)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def table_A():
    import re
    txt = open(os.path.join(ROOT, "stanford_raytracer_amd", "csrc", "srt_t04_tables.h")).read()
    body = re.search(r"T04D_T04_S_A\[69\] = \{(.*?)\};", txt, re.S).group(1)
    return np.array([np.float64(np.float32(t.replace("(double)", "").rstrip("f"))) for t in body.replace("\n", " ").split(",") if t.strip()])


def main():
    assert refharness.available()
    rng = np.random.default_rng(2005)
    n = 400
    rows = np.zeros((n, 14))
    rows[:, 0] = rng.uniform(0.5, 8, n)          # Pdyn
    rows[:, 1] = rng.uniform(-80, 10, n)         # Dst
    rows[:, 2:4] = rng.uniform(-8, 8, (n, 2))    # ByIMF, BzIMF
    rows[:, 4:10] = rng.uniform(0, 2, (n, 6))    # W1..W6
    rows[:, 10] = rng.uniform(-0.55, 0.55, n)    # tilt
    p = rng.normal(size=(n, 3))
    p /= np.linalg.norm(p, axis=1)[:, None]
    rows[:, 11:14] = p * rng.uniform(1.05, 16, (n, 1))
    rows[0, 11:14] = [0.0, 0.0, 3.0]             # on the z axis: the linear-approximation branches
    rows[1, 11:14] = [2.0, 0.0, 0.0]
    rows = rows.astype(np.float32).astype(np.float64)
    store = {"t04_in": rows, "t04_out": refharness.run_mode("t04", rows)}
    A = table_A()
    td = tempfile.mkdtemp()
    fin, fout = os.path.join(td, "in.txt"), os.path.join(td, "o.bin")
    ext = rows.copy()
    with open(fin, "w") as f:
        f.write(" ".join("%.17g" % v for v in A) + "\n")
        for r in ext:
            f.write(" ".join("%.17g" % v for v in r) + "\n")
    subprocess.run([refharness.EXE, "--mode=ext", "--in=%s" % fin, "--out=%s" % fout], check=True, stdout=subprocess.DEVNULL,
                   stderr=subprocess.DEVNULL)
    store["ext_in"], store["ext_out"] = ext, np.fromfile(fout).reshape(-1, 33)
    # the adapters' tail: harness PARMOD = Pdyn 4, Dst 1, ByIMF 0, BzIMF -5, W = .132 .303 .083 .07 .211 .308
    cfg = os.path.join(td, "newray.in")
    open(cfg, "w").write(wl.NEWRAY_PLASMAPAUSE)
    x = rng.normal(size=(300, 3))
    x /= np.linalg.norm(x, axis=1)[:, None]
    x *= wl.R_E * rng.uniform(1.05, 9.0, size=(300, 1))
    store["x"] = x
    store["parmod"] = np.array([4.0, 1.0, 0.0, -5.0, 0.132, 0.303, 0.083, 0.07, 0.211, 0.308])
    for tag, yd, ms, igrf in (("a", 2010001, 0, 0), ("b", 2022150, 37000000, 1)):
        ref = refharness.run_mode("params", x, {"kind": 1, "file": cfg, "use_igrf": igrf, "use_tsyganenko": 1, "yearday": yd, "msec": ms})
        store["B_" + tag] = ref[:, 16:19]
        store["date_" + tag] = np.array([yd, ms, igrf])
    # G4: fixed-step Ngo trajectories in dipole + T04 (the harness's PARMOD), first date
    pos0, dir0, w0 = wl.appendix_b_rays(8)
    rays = np.concatenate([pos0, dir0, w0[:, None]], axis=1)
    kw = dict(dt0=1e-3, dtmax=0.1, maxerr=5e-4, maxsteps=40, minalt=wl.MINALT, root=2, tmax=0.03, fixedstep=1)
    kw["del"] = 1e-4
    out, _ = refharness.run_rays({"kind": 1, "file": cfg, "use_tsyganenko": 1, "yearday": 2010001, "msec": 0}, rays, **kw)
    T = max(len(o["rows"]) for o in out)
    rr = np.zeros((len(out), T, 20))
    for i, o in enumerate(out):
        r = o["rows"]
        rr[i, :len(r)] = np.concatenate([r[:, :16], r[:, 24:28]], axis=1)
    store.update(run_pos0=pos0, run_dir0=dir0, run_w0=w0, run_rows=rr, run_nrows=np.array([len(o["rows"]) for o in out]),
                 run_stop=np.array([o["stopcond"] for o in out]))
    store["provenance"] = np.array("oracle/_ref/ref_harness (flang -O3): TS05_aka_TS04.for, geopack2008.for, ngo adapter")
    np.savez_compressed(os.path.join(HERE, "t04_golden.npz"), **store)
    print("wrote t04_golden.npz", {k: getattr(v, "shape", None) for k, v in store.items()})


if __name__ == "__main__":
    main()
