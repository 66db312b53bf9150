#!/usr/bin/env python3
"""Time the reference's OWN program (oracle/_ref/raytracer, built by oracle/build_ref.py) on 1 process and on 8 processes over
disjoint ray files -- the only parallel mode the reference supports (plan item 1).  Build container only.

    python tools/time_reference_procs.py [scale of the default ray counts]     -> one line per workload
"""
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from stanford_raytracer_amd import workloads as wl  # noqa: E402

EXE = os.path.join(ROOT, "oracle", "_ref", "raytracer")
TSY = ["--tsyganenko_Pdyn=4", "--tsyganenko_Dst=1", "--tsyganenko_ByIMF=0", "--tsyganenko_BzIMF=-5", "--tsyganenko_W1=0.132",
       "--tsyganenko_W2=0.303", "--tsyganenko_W3=0.083", "--tsyganenko_W4=0.07", "--tsyganenko_W5=0.211", "--tsyganenko_W6=0.308"]


def count_steps(path):
    """accepted steps = records beyond row 0 (outputper = 1)."""
    n, rays = 0, set()
    for line in open(path):
        n += 1
        rays.add(line[:10])
    return n - len(rays)


def run(model_flags, pos, d, w, nproc, per, td, maxsteps):
    procs, outs = [], []
    for k in range(nproc):
        rf = os.path.join(td, "rays_%d_%d.txt" % (nproc, k))
        wl.write_rays_file(rf, pos[k * per:(k + 1) * per], d[k * per:(k + 1) * per], w[k * per:(k + 1) * per])
        out = os.path.join(td, "out_%d_%d.ray" % (nproc, k))
        outs.append(out)
        cmd = [EXE, "--outputper=1", "--dt0=0.001", "--dtmax=0.1", "--tmax=0.5", "--root=2", "--fixedstep=0", "--maxerr=5e-4",
               "--maxsteps=%d" % maxsteps, "--minalt=%r" % wl.MINALT, "--inputraysfile=%s" % rf, "--outputfile=%s" % out,
               "--yearday=2010001", "--milliseconds_day=0", "--use_tsyganenko=0", "--use_igrf=0"] + TSY + model_flags
        procs.append(cmd)
    import threading

    el = [0.0] * nproc
    aborted = [False] * nproc

    def one(k):
        t0 = time.time()
        r = subprocess.run(procs[k], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True)
        el[k] = time.time() - t0
        assert r.returncode == 0
        aborted[k] = "STOP" in r.stderr   # the reference `stop`s the whole process on an SVD failure (blas.f95:208-211)

    th = [threading.Thread(target=one, args=(k,)) for k in range(nproc)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    steps = [count_steps(o) for o in outs]
    return steps, el, sum(aborted)


def main():
    scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
    td = tempfile.mkdtemp()
    cfg = os.path.join(td, "newray.in")
    open(cfg, "w").write(wl.NEWRAY_PLASMAPAUSE)
    F, b = wl.make_grid(64, half_width=10.0 * wl.R_E)
    gf = os.path.join(td, "grid64.txt")
    wl.write_grid_file(gf, F, b)
    for tag, flags, seed, maxsteps, per in (("config[1] ngo (seed 2, maxsteps 512)", ["--modelnum=1", "--ngo_configfile=%s" % cfg], 2, 512, 480),
                                             ("config[2] interp, 64^3 text grid (seed 3, maxsteps 256)", ["--modelnum=3", "--interp_interpfile=%s" % gf], 3, 256, 96)):
        per = max(1, int(per * scale))
        pos, d, w = wl.launch_set(8 * per, seed)
        # set-up cost (file parse) measured on a zero-ray file and subtracted from every process's own elapsed time;
        # aggregate rate = sum of the processes' own rates while they run side by side
        _, e0, _ = run(flags, pos[:0], d[:0], w[:0], 1, 0, td, maxsteps)
        s1, e1, a1 = run(flags, pos, d, w, 1, per, td, maxsteps)
        s8, e8, a8 = run(flags, pos, d, w, 8, per, td, maxsteps)
        r1 = s1[0] / max(e1[0] - e0[0], 1e-9)
        r8 = sum(s / max(e - e0[0], 1e-9) for s, e in zip(s8, e8))
        print("%s, %d rays per process: 1 process %d steps in %.1f s (set-up %.1f s) = %.0f steps/s%s; 8 processes %d steps, "
              "%.1f s each on average = %.0f steps/s in aggregate (%.1fx)%s"
              % (tag, per, s1[0], e1[0], e0[0], r1, " [process STOPped early]" if a1 else "", sum(s8), sum(e8) / 8, r8, r8 / r1,
                 " [%d of 8 STOPped early]" % a8 if a8 else ""))


if __name__ == "__main__":
    main()
