#!/usr/bin/env python3
"""`.ray` goldens written by the reference's OWN program (fortran/raytracer_driver.f95 compiled where it lies into
oracle/_ref/raytracer by oracle/build_ref.py): its flag parsing (:181-228, per-model blocks) and its record writer
(:1197-1217) -- not a restatement of them.  Run in the build container only.

  config1_outputper25.ray      BASELINE config[0]: 16 Appendix-B rays, Ngo + dipole, fixed RK4 (dt 1e-3, tmax 0.1,
                               outputper 25).  Must come out byte-identical to the committed file (which the harness's
                               restated writer produced in round 1) -- asserted here.
  driver_interp_adaptive.ray   the same 16 rays through modelnum 3 on the committed 16^3 grid (tests/golden/grid16.npz),
                               adaptive RKF45 (maxerr 5e-4, tmax 0.2, outputper 16)

    python tests/golden/make_driver_golden.py
"""
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from stanford_raytracer_amd import workloads as wl  # noqa: E402

EXE = os.path.join(ROOT, "oracle", "_ref", "raytracer")
TSY = ["--tsyganenko_Pdyn=4", "--tsyganenko_Dst=1", "--tsyganenko_ByIMF=0", "--tsyganenko_BzIMF=-5", "--tsyganenko_W1=0.132",
       "--tsyganenko_W2=0.303", "--tsyganenko_W3=0.083", "--tsyganenko_W4=0.07", "--tsyganenko_W5=0.211", "--tsyganenko_W6=0.308"]


def common(rays, out):
    return ["--dt0=0.001", "--dtmax=0.1", "--root=2", "--maxerr=5e-4", "--maxsteps=2000", "--minalt=%r" % wl.MINALT,
            "--inputraysfile=%s" % rays, "--outputfile=%s" % out, "--yearday=2010001", "--milliseconds_day=0",
            "--use_tsyganenko=0", "--use_igrf=0"] + TSY


def config1_flags(rays, out, cfg):
    return ["--outputper=25", "--tmax=0.1", "--fixedstep=1", "--modelnum=1", "--ngo_configfile=%s" % cfg] + common(rays, out)


def interp_flags(rays, out, grid):
    return ["--outputper=16", "--tmax=0.2", "--fixedstep=0", "--modelnum=3", "--interp_interpfile=%s" % grid] + common(rays, out)


def main():
    if not os.path.exists(EXE):
        raise SystemExit("oracle/_ref/raytracer missing: run python oracle/build_ref.py first")
    work = os.path.join(HERE, "_work")
    os.makedirs(work, exist_ok=True)
    cfg = os.path.join(work, "newray_plasmapause.in")
    open(cfg, "w").write(wl.NEWRAY_PLASMAPAUSE)
    p0, d0, w0 = wl.appendix_b_rays()
    rays = os.path.join(work, "rays16.txt")
    wl.write_rays_file(rays, p0, d0, w0)
    out1 = os.path.join(work, "config1_driver.ray")
    subprocess.run([EXE] + config1_flags(rays, out1, cfg), check=True, stdout=subprocess.DEVNULL)
    committed = os.path.join(HERE, "config1_outputper25.ray")
    assert open(out1, "rb").read() == open(committed, "rb").read(), "the reference driver's config-1 file differs from the committed golden"
    print("config1_outputper25.ray: byte-identical to the reference driver's own output")
    g = np.load(os.path.join(HERE, "grid16.npz"))
    grid = os.path.join(work, "grid16.txt")
    wl.write_grid_file(grid, g["F"], g["bounds"], g["qs"], g["ms"])
    out3 = os.path.join(HERE, "driver_interp_adaptive.ray")
    subprocess.run([EXE] + interp_flags(rays, out3, grid), check=True, stdout=subprocess.DEVNULL)
    os.chmod(out3, 0o644)
    print("driver_interp_adaptive.ray: %d records" % sum(1 for _ in open(out3)))


if __name__ == "__main__":
    main()
