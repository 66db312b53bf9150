#!/usr/bin/env python3
"""tools/cli_end_to_end.py [nrays] -- BASELINE config[2] through the drop-in executable, files in, files out: binary grid
(SRTGRID1), text ray file in, text .ray file out (the reference's record format), wall-clock per phase.  Run on the GPU box."""
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from stanford_raytracer_amd import api, workloads as wl  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
td = tempfile.mkdtemp(dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
t0 = time.time()
F, b = wl.make_grid(256, half_width=10.0 * wl.R_E)
gf = os.path.join(td, "grid256.bin")
api.write_grid_file(gf, F, b, wl.QS, wl.MS, binary=True)
del F
pos, d, w = wl.launch_set(n, 3)
rf = os.path.join(td, "rays.txt")
np.savetxt(rf, np.concatenate([pos, d, w[:, None]], axis=1), fmt="%.17g")
print("inputs written in %.1f s (grid %.2f GB, rays %.2f GB)" % (time.time() - t0, os.path.getsize(gf) / 1e9, os.path.getsize(rf) / 1e9))
out = os.path.join(td, "out.ray")
cmd = [os.path.join(ROOT, "stanford_raytracer_amd", "bin", "raytracer"), "--outputper=16", "--dt0=0.001", "--dtmax=0.1", "--tmax=0.5",
       "--root=2", "--fixedstep=0", "--maxerr=5e-4", "--maxsteps=256", "--minalt=%r" % wl.MINALT, "--inputraysfile=%s" % rf,
       "--outputfile=%s" % out, "--modelnum=3", "--interp_interpfile=%s" % gf, "--yearday=2010001", "--milliseconds_day=0",
       "--use_tsyganenko=0", "--use_igrf=0", "--ray_order=1"]
for extra in ([], ["--devices=0,0"]):
    t0 = time.time()
    r = subprocess.run(cmd + extra, stdout=subprocess.PIPE, text=True)
    dt = time.time() - t0
    print("%s: rc %d, %.1f s wall, output %.2f GB; %s" % (" ".join(extra) or "one device", r.returncode, dt, os.path.getsize(out) / 1e9,
                                                          r.stdout.strip().splitlines()[-1]))
for f in (gf, rf, out):
    os.remove(f)
