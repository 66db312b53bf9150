#!/usr/bin/env python3
"""What a fit of order 4 / 5 costs on the device (tests/golden/points5500.npz, window scale 2.5): funcPlasmaParams at 4 096 points,
and 64 rays x 6 fixed steps through the trace kernel; order 2 beside them."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from stanford_raytracer_amd import api, workloads as wl  # noqa: E402

g = np.load(os.path.join(ROOT, "tests", "golden", "points5500.npz"))
path = "/tmp/pts5500.txt"
wl.write_points_file(path, g["pts"], g["lnN"], g["bounds"], g["qs"], g["ms"])
pos, d, w = wl.launch_set(4096, 11)
pos = pos * 0.9
for order in (2, 4, 5):
    m = api.Model.scattered_file(path, order=order, window_scale=2.5)
    m.plasma_params(pos[:64])
    t0 = time.time()
    out = m.plasma_params(pos)
    t1 = time.time()
    ok = (out[:, 4] > 0) & (out[:, 4] != 1)
    kw = dict(dt0=1e-4, dtmax=1e-4, tmax=5.5e-4, maxerr=5e-4, minalt=wl.MINALT, maxsteps=6, root=2, fixedstep=1, del_=1e-6)
    t2 = time.time()
    rows, nrows, stop, _ = m.trace(pos[:64], d[:64], w[:64], outputper=1, **kw)
    t3 = time.time()
    print("order %d: params %d points %.3f s (%d with a fit); trace 64 rays x 6 steps %.3f s, rows %d" % (
        order, len(pos), t1 - t0, int(ok.sum()), t3 - t2, int(nrows.sum())))
