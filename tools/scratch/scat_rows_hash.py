#!/usr/bin/env python3
"""sha256 of every kept row of a scattered-model run (5 500-sample fixture; window scale 1.5: lists shorter than the LDS side arrays,
4.0: lists of ~1 000 samples, the record-resident loops): two builds of the library that do the same arithmetic must print the same
hashes (SRT_LIB_OVERRIDE selects the library)."""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from stanford_raytracer_amd import api, workloads as wl  # noqa: E402

g = np.load(os.path.join(ROOT, "tests", "golden", "points5500.npz"))
path = "/tmp/pts5500_hash.txt"
wl.write_points_file(path, g["pts"], g["lnN"], g["bounds"], g["qs"], g["ms"])
pos, d, w = wl.launch_set(3000, 17)
pos = pos * 0.9
for ws in (1.5, 4.0):
    m = api.Model.scattered_file(path, window_scale=ws)
    rows, nrows, stop, _ = m.trace(pos, d, w, outputper=1, dt0=1e-3, dtmax=0.05, tmax=0.2, maxerr=5e-4, minalt=wl.MINALT,
                                   maxsteps=40, root=2, fixedstep=0, del_=1e-6)
    h = hashlib.sha256()
    for i in range(len(nrows)):
        h.update(np.ascontiguousarray(rows[i, :nrows[i]]).tobytes())
    h.update(nrows.tobytes())
    h.update(stop.tobytes())
    print("window %.1f: rows %d, sha256 %s" % (ws, int(nrows.sum()), h.hexdigest()[:24]))
