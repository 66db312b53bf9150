"""tools/scratch/igrf_det.py -- is the IGRF trace kernel deterministic, and does a library variant change its rows?  (GPU box)"""
import sys
import numpy as np
sys.path.insert(0, ".")
from stanford_raytracer_amd import api, workloads as wl

api.init(0)
F, b = wl.make_grid(24, half_width=10 * wl.R_E)
m = api.Model.interp(F, b, wl.QS, wl.MS)
m.set_field(use_igrf=1)
pos, d, w = wl.launch_set(20000, 3)
kw = dict(fixedstep=0, dt0=1e-3, dtmax=0.1, tmax=0.5, maxerr=5e-4, maxsteps=64, del_=1e-6, outputper=8)
r1, n1, s1, a1 = m.trace(pos, d, w, **kw)
r2, n2, s2, a2 = m.trace(pos, d, w, **kw)
print("steps", a1, a2, "rows equal", np.array_equal(np.nan_to_num(r1), np.nan_to_num(r2)), "nrows equal", np.array_equal(n1, n2), "stop equal", np.array_equal(s1, s2))
bad = np.nonzero(n1 != n2)[0]
print("rays with different row counts:", len(bad), bad[:10])
np.save(sys.argv[1], np.concatenate([n1.astype(np.float64), s1.astype(np.float64), np.nan_to_num(r1).reshape(-1)[:2000000]]))
