"""tools/scratch/stepctl_diag.py -- where the GPU's first adaptive time stamps leave the oracle's (GPU box; uses the test fixtures' data)"""
import os, sys, tempfile
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from stanford_raytracer_amd import api, workloads as wl
from oracle import oracle
api.init(0)
G = "tests/golden"
g = np.load(os.path.join(G, "grid16.npz")); F, b, qs, ms = g["F"], g["bounds"], g["qs"], g["ms"]
gi, oi = api.Model.interp(F, b, qs, ms), oracle.Model.interp(F, b, qs, ms)
p5 = np.load(os.path.join(G, "points5500.npz"))
pf = os.path.join(tempfile.mkdtemp(), "p.txt"); wl.write_points_file(pf, p5["pts"], p5["lnN"], p5["bounds"], p5["qs"], p5["ms"])
gs, os_ = api.Model.scattered_file(pf), oracle.Model.scattered_file(pf, perm_seed=2 | 0x80000000)
pos, d, w = wl.launch_set(192, 23); pos = pos * 0.9
kw = dict(fixedstep=0, dt0=1e-3, dtmax=0.1, tmax=0.05, maxerr=5e-4, maxsteps=30, del_=1e-6)
for name, gm, om in (("interp", gi, oi), ("scattered", gs, os_)):
    rows, nrows, stop, _ = gm.trace(pos, d, w, outputper=1, **kw)
    orows, onrows, ostop, _ = om.trace(pos, d, w, capacity=30, **kw)
    both = (nrows > 4) & (onrows > 4)
    same = np.all(rows[:, 1:4, 0] == orows[:, 1:4, 0], axis=1)
    print(name, "both", both.sum(), "same", (same & both).sum())
    for i in np.nonzero(both & ~same)[0][:8]:
        print("  ray", i, "gpu t", rows[i, 1:5, 0], "oracle t", orows[i, 1:5, 0])
