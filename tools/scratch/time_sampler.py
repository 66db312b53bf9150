"""Time the device sample-set builder at the size of SURVEY 8(d) config 5 (825 k samples); run on the GPU box."""
import os, sys, tempfile, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stanford_raytracer_amd import api, workloads as wl

api.init(0)
cfg = os.path.join(tempfile.mkdtemp(), "newray.in")
open(cfg, "w").write(wl.NEWRAY_PLASMAPAUSE)
m = api.Model.ngo(cfg)
b = np.array([-10.0, 10.0] * 3) * wl.R_E
m.build_samples(b, n_initial_uniform=1000, seed=1)   # warm-up
for tol in (1.0,):
    t0 = time.time()
    s, c = m.build_samples(b, n_initial_uniform=200_000, adaptive_nmax=600_000, n_iri_pad=60_000, initial_tol=tol,
                           max_recursion=30, seed=5)
    dt = time.time() - t0
    print("samples", len(s), "stage counts", c, "seconds %.2f" % dt, "samples/s %.3g" % (len(s) / dt))
