#!/usr/bin/env python3
"""tools/scratch/dump_lengths.py -- trace the default bench workload once and save every ray's row count and stop code
(gpurun_out/r02_f/lengths.npz): data for scheduling experiments (which launch parameters predict a long ray)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from stanford_raytracer_amd import api, workloads as wl  # noqa: E402
from stanford_raytracer_amd.device_batch import DeviceBatch  # noqa: E402

api.init(0)
dev = torch.device("cuda", 0)
F, b = wl.make_grid(256, half_width=10.0 * wl.R_E)
m = api.Model.interp(F, b, wl.QS, wl.MS)
p = api.make_params(fixedstep=0, dt0=1e-3, dtmax=0.1, maxerr=5e-4, tmax=0.5, root=2, minalt=wl.MINALT, maxsteps=256,
                    outputper=16, del_=1e-6, ray_order=1)
pos, d, w = wl.launch_set(1_000_000, 3)
bt = DeviceBatch(m, p, pos, d, w, dev)
o = bt.launch()
torch.cuda.synchronize()
out = os.path.join(ROOT, "gpurun_out", "r02_f")
os.makedirs(out, exist_ok=True)
cnt = o["cnt"].cpu().numpy()
np.savez_compressed(os.path.join(out, "lengths.npz"), nrows=o["nrows"].cpu().numpy(), stop=o["stop"].cpu().numpy(), cnt=cnt)
print("kernel ms", m.last_kernel_ms(), cnt)
