import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle
from stanford_raytracer_amd import api, workloads as wl
np.set_printoptions(linewidth=220, precision=4)
def vrel(a, b): return np.linalg.norm(a - b, axis=-1) / np.maximum(np.linalg.norm(b, axis=-1), 1e-300)
api.init(0)
F, b = wl.make_grid(24, half_width=5 * wl.R_E)
gm, om = api.Model.interp(F, b, wl.QS, wl.MS), oracle.Model.interp(F, b, wl.QS, wl.MS)
del_ = 1e-6; dt = 1e-3
pos, d, w = wl.launch_set(512, 11)
od = np.array([om.disp(p, dd, ww) for p, dd, ww in zip(pos, d, w)])
ok = od[:, 8] > 0
x, k, ww = pos[ok], od[ok, 8:9] * d[ok], w[ok]
A = [[], [1/4], [3/32, 9/32], [1932/2197, -7200/2197, 7296/2197], [439/216, -8, 3680/513, -845/4104], [-8/27, 2, -3544/2565, 1859/4104, -11/40]]
st0 = np.concatenate([x, k], axis=1)
ks = []
for s in range(6):
    tmp = st0.copy()
    for j, a in enumerate(A[s]): tmp = tmp + a * ks[j]
    rg = gm.gradients(tmp[:, :3], tmp[:, 3:], ww, del_)
    rc = np.array([om.grad(t[:3], t[3:], c, del_) for t, c in zip(tmp, ww)])
    e_x = vrel(rg[:, 7:10], rc[:, 7:10]); e_k = vrel(rg[:, 10:13], rc[:, 10:13])
    e_dfdx = vrel(rg[:, 4:7], rc[:, 4:7])
    print("stage", s + 1, "rhs_x med %.2e max %.2e | rhs_k med %.2e max %.2e | dfdx med %.2e max %.2e" % (np.median(e_x), e_x.max(), np.median(e_k), e_k.max(), np.median(e_dfdx), e_dfdx.max()))
    ks.append(dt * rc[:, 7:13])
# sensitivity of the oracle itself: perturb initial k by 1e-10 relative and see change in rk45 result
args = np.concatenate([x, k, ww[:, None]], axis=1)
base = np.array([om.step(a, dt, del_) for a in args])
pert = args.copy(); pert[:, 3:6] *= (1 + 1e-10)
b2 = np.array([om.step(a, dt, del_) for a in pert])
for nm, o in (("rk4", 0), ("rk45_4", 7), ("rk45_5", 14)):
    e = vrel(b2[:, o + 3:o + 6] - pert[:, 3:6], base[:, o + 3:o + 6] - args[:, 3:6])
    print("oracle self-sensitivity (k*(1+1e-10))", nm, "k-incr med %.2e max %.2e" % (np.median(e), e.max()))
gs = gm.rk_step(args, dt, del_)
for nm, o in (("rk4", 0), ("rk45_4", 7), ("rk45_5", 14)):
    e = vrel(gs[:, o + 3:o + 6] - args[:, 3:6], base[:, o + 3:o + 6] - args[:, 3:6])
    i = np.argsort(e)[len(e) // 2]
    print("gpu vs oracle", nm, "k-incr med %.2e" % np.median(e), "sample", i, "gpu", gs[i, o + 3:o + 6] - args[i, 3:6], "cpu", base[i, o + 3:o + 6] - args[i, 3:6])
