import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle
from stanford_raytracer_amd import api, workloads as wl
np.set_printoptions(linewidth=200, precision=6)
def vrel(a, b):
    return np.linalg.norm(a - b, axis=-1) / np.maximum(np.linalg.norm(b, axis=-1), 1e-300)
api.init(0)
td = tempfile.mkdtemp()
cfg = os.path.join(td, "newray.in"); open(cfg, "w").write(wl.NEWRAY_PLASMAPAUSE)
F, b = wl.make_grid(24, half_width=5 * wl.R_E)
for tag, gm, om, del_ in (("ngo", api.Model.ngo(cfg), oracle.Model.ngo(cfg), 1e-4),
                          ("interp", api.Model.interp(F, b, wl.QS, wl.MS), oracle.Model.interp(F, b, wl.QS, wl.MS), 1e-6)):
    pos, d, w = wl.launch_set(512, 11)
    od = np.array([om.disp(p, dd, ww) for p, dd, ww in zip(pos, d, w)])
    ok = od[:, 8] > 0
    x, k, ww = pos[ok], od[ok, 8:9] * d[ok], w[ok]
    gg = gm.gradients(x, k, ww, del_); og = np.array([om.grad(a, b_, c, del_) for a, b_, c in zip(x, k, ww)])
    for nm, sl in (("dfdk", slice(0, 3)), ("dfdx", slice(4, 7)), ("rhs_x", slice(7, 10)), ("rhs_k", slice(10, 13))):
        e = vrel(gg[:, sl], og[:, sl]); i = np.argmax(e)
        print(tag, nm, "vec-rel max %.3e median %.3e 99pct %.3e" % (e.max(), np.median(e), np.percentile(e, 99)), "worst idx", i)
        if nm == "dfdx":
            print("   worst: gpu", gg[i, sl], "cpu", og[i, sl], "x", x[i] / wl.R_E)
    print(tag, "dfdw rel", np.max(np.abs(gg[:, 3] - og[:, 3]) / np.abs(og[:, 3])))
    args = np.concatenate([x, k, ww[:, None]], axis=1)
    gs = gm.rk_step(args, 1e-3, del_); os_ = np.array([om.step(a, 1e-3, del_) for a in args])
    for nm, o in (("rk4", 0), ("rk45_4", 7), ("rk45_5", 14)):
        # error of the increment relative to the increment
        dxg, dxo = gs[:, o:o + 3] - x, os_[:, o:o + 3] - x
        dkg, dko = gs[:, o + 3:o + 6] - k, os_[:, o + 3:o + 6] - k
        e1 = vrel(dxg, dxo); e2 = vrel(dkg, dko)
        print(tag, nm, "increment vec-rel: pos max %.2e med %.2e | k max %.2e med %.2e" % (e1.max(), np.median(e1), e2.max(), np.median(e2)),
              " state vec-rel pos %.2e k %.2e" % (vrel(gs[:, o:o + 3], os_[:, o:o + 3]).max(), vrel(gs[:, o + 3:o + 6], os_[:, o + 3:o + 6]).max()))
    p0, d0, w0 = wl.appendix_b_rays()
    for fixed in (1, 0):
        kw = dict(fixedstep=fixed, dt0=1e-3, dtmax=0.1, tmax=0.2, maxerr=5e-4, maxsteps=400, del_=del_)
        rows, nrows, stop, steps = gm.trace(p0, d0, w0, outputper=1, **kw)
        orows, onrows, ostop, osteps = om.trace(p0, d0, w0, capacity=400, **kw)
        print(tag, "fixed" if fixed else "adaptive", "row0 vg vec-rel", vrel(rows[:, 0, 7:10], orows[:, 0, 7:10]).max(), "B", vrel(rows[:, 0, 13:16], orows[:, 0, 13:16]).max(),
              "n", vrel(rows[:, 0, 10:13], orows[:, 0, 10:13]).max(), "Ns", (np.abs(rows[:, 0, 16:20] - orows[:, 0, 16:20]) / orows[:, 0, 16:20]).max())
        for r in (1, 2, 5, 10, 50, 100):
            sel = (nrows > r) & (onrows > r)
            if sel.any():
                print("   row %3d: pos %.2e  n %.2e  vg %.2e |dt| %.2e" % (r, vrel(rows[sel, r, 1:4], orows[sel, r, 1:4]).max(), vrel(rows[sel, r, 10:13], orows[sel, r, 10:13]).max(),
                      vrel(rows[sel, r, 7:10], orows[sel, r, 7:10]).max(), np.abs(rows[sel, r, 0] - orows[sel, r, 0]).max()))
