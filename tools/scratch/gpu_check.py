"""Quick GPU-vs-oracle comparison (development aid; the real gates are tests/ -m gpu)."""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle
from stanford_raytracer_amd import api, workloads as wl

def rel(a, b):
    return np.abs(a - b) / np.maximum(np.abs(b), 1e-300)

def report(name, mine, ref, cols=None):
    e = rel(mine, ref)
    e[np.isnan(mine) & np.isnan(ref)] = 0
    print("%-28s max rel %.3e   median %.3e  nanmismatch %d" % (name, np.nanmax(e), np.nanmedian(e), (np.isnan(mine) != np.isnan(ref)).sum()))
    if cols:
        print("     per col:", " ".join("%.1e" % v for v in np.nanmax(e, axis=0)))

def ladder(tag, gm, om, del_, n=512):
    pos, d, w = wl.launch_set(n, 11)
    gp = gm.plasma_params(pos)
    op = np.array([np.concatenate(om.plasma_params(p)) for p in pos])
    report(tag + " params", gp, op, True)
    # states on the dispersion surface
    od = np.array([om.disp(p, dd, ww) for p, dd, ww in zip(pos, d, w)])
    ok = od[:, 8] > 0
    x, k, ww = pos[ok], od[ok, 8:9] * d[ok], w[ok]
    gd = gm.dispersion(x, k, ww)
    od2 = np.array([om.disp(a, b, c) for a, b, c in zip(x, k, ww)])
    report(tag + " disp", gd, od2, True)
    gg = gm.gradients(x, k, ww, del_)
    og = np.array([om.grad(a, b, c, del_) for a, b, c in zip(x, k, ww)])
    report(tag + " grad", gg, og, True)
    args = np.concatenate([x, k, ww[:, None]], axis=1)
    gs = gm.rk_step(args, 1e-3, del_)
    os_ = np.array([om.step(a, 1e-3, del_) for a in args])
    report(tag + " step", gs, os_, True)

def traces(tag, gm, om, del_, fixed):
    p0, d0, w0 = wl.appendix_b_rays()
    kw = dict(fixedstep=fixed, dt0=1e-3, dtmax=0.1, tmax=0.2, maxerr=5e-4, maxsteps=400, del_=del_)
    t = time.time()
    rows, nrows, stop, steps = gm.trace(p0, d0, w0, outputper=1, **kw)
    print(tag, "fixed" if fixed else "adaptive", "gpu steps", steps, "kernel ms", gm.last_kernel_ms(), "wall", time.time() - t)
    orows, onrows, ostop, osteps = om.trace(p0, d0, w0, capacity=400, **kw)
    print("   nrows gpu", nrows.tolist()); print("   nrows cpu", onrows.tolist())
    print("   stop gpu", stop.tolist(), "cpu", ostop.tolist())
    for r in (1, 10, 100):
        sel = (nrows > r) & (onrows > r)
        if sel.any():
            e = rel(rows[sel, r, 1:4], orows[sel, r, 1:4]).max()
            en = rel(rows[sel, r, 10:13], orows[sel, r, 10:13]).max()
            et = np.abs(rows[sel, r, 0] - orows[sel, r, 0]).max()
            print("   row %3d: pos rel %.2e  n rel %.2e  |dt| %.2e" % (r, e, en, et))

if __name__ == "__main__":
    api.init(0)
    print(api.device_info())
    td = tempfile.mkdtemp()
    cfg = os.path.join(td, "newray.in"); open(cfg, "w").write(wl.NEWRAY_PLASMAPAUSE)
    cfg2 = os.path.join(td, "newray2.in"); open(cfg2, "w").write(wl.NEWRAY_DUCTS)
    gm, om = api.Model.ngo(cfg), oracle.Model.ngo(cfg)
    ladder("ngo", gm, om, 1e-4)
    gm2, om2 = api.Model.ngo(cfg2), oracle.Model.ngo(cfg2)
    ladder("ngo-ducts", gm2, om2, 1e-4, 256)
    traces("ngo", gm, om, 1e-4, 1); traces("ngo", gm, om, 1e-4, 0)
    F, b = wl.make_grid(24, half_width=5 * wl.R_E)
    gi, oi = api.Model.interp(F, b, wl.QS, wl.MS), oracle.Model.interp(F, b, wl.QS, wl.MS)
    ladder("interp", gi, oi, 1e-6)
    rng = np.random.default_rng(3)
    edge = rng.uniform(-5.6 * wl.R_E, 5.6 * wl.R_E, (512, 3))
    report("interp params (edges)", gi.plasma_params(edge), np.array([np.concatenate(oi.plasma_params(p)) for p in edge]), True)
    traces("interp", gi, oi, 1e-6, 1); traces("interp", gi, oi, 1e-6, 0)
