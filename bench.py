#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native many-ray Haselgrove integrator.

Contract: `python bench.py --gpus N --steps K --warmup W` (N>1: launched by torch.distributed.run,
one rank per GPU).  A "step" is one pass of the hot path over one batch of synthetic input: the whole
launch set is traced from t=0 to its stop conditions by one kernel launch.  W untimed steps, then
exactly K timed steps bracketed by barrier + synchronize; MAX over ranks; rank 0 prints ONE JSON line.

Workloads (BASELINE.json configs; --workload):
  interp256 (default)  config[2]: 1M rays, interp model on a 256^3 x 4-species ln N grid (tricubic),
                       adaptive RK45, maxsteps=256, outputper=16 -- the configuration the north-star
                       metric (ray-steps/s + HBM roofline) is quoted on.
  ngo100k              config[1]: 100k rays, Ngo model, adaptive RK45, maxsteps=512, outputper=8.
  Smaller variants for quick checks: --grid N --rays N.
Inputs are synthetic (seeded launch set + analytic plasmasphere) and resident in HBM
before the timed region starts.  N>1 is weak scaling: every rank traces its own launch set of the
same size against its own model replica, then the trajectory buffers are gathered to rank 0 (RCCL).
"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALGO_BYTES_PER_STEP = {  # algorithmic bytes per ACCEPTED ray-step
    # 44 distinct lookups x (8 corners x 8 arrays x 4 species x 8 B) + state r/w 160 B + row 256 B/outputper
    "interp": lambda outputper: 44 * 2048 + 160 + 256.0 / outputper,
    # no table: state r/w + emitted row only
    "ngo": lambda outputper: 160 + 256.0 / outputper,
    # scattered: data-dependent (visited samples x 64 B per lookup); reported from the neighbour statistics
    "scattered": lambda outputper: 160 + 256.0 / outputper,
}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="interp256", choices=["interp256", "ngo100k", "scattered825k"])
    ap.add_argument("--rays", type=int, default=0, help="override rays per GPU")
    ap.add_argument("--grid", type=int, default=0, help="override grid nodes per axis (interp)")
    ap.add_argument("--maxsteps", type=int, default=0)
    ap.add_argument("--points", type=int, default=0, help="override sample count (scattered)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU-baseline sample duration (0=skip)")
    ap.add_argument("--use-igrf", type=int, default=0, choices=[0, 1], help="IGRF main field instead of the dipole (driver flag --use_igrf)")
    ap.add_argument("--use-tsyganenko", type=int, default=0, choices=[0, 1],
                    help="add the T04_s external field (driver flag --use_tsyganenko; PARMOD = Pdyn 4, Dst -30, By 1, Bz -5, W .1-.3)")
    ap.add_argument("--no-gather", action="store_true")
    ap.add_argument("--damping-rays", type=int, default=100_000,
                    help="rays whose kept rows get the damping post-pass after the timed region (N=1 only; 0 = skip)")
    ap.add_argument("--refill", type=int, default=0)
    ap.add_argument("--streams", type=int, default=2, help="HIP streams the steps alternate on (1 = strictly serial)")
    ap.add_argument("--ray-order", type=int, default=1, choices=[0, 1],
                    help="srt_params.ray_order: 1 = the library works through the launch set sorted by launch cell "
                         "(device sort inside the timed region; SURVEY 8d allows this permutation), 0 = as given")
    return ap.parse_args()


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    # one rank per GPU.  (Rehearsal on a 1-GPU box: SRT_BENCH_BACKEND=gloo --no-gather folds the ranks onto the
    # devices that exist; RCCL itself refuses two ranks on one device.)
    backend = os.environ.get("SRT_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_

        dist = dist_
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    dev = torch.device("cuda", local_rank)

    from stanford_raytracer_amd import api, workloads as wl

    api.init(local_rank)

    # ---------------------------------------------------------------- workload
    if args.workload == "interp256":
        kind = "interp"
        nrays = args.rays or 1_000_000
        grid_n = args.grid or 256
        seed = 3
        p = api.make_params(fixedstep=0, dt0=1e-3, dtmax=0.1, maxerr=5e-4, tmax=0.5, root=2, minalt=wl.MINALT,
                            maxsteps=args.maxsteps or 256, outputper=16, del_=1e-6, refill_threshold=args.refill)
        t0 = time.time()
        F, bounds = wl.make_grid(grid_n, half_width=10.0 * wl.R_E)
        model = api.Model.interp(F, bounds, wl.QS, wl.MS)
        del F
        setup_s = time.time() - t0
        wname = "%d rays/GPU, interp_dens_model on %d^3 x4 lnN grid (tricubic), dipole B, adaptive RK45" % (nrays, grid_n)
    elif args.workload == "scattered825k":
        kind = "scattered"
        nrays = args.rays or 1_000_000
        grid_n = 0
        seed = 5
        p = api.make_params(fixedstep=0, dt0=1e-3, dtmax=0.1, maxerr=5e-4, tmax=0.5, root=2, minalt=wl.MINALT,
                            maxsteps=args.maxsteps or 64, outputper=8, del_=1e-6, refill_threshold=args.refill)
        t0 = time.time()
        npts = args.points or 825_000
        if npts == 825_000:  # 200 k uniform + 600 k importance-sampled + 25 k shell
            pts, lnN = wl.make_points_config5(5)
        else:
            pts, lnN = wl.make_points(int(npts * 0.97), npts - int(npts * 0.97), 5, half_width=10.0 * wl.R_E)
        pfile = os.path.join(tempfile.mkdtemp(), "points.txt")
        wl.write_points_file(pfile, pts, lnN, np.array([-10.0 * wl.R_E, 10.0 * wl.R_E] * 3))
        model = api.Model.scattered_file(pfile, window_scale=1.5, order=2, exact=0, local_window_scale=5.0)
        args.points_file = pfile
        setup_s = time.time() - t0
        wname = "%d rays/GPU, scattered_interp_dens_model (%d samples, order 2, window 1.5/5), dipole B, adaptive RK45" % (nrays, npts)
    else:
        kind = "ngo"
        nrays = args.rays or 100_000
        grid_n = 0
        seed = 2
        p = api.make_params(fixedstep=0, dt0=1e-3, dtmax=0.1, maxerr=5e-4, tmax=0.5, root=2, minalt=wl.MINALT,
                            maxsteps=args.maxsteps or 512, outputper=8, del_=1e-4, refill_threshold=args.refill)
        t0 = time.time()
        cfg = os.path.join(tempfile.mkdtemp(), "newray.in")
        with open(cfg, "w") as f:
            f.write(wl.NEWRAY_PLASMAPAUSE)
        model = api.Model.ngo(cfg)
        setup_s = time.time() - t0
        wname = "%d rays/GPU, ngo_dens_model, dipole B, adaptive RK45" % nrays

    if args.use_igrf or args.use_tsyganenko:
        model.set_field(use_igrf=args.use_igrf, use_tsyganenko=args.use_tsyganenko,
                        parmod=[4.0, -30.0, 1.0, -5.0, 0.132, 0.303, 0.083, 0.07, 0.211, 0.308])
        wname += (", IGRF main field" if args.use_igrf else "") + (", T04_s external field" if args.use_tsyganenko else "")
    pos0, dir0, w0 = wl.launch_set(nrays, seed + 1000 * rank)
    p.ray_order = args.ray_order
    slots = api.lib().srt_rows_per_ray(p)
    d_pos = torch.from_numpy(np.ascontiguousarray(pos0.T)).to(dev)  # SoA [3][n]
    d_dir = torch.from_numpy(np.ascontiguousarray(dir0.T)).to(dev)
    d_w = torch.from_numpy(w0).to(dev)
    # Steps are issued alternately on `--streams` HIP streams (default 2), each with its own output buffers: the
    # drain of one launch (queue empty, waves thinning out) overlaps the ramp-up of the next, and on N > 1 the RCCL
    # gather of step k overlaps the kernel of step k+1.  All K steps complete inside the timed region.
    nstream = max(1, args.streams)
    streams = [torch.cuda.Stream(dev) for _ in range(nstream)] if nstream > 1 else [torch.cuda.current_stream(dev)]
    outs = [{"rows": torch.zeros((nrays, slots, api.ROW), dtype=torch.float64, device=dev),
             "nrows": torch.zeros(nrays, dtype=torch.int32, device=dev),
             "stop": torch.zeros(nrays, dtype=torch.int32, device=dev)} for _ in range(nstream)]
    d_rows, d_nrows, d_stop = outs[0]["rows"], outs[0]["nrows"], outs[0]["stop"]
    gather_buf = None
    if dist is not None and not args.no_gather and rank == 0:
        gather_buf = [torch.empty_like(d_rows) for _ in range(world)]

    import ctypes as C

    def one_step(i, cnt):
        st, o = streams[i % nstream], outs[i % nstream]
        with torch.cuda.stream(st):
            rc = api.lib().srt_trace_batch_device(model.h, C.byref(p), nrays, d_pos.data_ptr(), d_dir.data_ptr(),
                                                  d_w.data_ptr(), o["rows"].data_ptr(), o["nrows"].data_ptr(),
                                                  o["stop"].data_ptr(), cnt.data_ptr(), st.cuda_stream)
            if rc != 0:
                raise RuntimeError(api.lib().srt_last_error().decode())
            if dist is not None and not args.no_gather:
                dist.gather(o["rows"], gather_buf, dst=0)

    def sync_all():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    torch.cuda.synchronize(dev)  # inputs are resident before anything is launched on the side streams
    wcnt = [torch.zeros(4, dtype=torch.int64, device=dev) for _ in range(args.warmup)]
    for i in range(args.warmup):
        one_step(i, wcnt[i])
    sync_all()
    cnts = [torch.zeros(4, dtype=torch.int64, device=dev) for _ in range(args.steps)]
    kernel_ms = [None] * args.steps
    LAG = 2  # a launch's duration is read two launches later (the library keeps the events of the last 4 launches)
    t_start = time.perf_counter()
    for i in range(args.steps):
        one_step(i, cnts[i])
        if i >= LAG:
            kernel_ms[i - LAG] = model.launch_ms(LAG)
    sync_all()
    elapsed = time.perf_counter() - t_start
    for i in range(max(0, args.steps - LAG), args.steps):
        kernel_ms[i] = model.launch_ms(args.steps - 1 - i)
    steps_acc = attempts = wave_attempts = 0
    for c in cnts:
        c = c.cpu().numpy()
        steps_acc += int(c[1])
        attempts += int(c[2])
        wave_attempts += int(c[3])

    tot_steps, tmax = steps_acc, elapsed
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        s = torch.tensor([steps_acc], dtype=torch.int64, device=dev)
        dist.all_reduce(s, op=dist.ReduceOp.SUM)
        tmax, tot_steps = float(t.item()), int(s.item())

    stream_gbs = None
    if rank == 0 and world == 1:
        # the box's own streaming rate ("measure a device-to-device copy and use THAT as 100 %"): read +
        # write bytes of a 2 GiB copy, outside the timed region
        try:
            src = torch.empty(1 << 28, dtype=torch.float64, device=dev)
            dst = torch.empty_like(src)
            dst.copy_(src)
            torch.cuda.synchronize(dev)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                dst.copy_(src)
            e1.record()
            torch.cuda.synchronize(dev)
            stream_gbs = 5 * 2 * src.numel() * 8 / (e0.elapsed_time(e1) * 1e-3) / 1e9
            del src, dst
        except Exception:
            stream_gbs = None
    if rank == 0:
        stop = d_stop.cpu().numpy()
        nrows = d_nrows.cpu().numpy()
        steps_per_launch = steps_acc / max(args.steps, 1)
        k_ms = float(np.mean(kernel_ms)) if kernel_ms else float("nan")
        algo = ALGO_BYTES_PER_STEP[kind](p.outputper) * steps_per_launch
        achieved = algo / (k_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic_%s.json" % args.workload)
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                if tj.get("rays") == nrays and tj.get("grid", 0) == grid_n:
                    traffic = tj["hbm_bytes_per_launch"]
            except Exception:
                traffic = None
        out = {
            "metric": "ray-steps/sec (whole node) + achieved HBM GB/s vs roofline",
            "value": tot_steps / tmax,
            "unit": "accepted ray-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * tmax / max(args.steps, 1),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": wname, "rays_per_gpu": nrays, "grid": grid_n, "maxsteps": p.maxsteps,
                       "outputper": p.outputper, "integrator": "rkf45 adaptive", "parallelism": "rays sharded x%d" % world,
                       "ray_order": "launch-cell Morton order, sorted on the device inside the timed region" if (args.ray_order and kind == "interp") else "as given",
                       "gather": bool(dist is not None and not args.no_gather), "streams": nstream},
            # achieved/frac: ALGORITHMIC bytes (every lookup counted at 2 KiB) / kernel time -- exceeds the
            # HBM peak because consecutive lookups of a ray re-read the same block.  traffic: fabric-side bytes per
            # launch from the PMC passes (profiles/traffic_*.json); traffic_GBs = traffic / kernel time is the physical
            # rate to hold against the 8 TB/s peak (traffic_frac).
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_GBs": (traffic / (k_ms * 1e-3) / 1e9) if traffic else None,
                         "traffic_frac": (traffic / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
                         "stream_copy_GBs": stream_gbs,
                         "kernel": "trace_kernel<%s,adaptive>" % kind, "kernel_ms": k_ms,
                         "algorithmic_bytes_per_accepted_step": ALGO_BYTES_PER_STEP[kind](p.outputper),
                         "accepted_steps_per_launch": steps_per_launch},
            "detail": {"attempts_per_launch": attempts / max(args.steps, 1),
                       "reject_ratio": 1.0 - steps_acc / max(attempts, 1),
                       "lane_occupancy": attempts / max(64 * wave_attempts, 1),
                       "mean_rows_per_ray": float(nrows.mean()),
                       "stopcond_hist": {str(int(k)): int(v) for k, v in zip(*np.unique(stop, return_counts=True))},
                       "model_setup_s": setup_s, "model_device_GB": model.device_bytes / 1e9},
        }
        if args.damping_rays > 0 and world == 1:
            out["detail"]["damping"] = damping_leg(args, api, model, p, slots, d_rows, d_nrows, d_w, dev, torch, nrows)
        out["cpu_baseline"] = cpu_baseline(args, kind, p, wl, pos0, dir0, w0, grid_n) if (args.cpu_seconds > 0 and world == 1) else None
        # the real reference (Fortran, one core -- its only mode), when its prebuilt harness travelled with the repo
        out["cpu_reference"] = cpu_reference(args, kind, p, wl, pos0, dir0, w0) if (args.cpu_seconds > 0 and world == 1) else None
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def damping_leg(args, api, model, p, slots, d_rows, d_nrows, d_w, dev, torch, nrows):
    """The step after the path, outside the timed region: hot-plasma damping (suprathermal electrons,
    m = -1, 0, 1, tol 1e-3: test_dampray.m's settings) along the kept rows of the first --damping-rays rays, straight
    from the row buffer the last launch left in HBM; the CPU oracle's restatement of the MATLAB scripts beside it."""
    import ctypes as C
    n = int(min(args.damping_rays, d_rows.shape[0]))
    qs, ms = model.species()
    d_rate = torch.empty((n, slots), dtype=torch.float64, device=dev)
    d_mag = torch.empty_like(d_rate)
    d_flag = torch.empty((n, slots), dtype=torch.int32, device=dev)
    dp = api.damping_params()
    st = torch.cuda.current_stream(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(dev)
    e0.record(st)
    rc = api.lib().srt_damping_device(C.byref(dp), len(qs), api._dp(api._f64(qs)), api._dp(api._f64(ms)), slots, p.outputper, n,
                                      d_rows.data_ptr(), d_nrows.data_ptr(), d_w.data_ptr(), d_rate.data_ptr(),
                                      d_mag.data_ptr(), d_flag.data_ptr(), st.cuda_stream)
    if rc != 0:
        return {"error": api.lib().srt_last_error().decode()}
    e1.record(st)
    torch.cuda.synchronize(dev)
    ms_gpu = e0.elapsed_time(e1)
    kept = (np.maximum(nrows[:n], 1) - 1) // p.outputper + 1
    nrow_eval = int((kept - 1).sum())
    flag = d_flag.cpu().numpy()
    out = {"rays": n, "rows_evaluated": nrow_eval, "kernel_ms": ms_gpu, "rows_per_s": nrow_eval / (ms_gpu * 1e-3),
           "flags": {str(int(k)): int(v) for k, v in zip(*np.unique(flag, return_counts=True))},
           "final_magnitude_median": float(np.nanmedian(d_mag.cpu().numpy()[np.arange(n), kept - 1]))}
    if args.cpu_seconds > 0:
        try:
            from oracle import oracle
            nc = int(min(n, 24))
            rows_h = d_rows[:nc].cpu().numpy()
            t0 = time.time()
            rk, _, _ = oracle.damping(qs, ms, p.outputper, rows_h, nrows[:nc], d_w[:nc].cpu().numpy())
            dt = time.time() - t0
            ne = int((kept[:nc] - 1).sum())
            g = d_rate[:nc].cpu().numpy()
            ok = np.isfinite(rk) & np.isfinite(g) & (rk != 0)
            out["cpu_port"] = {"rows_per_s": ne / max(dt, 1e-9), "cores": 1, "rows": ne,
                               "max_rel_diff_vs_gpu": float(np.max(np.abs(g[ok] - rk[ok]) / np.abs(rk[ok]))) if ok.any() else None,
                               "median_rel_diff_vs_gpu": float(np.median(np.abs(g[ok] - rk[ok]) / np.abs(rk[ok]))) if ok.any() else None}
        except Exception as e:  # pragma: no cover
            out["cpu_port"] = {"error": str(e)}
    return out


def cpu_baseline(args, kind, p, wl, pos0, dir0, w0, grid_n):
    """The CPU oracle (a port of the reference's algorithm, oracle/srt_oracle.c) timed on the host cores on
    a bounded sample of the SAME workload.  Only the checker/baseline leg touches oracle/."""
    try:
        from oracle import oracle
    except Exception as e:  # pragma: no cover
        return {"error": "oracle unavailable: %s" % e}
    cores = max(1, min(16, os.cpu_count() or 1))
    t0 = time.time()
    if kind == "interp":
        F, bounds = wl.make_grid(grid_n, half_width=10.0 * wl.R_E)
        om = oracle.Model.interp(F, bounds, wl.QS, wl.MS)
        del F
    elif kind == "scattered":
        om = oracle.Model.scattered_file(args.points_file, window_scale=1.5, order=2, exact=0, local_window_scale=5.0)
    else:
        cfg = os.path.join(tempfile.mkdtemp(), "newray.in")
        with open(cfg, "w") as f:
            f.write(wl.NEWRAY_PLASMAPAUSE)
        om = oracle.Model.ngo(cfg)
    if args.use_tsyganenko:
        return {"error": "the CPU port has no T04_s (parity for it is held against goldens of the reference, tests/test_t04.py)"}
    if args.use_igrf:
        om.set_igrf()
    setup = time.time() - t0
    kw = dict(dt0=p.dt0, dtmax=p.dtmax, tmax=p.tmax, maxerr=p.maxerr, minalt=p.minalt, del_=p.del_,
              maxsteps=p.maxsteps, root=p.root, fixedstep=p.fixedstep)
    # calibrate on a few rays, then size the sample for ~cpu_seconds
    n0 = min(4 * cores, len(w0))
    t0 = time.time()
    _, _, _, s0 = om.trace(pos0[:n0], dir0[:n0], w0[:n0], capacity=0, nthreads=cores, **kw)
    dt0 = max(time.time() - t0, 1e-3)
    n1 = int(min(len(w0), max(n0, n0 * args.cpu_seconds / dt0)))
    t0 = time.time()
    _, _, _, s1 = om.trace(pos0[:n1], dir0[:n1], w0[:n1], capacity=0, nthreads=cores, **kw)
    dt1 = time.time() - t0
    return {"value": s1 / dt1, "unit": "accepted ray-steps/s", "cores": cores, "kind": "port",
            "sample": "first %d rays of the same launch set, same model and integrator parameters, %d accepted steps in %.1f s "
                      "(oracle/srt_oracle.c, pthreads over rays; model setup %.1f s excluded)" % (n1, s1, dt1, setup)}


def cpu_reference(args, kind, p, wl, pos0, dir0, w0):
    """The reference itself -- oracle/_ref/ref_harness = rareid2/Stanford_Raytracer's own raytracer_run + adapters
    compiled with flang (oracle/build_ref.py), single-threaded as the reference is -- timed around its ray loop on
    the first rays of the same launch set.  The interp workload is served from a 64^3 text grid of the same analytic
    plasmasphere: the Fortran adapter needs minutes to parse the 1.7 GB text form of the 256^3 grid (its per-step
    cost does not depend on the grid size beyond the O(nx) cell search)."""
    try:
        from oracle import refharness
    except Exception as e:  # pragma: no cover
        return {"error": "refharness unavailable: %s" % e}
    if not refharness.available() or kind == "scattered":
        return None
    igrf = {"use_igrf": 1} if args.use_igrf else {}
    if args.use_tsyganenko:
        return None  # the harness runs T04_s with its own fixed PARMOD
    td = tempfile.mkdtemp()
    if kind == "interp":
        gn = 64
        F, bounds = wl.make_grid(gn, half_width=10.0 * wl.R_E)
        gfile = os.path.join(td, "grid64.txt")
        wl.write_grid_file(gfile, F, bounds)
        model = dict({"kind": 3, "file": gfile}, **igrf)
        note = "interp model on a %d^3 text grid of the same plasmasphere" % gn
    else:
        cfg = os.path.join(td, "newray.in")
        with open(cfg, "w") as f:
            f.write(wl.NEWRAY_PLASMAPAUSE)
        model = dict({"kind": 1, "file": cfg}, **igrf)
        note = "ngo model"
    kw = dict(dt0=p.dt0, dtmax=p.dtmax, tmax=p.tmax, maxerr=p.maxerr, minalt=p.minalt, maxsteps=p.maxsteps,
              root=p.root, fixedstep=p.fixedstep)
    kw["del"] = p.del_
    budget = min(args.cpu_seconds, 12.0)
    n0 = min(8, len(w0))
    rays = np.concatenate([pos0, dir0, w0[:, None]], axis=1)
    _, t0 = refharness.run_rays(model, rays[:n0], **kw)
    if not t0 or t0["seconds"] <= 0:
        return {"error": "reference harness gave no timing"}
    n1 = int(min(len(w0), max(n0, n0 * budget / max(t0["seconds"], 1e-3))))
    _, t1 = refharness.run_rays(model, rays[:n1], **kw)
    return {"value": t1["steps"] / t1["seconds"], "unit": "accepted ray-steps/s", "cores": 1, "kind": "reference",
            "sample": "first %d rays of the same launch set, same integrator parameters, %s: %d accepted steps in %.1f s "
                      "inside the reference's ray loop (oracle/_ref/ref_harness, flang -O3)" % (n1, note, t1["steps"], t1["seconds"])}


if __name__ == "__main__":
    main()
