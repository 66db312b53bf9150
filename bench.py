#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native many-ray Haselgrove integrator.

Contract: `python bench.py --gpus N --steps K --warmup W` (N>1: launched by torch.distributed.run, one rank per
GPU; --gpus must equal WORLD_SIZE).  A "step" is one pass of the hot path over one batch of synthetic input: the
whole launch set is traced from t=0 to its stop conditions (one kernel launch per GPU), and at N>1 the kept
trajectory rows are gathered to rank 0.  W untimed steps, then exactly K timed steps bracketed by barrier +
synchronize; MAX over ranks; rank 0 prints ONE JSON line.

Workloads (BASELINE.json configs; --workload):
  interp256 (default)  config[2]: 1M rays per GPU, interp model on a 256^3 x 4-species ln N grid (tricubic),
                       adaptive RK45, maxsteps=256, outputper=16 -- the configuration the north-star metric is
                       quoted on.  N>1: ONE launch set of N x 1M rays (seed 3) cut into contiguous shards (weak).
  interp4m             config[3]: 4M rays (seed 4) sharded over the N ranks, strong scaling, gather timed apart.
  ngo100k              config[1]: 100k rays, Ngo model, adaptive RK45, maxsteps=512, outputper=8 (compute-bound).
  scattered825k        config[4]: 1M rays, scattered model on the 825k-sample set, maxsteps=64, outputper=8.
  interp_igrf200k / interp_t04_64k   config[2]'s set and grid with --use_igrf=1 / --use_tsyganenko=1 (200k / 64k rays).
The default invocation additionally times one short run of the other configs (`other_configs`), so that every
BASELINE config is driver-timed.  Inputs are synthetic (seeded launch set + analytic plasmasphere)
and resident in HBM before the timed region starts.

Multi-GPU: every step goes through stanford_raytracer_amd.parallel.trace_sharded (shard -> trace -> pack ->
variable-length gather), the same function the gloo and 2-rank GPU tests run.

Roofline (N=1): `achieved` = bytes the kernel moved through the L2's fabric side per launch (rocprofv3 PMC,
FETCH_SIZE x2 + WRITE_SIZE, collected LIVE by two child passes of this script under rocprofv3 before this process
touches the GPU) / the kernel's HIP-event duration in the timed region; `frac` = achieved / 8 TB/s.  The SURVEY 8d
algorithmic figure (44 lookups x 2 KiB per accepted step) is reported beside it as `algorithmic_GBs` -- it is NOT a
fraction of anything: the 7-point stencil shares one block, so the kernel moves far fewer bytes than that figure.
"""
import argparse
import csv
import glob
import hashlib
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
# (RCCL between processes shares device memory through dmabuf handles on this driver stack; the legacy IPC mode fails with
# hipIpcGetMemHandle: invalid argument.  The pool exports this already; set here for launches that do not inherit it.)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8.0 TB/s spec
FP64_VALU_PEAK_TFLOPS = 78.6  # MI355X_MICROARCH.md: fp64 vector peak
# MI355X_MICROARCH.md, "Indexed rows: gather into LDS": what a gather of a 151 MB table into LDS reaches chip-wide
# (the resident rays' coefficient blocks are ~128 MiB and are served largely by the Infinity Cache)
LDS_GATHER_CEILING_GBS = (7400.0, 7900.0)
ALGO_INTERP = lambda outputper: 44 * 2048 + 160 + 256.0 / outputper  # SURVEY.md 8(d), bytes per ACCEPTED step

WORKLOADS = {
    "interp256": dict(kind="interp", rays=1_000_000, seed=3, scaling="weak", maxsteps=256, outputper=16, del_=1e-6),
    "interp4m": dict(kind="interp", rays=4_000_000, seed=4, scaling="strong", maxsteps=256, outputper=16, del_=1e-6),
    "ngo100k": dict(kind="ngo", rays=100_000, seed=2, scaling="weak", maxsteps=512, outputper=8, del_=1e-4),
    "scattered825k": dict(kind="scattered", rays=1_000_000, seed=5, scaling="weak", maxsteps=64, outputper=8, del_=1e-6),
    # config[2]'s launch set and grid with the reference's field options (short launches: the field tails cost
    # 4 x / 100 x the dipole kernel's time)
    "interp_igrf200k": dict(kind="interp", rays=200_000, seed=3, scaling="weak", maxsteps=256, outputper=16, del_=1e-6, field=(1, 0)),
    "interp_t04_64k": dict(kind="interp", rays=65_536, seed=3, scaling="weak", maxsteps=256, outputper=16, del_=1e-6, field=(0, 1)),
}
FP32_VALU_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: fp32 vector peak
VALU_MIX_COUNTERS = ["SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_FMA_F32",
                     "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_ADD_F32", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES"]
T04_PARMOD = [4.0, -30.0, 1.0, -5.0, 0.132, 0.303, 0.083, 0.07, 0.211, 0.308]  # Pdyn, Dst, ByIMF, BzIMF, W1..W6


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="interp256", choices=sorted(WORKLOADS))
    ap.add_argument("--rays", type=int, default=0, help="override rays (per GPU for weak workloads, total for interp4m)")
    ap.add_argument("--grid", type=int, default=0, help="override grid nodes per axis (interp)")
    ap.add_argument("--maxsteps", type=int, default=0)
    ap.add_argument("--points", type=int, default=0, help="override sample count (scattered)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU-baseline sample duration (0=skip)")
    ap.add_argument("--use-igrf", type=int, default=0, choices=[0, 1], help="IGRF main field instead of the dipole (driver flag --use_igrf)")
    ap.add_argument("--use-tsyganenko", type=int, default=0, choices=[0, 1],
                    help="add the T04_s external field (driver flag --use_tsyganenko; PARMOD = Pdyn 4, Dst -30, By 1, Bz -5, W .1-.3)")
    ap.add_argument("--no-gather", action="store_true", help="N>1: trace only (no pack, no gather)")
    ap.add_argument("--damping-rays", type=int, default=100_000,
                    help="rays whose kept rows get the damping post-pass after the timed region (N=1 only; 0 = skip)")
    ap.add_argument("--refill", type=int, default=0)
    ap.add_argument("--streams", type=int, default=1,
                    help="HIP streams the steps alternate on at N=1 (1 = strictly serial, the default: kernel time == step "
                         "time; 2 lets the drain of one launch overlap the ramp-up of the next, +5 %%)")
    ap.add_argument("--ray-order", type=int, default=1, choices=[0, 1, 2],
                    help="srt_params.ray_order: 1 = the library works through the launch set sorted by launch cell "
                         "(device sort inside the timed region; SURVEY 8d allows this permutation), 0 = as given")
    ap.add_argument("--traffic", default="auto", choices=["auto", "live", "file", "off"],
                    help="where roofline.traffic comes from: live = two rocprofv3 PMC child passes of this script "
                         "(FETCH_SIZE, WRITE_SIZE) before the timed run; file = profiles/traffic_<workload>.json if it was "
                         "collected for these kernel sources; auto = live at N=1 when rocprofv3 exists, else file")
    ap.add_argument("--other-configs", type=int, default=-1, choices=[-1, 0, 1],
                    help="time one short run of the other BASELINE configs as well (-1 = only in the default invocation)")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)  # internal: one bare launch under rocprofv3
    return ap.parse_args(argv)


# --------------------------------------------------------------------------------------------- provenance
def kernel_source_hash():
    """sha256 over the sources the kernels are compiled from: a traffic profile belongs to exactly one such hash."""
    h = hashlib.sha256()
    # what is compiled into device code: the kernel headers and the launch file (not the host-only file formats / CLI)
    csrc = os.path.join(ROOT, "stanford_raytracer_amd", "csrc")
    files = sorted(glob.glob(os.path.join(csrc, "*.hpp")) + glob.glob(os.path.join(csrc, "*.h")) + glob.glob(os.path.join(csrc, "*.hip")))
    files += [os.path.join(ROOT, "include", "srt.h")]
    for f in files:
        if os.path.isfile(f):
            h.update(os.path.basename(f).encode())
            h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


# --------------------------------------------------------------------------------------------- PMC child passes
def pmc_pass(counters, child_args, timeout=420):
    """One rocprofv3 --pmc pass over `python3 bench.py --pmc-child <child_args>`: {counter: sum over trace_kernel
    dispatches}, plus "_dispatches".  Runs as a child process; call before this process initialises the GPU."""
    exe = shutil.which("rocprofv3")
    if exe is None:
        return None, "rocprofv3 not found"
    out = tempfile.mkdtemp(prefix="srt_pmc_", dir="/tmp")
    cmd = [exe, "--pmc"] + list(counters) + ["--kernel-trace", "--output-format", "csv", "-d", out, "--",
                                              sys.executable, os.path.abspath(__file__), "--pmc-child"] + child_args
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    # if this process itself runs under a profiler, the child pass must not inherit its tool libraries
    for k in list(env):
        if k in ("HSA_TOOLS_LIB", "HSA_TOOLS_REPORT_LOAD_FAILURE") or k.startswith(("ROCP_", "ROCPROF", "ROCTRACER_", "ROCPROFILER_")):
            env.pop(k)
    if "LD_PRELOAD" in env:  # only the profiler's own entries; anything else preloaded on this machine stays
        keep = [x for x in env["LD_PRELOAD"].replace(":", " ").split() if "rocprof" not in x.lower() and "roctracer" not in x.lower()]
        if keep:
            env["LD_PRELOAD"] = ":".join(keep)
        else:
            env.pop("LD_PRELOAD")
    try:
        r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=timeout)
    except subprocess.TimeoutExpired:
        shutil.rmtree(out, ignore_errors=True)
        return None, "rocprofv3 pass timed out after %d s" % timeout
    vals, ndisp = {}, set()
    for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if "trace_kernel" in row["Kernel_Name"]:
                vals[row["Counter_Name"]] = vals.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
                ndisp.add(row["Dispatch_Id"])
    shutil.rmtree(out, ignore_errors=True)
    if r.returncode != 0 or not vals:
        return None, "rocprofv3 pass failed (rc %d): %s" % (r.returncode, r.stdout.decode(errors="replace")[-300:])
    vals["_dispatches"] = len(ndisp)
    for line in r.stdout.decode(errors="replace").splitlines():
        if line.startswith('{"pmc_child"'):
            vals["_child"] = json.loads(line)["pmc_child"]
    return vals, None


def child_args_for(args, workload, rays=0):
    a = ["--workload", workload, "--steps", "1", "--warmup", "0", "--ray-order", str(args.ray_order)]
    if rays or args.rays:
        a += ["--rays", str(rays or args.rays)]
    for k, v in (("--grid", args.grid), ("--maxsteps", args.maxsteps), ("--points", args.points), ("--refill", args.refill)):
        if v:
            a += [k, str(v)]
    if args.use_igrf:
        a += ["--use-igrf", "1"]
    if args.use_tsyganenko:
        a += ["--use-tsyganenko", "1"]
    return a


def progress(msg):
    if os.environ.get("RANK", "0") == "0":
        sys.stderr.write("bench.py: %s\n" % msg)
        sys.stderr.flush()


def live_traffic(args, workload, rays=0):
    """FETCH_SIZE and WRITE_SIZE in SEPARATE passes (they do not fit one pass on gfx950) with the corrections of
    MI355X_MICROARCH.md (HBM section): both are KiB; FETCH_SIZE tallies 128-B requests at 64 B -> x2; WRITE_SIZE as read."""
    ca = child_args_for(args, workload, rays)
    progress("counter passes (rocprofv3 --pmc FETCH_SIZE, then WRITE_SIZE) over one launch of %s" % workload)
    f, err = pmc_pass(["FETCH_SIZE"], ca)
    if f is None:
        return None, err
    w, err = pmc_pass(["WRITE_SIZE"], ca)
    if w is None:
        return None, err
    if f["_dispatches"] != 1 or w["_dispatches"] != 1:
        return None, "expected one trace_kernel dispatch per pass, saw %d / %d" % (f["_dispatches"], w["_dispatches"])
    return {"FETCH_SIZE_KB": f["FETCH_SIZE"], "WRITE_SIZE_KB": w["WRITE_SIZE"],
            "bytes_per_launch": f["FETCH_SIZE"] * 1024.0 * 2.0 + w["WRITE_SIZE"] * 1024.0,
            "accepted_steps_of_counted_launch": (f.get("_child") or {}).get("accepted"),
            "source": "live: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE child passes of this run"}, None


def file_traffic(workload, rays, grid_n, khash):
    """profiles/traffic_<workload>.json, only if it was collected for exactly these kernel sources and this size."""
    path = os.path.join(ROOT, "profiles", "traffic_%s.json" % workload)
    if not os.path.exists(path):
        return None, "no %s" % os.path.relpath(path, ROOT)
    try:
        tj = json.load(open(path))
    except Exception as e:
        return None, "unreadable %s: %s" % (path, e)
    if tj.get("kernel_source_sha16") != khash:
        return None, "STALE: %s was collected for kernel sources %s, this build is %s" % (
            os.path.relpath(path, ROOT), tj.get("kernel_source_sha16"), khash)
    if tj.get("rays") != rays or tj.get("grid", 0) != grid_n:
        return None, "%s is for %s rays / grid %s" % (os.path.relpath(path, ROOT), tj.get("rays"), tj.get("grid"))
    return {"FETCH_SIZE_KB": tj["FETCH_SIZE_KB"], "WRITE_SIZE_KB": tj["WRITE_SIZE_KB"],
            "bytes_per_launch": tj["hbm_bytes_per_launch"], "accepted_steps_of_counted_launch": tj.get("accepted_steps_per_launch"),
            "source": "file: %s (same kernel sources %s)" % (os.path.relpath(path, ROOT), khash)}, None


# --------------------------------------------------------------------------------------------- workloads
class Ctx:
    """Per-process state shared by the workloads of one bench run (device, ranks, cached models)."""

    def __init__(self, args, torch, dev, dist, rank, world):
        self.args, self.torch, self.dev, self.dist, self.rank, self.world = args, torch, dev, dist, rank, world
        self.models = {}
        self.field = {}  # id(model) -> (use_igrf, use_tsyganenko) currently set on it
        self.tmp = tempfile.mkdtemp(prefix="srt_bench_")


def make_model(ctx, kind, grid_n, npts):
    from stanford_raytracer_amd import api, workloads as wl

    key = (kind, grid_n, npts)
    if key in ctx.models:
        return ctx.models[key] + (0.0,)
    t0 = time.time()
    extra = {}
    if kind == "interp":
        F, bounds = wl.make_grid(grid_n, half_width=10.0 * wl.R_E)
        model = api.Model.interp(F, bounds, wl.QS, wl.MS)
        del F
    elif kind == "scattered":
        if npts == 825_000:  # 200 k uniform + 600 k importance-sampled + 25 k shell
            pts, lnN = wl.make_points_config5(5)
        else:
            pts, lnN = wl.make_points(int(npts * 0.97), npts - int(npts * 0.97), 5, half_width=10.0 * wl.R_E)
        pfile = os.path.join(ctx.tmp, "points_%d.bin" % npts)
        api.write_points_file(pfile, np.concatenate([pts, lnN], axis=1), np.array([-10.0 * wl.R_E, 10.0 * wl.R_E] * 3),
                              wl.QS, wl.MS, binary=True)
        model = api.Model.scattered_file(pfile, window_scale=1.5, order=2, exact=0, local_window_scale=5.0)
        extra["points_file"] = pfile
    else:
        cfg = os.path.join(ctx.tmp, "newray.in")
        with open(cfg, "w") as f:
            f.write(wl.NEWRAY_PLASMAPAUSE)
        model = api.Model.ngo(cfg)
    ctx.models[key] = (model, extra)
    return model, extra, time.time() - t0


def field_of(args, W):
    """(use_igrf, use_tsyganenko) of a workload: its own, else the command line's."""
    return W.get("field", (args.use_igrf, args.use_tsyganenko))


def run_workload(ctx, name, steps, warmup, rays_override=0, nstream=1, keep=False):
    """W untimed + K timed steps of one workload on this rank's shard.  Returns a dict (all ranks; whole-job numbers
    are reduced over ranks) -- or, with keep=True, also the DeviceBatch so the caller can post-process its rows."""
    from stanford_raytracer_amd import api, parallel, workloads as wl
    from stanford_raytracer_amd.device_batch import DeviceBatch

    args, torch, dev, dist, rank, world = ctx.args, ctx.torch, ctx.dev, ctx.dist, ctx.rank, ctx.world
    W = WORKLOADS[name]
    kind = W["kind"]
    grid_n = (args.grid or 256) if kind == "interp" else 0
    npts = (args.points or 825_000) if kind == "scattered" else 0
    per_gpu = rays_override or W["rays"]
    total = per_gpu * world if W["scaling"] == "weak" else per_gpu
    p = api.make_params(fixedstep=0, dt0=1e-3, dtmax=0.1, maxerr=5e-4, tmax=0.5, root=2, minalt=wl.MINALT,
                        maxsteps=args.maxsteps or W["maxsteps"], outputper=W["outputper"], del_=W["del_"],
                        refill_threshold=args.refill, ray_order=args.ray_order)
    model, extra, setup_s = make_model(ctx, kind, grid_n, npts)
    fld = field_of(args, W)
    if fld != ctx.field.get(id(model), (0, 0)):  # (the models are cached across workloads: the option is a switch on the handle)
        model.set_field(use_igrf=fld[0], use_tsyganenko=fld[1], parmod=T04_PARMOD if fld[1] else None)
        ctx.field[id(model)] = fld
    # ONE launch set for the whole job, cut into contiguous shards (ceil(n/world) rays per rank)
    pos0, dir0, w0 = wl.launch_set(total, W["seed"])
    lo, hi = parallel.shard_bounds(total, rank, world)
    nstream = max(1, nstream) if world == 1 else 1
    batch = DeviceBatch(model, p, pos0[lo:hi], dir0[lo:hi], w0[lo:hi], dev, nbuf=nstream)
    streams = [torch.cuda.Stream(dev) for _ in range(nstream)] if nstream > 1 else [torch.cuda.current_stream(dev)]
    gather = world > 1 and not args.no_gather
    pack = lambda rows, nrows: parallel.pack_rows_device(rows, nrows, p.outputper, slot=0)  # (one reused buffer: no allocation per step)

    def sync_all():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    last_gathered = [None]

    def one_step(i, cnt, tm):
        if gather:  # shard -> trace -> pack -> variable-length gather to rank 0: the tested multi-GPU path
            batch.out[0]["cnt"] = cnt
            last_gathered[0] = parallel.trace_sharded(dist, total, lambda a, b: batch.trace(0), pack, dst=0, timings=tm)
        else:
            with torch.cuda.stream(streams[i % nstream]):
                batch.launch(i % nstream, streams[i % nstream], counters=cnt)

    torch.cuda.synchronize(dev)  # inputs are resident before anything is launched
    for i in range(warmup):
        one_step(i, torch.zeros(4, dtype=torch.int64, device=dev), {})
    sync_all()
    cnts = [torch.zeros(4, dtype=torch.int64, device=dev) for _ in range(steps)]
    tms = [dict() for _ in range(steps)]
    kernel_ms = [None] * steps
    LAG = 2  # a launch's duration is read two launches later (the library keeps the events of the last 4 launches)
    t_start = time.perf_counter()
    for i in range(steps):
        one_step(i, cnts[i], tms[i])
        if gather:
            kernel_ms[i] = model.launch_ms(0) if hi > lo else 0.0  # trace_sharded has synchronised
        elif i >= LAG:
            kernel_ms[i - LAG] = model.launch_ms(LAG)
    sync_all()
    elapsed = time.perf_counter() - t_start
    if not gather:
        for i in range(max(0, steps - LAG), steps):
            kernel_ms[i] = model.launch_ms(steps - 1 - i)
    c = np.sum([x.cpu().numpy() for x in cnts], axis=0) if steps else np.zeros(4, dtype=np.int64)
    red = np.array([elapsed, float(np.mean(kernel_ms)) if steps else 0.0,
                    float(np.mean([t.get("gather_s", 0.0) for t in tms])) if steps else 0.0,
                    float(np.mean([t.get("pack_s", 0.0) for t in tms])) if steps else 0.0])
    tot = np.array([int(c[1]), int(c[2]), int(c[3])], dtype=np.int64)
    if dist is not None:
        rdev = dev if dist.get_backend() == "nccl" else torch.device("cpu")  # (gloo rehearsal: host tensors)
        t = torch.tensor(red, dtype=torch.float64, device=rdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        s = torch.tensor(tot, dtype=torch.int64, device=rdev)
        dist.all_reduce(s, op=dist.ReduceOp.SUM)
        red, tot = t.cpu().numpy(), s.cpu().numpy()
    o = batch.out[0]
    res = {
        "workload": name, "kind": kind, "grid": grid_n, "points": npts, "params": p, "model": model, "extra": extra,
        "total_rays": total, "rays_this_rank": hi - lo, "scaling": W["scaling"], "steps": steps, "warmup": warmup,
        "elapsed_s": float(red[0]), "ms_per_step": 1e3 * float(red[0]) / max(steps, 1),
        "kernel_ms": float(red[1]), "kernel_ms_rank0": float(np.mean(kernel_ms)) if steps else 0.0,
        "gather_ms": 1e3 * float(red[2]) if gather else None, "pack_ms": 1e3 * float(red[3]) if gather else None,
        "gather_bytes": int(np.mean([t.get("gather_bytes", 0) for t in tms])) if (gather and steps) else None,
        "accepted": int(tot[0]), "attempts": int(tot[1]), "wave_attempts": int(tot[2]),
        "value": float(tot[0]) / max(float(red[0]), 1e-12), "setup_s": setup_s, "nstream": nstream, "gather": gather,
        "stop": o["stop"].cpu().numpy(), "nrows": o["nrows"].cpu().numpy(),
        "launch": (pos0[lo:hi], dir0[lo:hi], w0[lo:hi]), "field": fld,
    }
    if gather and rank == 0 and last_gathered[0] is not None:
        # what rank 0 holds after the last step's gather, as a number a one-GPU run of the same launch set must reproduce
        pk, nr_all, st_all = last_gathered[0]
        res["rows_checksum"] = rows_checksum(torch, pk, nr_all, st_all)
        res["gathered_rows"] = int(pk.shape[0])
    if world == 1 and kind == "interp" and steps:
        # the fixed cost the multi-GPU step adds on every rank: packing the kept rows of this launch (srt_pack_rows_device)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        parallel.pack_rows_device(o["rows"], o["nrows"], p.outputper, slot=0)  # (allocates the reused buffer)
        torch.cuda.synchronize(dev)
        e0.record()
        pk, off = parallel.pack_rows_device(o["rows"], o["nrows"], p.outputper, slot=0)
        e1.record()
        torch.cuda.synchronize(dev)
        res["pack_ms_one_gpu"] = e0.elapsed_time(e1)
        res["packed_bytes_one_gpu"] = int(off[-1]) * 160
        res["rows_checksum"] = rows_checksum(torch, pk[:int(off[-1])], o["nrows"], o["stop"])
        res["gathered_rows"] = int(off[-1])
        parallel.release_pack_buffers(dev)
    if world == 1 and kind == "ngo" and steps and hi > lo:
        # what bounds this launch from below: its longest rays alone (<= 8 rays = one wave in tail mode, srt_models.hpp):
        # their sequential attempts x the tail-mode trip time
        nr = res["nrows"]
        longest = np.argsort(-nr.astype(np.int64), kind="stable")[:8]
        sub = DeviceBatch(model, p, pos0[lo:hi][longest], dir0[lo:hi][longest], w0[lo:hi][longest], dev)
        sub.launch(0)
        torch.cuda.synchronize(dev)
        sub.launch(0)
        torch.cuda.synchronize(dev)
        c8 = sub.out[0]["cnt"].cpu().numpy()
        res["latency_floor_ms"] = model.launch_ms(0)
        res["latency_floor_note"] = ("the launch's %d longest rays alone in one wave (tail mode): %d sequential attempt trips, "
                                     "%.1f us per trip" % (len(longest), int(c8[3]), 1e3 * res["latency_floor_ms"] / max(int(c8[3]), 1)))
    if keep:
        res["batch"] = batch
    return res


def rows_checksum(torch, packed, nrows, stop):
    """Position-weighted 64-bit sum (wrap-around) over the BITS of the packed kept rows [total, 20], the per-ray row counts and
    stop codes, in ray order: equal for a one-GPU launch and for the gather of any number of shards iff every row of every ray
    is bit-identical and in its place (rays are independent: raytracer_driver.f95:1144-1232 is a serial loop with no carried
    state).  Computed where the tensors live."""
    def wsum(t):
        v = t.contiguous().view(torch.int64).reshape(-1) if t.dtype == torch.float64 else t.reshape(-1).to(torch.int64)
        if v.numel() == 0:
            return 0
        wgt = torch.arange(v.numel(), dtype=torch.int64, device=v.device) % 1000003 + 1
        return int((v * wgt).sum().item())
    tot = (wsum(packed) * 3 + wsum(nrows) * 5 + wsum(stop) * 7) & 0xFFFFFFFFFFFFFFFF
    return "%016x" % tot


def pipelined_pass(ctx, name, rays_override=0, steps=3):
    """`steps` steps through parallel.trace_sharded_pipelined (two sets of output buffers; the pack + gather of step k on a
    side stream while step k + 1 traces), bracketed like the timed region.  Reported beside the serial headline; a failure
    is reported as text, it does not touch the headline."""
    from stanford_raytracer_amd import api, parallel, workloads as wl
    from stanford_raytracer_amd.device_batch import DeviceBatch

    args, torch, dev, dist, rank, world = ctx.args, ctx.torch, ctx.dev, ctx.dist, ctx.rank, ctx.world
    try:
        W = WORKLOADS[name]
        kind = W["kind"]
        per_gpu = rays_override or W["rays"]
        total = per_gpu * world if W["scaling"] == "weak" else per_gpu
        p = api.make_params(fixedstep=0, dt0=1e-3, dtmax=0.1, maxerr=5e-4, tmax=0.5, root=2, minalt=wl.MINALT,
                            maxsteps=args.maxsteps or W["maxsteps"], outputper=W["outputper"], del_=W["del_"],
                            refill_threshold=args.refill, ray_order=args.ray_order)
        model, _, _ = make_model(ctx, kind, (args.grid or 256) if kind == "interp" else 0, (args.points or 825_000) if kind == "scattered" else 0)
        pos0, dir0, w0 = wl.launch_set(total, W["seed"])
        lo, hi = parallel.shard_bounds(total, rank, world)
        batch = DeviceBatch(model, p, pos0[lo:hi], dir0[lo:hi], w0[lo:hi], dev, nbuf=2)
        launch = lambda k: batch.trace(k % 2)
        pack = lambda rows, nrows, slot: parallel.pack_rows_device(rows, nrows, p.outputper, slot=slot)
        parallel.trace_sharded_pipelined(dist, total, 2, launch, pack)  # warm-up (allocates the pack buffers)
        torch.cuda.synchronize(dev)
        dist.barrier()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        parallel.trace_sharded_pipelined(dist, total, steps, launch, pack)
        torch.cuda.synchronize(dev)
        dist.barrier()
        torch.cuda.synchronize(dev)
        el = time.perf_counter() - t0
        rdev = dev if dist.get_backend() == "nccl" else torch.device("cpu")
        t = torch.tensor([el], dtype=torch.float64, device=rdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        del batch
        parallel.release_pack_buffers(dev)
        return {"steps": steps, "ms_per_step": 1e3 * float(t.cpu()[0]) / steps,
                "note": "parallel.trace_sharded_pipelined: gather of step k overlapped with the trace of step k + 1; "
                        "gather_ms_exposed = this ms_per_step - the trace kernel's time (max over ranks)"}
    except Exception as e:  # pragma: no cover
        return {"ms_per_step": None, "error": "%s: %s" % (type(e).__name__, e)}


def describe(res):
    """config.workload text."""
    p, kind = res["params"], res["kind"]
    what = {"interp": "interp_dens_model on %d^3 x4 lnN grid (tricubic)" % res["grid"],
            "ngo": "ngo_dens_model", "scattered": "scattered_interp_dens_model (%d samples, order 2, window 1.5/5)" % res["points"]}[kind]
    fld = res.get("field", (0, 0))
    bname = {(0, 0): "dipole B", (1, 0): "IGRF B (use_igrf=1)", (0, 1): "dipole + T04_s B (use_tsyganenko=1)",
             (1, 1): "IGRF + T04_s B"}[tuple(fld)]
    return "%s: %d rays (%s scaling), %s, %s, adaptive RK45, maxsteps %d, outputper %d" % (
        res["workload"], res["total_rays"], res["scaling"], what, bname, p.maxsteps, p.outputper)


def detail_of(res):
    steps = max(res["steps"], 1)
    return {"accepted_steps_per_launch_all_ranks": res["accepted"] / steps,
            "attempts_per_launch": res["attempts"] / steps,
            "reject_ratio": 1.0 - res["accepted"] / max(res["attempts"], 1),
            "lane_occupancy": res["attempts"] / max(64 * res["wave_attempts"], 1),
            "mean_rows_per_ray_rank0": float(res["nrows"].mean()) if len(res["nrows"]) else 0.0,
            "stopcond_hist_rank0": {str(int(k)): int(v) for k, v in zip(*np.unique(res["stop"], return_counts=True))},
            "model_setup_s": res["setup_s"], "model_device_GB": res["model"].device_bytes / 1e9}


# --------------------------------------------------------------------------------------------- main
def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE is %d: for N > 1 launch with `python -m torch.distributed.run "
                         "--nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 bench.py --gpus N ...` (one rank per GPU)\n"
                         % (args.gpus, world))
        sys.exit(2)
    default_invocation = (args.workload == "interp256" and not (args.rays or args.grid or args.maxsteps or args.use_igrf
                                                                or args.use_tsyganenko))
    want_other = args.other_configs == 1 or (args.other_configs == -1 and default_invocation and not args.pmc_child)
    khash = kernel_source_hash()

    # ---- 0. counter passes: child processes under rocprofv3, BEFORE this process touches the GPU -------------------
    traffic, traffic_note, other_pmc = None, None, {}
    mode = args.traffic
    if args.pmc_child or world > 1:
        mode = "off" if args.pmc_child else ("file" if mode in ("auto", "live") else mode)
    elif mode == "auto":
        mode = "live" if shutil.which("rocprofv3") else "file"
    if mode == "live":
        traffic, traffic_note = live_traffic(args, args.workload)
        if want_other:  # config[1]: fp64 VALU mix of one launch; config[4]: fabric bytes of a 100k-ray launch
            v, err = pmc_pass(["SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_TRANS_F64",
                               "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES"],
                              child_args_for(args, "ngo100k"))
            other_pmc["ngo100k"] = v if v is not None else {"error": err}
            t, err = live_traffic(args, "scattered825k")  # the full 1M-ray launch: measured, not extrapolated
            other_pmc["scattered825k"] = t if t is not None else {"error": err}
            for name in ("interp_igrf200k", "interp_t04_64k"):  # field tails: fp32 + fp64 VALU mix of one launch
                progress("counter pass (VALU instruction mix) over one launch of %s" % name)
                v, err = pmc_pass(VALU_MIX_COUNTERS, child_args_for(args, name))
                other_pmc[name] = v if v is not None else {"error": err}

    import torch

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    # one rank per GPU.  (Rehearsal on a 1-GPU box: SRT_BENCH_BACKEND=gloo folds the ranks onto the devices that exist;
    # RCCL itself refuses two ranks on one device.)
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
    ctx = Ctx(args, torch, dev, dist, rank, world)

    progress("%s: %d warm-up + %d timed steps on %d GPU(s)" % (args.workload, args.warmup, args.steps, world))
    res = run_workload(ctx, args.workload, args.steps, args.warmup, rays_override=args.rays, nstream=args.streams, keep=True)
    if args.pmc_child:  # the parent (pmc_pass) reads this line from the child's stdout
        print(json.dumps({"pmc_child": {"accepted": res["accepted"], "kernel_ms": res["kernel_ms_rank0"], "rays": res["rays_this_rank"]}}))
        return
    if mode == "file":
        traffic, traffic_note = file_traffic(args.workload, res["rays_this_rank"], res["grid"], khash)
    if traffic is None and traffic_note and rank == 0:
        sys.stderr.write("bench.py: roofline.traffic unavailable -- %s\n" % traffic_note)

    # ---- N > 1: the same steps with the pack + gather of step k behind the trace of step k + 1 (outside the timed region) --
    pipe = None
    if world > 1 and res["gather"]:
        pipe = pipelined_pass(ctx, args.workload, rays_override=args.rays)
        if pipe.get("ms_per_step") is not None:
            pipe["gather_ms_exposed"] = pipe["ms_per_step"] - res["kernel_ms"]
            pipe["serial_ms_per_step"] = res["ms_per_step"]

    # ---- the other BASELINE configs, one short timed run each (outside the headline's timed region) ----------------
    other = {}
    if want_other:
        if world == 1:
            for name, st, wu in (("ngo100k", 3, 1), ("scattered825k", 1, 0), ("interp4m", 1, 1), ("interp_igrf200k", 2, 1),
                                 ("interp_t04_64k", 2, 1)):
                progress("other config %s" % name)
                other[name] = other_config_line(ctx, run_workload(ctx, name, st, wu), other_pmc.get(name))
        else:
            other["interp4m"] = other_config_line(ctx, run_workload(ctx, "interp4m", 2, 1), None)

    if want_other and world == 1 and os.environ.get("SRT_BENCH_CLI", "1") != "0":
        progress("other config cli_end_to_end (the drop-in executable, files in, .ray text out)")
        other["cli_end_to_end"] = cli_end_to_end(ctx, res)

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
        p, kind = res["params"], res["kind"]
        k_ms = res["kernel_ms_rank0"]
        steps_rank0 = float(res["nrows"].astype(np.int64).clip(1).sum() - len(res["nrows"]))  # accepted steps of THIS rank's launch
        out = {
            "metric": "ray-steps/sec (whole node) + achieved HBM GB/s vs roofline",
            "value": res["value"],
            "unit": "accepted ray-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": res["ms_per_step"],
            "higher_is_better": True,
            "scaling": res["scaling"],
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": describe(res), "total_rays": res["total_rays"], "rays_per_gpu": res["rays_this_rank"],
                       "grid": res["grid"], "maxsteps": p.maxsteps, "outputper": p.outputper, "integrator": "rkf45 adaptive",
                       "first_attempt_policy": int(p.first_attempt_policy),
                       "parallelism": "one launch set, contiguous shards x%d (parallel.trace_sharded)" % world if world > 1 else "1 GPU",
                       "ray_order": "launch-cell Morton order, sorted on the device inside the timed region" if (args.ray_order and kind == "interp") else "as given",
                       "gather": res["gather"], "streams": res["nstream"]},
            "roofline": roofline_of(res, k_ms, steps_rank0, traffic, traffic_note, stream_gbs, khash),
            "detail": detail_of(res),
        }
        if "rows_checksum" in res:
            out["detail"]["rows_checksum"] = res["rows_checksum"]
            out["detail"]["packed_rows"] = res["gathered_rows"]
        if world == 1 and "pack_ms_one_gpu" in res:
            out["multi_gpu_parts"] = {"pack_ms": res["pack_ms_one_gpu"], "packed_bytes": res["packed_bytes_one_gpu"],
                                      "note": "the fixed per-rank cost of a multi-GPU step measured here on one GPU: packing the kept rows "
                                              "of this launch (srt_pack_rows_device, after the timed region); a peer then sends "
                                              "packed_bytes over one xGMI link"}
        if world > 1:
            out["multi_gpu"] = {"gather_ms": res["gather_ms"], "pack_ms": res["pack_ms"], "kernel_ms_max_over_ranks": res["kernel_ms"],
                                "pipelined": pipe,
                                "gather_bytes_into_rank0": res["gather_bytes"],
                                "gather_GBs": (res["gather_bytes"] / (res["gather_ms"] * 1e-3) / 1e9) if res["gather_ms"] else None,
                                "note": "per step, max over ranks: trace (kernel_ms) + pack + variable-length gather of the kept rows "
                                        "(all_gather of counts, grouped send/recv of the packed rows, nrows and stop codes to rank 0)"}
        if other:
            out["other_configs"] = other
        if args.damping_rays > 0 and world == 1:
            b = res["batch"]
            out["detail"]["damping"] = damping_leg(args, api, res["model"], p, b.slots, b.out[0]["rows"], b.out[0]["nrows"], b.d_w, dev, torch, res["nrows"])
        if args.cpu_seconds > 0 and world == 1:
            progress("CPU baselines (oracle on the host cores, then the reference harness)")
            pos0, dir0, w0 = res["launch"]
            out["cpu_baseline"] = cpu_baseline(args, kind, p, wl, pos0, dir0, w0, res["grid"], res["extra"])
            # the real reference (Fortran, one core -- its only mode), when its prebuilt harness travelled with the repo
            out["cpu_reference"] = cpu_reference(args, kind, p, wl, pos0, dir0, w0)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
        if traffic is not None and traffic["source"].startswith("live"):
            save_traffic(args.workload, res, traffic, khash, k_ms)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def cli_end_to_end(ctx, res):
    """The product's own user-visible path for the headline config: `bin/raytracer` (the reference driver's flags) from a TEXT
    ray file + a binary grid file to a `.ray` TEXT file in the reference's record format (raytracer_driver.f95:1146,
    :1197-1217), all files in /dev/shm, phases from the executable's own --timing=1 line.  Run once per value of
    --first_attempt_policy: 1 is the executable's default (the reference's gfortran behaviour), 0 is what the API, the goldens
    and this bench's headline use (raytracer.f95:778-788)."""
    import subprocess

    from stanford_raytracer_amd import api, workloads as wl

    if res["kind"] != "interp":
        return {"skipped": "headline workload is not the interp model"}
    exe = os.path.join(ROOT, "stanford_raytracer_amd", "bin", "raytracer")
    if not os.path.exists(exe):
        return {"skipped": "stanford_raytracer_amd/bin/raytracer is not built"}
    p = res["params"]
    n = res["total_rays"]
    slots = (p.maxsteps - 1) // p.outputper + 2
    need = n * slots * 830 + res["grid"] ** 3 * 32 + n * 200          # worst-case .ray text + grid + ray file
    base = "/dev/shm" if os.path.isdir("/dev/shm") else ctx.tmp
    try:
        free = shutil.disk_usage(base).free
    except OSError:
        free = 0
    if free < need * 1.1:
        return {"skipped": "%s has %.1f GB free, the files need up to %.1f GB" % (base, free / 1e9, need / 1e9)}
    td = tempfile.mkdtemp(prefix="srt_cli_", dir=base)
    out = {"files_in": base}
    try:
        t0 = time.time()
        F, b = wl.make_grid(res["grid"], half_width=10.0 * wl.R_E)
        gf = os.path.join(td, "grid.bin")
        api.write_grid_file(gf, F, b, wl.QS, wl.MS, binary=True)
        del F
        pos0, dir0, w0 = wl.launch_set(n, WORKLOADS[res["workload"]]["seed"])
        rf = os.path.join(td, "rays.txt")
        np.savetxt(rf, np.concatenate([pos0, dir0, w0[:, None]], axis=1), fmt="%.17g")
        out["inputs"] = {"grid_file_GB": os.path.getsize(gf) / 1e9, "grid_format": "SRTGRID1 (binary side-format; text grids are accepted too)",
                         "rays_file_GB": os.path.getsize(rf) / 1e9, "write_s": time.time() - t0}
        ofile = os.path.join(td, "out.ray")
        cmd = [exe, "--outputper=%d" % p.outputper, "--dt0=%r" % p.dt0, "--dtmax=%r" % p.dtmax, "--tmax=%r" % p.tmax, "--root=%d" % p.root,
               "--fixedstep=0", "--maxerr=%r" % p.maxerr, "--maxsteps=%d" % p.maxsteps, "--minalt=%r" % p.minalt,
               "--inputraysfile=%s" % rf, "--outputfile=%s" % ofile, "--modelnum=3", "--interp_interpfile=%s" % gf,
               "--yearday=2010001", "--milliseconds_day=0", "--use_tsyganenko=0", "--use_igrf=0", "--ray_order=%d" % p.ray_order,
               "--timing=1"]
        for pol in (1, 0):
            t0 = time.time()
            r = subprocess.run(cmd + ["--first_attempt_policy=%d" % pol], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
            wall = time.time() - t0
            key = "first_attempt_policy_%d" % pol
            if r.returncode != 0:
                out[key] = {"error": "rc %d: %s" % (r.returncode, r.stderr.strip()[-300:])}
                continue
            tline = [ln for ln in r.stdout.splitlines() if ln.startswith(" timing: ")]
            sline = [ln for ln in r.stdout.splitlines() if "accepted steps" in ln]
            ph = json.loads(tline[-1][len(" timing: "):]) if tline else {}
            acc = int(sline[-1].split()[2]) if sline else None
            gb = os.path.getsize(ofile) / 1e9
            out[key] = {"process_wall_s": wall, "phases_s": {k: ph.get(k) for k in ("parse_s", "model_s", "trace_s", "write_s", "wall_s")},
                        "accepted_steps": acc, "ray_file_GB": gb, "records": int(os.path.getsize(ofile) // 823),
                        "writer_GBs": (gb / ph["write_s"]) if ph.get("write_s") else None,
                        "rays_per_s_end_to_end": n / wall, "accepted_steps_per_s_end_to_end": (acc / wall) if acc else None}
            os.remove(ofile)
        out["note"] = ("%d rays through bin/raytracer, one device; trace_s holds the host<->device copies of srt_trace_batch (PCIe-inclusive) "
                       "and the chunks' kernels, write_s the text formatting on all host cores, overlapped with the next chunk's trace; "
                       "policy 1 = the executable's default, policy 0 = the headline's" % n)
    except Exception as e:  # pragma: no cover
        out["error"] = "%s: %s" % (type(e).__name__, e)
    finally:
        shutil.rmtree(td, ignore_errors=True)
    return out


def roofline_of(res, k_ms, steps_per_launch, traffic, traffic_note, stream_gbs, khash):
    kind, p = res["kind"], res["params"]
    r = {"kernel": "trace_kernel<%s,adaptive>" % kind, "kernel_ms": k_ms, "accepted_steps_per_launch": steps_per_launch,
         "kernel_source_sha16": khash}
    if kind == "ngo":  # no table: compute-bound; the fp64 mix comes from the PMC pass (other_configs carries it in the default run)
        r.update({"bound": "fp64_valu", "achieved": None, "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": None, "traffic": None})
        return r
    tb = traffic["bytes_per_launch"] if traffic else None
    if traffic and traffic.get("accepted_steps_of_counted_launch") not in (None, steps_per_launch):
        r["traffic_warning"] = "the counted launch had %s accepted steps, this run's launches %s" % (
            traffic["accepted_steps_of_counted_launch"], steps_per_launch)
    achieved = (tb / (k_ms * 1e-3) / 1e9) if (tb and k_ms > 0) else None
    r.update({
        "bound": "hbm",
        # achieved = MEASURED bytes through the L2's fabric side per launch / kernel time (NOT the algorithmic figure)
        "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": (achieved / HBM_PEAK_GBS) if achieved else None,
        "traffic": tb, "traffic_source": traffic["source"] if traffic else ("unavailable: %s" % traffic_note),
        "traffic_bytes_per_accepted_step": (tb / steps_per_launch) if (tb and steps_per_launch) else None,
        "traffic_side": "L2 fabric side (TCC->EA requests): Infinity-Cache hits are included; rocprofv3 on gfx950 lists no "
                        "DRAM-side (memory-controller) counter, so DRAM bytes cannot be separated",
        "stream_copy_GBs": stream_gbs,
        "frac_of_stream_copy": (achieved / stream_gbs) if (achieved and stream_gbs) else None,
        "lds_gather_ceiling_GBs": list(LDS_GATHER_CEILING_GBS),
        "frac_of_lds_gather_ceiling": (achieved / LDS_GATHER_CEILING_GBS[1]) if achieved else None,
    })
    if kind == "interp":
        ab = ALGO_INTERP(p.outputper)
        r.update({"algorithmic_bytes_per_accepted_step": ab,
                  "algorithmic_GBs": (ab * steps_per_launch / (k_ms * 1e-3) / 1e9) if k_ms > 0 else None,
                  "algorithmic_note": "SURVEY 8d figure (44 lookups x 2 KiB + state + row): counts every stencil point as its own "
                                      "2-KiB read; the kernel stages 6 blocks per attempt, so this rate exceeds the peak and is not a fraction"})
    return r


def other_config_line(ctx, res, pmc):
    """One BASELINE config timed beside the headline: value, step time, kernel time and the roofline that bounds it."""
    k_ms = res["kernel_ms_rank0"]
    steps_rank0 = float(res["nrows"].astype(np.int64).clip(1).sum() - len(res["nrows"]))
    line = {"workload": describe(res), "value": res["value"], "unit": "accepted ray-steps/s", "steps": res["steps"],
            "warmup": res["warmup"], "ms_per_step": res["ms_per_step"], "kernel_ms": k_ms, "scaling": res["scaling"],
            "lane_occupancy": res["attempts"] / max(64 * res["wave_attempts"], 1),
            "accepted_steps_per_launch": res["accepted"] / max(res["steps"], 1)}
    if res["gather"]:
        line.update({"gather_ms": res["gather_ms"], "pack_ms": res["pack_ms"], "gather_bytes_into_rank0": res["gather_bytes"],
                     "kernel_ms_max_over_ranks": res["kernel_ms"]})
    kind = res["kind"]
    if kind == "ngo":
        rf = {"bound": "fp64_valu", "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "achieved": None, "frac": None}
        if pmc and "error" not in pmc:
            wave_flop_inst = 2.0 * pmc["SQ_INSTS_VALU_FMA_F64"] + pmc["SQ_INSTS_VALU_MUL_F64"] + pmc["SQ_INSTS_VALU_ADD_F64"]
            # instructions are counted per wave; lanes switched off by EXEC still occupy the slot.  The live-lane share of
            # VALU time is SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU).
            share = pmc["SQ_THREAD_CYCLES_VALU"] / max(64.0 * pmc["SQ_ACTIVE_INST_VALU"], 1.0)
            share = min(max(share, 0.0), 1.0)
            ach = wave_flop_inst * 64.0 * share / (k_ms * 1e-3) / 1e12 if k_ms > 0 else None
            rf.update({"achieved": ach, "frac": ach / FP64_VALU_PEAK_TFLOPS if ach else None,
                       "issue_slot_TFLOPs": wave_flop_inst * 64.0 / (k_ms * 1e-3) / 1e12 if k_ms > 0 else None,
                       "live_lane_share_of_valu": share,
                       "valu_busy": pmc["SQ_ACTIVE_INST_VALU"] / max(pmc["SQ_WAVE_CYCLES"], 1.0),
                       "fp64_wave_instructions": {k: pmc[k] for k in pmc if k.startswith("SQ_INSTS_VALU_")},
                       "source": "live: rocprofv3 --pmc child pass of one launch of this workload"})
        elif pmc:
            rf["source"] = "unavailable: %s" % pmc["error"]
        line["roofline"] = rf
        if "latency_floor_ms" in res:
            line["latency_floor_ms"] = res["latency_floor_ms"]
            line["latency_floor_note"] = res["latency_floor_note"]
    elif kind == "interp" and tuple(res.get("field", (0, 0))) != (0, 0):
        rf = {"bound": "valu", "unit": "TFLOP/s", "peak": {"fp64": FP64_VALU_PEAK_TFLOPS, "fp32": FP32_VALU_PEAK_TFLOPS},
              "achieved": None, "frac": None}
        if pmc and "error" not in pmc:
            f64 = 2.0 * pmc["SQ_INSTS_VALU_FMA_F64"] + pmc["SQ_INSTS_VALU_MUL_F64"] + pmc["SQ_INSTS_VALU_ADD_F64"]
            f32 = 2.0 * pmc["SQ_INSTS_VALU_FMA_F32"] + pmc["SQ_INSTS_VALU_MUL_F32"] + pmc["SQ_INSTS_VALU_ADD_F32"]
            # the counter pass traced one launch of this same workload; instructions per launch carry over, time is this run's
            t = k_ms * 1e-3
            a64, a32 = (f64 * 64.0 / t / 1e12, f32 * 64.0 / t / 1e12) if t > 0 else (None, None)
            rf.update({"achieved": {"fp64": a64, "fp32": a32},
                       # share of the vector ALU's issue capacity the arithmetic instructions take (wave-instruction slots;
                       # lanes switched off by EXEC still occupy theirs)
                       "frac": (a64 / FP64_VALU_PEAK_TFLOPS + a32 / FP32_VALU_PEAK_TFLOPS) if t > 0 else None,
                       "valu_busy": pmc["SQ_ACTIVE_INST_VALU"] / max(pmc["SQ_WAVE_CYCLES"], 1.0),
                       "wave_instructions": {k: pmc[k] for k in pmc if k.startswith("SQ_INSTS_VALU_")},
                       "source": "live: rocprofv3 --pmc child pass of one launch of this workload"})
            if tuple(res.get("field", (0, 0)))[0]:
                # the IGRF synthesis runs on v_pk_mul_f32 / v_pk_add_f32 (two operations per lane and instruction); the
                # counters count such an instruction once, so fp32 "achieved" at 64 operations per instruction is a floor
                rf["fp32_note"] = ("the IGRF synthesis issues packed fp32 instructions (2 operations per lane); SQ_INSTS_VALU_*_F32 "
                                   "counts each once: achieved.fp32 is a lower bound, up to 2x")
        elif pmc:
            rf["source"] = "unavailable: %s" % pmc["error"]
        line["roofline"] = rf
    elif kind == "scattered":
        rf = {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "achieved": None, "frac": None}
        if pmc and "error" not in pmc and pmc.get("accepted_steps_of_counted_launch"):
            # the counter passes traced this same launch (same rays, same sample set): measured bytes of the whole launch
            bps = pmc["bytes_per_launch"] / pmc["accepted_steps_of_counted_launch"]
            same = pmc["accepted_steps_of_counted_launch"] == steps_rank0
            tb = pmc["bytes_per_launch"] if same else bps * steps_rank0
            ach = tb / (k_ms * 1e-3) / 1e9 if k_ms > 0 else None
            rf.update({"achieved": ach, "frac": ach / HBM_PEAK_GBS if ach else None, "traffic_bytes_per_accepted_step": bps,
                       "traffic": tb, "FETCH_SIZE_KB": pmc.get("FETCH_SIZE_KB"), "WRITE_SIZE_KB": pmc.get("WRITE_SIZE_KB"),
                       "source": pmc["source"] + (" over this launch itself (%d rays)" % res["total_rays"] if same else
                                                  " on a launch with %s accepted steps, scaled to this one's" % pmc["accepted_steps_of_counted_launch"])})
        elif pmc:
            rf["source"] = "unavailable: %s" % pmc.get("error", "no step count from the counter pass")
        line["roofline"] = rf
    else:
        line["roofline"] = {"bound": "hbm", "note": "same kernel as the headline: see roofline"}
    if "pack_ms_one_gpu" in res:
        line["pack_ms"] = res["pack_ms_one_gpu"]
        line["packed_bytes"] = res["packed_bytes_one_gpu"]
    return line


def save_traffic(workload, res, traffic, khash, k_ms):
    """Keep what the live passes measured (gpurun_out/ is merged back): copy to profiles/ to commit it."""
    try:
        d = os.path.join(ROOT, "gpurun_out")
        os.makedirs(d, exist_ok=True)
        steps = float(res["nrows"].astype(np.int64).clip(1).sum() - len(res["nrows"]))
        json.dump({"workload": workload, "rays": res["rays_this_rank"], "grid": res["grid"], "kernel_source_sha16": khash,
                   "kernel": "srt::trace_kernel<%s>" % res["kind"], "FETCH_SIZE_KB": traffic["FETCH_SIZE_KB"],
                   "WRITE_SIZE_KB": traffic["WRITE_SIZE_KB"],
                   "correction": "gfx950: FETCH_SIZE doubled (MI355X_MICROARCH.md, HBM section: 128-B requests tallied at 64 B); "
                                 "WRITE_SIZE as read; x1024 B/KiB.  Counted at the L2's fabric side: Infinity-Cache hits included.",
                   "hbm_bytes_per_launch": traffic["bytes_per_launch"], "accepted_steps_per_launch": steps,
                   "hbm_bytes_per_accepted_step": traffic["bytes_per_launch"] / max(steps, 1.0), "kernel_ms_at_collection": k_ms,
                   "source": "bench.py live PMC passes"}, open(os.path.join(d, "traffic_%s.json" % workload), "w"), indent=1)
    except Exception:
        pass


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


def cpu_baseline(args, kind, p, wl, pos0, dir0, w0, grid_n, extra):
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
        from stanford_raytracer_amd import api
        txt = extra["points_file"] + ".txt"
        pts, lnN = wl.make_points_config5(5) if args.points in (0, 825_000) else wl.make_points(
            int(args.points * 0.97), args.points - int(args.points * 0.97), 5, half_width=10.0 * wl.R_E)
        api.write_points_file(txt, np.concatenate([pts, lnN], axis=1), np.array([-10.0 * wl.R_E, 10.0 * wl.R_E] * 3), wl.QS, wl.MS)
        om = oracle.Model.scattered_file(txt, window_scale=1.5, order=2, exact=0, local_window_scale=5.0)
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
