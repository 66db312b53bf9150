"""Run the real reference through oracle/_ref/ref_harness -- TEST INFRASTRUCTURE ONLY.

The binary is built by oracle/build_ref.py from /root/reference (only possible in the build
container); it travels to the GPU box as a prebuilt file.  Everything here degrades to
`available() == False` when it is absent.
"""
import os
import re
import subprocess
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
EXE = os.path.join(HERE, "_ref", "ref_harness")

NOUT = {"params": 19, "disp": 10, "grad": 14, "step": 21, "rh": 1, "t04": 3, "ext": 33}


def available():
    return os.path.isfile(EXE) and os.access(EXE, os.X_OK)


def _model_flags(model):
    kind = model["kind"]
    fl = ["--modelnum=%d" % kind, "--yearday=%d" % model.get("yearday", 2010001),
          "--milliseconds_day=%d" % model.get("msec", 0)]
    if model.get("use_igrf"):
        fl.append("--use_igrf=1")
    if model.get("use_tsyganenko"):
        fl.append("--use_tsyganenko=1")
    if kind == 1:
        fl.append("--ngo_configfile=%s" % model["file"])
    else:
        fl.append("--interp_interpfile=%s" % model["file"])
    if kind == 4:
        fl += ["--scattered_interp_window_scale=%r" % model.get("window_scale", 1.5),
               "--scattered_interp_order=%d" % model.get("order", 2),
               "--scattered_interp_exact=%d" % model.get("exact", 0),
               "--scattered_interp_local_window_scale=%r" % model.get("local_window_scale", 5.0)]
    return fl


def _write_rows(path, rows):
    with open(path, "w") as f:
        for r in np.atleast_2d(np.asarray(rows, dtype=np.float64)):
            f.write(" ".join("%.17g" % v for v in r) + "\n")


def run_mode(mode, rows, model=None):
    """rows: [N, ncol] inputs for the mode; returns [N, NOUT[mode]] float64."""
    with tempfile.TemporaryDirectory() as td:
        fin, fout = os.path.join(td, "in.txt"), os.path.join(td, "out.bin")
        _write_rows(fin, rows)
        cmd = [EXE, "--mode=%s" % mode, "--in=%s" % fin, "--out=%s" % fout]
        if model is not None:
            cmd += _model_flags(model)
        subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL)
        return np.fromfile(fout, dtype=np.float64).reshape(-1, NOUT[mode])


def run_rays(model, rays, rayout=None, threads=1, **kw):
    """rays: [N,7] (pos0, dir0, w).  Returns (list of per-ray dicts, timing dict).

    Each dict: raynum, stopcond, rows[T, 32] = t,pos3,vprel3,vgrel3,n3,B03,qs4,ms4,Ns4,nus4.
    """
    with tempfile.TemporaryDirectory() as td:
        fin, fout = os.path.join(td, "rays.txt"), os.path.join(td, "out.bin")
        _write_rows(fin, rays)
        cmd = [EXE, "--mode=run", "--in=%s" % fin, "--out=%s" % fout] + _model_flags(model)
        if rayout:
            cmd.append("--rayout=%s" % rayout)
        for k, v in kw.items():
            cmd.append("--%s=%r" % (k, v))
        res = subprocess.run(cmd, check=True, stdout=subprocess.PIPE, text=True)
        m = re.search(r"REF_TIMING accepted_steps=\s*(\d+) seconds=\s*([0-9.Ee+-]+)", res.stdout)
        timing = {"steps": int(m.group(1)), "seconds": float(m.group(2))} if m else {}
        raw = np.fromfile(fout, dtype=np.float64)
    out = []
    pos = 0
    while pos < raw.size:
        raynum, stop, T = int(raw[pos]), int(raw[pos + 1]), int(raw[pos + 2])
        pos += 3
        rows = raw[pos:pos + 32 * T].reshape(T, 32).copy()
        pos += 32 * T
        out.append({"raynum": raynum, "stopcond": stop, "rows": rows})
    return out, timing


def scattered_root(model):
    """The sample at the root of the reference's kd-tree for a modelnum-4 model: (point[3], vals[nspec+1], maxnearest)."""
    with tempfile.TemporaryDirectory() as td:
        fin = os.path.join(td, "in.txt")
        _write_rows(fin, np.zeros((1, 3)))
        cmd = [EXE, "--mode=scatroot", "--in=%s" % fin, "--out=%s" % os.path.join(td, "out.bin")] + _model_flags(model)
        res = subprocess.run(cmd, check=True, stdout=subprocess.PIPE, text=True)
    for line in res.stdout.splitlines():
        if line.startswith("SCAT_ROOT"):
            v = np.array([float(t) for t in line.split()[1:]])
            return v[:3], v[3:-1], float(v[-1])
    raise RuntimeError("ref_harness gave no SCAT_ROOT line")
