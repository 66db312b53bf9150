"""CPU: the parts of bench.py's contract that need no GPU -- world-size check, provenance of the traffic figure, and the
roofline arithmetic (frac = measured bytes / kernel time / peak, never the algorithmic figure)."""
import json
import os
import subprocess
import sys
import types

import numpy as np

from conftest import ROOT

sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_gpus_flag_must_match_the_world_size():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8"], env=env, stderr=subprocess.PIPE, text=True)
    assert r.returncode == 2 and "WORLD_SIZE is 1" in r.stderr and "torch.distributed.run" in r.stderr


def test_committed_traffic_profile_belongs_to_these_kernel_sources():
    """profiles/traffic_interp256.json is only used (N > 1, or no rocprofv3) when it was collected for exactly the sources
    the library is built from; the committed one must be current."""
    khash = bench.kernel_source_hash()
    t, why = bench.file_traffic("interp256", 1_000_000, 256, khash)
    assert t is not None, why
    assert t["bytes_per_launch"] == t["FETCH_SIZE_KB"] * 1024 * 2 + t["WRITE_SIZE_KB"] * 1024   # gfx950 correction, guide HBM section
    stale, why = bench.file_traffic("interp256", 1_000_000, 256, "0" * 16)
    assert stale is None and why.startswith("STALE")
    other, why = bench.file_traffic("interp256", 123, 256, khash)
    assert other is None and "rays" in why


def test_roofline_fraction_comes_from_measured_bytes():
    p = types.SimpleNamespace(outputper=16, maxsteps=256)
    res = {"kind": "interp", "params": p}
    traffic = {"bytes_per_launch": 2.4e12, "source": "live: test", "accepted_steps_of_counted_launch": 2.0e8}
    r = bench.roofline_of(res, 350.0, 2.0e8, traffic, None, 4900.0, "abc")
    assert abs(r["achieved"] - 2.4e12 / 0.35 / 1e9) < 1e-6 and r["peak"] == 8000.0
    assert 0.0 < r["frac"] <= 1.0 and abs(r["frac"] - r["achieved"] / 8000.0) < 1e-12
    assert r["algorithmic_GBs"] > r["peak"]                      # kept beside it, not a fraction of anything
    assert "frac" not in [k for k in r if k.startswith("algorithmic")]
    assert r["traffic_bytes_per_accepted_step"] == 12000.0 and "traffic_warning" not in r
    r2 = bench.roofline_of(res, 350.0, 1.9e8, traffic, None, None, "abc")
    assert "traffic_warning" in r2                               # the counted launch was not this launch
    r3 = bench.roofline_of(res, 350.0, 2.0e8, None, "STALE: x", None, "abc")
    assert r3["achieved"] is None and r3["frac"] is None and r3["traffic_source"].startswith("unavailable")
    ngo = bench.roofline_of({"kind": "ngo", "params": p}, 50.0, 2.8e6, None, None, None, "abc")
    assert ngo["bound"] == "fp64_valu" and ngo["unit"] == "TFLOP/s"


def test_workload_table_matches_baseline_json():
    cfgs = json.load(open(os.path.join(ROOT, "BASELINE.json")))["configs"]
    assert "100k rays" in cfgs[1] and bench.WORKLOADS["ngo100k"]["rays"] == 100_000
    assert "1M rays" in cfgs[2] and bench.WORKLOADS["interp256"]["rays"] == 1_000_000
    assert "4M rays" in cfgs[3] and bench.WORKLOADS["interp4m"]["rays"] == 4_000_000 and bench.WORKLOADS["interp4m"]["scaling"] == "strong"
    assert "1M rays" in cfgs[4] and bench.WORKLOADS["scattered825k"]["rays"] == 1_000_000
    assert bench.parse_args([]).workload == "interp256" and bench.parse_args([]).streams == 1
