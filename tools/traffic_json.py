#!/usr/bin/env python3
"""tools/traffic_json.py <measure dir> <workload> -- turn the FETCH_SIZE / WRITE_SIZE passes of tools/measure_round.sh
into profiles/traffic_<workload>.json (what bench.py reports as roofline.traffic), with the gfx950 corrections of
MI355X_MICROARCH.md (HBM section): FETCH_SIZE is in KiB and counts 128-B requests at 64 B -> x2; WRITE_SIZE as read."""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def counter(d, name):
    for f in glob.glob(os.path.join(d, "pmc_" + name, "*", "*counter_collection.csv")):
        for row in csv.DictReader(open(f)):
            if "trace_kernel" in row["Kernel_Name"] and row["Counter_Name"] == name:
                return float(row["Counter_Value"]), row["Kernel_Name"]
    raise SystemExit("no %s pass under %s" % (name, d))


def main():
    d, workload = sys.argv[1], sys.argv[2]
    bench = json.loads(open(os.path.join(d, "bench.log")).read().strip().splitlines()[-1])
    fetch, kname = counter(d, "FETCH_SIZE")
    write, _ = counter(d, "WRITE_SIZE")
    hbm = fetch * 1024.0 * 2.0 + write * 1024.0
    steps = bench["roofline"]["accepted_steps_per_launch"]
    out = {"workload": workload, "rays": bench["config"]["rays_per_gpu"], "grid": bench["config"]["grid"],
           "kernel": kname.split("(")[0].replace("void ", ""), "FETCH_SIZE_KB": fetch, "WRITE_SIZE_KB": write,
           "correction": "gfx950: FETCH_SIZE doubled (MI355X_MICROARCH.md, HBM section: 128-B requests tallied at 64 B); "
                         "WRITE_SIZE as read; x1024 B/KiB.  Counted at the L2's fabric side: Infinity-Cache hits included.",
           "hbm_bytes_per_launch": hbm, "accepted_steps_per_launch": steps, "hbm_bytes_per_accepted_step": hbm / steps,
           "kernel_ms_at_collection": bench["roofline"]["kernel_ms"],
           "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) via tools/measure_round.sh -> %s" % os.path.relpath(d, ROOT)}
    path = os.path.join(ROOT, "profiles", "traffic_%s.json" % workload)
    json.dump(out, open(path, "w"), indent=1)
    print(path, "%.3e B/launch, %.0f B/accepted step" % (hbm, hbm / steps))


if __name__ == "__main__":
    main()
