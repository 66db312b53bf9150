#!/usr/bin/env python3
"""The exact TEXT LAYOUT of the reference's two producers, as data: one tiny file from each.

    builder_grid_3x4x5_compder1.txt   gcpm_dens_model_buildgrid --nx=3 --ny=4 --nz=5 --compder=1  (header (5i10), bounds
                                      (6es24.15e3), qs, ms, then f and the seven derivative blocks one value per line,
                                      gcpm_dens_model_buildgrid.f95:302-327)
    builder_samples_60.txt            gcpm_dens_model_buildgrid_random, 60 samples (nspec on its own line, bounds, qs, ms, then
                                      "x y z lnN_1..4" per line, gcpm_dens_model_buildgrid_random.f95:196-225, ..helpermod.f95:39-43)

tests/test_host_formats.py holds srt_grid_file_write / srt_points_file_write (what `raytracer --buildgrid=1` and
`--buildsamples=1` write) to these bytes.  Binaries: oracle/build_ref.py; run in the build container only.

    python tests/golden/make_builder_layout_golden.py
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
from make_gcpm_golden import REFBIN, date_flags, scratch_dir  # noqa: E402


def main():
    work = os.path.join(HERE, "_work")
    os.makedirs(work, exist_ok=True)
    cwd = scratch_dir(work)
    b = ["--minx=-2.1e7", "--maxx=2.3e7", "--miny=-1.9e7", "--maxy=2.0e7", "--minz=-2.2e7", "--maxz=1.8e7"]
    g = os.path.join(HERE, "builder_grid_3x4x5_compder1.txt")
    subprocess.run([os.path.join(REFBIN, "gcpm_dens_model_buildgrid")] + b + ["--nx=3", "--ny=4", "--nz=5", "--compder=1",
                   "--filename=%s" % g] + date_flags(), check=True, cwd=cwd, stdout=subprocess.DEVNULL)
    s = os.path.join(HERE, "builder_samples_60.txt")
    subprocess.run([os.path.join(REFBIN, "gcpm_dens_model_buildgrid_random")] + b + ["--n_zero_altitude=10", "--n_iri_pad=20",
                   "--n_initial_radial=10", "--n_initial_uniform=20", "--initial_tol=1.0", "--max_recursion=3", "--adaptive_nmax=0",
                   "--filename=%s" % s] + date_flags(), check=True, cwd=cwd, stdout=subprocess.DEVNULL)
    for f in (g, s):
        os.chmod(f, 0o644)
        print(f, os.path.getsize(f), "bytes,", sum(1 for _ in open(f)), "lines")
    shutil.rmtree(work)


if __name__ == "__main__":
    main()
