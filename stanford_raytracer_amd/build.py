"""Build the HIP library (libsrt_hip.so) and the raytracer-compatible CLI for gfx950, in-tree.

hipcc cross-compiles without a GPU.  Outputs (git-ignored, shipped to the GPU box by gpurun):
    stanford_raytracer_amd/lib/libsrt_hip.so
    stanford_raytracer_amd/bin/raytracer
"""
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIBDIR = os.path.join(PKG, "lib")
BINDIR = os.path.join(PKG, "bin")
LIB = os.path.join(LIBDIR, "libsrt_hip.so")
CLI = os.path.join(BINDIR, "raytracer")
ARCH = "gfx950"

LIB_SRCS = ["srt_api.hip", "srt_host.cpp", "srt_scattered_host.cpp"]
CLI_SRCS = ["srt_cli.cpp"]
def _headers():
    """Every header the library is compiled from: all of csrc/*.hpp, csrc/*.h and include/*.h (globbed, so a new header
    cannot be forgotten: a stale .so under a fresh source hash is what this prevents)."""
    import glob
    hs = sorted(glob.glob(os.path.join(CSRC, "*.hpp")) + glob.glob(os.path.join(CSRC, "*.h")))
    hs += sorted(glob.glob(os.path.join(PKG, "..", "include", "*.h")))
    return [os.path.abspath(h) for h in hs]


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the MI355X path cannot be built (there is no CPU fallback)")
    return exe


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force=False, verbose=False):
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(BINDIR, exist_ok=True)
    deps = [os.path.join(CSRC, f) for f in LIB_SRCS] + _headers() + [os.path.abspath(__file__)]
    if force or _stale(LIB, deps):
        cmd = [_hipcc(), "-O3", "--offload-arch=" + ARCH, "-std=c++17", "-fPIC", "-shared", "-o", LIB] + \
              [os.path.join(CSRC, f) for f in LIB_SRCS]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    cli_src = [os.path.join(CSRC, f) for f in CLI_SRCS]
    if all(os.path.exists(s) for s in cli_src) and (force or _stale(CLI, cli_src + [LIB])):
        cmd = [_hipcc(), "-O2", "-std=c++17", "-pthread", "-o", CLI] + cli_src + \
              ["-L" + LIBDIR, "-lsrt_hip", "-Wl,-rpath,$ORIGIN/../lib"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    build_fortran(force, verbose)
    return LIB


FDRV = os.path.join(BINDIR, "srt_fortran_driver")


def build_fortran(force=False, verbose=False):
    """The Fortran-callable shim + its example driver (needs flang; skipped quietly when absent)."""
    fc = shutil.which("flang") or shutil.which("amdflang")
    fdir = os.path.join(PKG, "fortran")
    srcs = [os.path.join(fdir, "srt_bindc.f90"), os.path.join(fdir, "srt_fortran_driver.f90")]
    if fc is None or not all(os.path.exists(s) for s in srcs):
        return None
    if force or _stale(FDRV, srcs + [LIB]):
        moddir = os.path.join(PKG, "lib", "fmod")
        os.makedirs(moddir, exist_ok=True)
        cmd = [fc, "-O2", "-module-dir", moddir, "-o", FDRV] + srcs + ["-L" + LIBDIR, "-lsrt_hip",
                                                                    "-Wl,-rpath,$ORIGIN/../lib"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return FDRV


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
